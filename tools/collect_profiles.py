#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_cov.sh run from gpurun_out/ (scratch) into profiles/:
  profiles/<tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary, this library's kernels only
  profiles/<tag>_pmc_cov_accumulate.txt  mean PMC counters per cov_accumulate launch (separate --pmc passes)
  profiles/traffic.json            HBM bytes per cov_accumulate launch = 2*FETCH_SIZE + WRITE_SIZE (KB -> B);
                                   the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md §HBM
usage: tools/collect_profiles.py <tag> [bench args used]"""
import csv, glob, json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)
ours = ("cov_", "gc_", "db", "scan_", "sd_", "segmented", "radix", "rs_", "bgzf_", "bam_", "sig_", "med_", "region_")
rows = list(csv.reader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if any(k in r[0] for k in ours) and "at::" not in r[0]:
            w.writerow(r)
out = subprocess.check_output([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), src, "cov_accumulate"]).decode()
with open(os.path.join(dst, tag + "_pmc_cov_accumulate.txt"), "w") as f:
    f.write("# rocprofv3 --pmc passes (one counter group per run, kernel-trace only), mean per cov_accumulate launch\n")
    f.write("# command: tools/profile_cov.sh %s %s\n" % (tag, " ".join(sys.argv[2:])))
    f.write(out)
vals = {}
for line in out.splitlines():
    p = line.split()
    if len(p) >= 3 and p[1] == "mean":
        vals[p[0]] = float(p[2])
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    nreads = float(os.environ.get("COV_READS_PER_LAUNCH", "600000000"))
    json.dump({"cov_accumulate_bytes_per_launch": traffic, "reads_per_launch": nreads, "cov_accumulate_bytes_per_read": traffic / nreads, "source": "profiles/%s_pmc_cov_accumulate.txt" % tag,
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) KB; x2 = gfx950 FETCH_SIZE correction for wide coalesced reads",
               "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"]},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read()[:3000])
