#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_all.sh run from gpurun_out/ (scratch) into profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, this library's kernels only
  profiles/<tag>_pmc.txt            mean PMC counters per launch of the hot-path kernels (separate --pmc passes)
  profiles/traffic.json             HBM bytes per launch of each of them = (2 x FETCH_SIZE + WRITE_SIZE) KB; the x2 is the gfx950
                                    FETCH_SIZE correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section); bench.py reads it
usage: tools/collect_profiles.py <tag> [bench args used]"""
import csv, glob, json, os, sys
from collections import defaultdict
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)
ours = ("cov_", "gc_", "db", "scan_", "sd_", "rs_", "bgzf_", "bam_", "sig_", "med_", "region_", "seg_means", "tile_scan")
rows = list(csv.reader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if any(k in r[0] for k in ours) and "at::" not in r[0]:
            w.writerow(r)
hot = ("cov_accumulate", "cov_finalize", "dbt_tile", "dbt_finish1", "gc_small_bins")
acc = defaultdict(lambda: defaultdict(list))
for fn in sorted(glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if any(h in k for h in hot):
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, tag + "_pmc.txt"), "w") as f:
    f.write("# rocprofv3 --pmc passes (one counter group per run, kernel-trace only), mean per launch\n")
    f.write("# command: tools/profile_all.sh %s   (bench.py %s)\n" % (tag, " ".join(sys.argv[2:])))
    for k, d in acc.items():
        f.write(k + "\n")
        for c, v in sorted(d.items()):
            f.write("   %-28s mean %18.1f  n=%d\n" % (c, sum(v) / len(v), len(v)))
traffic = {"source": "profiles/%s_pmc.txt" % tag, "formula": "(2*FETCH_SIZE + WRITE_SIZE) KB per launch; x2 = gfx950 FETCH_SIZE correction for wide coalesced reads",
           "kernels": {}}
for k, d in acc.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        traffic["kernels"][k] = {"FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "bytes_per_launch": (2.0 * fe + wr) * 1024.0}
# the kernel sources these numbers were measured on: bench.py reports a kernel's traffic only while its source file is unchanged
import hashlib
cs = os.path.join(REPO, "tiddit_amd", "csrc")
traffic["sources_sha256"] = {f: hashlib.sha256(open(os.path.join(cs, f), "rb").read()).hexdigest() for f in sorted(os.listdir(cs))
                             if f.endswith((".hip", ".h"))}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read()[:2500])
print(json.dumps(traffic, indent=1)[:2000])
