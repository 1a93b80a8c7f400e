"""Which kernels ran UNDER each inflate launch: reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints, per `bgzf_inflate_lanes`
dispatch, its start / duration, the queue it ran on, and the other queues' kernels whose execution overlapped it (name, overlap in ms) —
the evidence that span k+1's inflate runs beside span k's record search, field decode, coverage and signal kernels.
python tools/kernel_timeline.py <kernel_trace.csv> [max_rows]"""
import csv, sys
from collections import defaultdict
path = sys.argv[1]
max_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[0][0]
short = lambda n: n.replace("void ", "")[:44]
infl = [r for r in rows if "bgzf_inflate_lanes" in r[2]]
others = [r for r in rows if "bgzf_inflate_lanes" not in r[2]]
print("# %s: %d dispatches, %d of bgzf_inflate_lanes on queues %s" % (path.split("/")[-1], len(rows), len(infl), sorted({r[3] for r in infl})))
busy_infl = 0
span_lo, span_hi = infl[0][0], infl[-1][1]
ev = sorted([(s, 1) for s, e, _, _ in infl] + [(e, -1) for s, e, _, _ in infl])
depth, last, two = 0, None, 0
for t, d in ev:
    if depth > 0:
        busy_infl += t - last
    if depth > 1:
        two += t - last
    depth += d
    last = t
print("# from the first inflate's start to the last one's end: %.2f ms; an inflate kernel resident %.2f ms of it (%.1f %%), two of them %.2f ms"
      % ((span_hi - span_lo) * 1e-6, busy_infl * 1e-6, 100.0 * busy_infl / (span_hi - span_lo), two * 1e-6))
under = defaultdict(float)
total = defaultdict(float)
for s, e, n, q in others:
    total[n] += (e - s) * 1e-6
    for a, b, _, qi in infl:
        if qi != q and a < e and s < b:
            under[n] += (min(e, b) - max(s, a)) * 1e-6
print("# kernels of the other queues: total ms, ms of it under an inflate launch")
for n in sorted(total, key=lambda k: -total[k])[:16]:
    print("  %-46s %9.3f %9.3f  (%3.0f %%)" % (short(n), total[n], under[n], 100 * under[n] / total[n] if total[n] else 0))
print("# per inflate launch: start ms, duration ms, queue | kernels of other queues that overlapped it (ms)")
for a, b, _, qi in infl[:max_rows]:
    ov = defaultdict(float)
    for s, e, n, q in others:
        if q != qi and a < e and s < b:
            ov[short(n)] += (min(e, b) - max(s, a)) * 1e-6
    nxt = [(s2, e2) for s2, e2, _, q2 in infl if q2 != qi and s2 < b and a < e2 and (s2, e2) != (a, b)]
    both = sum(min(b, e2) - max(a, s2) for s2, e2 in nxt) * 1e-6
    print("  %9.3f %8.3f q%s | other inflate %.3f; %s" % ((a - t0) * 1e-6, (b - a) * 1e-6, qi, both,
                                                        ", ".join("%s %.3f" % kv for kv in sorted(ov.items(), key=lambda kv: -kv[1])[:6])))
