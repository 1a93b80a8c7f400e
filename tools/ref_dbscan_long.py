#!/usr/bin/env python3
"""Time the UNMODIFIED reference DBSCAN.py (imported from /root/reference, build container only) on
gen_points(n) and record labels sha256 + timings into tests/golden/dbscan_gen.json (data only).
usage: python tools/ref_dbscan_long.py N"""
import hashlib, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/tiddit")
import DBSCAN as D          # pure-python reference module
from tiddit_amd import synth
n = int(sys.argv[1])
pts = synth.gen_points(n)
t0 = time.time(); xl, xid = D.x_coordinate_clustering(pts, 500, 3); t1 = time.time()
lab, yid = D.y_coordinate_clustering(pts, 500, 3, xid, xl); t2 = time.time()
res = {"eps": 500, "m": 3, "x_clusters": int(xid) + 1, "final_max_id": int(lab.max()), "n_noise": int((lab == -1).sum()),
       "labels_sha256": hashlib.sha256(lab.astype("<f8").tobytes()).hexdigest(),
       "ref_x_s": round(t1 - t0, 2), "ref_y_s": round(t2 - t1, 2), "ref_pts_per_s": round(n / (t2 - t0), 1),
       "ref_host": "1 core, build container (Xeon @ 2.10GHz)"}
print(n, res, flush=True)
p = os.path.join(REPO, "tests/golden/dbscan_gen.json")
d = json.load(open(p)); d.setdefault(str(n), {}).update(res); json.dump(d, open(p, "w"), indent=1)
