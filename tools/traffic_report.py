"""HBM traffic of the conv stack per training step from two rocprofv3 --pmc passes over bench.py
(FETCH_SIZE and WRITE_SIZE in separate passes; MI355X_MICROARCH.md 'HBM': FETCH_SIZE is reported in
KiB and counts HALF of the bytes of wide coalesced reads on gfx950 -> doubled here; WRITE_SIZE is exact).
usage: traffic_report.py <dir with fetch/ and write/ sub-directories> <steps run in the process>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest  # noqa: E402  (bench.py withholds roofline.traffic when the digest differs)

root, steps = sys.argv[1], int(sys.argv[2])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            k = r["Kernel_Name"]
            cls = ("conv" if ("igemm" in k or "gemm1" in k or "wgrad" in k or "first_" in k) else "other")
            tot[cls][ctr] += float(r["Counter_Value"]) * 1024.0
            if ctr == "FETCH_SIZE":
                cnt[cls] += 1
out = {}
for cls, d in tot.items():
    rd, wr = 2.0 * d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
    out[cls] = {"read_GB_per_step": round(rd / steps / 1e9, 3), "write_GB_per_step": round(wr / steps / 1e9, 3),
                "launches_per_step": round(cnt[cls] / steps, 1),
                "traffic_bytes_per_launch": round((rd + wr) / max(cnt[cls], 1))}
out["csrc_digest"] = csrc_digest()
out["command"] = "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps S --warmup W"
print(json.dumps(out, indent=1))
