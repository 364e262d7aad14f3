"""Per conv launch of the LAST step of a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ...` run of bench.py:
duration, held clock, matrix-pipe busy share.  usage: clock_report.py <rocprof output dir>"""
import collections, csv, glob, sys
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", ""),
                                                    "t0": int(r.get("Start_Timestamp", 0) or 0), "t1": int(r.get("End_Timestamp", 0) or 0)})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
conv = [i for i in ids if any(k in rows[i]["name"] for k in ("igemm2_kernel", "gemm1_kernel", "wgrad2_kernel", "first_fprop", "first_wgrad", "roll3d_kernel"))]
# the last step = the conv launches after the last optimizer step but one
sgd = [i for i in ids if "sgd" in rows[i]["name"].lower()]
lo = sgd[-2] if len(sgd) >= 2 else 0
hi = sgd[-1] if sgd else ids[-1]
print(f"{'#':>3s} {'us':>8s} {'MHz':>6s} {'mfma%':>6s} {'wait%':>6s}  kernel")
tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
k = 0
for i in conv:
    if not (lo < i < hi):
        continue
    d = rows[i]
    us = (d["t1"] - d["t0"]) / 1e3
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    mhz = cyc / us if us > 0 else 0.0
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc else 0.0
    wait = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"] if d.get("SQ_WAVE_CYCLES") else 0.0
    name = d["name"].replace("void ", "").split("(")[0][:70]
    print(f"{k:3d} {us:8.1f} {mhz:6.0f} {100*busy:6.1f} {100*wait:6.1f}  {name} grid {d['grid']}")
    t = tot[name]; t[0] += us; t[1] += cyc; t[2] += busy * us
    k += 1
print()
for name, (us, cw, bw) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{us/1e3:7.3f} ms  {cw/us if us else 0:6.0f} MHz (time-weighted)  mfma busy {100*bw/us:5.1f} %  {name}")
