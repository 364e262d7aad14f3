import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "igemm" in k or "wgrad" in k or "first" in k:
            agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    vals = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(vals):
        print(f"   {c:28s} {vals[c]:.4g}")
    if "SQ_WAVE_CYCLES" in vals:
        wc = vals["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in vals: print(f"   {c}/WAVE_CYCLES = {vals[c]/wc:.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
            print(f"   MFMA_BUSY/BUSY_CYCLES = {vals['SQ_VALU_MFMA_BUSY_CYCLES']/vals['SQ_BUSY_CYCLES']:.3f} (per-SE busy; see guide)")
