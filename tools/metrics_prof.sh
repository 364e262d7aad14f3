#!/bin/bash
# usage (GPU box): tools/metrics_prof.sh TAG -> gpurun_out/metrics_TAG.txt (call-level table + rocprofv3 kernel durations)
TAG=$1; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mp_$TAG -- python3 $GRAFT_REPO_ROOT/tools/metrics_bench.py > $O/metrics_$TAG.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$O/mp_$TAG" >> $O/metrics_$TAG.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
print("\nrocprofv3 --kernel-trace --stats (kernel durations only):")
n = 32 * 512 * 1024
byt = [("confusion_u8", 2 * n), ("class_confusion_kernel<unsigned char", 2 * n), ("class_confusion_kernel<long", 16 * n),
       (" confusion_kernel<long", 16 * n), ("sqdiff", 2 * n), ("column_absdiff", 2 * n)]
for r in csv.DictReader(open(f)):
    for k, b in byt:
        if k in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
            print(f"{r['Name'][:70]:70s} x{r['Calls']:>4s}  avg {us:8.1f} us  {b / us / 1e3:7.0f} GB/s  ({b / us / 1e3 / 6300:4.2f} of 6.3 TB/s)")
PY
cat $O/metrics_$TAG.txt
