#!/bin/bash
# usage (on the GPU box): tools/kstats.sh TAG [pattern]  -> per-kernel average durations of a 3-step bench run
TAG=$1; PAT=${2:-.}
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-h2d --no-parity-mode > $O/ks_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$O/ks_$TAG" "$PAT" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
steps = 6
tot = 0
for r in csv.DictReader(open(f)):
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    if re.search(sys.argv[2], r["Name"]) and ms > 0.05:
        print(f"{ms:7.3f} ms/step  x{int(r['Calls'])/steps:5.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
print(f"sum of all kernels: {tot:.2f} ms/step")
PY
