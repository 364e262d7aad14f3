#!/bin/bash
# usage (on the GPU box): tools/prof_cfg.sh TAG CONFIG [steps]  -> gpurun_out/ks_TAG/*kernel_stats.csv + a per-kernel table
# rocprofv3 kernel trace + stats of `bench.py --config CONFIG` (program itself after `--`, no wrapper: see the box rules)
TAG=$1; CFG=$2; STEPS=${3:-3}
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-h2d --no-parity-mode > $O/ks_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$O/ks_$TAG" "$STEPS" "$CFG" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
steps = int(sys.argv[2]) + (8 if sys.argv[3] in ("relaynet", "mgunet2") else 2) + (1 if sys.argv[3] == "cfg2" else 0)      # timed + warm-up (bench.py warms the autograd configs up for 8 steps) (+ the roofline step of cfg2)
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print(f"{sys.argv[3]}: sum of all kernels {tot:.2f} ms/step over {steps} steps")
for r in rows[:28]:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    print(f"{ms:8.3f} ms/step {100 * ms / tot:5.1f}%  x{int(r['Calls']) / steps:6.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
PY
