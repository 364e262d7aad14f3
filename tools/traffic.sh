#!/bin/bash
# usage (on the GPU box): tools/traffic.sh TAG  -> gpurun_out/traffic_TAG.json  (copy to profiles/r03_traffic.json)
# Two separate PMC passes over the bench command (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel
# trace NOT combined with them (MI355X_MICROARCH.md "rocprofv3 PMC slots").
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out
STEPS=2; WARM=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/traffic_$TAG/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-h2d --no-parity-mode > $O/traffic_$TAG.fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/traffic_$TAG/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-h2d --no-parity-mode > $O/traffic_$TAG.write.log 2>&1 &&
cd $GRAFT_REPO_ROOT && python3 tools/traffic_report.py $O/traffic_$TAG $((STEPS + WARM + 1)) > $O/traffic_$TAG.json && cat $O/traffic_$TAG.json
