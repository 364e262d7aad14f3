#!/bin/bash
# usage (on the GPU box): tools/clock_table.sh TAG [CONFIG] -> gpurun_out/clock_TAG.txt   (CONFIG: a bench.py --config name, default cfg2)
# ONE rocprofv3 --pmc pass (no kernel trace beside it) over a short bench run: per conv launch of the last step the
# clock the chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the share of matrix-pipe cycles that were busy
# (SQ_VALU_MFMA_BUSY_CYCLES / (held cycles x 1024 SIMDs)).
TAG=$1; CFG=${2:-cfg2}
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O/clock_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-parity-mode > $O/clock_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/clock_report.py $O/clock_$TAG > $O/clock_$TAG.txt && tail -80 $O/clock_$TAG.txt
