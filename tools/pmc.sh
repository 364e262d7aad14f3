#!/bin/bash
# usage: tools/pmc.sh TAG <conv_probe args...>   -> gpurun_out/pmc_TAG/{p1,p2,p3,p4}
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" 5 > $OUT.p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" 5 > $OUT.p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" 5 > $OUT.p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" 5 > $OUT.p4.log 2>&1
