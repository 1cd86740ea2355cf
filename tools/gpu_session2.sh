#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
O=$PWD/gpurun_out/prof
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $O/trace.log 2>&1; echo "trace rc=$?"
PROF_STEPS=0 PROF_ITERS=4 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc1 -o pmc1 -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
PROF_STEPS=0 PROF_ITERS=4 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD -d $O/pmc2 -o pmc2 -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
PROF_STEPS=0 PROF_ITERS=4 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc3 -o pmc3 -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $O/pmc3.log 2>&1; echo "pmc3 rc=$?"
cd $O; ls -R | head -50; find . -name "*.csv" | head; du -sh .
