#!/bin/bash
# SQ counters of the kernels of the LMM op stream (counter evidence for "the four-step Euler kernels are bound
# by vector issue"): 16 objective evaluations in lock-step batches of 8 behind one untimed batch, kernels from the build-time pack
# (FMHIP_JIT=sync: nothing is compiled in the background while the counters run).  One rocprofv3 --pmc pass per counter group, the
# driver binary directly behind `--`.  Output: gpurun_out/$1/pmc_{a,b}/…counter_collection.csv, summary by benchmarks/pmc_lmm_summary.py.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd/bin/lmm_hip
export FMHIP_JIT=sync
export FMHIP_COMMON_ROWS=0      # the replay values one parameter set eight times: identical rows would be computed once (runtime.cpp: common rows) and the launches would be one row tall
ARGS="--paths 1000000 --mode evaluate --evaluations 16 --jacobian-batch 8 --warmup-evaluations 8"
$L $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_a -o a -- $L $ARGS > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -o b -- $L $ARGS > /dev/null 2> $OUT/pmc_b.err
python3 $GRAFT_REPO_ROOT/benchmarks/pmc_lmm_summary.py $OUT/pmc_a $OUT/pmc_b > $OUT/pmc_lmm_summary.txt
cd $GRAFT_REPO_ROOT
