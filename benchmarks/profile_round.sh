#!/bin/bash
# The rocprofv3 evidence of a round, written under gpurun_out/$1 (copy what is to be judged into profiles/):
#   bench_stats/     --kernel-trace --stats of the default stream bench (the kernel behind roofline.achieved)
#   pmc_fetch/, pmc_write/   separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same workload  → hbm_traffic.json
#   lmm_stats/       --kernel-trace --stats of 16 LMM objective evaluations in lock-step batches of 8
#   bm_write/        --pmc WRITE_SIZE of the normal-increment generator (config 3)
# rocprofv3 gets the program itself after `--` (python3 / the driver binary), never a shell or env wrapper.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --workload stream --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python3 $B --steps 20 --warmup 5 > $OUT/bench_profiled_line.json 2> $OUT/bench_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $B --steps 5 --warmup 2 --sustained-seconds 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $B --steps 5 --warmup 2 --sustained-seconds 0 > /dev/null 2> $OUT/pmc_write.err
python3 $GRAFT_REPO_ROOT/benchmarks/hbm_traffic_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/hbm_traffic.json
L=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd/bin/lmm_hip
export FMHIP_JIT=sync
export FMHIP_COMMON_ROWS=0      # (the lock-step replay values one parameter set eight times: every row computed)
$L --paths 1000000 --mode evaluate --evaluations 8 --jacobian-batch 8 > /dev/null 2>&1
# 16 evaluations in lock-step batches of 8 behind one untimed batch (the first batch meets every graph shape for the first time)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lmm_stats -o lmm -- $L --paths 1000000 --mode evaluate --evaluations 16 --jacobian-batch 8 --warmup-evaluations 8 > $OUT/lmm_profiled_line.json 2> $OUT/lmm_stats.err
unset FMHIP_JIT FMHIP_COMMON_ROWS
# the whole calibration (621 evaluations), warm code-object cache
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lmm_calib_stats -o calib -- $L --paths 1000000 --mode calibrate --max-iterations 12 > $OUT/lmm_calib_line.json 2> $OUT/lmm_calib_stats.err
# the reference's smile calibration at its larger published path count (context workload)
S=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd/bin/lmm_smile_hip
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smile_stats -o smile -- $S --paths 163840 > $OUT/smile_line.json 2> $OUT/smile_stats.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bm_write -o bm -- python3 $GRAFT_REPO_ROOT/benchmarks/config3_heston.py > /dev/null 2> $OUT/bm_write.err
grep -h fm_bm_kernel $OUT/bm_write/*counter_collection.csv | head -3
cd $GRAFT_REPO_ROOT
