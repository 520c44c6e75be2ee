#!/bin/bash
# A/B on ONE box: the LMM calibration (1 M paths, 12 LM iterations) on the round-2 build kept under
# finmath-lib-cuda-extensions_amd/build/r2 (full IEEE expansion for every quotient) against the current build (ranged fast path
# of the division, fm_device_math.hpp: ueval_div_all); alternated, every launch bracketed by HIP events (--profile).
# usage: bash benchmarks/lmm_division_ab.sh <output file>
OUT=${1:-gpurun_out/lmm_division_ab.txt}
R=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd
OLD=$R/build/r2/bin/lmm_hip
NEW=$R/bin/lmm_hip
: > $OUT
for rep in 1 2; do
  for v in OLD NEW; do
    B=${!v}
    [ -x "$B" ] || continue
    echo "== $v run $rep (unprofiled, then profiled)" >> $OUT
    $B --paths 1000000 --mode calibrate --max-iterations 12 2>> $OUT.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print({k:r[k] for k in ('seconds','evaluations','mean_deviation','rms_deviation','kernel_launches','algorithmic_bytes')})" >> $OUT
    $B --paths 1000000 --mode calibrate --max-iterations 12 --profile 2>> $OUT.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print({k:r[k] for k in ('seconds','kernel_ms_total','achieved_GBps','profiled_launches','mean_deviation')}, 'frac', r['achieved_GBps']/8000, 'busy', r['kernel_ms_total']/1e3/r['seconds'])" >> $OUT
  done
done
cat $OUT
