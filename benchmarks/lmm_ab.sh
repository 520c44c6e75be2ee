#!/bin/bash
# A/B on ONE box: the LMM calibration (1 M paths, 12 LM iterations) on the previous round's build kept under
# finmath-lib-cuda-extensions_amd/build/$OLDTAG (default r3) against the current build; alternated, each unprofiled (wall time) and with
# every launch bracketed by HIP events (--profile: kernel time, algorithmic rate).  usage: bash benchmarks/lmm_ab.sh <output file> [repetitions]
OUT=${1:-gpurun_out/lmm_ab.txt}
REPS=${2:-2}
R=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd
OLD=$R/build/${OLDTAG:-r3}/bin/lmm_hip
NEW=$R/bin/lmm_hip
: > $OUT
for rep in $(seq 1 $REPS); do
  for v in OLD NEW; do
    B=${!v}
    [ -x "$B" ] || continue
    echo "== $v run $rep (unprofiled, then profiled)" >> $OUT
    $B --paths 1000000 --mode calibrate --max-iterations 12 2>> $OUT.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print({k:r.get(k) for k in ('seconds','evaluations','mean_deviation','rms_deviation','kernel_launches','specialised_kernels','specialisations_from_disk_cache')})" >> $OUT
    $B --paths 1000000 --mode calibrate --max-iterations 12 --profile 2>> $OUT.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print({k:r[k] for k in ('seconds','kernel_ms_total','achieved_GBps','profiled_launches','specialised_launches','mean_deviation')}, 'frac %.4f' % (r['achieved_GBps']/8000), 'busy %.3f' % (r['kernel_ms_total']/1e3/r['seconds']))" >> $OUT
  done
done
cat $OUT
