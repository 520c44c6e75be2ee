# The hint-free caller under a lagging collector (lmm_hip --finmath-like --release-lag): RAII, a collection every 100 ms, every 20 ms,
# "never until 256 MB of dead wrappers"; the native driver beside it.  Arguments as bench.py's lmm block (12 iterations).
# $1 = output directory, $2 = FMHIP_ESCAPE_POLICY (0: rounds 1-4, whatever has a handle is stored; 1: the engine learns)
O=${1:-gpurun_out/r5a}; POL=${2:-1}; mkdir -p $O
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12"
export FMHIP_ESCAPE_POLICY=$POL
timeout -k 10 300 $B $A --finmath-like > $O/raii.json 2> $O/raii.err && \
timeout -k 10 400 $B $A --finmath-like --release-lag 100 > $O/lag100.json 2> $O/lag100.err && \
timeout -k 10 400 $B $A --finmath-like --release-lag 20 > $O/lag20.json 2> $O/lag20.err && \
timeout -k 10 400 $B $A --finmath-like --release-lag-bytes 268435456 > $O/lagbytes.json 2> $O/lagbytes.err ; \
timeout -k 10 300 $B $A > $O/native.json 2> $O/native.err
for f in raii lag100 lag20 lagbytes native; do python3 -c "
import json,sys
try:
    d=json.load(open('$O/$f.json')); print('policy $POL', '$f', {k:d.get(k) for k in ['seconds','kernel_launches','mean_deviation','device_bytes_reserved','algorithmic_bytes','specialised_launches','specialised_kernels','release_lag','engine']})
except Exception as e: print('policy $POL', '$f', 'failed', e, open('$O/$f.err').read()[-400:])
"; done
