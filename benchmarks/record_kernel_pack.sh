#!/bin/bash
# Records the program descriptions of the kernel pack (csrc/jit.hpp) on a GPU box: every program the flagship workloads ask the
# specialised tier for.  Usage (repo root):  bash benchmarks/record_kernel_pack.sh gpurun_out/kernel_pack.txt
# then replace the body of finmath-lib-cuda-extensions_amd/csrc/kernel_pack.txt with the output and rebuild.
set -e
OUT=${1:-gpurun_out/kernel_pack.txt}
RAW=$(mktemp)
export FMHIP_JIT_RECORD=$RAW FMHIP_JIT_CACHE_DIR=off FMHIP_JIT_PACK_DIR=off
BIN=finmath-lib-cuda-extensions_amd/bin
$BIN/lmm_hip --paths 1000000 --max-iterations 12 > /dev/null
$BIN/lmm_hip --paths 1000000 --mode evaluate --evaluations 8 --jacobian-batch 1 > /dev/null
FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 1 --finmath-like > /dev/null     # the caller without hints: grouped time steps, peeled product chains with their expectations;
                                                                                             # every program it runs (a short run promotes few on its own)
$BIN/lmm_smile_hip --paths 163840 > /dev/null
$BIN/lmm_smile_hip --paths 1000000 > /dev/null
python3 bench.py --workload stream --steps 5 --warmup 2 --sustained-seconds 0.2 --no-cpu-baseline > /dev/null
python3 benchmarks/config3_heston.py > /dev/null 2>&1 || true
sort -u "$RAW" > "$OUT"
rm -f "$RAW"
wc -l "$OUT"
