#!/bin/bash
# Records the program descriptions of the kernel pack (csrc/jit.hpp) on a GPU box: every program the flagship workloads ask the
# specialised tier for.  Usage (repo root):  bash benchmarks/record_kernel_pack.sh gpurun_out/kernel_pack.txt
# then replace the body of finmath-lib-cuda-extensions_amd/csrc/kernel_pack.txt with the output and rebuild.
# Round 5: the caller without hints for three Levenberg-Marquardt iterations — the escape policy (runtime.hpp) moves a shape through two
# or three variants while it learns which handles are never used again —, with temporaries that die at once and with a collector that
# releases them late (a JVM's lifetime contract), on one thread and on four.
set -e
OUT=${1:-gpurun_out/kernel_pack.txt}
RAW=$OUT.raw; mkdir -p $(dirname $OUT); : > $RAW      # (under gpurun_out/: a long recording is seen to be alive by the file growing)
step() { echo "[record_kernel_pack] $* ($(wc -l < $RAW) descriptions so far)"; }
export FMHIP_JIT_RECORD=$RAW FMHIP_JIT_CACHE_DIR=off FMHIP_JIT_PACK_DIR=off
BIN=finmath-lib-cuda-extensions_amd/bin
$BIN/lmm_hip --paths 1000000 --max-iterations 12 > /dev/null
step "done: $BIN/lmm_hip --paths 1000000 --max-iterations 12"
$BIN/lmm_hip --paths 1000000 --mode evaluate --evaluations 8 --jacobian-batch 1 > /dev/null
step "done: $BIN/lmm_hip --paths 1000000 --mode evaluate --evaluations 8 --jacobian-batch 1"
FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 3 --finmath-like > /dev/null    # the caller without hints: grouped time steps, peeled product chains with their expectations;
step "done: FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 3 --finmath-like"
                                                                                             # every program it runs (a short run promotes few on its own)
FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 3 --finmath-like --release-lag 100 > /dev/null
step "done: FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 3 --finmath-like --release-la"
FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 2 --finmath-like --release-lag-bytes 67108864 > /dev/null
step "done: FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 2 --finmath-like --release-la"
FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 2 --finmath-like --threads 4 > /dev/null
step "done: FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 2 --finmath-like --threads 4"
FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 2 --devices 0,0 > /dev/null
step "done: FMHIP_JIT=sync $BIN/lmm_hip --paths 1000000 --max-iterations 2 --devices 0,0"
$BIN/lmm_smile_hip --paths 163840 > /dev/null
step "done: $BIN/lmm_smile_hip --paths 163840"
$BIN/lmm_smile_hip --paths 1000000 > /dev/null
step "done: $BIN/lmm_smile_hip --paths 1000000"
python3 bench.py --workload stream --steps 5 --warmup 2 --sustained-seconds 0.2 --no-cpu-baseline > /dev/null
step "done: python3 bench.py --workload stream --steps 5 --warmup 2 --sustained-seconds 0.2 --no-cpu-b"
python3 benchmarks/config3_heston.py > /dev/null 2>&1 || true
step "done: python3 benchmarks/config3_heston.py"
sort -u "$RAW" > "$OUT"
rm -f "$RAW"
wc -l "$OUT"
