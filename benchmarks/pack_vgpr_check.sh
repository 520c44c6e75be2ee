#!/bin/bash
# Which compiler built the stream-S kernel, and what an occupancy hint does: run-time compilation inside the Python process (PyTorch
# loads its own bundled ROCm compiler libraries first) against the build-time pack (the system's hiprtc / comgr).
set -e
B="python3 bench.py --workload stream --steps 20 --warmup 5 --sustained-seconds 1.0 --no-cpu-baseline"
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("  vgprs", d["roofline"]["vgprs"], "disk hits", d["jit"]["disk_cache_hits"], "timed %.1f us" % d["roofline"]["avg_kernel_us"], "sustained %.1f us (second half %.1f)" % (d["sustained"]["avg_kernel_us"], d["sustained"]["second_half_avg_us"]))'
for W in 0 4; do
  export FMHIP_JIT_WAVES=$W
  echo "waves hint $W, run-time compilation in the Python process:"; FMHIP_JIT_PACK_DIR=off FMHIP_JIT_CACHE_DIR=/tmp/c1_$W $B 2>/dev/null | python3 -c "$J"
  echo "waves hint $W, pack built by jit_pack_tool (system compiler):"; rm -rf /tmp/pack_$W; finmath-lib-cuda-extensions_amd/build/jit_pack_tool finmath-lib-cuda-extensions_amd/csrc/kernel_pack.txt /tmp/pack_$W > /dev/null
  FMHIP_JIT_PACK_DIR=/tmp/pack_$W FMHIP_JIT_CACHE_DIR=/tmp/c2_$W $B 2>/dev/null | python3 -c "$J"
done
