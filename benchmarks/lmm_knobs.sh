#!/bin/bash
# LMM op stream (16 objective evaluations in lock-step batches of 8, every program specialised, device time from HIP events)
# under different settings of the environment knobs of the runtime (csrc/jit.cpp: jit_shape; runtime.cpp: launch()).
# Usage: benchmarks/lmm_knobs.sh "VAR=value VAR=value" "..." ...      ("" = defaults)
B=$(dirname "$0")/../finmath-lib-cuda-extensions_amd/bin/lmm_hip
for cfg in "$@"; do
  env FMHIP_JIT=sync $cfg $B --paths 1000000 --mode evaluate --evaluations 8 --jacobian-batch 8 > /dev/null 2>&1
  echo "config [$cfg]: $(env FMHIP_JIT=sync $cfg $B --paths 1000000 --mode evaluate --evaluations 16 --jacobian-batch 8 --profile | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in d if k.startswith(("profiled","achieved","seconds_"))})')"
done
