#!/bin/bash
# The LMM calibration with the Euler-step groups as rolled loops (default) against the segmented launches (FMHIP_ROLL=0), for
# 2, 4 and 8 steps per group: wall time, launches, algorithmic bytes, summed kernel time.
B=./finmath-lib-cuda-extensions_amd/bin/lmm_hip
for S in 2 4 8; do for R in 0 1; do
  FMHIP_ROLL=$R $B --paths 1000000 --mode calibrate --max-iterations 1 --steps-per-launch $S > /dev/null 2>&1     # fills the code-object cache
  echo "== steps-per-launch $S, rolled $R"
  FMHIP_ROLL=$R $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','evaluations','mean_deviation','kernel_launches','algorithmic_bytes','specialised_kernels')})"
  FMHIP_ROLL=$R $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S --profile | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','kernel_ms_total','achieved_GBps','profiled_launches')})"
done; done
