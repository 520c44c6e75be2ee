#!/bin/bash
# A/B of the LMM op stream: steps per launch 1 vs 2 (profiled replays, every program specialised), then calibrations
B=./finmath-lib-cuda-extensions_amd/bin/lmm_hip
for S in 1 2; do
  FMHIP_JIT=sync $B --paths 1000000 --mode evaluate --evaluations 8 --jacobian-batch 8 --steps-per-launch $S > /dev/null 2>&1
  echo "== steps-per-launch $S, 8 in lock-step"
  FMHIP_JIT=sync FMHIP_PROFILE_DUMP=1 $B --paths 1000000 --mode evaluate --evaluations 16 --jacobian-batch 8 --steps-per-launch $S --profile 2> gpurun_out/r2f/dump_S${S}_K8.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('profiled_launches','kernel_ms_total','achieved_GBps','algorithmic_bytes','specialised_kernels','seconds_simulation_per_evaluation','seconds_valuation_per_evaluation')})"
  FMHIP_JIT=sync $B --paths 1000000 --mode evaluate --evaluations 1 --jacobian-batch 1 --steps-per-launch $S > /dev/null 2>&1
  echo "== steps-per-launch $S, one at a time"
  FMHIP_JIT=sync FMHIP_PROFILE_DUMP=1 $B --paths 1000000 --mode evaluate --evaluations 4 --jacobian-batch 1 --steps-per-launch $S --profile 2> gpurun_out/r2f/dump_S${S}_K1.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('profiled_launches','kernel_ms_total','achieved_GBps','algorithmic_bytes','specialised_kernels','seconds_simulation_per_evaluation','seconds_valuation_per_evaluation')})"
done
for S in 1 2; do
  echo "== calibration, steps-per-launch $S (warm code-object cache)"
  $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','iterations','evaluations','mean_deviation','rms_deviation','kernel_launches','specialised_kernels','specialisations_from_disk_cache','algorithmic_bytes')})"
done
echo "== calibration, steps-per-launch 2, COLD code-object cache"
FMHIP_JIT_CACHE_DIR=/tmp/coldcache $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','iterations','evaluations','mean_deviation','kernel_launches','specialised_kernels','specialisations_from_disk_cache','specialisations_pending','specialisation_seconds')})"
