B=./finmath-lib-cuda-extensions_amd/bin/lmm_hip
for S in 4 6 8 10; do
  $B --paths 1000000 --mode calibrate --max-iterations 1 --steps-per-launch $S > /dev/null 2>&1
  echo "== steps-per-launch $S (rolled)"
  $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','evaluations','mean_deviation','kernel_launches','algorithmic_bytes','specialised_kernels')})"
  $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S --profile | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','kernel_ms_total','achieved_GBps','profiled_launches')})"
done
echo "== profile dump S=4, 8 evaluations in lock-step"
FMHIP_JIT=sync $B --paths 1000000 --mode evaluate --evaluations 16 --jacobian-batch 8 --steps-per-launch 4 > /dev/null 2>&1
FMHIP_JIT=sync FMHIP_PROFILE_DUMP=1 $B --paths 1000000 --mode evaluate --evaluations 16 --jacobian-batch 8 --steps-per-launch 4 --profile 2>&1 >/dev/null | head -24
