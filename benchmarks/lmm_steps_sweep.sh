B=./finmath-lib-cuda-extensions_amd/bin/lmm_hip
for S in 2 4 6; do
  $B --paths 1000000 --mode calibrate --max-iterations 1 --steps-per-launch $S > /dev/null 2>&1
  echo "== steps-per-launch $S"
  $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','evaluations','mean_deviation','kernel_launches','algorithmic_bytes','specialised_kernels')})"
  $B --paths 1000000 --mode calibrate --max-iterations 12 --steps-per-launch $S --profile | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','kernel_ms_total','achieved_GBps','profiled_launches')})"
done
