#!/bin/bash
# A/B of the host side of the LMM calibration: parameter sets of a Jacobian batch recorded by hand (FMHIP_LMM_CLONE=0) or
# recorded once and replicated inside the engine (fmhip_graph_clone), for several batch sizes.  Same box, alternating.
B=./finmath-lib-cuda-extensions_amd/bin/lmm_hip
$B --paths 1000000 --mode calibrate --max-iterations 2 > /dev/null 2>&1     # fills the code-object cache
for K in 8 16 25; do for C in 0 1; do
  echo "== jacobian-batch $K, clone $C"
  FMHIP_LMM_CLONE=$C $B --paths 1000000 --mode calibrate --max-iterations 12 --jacobian-batch $K | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print({k:d[k] for k in ('seconds','evaluations','mean_deviation','rms_deviation','kernel_launches','device_bytes_reserved')})"
done; done
echo "== host profile, batch 8, clone 1, 3 iterations"
FMHIP_HOST_PROFILE=1 $B --paths 1000000 --mode calibrate --max-iterations 3 2>&1 >/dev/null | tail -16
