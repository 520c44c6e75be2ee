#!/bin/bash
# The LMM calibration as a caller WITHOUT knowledge of the engine would run it (lmm_hip --finmath-like: no hold / flush / graph
# replication / lock-step batches, every time step's state kept, one getAverage per product — what finmath-lib's Euler scheme and
# optimizer do through the Java interface), with and without the time-step grouping BrownianMotionHip does on the caller's behalf
# (FMHIP_BM_GROUP_STEPS), against the native driver with all its hints.  3 LM iterations each (154 evaluations).
L=finmath-lib-cuda-extensions_amd/bin/lmm_hip
J='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("  %.2f s for %d evaluations = %.2f ms each, %d launches, %.1f TB algorithmic, rms %.6e" % (d["seconds"], d["evaluations"], d["seconds"]/d["evaluations"]*1e3, d["kernel_launches"], d["algorithmic_bytes"]/1e12, d["rms_deviation"]))'
echo "native driver, all hints:";            $L --paths 1000000 --max-iterations 3 | python3 -c "$J"
echo "finmath-like caller:";                 $L --paths 1000000 --max-iterations 3 --finmath-like | python3 -c "$J"
for S in 1 2 4 8; do
  echo "finmath-like caller, BrownianMotionHip groups $S time steps:"; FMHIP_BM_GROUP_STEPS=$S $L --paths 1000000 --max-iterations 3 --finmath-like | python3 -c "$J"
done
