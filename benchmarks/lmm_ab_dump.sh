#!/bin/bash
# Per-shape A/B on ONE box: FMHIP_PROFILE_DUMP of the profiled calibration, old build (build/$OLDTAG, default r3) then new, twice.
OUT=${1:-gpurun_out/ab}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd
for rep in 1 2; do
  FMHIP_PROFILE_DUMP=1 $R/build/${OLDTAG:-r3}/bin/lmm_hip --paths 1000000 --mode calibrate --max-iterations 12 --profile > $OUT/old_$rep.json 2> $OUT/old_dump_$rep.txt
  FMHIP_PROFILE_DUMP=1 $R/bin/lmm_hip --paths 1000000 --mode calibrate --max-iterations 12 --profile > $OUT/new_$rep.json 2> $OUT/new_dump_$rep.txt
done
python3 $GRAFT_REPO_ROOT/benchmarks/lmm_dump_categories.py $OUT/old_dump_1.txt $OUT/new_dump_1.txt $OUT/old_dump_2.txt $OUT/new_dump_2.txt
