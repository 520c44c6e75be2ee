#!/bin/bash
# Per-shape tables (FMHIP_PROFILE_DUMP) of the profiled calibration under settings of ONE environment variable, on one box, alternated twice.
# usage: bash benchmarks/lmm_knob_dump.sh <output dir> <VARIABLE> <value> [<value> …]     ("-" = unset)
OUT=$1; VAR=$2; shift 2
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd/bin/lmm_hip
FILES=""
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    FMHIP_PROFILE_DUMP=1 $L --paths 1000000 --mode calibrate --max-iterations 12 --profile > $OUT/${VAR}_${v}_$rep.json 2> $OUT/${VAR}_${v}_dump_$rep.txt
    FILES="$FILES $OUT/${VAR}_${v}_dump_$rep.txt"
  done
done
unset $VAR
python3 $GRAFT_REPO_ROOT/benchmarks/lmm_dump_categories.py $FILES
