#!/bin/bash
# Round 5's own evidence, beside benchmarks/profile_round.sh: the caller without hints (RAII and under a collector that releases every
# 100 ms) — wall time, host profile, every launch bracketed by events with its shape — and rocprofv3 kernel statistics of the RAII run.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
$L $A > $OUT/nohints_line.json 2> $OUT/nohints.err; echo "raii done"
FMHIP_PROFILE_DUMP=1 $L $A --profile > $OUT/nohints_profiled_line.json 2> $OUT/nohints_profile_dump.txt; echo "raii profiled"
FMHIP_HOST_PROFILE=1 $L $A > $OUT/nohints_host_line.json 2> $OUT/nohints_host_profile.txt; echo "raii host profile"
$L $A --release-lag 100 > $OUT/lag100_line.json 2> $OUT/lag100.err; echo "lag100 done"
FMHIP_PROFILE_DUMP=1 $L $A --release-lag 100 --profile > $OUT/lag100_profiled_line.json 2> $OUT/lag100_profile_dump.txt; echo "lag100 profiled"
FMHIP_HOST_PROFILE=1 $L $A --release-lag 100 > $OUT/lag100_host_line.json 2> $OUT/lag100_host_profile.txt; echo "lag100 host profile"
FMHIP_ESCAPE_POLICY=0 $L $A --release-lag 100 > $OUT/lag100_round4_rule_line.json 2> $OUT/lag100_round4_rule.err; echo "lag100 with round 4's rule done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nohints_stats -o nohints -- $L $A > $OUT/nohints_rocprof_line.json 2> $OUT/nohints_rocprof.err; echo "rocprof done"
cd $GRAFT_REPO_ROOT
