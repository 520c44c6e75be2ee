# the configurations that ended in "malloc(): mismatching next->prev_size" on 2026-10-05, with a call stack on abort
O=$1; mkdir -p $O
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
export FMHIP_BACKTRACE=1
for late in 1 0; do
FMHIP_LATE_RELEASES=$late FMHIP_ESCAPE_POLICY=0 timeout -k 10 200 $B $A --release-lag 100 > $O/p0_lag100_late$late.json 2> $O/p0_lag100_late$late.err; echo "policy 0 lag100 late=$late rc $?"; tail -c 3000 $O/p0_lag100_late$late.err
FMHIP_LATE_RELEASES=$late FMHIP_ESCAPE_POLICY=1 timeout -k 10 200 $B $A --release-lag-bytes 268435456 > $O/p1_bytes_late$late.json 2> $O/p1_bytes_late$late.err; echo "policy 1 lagbytes late=$late rc $?"; tail -c 3000 $O/p1_bytes_late$late.err
done
