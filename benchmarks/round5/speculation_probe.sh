# Hint-free caller: the device's idleness as the trigger of speculative flushes (FMHIP_SPECULATE_IDLE_MIN; 0 = off), RAII and lag 100, host profile
O=$1; mkdir -p $O
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
show() { python3 -c "
import json,sys
d=json.load(open('$1')); print('$2', {k:d.get(k) for k in ['seconds','kernel_launches','specialised_kernels']}, d.get('engine',{}).get('interpreter_launches'))"; }
for idle in 0 512 1024 2048 4096; do
  FMHIP_SPECULATE_IDLE_MIN=$idle timeout -k 10 300 $B $A > $O/raii_$idle.json 2> $O/raii_$idle.err; show $O/raii_$idle.json "raii idle_min=$idle"
done
FMHIP_SPECULATE_IDLE_MIN=1024 timeout -k 10 300 $B $A --release-lag 100 > $O/lag100.json 2> $O/lag100.err; show $O/lag100.json "lag100 idle_min=1024"
FMHIP_HOST_PROFILE=1 timeout -k 10 300 $B $A > $O/raii_prof.json 2> $O/raii_prof.err; show $O/raii_prof.json "raii profiled"; grep -A14 "host profile" $O/raii_prof.err
FMHIP_HOST_PROFILE=1 timeout -k 10 300 $B $A --release-lag 100 > $O/lag_prof.json 2> $O/lag_prof.err; show $O/lag_prof.json "lag100 profiled"; grep -A14 "host profile" $O/lag_prof.err
timeout -k 10 300 $B $A --profile > $O/raii_dev.json 2> $O/raii_dev.err; python3 -c "
import json; d=json.load(open('$O/raii_dev.json')); print('device', {k:d.get(k) for k in ['seconds','profiled_launches','kernel_ms_total','achieved_GBps']})"
