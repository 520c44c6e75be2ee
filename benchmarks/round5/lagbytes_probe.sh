# Why does "never until 256 MB of dead wrappers" take 75 s?  The same run with more of the device left to the HIP runtime, with a smaller
# young generation, and with the host profile.
O=$1; mkdir -p $O
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
show() { python3 -c "
import json,sys
d=json.load(open('$1')); print('$2', {k:d.get(k) for k in ['seconds','kernel_launches','device_bytes_reserved','release_lag','engine']})"; }
FMHIP_POOL_HEADROOM_BYTES=34359738368 timeout -k 10 300 $B $A --release-lag-bytes 268435456 > $O/head32.json 2> $O/head32.err; show $O/head32.json "headroom 32 GiB"
timeout -k 10 300 $B $A --release-lag-bytes 67108864 > $O/b64.json 2> $O/b64.err; show $O/b64.json "64 MB young generation"
FMHIP_HOST_PROFILE=1 timeout -k 10 300 $B $A --release-lag-bytes 268435456 > $O/prof.json 2> $O/prof.err; show $O/prof.json "profiled"; grep -A14 "host profile" $O/prof.err
