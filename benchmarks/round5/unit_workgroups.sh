# hint-free caller: up to how many spans in all a launch with fused reductions takes a UNIT of the reduction tree per workgroup (FMHIP_UNIT_WORKGROUPS; default 128)
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12"
for v in 128 512 1024 128 512; do
FMHIP_UNIT_WORKGROUPS=$v $B $A --finmath-like | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hint-free, unit launches up to $v spans:', d['seconds'], 's', d['kernel_launches'], 'launches', d.get('engine',{}).get('interpreter_launches'), 'on the interpreter', d['mean_deviation'])"
done
for v in 128 512; do
FMHIP_UNIT_WORKGROUPS=$v $B $A | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('native, unit launches up to $v spans:', d['seconds'], 's', d['kernel_launches'], 'launches', d['mean_deviation'])"
done
