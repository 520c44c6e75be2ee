# hint-free caller: how many recorded methods without a new time step before the engine runs what is pending on its own (FMHIP_SPECULATE_PENDING)
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
for v in 5000 1200 2000 3000 4000 6500 9000 0 5000; do
FMHIP_SPECULATE_PENDING=$v $B $A | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('speculate after $v methods:', d['seconds'], 's', d['kernel_launches'], 'launches', d.get('engine',{}).get('interpreter_launches'), 'on the interpreter', d['specialised_kernels'], 'kernels')"
done
