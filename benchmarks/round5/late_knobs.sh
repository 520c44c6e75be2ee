B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
one() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('engine',{}); print('$1', d['seconds'], 's', d['kernel_launches'], 'launches', e.get('interpreter_launches'), 'interp', d['specialised_kernels'], 'kernels', e.get('values_demanded'), 'demanded', 'peak GB', e.get('peak_bytes_reserved',0)>>30, 'late: waiting', e.get('late_releases_while_waiting'), 'at once', e.get('late_releases_at_once'), e.get('late_release_seconds'), 's')"; }
$B $A | one "raii"
for eager in 4096 32768 131072; do for portion in 48; do
FMHIP_LATE_EAGER=$eager FMHIP_LATE_PORTION=$portion $B $A --release-lag 100 | one "lag100 eager $eager portion $portion"
done; done
$B $A | one "raii"
