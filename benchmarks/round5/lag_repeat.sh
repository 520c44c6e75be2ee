# run-to-run spread of the hint-free calibration: temporaries that die at once against a collector every 100 ms / 20 ms
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
one() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['seconds'], d['kernel_launches'], d.get('engine',{}).get('interpreter_launches'), d.get('engine',{}).get('algorithmic_bytes_written'), d.get('engine',{}).get('peak_bytes_reserved'), 'late: waiting', d.get('engine',{}).get('late_releases_while_waiting'), 'at once', d.get('engine',{}).get('late_releases_at_once'), d.get('engine',{}).get('late_release_seconds'), 's')"; }
for i in 1 2 3; do
$B $A | one "raii"
$B $A --release-lag 100 | one "lag100"
$B $A --release-lag 20 | one "lag20"
done
