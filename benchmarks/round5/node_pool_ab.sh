# lag 100 with the node pool capped at 65536 nodes (freed nodes beyond it go back to malloc) against 4 M (they stay), a RAII run between them
B=finmath-lib-cuda-extensions_amd/bin/lmm_hip
A="--paths 1000000 --mode calibrate --max-iterations 12 --finmath-like"
one() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['seconds'], d['kernel_launches'], d.get('engine',{}).get('interpreter_launches'))"; }
$B $A | one "raii"
FMHIP_NODE_POOL_CAP=65536 $B $A --release-lag 100 | one "lag100 cap 65536"
$B $A | one "raii"
FMHIP_NODE_POOL_CAP=4194304 $B $A --release-lag 100 | one "lag100 cap 4M"
$B $A | one "raii"
FMHIP_NODE_POOL_CAP=65536 $B $A --release-lag 100 | one "lag100 cap 65536"
$B $A | one "raii"
FMHIP_NODE_POOL_CAP=4194304 $B $A --release-lag 100 | one "lag100 cap 4M"
FMHIP_NODE_POOL_CAP=4194304 FMHIP_DRAIN_PREFETCH=0 $B $A --release-lag 100 | one "lag100 cap 4M no prefetch"
