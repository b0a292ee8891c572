# A/B over an environment variable: tools/env_ab.sh VAR v1 v2 ...
var=$1; shift
for val in "$@"; do env $var=$val python bench.py --no-extra --no-cpu-baseline --no-corr 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());s=d['config']['stage_ms'];print('$var=$val', round(d['value'],1), {k:round(v*1e3,1) for k,v in s.items()})"; done
