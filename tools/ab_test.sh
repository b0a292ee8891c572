# A/B timing of diagnostic builds: tools/ab_test.sh lib1 lib2 ...   (names under tools/libs, without lib_ / .so)
for n in "$@"; do DROID_HIP_LIB=$GRAFT_REPO_ROOT/tools/libs/lib_$n.so python bench.py --no-extra --no-cpu-baseline --no-corr 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());s=d['config']['stage_ms'];print('$n', round(d['value'],1), {k:round(v*1e3,1) for k,v in s.items()})"; done
python bench.py --no-extra --no-cpu-baseline --no-corr 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());s=d['config']['stage_ms'];print('product', round(d['value'],1), {k:round(v*1e3,1) for k,v in s.items()})"
