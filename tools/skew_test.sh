for sk in 1 2 4; do DROID_HIP_LIB=$GRAFT_REPO_ROOT/tools/libs/lib_skew$sk.so python bench.py --no-extra --no-cpu-baseline --no-corr 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('skew $sk', d['value'], d['config']['stage_ms']['schur'])"; done
python bench.py --no-extra --no-cpu-baseline --no-corr 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('base', d['value'], d['config']['stage_ms']['schur'])"
