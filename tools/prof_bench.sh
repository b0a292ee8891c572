# rocprofv3 kernel stats of the default BA bench (no corr / cpu / extras): tools/prof_bench.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-run}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-corr > $R/gpurun_out/prof_$T.json 2> $R/gpurun_out/prof_$T.err
f=$(ls $R/gpurun_out/prof_$T/*/*kernel_stats.csv | tail -1)
cp $f $R/gpurun_out/prof_${T}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'droid' in r['Name'] or float(r['Percentage'])>1: print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
