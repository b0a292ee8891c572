# rocprofv3 kernel stats of rank 0 of the W-rank shard of configs[3]: tools/prof_scale.sh W
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-8}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scale$W -- python3 $R/tools/scale_one.py $W 8 > /dev/null 2>&1
f=$(ls $R/gpurun_out/prof_scale$W/*/*kernel_stats.csv | tail -1)
cp $f $R/gpurun_out/prof_scale${W}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'droid' in r['Name']: print(f"{r['Name'][:75]:75s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
