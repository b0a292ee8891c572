# LDS counters per kernel of the default BA bench: tools/lds_pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/ldspmc -- python3 $R/bench.py --no-extra --no-cpu-baseline --no-corr > /dev/null 2>&1 || echo "pass failed"
f=$(ls -t $R/gpurun_out/ldspmc/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
print(f"{'kernel':60s} {'LDS insts':>12s} {'LDS active cyc':>15s} {'conflict cyc':>13s} {'conflict %':>10s} {'active/busy %':>13s}")
for k, v in acc.items():
    m = {c: sum(x)/len(x) for c, x in v.items()}
    a = m.get('SQ_LDS_IDX_ACTIVE', 0.0)
    print(f"{k:60s} {m.get('SQ_INSTS_LDS',0):12.0f} {a:15.0f} {m.get('SQ_LDS_BANK_CONFLICT',0):13.0f} {100*m.get('SQ_LDS_BANK_CONFLICT',0)/max(a,1):10.1f} {100*a/max(m.get('SQ_BUSY_CYCLES',1),1):13.1f}")
PY
