# LDS counters of the alt-corr / corr kernels: tools/alt_lds_pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/altldspmc -- python3 $R/tools/alt_bench.py > /dev/null 2>&1 || echo "pass failed"
f=$(ls -t $R/gpurun_out/altldspmc/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    m = {c: sum(x)/len(x) for c, x in v.items()}
    a = m.get('SQ_LDS_IDX_ACTIVE', 0.0)
    if a > 0: print(f"{k:70s} LDS insts {m.get('SQ_INSTS_LDS',0):10.0f} active {a:12.0f} conflict {m.get('SQ_LDS_BANK_CONFLICT',0):12.0f} ({100*m.get('SQ_LDS_BANK_CONFLICT',0)/max(a,1):5.1f} %)")
PY
