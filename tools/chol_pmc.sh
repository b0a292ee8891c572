# LDS counters of the single-launch Cholesky (n = 1530): tools/chol_pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $R/gpurun_out/cholpmc_$tag -- python3 $R/tools/chol_time.py > /dev/null 2>&1 || echo "pass failed: $grp"
  f=$(ls -t $R/gpurun_out/cholpmc_$tag/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'chol_factor_persistent' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f"{k:28s} per launch {sum(v)/len(v):14.0f}   (launches {len(v)})")
PY
done
