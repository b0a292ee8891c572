# LDS / MFMA counters of the dense-slot SYRK kernels (rank 0 of the W-rank shard): tools/syrk_lds_pmc.sh W
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-1}
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $R/gpurun_out/syrkpmc_$tag -- python3 $R/tools/scale_one.py $W 4 > /dev/null 2>&1 || echo "pass failed: $grp"
  f=$(ls -t $R/gpurun_out/syrkpmc_$tag/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'syrk' in r['Kernel_Name']: acc[r['Kernel_Name'][12:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c, x in v.items()})
PY
done
