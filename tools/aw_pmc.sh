# Instruction / stall counters of the alt-corr kernels (several rocprofv3 --pmc passes): tools/aw_pmc.sh [tag]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-aw}
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $R/gpurun_out/${tag}_pmc$i -- python3 $R/tools/alt_bench.py 128 > /dev/null 2>&1 || echo "pass $i failed: $grp"
done
python3 - $R/gpurun_out/${tag}_pmc* <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'altcorr' not in k: continue
            acc[k[:60] + f" grid={r.get('Grid_Size','?')}"][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {sum(x)/len(x):16.0f}   ({len(x)} launches)")
PY
