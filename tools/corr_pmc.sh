set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/corr_bench.py 256 f16 > $R/gpurun_out/corr_bench.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/cprof_fetch -- python3 $R/tools/corr_bench.py 256 f16 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/cprof_write -- python3 $R/tools/corr_bench.py 256 f16 > /dev/null 2>&1
tail -6 $R/gpurun_out/corr_bench.txt
