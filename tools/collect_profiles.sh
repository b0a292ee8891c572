set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 16 --warmup 3 > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_summary.json gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/pmc_summary.txt
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 600 gpurun_out/bench_final.json
