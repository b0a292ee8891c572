# The rocprofv3 passes behind profiles/r03_*: kernel stats of the default bench command, FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (no trace domains mixed in), then the plain bench line.  Run on the GPU box: bash tools/collect_profiles.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 16 --warmup 3 --no-extra > $O/stats.json 2> $O/stats.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra > $O/write.json 2> $O/write.err
cd $R && python3 tools/pmc_summary.py $O/pmc_summary.json $O/fetch $O/write > $O/pmc_summary.txt
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
