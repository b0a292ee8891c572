# gather16 calibration: wall-clock table + FETCH_SIZE per mode (separate rocprofv3 --pmc pass per mode)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gather16
mkdir -p $O
$R/tools/micro/gather16 > $O/timing.txt 2>&1
for m in 0 1 2 3 4 5; do
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_m$m -- $R/tools/micro/gather16 $m > /dev/null 2>&1
done
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rdreq_m1 -- $R/tools/micro/gather16 1 > $O/rdreq_m1.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rdreq_m3 -- $R/tools/micro/gather16 3 > $O/rdreq_m3.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rdreq_m0 -- $R/tools/micro/gather16 0 > $O/rdreq_m0.log 2>&1 || true
cat $O/timing.txt
