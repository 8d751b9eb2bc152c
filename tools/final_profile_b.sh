# Round-end measurement, part B: the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of one 768-picture launch,
# the traffic record stamped with the library's build id, then the driver's plain bench command (which quotes the record because the ids match).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_f -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r03_pmc_f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_w -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r03_pmc_w.log 2>&1 || exit 1
python3 $R/tools/make_traffic_json.py $R/gpurun_out/r03_pmc_f $R/gpurun_out/r03_pmc_w $R/gpurun_out/r03_traffic.json 768 1 hm355_ctu_kernel || exit 1
cp $R/gpurun_out/r03_traffic.json $R/profiles/r03_traffic.json
cd $R
timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || exit 1
tail -c 900 gpurun_out/r03_bench.json
