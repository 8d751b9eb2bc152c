# Round-end measurement, part C: the batched P / B workloads on this build, and the PMC traffic of the P launch (two passes).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 python3 bench.py --workload ldp_p --frames 320 --steps 1 --warmup 0 > gpurun_out/r03_bench_ldp_p_320.json 2> gpurun_out/r03_ldp.err || exit 1
timeout -k 10 300 python3 bench.py --workload ra_b --frames 128 --steps 1 --warmup 0 > gpurun_out/r03_bench_ra_b_128.json 2> gpurun_out/r03_rab.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_ldp_f -o f -- python3 $R/bench.py --workload ldp_p --frames 320 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r03_pmc_ldp_f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_ldp_w -o w -- python3 $R/bench.py --workload ldp_p --frames 320 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r03_pmc_ldp_w.log 2>&1 || exit 1
python3 $R/tools/rocprof_stats.py $R/gpurun_out/r03_pmc_ldp_f /tmp/ldp_stats.csv | head -4
python3 - <<PY
import json
for n in ("ldp_p_320", "ra_b_128"):
    d = json.load(open("$R/gpurun_out/r03_bench_%s.json" % n)); print(n, d["value"], d["ms_per_step"], d.get("cpu_baseline", {}).get("value"))
PY
