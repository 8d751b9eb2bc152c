# Round-end measurement, part A (run on the GPU box from the repo root): the driver's bench command under rocprofv3 --kernel-trace --stats,
# then the configuration of the round-1 abort (Encoder(max_batch=2560) on 64x64 pictures) once under the profiler.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout -k 10 560 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_kt -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_bench_under_rocprof.err || exit 1
python3 $R/tools/rocprof_stats.py $R/gpurun_out/r03_kt $R/gpurun_out/r03_bench_kernel_stats.csv > /dev/null || exit 1
tail -c 600 $R/gpurun_out/r03_bench_under_rocprof.json
head -3 $R/gpurun_out/r03_bench_kernel_stats.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_q1 -o q1 -- python3 $R/tools/quick_timing.py 64 64 2560 > $R/gpurun_out/r03_q1.log 2>&1
echo "q1 exit code $?" >> $R/gpurun_out/r03_q1.log
tail -n 3 $R/gpurun_out/r03_q1.log
