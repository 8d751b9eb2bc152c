set -x
for F in 32 64 128; do
  for M in "0 9" "1 9" "1 5"; do
    set -- $M
    HM355_TEAM=$1 HM355_TEAM_WAVES=$2 timeout -k 10 240 python bench.py --workload ldp_p --frames $F --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/sw_${F}_$1_$2.json 2> gpurun_out/sw_${F}_$1_$2.err || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/sw_${F}_$1_$2.json')); print('SWEEP frames $F team $1 waves $2:', round(d['value'],1), 'CTU/s', round(d['ms_per_step']), 'ms')"
  done
done
