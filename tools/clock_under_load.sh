#!/bin/bash
# Shader clock / power while bench.py's timed steps run, on live data and on the round 1-3 recipe (profiles/r04_data_dependence.json):
#   gpurun -- 'bash tools/clock_under_load.sh'  ->  gpurun_out/clock_under_load.txt
mkdir -p gpurun_out
OUT=gpurun_out/clock_under_load.txt
: > $OUT
for data in learnable survey; do
  echo "== --data $data" >> $OUT
  python3 bench.py --data $data --steps 600 --warmup 10 --no-cpu-baseline --val-dice-steps 0 --no-secondary --no-launch-timing > gpurun_out/clock_$data.json 2>/dev/null &
  PID=$!
  sleep 6
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr -s ' ' >> $OUT
    sleep 0.7
  done
  wait $PID
  python3 -c "import json; d=json.loads(open('gpurun_out/clock_$data.json').read().strip().splitlines()[-1]); print('patches/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'train dice', round(d['train_dice_last_step'],4))" >> $OUT
done
cat $OUT
