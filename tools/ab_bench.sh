#!/bin/bash
# A/B of engine switches on ONE box (boxes of the pool differ by several %): runs bench.py once per "VAR=value" argument (and once plain),
# interleaved twice, and prints patches/s + the exclusive per-kernel table.   usage: bash tools/ab_bench.sh FMRI_TAIL_FUSE=0 FMRI_PACK_OVERLAP=0
mkdir -p gpurun_out
for rep in 1 2; do
  for cfg in "" "$@"; do
    tag=${cfg:-default}
    env $cfg python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --val-dice-steps 0 --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
x=d.get('roofline_exclusive',{}).get('kernel_ms_per_step',{})
print('%-28s rep$rep  %.1f patches/s  %.3f ms   excl: %s' % ('$tag', d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(k.replace('conv_','').replace('_mfma',''),v) for k,v in x.items())))
"
  done
done
