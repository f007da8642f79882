#!/bin/bash
# steps/s of the pair-kernel variants over the small / medium size range (measurement tool)
for n in 1024 2048 4096 8192 16384 32768 65536; do
  for v in gather rt1 rt2 rt4; do
    case $v in
      gather) export LJMD_N3=0; unset LJMD_N3_ROW_TILES;;
      rt1) export LJMD_N3=1 LJMD_N3_MIN_N=1 LJMD_N3_ROW_TILES=1;;
      rt2) export LJMD_N3=1 LJMD_N3_MIN_N=1 LJMD_N3_ROW_TILES=2;;
      rt4) export LJMD_N3=1 LJMD_N3_MIN_N=1 LJMD_N3_ROW_TILES=4;;
    esac
    steps=$(( 40000000 / n )); [ $steps -gt 3000 ] && steps=3000; [ $steps -lt 100 ] && steps=100
    python bench.py --particles $n --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$n $v', round(d['value'], 1), 'steps/s | ms/step', round(d['ms_per_step'], 4), 'pair', round(r['kernel_ms_avg'], 4), 'reduce', round(r['reduce_kick_finalize_ms_avg'], 4), r['kernel'])
"
  done
done
