#!/bin/bash
# lines of 928 rows (four halo rounds): five pairs per thread (spills) against four, alternating runs of the driver's command at 928^3
mkdir -p gpurun_out
B="--ns 928 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-hetero --no-lean-block"
for rep in 1 2 3; do
for v in np5 np4; do
  if [ $v = np5 ]; then export FV_HR4_NP5=1; else unset FV_HR4_NP5; fi
  python bench.py $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
m = d['config'].get('multi_iteration', {})
print('rep $rep $v: %.4f ms/step, kernel %.4f ms, frac %.3f; dt = 1 h: %.3f ms/step, frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], m.get('ms_per_step', 0), m.get('roofline', {}).get('frac', 0)))
" >> gpurun_out/r5_hr4_pairs.log
done
done
