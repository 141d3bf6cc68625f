#!/bin/bash
# The loop's vectors chosen by their write class (fv_place.hip) against plain allocations: alternating runs of the driver's command.
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-hetero --no-multi-iteration ${EXTRA:-}"
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
for place in 0 1; do
  FV_PLACE=$place python bench.py --ns ${NS:-464} $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
m = d['config'].get('multi_iteration', {})
print('rep $rep place $place: %.4f ms/step, kernel %.4f ms, frac %.3f; dt = 1 h: %.3f ms/step, frac %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], m.get('ms_per_step', 0), m.get('roofline', {}).get('frac')))
" >> gpurun_out/r5_place_ab.log
done
done
