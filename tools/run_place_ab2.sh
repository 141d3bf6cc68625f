#!/bin/bash
# lean / CSR problem x vectors chosen by write class (FV_PLACE) x stagger inside the allocations (FV_ALLOC_SKEW): alternating runs of the driver's command
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-hetero --no-multi-iteration"
for rep in 1 2 3 4 5 6 7 8; do
for cfg in "off 0 0" "off 1 0" "off 1 4096" "on 1 4096" "on 1 0"; do
  set -- $cfg
  FV_PLACE=$2 FV_ALLOC_SKEW=$3 python bench.py --ns 464 $B --lean $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep lean $1 place $2 skew $3: %.4f ms/step, kernel %.4f ms, frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))
" >> gpurun_out/r5_place_ab2.log
done
done
