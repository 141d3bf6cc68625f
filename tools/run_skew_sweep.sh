#!/bin/bash
# Where the large arrays sit relative to each other: the same launches, the arrays staggered inside their allocations (FV_ALLOC_SKEW bytes x (k mod 16)).
# Alternating runs of the driver's command (one process each: the allocations are made when the problem is created).
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-hetero --no-multi-iteration"
for rep in 1 2 3 4; do
for lean in off on; do
for s in 0 1024 4096 1048576 1114112; do
  FV_ALLOC_SKEW=$s python bench.py --ns 464 $B --lean $lean 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep lean $lean skew $s: %.4f ms/step, kernel %.4f ms, frac %.3f, spmv %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['spmv']['avg_launch_ms']))
" >> gpurun_out/r5_skew_sweep2.log
done
done
done
