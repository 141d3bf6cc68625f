#!/bin/bash
# A/B of two builds of the library on the heterogeneous block of the bench (alternating processes): put the other build at .ab/libfvhip_base.so
# (mkdir .ab; cp finitevolume.jl_amd/libfvhip.so .ab/libfvhip_base.so before the change under test); the tree's build is the second one.
mkdir -p gpurun_out
cp finitevolume.jl_amd/libfvhip.so .ab/libfvhip_new.so
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-multi-iteration --no-lean-block"
for rep in 1 2 3 4; do
for which in base new; do
  cp .ab/libfvhip_$which.so finitevolume.jl_amd/libfvhip.so
  python bench.py $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
h = d['config']['heterogeneous_K']
print('rep $rep $which: headline %.4f ms; hetero: ' % d['ms_per_step'] + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items() if k in ('ms_per_step', 'pcg_iters_per_step')}) + ' one-iteration: ' + json.dumps(h.get('one_iteration_regime', {}).get('ms_per_step')))
" >> gpurun_out/r5_so_ab.log
done
done
cp .ab/libfvhip_new.so finitevolume.jl_amd/libfvhip.so
