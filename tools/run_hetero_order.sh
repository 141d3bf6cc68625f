# Does the order of the blocks in the default line (lean bench problem, csr_route_block, heterogeneous block) move the heterogeneous block's figures?
# (profiles/r05_default_line_block_order.log: no — 0.71-0.72 with the side block in front, 0.67-0.71 without)
for rep in 1 2; do
for extra in "" "--no-lean-block" "--lean off"; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']; h = c['heterogeneous_K']
print('[$extra] headline %.4f; hetero dt=60: %.3f ms frac %.3f; dt=7.5: %.3f ms frac %.3f' % (d['ms_per_step'], h['ms_per_step'], h['roofline']['frac'], h['one_iteration_regime']['ms_per_step'], h['one_iteration_regime']['roofline']['frac']))
"
done
done
