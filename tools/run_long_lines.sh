#!/bin/bash
# Lines longer than the chunk kernels' block (640^3: 640 rows): the chunk traversal with three halo rounds against the 2-D tiles.
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "chunk_traversal" > gpurun_out/r5_long_tests.log 2>&1
python bench.py --ns 640 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/r5_bench640_chunks.json 2> gpurun_out/r5_bench640_chunks.err
FV_TUNE="60=0" python bench.py --ns 640 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/r5_bench640_tiles.json 2> gpurun_out/r5_bench640_tiles.err
