#!/bin/bash
# Lean set-up (FV_OPT_LEAN_SETUP): the bench problem without faces / CSR in HBM, and boxes beyond the int32 CSR (7 N > 2^31) and
# beyond 2^32 bytes per vector (N > 5.4e8).
set -e
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-hetero"
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "lean_box" > gpurun_out/r5_lean_big_test.log 2>&1
python bench.py --ns 464 $B --lean on > gpurun_out/r5_bench464_lean.json 2> gpurun_out/r5_bench464_lean.err
timeout -k 10 400 python bench.py --ns 832 $B > gpurun_out/r5_bench832_lean.json 2> gpurun_out/r5_bench832_lean.err
timeout -k 10 400 python bench.py --ns 928 $B > gpurun_out/r5_bench928_lean.json 2> gpurun_out/r5_bench928_lean.err
