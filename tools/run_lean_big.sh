#!/bin/bash
# Lean set-up (FV_OPT_LEAN_SETUP): the bench problem without faces / CSR in HBM, and boxes beyond the int32 CSR (7 N > 2^31).
set -e
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-hetero"
python bench.py --ns 464 $B --lean on > gpurun_out/r5_bench464_lean.json 2> gpurun_out/r5_bench464_lean.err
python bench.py --ns 464 $B --lean off > gpurun_out/r5_bench464_csr.json 2> gpurun_out/r5_bench464_csr.err
timeout -k 10 400 python bench.py --ns 720 $B > gpurun_out/r5_bench720_lean.json 2> gpurun_out/r5_bench720_lean.err
