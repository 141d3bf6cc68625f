# separate --pmc passes over the bench's headline loop (read requests; write requests + L2 hit/miss; SQ wave-cycle split)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ARGS="--steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-profile --no-other-configs --no-multi-iteration --no-hetero"
rm -rf $O/bpmc_rd $O/bpmc_wr $O/bpmc_sq
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/bpmc_rd -- python3 $R/bench.py $ARGS > $O/bpmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/bpmc_wr -- python3 $R/bench.py $ARGS > $O/bpmc_wr.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/bpmc_sq -- python3 $R/bench.py $ARGS > $O/bpmc_sq.log 2>&1
cd $R
for d in rd wr sq; do python tools/pmc_summary.py gpurun_out/bpmc_$d; done | grep -E "fused_step|spmv_dia_kernel<true|q_to_v|pcg_pupdate|spmv_symdia_tile" > gpurun_out/bpmc_summary.txt
