# separate --pmc passes over the bench's headline loop (read requests; write requests + L2 hit/miss; SQ wave-cycle split)
# usage: bash tools/run_bench_pmc.sh [TAG [FV_TUNE string]]   -> gpurun_out/TAG_summary.txt
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
TAG=${1:-bpmc}
if [ -n "${2:-}" ]; then export FV_TUNE="$2"; fi
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
ARGS="${EXTRA_ARGS:-} --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-profile --no-other-configs --no-multi-iteration --no-hetero"
rm -rf $O/${TAG}_rd $O/${TAG}_wr $O/${TAG}_sq
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/${TAG}_rd -- python3 $R/bench.py $ARGS > $O/${TAG}_rd.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_wr -- python3 $R/bench.py $ARGS > $O/${TAG}_wr.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/${TAG}_sq -- python3 $R/bench.py $ARGS > $O/${TAG}_sq.log 2>&1
cd $R
for d in rd wr sq; do python tools/pmc_summary.py gpurun_out/${TAG}_$d; done | grep -E "fused_step|fused_chunk|spmv_dia_kernel<true|q_to_v|pcg_pupdate|spmv_symdia_tile" > gpurun_out/${TAG}_summary.txt
