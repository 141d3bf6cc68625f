set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
MASK=${1:-15364}
PM=${2:-1024}
timeout -k 10 120 $R/tools/fused_proto 9 70 200 3 2 0 $MASK > $O/fused_small.log 2>&1
timeout -k 10 300 $R/tools/fused_proto 462 464 464 12 2 0 $MASK > $O/fused_464.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/proto_pmc_rd -- $R/tools/fused_proto 462 464 464 2 2 0 $PM > $O/proto_pmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/proto_pmc_wr -- $R/tools/fused_proto 462 464 464 2 2 0 $PM > $O/proto_pmc_wr.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/proto_pmc_sq -- $R/tools/fused_proto 462 464 464 2 2 0 $PM > $O/proto_pmc_sq.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/proto_pmc_rd fused > gpurun_out/proto_pmc_summary.txt
python tools/pmc_summary.py gpurun_out/proto_pmc_wr fused >> gpurun_out/proto_pmc_summary.txt
python tools/pmc_summary.py gpurun_out/proto_pmc_sq fused >> gpurun_out/proto_pmc_summary.txt
