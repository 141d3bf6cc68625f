# kernel-trace stats + separate --pmc passes (read requests; write requests) over the 5M-cell irregular mesh's stepping loop
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/ipmc_ks $O/ipmc_rd $O/ipmc_wr
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ipmc_ks -- python3 $R/tools/irregular_rate.py 55=1 > $O/ipmc_ks.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/ipmc_rd -- python3 $R/tools/irregular_rate.py 55=1 > $O/ipmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/ipmc_wr -- python3 $R/tools/irregular_rate.py 55=1 > $O/ipmc_wr.log 2>&1
cd $R
head -12 $(ls gpurun_out/ipmc_ks/*/*kernel_stats.csv | head -1) > gpurun_out/ipmc_kernel_stats_head.csv
for d in rd wr; do python tools/pmc_summary.py gpurun_out/ipmc_$d; done | grep -E "fused_sell|spmv_sell|pcg_update" > gpurun_out/ipmc_summary.txt
