# round 5's evidence in one call: the driver's command, the default bench line, kernel-trace stats of the headline loop, PMC passes (read
# requests; write requests) of the headline loop (fused_chunk_kernel<512, 5, 0>), of the heterogeneous block (fused_chunkd_kernel<512, 4, 0, ..>
# in the one-iteration regime; fused_chunkd_kernel<512, 4, 3, ..> + pcg_carry_flush_kernel in the loop) and of the uniform many-iteration
# loop (fused_chunk_kernel<512, 5, 3>)
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench464_driver_command.json 2> $O/r05_bench464_driver_command.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r05_kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_kstats -- python3 $R/bench.py --no-other-configs --no-multi-iteration --no-hetero --no-cpu-baseline > $O/r05_bench464_headline_only_under_rocprof.json 2> $O/r05_kstats.err
cd $R
cp $(ls gpurun_out/r05_kstats/*/*kernel_stats.csv | head -1) gpurun_out/r05_bench464_kernel_stats_headline_only.csv
python tools/trace_gaps.py gpurun_out/r05_kstats > gpurun_out/r05_bench464_trace_gaps.txt 2>&1 || true
echo "stats done" > $O/r05_progress.log
bash tools/run_bench_pmc.sh r05pmc
echo "headline pmc done" >> $O/r05_progress.log
cd /tmp
for what in hetero iter; do
  if [ $what = iter ]; then CMD="$R/tools/iter_ab.py 63=1"; else CMD="$R/tools/hetero_rate.py --steps 10 --warmup 3"; fi
  rm -rf $O/r05_${what}_rd $O/r05_${what}_wr $O/r05_${what}_st
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_${what}_st -- python3 $CMD > $O/r05_${what}_st.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/r05_${what}_rd -- python3 $CMD > $O/r05_${what}_rd.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/r05_${what}_wr -- python3 $CMD > $O/r05_${what}_wr.log 2>&1
  echo "$what done" >> $O/r05_progress.log
done
cd $R
for what in hetero iter; do
  cp $(ls gpurun_out/r05_${what}_st/*/*kernel_stats.csv | head -1) gpurun_out/r05_${what}_kernel_stats.csv || true
  for d in rd wr; do python tools/pmc_summary.py --min-frac 0.75 gpurun_out/r05_${what}_$d; done | grep -E "fused_step|fused_chunk|pcg_update|pcg_carry|pcg_ploop|spmv_symdia_tile|pcg_init" > gpurun_out/r05_${what}_pmc_live_launches.txt || true
done
timeout -k 10 600 python bench.py > $O/r05_bench464_default_run.json 2> $O/r05_bench464_default_run.err
