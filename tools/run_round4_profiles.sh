# round 4's evidence in one call: the driver's command, the default bench line, kernel-trace stats of the headline loop, PMC passes of the
# headline loop (fused_chunk_kernel<512, 5, 0>), of the many-iteration loop (fused_chunk_kernel<512, 5, 1> + pcg_update_w_kernel) and of the
# heterogeneous block (fused_step_kernel<16, 0, false>, the loop with the matrix streamed)
set -eu
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun does)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench464_driver_command.json 2> $O/r04_bench464_driver_command.err
timeout -k 10 600 python bench.py > $O/r04_bench464_default_run.json 2> $O/r04_bench464_default_run.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r04_kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_kstats -- python3 $R/bench.py --no-other-configs --no-multi-iteration --no-hetero --no-cpu-baseline > $O/r04_bench464_headline_only_under_rocprof.json 2> $O/r04_kstats.err
cd $R
cp $(ls gpurun_out/r04_kstats/*/*kernel_stats.csv | head -1) gpurun_out/r04_bench464_kernel_stats_headline_only.csv
python tools/trace_gaps.py gpurun_out/r04_kstats > gpurun_out/r04_bench464_trace_gaps.txt 2>&1 || true
bash tools/run_bench_pmc.sh r04pmc
# the many-iteration loop and the heterogeneous block under the same counters
cd /tmp
for what in iter hetero; do
  if [ $what = iter ]; then CMD="$R/tools/iter_ab.py 46=1"; else CMD="$R/tools/hetero_rate.py --steps 10 --warmup 3"; fi
  rm -rf $O/r04_${what}_rd $O/r04_${what}_wr
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/r04_${what}_rd -- python3 $CMD > $O/r04_${what}_rd.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/r04_${what}_wr -- python3 $CMD > $O/r04_${what}_wr.log 2>&1
done
cd $R
for what in iter hetero; do
  for d in rd wr; do python tools/pmc_summary.py gpurun_out/r04_${what}_$d; done | grep -E "fused_step|fused_chunk|pcg_update|pcg_carry|spmv_symdia_tile|pcg_init" > gpurun_out/r04_${what}_pmc_summary.txt || true
done
