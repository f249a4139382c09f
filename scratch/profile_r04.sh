#!/bin/bash
# Round-4 evidence for profiles/: the bench line, rocprofv3 kernel stats of the production forward alone, HBM / SQ counters of the
# column NNConv and the GIN pair (separate --pmc passes, kernel-trace only, every profiler command under its own timeout), HBM bytes of
# the whole forward per kernel, the timeline of the benchmark shape, mid sizes (times, phases, timeline), config 3 (stats + counters),
# the greedy solve, the sharded step at world 1.  Output: gpurun_out/r04/ (copied to profiles/r04_*).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
rm -rf /tmp/r04_fw; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_fw -- python scratch/run_fwd_mode.py 1 20 > /tmp/r04_fw.log 2>&1
python scratch/kstats.py $(find /tmp/r04_fw -name "*kernel_stats.csv" | head -1) 26 > $O/forward_kernel_stats.txt
timeout 200 python scratch/run_stamped.py > $O/stamped_nnconv.txt 2>&1
: > $O/pmc_nnconv.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES" ; do
  i=$((i+1)); rm -rf /tmp/pmc_r04_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r04_$i -- python scratch/run_nnconv_only.py nnconv > /tmp/pmc_r04_$i.log 2>&1
  f=$(find /tmp/pmc_r04_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py cols_kernel $f >> $O/pmc_nnconv.txt
done
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_r04_g
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r04_g -- python scratch/run_nnconv_only.py gin > /tmp/pmc_r04_g.log 2>&1
  f=$(find /tmp/pmc_r04_g -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { python scratch/pmc.py gin32_aggregate $f | sed 's/^/gin32_aggregate /'; python scratch/pmc.py gin32_mlp $f | sed 's/^/gin32_mlp /'; } >> $O/pmc_nnconv.txt
done
bash scratch/pmc_forward.sh > $O/pmc_forward.txt 2>&1
# timelines: the benchmark shape, 10 000 and 20 000 nodes (the tracer serialises the chains: durations, not overlap)
bash scratch/mid_trace.sh 100000 > /dev/null 2>&1; cp gpurun_out/mid_trace_100000/timeline.txt $O/trace_100000.txt
bash scratch/mid_trace.sh 10000 > /dev/null 2>&1; cp gpurun_out/mid_trace_10000/timeline.txt $O/mid_trace_10000.txt
timeout 300 python scratch/time_mid.py > $O/mid_sizes.txt 2>&1
timeout 300 python scratch/mid_phases.py > $O/mid_phases.txt 2>&1
# fused GIN against the two kernels, alone and in the forward
timeout 300 python scratch/gin_isolated.py > $O/gin_fused.txt 2>&1
timeout 300 python scratch/time_gin_fused.py >> $O/gin_fused.txt 2>&1
# dense rows kernel
bash scratch/kstats_dense.sh r04_dense > $O/dense_rows.txt 2>&1
# config 3
bash scratch/kstats3.sh r04_config3 > $O/config3_kernel_stats.txt 2>&1
timeout 200 python scratch/time_config3.py 2>&1 | grep bfloat16 >> $O/config3_kernel_stats.txt
bash scratch/pmc_config3.sh > $O/config3_pmc.txt 2>&1
# greedy solve (host sweep against the device rounds), sharded step at world 1
timeout 600 python scratch/time_solve.py > $O/greedy_solve.txt 2>&1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --force-sharded --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_sharded_world1.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29578 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --force-sharded --scaling strong --config 4 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_strong_config4_world1.json
ls -la $O
