#!/bin/bash
# Round-6 evidence for profiles/: the bench line, rocprofv3 kernel stats of the bench command and of the production forward alone, HBM /
# SQ counters of the edge-group NNConv and the GIN pair (separate --pmc passes, kernel-trace only, every profiler command under its own
# timeout), HBM bytes of the whole forward per kernel, timelines (cached forward, full step with preparation, preparation alone),
# the final MLP's kernels, mid sizes, the sharded step at world 1 (bench + timeline).  Output: gpurun_out/r06/ (copied to profiles/r06_*).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
rm -rf /tmp/r06_ks; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_ks -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-extra-sizes > /tmp/r06_ks.log 2>&1
f=$(find /tmp/r06_ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp $f $O/kernel_stats.csv; python scratch/kstats.py $f 34 > $O/kernel_stats_summary.txt; }
rm -rf /tmp/r06_fw; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_fw -- python scratch/run_fwd_mode.py 1 20 > /tmp/r06_fw.log 2>&1
python scratch/kstats.py $(find /tmp/r06_fw -name "*kernel_stats.csv" | head -1) 26 > $O/forward_kernel_stats.txt
timeout 200 python scratch/run_stamped.py > $O/stamped_nnconv.txt 2>&1
: > $O/pmc_nnconv.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES" ; do
  i=$((i+1)); rm -rf /tmp/pmc_r06_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r06_$i -- python scratch/run_nnconv_only.py nnconv > /tmp/pmc_r06_$i.log 2>&1
  f=$(find /tmp/pmc_r06_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py nnconv32_eg $f >> $O/pmc_nnconv.txt
done
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_r06_g
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r06_g -- python scratch/run_nnconv_only.py gin > /tmp/pmc_r06_g.log 2>&1
  f=$(find /tmp/pmc_r06_g -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { python scratch/pmc.py gin32_aggregate $f | sed 's/^/gin32_aggregate /'; python scratch/pmc.py gin32_mlp $f | sed 's/^/gin32_mlp /'; } >> $O/pmc_nnconv.txt
done
bash scratch/pmc_forward.sh > $O/pmc_forward.txt 2>&1
bash scratch/mid_trace.sh 100000 > /dev/null 2>&1; cp gpurun_out/mid_trace_100000/timeline.txt $O/trace_100000.txt
bash scratch/step_trace.sh 100000 r06_step > /dev/null 2>&1; cp gpurun_out/r06_step/step_timeline.txt $O/step_trace_100000.txt
bash scratch/prep_trace_n.sh 100000 > $O/prep_trace_100000.txt 2>&1
bash scratch/prep_trace_n.sh 10000 > $O/prep_trace_10000.txt 2>&1
bash scratch/mid_trace.sh 10000 > /dev/null 2>&1; cp gpurun_out/mid_trace_10000/timeline.txt $O/mid_trace_10000.txt
timeout 300 python scratch/time_mid.py > $O/mid_sizes.txt 2>&1
AB_REPS=6 timeout 600 python scratch/ab_head.py 100000 0,3 2>&1 | grep -v amdgpu > $O/head_ab.txt
AB_REPS=6 timeout 600 python scratch/ab_tail.py 100000 0,2 2>&1 | grep -v amdgpu > $O/tail_ab.txt
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --force-sharded --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_sharded_world1.json
bash scratch/sharded_trace.sh > $O/sharded_trace.txt 2>&1
ls -la $O
