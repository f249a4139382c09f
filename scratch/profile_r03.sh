#!/bin/bash
# Round-3 evidence for profiles/: the bench line, rocprofv3 kernel stats of the same command and of the production forward alone,
# HBM / SQ counters of the column NNConv kernel (separate --pmc passes, kernel-trace only, every profiler command under its own
# timeout), preparation, sharded step, split modes, side-by-side small layouts.  Output: gpurun_out/r03/ (copied to profiles/r03_*).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
rm -rf $O/ks
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o r03 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-extra-sizes > $O/ks_bench.json 2> $O/ks.err
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python scratch/kstats.py $O/kernel_stats.csv 40 > $O/kernel_stats_summary.txt
# the production forward alone (cached layout, two chains): what the kernels cost INSIDE the schedule
rm -rf /tmp/r03_fw; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_fw -- python scratch/run_fwd_mode.py 1 20 > /tmp/r03_fw.log 2>&1
python scratch/kstats.py $(find /tmp/r03_fw -name "*kernel_stats.csv" | head -1) 24 > $O/forward_kernel_stats.txt
python scratch/run_stamped.py > $O/stamped_nnconv.txt 2>&1
: > $O/pmc_nnconv.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1)); rm -rf /tmp/pmc_r03_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r03_$i -- python scratch/run_nnconv_only.py nnconv > /tmp/pmc_r03_$i.log 2>&1
  f=$(find /tmp/pmc_r03_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py cols_kernel $f >> $O/pmc_nnconv.txt
done
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_r03_g
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r03_g -- python scratch/run_nnconv_only.py gin > /tmp/pmc_r03_g.log 2>&1
  f=$(find /tmp/pmc_r03_g -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { python scratch/pmc.py gin32_aggregate $f | sed 's/^/gin32_aggregate /'; python scratch/pmc.py gin32_mlp $f | sed 's/^/gin32_mlp /'; } >> $O/pmc_nnconv.txt
done
# preparation
timeout 200 python scratch/time_prep.py > $O/prep_times.txt 2>&1
timeout 200 python scratch/time_prep_parts.py 100000 >> $O/prep_times.txt 2>&1
bash scratch/kprof_prep.sh > $O/prep_kernel_stats.txt 2>&1
# split modes, scale invariance, accuracy of the fp16-pair kernels
timeout 400 python scratch/split_modes.py > $O/split_modes.txt 2>&1
timeout 200 python scratch/scale_invariance.py >> $O/split_modes.txt 2>&1
timeout 200 python scratch/test_cols_f16.py >> $O/split_modes.txt 2>&1
# sharded step at world 1 (library RCCL): one exchange per layer on one stream / split exchange, and its timeline
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python scratch/time_sharded.py 2>&1 | grep two_streams > $O/sharded.txt
TGNN_LIBRARY_RCCL=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29556 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python scratch/time_sharded.py 2>&1 | grep two_streams | sed 's/^/host callbacks into torch.distributed: /' >> $O/sharded.txt
bash scratch/sharded_trace.sh 2>&1 | grep -v "^$" >> $O/sharded.txt
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --force-sharded --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_sharded_world1.json
# small layouts: side by side, the persistent kernel's times
timeout 200 python scratch/time_many.py > $O/forward_many.txt 2>&1
timeout 300 python scratch/time_small.py 300 1254 2500 4096 > $O/small_layout_times.txt 2>&1
timeout 300 python scratch/time_fwd.py > $O/forward_sizes.txt 2>&1

timeout 200 python scratch/mid_size.py > $O/mid_sizes.txt 2>&1
ls -la $O
