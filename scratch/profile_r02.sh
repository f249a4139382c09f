#!/bin/bash
# Round-2 evidence for profiles/: the bench line, rocprofv3 kernel stats of the same command, HBM / SQ counters of the NNConv
# column kernel (separate --pmc passes, kernel-trace only, every profiler command under its own timeout).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 600 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err
rm -rf gpurun_out/r02/ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/ks -o r02 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-extra-sizes > gpurun_out/r02/ks.log 2>&1
cp $(find gpurun_out/r02/ks -name "*kernel_stats.csv" | head -1) gpurun_out/r02/kernel_stats.csv
python scratch/kstats.py gpurun_out/r02/kernel_stats.csv 40 > gpurun_out/r02/kernel_stats_summary.txt
i=0
: > gpurun_out/r02/pmc_nnconv.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_r02_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r02_$i -- python scratch/run_nnconv_only.py nnconv > /tmp/pmc_r02_$i.log 2>&1
  f=$(find /tmp/pmc_r02_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py cols_kernel $f >> gpurun_out/r02/pmc_nnconv.txt
done
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_r02_g
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_r02_g -- python scratch/run_nnconv_only.py gin > /tmp/pmc_r02_g.log 2>&1
  f=$(find /tmp/pmc_r02_g -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { python scratch/pmc.py gin32_aggregate $f | sed 's/^/gin32_aggregate /'; python scratch/pmc.py gin32_mlp $f | sed 's/^/gin32_mlp /'; } >> gpurun_out/r02/pmc_nnconv.txt
done
cat gpurun_out/r02/pmc_nnconv.txt
head -12 gpurun_out/r02/kernel_stats_summary.txt

# ---- the small-layout path: kernel stats of labyrinth forwards, phase timers of the persistent kernel, the timing table
rm -rf gpurun_out/r02/small_ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/small_ks -o small -- python scratch/small_trace.py > gpurun_out/r02/small_ks.log 2>&1
f=$(find gpurun_out/r02/small_ks -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp $f gpurun_out/r02/small_layout_kernel_stats.csv; python scratch/kstats.py $f 12 > gpurun_out/r02/small_layout_kernel_stats.txt; }
scratch/small_trace.sh > gpurun_out/r02/small_layout_timeline.txt 2>&1
timeout 300 python scratch/time_small.py 300 1254 2500 4096 > gpurun_out/r02/small_layout_times.txt 2>&1
scratch/prep_trace.sh > gpurun_out/r02/small_prep_timeline.txt 2>&1
if [ -f scratch/libs/libtgnn_SMALLTIME.so ]; then
  TGNN_LIB_PATH=$PWD/scratch/libs/libtgnn_SMALLTIME.so timeout 300 python scratch/small_phases.py laby 300 2500 > gpurun_out/r02/small_layout_phases.txt 2>&1
fi
timeout 300 python scratch/time_greedy.py > gpurun_out/r02/greedy_solve.txt 2>&1
