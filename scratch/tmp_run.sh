cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O; rm -rf $O/ks
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o r04 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-extra-sizes > $O/ks_bench.json 2> $O/ks.err
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python scratch/kstats.py $O/kernel_stats.csv 30 > $O/kernel_stats_summary.txt
cat $O/kernel_stats_summary.txt | head -12; tail -c 400 $O/ks_bench.json
