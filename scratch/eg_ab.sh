#!/bin/bash
# Same-box A/B of the NNConv on edge groups (default) against the NNConv on type columns (TGNN_GROUPS=0): the kernels alone
# (rocprofv3), inside the production two-stream forward (rocprofv3, 40 cached-layout forwards each), the forward's wall time,
# the bench line's headline.  Output: gpurun_out/eg_ab/summary.txt (-> profiles/r05_nnconv_eg.txt)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/eg_ab; mkdir -p $O; S=$O/summary.txt; : > $S
echo "== kernels alone, 100 000 nodes / 1 M edges, T = 13 (rocprofv3 --kernel-trace --stats -- python scratch/check_eg.py; 106 launches each)" >> $S
bash scratch/kstats_any.sh eg_ab_alone scratch/check_eg.py > /dev/null 2>&1
grep "nnconv32" gpurun_out/eg_ab_alone/stats.txt >> $S; grep -v "^W2026\|^E2026\|amdgpu.ids" gpurun_out/eg_ab_alone/log.txt | head -5 >> $S
echo "== kernels alone, 1 000 000 nodes / 10 M edges" >> $S
bash scratch/kstats_any.sh eg_ab_alone1m scratch/check_eg.py 1000000 10000000 > /dev/null 2>&1
grep "nnconv32" gpurun_out/eg_ab_alone1m/stats.txt >> $S; grep -v "^W2026\|^E2026\|amdgpu.ids" gpurun_out/eg_ab_alone1m/log.txt | sed -n 2,4p >> $S
for mode in columns groups; do
  echo "== inside the two-stream forward, 100 000 nodes, NNConv on $mode (rocprofv3 --kernel-trace --stats -- python scratch/run_fwd_groups.py $mode; 40 cached-layout forwards)" >> $S
  bash scratch/kstats_any.sh eg_ab_$mode scratch/run_fwd_groups.py $mode > /dev/null 2>&1
  head -6 gpurun_out/eg_ab_$mode/stats.txt >> $S
done
echo "== forward wall time, median of 30, A/B on this box (python scratch/time_eg_fwd.py)" >> $S
python scratch/time_eg_fwd.py 2>&1 | grep -v amdgpu.ids >> $S
echo "== bench.py headline (graph preparation inside the step), type columns (TGNN_GROUPS=0) then edge groups" >> $S
for g in 0 1; do
  TGNN_GROUPS=$g python bench.py --no-cpu-baseline --no-train-step --no-extra-sizes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('TGNN_GROUPS=$g: ms_per_step %.4f  value %.3e  cached %.4f  roofline: %s | avg_launch_us %.2f frac %.4f single-stream %.2f us gather_bound.frac %.3f' % (
    d['ms_per_step'], d['value'], d['cached_layout']['ms_per_step'], r['kernel'][:24], r['avg_launch_us'], r['frac'], r['single_stream']['avg_launch_us'], r.get('gather_bound', {}).get('frac', 0)))" >> $S
done
cat $S
