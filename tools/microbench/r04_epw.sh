cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for epw in 64 32 16 8; do
  export CW_EXP_EPW=$epw
  rm -rf gpurun_out/prof_epw$epw
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_epw$epw -o p -- python bench.py --quick --steps 600 --obs-mode state > /dev/null 2>&1
  echo "epw $epw state:"; grep "cw_step_fused" gpurun_out/prof_epw$epw/*/p_kernel_stats.csv gpurun_out/prof_epw$epw/p_kernel_stats.csv 2>/dev/null | cut -d, -f1-7
  python bench.py --quick --steps 600 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('epw $epw pixels  %.4e env-steps/s  ms/step %.4f  kernel avg %.4f' % (d['value'], d['ms_per_step'], r['avg_launch_ms']))"
done
