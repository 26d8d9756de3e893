# GPU box: bench.py on the other BASELINE shapes (1 x MI355X, eager launches unless noted; not the headline)
run() { python bench.py --quick --steps 600 "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-58s %.4e env-steps/s  %.4f ms/step  %s %.4f ms  frac %.3f' % ('$1', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac']))"; true; }
run "configs[2] headline: 65536 envs 21x21 full frames"
run "configs[3] per-GPU share: 131072 envs, eight mixed task menus" --envs-per-gpu 131072 --mixed-menus
run "131072 envs, one task list" --envs-per-gpu 131072
run "1048576 envs (66 GB of frames resident), mixed menus" --envs-per-gpu 1048576 --mixed-menus --steps 100 --warmup 10
run "configs[4]: 65536 envs 32x32 full frames (49 KB each)" --size 32
run "configs[1]: 4096 envs state-only" --envs-per-gpu 4096 --obs-mode state
run "configs[1]: 4096 envs state-only, HIP graphs of 16 steps" --envs-per-gpu 4096 --obs-mode state --graph-steps 16 --steps 608 --warmup 16
run "configs[1]: 4096 envs state-only, persistent rollout kernel" --envs-per-gpu 4096 --obs-mode state --rollout
run "65536 envs state-only, persistent rollout kernel" --obs-mode state --rollout
run "65536 envs full frames, phases spread out (--desync)" --desync
run "65536 envs dirty-cell, phases spread out" --obs-mode pixels_dirty --desync
run "65536 envs state-only, phases spread out" --obs-mode state --desync
run "65536 envs 8x8 full frames (the Flat class default grid)" --size 8 --max-steps 100
run "65536 envs 5x5 full frames" --size 5 --max-steps 50
run "65536 envs 64x64 full frames (196 KB each)" --size 64 --steps 100 --warmup 10
