"""GPU box only: how fast do candidate torch READERS of the observation array run?  (bench.py --consumer: the policy stand-in between two steps.)
The byte-wise sum the round-4 verdict prescribes reads 1.39 GB in 3.5 ms (0.4 TB/s: torch's uint8 reduction); what reads it at HBM speed?
    python tools/microbench/consumer_speed.py [envs] [size]"""
import sys
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 21
FB = 48 * S * S
obs = torch.randint(0, 255, (N, 4 * S, 4 * S, 3), dtype=torch.uint8, device='cuda')
cands = {
    'u8 sum(1, int32)': lambda: obs.view(N, -1).sum(1, dtype=torch.int32),
    'i32 view sum(1) (wraps)': lambda: obs.view(N, -1).view(torch.int32).sum(1, dtype=torch.int32),
    'i32 view sum(1, int64)': lambda: obs.view(N, -1).view(torch.int32).sum(1),
    'i64 view sum(1)': lambda: obs.view(N, -1).view(torch.int64).sum(1),
    'i32 view amax(1)': lambda: obs.view(N, -1).view(torch.int32).amax(1),
    'i64 view amax(1)': lambda: obs.view(N, -1).view(torch.int64).amax(1),
    'f32 view? (i32->float sum)': lambda: obs.view(N, -1).view(torch.int32).to(torch.float32).sum(1),
    'i16 view sum(1, int32)': lambda: obs.view(N, -1).view(torch.int16).sum(1, dtype=torch.int32),
    'bf16 matvec over u8->bf16': lambda: obs.view(N, -1).to(torch.bfloat16) @ torch.ones(FB, 1, dtype=torch.bfloat16, device='cuda'),
}
for name, fn in cands.items():
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('%-34s %8.3f ms  %7.1f GB/s' % (name, ms, N * FB / ms / 1e6), flush=True)
    except Exception as exc:  # noqa: BLE001
        print('%-34s failed: %s' % (name, str(exc).splitlines()[0]), flush=True)
