"""GPU box: the collectives bench.py makes for N > 1, in its order -- a gloo default group (the control plane), an RCCL group beside it
(`new_group(backend='nccl')`), one probing all-reduce on it, the MIN-reduced "ok" flag over gloo, then on the RCCL group the timing barrier,
the MAX-reduce of a float64 and the all_gather of the per-rank times -- exercised with a world of ONE rank, the most a 1-GPU box allows;
the code path is RCCL's all the same (the failure branch is exercised by `bench.py --gpus 2 --rehearse-on-one-gpu --dist-backend nccl`:
two ranks on one GPU, which RCCL refuses)."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_craftingworld_amd.sharding import gather_over_ranks, max_over_ranks  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=120))
group = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=60))
probe = torch.ones(1, device=dev)
dist.all_reduce(probe, group=group)
torch.cuda.synchronize(dev)
flag = torch.tensor([int(probe.item() == dist.get_world_size())], dtype=torch.int32)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
assert int(flag.item()) == 1
dist.barrier(group=group)
torch.cuda.synchronize(dev)
# a world of one short-circuits the helpers: call the collectives they wrap directly as well
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
out = [torch.zeros_like(t)]
dist.all_gather(out, t, group=group)
torch.cuda.synchronize(dev)
assert t.item() == 1.25 and out[0].item() == 1.25
assert max_over_ranks(1.25, device=dev, group=group) == 1.25 and gather_over_ranks(1.25, device=dev, group=group) == [1.25]
print('rccl ok: default group', dist.get_backend(), '| timing group', dist.get_backend(group), '| world', dist.get_world_size())
dist.barrier()
dist.destroy_process_group()
