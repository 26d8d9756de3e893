"""GPU box: the RCCL calls bench.py makes for N > 1 (init with device_id, barrier, all_reduce MAX f64 / SUM i64), exercised
with a world of ONE rank -- the most a 1-GPU box allows; the collectives' code path is RCCL's all the same."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
u = torch.tensor([3, 4, 5, 6], dtype=torch.int64, device=dev)
dist.all_reduce(u, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert t.item() == 1.25 and u.tolist() == [3, 4, 5, 6]
print('rccl ok: backend', dist.get_backend(), 'world', dist.get_world_size())
dist.barrier()
dist.destroy_process_group()
