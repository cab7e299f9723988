"""One-rank RCCL sanity of the calls bench.py makes at N > 1 (init with device_id, barrier, MAX all-reduce of a double):
python tools/nccl_smoke.py"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device='cuda:0')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print('nccl ok', t.item(), flush=True)
dist.destroy_process_group()
