"""RCCL smoke test on ONE GPU (world_size 1): the collectives of distributed.py applied to
library-owned buffers (hipMalloc'ed by libm3p2i_hip.so, wrapped as torch tensors through
__cuda_array_interface__) -- checks that RCCL accepts memory that did not come from torch's
allocator.  Multi-GPU runs are the driver's; this is the part of that path a 1-GPU box can exercise."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p2i_aip_amd.engine import HipEngine, make_config
from m3p2i_aip_amd import _lib as L
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
eng = HipEngine(make_config(K=2000, T=30, nu=2, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3]))
red = eng.buffer(L.BUF_REDUCE); red.fill_(1.5)
dist.all_reduce(red, op=dist.ReduceOp.SUM)
J, W = eng.buffer(L.BUF_TRAJ_COST), eng.buffer(L.BUF_WEIGHTS)
J.copy_(torch.arange(2000, device="cuda", dtype=torch.float32))
dist.all_gather_into_tensor(W, J)
torch.cuda.synchronize()
assert float(red[0]) == 1.5 and torch.equal(W, J)
dist.barrier(); dist.destroy_process_group()
print("rccl smoke ok")
