"""Can two RCCL ranks share ONE device?  (If they can, the strip driver's multi-process path runs on a one-GPU box with the product transport.)
Run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/archive/probe_rccl_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    t = torch.full((1024,), float(rank), device="cuda:0")
    r = torch.empty_like(t)
    ops = [dist.P2POp(dist.isend, t, 1 - rank), dist.P2POp(dist.irecv, r, 1 - rank)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    print(f"rank {rank}: received {r[0].item()} from rank {1 - rank}: two RCCL ranks on one device WORK", flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: two RCCL ranks on one device FAIL: {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(0)
