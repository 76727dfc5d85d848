"""Diagnostic: which rows of a loop-back strip hold young pixels (history < 4) in steady state, and what its moments launch works on."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, torch.distributed as dist
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F, strips
plan = argv[1] if len(argv) > 1 else "ghost"
periodic = "--aperiodic" not in argv
W, H = bench.WORKLOADS["8k"]
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)
params = F.Params(storage="f32", steps=5)
geo = strips.Geometry.make(W, H, 3, 8, 5, plan=plan, moments_radius=3, motion_reach=4)
gb, rads = bench.make_inputs(W, H, "f32", dev, row_begin=geo.y0, row_end=geo.y1)
if periodic:
    own_rows = geo.own[1] - geo.own[0]
    idx = torch.arange(geo.y0, geo.y1, device=dev)
    idx = (geo.own[0] - geo.y0) + torch.remainder(idx - geo.own[0], own_rows)
    take = lambda t: (t.view(torch.int16)[idx].contiguous().view(torch.uint16) if t.dtype == torch.uint16 else t[idx].contiguous())
    gb = F.GBuffer(take(gb.motion), take(gb.normal), take(gb.uv)); rads = [take(r) for r in rads]
gb2 = F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone()); gbs = [gb, gb2]
comm = strips.rccl_comm(1, 0, 0)
drv = strips.NativeStrips(W, H, 8, params, [3], [0], streams=[None], comms=[comm], plan=geo.plan, motion_reach=4, loopback=True)
drv.set_prev_guide(True)
for k in range(30):
    drv.frame([rads[k % len(rads)]], [gbs[k & 1]], [gbs[(k & 1) ^ 1]])
drv.sync()
hist = drv.state_plane(0, F.PLANE_HISTORY, 1 - drv.pingpong(0)).cpu()
nz = (gb.normal.cpu().view(torch.int16)[..., :3] != 0).any(-1)
young = (hist < 4)
print(f"plan {geo.plan}, periodic {periodic}: rows held {geo.y0}..{geo.y1}, own {geo.own}, ext_temporal {geo.ext_temporal}, halo_state {geo.halo_state}")
rows = young.sum(1)
surf_young = (young & nz).sum(1)
for y in range(0, young.shape[0], 1):
    if surf_young[y] > 0:
        print(f"  local row {y} (global {geo.y0 + y}): {int(rows[y])} young, {int(surf_young[y])} of them with a normal")
print("young with normal total", int(surf_young.sum()), "young total", int(young.sum()), "of", young.numel())
drv.close(); dist.destroy_process_group()
