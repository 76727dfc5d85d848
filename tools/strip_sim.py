"""What ONE middle strip of an N-strip frame costs per frame on ONE GPU — and the whole frame on the same GPU in the same call.

The geometry is that of rank `--rank` of `--world` strips of a WxH frame through the C++ strip driver (svgf_strips_frame); the halo
exchanges are real RCCL send/recv groups whose peer is this very rank (a loop-back communicator), so a strip's number holds its kernels
(ghost rows, edge tiles), the host's enqueue cost and RCCL's launches — everything but the xGMI wire time, which the wire model below
adds from the bytes per boundary.  Every (plan, edge-first) configuration and the whole frame are timed in one process on one device (devices of
the pool differ by +-4 %), one configuration after the other: medians of `--rounds` windows.

    python tools/strip_sim.py [--workload 8k] [--world 8] [--rank 3] [--plans ghost,grouped,per-iteration] [--edge-first both|0|1] [--rounds 3]
"""
import argparse
import os
import statistics
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
from svgf_amd import filter as F, strips  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="8k")
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--plans", default="ghost,grouped,per-iteration")
ap.add_argument("--edge-first", default="both", help="svgf_strips_set_edge_first: 1, 0 (round 4's three launches per exchanging iteration) or both")
ap.add_argument("--storage", default="f32")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--in-flight", type=int, default=1, help="svgf_strips_set_frames_in_flight")
ap.add_argument("--no-whole", action="store_true", help="skip the whole frame on this GPU")
ap.add_argument("--warm-ms", type=float, default=400.0, help="untimed frames for at least this long before anything is timed (DESIGN.md 6)")
ap.add_argument("--warm-frames", type=int, default=600, help="... and at least this many")
ap.add_argument("--per-frame", action="store_true", help="print the device and host time of every frame of the first configuration")
ap.add_argument("--aperiodic", action="store_true", help="keep the frame's real content in the halo rows (self-sent state is then inconsistent: see below)")
ap.add_argument("--link-gbps", type=float, default=153.0, help="wire model: one xGMI link between neighbouring GPUs, per direction (MI355X: 7 links x ~153 GB/s)")
ap.add_argument("--rccl-latency-us", type=float, default=-1.0, help="wire model: latency of one send/recv group; < 0 = measured here on the loop-back communicator")
args = ap.parse_args(argv[1:])

W, H = bench.WORKLOADS[args.workload]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)

side = torch.cuda.Stream(device=dev, priority=-1)          # a hardware queue of its own: RCCL's kernels run beside the filter kernels
torch.cuda.set_stream(side)
params = F.Params(storage=args.storage, steps=5)


class Config:
    """One (plan, edge-first) strip driver with inputs of its own layout."""

    def __init__(self, plan, edge_first):
        self.name = f"{plan}{'' if edge_first else ' (three launches)'}"
        self.plan, self.edge_first = plan, edge_first
        self.geo = geo = strips.Geometry.make(W, H, args.rank, args.world, 5, plan=plan, moments_radius=params.moments_radius, motion_reach=4)
        gb, rads = bench.make_inputs(W, H, args.storage, dev, row_begin=geo.y0, row_end=geo.y1)
        if not args.aperiodic:
            # Both neighbours are this rank: what arrives in the halo rows is this strip's OWN boundary rows.  With the frame's real content
            # that is the wrong state for those rows (e.g. the history of sky texels under surface texels: pixels that are "young" again every
            # frame).  So the strip's inputs are made PERIODIC in y with the period of the owned rows: a self-sent row IS the row a real
            # neighbour would send.
            own_rows = geo.own[1] - geo.own[0]
            idx = torch.arange(geo.y0, geo.y1, device=dev)
            idx = (geo.own[0] - geo.y0) + torch.remainder(idx - geo.own[0], own_rows)
            take = lambda t: (t.view(torch.int16)[idx].contiguous().view(torch.uint16) if t.dtype == torch.uint16 else t[idx].contiguous())   # noqa: E731
            gb = F.GBuffer(take(gb.motion), take(gb.normal), take(gb.uv))
            rads = [take(r) for r in rads]
        self.gbs = [gb, F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone())]      # current / previous G-buffer in distinct planes
        self.rads = rads
        # (a communicator of its own; a guess at the cause of what was measured — RCCL pays on the host for every change of the stream a communicator is used on —
        # six drivers taking turns on ONE loop-back communicator measured 0.74 ms of host time per frame instead of 0.36)
        self.comm = strips.rccl_comm(1, 0, 0)
        self.drv = strips.NativeStrips(W, H, args.world, params, [args.rank], [0], streams=[side.cuda_stream], comms=[self.comm], plan=geo.plan, motion_reach=4, loopback=True)
        # (svgf_set_prev_guide stays off, as in bench.py: the ABI default on both sides of every ratio)
        self.drv.set_frames_in_flight(args.in_flight)
        self.drv.set_edge_first(edge_first)
        self.k = 0
        self.ms, self.host = [], []

    def frame(self):
        k = self.k
        self.drv.frame([self.rads[k % len(self.rads)]], [self.gbs[k & 1]], [self.gbs[(k & 1) ^ 1]])
        self.k += 1


class Whole:
    """The whole frame on this GPU through svgf_denoise_frame (the denominator of every scaling figure)."""
    name = "whole frame, one GPU"

    def __init__(self):
        gb, self.rads = bench.make_inputs(W, H, args.storage, dev, nframes=2)
        self.gbs = [gb, F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone())]
        self.d = F.Denoiser(W, H, params, device=0, stream=side.cuda_stream)
        # (svgf_set_prev_guide off: see Config)
        self.k = 0
        self.ms, self.host = [], []

    def frame(self):
        k = self.k
        self.d.Render(self.rads[k & 1], self.gbs[k & 1], self.gbs[(k & 1) ^ 1])
        self.k += 1


def warm(c, ms, frames):
    """Two things must be behind a configuration before it is timed: the part's clock ramp after an idle gap, and the one-off
    20-65 ms stall of a process's first ~4 000 stream operations."""
    w0, n = time.perf_counter(), 0
    while (time.perf_counter() - w0) < ms * 1e-3 or n < frames:
        for _ in range(10):
            c.frame()
        n += 10
        torch.cuda.synchronize()


def timed(c, steps, per_frame=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        c.frame()
        if per_frame:
            ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((ev, time.perf_counter()))
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    c.ms.append(t / steps * 1e3); c.host.append(th / steps * 1e3)
    return marks


edge = {"both": (True, False), "1": (True,), "0": (False,)}[args.edge_first]
todo = [(pl, e) for pl in args.plans.split(",") for e in edge if e or len(strips.PLANS[pl](5)) > 1]
med = lambda v: statistics.median(v)           # noqa: E731
# The configurations run ONE AFTER THE OTHER, each created, primed, timed (`rounds` windows) and destroyed before the next: several strip drivers
# alive at once and taking turns measured up to twice as slow, host-bound (profiles/r05_small_experiments.txt block 1; the cause was not pursued).  The whole frame is timed before the first and after the last of them: the drift of the box across the call.
whole = None if args.no_whole else Whole()
if whole:
    warm(whole, args.warm_ms, args.warm_frames)
    for r in range(args.rounds):
        timed(whole, max(20, args.steps // 4))
configs = []
for pl, e in todo:
    c = Config(pl, e)
    warm(c, args.warm_ms, args.warm_frames)
    for r in range(args.rounds):
        timed(c, args.steps, per_frame=args.per_frame and r == 0 and not configs)
    c.drv.sync()
    c.drv.close()
    F.load_library().svgf_rccl_comm_destroy(c.comm)
    c.drv = None
    del c.gbs, c.rads
    configs.append(c)
whole_ms = None
if whole:
    warm(whole, 100.0, 40)
    for r in range(args.rounds):
        timed(whole, max(20, args.steps // 4))
    whole_ms = med(whole.ms)
    print(f"{W}x{H} {args.storage} whole frame on one GPU: {whole_ms:.4f} ms/frame (windows before and after the strips: {' '.join(f'{v:.4f}' for v in whole.ms)}); an ideal 1/{args.world}: {whole_ms / args.world:.4f} ms")


def rccl_group_latency_us(n=200):
    """GPU time of one RCCL group {ncclSend, ncclRecv} of 4 KiB to and from this rank itself on one stream (librccl called directly): launch +
    protocol latency, no wire.  The host enqueues the groups while the device is still busy with a queue of GEMMs."""
    import ctypes as C
    lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    for f in (lib.ncclSend, lib.ncclRecv):
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        f.restype = C.c_int
    c2 = strips.rccl_comm(1, 0, 0)
    st = torch.cuda.Stream(device=dev)
    a, b = torch.zeros(4096, device=dev, dtype=torch.int8), torch.zeros(4096, device=dev, dtype=torch.int8)
    xx = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    h = C.c_void_p(st.cuda_stream)

    def group():
        lib.ncclGroupStart()
        rc = lib.ncclSend(C.c_void_p(a.data_ptr()), 4096, 0, 0, c2, h) | lib.ncclRecv(C.c_void_p(b.data_ptr()), 4096, 0, 0, c2, h)
        lib.ncclGroupEnd()
        assert rc == 0
    with torch.cuda.stream(st):
        for _ in range(20):
            group()
        torch.cuda.synchronize()
        for _ in range(40):
            xx @ xx
        x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x0.record()
        for _ in range(n):
            group()
        x1.record()
        torch.cuda.synchronize()
    us = x0.elapsed_time(x1) * 1e3 / n
    F.load_library().svgf_rccl_comm_destroy(c2)
    return us


LAT = args.rccl_latency_us if args.rccl_latency_us >= 0 else rccl_group_latency_us()


def wire_model(c, ms_frame):
    """What the xGMI wire would add to the loop-back figure.  Per exchange: bytes per boundary and direction / link bandwidth + 2 x the
    latency of an RCCL group (send side and receive side), against the WINDOW in which the transfer runs beside compute: the state
    exchange is posted after iteration 0 and waited for at the start of the next frame (window: iterations 1..); the filter rows in
    front of iteration group g are posted when the edge rows of the iteration that produces them are done and waited for when group g
    starts (window: that iteration's interior).  Stage shares of the frame: an iteration 0.142 (the one-GPU stage table of bench.py)."""
    geo = c.geo
    msgs = strips.strip_messages(W, H, args.rank, args.world, 5, geo.plan, params.moments_radius, 4, args.storage)
    it_ms = 0.142 * ms_frame
    own = geo.own[1] - geo.own[0]
    total, lines = 0.0, []
    for ex in sorted({m["exchange"] for m in msgs}):
        nbytes = sum(m["bytes"] for m in msgs if m["exchange"] == ex and m["send"] and m["peer"] == args.rank + 1)
        if ex == 0:
            name, window = "state (colour, moments, history)", (geo.steps - 1) * it_ms
        else:
            h = geo.halo_group[ex]
            name, window = f"filter rows in front of iterations {geo.groups[ex]}", it_ms * max(0, own - 2 * h) / own
        wire = nbytes / (args.link_gbps * 1e9) * 1e3 + 2 * LAT * 1e-3
        exposed = max(0.0, wire - window)
        total += exposed
        lines.append(f"    {name}: {nbytes / 1e6:.2f} MB per boundary and direction, wire {wire * 1e3:.1f} us, window {window * 1e3:.0f} us, exposed {exposed * 1e3:.1f} us")
    print(f"  wire model ({args.link_gbps:.0f} GB/s per direction, RCCL group latency {LAT:.1f} us {'(measured on the loop-back communicator)' if args.rccl_latency_us < 0 else '(given)'}):")
    print("\n".join(lines))
    return total


for c in configs:
    ms = med(c.ms)
    own = c.geo.own[1] - c.geo.own[0]
    scale = f", {args.world} GPUs = x{whole_ms / ms:.2f} of one" if whole_ms else ""
    print(f"{W}x{H} strip {args.rank}/{args.world} ({own} rows, held {c.geo.y1 - c.geo.y0}), plan {c.name}{', two frames in flight' if args.in_flight == 2 else ''}: "
          f"{ms:.4f} ms/frame (rounds: {' '.join(f'{v:.4f}' for v in c.ms)}; host enqueue {med(c.host):.4f} ms){scale}")
    extra = wire_model(c, ms)
    print(f"  with wire: {ms + extra:.4f} ms/frame" + (f" = x{whole_ms / (ms + extra):.2f}" if whole_ms else ""))
dist.destroy_process_group()
