"""Diagnostic: what ONE middle strip of an N-strip frame costs per frame on ONE GPU (no 8-GPU node needed).

The geometry is that of rank `--rank` of `--world` strips of a WxH frame; the halo exchanges are real RCCL
send/recv batches, but both neighbours are this very rank (self send/recv), so the numbers hold the kernel
time of a strip (with its halo rows and interior/boundary splits), the host-side dispatch cost and the
RCCL launch cost — everything except the xGMI transfer time itself.  `--comm none` drops the exchanges.

    python tools/strip_sim.py [--workload 8k] [--world 8] [--rank 3] [--plan grouped] [--comm self|none]
"""
import argparse
import os
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
ap.add_argument("--plan", default="grouped")
ap.add_argument("--comm", default="self")
ap.add_argument("--storage", default="f32")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--stream", default="own", help="own = a non-blocking side stream; own-hi = the same at high priority (its own hardware queue); null = the legacy default stream")
ap.add_argument("--post", default="side", help="side = post RCCL batches from a stream of their own; inline = from the compute stream")
ap.add_argument("--prefill", type=int, default=0, help="N 8192^3 bf16 GEMMs queued before the timed frames (GPU-event timing)")
ap.add_argument("--warm-ms", type=float, default=400.0, help="untimed frames for at least this long before the timed ones")
ap.add_argument("--warm-frames", type=int, default=600, help="... and at least this many")
ap.add_argument("--per-frame", action="store_true", help="print the device and host time of every frame")
ap.add_argument("--driver", default="python", help="python = StripRunner; native = svgf_strip_* C driver if present")
ap.add_argument("--in-flight", type=int, default=1, help="native driver: svgf_strips_set_frames_in_flight")
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


class SelfComm(strips.DistComm):
    """Every peer is this rank."""
    def __init__(self):
        super().__init__(device=dev if args.post == "side" else None)

    def start(self, sends, recvs):
        if args.comm == "none":
            return []
        return super().start([(t, 0) for t, _ in sends], [(t, 0) for t, _ in recvs])


side = torch.cuda.Stream(device=dev, priority=-1 if args.stream == "own-hi" else 0) if args.stream.startswith("own") else torch.cuda.default_stream(dev)
torch.cuda.set_stream(side)
params = F.Params(storage=args.storage, steps=5)
geo = strips.Geometry.make(W, H, args.rank, args.world, 5, plan=args.plan, moments_radius=params.moments_radius, motion_reach=4)
gb, rads = bench.make_inputs(W, H, args.storage, dev, row_begin=geo.y0, row_end=geo.y1)
if args.comm == "self" and not args.aperiodic:
    # Both neighbours are this rank: what arrives in the halo rows is this strip's OWN boundary rows.  With the frame's real content
    # that is the wrong state for those rows (e.g. the history of sky texels under surface texels: pixels that are "young" again every
    # frame — the moments launch of the strip then costs 33 us instead of ~5).  So the strip's inputs are made PERIODIC in y with the
    # period of the owned rows: row y of the halo holds what row y -+ own of the strip holds, and a self-sent row IS the row a real
    # neighbour would send.
    own_rows = geo.own[1] - geo.own[0]
    idx = torch.arange(geo.y0, geo.y1, device=dev)
    idx = (geo.own[0] - geo.y0) + torch.remainder(idx - geo.own[0], own_rows)
    take = lambda t: (t.view(torch.int16)[idx].contiguous().view(torch.uint16) if t.dtype == torch.uint16 else t[idx].contiguous())   # noqa: E731 (no uint16 gather in torch)
    gb = F.GBuffer(take(gb.motion), take(gb.normal), take(gb.uv))
    rads = [take(r) for r in rads]
gb2 = F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone())      # current / previous G-buffer in distinct planes
gbs = [gb, gb2]
if args.driver == "native":
    # the C++ strip driver (svgf_strips_frame): loop-back communicator, every peer is this rank
    comm = strips.rccl_comm(1, 0, 0)
    drv = strips.NativeStrips(W, H, args.world, params, [args.rank], [0], streams=[side.cuda_stream], comms=[comm], plan=geo.plan, motion_reach=4, loopback=True)
    drv.set_prev_guide(True)          # the previous G-buffer below IS last frame's current one, untouched
    drv.set_frames_in_flight(args.in_flight)
    frame = lambda k: drv.frame([rads[k % len(rads)]], [gbs[k & 1]], [gbs[(k & 1) ^ 1]])      # noqa: E731
else:
    stages = strips.HipStages(geo, params, dev)
    runner = strips.StripRunner(geo, stages, SelfComm(), storage=args.storage, device=dev)
    frame = lambda k: runner.frame(rads[k % len(rads)], gbs[k & 1], gbs[(k & 1) ^ 1])          # noqa: E731
# --warm-ms (and >= --warm-frames) of untimed frames.  Two things must be behind us before anything is timed (--per-frame shows both):
# after the idle gaps of the set-up the part needs tens of milliseconds at load to be back at its clocks (tools/idle_gap.py: the next
# 10-40 frames run 5-20 % slower), and a process sees ONE stall of 20-65 ms when it has enqueued its first ~4 000 stream operations
# (launches, event records / waits, RCCL groups: ~300 frames of the ghost plan) - never again in the 1 500 frames after it.  With
# 12 untimed and 30 timed frames, as this tool used to run, the whole timed region sat in the ramp.
w0, k = time.perf_counter(), 0
while (time.perf_counter() - w0) < args.warm_ms * 1e-3 or k < args.warm_frames:
    for _ in range(10):
        frame(k)
        k += 1
    torch.cuda.synchronize()
if args.prefill:
    # keep the GPU busy for a while so that the host gets far ahead: the frame time seen by GPU events is then free of any
    # host-side launch latency (is a gap in the kernel trace the host's or the GPU's?)
    xx = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(args.prefill):
        xx @ xx
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
marks = []
for k in range(args.steps):
    frame(k)
    if args.per_frame:
        ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((ev, time.perf_counter()))
e1.record()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 if args.prefill else time.perf_counter() - t0
if args.per_frame:       # device time between the ends of consecutive frames, and when the host got there
    hp = t0
    last = e0
    dev_ms, host_ms = [], []
    for ev, th in marks:
        dev_ms.append(last.elapsed_time(ev)); host_ms.append((th - hp) * 1e3); last, hp = ev, th
    print("stalls (frame: device ms / host ms): " + ", ".join(f"{i}: {d:.1f}/{h:.1f}" for i, (d, h) in enumerate(zip(dev_ms, host_ms)) if d > 2.0 or h > 2.0))
    for i in range(0, len(dev_ms) if args.steps <= 200 else 0, 10):
        print(f"frames {i:4d}..: device " + " ".join(f"{v:.2f}" for v in dev_ms[i:i + 10]) + "   host " + " ".join(f"{v:.2f}" for v in host_ms[i:i + 10]))
own = geo.own[1] - geo.own[0]
print(f"{W}x{H} strip {args.rank}/{args.world} ({own} rows, held {geo.y1 - geo.y0}), plan {geo.plan}, comm {args.comm}/{args.post}, stream {args.stream}, driver {args.driver}{', two frames in flight' if args.in_flight == 2 else ''}: "
      f"{t / args.steps * 1e3:.4f} ms/frame (host enqueue {t_host / args.steps * 1e3:.4f} ms) -> "
      f"{W * own / (t / args.steps) / 1e6:.0f} Mpx/s per GPU, x{args.world} = {W * own * args.world / (t / args.steps) / 1e6:.0f} Mpx/s")


def rccl_group_latency_us(n=200):
    """GPU time of one RCCL group {ncclSend, ncclRecv} of 4 KiB to and from this rank itself on one stream (a loop-back communicator of the
    library's own, librccl called directly): launch + protocol latency, no wire.  The host enqueues the groups while the device is
    still busy with a queue of GEMMs, so that the time between the two events is the device's."""
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


def wire_model(ms_frame):
    """What the xGMI wire would add to the loop-back figure.  Per exchange: bytes per boundary and direction / link bandwidth + 2 x the
    latency of an RCCL group (send side and receive side), against the WINDOW in which the transfer runs beside compute: the state
    exchange is posted after iteration 0 and waited for at the start of the next frame (window: iterations 1..); the filter rows in
    front of iteration group g are posted behind the two edge launches of the iteration that produces them and waited for when group g
    starts (window: that iteration's interior).  Stage shares of the frame: temporal + moments 0.29, an iteration 0.142 (the one-GPU
    stage table of bench.py).  exposed = max(0, wire - window)."""
    cb, mb = (16, 8) if args.storage == "f32" else (8, 4)
    lat = args.rccl_latency_us if args.rccl_latency_us >= 0 else rccl_group_latency_us()
    it_ms = 0.142 * ms_frame
    own = geo.own[1] - geo.own[0]
    colour_held = geo.ext_atrous[0] if geo.steps else geo.ext_temporal
    ex = []
    state_bytes = W * ((geo.halo_state - colour_held) * cb + (geo.halo_state - geo.ext_temporal) * (mb + 1))
    ex.append(("state (colour, moments, history)", state_bytes, (geo.steps - 1) * it_ms))
    for gi in range(1, len(geo.groups)):
        h = geo.halo_group[gi]
        ex.append((f"filter rows in front of iterations {geo.groups[gi]}", W * h * cb, it_ms * max(0, own - 2 * h) / own))
    total = 0.0
    lines = []
    for name, nbytes, window in ex:
        wire = nbytes / (args.link_gbps * 1e9) * 1e3 + 2 * lat * 1e-3
        exposed = max(0.0, wire - window)
        total += exposed
        lines.append(f"    {name}: {nbytes / 1e6:.2f} MB per boundary and direction, wire {wire * 1e3:.1f} us, window {window * 1e3:.0f} us, exposed {exposed * 1e3:.1f} us")
    print(f"  wire model ({args.link_gbps:.0f} GB/s per direction, RCCL group latency {lat:.1f} us {'(measured on the loop-back communicator)' if args.rccl_latency_us < 0 else '(given)'}):")
    print("\n".join(lines))
    print(f"  with wire: {ms_frame + total:.4f} ms/frame (without: {ms_frame:.4f}) -> x{args.world} = {W * own * args.world / ((ms_frame + total) * 1e-3) / 1e6:.0f} Mpx/s")


wire_model(t / args.steps * 1e3)
if args.driver == "native":
    drv.sync()
    drv.close()
dist.destroy_process_group()
