"""How long does ONE halo exchange take on the device — alone, and beside a filter launch?  (tools/strip_trace.py shows the exchange's RCCL kernel
running as long as the a-trous interior beside it: is that the transfer, or the kernel waiting for the chip?)
A loop-back group {2 x ncclSend, 2 x ncclRecv} of `bytes` each (the two boundaries of a middle strip), through librccl directly, timed with events
on its stream: (a) on an idle device, (b) while 8K denoiser frames run on a high-priority stream.  Communication stream at normal and at high priority.
    python tools/archive/rccl_selfcopy.py"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
from svgf_amd import filter as F, strips  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)
lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
for f in (lib.ncclSend, lib.ncclRecv):
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
comm = strips.rccl_comm(1, 0, 0)
W, H = 7680, 1080
gb, rads = bench.make_inputs(W, H, "f32", dev, nframes=2)
gbs = [gb, F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone())]
hi = torch.cuda.Stream(device=dev, priority=-1)
d = F.Denoiser(W, H, F.Params(storage="f32", steps=5), device=0, stream=hi.cuda_stream)
for k in range(200):
    d.Render(rads[k & 1], gbs[k & 1], gbs[(k & 1) ^ 1])
torch.cuda.synchronize()
for prio in (0, -1):
    st = torch.cuda.Stream(device=dev, priority=prio)
    h = C.c_void_p(st.cuda_stream)
    for nbytes in (4096, 491520, 983040, 1966080, 3932160):
        a = [torch.zeros(nbytes, device=dev, dtype=torch.int8) for _ in range(4)]

        def group():
            lib.ncclGroupStart()
            rc = 0
            for i in (0, 1):
                rc |= lib.ncclSend(C.c_void_p(a[i].data_ptr()), nbytes, 0, 0, comm, h) | lib.ncclRecv(C.c_void_p(a[2 + i].data_ptr()), nbytes, 0, 0, comm, h)
            lib.ncclGroupEnd()
            assert rc == 0
        res = {}
        for busy in (False, True):
            with torch.cuda.stream(st):
                for _ in range(10):
                    group()
            torch.cuda.synchronize()
            ev = []
            k = 0
            for _ in range(40):
                if busy:
                    for _ in range(2):
                        d.Render(rads[k & 1], gbs[k & 1], gbs[(k & 1) ^ 1]); k += 1
                with torch.cuda.stream(st):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); group(); e1.record()
                ev.append((e0, e1))
                if not busy:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            t = sorted(x.elapsed_time(y) * 1e3 for x, y in ev)
            res[busy] = (t[len(t) // 2], t[0], t[-1])
        print(f"comm stream priority {prio:2d}, 2 x {nbytes / 1e6:5.2f} MB each way: idle device {res[False][0]:6.1f} us (min {res[False][1]:.1f}, max {res[False][2]:.1f});  "
              f"beside denoiser launches {res[True][0]:6.1f} us (min {res[True][1]:.1f}, max {res[True][2]:.1f})")
d.close()
F.load_library().svgf_rccl_comm_destroy(comm)
dist.destroy_process_group()
