"""Diagnostic: svgf_denoise_frame enqueued call by call against the same frames replayed from a hipGraph (two frames per graph: the context
ping-pongs, include/svgf.h "Stream capture"), per frame size.  Three figures per size: back to back (the device never idle: what bench.py
times), one frame at a time (host waits for every frame, as an interactive host does between its own passes), and the host's enqueue time.
    python3 tools/archive/graph_replay.py [f32|f16] [in_flight]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
sys.argv, argv = ["bench.py"], sys.argv
import bench
from svgf_amd import filter as F
storage = argv[1] if len(argv) > 1 else "f32"
in_flight = int(argv[2]) if len(argv) > 2 else 1
dev = torch.device("cuda:0")


def run(W, H, pairs):
    gb, rads = bench.make_inputs(W, H, storage, dev, nframes=2)
    gbs = [gb, F.GBuffer(gb.motion.clone(), gb.normal.clone(), gb.uv.clone())]
    s = torch.cuda.Stream()
    d = F.Denoiser(W, H, F.Params(storage=storage, steps=5), stream=s.cuda_stream)
    d.set_prev_guide(True)
    d.set_frames_in_flight(in_flight)

    def two():
        d.Render(rads[0], gbs[0], gbs[1])
        d.Render(rads[1], gbs[1], gbs[0])
        if in_flight == 2:
            d.flush()

    with torch.cuda.stream(s):
        for _ in range(300):
            two()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        two()

    def timed(fn, n, each_sync):
        with torch.cuda.stream(s):
            for _ in range(50):
                fn()
            s.synchronize()
            best = []
            for _ in range(5):
                t0 = time.perf_counter()
                host = 0.0
                for _ in range(n):
                    h0 = time.perf_counter()
                    fn()
                    host += time.perf_counter() - h0
                    if each_sync:
                        s.synchronize()
                s.synchronize()
                best.append(((time.perf_counter() - t0) / (2 * n) * 1e3, host / (2 * n) * 1e3))
        best.sort()
        return best[len(best) // 2]

    rows = []
    for each_sync in (False, True):
        a = timed(two, pairs, each_sync)
        b = timed(g.replay, pairs, each_sync)
        rows.append((a, b))
    (a0, b0), (a1, b1) = rows
    print(f"{W}x{H} {storage} in_flight={in_flight}: back to back  calls {a0[0]:.4f} ms/frame (host {a0[1]:.4f})  graph {b0[0]:.4f} (host {b0[1]:.4f})   "
          f"| one pair at a time  calls {a1[0]:.4f}  graph {b1[0]:.4f}", flush=True)
    del g
    d.close()


for (W, H, pairs) in ((640, 360, 400), (1280, 720, 300), (1920, 1080, 200), (3840, 2160, 100)):
    run(W, H, pairs)
