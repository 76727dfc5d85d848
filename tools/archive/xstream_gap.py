"""Diagnostic: does a cross-stream event dependency (what a collective post inserts between two kernels of the compute
stream) cost GPU time on the compute stream?  GPU-event timing with the host far ahead."""
import sys, torch
dev = torch.device("cuda:0")
hi = torch.cuda.Stream(device=dev, priority=-1)
lo = torch.cuda.Stream(device=dev, priority=0)
x = torch.randn(64 << 20, device=dev)          # 256 MB: one pass ~ 100 us
y = torch.zeros(1024, device=dev)
pre = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)

def run(mode, n=200):
    torch.cuda.synchronize()
    with torch.cuda.stream(hi):
        for _ in range(30):
            pre @ pre
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            x.mul_(1.0001)
            if mode != "none":
                ev = torch.cuda.Event()
                ev.record(hi)
                lo.wait_event(ev)
                if mode in ("kernel", "kernel+back"):
                    with torch.cuda.stream(lo):
                        y.add_(1.0)
                if mode == "kernel+back":
                    ev2 = torch.cuda.Event()
                    ev2.record(lo)
                    hi.wait_event(ev2)
            x.mul_(0.9999)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

run("none")
for rnd in range(3):
    print("  ".join(f"{mode} {run(mode):7.1f}" for mode in ("none", "event", "kernel", "kernel+back")), "  us per (kernel, [dependency], kernel) pair")
