import faulthandler
import os
import subprocess
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# A GPU test that stops making progress must say where (twice in ~60 full runs of round 2 the suite sat in one test until the
# 600 s timeout killed the process, with nothing to read afterwards): pytest's faulthandler plugin dumps every thread's Python stack
# after `faulthandler_timeout` seconds (pytest.ini), and the watchdog below records what the GPU side looks like at that moment.
HANG_SECONDS = 150
faulthandler.enable()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


def _gpu_snapshot(nodeid):
    """What a hung GPU test leaves behind: processes on the device, clocks / power, the kernel log tail (when readable)."""
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        name = "hang_" + "".join(c if c.isalnum() else "_" for c in nodeid)[-120:] + ".txt"
        with open(os.path.join(out_dir, name), "w") as f:
            f.write(f"{nodeid}: no progress for {HANG_SECONDS} s at {time.ctime()}\n")
            for cmd in (["rocm-smi", "--showpids"], ["rocm-smi", "--showuse", "--showpower", "--showclocks"], ["dmesg"], ["journalctl", "-k", "-n", "80", "--no-pager"]):
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
                    f.write(f"\n$ {' '.join(cmd)}  (rc {r.returncode})\n" + "\n".join((r.stdout + r.stderr).splitlines()[-80:]) + "\n")
                except Exception as e:  # noqa: BLE001
                    f.write(f"\n$ {' '.join(cmd)}: {type(e).__name__}: {e}\n")
            f.write("\nPython stacks of all threads:\n")
            f.flush()
            faulthandler.dump_traceback(file=f, all_threads=True)
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(autouse=True)
def _hang_watchdog(request):
    """GPU tests only: a timer that fires long before pytest-timeout ends the run."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    t = threading.Timer(HANG_SECONDS, _gpu_snapshot, args=(request.node.nodeid,))
    t.daemon = True
    t.start()
    try:
        yield
    finally:
        t.cancel()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
