"""profiles/hbm_traffic.json from the PMC passes of tools/prof.sh: HBM bytes per launch of the a-trous and temporal kernels,
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 — FETCH_SIZE is doubled per the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md
(HBM section: 128-B requests tallied at 64 B) — stamped with the hash of the kernel sources it was measured on; bench.py uses
the file only when that hash matches the sources it runs.
    python tools/make_traffic.py gpurun_out/prof_<tag> <WxH_storage> [round]"""
import collections
import csv
import glob
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402


def counters(d, sub, name):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = "atrous" if "atrous_lds_kernel" in r["Kernel_Name"] else "temporal" if "temporal_kernel" in r["Kernel_Name"] else None
                if k:
                    agg[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


SIMDS, XCDS = 1024, 8          # MI355X: 256 CUs x 4 SIMDs in 8 XCDs (GRBM_GUI_ACTIVE is summed over the XCDs)


def main(d, key, rnd):
    fetch, write = counters(d, "pmc_fetch", "FETCH_SIZE"), counters(d, "pmc_write", "WRITE_SIZE")
    insts, active, gui = counters(d, "pmc_sq", "SQ_INSTS_VALU"), counters(d, "pmc_sq", "SQ_ACTIVE_INST_VALU"), counters(d, "pmc_write", "GRBM_GUI_ACTIVE")
    W, H = (int(v) for v in key.split("_")[0].split("x"))
    path = os.path.join(R, "profiles", "hbm_traffic.json")
    rec = {}
    if os.path.exists(path):
        rec = json.load(open(path))
    sha = bench.kernel_source_sha()
    if rec.get("kernel_source_sha16") != sha:
        rec = {}
    rec["_comment_valu"] = ("*_valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the launch's shader cycles in which a SIMD's "
                            "vector ALU is executing; *_insts_valu_per_px = SQ_INSTS_VALU / (pixels / 64): vector instructions per pixel of the launch.")
    rec["_comment"] = ("HBM bytes per launch from rocprofv3 PMC passes (tools/prof.sh): (2 x FETCH_SIZE + WRITE_SIZE) x 1024, FETCH_SIZE doubled per the gfx950 "
                       "correction in /opt/skills/guides/MI355X_MICROARCH.md (HBM section). Mean over the launches of the profiled run (a-trous: the 5 launches of a frame).")
    rec["kernel_source_sha16"] = sha
    e = {"round": rnd}
    for k in ("atrous", "temporal"):
        if k in fetch and k in write:
            e[f"{k}_bytes_per_launch"] = int((2 * fetch[k] + write[k]) * 1024)
            e[f"{k}_fetch_size_kib"] = round(fetch[k], 1)
            e[f"{k}_write_size_kib"] = round(write[k], 1)
        if k in insts and k in active and k in gui and gui[k] > 0:
            # the second bound of the launch: how busy the vector ALUs are.  SQ_ACTIVE_INST_VALU counts quad-cycles (x4 = cycles) summed over
            # the SIMDs; GRBM_GUI_ACTIVE / XCDS = the launch's duration in shader cycles
            e[f"{k}_valu_busy"] = round(active[k] * 4 / (SIMDS * gui[k] / XCDS), 4)
            e[f"{k}_insts_valu_per_px"] = round(insts[k] / (W * H / 64.0), 1)
            e[f"{k}_sq_insts_valu"] = insts[k]
            e[f"{k}_sq_active_inst_valu"] = active[k]
            e[f"{k}_grbm_gui_active"] = gui[k]
    rec[key] = e
    json.dump(rec, open(path, "w"), indent=2)
    print(json.dumps(rec, indent=2))


if __name__ == "__main__":
    main(argv[1], argv[2], int(argv[3]) if len(argv) > 3 else 2)
