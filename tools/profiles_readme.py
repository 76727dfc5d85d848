#!/usr/bin/env python3
"""Regenerate one round's section of profiles/README.md FROM the files committed under profiles/ — every number in the section is
read out of a file named next to it, nothing is typed in:   python3 tools/profiles_readme.py r03
The section lives between `<!-- <tag>:begin -->` and `<!-- <tag>:end -->` (appended if absent)."""
import csv
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")


def load(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def trace_stats(tag, cfg):
    """kernel -> (calls, average ns) from the summary's kernel-stats block."""
    out = {}
    path = os.path.join(P, f"{tag}_rocprofv3_summary_{cfg}.txt")
    if not os.path.exists(path):
        return out
    for ln in open(path):
        m = re.match(r"(\S+)\s+calls\s+(\d+)\s+total_ns\s+(\d+)\s+avg_ns\s+(\d+)", ln)
        if m:
            out[m.group(1)] = (int(m.group(2)), int(m.group(4)))
    return out


def counters(tag, cfg, kernel_prefix, names):
    """mean per dispatch of the named counters for the kernels whose short name starts with kernel_prefix (averaged over them)."""
    path = os.path.join(P, f"{tag}_rocprofv3_summary_{cfg}.txt")
    acc = {}
    if not os.path.exists(path):
        return acc
    for ln in open(path):
        if ln.startswith(kernel_prefix):
            for n in names:
                m = re.search(rf"\b{n}=([0-9.e+]+)", ln)
                if m:
                    acc.setdefault(n, []).append(float(m.group(1)))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(tag):
    rows, notes = [], []
    b = load(f"{tag}_bench_4k_f32.json")
    files = [
        (f"{tag}_bench_4k_f32.json", "`bench.py` (defaults: 3840x2160 fp32, 5 windows of 50 frames, pan / 1080p / fp16 / general-path extras, cpu_baseline) at the round's final sources"),
        (f"{tag}_bench_4k_f16.json, {tag}_bench_1080p_f32.json, {tag}_bench_8k_f32.json", "`bench.py --storage f16 | --workload 1080p | --workload 8k` (`--no-cpu --no-extra`)"),
        (f"{tag}_bench_4k_f32_pair_launch.json", "`bench.py --fuse`: iterations 0 + 1 as one launch (`svgf_atrous_pair`), same call"),
        (f"{tag}_bench_4k_f32_two_in_flight.json", "`bench.py --frames-in-flight 2`: `svgf_set_frames_in_flight(2)` — iterations 1-4 of a frame beside the next frame's temporal launch (stage times are brackets of overlapping launches)"),
        (f"{tag}_bench_8k_f32_stripdriver_1gpu.json", "`bench.py --gpus 1 --strips --workload 8k`: the N > 1 code path (C++ strip driver, every halo plan, the pan, the one-GPU reference) on ONE GPU"),
        (f"{tag}_rocprofv3_summary_4k_f32.txt, _4k_f16.txt, _1080p_f32.txt", "`tools/prof.sh`: kernel stats (`rocprofv3 --kernel-trace --stats`) + SQ / LDS / FETCH_SIZE / WRITE_SIZE counters (separate `--pmc` passes) of `bench.py --steps 20 --warmup 3 --no-cpu --no-extra [...]`"),
        (f"{tag}_kernel_stats_4k_f32.csv", f"the raw `*_kernel_stats.csv` of the SAME trace run `{tag}_rocprofv3_summary_4k_f32.txt` was condensed from"),
        (f"{tag}_bench_under_rocprofv3_4k_f32.json", "the bench line that traced run printed itself (the library's HIP events under the tracer)"),
        ("hbm_traffic.json", "HBM bytes per launch from this round's PMC passes (`tools/make_traffic.py`), stamped with the hash of the kernel sources; `bench.py` attaches it (`roofline.traffic`, `traffic_source`) only when the sources it runs hash to the same value"),
        (f"{tag}_strip_sim_8k_over_8.txt", "`tools/strip_sim.py`, plans ghost / grouped / per-iteration (round 5: each with edge rows first and with round 4's three launches, and the whole 8K frame on the same GPU in the same call): the middle strip of an 8K/8 partition with loop-back RCCL groups"),
        (f"{tag}_strip_trace_per-iteration_three_launches.txt, _one_launch.txt, _two_launches.txt, {tag}_strip_trace_ghost.txt", "`tools/strip_trace.py`: a strip frame as the device ran it (rocprofv3 kernel trace of `strip_sim.py`): timeline of one frame, mean kernel durations, mean gap between consecutive filter kernels — round 4's schedule, edge rows first in one launch, and with the interior in two launches"),
        (f"{tag}_rccl_selfcopy.txt", "`tools/archive/rccl_selfcopy.py`: one loop-back halo exchange (2 sends + 2 receives, 4 KB - 3.9 MB) on an idle device and beside filter launches, communication stream at normal / highest priority"),
        (f"{tag}_probe_wait_value.txt, {tag}_probe_cu_mask.txt, {tag}_probe_rccl_two_ranks_one_gpu.txt", "`tools/ubench/wait_value.hip`, `tools/ubench/cu_mask.hip`, `tools/archive/probe_rccl_one_gpu.py`: stream memory operations against a running kernel; CU masks; two RCCL ranks on one device (refused)"),
        (f"{tag}_strip_sim_reserved_cus.txt", "`strip_sim.py` with the filter stream kept off one CU pair per XCD (`hipExtStreamCreateWithCUMask`): slower for every plan (the helper left the ABI)"),
        (f"{tag}_parity_envelope.json", "`tools/parity_envelope.py` (CPU): the oracle against its fp32 / fma / fp32fma / fused builds — and, from round 6, `hwulp` (transcendentals within 1 ulp, twelve seeds) — free-running on the parity frames"),
        (f"{tag}_fused_pair_ablations.txt", "iterations 0 + 1 as one launch: A/B against two launches, its knobs, what it is made of; the fp16 half-record experiment"),
        (f"{tag}_small_experiments.txt", "the `tools/abn.sh` / `tools/archive/strip_ab.sh` blocks of the round (interleaved A/B of prebuilt twins on one device)"),
        (f"{tag}_cold_frames.txt", "`tools/cold_frames.py`: per-frame stage times over the cold -> steady transition (fp32, fp16)"),
        (f"{tag}_repeat_suite_prefix.txt", "`tools/archive/repeat_suite_prefix.py 200`: the tests around the spot where two round-2 suite runs hung, 200 times in one process"),
        (f"{tag}_pytest_gpu.txt", "summary line of `pytest tests -q -m gpu` in the same call as the bench lines"),
        (f"{tag}_pytest_gpu_soak.txt", "five more full `pytest tests -q -m gpu` runs in one call at the final sources (the round-2 suite abort: not seen)"),
        (f"{tag}_parity_report.json", "`tests/test_gpu_round2.py::test_parity_report`: max / mean error per stage against the oracle, mask mismatch counts"),
        (f"{tag}_fuzz_parity.txt", "the seeded sweeps: `tests/fuzz_parity.py` on MI355X (every run's summary line, what each finding was), `tests/fuzz_oracle.py` and `tests/strip_geometry_checks.py` on the CPU"),
    ]
    for name, what in files:
        first = name.split(",")[0].strip()
        if os.path.exists(os.path.join(P, first)):
            rows.append(f"| `{name}` | {what} |")

    st = trace_stats(tag, "4k_f32")
    lds = {k: v for k, v in st.items() if k.startswith("atrous_lds_kernel<ST=0")}
    traced = load(f"{tag}_bench_under_rocprofv3_4k_f32.json")
    if lds:
        avg = sum(v[1] for v in lds.values()) / len(lds)
        step_of = lambda name: int(re.search(r"S=(\d+)", name).group(1))      # noqa: E731
        per = ", ".join("S=%d: %.1f" % (step_of(k), v[1] / 1e3) for k, v in sorted(lds.items(), key=lambda kv: step_of(kv[0])))
        alg = 491028480
        notes.append(f"* `{tag}_rocprofv3_summary_4k_f32.txt` / `{tag}_kernel_stats_4k_f32.csv` (one trace run): `atrous_lds_kernel` averages {per} us, "
                     f"mean **{avg / 1e3:.1f} us** over the five launches of a frame; on the contract's {alg / 1e6:.1f} MB of algorithmic bytes per launch that is "
                     f"{alg / avg:.0f} GB/s = **{alg / avg / 8000:.3f} of 8 TB/s**.")
        if traced and traced.get("roofline"):
            notes.append(f"* `{tag}_bench_under_rocprofv3_4k_f32.json`: the same traced run's own line reads {traced['ms_per_step']} ms per frame and "
                         f"{traced['roofline']['avg_launch_ms'] * 1e3:.1f} us per launch between the library's HIP events (frac {traced['roofline']['frac']}).")
    for k in ("temporal_kernel<ST=0>", "moments_young_kernel<ST=0>"):
        if k in st:
            notes.append(f"* `{k}`: {st[k][1] / 1e3:.1f} us average over {st[k][0]} calls in that trace.")
    if b:
        r = b["roofline"]
        notes.append(f"* `{tag}_bench_4k_f32.json` (no tracer, same call, same box): {b['ms_per_step']} ms per frame (windows {b.get('ms_per_step_min')}-{b.get('ms_per_step_max')}), "
                     f"{r['avg_launch_ms'] * 1e3:.1f} us per `{r['kernel']}` launch, `roofline.frac` {r['frac']}, stage sum {b.get('stage_sum_ms')} ms, "
                     f"{b.get('event_overhead_ms_per_step')} ms of event overhead per frame; traffic {r.get('traffic')} B from `{r.get('traffic_source')}`.")
        if r.get("atrous_x5_ms") is not None:
            notes.append(f"* BASELINE.md's line for the iterations alone: a-trous x5 = **{r['atrous_x5_ms']} ms** against the 60 % target of {r['atrous_x5_target_ms']} ms: "
                         f"{'met' if r['atrous_x5_target_met'] else 'MISSED'} (`roofline.atrous_x5_*`).")
        if b.get("uniform_normal_path_share"):
            notes.append(f"* share of the a-trous wave-steps on the uniform-normal tap path, counted on the device (`svgf_path_stats_enable`), headline scene: {b['uniform_normal_path_share']}.")
        cs = (b.get("also") or {}).get("curved_scene")
        if cs:
            notes.append(f"* also `curved_scene` (per-texel normals: terrain, sphere, cylinder): **{cs['ms_per_step']} ms** per frame = {cs['Mpixels/s']} Mpixel/s = **{cs['frac_of_8TBps']}** of the pass "
                         f"roofline (60 % target {'met' if cs['target_met'] else 'MISSED'}), a-trous x5 {cs.get('atrous_x5_ms')} ms, uniform-path share {cs['uniform_normal_path_share'].get('all')}.")
        if "pan" in b:
            notes.append(f"* pan ({b['pan']['mv']}): {b['pan']['ms_per_step']} ms per frame, moments launch {b['pan']['moments_ms'] * 1e3:.1f} us.")
        for k, v in (b.get("also") or {}).items():
            if k == "interactive":
                notes.append(f"* also `interactive` (one denoise per displayed frame, sum of the stage events): **{v['interleaved_ms']} ms** between a {v['producer_ms']} ms memory-bound producer "
                             f"on the same stream; **{v['isolated_ms']} ms** ({v['isolated_ms_min']}-{v['isolated_ms_max']}) with the device idle for 5 ms between frames.")
            elif k == "crowded_frames":
                notes.append(f"* also `crowded_frames` (every 8th column disoccluded in every frame: 12 % of the surface pixels young, some in every wave): **{v['adaptive']['ms_per_step']} ms** per frame "
                             f"(temporal {v['adaptive']['temporal_ms']}, moments {v['adaptive']['moments_ms']}: the streaming kernel by the sample); with `svgf_set_adaptive_moments(0)` "
                             f"{v['young_pixel_launch_only']['ms_per_step']} ms (moments {v['young_pixel_launch_only']['moments_ms']}).")
            elif k == "seven_iterations":
                notes.append(f"* also `seven_iterations` (steps 1..64, all LDS launches): {v['ms_per_step']} ms per frame; a-trous launch by step (ms): {v['atrous_launch_ms_by_step']}.")
            elif "ms_per_step" in v:
                notes.append(f"* also `{k}`: {v['ms_per_step']} ms per frame, {v['Mpixels/s']} Mpixel/s, pass frac {v.get('frac_of_8TBps')}, à-trous launch {v.get('atrous_avg_launch_ms')} ms (frac {v.get('atrous_roofline_frac')}).")
                if v.get("pan"):
                    notes.append(f"* also `{k}.pan` ({v['pan']['mv']}): {v['pan']['ms_per_step']} ms per frame, temporal {v['pan']['temporal_ms']}, moments {v['pan']['moments_ms']} ms.")
                g = v.get("hip_graph")
                if g:
                    notes.append(f"* also `{k}.hip_graph` (four frames captured once and replayed, ms per frame): one frame in flight {g['calls_1_in_flight_ms']} by calls / {g['graph_1_in_flight_ms']} replayed; "
                                 f"two in flight {g['calls_2_in_flight_ms']} / **{g['graph_2_in_flight_ms']}** (without stage events the calls read {v.get('ms_per_step_without_stage_events')}).")
        sec = (r or {}).get("secondary")
        if sec:
            notes.append(f"* `roofline.secondary` (the launch's second bound, from `{sec.get('source')}`): valu_busy **{sec['valu_busy']}**, {sec['insts_valu_per_px']} vector instructions per pixel.")
        if b.get("cold_frames_ms"):
            notes.append(f"* cold frames after a reset (ms, one frame at a time): {b['cold_frames_ms']['after_reset']}.")
        c = b.get("cpu_baseline")
        if c:
            notes.append(f"* cpu_baseline: {c['value']} Mpixel/s on {c['cores']} threads, {c['single_thread_value']} on one; configs[0] (256x256, one iteration, scalar C++ loop): "
                         f"{c['config0_256x256_one_atrous_iteration']['ms']} ms.")
    tr = None
    tpath = os.path.join(P, "hbm_traffic.json")
    if os.path.exists(tpath):
        tr = json.load(open(tpath))
        e = tr.get("3840x2160_f32")
        if e and "atrous_bytes_per_launch" in e:
            notes.append(f"* `hbm_traffic.json` (sources {tr.get('kernel_source_sha16')}): 4K fp32 à-trous launch 2 x {e['atrous_fetch_size_kib']:.0f} KiB fetched (gfx950 correction) + "
                         f"{e['atrous_write_size_kib']:.0f} KiB written = **{e['atrous_bytes_per_launch'] / 1e6:.1f} MB** per launch; temporal launch "
                         f"{e.get('temporal_bytes_per_launch', 0) / 1e6:.1f} MB.")
    for f, label in ((f"{tag}_bench_4k_f16.json", "4K fp16"), (f"{tag}_bench_1080p_f32.json", "1080p fp32"), (f"{tag}_bench_8k_f32.json", "8K fp32, one GPU"),
                     (f"{tag}_bench_4k_f32_pair_launch.json", "4K fp32 with the pair launch"),
                     (f"{tag}_bench_4k_f32_two_in_flight.json", "4K fp32 with two frames in flight")):
        d = load(f)
        if d:
            r = d.get("roofline") or {}
            notes.append(f"* `{f}` ({label}): {d['ms_per_step']} ms per frame, {d['value']} Mpixel/s, pass frac {d['pass_roofline'].get('frac_of_8TBps')}, "
                         f"`{r.get('kernel')}` {r.get('avg_launch_ms')} ms per launch (frac {r.get('frac')}).")
    d = load(f"{tag}_bench_8k_f32_stripdriver_1gpu.json")
    if d:
        hp = "; ".join(f"{k} {v['ms_per_step']} ms" for k, v in d["halo_plans"].items())
        notes.append(f"* `{tag}_bench_8k_f32_stripdriver_1gpu.json`: world size 1 through the C++ strip driver: {hp}; one GPU through `svgf_denoise_frame` {d['one_gpu_ms']} ms; "
                     f"pan (reach {d['pan']['motion_reach']}) {d['pan']['ms_per_step']} ms; headline plan {d['config'].get('halo_plan')}, verified against the one-GPU frame: {d.get('verified')}.")
    sim = os.path.join(P, f"{tag}_strip_sim_8k_over_8.txt")
    if os.path.exists(sim):
        for ln in open(sim):
            m = re.search(r"plan (\S+), .*?: ([0-9.]+) ms/frame \(host enqueue ([0-9.]+) ms\)", ln)
            if m:
                notes.append(f"* `{tag}_strip_sim_8k_over_8.txt`: middle strip of 8K/8, plan {m.group(1)}: {m.group(2)} ms per frame (host enqueue {m.group(3)} ms).")
            m = re.search(r"plan ([^:]+): ([0-9.]+) ms/frame \(rounds: [^;]*; host enqueue ([0-9.]+) ms\)(?:, (\d+) GPUs = x([0-9.]+) of one)?", ln)
            if m:
                notes.append(f"* `{tag}_strip_sim_8k_over_8.txt`: middle strip of 8K/8, plan {m.group(1)}: **{m.group(2)} ms** per frame (host enqueue {m.group(3)} ms)" +
                             (f" = **{m.group(5)}x** for {m.group(4)} GPUs against the whole frame of the same call." if m.group(5) else "."))
            m = re.search(r"whole frame on one GPU: ([0-9.]+) ms/frame", ln)
            if m:
                notes.append(f"* `{tag}_strip_sim_8k_over_8.txt`: the whole 8K frame on the same GPU in the same call: {m.group(1)} ms.")
    section = [f"<!-- {tag}:begin -->", f"## Round {int(tag[1:])}  (generated by `tools/profiles_readme.py {tag}` from the files it names)", "", "| file | what |", "|---|---|"] + rows + \
              ["", "Numbers read out of those files", ""] + notes + [f"<!-- {tag}:end -->", ""]
    path = os.path.join(P, "README.md")
    txt = open(path).read()
    pat = re.compile(rf"<!-- {tag}:begin -->.*?<!-- {tag}:end -->\n?", re.S)
    new = "\n".join(section)
    txt = pat.sub(lambda m: new, txt) if pat.search(txt) else txt.rstrip("\n") + "\n\n" + new
    open(path, "w").write(txt)
    print(new)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
