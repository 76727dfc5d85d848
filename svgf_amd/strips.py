"""Row strips across the GPUs of one node: Python's share of it.

The product path is C++: `svgf_strips_*` in svgf_amd/csrc/svgf_strip.hip runs the stage sequence of every strip, posts the halo
exchanges as RCCL send/recv groups and ties them to the filter stream.  This module holds
  * the bindings of that driver (`NativeStrips`, `strips_plan`, `strip_messages`) and the rank bootstrap (`rccl_comm`: the 128-byte
    unique id travels over torch.distributed — the only thing Python does for an exchange);
  * `bench_strips`, bench.py's N > 1 leg;
  * a Python restatement of the driver's SCHEDULE (`Geometry`, `StripRunner`, `LocalComm`, `DistComm`) that runs on any stage
    backend.  It is NOT the product and no GPU path uses it: tests/test_strips_cpu.py drives it with oracle stages over gloo (world
    size 2) and in lock step, which checks the halo plan's arithmetic — what must travel when for the strips to equal the whole
    frame — independently of the C++ code; `svgf_strips_plan` / `svgf_strips_messages` are checked against it (tests/test_abi.py).

Halo plan.  À-trous iteration i reaches 2*2^i rows.  A plan groups the iterations; a group's input halo (the sum of its iterations'
reaches) is exchanged once, before the group, and inside the group iteration i is computed on `ext_i` extra rows each side
(redundantly with the neighbour, bit-identically) so that the next iteration of the group finds its halo locally:
    "per-iteration"  [[0],[1],[2],[3],[4]]   one exchange before every iteration  (the literal north-star scheme)
    "grouped"        [[0,1,2],[3,4]]         2 à-trous exchanges per frame, <3 % redundant work
    "ghost"          [[0,1,2,3,4]]           no exchange between iterations, 62 ghost rows each side
The temporal and moments stages are computed redundantly on the first group's halo, so a frame needs ONE more exchange: its state
(colour feedback, moments, history) on the rows the next frame's reprojection can reach.  That state is final once iteration 0 has
written the feedback colour, so it is posted right there and waited for at the start of the NEXT frame.
"""
from __future__ import annotations

from dataclasses import dataclass, field

PLANS = {
    "per-iteration": lambda n: [[i] for i in range(n)],
    "grouped": lambda n: ([list(range(0, min(3, n)))] if n else []) + ([list(range(3, n))] if n > 3 else []),
    "ghost": lambda n: [list(range(n))] if n else [],
}
DEFAULT_PLAN = "grouped"
AUTO_ORDER = ("grouped", "per-iteration", "ghost")      # svgf_strips_plan's order (svgf_strip.hip): the fewest exchanges between iterations, but at least one


def partition(H: int, world: int):
    """Owned row range of every rank: contiguous, as even as possible."""
    return [(H * r // world, H * (r + 1) // world) for r in range(world)]


@dataclass
class Geometry:
    """Everything a rank needs to know about its strip under a plan."""
    W: int
    H: int
    rank: int
    world: int
    steps: int
    moments_radius: int
    motion_reach: int
    groups: list
    own: tuple = (0, 0)
    ext_atrous: list = field(default_factory=list)   # extra rows each side iteration i is computed on
    halo_group: list = field(default_factory=list)   # input halo of group g
    ext_moments: int = 0
    ext_temporal: int = 0
    halo_state: int = 0                               # previous-frame state rows needed beyond the owned rows
    halo_max: int = 0
    y0: int = 0
    y1: int = 0
    plan: object = None

    @staticmethod
    def make(W, H, rank, world, steps, plan=DEFAULT_PLAN, moments_radius=3, motion_reach=4):
        if plan == "auto":
            # the plan with the fewest exchanges between iterations — but at least one (BASELINE.json configs[3]) — whose halo fits the strips
            for cand in AUTO_ORDER:
                try:
                    g = Geometry.make(W, H, rank, world, steps, cand, moments_radius, motion_reach)
                    g.plan = cand
                    return g
                except ValueError:
                    continue
            raise ValueError(f"{world} strips of a {H}-row frame are shorter than every halo plan")
        groups = PLANS[plan](steps) if isinstance(plan, str) else [list(g) for g in plan]
        assert [i for g in groups for i in g] == list(range(steps)), "plan must list the iterations in order"
        g = Geometry(W, H, rank, world, steps, moments_radius, motion_reach, groups)
        g.plan = plan
        g.own = partition(H, world)[rank]
        g.ext_atrous = [0] * steps
        for grp in groups:
            for i in grp:
                g.ext_atrous[i] = sum(2 << k for k in grp if k > i)
        g.halo_group = [sum(2 << k for k in grp) for grp in groups]
        g.ext_moments = g.halo_group[0] if groups else 0
        g.ext_temporal = g.ext_moments + moments_radius
        g.halo_state = g.ext_temporal + motion_reach
        g.halo_max = max([g.halo_state] + g.halo_group)
        g.y0 = max(0, g.own[0] - g.halo_max)
        g.y1 = min(H, g.own[1] + g.halo_max)
        smallest = min(b - a for a, b in partition(H, world))
        if world > 1 and smallest < g.halo_max:
            raise ValueError(f"strips of {smallest} rows are shorter than the {g.halo_max}-row halo of plan {plan}: "
                             f"use fewer ranks or a plan with smaller groups")
        return g

    def rows(self, ext):
        """Owned rows grown by `ext` each side, clipped to the frame."""
        return max(0, self.own[0] - ext), min(self.H, self.own[1] + ext)

    @property
    def up(self):
        return self.rank - 1 if self.rank > 0 else None

    @property
    def down(self):
        return self.rank + 1 if self.rank < self.world - 1 else None


class DistComm:
    """Halo exchange over torch.distributed point-to-point ops (RCCL send/recv on MI355X, gloo on CPU).

    On a GPU the batch is posted from a communication stream of its own that waits for an event of the compute stream,
    and the compute stream later waits for an event of that stream: whatever the process group puts on its "current
    stream" when a batch is posted (measured: ~70 us during which neither the RCCL kernel nor the next filter kernel
    starts, GPU-side, tools/strip_sim.py) lands beside the filter kernels instead of between two of them."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group, self.torch = dist, group, torch
        self.post_stream = torch.cuda.Stream(device=device) if device is not None and torch.device(device).type == "cuda" else None

    def start(self, sends, recvs):
        """sends/recvs: lists of (tensor_slice, peer_rank).  Returns an opaque handle."""
        d, torch = self.dist, self.torch
        ops = [d.P2POp(d.irecv, t, p, group=self.group) for t, p in recvs] + [d.P2POp(d.isend, t, p, group=self.group) for t, p in sends]
        if not ops:
            return []
        if self.post_stream is None:
            return d.batch_isend_irecv(ops)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.post_stream):
            self.post_stream.wait_event(ready)
            return d.batch_isend_irecv(ops)

    def finish(self, handle):
        if not handle:
            return
        if self.post_stream is None:
            for w in handle:
                w.wait()
            return
        torch = self.torch
        with torch.cuda.stream(self.post_stream):
            for w in handle:
                w.wait()
            done = torch.cuda.Event()
            done.record(self.post_stream)
        torch.cuda.current_stream().wait_event(done)


class LocalComm:
    """In-process stand-in for DistComm driving N virtual ranks in lock step (tests: strips on ONE device).
    start() posts the sends in a mailbox, finish() copies them into the receivers."""

    def __init__(self):
        self.box = {}

    def for_rank(self, rank):
        parent = self

        class _C:
            """Exchanges are matched by their sequence number on the rank (every rank posts the same sequence), as RCCL
            matches sends and receives of a peer pair in posting order, so handles may be finished in any order."""
            seq = 0

            def start(self, sends, recvs):
                n = self.seq
                self.seq += 1
                for t, p in sends:
                    parent.box.setdefault((rank, p, n), []).append(t.clone())
                return n, recvs

            def finish(self, handle):
                n, recvs = handle
                for t, p in recvs:
                    t.copy_(parent.box[(p, rank, n)].pop(0))
        return _C()


class StripRunner:
    """One rank's share of the frame sequence application::Render runs (src/App.cu:552-556)."""

    def __init__(self, geo: Geometry, stages, comm, storage="f32", device="cpu"):
        import torch
        self.torch, self.geo, self.st, self.comm = torch, geo, stages, comm
        dt = torch.float32 if storage == "f32" else torch.float16
        n, W = geo.y1 - geo.y0, geo.W
        z = lambda ch, d: torch.zeros((n, W, ch) if ch else (n, W), dtype=d, device=device)   # noqa: E731
        self.colour = [z(4, dt), z(4, dt)]       # RenderBuffer[2]    App.h:138
        self.mom = [z(2, dt), z(2, dt)]          # MomentsBuffer[2]   App.h:139
        self.filt = [z(4, dt), z(4, dt)]         # FilterBuffer[2]    App.h:140
        self.hist = [z(0, torch.uint8), z(0, torch.uint8)]
        self.P = 0
        self.pending_state = None                # exchange of this frame's state, waited for by the next frame

    # -- halo plumbing -----------------------------------------------------------------------
    def _halo_ops(self, planes, h, lo=0):
        """Send my owned rows at distance [lo, h) from each strip edge, receive the neighbours' into my halo rows at the
        same distances.  lo > 0: the nearer halo rows are already held (computed redundantly, bit-identically)."""
        g = self.geo
        a, b = g.own[0] - g.y0, g.own[1] - g.y0          # local indices of the owned rows
        sends, recvs = [], []
        if h <= lo:
            return sends, recvs
        for t in planes:
            if g.up is not None:
                sends.append((t[a + lo:a + h], g.up))
                recvs.append((t[a - h:a - lo], g.up))
            if g.down is not None:
                sends.append((t[b - h:b - lo], g.down))
                recvs.append((t[b + lo:b + h], g.down))
        return sends, recvs

    def _split(self, rows, reach):
        """Rows of `rows` that need no neighbour data when the stage reaches `reach` rows, and the rest."""
        g = self.geo
        lo = rows[0] if g.up is None else max(rows[0], g.own[0] + reach)
        hi = rows[1] if g.down is None else min(rows[1], g.own[1] - reach)
        if hi <= lo:
            return None, [rows]
        edges = [r for r in ((rows[0], lo), (hi, rows[1])) if r[1] > r[0]]
        return (lo, hi), edges

    # -- one frame ---------------------------------------------------------------------------
    def frame_steps(self, radiance, gb_cur, gb_prev=None):
        """Generator: yields between posting an exchange and waiting for it (LocalComm lock step); the value of
        the finished generator (StopIteration.value) is the plane holding the result."""
        g, st, P = self.geo, self.st, self.P
        if gb_prev is None:
            gb_prev = gb_cur
        # previous-frame state halo (colour feedback, moments, history on the rows the reprojection can reach): posted by
        # the PREVIOUS frame right after its iteration 0, so the transfer ran beside that frame's remaining iterations
        if self.pending_state is not None:
            self.comm.finish(self.pending_state)
            self.pending_state = None
        if hasattr(st, "temporal_moments"):
            st.temporal_moments(g.rows(g.ext_temporal), g.rows(g.ext_moments), self.colour[1 - P], radiance, self.colour[P], self.filt[0],
                                gb_cur, gb_prev, self.hist[1 - P], self.hist[P], self.mom[P], self.mom[1 - P],
                                # rows beyond the reach of this rank's iteration 0 are replaced by the state exchange
                                feedback_follows=g.steps >= 1)
        else:
            st.temporal(g.rows(g.ext_temporal), self.colour[1 - P], radiance, self.colour[P], gb_cur, gb_prev, self.hist[1 - P], self.hist[P],
                        self.mom[P], self.mom[1 - P])
            st.moments(g.rows(g.ext_moments), self.colour[P], self.filt[0], self.mom[P], gb_cur, self.hist[P])
        pp = 0
        h = None                                   # the exchange of filter rows in flight for the group about to start
        for gi, grp in enumerate(g.groups):
            for k, i in enumerate(grp):
                rows = g.rows(g.ext_atrous[i])
                fb = self.colour[P] if i == 0 else None
                if k == 0 and h is not None:
                    yield
                    self.comm.finish(h)
                    h = None
                # The last iteration of a group that another group follows produces the rows its neighbours need FIRST (the rows within
                # the next group's halo of the strip boundaries), posts the exchange behind them and runs its interior beside the transfer
                # (the schedule of svgf_strips_frame, svgf_amd/csrc/svgf_strip.hip).
                if k + 1 == len(grp) and gi + 1 < len(g.groups) and g.world > 1:
                    hn = g.halo_group[gi + 1]
                    inner, edges = self._split(rows, hn)
                    for r in edges:
                        st.atrous(r, self.filt[pp], self.filt[1 - pp], fb, gb_cur, 1 << i, i)
                    h = self.comm.start(*self._halo_ops([self.filt[1 - pp]], hn))
                    if inner:
                        st.atrous(inner, self.filt[pp], self.filt[1 - pp], fb, gb_cur, 1 << i, i)
                else:
                    st.atrous(rows, self.filt[pp], self.filt[1 - pp], fb, gb_cur, 1 << i, i)
                pp ^= 1
                if i == 0:
                    self._post_state(P)            # this frame's state is final once iteration 0 has written the feedback colour
        if not g.steps:
            self._post_state(P)
        self.P ^= 1
        return self.filt[pp]

    def _post_state(self, P):
        """Post the exchange of this frame's state for the next frame's reprojection (reach: halo_state rows).  Only the rows
        this rank has NOT computed itself travel: it holds the feedback colour up to ext_atrous[0] rows beyond its strip
        and moments / history up to ext_temporal rows (bit-identical to the owner's), so per boundary and direction
        2 + moments_radius + motion_reach rows of colour and motion_reach rows of moments and history are sent — 1.4 MB at
        8K instead of the 13 MB of the full 69-row halo of the ghost plan."""
        g = self.geo
        if g.world <= 1:
            return
        colour_held = g.ext_atrous[0] if g.steps else g.ext_temporal
        sends, recvs = [], []
        for t, held in ((self.colour[P], colour_held), (self.mom[P], g.ext_temporal), (self.hist[P], g.ext_temporal)):
            s, r = self._halo_ops([t], g.halo_state, lo=held)
            sends += s
            recvs += r
        self.pending_state = self.comm.start(sends, recvs)

    def frame(self, radiance, gb_cur, gb_prev=None):
        it = self.frame_steps(radiance, gb_cur, gb_prev)
        try:
            while True:
                next(it)
        except StopIteration as e:
            return e.value

    def flush(self):
        """Wait for the exchange the last frame posted (call before tearing the process group down)."""
        if self.pending_state is not None:
            self.comm.finish(self.pending_state)
            self.pending_state = None

    def owned(self, plane):
        g = self.geo
        return plane[g.own[0] - g.y0:g.own[1] - g.y0]


def _subtract(rows, done):
    """Row ranges of `rows` not covered by the ranges in `done` (sorted, disjoint, inside rows)."""
    out, lo = [], rows[0]
    for a, b in sorted(done):
        if a > lo:
            out.append((lo, a))
        lo = max(lo, b)
    if lo < rows[1]:
        out.append((lo, rows[1]))
    return out


def run_virtual(runners, inputs):
    """Drive N virtual ranks (LocalComm) in lock step through one frame; inputs[r] = (radiance, gb_cur, gb_prev)."""
    its = [r.frame_steps(*inputs[k]) for k, r in enumerate(runners)]
    results = [None] * len(its)
    live = set(range(len(its)))
    while live:
        for k in sorted(live):
            try:
                next(its[k])
            except StopIteration as e:
                results[k] = e.value
                live.discard(k)
    return results


def strips_plan(W, H, rank, world, steps, plan="auto", moments_radius=3, motion_reach=0):
    """svgf_strips_plan (the C++ restatement of Geometry.make) -> dict; raises ValueError if the strips are too short."""
    import ctypes as C
    from . import filter as F
    lib = F.load_library()
    lay = F.StripLayoutC()
    rc = lib.svgf_strips_plan(W, H, rank, world, steps, F.HALO_PLAN[plan], moments_radius, motion_reach, C.byref(lay))
    if rc != 0:
        raise ValueError(f"svgf_strips_plan: {lib.svgf_status_string(rc).decode()}")
    st = lay.strip
    return dict(plan=F.HALO_PLAN_NAME[lay.plan], y0=st.y0, y1=st.y0 + st.rows, own=(st.own_begin, st.own_end), ext_atrous=list(lay.ext_atrous)[:steps],
                halo_group=list(lay.halo_group)[:lay.ngroups], group_first=list(lay.group_first)[:lay.ngroups], ext_moments=lay.ext_moments,
                ext_temporal=lay.ext_temporal, halo_state=lay.halo_state, halo_max=lay.halo_max)


def strip_messages(W, H, rank, world, steps, plan="auto", moments_radius=3, motion_reach=0, storage="f32"):
    """svgf_strips_messages -> the messages of one frame as `rank` posts them: list of dicts (exchange, send, peer, plane, rows, bytes)."""
    import ctypes as C
    from . import filter as F
    lib = F.load_library()
    cap = 64
    buf = (F.StripMessageC * cap)()
    n = C.c_int()
    rc = lib.svgf_strips_messages(W, H, rank, world, steps, F.HALO_PLAN[plan], moments_radius, motion_reach, F.STORAGE[storage], buf, cap, C.byref(n))
    if rc != 0:
        raise ValueError(f"svgf_strips_messages: {lib.svgf_status_string(rc).decode()}")
    return [dict(exchange=m.exchange, send=bool(m.send), peer=m.peer, plane=m.plane, rows=(m.row_begin, m.row_end), bytes=m.bytes) for m in buf[:n.value]]


def rccl_comm(world, rank, device_index, group=None):
    """An ncclComm_t for the C++ strip driver: rank 0 draws the unique id (svgf_rccl_unique_id), torch.distributed carries its
    128 bytes to the other ranks — the only thing Python does for the exchange — and every rank joins (svgf_rccl_comm_init)."""
    import ctypes as C
    import torch.distributed as dist
    from . import filter as F
    lib = F.load_library()
    buf = (C.c_char * 128)()
    ok = True
    if rank == 0:
        ok = lib.svgf_rccl_unique_id(buf) == 0
    box = [bytes(buf.raw) if ok else b""]            # rank 0's failure travels too: nobody is left waiting in the broadcast
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    if len(box[0]) != 128:
        raise F.SvgfError("svgf_rccl_unique_id failed on rank 0 (librccl not available?)")
    ident = (C.c_char * 128).from_buffer_copy(box[0])
    comm = C.c_void_p()
    rc = lib.svgf_rccl_comm_init(C.byref(comm), world, rank, ident, device_index)
    if rc != 0:
        raise F.SvgfError("svgf_rccl_comm_init failed")
    return comm


class NativeStrips:
    """The C++ strip driver of the library (svgf_strips_*, svgf_amd/csrc/svgf_strip.hip): the stage sequence of every local
    strip, the RCCL groups, the communication stream and the events all live in C++; Python hands over device pointers."""

    def __init__(self, W, H, world, params, ranks, devices, streams=None, comms=None, plan="auto", motion_reach=0, loopback=False, transport=None):
        """transport: "rccl" (default; one communicator per local rank), "rccl-loopback" (= loopback=True: ONE communicator of size 1), or
        "mailbox" (tests: every rank local, real peer addressing, sends matched to receives inside the library — include/svgf.h)."""
        import ctypes as C
        import torch
        from . import filter as F
        self.F, self.C, self.torch = F, C, torch
        self.lib = F.load_library()
        self.W, self.H, self.world, self.params, self.n = W, H, world, params, len(ranks)
        self.ranks, self.devices = list(ranks), list(devices)
        self.storage = params.storage
        n = self.n
        pc = params.to_c()
        r_arr, d_arr = (C.c_int * n)(*ranks), (C.c_int * n)(*devices)
        s_arr = (C.c_void_p * n)(*[(s if s is not None else None) for s in (streams or [None] * n)])
        transport = transport or ("rccl-loopback" if loopback else "rccl")
        self.transport = transport
        ncomm = 1 if transport == "rccl-loopback" else n
        c_arr = (C.c_void_p * ncomm)(*[(c.value if hasattr(c, "value") else c) for c in comms]) if comms and transport != "mailbox" else None
        h = C.c_void_p()
        rc = self.lib.svgf_strips_create(C.byref(h), W, H, world, C.byref(pc), F.HALO_PLAN[plan], motion_reach, n, r_arr, d_arr, s_arr, c_arr, F.TRANSPORT[transport])
        if rc != 0:
            raise F.SvgfError(f"svgf_strips_create: {self.lib.svgf_status_string(rc).decode()}")
        self._h = h
        self.layouts = []
        for k in range(n):
            lay = F.StripLayoutC()
            self._check(self.lib.svgf_strips_layout(self._h, k, C.byref(lay)))
            st = lay.strip
            self.layouts.append(dict(plan=F.HALO_PLAN_NAME[lay.plan], y0=st.y0, y1=st.y0 + st.rows, own=(st.own_begin, st.own_end), halo_state=lay.halo_state,
                                     halo_max=lay.halo_max, ext_atrous=list(lay.ext_atrous)[:params.steps], ext_temporal=lay.ext_temporal))
        self.plan = self.layouts[0]["plan"]

    def _check(self, rc, what="svgf_strips"):
        if rc != 0:
            raise self.F.SvgfError(f"{what}: {self.lib.svgf_status_string(rc).decode()}: {self.lib.svgf_strips_last_error(self._h).decode()}")

    def frame(self, radiance, cur, prev=None):
        """radiance: list of tensors, cur / prev: lists of filter.GBuffer (prev None on the first frame) -> list of result tensors
        (views of library-owned planes holding rows [y0, y1) of each strip)."""
        C, n = self.C, self.n
        rad = (C.c_void_p * n)(*[t.data_ptr() for t in radiance])
        gc = (self.F.GBufferC * n)(*[g._c for g in cur])
        gp = (self.F.GBufferC * n)(*[g._c for g in prev]) if prev is not None else None
        res = (C.c_void_p * n)()
        self._check(self.lib.svgf_strips_frame(self._h, rad, gc, gp, res), "svgf_strips_frame")
        return [self._wrap(k, res[k]) for k in range(n)]

    def _wrap(self, k, ptr, plane=None):
        torch, lay = self.torch, self.layouts[k]
        rows = lay["y1"] - lay["y0"]
        dt = torch.float32 if self.storage == "f32" else torch.float16
        if plane == self.F.PLANE_HISTORY:
            shape, typestr = (rows, self.W), "|u1"
        else:
            shape, typestr = (rows, self.W, 2 if plane == self.F.PLANE_MOMENTS else 4), "<f4" if dt == torch.float32 else "<f2"

        class _Holder:
            pass
        hld = _Holder()
        hld.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 3}
        hld.owner = self
        return torch.as_tensor(hld, device=torch.device("cuda", self.devices[k]))

    def state_plane(self, k, plane, index):
        ctx = self.lib.svgf_strips_context(self._h, k)
        p = self.lib.svgf_state_plane(ctx, plane, index)
        return self._wrap(k, p, plane) if p else None

    def pingpong(self, k=0):
        return self.lib.svgf_state_pingpong(self.lib.svgf_strips_context(self._h, k))

    def set_prev_guide(self, enable=True):
        """svgf_set_prev_guide on every local strip: the caller vouches that the previous G-buffer's planes are not rewritten between frames."""
        for k in range(self.n):
            self._check(self.lib.svgf_set_prev_guide(self.lib.svgf_strips_context(self._h, k), 1 if enable else 0))

    def set_frames_in_flight(self, frames=2):
        """svgf_strips_set_frames_in_flight: 2 = iterations 1.. of a frame on a side stream beside the next frame's temporal launch (same bits;
        a frame's result is ordered on the compute stream two calls later, or by sync())."""
        self._check(self.lib.svgf_strips_set_frames_in_flight(self._h, int(frames)), "svgf_strips_set_frames_in_flight")

    def set_edge_first(self, enable=True):
        """svgf_strips_set_edge_first: the iteration in front of an exchange as ONE launch whose first workgroups produce the edge rows and signal (an opt-in: default off, include/svgf_ext.h)."""
        self._check(self.lib.svgf_strips_set_edge_first(self._h, 1 if enable else 0))

    def set_iteration_fusion(self, enable=True):
        """Iterations 0 and 1 as one launch on every local strip (where the halo plan keeps them in one group)."""
        for k in range(self.n):
            self._check(self.lib.svgf_set_iteration_fusion(self.lib.svgf_strips_context(self._h, k), 1 if enable else 0))

    def transport_stats(self):
        """mailbox transport: (groups matched, copies enqueued, bytes copied) so far."""
        C = self.C
        g, c, b = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        self._check(self.lib.svgf_strips_transport_stats(self._h, C.byref(g), C.byref(c), C.byref(b)))
        return g.value, c.value, b.value

    def owned(self, k, t):
        lay = self.layouts[k]
        return t[lay["own"][0] - lay["y0"]:lay["own"][1] - lay["y0"]]

    def sync(self):
        self._check(self.lib.svgf_strips_sync(self._h), "svgf_strips_sync")

    def timing_enable(self, every):
        self._check(self.lib.svgf_strips_timing_enable(self._h, int(every)))

    def timing_read(self):
        C = self.C
        n, ms, pa, p0 = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.svgf_strips_timing_read(self._h, C.byref(n), C.byref(ms), C.byref(pa), C.byref(p0)))
        return n.value, ms.value, pa.value, p0.value

    def close(self):
        if getattr(self, "_h", None):
            self.lib.svgf_strips_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _NativeRunner:
    """One rank's strip through the C++ strip driver (svgf_strips_frame)."""
    name = "C++ (svgf_strips_frame)"

    def __init__(self, W, H, world, rank, params, device, stream, comm, plan, motion_reach):
        self.drv = NativeStrips(W, H, world, params, [rank], [device.index or 0], streams=[stream], comms=[comm] if comm else None, plan=plan, motion_reach=motion_reach)
        # (svgf_set_prev_guide stays off: the ABI's default is what the N > 1 line reports, like the N = 1 headline)
        self.lay = self.drv.layouts[0]

    def frame(self, rad, cur, prev):
        return self.drv.frame([rad], [cur], [prev] if prev is not None else None)[0]

    def owned(self, t):
        return self.drv.owned(0, t)

    def timing(self, on):
        self.drv.timing_enable(4 if on else 0)

    def frame_timing(self, k):
        pass

    def sync(self):
        self.drv.sync()

    def atrous_timing(self, bytes_iter, bytes_feedback):
        nl, ms, px_all, px0 = self.drv.timing_read()
        return nl, ms, px_all * bytes_iter + px0 * bytes_feedback

    def close(self):
        self.drv.close()


VERIFY_FRAMES = 6        # bench_strips' check of the strips against the one-GPU frame: frames from a fresh start (history passes 4: cold and steady moments paths)


def _checksum(t):
    """Two 64-bit sums over the raw 32-bit words of a device tensor (wrap-around arithmetic): a plain sum and a position-weighted one.  Equal
    checksums of 66 MB of pixels are, for every practical purpose, equal pixels; the position weights catch rows that arrive in the wrong place."""
    import torch
    v = t.contiguous().view(torch.int32).flatten().to(torch.int64) & 0xFFFFFFFF
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521) + 1
    return torch.stack([v.sum(), (v * w).sum()])


def _all_ranks_match(mine, ref_sums, rank, world):
    """bench_strips' verification, the part that talks to the other ranks: every rank contributes the checksum of its owned rows (`mine`, a tensor on
    the process group's device), rank 0 compares the gathered list with the checksums of the same rows of its one-GPU frame (`ref_sums`, [world, 2];
    None on the other ranks) and every rank learns the verdict.  -> bool, the same on every rank."""
    import torch
    import torch.distributed as dist
    sums = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(sums, mine)
    else:
        sums = [mine]
    good = rank != 0 or bool((torch.stack(sums).cpu() == ref_sums.cpu()).all())
    ok = torch.tensor([1 if good else 0], device=mine.device, dtype=torch.int32)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return bool(int(ok.item()))


def _strip_pan_frames(W, H, storage, device, mv, y0, y1):
    """Two consecutive frames of a camera pan (rows [y0, y1)), walked forth and back: frame n shows canvas n & 1; walking back, a
    frame's predecessor is the other one with the motion vector negated (bench.py's FramePool, pool of 2).
    -> (radiances[2], G-buffer of frame n as current: gbs[n & 1][direction])."""
    import numpy as np
    import torch
    from . import filter as F
    from . import synth
    npdt = np.float32 if storage == "f32" else np.float16
    rads, gbs = [], []
    for f in (0, 1):
        sc = synth.make_scene(W, H, f, mv=mv, row_begin=y0, row_end=y1)
        mo = torch.from_numpy(sc["motion"]).to(device)
        back = mo.clone()
        back[..., :2] *= -1.0
        no, uv = torch.from_numpy(sc["normal"]).to(device), torch.from_numpy(sc["uv"]).to(device)
        gbs.append((F.GBuffer(mo, no, uv), F.GBuffer(back, no, uv)))
        rads.append(torch.from_numpy(synth.make_radiance(sc["base"], W, f, row_begin=y0).astype(npdt)).to(device))
    return rads, gbs


def bench_strips(W, H, storage, iters, variant, steps, warmup, device, plan, make_inputs, prime_frames, motion_reach=None,
                 plans=("per-iteration", "grouped"), pan_mv=(1.5, -3.5), one_gpu_reference=True, busy=(400.0, 600),
                 on_phase=None, on_head=None):
    """bench.py's N > 1 leg: this rank's strip of a W x H frame; every measurement is `steps` frames between barriers, MAX over ranks.
    Measured: the headline plan (`plan`, static camera), the other halo plans (BASELINE config #4 names "per-iteration"), a camera
    pan whose state exchange really carries moments and history (motion reach >= 3), and — on rank 0 alone, before the strips —
    the whole frame on one GPU, which is what the strip-parallel speed-up is relative to.
    The driver is the C++ strip driver of the library (RCCL groups posted from C++); a rank that cannot bring its communicator up makes
    every rank raise (bench.py exits non-zero): there is nothing else to fall back to.
    on_phase(name): called before every leg (bench.py's watchdog: a leg that never returns ends the job with what is measured so far);
    on_head(result): called with the result so far once the headline plan is measured, and again after every further leg."""
    import math
    import time
    import torch
    import torch.distributed as dist
    from . import filter as F
    rank, world = dist.get_rank(), dist.get_world_size()
    # A high-priority side stream for the kernels: HIP maps a priority level to hardware queues of its own, so the RCCL
    # send/recv kernels (the communication stream, normal priority) run BESIDE the filter kernels.  On one shared
    # queue they run in line with them (tools/strip_sim.py: 0.66 vs 0.55 ms per 8K/8 strip).
    side = torch.cuda.Stream(device=device, priority=-1)
    torch.cuda.set_stream(side)
    params = F.Params(storage=storage, steps=iters, variant=variant)

    def max_over_ranks(x):
        t = torch.tensor([x], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the whole frame on ONE GPU (rank 0), through svgf_denoise_frame: the time the N-GPU speed-up is relative to
    phase = on_phase or (lambda name: None)
    one_gpu_ms, ref_sums = None, None
    if one_gpu_reference:
        phase("the whole frame on one GPU (rank 0)")
        if rank == 0:
            gb_w, rads_w = make_inputs(W, H, storage, device, nframes=2)
            gb2 = F.GBuffer(gb_w.motion.clone(), gb_w.normal.clone(), gb_w.uv.clone())
            d = F.Denoiser(W, H, params, device=device.index or 0, stream=side.cuda_stream)
            gbp = [gb_w, gb2]
            n = 0
            # primed like the strips it is compared with (`busy`: at least that many ms AND frames of untimed load: the post-idle clock ramp
            # and the one-off stall of a process's first ~4 000 stream operations), and the median of five windows
            t_w, k_w = time.perf_counter(), 0
            while (time.perf_counter() - t_w) * 1e3 < busy[0] or k_w < max(busy[1], min(prime_frames, 12) + warmup):
                for _ in range(10):
                    d.Render(rads_w[n & 1], gbp[n & 1], gbp[(n & 1) ^ 1])
                    n += 1
                k_w += 10
                torch.cuda.synchronize(device)
            wins = []
            for _ in range(5):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(steps):
                    d.Render(rads_w[n & 1], gbp[n & 1], gbp[(n & 1) ^ 1])
                    n += 1
                torch.cuda.synchronize(device)
                wins.append((time.perf_counter() - t0) * 1e3 / steps)
            one_gpu_ms = sorted(wins)[len(wins) // 2]
            # ... and, from a fresh start, the VERIFY_FRAMES frames every strip schedule below is checked against, bit for bit (verify())
            d.reset_history()
            for k in range(VERIFY_FRAMES):
                out_w = d.Render(rads_w[k & 1], gbp[k & 1], gbp[(k & 1) ^ 1] if k else None)
            torch.cuda.synchronize(device)
            parts = partition(H, world)
            ref_sums = torch.stack([_checksum(out_w[a:b]) for a, b in parts])       # [world, 2]: what rank r's owned rows must sum to
            d.close()
            del d, gb_w, gb2, rads_w, gbp, out_w
            torch.cuda.empty_cache()
        one_gpu_ms = max_over_ranks(one_gpu_ms or 0.0)          # every rank learns rank 0's figure (and waits for it)

    # ---- the strip's static inputs: its rows of the frame; the rows needed depend on the plan, which depends on the motion reach,
    # which is read off the inputs: generate the widest candidate (ghost plan, reach 8), measure, then cut
    phase("inputs and communicator")
    wide = "ghost" if plan == "auto" else plan
    probe = strips_plan(W, H, rank, world, iters, plan=wide, moments_radius=params.moments_radius, motion_reach=8) \
        if _plan_fits(W, H, rank, world, iters, wide, params.moments_radius, 8) else None
    y0p, y1p = (probe["y0"], probe["y1"]) if probe else (0, H)
    gb_all, rads_all = make_inputs(W, H, storage, device, row_begin=y0p, row_end=y1p)
    if motion_reach is None:
        mvy = gb_all.motion[..., 1].abs().max().reshape(1).to(torch.float32)
        if world > 1:
            dist.all_reduce(mvy, op=dist.ReduceOp.MAX)
        motion_reach = int(math.ceil(float(mvy.item())))

    comm, rccl_ranks = None, None
    if world > 1:
        # every rank brings the communicator up; if any rank cannot, ALL stop (no silent change of what is measured)
        err = ""
        try:
            comm = rccl_comm(world, rank, device.index or 0)
            rccl_ranks = rccl_comm_count(comm)
        except Exception as e:  # noqa: BLE001
            err, comm = f"{type(e).__name__}: {e}", None
        bad = torch.tensor([0 if comm is not None else 1], device=device, dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            raise F.SvgfError(f"rank {rank}: the C++ strip driver's RCCL communicator is unavailable on some rank ({err or 'see the other ranks'}): "
                              "nothing is measured (RCCL wants one device per rank)")
    Runner = _NativeRunner

    def measure(plan_name, reach, frame_of, keep_timing=False, edge_first=False):
        """One driver under one plan: prime, then `steps` frames between barriers.  frame_of(n) -> (radiance, cur, prev) of THIS layout.
        edge_first=False: the library's default schedule (three launches per exchanging iteration); True: svgf_strips_set_edge_first(1)."""
        lay = strips_plan(W, H, rank, world, iters, plan=plan_name, moments_radius=params.moments_radius, motion_reach=reach)
        run = Runner(W, H, world, rank, params, device, side.cuda_stream, comm, lay["plan"], reach)
        run.drv.set_edge_first(bool(edge_first))
        get = frame_of(lay)
        n = 0
        dist.barrier()                           # the ranks enter the untimed frames together ...
        torch.cuda.synchronize(device)
        w0 = time.perf_counter()
        for _ in range(prime_frames + warmup):
            run.frame(*get(n))
            n += 1
        torch.cuda.synchronize(device)
        # ... and keep the device busy for >= busy[0] = 400 ms and >= busy[1] = 600 frames before anything is timed: after the idle gaps of the set-up
        # (allocations, rank 0's whole-frame reference) the part needs tens of milliseconds at load to be back at its clocks
        # (tools/archive/idle_gap.py), and a process sees one stall of 20-65 ms when it has enqueued its first ~4 000 stream operations (~300
        # frames; tools/strip_sim.py --per-frame) that would otherwise land in the timed frames.  The number of extra frames comes from
        # the all-reduced per-frame time, so every rank runs the same count (frames exchange halos).
        done = prime_frames + warmup
        per = max(max_over_ranks((time.perf_counter() - w0) * 1e3 / done), 1e-3)
        for _ in range(int(min(3000, max(busy[1] - done, math.ceil((busy[0] - per * done) / per), 0)))):
            run.frame(*get(n))
            n += 1
        run.timing(True)
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        host = 0.0
        for k in range(steps):
            run.frame_timing(k)
            h0 = time.perf_counter()
            run.frame(*get(n))
            host += time.perf_counter() - h0
            n += 1
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        run.timing(False)
        out = run.frame(*get(n))
        run.sync()                               # raises if a reprojection left the strip (motion reach too small)
        assert bool(torch.isfinite(run.owned(out).float()).all())
        res = dict(ms_per_step=max_over_ranks((t1 - t0) * 1e3 / steps), plan=lay["plan"], rows_per_rank=lay["own"][1] - lay["own"][0],
                   rows_held=lay["y1"] - lay["y0"], host_ms=round(host * 1e3 / steps, 4), motion_reach=reach)
        if keep_timing:
            # read now and close: a second strip driver alive beside the one being timed (the other plans below) is not something the figures
            # of either should depend on (tools/strip_sim.py measured drivers taking turns in one process up to twice as slow)
            nl, ms_l, px_all, px0 = run.drv.timing_read()
            res["atrous_timing"] = lambda bytes_iter, bytes_feedback: (nl, ms_l, px_all * bytes_iter + px0 * bytes_feedback)
        run.close()
        return res

    def static_frames(lay, nrad=None):
        assert lay["y0"] >= y0p and lay["y1"] <= y1p, "the probe rows do not cover this layout"
        cut = slice(lay["y0"] - y0p, lay["y1"] - y0p)
        # current and previous G-buffer in DISTINCT planes, ping-ponged (src/App.cu:471-474), also with a static camera
        gbs = [F.GBuffer(gb_all.motion[cut].clone(), gb_all.normal[cut].clone(), gb_all.uv[cut].clone()) for _ in range(2)]
        rads = [r[cut].contiguous() for r in rads_all[:nrad]]
        return lambda n: (rads[n % len(rads)], gbs[n & 1], gbs[(n & 1) ^ 1])

    def verify(plan_name, edge_first):
        """VERIFY_FRAMES frames from a fresh start through the strips under one schedule; every rank's OWNED rows of the last result against the
        same rows of the one-GPU frame (rank 0's reference above), as 2 x 64-bit checksums over the raw bits: -> True / False on every rank."""
        lay = strips_plan(W, H, rank, world, iters, plan=plan_name, moments_radius=params.moments_radius, motion_reach=motion_reach)
        run = Runner(W, H, world, rank, params, device, side.cuda_stream, comm, lay["plan"], motion_reach)
        run.drv.set_edge_first(edge_first)
        get = static_frames(lay, nrad=2)
        dist.barrier()
        for k in range(VERIFY_FRAMES):
            rad, cur, prev = get(k)
            out = run.frame(rad, cur, prev if k else None)
        run.sync()
        ok = _all_ranks_match(_checksum(run.owned(out)).to(device), ref_sums, rank, world)
        run.close()
        return ok

    # ---- The headline FIRST, under the library's default schedule (every exchange ordered behind an event: three launches per exchanging iteration) —
    # whatever happens afterwards, this line exists (bench.py's watchdog prints what is measured when a later leg hangs).
    others, pan, verified = {}, None, None
    phase(f"headline: plan {plan}, static camera")
    head = measure(plan, motion_reach, static_frames, keep_timing=True, edge_first=False)
    head["edge_first"] = False

    def so_far():
        res = dict(head)
        res.update(driver=Runner.name, other_plans=dict(others), pan=pan, one_gpu_ms=one_gpu_ms, rccl_ranks=rccl_ranks, _comm=comm, verified=verified)
        return res
    if on_head:
        on_head(so_far())

    # ---- Then the check against the one-GPU frame, under both schedules.  The opt-in "edge rows first" (svgf_strips_set_edge_first) signals from inside
    # ONE launch and is faster in the one-GPU simulation: it replaces the headline — and times the other legs — if and only if THIS run, on THESE ranks,
    # reproduces the one-GPU frame bit for bit with it (include/svgf_ext.h).
    if ref_sums is not None or (one_gpu_reference and rank != 0):
        phase("verification against the one-GPU frame")
        verified = {"frames": VERIFY_FRAMES, "plan": head["plan"], "what": "owned rows of every rank after that many frames from a fresh start == the same rows of the one-GPU frame (raw bits, 2 x 64-bit checksums)",
                    "three_launches": verify(head["plan"], False)}
        if on_head:
            on_head(so_far())
        verified["edge_first"] = verify(head["plan"], True)
        if on_head:
            on_head(so_far())
    use_edge_first = bool(verified and verified["edge_first"])
    if use_edge_first and head["plan"] != "ghost" and world > 1:
        phase(f"headline again: plan {plan}, edge rows first")
        fast = measure(plan, motion_reach, static_frames, keep_timing=True, edge_first=True)
        fast["edge_first"], fast["ms_per_step_three_launches"] = True, head["ms_per_step"]
        head = fast
        if on_head:
            on_head(so_far())
    for pl in plans:
        if pl == head["plan"] or not _plan_fits(W, H, rank, world, iters, pl, params.moments_radius, motion_reach):
            continue
        phase(f"plan {pl}")
        r = measure(pl, motion_reach, static_frames, edge_first=use_edge_first)
        others[pl] = {k: r[k] for k in ("ms_per_step", "rows_held", "host_ms")}
        others[pl]["edge_first"] = use_edge_first
        if on_head:
            on_head(so_far())
        if pl != "ghost" and use_edge_first and world > 1:
            # the same plan under the default schedule: on real links this pair of numbers is what the edge-rows-first launch is worth
            phase(f"plan {pl}, three launches per exchanging iteration")
            others[pl]["ms_per_step_three_launches"] = measure(pl, motion_reach, static_frames, edge_first=False)["ms_per_step"]
            if on_head:
                on_head(so_far())

    if pan_mv is not None:
        reach = int(math.ceil(abs(pan_mv[1])))
        if _plan_fits(W, H, rank, world, iters, "auto", params.moments_radius, reach):
            def pan_frames(lay):
                rads, gbs = _strip_pan_frames(W, H, storage, device, pan_mv, lay["y0"], lay["y1"])
                state = {"prev": None}

                def get(n):
                    cur = gbs[n & 1][0 if (n & 1) else (1 if n else 0)]        # odd frames walk forth (0 -> 1), even ones back (1 -> 0)
                    prev = state["prev"] if state["prev"] is not None else cur
                    state["prev"] = cur
                    return rads[n & 1], cur, prev
                return get
            phase("camera pan")
            r = measure("auto", reach, pan_frames, edge_first=use_edge_first)
            pan = {"mv": list(pan_mv), "motion_reach": reach, "plan": r["plan"], "ms_per_step": r["ms_per_step"], "rows_held": r["rows_held"]}

    phase("done")
    return so_far()


def rccl_comm_count(comm):
    """The number of ranks RCCL itself reports for the communicator (ncclCommCount)."""
    import ctypes as C
    from . import filter as F
    n = C.c_int(-1)
    rc = F.load_library().svgf_rccl_comm_count(comm, C.byref(n))
    if rc != 0:
        raise F.SvgfError("svgf_rccl_comm_count failed")
    return n.value


def _plan_fits(W, H, rank, world, steps, plan, moments_radius, motion_reach):
    try:
        strips_plan(W, H, rank, world, steps, plan=plan, moments_radius=moments_radius, motion_reach=motion_reach)
        return True
    except ValueError:
        return False
