// svgf_strip.hip — multi-GPU row strips behind the C ABI (include/svgf.h, "Multi-GPU").
//
// The reference is single-GPU: application::Render runs TemporalFilter / FilterMoments / WaveletFilter on the whole frame
// (src/App.cu:552-556).  Here the frame is cut into `world` contiguous row strips and the same sequence runs on every strip,
// with the rows a strip needs from its neighbours travelling as RCCL point-to-point transfers
// (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI), posted from a communication stream of their own and tied to
// the filter stream by HIP events only.  There is no collective in the data path and no reduction: every stage is a
// bounded-reach gather, so the strips are bit-identical to the single-GPU frame.
//
// Halo plan.  À-trous iteration i reaches 2*2^i rows.  A plan groups the iterations; a group's input halo (the
// sum of its iterations' reaches) is exchanged once, before the group, and inside the group iteration i runs on `ext_i` extra
// rows each side (redundantly with the neighbour, bit-identically) so that the next iteration finds its halo locally.
//   ghost          [[0,1,2,3,4]]            no transfer between iterations
//   grouped        [[0,1,2],[3,4]]          one transfer between iterations
//   per-iteration  [[0],[1],[2],[3],[4]]    the literal "halo exchange between à-trous iterations"
// The temporal and moments stages run redundantly on the first group's halo, so a frame needs one more exchange: its STATE
// (colour feedback, moments, history) on the rows the next frame's reprojection can reach.  That state is final once
// iteration 0 has written the feedback colour: it is posted right there and waited for at the start of the NEXT frame — the
// transfer runs beside iterations 1.. of the frame that produced it.
//
// A driver object holds the strips of the ranks that live in THIS process: one (a rank per process, the bench.py layout),
// several on several devices (one host process driving the node, ncclCommInitAll-style), or several virtual ranks on one
// device sharing one loop-back communicator (tests: every peer is communicator rank 0).  Frames run in lock step over the
// local ranks so that each exchange is ONE RCCL group holding every local rank's transfers.
//
// RCCL is opened at run time (dlopen of the librccl the process already has — torch brings one — or of ROCm's): hosts that
// never attach a strip driver do not need it.
//
// Transports.  The four calls an exchange is made of — group start, send, receive, group end — go through a table (Transport).  The
// product transport is RCCL.  The second one, the MAILBOX, exists for tests: with every rank of the partition living in this process
// (nlocal == world, one device or several) each rank keeps its own communication stream and addresses its neighbours by their REAL rank
// numbers — the branch of post_exchange a multi-GPU run takes, which the loop-back communicator (every peer is rank 0, one shared
// stream) never enters — and the mailbox matches every send to the receive its peer posted for it, in posting order per (source,
// destination) pair and inside the same group: RCCL's matching rule.  A send without its receive, a receive without its send, or a pair
// whose sizes differ is what would deadlock (or corrupt) a real run: the mailbox refuses the group with SVGF_ERR_COMM and says which.
// A matched pair becomes a device-to-device copy on the RECEIVER's communication stream behind an event of the sender's, and the
// sender's stream waits for the copy (a send is complete when its buffer may be reused).

#include "svgf_ctx.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>

using namespace svgf_host;

namespace {

// ------------------------------------------------------------------ RCCL, resolved at run time ----------
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0 };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("SVGF_RCCL_LIBRARY");
        const char* loaded[] = {"librccl.so", "librccl.so.1"};
        if (env) r.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        for (const char* n : loaded) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the one already in the process (torch's)
        const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : fresh) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { const char* e = dlerror(); r.why = std::string("librccl not found: ") + (e ? e : ""); return; }
#define SVGF_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name)); if (!r.field) r.why = std::string("librccl lacks ") + name
        SVGF_SYM(GetUniqueId, "ncclGetUniqueId"); SVGF_SYM(CommInitRank, "ncclCommInitRank"); SVGF_SYM(CommDestroy, "ncclCommDestroy");
        SVGF_SYM(GroupStart, "ncclGroupStart"); SVGF_SYM(GroupEnd, "ncclGroupEnd"); SVGF_SYM(Send, "ncclSend"); SVGF_SYM(Recv, "ncclRecv");
        SVGF_SYM(GetErrorString, "ncclGetErrorString"); SVGF_SYM(CommCount, "ncclCommCount");
#undef SVGF_SYM
    });
    return &r;
}

}  // namespace

// ------------------------------------------------------------------ the driver -----------------------------
constexpr int kMaxAhead = 32, kAheadStride = 4;      // the host runs at most kMaxAhead (+ kAheadStride) frames ahead of the device

struct svgf_strip_plan_geo {
    int own0 = 0, own1 = 0;
    std::vector<std::vector<int>> groups;
    std::vector<int> ext_atrous, halo_group;
    int ext_moments = 0, ext_temporal = 0, halo_state = 0, halo_max = 0, y0 = 0, y1 = 0;
};

struct svgf_strips {
    int W = 0, H = 0, world = 1, steps = 0, plan = SVGF_PLAN_AUTO, motion_reach = 0, moments_radius = 3, storage = SVGF_F32;
    bool loopback = false;                 // SVGF_TRANSPORT_RCCL_LOOPBACK: one communicator of size 1, every peer is its rank 0, one shared communication stream
    bool mailbox = false;                  // SVGF_TRANSPORT_MAILBOX (tests): sends and receives matched in this process (see the head of the file)
    struct Posted { int rank, peer; char* buf; size_t bytes; };
    std::vector<Posted> mb_sends, mb_recvs;   // the open group, in posting order
    bool mb_open = false;
    int mb_fault_rank = -1, mb_fault = 0;  // svgf_strips_mailbox_fault: the next message of that kind posted by that rank is dropped / resized (tests of the matching itself)
    unsigned long long mb_groups = 0, mb_copies = 0, mb_bytes = 0;   // what the mailbox has matched so far (svgf_strips_transport_stats)
    bool broken = false;                   // a transport call failed inside a group: every later frame is refused
    struct Local {
        int rank = 0, device = 0;
        svgf_ctx* ctx = nullptr;
        ncclComm_t comm = nullptr;
        hipStream_t compute = nullptr, comm_stream = nullptr;
        bool own_comm_stream = false;
        // svgf_strips_set_frames_in_flight(2): iterations 1.. of a frame run on `side` beside the next frame's temporal launch; `cur` is the stream
        // the launches, the exchanges' ready records and their waits go to at the moment (compute, or side for a frame's tail)
        hipStream_t side = nullptr, cur = nullptr;
        hipEvent_t ev_first = nullptr, ev_tail = nullptr;     // iteration 0 of the frame being enqueued is on `compute`; the end of the tail in flight on `side`
        bool tail_pending = false;                            // ... which `compute` has not been made to wait for yet
        void* filter_alt[2] = {nullptr, nullptr};
        hipEvent_t ready = nullptr, halo_done = nullptr, state_done = nullptr;
        hipEvent_t mb_ready = nullptr, mb_done = nullptr;      // mailbox: this rank's communication stream has reached the group / has received what the group sends it
        // edge rows first (svgf_strips_set_edge_first): the iteration in front of an exchange is ONE launch whose first workgroups produce the rows the
        // neighbours wait for; the last of them writes edge_value into its slot's signal word and the communication stream waits for that word
        // (hipStreamWaitValue64) instead of for an event behind two extra launches
        // Two slots: launches on the filter stream use slot 0, launches on the side stream (two frames in flight: a frame's tail) slot 1 —
        // the two streams run CONCURRENTLY, and a word (or an arrival counter) shared between them is written out of order: the wait for the
        // smaller sequence number passes early and the one for the larger never (round 5's first version hung exactly there).
        // The two signal words are HSA signal memory (hipExtMallocWithFlags(hipMallocSignalMemory): what hipStreamWaitValue64 is documented for; plain
        // device memory works on this ROCm build too — tools/ubench/wait_value.hip — but is not promised), the arrival counters plain device memory.
        // No signal memory: edge_signal stays null and every exchanging iteration keeps the three-launch schedule.
        unsigned long long* edge_signal[2] = {nullptr, nullptr};
        unsigned* edge_arrivals = nullptr;                     // two counters, 256 B apart
        unsigned long long edge_value[2] = {0, 0};
        int edge_slot = 0;                                     // the slot of the launch just enqueued
        bool edge_pending = false;                             // the launch just enqueued signals: the next exchange waits for edge_value[edge_slot]
        bool state_pending = false;
        // the host never runs more than kMaxAhead frames ahead of the device: frame f waits for the end of frame f - kMaxAhead.  With ~100
        // frames of launches, events and RCCL groups queued the device starts to starve (0.43 -> 0.6 ms per 8K/8 strip, tools/strip_sim.py)
        std::vector<hipEvent_t> frame_done;
        svgf_strip_plan_geo g;
        // timing of the a-trous launches (bench.py's roofline block at N > 1)
        std::vector<hipEvent_t> tev;           // pairs
        std::vector<double> tbytes_px;         // pixels x (1 + feedback) weight per pair: (rows*W, iteration)
        std::vector<int> titer;
    };
    std::vector<Local> local;
    int timing_every = 0, timing_base = 0, frame_no = 0;       // timed: frames timing_base, timing_base + every, ...
    int frames_in_flight = 1;
    bool edge_first = false;               // svgf_strips_set_edge_first (opt-in, svgf_ext.h)
    double t_ms = 0, t_px_iter = 0, t_px_fb = 0;
    int t_launches = 0;
    std::string err;
};

namespace {

int sfail(svgf_strips* s, int code, const std::string& m) { if (s) s->err = m; return code; }
#define SVGF_SHIP(s, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return sfail((s), SVGF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
inline std::string nccl_text(int e) { Rccl* R = rccl(); return R->GetErrorString ? std::string(R->GetErrorString(e)) : "RCCL error " + std::to_string(e); }
#define SVGF_NCCL(s, call) do { int e_ = (call); if (e_ != ncclSuccess) return sfail((s), SVGF_ERR_COMM, std::string(#call) + ": " + nccl_text(e_)); } while (0)

std::vector<std::vector<int>> plan_groups(int plan, int n) {
    std::vector<std::vector<int>> g;
    if (n <= 0) return g;
    if (plan == SVGF_PLAN_GHOST) { g.emplace_back(); for (int i = 0; i < n; i++) g[0].push_back(i); }
    else if (plan == SVGF_PLAN_GROUPED) {
        g.emplace_back(); for (int i = 0; i < std::min(3, n); i++) g[0].push_back(i);
        if (n > 3) { g.emplace_back(); for (int i = 3; i < n; i++) g[1].push_back(i); }
    } else for (int i = 0; i < n; i++) g.push_back({i});
    return g;
}

// svgf_amd/strips.py:Geometry.make restated (the CPU tests compare the two)
bool make_geo(int W, int H, int rank, int world, int steps, int plan, int moments_radius, int motion_reach, svgf_strip_plan_geo& g) {
    (void)W;
    g = svgf_strip_plan_geo();
    g.groups = plan_groups(plan, steps);
    g.own0 = (int)((long long)H * rank / world); g.own1 = (int)((long long)H * (rank + 1) / world);
    g.ext_atrous.assign(steps, 0);
    for (auto& grp : g.groups) {
        int hsum = 0;
        for (int i : grp) { int e = 0; for (int k : grp) if (k > i) e += 2 << k; g.ext_atrous[i] = e; hsum += 2 << i; }
        g.halo_group.push_back(hsum);
    }
    g.ext_moments = g.groups.empty() ? 0 : g.halo_group[0];
    g.ext_temporal = g.ext_moments + moments_radius;
    g.halo_state = g.ext_temporal + motion_reach;
    g.halo_max = g.halo_state;
    for (int h : g.halo_group) g.halo_max = std::max(g.halo_max, h);
    g.y0 = std::max(0, g.own0 - g.halo_max); g.y1 = std::min(H, g.own1 + g.halo_max);
    int smallest = H;
    for (int r = 0; r < world; r++) smallest = std::min(smallest, (int)((long long)H * (r + 1) / world) - (int)((long long)H * r / world));
    return world <= 1 || smallest >= g.halo_max;
}

struct Rows { int a, b; };
Rows grown(const svgf_strip_plan_geo& g, int H, int ext) { return Rows{std::max(0, g.own0 - ext), std::min(H, g.own1 + ext)}; }


}  // namespace


namespace {

size_t row_bytes(const svgf_strips* s, int plane) {
    const size_t px = (size_t)s->W;
    switch (plane) {
        case SVGF_PLANE_COLOUR: case SVGF_PLANE_FILTER: return px * (s->storage == SVGF_F16 ? 8 : 16);
        case SVGF_PLANE_MOMENTS: return px * (s->storage == SVGF_F16 ? 4 : 8);
        default: return px;
    }
}

// ------------------------------------------------------------------ what an exchange consists of ------------
// The messages rank `rank` posts for ONE exchange, in posting order: for every plane, the rows at distance [held, h) from each of the strip's
// two boundaries — towards the upper neighbour first (send, then receive), then towards the lower one.  Rows nearer than `held` the receiver has
// computed itself (redundantly, bit-identically).  A pure function of the partition: svgf_strips_messages lists a frame's messages with it
// (tests/test_strips_cpu.py pairs every send of every rank with its neighbour's receive), and post_exchange posts exactly these.
//   rank b sends its rows [own1 - h, own1 - held) down and receives [own1 + held, own1 + h);  sends [own0 + held, own0 + h) up, receives [own0 - h, own0 - held)
// Between one pair of ranks the sends of one side and the receives of the other come in the same plane order: RCCL matches them in posting order.
struct MsgSpec { bool send; int peer; int plane, index; int g0, g1; };
struct PlaneSpec { int plane, index, held; };

void exchange_msgs(int H, int world, int rank, const std::vector<PlaneSpec>& planes, int h, std::vector<MsgSpec>& out) {
    const int own0 = (int)((long long)H * rank / world), own1 = (int)((long long)H * (rank + 1) / world);
    for (const PlaneSpec& pl : planes) {
        if (h <= pl.held) continue;
        if (rank > 0) {
            out.push_back(MsgSpec{true, rank - 1, pl.plane, pl.index, own0 + pl.held, own0 + h});
            out.push_back(MsgSpec{false, rank - 1, pl.plane, pl.index, own0 - h, own0 - pl.held});
        }
        if (rank + 1 < world) {
            out.push_back(MsgSpec{true, rank + 1, pl.plane, pl.index, own1 - h, own1 - pl.held});
            out.push_back(MsgSpec{false, rank + 1, pl.plane, pl.index, own1 + pl.held, own1 + h});
        }
    }
}

// the planes of the state exchange (posted once iteration 0 has written the feedback colour) — only rows a rank has NOT computed itself travel: it
// holds the feedback colour ext_atrous[0] rows beyond its strip and moments / history ext_temporal rows beyond (bit-identical to the owner's)
std::vector<PlaneSpec> state_planes(const svgf_strip_plan_geo& g, int steps, int P) {
    const int colour_held = steps ? g.ext_atrous[0] : g.ext_temporal;
    return {{SVGF_PLANE_COLOUR, P, colour_held}, {SVGF_PLANE_MOMENTS, P, g.ext_temporal}, {SVGF_PLANE_HISTORY, P, g.ext_temporal}};
}

// ------------------------------------------------------------------ transports ------------------------------
struct Transport {
    const char* name;
    int (*group_start)(svgf_strips*);
    int (*send)(svgf_strips*, svgf_strips::Local& from, const void* buf, size_t bytes, int peer);
    int (*recv)(svgf_strips*, svgf_strips::Local& to, void* buf, size_t bytes, int peer);
    int (*group_end)(svgf_strips*);      // SVGF_OK: every transfer of the group is enqueued on the communication streams of its two ends
    void (*abandon)(svgf_strips*);       // a call between group_start and group_end failed: leave nothing half open
};

// -- RCCL: the product transport.  Loop-back: one communicator of size 1 — every peer is its rank 0.
int rccl_group_start(svgf_strips* s) {
    Rccl* R = rccl();
    if (!R->Send) return sfail(s, SVGF_ERR_COMM, R->why.empty() ? "librccl not available" : R->why);
    SVGF_NCCL(s, R->GroupStart());
    return SVGF_OK;
}
int rccl_send(svgf_strips* s, svgf_strips::Local& from, const void* buf, size_t bytes, int peer) {
    if (int e = rccl()->Send(buf, bytes, ncclInt8, s->loopback ? 0 : peer, from.comm, from.comm_stream); e != ncclSuccess) return sfail(s, SVGF_ERR_COMM, "ncclSend: " + nccl_text(e));
    return SVGF_OK;
}
int rccl_recv(svgf_strips* s, svgf_strips::Local& to, void* buf, size_t bytes, int peer) {
    if (int e = rccl()->Recv(buf, bytes, ncclInt8, s->loopback ? 0 : peer, to.comm, to.comm_stream); e != ncclSuccess) return sfail(s, SVGF_ERR_COMM, "ncclRecv: " + nccl_text(e));
    return SVGF_OK;
}
int rccl_group_end(svgf_strips* s) { SVGF_NCCL(s, rccl()->GroupEnd()); return SVGF_OK; }
void rccl_abandon(svgf_strips*) { (void)rccl()->GroupEnd(); }      // (whatever it returns: the thread's RCCL state must not stay half open)
const Transport kRccl{"rccl", rccl_group_start, rccl_send, rccl_recv, rccl_group_end, rccl_abandon};

// -- mailbox: the test transport (head of the file)
svgf_strips::Local* local_of(svgf_strips* s, int rank) { for (auto& l : s->local) if (l.rank == rank) return &l; return nullptr; }
int mb_group_start(svgf_strips* s) {
    if (s->mb_open) return sfail(s, SVGF_ERR_COMM, "mailbox: group started inside a group");
    s->mb_sends.clear(); s->mb_recvs.clear();
    s->mb_open = true;
    return SVGF_OK;
}
int mb_post(svgf_strips* s, std::vector<svgf_strips::Posted>& box, int rank, const void* buf, size_t bytes, int peer, const char* what) {
    if (!s->mb_open) return sfail(s, SVGF_ERR_COMM, std::string("mailbox: ") + what + " outside a group");
    if (peer < 0 || peer >= s->world || peer == rank || !local_of(s, peer))
        return sfail(s, SVGF_ERR_COMM, std::string("mailbox: rank ") + std::to_string(rank) + " posts a " + what + " whose peer " + std::to_string(peer) + " is not a rank of this driver");
    if (s->mb_fault && s->mb_fault_rank == rank) {      // an injected defect of the schedule (svgf_strips_mailbox_fault): what group_end must catch
        const bool is_send = &box == &s->mb_sends;
        const int f = s->mb_fault;
        if ((f == SVGF_FAULT_DROP_SEND && is_send) || (f == SVGF_FAULT_DROP_RECV && !is_send)) { s->mb_fault = 0; return SVGF_OK; }
        if (f == SVGF_FAULT_SHORT_RECV && !is_send) { s->mb_fault = 0; bytes /= 2; }
    }
    box.push_back(svgf_strips::Posted{rank, peer, (char*)const_cast<void*>(buf), bytes});
    return SVGF_OK;
}
int mb_send(svgf_strips* s, svgf_strips::Local& from, const void* buf, size_t bytes, int peer) { return mb_post(s, s->mb_sends, from.rank, buf, bytes, peer, "send"); }
int mb_recv(svgf_strips* s, svgf_strips::Local& to, void* buf, size_t bytes, int peer) { return mb_post(s, s->mb_recvs, to.rank, buf, bytes, peer, "receive"); }
int mb_group_end(svgf_strips* s) {
    s->mb_open = false;
    // match: the k-th send of (src -> dst) to the k-th receive dst posted with peer src
    std::vector<int> taken(s->mb_recvs.size(), 0);
    struct Pair { const svgf_strips::Posted* snd; const svgf_strips::Posted* rcv; };
    std::vector<Pair> pairs;
    for (const auto& snd : s->mb_sends) {
        const svgf_strips::Posted* rcv = nullptr;
        for (size_t k = 0; k < s->mb_recvs.size(); k++)
            if (!taken[k] && s->mb_recvs[k].rank == snd.peer && s->mb_recvs[k].peer == snd.rank) { taken[k] = 1; rcv = &s->mb_recvs[k]; break; }
        if (!rcv) return sfail(s, SVGF_ERR_COMM, "mailbox: rank " + std::to_string(snd.rank) + " sends " + std::to_string(snd.bytes) + " bytes to rank " + std::to_string(snd.peer) +
                                                  ", which posts no receive for them in this group: a multi-GPU run would wait here for ever");
        if (rcv->bytes != snd.bytes) return sfail(s, SVGF_ERR_COMM, "mailbox: rank " + std::to_string(snd.rank) + " sends " + std::to_string(snd.bytes) + " bytes to rank " + std::to_string(snd.peer) +
                                                                    ", whose matching receive (posting order) expects " + std::to_string(rcv->bytes));
        pairs.push_back(Pair{&snd, rcv});
    }
    for (size_t k = 0; k < s->mb_recvs.size(); k++)
        if (!taken[k]) return sfail(s, SVGF_ERR_COMM, "mailbox: rank " + std::to_string(s->mb_recvs[k].rank) + " waits for " + std::to_string(s->mb_recvs[k].bytes) + " bytes from rank " +
                                                      std::to_string(s->mb_recvs[k].peer) + ", which sends none in this group: a multi-GPU run would wait here for ever");
    // every rank's communication stream has reached the group (post_exchange made it wait for the rows it sends and the rows it receives into)
    for (auto& l : s->local) { DeviceGuard dg(l.device); SVGF_SHIP(s, hipEventRecord(l.mb_ready, l.comm_stream)); }
    std::vector<char> receives(s->world, 0);
    for (const Pair& p : pairs) {
        svgf_strips::Local* src = local_of(s, p.snd->rank);
        svgf_strips::Local* dst = local_of(s, p.rcv->rank);
        DeviceGuard dg(dst->device);
        SVGF_SHIP(s, hipStreamWaitEvent(dst->comm_stream, src->mb_ready, 0));
        if (src->device == dst->device) SVGF_SHIP(s, hipMemcpyAsync(p.rcv->buf, p.snd->buf, p.snd->bytes, hipMemcpyDeviceToDevice, dst->comm_stream));
        else SVGF_SHIP(s, hipMemcpyPeerAsync(p.rcv->buf, dst->device, p.snd->buf, src->device, p.snd->bytes, dst->comm_stream));
        receives[dst->rank] = 1;
        s->mb_copies++; s->mb_bytes += p.snd->bytes;
    }
    for (auto& l : s->local) if (receives[l.rank]) { DeviceGuard dg(l.device); SVGF_SHIP(s, hipEventRecord(l.mb_done, l.comm_stream)); }
    // a send is complete when its buffer may be written again: the sender's stream waits for the copies out of it
    for (auto& l : s->local) {
        std::vector<char> waited(s->world, 0);
        for (const Pair& p : pairs) {
            if (p.snd->rank != l.rank || waited[p.rcv->rank]) continue;
            waited[p.rcv->rank] = 1;
            DeviceGuard dg(l.device);
            SVGF_SHIP(s, hipStreamWaitEvent(l.comm_stream, local_of(s, p.rcv->rank)->mb_done, 0));
        }
    }
    s->mb_groups++;
    return SVGF_OK;
}
void mb_abandon(svgf_strips* s) { s->mb_open = false; s->mb_sends.clear(); s->mb_recvs.clear(); }
const Transport kMailbox{"mailbox", mb_group_start, mb_send, mb_recv, mb_group_end, mb_abandon};

const Transport& transport_of(const svgf_strips* s) { return s->mailbox ? kMailbox : kRccl; }

// A call failed between group start and group end: close the group so that nothing is left half open, and mark the driver unusable — the
// events and transfers of this frame are in an unknown state.
int group_failed(svgf_strips* s, int rc) {
    const std::string why = s->err;
    transport_of(s).abandon(s);
    s->broken = true;
    return sfail(s, rc, why + " (the strip driver is unusable from here: destroy it)");
}

// Post ONE exchange for all local ranks: for every plane of `planes` the rows at distance [held, h) from each strip boundary (exchange_msgs).
// is_state selects which event the filter stream will wait for.
// in_order: the rows this exchange carries were final before the exchange posted last started (the communication stream runs in order: nothing to wait for)
int post_exchange(svgf_strips* s, const std::vector<PlaneSpec>& planes, int h, bool is_state, bool in_order = false) {
    const Transport& T = transport_of(s);
    // The transfers of a rank start when its filter stream has produced the rows it sends and is done with the halo rows it
    // receives into.  Loop-back: every virtual rank shares one communication stream, which then waits for all of them.
    // (An event record is a barrier packet on the filter stream — ~6 us between two launches that would otherwise run back to back,
    // profiles/r05_strip_trace_*.txt: it is only recorded where something will wait for it.)
    for (auto& l : s->local) {
        DeviceGuard dg(l.device);
        if (!in_order && !l.edge_pending) SVGF_SHIP(s, hipEventRecord(l.ready, l.cur));
    }
    // (edge rows first: the rows a rank sends are final when the first workgroups of the launch it has just enqueued have signalled — the
    // communication stream waits for that word, not for the launch; everything enqueued BEFORE that launch is complete by then, stream order)
    auto wait_for = [&](svgf_strips::Local& on, svgf_strips::Local& of) -> int {
        if (of.edge_pending) SVGF_SHIP(s, hipStreamWaitValue64(on.comm_stream, of.edge_signal[of.edge_slot], of.edge_value[of.edge_slot], hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
        else SVGF_SHIP(s, hipStreamWaitEvent(on.comm_stream, of.ready, 0));
        return SVGF_OK;
    };
    if (!in_order) for (auto& l : s->local) {
        DeviceGuard dg(l.device);
        if (!s->loopback) { if (int rc = wait_for(l, l); rc != SVGF_OK) return rc; }
        else if (&l == &s->local[0]) for (auto& m : s->local) { if (int rc = wait_for(l, m); rc != SVGF_OK) return rc; }
    }
    for (auto& l : s->local) l.edge_pending = false;
    if (int rc = T.group_start(s); rc != SVGF_OK) return rc;
    struct Post { svgf_strips::Local* l; MsgSpec m; };
    std::vector<Post> posts;
    std::vector<MsgSpec> msgs;
    for (auto& l : s->local) {
        msgs.clear();
        exchange_msgs(s->H, s->world, l.rank, planes, h, msgs);
        for (const MsgSpec& m : msgs) posts.push_back(Post{&l, m});
    }
    if (s->loopback) {
        // ONE communicator whose only peer is itself: RCCL matches ALL sends to ALL receives in posting order, so the sends and the receives are
        // each posted in the order of the messages they belong to, (source rank, destination rank), planes in plan order within a pair.  (With real
        // peers — RCCL or the mailbox — only the order within a pair of ranks matters, and exchange_msgs keeps that the same on both sides.)
        auto src_of = [](const Post& p) { return p.m.send ? p.l->rank : p.m.peer; };
        auto dst_of = [](const Post& p) { return p.m.send ? p.m.peer : p.l->rank; };
        std::stable_sort(posts.begin(), posts.end(), [&](const Post& a, const Post& b) {
            if (a.m.send != b.m.send) return a.m.send && !b.m.send;
            if (src_of(a) != src_of(b)) return src_of(a) < src_of(b);
            return dst_of(a) < dst_of(b);
        });
    }
    for (const Post& p : posts) {
        const MsgSpec& m = p.m;
        svgf_strips::Local& l = *p.l;
        const size_t rb = row_bytes(s, m.plane);
        char* base = (char*)svgf_state_plane(l.ctx, m.plane, m.index) + (size_t)(m.g0 - l.g.y0) * rb;
        const size_t bytes = (size_t)(m.g1 - m.g0) * rb;
        const int rc = m.send ? T.send(s, l, base, bytes, m.peer) : T.recv(s, l, base, bytes, m.peer);
        if (rc != SVGF_OK) return group_failed(s, rc);
    }
    if (int rc = T.group_end(s); rc != SVGF_OK) { s->broken = true; return rc; }
    for (auto& l : s->local) {
        DeviceGuard dg(l.device);
        SVGF_SHIP(s, hipEventRecord(is_state ? l.state_done : l.halo_done, l.comm_stream));
        if (is_state) l.state_pending = true;
    }
    return SVGF_OK;
}

int wait_exchange(svgf_strips* s, svgf_strips::Local& l, bool is_state) {
    DeviceGuard dg(l.device);
    SVGF_SHIP(s, hipStreamWaitEvent(l.cur, is_state ? l.state_done : l.halo_done, 0));
    if (is_state) l.state_pending = false;
    return SVGF_OK;
}

// pair: iterations 0 and 1 in one launch on `rows` (iteration 1's; iteration 0 and the feedback store cover 4 rows more either side)
// inner != nullptr: ONE launch over the two edge ranges [rows.a, inner->a), [inner->b, rows.b) — produced first and signalled — and the interior *inner
int launch_atrous_rows(svgf_strips* s, svgf_strips::Local& l, Rows rows, int src, int dst, int P, const svgf_gbuffer* cur, int i, bool pair = false, const Rows* inner = nullptr,
                       const Rows* left_out = nullptr) {
    if (rows.b <= rows.a) return SVGF_OK;
    svgf_ctx* c = l.ctx;
    DeviceGuard dg(l.device);
    c->rb = rows.a; c->re = rows.b;
    const bool timed = s->timing_every > 0 && ((s->frame_no - s->timing_base) % s->timing_every) == 0 && l.rank == s->local[0].rank;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        SVGF_SHIP(s, hipEventCreate(&e0));
        if (hipError_t e = hipEventCreate(&e1); e != hipSuccess) { (void)hipEventDestroy(e0); return sfail(s, SVGF_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
        if (hipError_t e = hipEventRecord(e0, l.cur); e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return sfail(s, SVGF_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e)); }
    }
    const void* guide = use_guide(c) ? c->guide : nullptr;
    int rc = SVGF_OK;
    if (inner) {
        svgf::AtrousRanges r{};
        auto add = [&](int a, int b) { if (b > a) { r.yb[r.n] = a; r.ye[r.n] = b; r.n++; } };
        add(rows.a, inner->a); add(left_out ? left_out->b : inner->b, rows.b);        // (left_out: interior rows [inner->b, left_out->b) go into the NEXT launch)
        r.nfirst = r.n;
        add(inner->a, inner->b);
        const int slot = l.cur == l.compute ? 0 : 1;
        r.signal = l.edge_signal[slot]; r.arrivals = l.edge_arrivals + 64 * slot; r.value = ++l.edge_value[slot];
        rc = atrous_ranges_impl(c, c->filter[src], c->filter[dst], i == 0 ? c->colour[P] : nullptr, cur, 1 << i, i, guide, r);
        if (rc == SVGF_OK) { l.edge_pending = r.nfirst > 0; l.edge_slot = slot; }
    } else rc = pair ? atrous_pair_impl(c, c->filter[src], c->filter[dst], c->colour[P], cur, guide)
                     : atrous_impl(c, c->filter[src], c->filter[dst], i == 0 ? c->colour[P] : nullptr, cur, 1 << i, i, guide);
    if (rc != SVGF_OK) {
        if (timed) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
        return sfail(s, rc, c->err);
    }
    if (timed) {
        if (hipError_t e = hipEventRecord(e1, l.cur); e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return sfail(s, SVGF_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e)); }
        l.tev.push_back(e0); l.tev.push_back(e1);
        // (rows of the interior that were left to the NEXT launch are that launch's: left_out)
        l.tbytes_px.push_back((double)((rows.b - rows.a) - (inner && left_out ? left_out->b - left_out->a : 0)) * s->W);
        l.titer.push_back(pair ? -1 : i);
    }
    return SVGF_OK;
}

}  // namespace

extern "C" {

int svgf_rccl_unique_id(void* id128) {
    Rccl* R = rccl();
    if (!id128 || !R->GetUniqueId) return SVGF_ERR_COMM;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return SVGF_ERR_COMM;
    std::memcpy(id128, &id, sizeof(id));
    return SVGF_OK;
}

int svgf_rccl_comm_init(void** comm, int world, int rank, const void* id128, int device) {
    Rccl* R = rccl();
    if (!comm || !id128 || !R->CommInitRank) return SVGF_ERR_COMM;
    DeviceGuard dg(device);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (R->CommInitRank(&c, world, id, rank) != ncclSuccess) return SVGF_ERR_COMM;
    *comm = c;
    return SVGF_OK;
}

int svgf_rccl_comm_destroy(void* comm) {
    Rccl* R = rccl();
    if (!comm || !R->CommDestroy) return SVGF_ERR_COMM;
    return R->CommDestroy((ncclComm_t)comm) == ncclSuccess ? SVGF_OK : SVGF_ERR_COMM;
}

int svgf_rccl_comm_count(void* comm, int* count) {
    Rccl* R = rccl();
    if (!comm || !count || !R->CommCount) return SVGF_ERR_COMM;
    return R->CommCount((ncclComm_t)comm, count) == ncclSuccess ? SVGF_OK : SVGF_ERR_COMM;
}

int svgf_strips_plan(int width, int height, int rank, int world, int steps, int plan, int moments_radius, int motion_reach, svgf_strip_layout* out) {
    if (!out || width <= 0 || height <= 0 || world < 1 || rank < 0 || rank >= world || steps < 0 || steps > SVGF_MAX_STEPS || motion_reach < 0 ||
        moments_radius < 0 || moments_radius > 3) return SVGF_ERR_INVALID;            // (the radius svgf_create accepts)
    svgf_strip_plan_geo g;
    int chosen = plan;
    if (plan == SVGF_PLAN_AUTO) {
        // the plan with the fewest exchanges BETWEEN iterations — but at least one: what BASELINE.json configs[3] names is a halo exchange between
        // a-trous iterations — whose halo fits the strips: grouped (one), else per-iteration (one in front of every iteration).  Grouped is also the
        // fastest of the three on an 8K/8 strip (profiles/r05_strip_sim_8k_over_8.txt: 6.19x against ghost 5.87x, per-iteration 5.71x).  Ghost (no
        // exchange between iterations, 62 ghost rows) needs the tallest halo of the three: it never fits where grouped does not; last, for completeness.
        const int order[3] = {SVGF_PLAN_GROUPED, SVGF_PLAN_PER_ITERATION, SVGF_PLAN_GHOST};
        bool ok = false;
        for (int cand : order) if (make_geo(width, height, rank, world, steps, cand, moments_radius, motion_reach, g)) { chosen = cand; ok = true; break; }
        if (!ok) return SVGF_ERR_HALO;
    } else {
        if (plan < SVGF_PLAN_GHOST || plan > SVGF_PLAN_PER_ITERATION) return SVGF_ERR_INVALID;
        if (!make_geo(width, height, rank, world, steps, plan, moments_radius, motion_reach, g)) return SVGF_ERR_HALO;
    }
    std::memset(out, 0, sizeof(*out));
    out->plan = chosen;
    out->strip = svgf_strip{g.y0, g.y1 - g.y0, g.own0, g.own1};
    out->ext_moments = g.ext_moments; out->ext_temporal = g.ext_temporal; out->halo_state = g.halo_state; out->halo_max = g.halo_max;
    out->ngroups = (int)g.groups.size();
    for (int i = 0; i < steps; i++) out->ext_atrous[i] = g.ext_atrous[i];
    for (size_t k = 0; k < g.groups.size(); k++) { out->halo_group[k] = g.halo_group[k]; out->group_first[k] = g.groups[k].front(); }
    return SVGF_OK;
}

// A frame's messages as rank `rank` posts them (exchange 0: the state, posted after iteration 0; exchange g >= 1: the filter rows in front of
// iteration group g), in posting order.  Pure geometry — no device needed.
int svgf_strips_messages(int width, int height, int rank, int world, int steps, int plan, int moments_radius, int motion_reach, int storage,
                         svgf_strip_message* out, int capacity, int* count) {
    if (!count || (capacity > 0 && !out) || (storage != SVGF_F32 && storage != SVGF_F16)) return SVGF_ERR_INVALID;
    svgf_strip_layout lay;
    int rc = svgf_strips_plan(width, height, rank, world, steps, plan, moments_radius, motion_reach, &lay);
    if (rc != SVGF_OK) return rc;
    svgf_strip_plan_geo g;
    make_geo(width, height, rank, world, steps, lay.plan, moments_radius, motion_reach, g);
    svgf_strips dims;
    dims.W = width; dims.storage = storage;
    int n = 0;
    auto emit = [&](int exchange, const std::vector<PlaneSpec>& planes, int h) {
        std::vector<MsgSpec> msgs;
        exchange_msgs(height, world, rank, planes, h, msgs);
        for (const MsgSpec& m : msgs) {
            if (n < capacity) out[n] = svgf_strip_message{exchange, m.send ? 1 : 0, m.peer, m.plane, m.g0, m.g1, (size_t)(m.g1 - m.g0) * row_bytes(&dims, m.plane)};
            n++;
        }
    };
    if (world > 1) {
        emit(0, state_planes(g, steps, 0), g.halo_state);
        for (size_t gi = 1; gi < g.groups.size(); gi++) emit((int)gi, {{SVGF_PLANE_FILTER, 0, 0}}, g.halo_group[gi]);
    }
    *count = n;
    return n <= capacity ? SVGF_OK : SVGF_ERR_INVALID;
}

int svgf_strips_transport_stats(const svgf_strips* s, unsigned long long* groups, unsigned long long* copies, unsigned long long* bytes) {
    if (!s || !s->mailbox) return SVGF_ERR_INVALID;
    if (groups) *groups = s->mb_groups;
    if (copies) *copies = s->mb_copies;
    if (bytes) *bytes = s->mb_bytes;
    return SVGF_OK;
}

int svgf_strips_mailbox_fault(svgf_strips* s, int rank, int fault) {
    if (!s || !s->mailbox || rank < 0 || rank >= s->world || fault < 0 || fault > SVGF_FAULT_SHORT_RECV) return SVGF_ERR_INVALID;
    s->mb_fault_rank = rank; s->mb_fault = fault;
    return SVGF_OK;
}

int svgf_strips_create(svgf_strips** out, int width, int height, int world, const svgf_params* params, int plan, int motion_reach,
                       int nlocal, const int* ranks, const int* devices, void* const* compute_streams, void* const* comms, int transport) {
    if (!out) return SVGF_ERR_INVALID;
    *out = nullptr;
    if (!params || nlocal < 1 || !ranks || !devices || world < 1 || nlocal > world) return SVGF_ERR_INVALID;
    if (transport < SVGF_TRANSPORT_RCCL || transport > SVGF_TRANSPORT_MAILBOX) return SVGF_ERR_INVALID;
    if (world > 1 && !comms && transport != SVGF_TRANSPORT_MAILBOX) return SVGF_ERR_INVALID;
    if (transport == SVGF_TRANSPORT_MAILBOX) {           // every rank of the partition lives here, once
        if (nlocal != world) return SVGF_ERR_INVALID;
        std::vector<char> seen(world, 0);
        for (int k = 0; k < nlocal; k++) { if (ranks[k] < 0 || ranks[k] >= world || seen[ranks[k]]) return SVGF_ERR_INVALID; seen[ranks[k]] = 1; }
    }
    {   // every device named must exist BEFORE anything is created on any of them
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return SVGF_ERR_NO_DEVICE; }
        for (int k = 0; k < nlocal; k++) if (devices[k] < 0 || devices[k] >= ndev) return SVGF_ERR_NO_DEVICE;
    }
    std::unique_ptr<svgf_strips> s(new (std::nothrow) svgf_strips());
    if (!s) return SVGF_ERR_ALLOC;
    s->W = width; s->H = height; s->world = world; s->steps = params->steps; s->motion_reach = motion_reach;
    s->moments_radius = params->moments_radius; s->storage = params->storage;
    s->loopback = transport == SVGF_TRANSPORT_RCCL_LOOPBACK; s->mailbox = transport == SVGF_TRANSPORT_MAILBOX;
    svgf_strip_layout lay;
    int rc = svgf_strips_plan(width, height, ranks[0], world, params->steps, plan, params->moments_radius, motion_reach, &lay);
    if (rc != SVGF_OK) return rc;
    s->plan = lay.plan;
    s->local.resize(nlocal);
    auto cleanup = [&]() { svgf_strips* p = s.release(); svgf_strips_destroy(p); };
    for (int k = 0; k < nlocal; k++) {
        auto& l = s->local[k];
        l.rank = ranks[k]; l.device = devices[k];
        if (l.rank < 0 || l.rank >= world) { cleanup(); return SVGF_ERR_INVALID; }
        make_geo(width, height, l.rank, world, params->steps, s->plan, params->moments_radius, motion_reach, l.g);
        l.compute = compute_streams ? (hipStream_t)compute_streams[k] : nullptr;
        l.cur = l.compute;
        l.comm = comms && !s->mailbox ? (ncclComm_t)comms[s->loopback ? 0 : k] : nullptr;
        svgf_strip st{l.g.y0, l.g.y1 - l.g.y0, l.g.own0, l.g.own1};
        rc = svgf_create_strip(&l.ctx, width, height, &st, params, l.device, l.compute);
        if (rc != SVGF_OK) { cleanup(); return rc; }
        l.ctx->strip_drv = reinterpret_cast<svgf_strip_driver*>(s.get());
        // previous-frame state is kept up to date own +- halo_state rows (computed here or received); the planes hold more
        // rows (the a-trous halos): the temporal stage must never take state from those
        l.ctx->vy0 = std::max(l.g.y0, l.g.own0 - l.g.halo_state); l.ctx->vy1 = std::min(l.g.y1, l.g.own1 + l.g.halo_state);
        DeviceGuard dg(l.device);
        hipError_t e = hipSuccess;
        if (s->loopback && k > 0) l.comm_stream = s->local[0].comm_stream;     // one communicator: one stream for its groups
        else {
            // at the highest priority: the exchange's kernels are a few workgroups that must find a slot while an a-trous launch, oversubscribed four
            // times, is being dispatched — behind a filter stream of higher priority they start when that grid has drained, i.e. when the transfer
            // should long be over (the exchange is then exposed in front of the next iteration)
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            e = hipStreamCreateWithPriority(&l.comm_stream, hipStreamNonBlocking, greatest);
            l.own_comm_stream = true;
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.halo_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.state_done, hipEventDisableTiming);
        if (e == hipSuccess && world > 1) {
            int can = 0;
            // (a part or runtime without stream memory operations keeps round 4's three launches per exchanging iteration)
            if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, l.device) == hipSuccess && can) {
                bool sig = true;
                for (int k = 0; k < 2 && sig; k++) {
                    sig = hipExtMallocWithFlags((void**)&l.edge_signal[k], 8, hipMallocSignalMemory) == hipSuccess && hipMemset(l.edge_signal[k], 0, 8) == hipSuccess;
                    if (!sig) (void)hipGetLastError();
                }
                if (!sig) {                                    // no signal memory here: the three-launch schedule (edge_signal[0] == null says so)
                    for (auto& p : l.edge_signal) { if (p) (void)hipFree(p); p = nullptr; }
                } else {
                    e = hipMalloc((void**)&l.edge_arrivals, 512);
                    if (e == hipSuccess) e = hipMemset(l.edge_arrivals, 0, 512);
                }
            }
        }
        if (e == hipSuccess && s->mailbox) e = hipEventCreateWithFlags(&l.mb_ready, hipEventDisableTiming);
        if (e == hipSuccess && s->mailbox) e = hipEventCreateWithFlags(&l.mb_done, hipEventDisableTiming);
        if (e != hipSuccess) { cleanup(); return SVGF_ERR_HIP; }
    }
    *out = s.release();
    return SVGF_OK;
}

void svgf_strips_destroy(svgf_strips* s) {
    if (!s) return;
    for (auto& l : s->local) {
        DeviceGuard dg(l.device);
        if (l.comm_stream) (void)hipStreamSynchronize(l.comm_stream);
        if (l.side) (void)hipStreamSynchronize(l.side);
        if (l.ctx) { (void)hipStreamSynchronize(l.compute); l.ctx->stream = l.compute; l.ctx->strip_drv = nullptr; svgf_destroy(l.ctx); }
        for (void* p : l.filter_alt) if (p) (void)hipFree(p);
        if (l.ev_first) (void)hipEventDestroy(l.ev_first);
        if (l.ev_tail) (void)hipEventDestroy(l.ev_tail);
        if (l.side) (void)hipStreamDestroy(l.side);
        for (auto e : l.tev) (void)hipEventDestroy(e);
        for (auto e : l.frame_done) if (e) (void)hipEventDestroy(e);
        if (l.ready) (void)hipEventDestroy(l.ready);
        if (l.halo_done) (void)hipEventDestroy(l.halo_done);
        if (l.state_done) (void)hipEventDestroy(l.state_done);
        for (void* p : l.edge_signal) if (p) (void)hipFree(p);
        if (l.edge_arrivals) (void)hipFree(l.edge_arrivals);
        if (l.mb_ready) (void)hipEventDestroy(l.mb_ready);
        if (l.mb_done) (void)hipEventDestroy(l.mb_done);
        if (l.own_comm_stream && l.comm_stream) (void)hipStreamDestroy(l.comm_stream);
    }
    delete s;
}

const char* svgf_strips_last_error(const svgf_strips* s) { return s ? s->err.c_str() : "null strip driver"; }

svgf_ctx* svgf_strips_context(svgf_strips* s, int local_index) {
    return s && local_index >= 0 && local_index < (int)s->local.size() ? s->local[local_index].ctx : nullptr;
}

int svgf_strips_layout(const svgf_strips* s, int local_index, svgf_strip_layout* out) {
    if (!s || !out || local_index < 0 || local_index >= (int)s->local.size()) return SVGF_ERR_INVALID;
    return svgf_strips_plan(s->W, s->H, s->local[local_index].rank, s->world, s->steps, s->plan, s->moments_radius, s->motion_reach, out);
}

// One frame on every local strip = application::Render's filter share (App.cu:552-556) per strip.
// radiance[k], cur[k], prev[k]: device planes of local rank k holding ITS rows [y0, y0+rows) (prev may be NULL on the first
// frame); results[k] receives the plane whose owned rows hold the result.
int svgf_strips_frame(svgf_strips* s, const void* const* radiance, const svgf_gbuffer* cur, const svgf_gbuffer* prev, const void** results) {
    if (!s || !radiance || !cur) return SVGF_ERR_INVALID;
    if (s->broken) return sfail(s, SVGF_ERR_COMM, "svgf_strips_frame: an earlier exchange failed; destroy this driver");
    const int n = (int)s->local.size();
    // state planes + scratch
    for (int k = 0; k < n; k++) {
        auto& l = s->local[k];
        svgf_ctx* c = l.ctx;
        DeviceGuard dg(l.device);
        if (!radiance[k]) return sfail(s, SVGF_ERR_INVALID, "svgf_strips_frame: null radiance");
        l.cur = l.compute; c->stream = l.compute;      // (a frame that failed behind its go_aside leaves them on the side stream)
        if (l.frame_done.empty()) l.frame_done.assign(kMaxAhead, nullptr);
        // the end of frame f - kMaxAhead (an event record is a barrier packet on the filter stream, ~6 us with nothing running: every kAheadStride-th frame carries one)
        if (s->frame_no % kAheadStride == 0) if (hipEvent_t old = l.frame_done[(s->frame_no / kAheadStride) % (kMaxAhead / kAheadStride)]) SVGF_SHIP(s, hipEventSynchronize(old));
        int rc0 = alloc_state(c);                 // svgf_denoise_frame's lazy allocation (exact size, zeroed)
        if (rc0 == SVGF_OK) rc0 = alloc_flags(c);
        if (rc0 != SVGF_OK) return sfail(s, rc0, c->err);
        if (s->frames_in_flight > 1) {
            // Two frames in flight: frames alternate between two pairs of filter planes (the context's pointers are swapped, so that every helper
            // keeps saying c->filter[]); the pair this frame takes over was last used by the frame before the previous one, whose tail the
            // filter stream was made to wait for in the previous call (go_aside).
            for (int i = 0; i < 2; i++) {
                if (!l.filter_alt[i]) {
                    SVGF_SHIP(s, hipMalloc(&l.filter_alt[i], colour_bytes(c)));
                    SVGF_SHIP(s, hipMemsetAsync(l.filter_alt[i], 0, colour_bytes(c), l.compute));
                }
                std::swap(c->filter[i], l.filter_alt[i]);
            }
        }
        // previous-frame state halo: posted by the PREVIOUS frame right after its iteration 0
        if (l.state_pending) { int rc = wait_exchange(s, l, true); if (rc != SVGF_OK) return rc; }
    }
    std::vector<int> pp(n, 0);
    for (int k = 0; k < n; k++) {
        auto& l = s->local[k];
        svgf_ctx* c = l.ctx;
        DeviceGuard dg(l.device);
        const int P = c->pingpong;
        const svgf_gbuffer* pv = prev ? &prev[k] : &cur[k];
        if (!pv->motion) pv = &cur[k];
        const Rows rt = grown(l.g, s->H, l.g.ext_temporal), rm = grown(l.g, s->H, l.g.ext_moments);
        c->rb = rt.a; c->re = rt.b;
        void* guide = use_guide(c) ? c->guide : nullptr;      // as svgf_denoise_frame: the temporal launch repacks {depth, ddepth, normal} for the iterations
        const void* guide_prev = prev_guide_for(c, &cur[k], pv);   // the previous frame's guide plane stands in for its G-buffer (all held rows)
        c->guide_prev_valid = false;                               // until this frame has written its own (commit_guide below)
        // which kernel serves the strip's young pixels (svgf_api.hip: every rank chooses for itself, the results do not depend on it)
        bool cold = false, crowded = false;
        choose_moments_kernel(c, &cold, &crowded);
        int rc = temporal_moments_impl(c, c->colour[1 - P], radiance[k], c->colour[P], c->filter[0], &cur[k], pv, c->hist[1 - P], c->hist[P],
                                       c->moments[P], c->moments[1 - P], rm.a, rm.b, s->steps >= 1 && !crowded, guide, guide_prev, cold || crowded);
        // (the temporal launch also writes the guide texels of the rows the strip holds beyond the temporal rows: the a-trous halos
        // of the later iteration groups and the next frame's reprojection read them)
        if (rc != SVGF_OK) return sfail(s, rc, c->err);
    }
    // The state exchange: final once iteration 0 has written the feedback colour, needed by the NEXT frame's temporal launch.  The communication stream
    // runs in order, so it is posted BEHIND the frame's last exchange of filter rows (in front of them it delayed the exchange the next iteration
    // waits for: 23 us of idle filter stream in the trace) — and needs no wait of its own there: an exchange posted for an iteration >= 1 started
    // after rows of that iteration were done, i.e. after iteration 0.  A frame without such an exchange posts it right behind iteration 0.
    int last_feed = -1;
    {
        const auto& gr = s->local[0].g.groups;
        for (size_t gi = 0; gi + 1 < gr.size(); gi++) last_feed = gr[gi].back();
        if (s->world <= 1) last_feed = -1;
    }
    auto post_state = [&](bool in_order) -> int {
        if (s->world <= 1) return SVGF_OK;
        const svgf_strip_plan_geo& g = s->local[0].g;
        const int P = s->local[0].ctx->pingpong;           // all local contexts advance together
        return post_exchange(s, state_planes(g, s->steps, P), g.halo_state, true, in_order);
    };
    // Two frames in flight: everything the NEXT frame's temporal launch reads is written once iteration 0 has stored the feedback colour and
    // the state exchange is posted; the remaining iterations (their exchanges included) go to the side stream.  Only when they read nothing
    // of the caller's: LDS launches on the guide plane (the direct kernel reads cur[k], which the caller may rewrite after this call).
    bool tail_ok = s->frames_in_flight > 1 && s->steps > 1;
    for (int k = 0; k < n && tail_ok; k++) {
        const svgf_ctx* c = s->local[k].ctx;
        tail_ok = use_guide(c) && c->p.variant != SVGF_VARIANT_DIRECT && c->p.phi_normal != 0.0f && (1 << (s->steps - 1)) <= 64;
    }
    bool aside = false;
    auto go_aside = [&]() -> int {
        if (!tail_ok || aside) return SVGF_OK;
        for (int k = 0; k < n; k++) {
            auto& l = s->local[k];
            DeviceGuard dg(l.device);
            SVGF_SHIP(s, hipEventRecord(l.ev_first, l.compute));
            // the frame that was on the side stream is ordered on the filter stream first: its result may be consumed after this call, its
            // pair of planes reused by the next one
            if (l.tail_pending) { SVGF_SHIP(s, hipStreamWaitEvent(l.compute, l.ev_tail, 0)); l.tail_pending = false; }
            SVGF_SHIP(s, hipStreamWaitEvent(l.side, l.ev_first, 0));
            l.cur = l.side;
            l.ctx->stream = l.side;
        }
        aside = true;
        return SVGF_OK;
    };
    const auto& groups = s->local[0].g.groups;
    // The exchange in front of iteration group g + 1 carries the rows within halo_group[g + 1] of every strip boundary of the LAST
    // iteration of group g.  That iteration therefore produces those rows FIRST (two edge launches), the exchange is posted behind
    // them, and the interior follows: the transfer runs beside the interior of the iteration that produced its rows — and has landed
    // when the next group's first iteration, which waits for it and then covers all of its rows in one launch, starts.  (Round 3 posted
    // the exchange after the WHOLE previous iteration and ran the next iteration's interior beside it: the tail of the producing
    // iteration and the head of the consuming one were serialised with the transfer.)
    bool posted = false;                              // an exchange of filter rows is in flight for the group about to start
    for (size_t gi = 0; gi < groups.size(); gi++) {
        for (size_t q = 0; q < groups[gi].size(); q++) {
            const int i = groups[gi][q];
            if (q == 0 && posted) {
                for (int k = 0; k < n; k++) { int rc = wait_exchange(s, s->local[k], false); if (rc != SVGF_OK) return rc; }
                posted = false;
            }
            // iterations 0 and 1 of one group: ONE launch on iteration 1's rows (svgf_atrous_pair); iteration 0 runs on 4 rows more
            // either side — grown(ext_atrous[0]) exactly — inside it
            if (i == 0 && q + 1 < groups[gi].size() && groups[gi][q + 1] == 1 && can_fuse01(s->local[0].ctx)) {
                for (int k = 0; k < n; k++) {
                    auto& l = s->local[k];
                    int rc = launch_atrous_rows(s, l, grown(l.g, s->H, l.g.ext_atrous[1]), pp[k], 1 - pp[k], l.ctx->pingpong, &cur[k], 0, true);
                    if (rc != SVGF_OK) return rc;
                    pp[k] ^= 1;
                }
                int rc = last_feed < 1 ? post_state(false) : SVGF_OK;
                if (rc == SVGF_OK) rc = go_aside();
                if (rc != SVGF_OK) return rc;
                q++;
                if (q + 1 == groups[gi].size() && gi + 1 < groups.size() && s->world > 1) {      // (a group of exactly {0, 1}: its output travels whole)
                    rc = post_exchange(s, {{SVGF_PLANE_FILTER, pp[0], 0}}, s->local[0].g.halo_group[gi + 1], false);
                    if (rc == SVGF_OK && last_feed == 1) rc = post_state(true);
                    if (rc != SVGF_OK) return rc;
                    posted = true;
                }
                continue;
            }
            const bool feeds_exchange = q + 1 == groups[gi].size() && gi + 1 < groups.size() && s->world > 1;
            // iteration 0 of a frame whose STATE exchange is posted right behind it (no exchange of filter rows later in the frame: the ghost plan): the rows
            // the state exchange carries — within halo_state of the boundaries — are produced first and signalled like the rows of any other exchange
            const bool feeds_state = i == 0 && last_feed < 1 && s->world > 1 && !feeds_exchange && s->edge_first;
            const int h = feeds_exchange ? s->local[0].g.halo_group[gi + 1] : feeds_state ? s->local[0].g.halo_state : 0;
            std::vector<Rows> inner(n, Rows{0, 0});
            bool split = feeds_exchange || feeds_state;
            if (split) {
                for (int k = 0; k < n; k++) {            // the rows the neighbours will need (the last iteration of a group runs on the owned rows)
                    auto& l = s->local[k];
                    const Rows rows = grown(l.g, s->H, l.g.ext_atrous[i]);
                    const int lo = l.rank == 0 ? rows.a : std::min(rows.b, l.g.own0 + h);
                    const int hi = l.rank == s->world - 1 ? rows.b : std::max(rows.a, l.g.own1 - h);
                    inner[k] = Rows{lo, std::max(lo, hi)};
                    if (hi <= lo) split = false;         // a strip shorter than its two edges: one launch, the exchange behind it
                }
            }
            // edge rows first, in ONE launch (round 5): the launch's first workgroups produce the two edge ranges and signal, the interior follows
            // in the same launch — instead of two edge launches, the exchange's event, and an interior launch (three launches' ramp and tail)
            std::vector<Rows> rest(n, Rows{0, 0});
            bool one_launch = split && s->edge_first;
            for (int k = 0; k < n && one_launch; k++) one_launch = atrous_ranges_ok(s->local[k].ctx, 1 << i) && s->local[k].edge_signal[0] != nullptr;
            for (int k = 0; k < n; k++) {
                auto& l = s->local[k];
                const Rows rows = grown(l.g, s->H, l.g.ext_atrous[i]);
                int rc = SVGF_OK;
                if (one_launch) {
                    // ... and only the first third of the interior: the rest is a launch of its own BEHIND the exchange's post.  An exchange's kernel
                    // (one RCCL workgroup: 132 registers per thread, 20 KB of LDS) finds no room beside a launch that oversubscribes every CU — whenever a
                    // filter workgroup retires, the next one takes its place — and so completed when the iteration drained, with the next iteration
                    // waiting behind it: ~20 us of idle filter stream per exchange (profiles/r05_strip_trace_*.txt).  At the boundary between the two
                    // launches the chip drains for a moment, the exchange gets its CU, and it has the second launch to finish in.
                    constexpr int kHeadPercent = 33;      // (20 / 33 / 50 measure the same within 1 %, 66 leaves the exchange too little time)
                    rest[k] = Rows{inner[k].a + (int)((long long)(inner[k].b - inner[k].a) * kHeadPercent / 100), inner[k].b};
                    if (feeds_state) rest[k] = Rows{inner[k].b, inner[k].b};      // (nobody waits for the state exchange before the next frame: one launch)
                    const Rows head{inner[k].a, rest[k].a};
                    rc = launch_atrous_rows(s, l, rows, pp[k], 1 - pp[k], l.ctx->pingpong, &cur[k], i, false, &head, &rest[k]);
                }
                else if (split) {
                    rc = launch_atrous_rows(s, l, Rows{rows.a, inner[k].a}, pp[k], 1 - pp[k], l.ctx->pingpong, &cur[k], i);
                    if (rc == SVGF_OK) rc = launch_atrous_rows(s, l, Rows{inner[k].b, rows.b}, pp[k], 1 - pp[k], l.ctx->pingpong, &cur[k], i);
                } else rc = launch_atrous_rows(s, l, rows, pp[k], 1 - pp[k], l.ctx->pingpong, &cur[k], i);
                if (rc != SVGF_OK) return rc;
            }
            if (feeds_exchange) {
                int rc = post_exchange(s, {{SVGF_PLANE_FILTER, 1 - pp[0], 0}}, h, false);
                if (rc == SVGF_OK && i == last_feed && i >= 1) rc = post_state(true);
                if (rc != SVGF_OK) return rc;
                posted = true;
            }
            if (one_launch) {
                split = false;                        // (the interior is part of that launch — and of the one behind it)
                for (int k = 0; k < n; k++) {
                    int rc = launch_atrous_rows(s, s->local[k], rest[k], pp[k], 1 - pp[k], s->local[k].ctx->pingpong, &cur[k], i);
                    if (rc != SVGF_OK) return rc;
                }
            }
            if (split) for (int k = 0; k < n; k++) {
                auto& l = s->local[k];
                int rc = launch_atrous_rows(s, l, inner[k], pp[k], 1 - pp[k], l.ctx->pingpong, &cur[k], i);
                if (rc != SVGF_OK) return rc;
            }
            for (int k = 0; k < n; k++) pp[k] ^= 1;
            if (i == 0) {                             // this frame's state is final once iteration 0 has written the feedback colour
                int rc = last_feed < 1 ? post_state(false) : SVGF_OK;
                if (rc == SVGF_OK) rc = go_aside();
                if (rc != SVGF_OK) return rc;
            }
        }
    }
    if (!s->steps) { int rc = post_state(false); if (rc != SVGF_OK) return rc; }
    for (int k = 0; k < n; k++) {
        auto& l = s->local[k];
        svgf_ctx* c = l.ctx;
        {
            DeviceGuard dg(l.device);
            if (s->frame_no % kAheadStride == 0) {
                hipEvent_t& done = l.frame_done[(s->frame_no / kAheadStride) % (kMaxAhead / kAheadStride)];
                if (!done) SVGF_SHIP(s, hipEventCreateWithFlags(&done, hipEventDisableTiming));
                SVGF_SHIP(s, hipEventRecord(done, l.cur));
            }
            if (l.cur != l.compute) {             // the end of this frame's tail on the side stream
                SVGF_SHIP(s, hipEventRecord(l.ev_tail, l.cur));
                l.tail_pending = true;
            } else if (l.tail_pending) {          // this frame never left the filter stream: the one in flight is ordered behind it now
                SVGF_SHIP(s, hipStreamWaitEvent(l.compute, l.ev_tail, 0));
                l.tail_pending = false;
            }
            l.cur = l.compute;
            c->stream = l.compute;
        }
        c->rb = c->strip.own_begin; c->re = c->strip.own_end;
        if (results) results[k] = c->filter[pp[k]];
        c->result_index = pp[k];
        commit_guide(c, &cur[k], use_guide(c));
        c->pingpong ^= 1;
        if (c->frames_since_reset < (1 << 30)) c->frames_since_reset++;
    }
    s->frame_no++;
    return SVGF_OK;
}

// Two frames in flight for the strips (svgf.h): 2 = iterations 1.. of a frame on a side stream of every local rank, beside the next frame's
// temporal launch; 1 = back to one frame at a time (waits for what is in flight).
int svgf_strips_set_frames_in_flight(svgf_strips* s, int frames) {
    if (!s) return SVGF_ERR_INVALID;
    if (frames != 1 && frames != 2) return sfail(s, SVGF_ERR_INVALID, "svgf_strips_set_frames_in_flight: 1 or 2");
    if (frames == s->frames_in_flight) return SVGF_OK;
    for (auto& l : s->local) {
        DeviceGuard dg(l.device);
        if (frames == 2) {
            if (!l.side) {                        // at the filter stream's priority (a default-priority stream beside a high-priority one is starved)
                int prio = 0;
                (void)hipStreamGetPriority(l.compute, &prio);
                SVGF_SHIP(s, hipStreamCreateWithPriority(&l.side, hipStreamNonBlocking, prio));
            }
            if (!l.ev_first) SVGF_SHIP(s, hipEventCreateWithFlags(&l.ev_first, hipEventDisableTiming));
            if (!l.ev_tail) SVGF_SHIP(s, hipEventCreateWithFlags(&l.ev_tail, hipEventDisableTiming));
        } else if (l.tail_pending) {              // back to one frame at a time: the filter stream waits for the tail in flight
            SVGF_SHIP(s, hipStreamWaitEvent(l.compute, l.ev_tail, 0));
            l.tail_pending = false;
        }
    }
    s->frames_in_flight = frames;
    return SVGF_OK;
}

int svgf_strips_set_edge_first(svgf_strips* s, int enable) {
    if (!s) return SVGF_ERR_INVALID;
    s->edge_first = enable != 0;
    return SVGF_OK;
}

// Wait for the exchange the last frame posted (before tearing the communicator down) and for the filter streams; reports
// SVGF_ERR_HALO if any local strip's temporal stage reprojected into rows it does not hold.
int svgf_strips_sync(svgf_strips* s) {
    if (!s) return SVGF_ERR_INVALID;
    unsigned long long total = 0;
    for (auto& l : s->local) {
        if (l.state_pending) { int rc = wait_exchange(s, l, true); if (rc != SVGF_OK) return rc; }
        DeviceGuard dg(l.device);
        unsigned long long n = 0;
        int rc = read_halo_violations(l.ctx, &n, 1);
        if (rc != SVGF_OK) return sfail(s, rc, l.ctx->err);
        if (l.tail_pending) { SVGF_SHIP(s, hipStreamWaitEvent(l.compute, l.ev_tail, 0)); l.tail_pending = false; }
        SVGF_SHIP(s, hipStreamSynchronize(l.compute));
        if (l.side) SVGF_SHIP(s, hipStreamSynchronize(l.side));
        SVGF_SHIP(s, hipStreamSynchronize(l.comm_stream));
        total += n;
    }
    if (total) return sfail(s, SVGF_ERR_HALO, "temporal stage: " + std::to_string(total) + " reprojection(s) reached beyond the state halo (motion_reach = " +
                                              std::to_string(s->motion_reach) + " rows): the strips differ from the whole frame there");
    return SVGF_OK;
}

int svgf_strips_timing_enable(svgf_strips* s, int every) {
    if (!s) return SVGF_ERR_INVALID;
    s->timing_every = every > 0 ? every : 0;
    s->timing_base = s->frame_no;                  // the first frame after this call is a timed one
    return SVGF_OK;
}

// -> launches timed on the first local rank, their summed milliseconds, and the pixels they covered (all iterations / iteration 0 only)
int svgf_strips_timing_read(svgf_strips* s, int* launches, double* ms, double* px_all, double* px_iter0) {
    if (!s || !launches || !ms || !px_all || !px_iter0) return SVGF_ERR_INVALID;
    auto& l = s->local[0];
    DeviceGuard dg(l.device);
    for (size_t i = 0; i + 1 < l.tev.size(); i += 2) {
        SVGF_SHIP(s, hipEventSynchronize(l.tev[i + 1]));
        float t = 0.f;
        SVGF_SHIP(s, hipEventElapsedTime(&t, l.tev[i], l.tev[i + 1]));
        s->t_ms += t; s->t_launches++;
        s->t_px_iter += l.tbytes_px[i / 2] * (l.titer[i / 2] < 0 ? 2.0 : 1.0);       // a pair launch covers its rows in two iterations
        if (l.titer[i / 2] <= 0) s->t_px_fb += l.tbytes_px[i / 2];
        (void)hipEventDestroy(l.tev[i]); (void)hipEventDestroy(l.tev[i + 1]);
    }
    l.tev.clear(); l.tbytes_px.clear(); l.titer.clear();
    *launches = s->t_launches; *ms = s->t_ms; *px_all = s->t_px_iter; *px_iter0 = s->t_px_fb;
    s->t_launches = 0; s->t_ms = 0; s->t_px_iter = 0; s->t_px_fb = 0;
    return SVGF_OK;
}

}  // extern "C"
