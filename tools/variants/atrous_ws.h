// The à-trous iteration with SPECIALISED WAVES (included by svgf_kernels.hip inside its namespaces when built with
// -DSVGF_WAVE_SPECIALISED=1; uses the helpers of the LDS-streaming kernel above it: the record layout, raw_load / commit_px, the tap
// arithmetic is the same, the results are bit-identical and the GPU parity suite passes with it as the library).
//
// STATUS: a measured alternative, NOT part of the product build.  On MI355X it is ~10 % SLOWER per launch than atrous_lds_kernel in
// every composition tried (4 compute + 1, 2 or 4 loader waves; tools/abn.sh WS1L1 / WS1 / WS44 in profiles/r02_atrous_ablations.txt):
// stamps without atomics show its compute waves 85 % of their time in the tap loop and its loaders 64 % waiting for them — the
// arithmetic of a workgroup, not its memory traffic, is then the longest chain, and three or four workgroups per CU (the loaders' wave
// slots and registers are taken from compute waves) hide less of it than the five of the kernel where every wave does everything.
//
// Why.  In atrous_lds_kernel every wave does everything: it requests the next ring rows, runs its 24 taps, waits for the rows,
// converts them into LDS records, and meets its siblings at two barriers per step.  In-kernel stamps (profiles/r02_stamps_*)
// show where a wave's time goes: 27 % stalled ISSUING its three to six buffer loads (the CU's memory pipe is backed up: an
// HBM-bound CU sustains ~10 B/cycle, /opt/skills/guides/MI355X_MICROARCH.md), 19 % in the tap loop, the rest waiting — for its
// rows or, at the barriers, for siblings that are themselves stalled in one of the two.  The arithmetic never runs in the shadow
// of the memory traffic of the same workgroup, only of other workgroups', and five workgroups per CU is all the registers give.
//
// Here a workgroup is 4 compute waves + 1 LOADER wave.  The loader is the only wave that reads global memory: per step it owns
// the two new ring rows (2 x (128 + 4S) texels = 3 load rounds of 64 lanes per row and plane), converts them and writes the LDS
// records; issue stalls and memory latency are its own business.  The compute waves only read LDS, do the taps and store
// the outputs.  There is no barrier after the prologue: the loader may overwrite the two oldest ring rows once every compute
// wave has consumed its taps of them — the first ten of a step, rows are walked oldest first — and a compute wave first reads
// the two newest rows with its last ten taps; each side signals with one LDS add and polls right before it needs the other
// (about a step of slack each way).  The prologue's six ring rows and the first refill are requested by four waves at once:
// one round of memory latency per workgroup instead of three.
#ifndef SVGF_WS_LOADERS
#define SVGF_WS_LOADERS 4                // loader waves per workgroup: SVGF_WS_SPLIT of them share a refill (one ring row each), the groups take the
#endif                                   // refills in turn, so each has its rows in flight for LOADERS / SPLIT steps
#ifndef SVGF_WS_SPLIT
#define SVGF_WS_SPLIT 2                  // 1: a loader stages both rows of its refills; 2: one row
#endif
constexpr int kWsTX = 128;               // columns per workgroup
constexpr int kWsCompute = 4;            // compute waves: wave w filters row group w >> 1, columns (w & 1) * 64 ..
constexpr int kWsLoaders = SVGF_WS_LOADERS, kWsSplit = SVGF_WS_SPLIT, kWsTurns = kWsLoaders / kWsSplit, kWsRowsPer = kRS / kWsSplit;
static_assert(kWsLoaders % kWsSplit == 0 && kRS % kWsSplit == 0, "loader waves");
constexpr int kWsThreads = (kWsCompute + kWsLoaders) * 64;
#ifndef SVGF_WS_WAVES
#define SVGF_WS_WAVES 6                 // waves per SIMD the kernel is compiled for (80 registers): three 8-wave workgroups per CU
#endif

template <int S> struct WsLds {
    static constexpr int WL = kWsTX + 4 * S;                 // staged columns per ring row
    static constexpr int RND = (WL + 63) / 64;               // load rounds of 64 lanes per ring row
    static constexpr size_t bytes = (size_t)kRing * WL * kRecBytes + (size_t)kRing * kWsTX * 4 + (kRing * 8 + 2 + 2 + kWsTurns) * sizeof(uint32_t);
};

template <int ST, int S>
__global__ __launch_bounds__(kWsThreads, SVGF_WS_WAVES) void atrous_ws_kernel(Geo g, AtrousArgs a, int band_rows, int nbands, int xgroup, int xrot) {
    constexpr int TX = kWsTX, WL = WsLds<S>::WL, RND = WsLds<S>::RND;
    constexpr int CB = ST == 0 ? 16 : 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* recA = (f32x4*)smem;                          // {r,g,b,variance} clamped
    f32x2* recL = (f32x2*)(recA + kRing * WL);           // {luminance, depth}
    f32x2* recN = recL + kRing * WL;                     // {(nx,ny) packed halfs, nz}
    float* recD = (float*)(recN + kRing * WL);           // ddepth of the workgroup's own columns (read once, when the row is a centre row)
    uint32_t* nflag = (uint32_t*)(recD + kRing * TX);    // [kRing][8]: a surface texel of this ring row differs from the reference normal
    uint32_t* nref = nflag + kRing * 8;
    uint32_t* sync = nref + 2;                           // {compute waves done with the two oldest rows: even steps, odd steps; refills completed by loader 0, 1, ..}
    // (two "done" counters: a compute wave signals step n+1 before it waits for refill n, so one cumulative counter would let a fast
    // wave's signal of step n+1 stand in for a slow wave's signal of step n; it cannot get as far as step n+2 before that refill)

#ifdef SVGF_STAMPS
    unsigned long long stamp_entry;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_entry) :: "memory");
#endif
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // tile order: as atrous_lds_kernel (XCD-aware groups of consecutive tiles; step 1 walks the frame bottom-up)
    const int xtiles = (g.W + TX - 1) / TX;
    const int ntiles = xtiles * nbands * S;
    const int wid = blockIdx.x >> 3;
    const int round = wid / xgroup;
    int v = (round * kXcds + ((blockIdx.x + xrot * round) & (kXcds - 1))) * xgroup + wid % xgroup;
    if (v >= ntiles) return;
    if ((SVGF_REVERSE_MASK / S) & 1) v = ntiles - 1 - v;
    const int x0 = (v % xtiles) * TX;
    const int band = (v / xtiles) % nbands;
    const int rv = v / (xtiles * nbands);
    const int nrows = g.ye - g.yb;
    const int nj = (nrows - rv + S - 1) / S;
    const int j0 = band * band_rows;
    if (j0 >= nj) return;
    const int j1 = min(nj, j0 + band_rows);
    const int ybase = g.yb + rv;

    const bool guided = a.guide != nullptr;
    const unsigned m_off = guided ? 0u : 8u, n_off = guided ? 8u : 0u, n_shift = guided ? 4u : 3u;
    const unsigned npx = (unsigned)g.rows * (unsigned)g.W;
    auto plane_rsrc = [&](bool rok) __attribute__((always_inline)) {
        PlaneRsrc r;
        r.colour = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, rok ? (int)(npx * CB) : 0, 0x00020000);
        r.motion = __builtin_amdgcn_make_buffer_rsrc(guided ? (void*)a.guide : (void*)a.motion, 0, rok ? (int)(npx * 16u) : 0, 0x00020000);
        r.normal = __builtin_amdgcn_make_buffer_rsrc(guided ? (void*)a.guide : (void*)a.normal, 0, rok ? (int)(npx << n_shift) : 0, 0x00020000);
        return r;
    };

    // ---- staging: a wave's share of two ring rows is RND texels per row, texel c = r*64 + lane of the row's WL columns
    typedef RawPx<ST, true> Px;
    struct Stage { Px px[kRS][RND]; };                       // the prologue's: two ring rows
    struct StageL { Px px[kWsRowsPer][RND]; };                // a loader's share of a refill
    auto col_offsets = [&](int r, unsigned& vc, unsigned& vm, unsigned& vn) __attribute__((always_inline)) {
        const int c = r * 64 + lane, x = x0 - 2 * S + c;
        const bool ok = c < WL && x >= 0 && x < g.W;
        vc = ok ? (unsigned)x * CB : kOob; vm = ok ? (unsigned)x * 16u + m_off : kOob; vn = ok ? ((unsigned)x << n_shift) + n_off : kOob;
    };
    auto stage_fetch = [&](int jn, auto& st, auto nrows_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < decltype(nrows_tag)::value; k++) {
            const int y = ybase + S * (jn + k), yl = y - g.y0;                               // scalar
            const bool rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
            const int srow = rok ? yl * g.W : 0;
            const PlaneRsrc rs = plane_rsrc(rok);
#pragma unroll
            for (int r = 0; r < RND; r++) {
                unsigned vc, vm, vn;
                col_offsets(r, vc, vm, vn);
                raw_load<ST, true>(st.px[k][r], rs, vc, vm, vn, srow, n_shift);
            }
        }
    };
    uint32_t ref01 = 0, refz = 0;
    auto stage_commit = [&](int sl, const auto& st, auto nrows_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < decltype(nrows_tag)::value; k++) {
            int so = sl + k; so = so >= kRing ? so - kRing : so;                             // scalar
            bool differs = false;
#pragma unroll
            for (int r = 0; r < RND; r++) {
                const int c = r * 64 + lane;
                if (c < WL) {
                    differs = commit_px<ST, true>(st.px[k][r], recA, recL, recN, so * WL + c, ref01, refz) || differs;
                    if (c >= 2 * S && c < 2 * S + TX) recD[so * TX + c - 2 * S] = __uint_as_float(st.px[k][r].zd.y);
                }
            }
            const bool wave_differs = __ballot(differs) != 0ull;
            if (lane < 8) nflag[so * 8 + lane] = (lane == 0 && wave_differs) ? 1u : 0u;      // a ring row has ONE stager here
        }
    };

    // ---- prologue: ring rows 0..5 = decimated rows j0-2 .. j0+3 by waves 1, 0, 2 (two rows each; wave 0 holds the workgroup's
    // reference texel: column x0 of row j0), requested at once: one round of memory latency
#ifdef SVGF_STAMPS
    const unsigned stamp_key = stamp_enter(wave, lane);
#endif
    if (t < kRing * 8) nflag[t] = 0u;
    if (t < 2 + kWsTurns) sync[t] = 0u;
    const bool loader = wave >= kWsCompute;
    {
        Stage st;
        const int pj = wave == 0 ? j0 : wave == 1 ? j0 - 2 : j0 + 2;
        if (wave <= 2) stage_fetch(pj, st, std::integral_constant<int, kRS>{});
        if (wave == 0 && lane == 2 * S) { nref[0] = st.px[0][0].n.x; nref[1] = st.px[0][0].n.y & 0xffffu; }   // 2S < 64: round 0
        __syncthreads();
        ref01 = nref[0]; refz = nref[1];
        if (wave <= 2) stage_commit(wave == 0 ? 2 : wave == 1 ? 0 : 4, st, std::integral_constant<int, kRS>{});
        __syncthreads();
    }

#ifdef SVGF_STAMPS
    // phase shares (tools/stamps.py): compute waves 0 = setup + first 14 taps, 1 = wait for the refill, 2 = last 10 taps + epilogue,
    // 3 = stores; loaders 4 = wait for the compute waves, 5 = wait for the rows + convert + LDS writes, 6 = signal + next requests
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t) :: "memory");
    const unsigned long long stamp_first = stamp_t;
    unsigned long long stamp_steps = 0;
#endif
    auto sync_signal = [&](int which) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_u32*)sync + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
    };
    auto sync_wait = [&](int which, unsigned target) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        for (int spin = 0; spin < (1 << 18); spin++) {             // bounded: a lost signal shows up as a wrong result, not as a hung device
            const unsigned seen = __builtin_amdgcn_readfirstlane(*((const volatile lds_u32*)sync + which));
            if ((int)(seen - target) >= 0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
    };

    if (!loader) {
        // ---- compute waves
        const int rg = wave >> 1;                            // scalar: row group (output row j + rg)
        const int col = (wave & 1) * 64 + lane;              // own column inside the tile
        const int gx = x0 + col;
        const unsigned vo_c = gx < g.W ? (unsigned)gx * CB : kOob;
        const float phi_n = a.phi_normal;
        int slot0 = 0;
        unsigned n = 0;
        for (int j = j0; j < j1; j += kRS, n++) {
            int rowbase[5];
    #pragma unroll
            for (int r = 0; r < 5; r++) { int sl = slot0 + rg + r; sl = sl >= kRing ? sl - kRing : sl; rowbase[r] = sl * WL + col; }
            int cslot = slot0 + rg + 2; cslot = cslot >= kRing ? cslot - kRing : cslot;
            // centre
            const f32x4 cA = recA[rowbase[2] + 2 * S];
            const f32x2 cL = recL[rowbase[2] + 2 * S], cN = recN[rowbase[2] + 2 * S];
            const float cdz = cL.y == kSkyZ ? 0.0f : recD[cslot * TX + col];                          // GetDepth: sky -> ddepth 0
            const f32x2 lzc = cL;
            const float ncz = cN.y;
            const uint32_t nc01 = __float_as_uint(cN.x);
            const float phi_l = a.phi_colour * sqrtf(fmaxf(0.0f, 1e-10f + cA.w));                     // :562
            const float il = fminf(hw_rcp(phi_l), 1e30f) * kLog2e;
            const float izb = hw_rcp(fmaxf(cdz, 1e-6f) * (float)S) * kLog2e;                          // :563
            const float iz[5] = {izb, izb * 0.70710678118654752f, izb * 0.5f, izb * 0.44721359549995794f, izb * 0.35355339059327376f};
            float sw = 1.0f;                                                                          // :567
            f32x2 srg = {cA.x, cA.y}, sbv = {cA.z, cA.w};                                             // :568
            const bool wave_has_surface = __ballot(cL.y != kSkyZ) != 0ull;

            float ebase[5];
            auto make_ebase = [&]() __attribute__((always_inline)) {
                const float lg = hw_log2(clamp01(fmaf(ncz, ncz, dot2_h2(nc01, nc01))));
                ebase[0] = fmaf(lg, phi_n, klog2(0, 1)); ebase[1] = fmaf(lg, phi_n, klog2(1, 1)); ebase[2] = fmaf(lg, phi_n, klog2(0, 2));
                ebase[3] = fmaf(lg, phi_n, klog2(1, 2)); ebase[4] = fmaf(lg, phi_n, klog2(2, 2));
            };
            // records [T0, T1) of the thread's 25 (row-major over its five ring rows) as a rolling pipeline: the LDS reads of record
            // t + D are issued before record t is consumed
            auto taps = [&](auto uni_tag, auto t0_tag, auto t1_tag) __attribute__((always_inline)) {
                constexpr bool UNI = decltype(uni_tag)::value;
                constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
                constexpr int D = SVGF_TAP_DEPTH > 0 ? SVGF_TAP_DEPTH : 1;
                f32x4 qA[25];
                f32x2 qL[25], qN[25];
                auto issue = [&](int tt) __attribute__((always_inline)) {
                    if (tt == 12) return;                                                             // the centre itself is no tap
                    const int r = tt / 5, c = tt % 5;
                    qA[tt] = recA[rowbase[r] + c * S];
                    qL[tt] = ((const volatile lds_f32x2*)recL)[rowbase[r] + c * S];
                    if (!UNI) qN[tt] = ((const volatile lds_f32x2*)recN)[rowbase[r] + c * S];
                };
    #pragma unroll
                for (int tt = T0; tt < T0 + D && tt < T1; tt++) issue(tt);
    #pragma unroll
                for (int tt = T0; tt < T1; tt++) {
                    if (tt + D < T1) issue(tt + D);
                    asm volatile("" ::: "memory");
                    if (tt == 10) sync_signal(n & 1);                    // ring rows 0 and 1 (records 0-9) are consumed: the loader may overwrite them
                    if (tt == 12) continue;
                    const int r = tt / 5, xx = tt % 5 - 2, yy = r - 2;
                    const int axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
                    const f32x4 A = qA[tt];
                    const f32x2 dlz = qL[tt] - lzc;
                    float e;
                    if constexpr (UNI) {
                        e = ebase[kernel_class(axx, ayy)];
                    } else {
                        const f32x2 N = qN[tt];
                        const float d = clamp01(fmaf(N.y, ncz, dot2_h2(__float_as_uint(N.x), nc01)));
                        e = fmaf(hw_log2(d), phi_n, klog2(axx, ayy));
                    }
                    e = fmaf(-fabsf(dlz.x), il, e);
                    e = fmaf(-fabsf(dlz.y), iz[len_class(xx, yy)], e);
                    const float w = hw_exp2(e);
                    const f32x2 ww = {w, w * w};
                    sw += w;                                                                          // :607
                    srg = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, srg);
                    sbv = __builtin_elementwise_fma(ww, (f32x2){A.z, A.w}, sbv);
                    asm volatile("" : "+v"(sw), "+v"(srg), "+v"(sbv) :: "memory");
                }
            };
            // uniform-normal fast path per segment: the flags of the ring rows a segment's taps and centres lie in — rows 0-3 (relative
            // to slot0) for records 0-14, rows 2-5 for records 15-24 (rows 4 and 5 are the ones the previous refill wrote)
            int frel = (lane >> 3) - slot0; frel = frel < 0 ? frel + kRing : frel;
            const bool flag_lane = lane < kRing * 8;
            using I0 = std::integral_constant<int, 0>;
            using IA = std::integral_constant<int, 15>;
            using IB = std::integral_constant<int, 25>;
            bool uni_a = false;
            if (wave_has_surface) {
                uni_a = !SVGF_NO_FASTPATH && __ballot(flag_lane && frel <= 3 && nflag[flag_lane ? lane : 0] != 0u) == 0ull;
                if (uni_a) { make_ebase(); taps(std::true_type{}, I0{}, IA{}); }
                else taps(std::false_type{}, I0{}, IA{});
            } else {
                sync_signal(n & 1);
            }
            SVGF_STAMP(0);
            if (n > 0) sync_wait(2 + (int)((n - 1) % kWsTurns), kWsSplit * ((n - 1) / kWsTurns + 1));
            SVGF_STAMP(1);                             // refill n-1 (ring rows 4, 5 of this step and their flags) is complete
            if (wave_has_surface) {
                const bool uni_b = !SVGF_NO_FASTPATH && __ballot(flag_lane && frel >= 2 && nflag[flag_lane ? lane : 0] != 0u) == 0ull;
                if (uni_b) { if (!uni_a) make_ebase(); taps(std::true_type{}, IA{}, IB{}); }
                else taps(std::false_type{}, IA{}, IB{});
            }

            float4 o;
            const bool sky = lzc.y == kSkyZ;
            if (sky) {
                o = make_float4(cA.x, cA.y, cA.z, cA.w);                                              // :554-558
            } else {
                const float inv = hw_rcp(sw);                                                         // sw >= 1
                o = make_float4(srg.x * inv, srg.y * inv, sbv.x * inv, sbv.y * (inv * inv));          // :615
            }
            SVGF_STAMP(2);
            if (j + rg < j1) {                                                                        // scalar
                const int srow = (ybase + S * (j + rg) - g.y0) * g.W;
                const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(npx * CB), 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_fb = __builtin_amdgcn_make_buffer_rsrc(a.feedback ? a.feedback : a.out, 0, a.feedback ? (int)(npx * CB) : 0, 0x00020000);
                if constexpr (ST == 0) {
                    const u32x4 raw = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, vo_c, srow * CB, SVGF_COLOUR_ST_AUX);             // :618
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_fb, sky ? kOob : vo_c, srow * CB, 0);                   // :619-622 (not for sky)
                } else {
                    const u32x2 raw = {pack_h2(o.x, o.y), pack_h2(o.z, o.w)};
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_out, vo_c, srow * CB, SVGF_COLOUR_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_fb, sky ? kOob : vo_c, srow * CB, 0);
                }
            }
            slot0 += kRS; if (slot0 >= kRing) slot0 -= kRing;
            SVGF_STAMP(3);
#ifdef SVGF_STAMPS
            stamp_steps++;
#endif
        }
#ifdef SVGF_STAMPS
        if (lane == 0) {
            for (int i = 0; i < 4; i++) stamp_add(wave, i, stamp_acc[i]);
            stamp_add(wave, 6, stamp_first - stamp_entry);
            stamp_add(wave, 7, stamp_t - stamp_entry);
            stamp_add(wave, 8, 1ull);
            stamp_add(wave, 10, stamp_steps);
        }
        stamp_leave(wave, lane, stamp_key);
#endif
    } else {
        // ---- a loader: refill n (during step n) replaces the two oldest ring rows by decimated rows j+4, j+5, if the band goes on.
        // Group `turn` of kWsSplit loaders takes the refills n = turn, turn + kWsTurns, ...: its rows are in flight while the other
        // groups' refills are consumed; inside a group every loader owns kWsRowsPer of the refill's rows.
        const int li = wave - kWsCompute, turn = li / kWsSplit, part = li % kWsSplit;
        using NR = std::integral_constant<int, kWsRowsPer>;
        StageL st;                                                 // (requested only now: a value live across the branch would be kept
        if (j0 + kRS * turn + kRS < j1) stage_fetch(j0 + kRS * turn + 4 + part * kWsRowsPer, st, NR{});   // alive through the compute waves' code)
        unsigned n = (unsigned)turn;
        for (int j = j0 + kRS * turn; j + kRS < j1; j += kRS * kWsTurns, n += kWsTurns) {
            int slot0 = (int)((kRS * n) % kRing) + part * kWsRowsPer;
            sync_wait((int)(n & 1), kWsCompute * ((n >> 1) + 1));  // every compute wave has consumed its taps of ring rows 0 and 1 of step n
            SVGF_STAMP(4);
            stage_commit(slot0, st, NR{});                         // (waits for the rows requested kWsTurns steps ago)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SVGF_STAMP(5);
            sync_signal(2 + turn);
            if (j + kRS * kWsTurns + kRS < j1) stage_fetch(j + kRS * kWsTurns + 4 + part * kWsRowsPer, st, NR{});
            SVGF_STAMP(6);
        }
#ifdef SVGF_STAMPS
        if (lane == 0) {
            stamp_add(wave, 4, stamp_acc[4]); stamp_add(wave, 5, stamp_acc[5]); stamp_add(wave, 13, stamp_acc[6]);
            stamp_add(wave, 14, stamp_t - stamp_entry);
            stamp_add(wave, 15, 1ull);
        }
#endif
    }
}

template <int ST, int S>
hipError_t launch_atrous_ws(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    constexpr size_t lds = WsLds<S>::bytes;
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(atrous_ws_kernel<ST, S>, lds, attr_done); e != hipSuccess) return e;
    constexpr int per_cu_lds = (int)((160 * 1024) / lds), per_cu_waves = 4 * (SVGF_WS_WAVES) / (kWsThreads / 64);
    constexpr int per_cu = per_cu_lds < per_cu_waves ? per_cu_lds : per_cu_waves;
    const int nrows = g.ye - g.yb;
    const int njmax = (nrows + S - 1) / S;
    const int xtiles = (g.W + kWsTX - 1) / kWsTX;
    int slots = per_cu * num_cus() * SVGF_OVERSUB;
#ifdef SVGF_DIAG
    slots = diag_env("SVGF_ATROUS_SLOTS", slots);
#endif
    int nbands = slots / (xtiles * S);
    if (nbands < 1) nbands = 1;
    int band = (njmax + nbands - 1) / nbands;
    if (band < SVGF_MIN_BAND) band = SVGF_MIN_BAND;
    band = (band + kRS - 1) / kRS * kRS;
    nbands = (njmax + band - 1) / band;
    const int xm = S <= 2 ? 16 : (S == 16 ? 2 : 1);
    const int xgroup = (xtiles * nbands * S + kXcds * xm - 1) / (kXcds * xm);
    const int ngroups = (xtiles * nbands * S + xgroup - 1) / xgroup;
    const dim3 grid((unsigned)((ngroups + kXcds - 1) / kXcds) * kXcds * xgroup);
    atrous_ws_kernel<ST, S><<<grid, dim3(kWsThreads), lds, s>>>(g, a, band, nbands, xgroup, 3);
    return hipGetLastError();
}
