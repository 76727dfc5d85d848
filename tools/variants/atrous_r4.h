// The à-trous iteration with FOUR decimated rows per step (included by svgf_kernels.hip inside its namespaces when built with
// -DSVGF_ROWS4=1; same records, same tap arithmetic as atrous_lds_kernel: bit-identical results).
//
// Why.  LDS, not registers, caps the resident waves of atrous_lds_kernel: a workgroup of 4 waves keeps a ring of 2 + 4 rows, three
// ring rows per output row — 5 workgroups per CU at steps 1-8, 4 at step 16.  With four output rows per step the ring is 4 + 4 rows
// (two per output row) for a workgroup of 8 waves: three workgroups per CU = 24 waves = six per SIMD at every step (80 registers),
// half as many barriers per output row, and bands twice as long for the same number of workgroups (row halo 1.19 instead of 1.31).
// The thread that stages a texel is no longer the one that filters it, so the centre's ddepth goes through LDS (recD).
//
// STATUS: a measured alternative, NOT part of the product build.  Parity-green; 4-5 % slower than atrous_lds_kernel at steps 1-8 and
// equal at step 16 (tools/abn.sh R4d1, profiles/r02_atrous_ablations.txt abn34): 80 registers leave the rolling tap pipeline one tap
// of read-ahead instead of three, and that costs more than the sixth wave per SIMD brings.
#ifndef SVGF_R4_TAP_DEPTH
#define SVGF_R4_TAP_DEPTH 3              // LDS reads of the rolling tap pipeline issued this many taps ahead
#endif
#ifndef SVGF_R4_WAVES
#define SVGF_R4_WAVES 6                  // waves per SIMD the kernel is compiled for: three 8-wave workgroups per CU
#endif
constexpr int kR4Rows = 4, kR4Ring = kR4Rows + 4, kR4TX = 128, kR4Threads = kR4TX * kR4Rows;

template <int S> struct R4Lds {
    static constexpr int WL = kR4TX + 4 * S;                 // staged columns per ring row
    static constexpr int HALF = (WL + 1) / 2;                // a wave stages one half of one new ring row per step
    static constexpr int RND = (HALF + 63) / 64;
    static constexpr size_t bytes = (size_t)kR4Ring * WL * kRecBytes + (size_t)kR4Ring * kR4TX * 4 + (kR4Ring * 8 + 2) * sizeof(uint32_t);
};

template <int ST, int S>
__global__ __launch_bounds__(kR4Threads, SVGF_R4_WAVES) void atrous_r4_kernel(Geo g, AtrousArgs a, int band_rows, int nbands, int xgroup, int xrot) {
    constexpr int TX = kR4TX, WL = R4Lds<S>::WL, HALF = R4Lds<S>::HALF, RND = R4Lds<S>::RND, KRS = kR4Rows, RING = kR4Ring;
    constexpr int CB = ST == 0 ? 16 : 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* recA = (f32x4*)smem;
    f32x2* recL = (f32x2*)(recA + RING * WL);
    f32x2* recN = recL + RING * WL;
    float* recD = (float*)(recN + RING * WL);            // ddepth of the workgroup's own columns
    uint32_t* nflag = (uint32_t*)(recD + RING * TX);     // [RING][8]: entries 0, 1 = the two waves that stage a ring row
    uint32_t* nref = nflag + RING * 8;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int rg = wave >> 1, half = wave & 1;           // scalar: output row j + rg, columns half * 64 .. ; stages ring row rg, half `half`
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

    // ---- staging: wave (rg, half) owns columns [half * HALF, min(WL, (half + 1) * HALF)) of the rg-th of a step's four new ring rows
    typedef RawPx<ST, true> Px;
    struct Stage { Px px[RND]; };
    unsigned vc[RND], vm[RND], vn[RND];
#pragma unroll
    for (int r = 0; r < RND; r++) {
        const int c = half * HALF + r * 64 + lane, x = x0 - 2 * S + c;
        const bool ok = r * 64 + lane < HALF && c < WL && x >= 0 && x < g.W;
        vc[r] = ok ? (unsigned)x * CB : kOob; vm[r] = ok ? (unsigned)x * 16u + m_off : kOob; vn[r] = ok ? ((unsigned)x << n_shift) + n_off : kOob;
    }
    auto stage_fetch = [&](int jrow, Stage& st) __attribute__((always_inline)) {
        const int y = ybase + S * jrow, yl = y - g.y0;                                       // scalar
        const bool rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
        const int srow = rok ? yl * g.W : 0;
        const PlaneRsrc rs = plane_rsrc(rok);
#pragma unroll
        for (int r = 0; r < RND; r++) raw_load<ST, true>(st.px[r], rs, vc[r], vm[r], vn[r], srow, n_shift);
    };
    uint32_t ref01 = 0, refz = 0;
    auto stage_commit = [&](int so, const Stage& st) __attribute__((always_inline)) {          // so: ring slot (scalar)
        bool differs = false;
#pragma unroll
        for (int r = 0; r < RND; r++) {
            const int c = half * HALF + r * 64 + lane;
            if (r * 64 + lane < HALF && c < WL) {
                differs = commit_px<ST, true>(st.px[r], recA, recL, recN, so * WL + c, ref01, refz) || differs;
                if (c >= 2 * S && c < 2 * S + TX) recD[so * TX + c - 2 * S] = __uint_as_float(st.px[r].zd.y);
            }
        }
        const bool wave_differs = __ballot(differs) != 0ull;
        if (lane == 0) nflag[so * 8 + half] = wave_differs ? 1u : 0u;
    };

    // ---- prologue: ring rows 0..7 = decimated rows j0-2 .. j0+5, two rounds of four rows (wave (rg, half) stages row rg of a round);
    // the workgroup's reference normal is column x0 of row j0 = ring row 2: wave (2, 0), lane 2S of its first round
    if (t < RING * 8) nflag[t] = 0u;
    {
        Stage s0, s1;
        stage_fetch(j0 - 2 + rg, s0);
        stage_fetch(j0 + 2 + rg, s1);
        if (rg == 2 && half == 0 && lane == 2 * S) { nref[0] = s0.px[0].n.x; nref[1] = s0.px[0].n.y & 0xffffu; }
        __syncthreads();
        ref01 = nref[0]; refz = nref[1];
        stage_commit(rg, s0);
        stage_commit(4 + rg, s1);
        __syncthreads();
    }

    const int col = half * 64 + lane;                    // own column inside the tile
    const int gx = x0 + col;
    const unsigned vo_c = gx < g.W ? (unsigned)gx * CB : kOob;
    const float phi_n = a.phi_normal;
    int slot0 = 0;
    Stage st;
    if (j0 + KRS < j1) stage_fetch(j0 + KRS + 2 + rg, st);               // the first refill: decimated rows j0+6 .. j0+9
    for (int j = j0; j < j1; j += KRS) {
        const bool more = j + KRS < j1;
        int rowbase[5];
#pragma unroll
        for (int r = 0; r < 5; r++) { int sl = slot0 + rg + r; sl = sl >= RING ? sl - RING : sl; rowbase[r] = sl * WL + col; }
        int cslot = slot0 + rg + 2; cslot = cslot >= RING ? cslot - RING : cslot;
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
        const bool uniform_normals = !SVGF_NO_FASTPATH && __ballot(lane < RING * 8 && nflag[lane < RING * 8 ? lane : 0] != 0u) == 0ull;

        float ebase[5];
        auto taps = [&](auto uni_tag) __attribute__((always_inline)) {
            constexpr bool UNI = decltype(uni_tag)::value;
            constexpr int D = SVGF_R4_TAP_DEPTH;
            if constexpr (UNI) {
                const float lg = hw_log2(clamp01(fmaf(ncz, ncz, dot2_h2(nc01, nc01))));
                ebase[0] = fmaf(lg, phi_n, klog2(0, 1)); ebase[1] = fmaf(lg, phi_n, klog2(1, 1)); ebase[2] = fmaf(lg, phi_n, klog2(0, 2));
                ebase[3] = fmaf(lg, phi_n, klog2(1, 2)); ebase[4] = fmaf(lg, phi_n, klog2(2, 2));
            }
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
            for (int tt = 0; tt < D; tt++) issue(tt);
#pragma unroll
            for (int tt = 0; tt < 25; tt++) {
                if (tt + D < 25) issue(tt + D);
                asm volatile("" ::: "memory");
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
        if (wave_has_surface) { if (uniform_normals) taps(std::true_type{}); else taps(std::false_type{}); }

        float4 o;
        const bool sky = lzc.y == kSkyZ;
        if (sky) {
            o = make_float4(cA.x, cA.y, cA.z, cA.w);                                              // :554-558
        } else {
            const float inv = hw_rcp(sw);                                                         // sw >= 1
            o = make_float4(srg.x * inv, srg.y * inv, sbv.x * inv, sbv.y * (inv * inv));          // :615
        }
        if (more) {
            lds_barrier();                                         // every wave is done reading the four oldest ring rows
            int so = slot0 + rg; so = so >= RING ? so - RING : so;
            stage_commit(so, st);
            slot0 += KRS; if (slot0 >= RING) slot0 -= RING;
            lds_barrier();
            if (j + 2 * KRS < j1) stage_fetch(j + 2 * KRS + 2 + rg, st);   // the rows of the step after the next, in flight during its taps
        }
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
    }
}

template <int ST, int S>
hipError_t launch_atrous_r4(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    constexpr size_t lds = R4Lds<S>::bytes;
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(atrous_r4_kernel<ST, S>, lds, attr_done); e != hipSuccess) return e;
    constexpr int per_cu_lds = (int)((160 * 1024) / lds), per_cu_waves = 4 * (SVGF_R4_WAVES) / (kR4Threads / 64);
    constexpr int per_cu = per_cu_lds < per_cu_waves ? per_cu_lds : per_cu_waves;
    const int nrows = g.ye - g.yb;
    const int njmax = (nrows + S - 1) / S;
    const int xtiles = (g.W + kR4TX - 1) / kR4TX;
    int slots = per_cu * num_cus() * SVGF_OVERSUB;
#ifdef SVGF_DIAG
    slots = diag_env("SVGF_ATROUS_SLOTS", slots);
#endif
    int nbands = slots / (xtiles * S);
    if (nbands < 1) nbands = 1;
    int band = (njmax + nbands - 1) / nbands;
    if (band < 2 * kR4Rows) band = 2 * kR4Rows;
    band = (band + kR4Rows - 1) / kR4Rows * kR4Rows;
    nbands = (njmax + band - 1) / band;
    const int xm = S <= 2 ? 16 : (S == 16 ? 2 : 1);
    const int xgroup = (xtiles * nbands * S + kXcds * xm - 1) / (kXcds * xm);
    const int ngroups = (xtiles * nbands * S + xgroup - 1) / xgroup;
    const dim3 grid((unsigned)((ngroups + kXcds - 1) / kXcds) * kXcds * xgroup);
    atrous_r4_kernel<ST, S><<<grid, dim3(kR4Threads), lds, s>>>(g, a, band, nbands, xgroup, 3);
    return hipGetLastError();
}
