// atrous_p2.h — the LDS-streaming à-trous kernel of svgf_atrous_lds.h with TWO output pixels per thread.  NOT the product kernel:
// compiled instead of it with -DSVGF_DIAG -DSVGF_P2 (atrous_lds_instrumented.h includes it), parity-green, 6-22 % SLOWER per launch
// (profiles/r03_small_experiments.txt): the ring in LDS caps the resident workgroups, and with two waves per workgroup that is 2.5-3 waves
// per SIMD - the kernel needs more waves than that to cover its barriers and its staging loads.  Same tiles, same ring, same records, same tap order per pixel (so the same bits); a workgroup is two
// waves instead of four and a thread filters its column of BOTH rows a step produces.  The two centres are one ring row apart: of the
// 6 x 5 records their windows cover, 20 serve both, so one pass over 30 records (ds_read_b128 + ds_read_b64 each) feeds 48 taps instead
// of 48 records; the LDS addresses, the barriers and the wave-wide tests are paid once per two pixels.  LDS still caps the resident
// workgroups (5-6 per CU), now 2.5-3 waves per SIMD with two independent pixel streams each and up to 168 registers per lane.
#pragma once
#include "../../svgf_amd/csrc/svgf_atrous_taps.h"

#ifndef SVGF_P2_DEPTH
#define SVGF_P2_DEPTH 4
#endif

namespace svgf {
namespace {

constexpr int kTX = 128;                 // columns of a workgroup = its threads
constexpr int kTapDepth = SVGF_P2_DEPTH; // LDS reads run this many records ahead of the arithmetic

// The taps of two centres (ring rows 2 and 3 of rowbase) in one pass over the 30 records; per centre the order of svgf_atrous_taps.h.
template <int CS, int D, bool UNI>
__device__ __forceinline__ void taps48(const f32x4* recA, const f32x2* recL, const f32x2* recN, const int (&rowbase)[6], const TapCentre (&c)[2], float phi_n,
                                       float (&sw)[2], f32x2 (&srg)[2], f32x2 (&sbv)[2], const UniBase* shared_base) {
    UniBase ub[2];
    if constexpr (UNI) {
        if (shared_base) { ub[0] = *shared_base; ub[1] = *shared_base; }
        else { ub[0] = uni_base(c[0].n01, c[0].nz, phi_n); ub[1] = uni_base(c[1].n01, c[1].nz, phi_n); }
    }
    constexpr int NT = 30;
    f32x4 qA[NT];
    f32x2 qL[NT], qN[NT];
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int r = t / 5, cc = t % 5;
        qA[t] = recA[rowbase[r] + cc * CS];
        qL[t] = ((const volatile lds_f32x2*)recL)[rowbase[r] + cc * CS];               // single ds_read_b64, see svgf_atrous_taps.h
        if (!UNI) qN[t] = ((const volatile lds_f32x2*)recN)[rowbase[r] + cc * CS];
    };
#pragma unroll
    for (int t = 0; t < D; t++) issue(t);
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t + D < NT) issue(t + D);
        asm volatile("" ::: "memory");
        const int r = t / 5, xx = t % 5 - 2;
        const f32x4 A = qA[t];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int yy = r - 2 - p;
            if (yy < -2 || yy > 2 || (yy == 0 && xx == 0)) continue;                    // outside this centre's window / the centre itself (:584)
            const int axx = xx < 0 ? -xx : xx, ayy = yy < 0 ? -yy : yy;
            const f32x2 dlz = qL[t] - c[p].lz;
            float e;
            if constexpr (UNI) {
                e = ub[p].e[kernel_class(axx, ayy)];
            } else {
                const f32x2 N = qN[t];
                const float d = clamp01(fmaf(N.y, c[p].nz, dot2_h2(__float_as_uint(N.x), c[p].n01)));
                e = fmaf(hw_log2(d), phi_n, klog2(axx, ayy));
            }
            e = fmaf(-fabsf(dlz.x), c[p].il, e);
            e = fmaf(-fabsf(dlz.y), c[p].iz[len_class(xx, yy)], e);
            const float w = hw_exp2(e);
            const f32x2 ww = {w, w * w};
            sw[p] += w;
            srg[p] = __builtin_elementwise_fma((f32x2){w, w}, (f32x2){A.x, A.y}, srg[p]);
            sbv[p] = __builtin_elementwise_fma(ww, (f32x2){A.z, A.w}, sbv[p]);
            asm volatile("" : "+v"(sw[p]), "+v"(srg[p]), "+v"(sbv[p]) :: "memory");
        }
    }
}

template <int ST, int S>
__global__ __launch_bounds__(kTX, 3) void atrous_lds_kernel(Geo g, AtrousArgs a, int band_rows, int nbands, int xgroup, int xrot) {
    constexpr int TX = kTX;
    constexpr int WL = TX + 4 * S;                 // staged columns per ring row
    constexpr int CB = ST == 0 ? 16 : 8;           // bytes per colour texel
    constexpr int NH = 4 * S;                      // halo pixels per ring row: wave w stages those of the step's row w (lanes 0..NH-1)
    static_assert(NH >= 1 && NH <= 64, "halo does not fit one wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* recA = (f32x4*)smem;
    f32x2* recL = (f32x2*)(recA + kRing * WL);
    f32x2* recN = recL + kRing * WL;
    uint32_t* nflag = (uint32_t*)(recN + kRing * WL);              // [kRing][8]
    uint32_t* nref = nflag + kRing * 8;                            // {(nx,ny) bits, nz bits}

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int col = t;
    const int wig = __builtin_amdgcn_readfirstlane(t >> 6);         // wave 0 / 1: also the row of a step whose halo it stages
    const int xtiles = (g.W + TX - 1) / TX;
    const int ntiles = xtiles * nbands * S;
    int v = xcd_tile(xgroup, xrot);
    if (v >= ntiles) return;
    if (S == 1) v = ntiles - 1 - v;
    const int x0 = (v % xtiles) * TX;
    const int band = (v / xtiles) % nbands;
    const int rv = v / (xtiles * nbands);
    const int nrows = g.ye - g.yb;
    const int nj = (nrows - rv + S - 1) / S;
    const int j0 = band * band_rows;
    if (j0 >= nj) return;
    const int j1 = min(nj, j0 + band_rows);
    const int ybase = g.yb + rv;

    const int gx = x0 + col;
    const int oli = col + 2 * S;
    const bool has_halo = lane < NH;
    const int hx = (lane < 2 * S) ? x0 - 2 * S + lane : x0 + TX + lane - 2 * S;
    const int hli = (lane < 2 * S) ? lane : TX + lane;
    const bool own_ok = gx < g.W, halo_ok = has_halo && hx >= 0 && hx < g.W;
    const GuideSel gs(a.guide != nullptr);
    const unsigned vo_c = own_ok ? (unsigned)gx * CB : kOob, vo_m = own_ok ? (unsigned)gx * 16u + gs.m_off : kOob, vo_n = own_ok ? ((unsigned)gx << gs.n_shift) + gs.n_off : kOob;
    const unsigned vh_c = halo_ok ? (unsigned)hx * CB : kOob, vh_m = halo_ok ? (unsigned)hx * 16u + gs.m_off : kOob, vh_n = halo_ok ? ((unsigned)hx << gs.n_shift) + gs.n_off : kOob;
    const unsigned npx = (unsigned)g.rows * (unsigned)g.W;

    // A thread's share of one staged step: its own pixel of rows jn and jn + 1, and a halo pixel of row jn + wig on lanes < NH.
    struct Staged { RawPx<ST, true> o[2]; RawPx<ST, false> h; };
    auto fetch = [&](int jn, Staged& st) __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < 2; rg++) {
            const int y = ybase + S * (jn + rg), yl = y - g.y0;                         // scalar
            const bool rok = y >= 0 && y < g.H && yl >= 0 && yl < g.rows;
            const int srow = rok ? yl * g.W : 0;
            const PlaneRsrc rs = plane_rsrc(a, npx, CB, gs.n_shift, rok);
            raw_load<ST, true>(st.o[rg], rs, vo_c, vo_m, vo_n, srow, gs.n_shift);
            if (wig == rg) raw_load<ST, false>(st.h, rs, vh_c, vh_m, vh_n, srow, gs.n_shift);
        }
    };
    uint32_t ref01 = 0, refz = 0;
    auto commit = [&](int sl, const Staged& st) __attribute__((always_inline)) {
        bool differs[2];
#pragma unroll
        for (int rg = 0; rg < 2; rg++) {
            int so = sl + rg; so = so >= kRing ? so - kRing : so;                       // scalar
            differs[rg] = commit_px<ST, true>(st.o[rg], recA, recL, recN, so * WL + oli, ref01, refz);
        }
        int soh = sl + wig; soh = soh >= kRing ? soh - kRing : soh;
        bool dh = false;
        if (has_halo) dh = commit_px<ST, false>(st.h, recA, recL, recN, soh * WL + hli, ref01, refz);
#pragma unroll
        for (int rg = 0; rg < 2; rg++) {
            int so = sl + rg; so = so >= kRing ? so - kRing : so;
            const bool wave_differs = wave_any(differs[rg] || (wig == rg && dh));
            if (lane == 0) nflag[so * 8 + wig] = wave_differs ? 1u : 0u;
        }
    };

    // ddepth of this thread's next centres: [row of the step][this step / the next one]
    float dq0[2] = {0.f, 0.f}, dq1[2] = {0.f, 0.f};
    if (t < kRing * 8) nflag[t] = 0u;
#pragma unroll 1
    for (int rr = 0; rr < kRing; rr += kRS) {
        const int r = rr == 0 ? 2 : (rr == 2 ? 0 : rr);
        Staged st;
        fetch(j0 - 2 + r, st);
        if (rr == 0) {
            if (t == 0) { nref[0] = st.o[0].n.x; nref[1] = st.o[0].n.y & 0xffffu; }
            __syncthreads();
            ref01 = nref[0]; refz = nref[1];
        }
        commit(r, st);
        if (r == 2) { dq0[0] = __uint_as_float(st.o[0].zd.y); dq0[1] = __uint_as_float(st.o[1].zd.y); }
        if (r == 4) { dq1[0] = __uint_as_float(st.o[0].zd.y); dq1[1] = __uint_as_float(st.o[1].zd.y); }
    }
    __syncthreads();

    const float phi_n = a.phi_normal;              // != 0 (launcher)
    const float inv_phi_c = hw_rcp(a.phi_colour) * kLog2e;
    UniBase ref_base = uni_base(ref01, unpack_h2(refz).x, phi_n);
#pragma unroll
    for (int k = 0; k < 5; k++) ref_base.e[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ref_base.e[k])));
    int slot0 = 0;
    Staged cs;
    for (int j = j0; j < j1; j += kRS) {
        const bool more = (j + kRS) < j1;
        if (more) fetch(j + kRS + 2, cs);

        // the centres are ring rows 2 and 3, their taps ring rows 0..4 and 1..5, columns oli-2S .. oli+2S
        int rowbase[6];
#pragma unroll
        for (int r = 0; r < 6; r++) { int sl = slot0 + r; sl = sl >= kRing ? sl - kRing : sl; rowbase[r] = sl * WL + col; }
        TapCentre c[2];
#pragma unroll
        for (int p = 0; p < 2; p++) { const int ci = rowbase[2 + p] + 2 * S; c[p] = centre_setup<S>(recA[ci], recL[ci], recN[ci], dq0[p], inv_phi_c); }
        const bool wave_has_surface = wave_any(!c[0].sky || !c[1].sky);
        const bool uniform = !a.no_fastpath && !wave_any(lane < kRing * 8 && nflag[lane < kRing * 8 ? lane : 0] != 0u);
        float sw[2] = {1.0f, 1.0f};
        f32x2 srg[2] = {{c[0].A.x, c[0].A.y}, {c[1].A.x, c[1].A.y}}, sbv[2] = {{c[0].A.z, c[0].A.w}, {c[1].A.z, c[1].A.w}};
        if (wave_has_surface) {
            if (uniform) taps48<S, kTapDepth, true>(recA, recL, recN, rowbase, c, phi_n, sw, srg, sbv, &ref_base);
            else taps48<S, kTapDepth, false>(recA, recL, recN, rowbase, c, phi_n, sw, srg, sbv, nullptr);
        }
        float4 o[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const float inv = hw_rcp(sw[p]);
            o[p] = make_float4(srg[p].x * inv, srg[p].y * inv, sbv[p].x * inv, sbv[p].y * (inv * inv));   // :615
        }
        const bool sky[2] = {c[0].sky, c[1].sky};

        if (more) {
            lds_barrier();                         // both waves are done reading the two oldest ring rows
            commit(slot0, cs);
#pragma unroll
            for (int p = 0; p < 2; p++) { dq0[p] = dq1[p]; dq1[p] = __uint_as_float(cs.o[p].zd.y); }
            slot0 += kRS; if (slot0 >= kRing) slot0 -= kRing;
            lds_barrier();
        }
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(npx * CB), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_fb = __builtin_amdgcn_make_buffer_rsrc(a.feedback ? a.feedback : a.out, 0, a.feedback ? (int)(npx * CB) : 0, 0x00020000);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            if (j + p < j1) {                                                            // scalar
                const int srow = (ybase + S * (j + p) - g.y0) * g.W;
                if constexpr (ST == 0) {
                    const u32x4 raw = {__float_as_uint(o[p].x), __float_as_uint(o[p].y), __float_as_uint(o[p].z), __float_as_uint(o[p].w)};
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_out, vo_c, srow * CB, 0);                   // :618
                    __builtin_amdgcn_raw_buffer_store_b128(raw, rs_fb, sky[p] ? kOob : vo_c, srow * CB, 0);    // :619-622 (not for sky)
                } else {
                    const u32x2 raw = {pack_h2(o[p].x, o[p].y), pack_h2(o[p].z, o[p].w)};
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_out, vo_c, srow * CB, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(raw, rs_fb, sky[p] ? kOob : vo_c, srow * CB, 0);
                }
            }
        }
    }
}

template <int ST, int S>
hipError_t launch_atrous_lds(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    constexpr int WL = kTX + 4 * S;
    constexpr size_t lds = (size_t)kRing * WL * kRecBytes + (kRing * 8 + 2) * sizeof(uint32_t);
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t e = allow_dynamic_lds(atrous_lds_kernel<ST, S>, lds, attr_done); e != hipSuccess) return e;
    constexpr int per_cu = (int)((160 * 1024) / lds);                  // LDS caps the resident (two-wave) workgroups
    const int nrows = g.ye - g.yb;
    const int njmax = (nrows + S - 1) / S;
    const int xtiles = (g.W + kTX - 1) / kTX;
    int nbands = per_cu * num_cus() * 4 / (xtiles * S);
    if (nbands < 1) nbands = 1;
    int band = (njmax + nbands - 1) / nbands;
    if (band < 8) band = 8;
    band = (band + kRS - 1) / kRS * kRS;
    nbands = (njmax + band - 1) / band;
    int xgroup;
    const dim3 grid = xcd_grid(xtiles * nbands * S, S <= 2 ? 16 : (S == 16 ? 2 : 1), xgroup);
    atrous_lds_kernel<ST, S><<<grid, dim3(kTX), lds, s>>>(g, a, band, nbands, xgroup, 3);
    return hipGetLastError();
}

template <int ST>
hipError_t launch_atrous_lds_step(const Geo& g, const AtrousArgs& a, hipStream_t s) {
    switch (a.step) {
        case 1: return launch_atrous_lds<ST, 1>(g, a, s);
        case 2: return launch_atrous_lds<ST, 2>(g, a, s);
        case 4: return launch_atrous_lds<ST, 4>(g, a, s);
        case 8: return launch_atrous_lds<ST, 8>(g, a, s);
        case 16: return launch_atrous_lds<ST, 16>(g, a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace
}  // namespace svgf
