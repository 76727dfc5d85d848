// Host-side C++ check of include/SVGF.h: drives gpupt::svgfDenoiser the way the reference's application::Render
// drives its filter stages (src/App.cu:552-556), on a flat grey wall, and checks hand-derivable answers.
// Built with g++ (no device code on the host side): see tests/test_cpp_shim.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "SVGF.h"

static uint16_t half_bits(float f) {   // exact for the few constants used here
    if (f == 0.0f) return 0;
    if (f == -1.0f) return 0xBC00;
    if (f == 1.0f) return 0x3C00;
    return 0;
}

int main() {
    const int W = 200, H = 120;
    try {
        gpupt::svgfDenoiser den(W, H, SVGF_F32);
        den.SpatialFilterSteps = 5;
        const size_t px = size_t(W) * H;
        std::vector<float> motion(px * 4, 0.0f);
        std::vector<uint16_t> normal(px * 4, 0), uv(px * 4, 0);
        for (size_t i = 0; i < px; i++) {
            motion[4 * i + 2] = 5.0f;      // depth
            motion[4 * i + 3] = 0.01f;     // ddepth
            normal[4 * i + 2] = half_bits(-1.0f);
            uv[4 * i + 3] = half_bits(1.0f);
        }
        gpupt::buffer dMotion(motion.size() * 4, motion.data()), dNormal(normal.size() * 2, normal.data()), dUV(uv.size() * 2, uv.data());
        svgf_gbuffer gb{dMotion.Data, dNormal.Data, dUV.Data};
        std::vector<float> radiance(px * 4);
        for (size_t i = 0; i < px; i++) { radiance[4 * i] = 0.25f; radiance[4 * i + 1] = 0.5f; radiance[4 * i + 2] = 0.75f; radiance[4 * i + 3] = 1.0f; }
        std::vector<float> out(px * 4);
        std::vector<uint8_t> hist(px);
        for (int frame = 0; frame < 6; frame++) {
            den.Buffers.ColourBuffer->updateData(radiance.data(), radiance.size() * 4);   // the path tracer's output (PathTrace.cuh:618)
            den.TemporalFilter(gb, gb);
            den.FilterMoments(gb);
            void* result = den.WaveletFilter(gb);
            if (hipMemcpy(out.data(), result, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            if (hipMemcpy(hist.data(), den.Buffers.HistoryLength[den.PingPongInx]->Data, px, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            den.EndFrame();
            for (size_t i = 0; i < px; i++) {
                if (hist[i] != frame + 1) { std::printf("frame %d: history %d at %zu\n", frame, hist[i], i); return 1; }
                const float* o = &out[4 * i];
                if (std::fabs(o[0] - 0.25f) > 2e-6f || std::fabs(o[1] - 0.5f) > 2e-6f || std::fabs(o[2] - 0.75f) > 2e-6f || std::fabs(o[3]) > 1e-6f) {
                    std::printf("frame %d: pixel %zu = %g %g %g %g\n", frame, i, o[0], o[1], o[2], o[3]);
                    return 1;
                }
            }
        }
        // ---- the G-buffer through array-backed render targets (CudaUtil.h:68-99) and a pitched surface, the result into a
        //      display array (App.cu:561), then ResizeRenderTextures (App.cu:742-778) and one more frame at the new size
        {
            hipChannelFormatDesc f4 = hipCreateChannelDesc(32, 32, 32, 32, hipChannelFormatKindFloat);
            hipChannelFormatDesc u4 = hipCreateChannelDesc(16, 16, 16, 16, hipChannelFormatKindUnsigned);
            hipArray_t aMotion = nullptr, aNormal = nullptr, aOut = nullptr;
            // CDNA parts have no image / texture hardware (hipDeviceAttributeImageSupport == 0 on MI300X and MI355X): hipMallocArray fails
            // there, and the array adapters have nothing to adapt; the pitched adapter is what such a host uses.
            const bool arrays = hipMallocArray(&aMotion, &f4, W, H, hipArrayDefault) == hipSuccess && hipMallocArray(&aNormal, &u4, W, H, hipArrayDefault) == hipSuccess &&
                                hipMallocArray(&aOut, &f4, W, H, hipArrayDefault) == hipSuccess;
            if (!arrays) { (void)hipGetLastError(); std::printf("note: no array-backed surfaces on this device; array adapters skipped\n"); }
            for (size_t i = 0; i < px; i++) motion[4 * i] = float(i % 251);                 // something to recognise
            if (arrays && hipMemcpy2DToArray(aMotion, 0, 0, motion.data(), size_t(W) * 16, size_t(W) * 16, H, hipMemcpyHostToDevice) != hipSuccess) return 4;
            if (arrays && hipMemcpy2DToArray(aNormal, 0, 0, normal.data(), size_t(W) * 8, size_t(W) * 8, H, hipMemcpyHostToDevice) != hipSuccess) return 4;
            gpupt::buffer lMotion(px * 16), lNormal(px * 8), lUV(px * 8);
            const size_t pitchM = size_t(W) * 16 + 512, pitchN = size_t(W) * 8 + 128;
            std::vector<uint8_t> pm(pitchM * H, 0xEE), pn(pitchN * H, 0xEE);
            for (int y = 0; y < H; y++) { std::memcpy(&pm[y * pitchM], &motion[size_t(y) * W * 4], size_t(W) * 16); std::memcpy(&pn[y * pitchN], &normal[size_t(y) * W * 4], size_t(W) * 8); }
            gpupt::buffer dPM(pm.size(), pm.data()), dPN(pn.size(), pn.data());
            if (arrays) { den.ImportGBufferPlane(SVGF_GBUF_MOTION, aMotion, lMotion.Data); den.ImportGBufferPlane(SVGF_GBUF_NORMAL, aNormal, lNormal.Data); }
            else { den.ImportGBufferPlane(SVGF_GBUF_MOTION, dPM.Data, pitchM, lMotion.Data); den.ImportGBufferPlane(SVGF_GBUF_NORMAL, dPN.Data, pitchN, lNormal.Data); }
            const size_t pitch = size_t(W) * 8 + 256;                                           // a pitched surface with padding at the row ends
            std::vector<uint8_t> pitched(pitch * H, 0xEE);
            for (int y = 0; y < H; y++) std::memcpy(&pitched[y * pitch], &uv[size_t(y) * W * 4], size_t(W) * 8);
            gpupt::buffer dPitched(pitched.size(), pitched.data());
            den.ImportGBufferPlane(SVGF_GBUF_UV, dPitched.Data, pitch, lUV.Data);
            den.Sync();
            std::vector<float> m2(px * 4); std::vector<uint16_t> n2(px * 4), u2(px * 4);
            if (hipMemcpy(m2.data(), lMotion.Data, px * 16, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(n2.data(), lNormal.Data, px * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(u2.data(), lUV.Data, px * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            if (std::memcmp(m2.data(), motion.data(), px * 16) || std::memcmp(n2.data(), normal.data(), px * 8) || std::memcmp(u2.data(), uv.data(), px * 8)) { std::printf("G-buffer import mismatch\n"); return 1; }
            // a frame on the imported planes, its result into the display array and back
            svgf_gbuffer gi{lMotion.Data, lNormal.Data, lUV.Data};
            den.Buffers.ColourBuffer->updateData(radiance.data(), radiance.size() * 4);
            den.TemporalFilter(gi, gi); den.FilterMoments(gi);
            void* res = den.WaveletFilter(gi);
            if (arrays) den.ExportToArray(res, aOut);
            den.Sync();
            std::vector<float> back(px * 4);
            if (hipMemcpy(out.data(), res, px * 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            if (arrays) {
                if (hipMemcpy2DFromArray(back.data(), size_t(W) * 16, aOut, 0, 0, size_t(W) * 16, H, hipMemcpyDeviceToHost) != hipSuccess) return 2;
                if (std::memcmp(back.data(), out.data(), px * 16)) { std::printf("array export mismatch\n"); return 1; }
                (void)hipFreeArray(aMotion); (void)hipFreeArray(aNormal); (void)hipFreeArray(aOut);
            }
            for (size_t i = 0; i < px; i++) if (std::fabs(out[4 * i] - 0.25f) > 2e-6f) { std::printf("frame on imported planes: pixel %zu = %g\n", i, out[4 * i]); return 1; }
            den.EndFrame();
        }
        {
            const int W2 = 333, H2 = 77;
            den.Resize(W2, H2);
            const size_t px2 = size_t(W2) * H2;
            std::vector<float> mo(px2 * 4, 0.0f), ra(px2 * 4), o2(px2 * 4);
            std::vector<uint16_t> no(px2 * 4, 0), uu(px2 * 4, 0);
            for (size_t i = 0; i < px2; i++) { mo[4 * i + 2] = 5.0f; mo[4 * i + 3] = 0.01f; no[4 * i + 2] = half_bits(-1.0f); uu[4 * i + 3] = half_bits(1.0f);
                                               ra[4 * i] = 0.5f; ra[4 * i + 1] = 0.25f; ra[4 * i + 2] = 0.125f; ra[4 * i + 3] = 1.0f; }
            gpupt::buffer dM(mo.size() * 4, mo.data()), dN(no.size() * 2, no.data()), dU(uu.size() * 2, uu.data());
            svgf_gbuffer g2{dM.Data, dN.Data, dU.Data};
            std::vector<uint8_t> h2(px2);
            for (int frame = 0; frame < 2; frame++) {
                den.Buffers.ColourBuffer->updateData(ra.data(), ra.size() * 4);
                den.TemporalFilter(g2, g2); den.FilterMoments(g2);
                void* r2 = den.WaveletFilter(g2);
                if (hipMemcpy(o2.data(), r2, o2.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
                if (hipMemcpy(h2.data(), den.Buffers.HistoryLength[den.PingPongInx]->Data, px2, hipMemcpyDeviceToHost) != hipSuccess) return 2;
                den.EndFrame();
                for (size_t i = 0; i < px2; i++)
                    if (h2[i] != frame + 1 || std::fabs(o2[4 * i] - 0.5f) > 2e-6f || std::fabs(o2[4 * i + 2] - 0.125f) > 2e-6f) { std::printf("after Resize: frame %d pixel %zu\n", frame, i); return 1; }
            }
        }
        std::printf("shim ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
}
