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
        std::printf("shim ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
}
