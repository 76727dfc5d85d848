// SVGF.h — source-compatible replacement of the reference's src/SVGF.h for the MI355X build.
//
// The reference declares `gpupt::svgfBuffers` (six std::shared_ptr<buffer> + Init(Width,Height), src/SVGF.h:7-16)
// but never defines Init, never compiles SVGF.cpp (CMakeLists.txt:88-111) and never instantiates the struct; the
// filter is driven by application::TemporalFilter/FilterMoments/WaveletFilter (src/App.cu:469-514) on buffers that
// application::ResizeRenderTextures allocates (src/App.cu:742-778).  This header keeps the struct and its six
// members, gives Init a body, and adds `gpupt::svgfDenoiser`, whose three methods carry the names and sequencing of
// those host methods but launch the gfx950 kernels through the C ABI of svgf.h.  Host code stays C++; nothing here
// is device code, so it compiles with g++ as well as hipcc (link libsvgf_mi355x.so + libamdhip64).
#pragma once
#include <stdint.h>

#include <memory>
#include <stdexcept>
#include <string>

#include <hip/hip_runtime_api.h>

#include "svgf.h"
#include "svgf_ext.h"      // svgf_set_prev_guide / svgf_set_adaptive_moments: the shim exposes the two opt-ins

namespace gpupt {

// Device buffer with the interface of the reference's `buffer` (src/Buffer.h:8-19: ctor(size, data), Destroy,
// updateData, Reallocate, public Data/Size) over hipMalloc instead of cudaMalloc (src/Buffer.cpp:12-24).
// Unlike the reference it zero-fills on allocation.
class buffer {
public:
    explicit buffer(size_t dataSize, const void* data = nullptr) : Data(nullptr), Size(0) { Reallocate(data, dataSize); }
    ~buffer() { Destroy(); }
    buffer(const buffer&) = delete;
    buffer& operator=(const buffer&) = delete;
    void Destroy() {
        if (Data) (void)hipFree(Data);
        Data = nullptr;
        Size = 0;
    }
    void updateData(const void* data, size_t dataSize) { updateData(0, data, dataSize); }
    void updateData(size_t offset, const void* data, size_t dataSize) {
        if (offset + dataSize > Size) throw std::out_of_range("gpupt::buffer::updateData beyond the allocation");
        check(hipMemcpy(static_cast<char*>(Data) + offset, data, dataSize, hipMemcpyHostToDevice), "hipMemcpy");
    }
    void Reallocate(const void* data, size_t dataSize) {
        Destroy();
        Size = dataSize;
        check(hipMalloc(&Data, dataSize ? dataSize : 1), "hipMalloc");
        check(hipMemset(Data, 0, dataSize), "hipMemset");
        if (data) check(hipMemcpy(Data, data, dataSize, hipMemcpyHostToDevice), "hipMemcpy");
    }
    void* Data;
    size_t Size;

private:
    static void check(hipError_t e, const char* what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
};

// src/SVGF.h:7-16, same six members.  Mapping onto the planes the reference's application owns (src/App.h:138-141):
//   ColourBuffer         RenderBuffer[P]      current colour+variance (temporal in/out, iteration-0 feedback)
//   HistoryBufferColour  RenderBuffer[1-P]    previous frame's colour
//   MomentsBuffer        MomentsBuffer[P]     current luminance moments
//   HistoryBufferMoments MomentsBuffer[1-P]   previous frame's moments
//   VarianceBuffer       FilterBuffer[0..1]   the two colour+variance planes the à-trous iterations ping-pong (one allocation)
//   MotionVectors        linear copy of the G-buffer's motion/depth plane (float4 per pixel); the caller fills it
// plus the two history-length planes the reference keeps outside the struct (HistoryLengthBuffer, ping-ponged here).
struct svgfBuffers {
    std::shared_ptr<buffer> ColourBuffer;
    std::shared_ptr<buffer> VarianceBuffer;
    std::shared_ptr<buffer> MomentsBuffer;
    std::shared_ptr<buffer> HistoryBufferColour;
    std::shared_ptr<buffer> HistoryBufferMoments;
    std::shared_ptr<buffer> MotionVectors;
    std::shared_ptr<buffer> HistoryLength[2];
    uint32_t Width = 0, Height = 0;
    int Storage = SVGF_F16;                      // the reference's half4/half2 (Filter.cuh:15-16)

    void Init(uint32_t Width_, uint32_t Height_) {
        Width = Width_;
        Height = Height_;
        const size_t px = static_cast<size_t>(Width) * Height;
        const size_t c4 = Storage == SVGF_F16 ? 8 : 16, c2 = Storage == SVGF_F16 ? 4 : 8;
        ColourBuffer = std::make_shared<buffer>(px * c4);            // exact size, not x4 (App.cu:763; App. B #10)
        HistoryBufferColour = std::make_shared<buffer>(px * c4);
        MomentsBuffer = std::make_shared<buffer>(px * c2);
        HistoryBufferMoments = std::make_shared<buffer>(px * c2);
        VarianceBuffer = std::make_shared<buffer>(2 * px * c4);
        MotionVectors = std::make_shared<buffer>(px * 16);
        HistoryLength[0] = std::make_shared<buffer>(px);
        HistoryLength[1] = std::make_shared<buffer>(px);
    }
    void* FilterPlane(int i) const {
        return static_cast<char*>(VarianceBuffer->Data) + static_cast<size_t>(i) * Width * Height * (Storage == SVGF_F16 ? 8 : 16);
    }
};

// The three host methods of the reference's application (src/App.cu:469-514) over svgfBuffers.
class svgfDenoiser {
public:
    // tunables, same names and defaults as src/App.h:109-114
    int SpatialFilterSteps = 3;
    float DepthThreshold = 0.8f;
    float NormalThreshold = 0.9f;
    int HistoryLength = 24;
    float PhiColour = 10.0f;
    float PhiNormal = 128.0f;
    int NanPolicy = SVGF_NAN_REFERENCE;             // SVGF_NAN_ZERO: the temporal stage reads a NaN radiance / history channel as 0 (an extension, svgf.h)

    svgfDenoiser(uint32_t Width, uint32_t Height, int Storage = SVGF_F16, int Device = 0, hipStream_t Stream = nullptr) {
        Buffers.Storage = Storage;
        Buffers.Init(Width, Height);
        svgf_params p;
        svgf_default_params(&p);
        p.storage = Storage;
        int rc = svgf_create(&Ctx, static_cast<int>(Width), static_cast<int>(Height), &p, Device, Stream);
        if (rc != SVGF_OK) throw std::runtime_error(std::string("svgf_create: ") + svgf_status_string(rc));
    }
    ~svgfDenoiser() { svgf_destroy(Ctx); }
    svgfDenoiser(const svgfDenoiser&) = delete;
    svgfDenoiser& operator=(const svgfDenoiser&) = delete;

    // application::TemporalFilter (App.cu:469-478): in place on ColourBuffer, which holds this frame's 1-spp radiance.
    void TemporalFilter(const svgf_gbuffer& Current, const svgf_gbuffer& Previous) {
        push_params();
        check(svgf_temporal(Ctx, Buffers.HistoryBufferColour->Data, Buffers.ColourBuffer->Data, Buffers.ColourBuffer->Data, &Current, &Previous,
                            static_cast<const uint8_t*>(Buffers.HistoryLength[1 - PingPongInx]->Data),
                            static_cast<uint8_t*>(Buffers.HistoryLength[PingPongInx]->Data), Buffers.MomentsBuffer->Data,
                            Buffers.HistoryBufferMoments->Data), "svgf_temporal");
    }
    // application::FilterMoments (App.cu:480-489)
    void FilterMoments(const svgf_gbuffer& Current) {
        check(svgf_moments(Ctx, Buffers.ColourBuffer->Data, Buffers.FilterPlane(0), Buffers.MomentsBuffer->Data, &Current,
                           static_cast<const uint8_t*>(Buffers.HistoryLength[PingPongInx]->Data)), "svgf_moments");
    }
    // application::WaveletFilter (App.cu:491-514); returns the plane holding the result (no odd-N copy)
    void* WaveletFilter(const svgf_gbuffer& Current) {
        int PingPong = 0;
        for (int i = 0; i < SpatialFilterSteps; i++) {
            check(svgf_atrous(Ctx, Buffers.FilterPlane(PingPong), Buffers.FilterPlane(1 - PingPong), Buffers.ColourBuffer->Data, &Current, 1 << i, i),
                  "svgf_atrous");
            PingPong = 1 - PingPong;
        }
        return Buffers.FilterPlane(PingPong);
    }
    // application::TAA (App.cu:516-522): Filtered -> Out with the previous Out as history
    void TAA(const void* Filtered, const void* History, void* Out) { check(svgf_taa(Ctx, Filtered, History, Out), "svgf_taa"); }
    // Albedo demodulation / re-modulation (the SVGF paper's; the reference has none, README.md:14): in place on the frame's
    // 1-spp radiance before TemporalFilter, and on the filtered plane after WaveletFilter.
    void Demodulate(const void* Albedo) { check(svgf_demodulate(Ctx, Buffers.ColourBuffer->Data, Albedo, Buffers.ColourBuffer->Data), "svgf_demodulate"); }
    void Modulate(void* Filtered, const void* Albedo) { check(svgf_modulate(Ctx, Filtered, Albedo, Filtered), "svgf_modulate"); }
    // application::ResizeRenderTextures (App.cu:742-778): every buffer reallocated at the new size (zero-filled), accumulation
    // restarted (ResetRender, App.cu:777); tunables stay.
    void Resize(uint32_t Width, uint32_t Height) {
        check(svgf_resize(Ctx, static_cast<int>(Width), static_cast<int>(Height)), "svgf_resize");
        Buffers.Init(Width, Height);
        PingPongInx = 0;
    }
    // The G-buffer of the reference lives in array-backed GL textures mapped into CUDA (CudaUtil.h:68-99); here an array-backed
    // (hipArray_t) render target is copied into the linear plane the filter reads, and the filtered plane back into the display
    // texture's array (cudaMemcpyToArray, App.cu:561).
    void ImportGBufferPlane(int Plane, hipArray_const_t Array, void* LinearPlane) { check(svgf_import_gbuffer_array(Ctx, Plane, Array, LinearPlane), "svgf_import_gbuffer_array"); }
    void ImportGBufferPlane(int Plane, const void* Pitched, size_t PitchBytes, void* LinearPlane) { check(svgf_import_gbuffer_pitched(Ctx, Plane, Pitched, PitchBytes, LinearPlane), "svgf_import_gbuffer_pitched"); }
    void ExportToArray(const void* Plane, hipArray_t Array) { check(svgf_export_to_array(Ctx, Plane, Array), "svgf_export_to_array"); }
    void Sync() { check(svgf_sync(Ctx), "svgf_sync"); }
    // svgf_denoise_frame / the strip driver may read the 16-byte guide plane kept of the previous frame's G-buffer instead of its three
    // planes when `prev` is that G-buffer (svgf.h, svgf_set_prev_guide: off by default; the host vouches that it does not rewrite
    // Framebuffer[1 - PingPongInx] between two frames, as the reference does not).  The three stage calls above never use it.
    void SetPrevGuide(bool Enable) { check(svgf_set_prev_guide(Ctx, Enable ? 1 : 0), "svgf_set_prev_guide"); }
    // svgf_denoise_frame serves a crowded frame (> 8 % young pixels by a sample of recent frames) with the LDS-streaming moments kernel instead
    // of the young-pixel launch: same results, bounded frame time (svgf.h, svgf_set_adaptive_moments: on by default)
    void SetAdaptiveMoments(bool Enable) { check(svgf_set_adaptive_moments(Ctx, Enable ? 1 : 0), "svgf_set_adaptive_moments"); }
    // application::EndFrame's share (App.cu:374): this frame's colour/moments/history become the previous frame's
    void EndFrame() {
        std::swap(Buffers.ColourBuffer, Buffers.HistoryBufferColour);
        std::swap(Buffers.MomentsBuffer, Buffers.HistoryBufferMoments);
        PingPongInx ^= 1;
    }

    svgfBuffers Buffers;
    int PingPongInx = 0;

private:
    void push_params() {
        svgf_params p;
        svgf_default_params(&p);
        p.steps = SpatialFilterSteps; p.depth_threshold = DepthThreshold; p.normal_threshold = NormalThreshold;
        p.history_base = HistoryLength; p.phi_colour = PhiColour; p.phi_normal = PhiNormal; p.storage = Buffers.Storage;
        p.nan_policy = NanPolicy;
        check(svgf_set_params(Ctx, &p), "svgf_set_params");
    }
    void check(int rc, const char* what) {
        if (rc != SVGF_OK) throw std::runtime_error(std::string(what) + ": " + svgf_status_string(rc) + ": " + svgf_last_error(Ctx));
    }
    svgf_ctx* Ctx = nullptr;
};

}  // namespace gpupt
