// Does a compute-only workgroup run slower when more CUs are busy?  N workgroups (one per CU: 160 KiB of LDS each) run a fixed
// dependent chain of (a) VALU fmas, (b) MFMAs, (c) LDS reads + MFMA like the recurrent kernels; per kernel: wall time, and
// per workgroup shader-clock cycles (s_memtime) against the 100 MHz constant clock (s_memrealtime).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/busy_cu_probe.hip -o build/busy_cu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct rec { unsigned long long cyc, ref; };

template <int MODE>
__global__ __launch_bounds__(256) void probe_k(rec* out, int iters, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 256) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    unsigned long long c0 = clock64(), r0 = wall_clock64();
    float v = lane * 1e-3f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (lane + k)); b[k] = (__bf16)(0.02f * (lane - k)); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 64; ++k) v = __builtin_fmaf(v, 0.999f, 0.001f);
        } else if (MODE == 3) {      // f32 MFMA, one accumulator (dependent chain)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, 1.0f, acc, 0, 0, 0);
        } else if (MODE == 4) {      // f32 MFMA, two accumulators alternating
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, 1.0f, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v, 2.0f, acc2, 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                f32x4 l = *reinterpret_cast<f32x4*>(&lds[((it * 16 + k) * 256 + lane) * 4 % 40960]);
                a[0] = (__bf16)l[0];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }
    unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (lane == 0) out[blockIdx.x] = {c1 - c0, r1 - r0};
    if (v + acc[0] + acc[1] + acc2[0] == 123.456f) sink[0] = v;
}

template <int MODE>
void run(const char* name, int grid, int iters, rec* d, float* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe_k<MODE><<<grid, 256, 163840>>>(d, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe_k<MODE><<<grid, 256, 163840>>>(d, iters, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<rec> h(grid);
    hipMemcpy(h.data(), d, grid * sizeof(rec), hipMemcpyDeviceToHost);
    unsigned long long cmin = ~0ull, cmax = 0, rmax = 0;
    for (auto& x : h) { cmin = std::min(cmin, x.cyc); cmax = std::max(cmax, x.cyc); rmax = std::max(rmax, x.ref); }
    printf("%-10s grid %3d  %8.3f ms/launch  s_memtime cycles min %llu max %llu  refclk(100MHz) max %llu -> %.0f s_memtime ticks/us\n", name,
           grid, ms / 5, cmin, cmax, rmax, (double)cmax / ((double)rmax / 100.0));
}

int main(int argc, char** argv) {
    rec* d; float* sink;
    hipMalloc(&d, 1024 * sizeof(rec)); hipMalloc(&sink, 4);
    const int grids[] = {8, 16, 32, 64, 96, 128, 160, 192, 256};
    for (int g : grids) run<0>("valu", g, 40000, d, sink);
    for (int g : grids) run<1>("mfma", g, 40000, d, sink);
    for (int g : grids) run<2>("lds+mfma", g, 20000, d, sink);
    // f32 MFMA issue rate (16 per iteration): cycles per MFMA = s_memtime cycles / (iters * 16)
    run<3>("f32 chain", 16, 40000, d, sink);
    run<4>("f32 2-acc", 16, 40000, d, sink);
    return 0;
}
