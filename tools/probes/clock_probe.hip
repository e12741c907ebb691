// clock_probe.hip - the shader clock one workgroup sees while N other workgroups keep their CUs busy (MFMA loop / VALU loop / sleeping).
// A recurrence occupies 16 CUs for a whole phase of the step; its time per step is cycles / clock - and the clock is not a constant.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void meter(unsigned long long* out, int iters) {
    float x = threadIdx.x * 1e-3f;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) x = x * 1.0001f + 0.5f;      // a dependent chain: time = cycles / clock
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (x == 123.f) out[2] = 1;
}
template <int KIND>     // 0: MFMA back to back, 1: VALU, 2: sleeping (a waiting persistent workgroup)
__global__ __launch_bounds__(256) void load(float* sink, volatile int* stop) {
    f32x4 acc[4] = {};
    s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    float x = threadIdx.x;
    for (long it = 0; it < (1L << 40); ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 64; ++j) x = x * 1.0001f + 0.5f;
        } else {
            __builtin_amdgcn_s_sleep(127);
        }
        if ((it & 1023) == 0 && *stop) break;
    }
    if (x == 123.f || acc[0][0] == 7.f) sink[0] = acc[1][1] + acc[2][2] + acc[3][3];
}

int main() {
    unsigned long long *out, h[3];
    float* sink;
    int* stop;
    hipMalloc(&out, 64); hipMalloc(&sink, 64);
    hipHostMalloc(&stop, sizeof(int), hipHostMallocMapped);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const char* names[3] = {"MFMA", "VALU", "sleep"};
    for (int kind = 0; kind < 3; ++kind)
        for (int n : {0, 16, 48, 96, 160, 240}) {
            *stop = 0;
            if (n) {
                if (kind == 0) load<0><<<n, 256, 0, s2>>>(sink, stop);
                if (kind == 1) load<1><<<n, 256, 0, s2>>>(sink, stop);
                if (kind == 2) load<2><<<n, 256, 0, s2>>>(sink, stop);
            }
            double mhz = 0;
            for (int rep = 0; rep < 3; ++rep) {      // the third measurement: the clock has had ~20 ms to settle
                meter<<<1, 256, 0, s1>>>(out, 40000);
                hipStreamSynchronize(s1);
                hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                mhz = (double)h[0] / ((double)h[1] / 100.0);
            }
            *stop = 1;
            hipDeviceSynchronize();
            printf("%-5s load on %3d CUs: meter workgroup runs at %.0f MHz (%.2f ms for a fixed dependent chain)\n", names[kind], n, mhz,
                   (double)h[1] / 1e5);
            if (kind && n == 0) continue;
        }
    return 0;
}
