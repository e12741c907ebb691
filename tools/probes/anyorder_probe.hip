// Does hipExtAnyOrderLaunch (packet without the AQL barrier bit) let kernels of ONE stream run concurrently on gfx950, and what
// does an ordered kernel behind them cost as a JOIN?  (The header says the flag is "not supported on GFX9xx boards" for one of the
// entry points; measured here.)  Compare with the same work on separate streams joined by events.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin_k(unsigned long long ticks, int* out) {      // wall_clock64: the constant 100 MHz counter
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, 1);
}
__global__ void tiny_k(int* out) { if (threadIdx.x == 0) atomicAdd(out, 1); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipStream_t s[4]; for (auto& x : s) CK(hipStreamCreate(&x));
    hipEvent_t ev[4]; for (auto& x : ev) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    const unsigned long long cyc = 30000;      // 300 us at the 100 MHz constant counter
    for (int rep = 0; rep < 3; ++rep) {
        // (a) three spinners in ONE stream, ordinary launches, then a tiny kernel
        CK(hipDeviceSynchronize()); double t = now();
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin_k, dim3(16), dim3(256), 0, s[0], cyc, d);
        hipLaunchKernelGGL(tiny_k, dim3(1), dim3(64), 0, s[0], d);
        CK(hipDeviceSynchronize()); double ta = now() - t;
        // (b) the same with hipExtAnyOrderLaunch on the 2nd and 3rd spinner; the tiny kernel ordered (= the join)
        t = now();
        hipExtLaunchKernelGGL(spin_k, dim3(16), dim3(256), 0, s[0], nullptr, nullptr, 0, cyc, d);
        for (int i = 1; i < 3; ++i) hipExtLaunchKernelGGL(spin_k, dim3(16), dim3(256), 0, s[0], nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
        hipExtLaunchKernelGGL(tiny_k, dim3(1), dim3(64), 0, s[0], nullptr, nullptr, 0, d);
        CK(hipDeviceSynchronize()); double tb = now() - t;
        // (c) three streams forked by an event and joined by events (what the engine does now)
        t = now();
        CK(hipEventRecord(ev[0], s[0]));
        for (int i = 1; i < 3; ++i) CK(hipStreamWaitEvent(s[i], ev[0], 0));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin_k, dim3(16), dim3(256), 0, s[i], cyc, d);
        for (int i = 1; i < 3; ++i) { CK(hipEventRecord(ev[i], s[i])); CK(hipStreamWaitEvent(s[0], ev[i], 0)); }
        hipLaunchKernelGGL(tiny_k, dim3(1), dim3(64), 0, s[0], d);
        CK(hipDeviceSynchronize()); double tc = now() - t;
        printf("rep %d: one stream ordered %.1f us | one stream, any-order launches + ordered join %.1f us | three streams + event joins %.1f us\n",
               rep, ta * 1e6, tb * 1e6, tc * 1e6);
    }
    int h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("kernels completed: %d (expected %d)\n", h, 3 * 3 * 4);
    return 0;
}
