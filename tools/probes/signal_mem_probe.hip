// Development probe: may one hipMallocSignalMemory allocation hold many 32-bit counters (hipStreamWaitValue32 on offsets)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); ok = 0; } } while (0)
__global__ void bump(uint32_t* p) { __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void mark(int* m, int v) { *m = v; }
int main() {
    int ok = 1;
    uint32_t* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 4096, hipMallocSignalMemory));
    printf("4096-byte signal allocation: %s\n", ok ? "ok" : "FAILED");
    if (!ok) { ok = 1; CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); }
    uint32_t* plain; CK(hipMalloc(&plain, 4096));
    int* m; CK(hipMalloc(&m, 4));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipMemsetAsync(plain, 0, 4096, a));
    for (int off : {0, 2, 16, 100}) {
        ok = 1;
        CK(hipMemsetAsync(sig, 0, 4096, a)); CK(hipMemsetAsync(m, 0, 4, a)); CK(hipStreamSynchronize(a));
        CK(hipStreamWaitValue32(b, sig + off, 3, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, b, m, 7);
        int h0 = -1; CK(hipMemcpyAsync(&h0, m, 4, hipMemcpyDeviceToHost, a)); CK(hipStreamSynchronize(a));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, a, sig + off);
        CK(hipStreamSynchronize(a));
        hipError_t e = hipSuccess; int spins = 0;
        while ((e = hipStreamQuery(b)) == hipErrorNotReady && ++spins < 2000000) {}
        int h1 = -1; CK(hipMemcpyAsync(&h1, m, 4, hipMemcpyDeviceToHost, a)); CK(hipStreamSynchronize(a));
        printf("signal offset %3d words: api ok %d, marker before bumps %d, after %d (%s)\n", off, ok, h0, h1, e == hipSuccess ? "released" : "STILL WAITING");
    }
    // plain device memory as a wait target (expected to be rejected or unsupported)
    ok = 1;
    hipError_t e = hipStreamWaitValue32(b, plain, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
    printf("wait on plain hipMalloc memory: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, a, plain); CK(hipStreamSynchronize(a));
        int spins = 0; hipError_t q;
        while ((q = hipStreamQuery(b)) == hipErrorNotReady && ++spins < 2000000) {}
        printf("  ... %s\n", q == hipSuccess ? "and it released" : "STILL WAITING (releasing it through the signal)");
    }
    // write value to plain memory
    e = hipStreamWriteValue32(a, plain + 5, 42, 0); CK(hipStreamSynchronize(a));
    uint32_t hv = 0; CK(hipMemcpyAsync(&hv, plain + 5, 4, hipMemcpyDeviceToHost, a)); CK(hipStreamSynchronize(a));
    printf("hipStreamWriteValue32 to plain memory: %s, value %u\n", hipGetErrorString(e), hv);
    return 0;
}
