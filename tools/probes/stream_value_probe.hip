// Development probe: can a stream wait on a value a RUNNING kernel publishes (hipStreamWaitValue32 on signal memory),
// and can a running kernel see a value a stream writes (hipStreamWriteValue32) and data a kernel launched meanwhile
// produced?  Every spin loop is bounded: a failure prints, it does not hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(uint32_t* progress, float* data, int chunks, int spin) {
    // each block publishes chunk k after some "work"
    for (int k = 0; k < chunks; ++k) {
        for (volatile int i = 0; i < spin; ++i) {}
        if (threadIdx.x == 0) data[k * gridDim.x + blockIdx.x] = 100.0f * (k + 1) + blockIdx.x;
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(progress, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void middle(const float* in, float* out, int n, int k, float mul, int fence) {   // "the GEMM" of chunk k
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[k * n + i] = in[k * n + i] * mul;
    if (fence) __threadfence_system();          // write back this XCD's L2 before the kernel is reported complete
}
__global__ void consumer(const uint32_t* flag, const float* data, float* result, int chunks, int n, uint32_t* timeouts, int inv) {
    for (int k = 0; k < chunks; ++k) {
        if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (uint32_t)(k + 1)) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > 20000000L) { atomicAdd(timeouts, 1u); break; }
            }
        }
        __syncthreads();
        if (inv) asm volatile("buffer_inv sc0 sc1" ::: "memory");   // drop this XCD's possibly stale lines
        if (threadIdx.x < n) result[k * n + threadIdx.x] = data[k * n + threadIdx.x];
        __syncthreads();
    }
}
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    uint32_t *progress, *flag, *timeouts;
    CK(hipExtMallocWithFlags((void**)&progress, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    CK(hipMalloc(&timeouts, 4));
    CK(hipMemset(progress, 0, 8)); CK(hipMemset(flag, 0, 8)); CK(hipMemset(timeouts, 0, 4));
    const int chunks = 8, nb = 16, n = nb;
    float *d0, *d1, *res;
    CK(hipMalloc(&d0, chunks * n * 4)); CK(hipMalloc(&d1, chunks * n * 4)); CK(hipMalloc(&res, chunks * n * 4));
    CK(hipMemset(d0, 0, chunks * n * 4)); CK(hipMemset(d1, 0, chunks * n * 4)); CK(hipMemset(res, 0, chunks * n * 4));
    hipStream_t sp, sm, sc;
    CK(hipStreamCreate(&sp)); CK(hipStreamCreate(&sm)); CK(hipStreamCreate(&sc));
    for (int rep = 0; rep < 12; ++rep) {
        const int fence = (rep / 3) & 1, inv = (rep / 6) & 1;
        const float mul = 2.0f + rep;
        CK(hipMemset(progress, 0, 8)); CK(hipMemset(flag, 0, 8));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, sc, flag, d1, res, chunks, n, timeouts, inv);       // persistent consumer first
        hipLaunchKernelGGL(producer, dim3(nb), dim3(64), 0, sp, progress, d0, chunks, 200000);
        for (int k = 0; k < chunks; ++k) {
            CK(hipStreamWaitValue32(sm, progress, (uint32_t)(nb * (k + 1)), hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(middle, dim3(1), dim3(64), 0, sm, d0, d1, n, k, mul, fence);
            CK(hipStreamWriteValue32(sm, flag, (uint32_t)(k + 1), 0));
        }
        CK(hipDeviceSynchronize());
        float h[chunks * n]; uint32_t to;
        CK(hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int k = 0; k < chunks; ++k) for (int i = 0; i < n; ++i) if (h[k * n + i] != mul * (100.0f * (k + 1) + i)) ++bad;
        if (bad) { for (int k = 0; k < chunks; ++k) { printf("  chunk %d:", k); for (int i = 0; i < n; ++i) printf(" %g", h[k * n + i]); printf("\n"); } }
        printf("rep %2d (producer-side fence %d, consumer-side buffer_inv %d): timeouts %u, wrong values %d of %d\n", rep, fence, inv, to, bad, chunks * n);
    }
    return 0;
}
