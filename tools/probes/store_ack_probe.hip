// store_ack_probe.hip - how long until a 16-byte-per-lane store is ACKNOWLEDGED (vmcnt drops), by cache policy.
// The resident recurrent kernels wait for vmcnt(0) once per time step (~2.5 us): a store whose acknowledgement takes longer than
// a step stalls the step.  plain / nt are acknowledged by the XCD's L2; sc1 / sc0 sc1 (write-through) by the fabric.
//   hipcc --offload-arch=gfx950 -O3 -o store_ack_probe store_ack_probe.hip && ./store_ack_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned char* buf, unsigned long long* out, int iters, size_t wave_bytes) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    unsigned char* base = buf + ((size_t)blockIdx.x * 4 + w) * wave_bytes;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    u32x4 v = {1u, 2u, 3u, (unsigned)l};
    unsigned long long acc = 0, mx = 0;
    for (int it = 0; it < iters; ++it) {
        const int off = (it * 1024 + l * 16) % (int)wave_bytes;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t0 = wall_clock64();
        if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
        if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);       // sc1
        if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 17);       // sc0 sc1
        if (MODE == 3) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 2);        // nt
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = wall_clock64();
        acc += t1 - t0;
        mx = t1 - t0 > mx ? t1 - t0 : mx;
        v[0] += 1;
        __builtin_amdgcn_s_sleep(64);
    }
    if (l == 0) {
        out[((size_t)blockIdx.x * 4 + w) * 2] = acc;
        out[((size_t)blockIdx.x * 4 + w) * 2 + 1] = mx;
    }
}

// Part 2: what a release of PLAIN stores costs (buffer_wbl2 writes back every dirty line of the XCD's L2) after a chunk's worth of
// saved activations (32 steps x 40 KB per workgroup), issued by all four waves of a workgroup or by one wave behind a barrier.
template <int WHO>      // 0: every wave fences; 1: wave 0 fences, the others meet it at the next barrier; 2: no fence (floor)
__global__ __launch_bounds__(256) void fence_k(unsigned char* buf, unsigned long long* out, int iters, size_t wg_bytes) {
    const int w = threadIdx.x >> 6;
    unsigned char* base = buf + (size_t)blockIdx.x * wg_bytes;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < 32; ++s) {                       // 32 "steps": 10 x 4 KB per workgroup each, ~2.5 us apart
#pragma unroll
            for (int j = 0; j < 10; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(((it & 1) * 32 + s) * 40960 + j * 4096 + threadIdx.x * 16), 0, 0);
            __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(127);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        if (WHO == 0 || (WHO == 1 && w == 0)) asm volatile("buffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += wall_clock64() - t0;
        v[0] += 1;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main() {
    for (int grid : {16, 64}) {
        const size_t wg_bytes = 64 * 40960;
        const int iters = 200;
        unsigned char* buf;
        unsigned long long* out;
        hipMalloc(&buf, wg_bytes * grid);
        hipMalloc(&out, sizeof(unsigned long long) * grid);
        const char* who[3] = {"every wave fences", "wave 0 fences", "no fence"};
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) fence_k<0><<<grid, 256>>>(buf, out, iters, wg_bytes);
                if (mode == 1) fence_k<1><<<grid, 256>>>(buf, out, iters, wg_bytes);
                if (mode == 2) fence_k<2><<<grid, 256>>>(buf, out, iters, wg_bytes);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), out, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < grid; ++i) s += (double)h[i];
            printf("grid %3d  %-18s release after 1.3 MB of plain stores per workgroup: %.2f us per hand-over\n", grid, who[mode],
                   s / ((double)grid * iters) / 100.0);
        }
        hipFree(buf);
        hipFree(out);
    }
    const int iters = 2000;
    const size_t wave_bytes = 1 << 20;
    const char* names[4] = {"plain", "sc1", "sc0 sc1", "nt"};
    for (int grid : {1, 16, 128, 256}) {
        unsigned char* buf;
        unsigned long long* out;
        hipMalloc(&buf, wave_bytes * 4 * grid);
        hipMalloc(&out, sizeof(unsigned long long) * 8 * grid);
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) k<0><<<grid, 256>>>(buf, out, iters, wave_bytes);
                if (mode == 1) k<1><<<grid, 256>>>(buf, out, iters, wave_bytes);
                if (mode == 2) k<2><<<grid, 256>>>(buf, out, iters, wave_bytes);
                if (mode == 3) k<3><<<grid, 256>>>(buf, out, iters, wave_bytes);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(8 * grid);
            hipMemcpy(h.data(), out, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
            double s = 0, m = 0;
            for (int i = 0; i < 4 * grid; ++i) { s += (double)h[2 * i]; m = h[2 * i + 1] > m ? (double)h[2 * i + 1] : m; }
            printf("grid %3d  %-8s store->ack  mean %.2f us  max %.2f us   (wall clock 100 MHz)\n", grid, names[mode],
                   s / (4.0 * grid * iters) / 100.0, m / 100.0);
        }
        hipFree(buf);
        hipFree(out);
    }
    return 0;
}
