// Round 4's L2-touch companion of the BPTT launches (mvae_l2_touch_bwd, ABI 6), REMOVED from the product in round 5: beside the
// decoder BPTT phase launch it made THAT launch 0.3 ms shorter (1.71 -> 1.40 ms) and the train step not at all (same-box A/B,
// profiles/r05_b_inkernel_touch.txt: 6.83-6.88 ms with it, 6.80-6.88 without) - the velocity head's BPTT, which it did not
// cover, and the joins behind the phase set the pace then.  Kept as a probe: the kernel and its argument struct, as they were.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
typedef struct {
    const void* base[3];
    uint32_t tile_bytes[3];
    int32_t T, tiles;            /* time steps (walked T-1 .. 0); 16-row tiles */
    int32_t chunk_steps;         /* of the recurrence; divides T */
    int32_t first_wg;
    int32_t lead;                /* time steps ahead of the recurrence's estimated position */
    uint32_t target;
    const uint32_t* counters;    /* [T / chunk_steps] */
    const uint32_t* status;      /* or NULL */
} mvae_l2_touch_args;
#define MVAE_OK 0
#define MVAE_E_ARG -1
#define MVAE_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return -2; } while (0)
// ---- L2 touch: a companion of a running BPTT launch (include/midivae_hip.h) -------------------------------------------------
struct l2_touch_multi {
    mvae_l2_touch_args p[8];
    int n;
};
__device__ __forceinline__ void l2_touch_walk(const mvae_l2_touch_args a, const int b, const int l) {
    const int cs = a.chunk_steps;
    const unsigned char* base[3] = {static_cast<const unsigned char*>(a.base[0]), static_cast<const unsigned char*>(a.base[1]),
                                    static_cast<const unsigned char*>(a.base[2])};
    const uint32_t tb[3] = {a.tile_bytes[0], a.tile_bytes[1], a.tile_bytes[2]};
    auto touch = [&](int s) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!base[k]) continue;
            const unsigned char* p = base[k] + ((size_t)s * a.tiles + b) * tb[k] + (size_t)l * 128u;
            for (uint32_t off = (uint32_t)l * 128u; off < tb[k]; off += 64u * 128u, p += 64u * 128u) {
                unsigned v;
                asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // (at most two steps of requests in flight)
    };
    const unsigned long long t0 = wall_clock64();              // 100 MHz
    int next = a.T - 1;                          // next step to touch
    int c_run = (a.T - 1) / cs;                  // the chunk the recurrence is in (it runs t = T-1 .. 0 and publishes chunk t / cs)
    long long tau = 0, per_step = 260;           // start of that chunk (ticks since t0); pace estimate: 2.6 us until a chunk was timed
    unsigned polls = 0;
    while (next >= 0) {
        const long long now = (long long)(wall_clock64() - t0);
        if (now > 400000000ll) break;            // 4 s
        if (a.status && (++polls & 63u) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
        if (c_run >= 0 && __hip_atomic_load(a.counters + c_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= a.target) {
            if (c_run < (a.T - 1) / cs && (now - tau) / cs > 50) per_step = (now - tau) / cs;
            tau = now;
            --c_run;
            continue;
        }
        long long in = (now - tau) / per_step;   // estimated position inside the running chunk, clamped to it
        if (in > cs - 1) in = cs - 1;
        const int pos = c_run >= 0 ? c_run * cs + (cs - 1) - (int)in : 0;
        if (next >= pos - a.lead) touch(next--);
        else __builtin_amdgcn_s_sleep(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__global__ __launch_bounds__(64) void l2_touch_bwd_k(const l2_touch_multi m) {
    const int wg = blockIdx.x, l = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < m.n && wg >= m.p[i].first_wg && wg < m.p[i].first_wg + m.p[i].tiles) {
            l2_touch_walk(m.p[i], wg - m.p[i].first_wg, l);
            return;
        }
}
extern "C" int mvae_l2_touch_bwd(const mvae_l2_touch_args* problems, int32_t n, void* stream) {
    if (!problems || n <= 0 || n > 8) return MVAE_E_ARG;
    l2_touch_multi m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        const mvae_l2_touch_args& a = problems[i];
        if (a.T <= 0 || a.tiles <= 0 || a.chunk_steps <= 0 || (a.T % a.chunk_steps) || a.first_wg < 0 || a.lead < 0 || !a.counters)
            return MVAE_E_ARG;
        for (int k = 0; k < 3; ++k)
            if (a.base[k] && (a.tile_bytes[k] == 0 || (a.tile_bytes[k] % 128))) return MVAE_E_ARG;
        m.p[i] = a;
        grid = a.first_wg + a.tiles > grid ? a.first_wg + a.tiles : grid;
    }
    hipLaunchKernelGGL(l2_touch_bwd_k, dim3(grid), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), m);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

