// Probe: does the BPTT kernel run faster when the saved activations it is about to read are already in its XCD's L2?
// One wave per batch tile (workgroup b lands on XCD b % 8 like workgroup b of the recurrent kernel) walks the time axis `lead`
// steps ahead of the recurrence and touches one dword of every 128-byte line of up to three TILE16-style arrays (element
// (step s, tile b) = tile_bytes contiguous bytes at base + (s * tiles + b) * tile_bytes).  Where the recurrence is: either a
// fixed pace (counters == NULL), or the chunk counters it publishes (mvae_rnn_bwd_args.signal_done: chunk c is done when
// counters[c] >= target) with the pace of the previous chunk in between.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/touch_paced.hip -o build/libtouch_paced.so
#include <hip/hip_runtime.h>
#include <stdint.h>
struct touch_args {
    const unsigned char* base[3];
    uint32_t tile_bytes[3];
    int32_t tiles, T, pace_ticks, lead, reverse;      // pace in 10 ns ticks (s_memrealtime); reverse: must be 1 with counters
    const uint32_t* counters;
    uint32_t target;
    int32_t chunk_steps;
};
__device__ __forceinline__ void touch_step(const touch_args& a, int s, int b, int l) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!a.base[k]) continue;
        const unsigned char* p = a.base[k] + ((size_t)s * a.tiles + b) * a.tile_bytes[k];
        for (uint32_t off = (uint32_t)l * 128u; off < a.tile_bytes[k]; off += 64u * 128u) {
            unsigned v;
            asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p + off) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__global__ __launch_bounds__(64) void touch_paced_k(const touch_args a) {
    const int b = blockIdx.x, l = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    if (!a.counters) {
        for (int i = 0; i < a.T; ++i) {
            const int s = a.reverse ? a.T - 1 - i : i;
            const long long due = (long long)(i - a.lead) * a.pace_ticks;
            while ((long long)(wall_clock64() - t0) < due) __builtin_amdgcn_s_sleep(8);
            touch_step(a, s, b, l);
        }
    } else {
        // the recurrence runs t = T-1 .. 0 and publishes chunk c = t / cs when its first step is done
        const int cs = a.chunk_steps;
        int next = a.T - 1;                         // next step to touch
        int c_run = (a.T - 1) / cs;                 // the chunk the recurrence is in
        long long tau = 0, per_step = a.pace_ticks; // start of that chunk (ticks since t0), pace estimate
        const long long deadline = 400000000;       // 4 s
        while (next >= 0) {
            const long long now = (long long)(wall_clock64() - t0);
            if (now > deadline) break;
            // has the running chunk been published?
            if (c_run >= 0 && __hip_atomic_load(a.counters + c_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= a.target) {
                if (c_run < (a.T - 1) / cs) per_step = (now - tau) / cs > 50 ? (now - tau) / cs : per_step;
                tau = now;
                --c_run;
                continue;
            }
            // estimated position of the recurrence inside chunk c_run (clamped to the chunk)
            long long in = per_step > 0 ? (now - tau) / per_step : 0;
            if (in > cs - 1) in = cs - 1;
            const int pos = c_run >= 0 ? c_run * cs + (cs - 1) - (int)in : 0;
            if (next >= pos - a.lead) touch_step(a, next--, b, l);
            else __builtin_amdgcn_s_sleep(4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
extern "C" int touch_paced(const touch_args* a, void* stream) {
    hipLaunchKernelGGL(touch_paced_k, dim3(a->tiles), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
