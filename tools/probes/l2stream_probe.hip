// Development probe (round 6): can LSTM weights that do not fit on chip with two waves per SIMD be STREAMED from L2 every time step?
// H = 256 LSTM: U is 512 KiB = the whole register file of a CU.  With 8 waves of 256 registers a wave keeps 32 of its 64 fragments
// in accumulator registers and <= 17 in LDS; the other ~15 (1 KiB each) would have to come from L2 every step: 120 KiB per CU and
// step, 120 global_load_dwordx4 wave-instructions.  Here: G workgroups of 8 waves, every wave runs per "step" 64 MFMAs (16x16x32
// bf16) and NLOAD 16-byte-per-lane loads from a 512 KiB buffer all workgroups share (L2-resident), consumed as MFMA A operands
// one step later (two register sets); step time vs NLOAD and G.
//   hipcc --offload-arch=gfx950 -O3 l2stream_probe.hip -o l2stream_probe && ./l2stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) unsigned short frag;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NLOAD>
__global__ __launch_bounds__(512, 1) void probe(const frag* __restrict__ w, float* out, unsigned long long* cyc, int steps) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    frag a0;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&a"(a0) : "v"(w + l) : "memory");
    frag b = w[l + 64];
    frag ring[2][NLOAD > 0 ? NLOAD : 1];
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const frag* base = w + (size_t)wv * 64 * 64 + l;          // this wave's 64 fragments of the shared 512 KiB
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) ring[0][i] = base[(size_t)i * 64];
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                // one streamed fragment requested per 64 / NLOAD MFMAs, into the OTHER register set
                if (NLOAD > 0 && (g * NLOAD) / 16 != ((g + 1) * NLOAD) / 16) {
#pragma unroll
                    for (int i = (g * NLOAD) / 16; i < ((g + 1) * NLOAD) / 16; ++i)
                        ring[par ^ 1][i] = base[(size_t)((i + s + par) & 63) * 64];
                }
                const frag& u = NLOAD > 0 ? ring[par][(g * NLOAD) / 16 < NLOAD ? (g * NLOAD) / 16 : 0] : b;
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "a"(a0), "v"(b));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(u), "v"(b));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "a"(a0), "v"(b));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "a"(a0), "v"(b));
            }
        }
    }
    asm volatile("s_nop 9" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <int NLOAD>
void run(int G, const frag* w, float* out, unsigned long long* cyc) {
    const int steps = 2000;
    float ms = 0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<NLOAD>), dim3(G), dim3(512), 0, 0, w, out, cyc, steps);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("workgroups %3d  streamed fragments per wave and step %2d (%3d KiB per CU and step): %.3f us per step (64 MFMAs per wave = "
           "128 per SIMD; pipe floor 0.85 us at 2.4 GHz)\n", G, NLOAD, NLOAD * 8, ms * 1e3 / steps);
}

int main() {
    frag* w; float* out; unsigned long long* cyc;
    (void)hipMalloc(&w, 8 << 20); (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 64);
    (void)hipMemset(w, 0x3c, 8 << 20);
    for (int G : {16, 64, 128}) {
        run<0>(G, w, out, cyc);
        run<4>(G, w, out, cyc);
        run<8>(G, w, out, cyc);
        run<16>(G, w, out, cyc);
    }
    return 0;
}
