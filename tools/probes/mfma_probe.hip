// Development probe: issue cost of 4-MFMA groups in one wave per SIMD with the A operand in AGPRs vs VGPRs, with and
// without the per-group s_nop, LDS B-fragment reads, and VALU fillers.   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) unsigned short frag;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MF(c, a, b) "v_mfma_f32_16x16x32_bf16 " c ", " a ", " b ", " c "\n\t"

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const frag* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[16 * 256 * 2];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 256 * 2; i += 256) tile[i] = (unsigned short)(0x3c00 + (i & 7));
    __syncthreads();
    frag a0, a1, a2, a3, a4, a5, a6, a7;
    if (MODE & 1) {
        asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\t"
                     "global_load_dwordx4 %2, %8, off offset:32\n\tglobal_load_dwordx4 %3, %8, off offset:48\n\t"
                     "global_load_dwordx4 %4, %8, off offset:64\n\tglobal_load_dwordx4 %5, %8, off offset:80\n\t"
                     "global_load_dwordx4 %6, %8, off offset:96\n\tglobal_load_dwordx4 %7, %8, off offset:112\n\ts_waitcnt vmcnt(0)"
                     : "=&a"(a0), "=&a"(a1), "=&a"(a2), "=&a"(a3), "=&a"(a4), "=&a"(a5), "=&a"(a6), "=&a"(a7) : "v"(src + l * 8) : "memory");
    } else {
        a0 = src[l * 8]; a1 = src[l * 8 + 1]; a2 = src[l * 8 + 2]; a3 = src[l * 8 + 3];
        a4 = src[l * 8 + 4]; a5 = src[l * 8 + 5]; a6 = src[l * 8 + 6]; a7 = src[l * 8 + 7];
    }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float f0 = l, f1 = 1.0f, f2 = 2.0f;
    frag b = src[l], b2 = src[l + 64];
    const frag* lp = reinterpret_cast<const frag*>(tile) + l;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE & 4) {   // B fragment from LDS, prefetched one group ahead
                frag nb = lp[((g + it) & 7) * 64];
                asm volatile("" : "+v"(nb));
                b2 = b; b = nb;
            }
#define GROUP(A0, A1, A2, A3, CONS)                                                                     \
    if (MODE & 2) asm volatile("s_nop 1\n\t" MF("%0", "%4", "%8") MF("%1", "%5", "%8") MF("%2", "%6", "%8") MF("%3", "%7", "%8") \
                               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : CONS(A0), CONS(A1), CONS(A2), CONS(A3), "v"(b2));       \
    else asm volatile(MF("%0", "%4", "%8") MF("%1", "%5", "%8") MF("%2", "%6", "%8") MF("%3", "%7", "%8")                          \
                      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : CONS(A0), CONS(A1), CONS(A2), CONS(A3), "v"(b2));
            if (MODE & 1) { if (g & 1) { GROUP(a4, a5, a6, a7, "a") } else { GROUP(a0, a1, a2, a3, "a") } }
            else { if (g & 1) { GROUP(a4, a5, a6, a7, "v") } else { GROUP(a0, a1, a2, a3, "v") } }
            if (MODE & 8) {   // 8 independent VALU fillers per group
#pragma unroll
                for (int k = 0; k < 4; ++k) { f0 = f0 * f1 + f2; f1 = f1 * f2 + f0; }
            }
            if (MODE & 16) {  // single-MFMA statements with 2 fillers after each: emulated by 4 extra fillers between groups
                asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %0" : "+v"(f0), "+v"(f1) : "v"(f2));
            }
        }
    }
    asm volatile("s_nop 9" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1;
}

// interleaved: every MFMA its own statement followed by NF VALU fillers
template <int NF, bool AG>
__global__ __launch_bounds__(256, 1) void probe_il(const frag* src, float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    frag a0, a1, a2, a3;
    if (AG) asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                         : "=&a"(a0), "=&a"(a1), "=&a"(a2), "=&a"(a3) : "v"(src + l * 8) : "memory");
    else { a0 = src[l * 8]; a1 = src[l * 8 + 1]; a2 = src[l * 8 + 2]; a3 = src[l * 8 + 3]; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float f0 = l, f1 = 1.0f, f2 = 2.0f, f3 = 0.5f;
    frag b = src[l];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#define ONE(C, A)                                                                                                  \
    if (AG) asm volatile(MF("%0", "%1", "%2") : "+v"(C) : "a"(A), "v"(b)); else asm volatile(MF("%0", "%1", "%2") : "+v"(C) : "v"(A), "v"(b)); \
    if (NF >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f2), "v"(f3));                           \
    if (NF >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));                           \
    if (NF >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f2), "v"(f3));                           \
    if (NF >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));                           \
    if (NF >= 5) asm volatile("v_exp_f32 %0, %0" : "+v"(f0));                                                     \
    if (NF >= 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
            ONE(c0, a0) ONE(c1, a1) ONE(c2, a2) ONE(c3, a3)
        }
    }
    asm volatile("s_nop 9" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1;
}

template <typename K>
void run(const char* name, K k, const frag* src, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %7.1f cycles per 4-MFMA group  (%5.1f per MFMA)\n", name, (double)h / (iters * 8), (double)h / (iters * 32));
}

int main() {
    frag* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    run("A=VGPR", probe<0>, src, out, cyc);
    run("A=AGPR", probe<1>, src, out, cyc);
    run("A=VGPR + s_nop1", probe<2>, src, out, cyc);
    run("A=AGPR + s_nop1", probe<3>, src, out, cyc);
    run("A=VGPR + s_nop1 + LDS B", probe<6>, src, out, cyc);
    run("A=AGPR + s_nop1 + LDS B", probe<7>, src, out, cyc);
    run("A=AGPR + s_nop1 + LDS B + 8 VALU after group", probe<15>, src, out, cyc);
    run("A=AGPR + s_nop1 + 8 VALU after group", probe<11>, src, out, cyc);
    run("A=AGPR + 2 asm VALU after group", probe<17>, src, out, cyc);
    run("interleaved A=VGPR NF=0", probe_il<0, false>, src, out, cyc);
    run("interleaved A=AGPR NF=0", probe_il<0, true>, src, out, cyc);
    run("interleaved A=AGPR NF=1", probe_il<1, true>, src, out, cyc);
    run("interleaved A=AGPR NF=2", probe_il<2, true>, src, out, cyc);
    run("interleaved A=AGPR NF=3", probe_il<3, true>, src, out, cyc);
    run("interleaved A=AGPR NF=4", probe_il<4, true>, src, out, cyc);
    run("interleaved A=AGPR NF=5 (4 fma + exp)", probe_il<5, true>, src, out, cyc);
    run("interleaved A=AGPR NF=6", probe_il<6, true>, src, out, cyc);
    run("interleaved A=VGPR NF=3", probe_il<3, false>, src, out, cyc);
    return 0;
}
