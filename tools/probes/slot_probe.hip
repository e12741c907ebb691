// Development probe: cost of one MFMA slot (16x16x32 bf16, one wave per SIMD, 4 accumulators round-robin) with
// different fillers issued after every MFMA.  hipcc --offload-arch=gfx950 -O3 slot_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) unsigned short frag;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define MFMA(C) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C) : "a"(a0), "v"(b));
#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(k1), "v"(k2));
#define EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X));
#define RCP(X) asm volatile("v_rcp_f32 %0, %0" : "+v"(X));
#define CVT(X, Y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(X), "v"(Y));
#define NOP0 asm volatile("s_nop 0");
#define DSR asm volatile("ds_read_b128 %0, %1" : "=v"(lb) : "v"(loff));
#define DSW asm volatile("ds_write_b64 %0, %1" :: "v"(loff), "v"(st2));
#define GST asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(goff), "v"(st2), "s"(gout) : "memory");
#define GLD asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(ld2) : "v"(goff), "s"(gout) : "memory");
#define GST4 asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(goff4), "v"(lb), "s"(gout) : "memory");
#define GLD4 asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(lb) : "v"(goff4), "s"(gout) : "memory");
#define GST1 asm volatile("global_store_dword %0, %1, %2" :: "v"(goff1), "v"(pk), "s"(gout) : "memory");

template <int PAT>
__global__ __launch_bounds__(256, 1) void probe(const frag* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[16 * 256 * 2];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 256 * 2; i += 256) tile[i] = (unsigned short)(0x3c00 + (i & 7));
    __syncthreads();
    frag a0;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&a"(a0) : "v"(src + l) : "memory");
    frag b = src[l + 64], lb = b;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = 0.1f * l, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f, k1 = 0.999f, k2 = 0.001f;
    unsigned pk = 0, loff = (threadIdx.x * 16) & 16383, goff = threadIdx.x * 8, goff4 = threadIdx.x * 16, goff1 = threadIdx.x * 4;
    u32x2 st2 = {1u, 2u}, ld2 = {0u, 0u};
    char* gout = reinterpret_cast<char*>(out) + 65536;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#define SLOT(C, X, Y)                                                                                   \
    MFMA(C)                                                                                             \
    if (PAT == 1) { EXP(X) }                                                                            \
    if (PAT == 2) { EXP(X) FMA(Y) }                                                                     \
    if (PAT == 3) { EXP(X) FMA(Y) FMA(Y) }                                                              \
    if (PAT == 4) { RCP(X) FMA(Y) }                                                                     \
    if (PAT == 5) { EXP(X) EXP(Y) }                                                                     \
    if (PAT == 6) { CVT(X, Y) CVT(Y, X) }                                                               \
    if (PAT == 7) { DSR }                                                                               \
    if (PAT == 8) { GST }                                                                               \
    if (PAT == 9) { NOP0 FMA(X) }                                                                       \
    if (PAT == 10) { FMA(X) FMA(Y) }                                                                    \
    if (PAT == 11) { FMA(X) FMA(Y) FMA(X) }                                                             \
    if (PAT == 12) { FMA(X) FMA(Y) FMA(X) FMA(Y) }                                                      \
    if (PAT == 13) { DSW }                                                                              \
    if (PAT == 14) { GLD }                                                                              \
    if (PAT == 15) { EXP(X) FMA(Y) FMA(Y) FMA(Y) }                                                      \
    if (PAT == 16) { FMA(X) NOP0 NOP0 FMA(Y) }                                                          \
    if (PAT == 17) { DSR FMA(X) FMA(Y) }                                                                \
    if (PAT == 18) { GST4 }                                                                             \
    if (PAT == 19) { GLD4 }                                                                             \
    if (PAT == 20) { GST1 }                                                                             \
    if (PAT == 21) { if ((g & 1) == 0) { GLD4 } }                                                       \
    if (PAT == 22) { if ((g & 3) == 0) { GLD4 } }                                                       \
    if (PAT == 23) { if ((g & 3) == 0) { GLD } }
            SLOT(c0, x0, x1) SLOT(c1, x2, x3) SLOT(c2, x1, x0) SLOT(c3, x3, x2)
        }
        if (PAT == 7 || PAT == 17 || PAT == 14 || PAT == 8 || PAT >= 18) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 9" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + pk + lb[0] + ld2[0];
}

template <typename K>
void run(const char* name, K k, const frag* src, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h;
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f cycles per MFMA slot\n", name, (double)h / (iters * 32));
}
int main() {
    frag* src; float* out; unsigned long long* cyc;
    (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 0, 1 << 20); (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    run("MFMA only", probe<0>, src, out, cyc);
    run("+ exp", probe<1>, src, out, cyc);
    run("+ exp fma", probe<2>, src, out, cyc);
    run("+ exp fma fma", probe<3>, src, out, cyc);
    run("+ exp fma fma fma", probe<15>, src, out, cyc);
    run("+ rcp fma", probe<4>, src, out, cyc);
    run("+ exp exp", probe<5>, src, out, cyc);
    run("+ cvt_pk cvt_pk", probe<6>, src, out, cyc);
    run("+ fma fma", probe<10>, src, out, cyc);
    run("+ fma fma fma", probe<11>, src, out, cyc);
    run("+ fma fma fma fma", probe<12>, src, out, cyc);
    run("+ s_nop0 fma", probe<9>, src, out, cyc);
    run("+ fma s_nop0 s_nop0 fma", probe<16>, src, out, cyc);
    run("+ ds_read_b128", probe<7>, src, out, cyc);
    run("+ ds_read_b128 fma fma", probe<17>, src, out, cyc);
    run("+ ds_write_b64", probe<13>, src, out, cyc);
    run("+ global_store_dwordx2 (saddr)", probe<8>, src, out, cyc);
    run("+ global_load_dwordx2 (saddr)", probe<14>, src, out, cyc);
    run("+ global_store_dwordx4", probe<18>, src, out, cyc);
    run("+ global_load_dwordx4", probe<19>, src, out, cyc);
    run("+ global_store_dword", probe<20>, src, out, cyc);
    run("+ global_load_dwordx4 every 8th slot", probe<21>, src, out, cyc);
    run("+ global_load_dwordx4 every 16th slot", probe<22>, src, out, cyc);
    run("+ global_load_dwordx2 every 16th slot", probe<23>, src, out, cyc);
    return 0;
}
