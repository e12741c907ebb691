// Development probe (round 6): what does a SECOND wave on a SIMD buy in instruction issue?  One workgroup on one CU, W waves
// per SIMD (256 / 512 threads), every wave runs  { MFMA 16x16x32 bf16 ; K fillers }  x N  on four accumulators round-robin.
// Printed: cycles per MFMA as the SIMD sees it (wave-0 s_memtime span / MFMAs issued on that SIMD).
//   hipcc --offload-arch=gfx950 -O3 issue2_probe.hip -o issue2_probe && ./issue2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) unsigned short frag;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(C) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C) : "a"(a0), "v"(b));
#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(k1), "v"(k2));
#define PKF(X) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(kk1), "v"(kk2));
#define EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X));
#define DSR asm volatile("ds_read_b128 %0, %1" : "=v"(lb) : "v"(loff));
#define SNOP asm volatile("s_nop 0");
#define SMOV asm volatile("s_mov_b32 %0, 5" : "=s"(sdummy));
#define WAITL asm volatile("s_waitcnt lgkmcnt(15)");

typedef __attribute__((ext_vector_type(2))) float f32x2;

// PAT: 0..6 = that many v_fma per MFMA; 10 = 1 ds_read_b128 per MFMA; 11 = 1 ds_read + 2 fma; 12 = 2 s_nop; 13 = 2 s_mov;
//      14 = 2 s_waitcnt lgkmcnt(15); 20..26: BURST - 8 MFMAs, then 8 x (PAT - 20) v_fma; 30..33: (PAT-30) v_pk_fma_f32 per MFMA;
//      40..43: (PAT-40) v_exp per MFMA;  50: 2 fma + 1 ds_read + 1 s_nop
template <int PAT, int TPB>
__global__ __launch_bounds__(TPB, 1) void probe(const frag* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[16 * 256 * 2];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 256 * 2; i += TPB) tile[i] = (unsigned short)(0x3c00 + (i & 7));
    __syncthreads();
    frag a0;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&a"(a0) : "v"(src + l) : "memory");
    frag b = src[l + 64], lb = b;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = 0.1f * l, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f, k1 = 0.999f, k2 = 0.001f;
    f32x2 p0 = {0.1f, 0.2f}, p1 = {0.3f, 0.4f}, kk1 = {0.999f, 0.999f}, kk2 = {0.001f, 0.001f};
    unsigned loff = (threadIdx.x * 16) & 16383;
    unsigned sdummy = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#define FILL(X, Y)                                                                                      \
    if (PAT >= 1 && PAT <= 6) { FMA(X) }                                                                \
    if (PAT >= 2 && PAT <= 6) { FMA(Y) }                                                                \
    if (PAT >= 3 && PAT <= 6) { FMA(X) }                                                                \
    if (PAT >= 4 && PAT <= 6) { FMA(Y) }                                                                \
    if (PAT >= 5 && PAT <= 6) { FMA(X) }                                                                \
    if (PAT >= 6 && PAT <= 6) { FMA(Y) }                                                                \
    if (PAT == 10) { DSR }                                                                              \
    if (PAT == 11) { DSR FMA(X) FMA(Y) }                                                                \
    if (PAT == 12) { SNOP SNOP }                                                                        \
    if (PAT == 13) { SMOV SMOV }                                                                        \
    if (PAT == 14) { WAITL WAITL }                                                                      \
    if (PAT >= 31 && PAT <= 33) { PKF(p0) }                                                             \
    if (PAT >= 32 && PAT <= 33) { PKF(p1) }                                                             \
    if (PAT >= 33 && PAT <= 33) { PKF(p0) }                                                             \
    if (PAT >= 41 && PAT <= 43) { EXP(X) }                                                              \
    if (PAT >= 42 && PAT <= 43) { EXP(Y) }                                                              \
    if (PAT >= 43 && PAT <= 43) { EXP(X) }                                                              \
    if (PAT == 50) { FMA(X) DSR FMA(Y) SNOP }
#define SLOT(C, X, Y) MFMA(C) FILL(X, Y)
            SLOT(c0, x0, x1) SLOT(c1, x2, x3) SLOT(c2, x1, x0) SLOT(c3, x3, x2)
        }
        if (PAT >= 20 && PAT <= 26) {
#pragma unroll
            for (int k = 0; k < 8 * (PAT - 20); ++k) { if (k & 1) { FMA(x1) } else { FMA(x0) } }
        }
        if (PAT == 10 || PAT == 11 || PAT == 50) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 9" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + lb[0] + p0[0] + p1[1] + sdummy;
}

template <int PAT>
void run(const char* name, const frag* src, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    double per[2];
    float ms[2];
    for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 2; ++r) {
            hipEventRecord(e0, 0);
            if (wps == 1) hipLaunchKernelGGL((probe<PAT, 256>), dim3(1), dim3(256), 0, 0, src, out, cyc, iters);
            else hipLaunchKernelGGL((probe<PAT, 512>), dim3(1), dim3(512), 0, 0, src, out, cyc, iters);
            hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
        }
        unsigned long long c;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        hipEventElapsedTime(&ms[wps - 1], e0, e1);
        per[wps - 1] = (double)c / ((double)iters * 8 * wps);
    }
    printf("%-44s 1 wave/SIMD %6.1f   2 waves/SIMD %6.1f   cycles per MFMA of the SIMD   (wall %.3f / %.3f ms)\n", name, per[0], per[1], ms[0], ms[1]);
}

int main() {
    frag* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    hipMemset(src, 0x3c, 1 << 20);
    run<0>("bare MFMA", src, out, cyc);
    run<1>("+1 v_fma per MFMA", src, out, cyc);
    run<2>("+2 v_fma", src, out, cyc);
    run<3>("+3 v_fma", src, out, cyc);
    run<4>("+4 v_fma", src, out, cyc);
    run<5>("+5 v_fma", src, out, cyc);
    run<6>("+6 v_fma", src, out, cyc);
    run<31>("+1 v_pk_fma_f32", src, out, cyc);
    run<32>("+2 v_pk_fma_f32", src, out, cyc);
    run<33>("+3 v_pk_fma_f32", src, out, cyc);
    run<41>("+1 v_exp_f32", src, out, cyc);
    run<42>("+2 v_exp_f32", src, out, cyc);
    run<43>("+3 v_exp_f32", src, out, cyc);
    run<10>("+1 ds_read_b128", src, out, cyc);
    run<11>("+1 ds_read_b128 +2 v_fma", src, out, cyc);
    run<12>("+2 s_nop 0", src, out, cyc);
    run<13>("+2 s_mov", src, out, cyc);
    run<14>("+2 s_waitcnt", src, out, cyc);
    run<50>("+2 v_fma +1 ds_read +1 s_nop", src, out, cyc);
    run<21>("burst: 8 MFMA then 8 v_fma", src, out, cyc);
    run<22>("burst: 8 MFMA then 16 v_fma", src, out, cyc);
    run<23>("burst: 8 MFMA then 24 v_fma", src, out, cyc);
    run<24>("burst: 8 MFMA then 32 v_fma", src, out, cyc);
    run<26>("burst: 8 MFMA then 48 v_fma", src, out, cyc);
    return 0;
}
