// Round 6 probe: the weight-gradient GEMM  C[M][N] += A^T B  with both operands row-contiguous ([k][row], the
// layout the recurrent kernels leave: hs (T*B, H), da (T*B, G*H)) on a 256 x 256 x 64 tile with 8 waves - two per SIMD, each a
// 128 x 64 output tile - instead of gemm_fast_k's 128 x 128 with 4.  global_load_lds (16 bytes per lane, the LDS image lane-linear,
// the swizzle on the SOURCE address), ds_read_b64_tr_b16 fragments, one barrier per K tile, split-K with f32 atomics.
//   hipcc --offload-arch=gfx950 -O3 -o gemm_tn256_probe gemm_tn256_probe.hip && ./gemm_tn256_probe [M N K splits]
#include <hip/hip_runtime.h>
#ifndef PIPE_FRAGS
#define PIPE_FRAGS 0
#endif
#ifndef STAGGER
#define STAGGER 0
#endif
#ifndef NO_RD
#define NO_RD 0
#endif
#ifndef NO_MM
#define NO_MM 0
#endif
#ifndef FAST_ADDR
#define FAST_ADDR 1
#endif
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int IMG = BK * 256 * 2;              // one operand image: [64 k][256 rows] bf16 = 32 KiB, k rows of 512 bytes

__device__ __forceinline__ f32x4 mfma(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 32-byte granule g of k row kr sits at granule position g ^ sw(kr): the 8 k rows one 32-lane pass of a transpose read touches
// (j = kr & 3 of two neighbouring q = bit 3) land in 8 different 32-byte bank groups = all 64 banks once
__device__ __forceinline__ int sw(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

// fragment of rows rbase .. rbase+15 (rbase % 16 == 0), k = kg*32 + q*8 .. +7
__device__ __forceinline__ u16x8 frag(const unsigned char* img, int rbase, int kg, int q, int r) {
    const int kr0 = kg * 32 + q * 8 + (r >> 2), kr1 = kr0 + 4;
    const int g = rbase >> 4;
    const unsigned char* p0 = img + kr0 * 512 + ((g ^ sw(kr0)) << 5) + (r & 3) * 8;
    const unsigned char* p1 = img + kr1 * 512 + ((g ^ sw(kr1)) << 5) + (r & 3) * 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    return u16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
}

// The same read from a PRECOMPUTED per-lane offset: kr & 3 and (kr >> 3) & 1 - the swizzle's inputs - do not depend on the k-group or on
// the half (kr = kg*32 + q*8 + h*4 + (r >> 2)), so the offset of fragment g is (lane part ^ (g << 5)) + kg*16384 + h*2048: one register
// per fragment row block, computed once per launch (frag() above costs ~20 VALU instructions per call: 1000 per K tile and wave)
__device__ __forceinline__ unsigned lane_off(int q, int r) {
    const int kr = q * 8 + (r >> 2);
    return (unsigned)(kr * 512 + (sw(kr) << 5) + (r & 3) * 8);
}
__device__ __forceinline__ u16x8 frag_at(const unsigned char* img, unsigned off, int kg) {
    const unsigned char* p0 = img + off + kg * 16384;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 2048));
    return u16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
}

// (__syncthreads() drains vmcnt(0) - every global_load_lds in flight; the staggered loop's barriers order LDS reads only)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool ATOMIC>
__global__ __launch_bounds__(512) void gemm_tn256_k(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                    float* __restrict__ C, int ldc, int M, int N, int K, int splits, int hot) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];     // [2 buffers][A image, B image]
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, q = l >> 4, r = l & 15;
    const int wm = w >> 2, wn = w & 3;
    const int tiles_n = N / BN, tiles = tiles_n * (M / BM);
    // the tiles of one k range share their panels: neighbours in the grid (one XCD takes workgroups b, b+8, ...)
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int per_xcd = (tiles * splits + 7) / 8;
    const int work = xcd * per_xcd + slot;                 // consecutive work items on one XCD
    if (work >= tiles * splits || slot >= per_xcd) return;
    const int tile = work % tiles, sp = work / tiles;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int per = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    const int kbeg = sp * per, kend = min(K, kbeg + per);
    if (kbeg >= kend) return;
    const int ntiles = (kend - kbeg) / BK;

    // staging: an operand image is 32 blocks of 1 KiB (2 k rows); wave w stages blocks 4w .. 4w+3 of each operand
    const int c = l & 31, half = l >> 5;
    auto stage = [&](int buf, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = w * 4 + i, kr = 2 * b + half;
            const int chunk = ((((c >> 1) ^ sw(kr)) << 1) | (c & 1)) * 8;          // first row of the 16-byte chunk this lane fetches
            const bf16_t* sa = A + (size_t)(k0 + kr) * lda + m0 + chunk;
            const bf16_t* sb = B + (size_t)(k0 + kr) * ldb + n0 + chunk;
            unsigned char* da = smem + buf * (2 * IMG) + b * 1024;
            unsigned char* db = da + IMG;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa, (__attribute__((address_space(3))) void*)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb, (__attribute__((address_space(3))) void*)db, 16, 0, 0);
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned offA[8], offB[4];
    {
        const unsigned lo = lane_off(q, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) offA[i] = lo ^ (unsigned)((wm * 8 + i) << 5);
#pragma unroll
        for (int j = 0; j < 4; ++j) offB[j] = lo ^ (unsigned)((wn * 4 + j) << 5);
    }
    stage(0, kbeg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
#if STAGGER
    // The two waves of a SIMD (w and w + 4: wave groups G0 = rows 0..127, G1 = rows 128..255 of the tile) run ONE barrier interval apart
    // - the same code, G1 behind one extra barrier at the start (G0 one at the end): while one group reads its fragments from LDS the
    // other issues its 32 MFMAs.  Intervals of K tile t:   G0:  R(t,0) | M | R(t,1) | M        G1:  - | R(t,0) | M | R(t,1) | M
    // A wave waits for its share of the next image behind its R(t,1): G0's first read of tile t+1 comes two barriers later.
    u16x8 fa[8], fb[4];
    const bool g1 = w >= 4;
    auto rd = [&](const unsigned char* Ai, const unsigned char* Bi, int kg) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = FAST_ADDR ? frag_at(Bi, offB[j], kg) : frag(Bi, wn * 64 + j * 16, kg, q, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = FAST_ADDR ? frag_at(Ai, offA[i], kg) : frag(Ai, wm * 128 + i * 16, kg, q, r);
    };
    auto mm = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma(fb[j], fa[i], acc[i][j]);
    };
    if (g1) lds_barrier();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage(cur ^ 1, hot ? kbeg : kbeg + (t + 1) * BK);
        const unsigned char* Ai = smem + cur * (2 * IMG);
        const unsigned char* Bi = Ai + IMG;
        if (!NO_RD || t == 0) rd(Ai, Bi, 0);
        lds_barrier();
        if (!NO_MM) mm();
        lds_barrier();
        if (!NO_RD) rd(Ai, Bi, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (!NO_MM || t == ntiles - 1) mm();
        lds_barrier();
        cur ^= 1;
    }
    if (!g1) lds_barrier();
#elif PIPE_FRAGS
    // register double buffering at k-group granularity: the fragments of k-group g+1 are requested before the MFMAs of k-group g,
    // across the K-tile boundary too (the next tile's image is complete behind the barrier at the end of the iteration)
    u16x8 fa[2][8], fb[2][4];
    auto load = [&](int set, const unsigned char* Ai, const unsigned char* Bi, int kg) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = FAST_ADDR ? frag_at(Bi, offB[j], kg) : frag(Bi, wn * 64 + j * 16, kg, q, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[set][i] = FAST_ADDR ? frag_at(Ai, offA[i], kg) : frag(Ai, wm * 128 + i * 16, kg, q, r);
    };
    auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma(fb[set][j], fa[set][i], acc[i][j]);
    };
    load(0, smem, smem + IMG, 0);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage(cur ^ 1, hot ? kbeg : kbeg + (t + 1) * BK);
        const unsigned char* Ai = smem + cur * (2 * IMG);
        const unsigned char* Bi = Ai + IMG;
        load(1, Ai, Bi, 1);
        mma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // the next tile's image is complete (and nobody reads k-group 0 of this one any more)
        cur ^= 1;
        if (t + 1 < ntiles) load(0, smem + cur * (2 * IMG), smem + cur * (2 * IMG) + IMG, 0);
        mma(1);
        // (k-group 1 of the old image was read before the barrier: the NEXT iteration's stage() may overwrite it - but only after
        //  every wave has passed this point's ... the barrier above already separates those reads from the next stage())
    }
#else
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage(cur ^ 1, hot ? kbeg : kbeg + (t + 1) * BK);     // hot: every tile re-reads the first one (L2 hits)
        const unsigned char* Ai = smem + cur * (2 * IMG);
        const unsigned char* Bi = Ai + IMG;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            u16x8 fa[8], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = FAST_ADDR ? frag_at(Bi, offB[j], kg) : frag(Bi, wn * 64 + j * 16, kg, q, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = FAST_ADDR ? frag_at(Ai, offA[i], kg) : frag(Ai, wm * 128 + i * 16, kg, q, r);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma(fb[j], fa[i], acc[i][j]);      // rows: n, cols: m
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#endif
    // lane holds C[m = .. + r][n = .. + q*4 + 0..3]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + q * 4;
            float* cp = C + (size_t)m * ldc + n;
            if (ATOMIC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(cp + e, acc[i][j][e]);
            } else *reinterpret_cast<f32x4*>(cp) = acc[i][j];
        }
    }
}

__global__ void ref_k(const bf16_t* A, int lda, const bf16_t* B, int ldb, float* C, int ldc, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        s += __uint_as_float((unsigned)A[(size_t)k * lda + m] << 16) * __uint_as_float((unsigned)B[(size_t)k * ldb + n] << 16);
    C[(size_t)m * ldc + n] = s;
}
static bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (bf16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1048576;
    int splits = argc > 4 ? atoi(argv[4]) : 64;
    bf16_t *A, *B; float *C, *R;
    CK(hipMalloc(&A, (size_t)K * M * 2)); CK(hipMalloc(&B, (size_t)K * N * 2));
    CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&R, (size_t)M * N * 4));
    {   // small-integer-ish values: exact in bf16, sums stay well inside f32
        std::vector<bf16_t> h((size_t)K * (M > N ? M : N));
        srand(1);
        for (size_t i = 0; i < (size_t)K * M; ++i) h[i] = f2bf((float)((rand() % 9) - 4) * 0.125f);
        CK(hipMemcpy(A, h.data(), (size_t)K * M * 2, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)K * N; ++i) h[i] = f2bf((float)((rand() % 7) - 3) * 0.25f);
        CK(hipMemcpy(B, h.data(), (size_t)K * N * 2, hipMemcpyHostToDevice));
    }
    CK(hipFuncSetAttribute((const void*)gemm_tn256_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * IMG));
    const int tiles = (M / BM) * (N / BN), grid = ((tiles * splits + 7) / 8) * 8;
    // check on a short K first (the reference kernel is slow)
    const int Kc = K < 4096 ? K : 4096;
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    gemm_tn256_k<true><<<grid, 512, 4 * IMG>>>(A, M, B, N, C, N, M, N, Kc, splits < Kc / BK ? splits : Kc / BK, 0);
    ref_k<<<dim3((N + 255) / 256, M), 256>>>(A, M, B, N, R, N, M, N, Kc);
    CK(hipDeviceSynchronize());
    std::vector<float> hc((size_t)M * N), hr((size_t)M * N);
    CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), R, hr.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mr = 0;
    for (size_t i = 0; i < hc.size(); ++i) { md = fmax(md, fabs(hc[i] - hr[i])); mr = fmax(mr, fabs(hr[i])); }
    printf("check K=%d: max |diff| %.3g of max |ref| %.3g  %s\n", Kc, md, mr, md <= 1e-3 * mr ? "OK" : "WRONG");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int hot = argc > 5 ? atoi(argv[5]) : 0;
    for (int sp : {splits / 2, splits, splits * 2}) {
        if (sp < 1) continue;
        const int g = ((tiles * sp + 7) / 8) * 8;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(C, 0, (size_t)M * N * 4));
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it) gemm_tn256_k<true><<<g, 512, 4 * IMG>>>(A, M, B, N, C, N, M, N, K, sp, hot);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%sM=%d N=%d K=%d splits=%d (%d workgroups): %.3f ms  %.1f TFLOP/s\n", hot ? "[L2-hot operands] " : "", M, N, K, sp, g, ms, 2.0 * M * N * K / ms * 1e-9);
    }
    return 0;
}
