// Shared device helpers for the gfx950 kernels of libmidivae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/midivae_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef unsigned short bf16_t;  // raw bf16 bits

#define MVAE_WAVE 64

// ---- scalar conversions (round-to-nearest-even; NaN not special-cased: inputs are finite) -----------------
// f32 -> bf16, round to nearest even.  Written as a cast to the native type so hipcc emits v_cvt_pk_bf16_f32
// (one instruction per TWO values) instead of a 5-instruction integer sequence per value.
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, static_cast<__bf16>(f)); }
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// ---- storage-type traits ---------------------------------------------------------------------------------
template <typename T> struct st;   // storage traits
template <> struct st<float> {
    typedef f32x4 vec4;            // 4 consecutive elements
    static __device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct st<bf16_t> {
    static __device__ __forceinline__ f32x4 load4(const bf16_t* p) {
        u16x4 r = *reinterpret_cast<const u16x4*>(p);
        f32x4 v = {bf2f(r[0]), bf2f(r[1]), bf2f(r[2]), bf2f(r[3])};
        return v;
    }
    static __device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4);
    }
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- activations (Keras semantics) ---------------------------------------------------
__device__ __forceinline__ float hard_sigmoid(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.0f), 1.0f); }
// derivative from the OUTPUT y: 0.2 strictly inside (0,1)
// derivative in terms of the OUTPUT y in [0, 1]: 0.2 * [0 < y < 1].  Without compares (2 VALU instead of 3 + 1 SALU):
// y - y^2 > 0 exactly for the interior (0 and 1 are exact), and med3(2^100 * (y - y^2), 0, 0.2) is 0.2 times that.
__device__ __forceinline__ float dhard_sigmoid(float y) {
    const float sq = y - y * y;
    return __builtin_amdgcn_fmed3f(sq * 0x1p100f, 0.0f, 0.2f);
}
__device__ __forceinline__ float tanh_f(float x) { return tanhf(x); }   // ocml: accurate near 0 (parity mode)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// bf16-mode tanh: 1 - 2/(1+e^{2x}) on the hardware exp2 / rcp units (5 VALU instead of ocml's ~40).  Absolute
// error ~1e-7 (cancellation near 0), far below the bf16 quantisation of everything this feeds.
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // e^{2x}
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ---- MFMA wrappers: D(16x16) = A(16xK) * B(Kx16) + C ------------------------------------------------------
// lane l:  A[row = l&15][k-slot = l>>4],  B[k-slot = l>>4][col = l&15],  C[row = (l>>4)*4 + i][col = l&15]
// bf16: a k-slot is 8 consecutive k (K = 32);   f32: a k-slot is ONE k (K = 4).
__device__ __forceinline__ f32x4 mfma_bf16(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}
__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One "k-group" of the contraction = 32 k for bf16 (one MFMA), 16 k for f32 (four MFMAs; lane slot q holds
// k = 16 s + 4 q + i, i = 0..3, so both operands are read as 4 consecutive floats).
template <typename T> struct op;
template <> struct op<bf16_t> {
    static constexpr int KG = 32;               // k per group
    typedef u16x8 frag;                         // per-lane fragment: 8 consecutive k
    static constexpr int FRAG_ELEMS = 8;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return mfma_bf16(a, b, c); }
};
template <> struct op<float> {
    static constexpr int KG = 16;
    typedef f32x4 frag;                         // 4 consecutive k
    static constexpr int FRAG_ELEMS = 4;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        c = mfma_f32(a[0], b[0], c);
        c = mfma_f32(a[1], b[1], c);
        c = mfma_f32(a[2], b[2], c);
        c = mfma_f32(a[3], b[3], c);
        return c;
    }
};

__host__ __device__ constexpr int mvae_gates(int cell) { return cell == MVAE_GRU ? 3 : (cell == MVAE_LSTM ? 4 : 1); }

// 16-lane-group reductions (lanes sharing l>>4 = one DPP row), every lane gets the result.  Data-parallel-primitive
// moves instead of __shfl_xor (= ds_bpermute: an LDS round trip per step - the head kernels' softmax is a chain of 20 of
// them per row): xor 1, xor 2 inside the quads, then the half-row and the row mirrored.  Both operands of every step are
// the two partial results swapped, so all 16 lanes end up with the same bits.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ int group16_min_i(int v) {
    v = min(v, dpp_i<DPP_XOR1>(v));
    v = min(v, dpp_i<DPP_XOR2>(v));
    v = min(v, dpp_i<DPP_HALF_MIRROR>(v));
    v = min(v, dpp_i<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Workgroup barrier for data exchanged through LDS ONLY.  __syncthreads() also releases global memory, i.e. it
// drains vmcnt(0): every in-flight prefetch load and every pending store of saved activations - once per time
// step that exposes a full HBM round trip (measured: 2 us of a 3.3 us step).  Here only LDS traffic is waited for.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#define MVAE_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return MVAE_E_LAUNCH;          \
    } while (0)

// ---- device-side hand-over between RUNNING kernels (time-pipelined stacks, include/midivae_hip.h) -------------------
// Single asm blocks with scalar control flow and one or two temporary VGPRs: as C++ (a thread-0 loop, barriers, an
// atomic) they cost the 256-VGPR LSTM backward kernel registers it does not have, and an extra basic block in its step
// loop breaks hipcc's allocation.  Every WAVE waits / publishes by itself: no barrier; a counter's consumer expects one
// increment per producer wave.
//
// wave_wait_ge: until *flag >= value (system-scope loads), then drop this XCD's possibly stale cache lines of the data the
// flag guards.  Bounded (~2-4 s: a host that stalls in the middle of enqueuing a step must not look like a dead producer): then *status = code (which kind of wait: 1 recurrent forward, 2 BPTT, 3 chunked GEMM, 4 K-streaming GEMM, 5 join) and the kernel carries on - it never hangs.
#ifdef MVAE_EXP_AGENT_INV
#define MVAE_ACQ_INV "buffer_inv sc1"
#else
#define MVAE_ACQ_INV "buffer_inv sc0 sc1"
#endif
template <int SLEEP = 8>        // (64: a throughput consumer that polls for most of its producer's run time)
__device__ __forceinline__ void wave_wait_ge(const uint32_t* flag, uint32_t value, uint32_t* status, uint32_t code = 1u) {
    // One give-up time (~3 s) for every waiter whatever its SLEEP: a look costs its system-scope load (~1.2 us) PLUS the
    // pause (SLEEP x 64 clocks), so the bound is 0x200000 x 53 / (45 + SLEEP) looks - 2.1 M at SLEEP 8, 1.4 M at 32, 1.0 M at 64.
    // (Round 6 first scaled it by 8 / SLEEP as if the pause were all of a look: the K-streaming GEMMs' SLEEP = 64 waiter then gave
    //  up after 0.8 s, and the live-producer test (tests/test_ops_gpu.py) lost a run now and then to a slow first enqueue on a fresh box.)
    constexpr unsigned LIMIT = (unsigned)(0x200000ull * 53ull / (45ull + (unsigned long long)(SLEEP > 0 ? SLEEP : 1)));
    unsigned tmp, spins, val;
    asm volatile(
        "s_mov_b32 %1, 0\n"
        "L_wait_%=:\n\t"
        "v_mov_b32 %0, 0\n\t"
        "global_load_dword %0, %0, %3 sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_readfirstlane_b32 %2, %0\n\t"
        "s_cmp_ge_u32 %2, %4\n\t"
        "s_cbranch_scc1 L_ready_%=\n\t"
        "s_sleep %5\n\t"
        "s_add_u32 %1, %1, 1\n\t"
        "s_cmp_lt_u32 %1, %6\n\t"
        "s_cbranch_scc1 L_wait_%=\n"
        "L_ready_%=:\n\t"
        MVAE_ACQ_INV
        : "=&v"(tmp), "=&s"(spins), "=&s"(val)
        : "s"(flag), "s"(value), "n"(SLEEP), "n"(LIMIT)
        : "memory", "scc");
    if (spins >= LIMIT && status) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the same under a scalar condition evaluated inside the block: if (t == bound) wait
__device__ __forceinline__ void wave_wait_ge_if(int t, int bound, const uint32_t* flag, uint32_t value, uint32_t* status, uint32_t code = 1u) {
    unsigned tmp, spins, val;
    asm volatile(
        "s_mov_b32 %1, 0\n\t"
        "s_cmp_lg_u32 %3, %4\n\t"
        "s_cbranch_scc1 L_skip_%=\n"
        "L_wait_%=:\n\t"
        "v_mov_b32 %0, 0\n\t"
        "global_load_dword %0, %0, %5 sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_readfirstlane_b32 %2, %0\n\t"
        "s_cmp_ge_u32 %2, %6\n\t"
        "s_cbranch_scc1 L_ready_%=\n\t"
        "s_sleep 8\n\t"
        "s_add_u32 %1, %1, 1\n\t"
        "s_cmp_lt_u32 %1, 0x200000\n\t"
        "s_cbranch_scc1 L_wait_%=\n"
        "L_ready_%=:\n\t"
        MVAE_ACQ_INV "\n"
        "L_skip_%=:"
        : "=&v"(tmp), "=&s"(spins), "=&s"(val)
        : "s"(t), "s"(bound), "s"(flag), "s"(value)
        : "memory", "scc");
    if (spins >= 0x200000u && status) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wave_signal_done<WB>: this wave's global stores so far are complete, then ONE increment (by its first lane) of the counter.
// WB = true: the wave's data left through PLAIN stores - dirty lines of this XCD's L2 are written back first (buffer_wbl2; it
// writes back EVERY dirty line of that L2, also the other workgroups': measured 7 % of a train step when the recurrent
// kernels published each 32-step chunk that way).  WB = false: the producer stored the handed-over data WRITE-THROUGH
// (store16_wt below) - nothing of it is left in L2, the drained vmcnt is the release.
#define MVAE_SIGNAL_ASM(PRE, WB_INSN, POST)                                                                          \
    PRE "s_waitcnt vmcnt(0)\n\t" WB_INSN "s_waitcnt vmcnt(0)\n\t"                                                    \
        "s_mov_b64 %2, exec\n\t"                                                                                     \
        "s_mov_b64 exec, 1\n\t"                                                                                      \
        "v_mov_b32 %0, 0\n\t"                                                                                        \
        "v_mov_b32 %1, 1\n\t" POST
template <bool WB = true>
__device__ __forceinline__ void wave_signal_done(uint32_t* counter) {
    unsigned t0, t1;
    unsigned long long save;
    if (WB)
        asm volatile(MVAE_SIGNAL_ASM("", "buffer_wbl2 sc0 sc1\n\t", "global_atomic_add %0, %1, %3 sc1\n\t"
                                                                     "s_mov_b64 exec, %2\n\t"
                                                                     "s_waitcnt vmcnt(0)")
                     : "=&v"(t0), "=&v"(t1), "=&s"(save)
                     : "s"(counter)
                     : "memory");
    else
        asm volatile(MVAE_SIGNAL_ASM("", "", "global_atomic_add %0, %1, %3 sc1\n\t"
                                             "s_mov_b64 exec, %2\n\t"
                                             "s_waitcnt vmcnt(0)")
                     : "=&v"(t0), "=&v"(t1), "=&s"(save)
                     : "s"(counter)
                     : "memory");
}
template <bool WB = true>
__device__ __forceinline__ void wave_signal_done_if(int t, int bound, uint32_t* counter) {
    unsigned t0, t1;
    unsigned long long save;
    if (WB)
        asm volatile(MVAE_SIGNAL_ASM("s_cmp_lg_u32 %3, %4\n\t"
                                     "s_cbranch_scc1 L_skip_%=\n\t",
                                     "buffer_wbl2 sc0 sc1\n\t",
                                     "global_atomic_add %0, %1, %5 sc1\n\t"
                                     "s_mov_b64 exec, %2\n\t"
                                     "s_waitcnt vmcnt(0)\n"
                                     "L_skip_%=:")
                     : "=&v"(t0), "=&v"(t1), "=&s"(save)
                     : "s"(t), "s"(bound), "s"(counter)
                     : "memory", "scc");
    else
        asm volatile(MVAE_SIGNAL_ASM("s_cmp_lg_u32 %3, %4\n\t"
                                     "s_cbranch_scc1 L_skip_%=\n\t",
                                     "",
                                     "global_atomic_add %0, %1, %5 sc1\n\t"
                                     "s_mov_b64 exec, %2\n\t"
                                     "s_waitcnt vmcnt(0)\n"
                                     "L_skip_%=:")
                     : "=&v"(t0), "=&v"(t1), "=&s"(save)
                     : "s"(t), "s"(bound), "s"(counter)
                     : "memory", "scc");
}
// 16-byte write-through store (buffer_store_dwordx4 ... sc1) to uniform base + per-lane byte offset: the bytes go to memory
// and are dropped from this XCD's L2, so a consumer on any XCD reads them after the producer's vmcnt has drained - the
// hand-over needs no L2 write-back.  A compiler-known instruction (waitcnt and hazard bookkeeping stay correct).
typedef unsigned int mvae_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mvae_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16_wt(const void* uniform_base, unsigned lane_off, u16x8 v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_base), 0, -1, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mvae_u32x4, v), r, (int)lane_off, 0, 16 /* sc1 */);
}
// the same for 4 bf16 of a GEMM epilogue (f32 accumulators rounded like st<bf16_t>::store4)
__device__ __forceinline__ void store4_bf16_wt(const void* uniform_base, unsigned lane_off, f32x4 v) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_base), 0, -1, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mvae_u32x2, __builtin_convertvector(v, bf16x4)), r, (int)lane_off, 0, 16);
}

// ---- weight preparation bodies (shared by the single kernels and the batched mvae_prepare_batch launch) -----------
// block `bid` of `nb` blocks (256 threads) of a grid-stride loop
template <typename WT>
__device__ __forceinline__ void pack_recurrent_body(const float* __restrict__ U, WT* __restrict__ out, int H, int GH, int direction,
                                                    int bid, int nb) {
    constexpr int KG = op<WT>::KG, FE = op<WT>::FRAG_ELEMS;
    const int rowsA = direction == 0 ? GH : H;   // A rows
    const int K = direction == 0 ? H : GH;       // contraction length
    const int S = K / KG;
    // one thread = one lane's fragment (FE consecutive k): 32-bit index arithmetic once per FE elements, one 16-byte store
    const unsigned groups = (unsigned)((size_t)rowsA * K / FE);
    for (unsigned g = (unsigned)bid * blockDim.x + threadIdx.x; g < groups; g += (unsigned)nb * blockDim.x) {
        const unsigned l = g & 63u, f = g >> 6, s = f % (unsigned)S, mt = f / (unsigned)S;
        const unsigned arow = mt * 16 + (l & 15), k0 = s * KG + (l >> 4) * FE;
        typename op<WT>::frag v;
#pragma unroll
        for (int j = 0; j < FE; ++j) {
            const float x = direction == 0 ? U[(size_t)(k0 + j) * GH + arow]      // A[gate col][h]   = U[h][gate col]
                                           : U[(size_t)arow * GH + k0 + j];      // A[unit][gate col] = U[unit][gate col]
            st<WT>::store(reinterpret_cast<WT*>(&v) + j, x);
        }
        *reinterpret_cast<typename op<WT>::frag*>(out + (size_t)g * FE) = v;
    }
}
template <typename D>
__device__ __forceinline__ void make_table_body(const float* W, const float* bias, D* table, int K, int N, int bid, int nb,
                                                int paired = 0) {
    // paired (MVAE_TABLE_PAIRED; N % 32 == 0): inside every block of 32 columns the two 16-column tiles are interleaved per lane -
    // column 16 h + 4 q + e sits at 8 q + 4 h + e - so that one lane's values of a tile PAIR are 16 contiguous bytes (bf16): the
    // indexed-input slot-interleaved kernels gather 8 x 16 bytes per row and step instead of 16 x 8
    const size_t n = (size_t)K * N;
    for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < n; e += (size_t)nb * blockDim.x) {
        const int c = (int)(e % N);
        // (paired == 2, MVAE_TABLE_PAIRED8; N % 256 == 0: the tiles j and j + 8 of every block of 256 columns)
        const int cp = paired == 2 ? (c & ~255) + ((c >> 4) & 7) * 32 + ((c & 15) >> 2) * 8 + ((c >> 7) & 1) * 4 + (c & 3)
                     : paired ? (c & ~31) + ((c & 15) >> 2) * 8 + ((c >> 4) & 1) * 4 + (c & 3) : c;
        st<D>::store(table + (e - c) + cp, W[e] + bias[c]);
    }
}
template <typename D>
__device__ __forceinline__ void transpose_convert_body(const float* W, D* out, int K, int N, int NPAD, int bid, int nb) {
    const size_t n = (size_t)NPAD * K;
    for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < n; e += (size_t)nb * blockDim.x) {
        const int nn = (int)(e / K), k = (int)(e % K);
        st<D>::store(out + e, nn < N ? W[(size_t)k * N + nn] : 0.0f);
    }
}
template <typename D>
__device__ __forceinline__ void convert_f32_body(const float* src, D* dst, size_t n, int bid, int nb) {
    for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < n; e += (size_t)nb * blockDim.x) st<D>::store(dst + e, src[e]);
}
