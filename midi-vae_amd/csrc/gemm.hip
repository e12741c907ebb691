// General matrix product on the gfx950 matrix cores for the time-parallel parts of the MIDI-VAE step:
// input projections of stacked recurrent layers (x_t W for all t at once), every parameter gradient
// (sum over all (t,b) rows), the Dense layers of the encoder tail / decoder initial states
// (reference vae_definition.py:484,487,506-507,563-567) and the one-hot "scatter" gradient of an index table.
//
//   C (M,N) = alpha * opA(A) (M,K) * opB(B) (K,N)  [+ bias(N)] [tanh]      row-major, leading dimensions
//
// Both operands are staged through LDS k-contiguous ([rows][BK] + 16 B pad -> conflict-free ds_read_b128),
// whichever way they lie in HBM; the MFMA is issued with the B tile as its row operand so that a lane's four
// accumulator values are four consecutive n of one m: the epilogue is one 8/16-byte store per fragment.
// f32 operands use v_mfma_f32_16x16x4_f32 (exact f32, parity mode), bf16 operands v_mfma_f32_16x16x32_bf16.
#include "common.h"

namespace {

constexpr int BK = 32;

template <typename OT> struct lpad { static constexpr int value = 16 / sizeof(OT); };

// element fetch with conversion to float; KIND: 0 f32, 1 bf16
template <int KIND> struct src;
template <> struct src<MVAE_F32> {
    typedef float type;
    static __device__ __forceinline__ void load8(const float* p, float* v) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
};
template <> struct src<MVAE_BF16> {
    typedef bf16_t type;
    static __device__ __forceinline__ void load8(const bf16_t* p, float* v) {
        const u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf2f(a[i]);
    }
    static __device__ __forceinline__ float load1(const bf16_t* p) { return bf2f(*p); }
};

template <typename OT> __device__ __forceinline__ void lds_store8(OT* dst, const float* v);
template <> __device__ __forceinline__ void lds_store8<float>(float* dst, const float* v) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void lds_store8<bf16_t>(bf16_t* dst, const float* v) {
    u16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f2bf(v[i]);
    *reinterpret_cast<u16x8*>(dst) = r;
}

// Stage one (ROWS x BK) operand tile into LDS as [row][k].
//   KCONTIG: the operand is k-contiguous in memory, element (row, k) at base[row*ld + k]
//   else   : row-contiguous,                         element (row, k) at base[k*ld + row]
template <typename OT, int KIND, bool KCONTIG, int ROWS>
__device__ __forceinline__ void stage_tile(OT* lds, const typename src<KIND>::type* base, int ld, int row0, int nrows,
                                           int k0, int kend, int tid) {
    constexpr int LD = BK + lpad<OT>::value;
    typedef typename src<KIND>::type ST;
    if (KCONTIG) {
        constexpr int CH = ROWS * (BK / 8);
        for (int c = tid; c < CH; c += 256) {
            const int rr = c / (BK / 8), kc = (c % (BK / 8)) * 8;
            const int row = row0 + rr, k = k0 + kc;
            float v[8];
            const ST* p = base + (size_t)row * ld + k;
            if (row < nrows && k + 8 <= kend && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                src<KIND>::load8(p, v);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (row < nrows && k + i < kend) ? src<KIND>::load1(p + i) : 0.0f;
            }
            lds_store8<OT>(lds + rr * LD + kc, v);
        }
    } else {
        constexpr int CH = BK * (ROWS / 8);
        for (int c = tid; c < CH; c += 256) {
            const int kk = c / (ROWS / 8), rc = (c % (ROWS / 8)) * 8;
            const int row = row0 + rc, k = k0 + kk;
            float v[8];
            const ST* p = base + (size_t)k * ld + row;
            if (k < kend && row + 8 <= nrows && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                src<KIND>::load8(p, v);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (k < kend && row + i < nrows) ? src<KIND>::load1(p + i) : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) st<OT>::store(lds + (rc + i) * LD + kk, v[i]);
        }
    }
}

// one-hot A^T: A(m,k) = (idx[k] == m)
template <typename OT, int ROWS>
__device__ __forceinline__ void stage_onehot(OT* lds, const uint8_t* idx, int row0, int nrows, int k0, int kend,
                                             int tid) {
    constexpr int LD = BK + lpad<OT>::value;
    for (int c = tid; c < BK * (ROWS / 8); c += 256) {
        const int kk = c / (ROWS / 8), rc = (c % (ROWS / 8)) * 8;
        const int k = k0 + kk;
        const int hot = (k < kend) ? (int)idx[k] - (row0 + rc) : -1;
#pragma unroll
        for (int i = 0; i < 8; ++i) st<OT>::store(lds + (rc + i) * LD + kk, (hot == i && row0 + rc + i < nrows) ? 1.0f : 0.0f);
    }
}

template <typename OT, int AKIND, int BKIND, bool TA, bool TB, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_k(const mvae_gemm_args a) {
    constexpr int LD = BK + lpad<OT>::value;
    constexpr int KG = op<OT>::KG, FE = op<OT>::FRAG_ELEMS;
    constexpr int MI = BM / 32, NI = BN / 32;
    typedef typename op<OT>::frag frag;
    __shared__ __attribute__((aligned(16))) OT As[BM * LD];
    __shared__ __attribute__((aligned(16))) OT Bs[BN * LD];

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, r = l & 15;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int M = a.M, N = a.N, K = a.K;
    int kbeg = 0, kend = K;
    if (a.split_k > 1) {
        const int per = ((K + a.split_k - 1) / a.split_k + BK - 1) / BK * BK;
        kbeg = blockIdx.z * per;
        kend = min(K, kbeg + per);
        if (kbeg >= kend) return;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (AKIND == MVAE_A_ONEHOT)
            stage_onehot<OT, BM>(As, reinterpret_cast<const uint8_t*>(a.A), m0, M, k0, kend, tid);
        else
            stage_tile<OT, (AKIND == MVAE_A_ONEHOT ? MVAE_F32 : AKIND), !TA, BM>(
                As, reinterpret_cast<const typename src<(AKIND == MVAE_A_ONEHOT ? MVAE_F32 : AKIND)>::type*>(a.A),
                a.lda, m0, M, k0, kend, tid);
        stage_tile<OT, BKIND, TB, BN>(Bs, reinterpret_cast<const typename src<BKIND>::type*>(a.B), a.ldb, n0, N, k0,
                                      kend, tid);
        __syncthreads();
#pragma unroll
        for (int kg = 0; kg < BK / KG; ++kg) {
            frag fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const frag*>(As + (wm * (BM / 2) + i * 16 + r) * LD + kg * KG + q * FE);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                fb[j] = *reinterpret_cast<const frag*>(Bs + (wn * (BN / 2) + j * 16 + r) * LD + kg * KG + q * FE);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = op<OT>::mma(fb[j], fa[i], acc[i][j]);   // rows: n, cols: m
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m = .. + r][n = .. + q*4 + 0..3]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + q * 4;
            if (n >= N) continue;
            f32x4 v = acc[i][j] * a.alpha;
            if (a.bias && (blockIdx.z == 0 || !a.accumulate)) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) v[e] += a.bias[n + e];
            }
            if (a.act == MVAE_ACT_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tanh_f(v[e]);
            }
            if (a.c_layout == MVAE_TILE16) {
                // the lane's four values are exactly its slot of the 16x16 tile image: one contiguous wave store
                const size_t off = ((((size_t)(m >> 4) * (N >> 4) + (n >> 4)) * 64) + (size_t)(q * 16 + (m & 15))) * 4;
                if (a.c_kind == MVAE_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.C) + off) = v;
                else st<bf16_t>::store4(reinterpret_cast<bf16_t*>(a.C) + off, v);
            } else if (a.accumulate) {
                float* cp = reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + n;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) atomicAdd(cp + e, v[e]);
            } else if (a.c_kind == MVAE_F32) {
                float* cp = reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + n;
                if (n + 4 <= N && ((reinterpret_cast<uintptr_t>(cp) & 15) == 0)) {
                    *reinterpret_cast<f32x4*>(cp) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) cp[e] = v[e];
                }
            } else {
                bf16_t* cp = reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n;
                if (n + 4 <= N && ((reinterpret_cast<uintptr_t>(cp) & 7) == 0)) {
                    st<bf16_t>::store4(cp, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) cp[e] = f2bf(v[e]);
                }
            }
        }
    }
}

template <typename OT, int AKIND, int BKIND, bool TA, bool TB>
int launch(const mvae_gemm_args& a, hipStream_t s) {
    const int sk = a.split_k > 1 ? a.split_k : 1;
    if ((long long)a.M * a.N >= 256 * 1024 && a.M >= 128 && a.N >= 128) {
        dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, sk);
        hipLaunchKernelGGL((gemm_k<OT, AKIND, BKIND, TA, TB, 128, 128>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((a.N + 63) / 64, (a.M + 63) / 64, sk);
        hipLaunchKernelGGL((gemm_k<OT, AKIND, BKIND, TA, TB, 64, 64>), grid, dim3(256), 0, s, a);
    }
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

template <typename OT, int AKIND, int BKIND>
int by_trans(const mvae_gemm_args& a, hipStream_t s) {
    if (AKIND == MVAE_A_ONEHOT) {
        if (!a.trans_a) return MVAE_E_ARG;
        return a.trans_b ? launch<OT, AKIND, BKIND, true, true>(a, s) : launch<OT, AKIND, BKIND, true, false>(a, s);
    }
    if (a.trans_a) return a.trans_b ? launch<OT, AKIND, BKIND, true, true>(a, s) : launch<OT, AKIND, BKIND, true, false>(a, s);
    return a.trans_b ? launch<OT, AKIND, BKIND, false, true>(a, s) : launch<OT, AKIND, BKIND, false, false>(a, s);
}

}  // namespace

extern "C" int mvae_gemm(const mvae_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return MVAE_E_ARG;
    if (a->accumulate && a->c_kind != MVAE_F32) return MVAE_E_ARG;
    if (a->split_k > 1 && !a->accumulate) return MVAE_E_ARG;
    if (a->accumulate && a->act != MVAE_ACT_NONE) return MVAE_E_ARG;
    if (a->c_layout == MVAE_TILE16 && (a->accumulate || (a->M % 16) || (a->N % 16))) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ak = a->a_kind, bk = a->b_kind;
    // operand type on the matrix cores: bf16 if any stored operand is bf16, else exact f32
    if (ak == MVAE_F32 && bk == MVAE_F32) return by_trans<float, MVAE_F32, MVAE_F32>(*a, s);
    if (ak == MVAE_BF16 && bk == MVAE_BF16) return by_trans<bf16_t, MVAE_BF16, MVAE_BF16>(*a, s);
    if (ak == MVAE_A_ONEHOT && bk == MVAE_F32) return by_trans<float, MVAE_A_ONEHOT, MVAE_F32>(*a, s);
    if (ak == MVAE_A_ONEHOT && bk == MVAE_BF16) return by_trans<bf16_t, MVAE_A_ONEHOT, MVAE_BF16>(*a, s);
    if (ak == MVAE_BF16 && bk == MVAE_F32) return by_trans<bf16_t, MVAE_BF16, MVAE_F32>(*a, s);
    if (ak == MVAE_F32 && bk == MVAE_BF16) return by_trans<bf16_t, MVAE_F32, MVAE_BF16>(*a, s);
    return MVAE_E_ARG;
}
