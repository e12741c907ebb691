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
#include <cstdlib>
#include <cstring>

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
    const int M = a.M, N = a.N, K = a.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int splits = a.split_k > 1 ? a.split_k : 1;
    const int total_tiles = tiles_n * tiles_m * splits;
    // persistent tile loop (one iteration when the grid covers every tile)
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int bx = tile % tiles_n, by = (tile / tiles_n) % tiles_m, bz = tile / (tiles_n * tiles_m);
    const int m0 = by * BM, n0 = bx * BN;
    int kbeg = 0, kend = K;
    if (splits > 1) {
        const int per = ((K + splits - 1) / splits + BK - 1) / BK * BK;
        kbeg = bz * per;
        kend = min(K, kbeg + per);
        if (kbeg >= kend) continue;
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
            if (a.bias && (bz == 0 || !a.accumulate)) {
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
    }   // tile loop
    if (a.sys_release) __threadfence_system();
}


// ===========================================================================================================
// Fast path: bf16 operands stored in bf16 (or one-hot A), 128x128 tiles, M%128 == N%128 == K%64 == 0, 16-byte
// aligned rows.  Differences from the generic kernel above:
//   * global -> LDS staging is a straight 16-byte copy (no float round trip), issued for tile k+1 BEFORE the MFMAs
//     of tile k (register prefetch) into the other half of a double-buffered LDS image: one barrier per K tile;
//   * BK = 64: 32 MFMAs per wave per barrier;
//   * an operand that is row-contiguous in memory ([k][row]: both operands of every weight-gradient GEMM) is
//     staged AS IS and read with ds_read_b64_tr_b16, the LDS transpose read of gfx950 - the generic kernel
//     transposes with eight 2-byte LDS stores per 16 bytes loaded.
// ===========================================================================================================
#include "ablations.h"      // GEMM_ABL_* timing switches: all 0 in the product build
// s_sleep argument of the persistent GEMMs' chunk polls (x 64 clocks between two system-scope loads of the counter): a chunk arrives
// every ~40 us; 256 waves of such a launch polling every 0.2 us (8) is traffic on one memory channel that buys nothing
#ifndef GEMM_POLL_SLEEP
#define GEMM_POLL_SLEEP 32
#endif
constexpr int FBM = 128, FBN = 128, FBK = 64;
// k-contiguous image: [row][FBK + 8] (144-byte rows).  NOT conflict-free for ds_read_b128 - its four lane groups are non-contiguous
// ({0-3, 12-15, 20-27}, ...: rows {0-3, 12-15} of one k chunk and rows {4-11} of the next share a group; PMC: SQ_LDS_BANK_CONFLICT 32-40 %
// of the LDS cycles of the NT GEMMs and of proj_ws_k).  Round 3 measured the fix - 128-byte rows, chunk c of row r at c ^ ((r >> 1) & 7):
// 0 % conflicts - and it bought nothing (proj_ws_k 0.535 -> 0.535 ms, dX 0.431 -> 0.434, decode 4.12 -> 4.14 us per step: these kernels
// wait for their global loads and barriers, wait-any 38 %, not for LDS; profiles/r03_q_lds_swizzle.txt), while the split-K row
// reduction of the epilogue is laid out for 18 KB images: the padded image stays.
constexpr int F_LDK = FBK + 8;
constexpr int F_LDR = 128 + 16;       // row-contiguous image: [k][128 + 16] (288 B rows = 32 mod 256: tr reads spread)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// image row of k row k (0..63) of a row-contiguous operand tile, see f_frag
__device__ __forceinline__ int f_krow(int k) { return (k & 32) | ((k & 4) << 2) | ((k >> 1) & 12) | (k & 3); }

template <bool RC>   // element count of one operand image
constexpr int f_img() { return RC ? FBK * F_LDR : 128 * F_LDK; }

// fragment (8 consecutive k for row base+r, k-slot q) of k-group kg (32 k) from an operand image
template <bool RC>
__device__ __forceinline__ u16x8 f_frag(const bf16_t* img, int rbase, int kg, int q, int r) {
    if (!RC) return *reinterpret_cast<const u16x8*>(img + (rbase + r) * F_LDK + kg * 32 + q * 8);
    // lane r of a 16-lane group hands in the 8-byte chunk (k = k0 + r/4, rows rbase + 4*(r%4) ..+3); the transpose
    // read returns k0..k0+3 of row rbase + r
    // The image holds k row kg*32 + q*8 + h*4 + j at position kg*32 + h*16 + q*4 + j (f_krow): the 8 rows one half-wave
    // pass of a transpose read touches (j = 0..3 of two neighbouring q) are then 8 CONSECUTIVE image rows, 8 banks apart
    // (row pitch 72 dwords): all 64 banks once.  With the rows in natural order the two q of a pass are 8 rows = 576
    // dwords = 0 banks apart: every pass a 2-way conflict (SQ_LDS_BANK_CONFLICT = 1/3 of the LDS cycles).
    const bf16_t* p0 = img + (kg * 32 + q * 4 + (r >> 2)) * F_LDR + rbase + (r & 3) * 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * F_LDR));
    return u16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2],
                 (bf16_t)hi[3]};
}

// One thread's share of an operand tile: 128 rows x 64 k bf16 = 16 KiB = 1024 chunks of 16 B -> 4 per thread.
template <bool RC, bool ONEHOT>
struct f_stage {
    u16x8 v[4];
    // chunk c (0..1023):  KC: row = c / 8, k8 = c % 8 ;  RC: k = c / 16, row8 = c % 16
    // rlim: first row index that may not be read (row-contiguous operands narrower than the tile re-read their last
    // 8 columns instead of running into the next k row; those output columns are never stored)
    __device__ __forceinline__ void load(const void* base, int ld, int row0, int k0, int tid, int rlim = 1 << 30) {
        if (GEMM_ABL_NOLOAD) k0 = 0;           // every tile re-reads the first one (L2 hits)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            if (ONEHOT) {
                const int k = c >> 4, r8 = (c & 15) * 8;
                const int hot = (int)reinterpret_cast<const uint8_t*>(base)[k0 + k] - (row0 + r8);
                u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (hot == e) ? (bf16_t)0x3f80 : (bf16_t)0;
                v[i] = z;
            } else if (RC) {
                const int k = c >> 4, r8 = min(row0 + (c & 15) * 8, rlim - 8);
                v[i] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16_t*>(base) + (size_t)(k0 + k) * ld + r8);
            } else {
                const int row = c >> 3, k8 = (c & 7) * 8;
                v[i] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16_t*>(base) + (size_t)(row0 + row) * ld + k0 + k8);
            }
        }
    }
    __device__ __forceinline__ void store(bf16_t* img, int tid) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            if (RC) *reinterpret_cast<u16x8*>(img + f_krow(c >> 4) * F_LDR + (c & 15) * 8) = v[i];
            else *reinterpret_cast<u16x8*>(img + (c >> 3) * F_LDK + (c & 7) * 8) = v[i];
        }
    }
};

// (the body is a device function of a VIRTUAL grid - block vb of nvb: gemm_kstream_multi_k runs several problems in one launch)
template <bool A_RC, bool B_RC, bool ONEHOT, bool CS = false>      // CS: also the column sums of B (mvae_gemm_args.colsum_b)
__device__ __forceinline__ void gemm_fast_body(const mvae_gemm_args a, const int vb, const int nvb, unsigned char* smem) {
    constexpr int IA = f_img<A_RC>(), IB = f_img<B_RC>();
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);                  // [2][IA]
    bf16_t* Bs = As + 2 * IA;                                      // [2][IB]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, r = l & 15;
    const int wm = w >> 1, wn = w & 1;
    const int M = a.M, N = a.N, K = a.K;
    const int tiles_n = (N + FBN - 1) / FBN, tiles_m = (M + FBM - 1) / FBM;   // M < 128: one-hot table gradient
    const int nlim = B_RC ? (a.ldb < tiles_n * FBN ? a.ldb : 1 << 30) : 1 << 30;   // N < 128: head kernels (accumulate)
    const int splits = a.split_k > 1 ? a.split_k : 1;
    // persistent chunked mode: a fixed grid walks the M tiles chunk by chunk, handing over with device-side counters
    const int tiles_mc = a.chunk_rows ? a.chunk_rows / FBM : tiles_m;          // M tiles per chunk
    const int nchunks = tiles_m / tiles_mc;
    // Split-K with >= 8 splits (the weight-gradient GEMMs: 16 output tiles, K = T*B): the tiles of one k-range share their
    // A and B panels, so they go to ONE XCD (workgroup b runs on XCD b % 8) - spread over all eight, every XCD's L2 fetches
    // every panel of A (2.4x the unique bytes from HBM for a 256 x 1024 output).
    const bool xcd_split = splits >= 8 && !a.chunk_rows && (nvb % 8) == 0;
    // Persistent chunked mode (the projection / dX GEMM between two time-pipelined layers): the tiles_n column tiles of one row
    // block read the same 128 x K panel of A.  Dealt out round-robin they land on eight different XCDs and every XCD's L2 fetches
    // the panel from the fabric; with the row blocks of residue x (mod 8) given to the workgroups of XCD x the panel is fetched
    // once and its other tiles_n - 1 readers hit that L2 (the weight panels, 64 KB each, stay resident in every L2).
    const bool xcd_rows = a.chunk_rows && splits == 1 && (nvb % 8) == 0 && (tiles_mc % 8) == 0;
    const int total_tiles = tiles_n * tiles_mc * (xcd_split ? (splits + 7) / 8 * 8 : splits);
    // The output chunk a RUNNING consumer picks up (chunk_done) is stored write-through: no L2 write-back before the counter
    // (csrc/common.h wave_signal_done) - the write-back of an XCD's L2 also stalled the recurrent workgroups on that XCD.
    const bool wt = a.chunk_done && a.c_layout == MVAE_TILE16 && a.c_kind != MVAE_F32 && !a.accumulate && (N % FBN) == 0;
    for (int ci = 0; ci < nchunks; ++ci) {
    const int chunk = a.chunk_reverse ? nchunks - 1 - ci : ci;
    if (a.chunk_wait) wave_wait_ge<GEMM_POLL_SLEEP>(a.chunk_wait + chunk, a.chunk_wait_value, a.chunk_status, 3u);
    for (int tile = vb; tile < total_tiles; tile += nvb) {
        int bx = tile % tiles_n, by = chunk * tiles_mc + (tile / tiles_n) % tiles_mc, bz = tile / (tiles_n * tiles_mc);
        if (xcd_rows) {
            const int x = tile & 7, i = tile >> 3;          // the i-th tile of XCD x's share of this chunk
            bx = i % tiles_n;
            by = chunk * tiles_mc + x + 8 * (i / tiles_n);
            bz = 0;
        }
        if (xcd_split) {
            const int x = tile & 7, i = tile >> 3, per_k = tiles_n * tiles_mc, within = i % per_k;
            bz = (i / per_k) * 8 + x;
            bx = within % tiles_n;
            by = within / tiles_n;
            if (bz >= splits) continue;
        }
        const int m0 = by * FBM, n0 = bx * FBN;
        int kbeg = 0, kend = K;
        if (splits > 1 && !a.k_wait) {
            const int per = ((K + splits - 1) / splits + FBK - 1) / FBK * FBK;
            kbeg = bz * per;
            kend = min(K, kbeg + per);
            if (kbeg >= kend) continue;
        }
        // K-streaming (mvae_gemm_args.k_wait): the K range arrives chunk by chunk from a running producer; this workgroup takes
        // partition bz of every chunk and keeps accumulating in registers
        const int nseg = a.k_wait ? K / a.k_chunk_rows : 1;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // column sums of B (bias gradient): the M tile by == 0, its waves wm == 0, multiply their B fragments by an all-ones
        // fragment as well - 4 more MFMAs per 16, every row of the result is the column sum
        // (a compile-time variant, and EVERY wave of it does the 4 extra MFMAs: a run-time branch here would split the loop
        // body into basic blocks and undo the instruction interleaving below; only the waves named above publish theirs)
        const bool want_cs = CS && by == 0 && wm == 0;
        f32x4 accs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) accs[j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Main loop.  One wave per SIMD runs this (a 128x128 tile per CU), so nothing overlaps unless the instruction
        // stream itself interleaves: issued in source order - 8 global loads, 16 LDS reads, wait, 16 MFMAs, 16 LDS reads,
        // wait, 16 MFMAs, 8 LDS writes, barrier - an iteration takes ~3400 cycles for 512 cycles of MFMAs (the loads alone
        // hold the CU's address unit for ~860).  Hence:
        //   * prefetch distance TWO K tiles through registers (requested in iteration k, stored to LDS at the end of k+1),
        //     and lds_barrier() instead of __syncthreads(), which drains vmcnt(0) - every prefetch in flight;
        //   * branch-free iteration halves (clamped addresses at the tail) that the scheduler is told to interleave:
        //     k-group 0's MFMAs carry the global loads and k-group 1's LDS reads, k-group 1's MFMAs the LDS writes.
        f_stage<A_RC, ONEHOT> sa0, sa1;
        f_stage<B_RC, false> sb0, sb1;
        for (int seg = 0; seg < nseg; ++seg) {
        if (a.k_wait) {
            const int c = a.k_reverse ? nseg - 1 - seg : seg, part = a.k_chunk_rows / splits;
            if (w == 0) wave_wait_ge<64>(a.k_wait + c, a.k_wait_value, a.chunk_status, 4u);     // one polling wave per workgroup
            __syncthreads();
            kbeg = c * a.k_chunk_rows + bz * part;
            kend = kbeg + part;
        }
        const int ntiles = (kend - kbeg + FBK - 1) / FBK, klast = kbeg + (ntiles - 1) * FBK;
        sa0.load(a.A, a.lda, m0, kbeg, tid);
        sb0.load(a.B, a.ldb, n0, kbeg, tid, nlim);
        sa1.load(a.A, a.lda, m0, min(kbeg + FBK, klast), tid);
        sb1.load(a.B, a.ldb, n0, min(kbeg + FBK, klast), tid, nlim);
        __syncthreads();                       // previous output tile's readers are done with both images
        sa0.store(As, tid);
        sb0.store(Bs, tid);
        __syncthreads();
        int cur = 0;
        (void)seg;
        auto frags = [&](int buf, int kg, u16x8 (&fa)[4], u16x8 (&fb)[4]) __attribute__((always_inline)) {
            const bf16_t* Ai = As + buf * IA;
            const bf16_t* Bi = Bs + buf * IB;
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = f_frag<A_RC>(Ai, wm * 64 + i * 16, kg, q, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = f_frag<B_RC>(Bi, wn * 64 + j * 16, kg, q, r);
        };
        auto mfmas = [&](u16x8 (&fa)[4], u16x8 (&fb)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!GEMM_ABL_NOMFMA || (i == 0 && j == 0)) acc[i][j] = mfma_bf16(fb[j], fa[i], acc[i][j]);   // rows: n, cols: m
            if (CS) {
                const u16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
                for (int j = 0; j < 4; ++j) accs[j] = mfma_bf16(fb[j], ones, accs[j]);
            }
        };
        // one K tile from LDS[cur]; requests tile kreq into (la, lb); stores (sta, stb) - requested an iteration ago - into
        // the other LDS image
        auto half = [&](f_stage<A_RC, ONEHOT>& la, f_stage<B_RC, false>& lb, const f_stage<A_RC, ONEHOT>& sta,
                        const f_stage<B_RC, false>& stb, int kreq) __attribute__((always_inline)) {
            static_assert(FBK == 64, "two k-groups per tile");
            u16x8 fa0[4], fb0[4], fa1[4], fb1[4];
            frags(cur, 0, fa0, fb0);
            la.load(a.A, a.lda, m0, kreq, tid);
            lb.load(a.B, a.ldb, n0, kreq, tid, nlim);
            frags(cur, 1, fa1, fb1);
            mfmas(fa0, fb0);
            mfmas(fa1, fb1);
            sta.store(As + (cur ^ 1) * IA, tid);
            stb.store(Bs + (cur ^ 1) * IB, tid);
            if (!ONEHOT && !GEMM_ABL_NOMFMA) {
                // DS read 0x100, MFMA 0x008, VMEM read 0x020, DS write 0x200
                __builtin_amdgcn_sched_group_barrier(0x100, (A_RC ? 8 : 4) + (B_RC ? 8 : 4), 0);      // k-group 0 fragments
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, ((A_RC ? 8 : 4) + (B_RC ? 8 : 4)) / 8, 0);
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            lds_barrier();
            cur ^= 1;
        };
        int t = 0, k0 = kbeg;
        for (; t + 2 <= ntiles; t += 2, k0 += 2 * FBK) {
            // tile t is in LDS[cur], tile t+1 in flight to (sa1, sb1): request t+2 into (sa0, sb0), store (sa1, sb1)
            half(sa0, sb0, sa1, sb1, min(k0 + 2 * FBK, klast));
            // tile t+1 is in LDS[cur], tile t+2 in flight to (sa0, sb0): request t+3 into (sa1, sb1), store (sa0, sb0)
            half(sa1, sb1, sa0, sb0, min(k0 + 3 * FBK, klast));
        }
        if (t < ntiles) {                      // odd tile count: the last tile is in LDS[cur]
            u16x8 fa[4], fb[4];
#pragma unroll
            for (int kg = 0; kg < FBK / 32; ++kg) {
                frags(cur, kg, fa, fb);
                mfmas(fa, fb);
            }
        }
        }   // K segments
        __syncthreads();                       // every wave is done with the images: the epilogue / the next prologue reuse them
        if (CS && want_cs && r == 0) {     // lane (q, r = 0) holds the sums of columns .. + q*4 + 0..3
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + q * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) atomicAdd(a.colsum_b + n + e, accs[j][e]);
            }
        }
        // Split-K accumulation: straight from the accumulator layout an atomic instruction touches 64 addresses in 16 rows
        // (35 of the weight-gradient GEMM's 180 us).  The tile goes through LDS (the operand images are free now) and leaves
        // as whole rows: one atomic instruction = 256 contiguous bytes.
        if (a.accumulate && a.c_layout == MVAE_ROWMAJOR && !GEMM_ABL_NOATOMIC) {
            float* ct = reinterpret_cast<float*>(smem);                    // [128][132] f32 = 66 KiB <= 2 (IA + IB) bf16
            constexpr int CP = 132;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(ct + (wm * 64 + i * 16 + r) * CP + wn * 64 + j * 16 + q * 4) = acc[i][j] * a.alpha;
            __syncthreads();
            float* cb = reinterpret_cast<float*>(a.C);
            const int mlim = ONEHOT ? M : 1 << 30;
            for (int rr = w * 32; rr < w * 32 + 32; ++rr) {
                const int m = m0 + rr;
                if (m >= mlim) break;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int n = n0 + h2 * 64 + l;
                    if (n < N) atomicAdd(cb + (size_t)m * a.ldc + n, ct[rr * CP + h2 * 64 + l]);
                }
            }
            __syncthreads();                   // the next output tile's prologue writes the images
            continue;
        }
        // epilogue: lane holds C[m = .. + r][n = .. + q*4 + 0..3]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + r;
            if (ONEHOT && m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + q * 4;
                if (n >= N) continue;
                f32x4 v = acc[i][j] * a.alpha;
                if (a.bias && (bz == 0 || !a.accumulate)) v += *reinterpret_cast<const f32x4*>(a.bias + n);
                if (a.act == MVAE_ACT_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tanh_f(v[e]);
                }
                if (a.c_layout == MVAE_TILE16) {
                    const size_t off = ((((size_t)(m >> 4) * (N >> 4) + (n >> 4)) * 64) + (size_t)(q * 16 + (m & 15))) * 4;
                    if (wt) {       // handed over to a running kernel: write-through, relative to this tile's (uniform) origin
                        const size_t off0 = (((size_t)(m0 >> 4) * (N >> 4) + (n0 >> 4)) * 64) * 4;
                        store4_bf16_wt(reinterpret_cast<bf16_t*>(a.C) + off0, (unsigned)(off - off0) * 2u, v);
                    } else if (a.c_kind == MVAE_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.C) + off) = v;
                    else st<bf16_t>::store4(reinterpret_cast<bf16_t*>(a.C) + off, v);
                } else if (a.accumulate) {
                    float* cp = reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!GEMM_ABL_NOATOMIC && n + e < N) atomicAdd(cp + e, v[e]);
                } else if (a.c_kind == MVAE_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + n) = v;
                } else {
                    st<bf16_t>::store4(reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n, v);
                }
            }
        }
    }
    if (a.chunk_done) {
        if (wt) wave_signal_done<false>(a.chunk_done + chunk);
        else wave_signal_done<true>(a.chunk_done + chunk);
    }
    }   // chunk loop
    if (a.sys_release) __threadfence_system();
}
template <bool A_RC, bool B_RC, bool ONEHOT, bool CS = false>
__global__ __launch_bounds__(256) void gemm_fast_k(const mvae_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemm_fast_body<A_RC, B_RC, ONEHOT, CS>(a, (int)blockIdx.x, (int)gridDim.x, smem);
}
// Several K-streaming weight-gradient GEMMs (mvae_gemm_args.k_wait) behind ONE pipelined stack as ONE launch on ONE queue: every
// such GEMM runs for the whole BPTT, so each needs its own queue otherwise - and every additional busy queue costs the step
// 0.14 ms of command-processor time (profiles/r02_n_ab_kstream_gradients.txt).  Workgroups [base[i], base[i+1]) run problem i.
constexpr int KS_MAX = 8;
struct kstream_multi {
    mvae_gemm_args p[KS_MAX];
    int32_t base[KS_MAX + 1];
    int32_t variant[KS_MAX];     // 0: dense A, 1: dense A + column sums of B, 2: one-hot A
    int32_t n;
};
__global__ __launch_bounds__(256) void gemm_kstream_multi_k(const kstream_multi m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.base[i + 1]) ++i;
    const int vb = (int)blockIdx.x - m.base[i], nvb = m.base[i + 1] - m.base[i];
    if (m.variant[i] == 1) gemm_fast_body<true, true, false, true>(m.p[i], vb, nvb, smem);
    else if (m.variant[i] == 2) gemm_fast_body<true, true, true, false>(m.p[i], vb, nvb, smem);
    else gemm_fast_body<true, true, false, false>(m.p[i], vb, nvb, smem);
}

// Up to GM_MAX ordinary (not K-streaming) weight-gradient GEMMs - C (M,N) f32 += A (K,M)^T B (K,N), split-K, bf16 or one-hot A - as
// ONE launch (mvae_gemm_multi).  Short sequences (the reference's shipped T = 64: K = T*B = 16384 rows) make every such GEMM a
// 20-120 us launch of which most is fill and drain; a dozen of them on two queues beside and behind the recurrences was two thirds
// of the step's tail, and each one's workgroups kept the recurrent launches (one workgroup per EMPTY CU) from being placed.
// Workgroups [base[i], base[i] + wgs[i]) run problem i; every base is a multiple of 8, so that a problem's virtual block b still
// runs on XCD b % 8 (the split-K panel sharing of gemm_fast_body).
constexpr int GM_MAX = 16;
struct gemm_multi {
    mvae_gemm_args p[GM_MAX];
    int32_t base[GM_MAX + 1];
    int32_t wgs[GM_MAX];
    int32_t variant[GM_MAX];     // 0: dense A, 1: dense A + column sums of B, 2: one-hot A
    int32_t n;
};
__global__ __launch_bounds__(256) void gemm_multi_k(const gemm_multi m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.base[i + 1]) ++i;
    const int vb = (int)blockIdx.x - m.base[i], nvb = m.wgs[i];
    if (vb >= nvb) return;                      // (padding up to the next multiple of 8)
    if (m.variant[i] == 1) gemm_fast_body<true, true, false, true>(m.p[i], vb, nvb, smem);
    else if (m.variant[i] == 2) gemm_fast_body<true, true, true, false>(m.p[i], vb, nvb, smem);
    else gemm_fast_body<true, true, false, false>(m.p[i], vb, nvb, smem);
}

// ===========================================================================================================
// Weights-stationary persistent projection: x*W + b of a stacked layer behind a time-pipelined lower layer, K = H = 256.
// In the fast kernel above a 128x128 tile with K = 256 is four k-iterations: prologue latency, the reload of the same 64 KB weight
// panel for every tile and the epilogue dominate (~8 us per tile, 10 % of a CU's MFMA rate) - decoder inference at 1024 windows
// per GPU was bound by THIS GEMM, not by the recurrences.  A workgroup owns ONE column tile for the whole
// launch.
//   Round 2: weight panel AND the current A tile in LDS, the tile staged through registers behind two barriers, every phase
//     (stage, multiply, store) serial: 4.2 us per 128-row block, 128 TFLOP/s on 64 workgroups.
//   Round 3 (profiles/r03_t_proj_ws.txt for each step):
//   * each wave's 32 weight fragments (its 64 columns x K) live in ACCUMULATOR registers for the whole launch, the 16 output
//     accumulators too (named as "a" operands of inline-asm MFMAs: the compiler does not place MFMA operands there by itself):
//     half the LDS fragment reads are gone, and LDS holds TWO A tiles - the next tile is written while this one multiplies, one
//     barrier per row block;
//   * the A tile is requested with UN-TRACKED loads (inline asm) and waited for by hand.  gfx950 counts loads and stores in ONE
//     counter (vmcnt, retired in issue order); the compiler's own wait has to hold for every way into the loop and is vmcnt(0) -
//     every row block then also waited for the previous block's 16 write-through stores to be acknowledged by memory.  Here
//     the loads of block i+2 are issued in the middle of block i, in front of its stores, and `vmcnt(16)` in the middle of block
//     i+1 says "they have arrived" with those stores still in flight;
//   * LDS-only barriers (__syncthreads() releases global memory: vmcnt(0)).
//   Tried: no LDS at all, A fragments straight from global memory in MFMA layout (a lane's fragment is 16 contiguous bytes): a
//     wave's request then touches 16 rows x 64 bytes - 16 half-used cache lines per instruction, 3.5 us per block for the loads
//     alone (WS_ABL_* ablations) against 0.85 us of MFMAs.  Row-contiguous requests (8 full lines per instruction) + LDS it is.
// Row blocks of residue x (mod 8) belong to the workgroups of XCD x (as xcd_rows above): the eight column tiles that read the
// same A rows share an L2.  Same MFMA order over k as the fast kernel: bit-identical results.
// EPI: the output as 0 TILE16 written through (a pipelined stack's hand-over), 1 TILE16 plain stores, 2 row-major - a template
// parameter, not three run-time branches per store.  The build fails if this kernel uses scratch (an asm load's output must
// never be spilled before its data has landed): csrc/Makefile.
// ===========================================================================================================
constexpr int WS_K = 256, WS_KT = WS_K / FBK;       // K = H; k-tiles of 64
// the hand-counted wait: vmcnt(16 * SEL), SEL a wave-uniform 0 / 1 - ONE statement with the branch inside (two statements in an
// if / else make the compiler merge two versions of the 64 "modified" registers)
#define WS_WAIT(SEL, B)                                                                                                           \
    asm volatile("s_cmp_eq_u32 %16, 1\n\ts_cbranch_scc1 L_ws16_%=\n\ts_waitcnt vmcnt(0)\n\ts_branch L_wsd_%=\n"                    \
                 "L_ws16_%=:\n\ts_waitcnt vmcnt(16)\nL_wsd_%=:"                                                                     \
                 : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5]), "+v"(B[6]), "+v"(B[7]), "+v"(B[8]),      \
                   "+v"(B[9]), "+v"(B[10]), "+v"(B[11]), "+v"(B[12]), "+v"(B[13]), "+v"(B[14]), "+v"(B[15])                        \
                 : "s"(SEL)                                                                                                         \
                 : "memory", "scc")
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <bool ZERO>
__device__ __forceinline__ void ws_mfma(f32x4& c, const u16x8& wgt, const u16x8& x) {
    if (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(c) : "a"(wgt), "v"(x));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(wgt), "v"(x));
}
template <int EPI>
__global__ __launch_bounds__(256) void proj_ws_k(const mvae_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IMG = f_img<false>();
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);                  // [2][WS_KT][IMG]   two A tiles
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, q = l >> 4, r = l & 15;
    const int wm = w >> 1, wn = w & 1;
    const int N = a.N;
    const int tiles_n = N / FBN, tiles_m = a.M / FBM;
    const int tiles_mc = a.chunk_rows / FBM, nchunks = tiles_m / tiles_mc;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;             // XCD, index within the XCD's workgroups
    const int bx = j % tiles_n, g = j / tiles_n, G = (gridDim.x >> 3) / tiles_n;
    const int n0 = bx * FBN;
    // The first chunk is waited for BEFORE anything is read: the launch may sit on its queue ahead of the weight preparation of
    // its step (no event orders it - one packet less on the critical queue per phase); the producer it polls runs behind that
    // preparation, so a published chunk says the weights are in place.
    if (a.chunk_wait) wave_wait_ge<GEMM_POLL_SLEEP>(a.chunk_wait + (a.chunk_reverse ? nchunks - 1 : 0), a.chunk_wait_value, a.chunk_status, 3u);
    u16x8 fb[4][8];                      // this wave's weights: column tile jj, k-group ks (32 k) - accumulator registers
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const unsigned char* pb = reinterpret_cast<const unsigned char*>(a.B) +
                                  ((size_t)(n0 + wn * 64 + jj * 16 + r) * a.ldb + q * 8) * 2;
        asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:64\n\t"
                     "global_load_dwordx4 %2, %8, off offset:128\n\tglobal_load_dwordx4 %3, %8, off offset:192\n\t"
                     "global_load_dwordx4 %4, %8, off offset:256\n\tglobal_load_dwordx4 %5, %8, off offset:320\n\t"
                     "global_load_dwordx4 %6, %8, off offset:384\n\tglobal_load_dwordx4 %7, %8, off offset:448\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&a"(fb[jj][0]), "=&a"(fb[jj][1]), "=&a"(fb[jj][2]), "=&a"(fb[jj][3]), "=&a"(fb[jj][4]), "=&a"(fb[jj][5]),
                       "=&a"(fb[jj][6]), "=&a"(fb[jj][7])
                     : "v"(pb)
                     : "memory");
    }
    f32x4 bias[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
        bias[jj] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * 64 + jj * 16 + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool wt = EPI == 0;        // (see gemm_fast_k)
    u16x8 sa[16];                        // the tile in flight: chunk tid + i * 256 (row = chunk / 8, 8 k at (chunk % 8) * 8) of k-tile kt at [kt * 4 + i]
    f32x4 acc[4][4];
    const size_t row_bytes = (size_t)a.lda * 2;
    for (int ci = 0; ci < nchunks; ++ci) {
        const int chunk = a.chunk_reverse ? nchunks - 1 - ci : ci;
        if (a.chunk_wait) wave_wait_ge<GEMM_POLL_SLEEP>(a.chunk_wait + chunk, a.chunk_wait_value, a.chunk_status, 3u);
        const int nloc = tiles_mc >> 3;                            // row blocks of this chunk on this XCD
        const int nt = g < nloc ? (nloc - g + G - 1) / G : 0;      // ... of this workgroup
        auto block_row = [&](const int t) { return (chunk * tiles_mc + x + 8 * (g + t * G)) * FBM; };
        auto request = [&](const int t) __attribute__((always_inline)) {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(a.A) + (size_t)(block_row(t) + (tid >> 3)) * row_bytes +
                                     (size_t)((tid & 7) * 8) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned char* pi = p + (size_t)i * 32 * row_bytes;
                asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                             "global_load_dwordx4 %1, %4, off offset:128\n\t"
                             "global_load_dwordx4 %2, %4, off offset:256\n\t"
                             "global_load_dwordx4 %3, %4, off offset:384"
                             : "=&v"(sa[0 * 4 + i]), "=&v"(sa[1 * 4 + i]), "=&v"(sa[2 * 4 + i]), "=&v"(sa[3 * 4 + i])
                             : "v"(pi)
                             : "memory");
            }
        };
        // The MFMA pipe runs by itself once an MFMA is issued (16 cycles each, 4 to issue): what else a row block needs - the
        // previous block's epilogue, this wave's share of the next block's LDS image, the requests for the block after - is issued
        // BETWEEN the k-groups of the multiply instead of behind it (one wave per SIMD: nothing else would overlap it).
        //   first half  (k 0..127):   after k-group kk the epilogue of row tile kk of the PREVIOUS block (accumulators copied out)
        //   second half (k 128..255): after k-group kk the LDS writes of k-tile kk of block t+1, then the requests of k-tile kk of t+2
        auto lds_part = [&](const int t, const int kt) __attribute__((always_inline)) {
            bf16_t* img = As + (t & 1) * WS_KT * IMG + kt * IMG;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + i * 256;
                *reinterpret_cast<u16x8*>(img + (c >> 3) * F_LDK + (c & 7) * 8) = sa[kt * 4 + i];
            }
        };
        auto request_part = [&](const int t, const int kt) __attribute__((always_inline)) {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(a.A) + (size_t)(block_row(t) + (tid >> 3)) * row_bytes +
                                     (size_t)(kt * FBK + (tid & 7) * 8) * 2;
            asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                         "global_load_dwordx4 %1, %5, off\n\t"
                         "global_load_dwordx4 %2, %6, off\n\t"
                         "global_load_dwordx4 %3, %7, off"
                         : "=&v"(sa[kt * 4 + 0]), "=&v"(sa[kt * 4 + 1]), "=&v"(sa[kt * 4 + 2]), "=&v"(sa[kt * 4 + 3])
                         : "v"(p), "v"(p + 32 * row_bytes), "v"(p + 64 * row_bytes), "v"(p + 96 * row_bytes)
                         : "memory");
        };
        f32x4 cv[4][4];                  // the previous block's accumulators, read out of the accumulator registers
        auto epilogue_part = [&](const int t, const int i) __attribute__((always_inline)) {   // lane holds C[m = .. + r][n = .. + q*4 + 0..3]
            const int m0 = block_row(t);
            const int m = m0 + wm * 64 + i * 16 + r;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int n = n0 + wn * 64 + jj * 16 + q * 4;
                const f32x4 v = cv[i][jj] * a.alpha + bias[jj];
                if constexpr (EPI < 2) {
                    const size_t off = ((((size_t)(m >> 4) * (N >> 4) + (n >> 4)) * 64) + (size_t)(q * 16 + (m & 15))) * 4;
                    if constexpr (wt) {
                        const size_t off0 = (((size_t)(m0 >> 4) * (N >> 4) + (n0 >> 4)) * 64) * 4;
                        store4_bf16_wt(reinterpret_cast<bf16_t*>(a.C) + off0, (unsigned)(off - off0) * 2u, v);
                    } else st<bf16_t>::store4(reinterpret_cast<bf16_t*>(a.C) + off, v);
                } else {
                    st<bf16_t>::store4(reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n, v);
                }
            }
        };
        auto kgroup = [&](const int t, auto ks_c) __attribute__((always_inline)) {
            constexpr int ks = decltype(ks_c)::value;
            const bf16_t* img = As + (t & 1) * WS_KT * IMG;
            u16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = f_frag<false>(img + (ks >> 1) * IMG, wm * 64 + i * 16, ks & 1, q, r);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (WS_ABL_NOMFMA && (i || jj)) continue;
                    if (ks == 0) ws_mfma<true>(acc[i][jj], fb[jj][ks], fa[i]);      // rows: n, cols: m
                    else ws_mfma<false>(acc[i][jj], fb[jj][ks], fa[i]);
                }
        };
        auto read_out = [&]() __attribute__((always_inline)) {
            asm volatile("s_nop 9"
                         : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
                           "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]),
                           "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) cv[i][jj] = acc[i][jj];
        };
        if (nt > 0) {
            if (!WS_ABL_NOLOAD) request(0);
            WS_WAIT(0, sa);
#pragma unroll
            for (int kt = 0; kt < WS_KT; ++kt) lds_part(0, kt);
            if (nt > 1 && !WS_ABL_NOLOAD) request(1);
            lds_barrier();
        }
        for (int t = 0; t < nt; ++t) {
            // block t is in LDS[t & 1]; block t+1 in flight to the registers (requested in the second half of block t-1, or above)
            static_for<0, 4>([&](auto kk) __attribute__((always_inline)) {
                kgroup(t, kk);
                if (t > 0 && !WS_ABL_NOSTORE) epilogue_part(t - 1, decltype(kk)::value);
            });
            // younger than block t+1's loads: block t-1's 16 stores (none in the chunk's first block).  LDS[(t+1) & 1] was last read
            // by block t-1, before the barrier that ended it.
            if (t + 1 < nt) WS_WAIT(__builtin_amdgcn_readfirstlane(t > 0 && !WS_ABL_NOSTORE ? 1 : 0), sa);
            static_for<0, 4>([&](auto kk) __attribute__((always_inline)) {
                constexpr int k = decltype(kk)::value;
                kgroup(t, std::integral_constant<int, 4 + k>{});
                if (t + 1 < nt) {
                    lds_part(t + 1, k);
                    if (t + 2 < nt && !WS_ABL_NOLOAD) request_part(t + 2, k);
                }
            });
            read_out();
            lds_barrier();               // block t+1 is complete in LDS; every wave is done reading block t
        }
        if (nt > 0 && !WS_ABL_NOSTORE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) epilogue_part(nt - 1, i);
        }
        if (a.chunk_done) {
            if constexpr (wt) wave_signal_done<false>(a.chunk_done + chunk);
            else wave_signal_done<true>(a.chunk_done + chunk);
        }
    }
    if (a.sys_release) __threadfence_system();
}
#undef WS_WAIT
// the problems proj_ws_k takes: the forward projection of a pipelined stack (A (M,256) and B (N,256) k-contiguous bf16, bf16 output)
bool ws_ok(const mvae_gemm_args& a) {
    if (!a.chunk_rows || a.trans_a || !a.trans_b || a.K != WS_K || a.c_kind != MVAE_BF16 || a.accumulate ||
        a.act != MVAE_ACT_NONE || a.split_k > 1)
        return false;
    const int tiles_n = a.N / FBN, tiles_mc = a.chunk_rows / FBM;
    return a.max_blocks >= 8 * tiles_n && (a.max_blocks % (8 * tiles_n)) == 0 && (tiles_mc % 8) == 0;
}
int launch_ws(const mvae_gemm_args& a, hipStream_t s) {
    const size_t lds = (size_t)2 * WS_KT * f_img<false>() * sizeof(bf16_t);
    static bool raised = false;
    if (!raised) {
        for (const void* f : {reinterpret_cast<const void*>(&proj_ws_k<0>), reinterpret_cast<const void*>(&proj_ws_k<1>),
                              reinterpret_cast<const void*>(&proj_ws_k<2>)})
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MVAE_E_LAUNCH;
        raised = true;
    }
    const dim3 grid((unsigned)a.max_blocks);
    if (a.c_layout != MVAE_TILE16) hipLaunchKernelGGL(proj_ws_k<2>, grid, dim3(256), lds, s, a);
    else if (a.chunk_done) hipLaunchKernelGGL(proj_ws_k<0>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(proj_ws_k<1>, grid, dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

template <bool A_RC, bool B_RC, bool ONEHOT, bool CS = false>
int launch_fast(const mvae_gemm_args& a, hipStream_t s) {
    const size_t lds = (size_t)2 * (f_img<A_RC>() + f_img<B_RC>()) * sizeof(bf16_t);
    static bool raised = false;
    if (!raised && lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fast_k<A_RC, B_RC, ONEHOT, CS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    const int sk = a.split_k > 1 ? a.split_k : 1;
    long long tiles = (long long)((a.N + FBN - 1) / FBN) * ((a.M + FBM - 1) / FBM) * (sk >= 8 ? (sk + 7) / 8 * 8 : sk);
    if (a.max_blocks > 0 && tiles > a.max_blocks) tiles = a.max_blocks;
    if (a.chunk_rows) tiles = a.max_blocks;          // persistent grid: every workgroup passes through every chunk
    hipLaunchKernelGGL((gemm_fast_k<A_RC, B_RC, ONEHOT, CS>), dim3((unsigned)tiles), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

// true when the fast kernel can take this problem
bool fast_ok(const mvae_gemm_args& a) {
    const bool onehot = a.a_kind == MVAE_A_ONEHOT;
    if (!(a.b_kind == MVAE_BF16 && (a.a_kind == MVAE_BF16 || onehot))) return false;
    if (a.K % FBK) return false;
    if (a.N % FBN) {    // one narrow N tile: row-contiguous B whose rows hold >= 8 columns, accumulate mode, no bias
        if (!(a.N < FBN && !a.trans_b && a.accumulate && !a.bias && a.c_layout == MVAE_ROWMAJOR && a.ldb >= 8 &&
              a.ldb >= ((a.N + 7) / 8) * 8)) return false;
    }
    if (onehot) { if (a.M > FBM || !a.trans_a) return false; }          // one M tile; rows >= M are never hot
    else if (a.M % FBM) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!onehot && (!al16(a.A) || (a.lda % 8))) return false;
    if (!al16(a.B) || (a.ldb % 8)) return false;
    if (a.bias && !al16(a.bias)) return false;
    if (a.c_layout == MVAE_ROWMAJOR && !a.accumulate) {     // (accumulate = scalar atomics: any ldc)
        if (a.c_kind == MVAE_F32 && (!al16(a.C) || (a.ldc % 4))) return false;
        if (a.c_kind == MVAE_BF16 && ((reinterpret_cast<uintptr_t>(a.C) & 7) || (a.ldc % 4))) return false;
    }
    return true;
}

int dispatch_fast(const mvae_gemm_args& a, hipStream_t s) {
    const bool a_rc = a.trans_a != 0, b_rc = a.trans_b == 0;     // row-contiguous = [k][row] in memory
    if (a.a_kind == MVAE_A_ONEHOT)      // a full 128-row tile is computed; only rows < M are stored
        return b_rc ? launch_fast<true, true, true>(a, s) : launch_fast<true, false, true>(a, s);
    if (a_rc && b_rc && a.colsum_b) return launch_fast<true, true, false, true>(a, s);
    if (a_rc) return b_rc ? launch_fast<true, true, false>(a, s) : launch_fast<true, false, false>(a, s);
    return b_rc ? launch_fast<false, true, false>(a, s) : launch_fast<false, false, false>(a, s);
}

template <typename OT, int AKIND, int BKIND, bool TA, bool TB>
int launch(const mvae_gemm_args& a, hipStream_t s) {
    const int sk = a.split_k > 1 ? a.split_k : 1;
    if ((long long)a.M * a.N >= 256 * 1024 && a.M >= 128 && a.N >= 128) {
        long long tiles = (long long)((a.N + 127) / 128) * ((a.M + 127) / 128) * sk;
        if (a.max_blocks > 0 && tiles > a.max_blocks) tiles = a.max_blocks;
        hipLaunchKernelGGL((gemm_k<OT, AKIND, BKIND, TA, TB, 128, 128>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    } else {
        long long tiles = (long long)((a.N + 63) / 64) * ((a.M + 63) / 64) * sk;
        if (a.max_blocks > 0 && tiles > a.max_blocks) tiles = a.max_blocks;
        hipLaunchKernelGGL((gemm_k<OT, AKIND, BKIND, TA, TB, 64, 64>), dim3((unsigned)tiles), dim3(256), 0, s, a);
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


// the argument checks of a K-streaming problem (shared by mvae_gemm and mvae_gemm_kstream_multi); workgroups it needs in *wgs
int kstream_check(const mvae_gemm_args* a, long long* wgs) {
    const int sk = a->split_k > 1 ? a->split_k : 1;
    if (!a->k_wait || !a->trans_a || !a->accumulate || a->c_kind != MVAE_F32 || a->c_layout != MVAE_ROWMAJOR || a->k_chunk_rows <= 0 ||
        (a->K % a->k_chunk_rows) || a->max_blocks != 0 || a->chunk_rows || (a->k_chunk_rows % sk) || ((a->k_chunk_rows / sk) % FBK))
        return MVAE_E_ARG;
    // every workgroup must be resident at once (each one waits for every chunk)
    *wgs = (long long)((a->N + FBN - 1) / FBN) * ((a->M + FBM - 1) / FBM) * (sk >= 8 ? (sk + 7) / 8 * 8 : sk);
    if (*wgs > 256) return MVAE_E_ARG;
    if (!fast_ok(*a)) return MVAE_E_UNSUPPORTED;
    return MVAE_OK;
}

}  // namespace

extern "C" int mvae_gemm_kstream_multi(const mvae_gemm_args* problems, int32_t n, void* stream) {
    if (!problems || n <= 0 || n > KS_MAX) return MVAE_E_ARG;
    kstream_multi m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        const mvae_gemm_args* a = problems + i;
        if (!a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->trans_b || a->bias || a->act != MVAE_ACT_NONE)
            return MVAE_E_ARG;
        long long wgs = 0;
        const int rc = kstream_check(a, &wgs);
        if (rc != MVAE_OK) return rc;
        if (a->a_kind == MVAE_A_ONEHOT) m.variant[i] = 2;
        else if (a->colsum_b) {
            if (a->N % FBN) return MVAE_E_UNSUPPORTED;
            m.variant[i] = 1;
        } else m.variant[i] = 0;
        m.p[i] = *a;
        m.base[i] = (int32_t)total;
        total += wgs;
    }
    m.base[n] = (int32_t)total;
    if (total > 256) return MVAE_E_ARG;         // all of them wait for the producers: they must fit beside them
    const size_t lds = (size_t)2 * (f_img<true>() + f_img<true>()) * sizeof(bf16_t);
    static bool raised = false;
    if (!raised && lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kstream_multi_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL(gemm_kstream_multi_k, dim3((unsigned)total), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), m);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

extern "C" int mvae_gemm_multi(const mvae_gemm_args* problems, int32_t n, void* stream) {
    if (!problems || n <= 0 || n > GM_MAX) return MVAE_E_ARG;
    static_assert(sizeof(gemm_multi) <= 4096, "kernel arguments");
    gemm_multi m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        const mvae_gemm_args* a = problems + i;
        if (!a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->bias || a->act != MVAE_ACT_NONE) return MVAE_E_ARG;
        if (!a->trans_a || a->trans_b || !a->accumulate || a->c_kind != MVAE_F32 || a->c_layout != MVAE_ROWMAJOR || a->k_wait ||
            a->chunk_rows || a->chunk_wait || a->chunk_done || a->max_blocks != 0 || a->sys_release)
            return MVAE_E_UNSUPPORTED;
        if (!fast_ok(*a)) return MVAE_E_UNSUPPORTED;
        if (a->a_kind == MVAE_A_ONEHOT) m.variant[i] = 2;
        else if (a->colsum_b) {
            if ((a->N % FBN) || a->a_kind != MVAE_BF16) return MVAE_E_UNSUPPORTED;
            m.variant[i] = 1;
        } else m.variant[i] = 0;
        const int sk = a->split_k > 1 ? a->split_k : 1;
        const long long wgs = (long long)((a->N + FBN - 1) / FBN) * ((a->M + FBM - 1) / FBM) * (sk >= 8 ? (sk + 7) / 8 * 8 : sk);
        if (wgs > (1 << 20)) return MVAE_E_ARG;
        m.p[i] = *a;
        m.base[i] = (int32_t)total;
        m.wgs[i] = (int32_t)wgs;
        total += (wgs + 7) / 8 * 8;
    }
    m.base[n] = (int32_t)total;
    const size_t lds = (size_t)2 * (f_img<true>() + f_img<true>()) * sizeof(bf16_t);
    static bool raised = false;
    if (!raised && lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_multi_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL(gemm_multi_k, dim3((unsigned)total), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), m);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

extern "C" int mvae_occupancy(int32_t which) {
    int n = 0;
    hipError_t e;
    if (which == 0) {            // the dX GEMM between two pipelined layers: da (R, G*H) x W (H, G*H)^T, both k-contiguous
        const size_t lds = (size_t)2 * (f_img<false>() + f_img<false>()) * sizeof(bf16_t);
        const void* f = reinterpret_cast<const void*>(&gemm_fast_k<false, false, false, false>);
        if (lds > 64 * 1024 && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MVAE_E_LAUNCH;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, lds);
    } else if (which == 1) {
        const size_t lds = (size_t)2 * WS_KT * f_img<false>() * sizeof(bf16_t);
        const void* f = reinterpret_cast<const void*>(&proj_ws_k<0>);
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MVAE_E_LAUNCH;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, lds);
    } else if (which == 2) {
        const size_t lds = (size_t)2 * (f_img<true>() + f_img<true>()) * sizeof(bf16_t);
        const void* f = reinterpret_cast<const void*>(&gemm_kstream_multi_k);
        if (lds > 64 * 1024 && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MVAE_E_LAUNCH;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, lds);
    } else return MVAE_E_ARG;
    return e == hipSuccess ? n : MVAE_E_LAUNCH;
}

extern "C" int mvae_gemm(const mvae_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return MVAE_E_ARG;
    if (a->accumulate && a->c_kind != MVAE_F32) return MVAE_E_ARG;
    if (a->split_k > 1 && !a->accumulate) return MVAE_E_ARG;
    if (a->accumulate && a->act != MVAE_ACT_NONE) return MVAE_E_ARG;
    if (a->c_layout == MVAE_TILE16 && (a->accumulate || (a->M % 16) || (a->N % 16))) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->chunk_rows || a->chunk_wait || a->chunk_done) {      // persistent chunked mode: fast path only
        if (a->chunk_rows <= 0 || (a->chunk_rows % FBM) || (a->M % a->chunk_rows) || a->split_k > 1 || a->max_blocks <= 0 ||
            a->max_blocks > 256 || a->accumulate)
            return MVAE_E_ARG;
        if (!fast_ok(*a)) return MVAE_E_UNSUPPORTED;
    }
    if (a->k_wait) {                                            // K-streaming behind a running producer: fast path only
        long long wgs = 0;
        const int rc = kstream_check(a, &wgs);
        if (rc != MVAE_OK) return rc;
    }
    if (a->colsum_b && !(fast_ok(*a) && a->trans_a && !a->trans_b && a->accumulate && a->a_kind == MVAE_BF16 && !(a->N % FBN)))
        return MVAE_E_UNSUPPORTED;
    if (fast_ok(*a) && ws_ok(*a)) return launch_ws(*a, s);
    if (fast_ok(*a)) return dispatch_fast(*a, s);
    // A handful of output tiles with a long K (the Dense layers around the latent: M = batch, K up to nInit*H = 2304) is
    // a few workgroups marching through K for 100+ us on an otherwise idle chip: zero C and split K over atomics.
    mvae_gemm_args split;
    if (!a->accumulate && a->split_k <= 1 && a->c_kind == MVAE_F32 && a->c_layout == MVAE_ROWMAJOR &&
        a->act == MVAE_ACT_NONE && a->K >= 512 && (long long)((a->M + 63) / 64) * ((a->N + 63) / 64) <= 32 && a->ldc == a->N) {
        if (hipMemsetAsync(a->C, 0, (size_t)a->M * a->N * sizeof(float), s) != hipSuccess) return MVAE_E_LAUNCH;
        split = *a;
        split.accumulate = 1;
        split.split_k = a->K / 128 < 16 ? a->K / 128 : 16;
        a = &split;
    }
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
