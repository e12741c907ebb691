// Output heads of the MIDI-VAE decoder, fused over all (t, b) rows at once:
//   Dense(H -> N) + softmax + Keras categorical cross-entropy + accuracy + argmax + d(logits)   (notes, instruments)
//   Dense(H -> 1) + sigmoid + squared error + binary accuracy + d(logit)                         (velocity)
// Replaces Dense(activation) on the top decoder cell (reference vae_definition.py:542,593,631), the losses
// (:338,:367,:374) with Keras' weighted-objective semantics, the 'accuracy' metrics (:339)
// and sample_vector(...,'argmax') of the decode path (reference vae_definition.py:1048-1067).
//
// One wave = 16 rows; logits come off the matrix cores with A = rows of h (read straight from HBM, k-contiguous)
// and B = W^T (L2-resident), so a row's N logits sit across the 16 lanes of a lane group and the softmax is a
// 4-step butterfly; nothing goes through LDS.  The (R, N) probability tensor is written only on request.
#include "common.h"

namespace {

constexpr float CE_EPS = 1e-7f;

// fragment (8 bf16 / 4 f32 consecutive k) from f32 values, or zeros
template <typename WT>
__device__ __forceinline__ typename op<WT>::frag frag_from(const float* src, bool valid);
template <>
__device__ __forceinline__ u16x8 frag_from<bf16_t>(const float* src, bool valid) {
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(src[e]);
    }
    return v;
}
template <>
__device__ __forceinline__ f32x4 frag_from<float>(const float* src, bool valid) {
    return valid ? f32x4{src[0], src[1], src[2], src[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
}

// row tiles per wave and pass (bf16 softmax heads up to 64 padded columns: registers and the 64 KiB of static LDS allow two)
template <typename WT, int NTL, int KIND>
__host__ __device__ constexpr int head_rb() { return (sizeof(WT) == 2 && KIND == 0 && NTL <= 4) ? 2 : 1; }

template <typename WT, int NTL, int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(head_rb<WT, NTL, KIND>()))) void head_k(const mvae_head_args a) {
    constexpr int KG = op<WT>::KG, FE = op<WT>::FRAG_ELEMS;
    typedef typename op<WT>::frag frag;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, r = l & 15;
    const int R = a.R, H = a.H, N = a.N, NP = NTL * 16;
    const WT* __restrict__ hs = reinterpret_cast<const WT*>(a.hs);
    const WT* __restrict__ wt = reinterpret_cast<const WT*>(a.wt);
    float loss_acc = 0.0f, hit_acc = 0.0f;
    WT* __restrict__ dl = reinterpret_cast<WT*>(a.dlogits);
    // RB row tiles per wave and pass share every weight fragment: the weight-fragment loads (all L1 / L2 hits, but each one
    // an instruction of the CU's address unit) are what this kernel waits for - 64 of the 92 VMEM instructions per 16 rows.
    constexpr int RB = head_rb<WT, NTL, KIND>();
    __shared__ __attribute__((aligned(16))) float stage[4 * RB * 16 * NTL * 16];      // [wave][RB][16 rows][NP] d(logits)
    __shared__ float wg_part[4][2];
    // A bounded grid walks the rows (16*RB per wave and pass): the two loss / accuracy scalars are ONE pair of atomics per
    // workgroup - 8192 waves adding to the same two addresses used to take 200 of this kernel's 250 us.
    for (int row00 = (blockIdx.x * 4 + w) * 16 * RB; row00 < R; row00 += gridDim.x * 64 * RB) {
    f32x4 acc_[RB][NTL];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int n = 0; n < NTL; ++n) acc_[b][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < H / KG; ++s) {
        frag fa[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b)
            fa[b] = *reinterpret_cast<const frag*>(hs + (size_t)min(row00 + b * 16 + r, R - 1) * H + s * KG + q * FE);
#pragma unroll
        for (int n = 0; n < NTL; ++n) {
            const frag fb = *reinterpret_cast<const frag*>(wt + (size_t)(n * 16 + r) * H + s * KG + q * FE);
#pragma unroll
            for (int b = 0; b < RB; ++b) acc_[b][n] = op<WT>::mma(fa[b], fb, acc_[b][n]);     // C[row = q*4+i][col = n*16 + r]
        }
    }
    float bias[NTL];
#pragma unroll
    for (int n = 0; n < NTL; ++n) bias[n] = (n * 16 + r < N) ? a.bias[n * 16 + r] : 0.0f;

#pragma unroll
    for (int b = 0; b < RB; ++b) {
    const int row0 = row00 + b * 16, wb = w * RB + b;
    const f32x4 (&acc)[NTL] = acc_[b];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + q * 4 + i;
        const bool rv = row < R;
        const bool counted = rv && (a.b_stride <= 0 || (row % a.b_stride) < a.b_valid);
        const int rc = rv ? row : R - 1;
        const float rw = a.row_weight ? a.row_weight[rc] : 1.0f;
        if (KIND == 0) {
            float lg[NTL], mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < NTL; ++n) {
                lg[n] = (n * 16 + r < N) ? acc[n][i] + bias[n] : -INFINITY;
                mx = fmaxf(mx, lg[n]);
            }
            mx = group16_max(mx);
            float p[NTL], sum = 0.0f;
#pragma unroll
            for (int n = 0; n < NTL; ++n) {
                p[n] = __builtin_amdgcn_exp2f((lg[n] - mx) * 1.4426950408889634f);     // v_exp_f32 (rel. error ~1e-6)
                sum += p[n];
            }
            sum = group16_sum(sum);
            const float inv = 1.0f / sum;
            const int tg = a.target_idx ? (int)a.target_idx[rc] : 255;
            // second hot column of a TWO-hot target row (attach_instruments: pitch + instrument, reference import_midi.py:288-292);
            // Keras' categorical cross-entropy is then -log p[tg] - log p[tg2], each probability clipped on its own
            const int tg2 = a.target_idx2 ? (int)a.target_idx2[rc] : 255;
            float pt = 0.0f, pm = 0.0f, pt2 = 0.0f;
#pragma unroll
            for (int n = 0; n < NTL; ++n) {
                p[n] *= inv;
                pm = fmaxf(pm, p[n]);
                if (n * 16 + r == tg) pt = p[n];
                if (n * 16 + r == tg2) pt2 = p[n];
            }
            pt = group16_sum(pt);
            if (a.target_idx2) pt2 = group16_sum(pt2);
            pm = group16_max(pm);
            int am = 1 << 30;
#pragma unroll
            for (int n = 0; n < NTL; ++n)
                if (p[n] == pm && n * 16 + r < N) am = min(am, n * 16 + r);
            am = group16_min_i(am);                                    // first maximum (NumPy argmax tie rule)
            const bool has_t = tg < N, has_t2 = tg2 < N;
            const bool inside = has_t && pt >= CE_EPS && pt <= 1.0f - CE_EPS;
            const bool inside2 = has_t2 && pt2 >= CE_EPS && pt2 <= 1.0f - CE_EPS;
            float ce = has_t ? -0.6931471805599453f * __builtin_amdgcn_logf(fminf(fmaxf(pt, CE_EPS), 1.0f - CE_EPS)) : 0.0f;
            if (has_t2) ce += -0.6931471805599453f * __builtin_amdgcn_logf(fminf(fmaxf(pt2, CE_EPS), 1.0f - CE_EPS));
            const int tfirst = has_t ? (has_t2 ? min(tg, tg2) : tg) : (has_t2 ? tg2 : 0);     // NumPy argmax of the target row
            if (rv && r == 0) {
                loss_acc += rw * ce;
                hit_acc += (counted && am == tfirst) ? 1.0f : 0.0f;
                if (a.argmax) a.argmax[row] = (uint8_t)am;
            }
            if (rv) {
#pragma unroll
                for (int n = 0; n < NTL; ++n) {
                    const int col = n * 16 + r;
                    if (a.probs && col < N) a.probs[(size_t)row * N + col] = p[n];
                    if (a.want_grad) {
                        float g = inside ? a.grad_scale * rw * (p[n] - (col == tg ? 1.0f : 0.0f)) : 0.0f;
                        if (inside2) g += a.grad_scale * rw * (p[n] - (col == tg2 ? 1.0f : 0.0f));
                        // staged in LDS: a lane owns single columns here, the row-major rows leave as 16-byte chunks
                        stage[(wb * 16 + q * 4 + i) * NP + col] = col < N ? g : 0.0f;
                    }
                }
            }
        } else {
            // sigmoid + squared error: only column 0 is real
            const float pr = sigmoid_f(acc[0][i] + bias[0]);
            const float y = a.target_val ? a.target_val[rc] : 0.0f;
            if (rv && r == 0) {
                loss_acc += rw * (pr - y) * (pr - y);
                const float rounded = rintf(pr);                          // Keras binary_accuracy: round half to even
                hit_acc += (counted && rounded == y) ? 1.0f : 0.0f;
                if (a.probs) a.probs[row] = pr;
                if (a.argmax) a.argmax[row] = (uint8_t)rounded;
                if (a.want_grad) st<WT>::store(dl + (size_t)row * NP, a.grad_scale * rw * 2.0f * (pr - y) * pr * (1.0f - pr));
            }
            if (a.want_grad && a.dhs)      // d(logits) tile for the fused input gradient: column 0 is real
                stage[(wb * 16 + q * 4 + i) * NP + r] = (rv && r == 0) ? a.grad_scale * rw * 2.0f * (pr - y) * pr * (1.0f - pr) : 0.0f;
        }
    }
    if (KIND == 0 && a.want_grad) {
        // this wave's 16 x NP tile, row-major: 4 values (8 / 16 bytes) per lane and pass
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float* mine = stage + wb * 16 * NP;
#pragma unroll
        for (int e = l * 4; e < 16 * NP; e += 256) {
            const int rr = e / NP, cc = e % NP;
            if (row0 + rr < R) st<WT>::store4(dl + (size_t)(row0 + rr) * NP + cc, *reinterpret_cast<const f32x4*>(mine + e));
        }
    }
    }   // row tiles of this pass
    if (a.want_grad && a.dhs) {
        // Fused gradient w.r.t. the top cell's h sequence: dhs^T tile (H x 16 rows) = Wc (H x NP) * dl^T, straight from the
        // staged d(logits) tile - the separate GEMM launch between this kernel and the decoder BPTT (and its pass over dl)
        // disappears.  Operand roles swapped so that a lane ends up with 4 consecutive h columns of one row: the TILE16 image.
        if (KIND != 0) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const WT* __restrict__ wc = reinterpret_cast<const WT*>(a.wc);
        WT* __restrict__ dhs = reinterpret_cast<WT*>(a.dhs);
        f32x4 dacc[RB][16];
        const int HT = H >> 4;                   // <= 16 (checked by the launcher)
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) dacc[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < (NP + KG - 1) / KG; ++s2) {
            const int k0 = s2 * KG + q * FE;
            const frag fz = frag_from<WT>(stage, false);
            frag fd[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b)
                fd[b] = frag_from<WT>(stage + ((w * RB + b) * 16 + r) * NP + (k0 < NP ? k0 : 0), k0 < NP);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < HT) {
                    const frag fw = k0 < NP ? *reinterpret_cast<const frag*>(wc + (size_t)(j * 16 + r) * NP + k0) : fz;
#pragma unroll
                    for (int b = 0; b < RB; ++b)
                        dacc[b][j] = op<WT>::mma(fw, fd[b], dacc[b][j]);   // C[row = h column q*4+i of tile j][col = batch-time row r]
                }
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row0 = row00 + b * 16;
            if (row0 < R) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < HT)
                        st<WT>::store4(dhs + ((((size_t)(row0 >> 4) * HT + j) * 64) + (size_t)(q * 16 + r)) * 4, dacc[b][j]);
            }
        }
    }
        __builtin_amdgcn_wave_barrier();        // the stage tile is reused by the next pass
    }   // row loop
    if (a.scalars) {
        loss_acc = wave_sum(loss_acc);
        hit_acc = wave_sum(hit_acc);
        if (l == 0) {
            wg_part[w][0] = loss_acc;
            wg_part[w][1] = hit_acc;
        }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(a.scalars, wg_part[0][0] + wg_part[1][0] + wg_part[2][0] + wg_part[3][0]);
            atomicAdd(a.scalars + 1, wg_part[0][1] + wg_part[1][1] + wg_part[2][1] + wg_part[3][1]);
        }
    }
}

template <typename WT, int KIND>
int launch(const mvae_head_args& a, hipStream_t s) {
    const int ntl = (a.N + 15) / 16;
    const dim3 block(256);
    auto grid = [&](int rb) {
        const int need = (a.R + 64 * rb - 1) / (64 * rb);
        const int cap = rb == 2 ? 512 : 1024;       // rb == 2: two workgroups per CU are resident - one round, half the atomics
        return dim3(need < cap ? need : cap);
    };
    switch (ntl) {
        case 1: hipLaunchKernelGGL((head_k<WT, 1, KIND>), grid(head_rb<WT, 1, KIND>()), block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((head_k<WT, 2, KIND>), grid(head_rb<WT, 2, KIND>()), block, 0, s, a); break;
        case 3:
        case 4: hipLaunchKernelGGL((head_k<WT, 4, KIND>), grid(head_rb<WT, 4, KIND>()), block, 0, s, a); break;
        case 5: case 6: case 7:
        case 8: hipLaunchKernelGGL((head_k<WT, 8, KIND>), grid(head_rb<WT, 8, KIND>()), block, 0, s, a); break;
        default: return MVAE_E_UNSUPPORTED;
    }
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

}  // namespace

// NB: dlogits has NP = 16*ceil(N/16) columns for N<=16 / N in (32,64] / (64,128]; N in (16,32] uses 32; the
// kernel instantiation for 3 tiles is the 4-tile one, so callers must size wt / dlogits with mvae_head_np().
extern "C" int mvae_head_np(int32_t N) {
    const int ntl = (N + 15) / 16;
    if (ntl <= 1) return 16;
    if (ntl == 2) return 32;
    if (ntl <= 4) return 64;
    if (ntl <= 8) return 128;
    return -1;
}

extern "C" int mvae_head(const mvae_head_args* a, void* stream) {
    if (!a || !a->hs || !a->wt || !a->bias || a->R <= 0 || a->N <= 0 || a->H <= 0) return MVAE_E_ARG;
    if (a->want_grad && !a->dlogits) return MVAE_E_ARG;
    if (a->dhs && (!a->wc || !a->want_grad || (a->R % 16) || (a->H % 16))) return MVAE_E_ARG;
    if (a->dhs && a->H > 256) return MVAE_E_UNSUPPORTED;
    if (a->kind == 1 && a->N != 1) return MVAE_E_ARG;
    if (a->kind == 0 && a->N > 128) return MVAE_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->dtype == MVAE_F32) {
        if (a->H % 16) return MVAE_E_UNSUPPORTED;
        return a->kind == 0 ? launch<float, 0>(*a, s) : launch<float, 1>(*a, s);
    }
    if (a->dtype == MVAE_BF16) {
        if (a->H % 32) return MVAE_E_UNSUPPORTED;
        return a->kind == 0 ? launch<bf16_t, 0>(*a, s) : launch<bf16_t, 1>(*a, s);
    }
    return MVAE_E_ARG;
}
