// Small kernels of the MIDI-VAE step: latent block (KL + reparameterisation + style classifier), reductions,
// conversions / weight preparation, Keras optimizers.  All HBM-bound, one pass over their operands.
#include <string.h>

#include "common.h"

namespace {

constexpr float CE_EPS = 1e-7f;

// ---- latent block (reference vae_definition.py:29-37 KL, :498-502 sampling, :730-734 style softmax) -----------
// one wave per batch row
__global__ __launch_bounds__(256) void latent_fwd_k(const mvae_latent_fwd_args a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + w;
    if (b >= a.B) return;
    const int Z = a.Z;
    const float plv = 2.0f * logf(a.prior_std), pvar = a.prior_std * a.prior_std;
    float kl = 0.0f;
    for (int j = l; j < Z; j += 64) {
        const float mu = a.mu[(size_t)b * Z + j], lv = a.logvar[(size_t)b * Z + j];
        const float d = mu - a.prior_mean;
        kl += 1.0f + lv - plv - (d * d + expf(lv)) / pvar;
        a.z[(size_t)b * (a.ldz ? a.ldz : Z) + j] = mu + expf(0.5f * lv) * a.eps[(size_t)b * Z + j];
    }
    kl = wave_sum(kl);
    if (l == 0) {
        atomicAdd(a.scalars, a.inv_batch * a.beta * (-0.5f) * kl);
        if (a.style_target && a.C > 0) {
            const int C = a.C;
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) {
                const float mu = a.mu[(size_t)b * Z + c], lv = a.logvar[(size_t)b * Z + c];
                mx = fmaxf(mx, mu + expf(0.5f * lv) * a.eps[(size_t)b * Z + c]);
            }
            float sum = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float mu = a.mu[(size_t)b * Z + c], lv = a.logvar[(size_t)b * Z + c];
                sum += expf(mu + expf(0.5f * lv) * a.eps[(size_t)b * Z + c] - mx);
            }
            const int tg = a.style_target[b];
            float pt = 0.0f, pm = -1.0f;
            int am = 0;
            for (int c = 0; c < C; ++c) {
                const float mu = a.mu[(size_t)b * Z + c], lv = a.logvar[(size_t)b * Z + c];
                const float p = expf(mu + expf(0.5f * lv) * a.eps[(size_t)b * Z + c] - mx) / sum;
                if (a.style_probs) a.style_probs[(size_t)b * C + c] = p;
                if (c == tg) pt = p;
                if (p > pm) { pm = p; am = c; }
            }
            const float rw = a.style_row_weight ? a.style_row_weight[b] : a.inv_batch;
            const float ce = tg < C ? -logf(fminf(fmaxf(pt, CE_EPS), 1.0f - CE_EPS)) : 0.0f;
            atomicAdd(a.scalars + 1, rw * ce);
            atomicAdd(a.scalars + 2, am == (tg < C ? tg : 0) ? 1.0f : 0.0f);
        }
    }
}

__global__ void latent_bwd_k(const mvae_latent_bwd_args a) {
    const size_t n = (size_t)a.B * a.Z;
    const float pvar = a.prior_std * a.prior_std;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / a.Z), j = (int)(e % a.Z);
        float dz = a.dz[(size_t)b * (a.lddz ? a.lddz : a.Z) + j];
        if (a.style_probs && a.style_target && j < a.C) {
            const int tg = a.style_target[b];
            if (tg < a.C) {
                const float pt = a.style_probs[(size_t)b * a.C + tg];
                if (pt >= CE_EPS && pt <= 1.0f - CE_EPS) {
                    const float rw = a.style_row_weight ? a.style_row_weight[b] : a.inv_batch;
                    dz += a.style_weight * rw * (a.style_probs[(size_t)b * a.C + j] - (j == tg ? 1.0f : 0.0f));
                }
            }
        }
        const float mu = a.mu[e], lv = a.logvar[e];
        a.dmu[e] = dz + a.beta * (mu - a.prior_mean) / pvar * a.inv_batch;
        a.dlogvar[e] = dz * a.eps[e] * 0.5f * expf(0.5f * lv) + a.beta * (-0.5f) * (1.0f - expf(lv) / pvar) * a.inv_batch;
    }
}

// ---- reductions -----------------------------------------------------------------------------------------------
template <typename WT>
__global__ void colsum_k(const WT* __restrict__ X, const float* __restrict__ wgt, int R, int N, int ldx, int rows_per_block,
                         float* __restrict__ out) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.0f;
        for (int rr = r0; rr < r1; ++rr) s += st<WT>::load(X + (size_t)rr * ldx + n) * (wgt ? wgt[rr] : 1.0f);
        atomicAdd(out + n, s);
    }
}
// bf16, N % 8 == 0, 16-byte aligned rows: each thread owns 8 columns (one 16-byte load per row), the block's 256
// threads cover 256 / (N/8) rows at a time; partial sums meet in LDS, one atomic per column per block.
__global__ __launch_bounds__(256) void colsum_bf16x8_k(const bf16_t* __restrict__ X, const float* __restrict__ wgt, int R, int N,
                                                       int Nv /* N rounded up to 8: columns read */, int ldx, int rows_per_block,
                                                       float* __restrict__ out) {
    __shared__ float red[256 * 8];
    const int c8 = Nv / 8, lanes_r = 256 / c8;          // c8 <= 256; threads beyond lanes_r * c8 (c8 = 96: a GRU's 768 columns) idle
    const int col = (threadIdx.x % c8) * 8, rsub = threadIdx.x / c8;
    const int r0 = blockIdx.x * rows_per_block, r1 = rsub < lanes_r ? min(R, r0 + rows_per_block) : 0;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rr = r0 + rsub;
    for (; rr + 3 * lanes_r < r1; rr += 4 * lanes_r) {        // 4 independent 16-byte loads in flight
        u16x8 v[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = *reinterpret_cast<const u16x8*>(X + (size_t)(rr + u * lanes_r) * ldx + col);
            w[u] = wgt ? wgt[rr + u * lanes_r] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf2f(v[u][e]) * w[u];
    }
    for (; rr < r1; rr += lanes_r) {
        const u16x8 v = *reinterpret_cast<const u16x8*>(X + (size_t)rr * ldx + col);
        const float w = wgt ? wgt[rr] : 1.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += bf2f(v[e]) * w;
    }
    if (rsub < lanes_r) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(rsub * c8 + threadIdx.x % c8) * 8 + e] = s[e];
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        float t = 0.0f;
        for (int k = 0; k < lanes_r; ++k) t += red[(k * c8 + n / 8) * 8 + n % 8];
        atomicAdd(out + n, t);
    }
}

template <typename WT>
__global__ void sum_time_k(const WT* __restrict__ X, int T, int BN, int t_per_block, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= BN) return;
    const int t0 = blockIdx.y * t_per_block, t1 = min(T, t0 + t_per_block);
    float s = 0.0f;
    for (int t = t0; t < t1; ++t) s += st<WT>::load(X + (size_t)t * BN + e);
    atomicAdd(out + e, s);
}
// bf16, BN % 8 == 0: 8 elements (16 bytes) per thread, the time range split over gridDim.y
__global__ __launch_bounds__(256) void sum_time_bf16x8_k(const bf16_t* __restrict__ X, int T, int BN, int t_per_block,
                                                         float* __restrict__ out) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (e >= BN) return;
    const int t0 = blockIdx.y * t_per_block, t1 = min(T, t0 + t_per_block);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int t = t0;
    for (; t + 3 < t1; t += 4) {
        u16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u16x8*>(X + (size_t)(t + u) * BN + e);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += bf2f(v[u][k]);
    }
    for (; t < t1; ++t) {
        const u16x8 v = *reinterpret_cast<const u16x8*>(X + (size_t)t * BN + e);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += bf2f(v[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(out + e + k, s[k]);
}

// ---- elementwise ----------------------------------------------------------------------------------------------
__global__ void tanh_bwd_k(const float* y, const float* dy, float* dx, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        dx[e] = dy[e] * (1.0f - y[e] * y[e]);
}

template <typename S, typename D>
__global__ void convert_k(const S* src, D* dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        st<D>::store(dst + e, st<S>::load(src + e));
}

template <typename D>
__global__ void make_table_k(const float* W, const float* bias, D* table, int K, int N) {
    make_table_body<D>(W, bias, table, K, N, blockIdx.x, gridDim.x);
}

template <typename D>
__global__ void transpose_convert_k(const float* W, D* out, int K, int N, int NPAD) {
    transpose_convert_body<D>(W, out, K, N, NPAD, blockIdx.x, gridDim.x);
}

template <typename D>
__global__ void relayout_k(const D* src, D* dst, int rows, int cols, int to_tile) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / cols), c = (int)(e % cols);
        const size_t lane = (size_t)(((c & 15) >> 2) * 16 + (m & 15));
        const size_t t = (to_tile & 4)
            ? ((((size_t)(m >> 4) * (cols >> 5) + (c >> 8) * 8 + ((c >> 4) & 7)) * 64) + lane) * 8 + ((c >> 7) & 1) * 4 + (c & 3)
            : (to_tile & 2)
            ? ((((size_t)(m >> 4) * (cols >> 5) + (c >> 5)) * 64) + lane) * 8 + ((c & 31) >> 4) * 4 + (c & 3)
            : ((((size_t)(m >> 4) * (cols >> 4) + (c >> 4)) * 64) + lane) * 4 + (c & 3);
        if (to_tile & 1) dst[t] = src[e]; else dst[e] = src[t];
    }
}

// ---- optimizers (Keras 2.0.8 formulas) ------------------------------------------------------
__global__ void adam_k(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
                       float gs) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float ge = g[e] * gs;
        const float me = b1 * m[e] + (1.0f - b1) * ge;
        const float ve = b2 * v[e] + (1.0f - b2) * ge * ge;
        m[e] = me;
        v[e] = ve;
        p[e] -= lr_t * me / (sqrtf(ve) + eps);
    }
}
// graph-replayable form: the step count lives in device memory (t_done = completed steps)
// ``guard``: status word of the time-pipelined stacks - non-zero means a kernel of THIS step gave up waiting for its producer and
// the gradients are garbage: the update is skipped (parameters and moments stay valid), the gradients are still zeroed.
__global__ void adam_dev_k(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                           float eps, float gs, const int* t_done, int zero_g, int vec, const uint32_t* guard) {
    const bool skip = guard && *guard != 0u;
    const float t = (float)(*t_done + 1);
    const float lr_t = lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
    auto one = [&](float ge, float& me, float& ve, float& pe) {
        if (skip) return;
        ge *= gs;
        me = b1 * me + (1.0f - b1) * ge;
        ve = b2 * ve + (1.0f - b2) * ge * ge;
        pe -= lr_t * me / (sqrtf(ve) + eps);
    };
    // 16 bytes per lane and access: four streams in, four out - the kernel is the HBM pass over 32 bytes per parameter
    const size_t n4 = vec ? n >> 2 : 0;          // vec: all four buffers 16-byte aligned
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const f32x4 ge = g4[e];
        f32x4 me = m4[e], ve = v4[e], pe = p4[e];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float mi = me[i], vi = ve[i], pi = pe[i];
            one(ge[i], mi, vi, pi);
            me[i] = mi;
            ve[i] = vi;
            pe[i] = pi;
        }
        m4[e] = me;
        v4[e] = ve;
        p4[e] = pe;
        if (zero_g) g4[e] = f32x4{0.f, 0.f, 0.f, 0.f};       // the next step accumulates into a clean buffer: no fill launch
    }
    for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float me = m[e], ve = v[e], pe = p[e];
        one(g[e], me, ve, pe);
        m[e] = me;
        v[e] = ve;
        p[e] = pe;
        if (zero_g) g[e] = 0.0f;
    }
}
__global__ void bump_k(int* t, const uint32_t* guard) { if (!(guard && *guard != 0u)) *t += 1; }
__global__ void rmsprop_k(float* p, float* g, float* v, size_t n, float lr, float rho, float eps, float gs, int zero_g, int vec,
                          const uint32_t* guard) {
    const bool skip = guard && *guard != 0u;
    auto one = [&](float ge, float& ve, float& pe) {
        if (skip) return;
        ge *= gs;
        ve = rho * ve + (1.0f - rho) * ge * ge;
        pe -= lr * ge / (sqrtf(ve) + eps);
    };
    const size_t n4 = vec ? n >> 2 : 0;          // vec: all three buffers 16-byte aligned
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const f32x4 ge = g4[e];
        f32x4 ve = v4[e], pe = p4[e];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float vi = ve[i], pi = pe[i];
            one(ge[i], vi, pi);
            ve[i] = vi;
            pe[i] = pi;
        }
        v4[e] = ve;
        p4[e] = pe;
        if (zero_g) g4[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float ve = v[e], pe = p[e];
        one(g[e], ve, pe);
        v[e] = ve;
        p[e] = pe;
        if (zero_g) g[e] = 0.0f;
    }
}

// epoch accumulators of the per-step loss / metric scalars (Keras BaseLogger: batch-size weighted means): one tiny launch
// after the step instead of a device->host read (= a full synchronisation) per minibatch
__global__ void scalars_accumulate_k(float* acc, const float* x, int n, float alpha, uint32_t plain_mask) {
    const int i = threadIdx.x;
    if (i < n) acc[i] += ((plain_mask >> i) & 1u) ? x[i] : alpha * x[i];
}
__global__ void copy2d_f32_k(float* dst, int ldd, const float* src, int lds, int rows, int cols, int src_row0, int zero_rows) {
    // dst[r, c] = src[src_row0 + r, c]; rows whose source index is negative (r < zero_rows) become zero
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)rows * cols; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        dst[(size_t)r * ldd + c] = r < zero_rows ? 0.0f : src[(size_t)(src_row0 + r) * lds + c];
    }
}

// ---- signature head: activation(z[:, off:off+SD]) against the song's signature vector, mean squared error ------------------
// one thread per window (B is a minibatch: a few hundred rows); loss / hits added to scalars[0..1]
__global__ void sig_head_fwd_k(const float* zh, int ldz, int off, int SD, int B, const float* target, const float* row_weight,
                               float* out, float* scalars) {
    float loss = 0.0f, hits = 0.0f;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        float se = 0.0f, best_p = -INFINITY, best_t = -INFINITY;
        int arg_p = 0, arg_t = 0;
        for (int j = 0; j < SD; ++j) {
            const float p = tanhf(zh[(size_t)b * ldz + off + j]);
            const float t = target ? target[(size_t)b * SD + j] : 0.0f;
            out[(size_t)b * SD + j] = p;
            se += (p - t) * (p - t);
            if (p > best_p) { best_p = p; arg_p = j; }
            if (t > best_t) { best_t = t; arg_t = j; }
        }
        const float rw = row_weight ? row_weight[b] : 0.0f;
        loss += rw * se / (float)SD;                        // Keras mse: mean over the last axis, then the weighted batch mean
        hits += (rw != 0.0f && arg_p == arg_t) ? 1.0f : 0.0f;     // 'accuracy' on a non-categorical output = categorical_accuracy
    }
    if (target && scalars) {
        if (loss != 0.0f) atomicAdd(scalars, loss);
        if (hits != 0.0f) atomicAdd(scalars + 1, hits);
    }
}
// dz[b, off + j] += weight * rw[b] * 2 (p - t) / SD * (1 - p^2)
__global__ void sig_head_bwd_k(float* dz, int lddz, int off, int SD, int B, const float* out, const float* target,
                               const float* row_weight, float weight) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * SD) return;
    const int b = e / SD, j = e % SD;
    const float p = out[e];
    dz[(size_t)b * lddz + off + j] += weight * row_weight[b] * 2.0f * (p - target[e]) / (float)SD * (1.0f - p * p);
}
// dl[r, n] += p[r, n] * (dp[r, n] - sum_j p[r, j] dp[r, j]): a gradient arriving at softmax PROBABILITIES folded into d(logits)
template <typename WT>
__global__ void softmax_bwd_add_k(const float* __restrict__ p, const float* __restrict__ dp, WT* __restrict__ dl, int R, int N, int NP) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        const float* pr = p + (size_t)r * N;
        const float* dr = dp + (size_t)r * N;
        float dot = 0.0f;
        for (int j = 0; j < N; ++j) dot += pr[j] * dr[j];
        WT* out = dl + (size_t)r * NP;
        for (int j = 0; j < N; ++j) st<WT>::store(out + j, st<WT>::load(out + j) + pr[j] * (dr[j] - dot));
    }
}

// ---- bidirectional encoder layers (Keras Bidirectional(..., merge_mode='concat')) -----------------------------------------
// cat[t, b, :H] = f[t, b, :], cat[t, b, H:] = r[T-1-t, b, :]  and  cat_rev[k] = cat[T-1-k]  (16-byte chunks; rows of H elements)
__global__ void bi_concat_k(const uint4* __restrict__ f, const uint4* __restrict__ r, uint4* __restrict__ cat, uint4* __restrict__ cat_rev,
                            int T, int B, int hc /* 16-byte chunks per H row */) {
    const size_t n = (size_t)T * B * hc;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % hc);
        const size_t row = e / hc;
        const int b = (int)(row % B), t = (int)(row / B);
        const uint4 vf = f[e];
        const uint4 vr = r[((size_t)(T - 1 - t) * B + b) * hc + c];
        const size_t o = ((size_t)t * B + b) * 2 * hc, orv = ((size_t)(T - 1 - t) * B + b) * 2 * hc;
        cat[o + c] = vf;
        cat[o + hc + c] = vr;
        if (cat_rev) { cat_rev[orv + c] = vf; cat_rev[orv + hc + c] = vr; }
    }
}
// dst[t] = (a ? a[t] : 0) + b[T-1-t] over T contiguous slabs of `slab` elements (row-major and TILE16 alike: a time step of a
// (T*B, H) sequence is one contiguous slab in both layouts when B is a multiple of 16)
template <typename WT>
__global__ void add_time_reversed_k(WT* __restrict__ dst, const WT* __restrict__ a, const WT* __restrict__ b, int T, size_t slab) {
    const size_t n = (size_t)T * slab;
    for (size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < n; e += (size_t)gridDim.x * blockDim.x * 4) {
        const size_t t = e / slab, w = e % slab;
        f32x4 v = st<WT>::load4(b + (size_t)(T - 1 - t) * slab + w);
        if (a) { const f32x4 u = st<WT>::load4(a + e); v = v + u; }
        st<WT>::store4(dst + e, v);
    }
}

inline int nblocks(size_t n, int per = 256, int cap = 2048) {
    size_t b = (n + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

extern "C" int mvae_abi_version(void) { return MVAE_ABI_VERSION; }
extern "C" int mvae_rnn_producer_waves(int32_t seq_layout) { return seq_layout == MVAE_TILE16Q ? 8 : 4; }
extern "C" const char* mvae_build_info(void) { return "libmidivae_hip gfx950 (CDNA4) built " __DATE__ " " __TIME__; }

extern "C" int mvae_latent_fwd(const mvae_latent_fwd_args* a, void* stream) {
    if (!a || !a->mu || !a->logvar || !a->eps || !a->z || !a->scalars || a->B <= 0 || a->Z <= 0) return MVAE_E_ARG;
    if (a->style_target && (a->C <= 0 || a->C > 64 || a->C > a->Z)) return MVAE_E_ARG;
    hipLaunchKernelGGL(latent_fwd_k, dim3((a->B + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_latent_bwd(const mvae_latent_bwd_args* a, void* stream) {
    if (!a || !a->mu || !a->logvar || !a->eps || !a->dz || !a->dmu || !a->dlogvar || a->B <= 0 || a->Z <= 0)
        return MVAE_E_ARG;
    hipLaunchKernelGGL(latent_bwd_k, dim3(nblocks((size_t)a->B * a->Z)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), *a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

static int colsum_impl(const void* X, int32_t kind, const float* wgt, int32_t R, int32_t N, int32_t ldx, float* out, hipStream_t s) {
    if (!X || !out || R <= 0 || N <= 0 || ldx < N) return MVAE_E_ARG;
    // (N not a multiple of 8 - 61 note classes in rows of 64: the pad columns are read and not written)
    const int Nv = (N + 7) / 8 * 8, c8 = Nv / 8;
    const bool vec = kind == MVAE_BF16 && Nv <= ldx && c8 <= 256 && (ldx % 8) == 0 &&
                     (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    if (vec) {
        // enough blocks to fill the chip, enough rows per block to amortise the LDS reduction and the atomics
        int rpb = (R + 2047) / 2048;
        const int lanes_r = 256 / c8;
        if (rpb < 16 * lanes_r) rpb = 16 * lanes_r;
        const int blocks = (R + rpb - 1) / rpb;
        hipLaunchKernelGGL(colsum_bf16x8_k, dim3(blocks), dim3(256), 0, s, (const bf16_t*)X, wgt, R, N, Nv, ldx, rpb, out);
        MVAE_CHECK_LAUNCH();
        return MVAE_OK;
    }
    int rpb = (R + 1023) / 1024;
    if (rpb < 8) rpb = 8;
    const int blocks = (R + rpb - 1) / rpb;
    if (kind == MVAE_F32)
        hipLaunchKernelGGL(colsum_k<float>, dim3(blocks), dim3(256), 0, s, (const float*)X, wgt, R, N, ldx, rpb, out);
    else if (kind == MVAE_BF16)
        hipLaunchKernelGGL(colsum_k<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)X, wgt, R, N, ldx, rpb, out);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_colsum(const void* X, int32_t kind, int32_t R, int32_t N, int32_t ldx, float* out, void* stream) {
    return colsum_impl(X, kind, nullptr, R, N, ldx, out, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int mvae_colsum_weighted(const void* X, int32_t kind, const float* wgt, int32_t R, int32_t N, int32_t ldx, float* out,
                                    void* stream) {
    if (!wgt) return MVAE_E_ARG;
    return colsum_impl(X, kind, wgt, R, N, ldx, out, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int mvae_sum_over_time(const void* X, int32_t kind, int32_t T, int32_t BN, float* out, int32_t accumulate, void* stream) {
    if (!X || !out || T <= 0 || BN <= 0) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!accumulate && hipMemsetAsync(out, 0, (size_t)BN * sizeof(float), s) != hipSuccess) return MVAE_E_LAUNCH;
    // split the time range so that >= ~1024 blocks exist; every block adds its partial sums atomically
    const bool vec = kind == MVAE_BF16 && (BN % 8) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    const int bx = vec ? (BN / 8 + 255) / 256 : (BN + 255) / 256;
    int by = (1024 + bx - 1) / bx;
    if (by > T) by = T;
    if (by < 1) by = 1;
    const int tpb = (T + by - 1) / by;
    by = (T + tpb - 1) / tpb;
    if (vec)
        hipLaunchKernelGGL(sum_time_bf16x8_k, dim3(bx, by), dim3(256), 0, s, (const bf16_t*)X, T, BN, tpb, out);
    else if (kind == MVAE_F32)
        hipLaunchKernelGGL(sum_time_k<float>, dim3(bx, by), dim3(256), 0, s, (const float*)X, T, BN, tpb, out);
    else if (kind == MVAE_BF16)
        hipLaunchKernelGGL(sum_time_k<bf16_t>, dim3(bx, by), dim3(256), 0, s, (const bf16_t*)X, T, BN, tpb, out);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

extern "C" int mvae_tanh_bwd(const float* y, const float* dy, float* dx, size_t n, void* stream) {
    if (!y || !dy || !dx) return MVAE_E_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(tanh_bwd_k, dim3(nblocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, dy, dx, n);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_convert(const void* src, int32_t sk, void* dst, int32_t dk, size_t n, void* stream) {
    if (!src || !dst) return MVAE_E_ARG;
    if (n == 0) return MVAE_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g(nblocks(n)), b(256);
    if (sk == MVAE_F32 && dk == MVAE_BF16)
        hipLaunchKernelGGL((convert_k<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (sk == MVAE_BF16 && dk == MVAE_F32)
        hipLaunchKernelGGL((convert_k<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, n);
    else if (sk == MVAE_F32 && dk == MVAE_F32)
        hipLaunchKernelGGL((convert_k<float, float>), g, b, 0, s, (const float*)src, (float*)dst, n);
    else if (sk == MVAE_BF16 && dk == MVAE_BF16)
        hipLaunchKernelGGL((convert_k<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_relayout(const void* src, void* dst, int32_t kind, int32_t rows, int32_t cols, int32_t to_tile16,
                             void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || (rows % 16) || (cols % ((to_tile16 & 4) ? 256 : (to_tile16 & 2) ? 32 : 16)) || to_tile16 < 0 || to_tile16 > 5)
        return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g(nblocks((size_t)rows * cols)), b(256);
    if (kind == MVAE_F32)
        hipLaunchKernelGGL(relayout_k<float>, g, b, 0, s, (const float*)src, (float*)dst, rows, cols, to_tile16);
    else if (kind == MVAE_BF16)
        hipLaunchKernelGGL(relayout_k<bf16_t>, g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, rows, cols, to_tile16);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
// ---- stream-ordered synchronisation with running kernels ------------------------------------------------------------
extern "C" int mvae_stream_wait_value32(void* stream, const uint32_t* addr, uint32_t value) {
    if (!addr) return MVAE_E_ARG;
    return hipStreamWaitValue32(reinterpret_cast<hipStream_t>(stream), const_cast<uint32_t*>(addr), value, hipStreamWaitValueGte,
                                0xFFFFFFFFu) == hipSuccess ? MVAE_OK : MVAE_E_LAUNCH;
}
extern "C" int mvae_stream_write_value32(void* stream, uint32_t* addr, uint32_t value) {
    if (!addr) return MVAE_E_ARG;
    return hipStreamWriteValue32(reinterpret_cast<hipStream_t>(stream), addr, value, 0) == hipSuccess ? MVAE_OK : MVAE_E_LAUNCH;
}

// ---- do two streams share a hardware queue? ---------------------------------------------------------------------------
// The HIP runtime maps streams onto a fixed number of hardware queues (GPU_MAX_HW_QUEUES); two streams on ONE queue run their
// kernels one after the other.  A phase launch on the critical stream contains BOTH the producer and the consumer of the
// persistent GEMM that runs on another stream beside it: on a shared queue the GEMM could only start when the launch has ended,
// and the launch would wait for it (bounded - a time-out and the per-launch schedule - but 2-4 s late).  So the engine asks
// once, when it creates its streams: a waiter on `a` (bounded: ~2 ms) and, enqueued AFTER it, a setter on `b`.
__global__ void alias_wait_k(const uint32_t* flag, uint32_t value, uint32_t max_spins, uint32_t* seen) {
    uint32_t spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != value && spins < max_spins) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
    }
    *seen = spins < max_spins ? 1u : 0u;
}
__global__ void alias_set_k(uint32_t* flag, uint32_t value) { __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
extern "C" int mvae_streams_alias(void* stream_a, void* stream_b, uint32_t* scratch /* 2 device words */, uint32_t tag) {
    if (!scratch || tag == 0u) return MVAE_E_ARG;
    hipStream_t a = reinterpret_cast<hipStream_t>(stream_a), b = reinterpret_cast<hipStream_t>(stream_b);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return MVAE_E_LAUNCH;
    hipLaunchKernelGGL(alias_wait_k, dim3(1), dim3(1), 0, a, scratch, tag, 20000u, scratch + 1);
    hipLaunchKernelGGL(alias_set_k, dim3(1), dim3(1), 0, b, scratch, tag);
    MVAE_CHECK_LAUNCH();
    uint32_t seen = 0;
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
        hipMemcpy(&seen, scratch + 1, sizeof(seen), hipMemcpyDeviceToHost) != hipSuccess)
        return MVAE_E_LAUNCH;
    return seen ? 0 : 1;
}

// ---- batched weight preparation: every derived copy of the parameters in ONE launch ------------------------------
// Workgroups [base[j], base[j+1]) run job j: a job gets workgroups in proportion to its output (a 256 x 1024 fragment pack 64,
// a 32-word fill one).
constexpr int PREP_MAX_JOBS = 64;      // (64 x 48-byte jobs + 65 offsets = 3.3 KiB of kernel arguments; the limit is 4 KiB)
struct prep_batch {
    int32_t n;
    int32_t base[PREP_MAX_JOBS + 1];
    mvae_prep_job jobs[PREP_MAX_JOBS];
};
template <typename D>
__device__ __forceinline__ void outer_bias_body(const float* __restrict__ xs, const float* __restrict__ w, const float* __restrict__ bias,
                                                D* __restrict__ out, int R, int N, int bid, int nb) {
    const size_t total = (size_t)R * N / 4;
    for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < total; e += (size_t)nb * blockDim.x) {
        const size_t tile = e >> 6;
        const int lane = (int)(e & 63), m = (int)(tile / (N >> 4)) * 16 + (lane & 15), n = (int)(tile % (N >> 4)) * 16 + (lane >> 4) * 4;
        const float x = xs[m];
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + n), bv = *reinterpret_cast<const f32x4*>(bias + n);
        st<D>::store4(out + e * 4, x * wv + bv);
    }
}
__global__ __launch_bounds__(256) void prepare_batch_k(const prep_batch pb) {
    int j = 0;
    while (j + 1 < pb.n && (int)blockIdx.x >= pb.base[j + 1]) ++j;
    const int bid = (int)blockIdx.x - pb.base[j], nb = pb.base[j + 1] - pb.base[j];
    const mvae_prep_job& job = pb.jobs[j];
    const float* src = reinterpret_cast<const float*>(job.src);
    const bool bf = job.kind == MVAE_BF16;
    switch (job.op) {
        case MVAE_PREP_PACK_RECURRENT:
            if (bf) pack_recurrent_body<bf16_t>(src, reinterpret_cast<bf16_t*>(job.dst), job.a, job.b, job.c, bid, nb);
            else pack_recurrent_body<float>(src, reinterpret_cast<float*>(job.dst), job.a, job.b, job.c, bid, nb);
            break;
        case MVAE_PREP_MAKE_TABLE:
            if (bf) make_table_body<bf16_t>(src, reinterpret_cast<const float*>(job.src2), reinterpret_cast<bf16_t*>(job.dst), job.a, job.b, bid, nb, job.c);
            else make_table_body<float>(src, reinterpret_cast<const float*>(job.src2), reinterpret_cast<float*>(job.dst), job.a, job.b, bid, nb, job.c);
            break;
        case MVAE_PREP_TRANSPOSE_CONVERT:
            if (bf) transpose_convert_body<bf16_t>(src, reinterpret_cast<bf16_t*>(job.dst), job.a, job.b, job.c, bid, nb);
            else transpose_convert_body<float>(src, reinterpret_cast<float*>(job.dst), job.a, job.b, job.c, bid, nb);
            break;
        case MVAE_PREP_CONVERT:
            if (bf) convert_f32_body<bf16_t>(src, reinterpret_cast<bf16_t*>(job.dst), (size_t)job.a * job.b, bid, nb);
            else convert_f32_body<float>(src, reinterpret_cast<float*>(job.dst), (size_t)job.a * job.b, bid, nb);
            break;
        case MVAE_PREP_CONVERT_PAD: {
            const size_t n = (size_t)job.a * job.c;
            for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < n; e += (size_t)nb * blockDim.x) {
                const int rr = (int)(e / job.c), cc = (int)(e % job.c);
                const float v = cc < job.b ? src[(size_t)rr * job.b + cc] : 0.0f;
                if (bf) st<bf16_t>::store(reinterpret_cast<bf16_t*>(job.dst) + e, v);
                else reinterpret_cast<float*>(job.dst)[e] = v;
            }
            break;
        }
        case MVAE_PREP_ADD_I32:        // (src = optional guard word: no increment while it is non-zero - the update was skipped too;
            if (bid == 0 && threadIdx.x == 0) {         //  src2 = optional latch word: the guard word is moved there and cleared)
                uint32_t* g = const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(job.src));
                const uint32_t st = g ? *g : 0u;
                if (st == 0u) *reinterpret_cast<int32_t*>(job.dst) += job.a;
                else if (job.src2) {
                    uint32_t* l = const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(job.src2));
                    if (st > *l) *l = st;
                    *g = 0u;
                }
            }
            break;
        case MVAE_PREP_BROADCAST_ROWS: {    // dst (a, b) = the row src (b) repeated: start*W + b of a cell stepped on an all-zero input
            const size_t n = (size_t)job.a * job.b;
            for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < n; e += (size_t)nb * blockDim.x) {
                const float v = src[e % job.b];
                if (bf) st<bf16_t>::store(reinterpret_cast<bf16_t*>(job.dst) + e, v);
                else reinterpret_cast<float*>(job.dst)[e] = v;
            }
            break;
        }
        case MVAE_PREP_ZERO: {
            const size_t n = (size_t)job.a * job.b * (bf ? 2 : 4) / 4;       // 32-bit words
            uint32_t* d = reinterpret_cast<uint32_t*>(job.dst);
            for (size_t e = (size_t)bid * blockDim.x + threadIdx.x; e < n; e += (size_t)nb * blockDim.x) d[e] = 0u;
            break;
        }
    }
}
extern "C" int mvae_prepare_batch(const mvae_prep_job* jobs, int32_t n_jobs, void* stream) {
    if (!jobs || n_jobs < 0) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int j0 = 0; j0 < n_jobs; j0 += PREP_MAX_JOBS) {
        prep_batch pb;
        pb.n = n_jobs - j0 < PREP_MAX_JOBS ? n_jobs - j0 : PREP_MAX_JOBS;
        int total = 0;
        for (int j = 0; j < pb.n; ++j) {
            const mvae_prep_job& job = jobs[j0 + j];
            if ((!job.src && job.op != MVAE_PREP_ZERO && job.op != MVAE_PREP_ADD_I32) || !job.dst || job.op < 0 ||
                (job.op == MVAE_PREP_ADD_I32 && job.src2 && !job.src) ||
                job.op > MVAE_PREP_BROADCAST_ROWS ||
                (job.op == MVAE_PREP_CONVERT_PAD && job.c < job.b) ||
                (job.kind != MVAE_F32 && job.kind != MVAE_BF16) || (job.op == MVAE_PREP_MAKE_TABLE && !job.src2) ||
                (job.op == MVAE_PREP_MAKE_TABLE && (job.c < 0 || job.c > 2 || (job.c == 1 && (job.b % 32)) || (job.c == 2 && (job.b % 256)))) ||
                (job.op == MVAE_PREP_ZERO && job.kind == MVAE_BF16 && (((size_t)job.a * job.b) & 1)))
                return MVAE_E_ARG;
            if (job.op == MVAE_PREP_PACK_RECURRENT) {
                const int K = job.c == 0 ? job.a : job.b, KG = job.kind == MVAE_BF16 ? 32 : 4;     // as mvae_pack_recurrent
                if (job.a <= 0 || (job.a % 16) || (job.b % 16) || (K % KG)) return MVAE_E_ARG;
            }
            pb.jobs[j] = job;
            // workgroups of the job: ~2048 output elements each, between 1 and 64
            size_t out = (size_t)(job.a > 0 ? job.a : 1) * (size_t)(job.b > 0 ? job.b : 1);
            if (job.op == MVAE_PREP_CONVERT_PAD) out = (size_t)job.a * job.c;
            if (job.op == MVAE_PREP_TRANSPOSE_CONVERT) out = (size_t)job.a * (job.c > job.b ? job.c : job.b);
            int nb = (int)((out + 2047) / 2048);
            nb = job.op == MVAE_PREP_ADD_I32 ? 1 : (nb < 1 ? 1 : (nb > 64 ? 64 : nb));
            pb.base[j] = total;
            total += nb;
        }
        pb.base[pb.n] = total;
        if (total > 0) hipLaunchKernelGGL(prepare_batch_k, dim3(total), dim3(256), 0, s, pb);
        MVAE_CHECK_LAUNCH();
    }
    return MVAE_OK;
}
// out (R, N) in TILE16 = xs[r] * w[n] + bias[n]: the input projection of a 1-feature layer, expanded so that the layer
// runs on the dense-input recurrent kernels.  One thread = 4 consecutive n of one row = its 8 / 16 bytes of a tile.
template <typename D>
__global__ void outer_bias_tile16_k(const float* __restrict__ xs, const float* __restrict__ w, const float* __restrict__ bias,
                                    D* __restrict__ out, int R, int N) {
    outer_bias_body<D>(xs, w, bias, out, R, N, (int)blockIdx.x, (int)gridDim.x);
}
extern "C" int mvae_outer_bias_tile16(const float* xs, const float* w, const float* bias, void* out, int32_t out_kind, int32_t R,
                                      int32_t N, void* stream) {
    if (!xs || !w || !bias || !out || R <= 0 || N <= 0 || (R % 16) || (N % 16)) return MVAE_E_ARG;
    if ((reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15)) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g(nblocks((size_t)R * N / 4)), b(256);
    if (out_kind == MVAE_F32)
        hipLaunchKernelGGL(outer_bias_tile16_k<float>, g, b, 0, s, xs, w, bias, (float*)out, R, N);
    else if (out_kind == MVAE_BF16)
        hipLaunchKernelGGL(outer_bias_tile16_k<bf16_t>, g, b, 0, s, xs, w, bias, (bf16_t*)out, R, N);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
// out (R, N) in TILE16 = table[idx[r]] + table2[idx2[r]]: the input projection of TWO-hot rows (attach_instruments: a pitch column and
// an instrument column per row, reference import_midi.py:288-292) written out for the dense-input recurrent kernels - a second
// table gather per step inside those kernels would double their VMEM instructions.  One thread = 4 consecutive n of one row.
template <typename D>
__global__ void gather2_tile16_k(const uint8_t* __restrict__ idx, const uint8_t* __restrict__ idx2, const D* __restrict__ table,
                                 const D* __restrict__ table2, D* __restrict__ out, int R, int N, int rowmajor) {
    const size_t total = (size_t)R * N / 4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = e >> 6;
        const int lane = (int)(e & 63), m = (int)(tile / (N >> 4)) * 16 + (lane & 15), n = (int)(tile % (N >> 4)) * 16 + (lane >> 4) * 4;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = st<D>::load(table + (size_t)idx[m] * N + n + i) + st<D>::load(table2 + (size_t)idx2[m] * N + n + i);
        st<D>::store4(rowmajor ? out + (size_t)m * N + n : out + e * 4, v);
    }
}
extern "C" int mvae_gather2_tile16(const uint8_t* idx, const uint8_t* idx2, const void* table, const void* table2, void* out,
                                   int32_t kind, int32_t R, int32_t N, int32_t layout, void* stream) {
    if (!idx || !idx2 || !table || !table2 || !out || R <= 0 || N <= 0 || (R % 16) || (N % 16) ||
        (layout != MVAE_TILE16 && layout != MVAE_ROWMAJOR))
        return MVAE_E_ARG;
    const int rowmajor = layout == MVAE_ROWMAJOR;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g(nblocks((size_t)R * N / 4)), b(256);
    if (kind == MVAE_F32)
        hipLaunchKernelGGL(gather2_tile16_k<float>, g, b, 0, s, idx, idx2, (const float*)table, (const float*)table2, (float*)out, R, N, rowmajor);
    else if (kind == MVAE_BF16)
        hipLaunchKernelGGL(gather2_tile16_k<bf16_t>, g, b, 0, s, idx, idx2, (const bf16_t*)table, (const bf16_t*)table2, (bf16_t*)out, R, N, rowmajor);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_make_table(const float* W, const float* bias, void* table, int32_t K, int32_t N, int32_t dk,
                               void* stream) {
    if (!W || !bias || !table || K <= 0 || N <= 0) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g(nblocks((size_t)K * N)), b(256);
    if (dk == MVAE_F32)
        hipLaunchKernelGGL(make_table_k<float>, g, b, 0, s, W, bias, (float*)table, K, N);
    else if (dk == MVAE_BF16)
        hipLaunchKernelGGL(make_table_k<bf16_t>, g, b, 0, s, W, bias, (bf16_t*)table, K, N);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_transpose_convert(const float* W, void* out, int32_t K, int32_t N, int32_t N_pad, int32_t dk,
                                      void* stream) {
    if (!W || !out || K <= 0 || N <= 0 || N_pad < N) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g(nblocks((size_t)K * N_pad)), b(256);
    if (dk == MVAE_F32)
        hipLaunchKernelGGL(transpose_convert_k<float>, g, b, 0, s, W, (float*)out, K, N, N_pad);
    else if (dk == MVAE_BF16)
        hipLaunchKernelGGL(transpose_convert_k<bf16_t>, g, b, 0, s, W, (bf16_t*)out, K, N, N_pad);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

extern "C" int mvae_signature_head_fwd(const float* zh, int32_t ldz, int32_t off, int32_t SD, int32_t B, const float* target,
                                       const float* row_weight, float* out, float* scalars, void* stream) {
    if (!zh || !out || SD <= 0 || B <= 0 || off < 0 || ldz < off + SD) return MVAE_E_ARG;
    hipLaunchKernelGGL(sig_head_fwd_k, dim3(nblocks(B, 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), zh, ldz, off, SD, B,
                       target, row_weight, out, scalars);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_signature_head_bwd(float* dz, int32_t lddz, int32_t off, int32_t SD, int32_t B, const float* out, const float* target,
                                       const float* row_weight, float weight, void* stream) {
    if (!dz || !out || !target || !row_weight || SD <= 0 || B <= 0 || off < 0 || lddz < off + SD) return MVAE_E_ARG;
    hipLaunchKernelGGL(sig_head_bwd_k, dim3(nblocks((size_t)B * SD)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dz, lddz,
                       off, SD, B, out, target, row_weight, weight);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_softmax_bwd_add(const float* probs, const float* dprobs, void* dlogits, int32_t kind, int32_t R, int32_t N,
                                    int32_t NP, void* stream) {
    if (!probs || !dprobs || !dlogits || R <= 0 || N <= 0 || NP < N) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (kind == MVAE_F32)
        hipLaunchKernelGGL(softmax_bwd_add_k<float>, dim3(nblocks(R)), dim3(256), 0, s, probs, dprobs, (float*)dlogits, R, N, NP);
    else if (kind == MVAE_BF16)
        hipLaunchKernelGGL(softmax_bwd_add_k<bf16_t>, dim3(nblocks(R)), dim3(256), 0, s, probs, dprobs, (bf16_t*)dlogits, R, N, NP);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_bi_concat(const void* f, const void* r, void* cat, void* cat_rev, int32_t kind, int32_t T, int32_t B, int32_t H,
                              void* stream) {
    const int esz = kind == MVAE_BF16 ? 2 : 4;
    if (!f || !r || !cat || T <= 0 || B <= 0 || H <= 0 || (H * esz) % 16 || (kind != MVAE_BF16 && kind != MVAE_F32)) return MVAE_E_ARG;
    const int hc = H * esz / 16;
    hipLaunchKernelGGL(bi_concat_k, dim3(nblocks((size_t)T * B * hc)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const uint4*)f, (const uint4*)r, (uint4*)cat, (uint4*)cat_rev, T, B, hc);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_add_time_reversed(void* dst, const void* a, const void* b, int32_t kind, int32_t T, size_t slab, void* stream) {
    if (!dst || !b || T <= 0 || slab == 0 || (slab % 4)) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nb = nblocks((size_t)T * slab / 4);
    if (kind == MVAE_F32)
        hipLaunchKernelGGL(add_time_reversed_k<float>, dim3(nb), dim3(256), 0, s, (float*)dst, (const float*)a, (const float*)b, T, slab);
    else if (kind == MVAE_BF16)
        hipLaunchKernelGGL(add_time_reversed_k<bf16_t>, dim3(nb), dim3(256), 0, s, (bf16_t*)dst, (const bf16_t*)a, (const bf16_t*)b, T,
                           slab);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_scalars_accumulate(float* acc, const float* x, int32_t n, float alpha, uint32_t plain_mask, void* stream) {
    if (!acc || !x || n < 0 || n > 32) return MVAE_E_ARG;
    if (n) hipLaunchKernelGGL(scalars_accumulate_k, dim3(1), dim3(32), 0, reinterpret_cast<hipStream_t>(stream), acc, x, n, alpha,
                              plain_mask);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_copy2d_f32(float* dst, int32_t ldd, const float* src, int32_t lds, int32_t rows, int32_t cols, int32_t src_row0,
                               int32_t zero_rows, void* stream) {
    if (!dst || !src || rows < 0 || cols < 0 || ldd < cols || lds < cols || zero_rows < 0 || src_row0 + zero_rows < 0)
        return MVAE_E_ARG;
    if (rows && cols)
        hipLaunchKernelGGL(copy2d_f32_k, dim3(nblocks((size_t)rows * cols)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dst,
                           ldd, src, lds, rows, cols, src_row0, zero_rows);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
// the fused history pre-pass (include/midivae_hip.h): z' = mu + exp(lv / 2) * eps2 - the expression of the latent kernels, so a
// pre-pass run on its own gives the same bits - stored and rolled by one window into the history columns
__global__ void history_from_latent_k(const float* mu, const float* lv, const float* eps2, int B, int Bp, int Z, float* hist, int ldh,
                                      const float* prev, float* z_out, int ldo) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)Bp * Z; e += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / Z), j = (int)(e % Z);
        if (b < B && z_out) z_out[(size_t)b * ldo + j] = mu[e] + expf(0.5f * lv[e]) * eps2[e];
        float h = 0.0f;
        if (b == 0) h = prev ? prev[j] : 0.0f;
        else if (b < B) {
            const size_t q = e - Z;
            h = mu[q] + expf(0.5f * lv[q]) * eps2[q];
        }
        hist[(size_t)b * ldh + j] = h;
    }
}
extern "C" int mvae_history_from_latent(const float* mu, const float* logvar, const float* eps2, int32_t B, int32_t B_pad, int32_t Z,
                                        float* hist, int32_t ldh, const float* prev, float* z_out, int32_t ldo, void* stream) {
    if (!mu || !logvar || !eps2 || !hist || B <= 0 || B_pad < B || Z <= 0 || ldh < Z || (z_out && ldo < Z)) return MVAE_E_ARG;
    hipLaunchKernelGGL(history_from_latent_k, dim3(nblocks((size_t)B_pad * Z)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mu,
                       logvar, eps2, B, B_pad, Z, hist, ldh, prev, z_out, ldo);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                              float eps, int32_t t, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || t < 1) return MVAE_E_ARG;
    if (n == 0) return MVAE_OK;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    hipLaunchKernelGGL(adam_k, dim3(nblocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n,
                       (float)lr_t, beta1, beta2, eps, grad_scale);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_adam_step_dev(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1,
                                  float beta2, float eps, int32_t* t_done, float grad_scale, int32_t zero_grad,
                                  const uint32_t* guard, void* stream) {
    if (!p || !g || !m || !v || !t_done) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (n) hipLaunchKernelGGL(adam_dev_k, dim3(nblocks(vec ? (n + 3) / 4 : n)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2,
                              eps, grad_scale, t_done, (int)(zero_grad & MVAE_ADAM_ZERO_GRAD), vec, guard);
    if (!(zero_grad & MVAE_ADAM_KEEP_COUNT)) hipLaunchKernelGGL(bump_k, dim3(1), dim3(1), 0, s, t_done, guard);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
extern "C" int mvae_rmsprop_step(float* p, float* g, float* v, size_t n, float lr, float rho, float eps,
                                 float grad_scale, int32_t zero_grad, const uint32_t* guard, void* stream) {
    if (!p || !g || !v) return MVAE_E_ARG;
    if (n == 0) return MVAE_OK;
    const int vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    hipLaunchKernelGGL(rmsprop_k, dim3(nblocks(vec ? (n + 3) / 4 : n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, v,
                       n, lr, rho, eps, grad_scale, (int)zero_grad, vec, guard);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
