// The Dense chain around the latent as ONE launch each way.
//
// Forward (reference vae_definition.py:483-516, 519-530): [h_notes | h_instr | h_vel] -> pack Dense (tanh) -> extra Dense
// (tanh) -> z_mean / z_log_var Denses (on the two halves of the vector when split_lstm_vector) -> KL term, sampling, style
// softmax -> the decoder's initial-state Denses (tanh) on [z | history].  Backward: the same chain reversed.
// As separate launches these are 6 and 9 tiny dependent kernels in the one stretch of the training step where nothing
// else can run ahead of them - every one of them queues behind the gradient GEMMs that fill the chip at that moment:
// 0.4 + 0.8 ms of a 10.8 ms step.  Here one workgroup owns LR batch rows through the whole chain, the row vectors stay
// in LDS, the f32 master weights stream from L2 (1.7 MB, read once per workgroup), plain f32 FMAs (the chain is ~0.2 GFLOP).
#include "common.h"

namespace {

constexpr int LR = 4;            // batch rows per workgroup (B/4 workgroups: the chain is FMA-bound, 64 of 256 CUs at B=256)
constexpr int LT = 512;          // threads per workgroup (256 registers per thread: the unrolled loads must not spill)
constexpr float CE_EPS = 1e-7f;

// y[r][n] = act(bias[n] + sum_k x[r][k] W[k][n]) for the LR rows.  x: LDS, row stride ldx (multiple of 4 floats).
// W: (K, N) row-major, row stride ldw; N % 4 == 0.  A thread owns 4 adjacent columns (16-byte weight loads, 16 of them in
// flight: 256 bytes per thread - the chain is bound by the latency of these L2 reads, not by their volume) and one of
// KS = LT / (N/4) slices of K; the slices meet in `red` (4 * LR * LT floats).  Results go to ys (LDS, stride ldy) and /
// or yg (global, stride ldg, 16-byte aligned rows).
template <bool TANH>
__device__ void dense_rows(const float* xs, int ldx, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                           int K, int N, float* red, float* ys, int ldy, float* yg, int ldg) {
    const int t = threadIdx.x, NV = N / 4;
    for (int c0 = 0; c0 < NV; c0 += LT) {
        const int cur = min(LT, NV - c0), KS = LT / cur, nvl = t % cur, ks = t / cur;
        const int per = ((K + KS - 1) / KS + 3) & ~3;
        if (ks < KS) {
            const int n = (c0 + nvl) * 4;
            f32x4 acc[LR];
#pragma unroll
            for (int r = 0; r < LR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int k0 = ks * per, k1 = min(K, k0 + per);
            int k = k0;
            for (; k + 16 <= k1; k += 16) {
                f32x4 wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = *reinterpret_cast<const f32x4*>(W + (size_t)(k + u) * ldw + n);
#pragma unroll
                for (int r = 0; r < LR; ++r)
#pragma unroll
                    for (int u = 0; u < 16; u += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(xs + r * ldx + k + u);
                        acc[r] += x[0] * wv[u] + x[1] * wv[u + 1] + x[2] * wv[u + 2] + x[3] * wv[u + 3];
                    }
            }
            for (; k + 4 <= k1; k += 4) {
                f32x4 wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wv[u] = *reinterpret_cast<const f32x4*>(W + (size_t)(k + u) * ldw + n);
#pragma unroll
                for (int r = 0; r < LR; ++r) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(xs + r * ldx + k);
                    acc[r] += x[0] * wv[0] + x[1] * wv[1] + x[2] * wv[2] + x[3] * wv[3];
                }
            }
            for (; k < k1; ++k) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + (size_t)k * ldw + n);
#pragma unroll
                for (int r = 0; r < LR; ++r) acc[r] += xs[r * ldx + k] * w0;
            }
#pragma unroll
            for (int r = 0; r < LR; ++r) *reinterpret_cast<f32x4*>(red + ((size_t)(ks * LR + r) * cur + nvl) * 4) = acc[r];
        }
        __syncthreads();
        for (int e = t; e < LR * cur; e += LT) {
            const int r = e / cur, c = e % cur, n = (c0 + c) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bias) v = f32x4{bias[n], bias[n + 1], bias[n + 2], bias[n + 3]};
            for (int s = 0; s < KS; ++s) v += *reinterpret_cast<const f32x4*>(red + ((size_t)(s * LR + r) * cur + c) * 4);
            if (TANH) v = f32x4{tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])};
            if (ys) *reinterpret_cast<f32x4*>(ys + r * ldy + n) = v;
            if (yg) *reinterpret_cast<f32x4*>(yg + (size_t)r * ldg + n) = v;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int pad4(int n) { return (n + 3) & ~3; }

__global__ __launch_bounds__(LT) void latent_chain_fwd_k(const mvae_latent_chain_fwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, Z = a.Z, zin = a.zin, ncH = a.ncat * H, t = threadIdx.x;
    const int r0 = blockIdx.x * LR;
    float* xcat = sm;                         // [LR][ncH]
    float* h1 = xcat + LR * pad4(ncH);        // [LR][H]
    float* h2 = h1 + LR * H;                  // [LR][H]
    float* mu = h2 + LR * H;                  // [LR][Z]
    float* lv = mu + LR * pad4(Z);
    float* zh = lv + LR * pad4(Z);            // [LR][zin]
    float* red = zh + LR * pad4(zin);         // [4 * LR * LT]
    for (int e = t; e < LR * ncH; e += LT) {
        const int r = e / ncH, c = e % ncH;
        xcat[r * pad4(ncH) + c] = a.cat[(size_t)(r0 + r) * ncH + c];
    }
    __syncthreads();
    const float* h = xcat;
    int ldh = pad4(ncH);
    if (a.w_pack) {
        dense_rows<true>(h, ldh, a.w_pack, H, a.b_pack, ncH, H, red, h1, H, a.pack + (size_t)r0 * H, H);
        h = h1;
        ldh = H;
    }
    if (a.w_extra) {
        dense_rows<true>(h, ldh, a.w_extra, H, a.b_extra, H, H, red, h2, H, a.extra + (size_t)r0 * H, H);
        h = h2;
        ldh = H;
    }
    const int h1w = a.split ? H / 2 : H, h2o = a.split ? h1w : 0, h2w = a.split ? H - h1w : H;
    dense_rows<false>(h, ldh, a.w_mu, Z, a.b_mu, h1w, Z, red, mu, pad4(Z), a.mu + (size_t)r0 * Z, Z);
    dense_rows<false>(h + h2o, ldh, a.w_lv, Z, a.b_lv, h2w, Z, red, lv, pad4(Z), a.logvar + (size_t)r0 * Z, Z);
    // KL, sampling, style softmax: one wave per row (as mvae_latent_fwd); rows >= B_valid are padding
    {
        const int w = t >> 6, l = t & 63;
        if (w < LR) {
            const int b = r0 + w;
            const float plv = 2.0f * logf(a.prior_std), pvar = a.prior_std * a.prior_std;
            float kl = 0.0f;
            for (int j = l; j < zin; j += 64) {
                float v;
                if (j < Z) {
                    const float m = mu[w * pad4(Z) + j], lg = lv[w * pad4(Z) + j];
                    const float d = m - a.prior_mean;
                    kl += 1.0f + lg - plv - (d * d + expf(lg)) / pvar;
                    v = m + expf(0.5f * lg) * a.eps[(size_t)b * Z + j];
                    a.zh[(size_t)b * zin + j] = v;
                } else {
                    v = a.zh[(size_t)b * zin + j];       // history columns, staged by the caller
                }
                zh[w * pad4(zin) + j] = v;
            }
            kl = wave_sum(kl);
            if (l == 0 && b < a.B_valid) {
                atomicAdd(a.scalars, a.inv_batch * a.beta * (-0.5f) * kl);
                if (a.style_target && a.C > 0) {
                    const int C = a.C;
                    const float* zr = zh + w * pad4(zin);
                    float mx = -INFINITY;
                    for (int c = 0; c < C; ++c) mx = fmaxf(mx, zr[c]);
                    float sum = 0.0f;
                    for (int c = 0; c < C; ++c) sum += expf(zr[c] - mx);
                    const int tg = a.style_target[b];
                    float pt = 0.0f, pm = -1.0f;
                    int am = 0;
                    for (int c = 0; c < C; ++c) {
                        const float p = expf(zr[c] - mx) / sum;
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
        __syncthreads();
    }
    if (a.w_init)
        dense_rows<true>(zh, pad4(zin), a.w_init, a.n_init, a.b_init, zin, a.n_init, red, nullptr, 0,
                         a.S + (size_t)r0 * a.n_init, a.n_init);
}

__global__ __launch_bounds__(LT) void latent_chain_bwd_k(const mvae_latent_chain_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, Z = a.Z, zin = a.zin, ncH = a.ncat * H, NI = a.n_init, t = threadIdx.x;
    const int r0 = blockIdx.x * LR;
    float* ds = sm;                           // [LR][NI]
    float* dzh = ds + LR * pad4(NI);          // [LR][zin]
    float* dml = dzh + LR * pad4(zin);        // [LR][2Z]: dmu | dlogvar
    float* dt = dml + LR * 2 * pad4(Z);       // [LR][H]
    float* dt2 = dt + LR * H;                 // [LR][H]
    float* dc = dt2 + LR * H;                 // [LR][ncH]
    float* red = dc + LR * pad4(ncH);         // [4 * LR * LT]
    // initial-state Denses: dS <- dS (1 - S^2)
    for (int e = t; e < LR * NI; e += LT) {
        const int r = e / NI, c = e % NI;
        const size_t g = (size_t)(r0 + r) * NI + c;
        const float s = a.S[g], v = a.dS[g] * (1.0f - s * s);
        a.dS[g] = v;
        ds[r * pad4(NI) + c] = v;
    }
    __syncthreads();
    // every "gradient times W^T" below is dense_rows on the transposed matrix (prepared once per step with the other
    // derived weight copies): coalesced weight reads, no cross-lane sums
    dense_rows<false>(ds, pad4(NI), a.wt_init, zin, nullptr, NI, zin, red, dzh, pad4(zin), nullptr, 0);
    // latent block (as mvae_latent_bwd)
    {
        const float pvar = a.prior_std * a.prior_std;
        for (int e = t; e < LR * Z; e += LT) {
            const int r = e / Z, j = e % Z, b = r0 + r;
            float dmu = 0.0f, dlv = 0.0f;
            if (b < a.B_valid) {
                float dz = dzh[r * pad4(zin) + j];
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
                const float m = a.mu[(size_t)b * Z + j], lg = a.logvar[(size_t)b * Z + j];
                dmu = dz + a.beta * (m - a.prior_mean) / pvar * a.inv_batch;
                dlv = dz * a.eps[(size_t)b * Z + j] * 0.5f * expf(0.5f * lg) +
                      a.beta * (-0.5f) * (1.0f - expf(lg) / pvar) * a.inv_batch;
            }
            a.dmu[(size_t)b * Z + j] = dmu;
            a.dlogvar[(size_t)b * Z + j] = dlv;
            dml[r * 2 * pad4(Z) + j] = dmu;
            dml[r * 2 * pad4(Z) + pad4(Z) + j] = dlv;
        }
        for (int e = t; e < LR * zin; e += LT) a.dzh[(size_t)(r0 + e / zin) * zin + e % zin] = dzh[(e / zin) * pad4(zin) + e % zin];
        __syncthreads();
    }
    // z_mean / z_log_var Denses -> d(tail)
    const int h1w = a.split ? H / 2 : H, h2w = a.split ? H - h1w : H;
    dense_rows<false>(dml, 2 * pad4(Z), a.wt_mu, h1w, nullptr, Z, h1w, red, dt, H, nullptr, 0);
    if (a.split) {
        dense_rows<false>(dml + pad4(Z), 2 * pad4(Z), a.wt_lv, h2w, nullptr, Z, h2w, red, dt + h1w, H, nullptr, 0);
    } else {
        dense_rows<false>(dml + pad4(Z), 2 * pad4(Z), a.wt_lv, H, nullptr, Z, H, red, dt2, H, nullptr, 0);
        for (int e = t; e < LR * H; e += LT) dt[e] += dt2[e];
        __syncthreads();
    }
    float* cur = dt;
    if (a.wt_extra) {
        for (int e = t; e < LR * H; e += LT) {
            const size_t g = (size_t)(r0 + e / H) * H + e % H;
            const float y = a.extra[g], v = dt[e] * (1.0f - y * y);
            dt[e] = v;
            a.d_extra[g] = v;
        }
        __syncthreads();
        dense_rows<false>(dt, H, a.wt_extra, H, nullptr, H, H, red, dt2, H, nullptr, 0);
        cur = dt2;
    }
    if (a.wt_pack) {
        for (int e = t; e < LR * H; e += LT) {
            const size_t g = (size_t)(r0 + e / H) * H + e % H;
            const float y = a.pack[g], v = cur[e] * (1.0f - y * y);
            cur[e] = v;
            a.d_pack[g] = v;
        }
        __syncthreads();
        dense_rows<false>(cur, H, a.wt_pack, ncH, nullptr, H, ncH, red, nullptr, 0, a.dcat + (size_t)r0 * ncH, ncH);
    } else {
        for (int e = t; e < LR * H; e += LT) a.dcat[(size_t)(r0 + e / H) * H + e % H] = cur[e];
    }
}

size_t fwd_lds(const mvae_latent_chain_fwd_args& a) {
    auto p4 = [](int n) { return (n + 3) & ~3; };
    return sizeof(float) * ((size_t)LR * (p4(a.ncat * a.H) + 2 * a.H + 2 * p4(a.Z) + p4(a.zin)) + (size_t)4 * LR * LT);
}
size_t bwd_lds(const mvae_latent_chain_bwd_args& a) {
    auto p4 = [](int n) { return (n + 3) & ~3; };
    return sizeof(float) * ((size_t)LR * (p4(a.n_init) + p4(a.zin) + 2 * p4(a.Z) + 2 * a.H + p4(a.ncat * a.H)) + (size_t)4 * LR * LT);
}

}  // namespace

extern "C" int mvae_latent_chain_fwd(const mvae_latent_chain_fwd_args* a, void* stream) {
    if (!a || a->B <= 0 || (a->B % LR) || a->H <= 0 || (a->H % 8) || a->Z <= 0 || (a->Z % 4) || (a->zin % 4) || a->zin < a->Z ||
        a->ncat < 1 || !a->cat || !a->w_mu || !a->w_lv || !a->mu || !a->logvar || !a->eps || !a->zh || !a->scalars)
        return MVAE_E_ARG;
    if ((a->w_pack && !a->pack) || (a->w_extra && !a->extra) || (a->w_init && (!a->S || a->n_init <= 0))) return MVAE_E_ARG;
    if (!a->w_pack && a->ncat != 1) return MVAE_E_ARG;
    if (a->w_init && (a->n_init % 4)) return MVAE_E_ARG;
    const size_t lds = fwd_lds(*a);
    if (lds > 160 * 1024) return MVAE_E_UNSUPPORTED;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&latent_chain_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL(latent_chain_fwd_k, dim3(a->B / LR), dim3(LT), lds, reinterpret_cast<hipStream_t>(stream), *a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

extern "C" int mvae_latent_chain_bwd(const mvae_latent_chain_bwd_args* a, void* stream) {
    if (!a || a->B <= 0 || (a->B % LR) || a->H <= 0 || (a->H % 8) || a->Z <= 0 || (a->Z % 4) || (a->zin % 4) || a->zin < a->Z ||
        a->ncat < 1 || a->n_init <= 0 || (a->n_init % 4) || !a->S || !a->dS || !a->wt_init || !a->wt_mu || !a->wt_lv || !a->mu ||
        !a->logvar || !a->eps || !a->dzh || !a->dmu || !a->dlogvar || !a->dcat)
        return MVAE_E_ARG;
    if ((a->wt_pack && (!a->pack || !a->d_pack)) || (a->wt_extra && (!a->extra || !a->d_extra))) return MVAE_E_ARG;
    if (!a->wt_pack && a->ncat != 1) return MVAE_E_ARG;
    const size_t lds = bwd_lds(*a);
    if (lds > 160 * 1024) return MVAE_E_UNSUPPORTED;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&latent_chain_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL(latent_chain_bwd_k, dim3(a->B / LR), dim3(LT), lds, reinterpret_cast<hipStream_t>(stream), *a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
