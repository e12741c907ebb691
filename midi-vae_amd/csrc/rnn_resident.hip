// Resident-weights recurrent kernels for the production shape (H = 256, bf16 MFMA operands) on gfx950.
//
// Same contract and numerics as rnn.hip (mvae_rnn_fwd / mvae_rnn_bwd dispatch here when H == 256 and
// dtype == MVAE_BF16), different data placement:
//
//   the recurrent kernel U (256 x G*256 bf16 = 384 KiB GRU / 512 KiB LSTM) is loaded ONCE per launch and stays
//   on chip for all T steps, split between the register file and LDS of the one CU that owns 16 batch rows.
//   A CU has 4 SIMDs x 512 registers x 64 lanes x 4 B = 512 KiB of registers and 160 KiB of LDS; the workgroup is
//   4 waves (one per SIMD, launch_bounds(256,1)) so every wave may use the whole 512-register budget.  Wave w owns
//   hidden units [64w, 64w+64) for all gates: NREG of its MFMA A-fragments live in registers (the matrix pipe
//   reads A straight from VGPR/AGPR), the remaining ones in a private LDS slab read back with ds_read_b128.
//
//   Per step nothing but x_t and the saved activations touch HBM: h_t travels through a 8 KiB LDS tile, the
//   cell state and the f32 master copy of h never leave registers, and the per-step critical path is
//   128 (LSTM) / 96 (GRU) MFMAs per wave + one (GRU: two) workgroup barrier.  The generic kernel re-streams U from
//   L2 every step (17 us/step measured at T=512,B=256); this one is bounded by the CU's MFMA rate (~1 us/step).
//
//   x_t (and in backward the saved gates / cell states / upstream gradient) are prefetched ONE STEP ahead into a
//   small register queue, re-issued right after they are consumed, so HBM latency is covered by a whole step.
#include "common.h"

namespace {

constexpr int RH = 256;           // hidden size this file is specialised for
constexpr int RS = RH / 32;       // k-groups of the forward contraction (8)
constexpr int RNT = 4;            // unit tiles (16 units) per wave
constexpr int RLDH = RH + 8;      // padded bf16 row of the h tile in LDS

typedef u16x8 frag;

// Four MFMAs that share one B fragment.  U fragments in the accumulator half of the register file are named
// with the "a" constraint so the matrix pipe reads them in place (hipcc, left alone, parks such values in AGPRs
// but copies them back with v_accvgpr_read before every MFMA: 2 VALU per MFMA).  hipcc neither pads hazards
// inside an asm statement nor models the MFMA: the leading s_nop 1 covers "VALU-written VGPR -> MFMA operand",
// chain_done() covers "MFMA result -> VALU reader" (4-pass XDL needs 8 states; 10 given).
template <bool AG>
__device__ __forceinline__ void mfma4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const frag& u0, const frag& u1,
                                      const frag& u2, const frag& u3, const frag& b) {
    if (AG)
        asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
            "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, %3"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
            : "a"(u0), "a"(u1), "a"(u2), "a"(u3), "v"(b));
    else
        asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
            "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, %3"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
            : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(b));
}
__device__ __forceinline__ void chain_done(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    asm volatile("s_nop 9" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
}
// load 4 fragments straight into AGPRs and wait for them (hipcc does not count asm loads)
__device__ __forceinline__ void load4_agpr(frag& u0, frag& u1, frag& u2, frag& u3, const frag* p0, const frag* p1,
                                           const frag* p2, const frag* p3) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                 : "=&a"(u0), "=&a"(u1), "=&a"(u2), "=&a"(u3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}

// Where the FPW fragments of a wave live, by their index f in order of use (all bounds multiples of 4):
//   [0, NA) accumulator registers | [NA, NA+NV) vector registers | [NA+NV, NA+NV+NL) LDS slab | rest: re-read
//   from L2 every step (only when the 512 KiB of LSTM weights + working set exceed registers + LDS).
#define RES_DECLARE_U(FPW_)                                                                                        \
    frag ua[NA > 0 ? NA : 4];                                                                                     \
    frag uv[NV > 0 ? NV : 4];                                                                                     \
    _Pragma("unroll") for (int f = 0; f < (FPW_); f += 4) {                                                       \
        const frag* p0 = up + (size_t)frag_src(f) * 64 + l;                                                       \
        const frag* p1 = up + (size_t)frag_src(f + 1) * 64 + l;                                                   \
        const frag* p2 = up + (size_t)frag_src(f + 2) * 64 + l;                                                   \
        const frag* p3 = up + (size_t)frag_src(f + 3) * 64 + l;                                                   \
        if (f < NA) load4_agpr(ua[f < NA ? f : 0], ua[f < NA ? f + 1 : 1], ua[f < NA ? f + 2 : 2],                \
                               ua[f < NA ? f + 3 : 3], p0, p1, p2, p3);                                           \
        else if (f < NA + NV) {                                                                                   \
            uv[f - NA < NV ? f - NA : 0] = *p0; uv[f - NA < NV ? f - NA + 1 : 1] = *p1;                           \
            uv[f - NA < NV ? f - NA + 2 : 2] = *p2; uv[f - NA < NV ? f - NA + 3 : 3] = *p3;                       \
        } else if (f < NA + NV + NL) {                                                                            \
            myl[(size_t)(f - NA - NV) * 64] = *p0; myl[(size_t)(f - NA - NV + 1) * 64] = *p1;                     \
            myl[(size_t)(f - NA - NV + 2) * 64] = *p2; myl[(size_t)(f - NA - NV + 3) * 64] = *p3;                 \
        }                                                                                                         \
    }

// acc{0..3} += U[f..f+3] * bf
#define RES_MFMA4(c0, c1, c2, c3, f, bf)                                                                           \
    do {                                                                                                          \
        if ((f) < NA) mfma4<true>(c0, c1, c2, c3, ua[(f) < NA ? (f) : 0], ua[(f) < NA ? (f) + 1 : 1],              \
                                  ua[(f) < NA ? (f) + 2 : 2], ua[(f) < NA ? (f) + 3 : 3], bf);                    \
        else if ((f) < NA + NV) mfma4<false>(c0, c1, c2, c3, uv[(f) - NA < NV ? (f) - NA : 0],                     \
                                             uv[(f) - NA < NV ? (f) - NA + 1 : 1], uv[(f) - NA < NV ? (f) - NA + 2 : 2], \
                                             uv[(f) - NA < NV ? (f) - NA + 3 : 3], bf);                           \
        else if ((f) < NA + NV + NL) {                                                                            \
            const frag t0 = myl[(size_t)((f) - NA - NV) * 64], t1 = myl[(size_t)((f) - NA - NV + 1) * 64];        \
            const frag t2 = myl[(size_t)((f) - NA - NV + 2) * 64], t3 = myl[(size_t)((f) - NA - NV + 3) * 64];    \
            mfma4<false>(c0, c1, c2, c3, t0, t1, t2, t3, bf);                                                     \
        } else {                                                                                                  \
            const frag t0 = up[(size_t)frag_src(f) * 64 + l], t1 = up[(size_t)frag_src((f) + 1) * 64 + l];        \
            const frag t2 = up[(size_t)frag_src((f) + 2) * 64 + l], t3 = up[(size_t)frag_src((f) + 3) * 64 + l];  \
            mfma4<false>(c0, c1, c2, c3, t0, t1, t2, t3, bf);                                                     \
        }                                                                                                         \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int CELL, int XMODE, int NA, int NV, int NL>
__global__ __launch_bounds__(256, 1) void rnn_fwd_res_k(const mvae_rnn_fwd_args a) {
    constexpr int G = mvae_gates(CELL), GH = G * RH;
    constexpr int FPW = G * RNT * RS;              // fragments per wave
    constexpr int NLDS = NL;
    static_assert(NA % 4 == 0 && NV % 4 == 0 && NL % 4 == 0 && NA + NV + NL <= FPW, "fragment classes");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* hbuf = reinterpret_cast<bf16_t*>(smem);                         // [2][16][RLDH]
    bf16_t* rhbuf = hbuf + 2 * 16 * RLDH;                                   // [16][RLDH]      (GRU)
    frag* ulds = reinterpret_cast<frag*>(rhbuf + (CELL == MVAE_GRU ? 16 * RLDH : 0));   // [4][NLDS][64]
    float* wb = reinterpret_cast<float*>(ulds + 4 * NLDS * 64);             // [2][GH]         (SCALAR)

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, r = l & 15;
    const int T = a.T, B = a.B;
    const int b = blockIdx.x * 16 + r;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.u_pack);
    bf16_t* __restrict__ hs = reinterpret_cast<bf16_t*>(a.hs);
    bf16_t* __restrict__ cs = reinterpret_cast<bf16_t*>(a.cs);
    bf16_t* __restrict__ acts = reinterpret_cast<bf16_t*>(a.acts);
    const bf16_t* __restrict__ xp = reinterpret_cast<const bf16_t*>(a.xp);
    const bf16_t* __restrict__ table = reinterpret_cast<const bf16_t*>(a.table);
    frag* myl = ulds + (size_t)w * NLDS * 64 + l;

    // fragment f of this wave, in order of use.  LSTM: f = (n*8 + ks)*4 + g.
    // GRU: phase A (z,r) f = ((np*8 + ks)*2 + nn)*2 + g for tile pair np, then phase B (candidate) f = 64 + ks*4 + n.
    auto frag_src = [&](int f) -> int {        // index of fragment f in the packed U (units of 64-lane fragments)
        int g, n, ks;
        if (CELL == MVAE_GRU) {
            if (f < 64) { g = f & 1; const int nn = (f >> 1) & 1; ks = (f >> 2) & 7; n = (f >> 5) * 2 + nn; }
            else { g = 2; n = (f - 64) & 3; ks = (f - 64) >> 2; }
        } else {
            g = f % G; ks = (f / G) % RS; n = f / (G * RS);
        }
        return (g * (RH / 16) + w * RNT + n) * RS + ks;
    };
    RES_DECLARE_U(FPW)

    int ub[RNT];
#pragma unroll
    for (int n = 0; n < RNT; ++n) ub[n] = w * 64 + n * 16 + q * 4;
    const int ld0 = a.h0_ld ? a.h0_ld : RH, ldl = a.h_last_ld ? a.h_last_ld : RH;

    f32x4 hreg[RNT], creg[RNT];
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        hreg[n] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + (size_t)bb * ld0 + ub[n]) : z4;
        creg[n] = (CELL == MVAE_LSTM && a.c0) ? *reinterpret_cast<const f32x4*>(a.c0 + (size_t)bb * ld0 + ub[n]) : z4;
        st<bf16_t>::store4(hbuf + r * RLDH + ub[n], hreg[n]);
        if (valid) {
            if (hs) st<bf16_t>::store4(hs + (size_t)b * RH + ub[n], hreg[n]);
            if (CELL == MVAE_LSTM && cs) st<bf16_t>::store4(cs + (size_t)b * RH + ub[n], creg[n]);
        }
    }
    if (XMODE == MVAE_X_SCALAR) {
        for (int i = tid; i < GH; i += 256) {
            wb[i] = a.w_row[i];
            wb[GH + i] = a.bias[i];
        }
    }

    // ---- x queue: packed bf16x4 per (tile, gate), always holding the NEXT step to be consumed --------------
    u16x4 xq[RNT][G];
    float xs_next = 0.0f;
    int i_next = 0, i_next2 = 0;
    auto xload = [&](int t, int n, int g, int irow) -> u16x4 {
        if (XMODE == MVAE_X_DENSE) return *reinterpret_cast<const u16x4*>(xp + ((size_t)t * B + bb) * GH + g * RH + ub[n]);
        if (XMODE == MVAE_X_INDEX) return *reinterpret_cast<const u16x4*>(table + (size_t)irow * GH + g * RH + ub[n]);
        return *reinterpret_cast<const u16x4*>(reinterpret_cast<const bf16_t*>(a.xp0) + (size_t)bb * GH + g * RH + ub[n]);
    };
    if (XMODE == MVAE_X_INDEX) {
        i_next = a.idx[bb];
        i_next2 = a.idx[(size_t)(T > 1 ? 1 : 0) * B + bb];
    }
    if (XMODE == MVAE_X_SCALAR) xs_next = a.xs[bb];
    if (XMODE != MVAE_X_SCALAR) {
#pragma unroll
        for (int n = 0; n < RNT; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g) xq[n][g] = xload(0, n, g, i_next);
    }
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : t;             // step whose inputs are (re)loaded during this step
        const float xs_cur = xs_next;
        int i_row = i_next2;                              // table row of step t+1
        if (XMODE == MVAE_X_SCALAR) xs_next = a.xs[(size_t)tn * B + bb];
        if (XMODE == MVAE_X_INDEX) {
            i_next = i_next2;
            i_next2 = a.idx[(size_t)(t + 2 < T ? t + 2 : T - 1) * B + bb];
        }
        // x for (gate g, tile n) of THIS step as f32x4
        auto xval = [&](int n, int g) -> f32x4 {
            if (XMODE == MVAE_X_SCALAR) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wb + g * RH + ub[n]);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(wb + GH + g * RH + ub[n]);
                return xs_cur * w4 + b4;
            }
            u16x4 p = xq[n][g];
            if (XMODE == MVAE_X_CONST) asm volatile("" : "+v"(p));   // keep the loop-invariant row PACKED (no LICM of the unpack)
            return f32x4{bf2f(p[0]), bf2f(p[1]), bf2f(p[2]), bf2f(p[3])};
        };
        auto refill = [&](int n) {
            if (XMODE == MVAE_X_DENSE || XMODE == MVAE_X_INDEX) {
#pragma unroll
                for (int g = 0; g < G; ++g) xq[n][g] = xload(tn, n, g, i_row);
            }
        };
        const bf16_t* hrow = hbuf + cur * 16 * RLDH + r * RLDH + q * 8;
        bf16_t* hnext = hbuf + (cur ^ 1) * 16 * RLDH + r * RLDH;

        if (CELL == MVAE_LSTM || CELL == MVAE_RNN) {
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                f32x4 acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < RS; ++ks) {
                    const frag bf = *reinterpret_cast<const frag*>(hrow + ks * 32);
                    RES_MFMA4(acc[0], acc[1], acc[2], acc[3], (n * RS + ks) * 4, bf);
                }
                chain_done(acc[0], acc[1], acc[2], acc[3]);
                f32x4 hnew;
                if (CELL == MVAE_LSTM) {
                    const f32x4 xi = xval(n, 0), xf = xval(n, 1), xg = xval(n, 2), xo = xval(n, 3);
                    f32x4 ig, fg, gg, og;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ig[i] = hard_sigmoid(acc[0][i] + xi[i]);
                        fg[i] = hard_sigmoid(acc[1][i] + xf[i]);
                        gg[i] = tanh_f(acc[2][i] + xg[i]);
                        og[i] = hard_sigmoid(acc[3][i] + xo[i]);
                        creg[n][i] = fg[i] * creg[n][i] + ig[i] * gg[i];
                        hnew[i] = og[i] * tanh_f(creg[n][i]);
                    }
                    if (valid) {
                        if (acts) {
                            bf16_t* ap = acts + ((size_t)t * B + b) * GH + ub[n];
                            st<bf16_t>::store4(ap, ig);
                            st<bf16_t>::store4(ap + RH, fg);
                            st<bf16_t>::store4(ap + 2 * RH, gg);
                            st<bf16_t>::store4(ap + 3 * RH, og);
                        }
                        if (cs) st<bf16_t>::store4(cs + ((size_t)(t + 1) * B + b) * RH + ub[n], creg[n]);
                    }
                } else {
                    const f32x4 x0 = xval(n, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) hnew[i] = tanh_f(acc[0][i] + x0[i]);
                    if (acts && valid) st<bf16_t>::store4(acts + ((size_t)t * B + b) * GH + ub[n], hnew);
                }
                refill(n);
                st<bf16_t>::store4(hnext + ub[n], hnew);
                if (valid) {
                    if (hs) st<bf16_t>::store4(hs + ((size_t)(t + 1) * B + b) * RH + ub[n], hnew);
                    // the f32 master copy of h is not needed by LSTM / SimpleRNN steps: emit the final state here
                    if (a.h_last && t == T - 1) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub[n]) = hnew;
                }
            }
        } else {
            // ---- GRU phase A: z, r for tile pairs -------------------------------------------------------------
            f32x4 zg[RNT], rg[RNT];
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                f32x4 acc[2][2];
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[nn][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < RS; ++ks) {
                    const frag bf = *reinterpret_cast<const frag*>(hrow + ks * 32);
                    RES_MFMA4(acc[0][0], acc[0][1], acc[1][0], acc[1][1], (np * RS + ks) * 4, bf);
                }
                chain_done(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
                    const int n = np * 2 + nn;
                    const f32x4 xz = xval(n, 0), xr = xval(n, 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        zg[n][i] = hard_sigmoid(acc[nn][0][i] + xz[i]);
                        rg[n][i] = hard_sigmoid(acc[nn][1][i] + xr[i]);
                    }
                    st<bf16_t>::store4(rhbuf + r * RLDH + ub[n], rg[n] * hreg[n]);
                }
            }
            __syncthreads();
            // ---- GRU phase B: candidate ---------------------------------------------------------------------
            f32x4 acc[RNT];
#pragma unroll
            for (int n = 0; n < RNT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16_t* rhrow = rhbuf + r * RLDH + q * 8;
#pragma unroll
            for (int ks = 0; ks < RS; ++ks) {
                const frag bf = *reinterpret_cast<const frag*>(rhrow + ks * 32);
                RES_MFMA4(acc[0], acc[1], acc[2], acc[3], 64 + ks * 4, bf);
            }
            chain_done(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                const f32x4 xh = xval(n, 2);
                f32x4 hh, hnew;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hh[i] = tanh_f(acc[n][i] + xh[i]);
                    hnew[i] = zg[n][i] * hreg[n][i] + (1.0f - zg[n][i]) * hh[i];
                }
                if (acts && valid) {
                    bf16_t* ap = acts + ((size_t)t * B + b) * GH + ub[n];
                    st<bf16_t>::store4(ap, zg[n]);
                    st<bf16_t>::store4(ap + RH, rg[n]);
                    st<bf16_t>::store4(ap + 2 * RH, hh);
                }
                refill(n);
                hreg[n] = hnew;
                st<bf16_t>::store4(hnext + ub[n], hnew);
                if (hs && valid) st<bf16_t>::store4(hs + ((size_t)(t + 1) * B + b) * RH + ub[n], hnew);
            }
        }
        cur ^= 1;
        __syncthreads();
    }
    if (CELL == MVAE_GRU && a.h_last && valid) {
#pragma unroll
        for (int n = 0; n < RNT; ++n) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub[n]) = hreg[n];
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward through time
// ---------------------------------------------------------------------------------------------------------
template <int CELL, bool HAS_EXT, int NA, int NV, int NL>
__global__ __launch_bounds__(256, 1) void rnn_bwd_res_k(const mvae_rnn_bwd_args a) {
    constexpr int G = mvae_gates(CELL), GH = G * RH;
    constexpr int S2 = GH / 32;                    // k-groups of the backward contraction (over gate columns)
    constexpr int FPW = RNT * S2;
    constexpr int NLDS = NL;
    constexpr int LDA = GH + 8;
    static_assert(NA % 4 == 0 && NV % 4 == 0 && NL % 4 == 0 && NA + NV + NL <= FPW, "fragment classes");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* dabuf = reinterpret_cast<bf16_t*>(smem);                        // [16][LDA]
    frag* ulds = reinterpret_cast<frag*>(dabuf + 16 * LDA);                 // [4][NLDS][64]

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, r = l & 15;
    const int T = a.T, B = a.B;
    const int b = blockIdx.x * 16 + r;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.ut_pack);
    const bf16_t* __restrict__ hs = reinterpret_cast<const bf16_t*>(a.hs);
    const bf16_t* __restrict__ cs = reinterpret_cast<const bf16_t*>(a.cs);
    const bf16_t* __restrict__ acts = reinterpret_cast<const bf16_t*>(a.acts);
    const bf16_t* __restrict__ dext = reinterpret_cast<const bf16_t*>(a.dhs_ext);
    bf16_t* __restrict__ da = reinterpret_cast<bf16_t*>(a.da);
    bf16_t* __restrict__ rh = reinterpret_cast<bf16_t*>(a.rh);
    frag* myl = ulds + (size_t)w * NLDS * 64 + l;

    // fragment f = ks*4 + n  (A rows = hidden units of tile w*4+n, k-group ks over gate columns)
    auto frag_src = [&](int f) -> int { return (w * RNT + (f & 3)) * S2 + (f >> 2); };
    RES_DECLARE_U(FPW)

    int ub[RNT];
#pragma unroll
    for (int n = 0; n < RNT; ++n) ub[n] = w * 64 + n * 16 + q * 4;

    f32x4 dh[RNT], dc[RNT];
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        dh[n] = a.dh_last ? *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)bb * (a.dh_last_ld ? a.dh_last_ld : RH) + ub[n]) : z4;
        dc[n] = z4;
    }
    auto ld4 = [&](const bf16_t* base, size_t row, int width, int col) -> u16x4 {
        return *reinterpret_cast<const u16x4*>(base + row * width + col);
    };
    auto unpack = [](u16x4 p) -> f32x4 { return f32x4{bf2f(p[0]), bf2f(p[1]), bf2f(p[2]), bf2f(p[3])}; };

    // queue of saved forward values for the NEXT step to be processed (one step of prefetch distance)
    u16x4 qa[RNT][G], qs[RNT], qd[RNT], carry[RNT];   // gates; c_{t-1} (LSTM) or h_{t-1} (GRU); upstream grad; c_t
    {
        const int t = T - 1;
        const size_t row = (size_t)t * B + bb;
#pragma unroll
        for (int n = 0; n < RNT; ++n) {
#pragma unroll
            for (int g = 0; g < G; ++g) qa[n][g] = ld4(acts, row, GH, g * RH + ub[n]);
            if (CELL == MVAE_LSTM) {
                qs[n] = ld4(cs, row, RH, ub[n]);
                carry[n] = ld4(cs, (size_t)(t + 1) * B + bb, RH, ub[n]);
            }
            if (CELL == MVAE_GRU) qs[n] = ld4(hs, row, RH, ub[n]);
            if (HAS_EXT) qd[n] = ld4(dext, row, RH, ub[n]);
        }
    }
    bf16_t* drow = dabuf + r * LDA;
    const bf16_t* brow = dabuf + r * LDA + q * 8;

    for (int t = T - 1; t >= 0; --t) {
        const int tp = t > 0 ? t - 1 : 0;                    // the step prefetched during this one
        const size_t prow = (size_t)tp * B + bb;
        f32x4 acc[RNT];
#pragma unroll
        for (int n = 0; n < RNT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (CELL == MVAE_LSTM) {
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                const f32x4 ig = unpack(qa[n][0]), fg = unpack(qa[n][1]), gg = unpack(qa[n][2]), og = unpack(qa[n][G > 3 ? 3 : 0]);
                const f32x4 c = unpack(carry[n]), cp = unpack(qs[n]);
                f32x4 d = dh[n];
                if (HAS_EXT) d += unpack(qd[n]);
                f32x4 di, df, dg, dO;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float tc = tanh_f(c[i]);
                    const float dct = dc[n][i] + d[i] * og[i] * (1.0f - tc * tc);
                    di[i] = dct * gg[i] * dhard_sigmoid(ig[i]);
                    df[i] = dct * cp[i] * dhard_sigmoid(fg[i]);
                    dg[i] = dct * ig[i] * (1.0f - gg[i] * gg[i]);
                    dO[i] = d[i] * tc * dhard_sigmoid(og[i]);
                    dc[n][i] = dct * fg[i];
                }
                st<bf16_t>::store4(drow + ub[n], di);
                st<bf16_t>::store4(drow + RH + ub[n], df);
                st<bf16_t>::store4(drow + 2 * RH + ub[n], dg);
                st<bf16_t>::store4(drow + 3 * RH + ub[n], dO);
                if (valid) {
                    bf16_t* gp = da + ((size_t)t * B + b) * GH + ub[n];
                    st<bf16_t>::store4(gp, di);
                    st<bf16_t>::store4(gp + RH, df);
                    st<bf16_t>::store4(gp + 2 * RH, dg);
                    st<bf16_t>::store4(gp + 3 * RH, dO);
                }
                carry[n] = qs[n];                         // c_{t-1} is the next step's c_t
#pragma unroll
                for (int g = 0; g < G; ++g) qa[n][g] = ld4(acts, prow, GH, g * RH + ub[n]);
                qs[n] = ld4(cs, prow, RH, ub[n]);
                if (HAS_EXT) qd[n] = ld4(dext, prow, RH, ub[n]);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < S2; ++ks) {
                const frag bf = *reinterpret_cast<const frag*>(brow + ks * 32);
                RES_MFMA4(acc[0], acc[1], acc[2], acc[3], ks * 4, bf);
            }
            chain_done(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int n = 0; n < RNT; ++n) dh[n] = acc[n];
        } else if (CELL == MVAE_GRU) {
            f32x4 z[RNT], rr[RNT], hp[RNT], hh[RNT], d[RNT];
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                z[n] = unpack(qa[n][0]);
                rr[n] = unpack(qa[n][1]);
                hh[n] = unpack(qa[n][G > 2 ? 2 : 0]);
                hp[n] = unpack(qs[n]);
                d[n] = dh[n];
                if (HAS_EXT) d[n] += unpack(qd[n]);
                f32x4 dah;
#pragma unroll
                for (int i = 0; i < 4; ++i) dah[i] = d[n][i] * (1.0f - z[n][i]) * (1.0f - hh[n][i] * hh[n][i]);
                st<bf16_t>::store4(drow + 2 * RH + ub[n], dah);
                if (valid) {
                    st<bf16_t>::store4(da + ((size_t)t * B + b) * GH + 2 * RH + ub[n], dah);
                    if (rh) st<bf16_t>::store4(rh + ((size_t)t * B + b) * RH + ub[n], rr[n] * hp[n]);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) qa[n][g] = ld4(acts, prow, GH, g * RH + ub[n]);
                qs[n] = ld4(hs, prow, RH, ub[n]);
                if (HAS_EXT) qd[n] = ld4(dext, prow, RH, ub[n]);
            }
            __syncthreads();
            constexpr int SH = RH / 32;
#pragma unroll
            for (int ks = 2 * SH; ks < 3 * SH; ++ks) {
                const frag bf = *reinterpret_cast<const frag*>(brow + ks * 32);
                RES_MFMA4(acc[0], acc[1], acc[2], acc[3], ks * 4, bf);
            }
            chain_done(acc[0], acc[1], acc[2], acc[3]);
            f32x4 drh[RNT];
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                drh[n] = acc[n];
                acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 daz, dar;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    daz[i] = d[n][i] * (hp[n][i] - hh[n][i]) * dhard_sigmoid(z[n][i]);
                    dar[i] = drh[n][i] * hp[n][i] * dhard_sigmoid(rr[n][i]);
                }
                st<bf16_t>::store4(drow + ub[n], daz);
                st<bf16_t>::store4(drow + RH + ub[n], dar);
                if (valid) {
                    bf16_t* gp = da + ((size_t)t * B + b) * GH + ub[n];
                    st<bf16_t>::store4(gp, daz);
                    st<bf16_t>::store4(gp + RH, dar);
                }
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2 * SH; ++ks) {
                const frag bf = *reinterpret_cast<const frag*>(brow + ks * 32);
                RES_MFMA4(acc[0], acc[1], acc[2], acc[3], ks * 4, bf);
            }
            chain_done(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int n = 0; n < RNT; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[n][i] = d[n][i] * z[n][i] + drh[n][i] * rr[n][i] + acc[n][i];
        }
        __syncthreads();
    }
    if (valid) {
        const int ldd = a.dh0_ld ? a.dh0_ld : RH;
#pragma unroll
        for (int n = 0; n < RNT; ++n) {
            if (a.dh0) *reinterpret_cast<f32x4*>(a.dh0 + (size_t)b * ldd + ub[n]) = dh[n];
            if (CELL == MVAE_LSTM && a.dc0) *reinterpret_cast<f32x4*>(a.dc0 + (size_t)b * ldd + ub[n]) = dc[n];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------------------
// fragment placement per kernel: NA in accumulator registers, NV in vector registers, NL in LDS (per wave)
template <int CELL> struct res_cfg;
template <> struct res_cfg<MVAE_LSTM> {   // 128 fragments per wave
    static constexpr int FA = 60, FV = 32, FL = 32;     // + 4 streamed;  LDS 17 + 128 KiB
    static constexpr int BA = 60, BV = 28, BL = 28;     // + 12 streamed; LDS 33 + 112 KiB
};
template <> struct res_cfg<MVAE_GRU> {    // 96 fragments per wave
    static constexpr int FA = 60, FV = 8, FL = 28;      // LDS 25 + 112 KiB
    static constexpr int BA = 60, BV = 8, BL = 28;      // LDS 25 + 112 KiB
};

template <int CELL, int XMODE>
int launch_fwd_res(const mvae_rnn_fwd_args& a, hipStream_t s) {
    typedef res_cfg<CELL> C;
    const size_t lds = (size_t)(2 + (CELL == MVAE_GRU ? 1 : 0)) * 16 * RLDH * sizeof(bf16_t) +
                       (size_t)4 * C::FL * 64 * sizeof(frag) +
                       (XMODE == MVAE_X_SCALAR ? (size_t)2 * mvae_gates(CELL) * RH * sizeof(float) : 0);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_fwd_res_k<CELL, XMODE, C::FA, C::FV, C::FL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((rnn_fwd_res_k<CELL, XMODE, C::FA, C::FV, C::FL>), dim3((a.B + 15) / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <int CELL>
int fwd_res_xmode(const mvae_rnn_fwd_args& a, hipStream_t s) {
    switch (a.xmode) {
        case MVAE_X_DENSE: return a.xp ? launch_fwd_res<CELL, MVAE_X_DENSE>(a, s) : MVAE_E_ARG;
        case MVAE_X_INDEX: return (a.idx && a.table) ? launch_fwd_res<CELL, MVAE_X_INDEX>(a, s) : MVAE_E_ARG;
        case MVAE_X_SCALAR: return (a.xs && a.w_row && a.bias) ? launch_fwd_res<CELL, MVAE_X_SCALAR>(a, s) : MVAE_E_ARG;
        case MVAE_X_CONST: return a.xp0 ? launch_fwd_res<CELL, MVAE_X_CONST>(a, s) : MVAE_E_ARG;
    }
    return MVAE_E_ARG;
}

template <int CELL, bool HAS_EXT>
int launch_bwd_res(const mvae_rnn_bwd_args& a, hipStream_t s) {
    typedef res_cfg<CELL> C;
    constexpr int G = mvae_gates(CELL);
    const size_t lds = (size_t)16 * (G * RH + 8) * sizeof(bf16_t) + (size_t)4 * C::BL * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_bwd_res_k<CELL, HAS_EXT, C::BA, C::BV, C::BL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((rnn_bwd_res_k<CELL, HAS_EXT, C::BA, C::BV, C::BL>), dim3((a.B + 15) / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

}  // namespace

// Entry points used by rnn.hip's dispatch.  Return MVAE_E_UNSUPPORTED when the shape is not this file's.
int mvae_rnn_fwd_resident(const mvae_rnn_fwd_args& a, hipStream_t s) {
    if (a.H != RH || a.dtype != MVAE_BF16) return MVAE_E_UNSUPPORTED;
    if (a.cell == MVAE_LSTM) return fwd_res_xmode<MVAE_LSTM>(a, s);
    if (a.cell == MVAE_GRU) return fwd_res_xmode<MVAE_GRU>(a, s);
    return MVAE_E_UNSUPPORTED;
}
int mvae_rnn_bwd_resident(const mvae_rnn_bwd_args& a, hipStream_t s) {
    if (a.H != RH || a.dtype != MVAE_BF16) return MVAE_E_UNSUPPORTED;
    if (a.cell == MVAE_LSTM) {
        if (!a.cs) return MVAE_E_ARG;
        return a.dhs_ext ? launch_bwd_res<MVAE_LSTM, true>(a, s) : launch_bwd_res<MVAE_LSTM, false>(a, s);
    }
    if (a.cell == MVAE_GRU) return a.dhs_ext ? launch_bwd_res<MVAE_GRU, true>(a, s) : launch_bwd_res<MVAE_GRU, false>(a, s);
    return MVAE_E_UNSUPPORTED;
}
