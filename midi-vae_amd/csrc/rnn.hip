// Recurrent layers of the MIDI-VAE hot path on gfx950: forward recurrence, BPTT, weight packing.
//
// Replaces keras.layers.{GRU,LSTM,SimpleRNN} (reference vae_definition.py:448-480) and recurrentshop
// {GRU,LSTM,SimpleRNN}Cell stepped by RecurrentModel (reference vae_definition.py:533-546 etc.).
//
// Mapping to the machine (generic kernel, any H in {64,128,256}):
//   * the recurrence is independent across batch rows, so ONE workgroup (4 waves) owns 16 batch rows for all
//     T steps: no inter-workgroup synchronisation exists anywhere in this file;
//   * per step the gate pre-activations are  gates^T (G*H x 16) = U^T (G*H x H) * h^T (H x 16)  on MFMA with
//     the WEIGHTS as the A operand (16 gate columns per tile) and the 16 batch rows as the N dimension, so a
//     lane's 4 accumulator values are 4 CONSECUTIVE hidden units of one batch row: every global load / store
//     of xp, gates, h, c is an 8- or 16-byte vector access on row-major (T,B,*) arrays;
//   * wave w owns hidden units [w*H/4, (w+1)*H/4) for ALL gates, so the gate arithmetic, the cell state c and
//     the f32 master copy of h stay in that lane's registers for all T steps; only the MFMA-operand copy of
//     h_t goes through LDS (double buffered, one barrier per step; GRU needs a second for r*h);
//   * U is pre-packed in A-fragment order (mvae_pack_recurrent) so each wave-load of U is 1 KiB contiguous;
//     it is re-read from L2 every step here (the H=256 bf16 resident-U kernel lives in rnn_resident.hip).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

// ----------------------------------------------------------------------------------------------------------
// packing
// ----------------------------------------------------------------------------------------------------------
template <typename WT>
__global__ void pack_recurrent_k(const float* __restrict__ U, WT* __restrict__ out, int H, int GH, int direction) {
    pack_recurrent_body<WT>(U, out, H, GH, direction, blockIdx.x, gridDim.x);
}

// ----------------------------------------------------------------------------------------------------------
// weight streaming
// ----------------------------------------------------------------------------------------------------------
// acc[j] += A(s, j) * B(s) for NS k-groups x NJ tiles, the A fragments (1 KiB per wave access) streamed from L2 through a ring
// of D fragment registers: fragment i + D is requested when fragment i is consumed, so D loads are in flight per wave instead
// of the one or two hipcc keeps ahead in a rolled loop (round 4: f32 LSTM H=256 forward 52 -> 19.4, BPTT 33 -> 21 us
// per time step, profiles/r04_s_f32_generic.txt; the f32 MFMAs alone are 13.7).  Fully unrolled: every ring slot is a register.
template <typename WT, int NS, int NJ, int D, typename ADDR, typename BFRAG>
__device__ __forceinline__ void stream_mma(const void* wave_base, unsigned lane_off, ADDR addr, BFRAG bfrag, f32x4* acc) {
    // addr(s, j): fragment index relative to wave_base (a compile-time constant once unrolled).  Buffer loads: ONE lane-offset
    // register + a scalar offset per fragment - as global loads hipcc hoists the 256 loop-invariant 64-bit addresses of a step
    // out of the time loop and spills ~1 KiB per lane.
    typedef typename op<WT>::frag frag;
    static_assert(sizeof(frag) == 16, "one 16-byte access per lane and fragment");
    constexpr int F = NS * NJ, DD = D < F ? D : F;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wave_base), 0, -1, 0x00020000);
    auto fetch = [&](int i) {
        return __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, addr(i / NJ, i % NJ) * 1024, 0));
    };
    frag ring[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) ring[i] = fetch(i);
    frag bf, bnext = bfrag(0);
#pragma unroll
    for (int i = 0; i < F; ++i) {
        if (i % NJ == 0) {              // the B fragment of a k-group is read from LDS one group ahead
            bf = bnext;
            if (i / NJ + 1 < NS) bnext = bfrag(i / NJ + 1);
        }
        const frag a = ring[i % DD];
        if (i + DD < F) ring[i % DD] = fetch(i + DD);
        acc[i % NJ] = op<WT>::mma(a, bf, acc[i % NJ]);
        __builtin_amdgcn_sched_barrier(0);      // keeps request i + D in iteration i (unfenced, hipcc sinks every load to its use: depth 2)
    }
}
// the same with the ring carried from call to call (same fragment list every call: one stream_mma per time step): the first D
// requests of the NEXT call are issued by the last D iterations of this one
template <typename WT, int NS, int NJ, int D, typename ADDR, typename BFRAG>
__device__ __forceinline__ void stream_mma_carry(typename op<WT>::frag (&ring)[D], bool preload, const void* wave_base, unsigned lane_off,
                                                 ADDR addr, BFRAG bfrag, f32x4* acc) {
    typedef typename op<WT>::frag frag;
    constexpr int F = NS * NJ;
    static_assert(F % D == 0 && F >= D, "ring slots line up across the wrap");
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wave_base), 0, -1, 0x00020000);
    auto fetch = [&](int i) {
        return __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, addr(i / NJ, i % NJ) * 1024, 0));
    };
    if (preload) {
#pragma unroll
        for (int i = 0; i < D; ++i) ring[i] = fetch(i);
    }
    frag bf, bnext = bfrag(0);
#pragma unroll
    for (int i = 0; i < F; ++i) {
        if (i % NJ == 0) {
            bf = bnext;
            if (i / NJ + 1 < NS) bnext = bfrag(i / NJ + 1);
        }
        const frag a = ring[i % D];
        ring[i % D] = fetch((i + D) % F);
        acc[i % NJ] = op<WT>::mma(a, bf, acc[i % NJ]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
#ifndef MVAE_STREAM_DEPTH
#define MVAE_STREAM_DEPTH 16
#endif
#ifndef MVAE_STREAM_CARRY
#define MVAE_STREAM_CARRY 1      /* f32 LSTM H=256 forward alone: inference 18.3 -> 16.2 us per time step, training 18.7 -> 18.4 (stores in flight) */
#endif
constexpr int STREAM_DEPTH = MVAE_STREAM_DEPTH;

// ----------------------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------------------
template <typename WT> struct lds_pad { static constexpr int value = 16 / sizeof(WT); };

// NW waves per workgroup, NT unit tiles per wave (H = 16 NT NW).  NW = 8 for the f32 H = 256 case: a wave cannot issue its next
// MFMA while one of its 256 fragment loads per step sits in the memory pipe's issue stage (~64 cycles each); with two waves per
// SIMD the other one's MFMAs fill the pipe meanwhile.
template <int CELL, typename WT, int XMODE, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void rnn_fwd_k(const mvae_rnn_fwd_args a) {
    constexpr int G = mvae_gates(CELL);
    constexpr int H = NT * 16 * NW, GH = G * H;
    constexpr int KG = op<WT>::KG, FE = op<WT>::FRAG_ELEMS, S = H / KG;
    constexpr int LDH = H + lds_pad<WT>::value;
    typedef typename op<WT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WT* hbuf = reinterpret_cast<WT*>(smem);                       // [2][16][LDH]
    WT* rhbuf = hbuf + 2 * 16 * LDH;                              // [16][LDH]       (GRU only)
    float* wb = reinterpret_cast<float*>(rhbuf + (CELL == MVAE_GRU ? 16 * LDH : 0));   // [2][GH] (SCALAR only)

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, r = l & 15;
    const int T = a.T, B = a.B;
    const int b = blockIdx.x * 16 + r;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;
    // this wave's fragments of a gate: tiles [w NT, w NT + NT), 1 KiB each (the wave index as a scalar: a uniform base address)
    const frag* wave_u = reinterpret_cast<const frag*>(a.u_pack) + (size_t)__builtin_amdgcn_readfirstlane(w) * NT * S * 64;
    WT* __restrict__ hs = reinterpret_cast<WT*>(a.hs);
    WT* __restrict__ cs = reinterpret_cast<WT*>(a.cs);
    WT* __restrict__ acts = reinterpret_cast<WT*>(a.acts);
    const WT* __restrict__ xp = reinterpret_cast<const WT*>(a.xp);

    int ub[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) ub[n] = w * (H / NW) + n * 16 + q * 4;
    const int ld0 = a.h0_ld ? a.h0_ld : H, ldl = a.h_last_ld ? a.h_last_ld : H;

    // ---- initial state -------------------------------------------------------------------------------
    f32x4 hreg[NT], creg[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        hreg[n] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + (size_t)bb * ld0 + ub[n]) : z4;
        creg[n] = (CELL == MVAE_LSTM && a.c0) ? *reinterpret_cast<const f32x4*>(a.c0 + (size_t)bb * ld0 + ub[n]) : z4;
        st<WT>::store4(hbuf + r * LDH + ub[n], hreg[n]);
        if (valid) {
            if (hs) st<WT>::store4(hs + (size_t)b * H + ub[n], hreg[n]);
            if (CELL == MVAE_LSTM && cs) st<WT>::store4(cs + (size_t)b * H + ub[n], creg[n]);
        }
    }
    if (XMODE == MVAE_X_SCALAR) {
        for (int i = tid; i < GH; i += NW * 64) {
            wb[i] = a.w_row[i];
            wb[GH + i] = a.bias[i];
        }
    }
    lds_barrier();

    int cur = 0;
    frag carry[STREAM_DEPTH];
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B + bb;
        // ---- x_t W + b for this lane's (gate, unit) positions; consumed after the MFMAs ----------------
        f32x4 xv[G][NT];
        if (XMODE == MVAE_X_DENSE) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int n = 0; n < NT; ++n) xv[g][n] = st<WT>::load4(xp + row * GH + g * H + ub[n]);
        } else if (XMODE == MVAE_X_INDEX) {
            const WT* trow = reinterpret_cast<const WT*>(a.table) + (size_t)a.idx[row] * GH;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int n = 0; n < NT; ++n) xv[g][n] = st<WT>::load4(trow + g * H + ub[n]);
        } else if (XMODE == MVAE_X_SCALAR) {
            const float x = a.xs[row];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wb + g * H + ub[n]);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(wb + GH + g * H + ub[n]);
                    xv[g][n] = x * w4 + b4;
                }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    xv[g][n] = st<WT>::load4(reinterpret_cast<const WT*>(a.xp0) + (size_t)bb * GH + g * H + ub[n]);
        }

        // ---- h_{t-1} U on the matrix cores -----------------------------------------------------------
        constexpr int GA = (CELL == MVAE_GRU) ? 2 : G;   // gates whose recurrent input is h itself
        f32x4 acc[G][NT];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const WT* hrow = hbuf + cur * 16 * LDH + r * LDH + q * FE;
        if constexpr (MVAE_STREAM_CARRY && CELL != MVAE_GRU && (S * GA * NT) % STREAM_DEPTH == 0)
            stream_mma_carry<WT, S, GA * NT, STREAM_DEPTH>(
                carry, t == 0, wave_u, l * 16u, [](int s, int j) { return ((j / NT) * (H / 16) + j % NT) * S + s; },
                [&](int s) { return *reinterpret_cast<const frag*>(hrow + s * KG); }, &acc[0][0]);
        else
        stream_mma<WT, S, GA * NT, STREAM_DEPTH>(
            wave_u, l * 16u, [](int s, int j) { return ((j / NT) * (H / 16) + j % NT) * S + s; },
            [&](int s) { return *reinterpret_cast<const frag*>(hrow + s * KG); }, &acc[0][0]);

        f32x4 hnew[NT];
        if (CELL == MVAE_GRU) {
            f32x4 z[NT], rr[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    z[n][i] = hard_sigmoid(acc[0][n][i] + xv[0][n][i]);
                    rr[n][i] = hard_sigmoid(acc[1][n][i] + xv[1][n][i]);
                }
                st<WT>::store4(rhbuf + r * LDH + ub[n], rr[n] * hreg[n]);
            }
            lds_barrier();
            const WT* rhrow = rhbuf + r * LDH + q * FE;
            stream_mma<WT, S, NT, STREAM_DEPTH>(
                wave_u, l * 16u, [](int s, int n) { return (2 * (H / 16) + n) * S + s; },
                [&](int s) { return *reinterpret_cast<const frag*>(rhrow + s * KG); }, &acc[2][0]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 hh;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hh[i] = tanh_f(acc[2][n][i] + xv[2][n][i]);
                    hnew[n][i] = z[n][i] * hreg[n][i] + (1.0f - z[n][i]) * hh[i];
                }
                if (acts && valid) {
                    WT* ap = acts + ((size_t)t * B + b) * GH + ub[n];
                    st<WT>::store4(ap, z[n]);
                    st<WT>::store4(ap + H, rr[n]);
                    st<WT>::store4(ap + 2 * H, hh);
                }
            }
        } else if (CELL == MVAE_LSTM) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 ig, fg, gg, og;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ig[i] = hard_sigmoid(acc[0][n][i] + xv[0][n][i]);
                    fg[i] = hard_sigmoid(acc[1][n][i] + xv[1][n][i]);
                    gg[i] = tanh_f(acc[2][n][i] + xv[2][n][i]);
                    og[i] = hard_sigmoid(acc[3][n][i] + xv[3][n][i]);
                    creg[n][i] = fg[i] * creg[n][i] + ig[i] * gg[i];
                    hnew[n][i] = og[i] * tanh_f(creg[n][i]);
                }
                if (valid) {
                    if (acts) {
                        WT* ap = acts + ((size_t)t * B + b) * GH + ub[n];
                        st<WT>::store4(ap, ig);
                        st<WT>::store4(ap + H, fg);
                        st<WT>::store4(ap + 2 * H, gg);
                        st<WT>::store4(ap + 3 * H, og);
                    }
                    if (cs) st<WT>::store4(cs + ((size_t)(t + 1) * B + b) * H + ub[n], creg[n]);
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hnew[n][i] = tanh_f(acc[0][n][i] + xv[0][n][i]);
                if (acts && valid) st<WT>::store4(acts + ((size_t)t * B + b) * GH + ub[n], hnew[n]);
            }
        }

        // ---- publish h_t: MFMA-operand copy to LDS, sequence copy to HBM ---------------------------------
        WT* hnext = hbuf + (cur ^ 1) * 16 * LDH + r * LDH;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            hreg[n] = hnew[n];
            st<WT>::store4(hnext + ub[n], hnew[n]);
            if (hs && valid) st<WT>::store4(hs + ((size_t)(t + 1) * B + b) * H + ub[n], hnew[n]);
        }
        cur ^= 1;
        lds_barrier();
    }
    if (valid) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (a.h_last) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub[n]) = hreg[n];
            if (CELL == MVAE_LSTM && a.c_last) *reinterpret_cast<f32x4*>(a.c_last + (size_t)b * ldl + ub[n]) = creg[n];
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// backward through time
// ----------------------------------------------------------------------------------------------------------
template <int CELL, typename WT, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void rnn_bwd_k(const mvae_rnn_bwd_args a) {
    constexpr int G = mvae_gates(CELL);
    constexpr int H = NT * 16 * NW, GH = G * H;
    constexpr int KG = op<WT>::KG, FE = op<WT>::FRAG_ELEMS, S2 = GH / KG;
    constexpr int LDA = GH + lds_pad<WT>::value;
    typedef typename op<WT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WT* dabuf = reinterpret_cast<WT*>(smem);                      // [16][LDA]

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, q = l >> 4, r = l & 15;
    const int T = a.T, B = a.B;
    const int b = blockIdx.x * 16 + r;
    const bool valid = b < B;
    const int bb = valid ? b : B - 1;
    const frag* wave_u = reinterpret_cast<const frag*>(a.ut_pack) + (size_t)__builtin_amdgcn_readfirstlane(w) * NT * S2 * 64;
    const WT* __restrict__ hs = reinterpret_cast<const WT*>(a.hs);
    const WT* __restrict__ cs = reinterpret_cast<const WT*>(a.cs);
    const WT* __restrict__ acts = reinterpret_cast<const WT*>(a.acts);
    const WT* __restrict__ dext = reinterpret_cast<const WT*>(a.dhs_ext);
    WT* __restrict__ da = reinterpret_cast<WT*>(a.da);
    WT* __restrict__ rh = reinterpret_cast<WT*>(a.rh);

    int ub[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) ub[n] = w * (H / NW) + n * 16 + q * 4;

    f32x4 dh[NT], dc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        dh[n] = a.dh_last ? *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)bb * (a.dh_last_ld ? a.dh_last_ld : H) + ub[n]) : z4;
        dc[n] = (CELL == MVAE_LSTM && a.dc_last)
                    ? *reinterpret_cast<const f32x4*>(a.dc_last + (size_t)bb * (a.dh_last_ld ? a.dh_last_ld : H) + ub[n]) : z4;
    }
    WT* drow = dabuf + r * LDA;
    const WT* brow = dabuf + r * LDA + q * FE;

    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)t * B + bb;
        const WT* ap = acts + row * GH;
        f32x4 d[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            d[n] = dh[n];
            if (dext) d[n] += st<WT>::load4(dext + row * H + ub[n]);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // acc[n] += U^T fragments of k-groups [S0, S0 + NS) x the da tile in LDS
        auto stream_dh = [&](auto s0, auto ns) {
            constexpr int S0 = decltype(s0)::value, NS = decltype(ns)::value;
            stream_mma<WT, NS, NT, STREAM_DEPTH>(
                wave_u, l * 16u, [](int s, int n) { return n * S2 + S0 + s; },
                [&](int s) { return *reinterpret_cast<const frag*>(brow + (S0 + s) * KG); }, acc);
        };

        if (CELL == MVAE_LSTM) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 ig = st<WT>::load4(ap + ub[n]), fg = st<WT>::load4(ap + H + ub[n]);
                const f32x4 gg = st<WT>::load4(ap + 2 * H + ub[n]), og = st<WT>::load4(ap + 3 * H + ub[n]);
                const f32x4 c = st<WT>::load4(cs + ((size_t)(t + 1) * B + bb) * H + ub[n]);
                const f32x4 cp = st<WT>::load4(cs + row * H + ub[n]);
                f32x4 di, df, dg, dO;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float tc = tanh_f(c[i]);
                    const float dct = dc[n][i] + d[n][i] * og[i] * (1.0f - tc * tc);
                    di[i] = dct * gg[i] * dhard_sigmoid(ig[i]);
                    df[i] = dct * cp[i] * dhard_sigmoid(fg[i]);
                    dg[i] = dct * ig[i] * (1.0f - gg[i] * gg[i]);
                    dO[i] = d[n][i] * tc * dhard_sigmoid(og[i]);
                    dc[n][i] = dct * fg[i];
                }
                st<WT>::store4(drow + ub[n], di);
                st<WT>::store4(drow + H + ub[n], df);
                st<WT>::store4(drow + 2 * H + ub[n], dg);
                st<WT>::store4(drow + 3 * H + ub[n], dO);
                if (valid) {
                    WT* gp = da + ((size_t)t * B + b) * GH + ub[n];
                    st<WT>::store4(gp, di);
                    st<WT>::store4(gp + H, df);
                    st<WT>::store4(gp + 2 * H, dg);
                    st<WT>::store4(gp + 3 * H, dO);
                }
            }
            lds_barrier();
            stream_dh(std::integral_constant<int, 0>{}, std::integral_constant<int, S2>{});
#pragma unroll
            for (int n = 0; n < NT; ++n) dh[n] = acc[n];
        } else if (CELL == MVAE_GRU) {
            f32x4 z[NT], rr[NT], hp[NT], hh[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                z[n] = st<WT>::load4(ap + ub[n]);
                rr[n] = st<WT>::load4(ap + H + ub[n]);
                hh[n] = st<WT>::load4(ap + 2 * H + ub[n]);
                hp[n] = st<WT>::load4(hs + row * H + ub[n]);
                f32x4 dah;
#pragma unroll
                for (int i = 0; i < 4; ++i) dah[i] = d[n][i] * (1.0f - z[n][i]) * (1.0f - hh[n][i] * hh[n][i]);
                st<WT>::store4(drow + 2 * H + ub[n], dah);
                if (valid) {
                    st<WT>::store4(da + ((size_t)t * B + b) * GH + 2 * H + ub[n], dah);
                    if (rh) st<WT>::store4(rh + ((size_t)t * B + b) * H + ub[n], rr[n] * hp[n]);
                }
            }
            lds_barrier();
            constexpr int SH = H / KG;
            stream_dh(std::integral_constant<int, 2 * SH>{}, std::integral_constant<int, SH>{});
            f32x4 drh[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                drh[n] = acc[n];
                acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 daz, dar;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    daz[i] = d[n][i] * (hp[n][i] - hh[n][i]) * dhard_sigmoid(z[n][i]);
                    dar[i] = drh[n][i] * hp[n][i] * dhard_sigmoid(rr[n][i]);
                }
                st<WT>::store4(drow + ub[n], daz);
                st<WT>::store4(drow + H + ub[n], dar);
                if (valid) {
                    WT* gp = da + ((size_t)t * B + b) * GH + ub[n];
                    st<WT>::store4(gp, daz);
                    st<WT>::store4(gp + H, dar);
                }
            }
            lds_barrier();
            stream_dh(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * SH>{});
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[n][i] = d[n][i] * z[n][i] + drh[n][i] * rr[n][i] + acc[n][i];
        } else {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 y = st<WT>::load4(ap + ub[n]);
                f32x4 dy;
#pragma unroll
                for (int i = 0; i < 4; ++i) dy[i] = d[n][i] * (1.0f - y[i] * y[i]);
                st<WT>::store4(drow + ub[n], dy);
                if (valid) st<WT>::store4(da + ((size_t)t * B + b) * GH + ub[n], dy);
            }
            lds_barrier();
            stream_dh(std::integral_constant<int, 0>{}, std::integral_constant<int, S2>{});
#pragma unroll
            for (int n = 0; n < NT; ++n) dh[n] = acc[n];
        }
        lds_barrier();   // all waves done reading dabuf before the next step overwrites it
    }
    if (valid) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ldd = a.dh0_ld ? a.dh0_ld : H;
            if (a.dh0) *reinterpret_cast<f32x4*>(a.dh0 + (size_t)b * ldd + ub[n]) = dh[n];
            if (CELL == MVAE_LSTM && a.dc0) *reinterpret_cast<f32x4*>(a.dc0 + (size_t)b * ldd + ub[n]) = dc[n];
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// dispatch
// ----------------------------------------------------------------------------------------------------------
template <int CELL, typename WT, int XMODE, int NT, int NW = 4>
int launch_fwd(const mvae_rnn_fwd_args& a, hipStream_t s) {
    constexpr int G = mvae_gates(CELL), H = NT * 16 * NW;
    constexpr int LDH = H + lds_pad<WT>::value;
    size_t lds = (size_t)(2 + (CELL == MVAE_GRU ? 1 : 0)) * 16 * LDH * sizeof(WT);
    if (XMODE == MVAE_X_SCALAR) lds += (size_t)2 * G * H * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised = false;   // gfx950 has 160 KiB of LDS per CU; anything above 64 KiB must be requested
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_fwd_k<CELL, WT, XMODE, NT, NW>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return MVAE_E_LAUNCH;
            raised = true;
        }
    }
    hipLaunchKernelGGL((rnn_fwd_k<CELL, WT, XMODE, NT, NW>), dim3((a.B + 15) / 16), dim3(NW * 64), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <int CELL, typename WT, int XMODE>
int fwd_nt(const mvae_rnn_fwd_args& a, hipStream_t s) {
    switch (a.H) {
        case 64: return launch_fwd<CELL, WT, XMODE, 1>(a, s);
        case 128: return launch_fwd<CELL, WT, XMODE, 2>(a, s);
        case 256:
            if constexpr (sizeof(WT) == 4) return launch_fwd<CELL, WT, XMODE, 2, 8>(a, s);
            else return launch_fwd<CELL, WT, XMODE, 4>(a, s);
    }
    return MVAE_E_UNSUPPORTED;
}
template <int CELL, typename WT>
int fwd_xmode(const mvae_rnn_fwd_args& a, hipStream_t s) {
    switch (a.xmode) {
        case MVAE_X_DENSE: return a.xp ? fwd_nt<CELL, WT, MVAE_X_DENSE>(a, s) : MVAE_E_ARG;
        case MVAE_X_INDEX: return (a.idx && a.table) ? fwd_nt<CELL, WT, MVAE_X_INDEX>(a, s) : MVAE_E_ARG;
        case MVAE_X_SCALAR: return (a.xs && a.w_row && a.bias) ? fwd_nt<CELL, WT, MVAE_X_SCALAR>(a, s) : MVAE_E_ARG;
        case MVAE_X_CONST: return a.xp0 ? fwd_nt<CELL, WT, MVAE_X_CONST>(a, s) : MVAE_E_ARG;
    }
    return MVAE_E_ARG;
}
template <typename WT>
int fwd_cell(const mvae_rnn_fwd_args& a, hipStream_t s) {
    switch (a.cell) {
        case MVAE_GRU: return fwd_xmode<MVAE_GRU, WT>(a, s);
        case MVAE_LSTM: return fwd_xmode<MVAE_LSTM, WT>(a, s);
        case MVAE_RNN: return fwd_xmode<MVAE_RNN, WT>(a, s);
    }
    return MVAE_E_ARG;
}

template <int CELL, typename WT, int NT, int NW = 4>
int launch_bwd(const mvae_rnn_bwd_args& a, hipStream_t s) {
    constexpr int GH = mvae_gates(CELL) * NT * 16 * NW;
    const size_t lds = (size_t)16 * (GH + lds_pad<WT>::value) * sizeof(WT);
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_bwd_k<CELL, WT, NT, NW>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return MVAE_E_LAUNCH;
            raised = true;
        }
    }
    hipLaunchKernelGGL((rnn_bwd_k<CELL, WT, NT, NW>), dim3((a.B + 15) / 16), dim3(NW * 64), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <int CELL, typename WT>
int bwd_nt(const mvae_rnn_bwd_args& a, hipStream_t s) {
    switch (a.H) {
        case 64: return launch_bwd<CELL, WT, 1>(a, s);
        case 128: return launch_bwd<CELL, WT, 2>(a, s);
        case 256:
            if constexpr (sizeof(WT) == 4) return launch_bwd<CELL, WT, 2, 8>(a, s);
            else return launch_bwd<CELL, WT, 4>(a, s);
    }
    return MVAE_E_UNSUPPORTED;
}
template <typename WT>
int bwd_cell(const mvae_rnn_bwd_args& a, hipStream_t s) {
    switch (a.cell) {
        case MVAE_GRU: return bwd_nt<MVAE_GRU, WT>(a, s);
        case MVAE_LSTM: return a.cs ? bwd_nt<MVAE_LSTM, WT>(a, s) : MVAE_E_ARG;
        case MVAE_RNN: return bwd_nt<MVAE_RNN, WT>(a, s);
    }
    return MVAE_E_ARG;
}

}  // namespace

// rnn_resident.hip: H = 256 / bf16 kernels with the recurrent weights resident in registers + LDS
int mvae_rnn_fwd_resident(const mvae_rnn_fwd_args& a, hipStream_t s);
int mvae_rnn_bwd_resident(const mvae_rnn_bwd_args& a, hipStream_t s);

extern "C" int mvae_rnn_fwd(const mvae_rnn_fwd_args* a, void* stream) {
    if (!a || !a->u_pack || a->T <= 0 || a->B <= 0) return MVAE_E_ARG;
    // time-pipelined stacks: only the slot-interleaved kernels (seq_layout TILE16P) poll / publish
    if (a->chunk_steps < 0 || ((a->wait_ready || a->signal_done) && a->chunk_steps == 0)) return MVAE_E_ARG;
    if ((a->wait_ready || a->signal_done) && a->seq_layout != MVAE_TILE16P && a->seq_layout != MVAE_TILE16Q) return MVAE_E_UNSUPPORTED;
    if (a->wait_ready && a->xmode != MVAE_X_DENSE) return MVAE_E_ARG;
    if (a->signal_done && !a->hs) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->H == 256 && a->dtype == MVAE_BF16 && a->cell != MVAE_RNN) {
        const int rc = mvae_rnn_fwd_resident(*a, s);
        if (rc != MVAE_E_UNSUPPORTED) return rc;
    }
    if (a->seq_layout != MVAE_ROWMAJOR) return MVAE_E_UNSUPPORTED;   // the generic kernels are row-major only
    if (a->xmode == MVAE_X_INDEX && a->table_layout != MVAE_TABLE_ROWMAJOR) return MVAE_E_ARG;     // ... and so are their lookup tables
    if (a->dtype == MVAE_F32) return fwd_cell<float>(*a, s);
    if (a->dtype == MVAE_BF16) return fwd_cell<bf16_t>(*a, s);
    return MVAE_E_ARG;
}

extern "C" int mvae_rnn_bwd(const mvae_rnn_bwd_args* a, void* stream) {
    if (!a || !a->ut_pack || !a->hs || !a->acts || !a->da || a->T <= 0 || a->B <= 0) return MVAE_E_ARG;
    if (a->chunk_steps < 0 || ((a->wait_ready || a->signal_done) && a->chunk_steps == 0)) return MVAE_E_ARG;
    if ((a->wait_ready || a->signal_done) && a->seq_layout != MVAE_TILE16P && a->seq_layout != MVAE_TILE16Q) return MVAE_E_UNSUPPORTED;
    if (a->wait_ready && !a->dhs_ext) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->H == 256 && a->dtype == MVAE_BF16 && a->cell != MVAE_RNN) {
        const int rc = mvae_rnn_bwd_resident(*a, s);
        if (rc != MVAE_E_UNSUPPORTED) return rc;
    }
    if (a->seq_layout != MVAE_ROWMAJOR) return MVAE_E_UNSUPPORTED;
    if (a->dtype == MVAE_F32) return bwd_cell<float>(*a, s);
    if (a->dtype == MVAE_BF16) return bwd_cell<bf16_t>(*a, s);
    return MVAE_E_ARG;
}

extern "C" int mvae_pack_recurrent(const float* U, void* out, int32_t cell, int32_t H, int32_t dtype,
                                   int32_t direction, void* stream) {
    if (!U || !out || (H % 64) != 0 || direction < 0 || direction > 1) return MVAE_E_ARG;
    if (cell != MVAE_GRU && cell != MVAE_LSTM && cell != MVAE_RNN) return MVAE_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int GH = mvae_gates(cell) * H;
    const size_t total = (size_t)GH * H;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (dtype == MVAE_F32)
        hipLaunchKernelGGL(pack_recurrent_k<float>, dim3(blocks), dim3(256), 0, s, U, (float*)out, H, GH, direction);
    else if (dtype == MVAE_BF16)
        hipLaunchKernelGGL(pack_recurrent_k<bf16_t>, dim3(blocks), dim3(256), 0, s, U, (bf16_t*)out, H, GH, direction);
    else
        return MVAE_E_ARG;
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
