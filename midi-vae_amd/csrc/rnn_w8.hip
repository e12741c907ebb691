// Resident-weights recurrent kernels, TWO WAVES PER SIMD (round 6): H = 256, bf16 MFMA operands, gfx950.
//
// Same contract, layouts and numerics as the slot-interleaved kernels of rnn_resident.hip (seq_layout MVAE_TILE16P);
// different occupancy.  There one wave per SIMD owns 64 hidden units and 512 registers: whatever that wave cannot issue
// under its own MFMAs (a 16x16x32 MFMA holds the matrix pipe 16 cycles, ~2 other instructions fit underneath) is dead
// time, and every dependency stall - MFMA result -> gate arithmetic, ds_read -> MFMA, the barrier - stalls the SIMD
// (round-5 counters: issue active 51 % of the cycles).  Here a workgroup is 8 waves of 256 registers: wave w owns the 32
// hidden units [32w, 32w+32) for all gates, so two waves share each SIMD's matrix pipe and the hardware issues one wave's
// gate arithmetic, LDS and memory instructions under the other's MFMAs.
//
// What the split costs: every wave needs ALL of h (and r*h) as MFMA B operand, so the workgroup reads its LDS tiles
// twice as often (16 ds_read_b128 per wave and step), and a wave has 256 registers for 48 weight fragments (192
// registers) plus its state: NLDS of the z / r fragments live in a private LDS slab (read back one per MFMA slot).
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

// Development switches (variant builds only: tools/build_w8_variants.sh; every W8_ABL_* makes the results WRONG)
#ifndef W8_ABL_NOBAR
#define W8_ABL_NOBAR 0      /* no barriers */
#endif
#ifndef W8_ABL_NOMATH
#define W8_ABL_NOMATH 0     /* no gate arithmetic */
#endif
#ifndef W8_ABL_NOL
#define W8_ABL_NOL 0        /* LDS-resident weight fragments replaced by register ones */
#endif
#ifndef W8_ABL_NOB
#define W8_ABL_NOB 0        /* B fragments read once per launch */
#endif
#ifndef W8_PRIO
#define W8_PRIO 0           /* s_setprio 1 for waves 4..7 (the second-dispatched wave of every SIMD) */
#endif
#ifndef MVAE_VARIANT_BUILD
static_assert(!W8_ABL_NOBAR && !W8_ABL_NOMATH && !W8_ABL_NOL && !W8_ABL_NOB, "timing ablations: variant builds only");
#endif

namespace {

constexpr int RH = 256;
typedef u16x8 frag;
typedef __attribute__((address_space(1))) unsigned char gbyte;
typedef __attribute__((address_space(1))) u16x4 g_u16x4;
typedef __attribute__((address_space(1))) u16x8 g_u16x8;
enum { SAVE_NONE = 0, SAVE_HS = 1, SAVE_ALL = 2 };

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
#define SF_LAMBDA(ic) [&](auto ic) __attribute__((always_inline))
template <bool AG>
__device__ __forceinline__ void mfma1(f32x4& c, const frag& u, const frag& b) {
    if (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(u), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(u), "v"(b));
}
__device__ __forceinline__ void load1_agpr_nowait(frag& u, const frag* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&a"(u) : "v"(p) : "memory");
}
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pinu(unsigned& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pini(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin1(u16x4& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pins(gbyte*& p) { asm volatile("" : "+s"(p)); }
__device__ __forceinline__ gbyte* to_global(const void* p) { return (gbyte*)(const_cast<void*>(p)); }
__device__ __forceinline__ void store16_wt(gbyte* uniform_base, unsigned lane_off, u16x8 v) {
    ::store16_wt((const void*)uniform_base, lane_off, v);
}
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ f32x4 unpack4(u16x4 p) { return f32x4{bf2f(p[0]), bf2f(p[1]), bf2f(p[2]), bf2f(p[3])}; }
__device__ __forceinline__ u16x4 pack4(f32x4 v) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4));
}
__device__ __forceinline__ u16x8 cat8(u16x4 a, u16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ void w8_barrier() {
    if (W8_ABL_NOBAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else lds_barrier();
}
__device__ __forceinline__ float hsig(float x) { return __builtin_amdgcn_fmed3f(__builtin_fmaf(0.2f, x, 0.5f), 0.0f, 1.0f); }

// ---------------------------------------------------------------------------------------------------------
// GRU forward (Keras 2.0.x GRU: reset gate applied BEFORE the candidate matmul; gate order [z | r | candidate])
// ---------------------------------------------------------------------------------------------------------
// Step t, per wave (tiles n = 0, 1 of 16 units each):
//   A  32 MFMA slots, B = h_{t-1} (all 8 k-groups read into registers behind the barrier): r over k-groups 0..3, then r
//      and z over k-groups 4..7, then z over k-groups 0..3 - r is complete 8 slots before the phase ends and its
//      arithmetic (hard_sigmoid, r*h -> rh tile) runs under the z slots of this wave and of its SIMD partner
//   -  barrier (the candidate needs every wave's r*h)
//   C  16 MFMA slots, B = r*h: tile 0 then tile 1; z's arithmetic and tile 0's tanh + h update run underneath
//   -  tile 1's tanh + h update, h -> LDS, barrier
// Phase-A slot s: gate, tile, k-group
struct slot_a { int g, n, ks; };
__host__ __device__ constexpr slot_a gru_slot_a(int s) {
    if (s < 8) return {1, s & 1, s >> 1};
    if (s < 24) return {((s - 8) >> 1) & 1 ? 0 : 1, s & 1, 4 + ((s - 8) >> 2)};
    return {0, s & 1, (s - 24) >> 1};
}
// NL of the 32 phase-A fragments live in LDS, spread evenly over the slots; the others in accumulator registers
__host__ __device__ constexpr bool gru_a_is_l(int s, int NL) { return ((s + 1) * NL) / 32 != (s * NL) / 32; }
__host__ __device__ constexpr int gru_a_lidx(int s, int NL) { return (s * NL) / 32; }

template <int XMODE, int SAVE, int NLDS>
__device__ __forceinline__ void gru_fwd_w8_body(const mvae_rnn_fwd_args& a, const unsigned bx) {
    constexpr int G = 3, GH = G * RH, NAA = 32 - NLDS;
    static_assert(XMODE != MVAE_X_SCALAR, "scalar inputs run on the phased kernel");
    static_assert(NLDS >= 0 && NLDS <= 16, "LDS-resident fragments");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* hbuf = smem;                                             // [2][16][RH] bf16, swizzled
    unsigned char* rhbuf = smem + 2 * 16 * RH * 2;                          // [16][RH]
    frag* ulds = reinterpret_cast<frag*>(smem + 3 * 16 * RH * 2);           // [8][NLDS][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                 // 0..7: units [32w, 32w + 32)
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.u_pack);
    frag* myl = ulds + (size_t)w * NLDS * 64 + l;
    auto src_frag = [&](int g, int n, int ks) -> const frag* {
        return up + (size_t)((g * (RH / 16) + w * 2 + n) * 8 + ks) * 64 + l;
    };
    frag ua[NAA > 0 ? NAA : 1], uc[16];
    static_for<0, 32>(SF_LAMBDA(sc) {
        constexpr int s = decltype(sc)::value;
        constexpr slot_a sa = gru_slot_a(s);
        if constexpr (gru_a_is_l(s, NLDS)) myl[(size_t)gru_a_lidx(s, NLDS) * 64] = *src_frag(sa.g, sa.n, sa.ks);
        else load1_agpr_nowait(ua[s - gru_a_lidx(s, NLDS)], src_frag(sa.g, sa.n, sa.ks));
    });
    static_for<0, 16>(SF_LAMBDA(sc) {
        constexpr int s = decltype(sc)::value;
        load1_agpr_nowait(uc[s], src_frag(2, s >> 3, s & 7));
    });

    const int ld0 = a.h0_ld ? a.h0_ld : RH, ldl = a.h_last_ld ? a.h_last_ld : RH;
    unsigned lane8 = (unsigned)l * 8u, lane16 = (unsigned)l * 16u;
    const int ub0 = w * 32 + q * 4;
    unsigned hw0 = (unsigned)r * 512u + ((((unsigned)w * 4u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    unsigned bf4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf4[j] = (unsigned)r * 512u + ((((unsigned)j * 4u + (unsigned)q) ^ (unsigned)r) << 4);
    // row-major copy of h_{t-1} (hs slot t): waves 4..7 (one per SIMD), two 16-byte chunks per lane - and only they publish,
    // so a consumer still counts 4 increments per workgroup and chunk
    const bool copier = w >= 4;
    const unsigned row0 = 4u * ((unsigned)w & 3u) + ((unsigned)l >> 5), ch0 = (unsigned)l & 31u;
    unsigned tl0 = row0 * 512u + ((ch0 ^ row0) << 4);
    unsigned tg0 = row0 * 512u + ch0 * 16u;

    f32x4 hreg[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        hreg[n] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * ld0 + ub0 + 16 * n) : z4;
        *reinterpret_cast<u16x4*>(hbuf + (hw0 ^ (n << 5))) = pack4(hreg[n]);
    }
    // ---- x queue: requested TWO steps ahead into alternating buffers (the step loop is unrolled by two) ----------------
    u16x4 xq[2][2][G];
    unsigned xoff = 0;
    int i_q = 0;                                  // X_INDEX: the index of step min(t+2, T-1)
    const unsigned char* xbase0;
    if (XMODE == MVAE_X_DENSE) {
        xoff = lane8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp) + ((size_t)bx * (GH / 16) + w * 2) * 512;
    } else if (XMODE == MVAE_X_INDEX) {
        xoff = (unsigned)a.idx[b] * (GH * 2) + q * 16;         // MVAE_TABLE_PAIRED: this wave's tile pair in one 16-byte gather
        xbase0 = reinterpret_cast<const unsigned char*>(a.table) + w * 64;
        i_q = a.idx[(size_t)(T > 2 ? 2 : T - 1) * B + b];
    } else {
        xoff = (unsigned)b * (GH * 2) + q * 8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp0) + w * 64;
    }
    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = 0, phi = cs_steps;
    if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready, wait_value, a.status);
    constexpr unsigned XG = XMODE == MVAE_X_DENSE ? (RH / 16) * 512 : RH * 2;
    constexpr unsigned XN = XMODE == MVAE_X_DENSE ? 512 : 32;
    const size_t x_step1 = (XMODE == MVAE_X_DENSE && T > 1) ? tps * (GH / 16) * 512 : 0;
    const unsigned xoff1 = XMODE == MVAE_X_INDEX ? (unsigned)a.idx[(size_t)(T > 1 ? 1 : 0) * B + b] * (GH * 2) + q * 16 : xoff;
    if (XMODE == MVAE_X_INDEX) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const u16x8 p0 = *reinterpret_cast<const u16x8*>(xbase0 + g * XG + xoff);
            const u16x8 p1 = *reinterpret_cast<const u16x8*>(xbase0 + g * XG + xoff1);
            xq[0][0][g] = __builtin_shufflevector(p0, p0, 0, 1, 2, 3);
            xq[0][1][g] = __builtin_shufflevector(p0, p0, 4, 5, 6, 7);
            xq[1][0][g] = __builtin_shufflevector(p1, p1, 0, 1, 2, 3);
            xq[1][1][g] = __builtin_shufflevector(p1, p1, 4, 5, 6, 7);
        }
    } else {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                xq[0][n][g] = *reinterpret_cast<const u16x4*>(xbase0 + g * XG + n * XN + xoff);
                if (XMODE != MVAE_X_CONST) xq[1][n][g] = *reinterpret_cast<const u16x4*>(xbase0 + x_step1 + g * XG + n * XN + xoff1);
            }
    }

    gbyte *acts_p[G], *hs_p, *hh_prev_p;          // step t: saved gates; h_{t-1} (slot t); the candidate tiles of step t-1
    gbyte* x_p[G];                                // step min(t+2, T-1): inputs
    const size_t acts_step = tps * (GH / 16) * 512, hs_step = (size_t)B * RH * 2;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        acts_p[g] = to_global(a.acts) + ((size_t)bx * (GH / 32) + g * (RH / 32) + w) * 1024;
        x_p[g] = to_global(xbase0) + g * XG + (XMODE == MVAE_X_DENSE ? (T > 2 ? 2 : T - 1) * acts_step : 0);
    }
    hh_prev_p = acts_p[2];
    hs_p = to_global(a.hs) + (size_t)bx * 16 * (RH * 2);

    constexpr float K2 = 2.8853900817779268f;
    f32x4 accR[2], accZ[2], accC[2];
    u16x8 hh_pk = u16x8{0, 0, 0, 0, 0, 0, 0, 0};  // candidate of the previous step (this wave's tile pair), stored during phase A
    frag bA[8], lt[2], cp[2];
    vm_drain();
    lds_barrier();
    if (W8_PRIO && w >= 4) __builtin_amdgcn_s_setprio(1);

    auto step = [&](const int t, auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;                     // t & 1: h_{t-1} sits in h buffer PAR
        constexpr int XB = XMODE == MVAE_X_CONST ? 0 : PAR;            // the buffer holding this step's inputs
        unsigned char* hcur = hbuf + PAR * 8192;
        unsigned char* hnext = hbuf + (1 - PAR) * 8192;
        pinu(hw0); pinu(tl0);
        if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready && t + 2 < T && t + 2 == phi)
            wave_wait_ge(uniform_ptr(a.wait_ready + pk + 1), wait_value, a.status);
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(hs_p); pins(hh_prev_p);
        pins(x_p[0]); pins(x_p[1]); pins(x_p[2]);
#pragma unroll
        for (int n = 0; n < 2; ++n) { pin1(xq[XB][n][0]); pin1(xq[XB][n][1]); pin1(xq[XB][n][2]); }
        if (XMODE == MVAE_X_INDEX) pini(i_q);
        // ---- phase A ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (!W8_ABL_NOB || t == 0) bA[ks] = *reinterpret_cast<const frag*>(hcur + bf4[ks & 3] + 256 * (ks >> 2));
        if (NLDS > 0 && !W8_ABL_NOL) lt[0] = myl[0];
        if (NLDS > 1 && !W8_ABL_NOL) lt[1] = myl[64];
        if (SAVE >= SAVE_HS && copier) {
            cp[0] = *reinterpret_cast<const frag*>(hcur + tl0);
            cp[1] = *reinterpret_cast<const frag*>(hcur + (tl0 ^ 1056u));
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            accZ[n] = unpack4(xq[XB][n][0]);
            accR[n] = unpack4(xq[XB][n][1]);
        }
        asm volatile("s_nop 1" : "+v"(accR[0]), "+v"(accR[1]), "+v"(accZ[0]), "+v"(accZ[1]));
        auto request_x = [&](int n, int g) __attribute__((always_inline)) {
            if (XMODE == MVAE_X_INDEX) {
                if (n & 1) return;
                if (g == 0) xoff = (unsigned)i_q * (GH * 2) + q * 16;
                pinu(xoff);
                const u16x8 pr = *reinterpret_cast<const g_u16x8*>(x_p[g] + xoff);
                xq[XB][0][g] = __builtin_shufflevector(pr, pr, 0, 1, 2, 3);
                xq[XB][1][g] = __builtin_shufflevector(pr, pr, 4, 5, 6, 7);
            } else if (XMODE != MVAE_X_CONST) {
                pinu(xoff);
                xq[XB][n][g] = *reinterpret_cast<const g_u16x4*>(x_p[g] + n * XN + xoff);
            }
        };
        static_for<0, 32>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value;
            constexpr slot_a sa = gru_slot_a(sl);
            constexpr bool isl = gru_a_is_l(sl, NLDS);
            constexpr int li = gru_a_lidx(sl, NLDS);
            f32x4& acc = sa.g == 1 ? accR[sa.n] : accZ[sa.n];
            if constexpr (isl && W8_ABL_NOL) {
                mfma1<true>(acc, uc[li], bA[sa.ks]);
            } else if constexpr (isl) {
                mfma1<false>(acc, lt[li & 1], bA[sa.ks]);
                if constexpr (li + 2 < NLDS) lt[li & 1] = myl[(size_t)(li + 2) * 64];
            } else {
                mfma1<true>(acc, ua[NAA > 0 ? sl - li : 0], bA[sa.ks]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- fillers ----
            if constexpr (sl == 1) {
                if (SAVE == SAVE_ALL && t > 0) {
                    pinu(lane16);
                    *reinterpret_cast<g_u16x8*>(hh_prev_p + lane16) = hh_pk;
                }
            }
            if constexpr (sl == 3 || sl == 5) {
                if (SAVE >= SAVE_HS && copier) {
                    pinu(tg0);
                    store16_wt(hs_p, tg0 + (sl == 5 ? 1024u : 0u), cp[sl == 5 ? 1 : 0]);
                }
            }
            if constexpr (sl >= 8 && sl < 16 && (sl & 1) == 0) {       // the z / r inputs of step t+2
                constexpr int k = (sl - 8) >> 1;
                request_x(k >> 1, k & 1);
            }
            // r of tiles 0, 1 (their last MFMA was slot 20 / 21): hard_sigmoid, r*h -> rh tile
            if constexpr (sl == 24 || sl == 26) {
                constexpr int n = (sl - 24) >> 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!W8_ABL_NOMATH) accR[n][e] = hsig(accR[n][e]);
            }
            if constexpr (sl == 25 || sl == 27) {
                constexpr int n = (sl - 25) >> 1;
                if (!W8_ABL_NOMATH) *reinterpret_cast<u16x4*>(rhbuf + (hw0 ^ (n << 5))) = pack4(accR[n] * hreg[n]);
            }
            if constexpr (sl == 28) {
                if (SAVE == SAVE_ALL) {
                    pinu(lane16);
                    *reinterpret_cast<g_u16x8*>(acts_p[1] + lane16) = cat8(pack4(accR[0]), pack4(accR[1]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        w8_barrier();
        // ---- phase C ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (!W8_ABL_NOB) bA[ks] = *reinterpret_cast<const frag*>(rhbuf + bf4[ks & 3] + 256 * (ks >> 2));
#pragma unroll
        for (int n = 0; n < 2; ++n) accC[n] = unpack4(xq[XB][n][2]);
        asm volatile("s_nop 1" : "+v"(accC[0]), "+v"(accC[1]));
        // candidate -> h for one element of tile n
        auto h_math = [&](int n, int e) __attribute__((always_inline)) {
            if (W8_ABL_NOMATH) return;
            const float ex = __builtin_amdgcn_exp2f(accC[n][e] * K2);
            const float hh = 1.0f - 2.0f * __builtin_amdgcn_rcpf(ex + 1.0f);
            accC[n][e] = hh;                                       // kept for the save
            hreg[n][e] = hh + accZ[n][e] * (hreg[n][e] - hh);      // z*h + (1-z)*hh
        };
        static_for<0, 16>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value, n = sl >> 3, ks = sl & 7;
            mfma1<true>(accC[n], uc[sl], bA[ks]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (sl == 3 || sl == 4) {                    // z (its last MFMA was phase A's slot 30 / 31)
                constexpr int m = sl - 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!W8_ABL_NOMATH) accZ[m][e] = hsig(accZ[m][e]);
            }
            if constexpr (sl == 5) {
                if (SAVE == SAVE_ALL) {
                    pinu(lane16);
                    *reinterpret_cast<g_u16x8*>(acts_p[0] + lane16) = cat8(pack4(accZ[0]), pack4(accZ[1]));
                }
            }
            if constexpr (sl == 6 || sl == 7) request_x(sl - 6, 2);        // the candidate inputs of step t+2
            if constexpr (sl == 8) {
                if (XMODE == MVAE_X_INDEX) i_q = a.idx[(size_t)(t + 3 < T ? t + 3 : T - 1) * B + b];
            }
            if constexpr (sl >= 11 && sl < 15) h_math(0, sl - 11);        // tile 0 (its last MFMA was slot 7)
            if constexpr (sl == 15) *reinterpret_cast<u16x4*>(hnext + hw0) = pack4(hreg[0]);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(accC[1]));
#pragma unroll
        for (int e = 0; e < 4; ++e) h_math(1, e);
        *reinterpret_cast<u16x4*>(hnext + (hw0 ^ 32u)) = pack4(hreg[1]);
        hh_pk = cat8(pack4(accC[0]), pack4(accC[1]));
        if (t == T - 1 && a.h_last) {
#pragma unroll
            for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub0 + 16 * n) = hreg[n];
        }
        hh_prev_p = acts_p[2];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acts_p[g] += acts_step;
            if (XMODE == MVAE_X_DENSE && t + 3 < T) x_p[g] += acts_step;
        }
        hs_p += hs_step;
        w8_barrier();
        if (cs_steps && t == phi) {
            if (SAVE >= SAVE_HS && a.signal_done && copier) wave_signal_done<false>(uniform_ptr(a.signal_done + pk));
            ++pk;
            phi += cs_steps;
        }
    };
    int t = 0;
    for (; t + 1 < T; t += 2) {
        step(t, std::integral_constant<int, 0>{});
        step(t + 1, std::integral_constant<int, 1>{});
    }
    if (t < T) step(t, std::integral_constant<int, 0>{});
    if (SAVE == SAVE_ALL) *reinterpret_cast<g_u16x8*>(hh_prev_p + lane16) = hh_pk;
    if (SAVE >= SAVE_HS && copier) {       // slot T = h_{T-1}
        const unsigned char* hfin = hbuf + (T & 1) * 8192;
        store16_wt(hs_p, tg0, *reinterpret_cast<const u16x8*>(hfin + tl0));
        store16_wt(hs_p, tg0 + 1024u, *reinterpret_cast<const u16x8*>(hfin + (tl0 ^ 1056u)));
        if (cs_steps && a.signal_done) wave_signal_done<false>(uniform_ptr(a.signal_done + pk));
    }
    vm_drain();
}

#ifndef GRU_W8_NLDS
#define GRU_W8_NLDS 16
#endif
template <int XMODE, int SAVE>
__global__ __launch_bounds__(512, 1) void gru_fwd_w8_k(const mvae_rnn_fwd_args a) {
    gru_fwd_w8_body<XMODE, SAVE, GRU_W8_NLDS>(a, blockIdx.x);
}

template <int XMODE, int SAVE>
int launch_gru_w8(const mvae_rnn_fwd_args& a, hipStream_t s) {
    const size_t lds = (size_t)3 * 16 * RH * sizeof(bf16_t) + (size_t)8 * GRU_W8_NLDS * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_w8_k<XMODE, SAVE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((gru_fwd_w8_k<XMODE, SAVE>), dim3(a.B / 16), dim3(512), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <int XMODE>
int gru_w8_save(const mvae_rnn_fwd_args& a, hipStream_t s) {
    if (XMODE == MVAE_X_INDEX && a.table_layout != MVAE_TABLE_PAIRED) return MVAE_E_ARG;
    if (a.acts) {
        if (!a.hs) return MVAE_E_UNSUPPORTED;
        return launch_gru_w8<XMODE, SAVE_ALL>(a, s);
    }
    if (a.cs) return MVAE_E_UNSUPPORTED;
    return a.hs ? launch_gru_w8<XMODE, SAVE_HS>(a, s) : launch_gru_w8<XMODE, SAVE_NONE>(a, s);
}

}  // namespace

// Entry point used by rnn_resident.hip's dispatch.  MVAE_E_UNSUPPORTED: not a shape of this file.
int mvae_rnn_fwd_w8(const mvae_rnn_fwd_args& a, hipStream_t s) {
    if (a.H != RH || a.dtype != MVAE_BF16 || (a.B % 16) != 0 || a.seq_layout != MVAE_TILE16P || a.cell != MVAE_GRU)
        return MVAE_E_UNSUPPORTED;
    switch (a.xmode) {
        case MVAE_X_DENSE: return a.xp ? gru_w8_save<MVAE_X_DENSE>(a, s) : MVAE_E_ARG;
        case MVAE_X_INDEX: return (a.idx && a.table) ? gru_w8_save<MVAE_X_INDEX>(a, s) : MVAE_E_ARG;
        case MVAE_X_CONST: return a.xp0 ? gru_w8_save<MVAE_X_CONST>(a, s) : MVAE_E_ARG;
    }
    return MVAE_E_UNSUPPORTED;
}
