// Resident-weights recurrent kernels, TWO WAVES PER SIMD (round 6): H = 256, bf16 MFMA operands, gfx950.  DESIGN.md section 3.5.
//
// Same contract and numerics as the slot-interleaved kernels of rnn_resident.hip; different occupancy.  There one wave per SIMD
// owns 64 hidden units and 512 registers: whatever that wave cannot issue under its own MFMAs (a 16x16x32 MFMA holds the matrix
// pipe 16 cycles, ~2 other instructions fit underneath) is dead time, and every dependency stall - MFMA result -> gate
// arithmetic, ds_read -> MFMA, the barrier - stalls the SIMD (round-5 counters: issue active 51 % of the cycles).  Here a workgroup
// is 8 waves of 256 registers (hipcc: 128 VGPRs + 128 AGPRs): two waves share each SIMD's matrix pipe and VALU port, and the
// hardware issues one wave's gate arithmetic, LDS and memory instructions under the other's MFMAs (tools/probes/issue2_probe.hip).
//
// GRU (the product; seq_layout MVAE_TILE16Q, one-hot tables MVAE_TABLE_PAIRED8): U^T is 384 KiB = 48 fragments per wave, 32 of them
// in AGPRs, 16 in a private LDS slab (every third MFMA slot), nothing streamed.  Wave w owns unit tiles a = w and b = 8 + w of
// every gate - tiles a of all waves are k-groups 0..3 of h, tiles b k-groups 4..7 - so that a step is a software pipeline: the
// tile-b half of step t-1 (tanh, h update, LDS write) runs under the first MFMAs of step t, and THREE half exchanges of h / r*h
// (barriers 2b, 1, 2a) replace two full ones and hide behind the z gate's MFMAs.  Every request of sequence data is an un-tracked
// asm load waited for by COUNT (vmcnt retires in issue order, loads and stores alike; hipcc's own bookkeeping gives up at loop
// headers), with the three pitfalls of that noted where they bit: xload8 / pin* below, the epilogues, the unconditional requests.
//
// LSTM BPTT (lstm_bwd_w8_k, MVAE_LSTM_BWD_W8=1, NOT the product): U^T is 512 KiB = the whole register file; a quarter of it streams
// from L2 every step and queues behind the HBM requests of the next step's saved activations: 3.9 vs 2.7 us per step
// (profiles/r06_g_lstm_bptt_w8.txt).  Kept with its parity test because the finding is the kernel.
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
#ifndef W8_B2B
#define W8_B2B 13           /* GRU forward: slot behind which barrier 2b stands (the h_b write is slot 6; h_b is needed from slot 16) */
#endif
#ifndef W8_B1
#define W8_B1 30            /* ... barrier 1 (the last r*h write is slot 27; r*h is needed from slot 32) */
#endif
#ifndef W8_TA0
#define W8_TA0 42           /* ... first slot of tile a's tanh + h update (3 slots, then the h_a write; its last MFMA is slot 39) */
#endif
#ifndef W8_BWD_Z_LATE
#define W8_BWD_Z_LATE 0     /* GRU BPTT: da_z (only M2z waits for it, slot 16) computed behind barrier 1 and exchanged by a barrier of its own in slot 8 */
#endif
#ifndef W8_ABL_NOSTREAM
#define W8_ABL_NOSTREAM 0   /* LSTM BPTT: the fragments streamed from L2 replaced by register ones (no requests) */
#endif
#ifndef W8_ABL_NOLOADS
#define W8_ABL_NOLOADS 0    /* LSTM BPTT: no requests of the next step's saved values (the stream alone in the memory queue) */
#endif
#ifndef W8_PRIO
#define W8_PRIO 0           /* s_setprio 1 for waves 4..7 (the second-dispatched wave of every SIMD) */
#endif
#ifndef MVAE_VARIANT_BUILD
static_assert(!W8_ABL_NOBAR && !W8_ABL_NOMATH && !W8_ABL_NOL && !W8_ABL_NOB && !W8_ABL_NOSTREAM && !W8_ABL_NOLOADS, "timing ablations: variant builds only");
#endif

// Phase stamps (build with -DW8_STAMPS): lane 0 of waves 0 and 4 of block 0 records s_memtime at marked points of steps
// [64, 72); read back with mvae_debug_stamps_w8().  Development tooling only.
#ifdef W8_STAMPS
__device__ unsigned long long mvae_w8_stamps[2 * 8 * 16];
#define W8_STAMP(k)                                                                                                   \
    do {                                                                                                              \
        if (bx == 0 && l == 0 && (w & 3) == 0 && t >= 64 && t < 72)                                                   \
            mvae_w8_stamps[((w >> 2) * 8 + (t - 64)) * 16 + (k)] = __builtin_readcyclecounter();                      \
    } while (0)
extern "C" int mvae_debug_stamps_w8(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mvae_w8_stamps), sizeof(mvae_w8_stamps)) == hipSuccess ? 0 : -3;
}
#else
#define W8_STAMP(k)
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
template <int N>
__device__ __forceinline__ void vm_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// un-tracked requests (the caller waits by count, then pins the destination): uniform base + 32-bit lane offset
__device__ __forceinline__ void xload8(u16x4& d, const __attribute__((address_space(1))) unsigned char* base, unsigned off) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void xload16(u16x8& d, const __attribute__((address_space(1))) unsigned char* base, unsigned off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void pin8(u16x8& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void iload1(int& d, const uint8_t* p) {
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
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

#include "rnn_multi.h"

// ---------------------------------------------------------------------------------------------------------
// GRU forward (Keras 2.0.x GRU: reset gate applied BEFORE the candidate matmul; gate order [z | r | candidate])
// ---------------------------------------------------------------------------------------------------------
// Wave w owns the unit tiles a = w (units [16w, 16w+16)) and b = 8 + w: the a tiles of all waves are the k-groups 0..3 of
// h, the b tiles the k-groups 4..7 - the two halves of h are exchanged at DIFFERENT points of the step, each behind
// MFMAs that do not need it.  A step is 48 MFMA slots per wave; what the dependency chain
//     candidate(tile) -> tanh, h update -> LDS -> barrier -> ds_read -> r = U_r h -> hard_sigmoid, r*h -> LDS -> barrier
//     -> ds_read -> candidate
// leaves idle (arithmetic, LDS round trips, barriers) is filled with the z MFMAs, which nothing waits for until the h update:
//   slots  0.. 7  r, both tiles, k-groups 0..3 (h_a)        fillers: tanh + h update of tile b of the PREVIOUS step -> h_b
//   -- barrier 2b; ds_read h_b (k-groups 4..7)
//   slots  8..15  z, k-groups 0..3                           (cover the read)
//   slots 16..23  r, tile a then tile b, k-groups 4..7       fillers: r_a -> r*h
//   slots 24..27  z, tile a, k-groups 4..7                   fillers: r_b -> r*h
//   -- barrier 1; ds_read r*h (all k-groups)
//   slots 28..31  z, tile b, k-groups 4..7                   (cover)
//   slots 32..39  candidate, tile a                          fillers: z
//   slots 40..47  candidate, tile b                          fillers: tanh + h update of tile a -> h_a
//   -- barrier 2a; ds_read h_a (k-groups 0..3) of the next step
struct w8_slot { int kind, n, ks; };        // kind 0 = z, 1 = r, 2 = candidate (the gate index of the packed kernel)
__host__ __device__ constexpr w8_slot gru_slot(int s) {
    if (s < 8) return {1, s & 1, s >> 1};
    if (s < 16) return {0, (s - 8) >> 2, (s - 8) & 3};
    if (s < 24) return {1, (s - 16) >> 2, 4 + ((s - 16) & 3)};
    if (s < 32) return {0, (s - 24) >> 2, 4 + ((s - 24) & 3)};
    return {2, (s - 32) >> 3, (s - 32) & 7};
}
// 16 of the 48 fragments of a wave live in LDS (every third slot), 32 in accumulator registers
__host__ __device__ constexpr bool gru_is_l(int s) { return s % 3 == 2; }
__host__ __device__ constexpr int gru_lidx(int s) { return s / 3; }

template <int XMODE, int SAVE>
__device__ __forceinline__ void gru_fwd_w8_body(const mvae_rnn_fwd_args& a, const unsigned bx) {
    constexpr int G = 3, GH = G * RH, NLDS = 16;
    static_assert(XMODE != MVAE_X_SCALAR, "scalar inputs run on the phased kernel");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* hbuf = smem;                                             // [2][16][RH] bf16, swizzled
    unsigned char* rhbuf = smem + 2 * 16 * RH * 2;                          // [16][RH]
    frag* ulds = reinterpret_cast<frag*>(smem + 3 * 16 * RH * 2);           // [8][NLDS][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                 // 0..7: unit tiles w and 8 + w
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.u_pack);
    frag* myl = ulds + (size_t)w * NLDS * 64 + l;
    auto src_frag = [&](int g, int n, int ks) -> const frag* {
        return up + (size_t)((g * (RH / 16) + n * 8 + w) * 8 + ks) * 64 + l;
    };
    // (the register-resident fragments first, un-waited; then the LDS-resident ones as plain loads - interleaved, every plain
    //  load behind a volatile one would be waited for by itself: 16 round trips in a row)
    frag ua[32];
    static_for<0, 48>(SF_LAMBDA(sc) {
        constexpr int s = decltype(sc)::value;
        constexpr w8_slot sa = gru_slot(s);
        if constexpr (!gru_is_l(s)) load1_agpr_nowait(ua[s - gru_lidx(s)], src_frag(sa.kind, sa.n, sa.ks));
    });
    {
        frag tmp[NLDS];
        static_for<0, 48>(SF_LAMBDA(sc) {
            constexpr int s = decltype(sc)::value;
            constexpr w8_slot sa = gru_slot(s);
            if constexpr (gru_is_l(s)) tmp[gru_lidx(s)] = *src_frag(sa.kind, sa.n, sa.ks);
        });
#pragma unroll
        for (int i = 0; i < NLDS; ++i) myl[(size_t)i * 64] = tmp[i];
    }

    const int ld0 = a.h0_ld ? a.h0_ld : RH, ldl = a.h_last_ld ? a.h_last_ld : RH;
    unsigned lane8 = (unsigned)l * 8u;
    unsigned lane16 = (unsigned)l * 16u;                                  // TILE16Q: this wave's tiles w and 8 + w are pair w
    const int ub0 = w * 16 + q * 4;
    // h tile write position of tile a (tile b: + 256 bytes); B fragment k-group ks: bf4[ks & 3] + 256 * (ks >> 2)
    unsigned hw0 = (unsigned)r * 512u + ((((unsigned)w * 2u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    unsigned bf4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf4[j] = (unsigned)r * 512u + ((((unsigned)j * 4u + (unsigned)q) ^ (unsigned)r) << 4);
    // row-major copy of h_{t-1} (hs slot t): one 16-byte chunk per lane, rows 2w and 2w + 1; every wave publishes its own
    // stores: a consumer counts 8 increments per workgroup and chunk (mvae_rnn_waves())
    const unsigned row0 = 2u * (unsigned)w + ((unsigned)l >> 5), ch0 = (unsigned)l & 31u;
    unsigned tl0 = row0 * 512u + ((ch0 ^ row0) << 4);
    unsigned tg0 = row0 * 512u + ch0 * 16u;

    f32x4 hreg[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        hreg[n] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * ld0 + ub0 + 128 * n) : z4;
        *reinterpret_cast<u16x4*>(hbuf + hw0 + 256 * n) = pack4(hreg[n]);
    }
    // ---- x queue: requested TWO steps ahead into alternating buffers (the step loop is unrolled by two): beside other
    // kernels an HBM round trip takes longer than the one step a single buffer gives it.  The requests are un-tracked asm loads
    // and every wait is hand-counted (all waves issue the same memory instructions in the same order, vmcnt retires in issue
    // order): hipcc's own bookkeeping loses count at the loop header and waits for all but the two youngest instructions there.
    u16x4 xq[2][2][G];            // [buffer][tile][gate]
    u16x8 xq8[2][G];              // X_INDEX: one 16-byte gather holds both tiles
    auto xv = [&](int buf, int n, int g) __attribute__((always_inline)) -> u16x4 {
        if (XMODE == MVAE_X_INDEX)
            return n ? __builtin_shufflevector(xq8[buf][g], xq8[buf][g], 4, 5, 6, 7) : __builtin_shufflevector(xq8[buf][g], xq8[buf][g], 0, 1, 2, 3);
        return xq[buf][n][g];
    };
    auto xpin = [&](int buf, int g) __attribute__((always_inline)) {
        if (XMODE == MVAE_X_INDEX) pin8(xq8[buf][g]);
        else { pin1(xq[buf][0][g]); pin1(xq[buf][1][g]); }
    };
    unsigned xoff = 0;
    int i_q = 0;                                  // X_INDEX: the index of step min(t+2, T-1)
    const unsigned char* xbase0;
    if (XMODE == MVAE_X_DENSE) {
        xoff = lane8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp) + ((size_t)bx * (GH / 16) + w) * 512;
    } else if (XMODE == MVAE_X_INDEX) {
        xoff = (unsigned)a.idx[b] * (GH * 2) + q * 16;         // MVAE_TABLE_PAIRED8: this wave's tiles w and 8 + w in one 16-byte gather
        xbase0 = reinterpret_cast<const unsigned char*>(a.table) + w * 64;
        i_q = a.idx[(size_t)(T > 2 ? 2 : T - 1) * B + b];
    } else {
        xoff = (unsigned)b * (GH * 2) + q * 8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp0) + w * 32;
    }
    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = 0, phi = cs_steps;
    if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready, wait_value, a.status);
    constexpr unsigned XG = XMODE == MVAE_X_DENSE ? (RH / 16) * 512 : RH * 2;      // gate stride
    constexpr unsigned XN = XMODE == MVAE_X_DENSE ? 8 * 512 : 256;                  // tile a -> tile b
    const size_t x_step1 = (XMODE == MVAE_X_DENSE && T > 1) ? tps * (GH / 16) * 512 : 0;
    const unsigned xoff1 = XMODE == MVAE_X_INDEX ? (unsigned)a.idx[(size_t)(T > 1 ? 1 : 0) * B + b] * (GH * 2) + q * 16 : xoff;
    if (XMODE == MVAE_X_INDEX) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            xq8[0][g] = *reinterpret_cast<const u16x8*>(xbase0 + g * XG + xoff);
            xq8[1][g] = *reinterpret_cast<const u16x8*>(xbase0 + g * XG + xoff1);
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
    // memory instructions per step, in issue order: [candidate save] r requests [h copy] z, candidate requests [index] [r save] [z save]
    constexpr int V_SV = SAVE == SAVE_ALL ? 1 : 0, V_CP = SAVE >= SAVE_HS ? 1 : 0, V_IX = XMODE == MVAE_X_INDEX ? 1 : 0;
    constexpr int V_RQ = XMODE == MVAE_X_CONST ? 0 : (XMODE == MVAE_X_INDEX ? 1 : 2);      // requests per gate
    constexpr int V_STEP = 3 * V_SV + 3 * V_RQ + V_CP + V_IX;

    gbyte *acts_p[G], *hs_p, *hh_prev_p;          // step t: saved gates; h_{t-1} (slot t); the candidate tiles of step t-1
    gbyte* x_p[G];                                // step min(t+2, T-1): inputs
    const size_t acts_step = tps * (GH / 16) * 512, hs_step = (size_t)B * RH * 2;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        acts_p[g] = to_global(a.acts) + ((size_t)bx * (GH / 32) + g * (RH / 32) + w) * 1024;
        x_p[g] = to_global(xbase0) + g * XG + (XMODE == MVAE_X_DENSE ? (T > 2 ? 2 : T - 1) * acts_step : 0);
    }
    hh_prev_p = acts_p[2];                        // (step 0 stores a placeholder where its own candidate goes one step later)
    hs_p = to_global(a.hs) + (size_t)bx * 16 * (RH * 2);

    constexpr float K2 = 2.8853900817779268f;
    // step -1 of tile b: candidate 0 and z = 1 make the deferred h update reproduce h0 exactly (0 + 1 * (h0 - 0))
    f32x4 accR[2], accZ[2], accC[2];
    accC[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    accC[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    accZ[1] = f32x4{1.f, 1.f, 1.f, 1.f};
    frag bh[8], lt[2], cp;     // B fragments: h k-groups 0..7, then (dead registers first) r*h: 0..3 behind barrier 1, 4..7 behind slot 31
    vm_drain();
    lds_barrier();
    if (W8_PRIO && w >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bh[ks] = *reinterpret_cast<const frag*>(hbuf + bf4[ks]);
#pragma unroll
    for (int n = 0; n < 2; ++n) accR[n] = unpack4(xv(0, n, 1));

    // tanh + h update for one element of tile n (accC[n] = candidate pre-activation, accZ[n] = z); the candidate stays in accC
    // in STAGES over the four elements of a tile (a stage per MFMA slot): one element's chain mul -> exp -> add -> rcp -> fma ->
    // sub -> fma is seven dependent instructions; four chains side by side are four independent instructions per link
    auto h_stage = [&](int n, int stage) __attribute__((always_inline)) {
        if (W8_ABL_NOMATH) return;
        // (element by element: arithmetic on the vector types becomes v_pk_*_f32, which costs a SIMD that also issues MFMAs
        //  ~16 cycles per instruction - tools/probes/issue2_probe.hip - against ~4.4 for a plain VALU instruction)
        f32x4& c = accC[n];
        if (stage == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_exp2f(c[e] * K2);
        } else if (stage == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_rcpf(c[e] + 1.0f);
        } else if (stage == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c[e] = __builtin_fmaf(c[e], -2.0f, 1.0f);                          // the candidate (kept for the save)
                hreg[n][e] = __builtin_fmaf(accZ[n][e], hreg[n][e] - c[e], c[e]);  // z*h + (1-z)*hh
            }
        }
    };
    auto mul4 = [&](const f32x4& x, const f32x4& y) __attribute__((always_inline)) {
        return f32x4{x[0] * y[0], x[1] * y[1], x[2] * y[2], x[3] * y[3]};
    };

    auto step = [&](const int t, auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;                     // t & 1: h_{t-1} sits in h buffer PAR
        constexpr int XB = XMODE == MVAE_X_CONST ? 0 : PAR;            // the buffer holding this step's inputs
        constexpr int XNX = XMODE == MVAE_X_CONST ? 0 : 1 - PAR;       // ... the next step's
        unsigned char* hcur = hbuf + PAR * 8192;
        unsigned char* hnext = hbuf + (1 - PAR) * 8192;
        pinu(hw0); pinu(tl0);
        // pipelined stack: x of step t+2 is requested during this step - its chunk must have been published
        if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready && t + 2 < T && t + 2 == phi)
            wave_wait_ge(uniform_ptr(a.wait_ready + pk + 1), wait_value, a.status);
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(hs_p); pins(hh_prev_p);
        pins(x_p[0]); pins(x_p[1]); pins(x_p[2]);
        // this step's z and candidate inputs were requested two steps ago: everything but the previous step's instructions has retired
        if (XMODE != MVAE_X_CONST) vm_wait<V_STEP + V_IX + 2 * V_SV>();
        xpin(XB, 0); xpin(XB, 2);
        lt[0] = myl[0];
        lt[1] = myl[64];
        asm volatile("s_nop 1" : "+v"(accR[0]), "+v"(accR[1]));
        auto request_x = [&](int n, int g) __attribute__((always_inline)) {
            if (XMODE == MVAE_X_CONST) return;
            if (XMODE == MVAE_X_INDEX) {              // one gather per gate: both tiles of this wave
                if (n) return;
                if (g == 1) {                         // (the first request of a step) the index was loaded one step ago
                    vm_wait<3 * V_SV>();
                    pini(i_q);
                    xoff = (unsigned)i_q * (GH * 2) + q * 16;
                }
                pinu(xoff);
                xload16(xq8[XB][g], x_p[g], xoff);
            } else {
                pinu(xoff);
                xload8(xq[XB][n][g], x_p[g] + n * XN, xoff);
            }
        };
        static_for<0, 48>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value;
            constexpr w8_slot sa = gru_slot(sl);
            constexpr int li = gru_lidx(sl);
            f32x4& acc = sa.kind == 1 ? accR[sa.n] : (sa.kind == 0 ? accZ[sa.n] : accC[sa.n]);
            const frag& bop = bh[sa.ks];
            if constexpr (gru_is_l(sl) && W8_ABL_NOL) {
                mfma1<true>(acc, ua[li], bop);
            } else if constexpr (gru_is_l(sl)) {
                mfma1<false>(acc, lt[li & 1], bop);
                lt[li & 1] = myl[(size_t)((li + 2) & 15) * 64];
            } else {
                mfma1<true>(acc, ua[sl - li], bop);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- fillers ---------------------------------------------------------------------------------------------
            if constexpr (sl == 0) { W8_STAMP(0); }
            if constexpr (sl == 15) { W8_STAMP(4); }
            if constexpr (sl == 23) { W8_STAMP(5); }
            if constexpr (sl == 31) { W8_STAMP(8); }
            if constexpr (sl == 39) { W8_STAMP(9); }
            if constexpr (sl == 0) accZ[0] = unpack4(xv(XB, 0, 0));
            if constexpr (sl >= 3 && sl < 6) h_stage(1, sl - 3);          // tile b of step t-1 (its last MFMA: slot 47)
            if constexpr (sl == 6) *reinterpret_cast<u16x4*>(hcur + hw0 + 256) = pack4(hreg[1]);
            if constexpr (sl == W8_B2B) {
                W8_STAMP(2);
                w8_barrier();                                             // ---- 2b: h_{t-1} is complete in hcur
                W8_STAMP(3);
#pragma unroll
                for (int ks = 4; ks < 8; ++ks)
                    if (!W8_ABL_NOB) bh[ks] = *reinterpret_cast<const frag*>(hcur + bf4[ks & 3] + 256);
                if (SAVE >= SAVE_HS) cp = *reinterpret_cast<const frag*>(hcur + tl0);
            }
            if constexpr (sl == 8) {                                      // the candidate of step t-1 (both tiles), then its registers
                if (SAVE == SAVE_ALL) {                                   // start the next accumulation
                    pinu(lane16);
                    *reinterpret_cast<g_u16x8*>(hh_prev_p + lane16) = cat8(pack4(accC[0]), pack4(accC[1]));
                }
                accZ[1] = unpack4(xv(XB, 1, 0));
            }
            if constexpr (sl == 9) {
                accC[0] = unpack4(xv(XB, 0, 2));
                accC[1] = unpack4(xv(XB, 1, 2));
            }
            if constexpr (sl == 13) {
                if (SAVE >= SAVE_HS) {
                    pinu(tg0);
                    store16_wt(hs_p, tg0, cp);
                }
            }
            // the inputs of step t+2 (this step's are consumed: r at the end of the previous step, z and candidate above)
            if constexpr (sl == 10) request_x(0, 1);
            if constexpr (sl == 11) request_x(1, 1);
            if constexpr (sl == 12) request_x(0, 0);
            if constexpr (sl == 14) request_x(1, 0);
            if constexpr (sl == 15) request_x(0, 2);
            if constexpr (sl == 16) request_x(1, 2);
            if constexpr (sl == 17) {
                if (XMODE == MVAE_X_INDEX) iload1(i_q, a.idx + (size_t)(t + 3 < T ? t + 3 : T - 1) * B + b);
            }
            // r (tile a: last MFMA slot 19, tile b: 23): hard_sigmoid, r*h -> rh tile
            if constexpr (sl == 22 || sl == 26) {
                constexpr int n = sl == 26;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!W8_ABL_NOMATH) accR[n][e] = hsig(accR[n][e]);
            }
            if constexpr (sl == 23 || sl == 27) {
                constexpr int n = sl == 27;
                if (!W8_ABL_NOMATH) *reinterpret_cast<u16x4*>(rhbuf + hw0 + 256 * n) = pack4(mul4(accR[n], hreg[n]));
            }
            if constexpr (sl == W8_B1) {
                W8_STAMP(6);
                w8_barrier();                                             // ---- 1: r*h is complete
                W8_STAMP(7);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (!W8_ABL_NOB) bh[ks] = *reinterpret_cast<const frag*>(rhbuf + bf4[ks]);
            }
            if constexpr (sl == 31) {
#pragma unroll
                for (int ks = 4; ks < 8; ++ks)
                    if (!W8_ABL_NOB) bh[ks] = *reinterpret_cast<const frag*>(rhbuf + bf4[ks & 3] + 256);
            }
            if constexpr (sl == 29) {
                if (SAVE == SAVE_ALL) {
                    pinu(lane16);
                    *reinterpret_cast<g_u16x8*>(acts_p[1] + lane16) = cat8(pack4(accR[0]), pack4(accR[1]));
                }
            }
            // z (tile a: last MFMA slot 27, tile b: 31)
            if constexpr (sl == 33 || sl == 35) {
                constexpr int n = sl == 35;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!W8_ABL_NOMATH) accZ[n][e] = hsig(accZ[n][e]);
            }
            if constexpr (sl == 37) {
                if (SAVE == SAVE_ALL) {
                    pinu(lane16);
                    *reinterpret_cast<g_u16x8*>(acts_p[0] + lane16) = cat8(pack4(accZ[0]), pack4(accZ[1]));
                }
            }
            if constexpr (sl >= W8_TA0 && sl < W8_TA0 + 3) h_stage(0, sl - W8_TA0);        // tile a (its last MFMA: slot 39)
            if constexpr (sl == W8_TA0 + 3) *reinterpret_cast<u16x4*>(hnext + hw0) = pack4(hreg[0]);
            if constexpr (sl == 47) {
                W8_STAMP(10);
                w8_barrier();                                             // ---- 2a: the a half of h_t is complete in hnext
                W8_STAMP(11);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (!W8_ABL_NOB) bh[ks] = *reinterpret_cast<const frag*>(hnext + bf4[ks]);
                // the next step's r inputs: the first requests of the previous step
                if (XMODE != MVAE_X_CONST) vm_wait<2 * V_STEP - V_SV - V_RQ>();
                xpin(XNX, 1);
                accR[0] = unpack4(xv(XNX, 0, 1));
                accR[1] = unpack4(xv(XNX, 1, 1));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        hh_prev_p = acts_p[2];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acts_p[g] += acts_step;
            if (XMODE == MVAE_X_DENSE && t + 3 < T) x_p[g] += acts_step;
        }
        hs_p += hs_step;
        if (cs_steps && t == phi) {
            if (SAVE >= SAVE_HS && a.signal_done) wave_signal_done<false>(uniform_ptr(a.signal_done + pk));
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
    // ---- the deferred half of the last step ----------------------------------------------------------------------------
    unsigned char* hfin = hbuf + (T & 1) * 8192;
    asm volatile("s_nop 9" : "+v"(accC[1]));
#pragma unroll
    for (int st = 0; st < 3; ++st) h_stage(1, st);
    *reinterpret_cast<u16x4*>(hfin + hw0 + 256) = pack4(hreg[1]);
    if (SAVE == SAVE_ALL) *reinterpret_cast<g_u16x8*>(hh_prev_p + lane16) = cat8(pack4(accC[0]), pack4(accC[1]));
    if (a.h_last) {
#pragma unroll
        for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub0 + 128 * n) = hreg[n];
    }
    lds_barrier();
    if (SAVE >= SAVE_HS) {       // slot T = h_{T-1}
        store16_wt(hs_p, tg0, *reinterpret_cast<const u16x8*>(hfin + tl0));
        if (cs_steps && a.signal_done) wave_signal_done<false>(uniform_ptr(a.signal_done + pk));
    }
    vm_drain();
    // The last steps' requests are never consumed.  Their destination registers must stay allocated until the data has
    // landed: a dead asm output is a free register to hipcc, and the load would arrive in whatever value was put there next.
    if (XMODE != MVAE_X_CONST) {
#pragma unroll
        for (int g = 0; g < G; ++g) { xpin(0, g); xpin(1, g); }
        if (XMODE == MVAE_X_INDEX) pini(i_q);
    }
}

// ---------------------------------------------------------------------------------------------------------
// GRU backward through time, two waves per SIMD
// ---------------------------------------------------------------------------------------------------------
// Per step (t from T-1 down), per unit of this wave's tiles a = w and b = 8 + w, with d = dh_t + the upstream gradient:
//     da_c = d (1-z)(1-hh^2)     da_z = d (h_{t-1} - hh) hs'(z)          -> da tile (LDS), exchange 1
//     M1   drh = U_c^T da_c                      (16 MFMA slots, k-groups 16..23 of the packed transposed kernel)
//     M2z  acc = U_z^T da_z                      (16 slots, k-groups 0..7: nothing of it waits for M1 - its slots carry E2)
//     E2   da_r = drh h_{t-1} hs'(r) -> da tile, exchange 2;   part = d z + drh r
//     M2r  acc += U_r^T da_r                     (16 slots, k-groups 8..15);   dh_{t-1} = acc + part
// What does not depend on dh - unpacking the saved values, (1-z)(1-hh^2), (h_{t-1} - hh) hs'(z) of the NEXT step - runs in
// the M2r slots; only d, two products per element and the exchange itself are exposed between two steps.
// Outputs as the 4-wave kernel: da (T,B,3H) and rh = r h_{t-1} (T,B,H) row-major (operands of the parameter-gradient GEMMs):
// assembled in LDS, one 16-byte chunk per lane, gate and step.
__host__ __device__ constexpr bool gbw_is_l(int s) { return s % 3 == 2; }
__host__ __device__ constexpr int gbw_lidx(int s) { return s / 3; }
// slot s: unit tile n, k-group ks (of the 24 k-groups over the gate columns [z | r | candidate])
struct gbw_slot { int n, ks; };
__host__ __device__ constexpr gbw_slot gru_bslot(int s) {
    if (s < 16) return {s & 1, 16 + (s >> 1)};            // M1
    if (s < 32) return {s & 1, (s - 16) >> 1};            // M2z
    return {s & 1, 8 + ((s - 32) >> 1)};                  // M2r
}

template <bool HAS_EXT>
__device__ __forceinline__ void gru_bwd_w8_body(const mvae_rnn_bwd_args& a, const unsigned bx) {
    constexpr int G = 3, GH = G * RH, S2 = GH / 32, NLDS = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* dabuf = smem;                                              // [16][GH] bf16, swizzled (24 KiB)
    unsigned char* rhbuf = smem + 16 * GH * 2;                                // [16][RH]
    frag* ulds = reinterpret_cast<frag*>(smem + 16 * GH * 2 + 16 * RH * 2);   // [8][NLDS][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.ut_pack);
    frag* myl = ulds + (size_t)w * NLDS * 64 + l;
    auto src_frag = [&](int n, int ks) -> const frag* { return up + (size_t)((n * 8 + w) * S2 + ks) * 64 + l; };
    frag ua[32];
    static_for<0, 48>(SF_LAMBDA(sc) {
        constexpr int s = decltype(sc)::value;
        constexpr gbw_slot sb = gru_bslot(s);
        if constexpr (!gbw_is_l(s)) load1_agpr_nowait(ua[s - gbw_lidx(s)], src_frag(sb.n, sb.ks));
    });
    {
        frag tmp[NLDS];
        static_for<0, 48>(SF_LAMBDA(sc) {
            constexpr int s = decltype(sc)::value;
            constexpr gbw_slot sb = gru_bslot(s);
            if constexpr (gbw_is_l(s)) tmp[gbw_lidx(s)] = *src_frag(sb.n, sb.ks);
        });
#pragma unroll
        for (int i = 0; i < NLDS; ++i) myl[(size_t)i * 64] = tmp[i];
    }

    const int ub0 = w * 16 + q * 4;
    unsigned lane8 = (unsigned)l * 8u, lane16 = (unsigned)l * 16u;
    // da tile (row stride 1536 B): this lane's 4 values of (gate g, tile n) at da_w0 + g * 512 + n * 256;  B fragment of
    // k-group ks at b_row + (b_ch ^ (ks << 6))
    unsigned da_w0 = (unsigned)r * (GH * 2) + ((((unsigned)w * 2u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    unsigned b_row = (unsigned)r * (GH * 2), b_ch = ((unsigned)q ^ (unsigned)r) << 4;
    unsigned rw0 = (unsigned)r * 512u + ((((unsigned)w * 2u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    // row-major copies: one 16-byte chunk per lane and gate (row 2w + l/32, chunk l%32 of the gate's 32), one of the rh tile
    const unsigned row0 = 2u * (unsigned)w + ((unsigned)l >> 5), ch0 = (unsigned)l & 31u;
    unsigned cl0 = row0 * (GH * 2) + ((ch0 ^ row0) << 4);        // da tile, gate g: cl0 + g * 512
    unsigned cg0 = row0 * (GH * 2) + ch0 * 16u;                  // global, gate g: + g * 512
    unsigned tl0 = row0 * 512u + ((ch0 ^ row0) << 4), tg0 = row0 * 512u + ch0 * 16u;
    unsigned hp_off = (unsigned)r * (RH * 2) + (unsigned)q * 8u;

    f32x4 dh[2];
    const int ldl = a.dh_last_ld ? a.dh_last_ld : RH;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        dh[n] = a.dh_last ? *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)b * ldl + ub0 + 128 * n) : z4;
    }
    gbyte *acts_p[G], *hs_p, *dx_p, *da_p, *rh_p;
    const size_t acts_step = tps * (GH / 32) * 1024, dx_step = tps * (RH / 16) * 512, hs_step = (size_t)B * RH * 2,
                 da_step = (size_t)B * GH * 2;
#pragma unroll
    for (int g = 0; g < G; ++g)
        acts_p[g] = to_global(a.acts) + (((size_t)(T - 1) * tps + bx) * (GH / 32) + g * (RH / 32) + w) * 1024;
    hs_p = to_global(a.hs) + ((size_t)(T - 1) * B + bx * 16) * (RH * 2) + w * 32;             // h_{t-1} = slot t
    dx_p = to_global(a.dhs_ext) + (((size_t)(T - 1) * tps + bx) * (RH / 16) + w) * 512;
    da_p = to_global(a.da) + ((size_t)(T - 1) * B + bx * 16) * (GH * 2);
    rh_p = to_global(a.rh) + ((size_t)(T - 1) * B + bx * 16) * (RH * 2);

    // pipelined stack bookkeeping, as the 4-wave kernels: chunk pk (first step plo) is the one being processed
    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = __builtin_amdgcn_readfirstlane(cs_steps ? (T - 1) / cs_steps : 0), plo = pk * cs_steps;
    if (HAS_EXT && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready + pk, wait_value, a.status, 2u);
    int pwait = (cs_steps && a.wait_ready && plo > 0) ? plo : -1;
    int psig = (cs_steps && a.signal_done) ? plo : -1;

    // saved values: z, r, hh as TILE16Q pairs (elements 0..3 tile a, 4..7 tile b); h_{t-1} (row-major) and the upstream gradient
    // (TILE16) per tile.  Un-tracked asm loads, hand-counted waits (as in the forward kernel).  Per step, in issue order:
    //   slot 0   [z hh h_a h_b dx_a dx_b] of step t-1: z, hh, h consumed from slot 36 on (factors of the two products that wait
    //            for dh; the same arithmetic in the M1 slots of the next step, fed two steps ahead: 2.00 / 2.18 instead of 1.87 /
    //            2.08 us), the upstream gradient at the next E1 - a whole step after its request
    //   slot 23  [r] of step t-1 (this step's is used until slot 21): consumed from the next step's slot 16 on
    //   slots 26, 29, 33, 35  the copy stores (da tile by gate, rh tile) - BEHIND the requests: vmcnt retires in order, a load
    //   behind a write-through store waits for that store's acknowledgement
    u16x8 qz, qh, qr;
    u16x4 qp[2], qd[2], hpk[2];
    auto issue_early = [&](size_t back_a, size_t back_h) __attribute__((always_inline)) {
        pinu(lane16); pinu(hp_off);
        xload16(qz, acts_p[0] - back_a, lane16);
        xload16(qh, acts_p[2] - back_a, lane16);
        xload8(qp[0], hs_p - back_h, hp_off);
        xload8(qp[1], hs_p - back_h + 256, hp_off);
        if (HAS_EXT) {
            pinu(lane8);
            xload8(qd[0], dx_p, lane8);
            xload8(qd[1], dx_p + 8 * 512, lane8);
        }
    };
    auto issue_late = [&]() __attribute__((always_inline)) {
        pinu(lane16);
        xload16(qr, acts_p[1], lane16);
    };
    issue_early(0, 0);
    issue_late();
    vm_drain();
    lds_barrier();

    auto lo4 = [&](const u16x8& v) __attribute__((always_inline)) { return unpack4(__builtin_shufflevector(v, v, 0, 1, 2, 3)); };
    auto hi4 = [&](const u16x8& v) __attribute__((always_inline)) { return unpack4(__builtin_shufflevector(v, v, 4, 5, 6, 7)); };
    // 0.2 [0 < y < 1] for a saved hard_sigmoid output
    auto dhs = [&](float y) __attribute__((always_inline)) { return __builtin_amdgcn_fmed3f((y - y * y) * 0x1p100f, 0.0f, 0.2f); };
    // per-step factors of the products that wait for dh: w1 = (1-z)(1-hh^2), kz = (h_{t-1} - hh) hs'(z), z; in 5 pieces per tile
    f32x4 w1[2], kz[2], zv[2], t_hh, t_hp;
    auto precompute = [&](int n, int piece) __attribute__((always_inline)) {
        if (piece == 0) {
            zv[n] = n ? hi4(qz) : lo4(qz);
            t_hh = n ? hi4(qh) : lo4(qh);
        } else if (piece == 1) {
            t_hp = unpack4(qp[n]);
            hpk[n] = qp[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) w1[n][e] = __builtin_fmaf(-t_hh[e], t_hh[e], 1.0f);
        } else if (piece == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) w1[n][e] *= 1.0f - zv[n][e];
        } else if (piece == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) kz[n][e] = (zv[n][e] - zv[n][e] * zv[n][e]) * 0x1p100f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) kz[n][e] = (t_hp[e] - t_hh[e]) * __builtin_amdgcn_fmed3f(kz[n][e], 0.0f, 0.2f);
        }
    };
    pin8(qz); pin8(qh); pin1(qp[0]); pin1(qp[1]);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int pc = 0; pc < 5; ++pc) precompute(n, pc);

    frag bq[3], lt[2], cp;
    f32x4 acc1[2], acc2[2], part[2], d[2], e_rv, e_hp;
    constexpr int V_EARLY = HAS_EXT ? 6 : 4;
    acc2[0] = dh[0];
    acc2[1] = dh[1];
    auto step = [&](const int t) __attribute__((always_inline)) {
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(hs_p); pins(da_p);
        if (HAS_EXT) pins(dx_p);
        if (a.rh) pins(rh_p);
        if (HAS_EXT) wave_wait_ge_if(t, __builtin_amdgcn_readfirstlane(pwait), uniform_ptr(a.wait_ready + (pk - 1)), wait_value, a.status, 2u);
        // ---- E1 (exposed): d = dh (+ the upstream gradient), da_c = d w1, da_z = d kz -----------------------------------
        // the upstream gradient of this step: the previous step's slot 0; behind it the r request and the copy stores
        if (HAS_EXT) {
            if (a.rh) vm_wait<5>(); else vm_wait<4>();
            pin1(qd[0]); pin1(qd[1]);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            d[n] = acc2[n];
            if (HAS_EXT) {
                const f32x4 dx = unpack4(qd[n]);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[n][e] += dx[e];
            }
            f32x4 dac, daz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dac[e] = d[n][e] * w1[n][e];
                if (!W8_BWD_Z_LATE) daz[e] = d[n][e] * kz[n][e];
            }
            *reinterpret_cast<u16x4*>(dabuf + (da_w0 + (2 * 512 + n * 256))) = pack4(dac);
            if (!W8_BWD_Z_LATE) *reinterpret_cast<u16x4*>(dabuf + (da_w0 + (0 * 512 + n * 256))) = pack4(daz);
        }
        w8_barrier();                                                     // ---- 1: da_c, da_z
        bq[0] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (16u << 6)));
        bq[1] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (17u << 6)));
        lt[0] = myl[0];
        lt[1] = myl[64];
        acc1[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_nop 1" : "+v"(acc1[0]), "+v"(acc1[1]));
        static_for<0, 48>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value;
            constexpr gbw_slot sb = gru_bslot(sl);
            constexpr int li = gbw_lidx(sl), kq = sl >> 1;               // kq: running k-group count 0..23 in order of use
            f32x4& acc = sl < 16 ? acc1[sb.n] : acc2[sb.n];
            // B ring of 3: k-group kq + 2 is requested when kq starts (M2r's first two behind barrier 2)
            if constexpr ((sl & 1) == 0 && kq + 2 < 16) {
                constexpr int kn = kq + 2, ksn = kn < 8 ? 16 + kn : kn - 8;
                bq[kn % 3] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ ((unsigned)ksn << 6)));
            }
            if constexpr ((sl & 1) == 0 && kq >= 16 && kq + 2 < 24) {
                constexpr int kn = kq + 2;
                bq[kn % 3] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ ((unsigned)(kn - 8) << 6)));
            }
            if constexpr (gbw_is_l(sl) && W8_ABL_NOL) {
                mfma1<true>(acc, ua[li], bq[kq % 3]);
            } else if constexpr (gbw_is_l(sl)) {
                mfma1<false>(acc, lt[li & 1], bq[kq % 3]);
                lt[li & 1] = myl[(size_t)((li + 2) & 15) * 64];
            } else {
                mfma1<true>(acc, ua[sl - li], bq[kq % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- fillers ---------------------------------------------------------------------------------------------
            // (requests are unconditional - at t = 0 step 0's values once more: a request under a branch is an asm output merged
            //  with the old value behind it, i.e. a register copy of data that has not landed)
            if constexpr (sl == 0) issue_early(0, 0);                     // (acts_p / hs_p / dx_p were moved to step t-1 above)
            if constexpr (W8_BWD_Z_LATE && (sl == 2 || sl == 4)) {
                constexpr int n = sl == 4;
                f32x4 daz;
#pragma unroll
                for (int e = 0; e < 4; ++e) daz[e] = d[n][e] * kz[n][e];
                *reinterpret_cast<u16x4*>(dabuf + (da_w0 + (0 * 512 + n * 256))) = pack4(daz);
            }
            if constexpr (W8_BWD_Z_LATE && sl == 9) w8_barrier();          // ---- 1b: da_z (its first B fragment is requested in slot 12)
            // M2's accumulators start at d z
            if constexpr (sl == 10 || sl == 12) {
                constexpr int n = sl == 12;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[n][e] = d[n][e] * zv[n][e];
            }
            // E2 (M1's last MFMAs: slots 14, 15): r, h_{t-1}: r h -> rh tile;  da_r = drh h hs'(r) -> da tile;  part = drh r
            if constexpr (sl == 16) {        // r: the previous step's slot 23; behind it its copy stores and this step's slot 0
                if (a.rh) vm_wait<4 + V_EARLY>(); else vm_wait<3 + V_EARLY>();
                pin8(qr);
            }
            if constexpr (sl == 16 || sl == 19) {
                constexpr int n = sl == 19;
                e_rv = n ? hi4(qr) : lo4(qr);
                e_hp = unpack4(hpk[n]);
                f32x4 p;
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = e_rv[e] * e_hp[e];
                if (a.rh) *reinterpret_cast<u16x4*>(rhbuf + rw0 + 256 * n) = pack4(p);
            }
            if constexpr (sl == 17 || sl == 20) {
#pragma unroll
                for (int e = 0; e < 4; ++e) e_hp[e] *= dhs(e_rv[e]);
            }
            if constexpr (sl == 18 || sl == 21) {
                constexpr int n = sl == 21;
                f32x4 dar;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dar[e] = acc1[n][e] * e_hp[e];
                    part[n][e] = acc1[n][e] * e_rv[e];
                }
                *reinterpret_cast<u16x4*>(dabuf + (da_w0 + (1 * 512 + n * 256))) = pack4(dar);
            }
            if constexpr (sl == 23) issue_late();
            // copies of the da tile's candidate and z columns (final since barrier 1)
            if constexpr (sl == 24 || sl == 27) {
                constexpr int g = sl == 24 ? 2 : 0;
                cp = *reinterpret_cast<const frag*>(dabuf + (cl0 + g * 512));
            }
            if constexpr (sl == 26 || sl == 29) {
                constexpr int g = sl == 26 ? 2 : 0;
                pinu(cg0);
                store16_wt(da_p, cg0 + g * 512, cp);
            }
            if constexpr (sl == 31) {
                w8_barrier();                                             // ---- 2: da_r (and the rh tile)
                bq[16 % 3] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (8u << 6)));
                bq[17 % 3] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (9u << 6)));
                cp = *reinterpret_cast<const frag*>(dabuf + (cl0 + 512u));
                // drh r joins the accumulation here, where the wave has just waited anyway (M2z's last MFMA: slot 31)
                asm volatile("s_nop 9" : "+v"(acc2[0]), "+v"(acc2[1]));
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[n][e] += part[n][e];
                asm volatile("s_nop 1" : "+v"(acc2[0]), "+v"(acc2[1]));
            }
            if constexpr (sl == 33) {
                pinu(cg0);
                store16_wt(da_p, cg0 + 512, cp);
                if (a.rh) cp = *reinterpret_cast<const frag*>(rhbuf + tl0);
            }
            if constexpr (sl == 35) {
                if (a.rh) {
                    pinu(tg0);
                    store16_wt(rh_p, tg0, cp);
                }
            }
            // the next step's factors: its z, hh, h (slot 0) have retired when only what was issued behind them is outstanding
            if constexpr (sl == 36) {
                if (a.rh) vm_wait<1 + 4>(); else vm_wait<1 + 3>();
                pin8(qz); pin8(qh); pin1(qp[0]); pin1(qp[1]);
            }
            if constexpr (sl >= 37 && sl < 47) precompute((sl - 37) / 5, (sl - 37) % 5);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(acc2[0]), "+v"(acc2[1]));
        da_p -= da_step;
        if (a.rh) rh_p -= hs_step;
        // pipelined stack: da of steps >= t is out; chunk t / cs is complete when t is its first step
        wave_signal_done_if<false>(t, __builtin_amdgcn_readfirstlane(psig), uniform_ptr(a.signal_done + pk));
        {
            const bool adv = cs_steps && t == plo;
            pk -= adv ? 1 : 0;
            plo -= adv ? cs_steps : 0;
            pwait = (a.wait_ready && plo > 0) ? plo : -1;
            psig = a.signal_done ? plo : -1;
        }
    };
    for (int t = T - 1; t >= 0; --t) {
        // pointers of the saved values move to step t-1 before step t's MFMA phases request them
#pragma unroll
        for (int g = 0; g < G; ++g) acts_p[g] -= (t > 0 ? acts_step : 0);
        hs_p -= (t > 0 ? hs_step : 0);
        if (HAS_EXT) dx_p -= (t > 0 ? dx_step : 0);
        step(t);
    }
    const int ldd = a.dh0_ld ? a.dh0_ld : RH;
#pragma unroll
    for (int n = 0; n < 2; ++n)
        if (a.dh0) *reinterpret_cast<f32x4*>(a.dh0 + (size_t)b * ldd + ub0 + 128 * n) = acc2[n];
    vm_drain();
    pin8(qz); pin8(qh); pin8(qr); pin1(qp[0]); pin1(qp[1]);
    if (HAS_EXT) { pin1(qd[0]); pin1(qd[1]); }
}
template <bool HAS_EXT>
__global__ __launch_bounds__(512, 1) void gru_bwd_w8_k(const mvae_rnn_bwd_args a) {
    gru_bwd_w8_body<HAS_EXT>(a, blockIdx.x);
}
template <bool HAS_EXT>
int launch_gru_bwd_w8(const mvae_rnn_bwd_args& a, hipStream_t s) {
    const size_t lds = (size_t)16 * 3 * RH * sizeof(bf16_t) + (size_t)16 * RH * sizeof(bf16_t) + (size_t)8 * 16 * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_w8_k<HAS_EXT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((gru_bwd_w8_k<HAS_EXT>), dim3(a.B / 16), dim3(512), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// LSTM backward through time, two waves per SIMD (the dominant kernel of a training step)
// ---------------------------------------------------------------------------------------------------------
// Same data as the 4-wave kernel lstm_bwd_il_k (seq_layout MVAE_TILE16P: a drop-in for it, the forward kernels stay as they are):
// wave w owns the unit tiles 2w and 2w + 1 - pair w of every TILE16P array.  Per step (t from T-1 down), with d = dh_t + upstream:
//     E   dO = d tanh(c_t) hs'(o);   dct = dc + d o (1 - tanh(c_t)^2);   di = dct g hs'(i);   df = dct c_{t-1} hs'(f);
//         dg = dct i (1 - g^2);   dc_{t-1} = dct f                                          -> da tile (LDS), barrier
//     M   dh_{t-1} = U^T-fragments x [di | df | dg | dO]: 64 MFMA slots per wave (32 k-groups x 2 unit tiles)
// U^T is 512 KiB - a CU's whole register file: a wave keeps 32 of its 64 fragments in accumulator registers, 16 in LDS, and
// STREAMS the other 16 from L2 every step through a ring of 4 (tools/probes/l2stream_probe.hip: 128 KiB per CU and step beside
// 128 MFMAs per SIMD cost 0.10 us per step on 16 ... 128 CUs).  The factors of d that do not depend on the recurrence -
// tanh(c_t) hs'(o) and o (1 - tanh(c_t)^2) of the NEXT step - are computed in the M slots; what E leaves exposed is the
// arithmetic on dct.  All memory instructions of a step are issued in a fixed order (table below) and waited for by count.
__host__ __device__ constexpr int lbw_class(int s) { return (s & 3) < 2 ? 0 : ((s & 3) == 2 ? 1 : 2); }      // 0 AGPR, 1 LDS, 2 streamed
// Memory instructions issued in the fillers of M slot s (at most one KIND per slot; within slot 0 the four requests in order):
//   streamed fragments, consumed in slots c = 4j + 3: an L2 round trip beside the LDS and store traffic of a step takes ~700 cycles =
//   ~22 MFMA slots (a ring of 4, every request 16 slots ahead, ran 3.5 us per step against 2.07 without the stream), so
//     c = 3 .. 15   are requested in slots 51 .. 63 of the PREVIOUS step (the E phase in between)  -> ring[0..3]
//     c = 19 .. 31  together in slot 0, 18 .. 30 slots ahead                                       -> ring[4..7]
//     c = 35 .. 47  in slots 3 .. 15, as ring[0..3] are consumed;  c = 51 .. 63 in slots 19 .. 31 as ring[4..7] are
//   the next step's requests: i f g o c_{t-2} in slots 1 .. 17, the upstream tiles in 21, 25
//   the 4 copy stores of the da tile in slots 4, 6, 8, 10 (their LDS reads two slots earlier, right behind the barrier: the next E
//   phase of a faster wave must not find them unread); vmcnt retires in issue order and a write-through store is acknowledged
//   after ~0.3 us, which every request behind them can afford: none is consumed less than 24 slots later
__host__ __device__ constexpr int lbw_vm_ops(int s, bool ext) {
    if (s == 0) return 4;
    if ((s & 3) == 3 && (s <= 31 || s >= 51)) return 1;
    if ((s & 3) == 1 && s < (ext ? 28 : 20)) return 1;
    if (s >= 4 && s <= 10 && (s & 1) == 0) return 1;
    return 0;
}
__host__ __device__ constexpr int lbw_vm_before(int s, bool ext) {      // issued in slots [0, s)
    int n = 0;
    for (int i = 0; i < s; ++i) n += lbw_vm_ops(i, ext);
    return n;
}
// memory instructions younger than the request of the streamed fragment that slot c (c % 4 == 3) uses, when slot c starts
__host__ __device__ constexpr int lbw_ring_younger(int c, bool ext) {
    if (c < 16) return lbw_vm_before(64, ext) - lbw_vm_before(c + 49, ext) + lbw_vm_before(c, ext);
    if (c < 32) return 3 - (c - 19) / 4 + lbw_vm_before(c, ext) - lbw_vm_before(1, ext);
    return lbw_vm_before(c, ext) - lbw_vm_before(c - 31, ext);
}
// ring entry of the fragment consumed in slot c
__host__ __device__ constexpr int lbw_ring_entry(int c) { return ((c >> 4) & 1) * 4 + ((c >> 2) & 3); }

template <bool HAS_EXT>
__device__ __forceinline__ void lstm_bwd_w8_body(const mvae_rnn_bwd_args& a, const unsigned bx) {
    constexpr int G = 4, GH = G * RH, S2 = GH / 32, NLDS = 16, NSTR = 16;
    constexpr int V_STEP = lbw_vm_before(64, HAS_EXT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* dabuf = smem;                                              // [16][GH] bf16, swizzled (32 KiB)
    frag* ulds = reinterpret_cast<frag*>(smem + 16 * GH * 2);                 // [8][NLDS][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.ut_pack);
    frag* myl = ulds + (size_t)w * NLDS * 64 + l;
    // slot s: k-group s / 2 (over the gate columns [i | f | g | o]), unit tile 2w + s % 2
    auto src_frag = [&](int s) -> const frag* { return up + (size_t)((2 * w + (s & 1)) * S2 + (s >> 1)) * 64 + l; };
    frag ua[32];
    static_for<0, 64>(SF_LAMBDA(sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (lbw_class(s) == 0) load1_agpr_nowait(ua[(s >> 2) * 2 + (s & 1)], src_frag(s));
    });
    {
        frag tmp[NLDS];
        static_for<0, 64>(SF_LAMBDA(sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (lbw_class(s) == 1) tmp[s >> 2] = *src_frag(s);
        });
#pragma unroll
        for (int i = 0; i < NLDS; ++i) myl[(size_t)i * 64] = tmp[i];
    }
    // the streamed fragments (slots 4j + 3): global byte offsets relative to this wave's first one are compile-time constants
    const gbyte* strm = to_global(up + (size_t)((2 * w + 1) * S2 + 1) * 64);       // (wave-uniform; + lane * 16)

    const int ub0 = w * 32 + q * 4;
    unsigned lane8 = (unsigned)l * 8u, lane16 = (unsigned)l * 16u;
    // da tile (row stride 2048 B): this lane's 4 values of (gate g, tile n) at (da_w0 ^ (n << 5)) + g * 512; B fragment of k-group
    // ks at b_row + (b_ch ^ (ks << 6))
    unsigned da_w0 = (unsigned)r * (GH * 2) + ((((unsigned)w * 4u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    unsigned b_row = (unsigned)r * (GH * 2), b_ch = ((unsigned)q ^ (unsigned)r) << 4;
    // row-major copy of the da tile: one 16-byte chunk per lane and gate (row 2w + l/32, chunk l%32 of the gate's 32)
    const unsigned row0 = 2u * (unsigned)w + ((unsigned)l >> 5), ch0 = (unsigned)l & 31u;
    unsigned cl0 = row0 * (GH * 2) + ((ch0 ^ row0) << 4), cg0 = row0 * (GH * 2) + ch0 * 16u;

    f32x4 dhv[2], dc[2];
    const int ldl = a.dh_last_ld ? a.dh_last_ld : RH;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        dhv[n] = a.dh_last ? *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)b * ldl + ub0 + 16 * n) : z4;
        dc[n] = a.dc_last ? *reinterpret_cast<const f32x4*>(a.dc_last + (size_t)b * ldl + ub0 + 16 * n) : z4;
    }
    gbyte *acts_p[G], *cs_p, *dx_p, *da_p;
    const size_t acts_step = tps * (GH / 16) * 512, cs_step = tps * (RH / 16) * 512, dx_step = cs_step, da_step = (size_t)B * GH * 2;
#pragma unroll
    for (int g = 0; g < G; ++g)
        acts_p[g] = to_global(a.acts) + (((size_t)(T - 1) * tps + bx) * (GH / 32) + g * (RH / 32) + w) * 1024;
    cs_p = to_global(a.cs) + (((size_t)(T - 1) * tps + bx) * (RH / 32) + w) * 1024;        // c_{t-1} of step T-1 (slot T-1)
    dx_p = to_global(a.dhs_ext) + (((size_t)(T - 1) * tps + bx) * (RH / 16) + 2 * w) * 512;
    da_p = to_global(a.da) + ((size_t)(T - 1) * B + bx * 16) * (GH * 2);

    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = __builtin_amdgcn_readfirstlane(cs_steps ? (T - 1) / cs_steps : 0), plo = pk * cs_steps;
    if (HAS_EXT && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready + pk, wait_value, a.status, 2u);
    int pwait = (cs_steps && a.wait_ready && plo > 0) ? plo : -1;
    int psig = (cs_steps && a.signal_done) ? plo : -1;

    // saved values as TILE16P pairs (elements 0..3 tile 2w, 4..7 tile 2w + 1): gates i f g o, c_{t-1}; `carry` = c_t (the previous
    // step's c_{t-1}); the upstream gradient per tile (TILE16)
    u16x8 qa[G], qs, carry;
    u16x4 qd[2];
#pragma unroll
    for (int g = 0; g < G; ++g) qa[g] = *reinterpret_cast<const g_u16x8*>(acts_p[g] + lane16);
    qs = *reinterpret_cast<const g_u16x8*>(cs_p + lane16);
    carry = *reinterpret_cast<const g_u16x8*>(cs_p + cs_step + lane16);
    if (HAS_EXT) {
        qd[0] = *reinterpret_cast<const g_u16x4*>(dx_p + lane8);
        qd[1] = *reinterpret_cast<const g_u16x4*>(dx_p + 512 + lane8);
    }
    // the ring of streamed fragments: slots 3, 7, 11, 15 of the first step
    frag ring[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) ring[j] = *reinterpret_cast<const g_u16x8*>(strm + (size_t)j * 2 * 64 * 16 + lane16);
    vm_drain();
    lds_barrier();
#pragma unroll
    for (int g = 0; g < G; ++g) acts_p[g] -= (T > 1 ? acts_step : 0);
    cs_p -= (T > 1 ? cs_step : 0);
    dx_p -= (T > 1 ? dx_step : 0);

    auto lo4 = [&](const u16x8& v) __attribute__((always_inline)) { return unpack4(__builtin_shufflevector(v, v, 0, 1, 2, 3)); };
    auto hi4 = [&](const u16x8& v) __attribute__((always_inline)) { return unpack4(__builtin_shufflevector(v, v, 4, 5, 6, 7)); };
    auto dhs = [&](float y) __attribute__((always_inline)) { return __builtin_amdgcn_fmed3f((y - y * y) * 0x1p100f, 0.0f, 0.2f); };
    // factors of d of the step about to be processed: ko = tanh(c_t) hs'(o), kc = o (1 - tanh(c_t)^2); in 4 pieces per tile
    f32x4 ko[2], kc[2], t_c;
    constexpr float K2 = 2.8853900817779268f;
    auto precompute = [&](int n, int piece) __attribute__((always_inline)) {
        if (W8_ABL_NOMATH) return;
        if (piece == 0) {
            t_c = n ? hi4(carry) : lo4(carry);
#pragma unroll
            for (int e = 0; e < 4; ++e) t_c[e] = __builtin_amdgcn_exp2f(t_c[e] * K2);
        } else if (piece == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) t_c[e] = __builtin_fmaf(__builtin_amdgcn_rcpf(t_c[e] + 1.0f), -2.0f, 1.0f);      // tanh(c_t)
        } else if (piece == 2) {
            const f32x4 o = n ? hi4(qa[3]) : lo4(qa[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kc[n][e] = o[e] * __builtin_fmaf(-t_c[e], t_c[e], 1.0f);
                ko[n][e] = (o[e] - o[e] * o[e]) * 0x1p100f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) ko[n][e] = t_c[e] * __builtin_amdgcn_fmed3f(ko[n][e], 0.0f, 0.2f);
        }
    };
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) precompute(n, pc);

    frag bq[3], lt, cp;          // (one staging register for the LDS-resident fragments: requested 4 slots before its use)
    f32x4 acc[2];
    acc[0] = dhv[0];
    acc[1] = dhv[1];
    auto step = [&](const int t) __attribute__((always_inline)) {
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(acts_p[3]); pins(cs_p); pins(da_p);
        if (HAS_EXT) pins(dx_p);
        if (HAS_EXT) wave_wait_ge_if(t, __builtin_amdgcn_readfirstlane(pwait), uniform_ptr(a.wait_ready + (pk - 1)), wait_value, a.status, 2u);
        // ---- E (exposed) ------------------------------------------------------------------------------------------------
        // this step's i, f, g, c_{t-1} and upstream gradient: requests 0..2, 4, 6.. of the previous step's M phase; behind the last of
        // them (slot 25) that phase issued what lbw_vm_before(64) - lbw_vm_before(26) counts
        vm_wait<V_STEP - lbw_vm_before(HAS_EXT ? 26 : 18, HAS_EXT)>();
        pin8(qa[0]); pin8(qa[1]); pin8(qa[2]); pin8(qs);
        if (HAS_EXT) { pin1(qd[0]); pin1(qd[1]); }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f32x4 ig = n ? hi4(qa[0]) : lo4(qa[0]), fg = n ? hi4(qa[1]) : lo4(qa[1]), gg = n ? hi4(qa[2]) : lo4(qa[2]);
            const f32x4 cpv = n ? hi4(qs) : lo4(qs);
            f32x4 d = acc[n], di, df, dg, dO;
            if (HAS_EXT) {
                const f32x4 dx = unpack4(qd[n]);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] += dx[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (W8_ABL_NOMATH) { di[e] = df[e] = dg[e] = dO[e] = d[e]; continue; }
                const float dct = __builtin_fmaf(d[e], kc[n][e], dc[n][e]);
                dO[e] = d[e] * ko[n][e];
                di[e] = dct * gg[e] * dhs(ig[e]);
                df[e] = dct * cpv[e] * dhs(fg[e]);
                dg[e] = dct * ig[e] * __builtin_fmaf(-gg[e], gg[e], 1.0f);
                dc[n][e] = dct * fg[e];
            }
            unsigned char* dan = dabuf + (da_w0 ^ (unsigned)(n << 5));
            *reinterpret_cast<u16x4*>(dan + 0 * 512) = pack4(di);
            *reinterpret_cast<u16x4*>(dan + 1 * 512) = pack4(df);
            *reinterpret_cast<u16x4*>(dan + 2 * 512) = pack4(dg);
            *reinterpret_cast<u16x4*>(dan + 3 * 512) = pack4(dO);
        }
        carry = qs;                                       // c_{t-1} is the next step's c_t
        w8_barrier();                                     // ---- the da tile is complete
        bq[0] = *reinterpret_cast<const frag*>(dabuf + b_row + b_ch);
        bq[1] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (1u << 6)));
        lt = myl[0];
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1]));
        static_for<0, 64>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value, kq = sl >> 1, n = sl & 1, j4 = sl >> 2;
            if constexpr (n == 0 && kq + 2 < 32)
                bq[(kq + 2) % 3] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ ((unsigned)(kq + 2) << 6)));
            if constexpr (lbw_class(sl) == 0) {
                mfma1<true>(acc[n], ua[j4 * 2 + n], bq[kq % 3]);
            } else if constexpr (lbw_class(sl) == 1 && W8_ABL_NOL) {
                mfma1<true>(acc[n], ua[j4 * 2], bq[kq % 3]);
            } else if constexpr (lbw_class(sl) == 1) {
                mfma1<false>(acc[n], lt, bq[kq % 3]);
                lt = myl[(size_t)((j4 + 1) & 15) * 64];
            } else if constexpr (W8_ABL_NOSTREAM) {
                mfma1<true>(acc[n], ua[j4 * 2 + 1], bq[kq % 3]);
            } else {
                // the streamed fragment of this slot was requested 16 slots ago (4 ring entries); behind it: what those slots issued
                vm_wait<lbw_ring_younger(sl, HAS_EXT)>();
                pin8(ring[lbw_ring_entry(sl)]);
                mfma1<false>(acc[n], ring[lbw_ring_entry(sl)], bq[kq % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- fillers (memory instructions exactly as lbw_vm_ops lists them, in this order) ---------------------------------
            // streamed fragments (lbw_vm_ops): fragment j sits 2j fragments behind the first one
            if constexpr (sl == 0 && !W8_ABL_NOSTREAM) {
                pinu(lane16);
                static_for<4, 8>(SF_LAMBDA(jc) {
                    constexpr int j = decltype(jc)::value;
                    xload16(ring[lbw_ring_entry(4 * j + 3)], strm + (size_t)(j * 2) * 64 * 16, lane16);
                });
            }
            if constexpr ((sl & 3) == 3 && sl <= 31 && !W8_ABL_NOSTREAM) {          // for slot sl + 32, into the entry just consumed
                pinu(lane16);
                xload16(ring[lbw_ring_entry(sl + 32)], strm + (size_t)(((sl + 32) >> 2) * 2) * 64 * 16, lane16);
            }
            if constexpr ((sl & 3) == 3 && sl >= 51 && !W8_ABL_NOSTREAM) {          // for slot sl - 48 of the next step
                pinu(lane16);
                xload16(ring[lbw_ring_entry(sl - 48)], strm + (size_t)(((sl - 48) >> 2) * 2) * 64 * 16, lane16);
            }
            // (W8_ABL_NOLOADS: the same instructions on ONE hot line - they stay in the count, their latency is an L2 hit's)
            if constexpr ((sl & 3) == 1 && sl < 16) { pinu(lane16); xload16(qa[sl >> 2], W8_ABL_NOLOADS ? strm : acts_p[sl >> 2], lane16); }     // i f g o of step t-1
            if constexpr (sl == 17) { pinu(lane16); xload16(qs, W8_ABL_NOLOADS ? strm : cs_p, lane16); }                                       // c_{t-2}
            if constexpr (sl == 21 && HAS_EXT) { pinu(lane8); xload8(qd[0], W8_ABL_NOLOADS ? strm : dx_p, lane8); }
            if constexpr (sl == 25 && HAS_EXT) { pinu(lane8); xload8(qd[1], W8_ABL_NOLOADS ? strm : dx_p + 512, lane8); }
            // the next step's factors: tanh(c_t) from `carry` (registers), o from request 3 (slot 13)
            // (slots 48 .. 55: four of the eight ring entries are dead between slot 47 and the next step - the registers the factors use)
            if constexpr (sl == 47) { vm_wait<lbw_vm_before(48, HAS_EXT) - lbw_vm_before(14, HAS_EXT)>(); pin8(qa[3]); }
            if constexpr (sl >= 48 && sl < 56) precompute((sl - 48) >> 2, (sl - 48) & 3);
            // row-major copy of the da tile (final since the barrier): gate (sl - 43) / 4
            if constexpr (sl >= 4 && sl <= 10 && (sl & 1) == 0) {
                pinu(cg0);
                store16_wt(da_p, cg0 + ((sl - 4) >> 1) * 512, cp);
            }
            if constexpr (sl >= 2 && sl <= 8 && (sl & 1) == 0) cp = *reinterpret_cast<const frag*>(dabuf + (cl0 + ((sl - 2) >> 1) * 512));
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(acc[0]), "+v"(acc[1]));
        da_p -= da_step;
        wave_signal_done_if<false>(t, __builtin_amdgcn_readfirstlane(psig), uniform_ptr(a.signal_done + pk));
        {
            const bool adv = cs_steps && t == plo;
            pk -= adv ? 1 : 0;
            plo -= adv ? cs_steps : 0;
            pwait = (a.wait_ready && plo > 0) ? plo : -1;
            psig = a.signal_done ? plo : -1;
        }
    };
    for (int t = T - 1; t >= 0; --t) {
        step(t);
#pragma unroll
        for (int g = 0; g < G; ++g) acts_p[g] -= (t > 1 ? acts_step : 0);
        cs_p -= (t > 1 ? cs_step : 0);
        if (HAS_EXT) dx_p -= (t > 1 ? dx_step : 0);
    }
    const int ldd = a.dh0_ld ? a.dh0_ld : RH;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        if (a.dh0) *reinterpret_cast<f32x4*>(a.dh0 + (size_t)b * ldd + ub0 + 16 * n) = acc[n];
        if (a.dc0) *reinterpret_cast<f32x4*>(a.dc0 + (size_t)b * ldd + ub0 + 16 * n) = dc[n];
    }
    vm_drain();
    pin8(qa[0]); pin8(qa[1]); pin8(qa[2]); pin8(qa[3]); pin8(qs);
    if (HAS_EXT) { pin1(qd[0]); pin1(qd[1]); }
    pin8(ring[0]); pin8(ring[1]); pin8(ring[2]); pin8(ring[3]); pin8(ring[4]); pin8(ring[5]); pin8(ring[6]); pin8(ring[7]);
}
template <bool HAS_EXT>
__global__ __launch_bounds__(512, 1) void lstm_bwd_w8_k(const mvae_rnn_bwd_args a) {
    lstm_bwd_w8_body<HAS_EXT>(a, blockIdx.x);
}
template <bool HAS_EXT>
int launch_lstm_bwd_w8(const mvae_rnn_bwd_args& a, hipStream_t s) {
    const size_t lds = (size_t)16 * 4 * RH * sizeof(bf16_t) + (size_t)8 * 16 * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_w8_k<HAS_EXT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((lstm_bwd_w8_k<HAS_EXT>), dim3(a.B / 16), dim3(512), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

constexpr int GRU_W8_NLDS = 16;
template <int XMODE, int SAVE>
__global__ __launch_bounds__(512, 1) void gru_fwd_w8_k(const mvae_rnn_fwd_args a) {
    gru_fwd_w8_body<XMODE, SAVE>(a, blockIdx.x);
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
    if (XMODE == MVAE_X_INDEX && a.table_layout != MVAE_TABLE_PAIRED8) return MVAE_E_ARG;
    if (a.acts) {
        if (!a.hs) return MVAE_E_UNSUPPORTED;
        return launch_gru_w8<XMODE, SAVE_ALL>(a, s);
    }
    if (a.cs) return MVAE_E_UNSUPPORTED;
    return a.hs ? launch_gru_w8<XMODE, SAVE_HS>(a, s) : launch_gru_w8<XMODE, SAVE_NONE>(a, s);
}

// ---- phase launches (as rnn_resident.hip's: every recurrence of a phase as ONE launch), 8 waves per workgroup -------------
template <int SAVE>
__global__ __launch_bounds__(512, 1) void gru_fwd_multi_w8_k(const rnn_fwd_multi m) {
    const int bid = (int)blockIdx.x, nx = m.nx;
    int i = 0;
    while (i + 1 < nx + m.n && bid >= m.base[i + 1]) ++i;
    const unsigned bx = (unsigned)(bid - m.base[i]);
    if (i < nx) {
        xpand_body<8>(m.xp[i], (int)bx, m.base[i + 1] - m.base[i]);
        return;
    }
    const mvae_rnn_fwd_args a = m.p[i - nx];       // (a copy: the fields then live in scalar registers for the whole launch)
    const int xm = __builtin_amdgcn_readfirstlane(a.xmode);
    if (xm == MVAE_X_DENSE) gru_fwd_w8_body<MVAE_X_DENSE, SAVE>(a, bx);
    else if (xm == MVAE_X_INDEX) gru_fwd_w8_body<MVAE_X_INDEX, SAVE>(a, bx);
    else gru_fwd_w8_body<MVAE_X_CONST, SAVE>(a, bx);
}
__global__ __launch_bounds__(512, 1) void gru_bwd_multi_w8_k(const rnn_bwd_multi m) {
    const int bid = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < m.n && bid >= m.base[i + 1]) ++i;
    const unsigned bx = (unsigned)(bid - m.base[i]);
    const mvae_rnn_bwd_args a = m.p[i];
    const bool ext = __builtin_amdgcn_readfirstlane(a.dhs_ext != nullptr);
    if (ext) gru_bwd_w8_body<true>(a, bx);
    else gru_bwd_w8_body<false>(a, bx);
}
template <int SAVE>
int launch_fwd_multi_w8(const rnn_fwd_multi& m, int total, hipStream_t s) {
    const size_t lds = (size_t)3 * 16 * RH * sizeof(bf16_t) + (size_t)8 * GRU_W8_NLDS * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_multi_w8_k<SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((gru_fwd_multi_w8_k<SAVE>), dim3(total), dim3(512), lds, s, m);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

}  // namespace

int mvae_rnn_bwd_w8(const mvae_rnn_bwd_args& a, hipStream_t s) {
    if (a.H != RH || a.dtype != MVAE_BF16 || (a.B % 16) != 0) return MVAE_E_UNSUPPORTED;
    if (a.cell == MVAE_LSTM && a.seq_layout == MVAE_TILE16P) {       // (the data of lstm_bwd_il_k: a drop-in for it)
        if (!a.cs) return MVAE_E_ARG;
        return a.dhs_ext ? launch_lstm_bwd_w8<true>(a, s) : launch_lstm_bwd_w8<false>(a, s);
    }
    if (a.seq_layout != MVAE_TILE16Q || a.cell != MVAE_GRU) return MVAE_E_UNSUPPORTED;
    return a.dhs_ext ? launch_gru_bwd_w8<true>(a, s) : launch_gru_bwd_w8<false>(a, s);
}
// Entry points used by rnn_resident.hip's dispatch.  MVAE_E_UNSUPPORTED: not a shape of this file.
int mvae_rnn_fwd_w8(const mvae_rnn_fwd_args& a, hipStream_t s) {
    if (a.H != RH || a.dtype != MVAE_BF16 || (a.B % 16) != 0 || a.seq_layout != MVAE_TILE16Q || a.cell != MVAE_GRU)
        return MVAE_E_UNSUPPORTED;
    switch (a.xmode) {
        case MVAE_X_DENSE: return a.xp ? gru_w8_save<MVAE_X_DENSE>(a, s) : MVAE_E_ARG;
        case MVAE_X_INDEX: return (a.idx && a.table) ? gru_w8_save<MVAE_X_INDEX>(a, s) : MVAE_E_ARG;
        case MVAE_X_CONST: return a.xp0 ? gru_w8_save<MVAE_X_CONST>(a, s) : MVAE_E_ARG;
    }
    return MVAE_E_UNSUPPORTED;
}

// (see rnn_resident.hip's mvae_rnn_fwd_multi / mvae_rnn_bwd_multi, which route seq_layout MVAE_TILE16Q here)
int mvae_rnn_fwd_multi_w8(const mvae_rnn_fwd_args* problems, int32_t n, const mvae_xpand_args* xpand, int32_t n_xpand, void* stream) {
    rnn_fwd_multi m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    m.nx = n_xpand;
    int total = 0;
    for (int i = 0; i < n_xpand; ++i) {
        const mvae_xpand_args& x = xpand[i];
        if (!((x.xs && x.w && x.bias) || (x.idx && x.table)) || !x.out || !x.chunk_done || x.out_kind != MVAE_BF16 || x.R <= 0 || x.N != 3 * RH ||
            x.chunk_rows <= 0 || (x.chunk_rows % 16) || (x.R % x.chunk_rows) || x.blocks <= 0 || x.blocks > 256 ||
            (!x.idx && ((reinterpret_cast<uintptr_t>(x.w) & 15) || (reinterpret_cast<uintptr_t>(x.bias) & 15))) ||
            (x.idx && ((reinterpret_cast<uintptr_t>(x.table) & 7) || (reinterpret_cast<uintptr_t>(x.idx) & 3))) ||
            (size_t)x.chunk_rows * x.N * 2 > 0x7fffffffull)
            return MVAE_E_ARG;
        m.xp[i] = x;
        m.base[i] = total;
        total += x.blocks;
    }
    const mvae_rnn_fwd_args& f = problems[0];
    for (int i = 0; i < n; ++i) {
        const mvae_rnn_fwd_args& a = problems[i];
        if (!a.u_pack || a.T <= 0 || a.B <= 0 || (a.B % 16) || a.chunk_steps < 0 ||
            ((a.wait_ready || a.signal_done) && a.chunk_steps == 0) || (a.wait_ready && a.xmode != MVAE_X_DENSE) ||
            (a.signal_done && !a.hs))
            return MVAE_E_ARG;
        if (a.H != RH || a.dtype != MVAE_BF16 || a.seq_layout != MVAE_TILE16Q || a.cell != MVAE_GRU || a.xmode == MVAE_X_SCALAR || !a.hs ||
            (a.acts != nullptr) != (f.acts != nullptr) || a.cs)
            return MVAE_E_UNSUPPORTED;
        if ((a.xmode == MVAE_X_DENSE && !a.xp) || (a.xmode == MVAE_X_INDEX && !(a.idx && a.table)) || (a.xmode == MVAE_X_CONST && !a.xp0))
            return MVAE_E_ARG;
        if (a.xmode == MVAE_X_INDEX && a.table_layout != MVAE_TABLE_PAIRED8) return MVAE_E_ARG;
        m.p[i] = a;
        m.base[n_xpand + i] = total;
        total += a.B / 16;
    }
    m.base[n_xpand + n] = total;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return f.acts ? launch_fwd_multi_w8<SAVE_ALL>(m, total, s) : launch_fwd_multi_w8<SAVE_HS>(m, total, s);
}
int mvae_rnn_bwd_multi_w8(const mvae_rnn_bwd_args* problems, int32_t n, void* stream) {
    rnn_bwd_multi m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const mvae_rnn_bwd_args& a = problems[i];
        if (!a.ut_pack || !a.hs || !a.acts || !a.da || a.T <= 0 || a.B <= 0 || (a.B % 16) || a.chunk_steps < 0 ||
            ((a.wait_ready || a.signal_done) && a.chunk_steps == 0) || (a.wait_ready && !a.dhs_ext))
            return MVAE_E_ARG;
        if (a.H != RH || a.dtype != MVAE_BF16 || a.seq_layout != MVAE_TILE16Q || a.cell != MVAE_GRU) return MVAE_E_UNSUPPORTED;
        m.p[i] = a;
        m.base[i] = total;
        total += a.B / 16;
    }
    m.base[n] = total;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)16 * 3 * RH * sizeof(bf16_t) + (size_t)16 * RH * sizeof(bf16_t) + (size_t)8 * 16 * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_multi_w8_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL(gru_bwd_multi_w8_k, dim3(total), dim3(512), lds, s, m);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
