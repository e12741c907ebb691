// Resident-weights recurrent kernels for the production shape (H = 256, bf16 MFMA operands) on gfx950.
//
// Same contract and numerics as rnn.hip (mvae_rnn_fwd / mvae_rnn_bwd dispatch here when H == 256,
// dtype == MVAE_BF16, B % 16 == 0 and the cell is GRU or LSTM); different data placement and schedule:
//
//  * The recurrent kernel U (256 x G*256 bf16 = 384 KiB GRU / 512 KiB LSTM) is loaded ONCE per launch and stays on
//    chip for all T steps, split between the register file and LDS of the one CU that owns 16 batch rows.  A CU has
//    4 SIMDs x 512 registers x 64 lanes x 4 B = 512 KiB of registers and 160 KiB of LDS; the workgroup is 4 waves
//    (one per SIMD, launch_bounds(256,1)) so each wave owns the full 512-register budget.  Wave w owns hidden units
//    [64w, 64w+64) for all gates.  Its MFMA A-fragments (1 KiB each) live, by index in order of use, in
//        [0,NA)        accumulator registers - loaded there by inline asm and read in place by v_mfma (hipcc on its
//                      own parks such values in AGPRs but copies them back with v_accvgpr_read before every MFMA),
//        [NA,NA+NV)    vector registers,
//        [NA+NV, ...)  a private LDS slab read back with ds_read_b128.
//    Nothing is re-read from L2/HBM inside the time loop except x_t and (backward) the saved activations.
//
//  * Memory ordering.  hipcc treats vmcnt as out-of-order once loads and stores are both in flight and then waits
//    for (almost) vmcnt(0) at every use of a loaded value: with per-tile prefetch loads next to per-tile stores
//    that exposed a store round trip per tile (measured 2 us of a 3.3 us step).  Here every per-step load is issued
//    ONE STEP ahead and all of a step's loaded registers are pinned at ONE point per step - placed where the
//    youngest outstanding store is oldest - behind one explicit s_waitcnt vmcnt(0).  Workgroup barriers wait for
//    LDS only (lds_barrier), never for global memory.
//
//  * HBM access shape.  Per-lane 8-byte accesses on row-major (T,B,*) arrays make every wave instruction touch 16
//    rows at a power-of-two stride (measured: 12 loads + 16 stores = 6000 of a step's 12000 cycles).  Everything
//    these kernels stream per step therefore uses the TILE16 layout (include/midivae_hip.h): xp, the saved gates,
//    the saved cell states and the upstream gradient are read / written as 512 contiguous bytes per wave access.
//    Row-major OUTPUTS consumed by the GEMMs (h sequence, da, r*h) are first assembled in LDS - where the step
//    needs them anyway - and written back as whole 512-byte row segments.
//
//  * Forward LSTM schedule: the gate arithmetic of unit tile n-1 is issued among the MFMA groups of tile n (the
//    matrix pipe runs a 4-MFMA group for 64 cycles), so only the last tile's arithmetic and the barrier are
//    exposed.  Backward has a true dependency (all of dh before any gate gradient) and stays phased.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

// Phase stamps (build with -DRES_STAMPS): block 0 / wave 0 / lane 0 records s_memtime at marked points of steps
// [64, 72); read back with mvae_debug_stamps().  Development tooling only.
#ifdef RES_STAMPS
__device__ unsigned long long mvae_stamps[8 * 16];
#define STAMP(k)                                                                                     \
    do {                                                                                             \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tstep >= 64 && tstep < 72)                        \
            mvae_stamps[(tstep - 64) * 16 + (k)] = __builtin_readcyclecounter();                     \
    } while (0)
extern "C" int mvae_debug_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mvae_stamps), sizeof(mvae_stamps)) == hipSuccess ? 0 : -3;
}
#else
#define STAMP(k)
#endif

#include "ablations.h"      // ABL_* timing switches: all 0 in the product build (variant builds only: tools/build_variants.sh)

namespace {

constexpr int RH = 256;           // hidden size this file is specialised for
constexpr int RS = RH / 32;       // k-groups of the forward contraction (8)
constexpr int RNT = 4;            // unit tiles (16 units) per wave

typedef u16x8 frag;

// ---- inline-asm building blocks ---------------------------------------------------------------------------
// Four MFMAs sharing one B fragment.  hipcc neither pads hazards inside an asm statement nor models the MFMA:
// the leading s_nop 1 covers "VALU-written VGPR -> MFMA operand", chain_done() covers "MFMA result -> VALU reader"
// (4-pass XDL: 8 wait states required; 10 given).
template <bool AG>
__device__ __forceinline__ void mfma4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const frag& u0, const frag& u1,
                                      const frag& u2, const frag& u3, const frag& b) {
    if (AG)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
                     "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                     : "a"(u0), "a"(u1), "a"(u2), "a"(u3), "v"(b));
    else
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
                     "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %8, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                     : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(b));
}
__device__ __forceinline__ void chain_done(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    asm volatile("s_nop 9" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
}
// 4 fragments straight into accumulator registers, waited for in the same statement
__device__ __forceinline__ void load4_agpr(frag& u0, frag& u1, frag& u2, frag& u3, const frag* p0, const frag* p1,
                                           const frag* p2, const frag* p3) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                 : "=&a"(u0), "=&a"(u1), "=&a"(u2), "=&a"(u3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}
// Same without the wait: the caller issues every group, then ONE vm_drain() before the first MFMA (a wait per group
// serialises 16 HBM/L2 round trips = tens of microseconds per launch).  ONLY for kernels without scratch: hipcc may
// spill an asm output right after the statement, i.e. before the data has landed.  The Makefile fails the build if a
// kernel named *_il_k has a non-zero scratch size.
__device__ __forceinline__ void load4_agpr_nowait(frag& u0, frag& u1, frag& u2, frag& u3, const frag* p0, const frag* p1,
                                                  const frag* p2, const frag* p3) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"
                 : "=&a"(u0), "=&a"(u1), "=&a"(u2), "=&a"(u3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}
// Prefetch loads (plain, so hipcc may spill / move them safely).  All values of a step are "consumed" together by
// the pin*() statements right after ONE vm_drain(): hipcc then has nothing left to wait for at the individual
// uses, instead of inserting a (store-draining) wait per tile.
__device__ __forceinline__ void aload8(u16x4& d, const void* p) { d = *reinterpret_cast<const u16x4*>(p); }
__device__ __forceinline__ void aload4(float& d, const void* p) { d = *reinterpret_cast<const float*>(p); }
__device__ __forceinline__ void aload1(int& d, const void* p) { d = *reinterpret_cast<const uint8_t*>(p); }
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ties asm-loaded registers to the drain: consumers can only be scheduled after this statement
__device__ __forceinline__ void pin4(u16x4& a, u16x4& b, u16x4& c, u16x4& d) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// LSTM BPTT filler slots (tools/build_variants.sh for A/B; profiles/r03_r_bptt_slots.txt): the write-through da stores as LATE as
// the staging registers allow - slots 16.. cost +0.33 ms per train step, 44 +0.03, 50 (round 2) 0, 58 -0.05: stores in flight slow
// the return of the next step's prefetched values more than their acknowledgement is missed at the T-fragment drain
#ifndef BWL_LOAD_STRIDE
#define BWL_LOAD_STRIDE 3
#endif
#ifndef BWL_COPY_SLOT
#define BWL_COPY_SLOT 58
#endif
#ifndef BWL_OLD_DHS
#define BWL_OLD_DHS 0
#endif
#ifndef BWL_COPY_STRIDE
#define BWL_COPY_STRIDE 4
#endif
#ifndef GB_DRAIN
#define GB_DRAIN 0
#endif
// a wave-uniform pointer the compiler has lost track of (state captured by a step lambda), back in scalar registers
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
#ifndef GRU_NVB
#define GRU_NVB 24     // candidate fragments of the GRU forward kernel held in vector registers (of 32; the rest in LDS)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int E>
__device__ __forceinline__ f32x2 lo_hi(const f32x4& v) { return __builtin_shufflevector(v, v, E, E + 1); }
// 4-element forms of tanh_fast / dhard_sigmoid (common.h): the same operations per element, written on vectors so that
// the multiplies / adds / fmas become packed instructions
__device__ __forceinline__ f32x4 tanh_fast4(f32x4 x) {
    const f32x4 t = x * 2.8853900817779268f;
    const f32x4 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    const f32x4 s1 = e + 1.0f;
    const f32x4 rc = {__builtin_amdgcn_rcpf(s1[0]), __builtin_amdgcn_rcpf(s1[1]), __builtin_amdgcn_rcpf(s1[2]), __builtin_amdgcn_rcpf(s1[3])};
    return 1.0f - 2.0f * rc;
}
__device__ __forceinline__ f32x4 dhard_sigmoid4(f32x4 y) {
    const f32x4 sq = (y - y * y) * 0x1p100f;
    return f32x4{__builtin_amdgcn_fmed3f(sq[0], 0.0f, 0.2f), __builtin_amdgcn_fmed3f(sq[1], 0.0f, 0.2f),
                 __builtin_amdgcn_fmed3f(sq[2], 0.0f, 0.2f), __builtin_amdgcn_fmed3f(sq[3], 0.0f, 0.2f)};
}
// The derivative of hard_sigmoid at a saved gate value y in [0, 1] is 0.2 where 0 < y < 1.  sat4(y) = 0.2 * 2^-100 there, else 0:
// y - y^2 is 0 exactly at the two clipped values and >= 2^-26 anywhere else a bf16 hard_sigmoid output can be, so one clamp
// against a tiny constant selects (round 4: the factor 2^100 that makes it 0.2 rides on the OTHER factor of the product - one
// multiply per element for all gates - and the negation is an operand modifier instead of the two v_xor hipcc emits).
// bf16 -> f32 on the MATRIX pipe (round 4).  In the gate-gradient phase of the backward kernels no MFMA is in flight and the one
// wave per SIMD is VALU-bound; a quarter of its instructions only widen saved bf16 values (v_lshlrev / v_and per element).
// v_mfma_f32_16x16x16_bf16 with the 16x16 IDENTITY as A returns its B operand widened: lane (q, r) holds B[k = 4q + i][n = r] and
// receives C[m = 4q + i][n = r] = sum_k I[m][k] B[k][r] = its own four elements, exactly (1.0 * x, fifteen zero products), plus the
// accumulator operand - one issue slot for four elements and an addition.  (A non-finite value anywhere in the 16 lanes of a
// column would spread as 0 * inf = NaN; saved activations are finite, or the step is lost already.)
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 identity_fragment(int l) {
    const int q = l >> 4, r = l & 15, j = r - 4 * q;
    return s16x4{(short)(j == 0 ? 0x3F80 : 0), (short)(j == 1 ? 0x3F80 : 0), (short)(j == 2 ? 0x3F80 : 0), (short)(j == 3 ? 0x3F80 : 0)};
}
__device__ __forceinline__ f32x4 widen4(s16x4 ident, u16x4 packed, f32x4 plus) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ident, __builtin_bit_cast(s16x4, packed), plus, 0, 0, 0);
}
#ifndef FWL_MFMA_WIDEN
#define FWL_MFMA_WIDEN 0     /* LSTM forward, single-launch inference variants: tiles 1..3 start their accumulators at x through widen4 (48 VALU per step less) - measured SLOWER (const input, no saves: 2.01 vs 1.94 us per step: 12 more MFMAs on a pipe the step is already waiting for); the training variants (252 VGPRs) and the phase launches spill with it */
#endif
#ifndef BWL_ASM_1MSQ
#define BWL_ASM_1MSQ 1
#endif
#ifndef BWL_MFMA_FIRST
#define BWL_MFMA_FIRST 0      /* C = 0 inline in the first MFMA of each chain instead of zeroed accumulators: 8 v_mov_b64 less, but 2.78 vs 2.71 us per step (profiles/r04_h_bptt_ab.txt) */
#endif
#ifndef RES_WGMAJOR
#define RES_WGMAJOR 0
#endif
#ifndef BWL_E_TILE_FENCE
#define BWL_E_TILE_FENCE 0
#endif
#ifndef BWL_MFMA_WIDEN
#define BWL_MFMA_WIDEN 0      /* 1: the upstream gradient only, 2: every saved value - both spill in the LSTM kernel (254 VGPRs + 256 AGPRs without them): profiles/r04_d_bptt_e_phase.txt */
#endif
#define DHS_TINY 0x1.99999ap-103f      /* 0.2f * 2^-100 */
#define DHS_BIG 0x1p100f
__device__ __forceinline__ f32x2 y_minus_y2(f32x2 y) {
    f32x2 t;
    asm("v_pk_fma_f32 %0, %1, %1, %1 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(t) : "v"(y));
    return t;
}
// 1 - y^2 for two elements with the negation as an operand modifier (hipcc: two v_xor + v_pk_fma)
__device__ __forceinline__ f32x2 one_minus_sq(f32x2 y) {
    f32x2 t;
    asm("v_pk_fma_f32 %0, %1, %1, 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(t) : "v"(y));
    return t;
}
__device__ __forceinline__ f32x4 one_minus_sq4(f32x4 y) {
    const f32x2 a = one_minus_sq(lo_hi<0>(y)), b = one_minus_sq(lo_hi<2>(y));
    return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ f32x4 sat4(f32x4 y) {
    const f32x2 a = y_minus_y2(lo_hi<0>(y)), b = y_minus_y2(lo_hi<2>(y));
    return f32x4{__builtin_amdgcn_fmed3f(a[0], 0.0f, DHS_TINY), __builtin_amdgcn_fmed3f(a[1], 0.0f, DHS_TINY),
                 __builtin_amdgcn_fmed3f(b[0], 0.0f, DHS_TINY), __builtin_amdgcn_fmed3f(b[1], 0.0f, DHS_TINY)};
}
__device__ __forceinline__ u16x8 cat8(u16x4 a, u16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ void pin1(u16x4& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pinf(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pini(int& a) { asm volatile("" : "+v"(a)); }

__device__ __forceinline__ f32x4 unpack4(u16x4 p) { return f32x4{bf2f(p[0]), bf2f(p[1]), bf2f(p[2]), bf2f(p[3])}; }
__device__ __forceinline__ u16x4 pack4(f32x4 v) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    return __builtin_bit_cast(u16x4, __builtin_convertvector(v, bf16x4));     // 2 x v_cvt_pk_bf16_f32
}

// ---- fragment residency -----------------------------------------------------------------------------------
#define RES_DECLARE_U(FPW_)                                                                                        \
    static_assert(NA % 4 == 0 && NV % 4 == 0 && NA + NV <= (FPW_) && NA <= 64, "fragment classes");              \
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
        } else {                                                                                                  \
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
        else if (ABL_NOL) mfma4<true>(c0, c1, c2, c3, ua[0], ua[1], ua[2], ua[3], bf);                            \
        else {                                                                                                    \
            const frag t0 = myl[(size_t)((f) - NA - NV) * 64], t1 = myl[(size_t)((f) - NA - NV + 1) * 64];        \
            const frag t2 = myl[(size_t)((f) - NA - NV + 2) * 64], t3 = myl[(size_t)((f) - NA - NV + 3) * 64];    \
            mfma4<false>(c0, c1, c2, c3, t0, t1, t2, t3, bf);                                                     \
        }                                                                                                         \
    } while (0)

// One MFMA phase: NG groups of 4 MFMAs, group gi uses fragments F0 + 4*gi .. +3 and B fragment BFRAG(gi).
// B fragments come from LDS through a 3-deep register ring: the read for group gi+2 is issued before the MFMAs of
// group gi, so the ~100-cycle ds_read latency is covered by two 64-cycle groups (hipcc does not hoist loads over
// volatile asm on its own: measured 164 instead of 64 cycles per group).  HOOK(gi) runs after group gi's MFMAs.
#define RES_PHASE(c0, c1, c2, c3, F0, NG, BFRAG, HOOK)                                                             \
    do {                                                                                                          \
        frag bq_[3];                                                                                              \
        bq_[0] = BFRAG(0);                                                                                        \
        bq_[1] = BFRAG((NG) > 1 ? 1 : 0);                                                                         \
        _Pragma("unroll") for (int gi_ = 0; gi_ < (NG); ++gi_) {                                                  \
            if (gi_ + 2 < (NG)) bq_[(gi_ + 2) % 3] = BFRAG(gi_ + 2);                                              \
            RES_MFMA4(c0, c1, c2, c3, (F0) + 4 * gi_, bq_[gi_ % 3]);                                              \
            HOOK(gi_);                                                                                            \
        }                                                                                                         \
        chain_done(c0, c1, c2, c3);                                                                               \
    } while (0)
#define RES_NOHOOK(gi)

enum { SAVE_NONE = 0, SAVE_HS = 1, SAVE_ALL = 2 };
__device__ __forceinline__ int t_next2(int t, int T) { return t + 2 < T ? t + 2 : T - 1; }
// Wave skew (round 4): the four waves of a workgroup leave every barrier together and run the same instruction stream, so their
// memory instructions reach the CU's one address unit (and their LDS accesses the LDS) in the same cycles and queue behind each
// other.  Wave w idles RES_WSKEW x 16 x w cycles behind the barrier that opens a memory-heavy phase: the streams stay de-phased
// until the next barrier (A/B: profiles/r04_b_wave_skew.txt).
#ifndef RES_WSKEW
#define RES_WSKEW 0
#endif
__device__ __forceinline__ void res_skew(int w) {
#if RES_WSKEW > 0
    for (int i = 0; i < w; ++i) {
#pragma unroll
        for (int j = 0; j < RES_WSKEW; ++j) asm volatile("s_nop 15" ::: "memory");
    }
#endif
}
__device__ __forceinline__ void res_barrier() {
    if (ABL_NOBAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else lds_barrier();
}

// [16 rows][W] bf16 tile in LDS with NO padding; the 16-byte chunk index is XOR-ed with the row so the 16 rows of a
// ds_read_b128 group hit 16 different bank slots.  col % 4 == 0.
template <int W>
__device__ __forceinline__ int sw_off(int row, int col) {
    return row * W + ((((col >> 3) ^ row) << 3) | (col & 7));
}
// Write rows [4w, 4w+4) of a swizzled LDS tile to a row-major global array as whole 16-byte chunks (coalesced).
template <int W>
__device__ __forceinline__ void tile_rows_to_global(const bf16_t* tile, bf16_t* gbase /* row 0 of this WG's 16 rows */,
                                                    int w, int l) {
    constexpr int CH = W / 8;                    // 16-byte chunks per row
    if (ABL_NOTRG) return;
#pragma unroll 2
    for (int j = 0; j < (4 * CH) / 64; ++j) {
        const int c = j * 64 + l, row = 4 * w + c / CH, ch = c % CH;
        const u16x8 v = *reinterpret_cast<const u16x8*>(tile + row * W + ((ch ^ row) << 3));
        *reinterpret_cast<u16x8*>(gbase + (size_t)row * W + ch * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int CELL, int XMODE, int SAVE, int NA, int NV>
__global__ __launch_bounds__(256, 1) void rnn_fwd_res_k(const mvae_rnn_fwd_args a) {
    constexpr int G = mvae_gates(CELL), GH = G * RH;
    constexpr int FPW = G * RNT * RS;              // fragments per wave (LSTM 128, GRU 96)
    constexpr int NLc = FPW - NA - NV;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* hbuf = reinterpret_cast<bf16_t*>(smem);                         // [2][16][RH] swizzled
    bf16_t* rhbuf = hbuf + 2 * 16 * RH;                                     // [16][RH] swizzled   (GRU)
    frag* ulds = reinterpret_cast<frag*>(rhbuf + (CELL == MVAE_GRU ? 16 * RH : 0));     // [4][NL][64]
    float* wb = reinterpret_cast<float*>(ulds + 4 * NLc * 64);              // [2][GH]         (SCALAR)
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: addresses become SGPR base + lane offset
    const int T = a.T, B = a.B;
    const int b = blockIdx.x * 16 + r;               // B % 16 == 0: every row is real
    const size_t tiles_per_step = (size_t)(B / 16);  // TILE16: row-tile index of (t, this WG) = t*B/16 + blockIdx.x
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.u_pack);
    bf16_t* __restrict__ hs = reinterpret_cast<bf16_t*>(a.hs);
    bf16_t* __restrict__ cs = reinterpret_cast<bf16_t*>(a.cs);
    bf16_t* __restrict__ acts = reinterpret_cast<bf16_t*>(a.acts);
    const bf16_t* __restrict__ xp = reinterpret_cast<const bf16_t*>(a.xp);
    const bf16_t* __restrict__ table = reinterpret_cast<const bf16_t*>(a.table);
    frag* myl = ulds + (size_t)w * NLc * 64 + l;

    // fragment f of this wave, in order of use.  LSTM: f = (n*8 + ks)*4 + g  (tile n, k-group ks, gate g).
    // GRU: phase A (z,r) f = ((np*8 + ks)*2 + nn)*2 + g for tile pair np; phase B (candidate) f = 64 + ks*4 + n.
    auto frag_src = [&](int f) -> int {
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
        hreg[n] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * ld0 + ub[n]) : z4;
        creg[n] = (CELL == MVAE_LSTM && a.c0) ? *reinterpret_cast<const f32x4*>(a.c0 + (size_t)b * ld0 + ub[n]) : z4;
        *reinterpret_cast<u16x4*>(hbuf + sw_off<RH>(r, ub[n])) = pack4(hreg[n]);
        if (CELL == MVAE_LSTM && SAVE == SAVE_ALL)     // c_0 -> tile row 0 of the (T+1, B, H) TILE16 array
            *reinterpret_cast<u16x4*>(cs + (((size_t)blockIdx.x * (RH / 16) + w * RNT + n) * 64 + l) * 4) = pack4(creg[n]);
    }
    if (XMODE == MVAE_X_SCALAR) {
        for (int i = tid; i < GH; i += 256) {
            wb[i] = a.w_row[i];
            wb[GH + i] = a.bias[i];
        }
    }

    // ---- x queue: packed bf16x4 per (tile, gate) for the step about to be computed ---------------------------
    u16x4 xq[RNT][G];
    float xs_q = 0.0f;        // SCALAR: x of the step about to be computed
    int i_q = 0;              // INDEX: table row of the step AFTER the one about to be computed
    auto xaddr = [&](int t, int n, int g, int irow) -> const bf16_t* {
        if (XMODE == MVAE_X_DENSE)      // TILE16 (T*B, GH): 512 contiguous bytes per wave
            return xp + ((((size_t)t * tiles_per_step + blockIdx.x) * (GH / 16) + g * (RH / 16) + w * RNT + n) * 64 + l) * 4;
        if (XMODE == MVAE_X_INDEX) return table + (size_t)irow * GH + g * RH + ub[n];
        return reinterpret_cast<const bf16_t*>(a.xp0) + (size_t)b * GH + g * RH + ub[n];
    };
    if (XMODE != MVAE_X_SCALAR) {
        const int i0 = XMODE == MVAE_X_INDEX ? (int)a.idx[b] : 0;
#pragma unroll
        for (int n = 0; n < RNT; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g) xq[n][g] = *reinterpret_cast<const u16x4*>(xaddr(0, n, g, i0));
        if (XMODE == MVAE_X_INDEX) i_q = a.idx[(size_t)(T > 1 ? 1 : 0) * B + b];
    } else {
        xs_q = a.xs[b];
    }
    vm_drain();
    lds_barrier();

    float xs_cur = 0.0f;
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const int tstep = t;
        STAMP(0);
        const int tn = t + 1 < T ? t + 1 : t;             // step whose inputs are fetched during this step
        const bf16_t* htile = hbuf + cur * 16 * RH;       // h_{t-1} (slot t of the saved sequence)
        bf16_t* hnext = hbuf + (cur ^ 1) * 16 * RH;
        auto hfrag = [&](const bf16_t* tile, int ks) -> frag {
            return *reinterpret_cast<const frag*>(tile + sw_off<RH>(r, ks * 32 + q * 8));
        };
        // saved-sequence slot t (= h_{t-1}; slot 0 = h0): whole rows straight from the LDS tile
        if (SAVE >= SAVE_HS) tile_rows_to_global<RH>(htile, hs + ((size_t)t * B + blockIdx.x * 16) * RH, w, l);
        const size_t otile = ((size_t)t * tiles_per_step + blockIdx.x);          // TILE16 row tile of step t

        // x of THIS step for (tile n, gate g)
        auto xval = [&](int n, int g) -> f32x4 {
            if (XMODE == MVAE_X_SCALAR) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wb + g * RH + ub[n]);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(wb + GH + g * RH + ub[n]);
                return xs_cur * w4 + b4;
            }
            u16x4 p = xq[n][g];
            if (XMODE == MVAE_X_CONST) asm volatile("" : "+v"(p));   // keep the loop-invariant row PACKED (no LICM of the unpack)
            return unpack4(p);
        };
        // every prefetch load of this step was issued one step ago; the youngest pending store is the previous step's
        auto step_inputs_ready = [&]() {
            STAMP(1);
            vm_drain();
            STAMP(2);
            if (XMODE == MVAE_X_DENSE || XMODE == MVAE_X_INDEX) {
#pragma unroll
                for (int n = 0; n < RNT; ++n) {
                    if (G == 4) pin4(xq[n][0], xq[n][1], xq[n][G > 2 ? 2 : 0], xq[n][G > 3 ? 3 : 0]);
                    else { pin1(xq[n][0]); pin1(xq[n][G > 1 ? 1 : 0]); pin1(xq[n][G > 2 ? 2 : 0]); }
                }
            }
            if (XMODE == MVAE_X_SCALAR) { pinf(xs_q); xs_cur = xs_q; }
            if (XMODE == MVAE_X_INDEX) pini(i_q);
        };
        auto request_next = [&](int n) {      // right after tile n's x was consumed
            if (ABL_NOX) return;
            if (XMODE == MVAE_X_DENSE || XMODE == MVAE_X_INDEX) {
#pragma unroll
                for (int g = 0; g < G; ++g) aload8(xq[n][g], xaddr(tn, n, g, i_q));
            }
            if (n == RNT - 1) {
                if (XMODE == MVAE_X_SCALAR) aload4(xs_q, a.xs + (size_t)tn * B + b);
                if (XMODE == MVAE_X_INDEX) aload1(i_q, a.idx + (size_t)(t + 2 < T ? t + 2 : T - 1) * B + b);
            }
        };

        if (CELL == MVAE_LSTM) {
            auto lstm_tile = [&](int n, const f32x4& a0, const f32x4& a1, const f32x4& a2, const f32x4& a3) {
                const f32x4 xi = xval(n, 0), xf = xval(n, G > 1 ? 1 : 0), xg = xval(n, G > 2 ? 2 : 0), xo = xval(n, G > 3 ? 3 : 0);
                f32x4 ig, fg, gg, og, hnew;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (ABL_NOMATH) { ig[i] = a0[i] + xi[i]; fg[i] = a1[i] + xf[i]; gg[i] = a2[i] + xg[i]; og[i] = a3[i] + xo[i];
                                      creg[n][i] += ig[i]; hnew[i] = og[i] + fg[i] + gg[i]; continue; }
                    ig[i] = hard_sigmoid(a0[i] + xi[i]);
                    fg[i] = hard_sigmoid(a1[i] + xf[i]);
                    gg[i] = tanh_fast(a2[i] + xg[i]);
                    og[i] = hard_sigmoid(a3[i] + xo[i]);
                    creg[n][i] = fg[i] * creg[n][i] + ig[i] * gg[i];
                    hnew[i] = og[i] * tanh_fast(creg[n][i]);
                }
                *reinterpret_cast<u16x4*>(hnext + sw_off<RH>(r, ub[n])) = pack4(hnew);
                if (SAVE == SAVE_ALL && !ABL_NOSAVE) {
                    bf16_t* ap = acts + ((otile * (GH / 16) + w * RNT + n) * 64 + l) * 4;     // gate 0 tile; gates 16 tiles apart
                    *reinterpret_cast<u16x4*>(ap) = pack4(ig);
                    *reinterpret_cast<u16x4*>(ap + 1 * (RH / 16) * 256) = pack4(fg);
                    *reinterpret_cast<u16x4*>(ap + 2 * (RH / 16) * 256) = pack4(gg);
                    *reinterpret_cast<u16x4*>(ap + 3 * (RH / 16) * 256) = pack4(og);
                    *reinterpret_cast<u16x4*>(cs + (((otile + tiles_per_step) * (RH / 16) + w * RNT + n) * 64 + l) * 4) = pack4(creg[n]);
                }
                if (t == T - 1) {
                    if (a.h_last) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub[n]) = hnew;
                    if (a.c_last) *reinterpret_cast<f32x4*>(a.c_last + (size_t)b * ldl + ub[n]) = creg[n];
                }
            };
#if RES_LSTM_PIPELINE
            // software pipeline over unit tiles: MFMAs of tile n with the arithmetic of tile n-1 issued among them
            f32x4 accA[4], accB[4];
#pragma unroll
            for (int n = 0; n <= RNT; ++n) {
                f32x4* acc = (n & 1) ? accB : accA;
                f32x4* prev = (n & 1) ? accA : accB;
                if (n < RNT) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#define HF_(ks) hfrag(htile, ks)
#define LSTM_HOOK_(gi)                                                         \
    if ((gi) == 1 && n > 0) {                                                 \
        if (n == 1) step_inputs_ready();                                      \
        lstm_tile(n - 1, prev[0], prev[1], prev[2], prev[3]);                 \
        request_next(n - 1);                                                  \
    }
                    RES_PHASE(acc[0], acc[1], acc[2], acc[3], n * RS * 4, RS, HF_, LSTM_HOOK_);
                    STAMP(8 + n);
                } else {
                    lstm_tile(n - 1, prev[0], prev[1], prev[2], prev[3]);
                    request_next(n - 1);
                }
            }
#else
            // register-lean schedule: one accumulator set; the gate arithmetic of a tile follows its own MFMAs
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                f32x4 acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef HF_
#define HF_(ks) hfrag(htile, ks)
#endif
                RES_PHASE(acc[0], acc[1], acc[2], acc[3], n * RS * 4, RS, HF_, RES_NOHOOK);
                if (n == 0) step_inputs_ready();
                lstm_tile(n, acc[0], acc[1], acc[2], acc[3]);
                request_next(n);
            }
#endif
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
#ifndef HF_
#define HF_(ks) hfrag(htile, ks)
#endif
                RES_PHASE(acc[0][0], acc[0][1], acc[1][0], acc[1][1], np * RS * 4, RS, HF_, RES_NOHOOK);
                if (np == 0) step_inputs_ready();
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
                    const int n = np * 2 + nn;
                    const f32x4 xz = xval(n, 0), xr = xval(n, G > 1 ? 1 : 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        zg[n][i] = hard_sigmoid(acc[nn][0][i] + xz[i]);
                        rg[n][i] = hard_sigmoid(acc[nn][1][i] + xr[i]);
                    }
                    *reinterpret_cast<u16x4*>(rhbuf + sw_off<RH>(r, ub[n])) = pack4(rg[n] * hreg[n]);
                }
            }
            STAMP(3);
            res_barrier();
            STAMP(4);
            // ---- GRU phase B: candidate ---------------------------------------------------------------------
            f32x4 acc[RNT];
#pragma unroll
            for (int n = 0; n < RNT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#define RHF_(ks) hfrag(rhbuf, ks)
            RES_PHASE(acc[0], acc[1], acc[2], acc[3], 64, RS, RHF_, RES_NOHOOK);
            STAMP(5);
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                const f32x4 xh = xval(n, G > 2 ? 2 : 0);
                f32x4 hh, hnew;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hh[i] = tanh_fast(acc[n][i] + xh[i]);
                    hnew[i] = zg[n][i] * hreg[n][i] + (1.0f - zg[n][i]) * hh[i];
                }
                hreg[n] = hnew;
                *reinterpret_cast<u16x4*>(hnext + sw_off<RH>(r, ub[n])) = pack4(hnew);
                if (SAVE == SAVE_ALL && !ABL_NOSAVE) {
                    bf16_t* ap = acts + ((otile * (GH / 16) + w * RNT + n) * 64 + l) * 4;
                    *reinterpret_cast<u16x4*>(ap) = pack4(zg[n]);
                    *reinterpret_cast<u16x4*>(ap + 1 * (RH / 16) * 256) = pack4(rg[n]);
                    *reinterpret_cast<u16x4*>(ap + 2 * (RH / 16) * 256) = pack4(hh);
                }
                request_next(n);
            }
        }
        cur ^= 1;
        STAMP(6);
        res_barrier();
        STAMP(7);
    }
    if (SAVE >= SAVE_HS) tile_rows_to_global<RH>(hbuf + cur * 16 * RH, hs + ((size_t)T * B + blockIdx.x * 16) * RH, w, l);
    if (CELL == MVAE_GRU && a.h_last) {
#pragma unroll
        for (int n = 0; n < RNT; ++n) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub[n]) = hreg[n];
    }
    vm_drain();
}

// ---------------------------------------------------------------------------------------------------------
// LSTM forward, slot-interleaved
// ---------------------------------------------------------------------------------------------------------
// One wave per SIMD issues in order: a 16x16x32 MFMA occupies the matrix pipe for ~16 cycles and only 2-3 other
// instructions can be issued underneath it (tools/probes/mfma_probe.hip: 65.6 cycles per 4 MFMAs bare, +0 for 2
// VALU per MFMA, ~+5 cycles per further VALU).  So every MFMA is its own asm statement, and the gate arithmetic
// of the PREVIOUS unit tile is cut into 32 pieces of ~3 instructions, one per MFMA slot.  Empty volatile asm
// statements ("pins") on the piece's operands keep hipcc from moving a piece out of its slot: volatile statements
// keep their order, and a piece sits between the pin that defines its inputs and the pin that uses its outputs.
// Global addresses are a wave-uniform base (SGPR pair) + one per-lane 32-bit offset + immediates.
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
// the first MFMA of an accumulation chain: C = 0 as the instruction's inline constant instead of a zeroed register quad
// (round 4: 8 v_mov_b64 per step and four registers of zeros less in the LSTM BPTT kernel)
__device__ __forceinline__ void mfma1_first(f32x4& c, const frag& u, const frag& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(u), "v"(b));
}
__device__ __forceinline__ void pinv(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pinq(frag& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pinu(unsigned& v) { asm volatile("" : "+v"(v)); }
// explicitly global (address space 1) views: a pointer that went through an asm pin is no longer provably global,
// and hipcc would fall back to flat_ instructions
typedef __attribute__((address_space(1))) unsigned char gbyte;
typedef __attribute__((address_space(1))) u16x4 g_u16x4;
typedef __attribute__((address_space(1))) u16x8 g_u16x8;
__device__ __forceinline__ gbyte* to_global(const void* p) { return (gbyte*)(const_cast<void*>(p)); }
__device__ __forceinline__ void pins(gbyte*& p) { asm volatile("" : "+s"(p)); }
// the data a pipelined stack hands over (saved h rows forward, gate gradients backward) leaves WRITE-THROUGH (common.h)
__device__ __forceinline__ void store16_wt(gbyte* uniform_base, unsigned lane_off, u16x8 v) {
    ::store16_wt((const void*)uniform_base, lane_off, v);
}

// Residency class of the 32 fragment groups of a step (group = the 4 gate fragments of one k-group of one unit tile, in
// order of use).  CLS_T: group 7 (tile 0's last).  With the NLG groups that live in LDS all at the end of the step (F0 = 32 -
// NLG) four waves stream 1 KiB per slot from LDS for 36 slots in a row and those slots take ~50 cycles instead of 16;
// spread evenly over groups F0..31 they overlap with the register-fed slots.  The training variants cannot afford it (the
// LDS-staging registers would be live through tile 2 as well: 17-21 registers spilled); the inference variants can.
enum { CLS_A = 0, CLS_V = 1, CLS_L = 2, CLS_T = 3 };
template <int NAG, int NVG, int NLG, int F0>       // numbers of groups per class (NAG + NVG + NLG + 1 == 32)
struct lstm_group_map {
    int cls[32], ord[32];
    constexpr lstm_group_map() : cls{}, ord{} {
        bool isl[32] = {};
        constexpr int SPAN = 32 - F0;
        for (int k = 0; k < NLG; ++k) isl[F0 + (k * SPAN + SPAN / 2) / (NLG > 0 ? NLG : 1)] = true;
        int na = 0, nv = 0, nl = 0;
        for (int g = 0; g < 32; ++g) {
            if (g == 7) { cls[g] = CLS_T; ord[g] = 0; }
            else if (isl[g]) { cls[g] = CLS_L; ord[g] = nl++; }
            else if (na < NAG) { cls[g] = CLS_A; ord[g] = na++; }
            else { cls[g] = CLS_V; ord[g] = nv++; }
        }
    }
};
#ifndef LSTM_L_FIRST
#define LSTM_L_FIRST 12
#endif

template <int XMODE, int SAVE, int NA, int NV, bool WIDEN = false>
__device__ __forceinline__ void lstm_fwd_il_body(const mvae_rnn_fwd_args& a, const unsigned bx) {
    constexpr int G = 4, GH = G * RH;
    constexpr int FPW = G * RNT * RS, NGRP = FPW / 4;
    // Residency class of fragment f (f = order of use: (tile*8 + kgroup)*4 + gate):
    //   T  the 4 fragments of tile 0's last k-group: streamed from L2 every step into the accumulator set that is
    //      idle while tile 0 runs (nothing else is free: 512 registers + 160 KiB hold U, the h tiles and the state)
    //   A  the first NA others in accumulator registers, V the next NV in vector registers, L the rest in LDS
    constexpr int TB = 28, NT = 4, NLc = FPW - NT - NA - NV;
    static_assert(XMODE != MVAE_X_SCALAR, "scalar inputs run on the phased kernel");
    static_assert(NA % 4 == 0 && NV % 4 == 0 && NLc >= 0 && NA <= 64, "fragment classes");
    constexpr lstm_group_map<NA / 4, NV / 4, NLc / 4, (SAVE == SAVE_ALL ? 32 - NLc / 4 : LSTM_L_FIRST)> GM{};

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* hbuf = smem;                                             // [2][16][RH] bf16, swizzled
    frag* ulds = reinterpret_cast<frag*>(smem + 2 * 16 * RH * 2);           // [4][NL][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.u_pack);
    frag* myl = ulds + (size_t)w * NLc * 64 + l;

    auto frag_ptr = [&](int f) -> const frag* {
        const int g = f % G, ks = (f / G) % RS, n = f / (G * RS);
        return up + (size_t)((g * (RH / 16) + w * RNT + n) * RS + ks) * 64 + l;
    };
    frag ua[NA > 0 ? NA : 4], uv[NV > 0 ? NV : 4];
    static_for<0, NGRP>(SF_LAMBDA(gc) {
        constexpr int gi = decltype(gc)::value, f = gi * 4, c = GM.cls[gi], i = GM.ord[gi] * 4;     // i: index within the class
        if constexpr (c == CLS_A) load4_agpr_nowait(ua[i], ua[i + 1], ua[i + 2], ua[i + 3], frag_ptr(f), frag_ptr(f + 1), frag_ptr(f + 2), frag_ptr(f + 3));
        else if constexpr (c == CLS_V) {
            uv[i] = *frag_ptr(f); uv[i + 1] = *frag_ptr(f + 1); uv[i + 2] = *frag_ptr(f + 2); uv[i + 3] = *frag_ptr(f + 3);
        } else if constexpr (c == CLS_L) {
            myl[(size_t)i * 64] = *frag_ptr(f); myl[(size_t)(i + 1) * 64] = *frag_ptr(f + 1);
            myl[(size_t)(i + 2) * 64] = *frag_ptr(f + 2); myl[(size_t)(i + 3) * 64] = *frag_ptr(f + 3);
        }
    });
    const frag* tsrc = frag_ptr(TB);            // T fragments: gates 0..3 are (RH/16)*RS*64 fragments apart

    const int ld0 = a.h0_ld ? a.h0_ld : RH, ldl = a.h_last_ld ? a.h_last_ld : RH;
    unsigned lane8 = (unsigned)l * 8u;          // this lane's 8 bytes inside a TILE16 tile
    unsigned lane16 = (unsigned)l * 16u;        // ... 16 bytes inside a TILE16P tile pair
    const int ub0 = w * 64 + q * 4;             // first of this lane's 4 units in tile 0 (tile n: + 16 n)

    // Swizzled LDS byte offsets, all derived by XOR from three per-lane values (chunk index ^ row is XOR-linear):
    //   h write of tile n      : hw0 ^ (n << 5)              B fragment of k-group ks : bf4[ks & 3] + 256 * (ks >> 2)
    //   row-major copy chunk j : tl0 ^ (j * 1056)  (rows 4w + 2j + l/32, chunk l % 32)
    unsigned hw0 = (unsigned)r * 512u + ((((unsigned)w * 8u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    unsigned bf4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf4[j] = (unsigned)r * 512u + ((((unsigned)j * 4u + (unsigned)q) ^ (unsigned)r) << 4);
    const unsigned row0 = 4u * w + ((unsigned)l >> 5), ch0 = (unsigned)l & 31u;
    unsigned tl0 = row0 * 512u + ((ch0 ^ row0) << 4);
    unsigned tg0 = row0 * 512u + ch0 * 16u;

    f32x4 creg[RNT];
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 h0v = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * ld0 + ub0 + 16 * n) : z4;
        creg[n] = a.c0 ? *reinterpret_cast<const f32x4*>(a.c0 + (size_t)b * ld0 + ub0 + 16 * n) : z4;
        *reinterpret_cast<u16x4*>(hbuf + (hw0 ^ (n << 5))) = pack4(h0v);
        if (SAVE == SAVE_ALL)      // c_0 -> slot 0 of the (T+1, B, H) TILE16P array: 8 bytes of the lane's 16 per tile pair
            *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned char*>(a.cs) +
                                      ((size_t)bx * (RES_WGMAJOR ? (T + 1) : 1) * (RH / 32) + w * 2 + (n >> 1)) * 1024 + (unsigned)l * 16u + (n & 1) * 8) = pack4(creg[n]);
    }

    // ---- x queue (packed bf16x4 per tile and gate) for the step about to be computed --------------------------
    u16x4 xq[RNT][G];
    unsigned xoff = 0;                          // per-lane byte offset of the x source
    int i_q = 0;
    const unsigned char* xbase0;                // wave-uniform base (step 0 for DENSE)
    if (XMODE == MVAE_X_DENSE) {
        xoff = lane8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp) + ((size_t)bx * (GH / 16) + w * RNT) * 512;
    } else if (XMODE == MVAE_X_INDEX) {
        // MVAE_TABLE_PAIRED (round 4): the lane's values of tiles 2j and 2j+1 are 16 contiguous bytes of the row - 8 gathers per
        // step instead of 16 (a gather touches 16 table rows whatever its width: it is the instruction count that costs)
        xoff = (unsigned)a.idx[b] * (GH * 2) + q * 16;
        xbase0 = reinterpret_cast<const unsigned char*>(a.table) + w * 128;
        i_q = a.idx[(size_t)(T > 1 ? 1 : 0) * B + b];
    } else {
        xoff = (unsigned)b * (GH * 2) + q * 8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp0) + w * 128;
    }
    // pipelined stack bookkeeping without divisions: chunk pk (ending before step phi) is the one being computed
    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = 0, phi = cs_steps;
    if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready, wait_value, a.status);      // chunk 0 of xp
    constexpr unsigned XG = XMODE == MVAE_X_DENSE ? (RH / 16) * 512 : RH * 2;     // bytes between gates / tiles
    constexpr unsigned XN = XMODE == MVAE_X_DENSE ? 512 : 32;
    if (XMODE == MVAE_X_INDEX) {
#pragma unroll
        for (int j = 0; j < RNT / 2; ++j)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u16x8 t = *reinterpret_cast<const u16x8*>(xbase0 + g * XG + j * 64 + xoff);
                xq[2 * j][g] = __builtin_shufflevector(t, t, 0, 1, 2, 3);
                xq[2 * j + 1][g] = __builtin_shufflevector(t, t, 4, 5, 6, 7);
            }
    } else {
#pragma unroll
        for (int n = 0; n < RNT; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g) xq[n][g] = *reinterpret_cast<const u16x4*>(xbase0 + g * XG + n * XN + xoff);
    }

    constexpr float K2 = 2.8853900817779268f;   // 2 / ln 2
    f32x4 accA[4], accB[4], hn = {0.f, 0.f, 0.f, 0.f};
    frag bq[3], lt[4];
    const s16x4 ident = identity_fragment(l);   // (FWL_MFMA_WIDEN: x -> f32 accumulators on the matrix pipe)
    auto request_t = [&]() __attribute__((always_inline)) {      // T fragments into the idle accumulator set
#pragma unroll
        for (int g = 0; g < 4; ++g) accB[g] = __builtin_bit_cast(f32x4, tsrc[(size_t)g * (RH / 16) * RS * 64]);
    };
    request_t();
    vm_drain();
    lds_barrier();

    gbyte *acts_p[G], *cs_p, *hs_p;              // step t:   saved gates (per gate), c_t (slot t+1), h_{t-1} (slot t)
    gbyte* x_p[G];                               // step t+1: inputs (per gate)
    const size_t x_step = tps * (GH / 16) * 512, hs_step = (size_t)B * RH * 2;
#if RES_WGMAJOR     // (experiment: saved activations workgroup-major - a workgroup's steps contiguous - instead of time-major)
    const size_t acts_step = (size_t)(GH / 32) * 1024, cs_step = (size_t)(RH / 32) * 1024;
    const size_t acts_b0 = (size_t)bx * T * (GH / 32), cs_b0 = (size_t)bx * (T + 1) * (RH / 32);
#else
    const size_t acts_step = x_step, cs_step = tps * (RH / 16) * 512;
    const size_t acts_b0 = (size_t)bx * (GH / 32), cs_b0 = (size_t)bx * (RH / 32);
#endif
#pragma unroll
    for (int g = 0; g < G; ++g) {
        acts_p[g] = to_global(a.acts) + (acts_b0 + g * (RH / 32) + w * 2) * 1024;
        x_p[g] = to_global(xbase0) + g * XG + (XMODE == MVAE_X_DENSE && T > 1 ? x_step : 0);
    }
    cs_p = to_global(a.cs) + (cs_b0 + w * 2) * 1024 + cs_step;
    hs_p = to_global(a.hs) + (size_t)bx * 16 * (RH * 2);

    for (int t = 0; t < T; ++t) {
        const int tstep = t;
        (void)tstep;
        // the three per-lane offset seeds are "redefined" every step so that hipcc derives the others where they are
        // used instead of keeping a dozen loop-invariant registers
        pinu(hw0); pinu(tl0);
        // pipelined stack: x of step t+1 is requested during this step - its chunk must have been published
        if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready && t + 1 < T && t + 1 == phi)
            wave_wait_ge(a.wait_ready + pk + 1, wait_value, a.status);
        unsigned char* hcur = hbuf;                                        // bf4 / tl0 / hw0 carry the buffer bit
        // Wave-uniform running pointers (SGPR pairs, one per gate: the 8 KiB between gates exceeds the instruction's
        // immediate range).  The pins stop hipcc from folding them back into per-lane 64-bit address arithmetic.
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(acts_p[3]); pins(cs_p); pins(hs_p);
        pins(x_p[0]); pins(x_p[1]); pins(x_p[2]); pins(x_p[3]);
        STAMP(0);
        bq[0] = *reinterpret_cast<const frag*>(hcur + bf4[0]);
        bq[1] = *reinterpret_cast<const frag*>(hcur + bf4[1]);
        if (SAVE >= SAVE_HS && !ABL_NOTRG) {     // row-major copy of h_{t-1}: staged in the (idle) LDS-fragment registers
            lt[0] = *reinterpret_cast<const frag*>(hcur + tl0);
            lt[1] = *reinterpret_cast<const frag*>(hcur + (tl0 ^ 1056u));
        }

        // next step's x of (tile m, gate g), into the registers whose values were just consumed
        auto request_x = [&](auto mc, auto gc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value, g = decltype(gc)::value;
            if (XMODE == MVAE_X_INDEX && !ABL_NOX) {
                // tile PAIRS: (0, 1) in tile 0's turn (slots 19.. of tile 1: both tiles' x are consumed by then), (2, 3) in tile 3's
                if constexpr (m == 0 || m == RNT - 1) {
                    constexpr int j = m == 0 ? 0 : 1;
                    if (g == 0) xoff = (unsigned)i_q * (GH * 2) + q * 16;      // (tile 3 overwrites i_q last)
                    pinu(xoff);
                    const u16x8 pr = *reinterpret_cast<const g_u16x8*>(x_p[g] + j * 64 + xoff);
                    xq[2 * j][g] = __builtin_shufflevector(pr, pr, 0, 1, 2, 3);
                    xq[2 * j + 1][g] = __builtin_shufflevector(pr, pr, 4, 5, 6, 7);
                    if (m == RNT - 1 && g == G - 1) i_q = a.idx[(size_t)(t_next2(t, T)) * B + b];
                }
            } else if (XMODE != MVAE_X_CONST && !ABL_NOX) {
                pinu(xoff);
                xq[m][g] = *reinterpret_cast<const g_u16x4*>(x_p[g] + m * XN + xoff);
            }
        };
        u16x4 held[G + 1];
        // Saved activations of tile m, quantity k (gates i,f,g,o from the tile's accumulator set Q, then c).  TILE16P:
        // an even tile parks its packed values, the odd tile stores the pair as ONE 16-byte access per lane.
        auto save = [&](auto mc, auto kc, f32x4* Q) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value, k = decltype(kc)::value;
            if (SAVE == SAVE_ALL && !ABL_NOSAVE) {
                const u16x4 mine = pack4(k < G ? Q[k < G ? k : 0] : creg[m]);
                if constexpr ((m & 1) == 0) held[k] = mine;
                else {
                    // (the 32-bit lane offset is re-defined in this basic block so that instruction selection sees
                    //  "uniform base + zext(lane offset)" and uses the SGPR-base addressing mode)
                    pinu(lane16);
                    const u16x8 pair = {held[k][0], held[k][1], held[k][2], held[k][3], mine[0], mine[1], mine[2], mine[3]};
                    if constexpr (k < G) *reinterpret_cast<g_u16x8*>(acts_p[k < G ? k : 0] + (m >> 1) * 1024 + lane16) = pair;
                    else *reinterpret_cast<g_u16x8*>(cs_p + (m >> 1) * 1024 + lane16) = pair;
                }
            }
        };
        // One piece of tile m's gate arithmetic.  P = that tile's accumulators (gate pre-activations).  Two elements
        // move in lockstep through 16 single-instruction stages, one stage per MFMA slot: 2 VALU per slot is what a
        // 16-cycle MFMA hides (tools/probes/slot_probe.hip), and the partner's instruction separates every exp / rcp
        // from its consumer (no hazard nops).
        auto piece = [&](auto mc, auto slc, f32x4* P) __attribute__((always_inline)) {
            // (the tail - no MFMAs to hide under - runs each stage for all 4 elements: 4 independent chains)
            constexpr int m = decltype(mc)::value, sl = decltype(slc)::value;
            constexpr bool tail = m == RNT - 1;
            constexpr int pr = tail ? 0 : sl >> 4, sub = tail ? sl >> 1 : sl & 15, ne = tail ? ((sl & 1) ? 0 : 4) : 2;
#pragma unroll
            for (int k = 0; k < ne; ++k) {
                const int e = 2 * pr + k;
                // tiles 1..3 start their accumulators at x (see below); tile 0 adds it here
                auto pre = [&](int g) -> float { return m == 0 ? P[g][e] + bf2f(xq[m][g][e]) : P[g][e]; };
                if constexpr (sub == 0) P[0][e] = hard_sigmoid(pre(0));
                if constexpr (sub == 1) P[1][e] = hard_sigmoid(pre(1));
                if constexpr (sub == 2) P[3][e] = hard_sigmoid(pre(3));
                if constexpr (sub == 3) P[2][e] = pre(2) * K2;
                if constexpr (sub == 4) P[2][e] = ABL_NOTRANS ? P[2][e] * 0.5f : __builtin_amdgcn_exp2f(P[2][e]);
                if constexpr (sub == 5) P[2][e] = P[2][e] + 1.0f;
                if constexpr (sub == 6) P[2][e] = ABL_NOTRANS ? P[2][e] * 0.5f : __builtin_amdgcn_rcpf(P[2][e]);
                if constexpr (sub == 7) P[2][e] = 1.0f - 2.0f * P[2][e];
                if constexpr (sub == 8) creg[m][e] = P[1][e] * creg[m][e];             // f * c_{t-1}, in place
                if constexpr (sub == 9) creg[m][e] = P[0][e] * P[2][e] + creg[m][e];   // + i * g
                if constexpr (sub == 10) hn[e] = creg[m][e] * K2;
                if constexpr (sub == 11) hn[e] = ABL_NOTRANS ? hn[e] * 0.5f : __builtin_amdgcn_exp2f(hn[e]);
                if constexpr (sub == 12) hn[e] = hn[e] + 1.0f;
                if constexpr (sub == 13) hn[e] = ABL_NOTRANS ? hn[e] * 0.5f : __builtin_amdgcn_rcpf(hn[e]);
                if constexpr (sub == 14) hn[e] = 1.0f - 2.0f * hn[e];
                if constexpr (sub == 15) hn[e] = P[3][e] * hn[e];
            }
            // tile 0's x is consumed by slot 19: its next-step loads go out one per 4 slots (a VMEM instruction costs
            // ~16 cycles of the CU's address unit; four waves bursting 9 of them stall each other)
            if constexpr (m == 0 && sl >= 19 && (sl - 19) % 4 == 0 && (sl - 19) / 4 < G)
                request_x(mc, std::integral_constant<int, (sl - 19) / 4 < G ? (sl - 19) / 4 : 0>{});
            // saves as soon as a quantity is final for all 4 elements (i: slot 16, f: 17, o: 18, g: 23, c: 25), apart
            if constexpr (!tail) {
                if constexpr (sl == 20) save(mc, std::integral_constant<int, 0>{}, P);
                if constexpr (sl == 22) save(mc, std::integral_constant<int, 1>{}, P);
                if constexpr (sl == 24) save(mc, std::integral_constant<int, 3>{}, P);
                if constexpr (sl == 26) save(mc, std::integral_constant<int, 2>{}, P);
                if constexpr (sl == 29) save(mc, std::integral_constant<int, 4>{}, P);
            } else {        // stage s runs at tail slot 2s: i final after stage 0, f 1, o 2, g 7, c 9
                if constexpr (sl == 7) save(mc, std::integral_constant<int, 0>{}, P);
                if constexpr (sl == 11) save(mc, std::integral_constant<int, 1>{}, P);
                if constexpr (sl == 15) save(mc, std::integral_constant<int, 3>{}, P);
                if constexpr (sl == 19) save(mc, std::integral_constant<int, 2>{}, P);
                if constexpr (sl == 23) save(mc, std::integral_constant<int, 4>{}, P);
            }
            if constexpr (sl == 31) {
                // tile m complete: h -> the other LDS tile
                *reinterpret_cast<u16x4*>(hbuf + (hw0 ^ ((m << 5) | 8192))) = pack4(hn);
                if (t == T - 1) {
                    if (a.h_last) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub0 + 16 * m) = hn;
                    if (a.c_last) *reinterpret_cast<f32x4*>(a.c_last + (size_t)b * ldl + ub0 + 16 * m) = creg[m];
                }
            }
        };
        static_for<0, RNT + 1>(SF_LAMBDA(nc) {
            constexpr int n = decltype(nc)::value;
            f32x4* acc = (n & 1) ? accB : accA;
            f32x4* P = (n & 1) ? accA : accB;
            if constexpr (n == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if constexpr (n < RNT) {
                // Accumulators start at this step's x (one unpack per value instead of zero + unpack + add), and the
                // next step's x of this tile is requested a whole step ahead (spread over slots 1, 5, 9, 13).  Tile 0
                // cannot: its x would have to be waited for at the very top of the step, behind the previous tail's stores.
                if constexpr (n == 1) asm volatile("s_nop 3");     // the T fragments were MFMA operands a moment ago
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = WIDEN ? widen4(ident, xq[n][g], f32x4{0.f, 0.f, 0.f, 0.f}) : unpack4(xq[n][g]);
            }
            static_for<0, 32>(SF_LAMBDA(slc) {
                constexpr int sl = decltype(slc)::value;
                if constexpr (n < RNT) {
                    constexpr int ks = sl >> 2, g = sl & 3, gi = n * RS + ks, f = gi * 4 + g;
                    constexpr int cl = GM.cls[gi], ci = GM.ord[gi] * 4 + g;      // class, index within the class
                    if constexpr (n == 0 && sl == TB) {
                        // One wait per step: this step's x and T fragments (requested during the previous step) and
                        // every older store.  The saved-sequence row-major copy of h_{t-1} follows it.
                        STAMP(1);
                        vm_drain();
                        STAMP(2);
#pragma unroll
                        for (int m = 0; m < RNT; ++m) pin4(xq[m][0], xq[m][1], xq[m][2], xq[m][3]);   // (CONST: keeps the rows packed)
                        if (XMODE == MVAE_X_INDEX) pini(i_q);
                        pinv(accB[0]); pinv(accB[1]); pinv(accB[2]); pinv(accB[3]);
                        if (SAVE >= SAVE_HS && !ABL_NOTRG) {
                            pinu(tg0);
                            store16_wt(hs_p, tg0, lt[0]);
                            store16_wt(hs_p, tg0 + 1024u, lt[1]);
                        }
                    }
                    if constexpr (g == 0 && gi + 2 < NGRP && !ABL_NOB)
                        bq[(gi + 2) % 3] = *reinterpret_cast<const frag*>(hcur + bf4[(ks + 2) & 3] + 256 * (((ks + 2) & 7) >> 2));
                    if constexpr (sl == 0)    // VALU-initialised accumulators -> first MFMA of the tile
                        asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                    constexpr int bi = ABL_NOB ? (gi & 1) : gi % 3;
                    if constexpr (cl == CLS_T) mfma1<false>(acc[g], __builtin_bit_cast(frag, accB[g]), bq[bi]);
                    else if constexpr (cl == CLS_A) mfma1<true>(acc[g], ua[ci], bq[bi]);
                    else if constexpr (cl == CLS_V) mfma1<false>(acc[g], uv[ci], bq[bi]);
                    else mfma1<false>(acc[g], lt[g], bq[bi]);
                    // the next group's fragments, if they live in LDS: into the staging register (idle, or just read)
                    if constexpr (gi + 1 < NGRP && !ABL_NOL) {
                        if constexpr (GM.cls[gi + 1 < NGRP ? gi + 1 : 0] == CLS_L)
                            lt[g] = myl[(size_t)(GM.ord[gi + 1 < NGRP ? gi + 1 : 0] * 4 + g) * 64];
                    }
                    if constexpr (n >= 1 && (sl & 3) == 1 && sl < 16) request_x(nc, std::integral_constant<int, (sl >> 2) & 3>{});
                    __builtin_amdgcn_sched_barrier(0);   // MFMA first, then the slot's fillers: strict alternation
                }
                if constexpr (n > 0 && !ABL_NOMATH) piece(std::integral_constant<int, (n > 0 ? n - 1 : 0)>{}, slc, P);
                __builtin_amdgcn_sched_barrier(0);       // nothing moves across a slot boundary
            });
            // the tail reads the last tile's accumulators right away (4-pass MFMA: 8 wait states required)
            if constexpr (n == RNT - 1) asm volatile("s_nop 9" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            STAMP(8 + n);
        });
        request_t();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acts_p[g] += acts_step;
            if (XMODE == MVAE_X_DENSE && t + 2 < T) x_p[g] += x_step;
        }
        cs_p += cs_step;
        hs_p += hs_step;
        // flip the h tiles: every swizzled offset carries the buffer bit
#pragma unroll
        for (int j = 0; j < 4; ++j) bf4[j] ^= 8192u;
        hw0 ^= 8192u;
        tl0 ^= 8192u;
        STAMP(6);
        res_barrier();
        res_skew(w);
        STAMP(7);
        // pipelined stack: hs slot t (= h_{t-1}) left this step, so the chunk ending at step t-1 is complete
        if (cs_steps && t == phi) {
            if (SAVE >= SAVE_HS && a.signal_done) wave_signal_done<false>(a.signal_done + pk);
            ++pk;
            phi += cs_steps;
        }
    }
    if (SAVE >= SAVE_HS) {       // slot T = h_{T-1}
        store16_wt(hs_p, tg0, *reinterpret_cast<const u16x8*>(hbuf + tl0));
        store16_wt(hs_p, tg0 + 1024u, *reinterpret_cast<const u16x8*>(hbuf + (tl0 ^ 1056u)));
        if (cs_steps && a.signal_done) wave_signal_done<false>(a.signal_done + pk);
    }
    vm_drain();
}
template <int XMODE, int SAVE, int NA, int NV>
__global__ __launch_bounds__(256, 1) void lstm_fwd_il_k(const mvae_rnn_fwd_args a) {
    // (x -> accumulators through widen4 in the single-launch inference variants: the training variants sit at 252 VGPRs and the phase
    //  launches' bodies share their registers with the dispatch - both spill with it)
    lstm_fwd_il_body<XMODE, SAVE, NA, NV, FWL_MFMA_WIDEN && SAVE != SAVE_ALL>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// GRU forward, slot-interleaved (Keras 2.0.x GRU: reset gate applied BEFORE the candidate matmul)
// ---------------------------------------------------------------------------------------------------------
// U is 384 KiB: the 64 z/r fragments of a wave sit in accumulator registers, its 32 candidate fragments in vector
// registers (GRU_NVB; the rest, if any, in LDS) - the whole recurrent kernel in registers, nothing here is starved like the
// LSTM kernels.
// Step t:  A  z, r pre-activations of tile pairs (0,1) then (2,3): 2 x 32 single-MFMA slots, B = h_{t-1} tile; the
//             gaps of pair 1 carry pair 0's gate arithmetic (z, r, r*h -> rh tile), the gaps of pair 0 the deferred
//             stores of the previous step and this step's row-major copy of h_{t-1}
//          -  pair 1's gate arithmetic, barrier (the candidate needs every wave's r*h)
//          B  candidate pre-activations, tiles (0,1) then (2,3): 2 x 16 slots, B = rh tile, fragments staged from LDS;
//             the gaps of the second half carry tanh + the h update of tiles 0,1
//          -  tanh + h update of tiles 2,3, barrier
// Inputs are TILE16, saved activations (z, r, candidate) TILE16P: one 16-byte store per gate and tile pair.
template <int XMODE, int SAVE>
__device__ __forceinline__ void gru_fwd_il_body(const mvae_rnn_fwd_args& a, const unsigned bx) {
    constexpr int G = 3, GH = G * RH, NLc = RNT * RS;             // 32 candidate fragments per wave in LDS
    static_assert(XMODE != MVAE_X_SCALAR, "scalar inputs run on the phased kernel");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* hbuf = smem;                                             // [2][16][RH] bf16, swizzled
    unsigned char* rhbuf = smem + 2 * 16 * RH * 2;                          // [16][RH]
    frag* ulds = reinterpret_cast<frag*>(smem + 3 * 16 * RH * 2);           // [4][NL][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.u_pack);
    frag* myl = ulds + (size_t)w * NLc * 64 + l;
    auto src_frag = [&](int g, int n, int ks) -> const frag* {
        return up + (size_t)((g * (RH / 16) + w * RNT + n) * RS + ks) * 64 + l;
    };
    // phase A fragment f = ((np*8 + ks)*2 + nn)*2 + g  (tile 2np+nn, gate g in {z, r});  phase B fragment (LDS) i =
    // (half*8 + ks)*2 + nn  (tile 2half+nn, candidate)
    frag ua[64];
    static_for<0, 16>(SF_LAMBDA(ic) {
        constexpr int f = decltype(ic)::value * 4, np = f >> 5, ks = (f >> 2) & 7;
        load4_agpr_nowait(ua[f], ua[f + 1], ua[f + 2], ua[f + 3], src_frag(0, 2 * np, ks), src_frag(1, 2 * np, ks),
                          src_frag(0, 2 * np + 1, ks), src_frag(1, 2 * np + 1, ks));
    });
    // candidate fragments: the first NVB of phase B's 32 slots read theirs from vector registers (the kernel has them to
    // spare: four waves streaming all 32 from LDS made the phase LDS-bandwidth bound), the rest from LDS
    constexpr int NVB = GRU_NVB;
    frag uv[NVB > 0 ? NVB : 1];
#pragma unroll
    for (int i = 0; i < NLc; ++i) {
        const frag f = *src_frag(2, 2 * (i >> 4) + (i & 1), (i >> 1) & 7);
        if (i < NVB) uv[i] = f;
        else myl[(size_t)i * 64] = f;
    }

    const int ld0 = a.h0_ld ? a.h0_ld : RH, ldl = a.h_last_ld ? a.h_last_ld : RH;
    unsigned lane8 = (unsigned)l * 8u, lane16 = (unsigned)l * 16u;      // TILE16 inputs; TILE16P saved activations
    const int ub0 = w * 64 + q * 4;
    unsigned hw0 = (unsigned)r * 512u + ((((unsigned)w * 8u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    unsigned bf4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf4[j] = (unsigned)r * 512u + ((((unsigned)j * 4u + (unsigned)q) ^ (unsigned)r) << 4);
    const unsigned row0 = 4u * w + ((unsigned)l >> 5), ch0 = (unsigned)l & 31u;
    unsigned tl0 = row0 * 512u + ((ch0 ^ row0) << 4);
    unsigned tg0 = row0 * 512u + ch0 * 16u;

    f32x4 hreg[RNT];
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        hreg[n] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + (size_t)b * ld0 + ub0 + 16 * n) : z4;
        *reinterpret_cast<u16x4*>(hbuf + (hw0 ^ (n << 5))) = pack4(hreg[n]);
    }
    // ---- x queue ------------------------------------------------------------------------------------------------
    // Inputs are requested TWO steps ahead into alternating buffers (the step loop is unrolled by two): beside the gradient
    // GEMMs an HBM round trip takes longer than the one step they used to get (in-step 3.3 us per time step against 1.8 alone)
    u16x4 xq[2][RNT][G];
    unsigned xoff = 0;
    int i_q = 0;                                  // X_INDEX: the index of step min(t+2, T-1)
    const unsigned char* xbase0;
    if (XMODE == MVAE_X_DENSE) {
        xoff = lane8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp) + ((size_t)bx * (GH / 16) + w * RNT) * 512;
    } else if (XMODE == MVAE_X_INDEX) {
        xoff = (unsigned)a.idx[b] * (GH * 2) + q * 16;         // MVAE_TABLE_PAIRED: tiles 2j, 2j+1 of a lane in one 16-byte gather
        xbase0 = reinterpret_cast<const unsigned char*>(a.table) + w * 128;
        i_q = a.idx[(size_t)(T > 2 ? 2 : T - 1) * B + b];
    } else {
        xoff = (unsigned)b * (GH * 2) + q * 8;
        xbase0 = reinterpret_cast<const unsigned char*>(a.xp0) + w * 128;
    }
    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = 0, phi = cs_steps;
    if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready, wait_value, a.status);
    constexpr unsigned XG = XMODE == MVAE_X_DENSE ? (RH / 16) * 512 : RH * 2;
    constexpr unsigned XN = XMODE == MVAE_X_DENSE ? 512 : 32;
    const size_t x_step1 = (XMODE == MVAE_X_DENSE && T > 1) ? tps * (GH / 16) * 512 : 0;       // step 1 (buffer 1)
    const unsigned xoff1 = XMODE == MVAE_X_INDEX ? (unsigned)a.idx[(size_t)(T > 1 ? 1 : 0) * B + b] * (GH * 2) + q * 16 : xoff;
    if (XMODE == MVAE_X_INDEX) {
#pragma unroll
        for (int j = 0; j < RNT / 2; ++j)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u16x8 p0 = *reinterpret_cast<const u16x8*>(xbase0 + g * XG + j * 64 + xoff);
                const u16x8 p1 = *reinterpret_cast<const u16x8*>(xbase0 + g * XG + j * 64 + xoff1);
                xq[0][2 * j][g] = __builtin_shufflevector(p0, p0, 0, 1, 2, 3);
                xq[0][2 * j + 1][g] = __builtin_shufflevector(p0, p0, 4, 5, 6, 7);
                xq[1][2 * j][g] = __builtin_shufflevector(p1, p1, 0, 1, 2, 3);
                xq[1][2 * j + 1][g] = __builtin_shufflevector(p1, p1, 4, 5, 6, 7);
            }
    } else {
#pragma unroll
        for (int n = 0; n < RNT; ++n)
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
        acts_p[g] = to_global(a.acts) + ((size_t)bx * (GH / 32) + g * (RH / 32) + w * 2) * 1024;
        x_p[g] = to_global(xbase0) + g * XG + (XMODE == MVAE_X_DENSE ? (T > 2 ? 2 : T - 1) * acts_step : 0);
    }
    hh_prev_p = acts_p[2];
    hs_p = to_global(a.hs) + (size_t)bx * 16 * (RH * 2);

    constexpr float K2 = 2.8853900817779268f;
    f32x4 accA[4], accB[4], zg[RNT];
    u16x8 hh_pk[2];                               // candidate of the previous step (tile pairs), stored during this step's phase A
    frag bq[3], lt[2], cp[2];                     // B-fragment ring; LDS-fragment staging; row-major copy staging
    hh_pk[0] = hh_pk[1] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    vm_drain();
    lds_barrier();

    auto step = [&](const int t, auto xbc) __attribute__((always_inline)) {
        constexpr int XB = XMODE == MVAE_X_CONST ? 0 : decltype(xbc)::value;       // the buffer holding this step's inputs
        const int tstep = t;
        (void)tstep;
        pinu(hw0); pinu(tl0);
        // pipelined stack: x of step t+2 is requested during this step - its chunk must have been published
        if (XMODE == MVAE_X_DENSE && cs_steps && a.wait_ready && t + 2 < T && t + 2 == phi)
            wave_wait_ge(uniform_ptr(a.wait_ready + pk + 1), wait_value, a.status);
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(hs_p); pins(hh_prev_p);
        pins(x_p[0]); pins(x_p[1]); pins(x_p[2]);
        unsigned char* hcur = hbuf;               // bf4 / tl0 / hw0 carry the buffer bit
        // (no drain: the compiler's counted waits cover this step's inputs, requested two steps ago; younger requests and
        // stores stay in flight)
#pragma unroll
        for (int n = 0; n < RNT; ++n) { pin1(xq[XB][n][0]); pin1(xq[XB][n][1]); pin1(xq[XB][n][2]); }
        if (XMODE == MVAE_X_INDEX) pini(i_q);
        if (SAVE >= SAVE_HS) {
            cp[0] = *reinterpret_cast<const frag*>(hcur + tl0);
            cp[1] = *reinterpret_cast<const frag*>(hcur + (tl0 ^ 1056u));
        }
        auto request_x = [&](int n, int g) __attribute__((always_inline)) {
            if (XMODE == MVAE_X_INDEX) {        // tile pairs: the even tile's turn fetches both (half the gathers)
                if (n & 1) return;
                if (n == 0 && g == 0) xoff = (unsigned)i_q * (GH * 2) + q * 16;
                pinu(xoff);
                const u16x8 pr = *reinterpret_cast<const g_u16x8*>(x_p[g] + (n >> 1) * 64 + xoff);
                xq[XB][n][g] = __builtin_shufflevector(pr, pr, 0, 1, 2, 3);
                xq[XB][n + 1][g] = __builtin_shufflevector(pr, pr, 4, 5, 6, 7);
            } else if (XMODE != MVAE_X_CONST) {
                pinu(xoff);
                xq[XB][n][g] = *reinterpret_cast<const g_u16x4*>(x_p[g] + n * XN + xoff);
            }
        };
        // ---- phase A ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            accA[j] = unpack4(xq[XB][j >> 1][j & 1]);         // accumulators start at x (tile j>>1, gate j&1)
            accB[j] = unpack4(xq[XB][2 + (j >> 1)][j & 1]);
        }
        bq[0] = *reinterpret_cast<const frag*>(hcur + bf4[0]);
        bq[1] = *reinterpret_cast<const frag*>(hcur + bf4[1]);
        asm volatile("s_nop 1" : "+v"(accA[0]), "+v"(accA[1]), "+v"(accA[2]), "+v"(accA[3]));
        // z, r and r*h of tile n from its two accumulators; element pairs in lockstep
        auto zr_math = [&](int n, f32x4& az, f32x4& ar, int e0, int ne) __attribute__((always_inline)) {
#pragma unroll
            for (int e = e0; e < e0 + ne; ++e) {
                az[e] = hard_sigmoid(az[e]);
                ar[e] = hard_sigmoid(ar[e]);
            }
        };
        static_for<0, 64>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value, np = sl >> 5, ks = (sl >> 2) & 7, j = sl & 3, gi = sl >> 2;
            f32x4* acc = np ? accB : accA;
            if constexpr (j == 0 && gi + 2 < 16)
                bq[(gi + 2) % 3] = *reinterpret_cast<const frag*>(hcur + bf4[(ks + 2) & 3] + 256 * (((ks + 2) & 7) >> 2));
            if constexpr (sl == 32) asm volatile("s_nop 1" : "+v"(accB[0]), "+v"(accB[1]), "+v"(accB[2]), "+v"(accB[3]));
            mfma1<true>(acc[j], ua[sl], bq[gi % 3]);
            __builtin_amdgcn_sched_barrier(0);
            // ---- fillers ----
            if constexpr (np == 0) {
                // the previous step's candidate tiles, the row-major copy of h_{t-1}, next step's z / r inputs
                if constexpr (sl == 2 || sl == 10) {
                    constexpr int pr = sl == 10;
                    if (SAVE == SAVE_ALL && t > 0) {
                        pinu(lane16);
                        *reinterpret_cast<g_u16x8*>(hh_prev_p + pr * 1024 + lane16) = hh_pk[pr];
                    }
                }
                if constexpr (sl == 20 || sl == 24) {
                    if (SAVE >= SAVE_HS) {
                        pinu(tg0);
                        store16_wt(hs_p, tg0 + (sl == 24 ? 1024u : 0u), cp[sl == 24 ? 1 : 0]);
                    }
                }
            } else {
                // pair 0: z, r (4 slots), r*h -> rh tile (slot 36..), z / r saves, next step's inputs of every tile
                constexpr int fs = sl - 32;
                if constexpr (fs < 4) zr_math(fs >> 1, accA[(fs >> 1) * 2], accA[(fs >> 1) * 2 + 1], (fs & 1) * 2, 2);
                if constexpr (fs == 4 || fs == 5) {
                    constexpr int n = fs - 4;
                    zg[n] = accA[n * 2];
                    *reinterpret_cast<u16x4*>(rhbuf + (hw0 & 8191u ^ (n << 5))) = pack4(accA[n * 2 + 1] * hreg[n]);
                }
                if constexpr (fs == 6 || fs == 10) {                                // z, r of the tile pair (0, 1)
                    constexpr int g = fs == 10;
                    if (SAVE == SAVE_ALL) {
                        pinu(lane16);
                        *reinterpret_cast<g_u16x8*>(acts_p[g] + lane16) = cat8(pack4(accA[g]), pack4(accA[2 + g]));
                    }
                }
                if constexpr (fs >= 15 && fs < 15 + 2 * 8 && (fs - 15) % 2 == 0) {   // next step's z / r inputs, all 4 tiles
                    constexpr int k = (fs - 15) / 2;
                    request_x(k >> 1, k & 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(accB[0]), "+v"(accB[1]), "+v"(accB[2]), "+v"(accB[3]));
        // pair 1's gate arithmetic has nothing to hide under: the candidate needs every wave's r*h first
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            zr_math(2 + nn, accB[nn * 2], accB[nn * 2 + 1], 0, 4);
            zg[2 + nn] = accB[nn * 2];
            *reinterpret_cast<u16x4*>(rhbuf + (hw0 & 8191u ^ ((2 + nn) << 5))) = pack4(accB[nn * 2 + 1] * hreg[2 + nn]);
        }
        res_barrier();
        // ---- phase B ------------------------------------------------------------------------------------------------
        f32x4 accC[RNT];
#pragma unroll
        for (int n = 0; n < RNT; ++n) accC[n] = unpack4(xq[XB][n][2]);
        frag rq[3];
        rq[0] = *reinterpret_cast<const frag*>(rhbuf + (bf4[0] & 8191u));
        rq[1] = *reinterpret_cast<const frag*>(rhbuf + (bf4[1] & 8191u));
        if (NVB < 1) lt[0] = myl[0];
        if (NVB < 2) lt[1] = myl[64];
        asm volatile("s_nop 1" : "+v"(accC[0]), "+v"(accC[1]), "+v"(accC[2]), "+v"(accC[3]));
        // candidate -> h for one element of tile n
        auto h_math = [&](int n, int e) __attribute__((always_inline)) {
            const float ex = __builtin_amdgcn_exp2f(accC[n][e] * K2);
            const float hh = 1.0f - 2.0f * __builtin_amdgcn_rcpf(ex + 1.0f);
            accC[n][e] = hh;                                       // kept for the save
            hreg[n][e] = hh + zg[n][e] * (hreg[n][e] - hh);        // z*h + (1-z)*hh
        };
        static_for<0, 32>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value, half = sl >> 4, ks = (sl >> 1) & 7, nn = sl & 1, n = half * 2 + nn;
            constexpr int gi = sl >> 1;                            // group of 2 MFMAs sharing rh fragment ks
            if constexpr (nn == 0 && gi + 2 < 16)
                rq[(gi + 2) % 3] = *reinterpret_cast<const frag*>(rhbuf + (bf4[(ks + 2) & 3] & 8191u) + 256 * (((ks + 2) & 7) >> 2));
            if constexpr (sl < NVB) mfma1<false>(accC[n], uv[sl < NVB ? sl : 0], rq[gi % 3]);
            else mfma1<false>(accC[n], lt[nn], rq[gi % 3]);
            if constexpr (sl + 2 < 32 && sl + 2 >= NVB) lt[nn] = myl[(size_t)(sl + 2) * 64];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (half == 0) {
                // z, r of tiles 2, 3; next step's candidate inputs
                if constexpr (sl == 0 || sl == 4) {                                 // z, r of the tile pair (2, 3)
                    constexpr int g = sl == 4;
                    if (SAVE == SAVE_ALL) {
                        pinu(lane16);
                        *reinterpret_cast<g_u16x8*>(acts_p[g] + 1024 + lane16) = cat8(pack4(accB[g]), pack4(accB[2 + g]));
                    }
                }
                // (the candidate inputs were consumed when the accumulators were initialised)
                if constexpr (sl >= 8 && (sl & 1) == 0) request_x((sl - 8) >> 1, 2);
                if constexpr (sl == 15) {
                    if (XMODE == MVAE_X_INDEX) i_q = a.idx[(size_t)(t + 3 < T ? t + 3 : T - 1) * B + b];
                }
            } else {
                // tanh + h update of tiles 0, 1 (their accumulators were finished by slot 15): one element per 2 slots
                constexpr int fs = sl - 16;
                if constexpr ((fs & 1) == 0) h_math(fs >> 3, (fs >> 1) & 3);
                if constexpr (fs == 15) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) *reinterpret_cast<u16x4*>(hbuf + (hw0 ^ ((m << 5) | 8192))) = pack4(hreg[m]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(accC[2]), "+v"(accC[3]));
#pragma unroll
        for (int n = 2; n < RNT; ++n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) h_math(n, e);
            *reinterpret_cast<u16x4*>(hbuf + (hw0 ^ ((n << 5) | 8192))) = pack4(hreg[n]);
        }
        hh_pk[0] = cat8(pack4(accC[0]), pack4(accC[1]));
        hh_pk[1] = cat8(pack4(accC[2]), pack4(accC[3]));
        if (t == T - 1 && a.h_last) {
#pragma unroll
            for (int n = 0; n < RNT; ++n) *reinterpret_cast<f32x4*>(a.h_last + (size_t)b * ldl + ub0 + 16 * n) = hreg[n];
        }
        hh_prev_p = acts_p[2];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acts_p[g] += acts_step;
            if (XMODE == MVAE_X_DENSE && t + 3 < T) x_p[g] += acts_step;
        }
        hs_p += hs_step;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf4[j] ^= 8192u;
        hw0 ^= 8192u;
        tl0 ^= 8192u;
        res_barrier();
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
    if (SAVE == SAVE_ALL) {
        *reinterpret_cast<g_u16x8*>(hh_prev_p + lane16) = hh_pk[0];
        *reinterpret_cast<g_u16x8*>(hh_prev_p + 1024 + lane16) = hh_pk[1];
    }
    if (SAVE >= SAVE_HS) {       // slot T = h_{T-1}
        store16_wt(hs_p, tg0, *reinterpret_cast<const u16x8*>(hbuf + tl0));
        store16_wt(hs_p, tg0 + 1024u, *reinterpret_cast<const u16x8*>(hbuf + (tl0 ^ 1056u)));
        if (cs_steps && a.signal_done) wave_signal_done<false>(uniform_ptr(a.signal_done + pk));
    }
    vm_drain();
}
template <int XMODE, int SAVE>
__global__ __launch_bounds__(256, 1) void gru_fwd_il_k(const mvae_rnn_fwd_args a) {
    gru_fwd_il_body<XMODE, SAVE>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// backward through time
// ---------------------------------------------------------------------------------------------------------
// The da tile in LDS is [16 rows][GH] bf16, swizzled, unpadded: LSTM needs every byte (32 KiB tile + 128 KiB of
// weights = all 160 KiB).

template <int CELL, bool HAS_EXT, int NA, int NV>
__global__ __launch_bounds__(256, 1) void rnn_bwd_res_k(const mvae_rnn_bwd_args a) {
    constexpr int G = mvae_gates(CELL), GH = G * RH;
    constexpr int S2 = GH / 32;                    // k-groups of the backward contraction (over gate columns)
    constexpr int FPW = RNT * S2;
    constexpr int NLc = FPW - NA - NV;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* dabuf = reinterpret_cast<bf16_t*>(smem);                        // [16][GH] swizzled
    frag* ulds = reinterpret_cast<frag*>(dabuf + 16 * GH);                  // [4][NL][64]
    bf16_t* rhtile = reinterpret_cast<bf16_t*>(ulds + 4 * NLc * 64);        // [16][RH] swizzled   (GRU)

    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: addresses become SGPR base + lane offset
    const int T = a.T, B = a.B;
    const int b = blockIdx.x * 16 + r;
    const size_t tiles_per_step = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.ut_pack);
    const bf16_t* __restrict__ hs = reinterpret_cast<const bf16_t*>(a.hs);
    const bf16_t* __restrict__ cs = reinterpret_cast<const bf16_t*>(a.cs);
    const bf16_t* __restrict__ acts = reinterpret_cast<const bf16_t*>(a.acts);
    const bf16_t* __restrict__ dext = reinterpret_cast<const bf16_t*>(a.dhs_ext);
    bf16_t* __restrict__ da = reinterpret_cast<bf16_t*>(a.da);
    bf16_t* __restrict__ rh = reinterpret_cast<bf16_t*>(a.rh);
    frag* myl = ulds + (size_t)w * NLc * 64 + l;

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
        dh[n] = a.dh_last ? *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)b * (a.dh_last_ld ? a.dh_last_ld : RH) + ub[n]) : z4;
        dc[n] = (CELL == MVAE_LSTM && a.dc_last)
                    ? *reinterpret_cast<const f32x4*>(a.dc_last + (size_t)b * (a.dh_last_ld ? a.dh_last_ld : RH) + ub[n]) : z4;
    }
    // queue of saved forward values for the step about to be processed (requested one step earlier)
    u16x4 qa[RNT][G], qs[RNT], qd[RNT], carry[RNT];   // gates; c_{t-1} (LSTM) / h_{t-1} (GRU); upstream grad; c_t
    // TILE16 addresses of this lane's 4-unit slot: saved gates (cols GH), cell states / upstream gradient (cols H)
    auto a_ptr = [&](int t, int n, int g) -> const bf16_t* {
        return acts + ((((size_t)t * tiles_per_step + blockIdx.x) * (GH / 16) + g * (RH / 16) + w * RNT + n) * 64 + l) * 4;
    };
    auto h_ptr = [&](const bf16_t* base, int t, int n) -> const bf16_t* {
        return base + ((((size_t)t * tiles_per_step + blockIdx.x) * (RH / 16) + w * RNT + n) * 64 + l) * 4;
    };
    {
        const int t = T - 1;
#pragma unroll
        for (int n = 0; n < RNT; ++n) {
#pragma unroll
            for (int g = 0; g < G; ++g) qa[n][g] = *reinterpret_cast<const u16x4*>(a_ptr(t, n, g));
            if (CELL == MVAE_LSTM) {
                qs[n] = *reinterpret_cast<const u16x4*>(h_ptr(cs, t, n));
                carry[n] = *reinterpret_cast<const u16x4*>(h_ptr(cs, t + 1, n));
            }
            if (CELL == MVAE_GRU) qs[n] = *reinterpret_cast<const u16x4*>(hs + ((size_t)t * B + b) * RH + ub[n]);
            if (HAS_EXT) qd[n] = *reinterpret_cast<const u16x4*>(h_ptr(dext, t, n));
        }
    }
    vm_drain();
    lds_barrier();

    // B fragment (k-group ks of the da tile) for this lane: row r, 8 columns from ks*32 + q*8
    auto bfrag = [&](int ks) -> frag { return *reinterpret_cast<const frag*>(dabuf + sw_off<GH>(r, ks * 32 + q * 8)); };
    auto put_da = [&](int col, f32x4 v) { *reinterpret_cast<u16x4*>(dabuf + sw_off<GH>(r, col)) = pack4(v); };

    for (int t = T - 1; t >= 0; --t) {
        const int tp = t > 0 ? t - 1 : 0;                           // the step requested during this one
        bf16_t* da_rows = da + ((size_t)t * B + blockIdx.x * 16) * GH;   // this WG's 16 rows of da[t], row-major
        const int tstep = T - 1 - t;
        STAMP(0);
        // this step's loads were issued a full step ago; the youngest store precedes the previous MFMA phase
        vm_drain();
        STAMP(1);
#pragma unroll
        for (int n = 0; n < RNT; ++n) {
            if (G == 4) pin4(qa[n][0], qa[n][G > 1 ? 1 : 0], qa[n][G > 2 ? 2 : 0], qa[n][G > 3 ? 3 : 0]);
            else { pin1(qa[n][0]); pin1(qa[n][G > 1 ? 1 : 0]); pin1(qa[n][G > 2 ? 2 : 0]); }
            pin1(qs[n]);
            if (HAS_EXT) pin1(qd[n]);
        }
        f32x4 acc[RNT];
#pragma unroll
        for (int n = 0; n < RNT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (CELL == MVAE_LSTM) {
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                const f32x4 ig = unpack4(qa[n][0]), fg = unpack4(qa[n][G > 1 ? 1 : 0]), gg = unpack4(qa[n][G > 2 ? 2 : 0]),
                            og = unpack4(qa[n][G > 3 ? 3 : 0]);
                const f32x4 c = unpack4(carry[n]), cp = unpack4(qs[n]);
                f32x4 d = dh[n];
                if (HAS_EXT) d += unpack4(qd[n]);
                f32x4 di, df, dg, dO;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float tc = tanh_fast(c[i]);
                    const float dct = dc[n][i] + d[i] * og[i] * (1.0f - tc * tc);
                    di[i] = dct * gg[i] * dhard_sigmoid(ig[i]);
                    df[i] = dct * cp[i] * dhard_sigmoid(fg[i]);
                    dg[i] = dct * ig[i] * (1.0f - gg[i] * gg[i]);
                    dO[i] = d[i] * tc * dhard_sigmoid(og[i]);
                    dc[n][i] = dct * fg[i];
                }
                put_da(ub[n], di);
                put_da(RH + ub[n], df);
                put_da(2 * RH + ub[n], dg);
                put_da(3 * RH + ub[n], dO);
                carry[n] = qs[n];                         // c_{t-1} is the next step's c_t
#pragma unroll
                for (int g = 0; g < G; ++g) aload8(qa[n][g], a_ptr(tp, n, g));
                aload8(qs[n], h_ptr(cs, tp, n));
                if (HAS_EXT) aload8(qd[n], h_ptr(dext, tp, n));
            }
            res_barrier();
            tile_rows_to_global<GH>(dabuf, da_rows, w, l);      // da[t]: whole rows, issued ahead of the MFMA phase
#define BF_(ks) bfrag(ks)
            RES_PHASE(acc[0], acc[1], acc[2], acc[3], 0, S2, BF_, RES_NOHOOK);
#pragma unroll
            for (int n = 0; n < RNT; ++n) dh[n] = acc[n];
        } else if (CELL == MVAE_GRU) {
            f32x4 z[RNT], rr[RNT], hp[RNT], hh[RNT], d[RNT];
#pragma unroll
            for (int n = 0; n < RNT; ++n) {
                z[n] = unpack4(qa[n][0]);
                rr[n] = unpack4(qa[n][G > 1 ? 1 : 0]);
                hh[n] = unpack4(qa[n][G > 2 ? 2 : 0]);
                hp[n] = unpack4(qs[n]);
                d[n] = dh[n];
                if (HAS_EXT) d[n] += unpack4(qd[n]);
                f32x4 dah;
#pragma unroll
                for (int i = 0; i < 4; ++i) dah[i] = d[n][i] * (1.0f - z[n][i]) * (1.0f - hh[n][i] * hh[n][i]);
                put_da(2 * RH + ub[n], dah);
                if (rh) *reinterpret_cast<u16x4*>(rhtile + sw_off<RH>(r, ub[n])) = pack4(rr[n] * hp[n]);
#pragma unroll
                for (int g = 0; g < G; ++g) aload8(qa[n][g], a_ptr(tp, n, g));
                aload8(qs[n], hs + ((size_t)tp * B + b) * RH + ub[n]);
                if (HAS_EXT) aload8(qd[n], h_ptr(dext, tp, n));
            }
            STAMP(2);
            res_barrier();
            STAMP(3);
            if (rh) tile_rows_to_global<RH>(rhtile, rh + ((size_t)t * B + blockIdx.x * 16) * RH, w, l);
            constexpr int SH = RH / 32;
#define BFH_(gi) bfrag(2 * SH + (gi))
            RES_PHASE(acc[0], acc[1], acc[2], acc[3], 2 * SH * 4, SH, BFH_, RES_NOHOOK);
            STAMP(4);
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
                put_da(ub[n], daz);
                put_da(RH + ub[n], dar);
            }
            STAMP(5);
            res_barrier();
            STAMP(6);
            tile_rows_to_global<GH>(dabuf, da_rows, w, l);
#ifndef BF_
#define BF_(ks) bfrag(ks)
#endif
            RES_PHASE(acc[0], acc[1], acc[2], acc[3], 0, 2 * SH, BF_, RES_NOHOOK);
#pragma unroll
            for (int n = 0; n < RNT; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[n][i] = d[n][i] * z[n][i] + drh[n][i] * rr[n][i] + acc[n][i];
        }
        STAMP(7);
        res_barrier();
        STAMP(8);
    }
    const int ldd = a.dh0_ld ? a.dh0_ld : RH;
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        if (a.dh0) *reinterpret_cast<f32x4*>(a.dh0 + (size_t)b * ldd + ub[n]) = dh[n];
        if (CELL == MVAE_LSTM && a.dc0) *reinterpret_cast<f32x4*>(a.dc0 + (size_t)b * ldd + ub[n]) = dc[n];
    }
    vm_drain();
}

// ---------------------------------------------------------------------------------------------------------
// LSTM backward, slot-interleaved
// ---------------------------------------------------------------------------------------------------------
// Step t:  E  gate gradients of the wave's 64 units from dh_t, dc_t and the saved forward values -> da tile (LDS)
//          M  dh_{t-1} = U^T-fragments x da tile: 128 single-MFMA slots; their gaps carry the row-major copy of the
//             da tile to HBM and the loads of step t-1's saved values (the registers those land in were consumed in
//             E, and a whole M phase - more than an HBM round trip - passes before the next E needs them)
// Fragment classes in order of use: T (first k-group; streamed from L2 during E into the registers that stage
// LDS-resident fragments during M), A accumulator registers, V vector registers, L LDS.
template <bool HAS_EXT, int NA, int NV>
__device__ __forceinline__ void lstm_bwd_il_body(const mvae_rnn_bwd_args& a, const unsigned bx) {
    constexpr int G = 4, GH = G * RH, S2 = GH / 32, FPW = RNT * S2;
    constexpr int NT = 4, NLc = FPW - NT - NA - NV;
    static_assert(NA % 4 == 0 && NV % 4 == 0 && NLc >= 0 && NA <= 64 && NLc <= 32, "fragment classes");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* dabuf = smem;                                              // [16][GH] bf16, swizzled (32 KiB)
    frag* ulds = reinterpret_cast<frag*>(smem + 16 * GH * 2);                 // [4][NL][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.ut_pack);
    frag* myl = ulds + (size_t)w * NLc * 64 + l;

    // fragment f = ks*4 + n  (A rows = hidden units of tile w*4+n, k-group ks over the gate columns)
    auto frag_ptr = [&](int f) -> const frag* { return up + (size_t)((w * RNT + (f & 3)) * S2 + (f >> 2)) * 64 + l; };
    frag ua[NA > 0 ? NA : 4], uv[NV > 0 ? NV : 4];
    static_for<0, (FPW - NT) / 4>(SF_LAMBDA(ic) {
        constexpr int i = decltype(ic)::value * 4, f = i + NT;
        if constexpr (i < NA) load4_agpr_nowait(ua[i], ua[i + 1], ua[i + 2], ua[i + 3], frag_ptr(f), frag_ptr(f + 1), frag_ptr(f + 2), frag_ptr(f + 3));
        else if constexpr (i < NA + NV) {
            uv[i - NA] = *frag_ptr(f); uv[i - NA + 1] = *frag_ptr(f + 1); uv[i - NA + 2] = *frag_ptr(f + 2); uv[i - NA + 3] = *frag_ptr(f + 3);
        } else {
            myl[(size_t)(i - NA - NV) * 64] = *frag_ptr(f); myl[(size_t)(i - NA - NV + 1) * 64] = *frag_ptr(f + 1);
            myl[(size_t)(i - NA - NV + 2) * 64] = *frag_ptr(f + 2); myl[(size_t)(i - NA - NV + 3) * 64] = *frag_ptr(f + 3);
        }
    });
    const frag* tsrc = frag_ptr(0);             // T fragments: tiles 0..3 of k-group 0 are S2*64 fragments apart

    const int ub0 = w * 64 + q * 4;
    unsigned lane16 = (unsigned)l * 16u;     // this lane's 16 bytes of a TILE16P pair / of a row-major copy chunk; half of it: TILE16
    // swizzled da-tile offsets (bytes), XOR-linear in (gate, tile) / k-group / copy chunk:
    //   this lane's 4 values of (gate g, tile n): da0 ^ (g*512 + n*32)      B fragment ks: (bb0 ^ ((ks&3) << 6)) + 256*(ks >> 2)
    //   copy chunk j (row 4w + j/2, 16-byte chunk (j&1)*64 + l): (tc0 ^ ((j>>1) << 4)) + (j>>1)*2048 + (j&1)*1024
    unsigned da0 = (unsigned)r * (GH * 2) + ((((unsigned)w * 8u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    unsigned bb0 = (unsigned)r * (GH * 2) + (((unsigned)q ^ (unsigned)r) << 4);     // B fragment ks: (bb0 ^ ((ks & 3) << 6)) + 256 * (ks >> 2)
    unsigned tc0 = 4u * w * (GH * 2) + ((((unsigned)l) ^ (4u * w)) << 4);

    f32x4 dh[RNT], dc[RNT];
    const int ldl = a.dh_last_ld ? a.dh_last_ld : RH;
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        dh[n] = a.dh_last ? *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)b * ldl + ub0 + 16 * n) : z4;
        dc[n] = a.dc_last ? *reinterpret_cast<const f32x4*>(a.dc_last + (size_t)b * ldl + ub0 + 16 * n) : z4;
    }
    // wave-uniform running pointers for step t-1 (the step whose values are fetched during step t)
    gbyte *acts_p[G], *cs_p, *dx_p, *da_p;
    const size_t dx_step = tps * (RH / 16) * 512, da_step = (size_t)B * GH * 2;
#if RES_WGMAJOR
    const size_t acts_step = (size_t)(GH / 32) * 1024, cs_step = (size_t)(RH / 32) * 1024;
    const size_t acts_bT = ((size_t)bx * T + (T - 1)) * (GH / 32), cs_bT = ((size_t)bx * (T + 1) + (T - 1)) * (RH / 32);
#else
    const size_t acts_step = tps * (GH / 16) * 512, cs_step = dx_step;
    const size_t acts_bT = ((size_t)(T - 1) * tps + bx) * (GH / 32), cs_bT = ((size_t)(T - 1) * tps + bx) * (RH / 32);
#endif
#pragma unroll
    for (int g = 0; g < G; ++g)
        acts_p[g] = to_global(a.acts) + (acts_bT + g * (RH / 32) + w * 2) * 1024;
    cs_p = to_global(a.cs) + (cs_bT + w * 2) * 1024;      // c_{t-1} of step T-1
    dx_p = to_global(a.dhs_ext) + (((size_t)(T - 1) * tps + bx) * (RH / 16) + w * RNT) * 512;
    da_p = to_global(a.da) + ((size_t)(T - 1) * B + bx * 16) * (GH * 2) + (size_t)w * 4 * (GH * 2);

    // pipelined stack bookkeeping: chunk pk (first step plo) is the one being processed; one division, before the loop
    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = __builtin_amdgcn_readfirstlane(cs_steps ? (T - 1) / cs_steps : 0), plo = pk * cs_steps;   // (the division runs on the VALU)
    if (HAS_EXT && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready + pk, wait_value, a.status, 2u);
    int pwait = (cs_steps && a.wait_ready && plo > 0) ? plo : -1;      // step at whose start chunk pk-1 must be ready (-1: never)
    int psig = (cs_steps && a.signal_done) ? plo : -1;                 // step after which chunk pk is published
    // saved forward values of the step about to be processed; acts / cs are TILE16P: element 0..3 of a 16-byte lane
    // chunk belong to tile 2j, 4..7 to tile 2j+1
    u16x8 qa[2][G], qs[2], carry[2];       // gates; c_{t-1}; c_t
    u16x4 qd[RNT];                         // upstream gradient (TILE16)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < G; ++g) qa[j][g] = *reinterpret_cast<const g_u16x8*>(acts_p[g] + j * 1024 + lane16);
        qs[j] = *reinterpret_cast<const g_u16x8*>(cs_p + j * 1024 + lane16);
        carry[j] = *reinterpret_cast<const g_u16x8*>(cs_p + cs_step + j * 1024 + lane16);
    }
    if (HAS_EXT) {
#pragma unroll
        for (int n = 0; n < RNT; ++n) qd[n] = *reinterpret_cast<const g_u16x4*>(dx_p + n * 512 + (lane16 >> 1));
    }
    // from here on the pointers address step t-1 while step t runs
#pragma unroll
    for (int g = 0; g < G; ++g) acts_p[g] -= (T > 1 ? acts_step : 0);
    cs_p -= (T > 1 ? cs_step : 0);
    dx_p -= (T > 1 ? dx_step : 0);

    float fillr = 0.f; (void)fillr;      // (ABL_FILL probe)
    const s16x4 ident = identity_fragment(l);
    frag bq[2], lt[4];      // B fragments: one k-group ahead (a third ring slot costs 4 registers this kernel lacks)
    vm_drain();
    lds_barrier();

    for (int t = T - 1; t >= 0; --t) {
        const int tstep = T - 1 - t;
        (void)tstep;
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(acts_p[3]); pins(cs_p); pins(da_p);
        if (HAS_EXT) pins(dx_p);
        pinu(da0); pinu(tc0); pinu(bb0);
        STAMP(0);
        // pipelined stack: the upstream gradient of step t-1 is requested during this step's M phase
        if (HAS_EXT) wave_wait_ge_if(t, pwait, a.wait_ready + (pk - 1), wait_value, a.status, 2u);
        // ---- E: everything requested during the previous M phase has had that whole phase to arrive (the compiler's
        // counted waits for the loads; no drain: the da stores issued at the end of that phase may still be in flight)
        if (GB_DRAIN) vm_drain();
        STAMP(1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pinq(qa[j][0]); pinq(qa[j][1]); pinq(qa[j][2]); pinq(qa[j][3]); pinq(qs[j]);
        }
        if (HAS_EXT) {
#pragma unroll
            for (int n = 0; n < RNT; ++n) pin1(qd[n]);
        }
        // T fragments for this step's first k-group, into the registers that are idle until the M phase
#pragma unroll
        for (int n = 0; n < 4; ++n) lt[n] = tsrc[(size_t)n * S2 * 64];
#pragma unroll
        for (int n = 0; n < RNT; ++n) {
            auto half = [&](const u16x8& v) -> f32x4 {
                const int o = (n & 1) * 4;
                if (BWL_MFMA_WIDEN > 1)
                    return widen4(ident, (n & 1) ? __builtin_shufflevector(v, v, 4, 5, 6, 7) : __builtin_shufflevector(v, v, 0, 1, 2, 3),
                                  f32x4{0.f, 0.f, 0.f, 0.f});
                return f32x4{bf2f(v[o]), bf2f(v[o + 1]), bf2f(v[o + 2]), bf2f(v[o + 3])};
            };
            const f32x4 ig = half(qa[n >> 1][0]), fg = half(qa[n >> 1][1]), gg = half(qa[n >> 1][2]), og = half(qa[n >> 1][3]);
            const f32x4 c = half(carry[n >> 1]), cp = half(qs[n >> 1]);
            f32x4 d = dh[n];
            if (HAS_EXT) d = BWL_MFMA_WIDEN ? widen4(ident, qd[n], d) : d + unpack4(qd[n]);
            // whole-vector expressions: no MFMA is in flight in this phase, so packed f32 instructions (two elements
            // each) are pure gain here - unlike in the MFMA gaps, where they cost issue slots
            f32x4 di, df, dg, dO;
            if (ABL_NOMATH) { di = d; df = d; dg = d; dO = d; dc[n] = d; }     // (timing ablation: the E phase without its arithmetic)
            else {
                const f32x4 tc = tanh_fast4(c);
                const f32x4 dct = dc[n] + d * og * (BWL_ASM_1MSQ ? one_minus_sq4(tc) : 1.0f - tc * tc);
#if BWL_OLD_DHS
                di = dct * (gg * dhard_sigmoid4(ig));
                df = dct * (cp * dhard_sigmoid4(fg));
                dO = d * (tc * dhard_sigmoid4(og));
#else
                const f32x4 dctK = dct * DHS_BIG, dK = d * DHS_BIG;       // (powers of two: exact; |dct| < 2^27 or the step is lost anyway)
                di = (dctK * gg) * sat4(ig);
                df = (dctK * cp) * sat4(fg);
                dO = (dK * tc) * sat4(og);
#endif
                dg = dct * ig * (BWL_ASM_1MSQ ? one_minus_sq4(gg) : 1.0f - gg * gg);
                dc[n] = dct * fg;
            }
            // (da0's 16-byte chunk index is < 32: the gate's 512 bytes never meet a set bit, so they are an instruction immediate
            //  instead of an exclusive-or per store: 12 v_xor + 12 v_add per step less)
            unsigned char* dan = dabuf + (da0 ^ (unsigned)(n * 32));
            *reinterpret_cast<u16x4*>(dan + 0 * 512) = pack4(di);
            *reinterpret_cast<u16x4*>(dan + 1 * 512) = pack4(df);
            *reinterpret_cast<u16x4*>(dan + 2 * 512) = pack4(dg);
            *reinterpret_cast<u16x4*>(dan + 3 * 512) = pack4(dO);
            if (n & 1) carry[n >> 1] = qs[n >> 1];      // c_{t-1} is the next step's c_t
#if BWL_E_TILE_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        STAMP(2);
        vm_drain();                                   // the T fragments (L2 hits issued a whole E phase ago)
        pinq(lt[0]); pinq(lt[1]); pinq(lt[2]); pinq(lt[3]);
        if (ABL_NOBAR1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else res_barrier();
        res_skew(w);
        STAMP(3);

        // ---- M -------------------------------------------------------------------------------------------------
        f32x4 acc[RNT];
#if !BWL_MFMA_FIRST
#pragma unroll
        for (int n = 0; n < RNT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
        bq[0] = *reinterpret_cast<const frag*>(dabuf + bb0);
        static_for<0, FPW>(SF_LAMBDA(sc) {
            constexpr int sl = decltype(sc)::value, gi = sl >> 2, n = sl & 3, ci = sl - NT;
            if constexpr (n == 0 && gi + 1 < S2 && !(ABL_NOB && gi >= 1))
                bq[(gi + 1) & 1] = *reinterpret_cast<const frag*>(dabuf + (bb0 ^ (((gi + 1) & 3) << 6)) + 256 * ((gi + 1) >> 2));
#if BWL_MFMA_FIRST
            if constexpr (sl == 0) asm volatile("s_nop 1");
            if constexpr (sl < NT) mfma1_first(acc[n], lt[n], bq[gi & 1]);       // (the accumulators start here: C = 0 inline)
#else
            if constexpr (sl == 0) asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            if constexpr (sl < NT) mfma1<false>(acc[n], lt[n], bq[gi & 1]);
#endif
            else if constexpr (ci < NA) mfma1<true>(acc[n], ua[ci < NA ? ci : 0], bq[gi & 1]);
            else if constexpr (ci < NA + NV) mfma1<false>(acc[n], uv[(ci >= NA && ci < NA + NV) ? ci - NA : 0], bq[gi & 1]);
            else mfma1<false>(acc[n], lt[n], bq[gi & 1]);
            constexpr int cn = ci + 4;      // the next group's fragment for this tile
            if constexpr (sl + 4 < FPW && cn >= NA + NV && !ABL_NOL) lt[n] = myl[(size_t)(cn - NA - NV) * 64];
            __builtin_amdgcn_sched_barrier(0);
            // fillers.  Slots 1..40: step t-1's saved values, one load per 3 slots; then the row-major copy of the da
            // tile, 8 chunks per lane staged through the (until the last groups idle) lt registers.
            if constexpr (!ABL_NOX && sl >= 1 && sl < 1 + BWL_LOAD_STRIDE * 14 && (sl - 1) % BWL_LOAD_STRIDE == 0) {
                constexpr int k = (sl - 1) / BWL_LOAD_STRIDE;                // 0..7 gates (pair j = k/4), 8..9 c_{t-1}, 10..13 upstream gradient
                if constexpr (k < 8) {
                    pinu(lane16);
                    qa[k >> 2][k & 3] = *reinterpret_cast<const g_u16x8*>(acts_p[k & 3] + (k >> 2) * 1024 + lane16);
                } else if constexpr (k < 10) {
                    pinu(lane16);
                    qs[k - 8] = *reinterpret_cast<const g_u16x8*>(cs_p + (k - 8) * 1024 + lane16);
                } else if constexpr (HAS_EXT) {
                    unsigned l8 = lane16 >> 1;
                    pinu(l8);
                    qd[k - 10] = *reinterpret_cast<const g_u16x4*>(dx_p + (k - 10) * 512 + l8);
                }
            }
            if constexpr (!ABL_NOTRG && sl >= BWL_COPY_SLOT && sl < BWL_COPY_SLOT + BWL_COPY_STRIDE * 10 && (sl - BWL_COPY_SLOT) % BWL_COPY_STRIDE == 0) {
                constexpr int j = (sl - BWL_COPY_SLOT) / BWL_COPY_STRIDE;               // read chunk j (j < 8), store chunk j - 2
                if constexpr (j >= 2) {
                    constexpr int js = j - 2;
                    pinu(lane16);
                    store16_wt(da_p, lane16 + (unsigned)((js >> 1) * 2048 + (js & 1) * 1024), lt[js & 3]);
                }
                if constexpr (j < 8)
                    lt[j & 3] = *reinterpret_cast<const frag*>(dabuf + (tc0 ^ ((j >> 1) << 4)) + (j >> 1) * 2048 + (j & 1) * 1024);
            }
            if constexpr (ABL_FILL > 0) {       // (timing probe: ABL_FILL semantically empty VALU instructions per MFMA slot)
                if constexpr (ABL_FILL > 0) asm volatile("v_max_f32 %0, %0, %0" : "+v"(fillr));
                if constexpr (ABL_FILL > 1) asm volatile("v_max_f32 %0, %0, %0" : "+v"(fillr));
                if constexpr (ABL_FILL > 2) asm volatile("v_max_f32 %0, %0, %0" : "+v"(fillr));
                if constexpr (ABL_FILL > 3) asm volatile("v_max_f32 %0, %0, %0" : "+v"(fillr));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
        for (int n = 0; n < RNT; ++n) dh[n] = acc[n];
#pragma unroll
        for (int g = 0; g < G; ++g) acts_p[g] -= (t > 1 ? acts_step : 0);
        cs_p -= (t > 1 ? cs_step : 0);
        if (HAS_EXT) dx_p -= (t > 1 ? dx_step : 0);
        da_p -= da_step;
        STAMP(7);
        if (ABL_NOBAR2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else res_barrier();
        STAMP(8);
        // pipelined stack: da of steps >= t is out; chunk t / cs is complete when t is its first step
        wave_signal_done_if<false>(t, psig, a.signal_done + pk);
        {   // scalar bookkeeping (s_cselect, no branch): next chunk once its first step has been processed
            const bool adv = cs_steps && t == plo;
            pk -= adv ? 1 : 0;
            plo -= adv ? cs_steps : 0;
            pwait = (a.wait_ready && plo > 0) ? plo : -1;
            psig = a.signal_done ? plo : -1;
        }
    }
    const int ldd = a.dh0_ld ? a.dh0_ld : RH;
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        if (a.dh0) *reinterpret_cast<f32x4*>(a.dh0 + (size_t)b * ldd + ub0 + 16 * n) = dh[n];
        if (a.dc0) *reinterpret_cast<f32x4*>(a.dc0 + (size_t)b * ldd + ub0 + 16 * n) = dc[n];
    }
    vm_drain();
}
template <bool HAS_EXT, int NA, int NV>
__global__ __launch_bounds__(256, 1) void lstm_bwd_il_k(const mvae_rnn_bwd_args a) {
    lstm_bwd_il_body<HAS_EXT, NA, NV>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// GRU backward, slot-interleaved
// ---------------------------------------------------------------------------------------------------------
// Step t (descending), d = dh_t (+ the gradient from the layer above), hp = h_{t-1}, saved z, r, hh:
//   E1  dah = d (1-z)(1-hh^2) -> da tile (candidate columns);  r*hp -> rh tile (left operand of the candidate dU GEMM)
//   M1  drh = Uh^T-fragments x dah: 32 single-MFMA slots (fragments staged from LDS - the phase is LDS-bandwidth bound,
//       its gaps are wide); they carry daz = d (hp-hh) hs'(z) -> da tile (z columns), the row-major copies of the rh
//       tile and of the da tile's candidate columns, and the first 8 loads of step t-1's saved values
//   E2  dar = drh hp hs'(r) -> da tile (r columns);  part = d z + drh r
//   M2  dh_{t-1} = part + Uzr^T-fragments x [daz|dar]: 64 slots (accumulator-register fragments); their gaps carry
//       the other 6 loads (early: a load needs ~1500 cycles) and the row-major copy of the da tile's z, r columns
// 64 + 32 fragments per wave: accumulator registers + LDS (24 KiB da tile + 8 KiB rh tile + 128 KiB = all of it).
// A memory instruction occupies the CU's address unit for ~27 cycles whatever its width (4 waves: ~108 cycles per wave
// and instruction), hence 16-byte accesses wherever the layout allows: acts TILE16P, copies in 16-byte chunks.
#ifndef GB_NOLOAD
#define GB_NOLOAD 0
#endif
#ifndef GB_NOCOPY
#define GB_NOCOPY 0
#endif
#ifndef GB_NODAZ
#define GB_NODAZ 0
#endif
template <bool HAS_EXT>
__device__ __forceinline__ void gru_bwd_il_body(const mvae_rnn_bwd_args& a, const unsigned bx) {
    constexpr int G = 3, GH = G * RH, S2 = GH / 32, NLc = RNT * (RH / 32);      // 24 k-groups; 32 LDS fragments per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* dabuf = smem;                                              // [16][GH] bf16, swizzled (24 KiB)
    unsigned char* rhbuf = smem + 16 * GH * 2;                                // [16][RH]
    frag* ulds = reinterpret_cast<frag*>(smem + 16 * GH * 2 + 16 * RH * 2);   // [4][NL][64]
    const int tid = threadIdx.x, l = tid & 63, q = l >> 4, r = l & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int b = bx * 16 + r;
    const size_t tps = (size_t)(B / 16);
    const frag* __restrict__ up = reinterpret_cast<const frag*>(a.ut_pack);
    frag* myl = ulds + (size_t)w * NLc * 64 + l;
    // packed fragment (unit tile w*4+n, k-group ks over the gate columns); M2 uses ks 0..15 (z, r), M1 ks 16..23
    auto src_frag = [&](int n, int ks) -> const frag* { return up + (size_t)((w * RNT + n) * S2 + ks) * 64 + l; };
    frag ua[64];
    static_for<0, 16>(SF_LAMBDA(ic) {
        constexpr int ks = decltype(ic)::value;
        load4_agpr_nowait(ua[ks * 4], ua[ks * 4 + 1], ua[ks * 4 + 2], ua[ks * 4 + 3], src_frag(0, ks), src_frag(1, ks),
                          src_frag(2, ks), src_frag(3, ks));
    });
#pragma unroll
    for (int i = 0; i < NLc; ++i) myl[(size_t)i * 64] = *src_frag(i & 3, 16 + (i >> 2));

    const int ub0 = w * 64 + q * 4;
    unsigned lane8 = (unsigned)l * 8u, lane16 = (unsigned)l * 16u;
    // da tile (row stride 1536 B): this lane's 4 values of (gate g, tile n) at da_row + (da_ch ^ ((g*32 + n*2) << 4));
    // B fragment of k-group ks at b_row + (b_ch ^ (ks << 6))
    unsigned da_row = (unsigned)r * (GH * 2) + ((unsigned)q & 1u) * 8u;
    unsigned da_ch = (((unsigned)w * 8u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4;
    unsigned b_row = (unsigned)r * (GH * 2), b_ch = ((unsigned)q ^ (unsigned)r) << 4;
    // rh tile (row stride 512 B), as the h tiles of the forward kernels
    unsigned rw0 = (unsigned)r * 512u + ((((unsigned)w * 8u + ((unsigned)q >> 1)) ^ (unsigned)r) << 4) + ((unsigned)q & 1u) * 8u;
    const unsigned row0 = 4u * w + ((unsigned)l >> 5), ch0 = (unsigned)l & 31u;
    unsigned tl0 = row0 * 512u + ((ch0 ^ row0) << 4), tg0 = row0 * 512u + ch0 * 16u;      // rh tile copy (2 chunks per lane)
    unsigned hp_off = (unsigned)r * (RH * 2) + (unsigned)q * 8u;                          // h_{t-1}: row r of the tile, row-major

    f32x4 dh[RNT];
    const int ldl = a.dh_last_ld ? a.dh_last_ld : RH;
#pragma unroll
    for (int n = 0; n < RNT; ++n) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        dh[n] = a.dh_last ? *reinterpret_cast<const f32x4*>(a.dh_last + (size_t)b * ldl + ub0 + 16 * n) : z4;
    }
    gbyte *acts_p[G], *hs_p, *dx_p, *da_p, *rh_p;
    const size_t acts_step = tps * (GH / 32) * 1024, dx_step = tps * (RH / 16) * 512, hs_step = (size_t)B * RH * 2,
                 da_step = (size_t)B * GH * 2;
#pragma unroll
    for (int g = 0; g < G; ++g)
        acts_p[g] = to_global(a.acts) + (((size_t)(T - 1) * tps + bx) * (GH / 32) + g * (RH / 32) + w * 2) * 1024;
    hs_p = to_global(a.hs) + ((size_t)(T - 1) * B + bx * 16) * (RH * 2) + w * 128;             // h_{t-1} = slot t
    dx_p = to_global(a.dhs_ext) + (((size_t)(T - 1) * tps + bx) * (RH / 16) + w * RNT) * 512;
    da_p = to_global(a.da) + ((size_t)(T - 1) * B + bx * 16) * (GH * 2) + (size_t)w * 4 * (GH * 2);
    rh_p = to_global(a.rh) + ((size_t)(T - 1) * B + bx * 16) * (RH * 2);

    // pipelined stack bookkeeping, as lstm_bwd_il_k: chunk pk (first step plo) is the one being processed
    const int cs_steps = a.chunk_steps;
    const unsigned wait_value = a.wait_value ? a.wait_value : 1u;
    int pk = __builtin_amdgcn_readfirstlane(cs_steps ? (T - 1) / cs_steps : 0), plo = pk * cs_steps;
    if (HAS_EXT && cs_steps && a.wait_ready) wave_wait_ge(a.wait_ready + pk, wait_value, a.status, 2u);
    int pwait = (cs_steps && a.wait_ready && plo > 0) ? plo : -1;      // step at whose start chunk pk-1 must be ready (-1: never)
    int psig = (cs_steps && a.signal_done) ? plo : -1;                 // step after which chunk pk is published

    // saved values of the step about to be processed: z, r, hh as TILE16P pairs (elements 0..3 tile 2j, 4..7 tile 2j+1);
    // h_{t-1} (row-major) and the upstream gradient (TILE16) per tile
    // z, hh and h_{t-1} are requested TWO steps ahead into alternating buffers (an HBM round trip beside the gradient GEMMs
    // takes longer than the half step they used to get); r and the upstream gradient one step ahead
    u16x8 qzh[2][2][2], qr[2];            // [buffer][pair][z, hh];  r per pair
    u16x4 qp[2][RNT], qd[RNT];
    // the 14 loads: 0..3 h, 4..7 z and hh (pair 0, pair 1) of the step `back2` behind into buffer bc; 8, 9 r; 10..13 the
    // upstream gradient of the next step.  acts_p / hs_p / dx_p point at the next step (t-1); back2 = one more step
    // (0 where that step does not exist: the values are never used then)
    auto issue_load = [&](auto kc, auto bc, size_t back_a, size_t back_h) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, bf_ = decltype(bc)::value;
        if constexpr (GB_NOLOAD) return;
        if constexpr (k < 4) {
            pinu(hp_off);
            qp[bf_][k] = *reinterpret_cast<const g_u16x4*>(hs_p - back_h + k * 32 + hp_off);
        } else if constexpr (k < 8) {
            constexpr int pr = (k - 4) >> 1, zh = (k - 4) & 1;
            pinu(lane16);
            qzh[bf_][pr][zh] = *reinterpret_cast<const g_u16x8*>(acts_p[zh * 2] - back_a + pr * 1024 + lane16);
        } else if constexpr (k < 10) {
            pinu(lane16);
            qr[k - 8] = *reinterpret_cast<const g_u16x8*>(acts_p[1] + (k - 8) * 1024 + lane16);
        } else if constexpr (HAS_EXT) {
            pinu(lane8);
            qd[k - 10] = *reinterpret_cast<const g_u16x4*>(dx_p + (k - 10) * 512 + lane8);
        }
    };
    // step T-1 complete into buffer 0; then the pointers move to T-2 and its z, hh, h go to buffer 1
    static_for<0, 14>(SF_LAMBDA(kc) { issue_load(kc, std::integral_constant<int, 0>{}, 0, 0); });
#pragma unroll
    for (int g = 0; g < G; ++g) acts_p[g] -= (T > 1 ? acts_step : 0);
    hs_p -= (T > 1 ? hs_step : 0);
    dx_p -= (T > 1 ? dx_step : 0);
    static_for<0, 8>(SF_LAMBDA(kc) { issue_load(kc, std::integral_constant<int, 1>{}, 0, 0); });

    // row-major copy of the da tile: rows 4w..4w+3, 96 chunks of 16 bytes each.  Passes 0..3: lane l copies chunk l of row
    // 4w+pass (z, r columns); passes 4, 5: chunk 64 + l%32 of row 4w + 2(pass-4) + l/32 (candidate columns)
    auto da_read = [&](auto jc) __attribute__((always_inline)) -> frag {
        constexpr int j = decltype(jc)::value;
        const unsigned row = 4u * w + (j < 4 ? (unsigned)j : 2u * (j - 4) + ((unsigned)l >> 5));
        const unsigned ch = j < 4 ? (unsigned)l : 64u + ((unsigned)l & 31u);
        return *reinterpret_cast<const frag*>(dabuf + row * (GH * 2) + ((ch ^ row) << 4));
    };
    auto da_store = [&](auto jc, frag v) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const unsigned row = j < 4 ? (unsigned)j : 2u * (j - 4) + ((unsigned)l >> 5);
        const unsigned ch = j < 4 ? (unsigned)l : 64u + ((unsigned)l & 31u);
        unsigned go = row * (GH * 2) + ch * 16u;
        pinu(go);
        store16_wt(da_p, go, v);
    };

    frag bq[3], lt[4], cp[4];
    vm_drain();
    lds_barrier();
    // values of the step about to be processed, unpacked one step ahead (in the gaps of M2): z, hh, h_{t-1} and
    // w1 = (1-z)(1-hh^2), the factor of the candidate gradient
    f32x4 zv[RNT], hh[RNT], hp[RNT], w1[RNT];
    auto half = [&](const u16x8& v, int n) __attribute__((always_inline)) -> f32x4 {
        const int o = (n & 1) * 4;
        return f32x4{bf2f(v[o]), bf2f(v[o + 1]), bf2f(v[o + 2]), bf2f(v[o + 3])};
    };
    // (whole-vector expressions: the compiler emits packed f32 instructions, two elements each)
    auto pre_unpack = [&](int n, int part, int bf_) __attribute__((always_inline)) {      // 4 parts per tile
        if (part == 0) zv[n] = half(qzh[bf_][n >> 1][0], n);
        if (part == 1) hh[n] = half(qzh[bf_][n >> 1][1], n);
        if (part == 2) hp[n] = unpack4(qp[bf_][n]);
        if (part == 3) w1[n] = (1.0f - zv[n]) * (1.0f - hh[n] * hh[n]);
    };
    // 0.2 * [0 < y < 1] for two elements: med3(2^100 (y - y^2), 0, 0.2) as in dhard_sigmoid
    auto dhs2 = [&](f32x2 y) __attribute__((always_inline)) -> f32x2 {
        const f32x2 sq = y_minus_y2(y) * 0x1p100f;       // (the negation as an operand modifier: hipcc emits two v_xor for y - y*y)
        return f32x2{__builtin_amdgcn_fmed3f(sq[0], 0.0f, 0.2f), __builtin_amdgcn_fmed3f(sq[1], 0.0f, 0.2f)};
    };
#pragma unroll
    for (int n = 0; n < RNT; ++n)
#pragma unroll
        for (int part = 0; part < 4; ++part) pre_unpack(n, part, 0);

    // one time step; BF = the buffer that held this step's z, hh, h (consumed a step ago) = the one step t-2's go to
    auto step = [&](const int t, auto bfc) __attribute__((always_inline)) {
        constexpr int BF = decltype(bfc)::value;
        const int tstep = T - 1 - t;
        (void)tstep;
        const size_t back_a = t >= 2 ? acts_step : 0, back_h = t >= 2 ? hs_step : 0;
        pins(acts_p[0]); pins(acts_p[1]); pins(acts_p[2]); pins(hs_p); pins(da_p);
        if (HAS_EXT) pins(dx_p);
        if (a.rh) pins(rh_p);
        STAMP(0);
        // pipelined stack: the upstream gradient of step t-1 is requested during this step's MFMA phases
        if (HAS_EXT) wave_wait_ge_if(t, __builtin_amdgcn_readfirstlane(pwait), uniform_ptr(a.wait_ready + (pk - 1)), wait_value, a.status, 2u);
        if (GB_DRAIN) vm_drain();       // (else the compiler's counted waits: the copy stores of M2 may still be in flight)
        STAMP(1);
        pinq(qr[0]); pinq(qr[1]);
        if (HAS_EXT) {
#pragma unroll
            for (int n = 0; n < RNT; ++n) pin1(qd[n]);
        }
        // ---- E1: only what M1 waits for -------------------------------------------------------------------------------
        f32x4 d[RNT], rv[RNT];
#pragma unroll
        for (int n = 0; n < RNT; ++n) {
            d[n] = dh[n];
            if (HAS_EXT) d[n] += unpack4(qd[n]);
            *reinterpret_cast<u16x4*>(dabuf + da_row + (da_ch ^ ((2 * 32 + n * 2) << 4))) = pack4(d[n] * w1[n]);
        }
        STAMP(2);
        res_barrier();
        STAMP(3);
        // ---- M1: drh -----------------------------------------------------------------------------------------------
        f32x4 acc1[RNT];
        f32x2 dzq;
#pragma unroll
        for (int n = 0; n < RNT; ++n) acc1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        bq[0] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (16u << 6)));
        bq[1] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (17u << 6)));
#pragma unroll
        for (int n = 0; n < 4; ++n) lt[n] = myl[(size_t)n * 64];
        if (!GB_NOCOPY) {
            cp[2] = da_read(std::integral_constant<int, 4>{});
            cp[3] = da_read(std::integral_constant<int, 5>{});
        }
        asm volatile("s_nop 1" : "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
        static_for<0, 32>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value, ks = sl >> 2, n = sl & 3;
            if constexpr (n == 0 && ks + 2 < 8)
                bq[(ks + 2) % 3] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ ((16u + ks + 2) << 6)));
            mfma1<false>(acc1[n], lt[n], bq[ks % 3]);
            if constexpr (sl + 4 < 32) lt[n] = myl[(size_t)(sl + 4) * 64];
            __builtin_amdgcn_sched_barrier(0);
            // daz = d (hp - hh) hs'(z): tile sl/8, element pair (sl/4)%2, in two pieces (4 + 3 instructions)
            if constexpr (!GB_NODAZ && (sl & 1) == 0) {
                constexpr int tn = sl >> 3, e = ((sl >> 2) & 1) * 2, piece = (sl >> 1) & 1;
                if constexpr (piece == 0) {
                    const f32x2 dz = dhs2(lo_hi<e>(zv[tn]));
                    dzq = dz;
                } else {
                    const f32x2 v = lo_hi<e>(d[tn]) * (lo_hi<e>(hp[tn]) - lo_hi<e>(hh[tn])) * dzq;
                    hh[tn][e] = v[0];                          // hh is dead after E1: reuse as daz
                    hh[tn][e + 1] = v[1];
                    if constexpr (e == 2)
                        *reinterpret_cast<u16x4*>(dabuf + da_row + (da_ch ^ ((0 * 32 + tn * 2) << 4))) = pack4(hh[tn]);
                }
            }
            // r, and r*hp -> rh tile (nothing in this step's recurrence waits for it): tile (sl-1)/4 on slots 1, 5, 9, 13
            if constexpr ((sl & 3) == 1 && sl < 16) {
                constexpr int tn = sl >> 2;
                rv[tn] = half(qr[tn >> 1], tn);
                if (a.rh) *reinterpret_cast<u16x4*>(rhbuf + (rw0 ^ (tn << 5))) = pack4(rv[tn] * hp[tn]);
            }
            // memory events: loads 0..7 from slot 16 on (their registers - h_{t-1}, z, hh raw - are dead since the
            // second half of the previous M2), the candidate columns of the da tile before.  (All 14 loads here, to give
            // the late ones the ~2000 cycles an HBM load takes, was measured slower: the phase is LDS-bound but not idle.)
            if constexpr (!GB_NOCOPY && sl == 3) da_store(std::integral_constant<int, 4>{}, cp[2]);
            if constexpr (!GB_NOCOPY && sl == 10) da_store(std::integral_constant<int, 5>{}, cp[3]);
            if constexpr (sl >= 16 && (sl & 1) == 0) issue_load(std::integral_constant<int, ((sl - 16) >> 1)>{}, bfc, back_a, back_h);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
        STAMP(4);
        // ---- E2 -------------------------------------------------------------------------------------------------------
        f32x4 part[RNT];
#pragma unroll
        for (int n = 0; n < RNT; ++n) {
            const f32x2 s0 = dhs2(lo_hi<0>(rv[n])), s1 = dhs2(lo_hi<2>(rv[n]));
            const f32x4 dar = acc1[n] * hp[n] * f32x4{s0[0], s0[1], s1[0], s1[1]};
            part[n] = d[n] * zv[n] + acc1[n] * rv[n];
            *reinterpret_cast<u16x4*>(dabuf + da_row + (da_ch ^ ((1 * 32 + n * 2) << 4))) = pack4(dar);
        }
        STAMP(5);
        res_barrier();
        STAMP(6);
        // ---- M2: dh_{t-1} ------------------------------------------------------------------------------------------
        f32x4 acc2[RNT];
#pragma unroll
        for (int n = 0; n < RNT; ++n) acc2[n] = part[n];
        bq[0] = *reinterpret_cast<const frag*>(dabuf + b_row + b_ch);
        bq[1] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ (1u << 6)));
        if (a.rh) {
            cp[0] = *reinterpret_cast<const frag*>(rhbuf + tl0);
            cp[1] = *reinterpret_cast<const frag*>(rhbuf + (tl0 ^ 1056u));
        }
        if (!GB_NOCOPY) {
            lt[0] = da_read(std::integral_constant<int, 0>{});
            lt[1] = da_read(std::integral_constant<int, 1>{});
            lt[2] = da_read(std::integral_constant<int, 2>{});
            lt[3] = da_read(std::integral_constant<int, 3>{});
        }
        asm volatile("s_nop 1" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]));
        static_for<0, 64>(SF_LAMBDA(slc) {
            constexpr int sl = decltype(slc)::value, ks = sl >> 2, n = sl & 3;
            if constexpr (n == 0 && ks + 2 < 16)
                bq[(ks + 2) % 3] = *reinterpret_cast<const frag*>(dabuf + b_row + (b_ch ^ ((unsigned)(ks + 2) << 6)));
            mfma1<true>(acc2[n], ua[sl], bq[ks % 3]);
            __builtin_amdgcn_sched_barrier(0);
            // memory events: loads 8..13 first (r, upstream gradient: used from the next E1 on), then the copies of the
            // rh tile and of the da tile's z, r columns
            if constexpr ((sl & 3) == 1 && sl < 24) issue_load(std::integral_constant<int, 8 + (sl >> 2)>{}, bfc, 0, 0);
            if constexpr (sl == 26 || sl == 31) {
                if (a.rh) {
                    pinu(tg0);
                    store16_wt(rh_p, tg0 + (sl == 31 ? 1024u : 0u), cp[sl == 31 ? 1 : 0]);    // (read by the K-streaming dU GEMM while this kernel runs)
                }
            }
            if constexpr (!GB_NOCOPY && sl >= 36 && (sl - 36) % 5 == 0 && sl < 36 + 20)
                da_store(std::integral_constant<int, (sl - 36) / 5>{}, lt[(sl - 36) / 5]);
            // step t-1's z, hh, h_{t-2} (requested in M1) unpacked, and w1: one part per 2 slots in the second half
            if constexpr (sl >= 32 && (sl & 1) == 0) pre_unpack((sl - 32) >> 3, ((sl - 32) >> 1) & 3, BF ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 9" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]));
        STAMP(7);
#pragma unroll
        for (int n = 0; n < RNT; ++n) dh[n] = acc2[n];
#pragma unroll
        for (int g = 0; g < G; ++g) acts_p[g] -= (t > 1 ? acts_step : 0);
        hs_p -= (t > 1 ? hs_step : 0);
        if (HAS_EXT) dx_p -= (t > 1 ? dx_step : 0);
        da_p -= da_step;
        if (a.rh) rh_p -= hs_step;
        res_barrier();
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
    int t = T - 1;
    for (; t >= 1; t -= 2) {
        step(t, std::integral_constant<int, 0>{});
        step(t - 1, std::integral_constant<int, 1>{});
    }
    if (t == 0) step(0, std::integral_constant<int, 0>{});
    const int ldd = a.dh0_ld ? a.dh0_ld : RH;
#pragma unroll
    for (int n = 0; n < RNT; ++n)
        if (a.dh0) *reinterpret_cast<f32x4*>(a.dh0 + (size_t)b * ldd + ub0 + 16 * n) = dh[n];
    vm_drain();
}
template <bool HAS_EXT>
__global__ __launch_bounds__(256, 1) void gru_bwd_il_k(const mvae_rnn_bwd_args a) {
    gru_bwd_il_body<HAS_EXT>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------------------
// fragment placement per kernel: A in accumulator registers, V in vector registers, the rest in LDS (per wave)
#ifndef RES_LSTM_FA
#define RES_LSTM_IA 64
#define RES_LSTM_IV 24
#define RES_LSTM_JA 64
#define RES_LSTM_JV 28
#define RES_LSTM_FA 64
#define RES_LSTM_FV 28
#define RES_LSTM_FVS 32
#define RES_LSTM_BA 64
#define RES_LSTM_BV 32
#define RES_GRU_FA 64
#define RES_GRU_FV 8
#define RES_GRU_BA 64
#define RES_GRU_BV 8
#endif
#ifndef RES_LSTM_PIPELINE
#define RES_LSTM_PIPELINE 1
#endif

template <int CELL> struct res_cfg;
template <> struct res_cfg<MVAE_LSTM> {   // 128 fragments per wave
    static constexpr int IA = RES_LSTM_IA, IV = RES_LSTM_IV;     // slot-interleaved forward (+4 streamed, rest LDS)
    static constexpr int JA = RES_LSTM_JA, JV = RES_LSTM_JV;     // slot-interleaved backward (+4 streamed, 32 LDS)
    static constexpr int FA = RES_LSTM_FA, FV = RES_LSTM_FV;     // rest in LDS (16 KiB h tiles + <= 144 KiB)
    static constexpr int FV_SCALAR = RES_LSTM_FVS;               // 8 KiB of LDS go to the scalar-input weights
    static constexpr int BA = RES_LSTM_BA, BV = RES_LSTM_BV;     // rest in LDS (32 KiB da tile + <= 128 KiB)
};
template <> struct res_cfg<MVAE_GRU> {    // 96 fragments per wave
    static constexpr int FA = RES_GRU_FA, FV = RES_GRU_FV;
    static constexpr int FV_SCALAR = RES_GRU_FV;
    static constexpr int BA = RES_GRU_BA, BV = RES_GRU_BV;
};

template <int CELL, int XMODE, int SAVE>
int launch_fwd_res(const mvae_rnn_fwd_args& a, hipStream_t s) {
    typedef res_cfg<CELL> C;
    constexpr int FVx = XMODE == MVAE_X_SCALAR ? C::FV_SCALAR : C::FV;
    constexpr int G = mvae_gates(CELL), NL = G * RNT * RS - C::FA - FVx;
    const size_t lds = (size_t)(2 + (CELL == MVAE_GRU ? 1 : 0)) * 16 * RH * sizeof(bf16_t) +
                       (size_t)4 * NL * 64 * sizeof(frag) + (XMODE == MVAE_X_SCALAR ? (size_t)2 * G * RH * sizeof(float) : 0);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_fwd_res_k<CELL, XMODE, SAVE, C::FA, FVx>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((rnn_fwd_res_k<CELL, XMODE, SAVE, C::FA, FVx>), dim3(a.B / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
// LSTM with dense / indexed / constant inputs and seq_layout TILE16P: the slot-interleaved kernel
template <int XMODE, int SAVE>
int launch_lstm_il(const mvae_rnn_fwd_args& a, hipStream_t s) {
    typedef res_cfg<MVAE_LSTM> C;
    constexpr int NL = 4 * RNT * RS - 4 - C::IA - C::IV;
    const size_t lds = (size_t)2 * 16 * RH * sizeof(bf16_t) + (size_t)4 * NL * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_il_k<XMODE, SAVE, C::IA, C::IV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((lstm_fwd_il_k<XMODE, SAVE, C::IA, C::IV>), dim3(a.B / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

// GRU with dense / indexed / constant inputs and seq_layout TILE16P: the slot-interleaved kernel
template <int XMODE, int SAVE>
int launch_gru_il(const mvae_rnn_fwd_args& a, hipStream_t s) {
    const size_t lds = (size_t)3 * 16 * RH * sizeof(bf16_t) + (size_t)4 * RNT * RS * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_il_k<XMODE, SAVE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((gru_fwd_il_k<XMODE, SAVE>), dim3(a.B / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <int CELL, int XMODE>
int fwd_res_save(const mvae_rnn_fwd_args& a, hipStream_t s) {
    // lookup-table layout: the slot-interleaved LSTM kernel gathers tile pairs (MVAE_TABLE_PAIRED), everything else row-major
    if (XMODE == MVAE_X_INDEX &&
        a.table_layout != (((CELL == MVAE_LSTM || CELL == MVAE_GRU) && a.seq_layout == MVAE_TILE16P) ? MVAE_TABLE_PAIRED : MVAE_TABLE_ROWMAJOR))
        return MVAE_E_ARG;
    if (a.acts) {
        if (!a.hs || (CELL == MVAE_LSTM && !a.cs)) return MVAE_E_UNSUPPORTED;     // partial saves: generic kernel
        if constexpr (CELL == MVAE_LSTM && XMODE != MVAE_X_SCALAR)
            if (a.seq_layout == MVAE_TILE16P) return launch_lstm_il<XMODE, SAVE_ALL>(a, s);
        if constexpr (CELL == MVAE_GRU && XMODE != MVAE_X_SCALAR)
            if (a.seq_layout == MVAE_TILE16P) return launch_gru_il<XMODE, SAVE_ALL>(a, s);
        if (a.seq_layout == MVAE_TILE16P) return MVAE_E_UNSUPPORTED;
        return launch_fwd_res<CELL, XMODE, SAVE_ALL>(a, s);
    }
    if (a.cs) return MVAE_E_UNSUPPORTED;
    if constexpr (CELL == MVAE_LSTM && XMODE != MVAE_X_SCALAR)
        if (a.seq_layout == MVAE_TILE16P) return a.hs ? launch_lstm_il<XMODE, SAVE_HS>(a, s) : launch_lstm_il<XMODE, SAVE_NONE>(a, s);
    if constexpr (CELL == MVAE_GRU && XMODE != MVAE_X_SCALAR)
        if (a.seq_layout == MVAE_TILE16P) return a.hs ? launch_gru_il<XMODE, SAVE_HS>(a, s) : launch_gru_il<XMODE, SAVE_NONE>(a, s);
    if (a.seq_layout == MVAE_TILE16P) return MVAE_E_UNSUPPORTED;
    return a.hs ? launch_fwd_res<CELL, XMODE, SAVE_HS>(a, s) : launch_fwd_res<CELL, XMODE, SAVE_NONE>(a, s);
}
template <int CELL>
int fwd_res_xmode(const mvae_rnn_fwd_args& a, hipStream_t s) {
    switch (a.xmode) {
        case MVAE_X_DENSE: return a.xp ? fwd_res_save<CELL, MVAE_X_DENSE>(a, s) : MVAE_E_ARG;
        case MVAE_X_INDEX: return (a.idx && a.table) ? fwd_res_save<CELL, MVAE_X_INDEX>(a, s) : MVAE_E_ARG;
        case MVAE_X_SCALAR: return (a.xs && a.w_row && a.bias) ? fwd_res_save<CELL, MVAE_X_SCALAR>(a, s) : MVAE_E_ARG;
        case MVAE_X_CONST: return a.xp0 ? fwd_res_save<CELL, MVAE_X_CONST>(a, s) : MVAE_E_ARG;
    }
    return MVAE_E_ARG;
}

template <bool HAS_EXT>
int launch_lstm_bwd_il(const mvae_rnn_bwd_args& a, hipStream_t s) {
    typedef res_cfg<MVAE_LSTM> C;
    constexpr int NL = RNT * (4 * RH / 32) - 4 - C::JA - C::JV;
    const size_t lds = (size_t)16 * 4 * RH * sizeof(bf16_t) + (size_t)4 * NL * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_il_k<HAS_EXT, C::JA, C::JV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((lstm_bwd_il_k<HAS_EXT, C::JA, C::JV>), dim3(a.B / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <bool HAS_EXT>
int launch_gru_bwd_il(const mvae_rnn_bwd_args& a, hipStream_t s) {
    const size_t lds = (size_t)16 * 3 * RH * sizeof(bf16_t) + (size_t)16 * RH * sizeof(bf16_t) +
                       (size_t)4 * RNT * (RH / 32) * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_il_k<HAS_EXT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((gru_bwd_il_k<HAS_EXT>), dim3(a.B / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <int CELL, bool HAS_EXT>
int launch_bwd_res(const mvae_rnn_bwd_args& a, hipStream_t s) {
    typedef res_cfg<CELL> C;
    if constexpr (CELL == MVAE_LSTM)
        if (a.seq_layout == MVAE_TILE16P) return launch_lstm_bwd_il<HAS_EXT>(a, s);
    if constexpr (CELL == MVAE_GRU)
        if (a.seq_layout == MVAE_TILE16P) return launch_gru_bwd_il<HAS_EXT>(a, s);
    if (a.seq_layout == MVAE_TILE16P) return MVAE_E_UNSUPPORTED;
    constexpr int G = mvae_gates(CELL), NL = RNT * (G * RH / 32) - C::BA - C::BV;
    const size_t lds = (size_t)16 * G * RH * sizeof(bf16_t) + (size_t)4 * NL * 64 * sizeof(frag) +
                       (CELL == MVAE_GRU ? (size_t)16 * RH * sizeof(bf16_t) : 0);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_bwd_res_k<CELL, HAS_EXT, C::BA, C::BV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((rnn_bwd_res_k<CELL, HAS_EXT, C::BA, C::BV>), dim3(a.B / 16), dim3(256), lds, s, a);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}


// ===========================================================================================================
// PHASE launches (round 3): every recurrence of one phase of the step - the notes stack's layers AND the velocity / instrument /
// held-notes branches beside it - as ONE launch on ONE queue.  Each recurrence used to be a launch on a queue of its own, forked
// from and joined into the critical queue by events: a cross-queue dependency costs the command processors 40-120 us to resolve
// when it resolves late (kernel timeline, profiles/r03_b_timeline_lstm_step.txt: ~0.5 ms of a 7.7 ms step between the phases),
// a same-queue boundary ~2-8 us.  Workgroups [base[i], base[i+1]) run problem i with the SAME code as the single launches (the
// kernel bodies above; problems of one launch share the cell type and the save mode and may differ in input mode and length).
// Problems are listed producers first - workgroups are dispatched in index order - and the hand-over between the layers of a
// stack (device-side counters) is what it was.  An XPAND problem is the x*W + b expansion of a 1-feature roll (mvae_outer_bias_
// tile16) as a chunk-publishing producer inside the launch: the recurrence that consumes it follows it chunk by chunk instead
// of waiting for a 0.12 ms launch in front of it.
// ===========================================================================================================
#include "rnn_multi.h"
template <int CELL, int SAVE>
__global__ __launch_bounds__(256, 1) void rnn_fwd_multi_k(const rnn_fwd_multi m) {
    const int bid = (int)blockIdx.x, nx = m.nx;
    int i = 0;
    while (i + 1 < nx + m.n && bid >= m.base[i + 1]) ++i;
    const unsigned bx = (unsigned)(bid - m.base[i]);
    if (i < nx) {
        xpand_body<4>(m.xp[i], (int)bx, m.base[i + 1] - m.base[i]);
        return;
    }
    // (a COPY: the fields then live in scalar registers for the whole launch, as a single launch's kernel arguments do - read
    //  through the reference they are re-loaded from the argument segment inside the step loop, each load behind an lgkmcnt wait)
    const mvae_rnn_fwd_args a = m.p[i - nx];
    const int xm = __builtin_amdgcn_readfirstlane(a.xmode);
    if constexpr (CELL == MVAE_LSTM) {
        typedef res_cfg<MVAE_LSTM> C;
        if (xm == MVAE_X_DENSE) lstm_fwd_il_body<MVAE_X_DENSE, SAVE, C::IA, C::IV>(a, bx);
        else if (xm == MVAE_X_INDEX) lstm_fwd_il_body<MVAE_X_INDEX, SAVE, C::IA, C::IV>(a, bx);
        else lstm_fwd_il_body<MVAE_X_CONST, SAVE, C::IA, C::IV>(a, bx);
    } else {
        if (xm == MVAE_X_DENSE) gru_fwd_il_body<MVAE_X_DENSE, SAVE>(a, bx);
        else if (xm == MVAE_X_INDEX) gru_fwd_il_body<MVAE_X_INDEX, SAVE>(a, bx);
        else gru_fwd_il_body<MVAE_X_CONST, SAVE>(a, bx);
    }
}
template <int CELL>
__global__ __launch_bounds__(256, 1) void rnn_bwd_multi_k(const rnn_bwd_multi m) {
    const int bid = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < m.n && bid >= m.base[i + 1]) ++i;
    const unsigned bx = (unsigned)(bid - m.base[i]);
    const mvae_rnn_bwd_args a = m.p[i];        // (a copy: see rnn_fwd_multi_k)
    const bool ext = __builtin_amdgcn_readfirstlane(a.dhs_ext != nullptr);
    if constexpr (CELL == MVAE_LSTM) {
        typedef res_cfg<MVAE_LSTM> C;
        if (ext) lstm_bwd_il_body<true, C::JA, C::JV>(a, bx);
        else lstm_bwd_il_body<false, C::JA, C::JV>(a, bx);
    } else {
        if (ext) gru_bwd_il_body<true>(a, bx);
        else gru_bwd_il_body<false>(a, bx);
    }
}
template <int CELL, int SAVE>
int launch_fwd_multi(const rnn_fwd_multi& m, int total, hipStream_t s) {
    typedef res_cfg<MVAE_LSTM> C;
    constexpr int NL = 4 * RNT * RS - 4 - C::IA - C::IV;
    const size_t lds = CELL == MVAE_LSTM ? (size_t)2 * 16 * RH * sizeof(bf16_t) + (size_t)4 * NL * 64 * sizeof(frag)
                                         : (size_t)3 * 16 * RH * sizeof(bf16_t) + (size_t)4 * RNT * RS * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_fwd_multi_k<CELL, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((rnn_fwd_multi_k<CELL, SAVE>), dim3(total), dim3(256), lds, s, m);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}
template <int CELL>
int launch_bwd_multi(const rnn_bwd_multi& m, int total, hipStream_t s) {
    typedef res_cfg<MVAE_LSTM> C;
    constexpr int NL = RNT * (4 * RH / 32) - 4 - C::JA - C::JV;
    const size_t lds = CELL == MVAE_LSTM ? (size_t)16 * 4 * RH * sizeof(bf16_t) + (size_t)4 * NL * 64 * sizeof(frag)
                                         : (size_t)16 * 3 * RH * sizeof(bf16_t) + (size_t)16 * RH * sizeof(bf16_t) +
                                               (size_t)4 * RNT * (RH / 32) * 64 * sizeof(frag);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_bwd_multi_k<CELL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MVAE_E_LAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((rnn_bwd_multi_k<CELL>), dim3(total), dim3(256), lds, s, m);
    MVAE_CHECK_LAUNCH();
    return MVAE_OK;
}

}  // namespace

// Entry points used by rnn.hip's dispatch.  Return MVAE_E_UNSUPPORTED when the shape is not this file's.
int mvae_rnn_fwd_w8(const mvae_rnn_fwd_args& a, hipStream_t s);      // rnn_w8.hip: two waves per SIMD (seq_layout MVAE_TILE16Q)
int mvae_rnn_fwd_resident(const mvae_rnn_fwd_args& a, hipStream_t s) {
    if (a.seq_layout == MVAE_TILE16Q) return mvae_rnn_fwd_w8(a, s);
    if (a.H != RH || a.dtype != MVAE_BF16 || (a.B % 16) != 0 || (a.seq_layout != MVAE_TILE16 && a.seq_layout != MVAE_TILE16P))
        return MVAE_E_UNSUPPORTED;
    if (a.cell == MVAE_LSTM) return fwd_res_xmode<MVAE_LSTM>(a, s);
    if (a.cell == MVAE_GRU) return fwd_res_xmode<MVAE_GRU>(a, s);
    return MVAE_E_UNSUPPORTED;
}
int mvae_rnn_bwd_w8(const mvae_rnn_bwd_args& a, hipStream_t s);
// MVAE_LSTM_BWD_W8=1 (default OFF; read at every call): the LSTM BPTT of seq_layout TILE16P on the two-waves-per-SIMD kernel of
// rnn_w8.hip (same data, same results).  It LOSES - 3.7-3.9 us per time step against 2.5-2.7 - because a quarter of U^T has to be
// streamed from L2 and gfx950 retires vector memory instructions in issue order: every streamed fragment waits behind the HBM-latency
// requests of the next step's saved activations (profiles/r06_g_lstm_bptt_w8.txt).  Kept as the measured experiment.
bool mvae_lstm_bwd_w8_enabled() {
    const char* e = getenv("MVAE_LSTM_BWD_W8");
    return e && atoi(e) != 0;
}
int mvae_rnn_bwd_resident(const mvae_rnn_bwd_args& a, hipStream_t s) {
    if (a.seq_layout == MVAE_TILE16Q) return mvae_rnn_bwd_w8(a, s);
    if (a.cell == MVAE_LSTM && a.seq_layout == MVAE_TILE16P && a.H == RH && a.dtype == MVAE_BF16 && (a.B % 16) == 0 &&
        mvae_lstm_bwd_w8_enabled())
        return mvae_rnn_bwd_w8(a, s);
    if (a.H != RH || a.dtype != MVAE_BF16 || (a.B % 16) != 0 || (a.seq_layout != MVAE_TILE16 && a.seq_layout != MVAE_TILE16P))
        return MVAE_E_UNSUPPORTED;
    if (a.cell == MVAE_LSTM) {
        if (!a.cs) return MVAE_E_ARG;
        return a.dhs_ext ? launch_bwd_res<MVAE_LSTM, true>(a, s) : launch_bwd_res<MVAE_LSTM, false>(a, s);
    }
    if (a.cell == MVAE_GRU) return a.dhs_ext ? launch_bwd_res<MVAE_GRU, true>(a, s) : launch_bwd_res<MVAE_GRU, false>(a, s);
    return MVAE_E_UNSUPPORTED;
}

// (see rnn_fwd_multi_k)  MVAE_E_UNSUPPORTED: a problem is not one the slot-interleaved kernels take - launch them one by one
int mvae_rnn_fwd_multi_w8(const mvae_rnn_fwd_args* problems, int32_t n, const mvae_xpand_args* xpand, int32_t n_xpand, void* stream);
int mvae_rnn_bwd_multi_w8(const mvae_rnn_bwd_args* problems, int32_t n, void* stream);
extern "C" int mvae_rnn_fwd_multi(const mvae_rnn_fwd_args* problems, int32_t n, const mvae_xpand_args* xpand, int32_t n_xpand,
                                  void* stream) {
    if (!problems || n <= 0 || n > RM_MAX || n_xpand < 0 || n_xpand > RM_XP_MAX || (n_xpand && !xpand)) return MVAE_E_ARG;
    if (problems[0].seq_layout == MVAE_TILE16Q) return mvae_rnn_fwd_multi_w8(problems, n, xpand, n_xpand, stream);
    rnn_fwd_multi m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    m.nx = n_xpand;
    int total = 0;
    for (int i = 0; i < n_xpand; ++i) {
        const mvae_xpand_args& x = xpand[i];
        if (!((x.xs && x.w && x.bias) || (x.idx && x.table)) || !x.out || !x.chunk_done || x.out_kind != MVAE_BF16 || x.R <= 0 || (x.N != 4 * RH && x.N != 3 * RH) ||
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
        if (a.H != RH || a.dtype != MVAE_BF16 || a.seq_layout != MVAE_TILE16P || (a.cell != MVAE_LSTM && a.cell != MVAE_GRU) ||
            a.cell != f.cell || a.xmode == MVAE_X_SCALAR || !a.hs || (a.acts != nullptr) != (f.acts != nullptr) ||
            (a.acts && a.cell == MVAE_LSTM && !a.cs) || (!a.acts && a.cs))
            return MVAE_E_UNSUPPORTED;
        if ((a.xmode == MVAE_X_DENSE && !a.xp) || (a.xmode == MVAE_X_INDEX && !(a.idx && a.table)) || (a.xmode == MVAE_X_CONST && !a.xp0))
            return MVAE_E_ARG;
        if (a.xmode == MVAE_X_INDEX && a.table_layout != MVAE_TABLE_PAIRED) return MVAE_E_ARG;       // (the slot-interleaved kernels gather tile pairs)
        m.p[i] = a;
        m.base[n_xpand + i] = total;
        total += a.B / 16;
    }
    m.base[n_xpand + n] = total;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (f.cell == MVAE_LSTM) return f.acts ? launch_fwd_multi<MVAE_LSTM, SAVE_ALL>(m, total, s) : launch_fwd_multi<MVAE_LSTM, SAVE_HS>(m, total, s);
    return f.acts ? launch_fwd_multi<MVAE_GRU, SAVE_ALL>(m, total, s) : launch_fwd_multi<MVAE_GRU, SAVE_HS>(m, total, s);
}
extern "C" int mvae_rnn_bwd_multi(const mvae_rnn_bwd_args* problems, int32_t n, void* stream) {
    if (!problems || n <= 0 || n > RM_MAX) return MVAE_E_ARG;
    if (problems[0].seq_layout == MVAE_TILE16Q) return mvae_rnn_bwd_multi_w8(problems, n, stream);
    rnn_bwd_multi m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    int total = 0;
    const mvae_rnn_bwd_args& f = problems[0];
    for (int i = 0; i < n; ++i) {
        const mvae_rnn_bwd_args& a = problems[i];
        if (!a.ut_pack || !a.hs || !a.acts || !a.da || a.T <= 0 || a.B <= 0 || (a.B % 16) || a.chunk_steps < 0 ||
            ((a.wait_ready || a.signal_done) && a.chunk_steps == 0) || (a.wait_ready && !a.dhs_ext))
            return MVAE_E_ARG;
        if (a.H != RH || a.dtype != MVAE_BF16 || a.seq_layout != MVAE_TILE16P || (a.cell != MVAE_LSTM && a.cell != MVAE_GRU) ||
            a.cell != f.cell)
            return MVAE_E_UNSUPPORTED;
        if (a.cell == MVAE_LSTM && !a.cs) return MVAE_E_ARG;
        m.p[i] = a;
        m.base[i] = total;
        total += a.B / 16;
    }
    m.base[n] = total;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return f.cell == MVAE_LSTM ? launch_bwd_multi<MVAE_LSTM>(m, total, s) : launch_bwd_multi<MVAE_GRU>(m, total, s);
}
