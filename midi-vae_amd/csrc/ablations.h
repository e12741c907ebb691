// Timing ablations of the resident recurrent kernels and the fast GEMM - DEVELOPMENT ONLY, results are wrong when any is set.
// The product build (csrc/Makefile) defines none of them: every switch below is 0 and the `if (ABL_...)` branches in the kernels
// are dead code.  A variant library for tools/rnn_microbench.py / tools/gemm_microbench.py (MVAE_LIB=...) is built by
// tools/build_variants.sh with -DMVAE_VARIANT_BUILD -DABL_...=1; without MVAE_VARIANT_BUILD a set switch is a compile error, so
// an ablation cannot reach libmidivae_hip.so by accident.
//   ABL_NOL     no LDS reads of LDS-resident weight fragments      ABL_NOTRG   no row-major write-back of h / da
//   ABL_NOSAVE  no saved-activation stores                         ABL_NOX     no input prefetch
//   ABL_NOMATH  no gate arithmetic                                 ABL_NOBAR   no barriers
//   ABL_NOB     B fragments: no LDS reads after the first two      ABL_NOTRANS exp / rcp replaced by multiplies
//   ABL_FILL=k  LSTM BPTT: k semantically empty VALU instructions per MFMA slot (how much filler room the M phase has)
//   GEMM_ABL_NOMFMA / NOLOAD / NOATOMIC                            the fast GEMM without its MFMAs / global loads / atomics
//   WS_ABL_NOSTORE / NOLOAD / NOMFMA                               proj_ws_k without its epilogue stores / A requests / MFMAs
#pragma once
#define MVAE_ABL_LIST(X) X(ABL_NOL) X(ABL_NOTRG) X(ABL_NOSAVE) X(ABL_NOX) X(ABL_NOMATH) X(ABL_NOBAR) X(ABL_NOB) X(ABL_NOTRANS) X(ABL_FILL) X(ABL_NOBAR1) X(ABL_NOBAR2) \
    X(GEMM_ABL_NOMFMA) X(GEMM_ABL_NOLOAD) X(GEMM_ABL_NOATOMIC) X(WS_ABL_NOSTORE) X(WS_ABL_NOLOAD) X(WS_ABL_NOMFMA)
#ifndef ABL_NOL
#define ABL_NOL 0
#endif
#ifndef ABL_NOTRG
#define ABL_NOTRG 0
#endif
#ifndef ABL_NOSAVE
#define ABL_NOSAVE 0
#endif
#ifndef ABL_NOX
#define ABL_NOX 0
#endif
#ifndef ABL_NOMATH
#define ABL_NOMATH 0
#endif
#ifndef ABL_NOBAR
#define ABL_NOBAR 0
#endif
#ifndef ABL_NOB
#define ABL_NOB 0
#endif
#ifndef ABL_NOTRANS
#define ABL_NOTRANS 0
#endif
#ifndef ABL_NOBAR1
#define ABL_NOBAR1 0      /* LSTM BPTT: without the barrier between the gate phase and the MFMA phase */
#endif
#ifndef ABL_NOBAR2
#define ABL_NOBAR2 0      /* ... without the barrier at the end of the step */
#endif
#ifndef ABL_FILL
#define ABL_FILL 0
#endif
#ifndef GEMM_ABL_NOMFMA
#define GEMM_ABL_NOMFMA 0
#endif
#ifndef GEMM_ABL_NOLOAD
#define GEMM_ABL_NOLOAD 0
#endif
#ifndef GEMM_ABL_NOATOMIC
#define GEMM_ABL_NOATOMIC 0
#endif
#ifndef WS_ABL_NOSTORE
#define WS_ABL_NOSTORE 0
#endif
#ifndef WS_ABL_NOLOAD
#define WS_ABL_NOLOAD 0
#endif
#ifndef WS_ABL_NOMFMA
#define WS_ABL_NOMFMA 0
#endif
#ifndef MVAE_VARIANT_BUILD
#define MVAE_ABL_CHECK(name) static_assert((name) == 0, #name " is a timing ablation: variant builds only (tools/build_variants.sh)");
MVAE_ABL_LIST(MVAE_ABL_CHECK)
#undef MVAE_ABL_CHECK
#endif
