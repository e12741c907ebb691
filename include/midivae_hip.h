/*
 * midivae_hip.h  --  C ABI of libmidivae_hip.so (gfx950 / MI355X).
 *
 * The reference (brunnergino/MIDI-VAE) has no FFI: its hot path is reached through four Keras Model
 * methods called from Python --
 *     autoencoder.fit       reference vae_training.py:804-809
 *     autoencoder.evaluate  reference vae_training.py:300
 *     encoder.predict       reference vae_training.py:289,795 ; vae_evaluation.py:2180-2181
 *     decoder.predict       reference vae_evaluation.py:2482
 * and everything below those calls is Keras/recurrentshop graph code (reference vae_definition.py:15-761).
 * This header is the operator set those four calls decompose into on the device.  Each entry names the
 * reference construct it replaces.  The Python host (midi-vae_amd/engine.py) binds it with ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing here allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, in order;
 *   - return value: 0 = enqueued, negative = rejected (MVAE_E_*), nothing enqueued; never throws;
 *   - sequences are TIME-MAJOR: element (t, b, c) of a (T, B, C) array lives at (t*B + b)*C + c;
 *   - `dtype` selects the MFMA operand type AND the storage type of sequence activations:
 *         MVAE_F32  : f32 operands (v_mfma_f32_16x16x4_f32), f32 storage  -- parity mode
 *         MVAE_BF16 : bf16 operands (v_mfma_f32_16x16x32_bf16), f32 accumulate, bf16 storage
 *     recurrent state, gate arithmetic, losses, gradients of parameters and the optimizer are always f32;
 *   - gate blocks along a 'G*H' axis: [z|r|h] GRU, [i|f|g|o] LSTM, [h] SimpleRNN.
 */
#ifndef MIDIVAE_HIP_H
#define MIDIVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVAE_ABI_VERSION 9

enum { MVAE_OK = 0, MVAE_E_ARG = -1, MVAE_E_UNSUPPORTED = -2, MVAE_E_LAUNCH = -3,
       MVAE_E_FORMAT = -4 /* host packers: a row of the caller's array is not one-hot */ };
enum { MVAE_GRU = 0, MVAE_LSTM = 1, MVAE_RNN = 2 };
enum { MVAE_F32 = 0, MVAE_BF16 = 1 };
/* where x_t W + b comes from in a recurrent layer */
enum {
    MVAE_X_DENSE = 0,  /* xp (T,B,G*H) precomputed by mvae_gemm (layers fed by another layer)              */
    MVAE_X_INDEX = 1,  /* one-hot rows: xp = table[idx[t,b]], table (K,G*H) dtype = W + b (notes, instruments) */
    MVAE_X_SCALAR = 2, /* 1-wide input: xp = xs[t,b]*w + bias                              (velocity roll)  */
    MVAE_X_CONST = 3   /* the same row every step: xp = xp0[b]                  (decoder cells, Appendix A.6) */
};

/* Layout of (rows, cols) sequence arrays exchanged with the recurrent kernels.
 *   MVAE_ROWMAJOR : element (m, n) at m*cols + n.
 *   MVAE_TILE16   : rows and cols in tiles of 16 (both must be multiples of 16); tile (m/16, n/16) is 256
 *                   contiguous elements holding the MFMA C-fragment image of the 16x16 block:
 *                       offset = ((m/16 * cols/16 + n/16) * 64 + ((n%16)/4)*16 + m%16) * 4 + n%4
 *                   i.e. lane (q = (n%16)/4, r = m%16) owns 4 consecutive n.  One wave reads / writes a whole tile as
 *                   512 contiguous bytes (bf16) instead of 16 segments at a power-of-two row stride.
 *   MVAE_TILE16P  : TILE16 with the tiles (m/16, 2j) and (m/16, 2j+1) interleaved per lane (cols % 32 == 0):
 *                       offset = ((m/16 * cols/32 + n/32) * 64 + ((n%16)/4)*16 + m%16) * 8 + ((n%32)/16)*4 + n%4
 *                   one lane's 8 values of a tile pair are 16 contiguous bytes: one memory instruction instead of two
 *                   (a VMEM instruction costs the CU's address unit the same 16 cycles whatever its width).
 *                   As a seq_layout it means: xp and dhs_ext TILE16, the saved activations (acts, cs) TILE16P - the
 *                   layout the slot-interleaved LSTM / GRU kernels use; forward and backward of a layer must agree.
 *   MVAE_TILE16Q  : as TILE16P with the tiles j and j + 8 of every block of 256 columns interleaved (cols % 256 == 0):
 *                       offset = ((m/16 * cols/32 + n/256 * 8 + (n/16)%8) * 64 + ((n%16)/4)*16 + m%16) * 8 + ((n/128)%2)*4 + n%4
 *                   As a seq_layout (round 6) it selects the TWO-WAVES-PER-SIMD kernels (GRU, H = 256, bf16; rnn_w8.hip): a
 *                   workgroup is 8 waves, wave w owns the unit tiles w and 8 + w of every gate - its pair; xp and dhs_ext
 *                   are TILE16, a one-hot layer's table MVAE_TABLE_PAIRED8, and a chunk of a time-pipelined stack is
 *                   published by 8 waves per workgroup (mvae_rnn_producer_waves). */
enum { MVAE_ROWMAJOR = 0, MVAE_TILE16 = 1, MVAE_TILE16P = 2, MVAE_TILE16Q = 3 };
/* lookup tables of one-hot input layers (mvae_rnn_fwd_args.table_layout).  PAIRED: tiles (2j, 2j+1) of a lane in 16 contiguous bytes;
 * PAIRED8: tiles (j, j+8) of every block of 256 columns (column 128 h + 16 j + 4 q + e of the block sits at 32 j + 8 q + 4 h + e) */
enum { MVAE_TABLE_ROWMAJOR = 0, MVAE_TABLE_PAIRED = 1, MVAE_TABLE_PAIRED8 = 2 };
/* waves per workgroup that publish a chunk of a time-pipelined stack (signal_done += 1 each): 8 for MVAE_TILE16Q, else 4 */
int mvae_rnn_producer_waves(int32_t seq_layout);

int mvae_abi_version(void);
/* human-readable build string (arch, compile date) */
const char* mvae_build_info(void);

/* ---------------------------------------------------------------------------------------------------------
 * Recurrent layer, forward.  Replaces keras.layers.{GRU,LSTM,SimpleRNN} (encoder, reference
 * vae_definition.py:448-480) and recurrentshop {GRU,LSTM,SimpleRNN}Cell stepped by RecurrentModel (decoder,
 * :533-546,584-594,622-632).  One workgroup owns 16 batch rows for all T steps; no inter-workgroup traffic.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t cell, dtype, xmode;
    int32_t T, B, H;
    const void* u_pack;    /* recurrent kernel packed by mvae_pack_recurrent (direction 0)                    */
    const void* xp;        /* DENSE: (T,B,G*H) dtype                                                          */
    const uint8_t* idx;    /* INDEX: (T,B)                                                                    */
    const void* table;     /* INDEX: (K,G*H) dtype (mvae_make_table)                                          */
    const float* xs;       /* SCALAR: (T,B)                                                                   */
    const float* w_row;    /* SCALAR: (G*H)                                                                   */
    const float* bias;     /* SCALAR: (G*H)                                                                   */
    const void* xp0;       /* CONST: (B,G*H) dtype                                                            */
    const float* h0;       /* (B,H) or NULL = zeros                                                           */
    const float* c0;       /* LSTM: (B,H) or NULL = zeros                                                     */
    void* hs;              /* (T+1,B,H) dtype or NULL; slot 0 receives h0, slot t+1 receives h_t              */
    void* cs;              /* LSTM: (T+1,B,H) dtype or NULL                                                   */
    void* acts;            /* (T,B,G*H) dtype post-activation gates, or NULL (inference)                      */
    float* h_last;         /* (B,H) or NULL                                                                   */
    float* c_last;         /* LSTM: final cell state (B,H) f32 or NULL (row stride h_last_ld): lets a sequence be
                              run as consecutive launches (time chunks) without rounding the carried state     */
    int32_t h0_ld;         /* row stride of h0 / c0 in floats (0 = H): states may be column blocks of a wider buffer */
    int32_t h_last_ld;     /* row stride of h_last (0 = H)                                                     */
    /* ---- time-pipelined stacks (slot-interleaved LSTM kernels only; all NULL / 0 otherwise) ----------------------
     * A stack of layers runs as ONE launch per layer, plus ONE persistent mvae_gemm launch per layer interface (its
     * chunk_* fields): layer l publishes chunk k (chunk_steps time steps) of hs in a counter, the GEMM waits for it,
     * projects the chunk and publishes xp, layer l+1 waits for that.  All device-side (system-scope loads / atomics,
     * ~microseconds per hand-over; stream-level wait/write values cost 50-100 us each).  Counters are plain 32-bit
     * words in device memory, zeroed by the caller before the launches.  Every producer stores the data it hands over
     * WRITE-THROUGH (sc1) and publishes behind a drained vmcnt - no L2 write-back: a buffer_wbl2 per hand-over writes back
     * the whole XCD's L2 and stalls the recurrent workgroups that share it (7.5 % of a training step). */
    int32_t chunk_steps;         /* time steps per pipeline chunk; chunk k = steps [k*chunk_steps, (k+1)*chunk_steps)    */
    const uint32_t* wait_ready;  /* [chunks] chunk k of xp may be read once wait_ready[k] >= wait_value (kernel polls)    */
    uint32_t wait_value;         /* 0 = 1                                                                                 */
    uint32_t* signal_done;       /* [chunks] += 1 per WAVE (4 * B/16 of them) once chunk k of hs is complete and visible  */
    uint32_t* status;            /* [1] set non-zero if a wait timed out (~2-4 s): results are invalid                    */
    int32_t seq_layout;    /* layout of xp, acts and cs (hs is always row-major): MVAE_ROWMAJOR, MVAE_TILE16 or
                              MVAE_TILE16P.  The tiled layouts need B % 16 == 0 and select the resident-weights kernels
                              (H=256, bf16): TILE16 the phased ones, TILE16P the slot-interleaved ones (LSTM, GRU; not
                              for MVAE_X_SCALAR inputs)                                                            */
    int32_t table_layout;  /* MVAE_X_INDEX: MVAE_TABLE_ROWMAJOR (0), or MVAE_TABLE_PAIRED (1: MVAE_PREP_MAKE_TABLE with c = 1) -
                              what the slot-interleaved LSTM and GRU kernels (MVAE_TILE16P) REQUIRE: one lane's values of two
                              neighbouring unit tiles are 16 contiguous bytes, 8 gathers per row and step instead of 16
                              (a memory instruction costs the CU's address unit the same whatever its width); every other
                              kernel takes the row-major table                                                          */
} mvae_rnn_fwd_args;
int mvae_rnn_fwd(const mvae_rnn_fwd_args* a, void* stream);

/* Recurrent layer, backward through time (the autodiff of the above that Keras/TF derives).
 * Produces d(xp) for every step; parameter gradients follow from it with mvae_gemm / mvae_colsum. */
typedef struct {
    int32_t cell, dtype;
    int32_t T, B, H;
    const void* ut_pack;   /* recurrent kernel packed by mvae_pack_recurrent (direction 1)                    */
    const void* hs;        /* (T+1,B,H) dtype, from forward                                                   */
    const void* cs;        /* LSTM: (T+1,B,H) dtype                                                           */
    const void* acts;      /* (T,B,G*H) dtype                                                                 */
    const void* dhs_ext;   /* (T,B,H) dtype gradient arriving at h_t from the layer above, or NULL            */
    const float* dh_last;  /* (B,H) gradient arriving at the final state, or NULL                             */
    const float* dc_last;  /* LSTM: gradient arriving at the final CELL state (time-chunked BPTT), or NULL     */
    void* da;              /* (T,B,G*H) dtype: gradient w.r.t. xp                                             */
    void* rh;              /* GRU: (T,B,H) dtype r_t*h_{t-1} (left operand of the candidate-kernel gradient)  */
    float* dh0;            /* (B,H) or NULL                                                                   */
    float* dc0;            /* LSTM: (B,H) or NULL                                                             */
    int32_t dh_last_ld;    /* row stride of dh_last (0 = H)                                                   */
    int32_t dh0_ld;        /* row stride of dh0 / dc0 (0 = H)                                                 */
    /* time-pipelined stacks, as in mvae_rnn_fwd_args: wait_ready gates dhs_ext (chunk k = steps [k*cs, (k+1)*cs), consumed
     * from the last chunk to the first), signal_done publishes da */
    int32_t chunk_steps;
    const uint32_t* wait_ready;
    uint32_t wait_value;
    uint32_t* signal_done;
    uint32_t* status;
    int32_t seq_layout;    /* layout of acts, cs and dhs_ext (hs, da, rh are always row-major)                */
} mvae_rnn_bwd_args;
int mvae_rnn_bwd(const mvae_rnn_bwd_args* a, void* stream);

/* Pack a recurrent kernel U (H, G*H) f32 row-major into MFMA A-fragment order.
 * direction 0: forward  (rows of A = gate columns, contraction over h)      -> G*H*H elements
 * direction 1: backward (rows of A = hidden units, contraction over gate columns) -> G*H*H elements */
int mvae_pack_recurrent(const float* U, void* out, int32_t cell, int32_t H, int32_t dtype, int32_t direction,
                        void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GEMM  C = alpha * opA(A) * opB(B) [+ bias] [-> tanh]   (Dense layers :484,487,506-507,563-567; input
 * projections of stacked layers; all parameter gradients).  Row-major with leading dimensions.
 *   a_kind: MVAE_F32 / MVAE_BF16, or MVAE_A_ONEHOT: A is never stored - opA(A)[m,k] = (idx[k] == m)
 *           (requires trans_a = 1), used for the gradient of an X_INDEX table.
 *   c_kind: MVAE_F32 / MVAE_BF16.   accumulate: 0 store, 1 atomic add into f32 C (split-K allowed).
 * --------------------------------------------------------------------------------------------------------- */
enum { MVAE_A_ONEHOT = 2 };
enum { MVAE_ACT_NONE = 0, MVAE_ACT_TANH = 1 };
typedef struct {
    int32_t M, N, K;
    int32_t trans_a, trans_b;     /* 0: stored (M,K)/(K,N); 1: stored (K,M)/(N,K)                             */
    int32_t a_kind, b_kind, c_kind;
    int32_t lda, ldb, ldc;
    int32_t accumulate, act, split_k;
    float alpha;
    const void* A;
    const void* B;
    void* C;
    const float* bias;            /* (N) or NULL                                                              */
    int32_t c_layout;             /* MVAE_ROWMAJOR (ldc applies) or MVAE_TILE16 (store only, M%16==0, N%16==0)  */
    int32_t max_blocks;           /* 0 = one workgroup per output tile; >0 = at most this many workgroups, each looping
                                     over tiles: keeps a throughput GEMM from occupying every CU while latency-critical
                                     recurrent launches (which need whole idle CUs) run beside it                 */
    int32_t sys_release;          /* != 0: every workgroup ends with a system-scope release (L2 write-back), so that a
                                     kernel ALREADY RUNNING on another XCD sees C after a stream-ordered flag write  */
    /* persistent chunked mode (fast bf16 path only, split_k <= 1, max_blocks > 0 = the persistent grid): the M rows are
     * processed in chunks of chunk_rows (multiple of 128); chunk c starts once chunk_wait[c] >= chunk_wait_value and is
     * published by chunk_done[c] += 1 per wave (4 * max_blocks in total).  chunk_reverse: last chunk first. */
    int32_t chunk_rows, chunk_reverse;
    const uint32_t* chunk_wait;
    uint32_t chunk_wait_value;
    uint32_t* chunk_done;
    uint32_t* chunk_status;       /* [1] set non-zero if a wait timed out                                          */
    float* colsum_b;              /* (N) f32 or NULL: += column sums of B over K (the bias gradient beside a weight-gradient
                                     GEMM C = A^T B, whose B tiles pass through the CU anyway: no second pass over B).
                                     Fast bf16 path with trans_a = 1, trans_b = 0 and accumulate only (else MVAE_E_UNSUPPORTED) */
    /* K-streaming (fast bf16 path, trans_a = 1, accumulate = 1, row-major C): a weight-gradient GEMM C += A^T B that FOLLOWS the
     * running BPTT kernel producing B (= da).  The K rows come in chunks of k_chunk_rows; chunk c may be read once
     * k_wait[c] >= k_wait_value (the counters a pipelined layer publishes, mvae_rnn_bwd_args.signal_done).  split_k = P
     * partitions: workgroup (tile, p) takes rows [c*k_chunk_rows + p*k_chunk_rows/P, ... + k_chunk_rows/P) of every chunk,
     * keeps its 128 x 128 accumulator in registers across ALL chunks and adds it to C once at the end - one workgroup per
     * (tile, partition), all resident for the whole launch (max_blocks must be 0).  k_reverse: last chunk first (BPTT order).
     * What is left when the producer ends is one chunk's share, not the whole GEMM.  chunk_status reports a timed-out wait. */
    const uint32_t* k_wait;
    uint32_t k_wait_value;
    int32_t k_chunk_rows, k_reverse;
} mvae_gemm_args;
int mvae_gemm(const mvae_gemm_args* a, void* stream);
/* n <= 8 K-streaming problems (k_wait set, trans_a = 1, trans_b = 0, accumulate, no bias) behind the kernels of ONE pipelined stack
 * as ONE launch on ONE queue (each would otherwise hold a queue of its own for the whole BPTT); at most 256 workgroups in all
 * (MVAE_E_ARG beyond) - they all wait, resident, for their producers.  Replaces the reference's implicit "gradients of every
 * weight of the encoder stack" inside K.gradients (vae_definition.py:1016-1045 via Keras' train_function). */
int mvae_gemm_kstream_multi(const mvae_gemm_args* problems, int32_t n, void* stream);
/* n <= 16 ORDINARY weight-gradient GEMMs as one launch (round 5): every problem C (M,N) f32 row-major += A^T B with trans_a = 1,
 * trans_b = 0, accumulate = 1, bf16 or MVAE_A_ONEHOT A, bf16 B, optional colsum_b and split_k - what mvae_gemm would run on its
 * fast kernel, with none of the chunk_* / k_wait fields (those producers must be DONE: order the launch behind them).  For short
 * sequences (reference settings.py:108-109: T = 64), where a dozen such launches of 20-120 us each were most of what follows the
 * last recurrence.  MVAE_E_UNSUPPORTED: a problem is not of this form (run it through mvae_gemm). */
int mvae_gemm_multi(const mvae_gemm_args* problems, int32_t n, void* stream);

/* Stream-ordered synchronisation with RUNNING kernels (hipStreamWaitValue32 / hipStreamWriteValue32 on plain device
 * memory): `stream` proceeds once *addr >= value / writes value to *addr after everything enqueued before it on `stream`. */
int mvae_stream_wait_value32(void* stream, const uint32_t* addr, uint32_t value);
int mvae_stream_write_value32(void* stream, uint32_t* addr, uint32_t value);
/* Do streams a and b share a hardware queue (1) or not (0; < 0: error)?  Synchronises both, then runs a bounded waiter (~2 ms) on
 * a and, enqueued after it, a setter on b: on one queue the setter cannot start before the waiter gives up.  `scratch`: 2 device
 * words, `tag` != 0 a value not used before on them.  The engine calls it when it creates its streams - kernels of one stack that
 * wait for each other across two streams need two queues (the reference has no counterpart: Keras runs one op at a time). */
int mvae_streams_alias(void* stream_a, void* stream_b, uint32_t* scratch, uint32_t tag);

/* Workgroups per CU of the library's RESIDENT kernels as the loaded code object and the device report it
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor with the launch's dynamic LDS size): which = 0 the persistent dX GEMM between two
 * pipelined layers (mvae_gemm chunk_*, NT), 1 the weights-stationary forward projection, 2 a K-streaming workgroup
 * (mvae_gemm_kstream_multi); the recurrent kernels are one workgroup per CU by construction (__launch_bounds__(256, 1)).
 * The host decides from these whether the kernels of a time-pipelined stack can all be resident at once (engine.py
 * _resident_cus) - the reference has no counterpart: Keras runs one op at a time.  < 0: error code. */
int mvae_occupancy(int32_t which);

/* out[n] (+)= sum_r X[r, n]  for X (R, N) of `kind`; ldx elements between rows; atomic f32 accumulate */
int mvae_colsum(const void* X, int32_t kind, int32_t R, int32_t N, int32_t ldx, float* out, void* stream);
/* out[n] += sum_r wgt[r] * X[r, n]: the kernel gradient of a 1-feature (scalar) input layer, dW = xs^T da */
int mvae_colsum_weighted(const void* X, int32_t kind, const float* wgt, int32_t R, int32_t N, int32_t ldx, float* out,
                         void* stream);
/* out[b, n] (+)= sum_t X[t, b, n]   (gradient of an X_CONST row; accumulate != 0 adds to out, e.g. per time chunk) */
int mvae_sum_over_time(const void* X, int32_t kind, int32_t T, int32_t BN, float* out, int32_t accumulate, void* stream);

/* PHASE launches: every recurrence of one phase of the step (reference vae_definition.py:443-480 - the encoder's notes stack and
 * its instrument / velocity / held-notes rolls - or :519-726 - the decoder's cell stacks - forward or backward) as ONE launch:
 * n <= 8 problems of the slot-interleaved kernels (H = 256, bf16, seq_layout MVAE_TILE16P, LSTM or GRU, dense / indexed / constant
 * inputs; all of one cell type, all saving activations or none), workgroups in problem order - list producers first.  The time-
 * pipelined hand-over fields of each problem work as in the single launches.  ``xpand``: up to 2 expansions of a 1-feature roll
 * (mvae_outer_bias_tile16) as chunk-publishing producers inside the forward launch: out (R, N) bf16 MVAE_TILE16 = xs[r] w[n] +
 * bias[n], chunk_done[c] += 4 * blocks when rows [c * chunk_rows, (c+1) * chunk_rows) are written (write-through); the problem that
 * reads `out` as its xp names chunk_done as wait_ready with wait_value = previous total + 4 * blocks.  The same producer can write
 * out the table rows of a one-hot input layer (idx / table): the indexed-input kernels gather 2 KB rows of the table every step
 * (2.67 us per time step alone against 2.22 for a dense input), and the bottom layer of the encoder stack sets the pace of its phase.
 * MVAE_E_UNSUPPORTED: some problem is not one of these kernels' (launch them one by one with mvae_rnn_fwd / mvae_rnn_bwd). */
typedef struct {
    const float* xs;           /* (R) f32 */
    const float* w;            /* (N) f32, 16-byte aligned */
    const float* bias;         /* (N) f32, 16-byte aligned */
    void* out;                 /* (R, N) out_kind (MVAE_BF16), MVAE_TILE16 */
    int32_t out_kind, R, N;    /* N = the consumer's G*H: 1024 (LSTM) or 768 (GRU), H = 256 */
    int32_t chunk_rows;        /* rows per published chunk (% 16 == 0, divides R): chunk_steps * B of the consumer */
    uint32_t* chunk_done;      /* [R / chunk_rows] counters */
    int32_t blocks, reserved;  /* workgroups of this producer (<= 256) */
    const uint8_t* idx;        /* or NULL (4-byte aligned).  With idx (R) and table (K, N) of out_kind (mvae_make_table): out[r] = table[idx[r]] - the input */
    const void* table;         /* projection of a ONE-HOT layer written out, so that the layer reads it like a dense one (xs / w / bias unused) */
} mvae_xpand_args;
int mvae_rnn_fwd_multi(const mvae_rnn_fwd_args* problems, int32_t n, const mvae_xpand_args* xpand, int32_t n_xpand, void* stream);
int mvae_rnn_bwd_multi(const mvae_rnn_bwd_args* problems, int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Output heads: Dense(H -> N) + activation + Keras weighted loss + metric + d(logits), fused, over all
 * (t, b) rows at once.  Replaces Dense(softmax)+categorical_crossentropy (:542,:593 with :338,:367),
 * Dense(sigmoid)+mse (:631,:374) and the per-output 'accuracy' metric (:339).
 *   kind 0: softmax + categorical cross-entropy (targets as class index per row, or -1 = all-zero target)
 *   kind 1: sigmoid + squared error (targets f32 per row), N must be 1
 * Row weights: rw[row] multiplies the row's score (already divided by the Keras normalisers on the host).
 * scalars[0] += sum rw*score, scalars[1] += number of rows whose argmax (or rounding) matches the target.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t kind, dtype;
    int32_t R, H, N;              /* R = T*B rows                                                            */
    int32_t want_grad;
    const void* hs;               /* (R,H) dtype                                                             */
    const void* wt;               /* (NP,H) dtype: W^T zero-padded to NP rows (mvae_transpose_convert)       */
    const float* bias;            /* (N)                                                                     */
    const uint8_t* target_idx;    /* kind 0: (R) class index, 255 = all-zero target row                      */
    const float* target_val;      /* kind 1: (R)                                                             */
    const float* row_weight;      /* (R) or NULL = 1                                                         */
    float grad_scale;             /* multiplies d(logits) (loss weight of the head)                          */
    float* probs;                 /* (R,N) f32 or NULL                                                       */
    uint8_t* argmax;              /* (R) or NULL: first-max index (kind 0), round(p) (kind 1)                */
    void* dlogits;                /* (R,NP) dtype, NP = N rounded up to 16; required if want_grad            */
    float* scalars;               /* (2) accumulated atomically                                              */
    int32_t b_stride, b_valid;    /* rows are (t, b) with b = row % b_stride; rows with b >= b_valid are padding and
                                     are excluded from the metric count (0,0 = every row counts)             */
    const void* wc;               /* (H,NP) dtype: W zero-padded to NP columns (MVAE_PREP_CONVERT_PAD), or NULL */
    void* dhs;                    /* (R,H) dtype in MVAE_TILE16 or NULL: d(logits) W^T, the gradient w.r.t. the top cell's
                                     h sequence, from the same launch (needs wc, want_grad, R % 16 == 0, H <= 256):
                                     no GEMM launch between the head and the decoder's BPTT                   */
    const uint8_t* target_idx2;   /* kind 0, or NULL: (R) the SECOND hot column of a two-hot target row (attach_instruments,
                                     reference import_midi.py:288-292: pitch one-hot | instrument one-hot), 255 = none:
                                     loss -log p[t1] - log p[t2], accuracy against the first hot column             */
} mvae_head_args;
int mvae_head(const mvae_head_args* a, void* stream);
/* padded column count NP used for `wt` rows and `dlogits` columns of an N-wide head (16/32/64/128; <0 = too wide) */
int mvae_head_np(int32_t N);

/* Latent block: KL term + reparameterisation + style classifier on z[:, :C], forward and backward
 * (reference vae_definition.py:29-37, 498-502, 514-515, 730-734 and their gradients). */
typedef struct {
    int32_t B, Z, C;
    float beta, prior_mean, prior_std, inv_batch;
    const float* mu;              /* (B,Z)                                                                   */
    const float* logvar;          /* (B,Z)                                                                   */
    const float* eps;             /* (B,Z) already scaled by epsilon_std                                     */
    const uint8_t* style_target;  /* (B) class index or NULL (no style head)                                 */
    const float* style_row_weight;/* (B) Keras sample weight / normalisers, or NULL = inv_batch              */
    float* z;                     /* (B,Z) out                                                               */
    float* style_probs;           /* (B,C) out or NULL                                                       */
    float* scalars;               /* (3): += inv_batch*sum_b kl_b, += sum_b rw*style CE, += style argmax hits */
    int32_t ldz;                  /* row stride of z (0 = Z): z may be the left block of [z | history]       */
} mvae_latent_fwd_args;
int mvae_latent_fwd(const mvae_latent_fwd_args* a, void* stream);

typedef struct {
    int32_t B, Z, C;
    float beta, prior_mean, prior_std, style_weight, inv_batch;
    const float* mu;
    const float* logvar;
    const float* eps;
    const float* dz;              /* (B,Z) gradient arriving from the decoder's initial-state Denses          */
    const float* style_probs;     /* (B,C) or NULL                                                           */
    const uint8_t* style_target;
    const float* style_row_weight;
    float* dmu;                   /* (B,Z)                                                                   */
    float* dlogvar;               /* (B,Z)                                                                   */
    int32_t lddz;                 /* row stride of dz (0 = Z)                                                */
} mvae_latent_bwd_args;
int mvae_latent_bwd(const mvae_latent_bwd_args* a, void* stream);

/* The Dense chain around the latent as ONE launch each way (reference vae_definition.py:483-516 encoder tail and
 * latent, :519-530 decoder initial states): cat = [h_notes | h_instr | h_vel] -> pack Dense (tanh, when w_pack) -> extra
 * Dense (tanh, when w_extra) -> z_mean / z_log_var (on the two halves of the vector when split) -> mvae_latent_fwd ->
 * S = tanh([z | history] w_init + b_init).  Same results as the separate mvae_gemm / mvae_latent_* calls, in f32 FMAs.
 * All matrices f32 row-major: w_pack (ncat*H, H), w_extra (H, H), w_mu / w_lv (H/2 or H, Z), w_init (zin, n_init).
 * B % 4 == 0 (rows >= B_valid are padding: computed, excluded from scalars and gradients); H % 8, Z % 4, zin % 4 == 0. */
typedef struct {
    int32_t B, B_valid, H, Z, C, ncat, zin, n_init /* columns of S */, split;
    float beta, prior_mean, prior_std, inv_batch;
    const float* cat;             /* (B, ncat*H)                                                             */
    const float *w_pack, *b_pack, *w_extra, *b_extra, *w_mu, *b_mu, *w_lv, *b_lv, *w_init, *b_init;
    const float* eps;             /* (B,Z)                                                                   */
    const uint8_t* style_target;  /* as mvae_latent_fwd_args                                                 */
    const float* style_row_weight;
    float *pack, *extra;          /* (B,H) out (kept for backward)                                           */
    float *mu, *logvar;           /* (B,Z) out                                                               */
    float* zh;                    /* (B,zin): columns [0,Z) out, [Z,zin) in (history)                        */
    float* style_probs;           /* (B,C) out or NULL                                                       */
    float* scalars;               /* (3) as mvae_latent_fwd_args                                             */
    float* S;                     /* (B,n_init) out                                                          */
} mvae_latent_chain_fwd_args;
int mvae_latent_chain_fwd(const mvae_latent_chain_fwd_args* a, void* stream);

typedef struct {
    int32_t B, B_valid, H, Z, C, ncat, zin, n_init, split;
    float beta, prior_mean, prior_std, style_weight, inv_batch;
    const float *wt_pack, *wt_extra, *wt_mu, *wt_lv, *wt_init;   /* the TRANSPOSED matrices, f32 row-major: (H, ncat*H), (H,H),
                                                                    (Z, H/2 or H) x2, (n_init, zin) - e.g. from
                                                                    mvae_transpose_convert / mvae_prepare_batch       */
    const float *S, *pack, *extra, *mu, *logvar, *eps, *style_probs;
    const uint8_t* style_target;
    const float* style_row_weight;
    float* dS;                    /* (B,n_init) in: gradient w.r.t. S; out: w.r.t. its pre-activation         */
    float* dzh;                   /* (B,zin) out                                                             */
    float *dmu, *dlogvar;         /* (B,Z) out                                                               */
    float *d_extra, *d_pack;      /* (B,H) out: gradients w.r.t. the pre-activations of the two tanh Denses   */
    float* dcat;                  /* (B, ncat*H) out                                                         */
} mvae_latent_chain_bwd_args;
int mvae_latent_chain_bwd(const mvae_latent_chain_bwd_args* a, void* stream);

/* Every derived copy of the parameters a step needs, in ONE launch (25 tiny dependent kernels cost 0.3 - 0.8 ms of queue
 * latency per training step otherwise).  Each job is one of the single calls above:
 *   MVAE_PREP_PACK_RECURRENT   src = U (a=H, b=G*H) f32, c = direction        -> dst as mvae_pack_recurrent(kind)
 *   MVAE_PREP_MAKE_TABLE       src = W (a=K, b=N), src2 = bias (N), c = layout -> dst (K, N) kind    (mvae_make_table; c = 1:
 *                              MVAE_TABLE_PAIRED, c = 2: MVAE_TABLE_PAIRED8, see mvae_rnn_fwd_args.table_layout)
 *   MVAE_PREP_TRANSPOSE_CONVERT src = W (a=K, b=N), c = N_pad                 -> dst (N_pad, K) kind (mvae_transpose_convert)
 *   MVAE_PREP_CONVERT          src (a*b) f32                                  -> dst (a*b) kind      (mvae_convert)
 *   MVAE_PREP_ZERO             (no src)                                       -> dst (a*b) kind, a*b even for bf16: zeros
 *                              (the step's loss / metric accumulators: one fill launch less)
 *   MVAE_PREP_CONVERT_PAD      src (a, b) f32, c = padded row length >= b     -> dst (a, c) kind, columns b..c-1 zero
 *   MVAE_PREP_ADD_I32          src = NULL or a guard word (uint32)            -> *(int32_t*)dst += a unless *src != 0 (the
 *                              optimizer's step count after mvae_adam_step_dev(MVAE_ADAM_KEEP_COUNT): no launch of its own);
 *                              src2 = NULL or a latch word (uint32): a non-zero *src is then moved there (max) and CLEARED -
 *                              the status word of the time-pipelined stacks lives for one step, the latch until the host reads it
 *   MVAE_PREP_BROADCAST_ROWS   src (b) f32                                    -> dst (a, b) kind: every row = src.  start*W + b of a
 *                              decoder cell whose constant input is all zeros - what the reference's packers always pass
 *                              (vae_definition.py:820,916): the bias row, no GEMM */
enum { MVAE_PREP_PACK_RECURRENT = 0, MVAE_PREP_MAKE_TABLE = 1, MVAE_PREP_TRANSPOSE_CONVERT = 2, MVAE_PREP_CONVERT = 3,
       MVAE_PREP_ZERO = 4, MVAE_PREP_CONVERT_PAD = 5, MVAE_PREP_ADD_I32 = 6, MVAE_PREP_BROADCAST_ROWS = 7 };
typedef struct {
    int32_t op, kind;          /* MVAE_PREP_*, element kind of dst (MVAE_F32 / MVAE_BF16) */
    int32_t a, b, c, reserved;
    const void* src;
    const void* src2;
    void* dst;
} mvae_prep_job;
int mvae_prepare_batch(const mvae_prep_job* jobs /* host array */, int32_t n_jobs, void* stream);

/* out (R, N) TILE16 of out_kind = xs[r] * w[n] + bias[n]: the input projection x*W + b of a 1-feature input (velocity
 * roll, reference vae_definition.py:456-470) written out, so that the layer can run on the MVAE_X_DENSE kernels
 * (R % 16 == 0, N % 16 == 0, w and bias 16-byte aligned) */
int mvae_outer_bias_tile16(const float* xs, const float* w, const float* bias, void* out, int32_t out_kind, int32_t R, int32_t N,
                           void* stream);
/* out (R, N) of `kind` in `layout` (MVAE_TILE16 / MVAE_ROWMAJOR) = table[idx[r]] + table2[idx2[r]] (tables (K, N) / (K2, N) of `kind`, mvae_make_table): x*W + b of
 * TWO-hot input rows - attach_instruments appends the voice's instrument one-hot to every pitch row (reference import_midi.py:
 * 288-292, settings.py:186-187,207-208) - written out so that the layer runs on the MVAE_X_DENSE kernels (R, N % 16 == 0) */
int mvae_gather2_tile16(const uint8_t* idx, const uint8_t* idx2, const void* table, const void* table2, void* out, int32_t kind,
                        int32_t R, int32_t N, int32_t layout, void* stream);
/* (rows, cols) row-major <-> tiled, same element kind on both sides.
 * to_tile16: 0 TILE16 -> row-major, 1 row-major -> TILE16, 2 TILE16P -> row-major, 3 row-major -> TILE16P,
 *            4 TILE16Q -> row-major, 5 row-major -> TILE16Q */
int mvae_relayout(const void* src, void* dst, int32_t kind, int32_t rows, int32_t cols, int32_t to_tile16, void* stream);

/* elementwise helpers */
int mvae_tanh_bwd(const float* y, const float* dy, float* dx, size_t n, void* stream);      /* dx = dy*(1-y^2) */
int mvae_convert(const void* src, int32_t src_kind, void* dst, int32_t dst_kind, size_t n, void* stream);
/* table (K,N) dst_kind = W (K,N) + bias (N): the X_INDEX lookup table of a one-hot input layer */
int mvae_make_table(const float* W, const float* bias, void* table, int32_t K, int32_t N, int32_t dst_kind,
                    void* stream);
/* out (N_pad,K) dst_kind = transpose of W (K,N) f32, rows N..N_pad-1 zero */
int mvae_transpose_convert(const float* W, void* out, int32_t K, int32_t N, int32_t N_pad, int32_t dst_kind,
                           void* stream);

/* Keras-2.0.8 optimizers on a flat f32 parameter buffer (reference vae_definition.py:174-175).
 * Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)   (epsilon outside the correction) */
int mvae_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                   float eps, int32_t t, float grad_scale, void* stream);
/* same, with the count of COMPLETED steps in device memory (incremented after the update): replayable in a hipGraph.
 * zero_grad = flags: MVAE_ADAM_ZERO_GRAD: g is zeroed as it is consumed (the next step accumulates into it: no separate
 * fill launch); MVAE_ADAM_KEEP_COUNT: *t_done is left alone - the caller adds 1 before the next step, e.g. as a
 * MVAE_PREP_ADD_I32 job of the weight-preparation launch that follows anyway (one dependent launch less per step).
 * guard (NULL = none): a device word, e.g. the `status` word of the time-pipelined stacks - while it is non-zero the update
 * (and the step count) is SKIPPED on the device, so a step whose gradients are invalid never reaches the parameters or the
 * moments; the gradients are still zeroed. */
enum { MVAE_ADAM_ZERO_GRAD = 1, MVAE_ADAM_KEEP_COUNT = 2 };
int mvae_adam_step_dev(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                       float eps, int32_t* t_done, float grad_scale, int32_t zero_grad, const uint32_t* guard, void* stream);
int mvae_rmsprop_step(float* p, float* g, float* v, size_t n, float lr, float rho, float eps,
                      float grad_scale, int32_t zero_grad, const uint32_t* guard, void* stream);

/* Epoch accumulators of the per-step loss / metric scalars - what Keras' BaseLogger keeps on the host for the `history` of
 * autoencoder.fit (reference vae_training.py:817-864 reads it per song): acc[i] += (bit i of plain_mask ? 1 : alpha) * x[i],
 * n <= 32.  One tiny launch behind the step instead of a device->host read (a full synchronisation) per minibatch. */
int mvae_scalars_accumulate(float* acc, const float* x, int32_t n, float alpha, uint32_t plain_mask, void* stream);
/* dst (rows, cols; row stride ldd) = src rows src_row0 .. (row stride lds); the first zero_rows rows of dst are ZERO instead
 * (their source row index may be negative).  Places the history latent - the previous window's z, zeros for the first
 * window (reference vae_training.py:795-798: H[1:] = z[:-1]) - or a caller's z into the [z | history] decoder input. */
int mvae_copy2d_f32(float* dst, int32_t ldd, const float* src, int32_t lds, int32_t rows, int32_t cols, int32_t src_row0,
                    int32_t zero_rows, void* stream);

/* The history pre-pass FUSED into the first train step of a song (reference vae_training.py:788-798: z' = encoder.predict(song) with
 * a fresh draw, H[1:] = z'[:-1], H[0] = 0; then fit on the same weights, :804-809): from the mu / logvar (B, Z) the step's own
 * encoder forward produced and the pre-pass's draw eps2 (B, Z; already scaled by epsilon_std):
 *   z_out[b] = mu[b] + exp(logvar[b] / 2) * eps2[b]          (rows of stride ldo; may be NULL)
 *   hist[b]  = b == 0 ? (prev ? prev : 0) : z'[b-1]          (rows of stride ldh: the history columns of [z | history]), b < B
 *   hist[b]  = 0 for B <= b < B_pad */
int mvae_history_from_latent(const float* mu, const float* logvar, const float* eps2, int32_t B, int32_t B_pad, int32_t Z,
                             float* hist, int32_t ldh, const float* prev, float* z_out, int32_t ldo, void* stream);

/* Signature head (reference vae_definition.py:737-745, loss :409-416; off by default): out (B,SD) = tanh(zh[:, off:off+SD]);
 * with a target (B,SD): scalars[0] += sum_b row_weight[b] * mean_j (out - target)^2, scalars[1] += rows whose argmax matches the
 * target's (Keras 'accuracy' on this output).  bwd: dz[b, off+j] += weight * row_weight[b] * 2 (out - target) / SD * (1 - out^2). */
int mvae_signature_head_fwd(const float* zh, int32_t ldz, int32_t off, int32_t SD, int32_t B, const float* target,
                            const float* row_weight, float* out, float* scalars, void* stream);
int mvae_signature_head_bwd(float* dz, int32_t lddz, int32_t off, int32_t SD, int32_t B, const float* out, const float* target,
                            const float* row_weight, float weight, void* stream);
/* dlogits (R,NP) kind += probs * (dprobs - rowsum(probs * dprobs)): a gradient that arrives at a softmax output - the classifiers
 * the reference can hang on the decoder's notes / instrument probabilities (vae_definition.py:747-761) - folded into d(logits) */
int mvae_softmax_bwd_add(const float* probs, const float* dprobs, void* dlogits, int32_t kind, int32_t R, int32_t N, int32_t NP,
                         void* stream);

/* Bidirectional encoder layers (keras.layers.Bidirectional(..., merge_mode='concat'), reference vae_definition.py:445-453): the
 * backward layer runs on the time-reversed sequence.  f, r: the (T, B, H) output sequences of the forward layer and of the
 * backward layer (in ITS time order), row-major, of `kind`; cat (T, B, 2H) = [f[t] | r[T-1-t]] - the time-aligned concatenation the
 * next layer reads - and cat_rev (optional) the same reversed in time (what the next BACKWARD layer reads).  H*elemsize % 16 == 0. */
int mvae_bi_concat(const void* f, const void* r, void* cat, void* cat_rev, int32_t kind, int32_t T, int32_t B, int32_t H, void* stream);
/* dst[t] = (a ? a[t] : 0) + b[T-1-t] for T contiguous slabs of `slab` elements (% 4 == 0) of `kind`: gradients that cross between
 * the two time directions.  A time step of a (T*B, H) sequence is one slab in row-major AND in MVAE_TILE16 layout (B % 16 == 0). */
int mvae_add_time_reversed(void* dst, const void* a, const void* b, int32_t kind, int32_t T, size_t slab, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * STEP PLANS (csrc/plan.cpp): the enqueue of a whole step as ONE call.
 * The reference reaches the device through ONE Python call per minibatch - the Keras train_function behind
 * autoencoder.fit (reference vae_training.py:804-809; test_function :300, predict_function :289,795) - and Keras
 * replays a graph it compiled once.  Here a step is ~70 launches of the entry points above plus the stream / event
 * packets that order them across the engine's queues; enqueued one by one from the host language they cost 2.5-3.1 ms
 * per step in CPython (profiles/r03_d_host_vs_device.txt) - more than the device needs at the reference's shipped
 * configuration (T=64).  A plan is that list recorded once - entry point, argument values, a private copy of every
 * argument struct - and replayed by mvae_plan_run on the host thread that calls it: same entry points, same streams,
 * same order, nothing else (no graph capture: kernels that wait for each other across queues must be launched the way
 * they were).  Values that advance from step to step - the cumulative progress counters the
 * time-pipelined kernels wait for, the sequence numbers of value joins - are PATCHES: a 32-bit field of an argument
 * struct (or a scalar argument) := key_values[key] + offset at every run; everything else is constant.
 *
 * Events for cross-queue ordering inside a plan are plain entry points too (mvae_event_record / mvae_stream_wait_event).
 * --------------------------------------------------------------------------------------------------------- */
int mvae_event_create(void** event);                 /* hipEventCreateWithFlags(hipEventDisableTiming) */
int mvae_event_create_timed(void** event);           /* hipEventCreate: a pair of them brackets a launch INSIDE a plan (bench.py's roofline leg) */
int mvae_event_elapsed_ms(void* first, void* second, float* ms);     /* waits for ``second``, then the time between the two records */
int mvae_event_destroy(void* event);
int mvae_event_record(void* event, void* stream);
int mvae_event_synchronize(void* event);             /* ABI 8: the HOST waits for the event's latest record (Engine._pace: a paced host) */
int mvae_stream_wait_event(void* stream, void* event);

typedef struct mvae_plan mvae_plan;
int mvae_plan_create(mvae_plan** out);
int mvae_plan_destroy(mvae_plan* p);
/* Append a call of `entry_point` (the name of any stream-taking entry point of this header, e.g. "mvae_gemm").  `slots`: one 64-bit
 * value per argument, in order: pointers and integers as they are, a float as its 32 bits, an argument-struct pointer as 0 (its
 * contents follow with mvae_plan_set_blob).  Returns the index of the call (>= 0) or MVAE_E_ARG (unknown name, wrong count). */
int mvae_plan_add_call(mvae_plan* p, const char* entry_point, const uint64_t* slots, int32_t n_slots);
/* argument `slot` of call `call` points at a struct (or array of structs, or a host job array): `bytes` bytes are copied into the plan */
int mvae_plan_set_blob(mvae_plan* p, int32_t call, int32_t slot, const void* data, size_t bytes);
/* at every run: the 32-bit word at byte `offset` of the blob of (`call`, `slot`) - or, with offset < 0, the scalar argument itself -
 * := (uint32_t)(key_values[key] + add) */
int mvae_plan_add_patch(mvae_plan* p, int32_t call, int32_t slot, int64_t offset, int32_t key, int64_t add);
/* enqueue calls [first, last) in order (last < 0: to the end).  Returns 0, or the first non-zero return value of an entry point
 * (nothing after it is enqueued; mvae_plan_failed_call tells which). */
int mvae_plan_run(mvae_plan* p, int32_t first, int32_t last, const uint64_t* key_values, int32_t n_keys);
int mvae_plan_size(const mvae_plan* p);              /* number of calls */
int mvae_plan_failed_call(const mvae_plan* p);       /* index of the call whose return value the last failing run reported (-1: none) */

/* ---------------------------------------------------------------------------------------------------------
 * HOST-side packers (csrc/hostpack.cpp): every pointer below is a HOST pointer, nothing touches the device.
 * The reference hands the Keras Models float64 NumPy windows - one-hot rows (n, T, K) (reference import_midi.py:245-286,
 * lists built by vae_definition.py:880-1045) - the engine consumes ONE BYTE per row, time-major, padded to 16 windows.
 * One multi-threaded pass over the caller's array validates, converts, transposes and pads, straight into the pinned
 * staging block the engine uploads with a single asynchronous copy (64 MB of float64 -> 128 KB per 256 x 512 rows).
 * --------------------------------------------------------------------------------------------------------- */
enum { MVAE_HOST_F64 = 0, MVAE_HOST_F32 = 1, MVAE_HOST_U8 = 2 };
/* size of the worker pool: n > 0 sets it, 0 restores the default, < 0 only queries; returns the size.  Default: a quarter of the
 * hardware threads this process may run on (its affinity mask) divided by LOCAL_WORLD_SIZE (one process per GPU under data
 * parallelism), at least 4 and at most 64 - 64 alone on a 256-thread host, 8 per rank with 8 ranks. */
int mvae_host_threads(int32_t n);
/* windows [lo, hi) of x (n, T, K) one-hot rows of xkind -> out (T, Bp) uint8: out[t*Bp + (b-lo)] = position of the 1;
 * columns hi-lo .. Bp-1 = fill.  A row that is not exactly one 1 among zeros: MVAE_E_FORMAT, *bad_row = the LOWEST such flat row.
 * The host packers are serialised internally: concurrent callers (threads) are safe, they take turns. */
int mvae_host_onehot_to_index_tm(const void* x, int32_t xkind, int64_t n, int32_t T, int32_t K, int64_t lo, int64_t hi,
                                 uint8_t* out, int32_t Bp, uint8_t fill, int64_t* bad_row);
/* the same for rows that already are indices: idx (n, T) uint8 */
int mvae_host_index_to_tm(const uint8_t* idx, int64_t n, int32_t T, int64_t lo, int64_t hi, uint8_t* out, int32_t Bp,
                          uint8_t fill);
/* two-hot rows (attach_instruments): x (n, T, K), the first K1 columns one-hot, the other K - K1 columns one-hot -> out1 / out2
 * (T, Bp) uint8: the position of each 1 within its block.  MVAE_E_FORMAT / *bad_row as above for any other row. */
int mvae_host_twohot_to_index_tm(const void* x, int32_t xkind, int64_t n, int32_t T, int32_t K, int32_t K1, int64_t lo, int64_t hi,
                                 uint8_t* out1, uint8_t* out2, int32_t Bp, uint8_t fill, int64_t* bad_row);
/* windows [lo, hi) of v (n, T) of vkind -> out (T, Bp) f32 = scale * v, pad columns zero (velocity roll, sample weights) */
int mvae_host_rows_to_tm_f32(const void* v, int32_t vkind, int64_t n, int32_t T, int64_t lo, int64_t hi, float scale,
                             float* out, int32_t Bp);

#ifdef __cplusplus
}
#endif
#endif /* MIDIVAE_HIP_H */
