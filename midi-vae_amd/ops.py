"""Tensor-level wrappers over the C ABI (hiplib): argument marshalling only, no arithmetic.

torch is used for what the task allows it for: device memory (tensors), streams.  Every function enqueues on
torch's CURRENT stream and returns immediately.  Shapes follow include/midivae_hip.h (time-major sequences).
"""
from __future__ import annotations

import ctypes as _C

import torch

from . import hiplib as hl

C_sizeof = _C.sizeof

_TORCH_DT = {hl.F32: torch.float32, hl.BF16: torch.bfloat16}


def torch_dtype(kind):
    return _TORCH_DT[kind]


def kind_of(t):
    if t.dtype == torch.float32:
        return hl.F32
    if t.dtype == torch.bfloat16:
        return hl.BF16
    raise TypeError("unsupported tensor dtype %s" % t.dtype)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class CounterValue(int):
    """a 32-bit wait / write value derived from one of the engine's monotonic counters (``key``): the fields it lands in are the
    patches of a step plan (plan.py) - field := counter before the step + offset.  Arithmetic on it gives a plain int."""
    def __new__(cls, value, key):
        self = int.__new__(cls, int(value))
        self.key = key
        return self


class ParamInt(int):
    """a 32-bit integer argument that is a PARAMETER of the call (``key`` = ("param", name)) - the real number of windows of a
    padded minibatch: every field it lands in is a patch of the step plan, field := the value the caller names at replay"""
    def __new__(cls, value, key):
        self = int.__new__(cls, int(value))
        self.key = key
        return self


class ParamFloat(float):
    """the same for a float argument (1 / global minibatch size); a plan carries it as its float32 bit pattern"""
    def __new__(cls, value, key):
        self = float.__new__(cls, float(value))
        self.key = key
        return self


PARAM_B, PARAM_INV_BATCH = ("param", "windows"), ("param", "inv_batch")


def f32_bits(v):
    import struct
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


def _tag(struct, field, value):
    """remember on an argument struct that ``field`` holds a counter value (consumed by _launch / the *_multi wrappers)"""
    key = getattr(value, "key", None)
    if key is not None:
        struct.__dict__.setdefault("_counter_fields", {})[field] = key


def _note_fields(arg_index, structs):
    """announce the counter fields of the struct argument(s) of the NEXT library call to the active plan recorder"""
    from . import plan
    rec = plan.active()
    if rec is None:
        return
    off = 0
    for st in structs:
        for field, key in getattr(st, "_counter_fields", {}).items():
            rec.note_field(arg_index, off + getattr(type(st), field).offset, key)
        off += C_sizeof(st)


def _note_scalar(arg_index, value):
    key = getattr(value, "key", None)
    if key is not None:
        from . import plan
        rec = plan.active()
        if rec is not None:
            rec.note_field(arg_index, -1, key)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
    return t.data_ptr()


def _pv(t):
    """pointer of a possibly strided VIEW (column block); the caller passes the row stride separately"""
    return None if t is None else t.data_ptr()


def pack_recurrent(U, cell, dtype, direction, out=None):
    """U (H, G*H) f32 -> fragment-ordered copy (flat, dtype)."""
    H = U.shape[0]
    if out is None:
        out = torch.empty(U.numel(), dtype=_TORCH_DT[dtype], device=U.device)
    hl.check(hl.load().mvae_pack_recurrent(_p(U), _p(out), cell, H, dtype, direction, _stream()), "mvae_pack_recurrent")
    return out


def rnn_fwd(cell, dtype, T, B, H, u_pack, *, xp=None, idx=None, table=None, xs=None, w_row=None, bias=None, xp0=None,
            h0=None, c0=None, hs=None, cs=None, acts=None, h_last=None, c_last=None, h0_ld=0, h_last_ld=0, seq_layout=0,
            chunk_steps=0, wait_ready=None, wait_value=0, signal_done=None, status=None, build_only=False, table_layout=0):
    if xp is not None:
        xmode = hl.X_DENSE
    elif idx is not None:
        xmode = hl.X_INDEX
    elif xs is not None:
        xmode = hl.X_SCALAR
    else:
        xmode = hl.X_CONST
    a = hl.RnnFwdArgs(cell, dtype, xmode, T, B, H, _p(u_pack), _pv(xp), _pv(idx), _p(table), _pv(xs), _p(w_row),
                      _p(bias), _p(xp0), _pv(h0), _pv(c0), _pv(hs), _pv(cs), _pv(acts), _pv(h_last), _pv(c_last), h0_ld, h_last_ld,
                      chunk_steps, _pv(wait_ready), int(wait_value), _pv(signal_done), _pv(status), seq_layout, int(table_layout))
    _tag(a, "wait_value", wait_value)
    if build_only:          # (a problem of rnn_fwd_multi; the tensors must stay alive until that launch)
        return a
    _note_fields(0, [a])
    hl.check(hl.load().mvae_rnn_fwd(a, _stream()), "mvae_rnn_fwd")


def rnn_bwd(cell, dtype, T, B, H, ut_pack, hs, cs, acts, da, *, dhs_ext=None, dh_last=None, dc_last=None, rh=None,
            dh0=None, dc0=None, dh_last_ld=0, dh0_ld=0, seq_layout=0, chunk_steps=0, wait_ready=None, wait_value=0,
            signal_done=None, status=None, build_only=False):
    a = hl.RnnBwdArgs(cell, dtype, T, B, H, _p(ut_pack), _pv(hs), _pv(cs), _pv(acts), _pv(dhs_ext), _pv(dh_last),
                      _pv(dc_last), _pv(da), _pv(rh), _pv(dh0), _pv(dc0), dh_last_ld, dh0_ld, chunk_steps, _pv(wait_ready),
                      int(wait_value), _pv(signal_done), _pv(status), seq_layout)
    _tag(a, "wait_value", wait_value)
    if build_only:
        return a
    _note_fields(0, [a])
    hl.check(hl.load().mvae_rnn_bwd(a, _stream()), "mvae_rnn_bwd")


def xpand(xs, w, bias, out, R, N, chunk_rows, chunk_done, blocks=64, idx=None, table=None):
    """an expansion of a 1-feature roll (outer_bias_tile16) - or, with idx / table, of a one-hot layer's table rows - as a
    chunk-publishing producer of rnn_fwd_multi"""
    return hl.XpandArgs(_p(xs), _p(w), _p(bias), _p(out), kind_of(out), int(R), int(N), int(chunk_rows), _pv(chunk_done), int(blocks), 0,
                        _p(idx), _p(table))


def rnn_fwd_multi(problems, xpands=()):
    """every recurrence of a phase as ONE launch on the current stream (``rnn_fwd(..., build_only=True)`` problems, producers
    first); False if the library does not take some problem (launch them singly)"""
    arr = (hl.RnnFwdArgs * len(problems))(*problems)
    xa = (hl.XpandArgs * len(xpands))(*xpands) if xpands else None
    _note_fields(0, problems)
    rc = hl.load().mvae_rnn_fwd_multi(arr, len(problems), xa, len(xpands), _stream())
    if rc == hl.E_UNSUPPORTED:
        return False
    hl.check(rc, "mvae_rnn_fwd_multi")
    return True


def rnn_bwd_multi(problems):
    arr = (hl.RnnBwdArgs * len(problems))(*problems)
    _note_fields(0, problems)
    rc = hl.load().mvae_rnn_bwd_multi(arr, len(problems), _stream())
    if rc == hl.E_UNSUPPORTED:
        return False
    hl.check(rc, "mvae_rnn_bwd_multi")
    return True


def gemm(A, B, C, M, N, K, *, trans_a=False, trans_b=False, lda=None, ldb=None, ldc=None, bias=None, act=hl.ACT_NONE,
         accumulate=False, split_k=1, alpha=1.0, a_kind=None, c_layout=0, max_blocks=0, sys_release=False, chunk_rows=0,
         chunk_reverse=False, chunk_wait=None, chunk_wait_value=0, chunk_done=None, chunk_status=None, colsum_b=None,
         k_wait=None, k_wait_value=0, k_chunk_rows=0, k_reverse=False, build_only=False):
    """C (M,N) = alpha * opA(A) opB(B) (+bias)(tanh).  Leading dimensions default to the packed row lengths.
    ``colsum_b`` (N,) f32 += column sums of B (weight-gradient GEMMs: the bias gradient from the same pass over B).
    ``k_wait`` ...: K-streaming behind a running producer of B (include/midivae_hip.h)."""
    a_kind = kind_of(A) if a_kind is None else a_kind
    if lda is None:
        lda = (M if trans_a else K)
    if ldb is None:
        ldb = (K if trans_b else N)
    if ldc is None:
        ldc = N
    g = hl.GemmArgs(M, N, K, int(trans_a), int(trans_b), a_kind, kind_of(B), kind_of(C), lda, ldb, ldc,
                    int(accumulate), act, split_k, float(alpha), A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(bias), c_layout, max_blocks,
                    int(sys_release), int(chunk_rows), int(chunk_reverse), _pv(chunk_wait), int(chunk_wait_value), _pv(chunk_done),
                    _pv(chunk_status), _pv(colsum_b), _pv(k_wait), int(k_wait_value), int(k_chunk_rows), int(k_reverse))
    _tag(g, "chunk_wait_value", chunk_wait_value)
    _tag(g, "k_wait_value", k_wait_value)
    if build_only:          # (for gemm_kstream_multi; the tensors must stay alive until that launch)
        return g
    _note_fields(0, [g])
    hl.check(hl.load().mvae_gemm(g, _stream()), "mvae_gemm")


def gemm_kstream_multi(problems):
    """several K-streaming GEMMs (``gemm(..., k_wait=..., build_only=True)``) as ONE launch on the current stream"""
    arr = (hl.GemmArgs * len(problems))(*problems)
    _note_fields(0, problems)
    hl.check(hl.load().mvae_gemm_kstream_multi(arr, len(problems), _stream()), "mvae_gemm_kstream_multi")


GEMM_MULTI_MAX = 16


def gemm_args(g):
    """launch a problem built with ``gemm(..., build_only=True)``"""
    _note_fields(0, [g])
    hl.check(hl.load().mvae_gemm(g, _stream()), "mvae_gemm")


def gemm_multi(problems, stream=None):
    """ordinary split-K weight-gradient GEMMs (``gemm(..., trans_a=True, accumulate=True, build_only=True)``) as ONE launch per
    GEMM_MULTI_MAX problems on the current (or given) stream.  Returns HOW MANY of the problems were launched: the library
    validates a part as a whole before it launches anything, so when it does not take a problem this way the parts in front of
    it have run (they accumulate: the caller must not launch them again) and ``problems[returned:]`` are the caller's."""
    st = _stream() if stream is None else stream.cuda_stream
    for i in range(0, len(problems), GEMM_MULTI_MAX):
        part = problems[i:i + GEMM_MULTI_MAX]
        arr = (hl.GemmArgs * len(part))(*part)
        rc = hl.load().mvae_gemm_multi(arr, len(part), st)
        if rc == hl.E_UNSUPPORTED:
            return i
        _note_fields(0, part)
        hl.check(rc, "mvae_gemm_multi")
    return len(problems)


def stream_wait_value32(word, value, stream=None):
    """the current (or given) stream proceeds once the 32-bit device word ``word`` (a 1-element view) is >= value"""
    _note_scalar(2, value)
    hl.check(hl.load().mvae_stream_wait_value32(_stream() if stream is None else stream.cuda_stream, _p(word), int(value)),
             "mvae_stream_wait_value32")


def stream_write_value32(word, value, stream=None):
    """writes ``value`` to the 32-bit device word after everything enqueued so far on the stream"""
    _note_scalar(2, value)
    hl.check(hl.load().mvae_stream_write_value32(_stream() if stream is None else stream.cuda_stream, _p(word), int(value)),
             "mvae_stream_write_value32")


def colsum(X, R, N, out, ldx=None):
    hl.check(hl.load().mvae_colsum(X.data_ptr(), kind_of(X), R, N, N if ldx is None else ldx, _p(out), _stream()),
             "mvae_colsum")


def outer_bias_tile16(xs, w, bias, out, R, N):
    """out (R, N) TILE16 = xs[r] * w[n] + bias[n]"""
    hl.check(hl.load().mvae_outer_bias_tile16(_p(xs), _p(w), _p(bias), _p(out), kind_of(out), R, N, _stream()),
             "mvae_outer_bias_tile16")


def gather2_tile16(idx, idx2, table, table2, out, R, N, layout=hl.TILE16):
    """out (R, N) in ``layout`` = table[idx[r]] + table2[idx2[r]]: x*W + b of two-hot input rows (attach_instruments)"""
    hl.check(hl.load().mvae_gather2_tile16(_p(idx), _p(idx2), _p(table), _p(table2), _p(out), kind_of(out), R, N, layout, _stream()),
             "mvae_gather2_tile16")


def colsum_weighted(X, wgt, R, N, out, ldx=None):
    """out[n] += sum_r wgt[r] * X[r, n]  (wgt f32)"""
    hl.check(hl.load().mvae_colsum_weighted(X.data_ptr(), kind_of(X), _p(wgt), R, N, N if ldx is None else ldx, _p(out),
                                            _stream()), "mvae_colsum_weighted")


def sum_over_time(X, T, BN, out, accumulate=False):
    hl.check(hl.load().mvae_sum_over_time(_p(X), kind_of(X), T, BN, _p(out), int(accumulate), _stream()),
             "mvae_sum_over_time")


def head_np(N):
    return hl.load().mvae_head_np(N)


def head(kind, dtype, R, H, N, hs, wt, bias, *, target_idx=None, target_val=None, row_weight=None, grad_scale=1.0,
         probs=None, argmax=None, dlogits=None, scalars=None, b_stride=0, b_valid=0, wc=None, dhs=None, target_idx2=None):
    a = hl.HeadArgs(kind, dtype, R, H, N, int(dlogits is not None), hs.data_ptr(), _p(wt), _p(bias), _p(target_idx),
                    _p(target_val), _p(row_weight), float(grad_scale), _p(probs), _p(argmax), _p(dlogits), _p(scalars),
                    b_stride, b_valid, _p(wc), _p(dhs), _p(target_idx2))
    _tag(a, "b_valid", b_valid)
    _note_fields(0, [a])
    hl.check(hl.load().mvae_head(a, _stream()), "mvae_head")


def latent_fwd(B, Z, C, beta, prior_mean, prior_std, inv_batch, mu, logvar, eps, z, scalars, *, style_target=None,
               style_row_weight=None, style_probs=None, ldz=0):
    a = hl.LatentFwdArgs(B, Z, C, beta, prior_mean, prior_std, inv_batch, _p(mu), _p(logvar), _p(eps),
                         _p(style_target), _p(style_row_weight), _pv(z), _p(style_probs), _p(scalars), ldz)
    hl.check(hl.load().mvae_latent_fwd(a, _stream()), "mvae_latent_fwd")


def latent_bwd(B, Z, C, beta, prior_mean, prior_std, style_weight, inv_batch, mu, logvar, eps, dz, dmu, dlogvar, *,
               style_probs=None, style_target=None, style_row_weight=None, lddz=0):
    a = hl.LatentBwdArgs(B, Z, C, beta, prior_mean, prior_std, style_weight, inv_batch, _p(mu), _p(logvar), _p(eps),
                         _pv(dz), _p(style_probs), _p(style_target), _p(style_row_weight), _p(dmu), _p(dlogvar), lddz)
    hl.check(hl.load().mvae_latent_bwd(a, _stream()), "mvae_latent_bwd")


def latent_chain_fwd(B, B_valid, H, Z, C, ncat, zin, n_init, split, beta, prior_mean, prior_std, inv_batch, **t):
    """Encoder tail Denses + latent block + decoder initial-state Denses in one launch (mvae_latent_chain_fwd); tensors by
    the field names of the C struct, absent / None = NULL.  Returns False if the library does not support the shape."""
    a = hl.LatentChainFwdArgs(B, B_valid, H, Z, C, ncat, zin, n_init, int(bool(split)), beta, prior_mean, prior_std, inv_batch)
    for name, _ in hl.LatentChainFwdArgs._fields_[13:]:
        setattr(a, name, _p(t.get(name)))
    _tag(a, "B_valid", B_valid)
    _tag(a, "inv_batch", inv_batch)
    _note_fields(0, [a])
    rc = hl.load().mvae_latent_chain_fwd(a, _stream())
    if rc == hl.E_UNSUPPORTED:
        return False
    hl.check(rc, "mvae_latent_chain_fwd")
    return True


def latent_chain_bwd(B, B_valid, H, Z, C, ncat, zin, n_init, split, beta, prior_mean, prior_std, style_weight, inv_batch, **t):
    a = hl.LatentChainBwdArgs(B, B_valid, H, Z, C, ncat, zin, n_init, int(bool(split)), beta, prior_mean, prior_std,
                              style_weight, inv_batch)
    for name, _ in hl.LatentChainBwdArgs._fields_[14:]:
        setattr(a, name, _p(t.get(name)))
    _tag(a, "B_valid", B_valid)
    _tag(a, "inv_batch", inv_batch)
    _note_fields(0, [a])
    rc = hl.load().mvae_latent_chain_bwd(a, _stream())
    if rc == hl.E_UNSUPPORTED:
        return False
    hl.check(rc, "mvae_latent_chain_bwd")
    return True


def relayout(src, dst, rows, cols, to_tile16, paired=False):
    """row-major <-> TILE16 (TILE16P with ``paired`` True, TILE16Q with ``paired`` == "q"); ``to_tile16`` True = row-major -> tiled"""
    hl.check(hl.load().mvae_relayout(_p(src), _p(dst), kind_of(src), rows, cols,
                                     int(bool(to_tile16)) + (4 if paired == "q" else 2 if paired else 0),
                                     _stream()), "mvae_relayout")


def tanh_bwd(y, dy, dx):
    hl.check(hl.load().mvae_tanh_bwd(_p(y), _p(dy), _p(dx), y.numel(), _stream()), "mvae_tanh_bwd")


def convert(src, dst):
    hl.check(hl.load().mvae_convert(_p(src), kind_of(src), _p(dst), kind_of(dst), src.numel(), _stream()),
             "mvae_convert")


def make_table_paired(W, bias, table):
    """lookup table in MVAE_TABLE_PAIRED column order (a one-job mvae_prepare_batch)"""
    pb = PrepBatch()
    pb.make_table(W, bias, table, paired=True)
    pb.run()


def make_table(W, bias, table):
    K, N = W.shape
    hl.check(hl.load().mvae_make_table(_p(W), _p(bias), _p(table), K, N, kind_of(table), _stream()), "mvae_make_table")


def transpose_convert(W, out, n_pad=None):
    K, N = W.shape
    hl.check(hl.load().mvae_transpose_convert(_p(W), _p(out), K, N, N if n_pad is None else n_pad, kind_of(out),
                                              _stream()), "mvae_transpose_convert")


class PrepBatch:
    """Job list for mvae_prepare_batch (every derived weight copy of a step in one launch).  The tensors are referenced by
    address: they must stay allocated and in place for as long as the batch is used."""

    def __init__(self):
        self.jobs = []
        self._keep = []
        self._arr = None

    def _add(self, op, dst, a, b, c, src, src2=None):
        self.jobs.append(hl.PrepJob(op, kind_of(dst), int(a), int(b), int(c), 0, _p(src), _p(src2), _p(dst)))
        self._keep += [src, src2, dst]
        self._arr = None

    def pack_recurrent(self, U, out, direction):
        H, GH = U.shape
        self._add(hl.PREP_PACK_RECURRENT, out, H, GH, direction, U)

    def make_table(self, W, bias, table, paired=False):
        """table (K, N) = W + bias; ``paired``: the column order the slot-interleaved LSTM / GRU kernels gather (hl.TABLE_PAIRED)"""
        self._add(hl.PREP_MAKE_TABLE, table, W.shape[0], W.shape[1], 2 if paired == 8 else int(bool(paired)), W, bias)

    def transpose_convert(self, W, out, n_pad=None):
        K, N = W.shape
        self._add(hl.PREP_TRANSPOSE_CONVERT, out, K, N, N if n_pad is None else n_pad, W)

    def convert(self, src, dst):
        self._add(hl.PREP_CONVERT, dst, src.numel(), 1, 0, src)

    def zero(self, dst):
        self._add(hl.PREP_ZERO, dst, dst.numel(), 1, 0, None)

    def add_i32(self, counter, value=1, guard=None, latch=None):
        """*counter (int32 device scalar) += value  (not while the device word ``guard`` is non-zero); with ``latch`` a non-zero
        guard word is then moved there (max) and cleared"""
        assert counter.dtype == torch.int32
        self.jobs.append(hl.PrepJob(hl.PREP_ADD_I32, hl.F32, int(value), 0, 0, 0, _p(guard), _p(latch), _p(counter)))
        self._keep += [counter, guard, latch]
        self._arr = None

    def broadcast_rows(self, row, out, rows):
        """out (rows, N) = ``row`` (N,) f32 repeated"""
        self._add(hl.PREP_BROADCAST_ROWS, out, rows, row.numel(), 0, row)

    def convert_pad(self, W, out, n_pad):
        K, N = W.shape
        self._add(hl.PREP_CONVERT_PAD, out, K, N, n_pad, W)

    def run(self):
        if self._arr is None:
            self._arr = (hl.PrepJob * len(self.jobs))(*self.jobs)
        hl.check(hl.load().mvae_prepare_batch(self._arr, len(self.jobs), _stream()), "mvae_prepare_batch")


def adam_step(p, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    hl.check(hl.load().mvae_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, t, grad_scale,
                                      _stream()), "mvae_adam_step")


def adam_step_dev(p, g, m, v, lr, t_done, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, zero_grad=False,
                  keep_count=False, guard=None):
    flags = (hl.ADAM_ZERO_GRAD if zero_grad else 0) | (hl.ADAM_KEEP_COUNT if keep_count else 0)
    hl.check(hl.load().mvae_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, _p(t_done),
                                          grad_scale, flags, _p(guard), _stream()), "mvae_adam_step_dev")


def rmsprop_step(p, g, v, lr, rho=0.9, eps=1e-8, grad_scale=1.0, zero_grad=False, guard=None):
    hl.check(hl.load().mvae_rmsprop_step(_p(p), _p(g), _p(v), p.numel(), lr, rho, eps, grad_scale, int(bool(zero_grad)),
                                         _p(guard), _stream()), "mvae_rmsprop_step")


def scalars_accumulate(acc, x, alpha, plain_mask=0):
    """acc[i] += (bit i of plain_mask ? 1 : alpha) * x[i]"""
    hl.check(hl.load().mvae_scalars_accumulate(_p(acc), _p(x), x.numel(), float(alpha), int(plain_mask), _stream()),
             "mvae_scalars_accumulate")


def signature_head_fwd(zh, off, SD, B, out, target=None, row_weight=None, scalars=None):
    hl.check(hl.load().mvae_signature_head_fwd(_pv(zh), zh.stride(0), int(off), int(SD), int(B), _p(target), _p(row_weight), _p(out),
                                               _p(scalars), _stream()), "mvae_signature_head_fwd")


def signature_head_bwd(dz, off, SD, B, out, target, row_weight, weight):
    hl.check(hl.load().mvae_signature_head_bwd(_pv(dz), dz.stride(0), int(off), int(SD), int(B), _p(out), _p(target), _p(row_weight),
                                               float(weight), _stream()), "mvae_signature_head_bwd")


def softmax_bwd_add(probs, dprobs, dlogits, R, N, NP):
    hl.check(hl.load().mvae_softmax_bwd_add(_p(probs), _p(dprobs), _p(dlogits), kind_of(dlogits), int(R), int(N), int(NP), _stream()),
             "mvae_softmax_bwd_add")


def bi_concat(f, r, cat, cat_rev, T, B, H):
    """cat (T,B,2H) = [f[t] | r[T-1-t]], cat_rev = cat reversed in time (or None)"""
    hl.check(hl.load().mvae_bi_concat(_p(f), _p(r), _p(cat), _p(cat_rev), kind_of(cat), int(T), int(B), int(H), _stream()),
             "mvae_bi_concat")


def add_time_reversed(dst, a, b, T, slab):
    """dst[t] = (a[t] if a is not None else 0) + b[T-1-t] over T slabs of ``slab`` elements"""
    hl.check(hl.load().mvae_add_time_reversed(_p(dst), _p(a), _p(b), kind_of(dst), int(T), int(slab), _stream()),
             "mvae_add_time_reversed")


def copy2d(dst, src, rows, cols, src_row0=0, zero_rows=0):
    """dst[:rows, :cols] = src[src_row0 : src_row0 + rows, :cols] (f32, row strides from the views); the first ``zero_rows`` rows
    of dst are zeroed instead"""
    hl.check(hl.load().mvae_copy2d_f32(_pv(dst), dst.stride(0), _pv(src), src.stride(0), int(rows), int(cols), int(src_row0),
                                       int(zero_rows), _stream()), "mvae_copy2d_f32")


def history_from_latent(mu, logvar, eps2, B, B_pad, Z, hist, z_out=None, prev=None):
    """the fused history pre-pass: z' = mu + exp(logvar / 2) * eps2 -> z_out rows [0, B); hist (a column-block view of
    [z | history]) row b = z'[b-1], row 0 = prev (or zeros), rows B.. zero"""
    _note_scalar(3, B)          # (the real number of windows: a call parameter of a step plan, ParamInt)
    hl.check(hl.load().mvae_history_from_latent(_p(mu), _p(logvar), _pv(eps2), int(B), int(B_pad), int(Z), _pv(hist), hist.stride(0),
                                                _pv(prev), _pv(z_out), z_out.stride(0) if z_out is not None else 0, _stream()),
             "mvae_history_from_latent")
