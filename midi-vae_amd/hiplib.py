"""ctypes binding of libmidivae_hip.so (C ABI declared in include/midivae_hip.h).

The library is built in-tree by ``make -C midi-vae_amd/csrc`` (or ``__graft_entry__.build()``).  Loading fails
LOUDLY when it is missing: there is no CPU fallback anywhere in the product path.

Device pointers are plain integers here (``tensor.data_ptr()``); streams are the raw ``hipStream_t`` value
(``torch.cuda.current_stream().cuda_stream``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVAE_LIB", os.path.join(_HERE, "libmidivae_hip.so"))   # MVAE_LIB: experiment builds

GRU, LSTM, RNN = 0, 1, 2
F32, BF16, ONEHOT = 0, 1, 2
X_DENSE, X_INDEX, X_SCALAR, X_CONST = 0, 1, 2, 3
ACT_NONE, ACT_TANH = 0, 1
ROWMAJOR, TILE16, TILE16P, TILE16Q = 0, 1, 2, 3
TABLE_ROWMAJOR, TABLE_PAIRED, TABLE_PAIRED8 = 0, 1, 2
CELL_CODE = {"GRU": GRU, "LSTM": LSTM, "SimpleRNN": RNN}
GATES = {GRU: 3, LSTM: 4, RNN: 1}
E_ARG, E_UNSUPPORTED, E_LAUNCH, E_FORMAT = -1, -2, -3, -4
HOST_F64, HOST_F32, HOST_U8 = 0, 1, 2
ABI_VERSION = 9
ERRORS = {-1: "MVAE_E_ARG (bad argument)", -2: "MVAE_E_UNSUPPORTED (shape/dtype not built)",
          -3: "MVAE_E_LAUNCH (HIP launch failed)", -4: "MVAE_E_FORMAT (a row is not one-hot)"}

_i32, _f32, _vp, _sz, _i64 = C.c_int32, C.c_float, C.c_void_p, C.c_size_t, C.c_int64


class RnnFwdArgs(C.Structure):
    _fields_ = [("cell", _i32), ("dtype", _i32), ("xmode", _i32), ("T", _i32), ("B", _i32), ("H", _i32),
                ("u_pack", _vp), ("xp", _vp), ("idx", _vp), ("table", _vp), ("xs", _vp), ("w_row", _vp),
                ("bias", _vp), ("xp0", _vp), ("h0", _vp), ("c0", _vp), ("hs", _vp), ("cs", _vp), ("acts", _vp),
                ("h_last", _vp), ("c_last", _vp), ("h0_ld", _i32), ("h_last_ld", _i32),
                ("chunk_steps", _i32), ("wait_ready", _vp), ("wait_value", C.c_uint32), ("signal_done", _vp), ("status", _vp),
                ("seq_layout", _i32), ("table_layout", _i32)]


class RnnBwdArgs(C.Structure):
    _fields_ = [("cell", _i32), ("dtype", _i32), ("T", _i32), ("B", _i32), ("H", _i32),
                ("ut_pack", _vp), ("hs", _vp), ("cs", _vp), ("acts", _vp), ("dhs_ext", _vp), ("dh_last", _vp),
                ("dc_last", _vp), ("da", _vp), ("rh", _vp), ("dh0", _vp), ("dc0", _vp), ("dh_last_ld", _i32), ("dh0_ld", _i32),
                ("chunk_steps", _i32), ("wait_ready", _vp), ("wait_value", C.c_uint32), ("signal_done", _vp), ("status", _vp),
                ("seq_layout", _i32)]


class GemmArgs(C.Structure):
    _fields_ = [("M", _i32), ("N", _i32), ("K", _i32), ("trans_a", _i32), ("trans_b", _i32),
                ("a_kind", _i32), ("b_kind", _i32), ("c_kind", _i32), ("lda", _i32), ("ldb", _i32), ("ldc", _i32),
                ("accumulate", _i32), ("act", _i32), ("split_k", _i32), ("alpha", _f32),
                ("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("c_layout", _i32), ("max_blocks", _i32),
                ("sys_release", _i32), ("chunk_rows", _i32), ("chunk_reverse", _i32), ("chunk_wait", _vp),
                ("chunk_wait_value", C.c_uint32), ("chunk_done", _vp), ("chunk_status", _vp), ("colsum_b", _vp),
                ("k_wait", _vp), ("k_wait_value", C.c_uint32), ("k_chunk_rows", C.c_int32), ("k_reverse", C.c_int32)]


class XpandArgs(C.Structure):
    _fields_ = [("xs", _vp), ("w", _vp), ("bias", _vp), ("out", _vp), ("out_kind", _i32), ("R", _i32), ("N", _i32),
                ("chunk_rows", _i32), ("chunk_done", _vp), ("blocks", _i32), ("reserved", _i32), ("idx", _vp), ("table", _vp)]


class PrepJob(C.Structure):
    _fields_ = [("op", _i32), ("kind", _i32), ("a", _i32), ("b", _i32), ("c", _i32), ("reserved", _i32),
                ("src", _vp), ("src2", _vp), ("dst", _vp)]


PREP_PACK_RECURRENT, PREP_MAKE_TABLE, PREP_TRANSPOSE_CONVERT, PREP_CONVERT, PREP_ZERO, PREP_CONVERT_PAD = 0, 1, 2, 3, 4, 5
PREP_ADD_I32, PREP_BROADCAST_ROWS = 6, 7
ADAM_ZERO_GRAD, ADAM_KEEP_COUNT = 1, 2


class HeadArgs(C.Structure):
    _fields_ = [("kind", _i32), ("dtype", _i32), ("R", _i32), ("H", _i32), ("N", _i32), ("want_grad", _i32),
                ("hs", _vp), ("wt", _vp), ("bias", _vp), ("target_idx", _vp), ("target_val", _vp),
                ("row_weight", _vp), ("grad_scale", _f32), ("probs", _vp), ("argmax", _vp), ("dlogits", _vp),
                ("scalars", _vp), ("b_stride", _i32), ("b_valid", _i32), ("wc", _vp), ("dhs", _vp), ("target_idx2", _vp)]


class LatentFwdArgs(C.Structure):
    _fields_ = [("B", _i32), ("Z", _i32), ("C", _i32), ("beta", _f32), ("prior_mean", _f32), ("prior_std", _f32),
                ("inv_batch", _f32), ("mu", _vp), ("logvar", _vp), ("eps", _vp), ("style_target", _vp),
                ("style_row_weight", _vp), ("z", _vp), ("style_probs", _vp), ("scalars", _vp), ("ldz", _i32)]


class LatentBwdArgs(C.Structure):
    _fields_ = [("B", _i32), ("Z", _i32), ("C", _i32), ("beta", _f32), ("prior_mean", _f32), ("prior_std", _f32),
                ("style_weight", _f32), ("inv_batch", _f32), ("mu", _vp), ("logvar", _vp), ("eps", _vp), ("dz", _vp),
                ("style_probs", _vp), ("style_target", _vp), ("style_row_weight", _vp), ("dmu", _vp),
                ("dlogvar", _vp), ("lddz", _i32)]


class LatentChainFwdArgs(C.Structure):
    _fields_ = ([(n, _i32) for n in ("B", "B_valid", "H", "Z", "C", "ncat", "zin", "n_init", "split")] +
                [(n, _f32) for n in ("beta", "prior_mean", "prior_std", "inv_batch")] +
                [(n, _vp) for n in ("cat", "w_pack", "b_pack", "w_extra", "b_extra", "w_mu", "b_mu", "w_lv", "b_lv", "w_init",
                                    "b_init", "eps", "style_target", "style_row_weight", "pack", "extra", "mu", "logvar", "zh",
                                    "style_probs", "scalars", "S")])


class LatentChainBwdArgs(C.Structure):
    _fields_ = ([(n, _i32) for n in ("B", "B_valid", "H", "Z", "C", "ncat", "zin", "n_init", "split")] +
                [(n, _f32) for n in ("beta", "prior_mean", "prior_std", "style_weight", "inv_batch")] +
                [(n, _vp) for n in ("wt_pack", "wt_extra", "wt_mu", "wt_lv", "wt_init", "S", "pack", "extra", "mu", "logvar", "eps",
                                    "style_probs", "style_target", "style_row_weight", "dS", "dzh", "dmu", "dlogvar",
                                    "d_extra", "d_pack", "dcat")])


# name -> (restype, argtypes); every symbol include/midivae_hip.h declares
SIGNATURES = {
    "mvae_abi_version": (_i32, []),
    "mvae_rnn_producer_waves": (_i32, [_i32]),
    "mvae_build_info": (C.c_char_p, []),
    "mvae_rnn_fwd": (_i32, [C.POINTER(RnnFwdArgs), _vp]),
    "mvae_rnn_bwd": (_i32, [C.POINTER(RnnBwdArgs), _vp]),
    "mvae_rnn_fwd_multi": (_i32, [C.POINTER(RnnFwdArgs), _i32, C.POINTER(XpandArgs), _i32, _vp]),
    "mvae_rnn_bwd_multi": (_i32, [C.POINTER(RnnBwdArgs), _i32, _vp]),
    "mvae_pack_recurrent": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mvae_gemm": (_i32, [C.POINTER(GemmArgs), _vp]),
    "mvae_gemm_kstream_multi": (_i32, [C.POINTER(GemmArgs), _i32, _vp]),
    "mvae_gemm_multi": (_i32, [C.POINTER(GemmArgs), _i32, _vp]),
    "mvae_colsum": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mvae_stream_wait_value32": (_i32, [_vp, _vp, C.c_uint32]),
    "mvae_stream_write_value32": (_i32, [_vp, _vp, C.c_uint32]),
    "mvae_occupancy": (_i32, [_i32]),
    "mvae_streams_alias": (_i32, [_vp, _vp, _vp, C.c_uint32]),
    "mvae_prepare_batch": (_i32, [_vp, _i32, _vp]),
    "mvae_outer_bias_tile16": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mvae_gather2_tile16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mvae_colsum_weighted": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    "mvae_sum_over_time": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "mvae_head": (_i32, [C.POINTER(HeadArgs), _vp]),
    "mvae_head_np": (_i32, [_i32]),
    "mvae_latent_fwd": (_i32, [C.POINTER(LatentFwdArgs), _vp]),
    "mvae_latent_bwd": (_i32, [C.POINTER(LatentBwdArgs), _vp]),
    "mvae_latent_chain_fwd": (_i32, [C.POINTER(LatentChainFwdArgs), _vp]),
    "mvae_latent_chain_bwd": (_i32, [C.POINTER(LatentChainBwdArgs), _vp]),
    "mvae_relayout": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mvae_tanh_bwd": (_i32, [_vp, _vp, _vp, _sz, _vp]),
    "mvae_convert": (_i32, [_vp, _i32, _vp, _i32, _sz, _vp]),
    "mvae_make_table": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mvae_transpose_convert": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mvae_adam_step": (_i32, [_vp, _vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "mvae_adam_step_dev": (_i32, [_vp, _vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _vp, _f32, _i32, _vp, _vp]),
    "mvae_rmsprop_step": (_i32, [_vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "mvae_scalars_accumulate": (_i32, [_vp, _vp, _i32, _f32, C.c_uint32, _vp]),
    "mvae_copy2d_f32": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mvae_history_from_latent": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp]),
    "mvae_signature_head_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mvae_signature_head_bwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _vp]),
    "mvae_softmax_bwd_add": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mvae_bi_concat": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mvae_add_time_reversed": (_i32, [_vp, _vp, _vp, _i32, _i32, _sz, _vp]),
    "mvae_event_create": (_i32, [C.POINTER(_vp)]),
    "mvae_event_create_timed": (_i32, [C.POINTER(_vp)]),
    "mvae_event_elapsed_ms": (_i32, [_vp, _vp, C.POINTER(C.c_float)]),
    "mvae_event_destroy": (_i32, [_vp]),
    "mvae_event_record": (_i32, [_vp, _vp]),
    "mvae_event_synchronize": (_i32, [_vp]),
    "mvae_stream_wait_event": (_i32, [_vp, _vp]),
    "mvae_plan_create": (_i32, [C.POINTER(_vp)]),
    "mvae_plan_destroy": (_i32, [_vp]),
    "mvae_plan_add_call": (_i32, [_vp, C.c_char_p, C.POINTER(C.c_uint64), _i32]),
    "mvae_plan_set_blob": (_i32, [_vp, _i32, _i32, _vp, _sz]),
    "mvae_plan_add_patch": (_i32, [_vp, _i32, _i32, _i64, _i32, _i64]),
    "mvae_plan_run": (_i32, [_vp, _i32, _i32, C.POINTER(C.c_uint64), _i32]),
    "mvae_plan_size": (_i32, [_vp]),
    "mvae_plan_failed_call": (_i32, [_vp]),
    "mvae_host_threads": (_i32, [_i32]),
    "mvae_host_onehot_to_index_tm": (_i32, [_vp, _i32, _i64, _i32, _i32, _i64, _i64, _vp, _i32, C.c_uint8, C.POINTER(_i64)]),
    "mvae_host_index_to_tm": (_i32, [_vp, _i64, _i32, _i64, _i64, _vp, _i32, C.c_uint8]),
    "mvae_host_twohot_to_index_tm": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _i64, _i64, _vp, _vp, _i32, C.c_uint8, C.POINTER(_i64)]),
    "mvae_host_rows_to_tm_f32": (_i32, [_vp, _i32, _i64, _i32, _i64, _i64, _f32, _vp, _i32]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load the C-ABI library once; raise HipLibraryMissing (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found - build it with `make -C midi-vae_amd/csrc` (or __graft_entry__.build()); "
            "the MIDI-VAE engine has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = ABI drift between header and library
        fn.restype = res
        fn.argtypes = args
    if lib.mvae_abi_version() != ABI_VERSION:
        raise HipLibraryMissing("ABI version mismatch: library %d, binding %d - rebuild with `make -C midi-vae_amd/csrc`"
                                % (lib.mvae_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s" % (what, ERRORS.get(code, code)))


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()
