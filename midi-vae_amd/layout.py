"""Model specification and parameter layout of the MIDI-VAE graph.

``ModelSpec`` is the validated subset of the reference's ``VAE.create`` keyword surface (reference
vae_definition.py:40-102) that the device engine implements; anything outside it raises NotImplementedError
naming the switch, instead of silently building a different model.

``ParamLayout`` places every trainable tensor in ONE flat f32 buffer (parameters, gradients and both Adam moments
use the same layout, so the optimizer is a single pass and the data-parallel all-reduce is one bucket per group).
Names are the contract shared with the oracle (oracle/vae_oracle.py:param_shapes).  All initial-state Denses of the
decoder read the same input [z | history] (reference vae_definition.py:548-568,596-604,634-642), so they are stored as
column blocks of ONE (2Z, nInit*H) matrix and evaluated by one GEMM; the per-Dense names are strided views of it.
"""
from __future__ import annotations

from collections import OrderedDict
import dataclasses
from dataclasses import dataclass, field

import numpy as np

GATES = {"GRU": 3, "LSTM": 4, "SimpleRNN": 1}
NSTATE = {"GRU": 1, "LSTM": 2, "SimpleRNN": 1}
_ALIGN = 64  # floats (256 B): every tensor starts on a boundary that keeps 16-byte vector access legal


@dataclass
class ModelSpec:
    cell: str = "GRU"
    H: int = 256
    Z: int = 256
    Din: int = 61
    Dout: int = 61
    T: int = 64
    V: int = 4
    ID: int = 16
    C: int = 2
    Le: int = 2
    Ld: int = 2
    meta_instrument: bool = True
    meta_velocity: bool = True
    extra_layer: bool = True
    split: bool = True
    history: bool = True
    style: bool = True
    w_instr: float = 0.1
    w_vel: float = 1.0
    w_style: float = 0.1
    beta: float = 0.1
    prior_mean: float = 0.0
    prior_std: float = 1.0
    epsilon_std: float = 0.01
    lr: float = 2e-4
    optimizer: str = "Adam"
    # off by default in the reference (settings.py:217,227): held-notes roll through the encoder + a 2-way decoder head
    # (vae_definition.py:476-480,648-683), a second notes stack predicting the NEXT window (:685-726)
    meta_held: bool = False
    w_held: float = 1.0
    meta_next: bool = False
    w_next: float = 1.0
    # the optional heads of reference vae_definition.py:737-761 and the decoder's additional input (:553-556), all off by default
    signature: bool = False          # tanh(z[:, off:off+SD]) against the song's signature vector, mse
    SD: int = 15
    w_sig: float = 1.0
    comp_notes: bool = False         # a Keras RNN + Dense softmax style classifier on the decoder's notes OUTPUT
    w_cnotes: float = 1.0
    comp_instr: bool = False         # ... on its instrument OUTPUT
    w_cinstr: float = 1.0
    add_dim: int = 0                 # width of decoder_additional_input (composer one-hot and / or signature vector)
    bidirectional: bool = False      # reference vae_definition.py:445-453 (Le-2 Bidirectional layers + one on top, as written)
    # attach_instruments (reference import_midi.py:288-292, settings.py:186-187,207-208): every notes row carries the one-hot
    # instrument category of its voice behind the pitch one-hot - Din and Dout INCLUDE these ``attach`` columns (two-hot rows)
    attach: int = 0

    def oracle_cfg(self):
        """dict accepted by oracle.vae_oracle.make_cfg (tests only)."""
        d = dict(self.__dict__)
        d.pop("epsilon_std")
        d.pop("attach")                 # (the oracle multiplies the dense two-hot rows: Din / Dout say it all)
        return d

    @property
    def G(self):
        return GATES[self.cell]

    @property
    def GH(self):
        return GATES[self.cell] * self.H

    @property
    def nstate(self):
        return NSTATE[self.cell]

    @property
    def zin(self):
        return (2 * self.Z if self.history else self.Z) + self.add_dim

    def enc_layers(self):
        """[[(prefix, reversed?, input width)]] of the encoder's notes stack, bottom to top (oracle.vae_oracle.enc_notes_layers):
        the reference's bidirectional loop is ``range(1, Le-1)`` - Le-2 Bidirectional(concat) layers and ONE plain layer on top."""
        if not self.bidirectional:
            return [[("enc.notes.%d" % l, False, self.Din if l == 0 else self.H)] for l in range(self.Le)]
        nbi = max(self.Le - 2, 0)
        out = []
        for l in range(nbi):
            k = self.Din if l == 0 else 2 * self.H
            out.append([("enc.notes.%d" % l, False, k), ("enc.notes.%d.rev" % l, True, k)])
        out.append([("enc.notes.%d" % nbi, False, self.Din if nbi == 0 else 2 * self.H)])
        return out

    @property
    def sig_off(self):
        """the signature head reads z behind the style classifier's columns (reference vae_definition.py:739-743)"""
        return self.C if self.style else 0

    @property
    def ncat(self):
        return 1 + int(self.meta_instrument) + int(self.meta_velocity) + int(self.meta_held)

    @property
    def has_pack(self):
        """the pack Dense exists when instrument or velocity rolls are on - the reference's condition repeats meta_instrument and
        omits meta_held_notes (vae_definition.py:483), reproduced as written"""
        return self.meta_instrument or self.meta_velocity

    @property
    def tail_in(self):
        """width of what the extra Dense (or, without it, the split) receives"""
        return self.H if self.has_pack else self.ncat * self.H


_UNSUPPORTED_SWITCHES = (
    ("use_embedding", False),
)


def spec_from_create_kwargs(kw: dict) -> ModelSpec:
    """Validate the reference's ``VAE.create`` keyword arguments and translate them.  Mirrors the asserts of
    reference vae_definition.py:177-208."""
    g = kw.get
    for name, allowed in _UNSUPPORTED_SWITCHES:
        if g(name, allowed) != allowed:
            raise NotImplementedError("VAE.create(%s=%r) is outside the implemented hot path (SURVEY section 8f)"
                                      % (name, g(name)))
    for name, want in (("lstm_activation", "tanh"), ("lstm_state_activation", "tanh"),
                       ("activation_before_splitting", "tanh"), ("activation", "softmax"),
                       ("meta_instrument_activation", "softmax"), ("meta_velocity_activation", "sigmoid"),
                       ("vae_loss", "categorical_crossentropy"), ("signature_activation", "tanh"),
                       ("composer_decoder_at_notes_activation", "softmax"),
                       ("composer_decoder_at_instrument_activation", "softmax")):
        if g(name, want) != want:
            raise NotImplementedError("VAE.create(%s=%r): only %r is implemented" % (name, g(name), want))
    if g("epsilon_factor", 0.0) != 0.0:
        raise NotImplementedError("epsilon_factor > 0 is dead code in the reference (vae_definition.py:509-512)")
    if not g("decode", True):
        raise NotImplementedError("decode=False")
    cell = g("cell_type", "LSTM")
    if cell not in GATES:
        raise ValueError("cell_type must be one of %s" % sorted(GATES))
    opt = g("optimizer", "Adam")
    if opt not in ("Adam", "RMSprop"):
        raise ValueError("optimizer must be 'Adam' or 'RMSprop' (reference vae_definition.py:174-175)")
    s = ModelSpec(
        cell=cell, H=int(g("lstm_size", 256)), Z=int(g("latent_rep_size", 256)), Din=int(g("input_dim", 64)),
        Dout=int(g("output_dim", 64)), T=int(g("output_length", 16)), V=int(g("meta_instrument_length", 0) or 1),
        ID=int(g("meta_instrument_dim", 0) or 1), C=int(g("num_composers", 0)), Le=int(g("num_layers_encoder", 1)),
        Ld=int(g("num_layers_decoder", 1)), meta_instrument=bool(g("meta_instrument", False)),
        meta_velocity=bool(g("meta_velocity", False)), extra_layer=bool(g("extra_layer", False)),
        split=bool(g("split_lstm_vector", True)), history=bool(g("history", True)),
        style=bool(g("include_composer_decoder", False)), w_instr=float(g("meta_instrument_weight", 1.0)),
        w_vel=float(g("meta_velocity_weight", 1.0)), w_style=float(g("composer_weight", 1.0)), beta=float(g("beta", 0.01)),
        prior_mean=float(g("prior_mean", 0.0)), prior_std=float(g("prior_std", 1.0)),
        epsilon_std=float(g("epsilon_std", 1.0)), lr=float(g("learning_rate", 0.001)), optimizer=opt,
        meta_held=bool(g("meta_held_notes", False)), w_held=float(g("meta_held_notes_weight", 1.0)),
        meta_next=bool(g("meta_next_notes", False)), w_next=float(g("meta_next_notes_weight", 1.0)),
        signature=bool(g("signature_decoder", False)), SD=int(g("signature_dim", 15) or 15), w_sig=float(g("signature_weight", 1.0)),
        comp_notes=bool(g("composer_decoder_at_notes_output", False)), w_cnotes=float(g("composer_decoder_at_notes_weight", 1.0)),
        comp_instr=bool(g("composer_decoder_at_instrument_output", False)),
        w_cinstr=float(g("composer_decoder_at_instrument_weight", 1.0)),
        add_dim=int(g("decoder_additional_input_dim", 0)) if g("decoder_additional_input", False) else 0,
        bidirectional=bool(g("bidirectional", False)), attach=int(g("attach_dim", 0) or 0))
    # the asserts of reference vae_definition.py:177-208
    assert s.Le > 0 and s.Ld > 0 and s.T > 0 and s.H > 0 and s.Z > 0 and s.beta > 0
    assert int(g("input_length", s.T)) > 0
    if int(g("input_length", s.T)) != s.T:
        raise NotImplementedError("input_length != output_length")
    if s.meta_instrument:
        assert g("meta_instrument_dim", 0) > 0 and s.w_instr > 0
        if not g("meta_instrument_length", 0) > 0:
            raise NotImplementedError("meta_instrument_length == 0 (non-sequence instrument input)")
    if s.meta_velocity:
        assert s.w_vel > 0 and g("meta_velocity_length", 0) > 0
        if int(g("meta_velocity_length")) != s.T:
            raise NotImplementedError("meta_velocity_length != output_length")
    if s.meta_held:
        assert s.w_held > 0 and g("meta_held_notes_length", 0) > 0
        if int(g("meta_held_notes_length")) != s.T:
            raise NotImplementedError("meta_held_notes_length != output_length")
        if g("meta_held_notes_activation", "softmax") != "softmax":
            raise NotImplementedError("meta_held_notes_activation: only 'softmax' is implemented")
        if not s.has_pack and not s.extra_layer:
            raise NotImplementedError("meta_held_notes without instrument / velocity rolls and without extra_layer: the reference "
                                      "then splits an un-packed 2H vector (vae_definition.py:483-492), which is not built")
    if s.meta_next:
        assert s.w_next > 0 and g("meta_next_notes_output_length", 0) > 0
        if int(g("meta_next_notes_output_length")) != s.T:
            raise NotImplementedError("meta_next_notes_output_length != output_length")
    # (teacher forcing - teacher_force / meta_next_notes_teacher_force - only changes what the unused readout state holds,
    #  the cell graph never reads it, so the switch is accepted and has no effect, as in the reference)
    if s.signature:
        assert s.w_sig > 0 and s.SD > 0
        if s.sig_off + s.SD > s.Z:
            raise ValueError("signature_dim does not fit the latent behind the style columns")
    if s.comp_notes or s.comp_instr:
        assert 0 < s.C
        if s.comp_instr and not s.meta_instrument:
            raise ValueError("composer_decoder_at_instrument_output needs meta_instrument")
    if s.style:
        assert 0 < s.C <= min(s.Z, 64)
    if s.H % 64 or s.H > 256:
        raise NotImplementedError("lstm_size must be 64, 128 or 256 (got %d)" % s.H)
    if s.attach:
        if not (0 < s.attach < min(s.Din, s.Dout)) or s.Din != s.Dout:
            raise ValueError("attach_dim must be the instrument columns appended to every notes row (input_dim == output_dim)")
        if s.bidirectional or s.meta_next or s.comp_notes:
            raise NotImplementedError("attach_instruments with a bidirectional encoder, the next-notes head or a classifier on the "
                                      "notes output")
    if s.Dout > 128 or s.ID > 128 or s.Din > 255:
        raise NotImplementedError("one-hot widths above 128 are not built")
    return s


def dec_init_blocks(spec: ModelSpec):
    """Ordered list of (param prefix) for every initial-state Dense; block k occupies columns [k*H,(k+1)*H)."""
    out = []
    for l in range(spec.Ld):
        for s in range(spec.nstate):
            out.append("dec.notes.init.%d.%d" % (l, s))
    if spec.meta_instrument:
        out += ["dec.instr.init.%d" % s for s in range(spec.nstate)]
    if spec.meta_velocity:
        out += ["dec.vel.init.%d" % s for s in range(spec.nstate)]
    if spec.meta_held:
        out += ["dec.held.init.%d" % s for s in range(spec.nstate)]
    if spec.meta_next:
        for l in range(spec.Ld):
            out += ["dec.next.init.%d.%d" % (l, s) for s in range(spec.nstate)]
    return out


@dataclass
class Entry:
    offset: int
    shape: tuple
    row_stride: int      # floats between rows (== shape[-1] unless the tensor is a column block)
    group: str = "enc"   # gradient bucket: 'dec' completes first in backward, 'enc' last


@dataclass
class ParamLayout:
    spec: ModelSpec
    entries: "OrderedDict[str, Entry]" = field(default_factory=OrderedDict)
    total: int = 0
    dec_begin: int = 0       # [dec_begin, total) holds decoder-side tensors (first gradient bucket)

    @staticmethod
    def build(spec: ModelSpec) -> "ParamLayout":
        L = ParamLayout(spec)
        H, GH, Z = spec.H, spec.GH, spec.Z
        cur = 0

        def add(name, shape, group):
            nonlocal cur
            n = int(np.prod(shape))
            L.entries[name] = Entry(cur, tuple(shape), shape[-1], group)
            cur += (n + _ALIGN - 1) // _ALIGN * _ALIGN

        def rnn(prefix, k, group):
            add(prefix + ".W", (k, GH), group)
            add(prefix + ".U", (H, GH), group)
            add(prefix + ".b", (GH,), group)

        for layer in spec.enc_layers():
            for prefix, _, k in layer:
                rnn(prefix, k, "enc")
        ncat = 1
        if spec.meta_instrument:
            rnn("enc.instr", spec.ID, "enc")
            ncat += 1
        if spec.meta_velocity:
            rnn("enc.vel", 1, "enc")
            ncat += 1
        if spec.meta_held:
            rnn("enc.held", 2, "enc")
            ncat += 1
        if spec.has_pack:
            add("enc.pack.W", (ncat * H, H), "enc")
            add("enc.pack.b", (H,), "enc")
        if spec.extra_layer:
            add("enc.extra.W", (spec.tail_in, H), "enc")
            add("enc.extra.b", (H,), "enc")
        h1 = H // 2 if spec.split else H
        h2 = H - H // 2 if spec.split else H
        add("enc.zmean.W", (h1, Z), "enc")
        add("enc.zmean.b", (Z,), "enc")
        add("enc.zlogvar.W", (h2, Z), "enc")
        add("enc.zlogvar.b", (Z,), "enc")
        L.dec_begin = cur
        blocks = dec_init_blocks(spec)
        nb = len(blocks)
        add("dec.init.W", (spec.zin, nb * H), "dec")
        add("dec.init.b", (nb * H,), "dec")
        wi, bi = L.entries["dec.init.W"], L.entries["dec.init.b"]
        for k, pre in enumerate(blocks):
            L.entries[pre + ".W"] = Entry(wi.offset + k * H, (spec.zin, H), nb * H, "dec")
            L.entries[pre + ".b"] = Entry(bi.offset + k * H, (H,), H, "dec")
        for l in range(spec.Ld):
            rnn("dec.notes.%d" % l, spec.Dout if l == 0 else H, "dec")
        add("dec.notes.out.W", (H, spec.Dout), "dec")
        add("dec.notes.out.b", (spec.Dout,), "dec")
        if spec.meta_instrument:
            rnn("dec.instr.cell", spec.ID, "dec")
            add("dec.instr.out.W", (H, spec.ID), "dec")
            add("dec.instr.out.b", (spec.ID,), "dec")
        if spec.meta_velocity:
            rnn("dec.vel.cell", 1, "dec")
            add("dec.vel.out.W", (H, 1), "dec")
            add("dec.vel.out.b", (1,), "dec")
        if spec.meta_held:
            rnn("dec.held.cell", 2, "dec")
            add("dec.held.out.W", (H, 2), "dec")
            add("dec.held.out.b", (2,), "dec")
        if spec.meta_next:
            for l in range(spec.Ld):
                rnn("dec.next.%d" % l, spec.Dout if l == 0 else H, "dec")
            add("dec.next.out.W", (H, spec.Dout), "dec")
            add("dec.next.out.b", (spec.Dout,), "dec")
        for key, flag, k in (("cnotes", spec.comp_notes, spec.Dout), ("cinstr", spec.comp_instr, spec.ID)):
            if flag:
                rnn(key + ".rnn", k, "dec")
                add(key + ".out.W", (H, spec.C), "dec")
                add(key + ".out.b", (spec.C,), "dec")
        L.total = cur
        return L

    # ---- names ---------------------------------------------------------------------------------------
    def oracle_names(self):
        """The per-tensor names (what the oracle uses); the two combined 'dec.init.*' holders are internal."""
        return [n for n in self.entries if n not in ("dec.init.W", "dec.init.b")]

    def n_params(self):
        return int(sum(np.prod(self.entries[n].shape) for n in self.oracle_names()))

    # ---- host <-> flat ---------------------------------------------------------------------------------
    def view(self, flat, name):
        """View of tensor ``name`` inside a flat torch tensor / numpy array (strided for column blocks)."""
        e = self.entries[name]
        if len(e.shape) == 1:
            return flat[e.offset:e.offset + e.shape[0]]
        rows, cols = e.shape
        if e.row_stride == cols:
            return flat[e.offset:e.offset + rows * cols].reshape(rows, cols)
        span = flat[e.offset:e.offset + (rows - 1) * e.row_stride + cols]
        if isinstance(span, np.ndarray):
            return np.lib.stride_tricks.as_strided(span, (rows, cols), (e.row_stride * span.itemsize, span.itemsize))
        return span.as_strided((rows, cols), (e.row_stride, 1))

    def pack(self, named: dict) -> np.ndarray:
        flat = np.zeros((self.total,), np.float32)
        for n in self.oracle_names():
            self.view(flat, n)[...] = np.asarray(named[n], np.float32)
        return flat

    def unpack(self, flat) -> "OrderedDict[str, np.ndarray]":
        flat = np.asarray(flat)
        return OrderedDict((n, np.array(self.view(flat, n))) for n in self.oracle_names())


@dataclass
class ClassifierSpec:
    """One of the reference's three style classifiers (pitch_classifier.py:89-103, velocity_classifier.py:110-125,
    instrument_classifier.py:93-107): ``L`` Keras GRU(H) layers over a roll -> Dense(C, softmax).  ``K`` = input width; ``xmode``
    'index' for one-hot rolls (pitch: K=61, instrument: K=16), 'scalar' for the velocity roll (K=1)."""
    K: int = 61
    T: int = 64
    C: int = 2
    H: int = 256
    L: int = 2
    cell: str = "GRU"
    xmode: str = "index"
    lr: float = 2e-5
    optimizer: str = "Adam"
    # what the shared engine code reads off a spec
    meta_held: bool = False
    meta_next: bool = False

    @property
    def G(self):
        return GATES[self.cell]

    @property
    def GH(self):
        return GATES[self.cell] * self.H

    @property
    def Le(self):
        return self.L

    @property
    def Ld(self):
        return 1

    def oracle_cfg(self):
        return dict(cell=self.cell, H=self.H, K=self.K, L=self.L, C=self.C)


def classifier_layout(spec: ClassifierSpec) -> ParamLayout:
    """rnn.<l>.{W,U,b}, cls.out.{W,b} in one flat f32 buffer (names shared with oracle/classifier_oracle.py)"""
    L = ParamLayout(spec)
    cur = 0
    for l in range(spec.L):
        for suffix, shape in ((".W", (spec.K if l == 0 else spec.H, spec.GH)), (".U", (spec.H, spec.GH)), (".b", (spec.GH,))):
            L.entries["rnn.%d%s" % (l, suffix)] = Entry(cur, tuple(shape), shape[-1], "enc")
            cur += (int(np.prod(shape)) + _ALIGN - 1) // _ALIGN * _ALIGN
    for name, shape in (("cls.out.W", (spec.H, spec.C)), ("cls.out.b", (spec.C,))):
        L.entries[name] = Entry(cur, tuple(shape), shape[-1], "enc")
        cur += (int(np.prod(shape)) + _ALIGN - 1) // _ALIGN * _ALIGN
    L.total, L.dec_begin = cur, 0
    return L


def init_classifier_params(spec: ClassifierSpec, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Keras defaults: kernels glorot_uniform, recurrent kernels orthogonal, biases zero"""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for n, e in classifier_layout(spec).entries.items():
        out[n] = (_orthogonal(rng, e.shape) if n.endswith(".U") else _glorot_uniform(rng, e.shape) if n.endswith(".W")
                  else np.zeros(e.shape)).astype(np.float32)
    return out


# ----------------------------------------------------------------------------------------------------------
# Keras / recurrentshop initialisers
# ----------------------------------------------------------------------------------------------------------

def _glorot_uniform(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, shape)


def _orthogonal(rng, shape):
    a = rng.standard_normal(shape)
    try:        # a 256 x 1024 SVD on a BLAS pool sized for a 256-thread host takes 0.16-0.3 s, on one thread 0.05 s
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            u, _, vt = np.linalg.svd(a, full_matrices=False)
    except ImportError:
        u, _, vt = np.linalg.svd(a, full_matrices=False)
    return u if u.shape == shape else vt


def init_params(spec: ModelSpec, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Kernels glorot_uniform over the FULL concatenated width, recurrent kernels orthogonal, biases zero; the
    encoder's Keras LSTM layers get unit_forget_bias (b_f = 1), recurrentshop's decoder cells do not."""
    key = (dataclasses.astuple(spec), int(seed))
    hit = _INIT_CACHE.get(key)
    if hit is not None:                 # (an Engine draws its initial parameters at creation: the same spec again and again in one process)
        return OrderedDict((n, v.copy()) for n, v in hit.items())
    rng = np.random.default_rng(seed)
    L = ParamLayout.build(spec)
    H = spec.H
    out = OrderedDict()
    for n in L.oracle_names():
        shape = L.entries[n].shape
        if n.endswith(".U"):
            out[n] = _orthogonal(rng, shape)
        elif n.endswith(".W"):
            out[n] = _glorot_uniform(rng, shape)
        else:
            out[n] = np.zeros(shape)
            if spec.cell == "LSTM" and n.startswith(("enc.", "cnotes.", "cinstr.")) and (n[:-2] + ".U") in L.entries:
                out[n][H:2 * H] = 1.0
        out[n] = out[n].astype(np.float32)
    if len(_INIT_CACHE) >= 16:
        _INIT_CACHE.pop(next(iter(_INIT_CACHE)))
    _INIT_CACHE[key] = OrderedDict((n, v.copy()) for n, v in out.items())
    return out


_INIT_CACHE = {}
