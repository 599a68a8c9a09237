"""Drop-in mirror of the reference's `model.py` hot-path classes, backed by libganffn.so.

Same class names, constructor signatures `(D_h, dropout=0.2)`, forward signatures, train()/eval()
behaviour and state_dict keys/shapes as /root/reference/model.py:1178-1462, so that
train_IEMOCAP.py's `from model import ...` (train_IEMOCAP.py:18-29) can bind to this module
(INTEGRATION.md shows how).  All arithmetic of forward/backward runs in the HIP library
through gan_ffn_amd.ops; there is no CPU fallback (forward raises on CPU tensors).

Parameters of one network live in ONE contiguous fp32 slab (encoder layers first, then object /
fc1 / fc2 / fc3); every nn.Parameter is a view into it, named like the reference's.  The slab is what
the kernels, the fused Adam and the gradient all-reduce operate on (engine.py).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .ops import FF, N_LAYERS, LAYER_KEYS, layer_shapes

MAX_LEN = 110


class PositionalEncoding(nn.Module):
    """model.py:1178-1197.  `pe` buffer [max_len, 1, d_model], dropout 0.2."""

    def __init__(self, d_model: int, dropout: float = 0.2, max_len: int = MAX_LEN):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def forward(self, x):
        # standalone use only (inside the networks the add+dropout is fused into the encoder call)
        return ops.DropoutFn.apply(x + self.pe[: x.size(0)], self.dropout.p, self.training, 0)


class _WB(nn.Module):
    """weight/bias holder (nn.Linear / nn.LayerNorm stand-in: parameters only, no compute)."""

    def __init__(self, *wshape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape))
        self.bias = nn.Parameter(torch.empty(wshape[0]))


class _SelfAttnParams(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * E))
        self.out_proj = _WB(E, E)


class _LayerParams(nn.Module):
    """parameter container with nn.TransformerEncoderLayer's state_dict layout"""

    def __init__(self, E, F=FF):
        super().__init__()
        self.self_attn = _SelfAttnParams(E)
        self.linear1 = _WB(F, E)
        self.linear2 = _WB(E, F)
        self.norm1 = _WB(E)
        self.norm2 = _WB(E)

    def ordered(self):
        sa = self.self_attn
        return [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, self.linear1.weight,
                self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias,
                self.norm2.weight, self.norm2.bias]


class _EncoderParams(nn.Module):
    def __init__(self, E, L):
        super().__init__()
        self.layers = nn.ModuleList([_LayerParams(E) for _ in range(L)])
        self.enc_dropout = ops.ENC_DROPOUT  # nn.TransformerEncoderLayer default


def _a4(n):
    return (n + 3) & ~3


class _Net(nn.Module):
    """Common body of the six networks.  Subclasses set KIND ('gen'|'disc'), D_MODEL (None -> D_h),
    NHEAD, FC (hidden widths) and HAS_OBJECT (+ OBJECT_IN, the raw-modality width `object` maps to D_h)."""
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "gen", 100, 10, (512,), False
    OBJECT_IN = 512

    def __init__(self, D_h, dropout=0.2, num_layers=N_LAYERS):
        super().__init__()
        E = self.D_MODEL or D_h
        self.d_model, self.nhead, self.num_layers, self.D_h = E, self.NHEAD, num_layers, D_h
        self.position_encoding = PositionalEncoding(E)
        # Initial values come from stock torch constructors, created in the reference's order
        # (model.py:1209-1216) so that the same torch.manual_seed gives the same initial weights;
        # like nn.TransformerEncoder's deep copies, all layers start identical to the template.
        tmpl = nn.TransformerEncoderLayer(d_model=E, nhead=self.NHEAD)
        self.encoder_layer = _LayerParams(E)          # registered-but-unused template (model.py:1210-1213)
        self.transformer_encoder = _EncoderParams(E, num_layers)
        tsd = tmpl.state_dict()
        with torch.no_grad():
            for holder in [self.encoder_layer] + list(self.transformer_encoder.layers):
                for k, p in zip(LAYER_KEYS, holder.ordered()):
                    p.copy_(tsd[k])
        if self.HAS_OBJECT:
            self.object = self._init_linear(self.OBJECT_IN, 100)  # model.py:1344
        dims = [E] + list(self.FC) + ([D_h] if self.KIND == "gen" else [])
        if self.KIND == "disc":
            dims = [E, 64, 16, 1]                       # model.py:1311-1313
        for i in range(len(dims) - 1):
            setattr(self, "fc%d" % (i + 1), self._init_linear(dims[i], dims[i + 1]))
        self.n_fc = len(dims) - 1
        self.gelu = nn.GELU()
        if self.KIND == "disc":
            self.sigmoid = nn.Sigmoid()
        self.dropout = nn.Dropout(dropout)
        self._slab = None
        self._pack()

    @staticmethod
    def _init_linear(din, dout):
        ref = nn.Linear(din, dout)
        m = _WB(dout, din)
        with torch.no_grad():
            m.weight.copy_(ref.weight)
            m.bias.copy_(ref.bias)
        return m

    # ---- slab management -------------------------------------------------------------------
    def _slab_params(self):
        """ordered [(param, shape)] of everything that lives in the slab (NOT the template layer)."""
        ps = []
        for layer in self.transformer_encoder.layers:
            ps += layer.ordered()
        if self.HAS_OBJECT:
            ps += [self.object.weight, self.object.bias]
        for i in range(self.n_fc):
            fc = getattr(self, "fc%d" % (i + 1))
            ps += [fc.weight, fc.bias]
        return ps

    def slab_layout(self):
        """-> (total floats, [(offset, shape)] in _slab_params order, encoder floats)."""
        E, L = self.d_model, self.num_layers
        per = sum(_a4(int(torch.Size(s).numel())) for s in layer_shapes(E))
        views, off = [], 0
        for p in self._slab_params():
            views.append((off, tuple(p.shape)))
            off += _a4(p.numel())
        return off, views, per * L

    def _pack(self):
        """(re)build the slab.  When a slab of the right size already exists on the parameters' device it is re-filled IN
        PLACE, so a no-op `.to(same device)` / `.float()` after an engine captured `module.slab` does not disconnect the
        engine (which trains the slab) from the module (whose parameters, state_dict and checkpoints are views of it)."""
        ps = self._slab_params()
        total, views, _ = self.slab_layout()
        dev = ps[0].device
        old = self._slab
        slab = old if (old is not None and old.device == dev and old.numel() == total) else torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, (off, shape) in zip(ps, views):
                dst = slab[off:off + p.numel()]
                if p.data_ptr() != dst.data_ptr() or p.device != dev:
                    dst.copy_(p.detach().reshape(-1).float())
                p.data = dst.view(shape)
        self._slab = slab
        self._views = views

    def _ensure_packed(self):
        s = self._slab
        base = s.data_ptr()
        for p, (off, _) in zip(self._slab_params(), self._views):
            if p.data_ptr() != base + 4 * off or p.device != s.device:
                self._pack()
                return

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._pack()
        return r

    def __setstate__(self, state):
        super().__setstate__(state)
        self._pack()

    def _replicate_for_data_parallel(self):
        """nn.DataParallel with MORE than one device replicates the module and scatters the inputs along dim 0.  The trainer's
        tensors are (seq_len, batch, dim) (train_IEMOCAP.py:142-147), so that wrap (train_IEMOCAP.py:587-593) cuts every
        DIALOGUE into sequence fragments, one per GPU, each with its positional encoding restarting at 0 — the reference's own
        README reports the F1 drop (README.md:82-83).  The build refuses to reproduce that: one process per GPU, dialogues
        (dim 1) sharded, gradients all-reduced over RCCL.  (With ONE device — `nn.DataParallel(m, device_ids=[0])`, or the
        reference's plain `nn.DataParallel(m)` on a one-GPU machine — torch never replicates and the wrap is harmless:
        tests/test_hip_module_path.py.)"""
        raise RuntimeError(
            "%s: nn.DataParallel over several GPUs scatters dim 0, which for this model's (seq_len, batch, dim) tensors is the "
            "SEQUENCE axis (the reference's train_IEMOCAP.py:587-593 bug: every dialogue is cut into per-GPU fragments; its README "
            "reports the F1 drop).  Use one process per GPU with dialogues sharded along dim 1 — `python bench.py --gpus N` / "
            "engine.GanEngine(process_group=...), INTEGRATION.md section 3 — or restrict the wrap to one device "
            "(device_ids=[0])." % type(self).__name__)

    @property
    def slab(self):
        self._ensure_packed()
        return self._slab

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, x):
        self._ensure_packed()
        if self.HAS_OBJECT and x.size(-1) == self.OBJECT_IN:      # model.py:1355-1356
            x = ops.LinearFn.apply(x, self.object.weight, self.object.bias)
        if x.size(-1) != self.d_model:
            raise ValueError("%s expects last dim %d, got %d" % (type(self).__name__, self.d_model, x.size(-1)))
        if x.size(0) > MAX_LEN:
            raise ValueError("sequence length %d > %d (PositionalEncoding max_len, model.py:1179)" % (x.size(0), MAX_LEN))
        L = self.num_layers
        n_enc = 12 * L
        meta = {"E": self.d_model, "H": self.nhead, "L": L, "train": self.training,
                "p_pe": self.position_encoding.dropout.p, "p_enc": self.transformer_encoder.enc_dropout,
                "views": self._views[:n_enc]}
        enc_params = self._slab_params()[:n_enc]
        h = ops.EncoderFn.apply(x, self.position_encoding.pe, self._slab, meta, *enc_params)
        fc3 = getattr(self, "fc3", None)
        return ops.HeadFn.apply(h, 0 if self.KIND == "gen" else 1, float(self.dropout.p), self.training,
                                self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                                fc3.weight if fc3 is not None else None, fc3.bias if fc3 is not None else None)


class AcousticGenerator(_Net):      # model.py:1200-1231
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "gen", 100, 10, (512,), False


class VisualGenerator(_Net):        # model.py:1234-1263
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "gen", 512, 8, (1024,), False


class TextGenerator(_Net):          # model.py:1266-1294
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "gen", 100, 10, (512,), False


class AcousticDiscriminator(_Net):  # model.py:1297-1327
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "disc", None, 10, (64, 16), False


class VisualDiscriminator(_Net):    # model.py:1330-1364
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "disc", None, 10, (64, 16), True


class TextDiscriminator(_Net):      # model.py:1367-1397
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "disc", None, 10, (64, 16), False


# ---- extension: MELD-dimension generator / discriminator stacks (BASELINE.json configs[2]) -----------------------------
# The reference has NO GAN path for MELD (train_MELD.py trains MELDLSTMModel on text only).  These are the
# reference's generic stack (PE -> 8 post-LN encoder layers, 10 heads -> GELU head) instantiated at MELD's feature widths
# (text 600, train_MELD.py:143; audio 300, dataloader.py:93-95), with the VisualGenerator / VisualDiscriminator recipe:
# generator head d -> fc -> D_h, discriminator `object` Linear(d -> D_h) applied to the raw modality.
class MELDTextGenerator(_Net):
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "gen", 600, 10, (1024,), False


class MELDAudioGenerator(_Net):
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT = "gen", 300, 10, (512,), False


class MELDTextDiscriminator(_Net):
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT, OBJECT_IN = "disc", None, 10, (64, 16), True, 600


class MELDAudioDiscriminator(_Net):
    KIND, D_MODEL, NHEAD, FC, HAS_OBJECT, OBJECT_IN = "disc", None, 10, (64, 16), True, 300


class MaskedNLLLoss(nn.Module):
    """model.py:62-81 — NLL(sum, class weights)(pred*mask, target) / sum(weight[target]*mask)."""

    def __init__(self, weight=None):
        super().__init__()
        self.weight = weight

    def forward(self, pred, target, mask):
        mask_ = mask.reshape(-1).to(pred.dtype)
        picked = pred.gather(1, target.reshape(-1, 1)).squeeze(1) * mask_
        if self.weight is None:
            return -picked.sum() / mask.sum()
        w = self.weight.to(pred.device)[target.reshape(-1)]
        return -(w * picked).sum() / (w * mask_).sum()


class GAN_FFN(nn.Module):
    """model.py:1405-1462: log_softmax(fc(G_a(a) + G_v(v) + G_t(t)), 2).  `lstm`, `smax_fc`, dropout are
    constructed-but-unused members of the reference (model.py:1425-1430) kept for parameter-count parity."""

    def __init__(self, acoustic_generator, visual_generator, text_generator, n_classes=6, dropout=0.2):
        super().__init__()
        self.n_classes = n_classes
        self.acoustic_generator = acoustic_generator
        self.visual_generator = visual_generator
        self.text_generator = text_generator
        self.lstm = nn.LSTM(100, n_classes, bidirectional=False)
        self.gelu = nn.GELU()
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.smax_fc = nn.Linear(32 * 2, n_classes)
        self.fc = nn.Linear(100, n_classes)

    def forward(self, acoustic, visual, text):
        fusion = ops.Add3Fn.apply(self.acoustic_generator(acoustic), self.visual_generator(visual),
                                  self.text_generator(text))
        logits = ops.LinearFn.apply(fusion, self.fc.weight, self.fc.bias)
        log_prob = ops.LogSoftmaxFn.apply(logits)
        return log_prob, [], [], []


# configuration 5 (train_IEMOCAP_DialogueRNN.py:705-720): the DialogueRNN head lives in dialogue_rnn.py; re-exported so
# that `from model import GAN_FFN_DialogueRNN, BiModel, ...` keeps working for code written against the reference
from .dialogue_rnn import (BiModel, DialogueRNN, DialogueRNNCell, GAN_FFN_DialogueRNN, MatchingAttention,  # noqa: E402,F401
                           MELDLSTMModel, SimpleAttention)
