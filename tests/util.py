"""Shared helpers for the test-suite: golden loading, formula weights, comparisons."""
import os

import numpy as np
import torch

import formula as F_

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# (kind, d_in, d_model, heads, fc dims, has_object)
NETS = {
    "AcousticGenerator": ("gen", 100, 100, 10, (512, 100), False),
    "TextGenerator": ("gen", 100, 100, 10, (512, 100), False),
    "VisualGenerator": ("gen", 512, 512, 8, (1024, 100), False),
    "AcousticDiscriminator": ("disc", 100, 100, 10, (64, 16, 1), False),
    "TextDiscriminator": ("disc", 100, 100, 10, (64, 16, 1), False),
    "VisualDiscriminator": ("disc", 100, 100, 10, (64, 16, 1), True),
    # extension: MELD-dimension stacks (BASELINE.json configs[2]); has_object = `object`'s input width
    "MELDTextGenerator": ("gen", 600, 600, 10, (1024, 100), False),
    "MELDAudioGenerator": ("gen", 300, 300, 10, (512, 100), False),
    "MELDTextDiscriminator": ("disc", 100, 100, 10, (64, 16, 1), 600),
    "MELDAudioDiscriminator": ("disc", 100, 100, 10, (64, 16, 1), 300),
}
MELD_GEN = {"acoustic": "MELDAudioGenerator", "text": "MELDTextGenerator"}
MELD_DISC = {"acoustic": "MELDAudioDiscriminator", "text": "MELDTextDiscriminator"}
MELD_DIN = {"acoustic": 300, "text": 600}
GEN = {"acoustic": "AcousticGenerator", "visual": "VisualGenerator", "text": "TextGenerator"}
DISC = {"acoustic": "AcousticDiscriminator", "visual": "VisualDiscriminator", "text": "TextDiscriminator"}
DIN = {"acoustic": 100, "visual": 512, "text": 100}
FF = 2048


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def state_shapes(cls_name, n_layers=8):
    """state_dict key -> shape for a reference-layout network (without the pe buffer)."""
    kind, din, E, H, fcs, has_obj = NETS[cls_name]
    sh = {}

    def layer(pre):
        sh[pre + "self_attn.in_proj_weight"] = (3 * E, E)
        sh[pre + "self_attn.in_proj_bias"] = (3 * E,)
        sh[pre + "self_attn.out_proj.weight"] = (E, E)
        sh[pre + "self_attn.out_proj.bias"] = (E,)
        sh[pre + "linear1.weight"] = (FF, E)
        sh[pre + "linear1.bias"] = (FF,)
        sh[pre + "linear2.weight"] = (E, FF)
        sh[pre + "linear2.bias"] = (E,)
        sh[pre + "norm1.weight"] = (E,)
        sh[pre + "norm1.bias"] = (E,)
        sh[pre + "norm2.weight"] = (E,)
        sh[pre + "norm2.bias"] = (E,)

    layer("encoder_layer.")
    for l in range(n_layers):
        layer("transformer_encoder.layers.%d." % l)
    if has_obj:
        sh["object.weight"] = (100, 512 if has_obj is True else int(has_obj))
        sh["object.bias"] = (100,)
    prev = E
    for i, d in enumerate(fcs):
        sh["fc%d.weight" % (i + 1)] = (d, prev)
        sh["fc%d.bias" % (i + 1)] = (d,)
        prev = d
    return sh


class _Shape:
    def __init__(self, s):
        self.shape = s


def formula_sd(cls_name, n_layers=8):
    return F_.formula_state_dict({k: _Shape(s) for k, s in state_shapes(cls_name, n_layers).items()})


def _assert_close(t, ref, rtol, atol, label, outlier_frac=0.02, outlier_mult=100.0):
    """max-norm relative check that tolerates relu-kink flips: an fp32 hidden unit whose
    pre-activation is ~1e-7 from 0 may land on the other side of relu in another
    implementation, perturbing ONE token's gradient row by ~1e-3 relative.  At most
    `outlier_frac` of the elements may exceed the tolerance, and none by more than
    `outlier_mult` x."""
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(t - ref)
    tol = atol + rtol * scale
    bad = err > tol
    assert bad.mean() <= outlier_frac, "%s: %.3f%% of elements exceed tol %.3e (max err %.3e, scale %.3e)" % (
        label, 100 * bad.mean(), tol, err.max(), scale)
    assert err.max() <= outlier_mult * tol, "%s: max err %.3e > %g x tol %.3e (scale %.3e)" % (
        label, err.max(), outlier_mult, tol, scale)
    return err.max() / scale


def check_summary(g, prefix, t, rtol=1e-4, atol=1e-5, what="", strict=False, outlier_frac=0.02, l2_rtol=1e-3, outlier_mult=100.0):
    """compare tensor `t` with the fixture summary stored under prefix/..."""
    t = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float64)
    kw = dict(outlier_frac=0.0, outlier_mult=1.0) if strict else dict(outlier_frac=outlier_frac, outlier_mult=outlier_mult)
    if prefix + "/full" in g.files:
        ref = g[prefix + "/full"].astype(np.float64)
        assert ref.shape == t.shape, (prefix, ref.shape, t.shape)
        return _assert_close(t, ref, rtol, atol, "%s %s" % (what, prefix), **kw)
    flat = t.reshape(-1)
    idx = np.minimum(F_.sample_indices(flat.size), flat.size - 1)
    ref = g[prefix + "/sample"].astype(np.float64)
    r = _assert_close(flat[idx], ref, rtol, atol, "%s %s (sample)" % (what, prefix), **kw)
    l2 = float(g[prefix + "/l2"])
    assert abs(np.sqrt((flat ** 2).sum()) - l2) <= l2_rtol * max(l2, 1e-30) + atol, (what, prefix, "l2")
    return r
