"""CPU-side checks of the drop-in boundary: libganffn.so loads, exports every symbol include/ganffn.h
declares, struct layouts agree, the slab layout matches the reference's state_dict, and the module
mirror keeps the reference's names/shapes.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

from util import NETS, state_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ganffn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ganffn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gan_ffn_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libganffn.so lacks %s" % s
        assert s in _lib.SIGNATURES, "ctypes binding lacks %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.ganffn_version() == 100


def test_struct_layouts_match_header():
    from gan_ffn_amd._lib import EncCfg, HeadCfg
    assert C.sizeof(EncCfg) == 10 * 4
    assert C.sizeof(HeadCfg) == 7 * 4


def test_argument_errors_are_reported_not_crashed():
    from gan_ffn_amd import _lib
    lib = _lib.load()
    assert lib.ganffn_layer_param_offsets(100, 2048, None) < 0
    assert b"null" in lib.ganffn_last_error()
    bad = _lib.EncCfg(200, 2, 100, 10, 2048, 8, 0.2, 0.1, 1e-5, 0)   # S > 110
    assert lib.ganffn_encoder_saved_floats(C.byref(bad)) < 0
    assert b"110" in lib.ganffn_last_error()
    bad = _lib.EncCfg(50, 2, 102, 10, 2048, 8, 0.2, 0.1, 1e-5, 0)    # E % 4
    assert lib.ganffn_encoder_workspace_floats(C.byref(bad)) < 0
    with pytest.raises(_lib.GanffnError):
        _lib.call("ganffn_encoder_fwd", C.byref(bad), None, None, None, None, None, None, None, C.c_uint64(0), None)


@pytest.mark.parametrize("E", [100, 512])
def test_layer_layout_matches_state_dict_shapes(E):
    from gan_ffn_amd import ops
    total, offs = ops.layer_layout(E)
    shapes = ops.layer_shapes(E)
    pos = 0
    for o, s in zip(offs, shapes):
        assert o == pos and o % 4 == 0
        n = 1
        for d in s:
            n *= d
        pos += (n + 3) // 4 * 4
    assert pos == total
    assert total == sum(torch.Size(s).numel() for s in shapes)  # no padding needed when E % 4 == 0


@pytest.mark.parametrize("cls_name", sorted(NETS))
def test_module_mirror_keeps_reference_state_dict(cls_name):
    from gan_ffn_amd import model
    m = getattr(model, cls_name)(100, dropout=0.2)
    sd = m.state_dict()
    want = state_shapes(cls_name)
    want["position_encoding.pe"] = (110, 1, NETS[cls_name][2])
    assert set(sd) == set(want)
    for k, s in want.items():
        assert tuple(sd[k].shape) == tuple(s), k
    # parameters are views into one slab; the unused template layer is outside it
    total, views, enc = m.slab_layout()
    n_active = sum(p.numel() for k, p in m.named_parameters() if not k.startswith("encoder_layer."))
    assert 0 <= total - n_active < 16   # 16-byte alignment padding only (fc3 bias)
    base = m.slab.data_ptr()
    for p, (off, shape) in zip(m._slab_params(), views):
        assert p.data_ptr() == base + 4 * off
    # train()/eval() reach every dropout; ctor signature (D_h, dropout=0.2)
    m.eval()
    assert not m.training and not m.position_encoding.dropout.training
    m.train()
    assert m.dropout.p == 0.2 and m.position_encoding.dropout.p == 0.2


def test_param_counts_match_survey():
    from gan_ffn_amd import model
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(model.AcousticGenerator(100)) == 4175944
    assert n(model.VisualGenerator(100)) == 28999268
    assert n(model.TextDiscriminator(100)) == 4080453
    assert n(model.VisualDiscriminator(100)) == 4131753
    ffn = model.GAN_FFN(model.AcousticGenerator(100), model.VisualGenerator(100), model.TextGenerator(100))
    assert n(ffn) == 37354744      # printed by train_IEMOCAP.py:647-649


def test_same_seed_same_init_as_stock_construction_order():
    """initial weights follow the reference's construction order under torch.manual_seed"""
    from gan_ffn_amd import model
    from oracle import stock_modules
    torch.manual_seed(3407)
    a = model.VisualDiscriminator(100)
    torch.manual_seed(3407)
    b = stock_modules.StockNet("VisualDiscriminator")
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_model_shim_binds_every_name_the_trainers_import():
    """gan_ffn_amd/shims/model.py under the module name `model` serves the import lists of
    /root/reference/train_IEMOCAP.py:18-29 and train_IEMOCAP_DialogueRNN.py:18-30"""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gan_ffn_amd", "shims", "model.py")
    spec = importlib.util.spec_from_file_location("model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from gan_ffn_amd import model as pkg
    for name in ("MaskedNLLLoss", "FocalLoss", "LSTMModel2", "AcousticGenerator", "AcousticDiscriminator", "TextGenerator",
                 "TextDiscriminator", "VisualGenerator", "VisualDiscriminator", "GAN_FFN", "GAN_FFN_DialogueRNN"):
        assert hasattr(mod, name), name
        if name not in ("FocalLoss", "LSTMModel2"):
            assert getattr(mod, name) is getattr(pkg, name)
    with pytest.raises(NotImplementedError):
        mod.LSTMModel2()
