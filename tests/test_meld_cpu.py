"""MELD-dimension extension (BASELINE.json configs[2]) on the CPU: the reference has NO GAN path for MELD (train_MELD.py
trains a text-only BiLSTM), so there is no reference output to pin.  What can be pinned — and is, here —
is that the oracle restates stock torch's nn.TransformerEncoder stack at MELD's widths (text 600 / audio 300, 10 heads:
head_dim 60 / 30) exactly as it does at IEMOCAP's, and that the HIP module mirror exposes the same state_dict."""
import numpy as np
import pytest
import torch

import formula as F_
from oracle import ganffn_oracle as O
from oracle import stock_modules as SM
from util import NETS, formula_sd


def _load_formula(stock, name):
    sd = formula_sd(name)
    missing = stock.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert missing.missing_keys == ["position_encoding.pe"] and not missing.unexpected_keys
    return sd


@pytest.mark.parametrize("name,din", [("MELDTextGenerator", 600), ("MELDAudioGenerator", 300),
                                      ("MELDTextDiscriminator", 600), ("MELDTextDiscriminator", 100),
                                      ("MELDAudioDiscriminator", 300)])
def test_oracle_matches_stock_transformer_at_meld_dims(name, din):
    kind, _, E, H, fcs, has_obj = NETS[name]
    S, B = 9, 2
    stock = SM.StockNet(name).eval()
    sd = _load_formula(stock, name)
    tag = "meld.%s.%d" % (name, din)
    x_np = F_.formula_input(tag, S, B, din, pad_from=6)
    x = torch.from_numpy(x_np).requires_grad_(True)
    y = stock(x)
    gy = torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5
    (y * gy).sum().backward()

    onet = O.OracleNet(kind, sd, H, 0.2, torch.float64)
    xo = torch.from_numpy(x_np).double().requires_grad_(True)
    yo = onet(xo)
    (yo * gy.double()).sum().backward()
    assert float((y.detach().double() - yo.detach()).abs().max()) < 2e-5
    sc = float(xo.grad.abs().max())
    assert float((x.grad.double() - xo.grad).abs().max()) < 2e-4 * sc
    named = dict(stock.named_parameters())
    for k in ("transformer_encoder.layers.0.self_attn.in_proj_weight", "transformer_encoder.layers.7.linear1.weight",
              "transformer_encoder.layers.3.norm1.weight", "fc1.weight", "fc2.bias"):
        g_ref, g_o = named[k].grad.double(), onet.P[k].grad
        assert float((g_ref - g_o).abs().max()) < 2e-3 * float(g_o.abs().max()), k
    assert all(p.grad is None for k, p in named.items() if k.startswith("encoder_layer."))


@pytest.mark.parametrize("name", ["MELDTextGenerator", "MELDAudioGenerator", "MELDTextDiscriminator",
                                  "MELDAudioDiscriminator"])
def test_meld_module_mirror_has_the_stock_state_dict(name):
    from gan_ffn_amd import model
    m = getattr(model, name)(100, dropout=0.2)
    ref = SM.StockNet(name)
    a = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert a == b
    assert sum(p.numel() for p in m.parameters()) == sum(p.numel() for p in ref.parameters())


def test_bimodal_schedule_is_the_text_acoustic_part_of_the_reference_order():
    from gan_ffn_amd import engine
    assert engine.SCHEDULE_BIMODAL == [("D", "text", "acoustic"), ("G", "acoustic", "text"),
                                       ("D", "acoustic", "text"), ("G", "text", "acoustic")]
    assert engine.SCHEDULE_BIMODAL == engine.SCHEDULE[4:8]      # train_IEMOCAP.py:363-370
