"""N4: the oracle restatement of the reference's 4-layer bidirectional LSTM (oracle/lstm_oracle.py) pinned
(a) against torch.nn.LSTM itself — the third-party module the reference calls (/root/reference/model.py:527-533, 546) — in
float64, forward and every gradient, and (b) against the REFERENCE-generated MELDLSTMModel fixture (tests/golden/dialogue_rnn.npz
`meld/*`, made by tests/golden/make_golden.py from /root/reference/model.py:520-562) with the oracle LSTM swapped in for nn.LSTM."""
import numpy as np
import torch
import torch.nn.functional as Fn

import formula as F_
from oracle import lstm_oracle as LO
from test_dialogue_rnn_cpu import _meld_model, close, inputs
from util import golden


def test_oracle_lstm_equals_torch_lstm_in_float64():
    torch.manual_seed(5)
    S, B, In, H, L = 9, 3, 12, 8, 4
    lstm = torch.nn.LSTM(In, H, num_layers=L, bidirectional=True, dropout=0.3).double().eval()
    x = torch.randn(S, B, In, dtype=torch.float64, requires_grad=True)
    y, _ = lstm(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    gx = x.grad.clone()
    gref = {k: p.grad.clone() for k, p in lstm.named_parameters()}
    P = {k: p.detach().clone().requires_grad_(True) for k, p in lstm.named_parameters()}
    x2 = x.detach().clone().requires_grad_(True)
    y2 = LO.lstm_forward(x2, P, L)
    (y2 * gy).sum().backward()
    assert float((y2 - y).abs().max()) < 1e-12
    assert float((x2.grad - gx).abs().max()) < 1e-12
    for k in gref:
        assert float((P[k].grad - gref[k]).abs().max()) < 1e-11, k


def test_reference_meld_fixture_with_the_oracle_lstm():
    """the build's MELDLSTMModel mirror with its nn.LSTM replaced by the oracle LSTM reproduces the reference's numbers"""
    g = golden("dialogue_rnn")
    m = _meld_model()
    _, _, umask = inputs()
    P = {k: v for k, v in m.lstm.named_parameters()}
    Um = torch.from_numpy(F_.formula_input("meld.U", 7, 3, 600)).requires_grad_(True)
    emotions = LO.lstm_forward(Um, P, 4)
    att, a = m.matchatt.general2_all_queries(emotions, torch.from_numpy(umask))
    hidden = Fn.hardswish(emotions + Fn.hardswish(att))
    lp = Fn.log_softmax(m.smax_fc(hidden), 2)
    close(lp.detach().numpy(), g["meld/log_prob"], 5e-5, "log_prob")
    gy = torch.from_numpy(F_.formula_input("meld.grad", 7, 3, 7)) - 0.5
    (lp * gy).sum().backward()
    close(Um.grad.numpy(), g["meld/dU"], 3e-4, "dU")
    for k in ("lstm.weight_ih_l0", "lstm.weight_hh_l3_reverse", "lstm.bias_ih_l2"):
        gk = dict(m.named_parameters())[k].grad
        got = gk.numpy() if gk.numel() <= 4096 else gk.reshape(-1)[F_.sample_indices(gk.numel())].numpy()
        close(got, g["meld/grad/" + k], 5e-4, "grad " + k)
