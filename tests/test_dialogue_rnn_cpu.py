"""N2: the DialogueRNN head against fixtures produced by the reference's own BiModel /
MatchingAttention (tests/golden/make_golden.py dialogue_rnn).  The head is device-agnostic torch code (the HIP
generators are not involved here), so this parity check runs on the CPU."""
import numpy as np
import pytest
import torch

import formula as F_
from util import golden

DIMS = dict(D_m=100, D_g=500, D_p=500, D_e=100, D_h=100, n_classes=6, D_a=100, dropout_rec=0.1, dropout=0.6)
CASES = {"general": dict(context_attention="general", listener_state=False),
         "simple_listener": dict(context_attention="simple", listener_state=True),
         "simple": dict(context_attention="simple", listener_state=False)}
LENS = [7, 4, 6]


def inputs():
    S, B = max(LENS), len(LENS)
    U = F_.formula_input("drnn.U", S, B, 100)
    umask = np.zeros((B, S), np.float32)
    for b, L in enumerate(LENS):
        umask[b, :L] = 1
        U[L:, b] = 0
    spk = (np.arange(S)[:, None] * 3 + np.arange(B)[None, :] * 2 + (np.arange(S)[:, None] // 3)) % 2
    qmask = np.stack([1 - spk, spk], -1).astype(np.float32) * umask.T[:, :, None]
    return U, qmask, umask


def close(a, ref, rtol, what):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(a - ref).max()
    assert a.shape == ref.shape and err <= rtol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("tag", list(CASES))
def test_bimodel_matches_reference_fixture(tag):
    from gan_ffn_amd import dialogue_rnn as DR
    g = golden("dialogue_rnn")
    torch.manual_seed(1)
    m = DR.BiModel(**DIMS, **CASES[tag]).eval()
    sd = F_.formula_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})          # same keys and shapes as the reference
    U, qmask, umask = inputs()
    Ut = torch.from_numpy(U).requires_grad_(True)
    lp, alpha, alpha_f, alpha_b = m(Ut, torch.from_numpy(qmask), torch.from_numpy(umask))
    close(lp.detach().numpy(), g["%s/log_prob" % tag], 2e-5, "log_prob")
    close(torch.stack(alpha, 0).detach().numpy(), g["%s/alpha" % tag], 2e-5, "alpha")
    for name, al in (("alpha_f", alpha_f), ("alpha_b", alpha_b)):
        assert len(al) == int(g["%s/%s/n" % (tag, name)])
        for t, a in enumerate(al):
            close(a.detach().numpy(), g["%s/%s/%d" % (tag, name, t)], 2e-5, "%s[%d]" % (name, t))
    gy = torch.from_numpy(F_.formula_input("drnn.grad", lp.shape[0], lp.shape[1], lp.shape[2])) - 0.5
    (lp * gy).sum().backward()
    close(Ut.grad.numpy(), g["%s/dU" % tag], 1e-4, "dU")
    n = 0
    for k, p in m.named_parameters():
        key = "%s/grad/%s" % (tag, k)
        if p.grad is None:
            assert key not in g.files, k
            continue
        got = p.grad.numpy() if p.grad.numel() <= 4096 else p.grad.reshape(-1)[F_.sample_indices(p.grad.numel())].numpy()
        close(got, g[key], 2e-4, "grad " + k)
        n += 1
    assert n >= 20


BIG_S, BIG_B = 94, 30


def big_inputs():
    """the ragged (94, 30) batch of tests/golden/make_golden.py drnn_big_inputs (configuration 5's real size)"""
    S, B = BIG_S, BIG_B
    lens = [S] + [12 + (b * 37) % 82 for b in range(1, B)]
    U = F_.formula_input("drnn.bigU", S, B, 100)
    umask = np.zeros((B, S), np.float32)
    for b, L in enumerate(lens):
        umask[b, :L] = 1
        U[L:, b] = 0
    spk = (np.arange(S)[:, None] * 3 + np.arange(B)[None, :] * 2 + (np.arange(S)[:, None] // 3)) % 2
    qmask = np.stack([1 - spk, spk], -1).astype(np.float32) * umask.T[:, :, None]
    return U, qmask, umask


def check_big(m, dev, rtol=5e-5, grtol=5e-4):
    """BiModel `m` (formula weights, eval) at (94, 30) against the reference-generated summaries of
    tests/golden/dialogue_rnn_big.npz: log-probabilities, attention maps, input gradient, every parameter gradient"""
    from util import check_summary
    g = golden("dialogue_rnn_big")
    U, qmask, umask = big_inputs()
    Ut = torch.from_numpy(U).to(dev).requires_grad_(True)
    lp, alpha, alpha_f, alpha_b = m(Ut, torch.from_numpy(qmask).to(dev), torch.from_numpy(umask).to(dev))
    check_summary(g, "big/log_prob", lp, rtol=rtol, atol=1e-6, what="log_prob", strict=True)
    check_summary(g, "big/alpha", torch.stack(alpha, 0), rtol=rtol, atol=1e-7, what="alpha", strict=True)
    check_summary(g, "big/alpha_f_last", alpha_f[-1], rtol=rtol, atol=1e-7, what="alpha_f", strict=True)
    check_summary(g, "big/alpha_b_last", alpha_b[-1], rtol=rtol, atol=1e-7, what="alpha_b", strict=True)
    gy = torch.from_numpy(F_.formula_input("drnn.biggrad", lp.shape[0], lp.shape[1], lp.shape[2])) - 0.5
    (lp * gy.to(dev)).sum().backward()
    check_summary(g, "big/dU", Ut.grad, rtol=grtol, atol=1e-7, what="dU", strict=True)
    n = 0
    for k, p in m.named_parameters():
        if p.grad is None:
            assert not any(f.startswith("big/grad/%s/" % k) for f in g.files), k
            continue
        check_summary(g, "big/grad/" + k, p.grad, rtol=grtol, atol=1e-7, what="grad " + k, strict=True, l2_rtol=2e-3)
        n += 1
    assert n >= 20


def test_bimodel_at_configuration_5_size_matches_reference_fixture():
    """(94, 30), ragged, general attention / no listener — the trained configuration at its real size"""
    from gan_ffn_amd import dialogue_rnn as DR
    torch.manual_seed(1)
    m = DR.BiModel(**DIMS, **CASES["general"]).eval()
    sd = F_.formula_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    check_big(m, "cpu")


def test_general2_attention_single_query_and_batched_agree_with_reference():
    from gan_ffn_amd import dialogue_rnn as DR
    g = golden("dialogue_rnn")
    att = DR.MatchingAttention(200, 200, att_type="general2").eval()
    sd = F_.formula_state_dict(att.state_dict())
    att.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    _, _, umask = inputs()
    M = torch.from_numpy(F_.formula_input("drnn.M", 7, 3, 200)) - 0.5
    mask = torch.from_numpy(umask)
    pool, al = att(M, M[2], mask=mask)
    close(pool.detach().numpy(), g["general2/pool"], 2e-6, "pool")
    close(al.detach().numpy(), g["general2/alpha"], 2e-6, "alpha")
    # masked positions get exactly zero weight, the rest sums to one
    assert float((al[:, 0, :] * (1 - mask)).abs().max()) == 0.0 and torch.allclose(al.sum(2), torch.ones(3, 1), atol=1e-6)
    # batched form == one query per step
    allp, alla = att.general2_all_queries(M, mask)
    close(allp[2].detach().numpy(), pool.detach().numpy(), 1e-6, "batched pool")
    close(alla[:, 2, :].detach().numpy(), al[:, 0, :].detach().numpy(), 1e-6, "batched alpha")


def test_reverse_valid_prefix():
    from gan_ffn_amd import dialogue_rnn as DR
    X = torch.arange(7 * 3 * 2, dtype=torch.float32).view(7, 3, 2)
    mask = torch.from_numpy(inputs()[2])
    R = DR.reverse_valid_prefix(X, mask)
    assert R.shape == (7, 3, 2)
    for b, L in enumerate(LENS):
        assert torch.equal(R[:L, b], X[:L, b].flip(0)) and float(R[L:, b].abs().sum()) == 0
    # trimmed to the longest valid length, like pad_sequence
    assert DR.reverse_valid_prefix(X, mask[:, :7] * torch.tensor([[1.] * 5 + [0.] * 2])).shape[0] == 5


def _meld_model(dropout=0.5):
    from gan_ffn_amd import dialogue_rnn as DR
    torch.manual_seed(2)
    m = DR.MELDLSTMModel(600, 300, 600, n_classes=7, dropout=dropout).eval()
    sd = F_.formula_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


def check_meld(m, dev):
    g = golden("dialogue_rnn")
    _, _, umask = inputs()
    Um = torch.from_numpy(F_.formula_input("meld.U", 7, 3, 600)).to(dev).requires_grad_(True)
    lp, alpha, af, ab = m(Um, None, torch.from_numpy(umask).to(dev))
    assert af == [] and ab == []
    close(lp.detach().cpu().numpy(), g["meld/log_prob"], 5e-5, "log_prob")
    close(torch.stack(alpha, 0).detach().cpu().numpy(), g["meld/alpha"], 5e-5, "alpha")
    gy = torch.from_numpy(F_.formula_input("meld.grad", 7, 3, 7)) - 0.5
    (lp * gy.to(dev)).sum().backward()
    close(Um.grad.cpu().numpy(), g["meld/dU"], 3e-4, "dU")
    P = dict(m.named_parameters())
    for k in ("lstm.weight_ih_l0", "lstm.weight_hh_l3_reverse", "lstm.bias_ih_l2", "matchatt.transform.weight", "smax_fc.weight"):
        gk = P[k].grad.cpu()
        got = gk.numpy() if gk.numel() <= 4096 else gk.reshape(-1)[F_.sample_indices(gk.numel())].numpy()
        close(got, g["meld/grad/" + k], 5e-4, "grad " + k)


def test_meld_lstm_model_matches_reference_fixture():
    """N4: MELDLSTMModel (model.py:520-562) — eval mode, formula weights, ragged mask"""
    check_meld(_meld_model(), "cpu")
