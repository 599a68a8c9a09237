"""GPU parity of the hand-written bidirectional LSTM (csrc/lstm.hip, SURVEY.md §8f row N4: the recurrence of nn.LSTM inside
MELDLSTMModel, /root/reference/model.py:520-562) against the float64 oracle (oracle/lstm_oracle.py, itself pinned to
torch.nn.LSTM and to the reference fixture by tests/test_lstm_cpu.py): one layer through the C ABI wrapper, the 4-layer stack in
eval and in train mode (inter-layer dropout with the SAME Philox masks), batches beyond the 32-dialogue tile, bit-reproducibility."""
import numpy as np
import pytest
import torch

from oracle import ganffn_oracle as O
from oracle import lstm_oracle as LO

pytestmark = pytest.mark.gpu


def rel(a, b):
    b = b.double()
    return float((a.double().cpu() - b).abs().max() / max(float(b.abs().max()), 1e-30))


def make_lstm(In, H, L, seed, dropout=0.0):
    torch.manual_seed(seed)
    return torch.nn.LSTM(In, H, num_layers=L, bidirectional=True, dropout=dropout)


@pytest.mark.parametrize("S,B,In,H", [(7, 3, 600, 300), (1, 2, 600, 300), (33, 32, 600, 300), (94, 5, 600, 300), (12, 40, 600, 300),
                                      (10, 33, 64, 20), (5, 1, 8, 4)])
def test_one_bidirectional_layer_forward_and_backward(S, B, In, H):
    """(12, 40) and (10, 33): more dialogues than the kernels' 32-dialogue tile -> chunks; (5, 1, 8, 4): the smallest legal sizes"""
    from gan_ffn_amd import ops
    lstm = make_lstm(In, H, 1, seed=S * 7 + B)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    params = [getattr(lstm, n) for n in names] + [getattr(lstm, n + "_reverse") for n in names]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(S, B, In, generator=g)
    gy = torch.randn(S, B, 2 * H, generator=g)
    # oracle, float64
    P = {k: p.detach().double().requires_grad_(True) for k, p in lstm.named_parameters()}
    xo = x.double().requires_grad_(True)
    yo = LO.lstm_forward(xo, P, 1)
    (yo * gy.double()).sum().backward()
    # HIP
    pc = [p.detach().cuda().requires_grad_(True) for p in params]
    xc = x.cuda().requires_grad_(True)
    y = ops.LstmLayerFn.apply(xc, *pc)
    (y * gy.cuda()).sum().backward()
    assert rel(y.detach(), yo.detach()) < 2e-6
    assert rel(xc.grad, xo.grad) < 2e-5
    keys = [n for n in names] + [n + "_reverse" for n in names]
    for k, t in zip(keys, pc):
        assert rel(t.grad, P[k].grad) < 3e-5, k
    # deterministic: the same call again gives the same bits (no atomics anywhere)
    xc2 = x.cuda().requires_grad_(True)
    pc2 = [p.detach().cuda().requires_grad_(True) for p in params]
    y2 = ops.LstmLayerFn.apply(xc2, *pc2)
    (y2 * gy.cuda()).sum().backward()
    assert torch.equal(y2, y) and torch.equal(xc2.grad, xc.grad)
    for a, b in zip(pc, pc2):
        assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("train", [False, True])
def test_four_layer_stack_matches_oracle_with_the_same_dropout_masks(train):
    """the MELD classifier's LSTM (D_m = 600, D_e = 300, 4 layers, dropout 0.5: train_MELD.py:143-151) on a (33, 32) batch
    (MELD's longest dialogue; the reference's batch): eval, and TRAIN mode with the inter-layer dropout drawn from the same
    Philox stream on both sides (site SITE_LSTM + layer, offsets 0, 1, 2)"""
    from gan_ffn_amd import ops
    S, B, In, H, L, p = 33, 32, 600, 300, 4, 0.5
    lstm = make_lstm(In, H, L, seed=11, dropout=p)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(S, B, In, generator=g) * 0.5
    gy = torch.randn(S, B, 2 * H, generator=g)
    seed = 424242
    P = {k: v.detach().double().requires_grad_(True) for k, v in lstm.named_parameters()}
    xo = x.double().requires_grad_(True)
    yo = LO.lstm_forward(xo, P, L, p, rng=O.Rng(seed, 0, train))
    (yo * gy.double()).sum().backward()
    m = lstm.cuda()
    m.train(train)
    ops.manual_seed(seed)
    xc = x.cuda().requires_grad_(True)
    y = ops.lstm_forward(xc, m, train)
    (y * gy.cuda()).sum().backward()
    assert rel(y.detach(), yo.detach()) < 1e-5
    assert rel(xc.grad, xo.grad) < 1e-4
    for k, v in m.named_parameters():
        assert rel(v.grad, P[k].grad) < 2e-4, k
    if train:
        # dropout really happened: exactly-zero outputs of layer 0..2 show up as a different result from eval mode
        m.eval()
        with torch.no_grad():
            assert not torch.equal(ops.lstm_forward(xc.detach(), m, False), y.detach())


def test_cuda_lstm_matches_stock_torch_lstm_on_the_cpu():
    """the same nn.LSTM object: stock torch on the CPU (what the reference runs, model.py:546) against the HIP kernels on the GPU"""
    from gan_ffn_amd import ops
    lstm = make_lstm(600, 300, 4, seed=2).eval()
    x = torch.randn(20, 6, 600, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y_cpu, _ = lstm(x)
        y_gpu = ops.lstm_forward(x.cuda(), lstm.cuda(), False)
    assert rel(y_gpu, y_cpu) < 1e-5
