"""N2 on the GPU: the HIP DialogueRNN recurrence (csrc/dialogue_rnn.hip; general context attention, no listener — the
configuration train_IEMOCAP_DialogueRNN.py runs) against the torch restatement of model.py:828-972 in fp64 on the CPU:
emotions, attention weights, input gradient and every parameter gradient; eval mode and train mode with the SAME Philox
dropout masks; ragged dialogues (padded steps keep the party states); more than 32 dialogues (chunked)."""
import numpy as np
import pytest
import torch

from oracle import philox

pytestmark = pytest.mark.gpu

DIMS = dict(D_m=100, D_g=500, D_p=500, D_e=100)


def make_inputs(S, B, seed, Dm=100):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(1, S // 3), S + 1, (B,), generator=g)
    lens[0] = S
    valid = (torch.arange(S).unsqueeze(1) < lens.unsqueeze(0)).float()            # (S, B)
    U = (torch.rand(S, B, Dm, generator=g) - 0.3) * valid.unsqueeze(2)
    spk = torch.randint(0, 2, (S, B), generator=g)
    qmask = torch.nn.functional.one_hot(spk, 2).float() * valid.unsqueeze(2)
    return U, qmask


def build(seed=7, dropout=0.1, dims=None):
    from gan_ffn_amd import dialogue_rnn as DR
    torch.manual_seed(seed)
    m = DR.DialogueRNN(context_attention="general", listener_state=False, dropout=dropout, **(dims or DIMS))
    with torch.no_grad():                       # livelier recurrent weights than the default init
        for p in m.parameters():
            p.mul_(1.5)
    return m


class _MaskSeq(torch.nn.Module):
    """stands in for DialogueRNNCell.dropout on the CPU: multiplies by the next prepared mask (call order per step:
    g (B,H), qs (B,P,H), e (B,He))"""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.i = masks, 0
        self.p = 0.0

    def forward(self, x):
        m = self.masks[self.i]
        self.i += 1
        return x * m


def philox_masks(S, B, H, He, p, seed, offset, direction=0):
    out = []
    kg = torch.from_numpy(philox.keep_mask(S * B, H, p, 8 + 4 * direction, seed, offset)).double().view(S, B, H) / (1 - p)
    kp = torch.from_numpy(philox.keep_mask(S * B, H, p, 9 + 4 * direction, seed, offset)).double().view(S, B, H) / (1 - p)
    ke = torch.from_numpy(philox.keep_mask(S * B, He, p, 10 + 4 * direction, seed, offset)).double().view(S, B, He) / (1 - p)
    for t in range(S):
        out += [kg[t], kp[t].unsqueeze(1).expand(-1, 2, -1), ke[t]]
    return out


def compare(m_gpu, m_cpu, U, qmask, e_tol=2e-5, g_tol=3e-4):
    Ug = U.cuda().requires_grad_(True)
    e, alpha = m_gpu(Ug, qmask.cuda())
    Uc = U.double().requires_grad_(True)
    e_ref, alpha_ref = m_cpu(Uc, qmask.double())
    S, B = U.shape[:2]
    assert e.shape == e_ref.shape and len(alpha) == len(alpha_ref) == S - 1

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-30))
    assert rel(e, e_ref) < e_tol
    for t, (a, ar) in enumerate(zip(alpha, alpha_ref)):
        assert a.shape == ar.shape and float((a.detach().cpu().double() - ar.detach()).abs().max()) < 2e-5, t
    gy = torch.rand(e_ref.shape, generator=torch.Generator().manual_seed(3)) - 0.5
    (e * gy.cuda()).sum().backward()
    (e_ref * gy.double()).sum().backward()
    assert rel(Ug.grad, Uc.grad) < g_tol
    pc = dict(m_cpu.named_parameters())
    for k, p in m_gpu.named_parameters():
        assert p.grad is not None, k
        if pc[k].grad is None:          # S = 1: no attention step, torch leaves the transform's gradient unset
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel(p.grad, pc[k].grad) < g_tol, k


@pytest.mark.parametrize("S,B", [(7, 3), (23, 5), (94, 30), (33, 40), (1, 2), (110, 4)])
def test_eval_mode_matches_torch_restatement(S, B):
    import copy
    U, qmask = make_inputs(S, B, seed=S * 100 + B)
    m_cpu = build().double().eval()
    m_gpu = copy.deepcopy(m_cpu).float().cuda().eval()
    compare(m_gpu, m_cpu, U, qmask)


@pytest.mark.parametrize("S,B", [(9, 4), (94, 30)])
def test_train_mode_matches_torch_restatement_with_the_same_philox_masks(S, B):
    import copy
    from gan_ffn_amd import ops
    U, qmask = make_inputs(S, B, seed=S + B)
    p, seed = 0.1, 20261004
    m_cpu = build(dropout=p).double().train()
    m_gpu = copy.deepcopy(m_cpu).float().cuda().train()
    m_cpu.dialogue_cell.dropout = _MaskSeq(philox_masks(S, B, 500, 100, p, seed, 0))
    ops.manual_seed(seed)                       # the call below takes rng offset 0
    compare(m_gpu, m_cpu, U, qmask)


def test_bimodel_uses_the_hip_recurrence_and_matches_its_cpu_self():
    """BiModel on the GPU (both directions through one chain of launches) == the same module on the CPU (torch ops),
    forward and every gradient, ragged batch"""
    import copy
    from gan_ffn_amd import dialogue_rnn as DR
    torch.manual_seed(2)
    m_cpu = DR.BiModel(D_m=100, D_g=500, D_p=500, D_e=100, D_h=100, n_classes=6, context_attention="general", listener_state=False,
                       dropout_rec=0.1, dropout=0.6).double().eval()
    m_gpu = copy.deepcopy(m_cpu).float().cuda().eval()
    S, B = 19, 6
    U, qmask = make_inputs(S, B, seed=77)
    umask = (qmask.sum(2) > 0).float().t().contiguous()
    Ug = U.cuda().requires_grad_(True)
    lp, alpha, af, ab = m_gpu(Ug, qmask.cuda(), umask.cuda())
    Uc = U.double().requires_grad_(True)
    lp_r, alpha_r, af_r, ab_r = m_cpu(Uc, qmask.double(), umask.double())
    assert float((lp.detach().cpu().double() - lp_r.detach()).abs().max()) < 5e-5
    for a, ar in zip(af + ab, af_r + ab_r):
        assert float((a.detach().cpu().double() - ar.detach()).abs().max()) < 2e-5
    gy = torch.rand(lp_r.shape, generator=torch.Generator().manual_seed(5)) - 0.5
    (lp * gy.cuda()).sum().backward()
    (lp_r * gy.double()).sum().backward()
    sc = float(Uc.grad.abs().max())
    assert float((Ug.grad.cpu().double() - Uc.grad).abs().max()) < 5e-4 * sc
    pc = dict(m_cpu.named_parameters())
    for k, p in m_gpu.named_parameters():
        ref = pc[k].grad
        assert float((p.grad.cpu().double() - ref).abs().max()) < 5e-4 * max(float(ref.abs().max()), 1e-30), k


@pytest.mark.parametrize("nn", [0, 1])
@pytest.mark.parametrize("M,N,K", [(30, 1500, 500), (32, 300, 100), (3, 500, 1500), (30, 500, 1500), (1, 100, 300), (17, 52, 36)])
def test_skinny_products(nn, M, N, K):
    """the recurrence's skinny MFMA products against fp64 matmul (8 weight copies per launch, like a step: 2 directions x
    2 cells x 2 products; K = 1500 with nn = 0 is the backward step on transposed weights, 12 waves per workgroup)"""
    import ctypes as C
    from gan_ffn_amd import _lib, ops
    g = torch.Generator().manual_seed(M * 7 + N + K + nn)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(8, K, N, generator=g) if nn else torch.randn(8, N, K, generator=g)
    Ad, Wd = A.cuda(), W.cuda().contiguous()
    Cd = torch.full((8, M, N), float("nan"), device="cuda")
    _lib.call("ganffn_drnn_skinny", nn, 8, ops._ptr(Ad), ops._ptr(Wd), ops._ptr(Cd), M, N, K, ops._stream())
    for i in range(8):
        ref = A.double() @ (W[i].double() if nn else W[i].double().T)
        assert float((Cd[i].cpu().double() - ref).abs().max() / ref.abs().max()) < 3e-6 * max(1.0, K ** 0.5)


class _SeededDrop(torch.nn.Module):
    """stands in for nn.Dropout on BOTH sides of the composite test: inverted dropout with a mask drawn on the CPU from
    a seeded generator (call order is the same on both sides)"""

    def __init__(self, p, seed):
        super().__init__()
        self.p, self.g = p, torch.Generator().manual_seed(seed)

    def forward(self, x):
        keep = (torch.rand(x.shape, generator=self.g) >= self.p).to(x.device, x.dtype)
        return x * keep / (1.0 - self.p)


@pytest.mark.parametrize("S,B", [(23, 5), (94, 30)])
def test_composite_train_mode_step_matches_cpu_bimodel_on_the_same_fusion_and_masks(S, B):
    """configuration 5 in TRAIN mode on the device (three HIP generators with dropout -> HIP bidirectional recurrence
    with Philox dropout -> general2 attention -> head; train_IEMOCAP_DialogueRNN.py:705-721 settings) against the
    fp64 CPU BiModel mirror fed the SAME fusion tensor and the SAME dropout masks: log-probabilities, loss, the gradient
    arriving at the fusion (what the generators back-propagate) and every BiModel parameter gradient"""
    import copy
    from gan_ffn_amd import data as D, model as M, ops
    torch.manual_seed(11)
    net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), 100, 500, 500, 100, 100,
                                100, n_classes=6, listener_state=False, context_attention="general", dropout_rec=0.1,
                                dropout=0.6).cuda().train()
    cpu = copy.deepcopy(net.bi_model).double().cpu().train()
    for bm in (net.bi_model, cpu):
        bm.dropout_rec, bm.dropout = _SeededDrop(0.75, 5), _SeededDrop(0.6, 6)
    b = D.synthetic_batch(B=B, S_max=S, seed=S + B, device="cuda")
    seed, seen = 20261004, {}

    def grab(mod, args):
        seen["add"] = ops.DeviceRng.get(args[0].device).counter      # the rng offset the recurrence is about to take
        args[0].retain_grad()
        seen["fusion"] = args[0]
    net.bi_model.register_forward_pre_hook(grab)
    ops.manual_seed(seed)
    lp = net(b["acoustic"], b["visual"], b["text"], b["qmask"], b["umask"])[0]
    w = torch.tensor([1.2, 0.60072, 0.38066, 0.94019, 0.67924, 0.34332])
    loss = M.MaskedNLLLoss(w.cuda())(lp.transpose(0, 1).contiguous().view(-1, 6), b["label"].view(-1), b["umask"])
    loss.backward()
    for g in (net.acoustic_generator, net.visual_generator, net.text_generator):
        assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in g.parameters())

    # CPU mirror: same fusion, per-direction Philox masks of the recurrence, same seeded head masks
    S_, B_ = seen["fusion"].shape[:2]
    cpu.dialog_rnn_f.dialogue_cell.dropout = _MaskSeq(philox_masks(S_, B_, 500, 100, 0.1, seed, seen["add"], 0))
    cpu.dialog_rnn_r.dialogue_cell.dropout = _MaskSeq(philox_masks(S_, B_, 500, 100, 0.1, seed, seen["add"], 1))
    Uc = seen["fusion"].detach().cpu().double().requires_grad_(True)
    lp_r = cpu(Uc, b["qmask"].cpu().double(), b["umask"].cpu().double())[0]
    loss_r = M.MaskedNLLLoss(w.double())(lp_r.transpose(0, 1).contiguous().view(-1, 6), b["label"].cpu().view(-1), b["umask"].cpu().double())
    loss_r.backward()
    assert float((lp.detach().cpu().double() - lp_r.detach()).abs().max()) < 1e-4
    assert abs(float(loss) - float(loss_r)) < 1e-5 * max(1.0, abs(float(loss_r)))

    def rel(a, ref):
        return float((a.cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    assert rel(seen["fusion"].grad, Uc.grad) < 1e-3
    pc = dict(cpu.named_parameters())
    for k, p in net.bi_model.named_parameters():
        assert rel(p.grad, pc[k].grad) < 1e-3, k


@pytest.mark.parametrize("dims", [dict(D_m=52, D_g=128, D_p=128, D_e=128), dict(D_m=100, D_g=256, D_p=256, D_e=36),
                                  dict(D_m=20, D_g=64, D_p=64, D_e=4)])
def test_other_widths_match_torch_restatement(dims):
    """the kernels are not specialised to the (100, 500, 500, 100) widths of configuration 5: emotion chain at its largest
    (D_e = 128) and smallest width, short K ranges in the skinny products"""
    import copy
    S, B = 11, 5
    U, qmask = make_inputs(S, B, seed=dims["D_e"], Dm=dims["D_m"])
    m_cpu = build(dims=dims).double().eval()
    m_gpu = copy.deepcopy(m_cpu).float().cuda().eval()
    from gan_ffn_amd import ops
    assert ops.dialogue_rnn_supported(m_gpu.dialogue_cell, U.cuda(), qmask.cuda())
    compare(m_gpu, m_cpu, U, qmask)
