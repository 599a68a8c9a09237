"""N2 on the GPU: GAN_FFN_DialogueRNN = HIP generators + DialogueRNN head (configuration 5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIMS = dict(D_m=100, D_g=500, D_p=500, D_e=100, D_h=100, D_a=100)


def build(seed=3):
    from gan_ffn_amd import model as M
    torch.manual_seed(seed)
    net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), n_classes=6,
                                listener_state=False, context_attention="general", dropout_rec=0.1, dropout=0.6, **DIMS)
    return net.cuda()


def batch(S=13, B=4, seed=5):
    from gan_ffn_amd import data as D
    b = D.synthetic_batch(B=B, S_max=S, seed=seed, device="cuda")
    return b


def test_forward_backward_shapes_and_head_consistency_with_cpu():
    from gan_ffn_amd import dialogue_rnn as DR
    net = build().eval()
    b = batch()
    lp, alpha, alpha_f, alpha_b = net(b["acoustic"], b["visual"], b["text"], b["qmask"], b["umask"])
    S, B = b["text"].shape[:2]
    assert lp.shape == (S, B, 6) and len(alpha) == S and alpha[0].shape == (B, S) and len(alpha_f) == S - 1
    assert torch.allclose(lp.exp().sum(2), torch.ones(S, B, device="cuda"), atol=1e-5)
    # the head on the GPU == the same head on the CPU for the same fusion input (device-agnostic torch code)
    fusion = (net.acoustic_generator(b["acoustic"]) + net.visual_generator(b["visual"]) + net.text_generator(b["text"])).detach()
    head_cpu = DR.BiModel(**{k: v for k, v in DIMS.items()}, n_classes=6, context_attention="general", dropout_rec=0.1,
                          dropout=0.6).eval()
    head_cpu.load_state_dict({k: v.cpu() for k, v in net.bi_model.state_dict().items()})
    lp_cpu = head_cpu(fusion.cpu(), b["qmask"].cpu(), b["umask"].cpu())[0]
    assert float((lp.detach().cpu() - lp_cpu).abs().max()) < 2e-4


def test_one_training_step_updates_generators_and_head():
    """train_or_eval_model of train_IEMOCAP_DialogueRNN.py: MaskedNLLLoss(class weights) + Adam(lr 1e-4, l2 1e-5)"""
    from gan_ffn_amd import model as M, ops
    ops.manual_seed(11)
    net = build().train()
    b = batch()
    w = torch.tensor([1.2, 0.60072, 0.38066, 0.94019, 0.67924, 0.34332], device="cuda")
    loss_fn = M.MaskedNLLLoss(w)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-5)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt.zero_grad()
    lp = net(b["acoustic"], b["visual"], b["text"], b["qmask"], b["umask"])[0]
    loss = loss_fn(lp.transpose(0, 1).contiguous().view(-1, 6), b["label"].view(-1), b["umask"])
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    moved = {k: float((v.detach() - before[k]).abs().max()) for k, v in net.named_parameters()}
    for k in ("text_generator.transformer_encoder.layers.0.linear1.weight", "visual_generator.fc2.weight",
              "bi_model.dialog_rnn_f.dialogue_cell.g_cell.weight_hh", "bi_model.dialog_rnn_r.dialogue_cell.e_cell.weight_ih",
              "bi_model.matchatt.transform.weight", "bi_model.smax_fc.bias"):
        assert moved[k] > 0, k
    assert moved["fc1.weight"] == 0           # present on the object, unused by forward (as in the reference)


@pytest.mark.parametrize("S,B,D", [(7, 3, 200), (94, 30, 200), (128, 2, 256), (1, 1, 4), (5, 2, 50), (112, 2, 600), (33, 3, 600)])
def test_general2_attention_kernel_vs_fp64_torch(S, B, D):
    """HIP kernel (fwd + bwd) against the torch restatement of model.py:169-182,193 in fp64"""
    from gan_ffn_amd import dialogue_rnn as DR, ops
    g = torch.Generator().manual_seed(S * 1000 + D)
    M = (torch.rand(S, B, D, generator=g) - 0.5)
    X = (torch.rand(S, B, D, generator=g) - 0.5) * 0.7
    lens = torch.randint(1, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).float()
    gy = torch.rand(S, B, D, generator=g) - 0.5
    M64, X64 = M.double().requires_grad_(True), X.double().requires_grad_(True)
    a64 = DR.general2_scores(X64.transpose(0, 1), M64, mask.double())
    att64 = torch.bmm(a64, M64.transpose(0, 1)).transpose(0, 1)
    (att64 * gy.double()).sum().backward()
    Md, Xd = M.cuda().requires_grad_(True), X.cuda().requires_grad_(True)
    att, alpha = ops.General2AttnFn.apply(Xd, Md, mask.cuda())
    (att * gy.cuda()).sum().backward()

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-30))
    assert rel(alpha, a64) < 5e-6 and rel(att, att64) < 5e-6
    assert rel(Xd.grad, X64.grad) < 2e-5 and rel(Md.grad, M64.grad) < 2e-5
    assert float((alpha.cpu() * (1 - mask).unsqueeze(1)).abs().max()) == 0.0          # masked steps: exactly zero weight


@pytest.mark.parametrize("tag", ["general", "simple_listener", "simple"])
def test_bimodel_on_gpu_matches_reference_fixture(tag):
    """the same reference fixture as tests/test_dialogue_rnn_cpu.py, with the head on the GPU (HIP general2 kernel).  "general"
    and (round 5) "simple" — the scalar-score attention of model.py:117-131, run as general attention with a constant query —
    go through the HIP recurrence; listener state is the one variant left on torch ops (not the trained configuration)"""
    import test_dialogue_rnn_cpu as T
    import formula as F_
    from util import golden
    from gan_ffn_amd import dialogue_rnn as DR
    g = golden("dialogue_rnn")
    m = DR.BiModel(**T.DIMS, **T.CASES[tag]).eval()
    sd = F_.formula_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    U, qmask, umask = T.inputs()
    Ut = torch.from_numpy(U).cuda().requires_grad_(True)
    from gan_ffn_amd import ops
    assert ops.dialogue_rnn_supported(m.dialog_rnn_f.dialogue_cell, Ut, torch.from_numpy(qmask).cuda()) == (tag != "simple_listener")
    lp, alpha, alpha_f, alpha_b = m(Ut, torch.from_numpy(qmask).cuda(), torch.from_numpy(umask).cuda())
    T.close(lp.detach().cpu().numpy(), g["%s/log_prob" % tag], 5e-5, "log_prob")
    T.close(torch.stack(alpha, 0).detach().cpu().numpy(), g["%s/alpha" % tag], 5e-5, "alpha")
    gy = torch.from_numpy(F_.formula_input("drnn.grad", lp.shape[0], lp.shape[1], lp.shape[2])) - 0.5
    (lp * gy.cuda()).sum().backward()
    T.close(Ut.grad.cpu().numpy(), g["%s/dU" % tag], 2e-4, "dU")
    keys = ("matchatt.transform.weight", "matchatt.transform.bias", "linear.weight", "dialog_rnn_f.dialogue_cell.g_cell.weight_ih")
    if tag == "simple":     # the scalar-score weight gets its gradient through the constant-query column; the padded input weights theirs
        keys += ("dialog_rnn_f.dialogue_cell.attention.scalar.weight", "dialog_rnn_r.dialogue_cell.attention.scalar.weight",
                 "dialog_rnn_r.dialogue_cell.p_cell.weight_ih", "dialog_rnn_f.dialogue_cell.e_cell.weight_hh")
    for k in keys:
        p = dict(m.named_parameters())[k]
        got = p.grad.cpu().numpy() if p.grad.numel() <= 4096 else p.grad.cpu().reshape(-1)[F_.sample_indices(p.grad.numel())].numpy()
        T.close(got, g["%s/grad/%s" % (tag, k)], 5e-4, "grad " + k)


def test_hip_recurrence_at_configuration_5_size_matches_reference_fixture():
    """the HIP recurrence (csrc/dialogue_rnn.hip) + HIP general2 at (94, 30) against summaries the REFERENCE's BiModel
    produced at that size (tests/golden/dialogue_rnn_big.npz, make_golden.py dialogue_rnn_big)"""
    import test_dialogue_rnn_cpu as T
    import formula as F_
    from gan_ffn_amd import dialogue_rnn as DR, ops
    m = DR.BiModel(**T.DIMS, **T.CASES["general"]).eval()
    sd = F_.formula_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    U, qmask, _ = T.big_inputs()
    cell = m.dialog_rnn_f.dialogue_cell
    assert ops.dialogue_rnn_supported(cell, torch.from_numpy(U).cuda(), torch.from_numpy(qmask).cuda()), "HIP recurrence must be the path under test"
    T.check_big(m, "cuda", rtol=1e-4, grtol=1e-3)


def test_meld_lstm_model_on_gpu_matches_reference_fixture():
    """N4 on the device: the build's own LSTM recurrence (csrc/lstm.hip, since round 5; MIOpen's until then) + the HIP general2
    kernel (D = 600) against the REFERENCE's MELDLSTMModel fixture (/root/reference/model.py:520-562): log-probabilities,
    attention weights, dU and the sampled LSTM / attention / head gradients — in eval mode (as the fixture was made) and in train
    mode with dropout 0 (the same arithmetic through the train-mode code path)"""
    import test_dialogue_rnn_cpu as T
    T.check_meld(T._meld_model().cuda().eval(), "cuda")
    T.check_meld(T._meld_model(dropout=0.0).cuda().train(), "cuda")
