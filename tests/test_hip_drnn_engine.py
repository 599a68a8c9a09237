"""Configuration 5 on the C-ABI step runner (engine.DrnnEngine, the counterpart of train_or_eval_model in
/root/reference/train_IEMOCAP_DialogueRNN.py:705-760 for GAN_FFN_DialogueRNN) against the module path under autograd —
which tests/test_hip_dialogue_rnn.py and tests/test_hip_drnn_kernel.py pin to the reference's fixtures."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIMS = dict(D_m=100, D_g=500, D_p=500, D_e=100, D_h=100, D_a=100)
W = [1.2, 0.60072, 0.38066, 0.94019, 0.67924, 0.34332]          # train_IEMOCAP_DialogueRNN.py:738


def build(seed=3, dropout_off=False):
    from gan_ffn_amd import model as M
    torch.manual_seed(seed)
    net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), n_classes=6,
                                listener_state=False, context_attention="general", dropout_rec=0.1, dropout=0.6, **DIMS)
    if dropout_off:
        for mod in net.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        for g in (net.acoustic_generator, net.visual_generator, net.text_generator):
            g.transformer_encoder.enc_dropout = 0.0
    return net.cuda()


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("S,B,streams", [(13, 4, 1), (94, 30, 3), (33, 7, 3)])
def test_engine_step_matches_module_path_autograd(S, B, streams):
    """dropout off (both paths then compute the same function): loss, log-probabilities and EVERY gradient of the engine's
    step equal the module path's autograd results; Adam's first update has the module path's sign pattern"""
    from gan_ffn_amd import data as D, engine as E, model as M
    net = build(dropout_off=True).train()
    ref = copy.deepcopy(net)
    b = D.synthetic_batch(B=B, S_max=S, seed=5, device="cuda")
    # module path
    lp = ref(b["acoustic"], b["visual"], b["text"], b["qmask"], b["umask"])[0]
    loss_ref = M.MaskedNLLLoss(torch.tensor(W, device="cuda"))(lp.transpose(0, 1).contiguous().view(-1, 6), b["label"].view(-1), b["umask"])
    loss_ref.backward()
    # engine
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    eng = E.DrnnEngine(net, n_streams=streams)
    loss, log_prob = eng.step(b, train=True)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-5 * max(1.0, abs(float(loss_ref)))
    assert rel(log_prob, lp) < 1e-4
    refp = dict(ref.named_parameters())
    # head gradients: the engine's flat gradient slab, tensor by tensor
    names = {id(p): n for n, p in net.named_parameters()}
    n_head = 0
    for i, p in enumerate(eng._hparams):
        g_ref = refp[names[id(p)]].grad
        assert g_ref is not None, names[id(p)]
        assert rel(eng._hp(i, True).view_as(p), g_ref) < 2e-3, names[id(p)]
        n_head += 1
    assert n_head == 32
    # generator gradients (flat slabs; `named` maps the module's parameter names to slab ranges)
    for k, pre in (("acoustic", "acoustic_generator."), ("visual", "visual_generator."), ("text", "text_generator.")):
        st = eng.G[k]
        for name in ("transformer_encoder.layers.0.self_attn.in_proj_weight", "transformer_encoder.layers.7.linear2.weight",
                     "transformer_encoder.layers.3.norm1.weight", "fc1.weight", "fc2.bias"):
            assert rel(st.w(name, True).view_as(refp[pre + name]), refp[pre + name].grad) < 2e-3, (k, name)
    # first Adam step (lr 1e-4, weight decay 1e-5): |delta| ~ lr where the gradient is not rounding noise, sign = -sign(g)
    for name in ("bi_model.smax_fc.weight", "bi_model.dialog_rnn_f.dialogue_cell.g_cell.weight_hh", "text_generator.fc1.weight"):
        d = (dict(net.named_parameters())[name].detach() - before[name]).cpu().numpy().reshape(-1)
        g = (refp[name].grad + 1e-5 * refp[name].detach()).cpu().numpy().reshape(-1)
        big = np.abs(g) > 1e-3 * np.abs(g).max()
        assert big.sum() > 0 and (np.sign(d[big]) == -np.sign(g[big])).mean() > 0.995, name
        assert np.abs(np.abs(d[big]) - 1e-4).max() < 2e-5, name
    # parameters the forward never touches stay put (torch.optim.Adam skips tensors without a gradient)
    assert torch.equal(dict(net.named_parameters())["fc1.weight"], before["fc1.weight"])


def test_engine_eval_step_and_train_mode_progress():
    """eval step = the module path in eval mode; a few train-mode steps (dropout on) lower the loss on a fixed batch and
    replicas with the same seed stay bit-identical"""
    from gan_ffn_amd import data as D, engine as E, model as M, ops
    b = D.synthetic_batch(B=6, S_max=21, seed=9, device="cuda")
    net = build().eval()
    eng = E.DrnnEngine(net)
    loss_e, lp_e = eng.step(b, train=False)
    lp = net(b["acoustic"], b["visual"], b["text"], b["qmask"], b["umask"])[0]
    assert rel(lp_e, lp) < 1e-4
    runs = []
    for _ in range(2):
        ops.manual_seed(77)
        net2 = build(seed=4).train()
        eng2 = E.DrnnEngine(net2, lr=1e-3)
        ls = []
        for _i in range(6):
            ls.append(float(eng2.step(b, train=True)[0]))
        torch.cuda.synchronize()
        runs.append((ls, net2.bi_model.smax_fc.weight.detach().clone(), net2.visual_generator.slab.detach().clone()))
    assert np.isfinite(runs[0][0]).all() and min(runs[0][0][3:]) < runs[0][0][0]
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
