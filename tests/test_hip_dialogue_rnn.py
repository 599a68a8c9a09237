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
