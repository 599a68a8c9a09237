"""The drop-in MODULE path as a training loop: the nn.Module mirror (gan_ffn_amd.model) driven exactly the way the
reference's trainer drives its own classes — `.train()/.eval()` toggles, `fusion.detach()`, `torch.nn.BCELoss()`,
`torch.optim.Adam(m.parameters(), ...)` on the slab-view parameters, `opt.zero_grad()`, `loss.backward()`, `opt.step()`
(/root/reference/train_IEMOCAP.py:200-252 train_disc / train_gen, :292-300 optimizers and loss, :355-382 schedule) —
against the trajectory the REFERENCE produced with those functions (tests/golden/gan_steps.npz, dropout p -> 0).
Bound twice: the classes imported from the package, and the same classes imported `from model import ...` through the
committed shim (gan_ffn_amd/shims/model.py), the way train_IEMOCAP.py:18-29 would bind them."""
import importlib.util
import io
import os
import sys

import numpy as np
import pytest
import torch

import formula as F_
from util import DIN, DISC, GEN, formula_sd, golden
from test_oracle_golden import GAN_LOSS_TOL, check_first_update, run_gan_trajectory
from oracle.ganffn_oracle import SCHEDULE

pytestmark = pytest.mark.gpu
SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gan_ffn_amd", "shims", "model.py")


def bind_model(how):
    """the namespace the trainer's `from model import ...` sees"""
    if how == "package":
        from gan_ffn_amd import model
        return model
    spec = importlib.util.spec_from_file_location("model", SHIM)
    mod = importlib.util.module_from_spec(spec)
    old = sys.modules.get("model")
    sys.modules["model"] = mod
    try:
        spec.loader.exec_module(mod)
        # exactly the import statement of train_IEMOCAP.py:18-29
        from model import (MaskedNLLLoss, FocalLoss, LSTMModel2, AcousticGenerator, AcousticDiscriminator, TextGenerator,   # noqa: F401
                           TextDiscriminator, VisualGenerator, VisualDiscriminator, GAN_FFN)
        from model import GAN_FFN_DialogueRNN                                                                          # noqa: F401  (train_IEMOCAP_DialogueRNN.py:29)
    finally:
        if old is None:
            del sys.modules["model"]
        else:
            sys.modules["model"] = old
    return mod


def build_six(ns, zero_dropout=True):
    """train_IEMOCAP.py:580-585 with formula weights; dropout p -> 0 on the instances (as make_golden.py did on the
    reference's), `.cuda()` as :587-593 without the DataParallel wrap"""
    nets = {"G": {}, "D": {}}
    for grp, table in (("G", GEN), ("D", DISC)):
        for k, cls in table.items():
            m = getattr(ns, cls)(100, dropout=0.2)
            miss = m.load_state_dict({a: torch.from_numpy(b) for a, b in formula_sd(cls).items()}, strict=False)
            assert miss.missing_keys == ["position_encoding.pe"] and not miss.unexpected_keys
            if zero_dropout:
                m.dropout.p = 0.0
                m.position_encoding.dropout.p = 0.0
                m.transformer_encoder.enc_dropout = 0.0
            nets[grp][k] = m.cuda()
    return nets["G"], nets["D"]


def make_optimizers(gens, discs, lr=1e-4, b1=0.5, b2=0.6):
    """train_IEMOCAP.py:292-297 (call site :603-606): stock torch.optim.Adam over module.parameters()"""
    A = torch.optim.Adam
    opt = {}
    for k, m in gens.items():
        opt[("G", k)] = A(m.parameters(), lr=lr * (1.1 if k == "text" else 1.0), betas=(b1, b2))
    for k, m in discs.items():
        opt[("D", k)] = A(m.parameters(), lr=lr / 2, betas=(b1, b2))
    return opt


adversarial_loss = torch.nn.BCELoss()          # train_IEMOCAP.py:300


def train_disc(disc, real_disc, gen, real_gen, opt, valid, fake):
    """the steps of train_IEMOCAP.py:200-227, in its order"""
    disc.train()
    gen.eval()
    opt.zero_grad()
    real_prob = disc(real_disc)
    fusion = gen(real_gen)
    fake_prob = disc(fusion.detach())
    d_loss = (adversarial_loss(real_prob, valid) + adversarial_loss(fake_prob, fake)) / 2.0
    res = d_loss.cpu().detach().numpy()
    d_loss.backward()
    opt.step()
    return res


def train_gen(gen, real_gen, disc, opt, valid, fake):
    """the steps of train_IEMOCAP.py:230-252, in its order"""
    gen.train()
    disc.eval()
    opt.zero_grad()
    prob = disc(gen(real_gen))
    g_loss = adversarial_loss(prob, valid)
    res = g_loss.cpu().detach().numpy()
    g_loss.backward()
    opt.step()
    return res


@pytest.mark.parametrize("how", ["package", "shim"])
def test_module_path_reproduces_reference_gan_trajectory(how):
    """24 sub-steps at (7, 2): losses within GAN_LOSS_TOL of the reference's own run and every network's first-update
    parameter deltas (the same checks the engine and the oracle pass)"""
    g = golden("gan_steps")
    ns = bind_model(how)
    gens, discs = build_six(ns)
    opts = make_optimizers(gens, discs)
    batch = {k: torch.from_numpy(F_.formula_input("gan." + k, 7, 2, DIN[k], pad_from=5)).cuda() for k in DIN}
    slabs = {(grp, k): m.slab.data_ptr() for grp, d in (("G", gens), ("D", discs)) for k, m in d.items()}

    def hook(kind, who, net):
        sd = dict(net.named_parameters())
        check_first_update(g, kind, who, lambda k: sd[k].detach().cpu().numpy())
        # the frozen partner of a train_gen step accumulated weight gradients it never applies (reference behaviour,
        # cleared by its own next zero_grad); the template layer never gets one
        assert all(p.grad is None for k, p in sd.items() if k.startswith("encoder_layer."))

    losses = run_gan_trajectory(gens, discs, opts, batch, train_disc, train_gen, SCHEDULE, hook)
    err = np.abs(np.array(losses) - g["gan/losses"])
    assert (err <= np.array(GAN_LOSS_TOL)).all(), err
    # torch.optim.Adam stepped the slab VIEWS in place: the slab did not move, the views still alias it
    for (grp, k), ptr in slabs.items():
        m = (gens if grp == "G" else discs)[k]
        assert m.slab.data_ptr() == ptr
        assert m.fc1.weight.data_ptr() >= ptr and m.fc1.weight.data_ptr() < ptr + 4 * m.slab.numel()


def test_module_path_equals_engine_bit_for_bit_on_the_first_substeps():
    """module path (autograd Functions + torch Adam + torch BCELoss) against the C-ABI step runner from identical states:
    the same kernels run underneath, so the first train_disc / train_gen losses agree to fp32 rounding of the loss
    reduction (torch's BCELoss kernel sums in another order than ganffn_bce2) and the first updates agree to 1 ulp-level
    Adam differences"""
    from gan_ffn_amd import engine
    from test_hip_engine import build_all
    batch = {k: torch.from_numpy(F_.formula_input("gan." + k, 7, 2, DIN[k], pad_from=5)).cuda() for k in DIN}
    gens, discs = build_six(bind_model("package"))
    opts = make_optimizers(gens, discs)
    valid, fake = torch.ones(7, 2, 1, device="cuda"), torch.zeros(7, 2, 1, device="cuda")
    l_mod = [float(train_disc(discs["visual"], batch["visual"], gens["acoustic"], batch["acoustic"], opts[("D", "visual")], valid, fake)),
             float(train_gen(gens["acoustic"], batch["acoustic"], discs["visual"], opts[("G", "acoustic")], valid, fake))]
    g2, d2 = build_all(zero_dropout=True)
    eng = engine.GanEngine(g2, d2)
    eng._prepare(7, 2)
    eng._adds = 0
    eng.train_disc("visual", "acoustic", batch, 0)
    eng.train_gen("acoustic", "visual", batch, 1)
    l_eng = eng.losses[:2].tolist()
    assert np.abs(np.array(l_mod) - np.array(l_eng)).max() <= 2e-6, (l_mod, l_eng)
    for a, b in ((discs["visual"], d2["visual"]), (gens["acoustic"], g2["acoustic"])):
        d = (a.slab - b.slab).abs().max().item()
        assert d <= 2.5e-4, d          # a +-lr flip of a ~0-gradient element at worst (lr 1e-4 / 5e-5, two steps)
        assert ((a.slab - b.slab).abs() > 1e-6).float().mean().item() < 0.02


def test_shim_modules_pickle_and_reload_like_the_trainer_does():
    """whole-object torch.save / torch.load (train_IEMOCAP.py:438, :528-533) of a module built through the shim"""
    ns = bind_model("shim")
    m = ns.TextDiscriminator(100, dropout=0.2).cuda().eval()
    x = torch.rand(5, 2, 100, device="cuda")
    y0 = m(x)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False).eval()
    assert type(m2).__name__ == "TextDiscriminator" and torch.equal(m2(x), y0)
    with pytest.raises(NotImplementedError):
        ns.FocalLoss()


def test_literal_trainer_binding_dataparallel_wrap_and_cpu_batches():
    """the trainer's own two lines that INTEGRATION.md section 1 used to ask a maintainer to edit, taken literally on a one-GPU
    machine: every module wrapped as `nn.DataParallel(m).cuda()` (/root/reference/train_IEMOCAP.py:587-593; device_ids = [0]
    here so that the test means the same on a multi-GPU box) and the batch handed over as CPU FloatTensors
    (`real_text = Variable(textf.type(FloatTensor))`, :349-351, FloatTensor = the CPU alias of :40), labels on the device
    (:341-346).  torch's single-device DataParallel scatters the inputs to cuda:0 and calls the module: 4 sub-steps against
    the reference's own trajectory (tests/golden/gan_steps.npz)."""
    from torch import nn
    g = golden("gan_steps")
    gens, discs = build_six(bind_model("shim"))
    inner = {("G", k): m for k, m in gens.items()}
    inner.update({("D", k): m for k, m in discs.items()})
    gens = {k: nn.DataParallel(m, device_ids=[0]).cuda() for k, m in gens.items()}
    discs = {k: nn.DataParallel(m, device_ids=[0]).cuda() for k, m in discs.items()}
    opts = make_optimizers(gens, discs)                          # Adam over wrapper.parameters() = the slab views
    FloatTensor = torch.FloatTensor
    batch = {k: torch.from_numpy(F_.formula_input("gan." + k, 7, 2, DIN[k], pad_from=5)).cuda().type(FloatTensor) for k in DIN}
    assert all(not v.is_cuda for v in batch.values())
    valid = torch.ones(7, 2, 1, device="cuda")
    fake = torch.zeros(7, 2, 1, device="cuda")
    losses, seen = [], set()
    for kind, who, partner in SCHEDULE[:4]:
        if kind == "D":
            losses.append(float(train_disc(discs[who], batch[who], gens[partner], batch[partner], opts[("D", who)], valid, fake)))
        else:
            losses.append(float(train_gen(gens[who], batch[who], discs[partner], opts[("G", who)], valid, fake)))
        if (kind, who) not in seen:                              # (the visual discriminator steps twice in these four)
            seen.add((kind, who))
            sd = dict(inner[(kind, who)].named_parameters())
            check_first_update(g, kind, who, lambda k: sd[k].detach().cpu().numpy())
    err = np.abs(np.array(losses) - g["gan/losses"][:4])
    assert (err <= np.array(GAN_LOSS_TOL[:4])).all(), err
    for (grp, k), m in inner.items():                            # the wrap did not move or re-pack anything
        assert m.slab.is_cuda and m.fc1.weight.data_ptr() >= m.slab.data_ptr()


def test_dataparallel_over_several_devices_is_refused_with_the_reason():
    """nn.DataParallel with more than one device would replicate() the module and scatter dim 0 = the SEQUENCE axis
    (train_IEMOCAP.py:587-593 on (seq_len, batch, dim) tensors): the build refuses and says what to do instead"""
    m = bind_model("package").TextGenerator(100, dropout=0.2).cuda()
    with pytest.raises(RuntimeError) as e:
        m._replicate_for_data_parallel()              # what torch.nn.parallel.replicate calls on every module of the network
    msg = str(e.value)
    assert "SEQUENCE axis" in msg and "INTEGRATION.md section 3" in msg and "587-593" in msg
    if torch.cuda.device_count() > 1:
        from torch import nn
        with pytest.raises(RuntimeError):
            nn.DataParallel(m, device_ids=[0, 1])(torch.rand(7, 2, 100))
