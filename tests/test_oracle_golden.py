"""Pin the oracle (oracle/ganffn_oracle.py) and the stock re-declaration
(oracle/stock_modules.py) against the golden fixtures that tests/golden/make_golden.py
produced by running the reference itself.  CPU only."""
import numpy as np
import pytest
import torch

import formula as F_
from oracle import ganffn_oracle as O
from oracle import philox, stock_modules
from util import DIN, DISC, GEN, NETS, check_summary, formula_sd, golden

torch.set_num_threads(8)


def make_net(cls_name, dtype=torch.float32):
    kind, din, E, H, fcs, has_obj = NETS[cls_name]
    return O.OracleNet(kind, formula_sd(cls_name), H, 0.2, dtype)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    r = philox.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_mask_rate_and_layout():
    m = philox.keep_mask(1001, 256, 0.1, 7, 1234, 5)
    assert m.shape == (1001, 256)
    assert abs(m.mean() - 0.9) < 3e-3
    m2 = philox.keep_mask(1001, 256, 0.1, 8, 1234, 5)
    assert (m != m2).mean() > 0.1
    a = philox.attn_keep_mask(2, 3, 7, 0.1, 16, 1, 2)
    assert a.shape == (6, 7, 7)
    full = philox.keep_mask(6 * 112, 128, 0.1, 16, 1, 2).reshape(6, 112, 128)
    assert (a == full[:, :7, :7]).all()


def test_pe_table():
    g = golden("misc")
    for d in (100, 512):
        pe = O.pe_table(d)[:, 0, :].numpy()
        assert np.abs(pe - g["pe/%d" % d]).max() == 0.0


@pytest.mark.parametrize("case", [
    ("AcousticGenerator", 100), ("TextGenerator", 100), ("VisualGenerator", 512),
    ("AcousticDiscriminator", 100), ("TextDiscriminator", 100),
    ("VisualDiscriminator", 512), ("VisualDiscriminator", 100)])
@pytest.mark.parametrize("shape", [(7, 2), (110, 3)])
def test_module_forward_backward(case, shape):
    cls_name, din = case
    S, B = shape
    g = golden("modules")
    tag = "%s.%d.%dx%d" % (cls_name, din, S, B)
    net = make_net(cls_name)
    x = torch.from_numpy(F_.formula_input(tag, S, B, din, pad_from=max(1, S - 3))).requires_grad_(True)
    y = net(x)
    gy = torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5
    (y * gy).sum().backward()
    check_summary(g, tag + "/out", y, rtol=2e-5, atol=1e-6, what="oracle")
    check_summary(g, tag + "/dx", x.grad, rtol=1e-4, atol=1e-7, what="oracle")
    n = 0
    for k in [f[len(tag) + 6:-5] for f in g.files if f.startswith(tag + "/grad/") and f.endswith("/full")] + \
             [f[len(tag) + 6:-7] for f in g.files if f.startswith(tag + "/grad/") and f.endswith("/sample")]:
        # parameter grads are sums over tokens: ONE relu-kink flip (see util._assert_close) moves every
        # element by ~5e-4 of scale, so the bound here is 1e-3; without a flip the error is ~3e-6.
        check_summary(g, tag + "/grad/" + k, net.P[k].grad, rtol=2e-3, atol=1e-7, what="oracle", outlier_frac=0.10)
        n += 1
    assert n >= 12
    assert bool(g[tag + "/template_grad_is_none"])
    assert all(net.P[k].grad is None for k in net.P if k.startswith("encoder_layer."))


@pytest.mark.parametrize("cls_name", ["AcousticGenerator", "VisualDiscriminator"])
def test_stock_redeclaration_matches_reference(cls_name):
    """oracle/stock_modules.py shares the reference's state_dict layout and arithmetic."""
    g = golden("modules")
    din = NETS[cls_name][1] if cls_name != "VisualDiscriminator" else 512
    S, B = 7, 2
    tag = "%s.%d.%dx%d" % (cls_name, din, S, B)
    m = stock_modules.StockNet(cls_name).eval()
    sd = formula_sd(cls_name)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert missing.missing_keys == ["position_encoding.pe"] and not missing.unexpected_keys
    x = torch.from_numpy(F_.formula_input(tag, S, B, din, pad_from=max(1, S - 3)))
    with torch.no_grad():
        y = m(x)
    check_summary(g, tag + "/out", y, rtol=2e-5, atol=1e-6, what="stock")


def test_bce_edge_cases():
    g = golden("misc")
    p = torch.from_numpy(g["bce/probs"])
    for tgt in (0, 1):
        y = torch.full_like(p, float(tgt))
        v = O.bce_mean(p, y)
        assert abs(float(v) - float(g["bce/target%d" % tgt])) <= 1e-5 * abs(float(g["bce/target%d" % tgt]))


def test_adam_matches_torch():
    g = golden("misc")
    for tag, kw in (("gan", dict(lr=1e-4, betas=(0.5, 0.6))), ("phase2", dict(lr=1e-4, weight_decay=0.008))):
        w = torch.from_numpy(F_.formula_tensor("adam.w", (37, 11))).clone().requires_grad_(True)
        o = O.Adam([w], **kw)
        for step in range(3):
            w.grad = torch.from_numpy(F_.formula_tensor("adam.g%d" % step, (37, 11))).clone()
            o.step()
            ref = g["adam/%s/step%d" % (tag, step)]
            assert np.abs(w.detach().numpy() - ref).max() <= 2e-7


# Loss tolerances along the 24-sub-step trajectory.  Adam's first steps are sign-like
# (delta = -lr*g/(|g|+eps)), so rounding noise on ~0 gradients flips +-lr updates and any two fp32
# (or fp64) implementations separate chaotically: measured drift of an fp64 restatement from the
# reference run is 4e-7 at sub-step 8, 2e-3 at 11, 5e-2 at 23.  Sub-steps 0-8 pin the arithmetic
# at 1e-4; later ones only pin the schedule (a wrong order / missing update moves losses by >0.1).
GAN_LOSS_TOL = [1e-4] * 9 + [1e-2] * 3 + [0.15] * 12


def run_gan_trajectory(gens, discs, opts, batch, train_disc, train_gen, schedule, first_update_hook=None):
    S_, B_ = batch["text"].shape[:2]
    valid = torch.ones(S_, B_, 1, dtype=batch["text"].dtype, device=batch["text"].device)
    fake = torch.zeros_like(valid)
    losses, seen = [], set()
    for it in range(2):
        for kind, who, partner in schedule:
            if kind == "D":
                v = train_disc(discs[who], batch[who], gens[partner], batch[partner], opts[("D", who)], valid, fake)
            else:
                v = train_gen(gens[who], batch[who], discs[partner], opts[("G", who)], valid, fake)
            losses.append(float(v))
            if (kind, who) not in seen:
                seen.add((kind, who))
                if first_update_hook:
                    first_update_hook(kind, who, (discs if kind == "D" else gens)[who])
    return losses


def check_first_update(g, kind, who, get_param, outlier_frac=0.06, l2_rtol=1e-3):
    """parameter delta after the module's first Adam step vs the reference's."""
    pre = "gan/%s_%s/delta1/" % (kind, who)
    n = 0
    for f in g.files:
        if f.startswith(pre) and (f.endswith("/full") or f.endswith("/sample")):
            k = f[len(pre):].rsplit("/", 1)[0]
            w = get_param(k)
            delta = w - F_.formula_tensor(k, tuple(w.shape))
            # t=1 deltas are ~ +-lr; elements whose gradient is rounding noise may flip sign (outliers)
            check_summary(g, pre + k, delta, rtol=3e-2, atol=1e-7, what="gan-delta1", outlier_frac=outlier_frac, l2_rtol=l2_rtol)
            n += 1
    assert n >= 12


def test_gan_two_iterations_match_reference():
    """24 sub-step losses + every module's first-update parameter delta (dropout p = 0)."""
    g = golden("gan_steps")
    S, B = 7, 2
    gens = {k: make_net(v) for k, v in GEN.items()}
    discs = {k: make_net(v) for k, v in DISC.items()}
    opts = O.make_optimizers(gens, discs)
    batch = {k: torch.from_numpy(F_.formula_input("gan." + k, S, B, DIN[k], pad_from=5)) for k in DIN}

    def hook(kind, who, net):
        check_first_update(g, kind, who, lambda k: net.P[k].detach().numpy())

    losses = run_gan_trajectory(gens, discs, opts, batch, O.train_disc, O.train_gen, O.SCHEDULE, hook)
    ref = g["gan/losses"]
    err = np.abs(np.array(losses) - ref)
    assert (err <= np.array(GAN_LOSS_TOL)).all(), err


def test_phase2_forward_and_loss():
    g = golden("misc")
    S, B = 7, 2
    gens = {k: make_net(v) for k, v in GEN.items()}
    batch = {k: torch.from_numpy(F_.formula_input("gan." + k, S, B, DIN[k], pad_from=5)) for k in DIN}
    fc_w = torch.from_numpy(F_.formula_tensor("phase2.fc.weight", (6, 100))).requires_grad_(True)
    fc_b = torch.from_numpy(F_.formula_tensor("phase2.fc.bias", (6,))).requires_grad_(True)
    lp = O.gan_ffn_forward(batch["acoustic"], batch["visual"], batch["text"], gens, fc_w, fc_b)
    assert np.abs(lp.detach().numpy() - g["phase2/log_prob"]).max() <= 2e-5
    umask = torch.from_numpy(g["phase2/umask"])
    label = torch.from_numpy(g["phase2/label"])
    w = torch.tensor(O.CLASS_WEIGHTS)
    lw = O.masked_nll(lp, label, umask, w)
    lu = O.masked_nll(lp, label, umask, None)
    assert abs(float(lw) - float(g["phase2/loss_weighted"])) <= 2e-5
    assert abs(float(lu) - float(g["phase2/loss_unweighted"])) <= 2e-5
    lw.backward()
    assert np.abs(fc_w.grad.numpy() - g["phase2/grad_fc_weight"]).max() <= 2e-6
    check_summary(g, "phase2/grad_text_fc2_weight", gens["text"].P["fc2.weight"].grad, rtol=2e-4, atol=1e-8)
    check_summary(g, "phase2/grad_visual_l0_inproj",
                  gens["visual"].P["transformer_encoder.layers.0.self_attn.in_proj_weight"].grad, rtol=2e-4, atol=1e-9)


# ---- headline size (94, 32): the oracle against fixtures the reference produced at BASELINE.json configs[1]'s size ------
# Tolerances at this size are set by the REFERENCE's own fp32 run, not by the implementation under test: an fp64 run of
# the oracle differs from the reference fixture by exactly as much as an fp32 run does (measured, round 4: output 2e-6;
# dx 1.8e-3 .. 5.5e-3 of scale; token-summed parameter gradients 1.5e-3 .. 5e-3 of scale, and single elements of a
# linear1 bias gradient by 6e-2 .. 9e-2 — one hidden unit whose pre-activation is within rounding of zero on one token,
# see util._assert_close).  3008 tokens x 2048 hidden units x 8 layers hold ~50 M ReLU decisions: a few land on the other side.
HEAD_DX_TOL = dict(rtol=1e-2, atol=1e-7, outlier_frac=0.02, l2_rtol=2e-2)
HEAD_GRAD_TOL = dict(rtol=1e-2, atol=1e-7, outlier_frac=0.05, l2_rtol=2e-2, outlier_mult=20.0)
# Losses of the 12 sub-steps: 1e-4 while the trajectory is pinned (sub-steps 0-4); from the second update of a network on,
# Adam's sign-like first steps (delta = -lr g / (|g| + eps)) amplify rounding noise on ~0 gradients — measured drift of
# an fp64 oracle from the reference's fp32 run at this size: 7e-5 at sub-step 5, 3e-5 at 7, 4.7e-4 at 9, 1.2e-3 at 11
# (fp32 oracle: 1.5e-4, 1.2e-4, 2.7e-4, 1.9e-3).
HEAD_LOSS_TOL = [1e-4] * 5 + [5e-4] * 4 + [2e-3, 1e-3, 1e-2]
HEAD_DELTA_OUTLIERS = 0.10


@pytest.mark.parametrize("case", [
    ("AcousticGenerator", 100), ("TextGenerator", 100), ("VisualGenerator", 512), ("AcousticDiscriminator", 100),
    ("TextDiscriminator", 100), ("VisualDiscriminator", 512), ("VisualDiscriminator", 100)])
def test_module_forward_backward_at_headline_size(case):
    """the seven module cases of modules_big.npz (the HIP path is compared with the same fixture in
    tests/test_hip_headline.py)"""
    cls_name, din = case
    S, B = 94, 32
    g = golden("modules_big")
    tag = "%s.%d.%dx%d" % (cls_name, din, S, B)
    net = make_net(cls_name)
    x = torch.from_numpy(F_.formula_input(tag, S, B, din, pad_from=61)).requires_grad_(True)
    y = net(x)
    gy = torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5
    (y * gy).sum().backward()
    check_summary(g, tag + "/out", y, rtol=2e-5, atol=1e-6, what="oracle")
    check_summary(g, tag + "/dx", x.grad, **HEAD_DX_TOL, what="oracle")
    n = 0
    for f in g.files:
        if f.startswith(tag + "/grad/") and (f.endswith("/full") or f.endswith("/sample")):
            k = f[len(tag) + 6:].rsplit("/", 1)[0]
            check_summary(g, tag + "/grad/" + k, net.P[k].grad, **HEAD_GRAD_TOL, what="oracle")
            n += 1
    assert n >= 12


def test_gan_iteration_matches_reference_at_headline_size():
    """the 12 sub-steps of gan_steps_big.npz (the reference's train_disc / train_gen at (94, 32), dropout p = 0): losses
    (HEAD_LOSS_TOL above) and every network's first-update deltas"""
    g = golden("gan_steps_big")
    S, B = 94, 32
    gens = {k: make_net(v) for k, v in GEN.items()}
    discs = {k: make_net(v) for k, v in DISC.items()}
    opts = O.make_optimizers(gens, discs)
    batch = {k: torch.from_numpy(F_.formula_input("ganbig." + k, S, B, DIN[k], pad_from=61)) for k in DIN}
    valid = torch.ones(S, B, 1)
    fake = torch.zeros_like(valid)
    seen = set()
    for i, (kind, who, partner) in enumerate(O.SCHEDULE):
        if kind == "D":
            v = O.train_disc(discs[who], batch[who], gens[partner], batch[partner], opts[("D", who)], valid, fake)
        else:
            v = O.train_gen(gens[who], batch[who], discs[partner], opts[("G", who)], valid, fake)
        assert abs(float(v) - float(g["gan/losses"][i])) <= HEAD_LOSS_TOL[i], (i, float(v), float(g["gan/losses"][i]))
        if (kind, who) not in seen:
            seen.add((kind, who))
            net = (discs if kind == "D" else gens)[who]
            # (the visual generator's first update is sub-step 9: its gradient comes through a discriminator that has
            # already taken two sign-like steps, so a few % more of its own +-lr updates flip than at sub-steps 0-4)
            check_first_update(g, kind, who, lambda k: net.P[k].detach().numpy(), outlier_frac=HEAD_DELTA_OUTLIERS, l2_rtol=2e-2)
