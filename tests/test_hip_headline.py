"""GPU parity at the HEADLINE size (S = 94, B = 32: BASELINE.json configs[1]) against fixtures the REFERENCE produced at
that size (tests/golden/make_golden.py modules_big / gan_steps_big: /root/reference/model.py:1200-1397 modules in eval
mode; /root/reference/train_IEMOCAP.py:200-252 train_disc / train_gen with dropout p -> 0): HIP -> reference directly,
without the oracle in between (SURVEY.md §8c "(94,32 -> checksums only)")."""
import numpy as np
import pytest
import torch

import formula as F_
from util import DIN, check_summary, golden
from test_hip_modules import build
from test_hip_engine import build_all
from test_oracle_golden import HEAD_DX_TOL, HEAD_GRAD_TOL, HEAD_DELTA_OUTLIERS, HEAD_LOSS_TOL, check_first_update

pytestmark = pytest.mark.gpu
S, B = 94, 32


@pytest.mark.parametrize("case", [
    ("AcousticGenerator", 100), ("TextGenerator", 100), ("VisualGenerator", 512), ("AcousticDiscriminator", 100),
    ("TextDiscriminator", 100), ("VisualDiscriminator", 512), ("VisualDiscriminator", 100)])
def test_module_matches_reference_at_headline_size(case):
    """eval mode, nn.Module mirror under autograd (every op one C-ABI call): output strict 1e-4 (north_star); input and
    parameter gradients within the reference's own fp32 noise at this size (HEAD_*_TOL in tests/test_oracle_golden.py: an
    fp64 oracle differs from the fixture by as much; the strict gradient comparison on the HIP forward's own ReLU pattern
    is tests/test_hip_properties.py)"""
    cls_name, din = case
    g = golden("modules_big")
    tag = "%s.%d.%dx%d" % (cls_name, din, S, B)
    net = build(cls_name).eval()
    x = torch.from_numpy(F_.formula_input(tag, S, B, din, pad_from=61)).cuda().requires_grad_(True)
    y = net(x)
    gy = torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1]) - 0.5).cuda()
    (y * gy).sum().backward()
    check_summary(g, tag + "/out", y, rtol=1e-4, atol=1e-6, what="hip", strict=True)
    check_summary(g, tag + "/dx", x.grad, **HEAD_DX_TOL, what="hip")
    sd = dict(net.named_parameters())
    n = 0
    for f in g.files:
        if f.startswith(tag + "/grad/") and (f.endswith("/full") or f.endswith("/sample")):
            k = f[len(tag) + 6:].rsplit("/", 1)[0]
            assert sd[k].grad is not None, k
            check_summary(g, tag + "/grad/" + k, sd[k].grad, **HEAD_GRAD_TOL, what="hip")
            n += 1
    assert n >= 12


@pytest.mark.parametrize("n_streams", [1, 3])
def test_engine_reproduces_reference_iteration_at_headline_size(n_streams):
    """one full 12-sub-step iteration of the step runner at (94, 32), dropout p = 0, against the reference's own
    train_disc / train_gen: the 12 losses (HEAD_LOSS_TOL: 1e-4 for sub-steps 0-4, then the measured Adam-chaos
    allowance) and every network's first-update parameter deltas"""
    from gan_ffn_amd import engine
    g = golden("gan_steps_big")
    gens, discs = build_all(zero_dropout=True)
    eng = engine.GanEngine(gens, discs, n_streams=n_streams)
    batch = {k: torch.from_numpy(F_.formula_input("ganbig." + k, S, B, DIN[k], pad_from=61)).cuda() for k in DIN}
    if n_streams == 1:
        eng._prepare(S, B)
        eng._adds = 0
        losses, seen = [], set()
        for i, (kind, who, partner) in enumerate(engine.SCHEDULE):
            (eng.train_disc if kind == "D" else eng.train_gen)(who, partner, batch, i)
            losses.append(float(eng.losses[i]))
            if (kind, who) not in seen:
                seen.add((kind, who))
                sd = dict((discs if kind == "D" else gens)[who].named_parameters())
                check_first_update(g, kind, who, lambda k: sd[k].detach().cpu().numpy(), outlier_frac=HEAD_DELTA_OUTLIERS, l2_rtol=2e-2)
    else:
        eng.iteration(batch)
        eng.synchronize()
        torch.cuda.synchronize()
        losses = eng.losses.tolist()
    err = np.abs(np.array(losses) - g["gan/losses"])
    assert (err <= np.array(HEAD_LOSS_TOL)).all(), err


@pytest.mark.parametrize("cls_name,din", [("TextDiscriminator", 100), ("AcousticGenerator", 100)])
def test_weight_gradients_over_all_32_dialogues_are_strict_on_the_hip_relu_pattern(cls_name, din):
    """at the headline size the weight gradients were pinned to the reference fixture only at rtol 1e-2 with
    5 % outliers (the fixture's own fp32 ReLU-kink noise), and the STRICT comparison — fp64 oracle on the HIP forward's own
    ReLU pattern and the same Philox masks — ran on 2 of the 32 dialogues.  Here it runs on ALL 32, train mode: every one of the
    3008 tokens contributes to the token-summed weight gradients, which are held to rtol 1e-3 of scale with NO outliers
    (so are the output at 1e-4 and dx at 2e-4).  Reference ops: /root/reference/model.py:1200-1231, 1367-1397 under
    train_IEMOCAP.py:200-252's backward."""
    from test_hip_properties import train_mode_backward_vs_oracle
    train_mode_backward_vs_oracle(cls_name, din, list(range(B)), grad_rtol=1e-3)
