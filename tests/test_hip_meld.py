"""BASELINE.json configs[2] on the GPU: the generic generator / discriminator stack at MELD's feature widths (text 600,
audio 300; 10 heads -> head_dim 60 / 30), labelled "extension — no reference GAN path for MELD" (SURVEY.md §8d).
HIP modules and the bi-modal step runner against the oracle (which tests/test_meld_cpu.py pins to stock torch at these
widths), eval mode and train mode with shared Philox masks, up to the full B = 32 batch."""
import ctypes as C

import numpy as np
import pytest
import torch

import formula as F_
from oracle import ganffn_oracle as O
from util import MELD_DIN, MELD_DISC, MELD_GEN, NETS, _assert_close, formula_sd

pytestmark = pytest.mark.gpu


def build(cls_name, zero_dropout=False):
    from gan_ffn_amd import model
    m = getattr(model, cls_name)(100, dropout=0.2)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in formula_sd(cls_name).items()}, strict=False)
    assert missing.missing_keys == ["position_encoding.pe"] and not missing.unexpected_keys
    if zero_dropout:
        m.dropout.p = 0.0
        m.position_encoding.dropout.p = 0.0
        m.transformer_encoder.enc_dropout = 0.0
    return m.cuda()


GRAD_KEYS = ("transformer_encoder.layers.0.self_attn.in_proj_weight", "transformer_encoder.layers.7.linear1.weight",
             "transformer_encoder.layers.4.linear2.bias", "transformer_encoder.layers.2.norm1.weight",
             "transformer_encoder.layers.5.norm2.bias", "transformer_encoder.layers.3.self_attn.out_proj.weight",
             "fc1.weight", "fc2.bias")


@pytest.mark.parametrize("cls_name,din,S,B,train", [
    ("MELDTextGenerator", 600, 13, 2, False), ("MELDTextGenerator", 600, 33, 3, True),
    ("MELDAudioGenerator", 300, 13, 2, False), ("MELDAudioGenerator", 300, 33, 3, True),
    ("MELDTextDiscriminator", 600, 13, 2, False), ("MELDTextDiscriminator", 100, 33, 4, True),
    ("MELDAudioDiscriminator", 300, 21, 2, True), ("MELDTextGenerator", 600, 110, 2, False)])
def test_meld_module_matches_oracle(cls_name, din, S, B, train):
    """eval mode, and train mode with dropout ON (same (seed, offset) -> identical Philox masks in kernel and oracle)"""
    from gan_ffn_amd import _lib, ops
    kind, _, E, H, fcs, has_obj = NETS[cls_name]
    net = build(cls_name)
    net = net.train() if train else net.eval()
    seed = 99173
    ops.manual_seed(seed)
    tag = "meldhip.%s.%d.%d" % (cls_name, din, S)
    x_np = F_.formula_input(tag, S, B, din, pad_from=max(1, S - 4))
    x = torch.from_numpy(x_np).cuda().requires_grad_(True)
    y = net(x)
    gy = torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5
    (y * gy.cuda()).sum().backward()

    onet = O.OracleNet(kind, formula_sd(cls_name), H, 0.2, torch.float64)
    xo = torch.from_numpy(x_np).double().requires_grad_(True)
    xin = xo
    if has_obj and din == has_obj:
        xin = xo @ onet.P["object.weight"].T + onet.P["object.bias"]
    # the oracle takes the ReLU pattern the HIP forward actually took (see tests/test_hip_modules.py)
    enc_node = y.grad_fn.next_functions[0][0]
    T_, F_hid = S * B, int(enc_node.cfg.F)
    masks = []
    for l in range(8):
        off = int(_lib.load().ganffn_encoder_saved_hidden_offset(C.byref(enc_node.cfg), l))
        hsav = enc_node.saved[off:off + T_ * F_hid].view(S, B, F_hid)
        masks.append((hsav != 0).double().cpu())
    # (eval mode too: a hidden unit within rounding of zero may land on the other side of relu than in fp64 and would move
    # one token's gradient row; the audit below ties the HIP pattern to the oracle's own)
    h = O.encoder_stack(xin, onet.P, H, O.Rng(seed, 0, train), relu_masks=masks)
    r1 = O.Rng(seed, 1, train)
    P = onet.P
    if kind == "gen":
        t = O._drop(O.gelu(h), 0.2, O.SITE_HEAD0, r1)
        t = O.gelu(O._drop(t @ P["fc1.weight"].T + P["fc1.bias"], 0.2, O.SITE_HEAD1, r1))
        yo = O.gelu(O._drop(t @ P["fc2.weight"].T + P["fc2.bias"], 0.2, O.SITE_HEAD2, r1))
    else:
        t = O.gelu(h)
        t = O.gelu(O._drop(t @ P["fc1.weight"].T + P["fc1.bias"], 0.2, O.SITE_HEAD1, r1))
        t = O.gelu(O._drop(t @ P["fc2.weight"].T + P["fc2.bias"], 0.2, O.SITE_HEAD2, r1))
        yo = torch.sigmoid(O._drop(t @ P["fc3.weight"].T + P["fc3.bias"], 0.2, O.SITE_HEAD3, r1))
    (yo * gy.double()).sum().backward()
    _assert_close(y.detach().cpu().double().numpy(), yo.detach().numpy(), 1e-4, 1e-6, "out", 0.0, 1.0)   # north_star 1e-4
    _assert_close(x.grad.cpu().double().numpy(), xo.grad.numpy(), 2e-4, 1e-8, "dx", 0.0, 1.0)          # strict
    sd = dict(net.named_parameters())
    keys = GRAD_KEYS + (("object.weight",) if has_obj and din == has_obj else ())
    for k in keys:
        _assert_close(sd[k].grad.cpu().double().numpy(), P[k].grad.numpy(), 1e-3, 1e-8, "grad " + k, 0.0, 1.0)
    if not train:
        # kink audit: wherever the HIP relu pattern differs from the oracle's own, the pre-activation is rounding noise
        trace = []
        with torch.no_grad():
            O.encoder_stack(xin.detach(), onet.P, H, None, trace=trace)
        flips = 0
        for l in range(8):
            diff = (trace[l] > 0) != (masks[l] > 0)
            flips += int(diff.sum())
            if diff.any():
                assert float(trace[l][diff].abs().max()) < 2e-5 * max(1.0, float(trace[l].abs().max())), l
        assert flips <= 1e-4 * 8 * S * B * 2048, flips


def _build_all(zero_dropout):
    gens = {k: build(c, zero_dropout) for k, c in MELD_GEN.items()}
    discs = {k: build(c, zero_dropout) for k, c in MELD_DISC.items()}
    return gens, discs


def _batch(S, B):
    return {k: torch.from_numpy(F_.formula_input("meldgan." + k, S, B, d, pad_from=max(1, S - 6))).cuda()
            for k, d in MELD_DIN.items()}


@pytest.mark.parametrize("S,B", [(9, 2), (33, 32)])
def test_bimodal_engine_matches_oracle_iteration(S, B):
    """The 4 sub-steps of the bi-modal schedule (dropout p = 0) against the oracle's train_disc / train_gen on the same
    formula weights and batch — at (33, 32) this is BASELINE.json configs[2]'s full batch.  Losses at 1e-4; every
    network's first Adam update against the oracle's."""
    from gan_ffn_amd import engine
    gens, discs = _build_all(zero_dropout=True)
    before = {(g_, k): {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
              for g_, grp in (("G", gens), ("D", discs)) for k, m in grp.items()}
    eng = engine.GanEngine(gens, discs, n_streams=1)
    assert eng.schedule == engine.SCHEDULE_BIMODAL
    batch = _batch(S, B)
    eng.iteration(batch)
    got = eng.loss_dict()

    ogens = {k: O.OracleNet("gen", formula_sd(c), 10, 0.0, torch.float64) for k, c in MELD_GEN.items()}
    odiscs = {k: O.OracleNet("disc", formula_sd(c), 10, 0.0, torch.float64) for k, c in MELD_DISC.items()}
    opts = O.make_optimizers(ogens, odiscs)
    ob = {k: v.cpu().double() for k, v in batch.items()}
    want = O.gan_iteration(ogens, odiscs, opts, ob, None, schedule=engine.SCHEDULE_BIMODAL)
    for k, v in want.items():
        assert abs(got[k] - v) < 1e-4, (k, got[k], v)
    # first update of each network: Adam's first step is ~lr * sign(g); compare the update direction where the
    # gradient is not within rounding of zero (same policy as tests/test_oracle_golden.check_first_update)
    for (g_, k), sd0 in before.items():
        net = (gens if g_ == "G" else discs)[k]
        onet = (ogens if g_ == "G" else odiscs)[k]
        lr = 1e-4 * (1.1 if (g_ == "G" and k == "text") else 1.0) * (0.5 if g_ == "D" else 1.0)
        for n in ("transformer_encoder.layers.7.linear2.weight", "transformer_encoder.layers.0.self_attn.in_proj_weight",
                  "fc1.weight"):
            d_hip = (dict(net.named_parameters())[n].detach().cpu() - sd0[n]).double().numpy().reshape(-1)
            d_ora = (onet.P[n].detach() - torch.from_numpy(formula_sd(MELD_GEN[k] if g_ == "G" else MELD_DISC[k])[n]).double()).numpy().reshape(-1)
            # t = 1 deltas are ~ +-lr; elements whose gradient is rounding noise may flip sign (outliers), as in
            # tests/test_oracle_golden.check_first_update
            bad = np.abs(d_hip - d_ora) > 3e-2 * lr
            assert bad.mean() < 0.06, (g_, k, n, bad.mean())


def test_bimodal_engine_train_mode_full_batch_two_streams():
    """configs[2] shape (B = 32, S = 33) in TRAIN mode on 2 streams, against the oracle with the SAME Philox masks.

    Sub-step 0 = train_disc(D_text | G_acoustic): G in eval mode (no dropout), then ONE discriminator pass over the
    [real | fake] batch of 2B = 64 dialogues with the dropout offsets the engine allocates (base + 2 encoder, base + 3
    head; base + 0 / + 1 go to the eval-mode generator pass and draw nothing).  The oracle draws the masks of that
    64-dialogue layout, so the train-mode loss is checked at north_star's 1e-4, then repeats D's Adam step and checks
    sub-step 1 = train_gen(G_acoustic | D_text) (generator offsets base + 4 / + 5, the frozen D in eval mode) the same way.

    History: this test used to assert 0.2 < loss < 3.0 on every sub-step, failed on its first MELD run
    (gpurun_out/r2_round_a.log) and commit 39a10e4 widened the bound to (1e-3, 10) without a recorded cause.  Cause, from
    that commit's own diff: the first version cloned the loss tensor BEFORE eng.synchronize() — on the current stream,
    while the two side streams were still running the sub-steps — so it read loss slots that had not been written yet
    (zeros: "a > 0.2" false).  The same commit moved the clone behind synchronize(), which was the whole fix; the wider
    bound was never needed: on the fixed test the four train-mode losses at (33, 32) are 0.7043, 0.7308, 0.7004, 0.7428
    (measured, round 3).  A range is a weak check either way: it is gone, and the losses are compared with the oracle."""
    import test_hip_modules as M
    from gan_ffn_amd import engine, ops
    S, B, seed = 33, 32, 11
    gens, discs = _build_all(zero_dropout=False)
    ops.manual_seed(seed)
    eng = engine.GanEngine(gens, discs, n_streams=2)
    assert eng.schedule[:2] == [("D", "text", "acoustic"), ("G", "acoustic", "text")]
    batch = _batch(S, B)
    base = eng.rng.counter                      # the block of offsets this iteration will get
    la = eng.iteration(batch)
    eng.synchronize()                 # side streams -> current stream before reading the loss slots
    torch.cuda.synchronize()
    a = la.clone()
    b = eng.iteration(batch)
    eng.synchronize()
    torch.cuda.synchronize()
    assert set(eng.loss_dict()) == {"text_D_loss", "acoustic_G_loss", "acoustic_D_loss", "text_G_loss"}
    assert torch.isfinite(a).all() and torch.isfinite(b).all(), (a, b)
    assert not torch.allclose(a, b)             # offsets advanced: new masks (and one Adam step further)
    print("train-mode losses at (33, 32), formula weights:", [round(float(v), 4) for v in a])

    # ---- oracle, sub-step 0: train_disc(D_text | G_acoustic) with the engine's offsets and the 2B layout
    Ga = O.OracleNet("gen", formula_sd(MELD_GEN["acoustic"]), 10, 0.2, torch.float64)
    Dt = O.OracleNet("disc", formula_sd(MELD_DISC["text"]), 10, 0.2, torch.float64)
    xa, xt = batch["acoustic"].cpu().double(), batch["text"].cpu().double()
    with torch.no_grad():
        fusion = O.generator_forward(xa, Ga.P, 10, 0.2, None)                       # gen.eval(), detached
    real_in = xt @ Dt.P["object.weight"].T + Dt.P["object.bias"]                   # model.py:1355-1356 at MELD's text width
    xcat = torch.cat((real_in, fusion), dim=1)
    h = O.encoder_stack(xcat, Dt.P, 10, O.Rng(seed, base + 2, True))
    prob = M.oracle_head(Dt, "disc", h, O.Rng(seed, base + 3, True))
    want0 = (O.bce_mean(prob[:, :B], torch.ones(S, B, 1, dtype=torch.float64)) +
             O.bce_mean(prob[:, B:], torch.zeros(S, B, 1, dtype=torch.float64))) / 2.0
    assert abs(float(a[0]) - float(want0)) < 1e-4 * max(1.0, abs(float(want0))), (float(a[0]), float(want0))
    # D_text's Adam step (lr / 2, betas (0.5, 0.6): train_IEMOCAP.py:292-297,603-606), then sub-step 1
    opt = O.Adam(Dt.parameters(), 1e-4 / 2, (0.5, 0.6))
    want0.backward()
    opt.step()
    hg = O.encoder_stack(xa, Ga.P, 10, O.Rng(seed, base + 4, True))
    fus = M.oracle_head(Ga, "gen", hg, O.Rng(seed, base + 5, True))
    p1 = O.discriminator_forward(fus, Dt.P, 10, 0.2, None)                          # disc.eval()
    want1 = O.bce_mean(p1, torch.ones(S, B, 1, dtype=torch.float64))
    assert abs(float(a[1]) - float(want1)) < 1e-4 * max(1.0, abs(float(want1))), (float(a[1]), float(want1))
