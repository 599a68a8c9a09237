"""GPU parity of the nn.Module mirror (gan_ffn_amd.model, all arithmetic in libganffn.so) against
(1) the golden fixtures generated from the reference itself (eval mode), and
(2) the oracle with identical Philox dropout masks (train mode)."""
import numpy as np
import pytest
import torch

import formula as F_
from oracle import ganffn_oracle as O
from util import NETS, check_summary, formula_sd, golden

pytestmark = pytest.mark.gpu


def build(cls_name):
    from gan_ffn_amd import model
    m = getattr(model, cls_name)(100, dropout=0.2)
    sd = formula_sd(cls_name)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert missing.missing_keys == ["position_encoding.pe"] and not missing.unexpected_keys
    return m.cuda()


def hip_relu_masks(y, S, B):
    """the 0/1 ReLU(+dropout) pattern the HIP forward actually took, per layer, read from its saved hidden activations"""
    import ctypes as C
    from gan_ffn_amd import _lib
    enc_node = y.grad_fn.next_functions[0][0]
    T_, F_hid = S * B, int(enc_node.cfg.F)
    masks = []
    for l in range(8):
        off = int(_lib.load().ganffn_encoder_saved_hidden_offset(C.byref(enc_node.cfg), l))
        assert off >= 0
        hsav = enc_node.saved[off:off + T_ * F_hid].view(S, B, F_hid)
        masks.append((hsav != 0).double().cpu())      # dropped units read 0: their gradient is 0 whatever the pattern
    return masks


def oracle_head(onet, kind, h, r1):
    P = onet.P
    if kind == "gen":
        t = O._drop(O.gelu(h), 0.2, O.SITE_HEAD0, r1)
        t = O.gelu(O._drop(t @ P["fc1.weight"].T + P["fc1.bias"], 0.2, O.SITE_HEAD1, r1))
        return O.gelu(O._drop(t @ P["fc2.weight"].T + P["fc2.bias"], 0.2, O.SITE_HEAD2, r1))
    t = O.gelu(h)
    t = O.gelu(O._drop(t @ P["fc1.weight"].T + P["fc1.bias"], 0.2, O.SITE_HEAD1, r1))
    t = O.gelu(O._drop(t @ P["fc2.weight"].T + P["fc2.bias"], 0.2, O.SITE_HEAD2, r1))
    return torch.sigmoid(O._drop(t @ P["fc3.weight"].T + P["fc3.bias"], 0.2, O.SITE_HEAD3, r1))


GRAD_KEYS = ("transformer_encoder.layers.0.self_attn.in_proj_weight", "transformer_encoder.layers.7.linear1.weight",
             "transformer_encoder.layers.4.linear2.bias", "transformer_encoder.layers.2.norm1.weight",
             "transformer_encoder.layers.5.norm2.bias", "transformer_encoder.layers.3.self_attn.out_proj.weight",
             "transformer_encoder.layers.6.self_attn.in_proj_bias", "transformer_encoder.layers.1.linear1.bias",
             "transformer_encoder.layers.0.linear2.weight", "fc1.weight", "fc2.bias")


@pytest.mark.parametrize("case", [
    ("AcousticGenerator", 100), ("TextGenerator", 100), ("VisualGenerator", 512),
    ("AcousticDiscriminator", 100), ("TextDiscriminator", 100),
    ("VisualDiscriminator", 512), ("VisualDiscriminator", 100)])
@pytest.mark.parametrize("shape", [(7, 2), (110, 3)])
def test_module_matches_reference_fixture(case, shape, fixture_grads=True):
    """Eval mode.  (1) forward vs the reference's own output: strict 1e-4 (north_star).  (2) input / weight gradients
    vs the reference fixture, tolerating ReLU-kink rows (a hidden unit within rounding of zero lands on the other side
    of relu in another fp32 implementation and moves one token's gradient row).  (3) the SAME gradients vs the fp64
    oracle run on the ReLU pattern the HIP forward took: STRICT, no outliers.  (4) the kink audit that ties (2) to
    (3): every unit whose pattern differs from the oracle's own has a pre-activation within rounding of zero, and
    there are only a handful of them."""
    cls_name, din = case
    S, B = shape
    g = golden("modules")
    tag = "%s.%d.%dx%d" % (cls_name, din, S, B)
    kind, _, E, H, fcs, has_obj = NETS[cls_name]
    net = build(cls_name).eval()
    x_np = F_.formula_input(tag, S, B, din, pad_from=max(1, S - 3))
    x = torch.from_numpy(x_np).cuda().requires_grad_(True)
    y = net(x)
    gy_np = F_.formula_input("grad." + tag, S, B, y.shape[-1]) - 0.5
    gy = torch.from_numpy(gy_np).cuda()
    (y * gy).sum().backward()
    # (1) forward: the 1e-4 bound of BASELINE.json's north_star (fused features / discriminator outputs)
    check_summary(g, tag + "/out", y, rtol=1e-4, atol=1e-6, what="hip", strict=True)
    # (3) strict: fp64 oracle on the HIP forward's ReLU pattern
    sd = dict(net.named_parameters())
    masks = hip_relu_masks(y, S, B)
    onet = O.OracleNet(kind, formula_sd(cls_name), H, 0.2, torch.float64)
    xo = torch.from_numpy(x_np).double().requires_grad_(True)
    xin = xo
    if has_obj and din == 512:
        xin = xo @ onet.P["object.weight"].T + onet.P["object.bias"]
    yo = oracle_head(onet, kind, O.encoder_stack(xin, onet.P, H, None, relu_masks=masks), None)
    (yo * torch.from_numpy(gy_np).double()).sum().backward()
    from util import _assert_close
    _assert_close(y.detach().cpu().double().numpy(), yo.detach().numpy(), 1e-4, 1e-6, "out vs oracle", 0.0, 1.0)
    _assert_close(x.grad.cpu().double().numpy(), xo.grad.numpy(), 2e-4, 1e-8, "dx vs oracle(hip relu pattern)", 0.0, 1.0)
    for k in GRAD_KEYS:
        _assert_close(sd[k].grad.cpu().double().numpy(), onet.P[k].grad.numpy(), 1e-3, 1e-8,
                      "grad %s vs oracle(hip relu pattern)" % k, 0.0, 1.0)
    # (4) kink audit: where the HIP pattern differs from the oracle's own, the pre-activation is rounding noise
    trace = []
    with torch.no_grad():
        O.encoder_stack(xin.detach(), onet.P, H, None, trace=trace)
    flips = 0
    for l in range(8):
        own = trace[l] > 0
        diff = own != (masks[l] > 0)
        flips += int(diff.sum())
        if diff.any():
            assert float(trace[l][diff].abs().max()) < 2e-5 * max(1.0, float(trace[l].abs().max())), (l, float(trace[l][diff].abs().max()))
    assert flips <= 1e-4 * 8 * S * B * 2048, flips
    # (2) vs the reference fixture (kink-tolerant: the reference's own fp32 run has its own set of kink units)
    if not fixture_grads:
        return
    check_summary(g, tag + "/dx", x.grad, rtol=2e-4, atol=1e-7, what="hip", outlier_frac=0.05, l2_rtol=2e-2)
    n = 0
    for f in g.files:
        if f.startswith(tag + "/grad/") and (f.endswith("/full") or f.endswith("/sample")):
            k = f[len(tag) + 6:].rsplit("/", 1)[0]
            assert sd[k].grad is not None, k
            # see tests/test_oracle_golden.py: one relu-kink flip moves every element of a summed grad by ~1e-3 of scale (TextDiscriminator 110x3 has such a unit)
            check_summary(g, tag + "/grad/" + k, sd[k].grad, rtol=2e-3, atol=1e-7, what="hip", outlier_frac=0.10, l2_rtol=2e-2)
            n += 1
    assert n >= 12
    assert all(p.grad is None for k, p in sd.items() if k.startswith("encoder_layer."))


# (3, 32) and (5, 32): T = 96 and 160 tokens, T % 64 == 32 — the token counts of real variable-length IEMOCAP batches at B = 32 with an odd
# S, at which the wave holding rows 32..63 of the last 64-row tile of the linear2 dgrad lies wholly past the matrix; its 1-bit-pattern
# read is clamped onto the last row tile (it read one tile past the pattern, the last field of the exactly-sized saved block)
@pytest.mark.parametrize("cls_name,din,S,B", [("TextGenerator", 100, 23, 3), ("VisualGenerator", 512, 38, 2),
                                              ("AcousticDiscriminator", 100, 94, 4), ("VisualDiscriminator", 512, 17, 2),
                                              ("TextDiscriminator", 100, 3, 32), ("VisualGenerator", 512, 5, 32)])
def test_train_mode_matches_oracle_with_same_masks(cls_name, din, S, B):
    """dropout ON: same (seed, offset) -> identical Philox masks in kernel and oracle."""
    from gan_ffn_amd import ops
    kind, _, E, H, fcs, has_obj = NETS[cls_name]
    net = build(cls_name).train()
    seed = 424242
    ops.manual_seed(seed)
    tag = "train.%s" % cls_name
    x_np = F_.formula_input(tag, S, B, din, pad_from=max(1, S - 4))
    x = torch.from_numpy(x_np).cuda().requires_grad_(True)
    y = net(x)                      # consumes rng offsets 0 (encoder) and 1 (head)
    gy = (torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5)
    (y * gy.cuda()).sum().backward()

    # oracle with the same offsets: encoder sites read offset 0, head sites offset 1
    onet = O.OracleNet(kind, formula_sd(cls_name), H, 0.2, torch.float64)
    xo = torch.from_numpy(x_np).double().requires_grad_(True)
    xin = xo
    if has_obj and din == 512:
        xin = xo @ onet.P["object.weight"].T + onet.P["object.bias"]
    # The oracle takes the ReLU pattern the HIP forward pass actually took (read back from its saved hidden
    # activations): of ~6M hidden units per pass a few sit within fp32 rounding of zero, and which side they land on
    # is implementation noise that would otherwise show up as a one-token gradient difference.
    masks = hip_relu_masks(y, S, B)
    h = O.encoder_stack(xin, onet.P, H, O.Rng(seed, 0, True), relu_masks=masks)
    yo = oracle_head(onet, kind, h, O.Rng(seed, 1, True))
    (yo * gy.double()).sum().backward()

    from util import _assert_close
    _assert_close(y.detach().cpu().double().numpy(), yo.detach().numpy(), 1e-4, 1e-6, "train out", 0.0, 1.0)
    _assert_close(x.grad.cpu().double().numpy(), xo.grad.numpy(), 2e-4, 1e-8, "train dx", 0.0, 1.0)
    sd = dict(net.named_parameters())
    for k in ("transformer_encoder.layers.0.self_attn.in_proj_weight", "transformer_encoder.layers.7.linear1.weight",
              "transformer_encoder.layers.4.linear2.bias", "transformer_encoder.layers.2.norm1.weight",
              "transformer_encoder.layers.5.norm2.bias", "transformer_encoder.layers.3.self_attn.out_proj.weight",
              "fc1.weight", "fc2.bias"):
        _assert_close(sd[k].grad.cpu().double().numpy(), onet.P[k].grad.numpy(), 1e-3, 1e-8, "train grad " + k, 0.0, 1.0)


def test_state_dict_roundtrip_and_pickle(tmp_path):
    from gan_ffn_amd import model
    torch.manual_seed(3407)
    a = model.AcousticDiscriminator(100).cuda().eval()
    x = torch.rand(9, 2, 100, device="cuda")
    y0 = a(x)
    torch.save(a, tmp_path / "d.pt")                 # the reference pickles whole modules (train_IEMOCAP.py:438)
    b = torch.load(tmp_path / "d.pt", weights_only=False).eval()
    assert torch.equal(b(x), y0)
    c = model.AcousticDiscriminator(100)
    c.load_state_dict(a.state_dict())
    assert torch.equal(c.cuda().eval()(x), y0)
    # all 8 layers start identical to the template (nn.TransformerEncoder deep copies, model.py:1211)
    sd = a.state_dict()
    for l in range(8):
        assert torch.equal(sd["transformer_encoder.layers.%d.linear1.weight" % l], sd["encoder_layer.linear1.weight"])


def test_cpu_tensor_fails_loudly():
    from gan_ffn_amd import model
    from gan_ffn_amd._lib import GanffnError
    m = model.TextGenerator(100)
    with pytest.raises(GanffnError):
        m(torch.zeros(4, 2, 100))


@pytest.mark.parametrize("mode", [0, 2, 4, 6, 8, 32, 46, 64, 1 << 23, 1 << 25, 1 << 24 | 1 << 25, 1 << 28])
def test_older_launch_sequences_agree_with_fixture(mode):
    """ganffn_debug_set_ffn_mode keeps the launch sequences that newer kernels replaced selectable, and every one of them
    stays pinned to the reference fixture and to the oracle in train mode: the token-local chains around the LayerNorms of a
    d_model-100 layer as separate GEMM + LayerNorm launches (bit 1) instead of rowchain.hip; the [T x 2048] x [2048 x 100]
    products on the generic 64 x 64 tiles (bit 2) instead of gemm_n100.hip; likewise the grouped weight gradients (bit 3,
    gemm_tn100.hip), the discriminator head (bit 5, disc_head.hip) and the head of the stack (bit 6); everything at once (46);
    the 100-wide products with their last four rows on a padded seventh MFMA tile (bit 23) instead of v_mfma_f32_4x4x1; the
    linear2 dgrad taking its ReLU / dropout pattern from the saved activation (bit 25) instead of the 1-bit copy beside it, and
    the 512-wide out-proj unsplit (bit 24; these two also on the 512-wide generator); weight gradients always through the
    reduce launch (bit 28).  (Bits 0, 7 and 22 selected the fused feed-forward kernels until round 5; they are reserved.)"""
    from gan_ffn_amd import _lib
    lib = _lib.load()
    lib.ganffn_debug_set_ffn_mode(mode)
    try:
        test_module_matches_reference_fixture(("TextGenerator", 100), (110, 3))
        test_train_mode_matches_oracle_with_same_masks("AcousticDiscriminator", 100, 94, 4)
        if mode & (3 << 24):
            test_module_matches_reference_fixture(("VisualGenerator", 512), (110, 3))
            test_train_mode_matches_oracle_with_same_masks("VisualGenerator", 512, 38, 2)
    finally:
        lib.ganffn_debug_set_ffn_mode(0)


@pytest.mark.parametrize("case,d_in", [("TextGenerator", 100), ("VisualGenerator", 512), ("VisualDiscriminator", 512)])
def test_parameter_gradients_do_not_depend_on_whether_the_input_wants_a_gradient(case, d_in):
    """ganffn_encoder_bwd2 with need_dx_in = 0 (the input does not require grad: a network trained on data,
    train_IEMOCAP.py:200-252) skips the bottom layer's in-proj dgrad and the PE-dropout backward — every parameter gradient
    must keep its bits, in train mode with dropout (same Philox offsets on both runs)"""
    from gan_ffn_amd import model, ops
    torch.manual_seed(11)
    m = getattr(model, case)(100).cuda().train()
    x = torch.randn(41, 3, d_in, device="cuda")
    grads = []
    for want in (True, False):
        ops.manual_seed(5, "cuda")
        xi = x.clone().requires_grad_(want)
        m.zero_grad(set_to_none=True)
        out = m(xi)
        out.square().sum().backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        assert len(grads[-1]) >= 100
        assert (xi.grad is not None) == want
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


@pytest.mark.parametrize("case,d_in,S,B", [("TextGenerator", 100, 41, 3), ("VisualGenerator", 512, 33, 3), ("AcousticDiscriminator", 100, 94, 5)])
def test_one_bit_relu_pattern_gives_the_bits_of_the_saved_activation(case, d_in, S, B):
    """linear1's epilogue leaves [h > 0] as one bit per hidden unit beside the saved activation and the linear2 dgrad
    (dh = (dy W2) * [h > 0] / (1 - p)) reads that instead of h (24.6 MB per layer at the headline size).  Same predicate on
    the same values: every gradient keeps its BITS against the run that reads h (debug bit 25), train mode with dropout,
    ragged token counts (S * B not a multiple of 32), both kernels that carry the epilogue (K = 100 weight-resident and the
    generic tiles of the 512-wide stack)"""
    from gan_ffn_amd import _lib, model, ops
    lib = _lib.load()
    torch.manual_seed(23)
    m = getattr(model, case)(100).cuda().train()
    x = torch.randn(S, B, d_in, device="cuda")
    runs = []
    try:
        for bits in (0, 1 << 25):
            lib.ganffn_debug_set_ffn_mode(bits)
            ops.manual_seed(9, "cuda")
            xi = x.clone().requires_grad_(True)
            m.zero_grad(set_to_none=True)
            out = m(xi)
            out.square().sum().backward()
            runs.append((out.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    finally:
        lib.ganffn_debug_set_ffn_mode(0)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert len(runs[0][2]) >= 100
    for k in runs[0][2]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k
