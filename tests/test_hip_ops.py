"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes), against the oracle /
plain torch fp32 references on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ganffn_oracle as O
from oracle import philox
from util import golden
import formula as F_

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from gan_ffn_amd import _lib
    return _lib


def dev(t):
    return t.cuda().contiguous()


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


GEMM_SHAPES = [(3008, 300, 100), (3008, 2048, 100), (3008, 100, 2048), (282, 1536, 512), (14, 300, 100),
               (5, 4, 4), (65, 68, 20), (3008, 64, 100), (3008, 16, 64), (330, 100, 512), (6016, 2048, 512),
               (3025, 1100, 100), (6016, 2048, 100), (40, 1024, 100)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt(lib, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    Ad, Wd, bd = dev(A), dev(W), dev(b)
    Cd = torch.full((M, N), float("nan"), device="cuda")
    lib.call("ganffn_gemm_nt", ptr(Ad), ptr(Wd), ptr(bd), ptr(Cd), M, N, K, stream())
    assert rel_err(Cd, ref) < 2e-6 * max(1, K ** 0.5)


@pytest.mark.parametrize("M,N,K", [(3008, 100, 300), (3008, 2048, 100), (3008, 100, 2048), (282, 512, 1536),
                                   (14, 100, 300), (5, 4, 4), (65, 68, 20), (3008, 64, 16), (3008, 512, 100),
                                   (3025, 1100, 100), (6016, 2048, 100), (40, 1024, 100)])
def test_gemm_nn(lib, M, N, K):
    g = torch.Generator().manual_seed(M + N * 5 + K * 11)
    A, Bm = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    ref = A.double() @ Bm.double()
    Cd = torch.full((M, N), float("nan"), device="cuda")
    Ad, Bd = dev(A), dev(Bm)   # keep alive: a temporary would be freed (and its block reused) before the launch
    lib.call("ganffn_gemm_nn", ptr(Ad), ptr(Bd), ptr(Cd), M, N, K, stream())
    assert rel_err(Cd, ref) < 2e-6 * max(1, K ** 0.5)


def test_gemm_refuses_an_operand_of_4_gib(lib):
    """the K loops read their operands through buffer descriptors with 32-bit byte offsets (csrc/gemm.hip `fits32`): an operand
    of 4 GiB or more must be refused with a message, not wrap around (the path's matrices are 1-50 MB)"""
    from gan_ffn_amd._lib import GanffnError
    M = K = 32768                                   # M * K * 4 bytes = 4 GiB exactly
    A = torch.empty(M, K, device="cuda")            # never read
    W, b, Cd = torch.zeros(4, K, device="cuda"), torch.zeros(4, device="cuda"), torch.zeros(M, 4, device="cuda")
    with pytest.raises(GanffnError, match="4 GiB"):
        lib.call("ganffn_gemm_nt", ptr(A), ptr(W), ptr(b), ptr(Cd), M, 4, K, stream())
    del A
    torch.cuda.empty_cache()


def test_weight_resident_short_k_kernel_gives_the_generic_kernels_bits(lib):
    """K = 100, N >= 1024 runs on the persistent weight-resident kernel (gemm_wres_kernel); the same rows through the
    generic kernel (N < 1024) must give identical bits — both use the same k order inside the MFMA chain"""
    g = torch.Generator().manual_seed(5)
    M, K = 3025, 100
    A, W, b = torch.randn(M, K, generator=g), torch.randn(2048, K, generator=g), torch.randn(2048, generator=g)
    Ad, Wd, bd = dev(A), dev(W), dev(b)
    big = torch.empty(M, 2048, device="cuda")
    lib.call("ganffn_gemm_nt", ptr(Ad), ptr(Wd), ptr(bd), ptr(big), M, 2048, K, stream())
    Ws, bs = Wd[:1000].contiguous(), bd[:1000].contiguous()
    small = torch.empty(M, 1000, device="cuda")
    lib.call("ganffn_gemm_nt", ptr(Ad), ptr(Ws), ptr(bs), ptr(small), M, 1000, K, stream())
    assert torch.equal(big[:, :1000], small)
    Bm = dev(torch.randn(K, 2048, generator=g))
    big2 = torch.empty(M, 2048, device="cuda")
    lib.call("ganffn_gemm_nn", ptr(Ad), ptr(Bm), ptr(big2), M, 2048, K, stream())
    Bs = Bm[:, :1000].contiguous()
    small2 = torch.empty(M, 1000, device="cuda")
    lib.call("ganffn_gemm_nn", ptr(Ad), ptr(Bs), ptr(small2), M, 1000, K, stream())
    assert torch.equal(big2[:, :1000], small2)


@pytest.mark.parametrize("mode_bits", [0])
@pytest.mark.parametrize("T,p", [(3008, 0.1), (6016, 0.0), (21, 0.1), (1, 0.1), (333, 0.1)])
def test_ffn_linear1_fwd_epilogue(lib, T, p, mode_bits):
    """h = dropout(relu(x W1^T + b1)) (K = 100 -> 2048, the weight-resident kernel of gemm.hip): against fp64 with the Philox
    mask of the contract, a second launch bit-identical, and a row's bits independent of its position (rows permuted in
    -> rows permuted out)"""
    E, F = 100, 2048
    g = torch.Generator().manual_seed(T)
    x, w1, b1 = torch.randn(T, E, generator=g), torch.randn(F, E, generator=g) / 10, torch.randn(F, generator=g) / 10
    seed, off, add, site = 4242, 3, 11, 18
    keep = torch.from_numpy(philox.keep_mask(T, F, p, site, seed, off + add)).double() / (1 - p) if p > 0 else torch.ones(T, F).double()
    ref = torch.relu(x.double() @ w1.double().T + b1.double()) * keep
    rng = torch.tensor([seed, off], dtype=torch.int64, device="cuda")
    xd, wd, bd = dev(x), dev(w1), dev(b1)
    lib.load().ganffn_debug_set_ffn_mode(mode_bits)
    try:
        h = torch.full((T, F), float("nan"), device="cuda")
        lib.call("ganffn_ffn_linear1_fwd", ptr(xd), ptr(wd), ptr(bd), ptr(h), T, E, F, C.c_float(p), C.c_uint32(site), ptr(rng),
                 C.c_uint64(add), 1, stream())
        h2 = torch.empty_like(h)
        lib.call("ganffn_ffn_linear1_fwd", ptr(xd), ptr(wd), ptr(bd), ptr(h2), T, E, F, C.c_float(p), C.c_uint32(site), ptr(rng),
                 C.c_uint64(add), 1, stream())
        assert torch.equal(h, h2)
        # relu kinks: an element within rounding of zero may differ in sign; compare where |pre-activation| is not tiny
        pre = x.double() @ w1.double().T + b1.double()
        ok = pre.abs() > 1e-5
        assert float(((h.double().cpu() - ref).abs() * ok).max()) < 5e-6 * float(ref.abs().max())
        if p == 0.0 and T >= 64:
            perm = torch.randperm(T, generator=g)
            hp = torch.empty_like(h)
            xp = dev(x[perm])
            lib.call("ganffn_ffn_linear1_fwd", ptr(xp), ptr(wd), ptr(bd), ptr(hp), T, E, F, C.c_float(0.0), C.c_uint32(site), ptr(rng),
                     C.c_uint64(add), 1, stream())
            assert torch.equal(hp, h[perm.cuda()])
    finally:
        lib.load().ganffn_debug_set_ffn_mode(0)


@pytest.mark.parametrize("M,N,K", [(300, 100, 3008), (2048, 100, 3008), (100, 2048, 3008), (1536, 512, 282),
                                   (100, 100, 14), (4, 4, 5), (68, 20, 65), (16, 64, 3008), (2048, 512, 6016)])
def test_gemm_tn_acc(lib, M, N, K):
    g = torch.Generator().manual_seed(M * 13 + N + K)
    At, Bm = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    C0, s0 = torch.randn(M, N, generator=g), torch.randn(M, generator=g)
    ref = C0.double() + At.double().T @ Bm.double()
    refs = s0.double() + At.double().sum(0)
    Cd, sd = dev(C0.clone()), dev(s0.clone())
    Ad, Bd = dev(At), dev(Bm)
    lib.call("ganffn_gemm_tn_acc", ptr(Ad), ptr(Bd), ptr(Cd), ptr(sd), M, N, K, stream())
    assert rel_err(Cd, ref) < 3e-6 * max(1, K ** 0.5)
    assert rel_err(sd, refs) < 3e-6 * max(1, K ** 0.5)


@pytest.mark.parametrize("R,Cn,p", [(3008, 2048, 0.1), (13, 100, 0.2), (7, 1, 0.2), (330, 512, 0.5)])
def test_dropout_mask_matches_philox_contract(lib, R, Cn, p):
    seed, off, add, site = 0x1234567890ABCDEF, 5, 3, 18
    rng = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed, off], dtype=torch.int64, device="cuda")
    x = torch.ones(R, Cn, device="cuda")
    y = torch.empty_like(x)
    lib.call("ganffn_dropout", ptr(x), ptr(y), R, Cn, C.c_float(p), C.c_uint32(site), ptr(rng), C.c_uint64(add), stream())
    keep = philox.keep_mask(R, Cn, p, site, seed, off + add)
    got = y.cpu().numpy()
    assert ((got != 0) == keep).all()
    assert np.allclose(got[keep], np.float32(1.0) / (np.float32(1.0) - np.float32(p)), rtol=1e-6)


ATTN_CASES = [(7, 2, 100, 10), (110, 3, 100, 10), (94, 4, 512, 8), (33, 2, 100, 10), (1, 1, 100, 10), (64, 2, 512, 8),
              (32, 1, 100, 10), (96, 2, 100, 10), (97, 1, 512, 8), (16, 3, 100, 10), (17, 1, 100, 10), (94, 32, 100, 10),
              (80, 2, 100, 10), (49, 5, 100, 10), (94, 64, 100, 10),
              # MELD-dimension stacks (BASELINE.json configs[2]): text E = 600 (head_dim 60), audio E = 300 (head_dim 30)
              (94, 2, 600, 10), (33, 3, 600, 10), (110, 2, 600, 10), (94, 3, 300, 10), (7, 2, 300, 10), (110, 2, 300, 10),
              # head_dim 60 / 64 at S <= 48 run on the 16x16x4 kernels, longer sequences on attention.hip
              (48, 2, 600, 10), (49, 2, 600, 10), (40, 3, 512, 8), (16, 2, 512, 8), (5, 1, 600, 10)]


@pytest.mark.parametrize("S,B,E,H", ATTN_CASES)
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_fwd_bwd(lib, S, B, E, H, p):
    g = torch.Generator().manual_seed(S * 131 + B * 17 + E)
    qkv = torch.randn(S, B, 3 * E, generator=g) * 1.5
    do = torch.randn(S, B, E, generator=g)
    seed, off, add, layer = 777, 11, 4, 2
    site = O.SITE_LAYER0 + 4 * layer
    # oracle (float64 for a tight reference; same Philox mask)
    q64 = qkv.double().requires_grad_(True)
    saved = O.ENC_DROPOUT
    O.ENC_DROPOUT = p
    try:
        o_ref = O.attention(q64, B, H, layer, O.Rng(seed, off + add, train=p > 0))
    finally:
        O.ENC_DROPOUT = saved
    (o_ref * do.double()).sum().backward()
    rng = torch.tensor([seed, off], dtype=torch.int64, device="cuda")
    qd, dod = dev(qkv), dev(do)
    od = torch.full((S, B, E), float("nan"), device="cuda")
    lse = torch.full((B * H, S), float("nan"), device="cuda")
    lib.call("ganffn_attention_fwd", ptr(qd), ptr(od), ptr(lse), S, B, E, H, C.c_float(p), C.c_uint32(site), ptr(rng),
             C.c_uint64(add), stream())
    assert rel_err(od, o_ref.detach()) < 2e-5
    if E // H <= 32:   # small-head kernels keep the log-sum-exp of every (dialogue, head, query) score row
        hd = E // H
        q = qkv[..., :E].double().reshape(S, B * H, hd).transpose(0, 1)
        k = qkv[..., E:2 * E].double().reshape(S, B * H, hd).transpose(0, 1)
        lse_ref = torch.logsumexp(q @ k.transpose(1, 2) / hd ** 0.5, dim=-1)
        assert float((lse.double().cpu() - lse_ref).abs().max()) < 2e-5
    dq = torch.full((S, B, 3 * E), float("nan"), device="cuda")
    lib.call("ganffn_attention_bwd", ptr(qd), ptr(od), ptr(lse), ptr(dod), ptr(dq), S, B, E, H, C.c_float(p),
             C.c_uint32(site), ptr(rng), C.c_uint64(add), stream())
    assert rel_err(dq, q64.grad) < 5e-5
    # the pair that hands the dropout keep bits from the forward to the backward (what the encoder stack runs): the
    # same bits, so output, log-sum-exp and gradient are IDENTICAL to the Philox-recomputing pair above
    nk = int(lib.load().ganffn_attention_keep_words(B, H))
    keep = torch.zeros(nk, dtype=torch.int32, device="cuda")
    od2, lse2 = torch.full_like(od, float("nan")), torch.full_like(lse, float("nan"))
    lib.call("ganffn_attention_fwd_keep", ptr(qd), ptr(od2), ptr(lse2), ptr(keep), S, B, E, H, C.c_float(p), C.c_uint32(site),
             ptr(rng), C.c_uint64(add), stream())
    dq2 = torch.full_like(dq, float("nan"))
    lib.call("ganffn_attention_bwd_keep", ptr(qd), ptr(od2), ptr(lse2), ptr(dod), ptr(keep), ptr(dq2), S, B, E, H, C.c_float(p),
             C.c_uint32(site), ptr(rng), C.c_uint64(add), stream())
    assert torch.equal(od2, od) and torch.equal(dq2, dq)
    if E // H <= 32:
        assert torch.equal(lse2, lse)
        if p > 0 and B * H <= 384:
            assert int((keep != 0).sum()) > 0        # the forward did store its keep words (small launches only: see attn16_use_keep)


@pytest.mark.parametrize("T,E", [(3008, 100), (3008, 512), (14, 100), (5, 512), (331, 100)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_add_dropout_layernorm(lib, T, E, p):
    g = torch.Generator().manual_seed(T + E)
    x, y = torch.randn(T, E, generator=g) * 2 + 0.3, torch.randn(T, E, generator=g)
    w, b = 1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    dout = torch.randn(T, E, generator=g)
    seed, off, add, site = 99, 3, 7, 21
    keep = torch.from_numpy(philox.keep_mask(T, E, p, site, seed, off + add)).double() / (1 - p)
    y64 = y.double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    out_ref = O.layer_norm(x64 + y64 * keep, w64, b64)
    (out_ref * dout.double()).sum().backward()
    rng = torch.tensor([seed, off], dtype=torch.int64, device="cuda")
    out = torch.empty(T, E, device="cuda")
    xhat = torch.empty(T, E, device="cuda")
    rstd = torch.empty(T, device="cuda")
    wd, bd, xd, yd, doutd = dev(w), dev(b), dev(x), dev(y), dev(dout)
    lib.call("ganffn_add_dropout_layernorm_fwd", ptr(xd), ptr(yd), ptr(wd), ptr(bd), ptr(out), ptr(xhat), ptr(rstd),
             T, E, C.c_float(1e-5), C.c_float(p), C.c_uint32(site), ptr(rng), C.c_uint64(add), stream())
    assert rel_err(out, out_ref.detach()) < 5e-6
    dz = torch.empty(T, E, device="cuda")
    dy = torch.empty(T, E, device="cuda")
    gw = torch.zeros(E, device="cuda")
    gb = torch.zeros(E, device="cuda")
    lib.call("ganffn_add_dropout_layernorm_bwd", ptr(doutd), ptr(xhat), ptr(rstd), ptr(wd), ptr(dz), ptr(dy), ptr(gw),
             ptr(gb), T, E, C.c_float(p), C.c_uint32(site), ptr(rng), C.c_uint64(add), stream())
    assert rel_err(dz, x64.grad) < 2e-5
    assert rel_err(dy, y64.grad) < 2e-5
    assert rel_err(gw, w64.grad) < 2e-5
    assert rel_err(gb, b64.grad) < 2e-5


def test_pe_table(lib):
    g = golden("misc")
    for d in (100, 512):
        pe = torch.empty(110, d, device="cuda")
        lib.call("ganffn_pe_table", ptr(pe), 110, d, stream())
        # device sinf/cosf/expf vs torch CPU: argument rounding at |a| ~ 100 gives ~1e-5 absolute
        assert np.abs(pe.cpu().numpy() - g["pe/%d" % d]).max() < 3e-5


def test_bce_edge_cases_and_backward(lib):
    g = golden("misc")
    p = torch.from_numpy(g["bce/probs"]).cuda()
    n = p.numel()
    for tgt in (0, 1):
        loss = torch.zeros(1, device="cuda")
        lib.call("ganffn_bce_fwd", ptr(p), C.c_float(tgt), n, C.c_float(1.0), ptr(loss), 0, stream())
        ref = float(g["bce/target%d" % tgt])
        assert abs(float(loss) - ref) <= 1e-5 * abs(ref)
        # accumulate + scale: (a + b)/2 form used by the D loss
        lib.call("ganffn_bce_fwd", ptr(p), C.c_float(tgt), n, C.c_float(0.5), ptr(loss), 0, stream())
        lib.call("ganffn_bce_fwd", ptr(p), C.c_float(tgt), n, C.c_float(0.5), ptr(loss), 1, stream())
        assert abs(float(loss) - ref) <= 1e-5 * abs(ref)
    pp = torch.tensor([0.3, 0.9, 0.5, 1e-3, 0.999], requires_grad=True)
    for tgt in (0.0, 1.0):
        l = torch.nn.BCELoss()(pp, torch.full_like(pp, tgt))
        (gr,) = torch.autograd.grad(l, pp)
        d = torch.empty(5, device="cuda")
        ppd = pp.detach().cuda()
        lib.call("ganffn_bce_bwd", ptr(ppd), C.c_float(tgt), 5, C.c_float(1.0), ptr(d), stream())
        assert rel_err(d, gr) < 1e-5


def test_adam_matches_reference_fixture(lib):
    g = golden("misc")
    for tag, kw in (("gan", dict(lr=1e-4, b1=0.5, b2=0.6, wd=0.0)), ("phase2", dict(lr=1e-4, b1=0.9, b2=0.999, wd=0.008))):
        w = torch.from_numpy(F_.formula_tensor("adam.w", (37, 11))).cuda().contiguous()
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        for s in range(3):
            gr = torch.from_numpy(F_.formula_tensor("adam.g%d" % s, (37, 11))).cuda().contiguous()
            lib.call("ganffn_adam_step", ptr(w), ptr(gr), ptr(m), ptr(v), ptr(step), C.c_int64(w.numel()),
                     C.c_float(kw["lr"]), C.c_float(kw["b1"]), C.c_float(kw["b2"]), C.c_float(1e-8), C.c_float(kw["wd"]),
                     C.c_float(1.0), stream())
            ref = g["adam/%s/step%d" % (tag, s)]
            assert np.abs(w.cpu().numpy() - ref).max() <= 3e-7
        assert int(step) == 3


def test_logsoftmax_nll_matches_reference_fixture(lib):
    g = golden("misc")
    lp_ref = torch.from_numpy(g["phase2/log_prob"])          # (S, B, 6)
    S, B, Cn = lp_ref.shape
    logits = (lp_ref + 0.37).cuda().contiguous()             # log_softmax is shift-invariant
    labels = torch.from_numpy(g["phase2/label"]).cuda().contiguous()
    umask = torch.from_numpy(g["phase2/umask"]).cuda().contiguous()
    w = torch.tensor(O.CLASS_WEIGHTS, device="cuda")
    for cw, key in ((w, "phase2/loss_weighted"), (None, "phase2/loss_unweighted")):
        lp = torch.empty_like(logits)
        loss = torch.zeros(1, device="cuda")
        dl = torch.empty_like(logits)
        ws2 = torch.zeros(2, device="cuda")
        lib.call("ganffn_logsoftmax_nll", ptr(logits), ptr(labels), ptr(umask), ptr(cw), ptr(lp), ptr(loss), ptr(dl), ptr(ws2),
                 S, B, Cn, stream())
        assert np.abs(lp.cpu().numpy() - g["phase2/log_prob"]).max() < 2e-6
        assert abs(float(loss) - float(g[key])) < 2e-6
        x = logits.cpu().double().requires_grad_(True)
        l = O.masked_nll(torch.log_softmax(x, 2), labels.cpu(), umask.cpu().double(), None if cw is None else cw.cpu().double())
        l.backward()
        assert rel_err(dl, x.grad) < 1e-5


@pytest.mark.parametrize("T,K,kmajor,cap", [(3008, 2048, 0, 16), (6016, 2048, 1, 16), (6016, 2048, 0, 8), (3008, 2048, 1, 5),
                                            (21, 256, 0, 16), (70, 2048, 1, 1), (1, 320, 1, 16), (2112, 2048, 0, 16)])
@pytest.mark.parametrize("kw", [0, 1, 2, 2 | 8])
def test_gemm_n100_slabs_sum_to_the_product(lib, T, K, kmajor, cap, kw):
    """[T x K] x [K x 100] on the 112-wide 16x16x4 kernel (csrc/gemm_n100.hip): the sum of its K-chunk slabs against fp64
    torch, both weight layouts (rows of K = linear2's W2; K-major = linear1's W1 in the dgrad), bias on chunk 0; ragged T,
    a single chunk, chunk counts that do not divide K / 32; kw: the launch heuristic's choice (0), four waves per
    workgroup (1), eight = two waves per K tile whose halves are added through LDS in fixed order (2); features 96..99
    on v_mfma_f32_4x4x1 (the eight-wave default) or on a padded seventh 16-wide tile (2 | 8: bit 23)"""
    lib.load().ganffn_debug_set_ffn_mode(kw << 20)
    try:
        _n100_case(lib, T, K, kmajor, cap)
    finally:
        lib.load().ganffn_debug_set_ffn_mode(0)


def _n100_case(lib, T, K, kmajor, cap):
    g = torch.Generator().manual_seed(T * 7 + K + kmajor)
    A = torch.randn(T, K, generator=g)
    W = (torch.randn(K, 100, generator=g) if kmajor else torch.randn(100, K, generator=g)) / (K ** 0.5)
    b = torch.randn(100, generator=g)
    ref = A.double() @ (W.double() if kmajor else W.double().T) + b.double()
    slabs = torch.full((cap, T, 100), float("nan"), device="cuda")
    n = C.c_int(0)
    Ad, Wd, bd = dev(A), dev(W), dev(b)
    lib.call("ganffn_gemm_n100", ptr(Ad), ptr(Wd), kmajor, ptr(bd), ptr(slabs), C.c_int64(T * 100), T, K, cap, C.byref(n), stream())
    assert 1 <= n.value <= cap
    y = slabs[:n.value].sum(0)
    assert bool(torch.isfinite(y).all())
    assert rel_err(y, ref) < 2e-6
    if n.value < cap:
        assert bool(torch.isnan(slabs[n.value:]).all())          # nothing written beyond the slabs it reports
    # deterministic: a second launch gives the same bits
    slabs2 = torch.empty_like(slabs)
    lib.call("ganffn_gemm_n100", ptr(Ad), ptr(Wd), kmajor, ptr(bd), ptr(slabs2), C.c_int64(T * 100), T, K, cap, C.byref(n), stream())
    assert torch.equal(slabs2[:n.value], slabs[:n.value])


@pytest.mark.parametrize("mode_bits", [0, 8, 2 << 16, 5 << 16, 8 << 16, 9 << 16, 1 << 23, 1 << 23 | 3 << 16])
@pytest.mark.parametrize("K", [6016, 3008, 333, 40])
def test_gemm_tn_grouped_d100_group(lib, K, mode_bits):
    """the weight-gradient group of a d_model-100 encoder pass (every problem 100-wide on one side; two layers' worth, some
    without a bias gradient, one with a strided gradient) on the 112-wide kernel (csrc/gemm_tn100.hip: default, and with
    forced token-chunk counts — 9 is clamped to the kernel's 8; rows 96..99 of the 100-wide dimension on v_mfma_f32_4x4x1,
    or with bit 23 on a padded seventh tile) and on the generic 64 x 64 tiles (bit 3): gradients and
    bias gradients against fp64, accumulation into non-zero slabs, ragged token counts (K % 32 != 0), bit-reproducible;
    and the in-kernel slab sum (the last-arriving workgroup of a tile: bit 4, opt-in — measured slower) against the
    separate reduce launch: the SAME bits"""
    probs = [(100, 2048), (2048, 100), (100, 100), (300, 100)] * 2
    g = torch.Generator().manual_seed(K)
    At = [dev(torch.randn(K, m, generator=g)) for (m, n) in probs]
    Bm = [dev(torch.randn(K, n, generator=g)) for (m, n) in probs]
    C0 = [torch.randn(m, n, generator=g) for (m, n) in probs]
    S0 = [torch.randn(m, generator=g) for (m, n) in probs]
    n = len(probs)
    nws = int(lib.load().ganffn_gemm_tn_grouped_workspace_floats())
    ws = torch.full((nws,), float("nan"), device="cuda")
    arr = lambda ts: (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts])
    Ms, Ns, Ks = (C.c_int * n)(*[p[0] for p in probs]), (C.c_int * n)(*[p[1] for p in probs]), (C.c_int * n)(*[K] * n)

    def run():
        Cd = [dev(c.clone()) for c in C0]
        Sd = [dev(s_.clone()) if i != 5 else None for i, s_ in enumerate(S0)]         # problem 5: no bias gradient wanted
        lib.call("ganffn_gemm_tn_grouped", n, arr(At), arr(Bm), arr(Cd), arr(Sd), Ms, Ns, Ks, ptr(ws), nws, stream())
        return Cd, Sd
    lib.load().ganffn_debug_set_ffn_mode(mode_bits)
    try:
        Cd, Sd = run()
        Cd2, Sd2 = run()
        lib.load().ganffn_debug_set_ffn_mode(mode_bits | 16)          # partial slabs added in-kernel by the last-arriving workgroup
        Cd3, Sd3 = run()
    finally:
        lib.load().ganffn_debug_set_ffn_mode(0)
    for i in range(n):
        assert torch.equal(Cd[i], Cd3[i]) and (Sd[i] is None or torch.equal(Sd[i], Sd3[i])), i
    for i, (m, nn) in enumerate(probs):
        ref = C0[i].double() + At[i].double().cpu().T @ Bm[i].double().cpu()
        assert rel_err(Cd[i], ref) < 3e-6 * max(1, K ** 0.5), (i, probs[i])
        if Sd[i] is not None:
            assert rel_err(Sd[i], S0[i].double() + At[i].double().cpu().sum(0)) < 3e-6 * max(1, K ** 0.5), (i, probs[i])
        assert torch.equal(Cd[i], Cd2[i]) and (Sd[i] is None or torch.equal(Sd[i], Sd2[i]))


@pytest.mark.parametrize("T,K,N", [(2820, 200, 6), (3008, 100, 6), (37, 13, 3), (1, 4, 1), (500, 70, 7)])
def test_linear_bwd_small_class_head(lib, T, K, N):
    """ganffn_linear_fwd / _bwd on shapes the MFMA kernels do not take (N or K not a multiple of 4: the 6-class heads of
    GAN_FFN.fc and BiModel.smax_fc, model.py:1432,1062): y, dx, dW (+=) and db (+=) against fp64; the weight gradient is
    reduced over the tokens in a fixed order (second run bit-identical)"""
    g = torch.Generator().manual_seed(T + K + N)
    x, w, b, dy = torch.randn(T, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g), torch.randn(T, N, generator=g)
    gw0, gb0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    y = torch.empty(T, N, device="cuda")
    lib.call("ganffn_linear_fwd", ptr(xd), ptr(wd), ptr(bd), ptr(y), T, K, N, stream())
    assert rel_err(y, x.double() @ w.double().T + b.double()) < 3e-6

    def run():
        dx, gw, gb = torch.empty(T, K, device="cuda"), dev(gw0.clone()), dev(gb0.clone())
        lib.call("ganffn_linear_bwd", ptr(dyd), ptr(xd), ptr(wd), ptr(dx), ptr(gw), ptr(gb), T, K, N, None, C.c_int64(0), stream())
        return dx, gw, gb
    dx, gw, gb = run()
    assert rel_err(dx, dy.double() @ w.double()) < 3e-6
    assert rel_err(gw, gw0.double() + dy.double().T @ x.double()) < 3e-6 * max(1, T ** 0.5)
    assert rel_err(gb, gb0.double() + dy.double().sum(0)) < 3e-6 * max(1, T ** 0.5)
    dx2, gw2, gb2 = run()
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2) and torch.equal(dx, dx2)


@pytest.mark.parametrize("use_ws", [False, True])
def test_gemm_tn_grouped(lib, use_ws):
    """several weight-gradient GEMMs in one launch == each computed separately; with a workspace the narrow group splits
    its token ranges (problems of different K in one group: short ones get empty splits) and reduces in split order"""
    probs = [(100, 2048, 3300), (2048, 100, 3300), (100, 100, 330), (300, 100, 3300), (16, 64, 75), (512, 512, 1280)]
    g = torch.Generator().manual_seed(5)
    At = [dev(torch.randn(k, m, generator=g)) for (m, n, k) in probs]
    Bm = [dev(torch.randn(k, n, generator=g)) for (m, n, k) in probs]
    C0 = [torch.randn(m, n, generator=g) for (m, n, k) in probs]
    Cd = [dev(c.clone()) for c in C0]
    Sd = [torch.zeros(m, device="cuda") for (m, n, k) in probs]
    n = len(probs)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    ints = lambda i: (C.c_int * n)(*[p[i] for p in probs])
    nws = int(lib.load().ganffn_gemm_tn_grouped_workspace_floats())
    ws = torch.full((nws,), float("nan"), device="cuda") if use_ws else None
    lib.call("ganffn_gemm_tn_grouped", n, arr(At), arr(Bm), arr(Cd), arr(Sd), ints(0), ints(1), ints(2), ptr(ws), nws if use_ws else 0,
             stream())
    for i, (m, nn, k) in enumerate(probs):
        ref = C0[i].double() + At[i].double().cpu().T @ Bm[i].double().cpu()
        assert rel_err(Cd[i], ref) < 3e-6 * max(1, k ** 0.5), probs[i]
        assert rel_err(Sd[i], At[i].double().cpu().sum(0)) < 3e-6 * max(1, k ** 0.5)
    if use_ws:       # deterministic: a second run from the same inputs gives the same bits
        Cd2 = [dev(c.clone()) for c in C0]
        Sd2 = [torch.zeros(m, device="cuda") for (m, n, k) in probs]
        lib.call("ganffn_gemm_tn_grouped", n, arr(At), arr(Bm), arr(Cd2), arr(Sd2), ints(0), ints(1), ints(2), ptr(ws), nws, stream())
        assert all(torch.equal(a, b) for a, b in zip(Cd, Cd2)) and all(torch.equal(a, b) for a, b in zip(Sd, Sd2))
