import os, sys, ctypes as C, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, ops
lib = _lib.load(); st = ops._stream(); P = ops._ptr
S, B = 94, 32
perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
def permrows(x, E):      # x [T, E] token t = s*B+b
    return x.view(S, B, E)[:, perm].reshape(S * B, E).contiguous()
for (N, K) in ((100, 100), (300, 100), (2048, 100), (100, 2048), (512, 512)):
    a = torch.randn(S * B, K, device="cuda"); w = torch.randn(N, K, device="cuda"); bias = torch.randn(N, device="cuda")
    c1 = torch.empty(S * B, N, device="cuda"); c2 = torch.empty_like(c1)
    _lib.call("ganffn_gemm_nt", P(a), P(w), P(bias), P(c1), S * B, N, K, st)
    ap = permrows(a, K)
    _lib.call("ganffn_gemm_nt", P(ap), P(w), P(bias), P(c2), S * B, N, K, st)
    print("gemm_nt N=%d K=%d differing: %d" % (N, K, int((c2 != permrows(c1, N)).sum())))
for E in (100, 512):
    T = S * B
    x = torch.randn(T, E, device="cuda"); y = torch.randn(T, E, device="cuda"); w = torch.rand(E, device="cuda"); b = torch.rand(E, device="cuda")
    rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
    def ln(x, y):
        out = torch.empty(T, E, device="cuda"); xh = torch.empty(T, E, device="cuda"); rs = torch.empty(T, device="cuda")
        _lib.call("ganffn_add_dropout_layernorm_fwd", P(x), P(y), P(w), P(b), P(out), P(xh), P(rs), T, E, C.c_float(1e-5), C.c_float(0.0), C.c_uint32(3), P(rng), C.c_uint64(0), st)
        return out
    o1 = ln(x, y); o2 = ln(permrows(x, E), permrows(y, E))
    print("layernorm E=%d differing: %d" % (E, int((o2 != permrows(o1, E)).sum())))
for (E, H) in ((100, 10), (512, 8)):
    qkv = torch.randn(S, B, 3 * E, device="cuda"); rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
    def att(q):
        o = torch.empty(S, B, E, device="cuda")
        _lib.call("ganffn_attention_fwd", P(q), P(o), None, S, B, E, H, C.c_float(0.0), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        return o
    o1 = att(qkv); o2 = att(qkv[:, perm].contiguous())
    print("attention E=%d differing: %d" % (E, int((o2 != o1[:, perm]).sum())))
E = 100; T = S * B
x = torch.randn(T, E, device="cuda"); y = torch.randn(T, E, device="cuda"); w = torch.rand(E, device="cuda"); b = torch.rand(E, device="cuda")
o1 = ln(x, y); o2 = ln(permrows(x, E), permrows(y, E))
d = (o2 != permrows(o1, E)).nonzero()
cols = torch.bincount(d[:, 1], minlength=E)
rows = d[:, 0].unique()
print("cols with diffs:", (cols > 0).sum().item(), "rows with diffs:", len(rows), "row%4 histogram:", torch.bincount(rows % 4, minlength=4).tolist(), "first rows:", rows[:10].tolist())
print("max abs diff", float((o2 - permrows(o1, E)).abs().max()))
def ln_full(x, y):
    out = torch.empty(T, E, device="cuda"); xh = torch.empty(T, E, device="cuda"); rs = torch.empty(T, device="cuda")
    _lib.call("ganffn_add_dropout_layernorm_fwd", P(x), P(y), P(w), P(b), P(out), P(xh), P(rs), T, E, C.c_float(1e-5), C.c_float(0.0), C.c_uint32(3), P(rng), C.c_uint64(0), st)
    return out, xh, rs
o1, xh1, rs1 = ln_full(x, y); o2, xh2, rs2 = ln_full(permrows(x, E), permrows(y, E))
rs1p = rs1.view(S, B)[:, perm].reshape(-1)
print("rstd differing rows:", int((rs2 != rs1p).sum()), " xhat differing elems:", int((xh2 != permrows(xh1, E)).sum()))
# which q (row % 4) in ORIGINAL vs PERMUTED position for differing rows
dr = (rs2 != rs1p).nonzero().flatten()
tperm = torch.arange(T, device="cuda").view(S, B)[:, perm].reshape(-1)     # permuted row r came from original row tperm[r]
print("pairs (q_new, q_old) histogram:", torch.bincount((dr % 4) * 4 + (tperm[dr] % 4), minlength=16).view(4, 4).tolist())
eq = (rs2 == rs1p).nonzero().flatten()
print("pairs for EQUAL rows:", torch.bincount((eq % 4) * 4 + (tperm[eq] % 4), minlength=16).view(4, 4).tolist())
