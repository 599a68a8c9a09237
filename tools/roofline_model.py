"""Kernel FAMILIES of the GAN step and their algorithmic work — the table bench.py and tests/test_bench_roofline_cpu.py
share, so that the bench line follows the committed single-stream profile instead of asserting what it shows.

A family = the kernels of one logical operation (symbol prefixes), its bound, and the algorithmic FLOPs (or bytes) ONE
ITERATION of the headline workload (12 sub-steps, /root/reference/train_IEMOCAP.py:355-382) asks of it, from the same
counts SURVEY.md §8d uses: per iteration the d_model-100 networks run 14 forward stack passes over T1 = S*B tokens (4
train-mode + 4 no-save generator passes, 6 frozen-discriminator passes) and 6 over T2 = 2*S*B ([real | fake]
discriminator passes), 10 + 6 backward passes (no weight gradients for the 6 frozen ones), the d_model-512 generator 4
forward and 2 backward passes over T1; 8 layers per stack.

`in_step` figures come from a committed rocprofv3 summary of the single-stream step (tools/prof_summary.py output with an
`iterations N` header): achieved = family work per iteration / family kernel time per iteration.
"""
import os
import re

FP32_MFMA_PEAK = 157.3e12      # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 flop/clk
HBM_PEAK = 8.0e12              # B/s
L, F = 8, 2048


def _code_only(text):
    """source text without comments and whitespace: the hash below identifies CODE, so that editing a comment does not make
    a committed profile look stale"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources' code (comments and whitespace stripped), in name order:
    identifies what a profile was taken on"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gan_ffn_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")):
            h.update(n.encode())
            h.update(_code_only(open(os.path.join(d, n)).read()).encode())
    h.update(_code_only(open(os.path.join(root, "include", "ganffn.h")).read()).encode())
    # the compiler flags decide the code as much as the sources do (round 5: -amdgpu-mfma-vgpr-form): the Makefile's CXXFLAGS line
    for line in open(os.path.join(d, "Makefile")):
        if line.startswith("CXXFLAGS"):
            h.update(line.strip().encode())
    return h.hexdigest()[:16]


def summary_sha(path):
    for line in open(path):
        f = line.split()
        if len(f) >= 2 and f[0] == "csrc_sha16":
            return f[1]
    return None


def parse_summary(path):
    """-> (rows, total_us, iterations); rows = dicts share / launches / avg_us / grid / name of the (kernel, grid) lines"""
    rows, total, iters = [], None, None
    for line in open(path):
        f = line.split()
        if len(f) >= 2 and f[0] == "iterations":
            iters = int(f[1])
        elif len(f) >= 5 and f[0].endswith("%") and f[3].startswith("("):
            rows.append(dict(share=float(f[0][:-1]), launches=int(f[1]), avg_us=float(f[2]), grid=f[3], name=" ".join(f[4:])))
        elif line.startswith("total GPU kernel time"):
            total = float(f[4]) * 1e3
        elif line.startswith("--- by kernel"):
            break
    return rows, total, iters


def _layer512(T):
    E = 512
    return dict(inp=2.0 * T * E * 3 * E, out=2.0 * T * E * E, l1=2.0 * T * E * F, l2=2.0 * T * F * E)


def family_table(S, B):
    """[family dict]: key, title, prefixes (kernel symbol prefixes of its launches), bound, per-iteration algorithmic
    work (`flops` or `bytes`) and, where the work per launch is needed, `launches` per iteration"""
    T1, T2 = S * B, 2 * S * B
    fwd100, bwd100 = 14 * T1 + 6 * T2, 10 * T1 + 6 * T2          # token-passes per layer, d_model 100
    fwd512, bwd512 = 4 * T1, 2 * T1
    ffn100 = 2.0 * 100 * F                                         # one K = 100 <-> 2048 product, per token
    l5 = _layer512(T1)
    fams = []
    # generic 64 x 64-tile GEMM: every nn.Linear of the d_model-512 generator (forward + dgrad) and the generator heads /
    # `object` of all networks (the d_model-100 layers have their own kernels below)
    g_fwd = 4 * L * (l5["inp"] + l5["out"] + l5["l1"] + l5["l2"])
    g_bwd = 2 * L * (l5["out"] + l5["l1"] + l5["l2"]) + 2 * (L - 1) * l5["inp"]       # the bottom layer's in-proj dgrad is skipped
    heads = 4 * (2.0 * T1 * 512 * 1024 + 2.0 * T1 * 1024 * 100) + 2 * (2.0 * T1 * 100 * 1024 + 2.0 * T1 * 1024 * 512) \
        + (8 + 4) * (2.0 * T1 * 100 * 512 + 2.0 * T1 * 512 * 100) + 2 * 2.0 * T1 * 512 * 100
    fams.append(dict(key="gemm_generic", bound="mfma", prefixes=["gemm_kernel<0", "gemm_kernel<1"], flops=g_fwd + g_bwd + heads,
                     title="generic 64x64-tile GEMM (gemm_kernel<NT|NN,...>): the d_model-512 generator's in-proj / out-proj / "
                           "linear1 / linear2 forward and dgrad, generator heads, `object`"))
    fams.append(dict(key="ffn_k100", bound="mfma", prefixes=["gemm_wres_kernel"], flops=ffn100 * L * (fwd100 + bwd100),
                     launches=L * (14 + 6 + 10 + 6),
                     title="gemm_wres_kernel: the K = 100 -> 2048 products of the d_model-100 feed-forward block (linear1 forward "
                           "with bias + ReLU + dropout, linear2 dgrad with the ReLU/dropout mask); persistent, weights register-resident"))
    fams.append(dict(key="ffn_n100", bound="mfma", prefixes=["gemm_n100_kernel"], flops=ffn100 * L * (fwd100 + bwd100),
                     launches=L * (14 + 6 + 10 + 6),
                     title="gemm_n100_kernel: the 2048 -> 100 products of the d_model-100 feed-forward block (linear2 forward, "
                           "linear1 dgrad) on 112-wide 16x16x4 tiles, K-chunk slabs"))
    w100 = lambda T: L * 2.0 * T * (100 * F + F * 100 + 100 * 100 + 300 * 100)
    w512 = L * 2.0 * T1 * (512 * F + F * 512 + 512 * 512 + 1536 * 512)
    fams.append(dict(key="wgrad", bound="mfma", prefixes=["tn100_kernel", "tn100_reduce_kernel", "gemm_tn_grouped_kernel", "tn_reduce_grouped_kernel"],
                     flops=6 * w100(T2) + 4 * w100(T1) + 2 * w512, launches=12,
                     title="grouped weight-gradient launch (tn100_kernel + tn100_reduce_kernel for d_model 100, "
                           "gemm_tn_grouped_kernel for d_model 512): all 32 weight and bias gradients of one encoder backward pass"))
    att = lambda E, fw, bw: L * 4.0 * S * E * (fw + 2.5 * bw)
    fams.append(dict(key="attention", bound="mfma", prefixes=["attn16_fwd_kernel", "attn16_bwd_kernel", "attention_fwd_kernel", "attention_bwd_kernel"],
                     flops=att(100, fwd100, bwd100) + att(512, fwd512, bwd512),
                     title="attention core (attn16_fwd/bwd_kernel<10,..> for the five d_model-100 networks, attention_fwd/bwd_kernel<64,..> "
                           "for the generator with head_dim 64): softmax(QK^T)V and its backward, S = %d keys — a latency chain per "
                           "(dialogue, head), priced against the MFMA peak as the largest roof it could have" % S))
    # rowchain: token-local chains of a d_model-100 layer; HBM-side operand bytes per token per layer: forward out-proj + LN1
    # reads attn_o, x, writes x1, xhat1 (4 E); LN2 + next in-proj reads 5 slabs + x1, writes x, xhat2, qkv (11 E); backward
    # mirrors them (LN backward reads dy, xhat, writes dx: + the slabs of the dgrad products): ~ 15 E + 15 E floats
    rc_bytes = 4.0 * 100 * L * (15 * fwd100 + 15 * bwd100)
    fams.append(dict(key="rowchain", bound="hbm", prefixes=["rc_fwd_kernel", "rc_bwd_kernel"], bytes=rc_bytes,
                     title="rowchain (rc_fwd/bwd_kernel): out-proj + residual + dropout + LN1, LN2 + next in-proj and their backward "
                           "mirrors for d_model 100, 16 token rows per workgroup — latency-sized (5-17 us) kernels, priced on their "
                           "operand bytes against HBM"))
    return fams


def family_of(name, fams):
    for f in fams:
        if any(name.startswith(p) for p in f["prefixes"]):
            return f["key"]
    return None


def in_step(path, S, B, min_share=5.0):
    """per family >= min_share % of the profiled kernel time: share, in-step time per iteration, achieved and frac"""
    rows, total_us, iters = parse_summary(path)
    if not rows or not iters or not total_us:
        return None
    fams = family_table(S, B)
    out = []
    for f in fams:
        mine = [r for r in rows if family_of(r["name"], fams) == f["key"]]
        t_us = sum(r["avg_us"] * r["launches"] for r in mine)
        share = 100.0 * t_us / total_us
        if share < min_share:
            continue
        per_iter_s = t_us / iters * 1e-6
        work = f.get("flops", f.get("bytes"))
        peak = FP32_MFMA_PEAK if f["bound"] == "mfma" else HBM_PEAK
        ach = work / per_iter_s
        d = dict(family=f["key"], kernel=f["title"], bound=f["bound"], share_pct=round(share, 2),
                 in_step_ms_per_iteration=round(per_iter_s * 1e3, 4),
                 achieved=round(ach / (1e12 if f["bound"] == "mfma" else 1e9), 2), peak=peak / (1e12 if f["bound"] == "mfma" else 1e9),
                 unit="TFLOP/s" if f["bound"] == "mfma" else "GB/s", frac=round(ach / peak, 4),
                 algorithmic_per_iteration=work, launches_profiled=sum(r["launches"] for r in mine))
        if "launches" in f:
            d["launches_per_iteration"] = f["launches"]
            # tn100_reduce is part of the logical launch, not a launch of its own
            d["in_step_avg_us"] = round(t_us / iters / f["launches"], 2)
        out.append(d)
    out.sort(key=lambda d: -d["share_pct"])
    return dict(profile=os.path.basename(path), profile_csrc_sha16=summary_sha(path), iterations=iters, kernel_time_ms_per_iteration=round(total_us / iters / 1e3, 3),
                families=out)
