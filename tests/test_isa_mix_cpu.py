"""The instruction mix of the hot K loops, from hipcc's own listing (tools/isa_mix.py; no GPU needed).

fp32 MFMAs share the SIMD's vector ALU, so a vector instruction the compiler adds next to them costs MFMA
time; round 5 removed the ones it found — accumulator shuffles, 64-bit per-lane addresses in
every K loop, select chains, exec-masked dead work.  These tests keep them out: they fail when a change (or a compiler update)
brings such instructions back into a steady-state block."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix  # noqa: E402

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")


def _kernels(fname):
    text = isa_mix.listing(os.path.join(isa_mix.CSRC, fname))
    found = list(isa_mix.blocks_of(text))
    names = isa_mix.demangle([n for n, _ in found])
    return {names[n].replace("ganffn::", "").replace("(anonymous namespace)::", ""): rows for n, rows in found}, text


@pytest.fixture(scope="module")
def gemm():
    return _kernels("gemm.hip")


def test_no_accumulator_moves_anywhere_in_gemm(gemm):
    """-amdgpu-mfma-vgpr-form: accumulators live in ordinary registers, nothing copies them in front of an epilogue"""
    _, text = gemm
    assert "v_accvgpr" not in text


def test_generic_gemm_k_loop_has_no_vector_alu_work_beside_its_mfmas(gemm):
    kernels, _ = gemm
    for name in ("void gemm_kernel<0, 64, 64, 16, 0, 2, 2, 0>(GemmArgs)", "void gemm_kernel<1, 64, 64, 16, 0, 2, 2, 0>(GemmArgs)"):
        rows = kernels[name]
        steady = [c for lab, n, c in rows if c["mfma"] == 8]
        assert len(steady) >= 4, rows
        # the four unrolled steady-state steps: 8 MFMAs, their LDS traffic, two buffer loads; the first carries the loop's few setup moves
        assert sorted(c["valu"] for c in steady)[:3] == [0, 0, 0] and max(c["valu"] for c in steady) <= 6, rows
        assert all(c["accmov"] == 0 for c in steady)


def test_weight_resident_epilogues_stay_lean(gemm):
    kernels, _ = gemm
    lin1 = [c for lab, n, c in kernels["void gemm_wres_kernel<0, 1, 100>(GemmArgs, int, int)"] if c["mfma"] == 50]
    dgrad = [c for lab, n, c in kernels["void gemm_wres_kernel<1, 3, 100>(GemmArgs, int, int)"] if c["mfma"] == 50]
    assert lin1 and dgrad
    # per tile of 50 MFMAs: train-mode linear1 epilogue ~118 vector instructions (was 184 + 16 accumulator reads), dgrad ~53 (was 173 + 16)
    assert max(c["valu"] for c in lin1) <= 130 and max(c["valu"] for c in dgrad) <= 64, (lin1, dgrad)


def test_d100_kernels_k_loops():
    n100, _ = _kernels("gemm_n100.hip")
    for name, rows in n100.items():
        if "gemm_n100_kernel<" in name and ", 2, " in name:                  # the 8-wave form the engine launches
            assert all(c["valu"] == 0 and c["accmov"] == 0 for lab, n, c in rows if c["mfma"] == 28), (name, rows)
    tn100, text = _kernels("gemm_tn100.hip")
    rows = tn100["void tn100_kernel<true>(W100Group)"]
    groups = [c for lab, n, c in rows if c["mfma"] == 14]
    assert len(groups) >= 8
    # 14 MFMAs per group; the vector-ALU work in these blocks is the one owner wave's bias sums (8 packed adds behind a scalar
    # branch) and, in the second stage's blocks, a few LDS address adds (round 4's loop: 2 per MFMA in every block)
    assert all(c["valu"] <= 14 and c["vmem"] == 0 for c in groups), groups
    assert sum(c["valu"] for c in groups) <= 0.8 * sum(c["mfma"] for c in groups), groups
    assert "v_accvgpr" not in text
