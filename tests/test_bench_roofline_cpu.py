"""The bench line must FOLLOW the committed profile: every roofline fraction of the committed bench JSON is recomputed
here from the committed rocprofv3 summary (in-step figures) or from the line's own live fields, the dominant family is the
one with the largest share of that summary, and `roofline_worst` is the minimum over the listed families.
CPU only: reads profiles/ and tools/roofline_model.py."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import roofline_model as RM  # noqa: E402

BENCH = os.path.join(ROOT, "profiles", "r05_bench_default.json")
SUMMARY = os.path.join(ROOT, "profiles", "r05_bench_streams1_by_launch_shape.txt")
needs_profiles = pytest.mark.skipif(not (os.path.exists(BENCH) and os.path.exists(SUMMARY)), reason="round-5 profiles not committed yet")


def load_line():
    for line in open(BENCH):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in " + BENCH)


def test_family_table_adds_up_to_the_survey_flop_count():
    """the GEMM-shaped families' per-iteration FLOPs reproduce SURVEY.md §8d's 954.3 MFLOP per padded token (S = 94) minus
    what the build does not execute: the frozen discriminators' weight gradients in train_gen (6 passes)"""
    S, B = 94, 32
    fams = {f["key"]: f for f in RM.family_table(S, B)}
    total = sum(fams[k]["flops"] for k in ("gemm_generic", "ffn_k100", "ffn_n100", "wgrad", "attention"))
    T = S * B
    # rowchain's tiny GEMMs (in-/out-proj of d_model 100) and the discriminator heads are the rest
    E = 100
    rc = 8 * (2.0 * E * E + 6.0 * E * E) * ((14 + 10) * T + (6 + 6) * 2 * T) - 8 * 6.0 * E * E * (4 * T + 6 * 2 * T)   # skipped bottom in-proj dgrads approx
    survey = 954.3e6 * T
    frozen_wgrad = 6 * 8 * 2.0 * T * (100 * 2048 * 2 + 100 * 100 + 300 * 100)
    assert abs((total + rc) - (survey - frozen_wgrad)) / survey < 0.03, (total + rc, survey - frozen_wgrad)


@needs_profiles
def test_summary_has_header_and_families_cover_the_step():
    rows, total_us, iters = RM.parse_summary(SUMMARY)
    assert iters and iters >= 10 and total_us > 0 and RM.summary_sha(SUMMARY)
    r = RM.in_step(SUMMARY, 94, 32)
    shares = {f["family"]: f["share_pct"] for f in r["families"]}
    assert sum(shares.values()) > 85.0, shares                      # the listed families are (nearly) the whole step
    assert all(s >= 5.0 for s in shares.values())


@needs_profiles
def test_bench_line_fractions_follow_the_committed_profile():
    line = load_line()
    r = RM.in_step(SUMMARY, 94, 32)
    fams = {f["family"]: f for f in line["roofline_families"]}
    prof = {f["family"]: f for f in r["families"]}
    # every family with >= 5 % of the profiled kernel time is in the line, with the in-step fraction of the profile
    assert set(prof) == set(fams), (sorted(prof), sorted(fams))
    for k, p in prof.items():
        assert abs(fams[k]["in_step_frac"] - p["frac"]) <= 1e-3 + 0.01 * p["frac"], (k, fams[k]["in_step_frac"], p["frac"])
        assert abs(fams[k]["share_pct"] - p["share_pct"]) < 0.05
        if fams[k].get("avg_kernel_us"):          # live replay: frac = gflop per launch / avg launch duration / peak
            peak = RM.FP32_MFMA_PEAK
            frac = fams[k]["avg_gflop_per_launch"] * 1e9 / (fams[k]["avg_kernel_us"] * 1e-6) / peak
            assert abs(frac - fams[k]["frac"]) <= 2e-3 + 0.01 * frac, (k, frac, fams[k]["frac"])
    # the headline roofline object names the family with the LARGEST share ...
    dominant = max(prof.values(), key=lambda f: f["share_pct"])["family"]
    assert line["roofline"]["family"] == dominant
    assert abs(line["roofline"]["frac"] - fams[dominant]["frac"]) < 1e-9
    assert abs(line["roofline"]["achieved"] / line["roofline"]["peak"] - line["roofline"]["frac"]) < 2e-3
    # ... and roofline_worst the one with the LOWEST in-step fraction AMONG THE FAMILIES ON THE SAME ROOF (a fraction of the MFMA
    # peak and a fraction of the HBM peak are not comparable: ADVICE r4); the HBM-bound families have roofline_worst_hbm
    for bound, key in (("mfma", "roofline_worst"), ("hbm", "roofline_worst_hbm")):
        cands = [f for f in fams.values() if f["bound"] == bound]
        if cands:
            assert line[key]["family"] == min(cands, key=lambda f: f["in_step_frac"])["family"]
            assert line[key]["bound"] == bound
    # the profile the line quotes was taken on the kernel sources the line was produced with
    assert line["profile"]["file"].endswith(os.path.basename(SUMMARY))
    # below peak, as it must be
    for f in fams.values():
        assert 0.0 < f["in_step_frac"] < 1.0 and 0.0 < f["frac"] < 1.0
    step = line["config"]["step_frac_of_fp32_mfma_peak"]
    assert 0.3 < step < 1.0
    assert abs(step - 954.3e6 * 94 * 32 / (line["ms_per_step"] * 1e-3) / RM.FP32_MFMA_PEAK) < 5e-3
    # the EXECUTED fraction leaves out the frozen discriminators' weight gradients (6 passes the reference computes and drops)
    frozen = 6 * (8 * 2.0 * (100 * 2048 * 2 + 100 * 100 + 300 * 100) + 2.0 * (100 * 64 + 64 * 16 + 16))
    ex = line["config"]["step_frac_executed"]
    assert abs(ex - (954.3e6 - frozen) * 94 * 32 / (line["ms_per_step"] * 1e-3) / RM.FP32_MFMA_PEAK) < 5e-3 and ex < step
    # the median of device-synchronised iterations is reported beside the headline, and is the slower of the two
    assert line["config"]["synchronised_iterations"] >= 20
    assert line["config"]["median_ms_per_step_synchronised"] >= 0.98 * line["ms_per_step"]


@needs_profiles
def test_committed_profiles_were_taken_on_the_kernel_code_in_the_tree():
    """the in-step figures of the bench line come from these summaries; their header carries a hash of the kernel sources'
    CODE (comments and whitespace stripped) — a kernel change without a fresh profile fails here, and bench.py would flag the
    in-step fields as stale"""
    assert RM.summary_sha(SUMMARY) == RM.csrc_sha16()
    drnn = os.path.join(ROOT, "profiles", "r05_drnn_by_launch_shape.txt")
    assert RM.summary_sha(drnn) == RM.csrc_sha16()
