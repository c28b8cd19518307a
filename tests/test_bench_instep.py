"""bench.py's in-step roofline bookkeeping on a synthetic kernel trace (no GPU): family tagging, the frozen front end as
a family of its own, overlap attribution (shares add up to the busy time), the dominant kernel symbol, and the hash that
ties tracked counter summaries to the kernel sources."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

FWD = "void mxdet::conv_igemm_kernel<64, 128, 2, 2, 2, false, false, 1, 0>(mxdet::ConvP)"
DGR = "void mxdet::conv_igemm_grouped_kernel<64, 128, 2, 2, 2, true, 9>(mxdet::ConvG const*, int)"
WGR = "void mxdet::wgrad_mixed_grouped_kernel<2, 3>(mxdet::WgradG const*, int, unsigned char*, int, int)"
STEM = "void mxdet::stem_pool_kernel<float>(float const*, int)"
OTHER = "mxdet::nms_mask_kernel(float4 const*)"


def one_step(t0):
    """A 1000-us step: stem 0-100 (q2), its front-end conv 100-300 (q2), an unrelated kernel ends the front end on q2,
    a later forward conv on q2 is ordinary; main queue q1: fwd 300-500, dgrad 500-800 beside wgrad 600-900 (q3)."""
    us = 1000
    return [(t0 + 0 * us, t0 + 100 * us, STEM, "2"), (t0 + 100 * us, t0 + 300 * us, FWD, "2"),
            (t0 + 300 * us, t0 + 310 * us, OTHER, "2"), (t0 + 310 * us, t0 + 330 * us, FWD, "2"),
            (t0 + 300 * us, t0 + 500 * us, FWD, "1"), (t0 + 500 * us, t0 + 800 * us, DGR, "1"),
            (t0 + 600 * us, t0 + 900 * us, WGR, "3")]


def test_families_front_end_and_attribution():
    rows = []
    for k in range(5):
        rows += one_step(k * 1_000_000)
    r = bench.instep_from_rows(rows)
    assert r is not None and r["steps"] == 3
    assert abs(r["step_ms"] - 1.0) < 1e-9
    f = r["families_ms"]
    assert abs(f["conv_frozen_front"] - 0.2) < 1e-9          # only the conv right behind the stem on the stem's queue
    assert abs(f["conv_igemm_fwd"] - 0.22) < 1e-9            # main-queue forward + the later one on q2
    assert abs(f["conv_igemm_dgrad"] - 0.3) < 1e-9
    assert abs(f["conv_wgrad"] - 0.3) < 1e-9
    assert r["kernels"]["conv_igemm_fwd"] == 2 and r["kernels"]["conv_frozen_front"] == 1
    s = r["families_share_ms"]
    # dgrad 500-800 and wgrad 600-900 overlap for 200 us: each gets 100 of them
    assert abs(s["conv_igemm_dgrad"] - 0.2) < 1e-9 and abs(s["conv_wgrad"] - 0.2) < 1e-9
    # forward: 300-500 alone except 300-330 beside two short kernels on q2
    assert s["conv_igemm_fwd"] < f["conv_igemm_fwd"]
    assert r["dominant_kernel"]["name"].startswith("mxdet::conv_igemm_kernel<64, 128")   # 0.42 ms of forward symbol per step
    assert abs(r["dominant_kernel"]["ms_per_step"] - 0.42) < 1e-9


def test_too_short_a_trace_is_refused():
    rows = one_step(0) + one_step(1_000_000) + one_step(2_000_000)
    assert bench.instep_from_rows(rows) is None


def test_kernel_family_names():
    assert bench.kernel_family(FWD) == "conv_igemm_fwd"
    assert bench.kernel_family(DGR) == "conv_igemm_dgrad"
    assert bench.kernel_family(WGR) == "conv_wgrad"
    assert bench.kernel_family("mxdet::wgrad_reduce_grouped_kernel(mxdet::WgradG const*, int, unsigned char const*)") == "conv_wgrad"
    assert bench.kernel_family(OTHER) is None and bench.kernel_family(STEM) is None


def test_source_hash_follows_the_kernel_sources(tmp_path, monkeypatch):
    h = bench.csrc_sha16()
    assert len(h) == 16 and h == bench.csrc_sha16()
    import json
    p = os.path.join(bench.ROOT, "profiles", "r03_pmc_step.json")
    if os.path.exists(p):          # a tracked summary either matches the sources or bench.py reports traffic null
        assert isinstance(json.load(open(p)).get("csrc_sha16"), str)
