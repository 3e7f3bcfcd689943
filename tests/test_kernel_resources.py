"""The register budgets of the production kernels, read from the built library's code-object metadata (tools/kernel_resources.py):
the kernels the BASELINE configurations launch must not spill vector registers -- a spill in a row loop is a scratch round trip per
row.  Runs without a GPU: hipcc cross-compiles, the metadata is in the ELF notes."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources
    from rankfm_amd import _build
    rows = kernel_resources.kernels(_build.build())
    assert rows, "no device code in the library"
    return rows


def test_the_library_carries_gfx950_code_only(kernels):
    assert {r["triple"].split("--")[-1] for r in kernels} == {"gfx950"}, sorted({r["triple"] for r in kernels})


def pick(kernels, prefix):
    out = [r for r in kernels if r["kernel"].startswith(prefix)]
    assert out, prefix
    return out


def test_the_bench_kernel_and_the_warp_kernels_do_not_spill(kernels):
    # config 2 / config 1: BPR with hot-row accumulators, 16-lane row groups, k = 64 / 32 / 16 -- on segment-major item rows (the bench
    # kernel since round 5) and on row-major ones
    for kpl in (1, 2, 4):
        for fresh in ("false", "true"):
            for split in ("true", "false"):
                for r in pick(kernels, "sgd_segments_kernel<16, %d, %s, true, false, false, %s>" % (kpl, fresh, split)):
                    assert r["vgpr_spill"] == 0 and r["scratch"] == 0 and r["vgpr"] <= 128, r
    # ... and no segment-major BPR instantiation the host can launch (k = 16 ... 96) spills either (k = 128 stays row-major: rfm_api.hip)
    for kpl in (3, 6):
        for r in pick(kernels, "sgd_segments_kernel<16, %d, false, true, false, false, true>" % kpl):
            assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r
    assert not [r for r in kernels if r["kernel"].startswith("sgd_segments_kernel<16, 8, false, true, false, false, true>")]
    # configs 3 / 5 and every other factor count: the WARP state machine
    for r in pick(kernels, "sgd_warp_kernel<16, "):
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r


def test_the_pipelined_feature_row_loop_fits_three_wavefronts_per_simd(kernels):
    # config 4: 768-thread workgroups = 168 registers, nothing in scratch
    for r in pick(kernels, "sgd_features_fast_kernel<16, 4, "):
        if r["max_wg"] == 768:
            assert r["vgpr"] <= 168 and r["vgpr_spill"] == 0 and r["scratch"] == 0, r
    for r in pick(kernels, "feat_tables_kernel<16, 4, false>"):
        assert r["vgpr_spill"] == 0, r
