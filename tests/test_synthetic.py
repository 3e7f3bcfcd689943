"""workload generator + counter RNG spec properties (CPU)"""
import numpy as np
import pytest

from rankfm_amd import synthetic


@pytest.mark.parametrize("zipf", [0.0, 1.0])
def test_interactions_are_unique_exact_and_unsaturated(zipf):
    U, I, N = 500, 200, 20000
    pairs, csr = synthetic.make_interactions(U, I, N, seed=3, zipf_s=zipf)
    assert pairs.shape == (N, 2) and pairs.dtype == np.int32
    assert len(np.unique(pairs[:, 0].astype(np.int64) * I + pairs[:, 1])) == N
    assert pairs[:, 0].min() >= 0 and pairs[:, 0].max() < U and pairs[:, 1].max() < I
    deg = np.diff(csr.offsets)
    assert deg.sum() == N and deg.max() < I
    for u in (0, 17, U - 1):
        mine = np.sort(pairs[pairs[:, 0] == u, 1])
        assert np.array_equal(csr[u], mine)
    again, _ = synthetic.make_interactions(U, I, N, seed=3, zipf_s=zipf)
    assert np.array_equal(pairs, again)
    if zipf:
        cnt = np.sort(np.bincount(pairs[:, 1], minlength=I))[::-1]
        assert cnt[0] > 5 * np.median(cnt)          # a popularity head exists


def test_too_dense_is_rejected():
    with pytest.raises(ValueError):
        synthetic.make_interactions(10, 5, 45)


def test_init_weights_follow_reference_law():
    w = synthetic.init_weights(1000, 800, 16, n_user_features=4, n_item_features=0, sigma=0.1, alpha=0.01, beta=0.1, seed=1)
    assert w["v_u"].shape == (1000, 16) and w["v_u"].dtype == np.float32
    assert abs(w["v_u"].std() - 0.1) < 0.005 and abs(w["v_uf"].std() - 0.01) < 0.004      # (alpha/beta) * sigma
    assert not w["w_i"].any() and not w["w_if"].any() and not w["v_if"].any() and w["v_if"].shape == (1, 16)


def test_configs_match_baseline_json():
    import json
    import os
    from conftest import ROOT
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == len([k for k in synthetic.CONFIGS if k != "C4S"]) == 5
    c2 = synthetic.CONFIGS["C2"]
    assert (c2["n_users"], c2["n_items"], c2["n_interactions"], c2["factors"], c2["loss"]) == (100_000, 50_000, 5_000_000, 64, "bpr")
    assert "100k users" in base["configs"][1] and "5M interactions" in base["configs"][1] and "factors=64" in base["configs"][1]


def test_order_mirror_matches_the_c_spec(tmp_path):
    """rankfm_amd/order.py (numpy) must be the same function as include/rfm_rng.h (C, compiled into the kernels and the oracle):
    compile a probe against the header and compare epoch keys, permutations over many domain sizes, and a full epoch order"""
    import os
    import subprocess
    from conftest import ROOT
    from rankfm_amd import order
    src = tmp_path / "probe.c"
    src.write_text(r'''
#include <stdio.h>
#include "rfm_rng.h"
int main(void) {
    const uint32_t ek = rfm_epoch_key(77, 3);
    printf("%u\n", ek);
    for (uint32_t n = 1; n < 70; ++n) for (uint32_t p = 0; p < n; ++p) printf("%u ", rfm_perm(p, n, rfm_perm_bits(n), ek ^ n));
    printf("\n");
    for (uint32_t p = 0; p < 5000; ++p) printf("%u ", rfm_perm(p, 5000, rfm_perm_bits(5000), ek));
    printf("\n%u %u %u\n", rfm_row_key(ek, 12345), rfm_draw(rfm_row_key(ek, 12345), 7), rfm_draw_to_item(0xDEADBEEFu, 50000));
    return 0;
}''')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    ek = order.epoch_key(77, 3)
    assert ek == int(out[0])
    got = np.concatenate([order.perm(np.arange(n), n, order.perm_bits(n), ek ^ n) for n in range(1, 70)])
    assert np.array_equal(got, np.array(out[1].split(), dtype=np.int64))
    full = order.perm(np.arange(5000), 5000, order.perm_bits(5000), ek)
    assert np.array_equal(full, np.array(out[2].split(), dtype=np.int64)) and sorted(full.tolist()) == list(range(5000))
    assert int((0xDEADBEEF * 50000) >> 32) == int(out[3].split()[2])
    # a whole epoch of the segments kernel's order is a permutation of the CSR positions, user runs of <= 32 rows
    pairs, csr = synthetic.make_interactions(300, 200, 12000, seed=1)
    pos = order.epoch_positions(csr.offsets, 5, 0)
    assert sorted(pos.tolist()) == list(range(12000))
    seg_user, seg_begin, seg_len = order.segments(csr.offsets)
    assert seg_len.max() <= order.SEGMENT_ROWS and seg_len.min() >= 1 and seg_len.sum() == 12000
    assert np.array_equal(np.bincount(seg_user, weights=seg_len, minlength=300).astype(np.int64), np.diff(csr.offsets))
    assert not np.array_equal(pos, order.epoch_positions(csr.offsets, 5, 1))
