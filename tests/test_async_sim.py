"""oracle/rfm_async_sim.c -- the CPU model of how the stripe kernel executes an epoch (analysis infrastructure, see its header) --
is pinned to the oracle where it must coincide with it: restricted to ONE row group it is the sequential algorithm, with and
without negative stripes, on the engine's order and draws.  (What the model is for -- many groups in lock-step rounds, deferred
atomics, stripes, skewed workgroups -- has no CPU ground truth; it is checked against GPU measurements in profiles/r02_notes.md.)"""
import numpy as np
import pytest

from rankfm_amd import order, synthetic


@pytest.mark.parametrize("stripes", [False, True])
def test_one_group_of_the_model_is_the_oracle(oracle, stripes):
    from oracle import async_sim as sim
    U, I, N, F, seed = 300, 200, 9000, 8, 5
    pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
    sw = np.random.default_rng(1).uniform(0.5, 1.5, N).astype(np.float32)
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs_csr, sw_csr = np.ascontiguousarray(pairs[by_csr]), np.ascontiguousarray(sw[by_csr])
    n_seg = len(order.segments(csr.offsets)[0])
    geo = dict(workgroups=1, groups_per_workgroup=64, working_groups=1, units_per_launch=n_seg, n_units=n_seg,
               stripe_rows=16 if stripes else 0, stripe_window=1, single_group=True, epoch_part=None, n_items=I)
    perms = np.stack([order.epoch_positions(csr.offsets, seed, e) for e in range(2)]).astype(np.int32)
    w = synthetic.init_weights(U, I, F, seed=1)
    out = oracle.fit(pairs_csr, sw_csr, csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w["w_i"], w["w_if"],
                     w["v_u"], w["v_i"], w["v_uf"], w["v_if"], 0.01, 0.1, 0.1, "constant", 0.25, 1, 2, perms=perms,
                     rng_mode=oracle.RNG_COUNTER, seed=seed, membership="binary", **order.oracle_stripes(csr.offsets, seed, range(2), geo, I))
    s = synthetic.init_weights(U, I, F, seed=1)
    ll = [sim.epoch(pairs_csr, sw_csr, csr.offsets, csr.items, s, seed, e, geo, mean_view=0.0)[0] for e in range(2)]
    for k in ("v_u", "v_i", "w_i"):
        np.testing.assert_allclose(s[k], w[k], rtol=2e-6, atol=2e-6, err_msg=k)      # fp32 rounding (different expression order)
    np.testing.assert_allclose(ll, out["ll64"], rtol=1e-7)


def test_row_schedule_is_a_walk_of_every_group_through_its_segments():
    U, I, N = 500, 300, 20000
    pairs, csr = synthetic.make_interactions(U, I, N, seed=2)
    n_seg = len(order.segments(csr.offsets)[0])
    geo = dict(workgroups=2, groups_per_workgroup=64, working_groups=100, units_per_launch=n_seg, n_units=n_seg, stripe_rows=0,
               stripe_window=1, single_group=False, epoch_part=None, n_items=I)
    sch = order.row_schedule(csr.offsets, 3, 0, geo)
    assert sorted(sch["pos"].tolist()) == list(range(N))                         # every row once
    assert sch["group"].max() == 99 and np.array_equal(sch["workgroup"], sch["group"] // 64)
    for g in (0, 37, 99):                                                        # a group's rows: iterations 0, 1, 2, ... without gaps
        its = np.sort(sch["it"][sch["group"] == g])
        assert np.array_equal(its, np.arange(len(its)))
    # group g takes the segments at order positions g, g + 100, ...
    assert np.array_equal(sch["group"], sch["sp"] % 100)
