"""Negative stripes on the CPU side: the spec functions of include/rfm_rng.h (through the oracle, which includes the header),
the host mirror of the window schedule (rankfm_amd/order.py) and the statistics the scheme has to keep -- every item offered as
a negative equally often, the sequential oracle learning the same model with and without stripes."""
import numpy as np
import pytest

from rankfm_amd import order, synthetic


def _geometry(R, RW, groups, gpw=64, single=False, upl=10**9, part=None):
    return dict(stripe_rows=R, stripe_window=RW, single_group=single, working_groups=groups, groups_per_workgroup=gpw,
                workgroups=1 if single else (groups + gpw - 1) // gpw, units_per_launch=upl, n_units=0, epoch_part=part)


def _striped_fit(oracle, pairs, csr, w, geo, seed, epochs, want_negatives=False):
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs_csr = np.ascontiguousarray(pairs[by_csr])
    U, I = w["v_u"].shape[0], w["v_i"].shape[0]
    perms = np.stack([order.epoch_positions(csr.offsets, seed, e) for e in range(epochs)]).astype(np.int32)
    kw = order.oracle_stripes(csr.offsets, seed, range(epochs), geo, I) if geo else {}
    return oracle.fit(pairs_csr, np.ones(len(pairs), np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32),
                      np.zeros((I, 1), np.float32), w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"], 0.01, 0.1, 0.1,
                      "constant", 0.25, 1, epochs, perms=perms, rng_mode=oracle.RNG_COUNTER, seed=seed, membership="binary",
                      want_negatives=want_negatives, **kw), kw


@pytest.mark.parametrize("R, RW, groups", [(64, 4, 300), (190, 12, 1235), (37, 7, 100)])
def test_mirror_tiles_the_item_cycle(R, RW, groups):
    """rfm_stripe_start: the window slots of a launch start R positions apart around the cycle of I positions, so the stripes
    used during any stretch of the launch cover every position equally often (+-1)"""
    U, I, N = 3000, 1500, 90_000
    pairs, csr = synthetic.make_interactions(U, I, N, seed=5)
    geo = dict(_geometry(R, RW, groups), n_items=I)
    starts = order.row_stripes(csr.offsets, 7, 0, geo)
    assert starts.shape == (N,) and starts.min() >= 0 and starts.max() < I
    slots = np.unique(starts)
    # consecutive slot starts differ by R (mod I): sort by slot index through the known arithmetic progression
    cover = np.zeros(I, dtype=np.int64)
    for s in slots:
        cover[(s + np.arange(R)) % I] += 1
    assert cover.max() - cover.min() <= 2, (cover.min(), cover.max())     # (+1 for the slots the launch did not finish)
    # the rows a group visits within one window share a stripe; its next window has another one
    pos, sp, t, seg_len, _ = order._visit(csr.offsets, 7, 0)
    first = pos[(sp == 0)]                                       # rows of the first segment of group 0, in visiting order
    k = min(len(first), RW)
    assert len(np.unique(starts[first[:k]])) == 1
    if len(first) > RW:
        assert starts[first[RW]] != starts[first[0]]


def test_single_group_mirror_advances_one_stripe_per_row():
    pairs, csr = synthetic.make_interactions(50, 40, 600, seed=1)
    geo = dict(_geometry(16, 1, 1, gpw=1, single=True), n_items=40)
    starts = order.row_stripes(csr.offsets, 3, 0, geo)
    visit = order.epoch_positions(csr.offsets, 3, 0)
    d = np.diff(starts[visit].astype(np.int64)) % 40
    assert np.all(d == 16)                                       # slot n starts at n * R + offset (mod I)


def test_oracle_draws_stay_inside_the_stripe_and_reject_the_users_items(oracle):
    U, I, N = 400, 300, 12_000
    pairs, csr = synthetic.make_interactions(U, I, N, seed=2)
    w = synthetic.init_weights(U, I, 8, seed=3)
    geo = _geometry(32, 4, 100)
    out, kw = _striped_fit(oracle, pairs, csr, w, geo, seed=11, epochs=1, want_negatives=True)
    neg = out["neg"][0]
    visit = order.epoch_positions(csr.offsets, 11, 0)
    starts = kw["row_stripe"][0][visit]                           # stripe start of the row visited at each position
    # position of each negative in the epoch's item permutation must lie in [start, start + R)
    ek = order.epoch_key(11, 0)
    inv = np.empty(I, dtype=np.int64)
    inv[order.perm(np.arange(I), I, order.perm_bits(I), ek ^ 0x2545F491)] = np.arange(I)
    off = (inv[neg] - starts) % I
    assert np.all(off < 32)
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    users_csr = pairs[by_csr][:, 0][visit]
    for r in range(0, N, 97):                                     # never one of the user's own items
        u = users_csr[r]
        assert neg[r] not in csr.items[csr.offsets[u]:csr.offsets[u + 1]]


def test_stripes_offer_every_item_equally_often_and_learn_the_same_model(oracle):
    """per-item negative counts keep the Poisson spread of uniform sampling (a hashed slot -> stripe assignment tripled it and
    cost the sequential oracle itself 1-2 points of hit rate), and two epochs of the sequential oracle with and without stripes
    end at the same log-likelihood (1.5 %) and norms (2.5 %; measured +1.8 % on v_i: in the ORACLE's sequential order the rows of
    one workgroup's window come in a burst, on the GPU they are spread over the window's span)"""
    U, I, N = 2000, 1200, 120_000
    pairs, csr = synthetic.make_interactions(U, I, N, seed=4, zipf_s=0.0)
    res = {}
    for name, geo in (("plain", None), ("stripes", _geometry(96, 6, 400))):
        w = synthetic.init_weights(U, I, 16, seed=9)
        out, _ = _striped_fit(oracle, pairs, csr, w, geo, seed=21, epochs=2, want_negatives=True)
        cnt = np.bincount(out["neg"][0], minlength=I)
        res[name] = (cnt, out["ll"], {k: np.linalg.norm(w[k]) for k in ("v_u", "v_i", "w_i")})
    mean = N / I
    for name in res:
        cnt = res[name][0]
        assert abs(cnt.mean() - mean) < 1e-9 and cnt.std() < 1.35 * np.sqrt(mean), (name, cnt.std(), np.sqrt(mean))
    np.testing.assert_allclose(res["stripes"][1], res["plain"][1], rtol=0.015)
    for k in ("v_u", "v_i", "w_i"):
        assert abs(res["stripes"][2][k] / res["plain"][2][k] - 1.0) < 0.025, k


def test_saturated_stripe_falls_back_to_the_catalogue(oracle):
    """a user who holds a whole stripe: after RFM_STRIPE_ATTEMPTS rejected draws the row draws from the whole catalogue"""
    I = 64
    heavy = np.stack([np.zeros(56, np.int64), np.arange(56)], 1)                # user 0 holds 56 of 64 items
    light = np.stack([np.arange(1, 30), np.arange(1, 30) % I], 1)
    pairs = np.concatenate([heavy, light]).astype(np.int32)
    from rankfm_amd import UserItemsCSR
    csr = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], 30)
    w = synthetic.init_weights(30, I, 4, seed=1)
    out, _ = _striped_fit(oracle, pairs, csr, w, dict(_geometry(8, 2, 1, gpw=1, single=True)), seed=5, epochs=1, want_negatives=True)
    neg = out["neg"][0]
    visit = order.epoch_positions(csr.offsets, 5, 0)
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    users_csr = pairs[by_csr][:, 0][visit]
    assert np.all(neg[users_csr == 0] >= 56) and np.isfinite(out["ll"]).all()


def test_segments_follow_the_plans_segment_length():
    """rfm_fit_report.segment_rows: a plan that uses negative stripes cuts 16-row segments (a user's rows then span several
    stripes); the host mirror must cut the same ones"""
    from rankfm_amd import order, synthetic
    pairs, csr = synthetic.make_interactions(300, 200, 12000, seed=4)
    for rows in (32, 16, 8):
        users, begin, length = order.segments(csr.offsets, rows)
        assert length.max() <= rows and length.min() >= 1 and length.sum() == 12000
        deg = np.diff(csr.offsets)
        assert len(users) == int(((deg + rows - 1) // rows).sum())
        assert np.array_equal(np.sort(order.epoch_positions(csr.offsets, 3, 0, rows)), np.arange(12000))
    assert np.array_equal(order.epoch_positions(csr.offsets, 3, 0), order.epoch_positions(csr.offsets, 3, 0, 32))
