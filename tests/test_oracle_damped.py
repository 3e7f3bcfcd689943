"""The oracle's damped variant (oracle/rfm_oracle.c: rfm_oracle_fit_ex) -- the engine's Hogwild step damping applied to the
SEQUENTIAL algorithm, so that GPU tests can tell the deliberate change of the optimiser from the effect of asynchrony.
CPU-only checks of the variant itself; the reference-pinned entry point is unchanged (tests/test_oracle_golden.py)."""
import numpy as np

from rankfm_amd import synthetic


def _fit(oracle, w, pairs, csr, epochs=2, **kw):
    g = {k: v.copy() for k, v in w.items()}
    U, I = len(w["v_u"]), len(w["w_i"])
    out = oracle.fit(pairs, np.ones(len(pairs), np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32),
                     np.zeros((I, 1), np.float32), g["w_i"], g["w_if"], g["v_u"], g["v_i"], g["v_uf"], g["v_if"], 0.01, 0.1, 0.1,
                     "constant", 0.25, 1, epochs, perms=None, rng_mode=oracle.RNG_COUNTER, seed=7, membership="binary", **kw)
    return g, out


def test_unit_scales_are_the_plain_oracle_bit_for_bit(oracle):
    U, I, N, F = 60, 40, 1500, 8
    pairs, csr = synthetic.make_interactions(U, I, N, seed=3)
    w = synthetic.init_weights(U, I, F, seed=4)
    a, oa = _fit(oracle, w, pairs, csr)
    b, ob = _fit(oracle, w, pairs, csr, pos_step=np.ones(I, np.float32), user_step=np.ones(U, np.float32))
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(oa["ll"], ob["ll"])


def test_zero_scales_freeze_exactly_what_they_scale(oracle):
    U, I, N, F = 60, 40, 1500, 8
    pairs, csr = synthetic.make_interactions(U, I, N, seed=3)
    w = synthetic.init_weights(U, I, F, seed=4)
    g, _ = _fit(oracle, w, pairs, csr, pos_step=np.ones(I, np.float32), user_step=np.zeros(U, np.float32))
    assert np.array_equal(g["v_u"], w["v_u"]) and not np.array_equal(g["v_i"], w["v_i"])
    # one interaction (user 0, item 0) over three items: with pos_step[0] = 0 item 0 never moves, the drawn negative does
    pairs1 = np.array([[0, 0]], np.int32)
    csr1 = type(csr).from_pairs(pairs1[:, 0], pairs1[:, 1], 1)
    w1 = synthetic.init_weights(1, 3, 4, seed=5)
    g, out = _fit(oracle, w1, pairs1, csr1, epochs=1, pos_step=np.array([0, 1, 1], np.float32), user_step=np.ones(1, np.float32),
                  want_negatives=True)
    j = int(out["neg"][0, 0])
    assert j in (1, 2)
    assert np.array_equal(g["v_i"][0], w1["v_i"][0]) and g["w_i"][0] == w1["w_i"][0]
    assert not np.array_equal(g["v_i"][j], w1["v_i"][j]) and not np.array_equal(g["v_u"], w1["v_u"])


def test_one_damped_step_by_hand(oracle):
    """a single interaction: the update block (rankfm/_rankfm.pyx:276-310) recomputed in numpy with eta_i = s_i eta, eta_u = s_u eta"""
    pairs1 = np.array([[0, 0]], np.int32)
    from rankfm_amd._rankfm import UserItemsCSR
    csr1 = UserItemsCSR.from_pairs(pairs1[:, 0], pairs1[:, 1], 1)
    w = synthetic.init_weights(1, 3, 4, seed=6)
    for k in ("v_u", "v_i"):
        w[k] = (w[k] * 30).astype(np.float32)              # large enough for the logistic term to matter
    s_i, s_u, eta, alpha = 0.25, 0.5, 0.1, 0.01
    g, out = _fit(oracle, w, pairs1, csr1, epochs=1, pos_step=np.array([s_i, 1, 1], np.float32), user_step=np.array([s_u], np.float32),
                  want_negatives=True)
    j = int(out["neg"][0, 0])
    vu, vi, vj = (w["v_u"][0].astype(np.float64), w["v_i"][0].astype(np.float64), w["v_i"][j].astype(np.float64))
    pu = (w["w_i"][0] + vu @ vi) - (w["w_i"][j] + vu @ vj)
    d = 1.0 / (np.exp(pu) + 1.0)
    mult = np.log((3 - 1) // 1) / np.log(3)
    reg = 2 * alpha
    np.testing.assert_allclose(g["v_u"][0], vu + s_u * eta * (mult * d * (vi - vj) - reg * vu), rtol=2e-6)
    np.testing.assert_allclose(g["v_i"][0], vi + s_i * eta * (mult * d * vu - reg * vi), rtol=2e-6)
    np.testing.assert_allclose(g["v_i"][j], vj + eta * (mult * d * -vu - reg * vj), rtol=2e-6)
    np.testing.assert_allclose(g["w_i"][0], w["w_i"][0] + s_i * eta * (mult * d - reg * w["w_i"][0]), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(g["w_i"][j], w["w_i"][j] + eta * (-mult * d - reg * w["w_i"][j]), rtol=2e-6, atol=1e-9)


def test_negative_side_scale_by_hand(oracle):
    """`neg_step` (the engine scales an item's step whichever side of the pair it is on): the same single interaction with the drawn
    negative's step scaled -- bias and factor row -- and nothing else changed"""
    pairs1 = np.array([[0, 0]], np.int32)
    from rankfm_amd._rankfm import UserItemsCSR
    csr1 = UserItemsCSR.from_pairs(pairs1[:, 0], pairs1[:, 1], 1)
    w = synthetic.init_weights(1, 3, 4, seed=6)
    for k in ("v_u", "v_i"):
        w[k] = (w[k] * 30).astype(np.float32)
    s_j, eta, alpha = 0.125, 0.1, 0.01
    one_i, one_u = np.ones(3, np.float32), np.ones(1, np.float32)
    plain, out = _fit(oracle, w, pairs1, csr1, epochs=1, pos_step=one_i, user_step=one_u, want_negatives=True)
    g, out2 = _fit(oracle, w, pairs1, csr1, epochs=1, pos_step=one_i, user_step=one_u, neg_step=np.full(3, s_j, np.float32), want_negatives=True)
    j = int(out["neg"][0, 0])
    assert int(out2["neg"][0, 0]) == j
    vu, vi, vj = (w["v_u"][0].astype(np.float64), w["v_i"][0].astype(np.float64), w["v_i"][j].astype(np.float64))
    pu = (w["w_i"][0] + vu @ vi) - (w["w_i"][j] + vu @ vj)
    d = 1.0 / (np.exp(pu) + 1.0)
    mult = np.log((3 - 1) // 1) / np.log(3)
    reg = 2 * alpha
    np.testing.assert_allclose(g["v_i"][j], vj + s_j * eta * (mult * d * -vu - reg * vj), rtol=2e-6)
    np.testing.assert_allclose(g["w_i"][j], w["w_i"][j] + s_j * eta * (-mult * d - reg * w["w_i"][j]), rtol=2e-6, atol=1e-9)
    for k in ("v_u",):
        assert np.array_equal(g[k], plain[k])                       # the user's and the positive's steps are untouched
    assert np.array_equal(g["v_i"][0], plain["v_i"][0]) and g["w_i"][0] == plain["w_i"][0]
    # unit negative scales are the plain damped loop bit for bit
    h, _ = _fit(oracle, w, pairs1, csr1, epochs=1, pos_step=one_i, user_step=one_u, neg_step=one_i)
    assert all(np.array_equal(h[k], plain[k]) for k in plain)


def test_ll64_is_the_float_accumulators_sum_without_its_rounding(oracle):
    """`ll` restates the reference's float accumulator (rankfm/_rankfm.pyx:228, :270); `ll64` sums the same terms in double.  On a
    small problem they agree to float precision; the divergence at millions of rows (the accumulator's spacing passes the size of
    the small terms: ~0.5 % at 5 M rows) is documented in profiles/r02_notes.md."""
    U, I, N, F = 60, 40, 1500, 8
    pairs, csr = synthetic.make_interactions(U, I, N, seed=3)
    w = synthetic.init_weights(U, I, F, seed=4)
    _, out = _fit(oracle, w, pairs, csr, epochs=3)
    assert out["ll64"].shape == (3,) and np.all(out["ll64"] < 0)
    np.testing.assert_allclose(out["ll"], out["ll64"], rtol=2e-6)


def _fit_features(oracle, epochs=2, **kw):
    rng = np.random.default_rng(11)
    U, I, N, F, P, Q = 50, 30, 1200, 8, 5, 4
    pairs, csr = synthetic.make_interactions(U, I, N, seed=5)
    w = synthetic.init_weights(U, I, F, seed=6, n_user_features=P, n_item_features=Q)
    x_uf = (rng.random((U, P)) < 0.4).astype(np.float32)
    x_if = (rng.random((I, Q)) < 0.4).astype(np.float32)
    g = {k: v.copy() for k, v in w.items()}
    oracle.fit(pairs, np.ones(len(pairs), np.float32), csr.offsets, csr.items, x_uf, x_if, g["w_i"], g["w_if"], g["v_u"], g["v_i"], g["v_uf"],
               g["v_if"], 0.01, 0.1, 0.03, "constant", 0.25, 1, epochs, perms=None, rng_mode=oracle.RNG_COUNTER, seed=7, membership="binary", **kw)
    return w, g


def test_table_sampling_options_of_the_stand_in(oracle):
    """`table_every` / `table_head_every` / `table_head_rows` (analysis only: the dense tables trained on every k-th visited row, with
    another k for the first rows of the call's first epoch -- the sequential stand-in behind the opening launch of feature models,
    profiles/r03_notes.md section 7): the defaults are the reference bit for bit, a head at k = 1 over a whole one-epoch call is the
    reference, frozen tables stay at their initial values, and a head changes only what the tables see."""
    w, ref = _fit_features(oracle)
    _, a = _fit_features(oracle, table_every=0, table_head_every=0, table_head_rows=0)
    assert all(np.array_equal(ref[k], a[k]) for k in ref)
    _, one = _fit_features(oracle, epochs=1)
    _, b = _fit_features(oracle, epochs=1, table_every=7, table_head_every=1, table_head_rows=10**9)
    assert all(np.array_equal(one[k], b[k]) for k in one)
    _, frozen = _fit_features(oracle, table_every=-1)
    assert all(np.array_equal(frozen[k], w[k]) for k in ("v_uf", "v_if", "w_if")) and not np.array_equal(frozen["v_i"], w["v_i"])
    _, sparse = _fit_features(oracle, table_every=7)
    _, head = _fit_features(oracle, table_every=7, table_head_every=1, table_head_rows=300)
    assert not np.array_equal(sparse["v_uf"], ref["v_uf"]) and not np.array_equal(head["v_uf"], sparse["v_uf"])


def test_table_tail_of_the_stand_in_moves_the_tables_only(oracle):
    """`table_tail` (analysis only: table-only visits of random rows behind every epoch -- the sequential stand-in for a table trainer
    that outlasts the row loops, tools/table_quota_standin.py): none by default; after ONE epoch biases and factor rows are those of
    the plain call bit for bit and only the dense tables have moved; over two epochs the rows feel the moved tables."""
    _, ref = _fit_features(oracle, epochs=1)
    _, none = _fit_features(oracle, epochs=1, table_tail=0)
    assert all(np.array_equal(ref[k], none[k]) for k in ref)
    _, tail = _fit_features(oracle, epochs=1, table_tail=400)
    assert all(np.array_equal(ref[k], tail[k]) for k in ("w_i", "v_u", "v_i"))
    assert all(not np.array_equal(ref[k], tail[k]) for k in ("v_uf", "v_if", "w_if"))
    _, ref2 = _fit_features(oracle, epochs=2)
    _, tail2 = _fit_features(oracle, epochs=2, table_tail=400)
    assert not np.array_equal(ref2["v_i"], tail2["v_i"])


def test_table_quiet_rows_of_the_stand_in(oracle):
    """`table_quiet_rows` (analysis only: the last K visited rows of every epoch read the tables and do not train them -- a table trainer
    that finishes its quota before the row loops; tools/table_quota_standin.py): none by default, a quiet period as long as the epoch is
    frozen tables, and a shorter one changes what the tables see."""
    w, ref = _fit_features(oracle)
    _, none = _fit_features(oracle, table_quiet_rows=0)
    assert all(np.array_equal(ref[k], none[k]) for k in ref)
    _, frozen = _fit_features(oracle, table_every=-1)
    _, quiet_all = _fit_features(oracle, table_quiet_rows=10**9)
    assert all(np.array_equal(frozen[k], quiet_all[k]) for k in frozen)
    _, quiet = _fit_features(oracle, table_quiet_rows=300)
    assert not np.array_equal(quiet["v_uf"], ref["v_uf"]) and not np.array_equal(quiet["v_uf"], w["v_uf"])


def test_table_batch_of_the_stand_in(oracle):
    """`table_batch` (analysis only: the tables' updates scored on a snapshot of the tables taken every B table-training visits -- the
    engine's trainer applies batches of staged steps that were scored on one table state; tools/table_quota_standin.py): 0 and 1 are the
    reference bit for bit, a larger batch changes the tables (and, through them, the rows)."""
    _, ref = _fit_features(oracle)
    for b in (0, 1):
        _, same = _fit_features(oracle, table_batch=b)
        assert all(np.array_equal(ref[k], same[k]) for k in ref)
    _, stale = _fit_features(oracle, table_batch=16)
    assert not np.array_equal(stale["v_uf"], ref["v_uf"]) and not np.array_equal(stale["v_i"], ref["v_i"])
    assert all(np.isfinite(stale[k]).all() for k in stale)
