"""Edge cases of the training path on the GPU, each against the sequential oracle in the engine's own order (tight
tolerance, single-group mode of the production kernel) and -- where it makes sense -- as a full Hogwild run."""
import numpy as np
import pytest

from conftest import WEIGHTS
from test_gpu_parity import SERIAL_ATOL, SERIAL_RTOL, _oracle_in_engine_order

pytestmark = pytest.mark.gpu


def _run(pairs, csr, sw, F, max_samples, epochs, seed, engine_kw, oracle, sigma=0.1, schedule="constant"):
    from rankfm_amd import EngineOptions, synthetic
    from rankfm_amd._rankfm import _fit
    U, I = len(csr.offsets) - 1, int(max(pairs[:, 1].max() + 1, 2))
    I = max(I, int(csr.items.max()) + 1 if len(csr.items) else I)
    w0 = synthetic.init_weights(U, I, F, sigma=sigma, seed=seed)
    x_uf, x_if = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
    g = {k: v.copy() for k, v in w0.items()}
    rep = {}
    _fit(pairs, sw, csr, x_uf, x_if, g["w_i"], g["w_if"], g["v_u"], g["v_i"], g["v_uf"], g["v_if"], 0.01, 0.1, 0.1, schedule, 0.25,
         max_samples, epochs, False, engine=EngineOptions(seed=seed, **engine_kw), report=rep)
    o, out = _oracle_in_engine_order(oracle, (pairs, csr, sw, x_uf, x_if, None), w0, max_samples, epochs, seed, schedule=schedule,
                                     geometry=rep["geometry"])
    return g, rep, o, out


def _csr(pairs, U):
    from rankfm_amd import UserItemsCSR
    return UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], U)


@pytest.mark.parametrize("F", [1, 2, 64])
def test_tiny_problems(oracle, F):
    """one interaction; two items only; every user a single row"""
    for pairs, U in ((np.array([[0, 1]], np.int32), 1),
                     (np.array([[0, 0], [1, 1], [2, 0]], np.int32), 3),
                     (np.array([[u, (3 * u) % 7] for u in range(40)], np.int32), 40)):
        csr = _csr(pairs, U)
        sw = np.ones(len(pairs), np.float32)
        for flags in (1, 0):          # sequential single group, then plain Hogwild (a handful of rows: still near-sequential)
            g, rep, o, out = _run(pairs, csr, sw, F, 1, 3, 4, dict(debug_flags=flags), oracle)
            # Hogwild on 1-40 rows: the concurrency cap min(U, I)/3 leaves one or two groups working -> still close
            tol = dict(rtol=SERIAL_RTOL, atol=SERIAL_ATOL) if flags else dict(rtol=0.05, atol=5e-3)
            for k in WEIGHTS:
                np.testing.assert_allclose(g[k], o[k], err_msg="%s F=%d flags=%d" % (k, F, flags), **tol)


def test_independent_rows_in_hogwild_equal_the_sequential_program(oracle):
    """Rows that share neither a user nor an item, negatives drawn from a catalogue large enough that no drawn item is another row's:
    the order of execution cannot matter, so plain Hogwild must land on the sequential oracle's weights at the single-group tolerance.
    This pins the WARP kernel's deferred row updates (rfm_sgd_warp.hpp, DEFER: a finished row's atomics are issued behind the group's NEXT
    gathers, the last ones behind the loop) -- an update issued late, twice or never shows here -- and the BPR kernel's likewise."""
    U, I, F = 64, 200_000, 64
    pairs = np.array([[u, 1000 + 997 * u] for u in range(U - 1)] + [[U - 1, I - 1]], np.int32)      # (the last item id sets the catalogue's size)
    csr = _csr(pairs, U)
    sw = np.ones(U, np.float32)
    for max_samples in (10, 1):
        g, rep, o, out = _run(pairs, csr, sw, F, max_samples, 2, 21, dict(debug_flags=0), oracle)
        for k in WEIGHTS:
            np.testing.assert_allclose(g[k], o[k], rtol=SERIAL_RTOL, atol=SERIAL_ATOL, err_msg="%s max_samples=%d" % (k, max_samples))
        assert len(rep["log_likelihood"]) == 2


def test_duplicate_interactions_and_zero_weights(oracle):
    """the reference keeps repeated (user, item) rows -- each is a step (rankfm/rankfm.py:174 sorts, does not dedupe) -- and a
    zero sample weight leaves only the L2 shrink"""
    rng = np.random.default_rng(0)
    base = np.stack([rng.integers(0, 30, 300), rng.integers(0, 50, 300)], 1).astype(np.int32)
    pairs = np.concatenate([base, base[:120], base[:40]])            # up to three copies of a pair
    sw = np.ones(len(pairs), np.float32)
    sw[rng.random(len(pairs)) < 0.2] = 0.0
    # copies of one pair must share a weight for a reproducible run: their CSR slots are interchangeable
    key = pairs[:, 0].astype(np.int64) * 1000 + pairs[:, 1]
    _, first = np.unique(key, return_index=True)
    sw = sw[first][np.searchsorted(key[first], key)]
    csr = _csr(pairs, 30)
    assert len(csr.items) == len(pairs)
    g, rep, o, out = _run(pairs, csr, sw, 16, 4, 2, 7, dict(debug_flags=1), oracle, sigma=0.4)
    for k in WEIGHTS:
        np.testing.assert_allclose(g[k], o[k], rtol=SERIAL_RTOL, atol=SERIAL_ATOL, err_msg=k)


def test_heavy_user_spans_many_segments(oracle):
    """one user holds 80 % of the catalogue (60+ segments, sampler rejects 4 draws in 5); the rest are light"""
    rng = np.random.default_rng(1)
    I, U = 400, 200
    heavy = np.stack([np.zeros(320, np.int64), rng.permutation(I)[:320]], 1)
    light = np.stack([rng.integers(1, U, 3000), rng.integers(0, I, 3000)], 1)
    light = np.unique(light, axis=0)
    pairs = np.concatenate([heavy, light]).astype(np.int32)
    pairs = pairs[rng.permutation(len(pairs))]
    csr = _csr(pairs, U)
    sw = np.ones(len(pairs), np.float32)
    g, rep, o, out = _run(pairs, csr, sw, 32, 1, 2, 11, dict(debug_flags=1), oracle)
    for k in WEIGHTS:
        np.testing.assert_allclose(g[k], o[k], rtol=SERIAL_RTOL, atol=SERIAL_ATOL, err_msg=k)
    # full Hogwild on the same data stays finite and close (the heavy user's concurrent segments are damped)
    g, rep, o, out = _run(pairs, csr, sw, 32, 1, 3, 11, {}, oracle)
    assert all(np.isfinite(g[k]).all() for k in WEIGHTS)
    np.testing.assert_allclose(rep["log_likelihood"], out["ll64"], rtol=0.03)
    assert abs(np.linalg.norm(g["v_i"]) - np.linalg.norm(o["v_i"])) < 0.03 * np.linalg.norm(o["v_i"])


def test_invscaling_schedule_and_resumed_epochs(oracle):
    """eta = lr / (epoch+1)^0.25 (rankfm/_rankfm.pyx:222-223); a second call continues the counter streams at epoch_begin"""
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    pairs, csr = synthetic.make_interactions(150, 120, 4000, seed=5)
    sw = np.ones(len(pairs), np.float32)
    g, rep, o, out = _run(pairs, csr, sw, 20, 1, 3, 13, dict(debug_flags=1), oracle, schedule="invscaling")
    for k in WEIGHTS:
        np.testing.assert_allclose(g[k], o[k], rtol=SERIAL_RTOL, atol=SERIAL_ATOL, err_msg=k)
    w0 = synthetic.init_weights(150, 120, 20, seed=13)
    z_u, z_i = np.zeros((150, 1), np.float32), np.zeros((120, 1), np.float32)
    one = DeviceSession(pairs, sw, csr.offsets, csr.items, z_u, z_i, w0, learning_schedule="invscaling", seed=13, debug_flags=1)
    one.run(epochs=3)
    two = DeviceSession(pairs, sw, csr.offsets, csr.items, z_u, z_i, w0, learning_schedule="invscaling", seed=13, debug_flags=1)
    two.run(epochs=1)
    two.run(epochs=2, epoch_begin=1)
    a, b = one.weights_to_host(), two.weights_to_host()
    for k in WEIGHTS:
        assert np.array_equal(a[k], b[k]), k          # sequential single-group mode is bit-reproducible
    for k in WEIGHTS:
        np.testing.assert_allclose(a[k], g[k], rtol=SERIAL_RTOL, atol=SERIAL_ATOL)


def test_max_samples_beyond_catalogue_is_reported_not_silent(oracle):
    """(I-1) // sampled == 0 makes the reference's multiplier log(0) = -inf: weights go non-finite and
    the epoch-end check must say so"""
    from rankfm_amd import EngineOptions, UserItemsCSR, synthetic
    from rankfm_amd._rankfm import _fit
    pairs = np.array([[u, i] for u in range(3) for i in range(3)], np.int32)      # every user holds items 0..2
    csr = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], 3)
    w = synthetic.init_weights(3, 5, 4, sigma=2.0, seed=1)
    w["w_i"][:] = np.array([9, 9, 9, 0, 0], np.float32)          # positives far ahead of both negatives: no draw violates the margin
    w["v_u"][:] = 0
    z_u, z_i = np.zeros((3, 1), np.float32), np.zeros((5, 1), np.float32)
    with pytest.raises(AssertionError, match="not finite"):
        _fit(pairs, np.ones(len(pairs), np.float32), csr, z_u, z_i, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"], 0.01, 0.1,
             0.1, "constant", 0.25, 6, 1, False, engine=EngineOptions(mode="serial", seed=1))


def test_epoch_parts_compose_to_the_whole_epoch():
    """epoch_parts = n runs the k-th slice of the visiting order (multi-GPU callers exchange deltas between slices): the n slices in
    turn must equal one whole-epoch run bit for bit in the sequential single-group mode"""
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    pairs, csr = synthetic.make_interactions(200, 150, 6000, seed=8)
    sw = np.ones(len(pairs), np.float32)
    w0 = synthetic.init_weights(200, 150, 16, seed=2)
    z_u, z_i = np.zeros((200, 1), np.float32), np.zeros((150, 1), np.float32)
    whole = DeviceSession(pairs, sw, csr.offsets, csr.items, z_u, z_i, w0, seed=3, debug_flags=1)
    r0 = whole.run(epochs=1)
    sliced = DeviceSession(pairs, sw, csr.offsets, csr.items, z_u, z_i, w0, seed=3, debug_flags=1)
    ll, draws = 0.0, 0
    for k in range(5):
        r = sliced.run(epochs=1, part=(k, 5))
        ll += r["log_likelihood"][0]
        draws += r["n_draws"][0]
    a, b = whole.weights_to_host(), sliced.weights_to_host()
    for k in WEIGHTS:
        assert np.array_equal(a[k], b[k]), k
    assert draws == r0["n_draws"][0] == len(pairs) and ll == pytest.approx(r0["log_likelihood"][0], rel=1e-9)
    with pytest.raises(ValueError):
        sliced.run(epochs=1, part=(5, 5))


def test_duplicate_heavy_user_with_more_rows_than_items_trains(oracle):
    """A user with 60 rows over 20 DISTINCT items of a 50-item catalogue is not saturated: the reference's rejection sampler
    terminates on such data (rankfm/_rankfm.pyx:250-253) and so must the engine -- the saturation guard counts distinct
    items, not rows.  A user who really holds every item (with duplicates on top) is still refused."""
    rng = np.random.default_rng(3)
    heavy = np.stack([np.zeros(60, np.int64), rng.integers(0, 20, 60)], 1)
    heavy[:20, 1] = np.arange(20)                                       # all 20 items present
    rest = np.stack([rng.integers(1, 25, 400), rng.integers(0, 50, 400)], 1)
    pairs = np.concatenate([heavy, rest]).astype(np.int32)
    key = pairs[:, 0].astype(np.int64) * 1000 + pairs[:, 1]
    sw = np.ones(len(pairs), np.float32)
    csr = _csr(pairs, 25)
    assert csr.offsets[1] - csr.offsets[0] == 60 and len(np.unique(csr[0])) == 20
    g, rep, o, out = _run(pairs, csr, sw, 16, 1, 2, 5, dict(debug_flags=1), oracle)
    for k in WEIGHTS:
        np.testing.assert_allclose(g[k], o[k], rtol=SERIAL_RTOL, atol=SERIAL_ATOL, err_msg=k)
    g, rep, o, out = _run(pairs, csr, sw, 16, 1, 2, 5, {}, oracle)       # and as a Hogwild run
    assert np.isfinite(rep["log_likelihood"]).all() and rep["n_draws"][0] == len(pairs)
    full = np.concatenate([np.stack([np.zeros(70, np.int64), np.concatenate([np.arange(50), rng.integers(0, 50, 20)])], 1), rest]).astype(np.int32)
    with pytest.raises(ValueError, match="every item"):
        _run(full, _csr(full, 25), np.ones(len(full), np.float32), 16, 1, 1, 5, {}, oracle)


def test_recommend_for_a_user_who_has_seen_nearly_everything(oracle):
    """top-n with `filter_previous` when fewer than n_rec of the 4096 score-row segments of the threshold selection
    (rfm_infer.hip, topn_select_kernel) hold an unseen item: the selection must fall back to ranking every unseen item instead
    of cutting at the last segment maximum it found (which dropped valid items and returned NaN slots)."""
    from rankfm_amd import synthetic
    from rankfm_amd._rankfm import UserItemsCSR, _recommend
    U, I, F, n_rec = 3, 5000, 8, 10                           # 5000 items -> 2500 segments of two elements (i, i + 2500)
    w = synthetic.init_weights(U, I, F, sigma=0.5, seed=4)
    w["w_i"] = np.random.default_rng(1).normal(0, 0.3, I).astype(np.float32)
    unseen = np.array([0, 2500, 1, 2501, 2, 2502, 3, 2503, 4, 2504, 5, 2505])      # 12 items in SIX segments
    lists = [np.setdiff1d(np.arange(I), unseen).astype(np.int32), np.arange(0, I, 7, dtype=np.int32), np.zeros(0, np.int32)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    csr = UserItemsCSR(off, np.concatenate(lists))
    args = (np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])
    users = np.array([0, 1, 2], np.float32)
    rec = _recommend(users, csr, n_rec, True, *args)
    ro = oracle.recommend(users, csr.offsets, csr.items, n_rec, True, *args)
    assert not np.isnan(rec).any()
    assert set(rec[0].astype(int)) <= set(unseen.tolist()) and len(set(rec[0].astype(int))) == n_rec
    assert np.array_equal(rec, ro)


def test_kept_engine_layout_is_the_same_training_bit_for_bit_on_one_group():
    """rfm_fit_config.keep_layout (DeviceSession(keep_layout=True)): between calls the item-side weights stay in the engine's working layout
    inside the workspace -- here the padded biases; the caller's arrays are stale until they are exported.  One row group is a sequential
    program, so a session that keeps the layout over calls of DIFFERENT lengths (the workspace is re-allocated in between: the session
    exports first) must land bit for bit on the weights of a session that converts on every call, and `predict` on the live session must
    serve the current weights."""
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    U, I, F = 200, 150, 16
    pairs, csr = synthetic.make_interactions(U, I, 6000, seed=8)
    sw = np.ones(len(pairs), np.float32)
    w0 = synthetic.init_weights(U, I, F, seed=2)
    z_u, z_i = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
    plain = DeviceSession(pairs, sw, csr.offsets, csr.items, z_u, z_i, w0, seed=3, debug_flags=1)
    kept = DeviceSession(pairs, sw, csr.offsets, csr.items, z_u, z_i, w0, seed=3, debug_flags=1, keep_layout=True)
    e = 0
    for n in (1, 1, 3, 2, 9):
        ra, rb = plain.run(epochs=n, epoch_begin=e), kept.run(epochs=n, epoch_begin=e)
        assert kept._layout_token != 0                                        # (the call did keep it)
        np.testing.assert_array_equal(ra["log_likelihood"], rb["log_likelihood"])
        e += n
    stale = kept.weights["w_i"].cpu().numpy()
    a, b = plain.weights_to_host(), kept.weights_to_host()
    assert not np.array_equal(stale, b["w_i"])                                # (the caller's array WAS stale, and the export refreshed it)
    for k in WEIGHTS:
        assert np.array_equal(a[k], b[k]), k
    idx = np.stack([np.arange(50) % U, np.arange(50) % I], 1).astype(np.float32)
    kept.run(epochs=1, epoch_begin=e)
    plain.run(epochs=1, epoch_begin=e)
    np.testing.assert_array_equal(kept.predict(idx), plain.predict(idx))
    # a call that is refused before anything runs leaves plan and layout as they were
    with pytest.raises(ValueError):
        kept.run(epochs=1, part=(5, 5))
    assert kept._layout_token != 0
    np.testing.assert_array_equal(kept.weights_to_host()["v_i"], plain.weights_to_host()["v_i"])


def test_kept_engine_layout_loses_no_update_at_config2_scale(c2_problem):
    """The production kernel of config 2 on a kept layout: segment-major factor rows, padded biases and the hot-row bins stay in the
    workspace over four calls (1 + 1 + 2 + 5 epochs; the last one re-allocates the workspace); what the workgroups leave in the bins when a
    launch ends is folded in by the epoch tail and by the export.  With alpha = 0 the column sums of v_i and the sum of w_i are invariants of
    any interleaving iff no update is lost (tests/test_gpu_parity.py::test_hogwild_conserves_item_factor_sums) -- also across the layout's
    round trips."""
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    U, I, N, F, pairs, csr = c2_problem
    w = synthetic.init_weights(U, I, F, seed=1492)
    before = w["v_i"].astype(np.float64).sum(axis=0)
    sess = DeviceSession(pairs, np.ones(N, np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w,
                         alpha=0.0, beta=0.0, max_samples=1, seed=1492, hogwild_damping=1e9, keep_layout=True)
    lls, e = [], 0
    for n in (1, 1, 2, 5):
        rep = sess.run(epochs=n, epoch_begin=e)
        assert sess._layout_token & 4 and sess._layout_token & 2 and (sess._layout_token >> 8) > 0      # segment-major rows, padded biases, hot slots
        lls += list(rep["log_likelihood"])
        e += n
    h = sess.weights_to_host()
    after = h["v_i"].astype(np.float64).sum(axis=0)
    moved = np.abs(h["v_i"] - w["v_i"]).astype(np.float64).sum(axis=0)
    assert np.all(np.abs(after - before) <= 2e-5 * moved + 1e-3), (np.abs(after - before).max(), moved.min())
    assert abs(float(h["w_i"].astype(np.float64).sum())) <= 2e-5 * float(np.abs(h["w_i"]).astype(np.float64).sum()) + 1e-3
    assert all(np.isfinite(lls)) and lls[-1] > lls[0]
    # ... and the exported weights are what the engine goes on training from: one more epoch on a fresh session started from them tracks
    # the kept session's next epoch
    rep_k = sess.run(epochs=1, epoch_begin=e)
    fresh = DeviceSession(pairs, np.ones(N, np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), h,
                          alpha=0.0, beta=0.0, max_samples=1, seed=1492, hogwild_damping=1e9)
    rep_f = fresh.run(epochs=1, epoch_begin=e)
    assert rep_f["log_likelihood"][0] == pytest.approx(rep_k["log_likelihood"][0], rel=2e-3)


def test_kept_engine_layout_with_a_feature_model_trains_the_same_model():
    """keep_layout on a model with features (padded biases + hot-row bins kept; the feature tables themselves are never re-laid): the
    engine's Hogwild run is not bit-reproducible, so a kept session is held to a converting session statistically -- five one-epoch calls
    each from the same weights: log-likelihood per epoch within 1.5 %, the row norms within 1 % -- and its exported weights must be what
    `predict` serves."""
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    U, I, F, P, Q = 6000, 4000, 32, 8, 8
    pairs, csr = synthetic.make_interactions(U, I, 400_000, seed=5)
    sw = np.ones(len(pairs), np.float32)
    w0 = synthetic.init_weights(U, I, F, P, Q, seed=2)
    x_uf, x_if = synthetic.make_features(U, P, 3), synthetic.make_features(I, Q, 4)
    lls = {}
    sessions = {}
    for keep in (False, True):
        s = DeviceSession(pairs, sw, csr.offsets, csr.items, x_uf, x_if, w0, seed=3, learning_rate=0.03, keep_layout=keep)
        lls[keep] = np.concatenate([s.run(epochs=1, epoch_begin=e)["log_likelihood"] for e in range(5)])
        assert (s._layout_token != 0) == keep
        sessions[keep] = s
    np.testing.assert_allclose(lls[True], lls[False], rtol=0.015)
    a, b = sessions[False].weights_to_host(), sessions[True].weights_to_host()
    for k in ("w_i", "v_u", "v_i"):
        assert abs(np.linalg.norm(a[k]) / np.linalg.norm(b[k]) - 1.0) <= 0.01, k
    assert all(np.isfinite(b[k]).all() for k in b)
    idx = np.stack([np.arange(64) % U, np.arange(64) % I], 1).astype(np.float32)
    from rankfm_amd._rankfm import _predict
    np.testing.assert_allclose(sessions[True].predict(idx), _predict(idx, x_uf, x_if, b["w_i"], b["w_if"], b["v_u"], b["v_i"], b["v_uf"], b["v_if"]),
                               rtol=1e-6, atol=1e-6)
