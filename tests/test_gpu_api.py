"""Public-API conformance on the GPU: the behaviours the reference's own test-suite checks (tests/test_rankfm.py, 23 cases
on a 3x6 toy set) re-expressed against rankfm_amd.RankFM on our own toy data, plus end-to-end parity with the reference
through the golden api_*.npz fixtures (fit -> predict -> recommend -> evaluation)."""
import numpy as np
import pandas as pd
import pytest

from conftest import WEIGHTS, load_golden

pytestmark = pytest.mark.gpu

INTX = pd.DataFrame([(10, 1), (10, 3), (10, 5), (20, 1), (20, 2), (20, 6), (30, 3), (30, 6), (30, 4)], columns=["user_id", "item_id"])
INTX_STR = pd.DataFrame({"user_id": INTX.user_id.map({10: "X", 20: "Y", 30: "Z"}), "item_id": INTX.item_id.map(lambda i: "ABCDEF"[i - 1])})
UF = pd.DataFrame([(10, 0, 1, 5, 3.1), (20, 1, 0, 6, 2.7), (30, 0, 0, 4, 1.6)], columns=["user_id", "b1", "b2", "n", "c"])
IF = pd.DataFrame([(1, 0, 1, 5, 3.1), (2, 1, 0, 6, 2.7), (3, 0, 0, 4, 1.6), (4, 1, 1, 3, 1.0), (5, 1, 0, 6, 0.3), (6, 0, 0, 0, 0.0)],
                  columns=["item_id", "b1", "b2", "n", "c"])
DISJOINT = pd.DataFrame([(10, 1), (10, 3), (10, 5), (20, 1), (20, 2), (20, 7), (40, 3), (40, 7), (40, 4)], columns=["user_id", "item_id"])
TRAIN_USERS, VALID_USERS = np.array([10, 20, 30]), np.array([10, 20, 40, 50])


def _model(**kw):
    from rankfm_amd import RankFM
    return RankFM(factors=2, **kw)


@pytest.mark.parametrize("intx, uf, itf", [(INTX, None, None), (INTX_STR, None, None), (INTX.values, None, None),
                                           (INTX, UF, None), (INTX, None, IF), (INTX, UF, IF), (INTX, UF.values, IF.values)])
@pytest.mark.parametrize("loss", ["bpr", "warp"])
def test_fit_accepts_the_reference_input_kinds(intx, uf, itf, loss, capsys):
    m = _model(loss=loss, max_samples=3).fit(intx, uf, itf, epochs=2, verbose=True)
    assert m.is_fit and capsys.readouterr().out.count("training epoch:") == 2
    for k in WEIGHTS:
        w = getattr(m, k)
        assert w.dtype == np.float32 and w.flags.c_contiguous and np.isfinite(w).all()
    assert m.v_u.shape == (3, 2) and m.v_i.shape == (6, 2) and m.w_i.shape == (6,)


def test_fit_partial_and_sample_weight():
    m = _model().fit(INTX, sample_weight=np.linspace(0.5, 1.5, 9).astype(np.float32), epochs=1)
    before = m.v_u.copy()
    m.fit_partial(INTX.iloc[:5], epochs=2)
    assert m.is_fit and not np.array_equal(before, m.v_u) and m.epochs_trained == 3
    assert len(m.interactions) == 5 and m.user_items[0].tolist() == [0, 2, 4]


def test_resumed_fit_with_a_fixed_seed_does_not_replay_order_and_draws():
    """fit_partial keys the counter RNG and the keyed order by the epochs already trained (rfm_fit_config.rng_epoch_offset)
    while the learning-rate schedule restarts at 0 like the reference's (rankfm/_rankfm.pyx:218-223): from identical weights,
    a second call must take a different trajectory than the first, and exactly the one of `_fit(..., rng_epoch_offset=1)`.
    Single-group mode, so both runs are deterministic."""
    from rankfm_amd import EngineOptions, RankFM, synthetic
    from rankfm_amd._rankfm import _fit
    d = synthetic.make_planted(300, 200, seed=1, mean_degree=30.0)
    train = pd.DataFrame(d["train"], columns=["u", "i"])
    eng = EngineOptions(seed=5, debug_flags=1)
    m = RankFM(factors=8, learning_schedule="invscaling", engine=eng)
    np.random.seed(0)
    m._init_all(train)
    init = {k: getattr(m, k).copy() for k in WEIGHTS}
    m.fit_partial(train, epochs=1)
    first = {k: getattr(m, k).copy() for k in WEIGHTS}
    assert m.epochs_trained == 1
    for k in WEIGHTS:
        getattr(m, k)[...] = init[k]
    m.fit_partial(train, epochs=1)                                   # same weights, same data, same seed: second call
    second = {k: getattr(m, k).copy() for k in WEIGHTS}
    assert not np.allclose(first["v_i"], second["v_i"], atol=1e-4)    # ... is not a replay of the first
    w = {k: v.copy() for k, v in init.items()}
    _fit(m.interactions, m.sample_weight, m.user_items, m.x_uf, m.x_if, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"],
         m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent, 1, 1, False, engine=eng, rng_epoch_offset=1)
    for k in WEIGHTS:
        np.testing.assert_allclose(second[k], w[k], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", ["bpr_feat_then_none", "warp_feat_then_feat"])
def test_fit_then_fit_partial_reproduces_the_reference(case):
    """resumed training against the reference's own numbers (tests/golden/partial_*.npz): fit(A, features) then
    fit_partial(B[, features]) in REFERENCE_ENGINE mode -- every call restarts MT19937 at 1492 and its learning-rate schedule at
    epoch 0 (rankfm/_rankfm.pyx:182, 218-223), numpy's shuffle stream runs on, the item sets are extended, and a second call
    without features switches the feature terms off while v_uf / v_if keep their values (rankfm/rankfm.py:286, 199, 211)"""
    from rankfm_amd import REFERENCE_ENGINE, RankFM
    z = load_golden("partial", case)
    fa = pd.DataFrame({"user_id": z["a_users"], "item_id": z["a_items"]})
    fb = pd.DataFrame({"user_id": z["b_users"], "item_id": z["b_items"]})
    uf = pd.concat([pd.DataFrame({"user_id": z["user_id"]}), pd.DataFrame(z["uf_vals"])], axis=1)
    itf = pd.concat([pd.DataFrame({"item_id": z["item_id"]}), pd.DataFrame(z["if_vals"])], axis=1)
    m = RankFM(factors=int(z["factors"]), loss=str(z["loss"]), max_samples=int(z["max_samples"]), learning_schedule="invscaling", sigma=0.5,
               engine=REFERENCE_ENGINE)
    np.random.seed(31)
    m.fit(fa, uf, itf, epochs=2)
    for k in WEIGHTS:
        np.testing.assert_allclose(getattr(m, k), z["first_" + k], rtol=1e-4, atol=2e-5, err_msg="first call: " + k)
    with_feat = bool(int(z["second_with_features"]))
    own_first = {k: getattr(m, k).copy() for k in ("v_uf", "v_if", "w_if")}
    m.fit_partial(fb, uf if with_feat else None, itf if with_feat else None, z["b_sw"], epochs=2)
    for k in WEIGHTS:
        np.testing.assert_allclose(getattr(m, k), z["second_" + k], rtol=1e-4, atol=2e-5, err_msg="second call: " + k)
    if not with_feat:                                         # feature terms off: the tables are not touched at all
        assert all(np.array_equal(getattr(m, k), own_first[k]) for k in own_first)


def test_predict_shapes_dtypes_and_cold_start():
    m = _model().fit(INTX)
    s = m.predict(INTX)
    assert s.shape == (9,) and s.dtype == np.float32 and not np.isnan(s).any()
    s = m.predict(DISJOINT, cold_start="nan")
    assert s.shape == (9,) and s.dtype == np.float32 and int(np.isnan(s).sum()) == 4
    s = m.predict(DISJOINT, cold_start="drop")
    assert s.shape == (5,) and not np.isnan(s).any()
    with pytest.raises(ValueError):
        m.predict(INTX, cold_start="zero")
    # the score is the reference's pointwise utility (rankfm/_rankfm.pyx:48-89) without features
    u, i = m.user_to_index.loc[20], m.item_to_index.loc[6]
    assert m.predict(np.array([[20, 6]]))[0] == pytest.approx(m.w_i[i] + m.v_u[u] @ m.v_i[i], abs=1e-6)


def test_recommend_frames_filtering_and_cold_start():
    m = _model().fit(INTX)
    recs = m.recommend(TRAIN_USERS, n_items=3)
    assert isinstance(recs, pd.DataFrame) and recs.shape == (3, 3) and np.array_equal(recs.index.values, TRAIN_USERS)
    assert recs.isin(INTX.item_id.values).all().all()
    assert all(len(set(r)) == 3 for r in recs.values)
    recs = m.recommend(TRAIN_USERS, n_items=3, filter_previous=True)
    long = recs.stack().reset_index().drop("level_1", axis=1)
    long.columns = ["user_id", "item_id"]
    assert pd.merge(INTX, long.astype({"item_id": INTX.item_id.dtype}), on=["user_id", "item_id"]).empty
    recs = m.recommend(VALID_USERS, n_items=3, cold_start="nan")
    assert recs.shape == (4, 3) and recs.loc[[40, 50]].isnull().all().all() and recs.dropna().isin(INTX.item_id.values).all().all()
    recs = m.recommend(VALID_USERS, n_items=3, cold_start="drop")
    assert recs.shape == (2, 3) and sorted(recs.index.values) == [10, 20]
    # ranking really is by descending utility
    top = m.recommend([30], n_items=6).values[0]
    scores = m.predict(np.stack([np.full(6, 30), top], 1))
    assert np.all(np.diff(scores) <= 1e-7)


def test_similar_items_and_users():
    m = _model().fit(INTX, UF, IF)
    sim = m.similar_items(1, n_items=3)
    assert sim.shape == (3,) and np.isin(sim, INTX.item_id.unique()).all() and 1 not in sim
    sim = m.similar_users(10, n_users=2)
    assert sim.shape == (2,) and np.isin(sim, INTX.user_id.unique()).all() and 10 not in sim
    with pytest.raises(AssertionError):
        m.similar_items(99)
    with pytest.raises(AssertionError):
        m.similar_users(9)


@pytest.mark.parametrize("n", [5, 40])
def test_similar_items_and_users_rank_like_the_reference_formula(n):
    """similar_items / similar_users on the device (rfm_similar_host: representation v + x . v_f, dot products, top-n with the
    query row excluded) against the reference's formula (rankfm/rankfm.py:418-426, 444-452) evaluated in numpy; n = 5 takes the
    single-pass selection, n = 40 the multi-round one"""
    from rankfm_amd import RankFM, synthetic
    d = synthetic.make_planted(400, 300, seed=3, mean_degree=40.0, n_tags=6)
    train = pd.DataFrame(d["train"], columns=["u", "i"])
    uf = pd.DataFrame(np.column_stack([np.arange(400), d["user_tags"]]))
    itf = pd.DataFrame(np.column_stack([np.arange(300), d["item_tags"]]))
    m = RankFM(factors=12, learning_rate=0.03)
    np.random.seed(1)
    m.fit(train, uf, itf, epochs=2)
    for kind, ids, v, x, vf, fn in (("item", m.item_id.values, m.v_i, m.x_if, m.v_if, m.similar_items),
                                    ("user", m.user_id.values, m.v_u, m.x_uf, m.v_uf, m.similar_users)):
        rep = v + x @ vf
        for q in (0, 17, len(ids) - 1):
            sims = rep @ rep[q]
            want = np.argsort(-sims, kind="stable")
            want = want[want != q][:n]
            got = fn(ids[q], n)
            assert got.shape == (n,) and ids[q] not in got
            # equal up to ties / last-ulp differences of the dot products: the similarity of every returned row must be as
            # large as the n-th best
            np.testing.assert_allclose(np.sort(sims[np.searchsorted(ids, got)])[::-1], sims[want], rtol=1e-5, atol=1e-6)


def test_save_load_round_trip_predicts_and_recommends_identically(tmp_path):
    """RankFM.save -> RankFM.load (the reference's weight layout + id maps in one .npz): the loaded model's predict / recommend /
    similar_items on the GPU are bit-equal to the unsaved model's, and training resumes from it"""
    from rankfm_amd import RankFM
    m = _model(loss="warp", max_samples=5).fit(INTX, UF, IF, epochs=3)
    path = str(tmp_path / "model.npz")
    m.save(path)
    r = RankFM.load(path)
    pairs = pd.DataFrame([(10, 1), (20, 6), (30, 2), (99, 1)], columns=["user_id", "item_id"])
    assert np.array_equal(m.predict(pairs), r.predict(pairs), equal_nan=True)
    a, b = m.recommend(VALID_USERS, n_items=3, filter_previous=True), r.recommend(VALID_USERS, n_items=3, filter_previous=True)
    assert a.equals(b)
    assert np.array_equal(m.similar_items(1, 3), r.similar_items(1, 3))
    for k in WEIGHTS:
        assert np.array_equal(getattr(m, k), getattr(r, k))
    r.fit_partial(INTX, UF, IF, epochs=1)
    assert r.epochs_trained == m.epochs_trained + 1 and not np.array_equal(r.v_u, m.v_u)


def _golden_frames(g):
    train = pd.DataFrame({"user_id": g["train_users"], "item_id": g["train_items"]})
    test = pd.DataFrame({"user_id": g["test_users"], "item_id": g["test_items"]})
    uf = itf = None
    if int(g["with_features"]):
        uf = pd.concat([pd.DataFrame({"user_id": g["uf_ids"]}), pd.DataFrame(g["uf_vals"])], axis=1)
        itf = pd.concat([pd.DataFrame({"item_id": g["if_ids"]}), pd.DataFrame(g["if_vals"])], axis=1)
    return train, test, uf, itf


@pytest.mark.parametrize("case", ["bpr_int_nofeat", "warp_str_feat"])
def test_end_to_end_drop_in_reproduces_the_reference(case):
    """np.random.seed(21); RankFM(...).fit(...) with REFERENCE_ENGINE must reproduce the reference's fit() -- same initial
    weights (numpy stream), same shuffles (continuing stream), same MT19937 negatives -- to fp32 tolerance, and then the
    reference's predict / recommend / hit_rate / MRR / DCG / precision / recall outputs on it."""
    from rankfm_amd import REFERENCE_ENGINE, RankFM, evaluation
    g = load_golden("api", case)
    train, test, uf, itf = _golden_frames(g)
    m = RankFM(factors=int(g["factors"]), loss=str(g["loss"]), max_samples=int(g["max_samples"]), learning_schedule="invscaling",
               engine=REFERENCE_ENGINE)
    np.random.seed(21)
    m.fit(train, uf, itf, g["train_sw"], epochs=int(g["epochs"]))
    for k in WEIGHTS:
        np.testing.assert_allclose(getattr(m, k), g["final_" + k], rtol=1e-4, atol=2e-5, err_msg=k)

    # scoring / ranking kernels on the reference's own final weights (isolates _predict/_recommend from training noise)
    for k in WEIGHTS:
        setattr(m, k, np.ascontiguousarray(g["final_" + k]))
    pred = pd.DataFrame({"user_id": g["pred_users"], "item_id": g["pred_items"]})
    s = m.predict(pred)
    assert np.array_equal(np.isnan(s), np.isnan(g["pred_scores"]))
    np.testing.assert_allclose(s[~np.isnan(s)], g["pred_scores"][~np.isnan(s)], rtol=1e-5, atol=1e-6)
    str_ids = bool(int(g["str_ids"]))
    for key, flt in (("rec_all", False), ("rec_new", True)):
        recs = m.recommend(list(g["rec_users"]), n_items=7, filter_previous=flt)
        got = recs.values.astype("U16") if str_ids else recs.values.astype(np.float64)
        want = g[key]
        if str_ids:
            assert np.array_equal(got, want), key
        else:
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]), key
    # similar_items / similar_users: the reference's own output (rankfm/rankfm.py:405-454) for a few query rows
    for q, want in zip(g["sim_item_queries"], g["sim_items"]):
        got = m.similar_items(q if str_ids else int(q), n_items=6)
        assert np.array_equal(np.asarray(got).astype(want.dtype), want), ("similar_items", q, got, want)
    for q, want in zip(g["sim_user_queries"], g["sim_users"]):
        got = m.similar_users(q if str_ids else int(q), n_users=6)
        assert np.array_equal(np.asarray(got).astype(want.dtype), want), ("similar_users", q, got, want)
    assert evaluation.hit_rate(m, test, k=7) == pytest.approx(float(g["hit_rate"]), abs=1e-12)
    assert evaluation.hit_rate(m, test, k=7, filter_previous=True) == pytest.approx(float(g["hit_rate_new"]), abs=1e-12)
    assert evaluation.reciprocal_rank(m, test, k=7) == pytest.approx(float(g["reciprocal_rank"]), abs=1e-9)
    assert evaluation.discounted_cumulative_gain(m, test, k=7) == pytest.approx(float(g["dcg"]), abs=1e-9)
    assert evaluation.precision(m, test, k=7) == pytest.approx(float(g["precision"]), abs=1e-9)
    assert evaluation.recall(m, test, k=7) == pytest.approx(float(g["recall"]), abs=1e-9)


def test_predict_and_recommend_match_oracle_at_scale(oracle):
    """scoring kernels vs the CPU oracle on a mid-size model with features and NaN (cold-start) entries"""
    from rankfm_amd import synthetic
    from rankfm_amd._rankfm import _predict, _recommend
    U, I, F, P, Q = 1500, 900, 24, 5, 6
    rng = np.random.default_rng(3)
    pairs, csr = synthetic.make_interactions(U, I, 40000, seed=1)
    w = synthetic.init_weights(U, I, F, P, Q, sigma=0.5, seed=2)
    w["w_i"] = rng.normal(0, 0.3, I).astype(np.float32)
    w["w_if"] = rng.normal(0, 0.3, Q).astype(np.float32)
    x_uf, x_if = synthetic.make_features(U, P, 4), synthetic.make_features(I, Q, 5)
    args = (x_uf, x_if, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])
    idx = np.stack([rng.integers(0, U, 5000), rng.integers(0, I, 5000)], 1).astype(np.float32)
    idx[::97, 0] = np.nan
    idx[::89, 1] = np.nan
    s, so = _predict(idx, *args), oracle.predict(idx, *args)
    assert np.array_equal(np.isnan(s), np.isnan(so))
    np.testing.assert_allclose(s[~np.isnan(s)], so[~np.isnan(so)], rtol=1e-5, atol=2e-5)
    users = rng.integers(0, U, 64).astype(np.float32)
    users[5] = np.nan
    for flt in (False, True):
        for n_rec in (10, 25):              # one-pass top-n (<= 16) and the multi-pass kernel
            rec = _recommend(users, csr, n_rec, flt, *args)
            ro = oracle.recommend(users, csr.offsets, csr.items, n_rec, flt, *args)
            assert np.isnan(rec[5]).all() and np.isnan(ro[5]).all()
            same = rec[~np.isnan(users)] == ro[~np.isnan(users)]
            assert same.mean() > 0.995      # identical up to fp32 near-ties in the ranking


def test_resident_session_serves_like_the_host_entry_points():
    """`DeviceSession.predict / recommend` (rfm_predict_device / rfm_recommend_device on the session's resident model, feature matrices
    and item lists: no uploads) return what `_predict` / `_recommend` (the host entry points, staged through the per-device serving
    arena) return on the same weights -- also when the host entry points are called again and again (the arena is reused and regrown)."""
    from rankfm_amd import synthetic
    from rankfm_amd._rankfm import _predict, _recommend
    from rankfm_amd.engine import DeviceSession
    U, I, F, P, Q = 1500, 900, 24, 5, 6
    rng = np.random.default_rng(3)
    pairs, csr = synthetic.make_interactions(U, I, 40000, seed=1)
    w = synthetic.init_weights(U, I, F, P, Q, sigma=0.5, seed=2)
    w["w_i"] = rng.normal(0, 0.3, I).astype(np.float32)
    w["w_if"] = rng.normal(0, 0.3, Q).astype(np.float32)
    x_uf, x_if = synthetic.make_features(U, P, 4), synthetic.make_features(I, Q, 5)
    args = (x_uf, x_if, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])
    s = DeviceSession(pairs, np.ones(len(pairs), np.float32), csr.offsets, csr.items, x_uf, x_if, w)
    idx = np.stack([rng.integers(0, U, 5000), rng.integers(0, I, 5000)], 1).astype(np.float32)
    idx[::97, 0] = np.nan
    idx[::89, 1] = np.nan
    host = _predict(idx, *args)
    np.testing.assert_array_equal(s.predict(idx), host)
    users = rng.integers(0, U, 300).astype(np.float32)
    users[5] = np.nan
    for flt in (False, True):
        for n_rec in (10, 25):
            host = _recommend(users, csr, n_rec, flt, *args)
            np.testing.assert_array_equal(s.recommend(users, n_rec, flt), host)
            np.testing.assert_array_equal(_recommend(users[:7], csr, n_rec, flt, *args), host[:7])      # (a smaller call on the kept arena)
    assert np.isnan(s.recommend(users, 10, True)[5]).all()
    with pytest.raises(ValueError):
        s.recommend(users, 0)
    # ... and after training a little, the session serves the weights it trained
    s.run(epochs=1)
    trained = s.weights_to_host()
    args2 = (x_uf, x_if, trained["w_i"], trained["w_if"], trained["v_u"], trained["v_i"], trained["v_uf"], trained["v_if"])
    np.testing.assert_array_equal(s.recommend(users, 10, True), _recommend(users, csr, 10, True, *args2))


@pytest.mark.parametrize("n_items", [50, 77, 100, 147, 192, 250])
@pytest.mark.parametrize("factors, item_features", [(8, 0), (24, 3), (50, 0), (64, 4)])
def test_recommend_for_a_handful_of_users_on_small_catalogues(oracle, n_items, factors, item_features):
    """with 1 - 10 users and 42 - 256 items the matrix-free path's per-block arrays, each rounded up to 64 floats, did
    not fit the `scores` area they are carved from and ran into `veff` -- recommendations were silently wrong (item 2's effective
    factors replaced by the user's row, `filter_previous` zeroing veff[0][0..3]).  Every (users, items, padded k) of the advisor's
    enumeration against the oracle, with and without the filter; the calls are in ascending user count so that each one runs on the
    arena the previous, smaller one left behind."""
    from rankfm_amd import synthetic
    from rankfm_amd._rankfm import _recommend
    U, I, F, P, Q = 40, n_items, factors, 1, max(item_features, 1)
    rng = np.random.default_rng(n_items * 131 + factors)
    pairs, csr = synthetic.make_interactions(U, I, 300, seed=3)
    w = synthetic.init_weights(U, I, F, 0, item_features, sigma=0.5, seed=2)
    w["w_i"] = rng.normal(0, 0.3, I).astype(np.float32)
    x_uf = np.zeros((U, 1), np.float32)
    x_if = synthetic.make_features(I, Q, 5) if item_features else np.zeros((I, 1), np.float32)
    if item_features:
        w["w_if"] = rng.normal(0, 0.3, Q).astype(np.float32)
    args = (x_uf, x_if, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])
    for n_users in (1, 2, 3, 5, 8, 10):
        users = rng.choice(U, n_users, replace=False).astype(np.float32)
        for flt in (False, True):
            rec = _recommend(users, csr, 10, flt, *args)
            ro = oracle.recommend(users, csr.offsets, csr.items, 10, flt, *args)
            assert (rec == ro).mean() > 0.97, (n_users, flt, rec, ro)      # (identical up to fp32 near-ties in the ranking)
            if flt:
                for r, u in zip(rec, users.astype(int)):
                    assert not set(r.astype(int)) & set(csr.items[csr.offsets[u]:csr.offsets[u + 1]].tolist())


def test_matrix_free_and_matrix_recommend_paths_agree():
    """the matrix-free path picks its candidate blocks from the matrix cores' accumulators and ranks them from fp32 FMA
    chains; the matrix path (lists longer than 16) ranks the matrix cores' scores themselves.  The same model through both: the first
    ten of a top-17 (matrix path) are the top-10 (matrix-free path), item for item, on a catalogue with many near-ties."""
    from rankfm_amd import synthetic
    from rankfm_amd._rankfm import _recommend
    U, I, F = 700, 3000, 50
    rng = np.random.default_rng(11)
    pairs, csr = synthetic.make_interactions(U, I, 30000, seed=1)
    w = synthetic.init_weights(U, I, F, 0, 0, sigma=0.05, seed=2)      # (small factors: scores 1e-2 apart and closer)
    w["w_i"] = (rng.integers(0, 40, I) * 0.01).astype(np.float32)
    args = (np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])
    users = np.arange(U, dtype=np.float32)
    for flt in (False, True):
        short, long_ = _recommend(users, csr, 10, flt, *args), _recommend(users, csr, 17, flt, *args)
        assert (short == long_[:, :10]).mean() > 0.999
