"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    oracle/build_ref.sh && python tests/golden/make_golden.py

The reference (etlundquist/rankfm @ /root/reference) is imported, never copied: its compiled
`_fit/_predict/_recommend` come from oracle/_ref/ and its Python from /root/reference.  Only DATA
is written here: inputs, initial weights, the per-epoch permutations numpy produced, final weights,
log-likelihoods, predict/recommend/hit_rate outputs.  The fixtures are what pins the CPU oracle
(tests/test_oracle_golden.py) and the HIP engine (tests/test_gpu_parity.py) to the reference.

Capture protocol (verified to reproduce the reference bit for bit):
  1. np.random.seed(S_init); RankFM._init_all(...)                -> inputs + initial weights
  2. np.random.seed(S_shuf); _fit(...)                            -> final weights
  3. perms replayed with np.random.seed(S_shuf) + cumulative np.random.shuffle of one arange(N)
     (rankfm/_rankfm.pyx:197,227)
  4. negatives come from MT19937 seeded 1492 inside _fit (rankfm/_rankfm.pyx:182)
"""
import contextlib
import io
import os
import re
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_loader  # noqa: E402

RankFM, ref_ext, ref_eval = ref_loader.load_reference()
WEIGHTS = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")


def make_interactions(rng, U, I, N, zipf=True):
    """random (user, item) pairs, every user and item present at least once, no user saturated"""
    users = np.concatenate([np.arange(U), rng.integers(0, U, N - U)])
    if zipf:
        pr = 1.0 / np.arange(1, I + 1)
        pr /= pr.sum()
        items = rng.choice(I, size=N, p=pr)
    else:
        items = rng.integers(0, I, N)
    items[:I] = np.arange(I)          # every item observed -> item index == item id
    pairs = np.stack([users, items], 1)
    rng.shuffle(pairs)
    return pairs.astype(np.int64)


def csr_of(user_items, U):
    off = np.zeros(U + 1, dtype=np.int64)
    for u in range(U):
        off[u + 1] = off[u] + len(user_items[u])
    items = np.concatenate([np.asarray(user_items[u], dtype=np.int32) for u in range(U)])
    return off, items


def replay_perms(seed, N, epochs):
    np.random.seed(seed)
    idx = np.arange(N, dtype=np.int32)
    out = []
    for _ in range(epochs):
        np.random.shuffle(idx)
        out.append(idx.copy())
    return np.stack(out)


def run_fit(m, max_samples, epochs, w, s_shuf, verbose=True):
    """call the reference's private _fit on copies of the initial weights `w`"""
    cur = {k: w[k].copy() for k in WEIGHTS}
    np.random.seed(s_shuf)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref_ext._fit(m.interactions, m.sample_weight, m.user_items, m.x_uf, m.x_if,
                     cur["w_i"], cur["w_if"], cur["v_u"], cur["v_i"], cur["v_uf"], cur["v_if"],
                     m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent,
                     max_samples, epochs, verbose)
    printed = [float(x) for x in re.findall(r"log likelihood: (-?[0-9.eE+-]+)", buf.getvalue())]
    return cur, np.array(printed, dtype=np.float64)


def penalty(m, w):
    return float(ref_ext.reg_penalty(m.alpha, m.beta, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"]))


def fit_case(name, *, U, I, N, F, loss, max_samples=10, user_feats=0, item_feats=0, sample_weights=False,
             schedule="constant", sigma=0.1, lr=0.1, epochs=3, data_seed=0, s_init=11, s_shuf=12, zipf=True,
             alpha=0.01, beta=0.1, exponent=0.25, feat_kind="tags"):
    rng = np.random.default_rng(data_seed)
    pairs = make_interactions(rng, U, I, N, zipf)
    uf = itf = None
    if user_feats:
        vals = (rng.random((U, user_feats)) < 0.35).astype(np.float32) if feat_kind == "tags" else rng.normal(size=(U, user_feats)).astype(np.float32)
        uf = pd.DataFrame(np.column_stack([np.arange(U), vals]))
    if item_feats:
        vals = (rng.random((I, item_feats)) < 0.35).astype(np.float32) if feat_kind == "tags" else rng.normal(size=(I, item_feats)).astype(np.float32)
        itf = pd.DataFrame(np.column_stack([np.arange(I), vals]))
    sw = rng.uniform(0.25, 2.0, N).astype(np.float32) if sample_weights else None

    m = RankFM(factors=F, loss=loss, max_samples=max_samples, alpha=alpha, beta=beta, sigma=sigma,
               learning_rate=lr, learning_schedule=schedule, learning_exponent=exponent)
    np.random.seed(s_init)
    m._init_all(pairs, uf, itf, sw)
    assert np.array_equal(m.interactions, pairs.astype(np.int32)), "ids are already 0-based indexes"
    ms = 1 if loss == "bpr" else max_samples                     # rankfm.py:294-297
    w0 = {k: getattr(m, k).copy() for k in WEIGHTS}

    # weights after each epoch count (fresh run per count: the MT stream restarts inside every _fit call)
    ll_printed = pen = None
    per_epoch = []
    for k in range(1, epochs + 1):
        wk, printed = run_fit(m, ms, k, w0, s_shuf)
        per_epoch.append(wk)
        if k == epochs:
            ll_printed = printed
    pen = np.array([penalty(m, wk) for wk in per_epoch])
    final = per_epoch[-1]
    off, items = csr_of(m.user_items, U)
    out = dict(
        interactions=m.interactions, sample_weight=m.sample_weight, csr_off=off, csr_items=items,
        x_uf=m.x_uf, x_if=m.x_if, perms=replay_perms(s_shuf, N, epochs),
        ll_printed=ll_printed, reg_penalty=pen,
        alpha=np.float32(alpha), beta=np.float32(beta), learning_rate=np.float32(lr),
        learning_exponent=np.float32(exponent), learning_schedule=np.array(schedule),
        max_samples=np.int32(ms), epochs=np.int32(epochs), loss=np.array(loss),
        has_uf=np.int32(int(m.x_uf.any())), has_if=np.int32(int(m.x_if.any())),
    )
    for k in WEIGHTS:
        out["init_" + k] = w0[k]
        out["final_" + k] = final[k]
        out["epoch1_" + k] = per_epoch[0][k]
    np.savez_compressed(os.path.join(HERE, "fit_%s.npz" % name), **out)
    print("fit_%-28s N=%d F=%d ms=%d  |v_u| %.4f -> %.4f  ll %s" % (
        name, N, F, ms, np.linalg.norm(w0["v_u"]), np.linalg.norm(final["v_u"]), ll_printed))


def api_case(name, *, loss, with_features, str_ids, seed=5):
    """public-API fixture: raw ids in, _init_all products + fit/predict/recommend/hit_rate out"""
    rng = np.random.default_rng(seed)
    U, I, N, F = 30, 45, 420, 6
    pairs = make_interactions(rng, U, I, N)
    uid = np.array(["u%03d" % (3 * k + 7) for k in range(U)]) if str_ids else (np.arange(U) * 3 + 100)
    iid = np.array(["i%03d" % (2 * k + 1) for k in range(I)]) if str_ids else (np.arange(I) * 2 + 1000)
    raw = pd.DataFrame({"user_id": uid[pairs[:, 0]], "item_id": iid[pairs[:, 1]]})
    order_u, order_i = rng.permutation(U), rng.permutation(I)     # feature frames arrive in arbitrary row order
    uf = itf = None
    if with_features:
        ufv = (rng.random((U, 4)) < 0.4).astype(np.float32)
        ifv = (rng.random((I, 5)) < 0.4).astype(np.float32)
        uf = pd.concat([pd.DataFrame({"user_id": uid[order_u]}), pd.DataFrame(ufv[order_u])], axis=1)
        itf = pd.concat([pd.DataFrame({"item_id": iid[order_i]}), pd.DataFrame(ifv[order_i])], axis=1)
    sw = rng.uniform(0.5, 1.5, N).astype(np.float32)
    train, test = raw.iloc[:340], raw.iloc[340:]
    sw_train = sw[:340]
    # every user/item must appear in train for the feature check; rebuild train to guarantee it
    missing_u = set(uid) - set(train.user_id)
    missing_i = set(iid) - set(train.item_id)
    extra = pd.DataFrame({"user_id": [u for u in missing_u] + [uid[0]] * len(missing_i),
                          "item_id": [iid[0]] * len(missing_u) + [i for i in missing_i]})
    train = pd.concat([train, extra], ignore_index=True)
    sw_train = np.concatenate([sw_train, np.ones(len(extra), dtype=np.float32)])

    m = RankFM(factors=F, loss=loss, max_samples=6, learning_schedule="invscaling")
    np.random.seed(21)
    m._init_all(train, uf, itf, sw_train)
    init = {k: getattr(m, k).copy() for k in WEIGHTS}
    off, items = csr_of(m.user_items, len(m.user_id))
    np.random.seed(22)
    m2 = RankFM(factors=F, loss=loss, max_samples=6, learning_schedule="invscaling")
    np.random.seed(21)      # fit() = _reset_state + _init_all (same seed -> same init) then _fit (shuffle continues the stream)
    m2.fit(train, uf, itf, sw_train, epochs=4)
    # the shuffle stream continues from the state left by _init_all: capture the perms by replaying the init draws
    np.random.seed(21)
    m3 = RankFM(factors=F, loss=loss, max_samples=6, learning_schedule="invscaling")
    m3._init_all(train, uf, itf, sw_train)
    idx = np.arange(len(train), dtype=np.int32)
    perms = []
    for _ in range(4):
        np.random.shuffle(idx)
        perms.append(idx.copy())

    unknown_u = "zz_user" if str_ids else 99999
    unknown_i = "zz_item" if str_ids else 99998
    pred_pairs = pd.concat([test, pd.DataFrame({"user_id": [unknown_u, uid[1]], "item_id": [iid[2], unknown_i]})], ignore_index=True)
    scores = m2.predict(pred_pairs, cold_start="nan")
    rec_users = list(uid[:12]) + [unknown_u]
    rec_all = m2.recommend(rec_users, n_items=7, filter_previous=False, cold_start="nan")
    rec_new = m2.recommend(rec_users, n_items=7, filter_previous=True, cold_start="nan")
    # similar_items / similar_users (rankfm/rankfm.py:405-454): the reference's own ranking for a few query rows
    sim_item_q, sim_user_q = list(iid[[0, 3, 7, 11]]), list(uid[[0, 2, 5, 9]])
    sim_items = np.stack([np.asarray(m2.similar_items(q, n_items=6)) for q in sim_item_q])
    sim_users = np.stack([np.asarray(m2.similar_users(q, n_users=6)) for q in sim_user_q])
    out = dict(
        train_users=train.user_id.values.astype("U16" if str_ids else np.int64),
        train_items=train.item_id.values.astype("U16" if str_ids else np.int64),
        train_sw=sw_train,
        test_users=test.user_id.values.astype("U16" if str_ids else np.int64),
        test_items=test.item_id.values.astype("U16" if str_ids else np.int64),
        user_id=m.user_id.values.astype("U16" if str_ids else np.int64),
        item_id=m.item_id.values.astype("U16" if str_ids else np.int64),
        interactions=m.interactions, sample_weight=m.sample_weight, csr_off=off, csr_items=items,
        x_uf=m.x_uf, x_if=m.x_if, perms=np.stack(perms),
        pred_users=pred_pairs.user_id.values.astype("U16" if str_ids else np.int64),
        pred_items=pred_pairs.item_id.values.astype("U16" if str_ids else np.int64),
        pred_scores=scores,
        rec_users=np.array(rec_users).astype("U16" if str_ids else np.int64),
        rec_all=rec_all.values.astype("U16" if str_ids else np.float64),
        rec_new=rec_new.values.astype("U16" if str_ids else np.float64),
        sim_item_queries=np.array(sim_item_q).astype("U16" if str_ids else np.int64),
        sim_user_queries=np.array(sim_user_q).astype("U16" if str_ids else np.int64),
        sim_items=sim_items.astype("U16" if str_ids else np.int64), sim_users=sim_users.astype("U16" if str_ids else np.int64),
        hit_rate=np.float64(ref_eval.hit_rate(m2, test, k=7)),
        hit_rate_new=np.float64(ref_eval.hit_rate(m2, test, k=7, filter_previous=True)),
        reciprocal_rank=np.float64(ref_eval.reciprocal_rank(m2, test, k=7)),
        dcg=np.float64(ref_eval.discounted_cumulative_gain(m2, test, k=7)),
        precision=np.float64(ref_eval.precision(m2, test, k=7)),
        recall=np.float64(ref_eval.recall(m2, test, k=7)),
        loss=np.array(loss), factors=np.int32(F), max_samples=np.int32(6), epochs=np.int32(4),
        str_ids=np.int32(int(str_ids)), with_features=np.int32(int(with_features)),
    )
    if with_features:
        out.update(uf_ids=uid[order_u].astype("U16" if str_ids else np.int64), uf_vals=ufv[order_u],
                   if_ids=iid[order_i].astype("U16" if str_ids else np.int64), if_vals=ifv[order_i])
    for k in WEIGHTS:
        out["init_" + k] = init[k]
        out["final_" + k] = getattr(m2, k)
    np.savez_compressed(os.path.join(HERE, "api_%s.npz" % name), **out)
    print("api_%-28s hit_rate@7 %.4f  mrr %.4f  nan scores %d" % (name, out["hit_rate"], out["reciprocal_rank"], int(np.isnan(scores).sum())))


def partial_case(name, *, loss, second_with_features, seed=9):
    """public-API fixture for RESUMED training: fit(A, features) then fit_partial(B[, features]) (rankfm/rankfm.py:269-327).
    Every call restarts the MT19937 stream at 1492 (rankfm/_rankfm.pyx:182) and its learning-rate schedule at epoch 0; the
    numpy shuffle stream runs on; the per-user item sets are extended with B's items (rankfm/rankfm.py:170-172); when the
    second call omits the features, x_uf / x_if are rebuilt as zeros and the feature terms switch off while v_uf / v_if keep
    their values (rankfm/rankfm.py:286, 199, 211).  B holds every user (the reference raises KeyError otherwise, :172)."""
    rng = np.random.default_rng(seed)
    U, I, F = 28, 40, 6
    A = make_interactions(rng, U, I, 380)
    B = np.concatenate([np.stack([np.arange(U), rng.integers(0, I, U)], 1), np.stack([rng.integers(0, U, 90), rng.integers(0, I, 90)], 1)])
    rng.shuffle(B)
    uid, iid = np.arange(U) * 3 + 100, np.arange(I) * 2 + 1000
    fa = pd.DataFrame({"user_id": uid[A[:, 0]], "item_id": iid[A[:, 1]]})
    fb = pd.DataFrame({"user_id": uid[B[:, 0]], "item_id": iid[B[:, 1]]})
    ufv, ifv = (rng.random((U, 3)) < 0.4).astype(np.float32), (rng.random((I, 4)) < 0.4).astype(np.float32)
    uf = pd.concat([pd.DataFrame({"user_id": uid}), pd.DataFrame(ufv)], axis=1)
    itf = pd.concat([pd.DataFrame({"item_id": iid}), pd.DataFrame(ifv)], axis=1)
    swb = rng.uniform(0.5, 1.5, len(fb)).astype(np.float32)
    m = RankFM(factors=F, loss=loss, max_samples=5, learning_schedule="invscaling", sigma=0.5)
    np.random.seed(31)
    m.fit(fa, uf, itf, epochs=2)
    w1 = {k: getattr(m, k).copy() for k in WEIGHTS}
    m.fit_partial(fb, uf if second_with_features else None, itf if second_with_features else None, swb, epochs=2)
    w2 = {k: getattr(m, k).copy() for k in WEIGHTS}
    off, items = csr_of(m.user_items, U)
    out = dict(a_users=fa.user_id.values, a_items=fa.item_id.values, b_users=fb.user_id.values, b_items=fb.item_id.values, b_sw=swb,
               uf_vals=ufv, if_vals=ifv, user_id=uid, item_id=iid, loss=np.array(loss), factors=np.int32(F), max_samples=np.int32(5),
               second_with_features=np.int32(int(second_with_features)), interactions_after=m.interactions, csr_off_after=off,
               csr_items_after=items, x_uf_after=m.x_uf, x_if_after=m.x_if)
    for k in WEIGHTS:
        out["first_" + k] = w1[k]
        out["second_" + k] = w2[k]
    np.savez_compressed(os.path.join(HERE, "partial_%s.npz" % name), **out)
    print("partial_%-24s |v_u| %.4f -> %.4f   |v_uf| %.4f -> %.4f" % (name, np.linalg.norm(w1["v_u"]), np.linalg.norm(w2["v_u"]),
                                                                   np.linalg.norm(w1["v_uf"]), np.linalg.norm(w2["v_uf"])))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "partial":        # only the resumed-training fixtures
        partial_case("bpr_feat_then_none", loss="bpr", second_with_features=False)
        partial_case("warp_feat_then_feat", loss="warp", second_with_features=True)
        sys.exit(0)
    # --- _fit-level fixtures: {bpr, warp} x {no features, features} x {constant, invscaling} (+ sample weights, odd F)
    fit_case("bpr_nofeat_const_f8", U=40, I=60, N=600, F=8, loss="bpr")
    fit_case("bpr_nofeat_inv_f10_sw", U=40, I=60, N=600, F=10, loss="bpr", schedule="invscaling", sample_weights=True, data_seed=1)
    fit_case("bpr_feat_const_f8", U=40, I=60, N=600, F=8, loss="bpr", user_feats=4, item_feats=5, data_seed=2)
    fit_case("bpr_feat_inv_f16_sw", U=36, I=50, N=500, F=16, loss="bpr", user_feats=3, item_feats=6, schedule="invscaling",
             sample_weights=True, data_seed=3, feat_kind="dense")
    fit_case("bpr_nofeat_const_f64", U=50, I=80, N=800, F=64, loss="bpr", data_seed=4)
    fit_case("bpr_nofeat_const_f20", U=50, I=80, N=800, F=20, loss="bpr", data_seed=9)
    fit_case("bpr_nofeat_const_f128", U=30, I=64, N=500, F=128, loss="bpr", data_seed=10, epochs=2)
    fit_case("warp_nofeat_const_f8", U=40, I=60, N=600, F=8, loss="warp", max_samples=8, sigma=1.5, data_seed=5)
    fit_case("warp_nofeat_inv_f64_sw", U=50, I=80, N=700, F=64, loss="warp", max_samples=12, sigma=0.5, schedule="invscaling",
             sample_weights=True, data_seed=6)
    fit_case("warp_feat_const_f8", U=40, I=60, N=600, F=8, loss="warp", max_samples=8, sigma=1.0, user_feats=4, item_feats=5, data_seed=7)
    fit_case("warp_feat_inv_f12", U=36, I=50, N=500, F=12, loss="warp", max_samples=10, sigma=1.0, user_feats=3, item_feats=4,
             schedule="invscaling", data_seed=8, feat_kind="dense", lr=0.05)
    fit_case("bpr_ufeat_only_f8", U=40, I=60, N=500, F=8, loss="bpr", user_feats=5, data_seed=11)
    fit_case("warp_ifeat_only_f8", U=40, I=60, N=500, F=8, loss="warp", max_samples=6, sigma=1.0, item_feats=5, data_seed=12)
    # --- public-API fixtures
    api_case("bpr_int_nofeat", loss="bpr", with_features=False, str_ids=False)
    api_case("warp_str_feat", loss="warp", with_features=True, str_ids=True)
    # --- resumed training through the public API
    partial_case("bpr_feat_then_none", loss="bpr", second_with_features=False)
    partial_case("warp_feat_then_feat", loss="warp", second_with_features=True)
