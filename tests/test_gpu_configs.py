"""BASELINE.json configs 3, 4 and 5 on the GPU (config 2 at full size lives in test_gpu_parity.py).

Config 3 runs at full size.  Configs 4 and 5 are 8-GPU configurations: what ONE GPU of the eight holds -- the user shard
rank 0 of 8 of the config's data set (rankfm_amd.synthetic.make_config_shard: 1/8 of the users and interactions over the whole
item catalogue) -- runs at full size here, against the sequential CPU oracle where the oracle finishes in about a minute and
through size-independent properties where it does not.

Every comparison with the oracle is on the engine's own visiting order and counter-based draws (rankfm_amd.order), like the
config-2 test; tolerances are stated per check and are statistical by construction (Hogwild).  Log-likelihoods are compared with
the oracle's DOUBLE sum (`ll64`: the reference's float accumulator is off by +0.4 % on config 3 and -1.7 % on config 4's share,
profiles/r03_notes.md).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _norm_ratio(a, b):
    return float(np.linalg.norm(a) / np.linalg.norm(b))


def _oracle_epoch(oracle, sh, w, max_samples, epoch, seed, lr, geometry=None, **oracle_kw):
    """one epoch of the sequential oracle, in place on `w`, in the engine's order of epoch `epoch`"""
    from rankfm_amd import order
    pairs = sh["interactions"]
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs_csr = np.ascontiguousarray(pairs[by_csr])
    assert np.array_equal(pairs_csr[:, 1], sh["csr_items"])
    sw_csr = np.ascontiguousarray(sh["sample_weight"][by_csr])
    perms = order.epoch_positions(sh["csr_offsets"], seed, epoch, (geometry or {}).get("segment_rows") or None)[None, :].astype(np.int32)
    return oracle.fit(pairs_csr, sw_csr, sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"], w["w_i"], w["w_if"], w["v_u"],
                      w["v_i"], w["v_uf"], w["v_if"], 0.01, 0.1, lr, "constant", 0.25, max_samples, 1, perms=perms,
                      rng_mode=oracle.RNG_COUNTER, seed=seed, membership="binary", epoch_begin=epoch, want_negatives=True, **oracle_kw)


def _trained_then_one_epoch(oracle, sh, max_samples, warm_epochs, seed, lr=0.1, damped=False):
    """`warm_epochs` epochs on the GPU, then ONE more epoch on the GPU and on the oracle from the same trained weights; with `damped`
    also on the oracle under the engine's step damping (returned last)"""
    from rankfm_amd.engine import DeviceSession
    sess = DeviceSession(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"],
                         sh["weights"], max_samples=max_samples, seed=seed, learning_rate=lr)
    warm = sess.run(epochs=warm_epochs) if warm_epochs else None
    w0 = sess.weights_to_host()
    rep = sess.run(epochs=1, epoch_begin=warm_epochs)
    g = sess.weights_to_host()
    o = {k: v.copy() for k, v in w0.items()}
    out = _oracle_epoch(oracle, sh, o, max_samples, warm_epochs, seed, lr, sess.geometry())
    if damped:
        pos_step, user_step = sess.step_scales()
        od = {k: v.copy() for k, v in w0.items()}
        outd = _oracle_epoch(oracle, sh, od, max_samples, warm_epochs, seed, lr, sess.geometry(), pos_step=pos_step, user_step=user_step, neg_step=pos_step)
        return w0, g, rep, o, out, warm, (od, outd)
    return w0, g, rep, o, out, warm


def _assert_epoch_tracks_oracle(w0, g, rep, o, out, names, norm_tol, ll_tol, delta_corr, draws_tol=None):
    for k in names:
        r = _norm_ratio(g[k], o[k])
        assert abs(r - 1.0) <= norm_tol, "|%s| gpu / oracle = %.4f" % (k, r)
        # the epoch's MOVE of every weight, not the weights (which share their starting point)
        c = np.corrcoef((g[k] - w0[k]).ravel(), (o[k] - w0[k]).ravel())[0, 1]
        assert c > delta_corr, "%s: correlation of the epoch's updates with the sequential oracle's %.4f" % (k, c)
    np.testing.assert_allclose(rep["log_likelihood"], out["ll64"], rtol=ll_tol)
    if draws_tol is not None:
        np.testing.assert_allclose(float(rep["n_draws"][0]), float(out["nsamp"].sum()), rtol=draws_tol)


def test_config3_full_size_warp_tracks_sequential_oracle(oracle, c2_problem):
    """BASELINE config 3 = config 2's data, loss='warp', max_samples=50, at FULL size (rankfm/_rankfm.pyx:244-270).  Three
    epochs of training first, so that the model is past the stage where every first draw violates the margin: the compared
    epoch evaluates several candidates per update (the count is printed and checked against the oracle's).  Two comparisons of
    that epoch, from the same weights, on the engine's order and draws:
      (a) the sequential oracle under the ENGINE'S STEP DAMPING (hot items' steps -- on either side of a pair -- and heavy users'
          steps scaled like the plan scales them, DeviceSession.step_scales): what is left is asynchronous execution alone;
      (b) the reference's algorithm itself (rankfm/_rankfm.pyx:244-326, undamped).
    Both: log-likelihood 1 %, accepted draws 1.5 %, |v_u|, |v_i|, |w_i| 1 %.  Round 3 needed 2.75 % / 5 % / 4 % for (b): its damping
    scaled an item's step only when it was the POSITIVE item, which moves the fixed point of a hot item's bias (+1.9 % log-likelihood,
    -3.7 % draws, +2.7 % |w_i| on this epoch; +2 % / +9 % over four epochs from the initial weights).  Since round 4 an item's scale
    applies to its step on either side, the fixed point is the reference's, and four epochs from the initial weights end +0.08 %,
    0.0 %, +0.44 % from the reference algorithm (profiles/r04_notes.md).
    Correlation of the epoch's weight updates with the oracle's > 0.8 (WARP's discrete decisions -- first violating draw,
    rank-dependent multiplier -- turn stale reads into different-but-equivalent steps)."""
    U, I, N, F, pairs, csr = c2_problem
    from rankfm_amd import synthetic
    sh = dict(interactions=pairs, sample_weight=np.ones(N, np.float32), csr_offsets=csr.offsets, csr_items=csr.items,
              x_uf=np.zeros((U, 1), np.float32), x_if=np.zeros((I, 1), np.float32), weights=synthetic.init_weights(U, I, F, seed=1492))
    w0, g, rep, o, out, warm, (od, outd) = _trained_then_one_epoch(oracle, sh, max_samples=50, warm_epochs=3, seed=1492, damped=True)
    for name, oo, oout in (("reference algorithm", o, out), ("oracle with the engine's step damping", od, outd)):
        print("config 3 vs %s: draws per update gpu %.2f oracle %.2f (warm-up epochs %s); LL gpu/oracle - 1 = %+.4f; norms gpu/oracle %s"
              % (name, rep["n_draws"][0] / N, oout["nsamp"].sum() / N, np.round(warm["n_draws"] / N, 2),
                 rep["log_likelihood"][0] / oout["ll64"][0] - 1.0, [round(_norm_ratio(g[k], oo[k]), 4) for k in ("v_u", "v_i", "w_i")]))
    assert out["nsamp"].sum() > 1.5 * N                      # the multi-draw path is what is being compared
    _assert_epoch_tracks_oracle(w0, g, rep, od, outd, ("v_u", "v_i", "w_i"), norm_tol=0.01, ll_tol=0.01, delta_corr=0.8, draws_tol=0.015)
    _assert_epoch_tracks_oracle(w0, g, rep, o, out, ("v_u", "v_i", "w_i"), norm_tol=0.01, ll_tol=0.01, delta_corr=0.8, draws_tol=0.015)


def test_config4_one_gpu_share_with_features_tracks_sequential_oracle(oracle):
    """BASELINE config 4 (1 M users x 200 k items x 50 M interactions + 32-d user/item features, k=64, BPR over 8 GPUs): the
    share of ONE GPU -- 125 k users, 6.25 M interactions, all 200 k items, P = Q = 32 dense Bernoulli(0.25) tags -- on the GPU
    and on the sequential oracle (rankfm/_rankfm.pyx:283-286, 297-326).  Learning rate 0.03: at the reference's default 0.1
    the reference algorithm itself diverges on these tags.

    Two comparisons, both in the engine's order and draws.  (1) The first epoch from the initial weights: the dense tables are
    trained by ONE sequential stream (the table trainer of sgd_features_kernel) on a sample of the rows -- every ~280th at this
    size (rfm_fit_report.table_steps) -- while the sequential algorithm's tables follow within a few hundred rows of the WHOLE
    stream.  What that costs is decided in the first ~1 % of the rows, while the tables leave their initial values (the item
    biases pick up what the tables would carry: the two are degenerate, 8 active tags x the mean table row is an item bias), so
    the fit's opening rows run as a launch of their own in which the trainer sees every ~20th row (rfm_api.hip, "opening"):
    log-likelihood within 1.5 %, |w_i| within 2.5 %, factor norms within 1 % (measured in round 4, three runs: +0.53 ... +0.56 %,
    +0.12 ... +0.26 %, +0.10 / -0.04 %; round 3: +0.1 % / -2 %; without the opening +6.8 ... +7.7 % and +20 ... +23 %:
    profiles/r03_notes.md section 7).  (2) The second epoch, GPU and oracle both from the GPU's weights after the first:
    log-likelihood 1 %, every row norm 1.5 % (measured +0.10 ... +0.13 %, <= 0.22 %).
    Round 4: the trainer and its producers are a kernel of their own beside the row-loop kernel (two streams) and the trainer applies
    a FIXED quota of staged steps per launch: the step count repeats exactly from run to run (asserted), the two kernels must really
    overlap (asserted from the kernels' own clock stamps: an XCD that is handed more one-per-CU workgroups than it has CUs runs them
    one after the other, and the tables are then trained BEFORE the rows -- first-epoch log-likelihood -1.7 %, |w_i| -11 %), and the
    tables' norms are held to a stated tolerance instead of a factor 2.5."""
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    sh = synthetic.make_config_shard("C4", rank=0, world=8)
    assert sh["interactions"].shape == (6_250_000, 2) and sh["x_uf"].shape == (125_000, 32) and sh["x_if"].shape == (200_000, 32)
    lr = sh["config"]["learning_rate"]
    sess = DeviceSession(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"],
                         sh["weights"], max_samples=1, seed=1492, learning_rate=lr)
    w0 = {k: np.array(v, copy=True) for k, v in sh["weights"].items()}
    rep1 = sess.run(epochs=1)
    g1 = sess.weights_to_host()
    geo = sess.geometry()
    rep2 = sess.run(epochs=1, epoch_begin=1)
    g2 = sess.weights_to_host()
    o1 = {k: v.copy() for k, v in w0.items()}
    out1 = _oracle_epoch(oracle, sh, o1, 1, 0, 1492, lr, sess.geometry())
    o2 = {k: v.copy() for k, v in g1.items()}
    out2 = _oracle_epoch(oracle, sh, o2, 1, 1, 1492, lr, sess.geometry())
    r1 = {k: round(_norm_ratio(g1[k], o1[k]), 4) for k in g1}
    r2 = {k: round(_norm_ratio(g2[k], o2[k]), 4) for k in g2}
    print("config 4 share: epoch 1 LL gpu/oracle - 1 = %+.4f norms %s | epoch 2 (same start) LL %+.4f norms %s; SGD kernel %.1f ms; the trainer "
          "(%d producers) applied %d staged steps in the first epoch = every %.0f-th row"
          % (rep1["log_likelihood"][0] / out1["ll64"][0] - 1.0, r1, rep2["log_likelihood"][0] / out2["ll64"][0] - 1.0, r2, rep2["sgd_kernel_ms"][0],
             geo["table_producers"], geo["table_steps"], 6_250_000 / max(geo["table_steps"], 1)))
    assert geo["table_producers"] >= 1 and geo["table_steps"] > 6_250_000 / 1000      # the trainer keeps up with >= every 1000th row
    # the trainer works to a quota fixed by the launch geometry (round 4): the same number of staged steps in every run
    again = DeviceSession(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"],
                          sh["weights"], max_samples=1, seed=1492, learning_rate=lr)
    again.run(epochs=1)
    assert again.geometry()["table_steps"] == geo["table_steps"], (again.geometry()["table_steps"], geo["table_steps"])
    del again
    # and the tables kernel runs BESIDE the row loops (two streams): most of the shorter one's time on the boxes measured.  How the
    # hardware deals the workgroups of two concurrent kernels round the XCDs is not contractual and a profiler serialises them
    # (rfm_api.hip "Placement is not contractual": less overlap is slower, still correct -- the quota is fixed): a warning, not a failure
    if geo["table_overlap_us"] < 0.6 * min(geo["table_span_us"]):
        import warnings
        warnings.warn("tables kernel overlapped the row loops for %d us of %s us (placement / profiler?)" % (geo["table_overlap_us"], geo["table_span_us"]))
    np.testing.assert_allclose(rep1["log_likelihood"], out1["ll64"], rtol=0.015)
    assert abs(r1["w_i"] - 1.0) <= 0.025 and abs(r1["v_u"] - 1.0) <= 0.01 and abs(r1["v_i"] - 1.0) <= 0.01, r1
    np.testing.assert_allclose(rep2["log_likelihood"], out2["ll64"], rtol=0.01)
    assert all(abs(r2[k] - 1.0) <= 0.015 for k in ("w_i", "v_u", "v_i")), r2
    # The feature tables, against the REFERENCE'S OWN run-to-run spread.  tests/golden/quality_tags_spread.npz holds
    # eight runs of the reference itself on config 4 reduced 1 : 80 (same proportions, tags, learning rate) that differ in the order of the
    # rows only: the tables are an exponential moving average of the last ~170 rows' gradient noise, and the reference's own |v_uf|, |v_if|,
    # |w_if| move by 3.2 / 4.2 / 17 % (one sigma) after the first epoch and 1.8 / 4.6 / 12 % after the second from one order to the next
    # (|w_i| 0.09 %, |v_u|, |v_i| 0.01 %) -- whatever the problem's size.  The engine's tables are held to max(2 %, 2 sigma_ref) of the
    # oracle's: 6.4 / 8.4 / 34 % and 3.6 / 9.2 / 24 % (rounds 4 - 5 asserted 20 %, then 10 / 15 %, unanchored).  Measured in round 5 (three
    # runs on two builds): after the first epoch 1.044 ... 1.045 / 1.045 / 0.908, after the second 1.000 ... 1.002 / 0.923 ... 0.927 /
    # 1.072 ... 1.086; with the trainer's fixed quota they repeat to half a percent.
    from conftest import load_golden
    z = load_golden("quality", "tags_spread")
    cols = [str(c) for c in z["c4r_columns"]]
    rel_sigma = z["c4r_order_only"].std(axis=0, ddof=1) / np.abs(z["c4r_order_only"].mean(axis=0))
    for epoch, r in ((1, r1), (2, r2)):
        for k in ("v_uf", "v_if", "w_if"):
            tol = max(0.02, 2.0 * float(rel_sigma[cols.index("e%d_norm_%s" % (epoch, k))]))
            assert abs(r[k] - 1.0) <= tol, ("epoch %d |%s| gpu / oracle = %.4f, tolerance max(2 %%, 2 sigma_ref) = %.3f" % (epoch, k, r[k], tol), r)
    assert all(np.isfinite(g2[k]).all() for k in g2)


@pytest.fixture(scope="module")
def c5_share():
    from rankfm_amd import synthetic
    return synthetic.make_config_shard("C5", rank=0, world=8)


def test_config5_one_gpu_share_properties(c5_share):
    """BASELINE config 5 (5 M users x 1 M items x 500 M interactions, k=128, WARP over 8 GPUs): ONE GPU's share at full size
    -- 625 k users x 1 M items x 62.5 M interactions -- through size-independent properties (the oracle would need an hour):
    with alpha = 0 every step adds +d to v_i[i] and -d to v_i[j] and +-g to w_i (rankfm/_rankfm.pyx:279-280, 309-310), so
    the column sums of v_i and the sum of w_i are invariants of ANY interleaving iff no update is lost; every update accepts
    at least one draw; nothing goes non-finite; the mean log-likelihood per update stays above the untrained log(0.5)."""
    from rankfm_amd.engine import DeviceSession
    sh = c5_share
    N = len(sh["interactions"])
    assert N == 62_500_000 and sh["weights"]["v_u"].shape == (625_000, 128) and sh["weights"]["v_i"].shape == (1_000_000, 128)
    w = sh["weights"]
    before = w["v_i"].astype(np.float64).sum(axis=0)
    sess = DeviceSession(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"], w,
                         alpha=0.0, beta=0.0, max_samples=50, seed=1492, hogwild_damping=1e9)
    rep = sess.run(epochs=3)
    h = sess.weights_to_host()
    after = h["v_i"].astype(np.float64).sum(axis=0)
    moved = np.abs(h["v_i"] - w["v_i"]).astype(np.float64).sum(axis=0)
    assert np.all(np.abs(after - before) <= 2e-5 * moved + 1e-3), (np.abs(after - before).max(), moved.min())
    assert abs(float(h["w_i"].astype(np.float64).sum())) <= 2e-5 * float(np.abs(h["w_i"]).astype(np.float64).sum()) + 1e-3
    assert all(np.isfinite(h[k]).all() for k in h)
    assert np.all(rep["n_draws"] >= N) and np.all(rep["n_draws"] <= 50 * N)
    # (WARP's log-likelihood is that of the hardest negative found among the draws, and a better model makes the search go on
    # for more draws: it is not monotone from epoch to epoch -- measured -0.3481 / -0.3495 / ... per update)
    assert np.isfinite(rep["log_likelihood"]).all() and np.all(rep["log_likelihood"] < 0) and np.all(rep["log_likelihood"] > -0.7 * N)
    print("config 5 share: draws per update %s, mean LL per update %s, SGD kernel ms %s"
          % (np.round(rep["n_draws"] / N, 2), np.round(rep["log_likelihood"] / N, 4), np.round(rep["sgd_kernel_ms"], 1)))


def test_config5_subsample_tracks_sequential_oracle(oracle, c5_share):
    """... and against the oracle on a sub-sample the oracle finishes in a minute: the first 31,250 users of the share (3.1 M
    interactions) over the whole 1 M-item catalogue, k=128, WARP(50), three epochs of training and then one compared epoch."""
    sh = c5_share
    n_users = 31_250
    hi = int(sh["csr_offsets"][n_users])
    sel = sh["interactions"][:, 0] < n_users
    sub = dict(interactions=np.ascontiguousarray(sh["interactions"][sel]), sample_weight=np.ones(hi, np.float32),
               csr_offsets=sh["csr_offsets"][:n_users + 1].copy(), csr_items=sh["csr_items"][:hi].copy(),
               x_uf=np.zeros((n_users, 1), np.float32), x_if=sh["x_if"],
               weights=dict(sh["weights"], v_u=sh["weights"]["v_u"][:n_users].copy()))
    assert len(sub["interactions"]) == hi
    w0, g, rep, o, out, warm = _trained_then_one_epoch(oracle, sub, max_samples=50, warm_epochs=3, seed=77)
    print("config 5 sub-sample: draws per update gpu %.2f oracle %.2f; LL gpu/oracle - 1 = %+.4f; norms gpu/oracle %s"
          % (rep["n_draws"][0] / hi, out["nsamp"].sum() / hi, rep["log_likelihood"][0] / out["ll64"][0] - 1.0,
             [round(_norm_ratio(g[k], o[k]), 4) for k in ("v_u", "v_i", "w_i")]))
    _assert_epoch_tracks_oracle(w0, g, rep, o, out, ("v_u", "v_i", "w_i"), norm_tol=0.02, ll_tol=0.02, delta_corr=0.9, draws_tol=0.05)


def test_config4_eight_shards_merged_at_its_own_size_track_one_gpu():
    """BASELINE config 4 is an 8-GPU configuration, and no 8-GPU node has been available: rounds 1 - 5 ran its data as ONE rank's share and
    exercised the merge across ranks at config 2's shape only.  Here the WHOLE data set -- 1 M users x 200 k items x
    50 M interactions, 32 + 32 tags, learning rate 0.03 -- is trained (a) by one engine session on one GPU and (b) as EIGHT user shards,
    each by the real engine with a rank's own concurrency plan and its own copy of the item-side tables, merged after every exchange window
    like the ranks merge (distributed.emulate_ranks_on_one_device: the curvature rule on v_i / w_i over the 52 MB bucket, the default
    cadence of a fit's early epochs -- 8 blocking exchanges per epoch -- and the MEAN of the ranks' feature-table deltas,
    SharedTables.table_merge).  Two epochs from the same initial weights, the phase in which the model moves fastest.

    Measured, merged / one GPU (tools/merge_c4_scan.py, profiles/r06_notes.md sections 7 - 8): |v_u| 1.000, |v_i| 1.015, |w_i| 1.045, |v_uf|
    0.98, corr(w_i) 0.963 (the biases of a merged model run ahead of a single GPU's in the first epochs: the curvature rule SUMS the ranks'
    deltas while steps are few) -- and |v_if| 0.375, |w_if| 0.383: the item-feature tables are mostly gradient noise with a memory of ~170
    trainer steps, and the mean of eight ranks' independent noises is a third of one.  That shrink is a STATED deviation from a one-GPU fit:
    the alternatives that keep the norms -- one rank's tables per exchange, or the ranks taking turns training them -- were built and
    measured, and they rank 4.7 - 6.9 points of hit_rate@10 WORSE at config 2's shape with the default cadence where the mean ranks 1.5
    points better than one GPU (tests/test_gpu_quality.py; here: |w_i| 1.08 / 1.29).  The one-window-late merge is not taken at this size
    (ShardedTrainer.LATE_MOVEMENT).  Asserted: nothing diverges; |v_u| 2 %, |v_i| 4 %, |w_i| 10 %, |v_uf| 12 %, corr(w_i) >= 0.95; |v_if|
    and |w_if| inside the band the averaging predicts (0.25 ... 0.60 of one GPU's)."""
    import torch
    from rankfm_amd import synthetic
    from rankfm_amd.distributed import emulate_ranks_on_one_device
    from rankfm_amd.engine import DeviceSession
    sh = synthetic.make_config_shard("C4", rank=0, world=1)
    assert sh["interactions"].shape == (50_000_000, 2) and sh["x_uf"].shape == (1_000_000, 32)
    lr, E = sh["config"]["learning_rate"], 2
    w0 = {k: np.array(v, copy=True) for k, v in sh["weights"].items()}
    one = DeviceSession(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"],
                        {k: v.copy() for k, v in w0.items()}, max_samples=1, seed=1492, learning_rate=lr)
    rep = one.run(epochs=E)
    g = one.weights_to_host()
    del one
    torch.cuda.empty_cache()
    hyper = dict(alpha=0.01, beta=0.1, learning_rate=lr, learning_schedule="constant", learning_exponent=0.25, max_samples=1)
    problem = dict(interactions=sh["interactions"], sample_weight=sh["sample_weight"], csr_offsets=sh["csr_offsets"], csr_items=sh["csr_items"],
                   x_uf=sh["x_uf"], x_if=sh["x_if"], weights=w0)
    m = emulate_ranks_on_one_device(problem, 8, hyper, E, torch.device("cuda", 0), syncs_per_epoch="auto", seed=1492,
                                    has_user_features=1, has_item_features=1)
    ratio = {k: float(np.linalg.norm(m[k].astype(np.float64)) / np.linalg.norm(g[k].astype(np.float64))) for k in g}
    corr = float(np.corrcoef(m["w_i"], g["w_i"])[0, 1])
    print("config 4 whole, eight shards merged / one GPU after %d epochs: norms %s, corr(w_i) %.4f; one GPU: LL per update %s, SGD ms %s"
          % (E, {k: round(v, 4) for k, v in ratio.items()}, corr, np.round(rep["log_likelihood"] / 5e7, 4), np.round(rep["sgd_kernel_ms"], 1)))
    assert all(np.isfinite(m[k]).all() for k in m)
    tol = dict(v_u=0.02, v_i=0.04, w_i=0.10, v_uf=0.12)
    for k, t in tol.items():
        assert abs(ratio[k] - 1.0) <= t, (k, ratio[k], t, ratio)
    for k in ("v_if", "w_if"):             # (the mean of eight ranks' noise-dominated tables: see the docstring)
        assert 0.25 <= ratio[k] <= 0.60, (k, ratio[k], ratio)
    assert corr >= 0.95, corr
