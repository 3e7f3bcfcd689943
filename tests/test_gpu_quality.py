"""The north star's quality bar against the REFERENCE ITSELF: hit_rate@10 within 1 point (abs, seed-averaged) and factor norms
within 2 % of etlundquist/rankfm's own fit on the same data.

tests/golden/quality_planted.npz holds what the reference's compiled `_fit` + its own evaluation functions produced in the build
container (tests/golden/make_quality_golden.py): BASELINE.json config 1 hyper-parameters (factors=20, epochs=5) on the seeded planted
MovieLens-1M-shaped surrogate, five seeds, BPR and WARP(20).  The data are regenerated here from the seeds; the GPU side is the
default production engine (Hogwild, counter RNG, keyed order) -- a different visiting order and different negatives than the
reference's, so the comparison is statistical by construction."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("loss", ["bpr", "warp"])
def test_hit_rate_and_norms_match_the_reference(loss):
    from rankfm_amd import RankFM, evaluation, synthetic
    z = load_golden("quality", "planted")
    cols = list(z["columns"])
    ref = z[loss]
    got = []
    for seed in range(5):
        d = synthetic.make_planted(seed=seed)
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        assert len(train) == int(ref[seed, cols.index("n_train")])           # same data as the reference saw
        m = RankFM(factors=20, loss=loss, max_samples=20)
        np.random.seed(seed)
        m.fit(train, epochs=5)
        got.append([evaluation.hit_rate(m, test, k=10), evaluation.precision(m, test, k=10), evaluation.recall(m, test, k=10),
                    np.linalg.norm(m.v_u), np.linalg.norm(m.v_i), np.linalg.norm(m.w_i)])
    got, want = np.mean(got, axis=0), ref[:, :6].mean(axis=0)
    assert abs(got[0] - want[0]) <= 0.01, ("hit_rate@10", got[0], want[0])
    assert abs(got[1] - want[1]) <= 0.01 and abs(got[2] - want[2]) <= 0.005, ("precision/recall@10", got[1:3], want[1:3])
    np.testing.assert_allclose(got[3:5], want[3:5], rtol=0.02)                 # |v_u|, |v_i|
    np.testing.assert_allclose(got[5], want[5], rtol=0.04)                     # |w_i|: the biases are the most order-sensitive table


def test_feature_model_matches_the_reference():
    """Same bar for a model WITH user and item features (8 + 8 binary tags that carry signal): the reference's numbers are in
    tests/golden/quality_planted_tags.npz (make_quality_tags_golden.py; learning_rate 0.03 -- at its default 0.1 the reference
    itself diverges on dense tags).  The GPU side runs the production feature kernel (sgd_features_kernel: one table trainer
    workgroup fed by step-producer workgroups, every other workgroup reading the tables from an LDS copy; the fit's opening
    rows as a table-friendly launch of their own).  Measured over 30 runs (tools/feature_quality.py): hit_rate@10 0.4759
    against the reference's 0.4784, |v_u| +0.6 %, |v_i| -0.2 %, |w_i| -0.9 %."""
    from rankfm_amd import RankFM, evaluation, synthetic
    z = load_golden("quality", "planted_tags")
    cols = list(z["columns"])
    ref = z["bpr"]
    # (round 6) the reference's forty runs on the same five data sets -- eight visiting orders each from the initial weights of
    # `np.random.seed(seed); fit(...)` -- instead of its five single runs: one run's hit rate moves by 1.6 points with the order alone
    # (tests/golden/make_quality_tags_spread.py), so the mean of five carried +-0.7 point of its own against the 1-point bar
    spread = load_golden("quality", "tags_spread")["tags_order_only"]              # [data seed, run, hit_rate | six norms]
    got = []
    for seed in range(5):
        d = synthetic.make_planted(seed=seed, n_users=3000, n_items=2000, mean_degree=100.0, n_tags=8)
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        assert len(train) == int(ref[seed, cols.index("n_train")])
        uf = pd.DataFrame(np.column_stack([np.arange(len(d["user_tags"])), d["user_tags"]]))
        itf = pd.DataFrame(np.column_stack([np.arange(len(d["item_tags"])), d["item_tags"]]))
        # eight engine runs per data seed: on 3000 test users ONE Hogwild run's hit rate moves by +-2 points from run to run (measured:
        # tools/feature_quality.py), so a five-run mean would be a coin toss against the 1-point bar; forty runs put the mean's own
        # spread at ~0.3 point
        for run in range(8):
            m = RankFM(factors=20, loss="bpr", learning_rate=0.03)
            np.random.seed(seed)
            m.fit(train, user_features=uf, item_features=itf, epochs=5)
            got.append([evaluation.hit_rate(m, test, k=10)] + [np.linalg.norm(getattr(m, k)) for k in ("v_u", "v_i", "w_i", "v_uf", "v_if", "w_if")])
    got, want = np.mean(got, axis=0), spread.mean(axis=(0, 1))
    # one run's sigma around its data seed's mean, relative: the reference's own run-to-run spread (order only)
    sigma = np.sqrt(((spread - spread.mean(axis=1, keepdims=True)) ** 2).sum(axis=(0, 1)) / (spread.shape[0] * (spread.shape[1] - 1))) / np.abs(want)
    print("feature model: got", np.round(got, 4), "reference (40 runs)", np.round(want, 4), "its five single runs", np.round(ref[:, :7].mean(axis=0), 4),
          "reference's relative sigma of one run", np.round(sigma, 4))
    assert abs(got[0] - want[0]) <= 0.010, ("hit_rate@10", got[0], want[0])
    # every norm within max(2 %, 2 sigma_ref) of the reference's mean: |v_u|, |v_i|, |w_i| 2 % (sigma_ref 0.4 - 0.6 %; rounds 4 - 5: 2.5 %);
    # the tables -- mostly gradient noise with a memory of ~1 / (2 beta eta) rows, DESIGN.md section 5.3 -- 9 % / 8 % and, for the eight
    # numbers of w_if, 2 sigma_ref = 70 % (rounds 4 - 5 allowed a factor 2 on all three, unanchored)
    for k in range(1, 7):
        tol = max(0.02, 2.0 * float(sigma[k]))
        assert abs(got[k] / want[k] - 1.0) <= tol, (cols[k], got[k], want[k], tol)


def test_default_sampler_holds_the_quality_bar_at_a_chip_filling_size():
    """Ranking quality at a size whose launches fill a good part of the chip -- 30,000 users x 12,000 items, 3.7 M interactions,
    BPR, k = 32, 5 epochs, FIVE seeds -- against the sequential
    oracle drawing its negatives like the reference (uniformly over the catalogue, rankfm/_rankfm.pyx:250-253), from the same
    initial weights.
    hit_rate@10 within 1.0 point (measured -0.17: profiles/r03_notes.md), |v_u|, |v_i| 2 %, |w_i| 4 %.  (The stripe sampler of rounds 2 - 5,
    which this test used to run beside the default, measured -1.03 points here and -2.3 at 100 k x 50 k and was removed in round 6.)
    The five seeds' data and oracle fits are CPU work and run in a process pool; the engine runs in this process."""
    import multiprocessing as mp
    from oracle.planted_worker import fit_planted
    from rankfm_amd import EngineOptions, RankFM, evaluation
    U, I, F, E, SEEDS = 30_000, 12_000, 32, 5, 5
    with mp.get_context("spawn").Pool(SEEDS) as pool:
        jobs = pool.map(fit_planted, [(s, U, I, F, E) for s in range(SEEDS)])
    sides = ["oracle", "default"]
    hits = {k: [] for k in sides}
    norms = {k: [] for k in sides}
    for job in jobs:
        seed = job["seed"]
        train, test = pd.DataFrame(job["train"], columns=["u", "i"]), pd.DataFrame(job["test"], columns=["u", "i"])
        for side in hits:
            rep = {}
            m = RankFM(factors=F, loss="bpr", engine=EngineOptions(seed=100 + seed))
            np.random.seed(seed)
            if side == "oracle":
                m._init_all(train)
                for k, v in job["weights"].items():
                    setattr(m, k, np.ascontiguousarray(v))
                m.is_fit = True
            else:
                m.fit(train, epochs=E)
            hits[side].append(evaluation.hit_rate(m, test, k=10))
            norms[side].append([np.linalg.norm(m.v_u), np.linalg.norm(m.v_i), np.linalg.norm(m.w_i)])
    mean = {k: float(np.mean(v)) for k, v in hits.items()}
    print("hit_rate@10 over %d seeds:" % SEEDS, {k: np.round(v, 4).tolist() for k, v in hits.items()}, "means", mean,
          "norms / oracle - 1:", {k: np.round(np.mean(norms[k], axis=0) / np.mean(norms["oracle"], axis=0) - 1.0, 4).tolist() for k in sides[1:]})
    assert mean["oracle"] > 0.7                                              # the task is learnable
    assert abs(mean["default"] - mean["oracle"]) <= 0.010, mean
    got, want = np.mean(norms["default"], axis=0), np.mean(norms["oracle"], axis=0)
    np.testing.assert_allclose(got[:2], want[:2], rtol=0.02)
    np.testing.assert_allclose(got[2], want[2], rtol=0.04)


# ---- the quality bar AT BASELINE config 2's shape: 100,000 users x 50,000 items, ~4.5 M training rows --------------
C2_SHAPE = dict(U=100_000, I=50_000, E=5, SEEDS=5)
C2_VARIANTS = {"bpr_k32": ("bpr", 32, 1), "bpr_k64": ("bpr", 64, 1), "warp_k32": ("warp", 32, 50),
               "warp_k64": ("warp", 64, 50),        # (config 3's actual model: WARP, max_samples 50, k = 64 -- VERDICT r05 item 5)
               # config 4's kind of model at this size: 8 + 8 binary user / item tags that carry signal, learning rate 0.05 and ten epochs
               # (at the default 0.1 the REFERENCE ALGORITHM goes non-finite on tag features -- here as on config 4's, BASELINE.md
               # section 5 -- and at 0.03 five epochs learn too little to rank) -- the features kernels; three seeds
               "bpr_k32_tags": ("bpr", 32, 1),
               # the sequential oracle in the ENGINE'S visiting order (user segments of <= 32 rows): the asynchrony term by itself
               "bpr_k32_engine_order": ("bpr", 32, 1)}
C2_TAGS, C2_TAG_SEEDS, C2_TAG_LR, C2_TAG_EPOCHS = 8, 3, 0.05, 10


@pytest.fixture(scope="module")
def c2_shape_jobs():
    """Five seeds of a planted ranking problem of config 2's shape (generated on the GPU: synthetic.make_planted_large_device) and the
    sequential oracle's fits on them -- reference sampler, same initial weights as the engine side -- for every variant below, all
    started at once in one process pool (fifteen CPU jobs, the slowest -- WARP with up to 50 draws -- about two minutes)."""
    import multiprocessing as mp
    from oracle.planted_worker import fit_pairs
    from rankfm_amd import synthetic
    data = {seed: synthetic.make_planted_large_device(C2_SHAPE["U"], C2_SHAPE["I"], seed=seed, n_tags=C2_TAGS) for seed in range(C2_SHAPE["SEEDS"])}
    pool = mp.get_context("spawn").Pool(len(C2_VARIANTS) * C2_SHAPE["SEEDS"])
    pending = {}
    for tag, (loss, F, ms) in C2_VARIANTS.items():
        for seed in data:
            if tag.endswith("_tags"):
                if seed < C2_TAG_SEEDS:      # (the oracle with features is ~3x the work: three seeds)
                    pending[(tag, seed)] = pool.apply_async(fit_pairs, ((tag, seed, data[seed]["train"], F, C2_TAG_EPOCHS, loss, ms,
                                                                         data[seed]["user_tags"], data[seed]["item_tags"], C2_TAG_LR),))
            elif tag.endswith("_engine_order"):
                pending[(tag, seed)] = pool.apply_async(fit_pairs, ((tag, seed, data[seed]["train"], F, C2_SHAPE["E"], loss, ms, None, None, 0.1, 32),))
            else:
                pending[(tag, seed)] = pool.apply_async(fit_pairs, ((tag, seed, data[seed]["train"], F, C2_SHAPE["E"], loss, ms),))
    yield data, pending
    pool.terminate()


@pytest.mark.parametrize("tag", [t for t in C2_VARIANTS if not t.endswith("_engine_order")])
def test_default_engine_holds_the_quality_bar_at_config2_shape(c2_shape_jobs, tag):
    """hit_rate@10 of the production default (uniform sampler, item damping, dynamic segment order) within 1.0 point of the sequential
    oracle with the reference's sampler (rankfm/_rankfm.pyx:250-253, evaluation.py:9-33), mean over FIVE seeds, at config 2's shape,
    for BPR at k = 32 and k = 64, for WARP (max_samples 50, config 3's loss) at k = 32 and k = 64, and for a BPR model with 8 + 8 user / item tags
    (config 4's kind of model: the features kernels, three seeds); |v_u|, |v_i| within 2 %, |w_i| within 4 %."""
    from rankfm_amd import EngineOptions, RankFM, evaluation
    data, pending = c2_shape_jobs
    loss, F, ms = C2_VARIANTS[tag]
    hits = {"oracle": [], "default": []}
    norms = {"oracle": [], "default": []}
    tags = tag.endswith("_tags")
    lr = C2_TAG_LR if tags else 0.1
    for seed, d in data.items():
        if tags and seed >= C2_TAG_SEEDS:
            continue
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        uf = itf = None
        if tags:      # (feature rows for exactly the users / items of the training data, as the reference demands: rankfm/rankfm.py:181-211)
            us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
            uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
            itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
        # (a model with tags is scored over FOUR engine seeds per data seed: its trajectory is chaotic in the engine's settings -- the same
        #  data and weights under a slightly different table quota or build move a seed's hit rate by +-0.5 point, profiles/r04_notes.md
        #  section 11 -- and three single runs would leave the mean +-0.3 point of that alone)
        runs_hit, runs_norm = [], []
        # (WARP: two engine seeds per data seed -- its 16 k rows in flight put k = 32 at -0.7 ... -0.8 point, and five single runs leave the
        #  mean +-0.15)
        for engine_seed in ([100 + seed, 1100 + seed, 2100 + seed, 3100 + seed] if tags else ([100 + seed, 1100 + seed] if loss == "warp" else [100 + seed])):
            m = RankFM(factors=F, loss=loss, max_samples=ms, learning_rate=lr, engine=EngineOptions(seed=engine_seed))
            np.random.seed(seed)
            m.fit(train, uf, itf, epochs=C2_TAG_EPOCHS if tags else C2_SHAPE["E"])
            assert (m.last_fit_report["geometry"]["table_producers"] > 0) == tags      # (the features kernels ran iff there are features)
            runs_hit.append(evaluation.hit_rate(m, test, k=10))
            runs_norm.append([np.linalg.norm(m.v_u), np.linalg.norm(m.v_i), np.linalg.norm(m.w_i)])
        hits["default"].append(float(np.mean(runs_hit)))
        norms["default"].append(np.mean(runs_norm, axis=0))
        job = pending[(tag, seed)].get(timeout=1500)
        o = RankFM(factors=F, loss=loss, max_samples=ms, learning_rate=lr, engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        o._init_all(train, uf, itf)
        for k, v in job["weights"].items():
            setattr(o, k, np.ascontiguousarray(v))
        o.is_fit = True
        hits["oracle"].append(evaluation.hit_rate(o, test, k=10))
        norms["oracle"].append([np.linalg.norm(o.v_u), np.linalg.norm(o.v_i), np.linalg.norm(o.w_i)])
    mean = {k: float(np.mean(v)) for k, v in hits.items()}
    got, want = np.mean(norms["default"], axis=0), np.mean(norms["oracle"], axis=0)
    print("config-2 shape %s: hit_rate@10 %s means %s  norms / oracle - 1 %s" % (tag, {k: np.round(v, 4).tolist() for k, v in hits.items()}, mean,
                                                                             np.round(got / want - 1.0, 4).tolist()))
    assert mean["oracle"] > 0.25                                             # the task is learnable (a popularity ranking scores ~0.1 here)
    assert abs(mean["default"] - mean["oracle"]) <= 0.010, mean
    np.testing.assert_allclose(got[:2], want[:2], rtol=0.02)
    np.testing.assert_allclose(got[2], want[2], rtol=0.04)


@pytest.mark.parametrize("syncs, late", [("auto", True), ("auto", False)])
def test_eight_engine_shards_merged_like_the_ranks_hold_the_quality_bar(c2_shape_jobs, syncs, late):
    """BASELINE configs 4 / 5 run on eight ranks.  No 8-GPU node: eight user shards of a config-2-shaped problem, each trained by the REAL
    engine on this one GPU (its own session, its own copy of the item-side tables, a rank's concurrency plan) and merged after every
    exchange like the ranks merge (distributed.emulate_ranks_on_one_device = the curvature rule of SharedTables.exchange_fused with the
    all-reduce replaced by a loop), at the DEFAULT cadence (syncs_per_epoch "auto": eight exchanges per epoch during a fit's first
    eight epochs): hit_rate@10 within 1.0 point of the sequential oracle on the whole data after the five epochs of the other tests
    here (two seeds; the oracle fits are the fixture's), |v_i|, |w_i| within 5 %.  Measured (tools/merge_engine_scan.py,
    profiles/r04_notes.md): +0.2 point against one GPU on the whole data; with ONE exchange per epoch -13.9 after 5 epochs and +2.2
    after 15 -- which is why one exchange per epoch is not the default while the model moves fast.  (Round 3 tuned the rule against
    shards trained by the sequential oracle; this closes the loop with the asynchronous engine in every shard.)
    `late` = the one-window-late merge (the all-reduce of a window runs beside the next window's SGD, SharedTables.exchange_late; opt-in
    since round 6: at THIS problem's 4 positive updates per item per late window it is stable and holds the same bar, at configs 2 - 5's own
    sizes it rings and `overlap="auto"` refuses it: ShardedTrainer.LATE_MOVEMENT); False = every exchange blocks (the default)."""
    import torch
    from rankfm_amd import EngineOptions, RankFM, evaluation
    from rankfm_amd.distributed import emulate_ranks_on_one_device
    data, pending = c2_shape_jobs
    loss, F, ms = C2_VARIANTS["bpr_k32"]
    hits, norms = {"oracle": [], "merged": []}, {"oracle": [], "merged": []}
    for seed in (0, 1):
        d = data[seed]
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        m = RankFM(factors=F, loss=loss, engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        m._init_all(train)
        problem = dict(interactions=m.interactions, sample_weight=m.sample_weight, csr_offsets=m.user_items.offsets, csr_items=m.user_items.items,
                       x_uf=m.x_uf, x_if=m.x_if, weights={k: getattr(m, k) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")})
        hyper = dict(alpha=m.alpha, beta=m.beta, learning_rate=m.learning_rate, learning_schedule=m.learning_schedule,
                     learning_exponent=m.learning_exponent, max_samples=1)
        out = emulate_ranks_on_one_device(problem, 8, hyper, C2_SHAPE["E"], torch.device("cuda", 0), syncs_per_epoch=syncs, seed=100 + seed, late=late)
        for side, weights in (("merged", out), ("oracle", pending[("bpr_k32", seed)].get(timeout=1500)["weights"])):
            o = RankFM(factors=F, loss=loss, engine=EngineOptions(seed=100 + seed))
            np.random.seed(seed)
            o._init_all(train)
            for k, v in weights.items():
                setattr(o, k, np.ascontiguousarray(v))
            o.is_fit = True
            hits[side].append(evaluation.hit_rate(o, test, k=10))
            norms[side].append([np.linalg.norm(o.v_u), np.linalg.norm(o.v_i), np.linalg.norm(o.w_i)])
    mean = {k: float(np.mean(v)) for k, v in hits.items()}
    got, want = np.mean(norms["merged"], axis=0), np.mean(norms["oracle"], axis=0)
    print("eight engine shards, %s exchange(s) per epoch%s: hit_rate@10 %s means %s  norms / oracle - 1 %s"
          % (syncs, ", late merge" if late else "", {k: np.round(v, 4).tolist() for k, v in hits.items()}, mean, np.round(got / want - 1.0, 4).tolist()))
    assert abs(mean["merged"] - mean["oracle"]) <= 0.010, mean
    np.testing.assert_allclose(got[1:], want[1:], rtol=0.05)


def test_eight_engine_shards_of_a_model_with_tags_hold_the_quality_bar(c2_shape_jobs):
    """The same eight-shard emulation for config 4's KIND of model (8 + 8 tags that carry signal, the features kernels and the table trainer in
    every shard; two data seeds, ten epochs at learning rate 0.05 like the one-GPU tags test), default cadence, blocking exchanges, the feature
    tables merged as the MEAN of the ranks' deltas (SharedTables.table_merge).  This test is what settled that rule (tools/merge_tags_scan.py,
    profiles/r06_notes.md section 8; against one engine on the whole data, which is ~1 point under the oracle): mean +1.5 points, one rank's
    tables per exchange -4.7, the ranks taking turns training them -6.9 -- the two that keep the tables' norms need eight exchanges per epoch
    for the whole fit to rank within a point.  It also found two defects of an epoch trained in PARTS, fixed in rfm_api.hip: every part
    trained the tables on the same sampled rows, and every part kept a table schedule of its own (test below).
    hit_rate@10 within 1.5 points of the sequential oracle with tags (one merged run per seed), |v_i|, |w_i| within 5 %."""
    import torch
    from rankfm_amd import EngineOptions, RankFM, evaluation
    from rankfm_amd.distributed import emulate_ranks_on_one_device
    data, pending = c2_shape_jobs
    loss, F, ms = C2_VARIANTS["bpr_k32_tags"]
    hits, norms = {"oracle": [], "merged": []}, {"oracle": [], "merged": []}
    for seed in (0, 1):
        d = data[seed]
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
        uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
        itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
        m = RankFM(factors=F, loss=loss, learning_rate=C2_TAG_LR, engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        m._init_all(train, uf, itf)
        problem = dict(interactions=m.interactions, sample_weight=m.sample_weight, csr_offsets=m.user_items.offsets, csr_items=m.user_items.items,
                       x_uf=m.x_uf, x_if=m.x_if, weights={k: getattr(m, k) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")})
        hyper = dict(alpha=m.alpha, beta=m.beta, learning_rate=m.learning_rate, learning_schedule=m.learning_schedule,
                     learning_exponent=m.learning_exponent, max_samples=1)
        out = emulate_ranks_on_one_device(problem, 8, hyper, C2_TAG_EPOCHS, torch.device("cuda", 0), syncs_per_epoch="auto", seed=100 + seed,
                                          has_user_features=1, has_item_features=1)
        for side, weights in (("merged", out), ("oracle", pending[("bpr_k32_tags", seed)].get(timeout=1500)["weights"])):
            o = RankFM(factors=F, loss=loss, learning_rate=C2_TAG_LR, engine=EngineOptions(seed=100 + seed))
            np.random.seed(seed)
            o._init_all(train, uf, itf)
            for k, v in weights.items():
                setattr(o, k, np.ascontiguousarray(v))
            o.is_fit = True
            hits[side].append(evaluation.hit_rate(o, test, k=10))
            norms[side].append([np.linalg.norm(getattr(o, k)) for k in ("v_u", "v_i", "w_i", "v_uf", "v_if", "w_if")])
    mean = {k: float(np.mean(v)) for k, v in hits.items()}
    got, want = np.mean(norms["merged"], axis=0), np.mean(norms["oracle"], axis=0)
    print("eight engine shards WITH TAGS: hit_rate@10 %s means %s  norms / oracle - 1 (v_u v_i w_i v_uf v_if w_if) %s"
          % ({k: np.round(v, 4).tolist() for k, v in hits.items()}, mean, np.round(got / want - 1.0, 4).tolist()))
    assert abs(mean["merged"] - mean["oracle"]) <= 0.015, mean
    np.testing.assert_allclose(got[1:3], want[1:3], rtol=0.05)


def test_an_epoch_trained_in_parts_ranks_like_an_epoch_in_one_call_with_tags(c2_shape_jobs):
    """A multi-GPU rank trains an epoch in PARTS (rfm_fit_config.epoch_parts: one exchange window each).  On ONE GPU, no merge involved, the
    tags model of config 2's shape trained in 8 and in 24 parts per epoch must rank like the same model trained an epoch per launch.  It did
    not (tools/merge_tags_scan.py, profiles/r06_notes.md section 8): eight parts cost 5.0 points of hit_rate@10 with every norm in place --
    (1) the step producers' row sample was keyed by the launch's index within the CALL, so all parts of an epoch trained the tables on the
    same sampled rows, and (2) every part ran the epoch's table schedule in small (quota over its first 65 %, then quiet), which leaves the
    model a final quiet period an eighth as long.  Now the sample is keyed by the part as well and a part's launch takes its share of the
    EPOCH's schedule: measured +0.9 / +0.6 point at 8 / 24 parts against the one-launch epochs.  Two data seeds; neither more than 1.0 point below
    the one-launch engine, both within 1.5 of the oracle."""
    from rankfm_amd import EngineOptions, RankFM, evaluation
    from rankfm_amd.engine import DeviceSession
    import torch
    data, pending = c2_shape_jobs
    loss, F, ms = C2_VARIANTS["bpr_k32_tags"]
    hits = {"oracle": [], "one launch": [], "8 parts": [], "24 parts": []}
    for seed in (0, 1):
        d = data[seed]
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
        uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
        itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)

        def scored(weights):
            o = RankFM(factors=F, loss=loss, learning_rate=C2_TAG_LR, engine=EngineOptions(seed=100 + seed))
            np.random.seed(seed)
            o._init_all(train, uf, itf)
            for k, v in weights.items():
                setattr(o, k, np.ascontiguousarray(v))
            o.is_fit = True
            return evaluation.hit_rate(o, test, k=10)
        hits["oracle"].append(scored(pending[("bpr_k32_tags", seed)].get(timeout=1500)["weights"]))
        for name, parts in (("one launch", 1), ("8 parts", 8), ("24 parts", 24)):
            m = RankFM(factors=F, loss=loss, learning_rate=C2_TAG_LR, engine=EngineOptions(seed=100 + seed))
            np.random.seed(seed)
            m._init_all(train, uf, itf)
            w = {k: np.array(getattr(m, k), copy=True) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")}
            sess = DeviceSession(m.interactions, m.sample_weight, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, w, device=torch.device("cuda", 0),
                                 alpha=m.alpha, beta=m.beta, learning_rate=C2_TAG_LR, learning_schedule="constant", learning_exponent=0.25, max_samples=1,
                                 seed=100 + seed, has_user_features=1, has_item_features=1)
            for e in range(C2_TAG_EPOCHS):
                for k in range(parts):
                    sess.run(epochs=1, epoch_begin=e, part=(k, parts) if parts > 1 else None)
            hits[name].append(scored(sess.weights_to_host()))
    mean = {k: float(np.mean(v)) for k, v in hits.items()}
    print("tags model, epochs in parts on one GPU: hit_rate@10 %s means %s" % ({k: np.round(v, 4).tolist() for k, v in hits.items()}, mean))
    for name in ("8 parts", "24 parts"):
        # (not WORSE than the one-launch epochs by a point -- the defects cost five; parts rank +0.6 ... +1.0 above them here because an eighth
        #  of this small epoch is 1.4 segments per row group, i.e. fewer rows in flight: profiles/r06_notes.md section 8 -- and within 1.5 of the oracle)
        assert mean[name] >= mean["one launch"] - 0.010 and abs(mean[name] - mean["oracle"]) <= 0.015, mean


def test_asynchrony_term_by_itself_at_config2_shape(c2_shape_jobs):
    """The engine's -0.6 point against the reference at config 2's shape is the sum of two unrelated effects: its
    visiting order -- user segments of <= 32 rows in a keyed order -- ranks ~1 point BETTER than the reference's row-level shuffle even
    when run sequentially, and ~16 k rows in flight cost ~1.5 - 2 points.  This test isolates the second: the engine against the sequential
    oracle run in the ENGINE'S order (rankfm_amd.order.epoch_positions; reference sampler, same initial weights), BPR k = 32, five
    seeds.  Asserted: the order bonus is what the notes say (the ordered oracle ranks 0.3 ... 2.0 points above the row-shuffled one), and
    asynchronous execution + step damping cost at most 2.0 points against the ordered oracle (round 5 measured -1.87 at 16 k rows in
    flight, bound 2.5; -1.63 since the BPR kernel runs 192 workgroups = 12 k rows in flight: the bound follows) --
    the bound the production concurrency plan (rfm_api.hip "launch geometry") is held to; the user-visible bar, 1.0 point against the
    reference's algorithm, is test_default_engine_holds_the_quality_bar_at_config2_shape."""
    from rankfm_amd import EngineOptions, RankFM, evaluation
    data, pending = c2_shape_jobs
    loss, F, ms = C2_VARIANTS["bpr_k32_engine_order"]
    hits = {"ordered oracle": [], "shuffled oracle": [], "engine": []}
    for seed, d in data.items():
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        m = RankFM(factors=F, loss=loss, engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        m.fit(train, epochs=C2_SHAPE["E"])
        assert m.last_fit_report["geometry"]["segment_rows"] in (0, 32)
        hits["engine"].append(evaluation.hit_rate(m, test, k=10))
        for side, tag in (("ordered oracle", "bpr_k32_engine_order"), ("shuffled oracle", "bpr_k32")):
            o = RankFM(factors=F, loss=loss, engine=EngineOptions(seed=100 + seed))
            np.random.seed(seed)
            o._init_all(train)
            for k, v in pending[(tag, seed)].get(timeout=1500)["weights"].items():
                setattr(o, k, np.ascontiguousarray(v))
            o.is_fit = True
            hits[side].append(evaluation.hit_rate(o, test, k=10))
    mean = {k: float(np.mean(v)) for k, v in hits.items()}
    print("config-2 shape, asynchrony by itself: hit_rate@10 %s means %s: order bonus %+.2f point, engine against the ordered oracle %+.2f point"
          % ({k: np.round(v, 4).tolist() for k, v in hits.items()}, mean, 100 * (mean["ordered oracle"] - mean["shuffled oracle"]),
             100 * (mean["engine"] - mean["ordered oracle"])))
    assert 0.003 <= mean["ordered oracle"] - mean["shuffled oracle"] <= 0.020, mean
    assert mean["ordered oracle"] - mean["engine"] <= 0.020, mean


# ---- the table trainer's quota: no hump from x 0.5 to x 2 of the default -----------------------------------------
def _every_of(model, n_rows, epochs):
    """the trainer's effective quota of the last fit: one staged step per this many rows"""
    steps = model.last_fit_report["geometry"]["table_steps"]
    return max(1, int(round(float(n_rows) * epochs / max(steps, 1))))


def test_table_quota_sweep_on_the_reference_backed_feature_fixture():
    """The REFERENCE's own fits (tests/golden/quality_tags_spread.npz: forty runs, mean pinned to +-0.25 point): hit_rate@10 of the feature
    model at HALF and at TWICE the trainer's default quota as well (rfm_fit_tuning.table_every; four engine runs per data seed and setting:
    twenty runs, whose mean still moves by +-0.3 point).  Measured over four sweeps in round 5 (tools/feature_quality.py): half the
    default's spacing 0.4795 ... 0.4829, twice 0.4824 ... 0.4874.  Held to 1.0 point (round 5: 1.5, against the reference's five single
    runs); the default itself is held to 1.0 over forty runs by test_feature_model_matches_the_reference."""
    from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic
    want = float(load_golden("quality", "tags_spread")["tags_order_only"][:, :, 0].mean())      # (the reference's forty runs: see above)
    data = []
    for seed in range(5):
        d = synthetic.make_planted(seed=seed, n_users=3000, n_items=2000, mean_degree=100.0, n_tags=8)
        data.append((pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"]),
                     pd.DataFrame(np.column_stack([np.arange(len(d["user_tags"])), d["user_tags"]])),
                     pd.DataFrame(np.column_stack([np.arange(len(d["item_tags"])), d["item_tags"]]))))
    probe = RankFM(factors=20, loss="bpr", learning_rate=0.03)
    np.random.seed(0)
    probe.fit(data[0][0], user_features=data[0][2], item_features=data[0][3], epochs=5)
    every = _every_of(probe, len(data[0][0]), 5)
    got = {}
    for name, ev in (("half", max(1, every // 2)), ("twice", every * 2)):
        hits = []
        for seed, (train, test, uf, itf) in enumerate(data):
            for run in range(4):
                m = RankFM(factors=20, loss="bpr", learning_rate=0.03, engine=EngineOptions(tune={"table_every": ev}))
                np.random.seed(seed)
                m.fit(train, user_features=uf, item_features=itf, epochs=5)
                hits.append(evaluation.hit_rate(m, test, k=10))
        got[name] = float(np.mean(hits))
    print("feature fixture, table quota sweep: default every %d-th row; hit_rate@10 %s, reference %.4f" % (every, got, want))
    for name, h in got.items():
        assert abs(h - want) <= 0.010, (name, h, want)


def test_table_quota_sweep_at_config2_shape_with_tags(c2_shape_jobs):
    """The same sweep at config 2's shape with 8 + 8 tags (the features kernels on a full chip) against the sequential oracle: three data
    seeds x FOUR engine seeds per setting (six runs per setting left the mean +-0.35 point of run-to-run noise: sweeps of the same build
    measured -0.75 ... -1.60 at one setting; the asynchrony term at this shape is ~ -1 point by itself, test_asynchrony_term_by_itself).  Since round 5 a quota DENSER than the default makes the trainer stop by itself once 80 % of a
    launch's segments are handed out (kTableQuietFrom), so that it no longer costs the rows their quiet period -- round 4 measured -3.8
    points at every 250th row.  A tags model's hit rate moves by +-0.5 point with the engine's seed and more under a dense quota
    (profiles/r05_notes.md section 8), so with six runs per setting the default is held to 1.5 points (measured -0.17 ... -0.88 over five
    sweeps; the four-seed test above holds it to 1.0 and is the bar proper).  Round 6: the step producers PACE the quota over the first 65 %
    of a launch's segments (SgdArgs::table_pace) -- a sparser quota used to be worked off in the launch's first third, which was the whole of
    its cost (twice the spacing: -1.4 points bunched, -0.8 spread; profiles/r06_notes.md section 4: twelve runs per setting 0.3712 / 0.3714 /
    0.3681 at twice / once / half the spacing against 0.3647 unpaced) -- and all three settings are held to 1.5 points (round 5: 2.0 at
    half and at twice the spacing)."""
    from rankfm_amd import EngineOptions, RankFM, evaluation
    data, pending = c2_shape_jobs
    loss, F, ms = C2_VARIANTS["bpr_k32_tags"]
    frames, oracle_hits = {}, []
    for seed in range(C2_TAG_SEEDS):
        d = data[seed]
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
        uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
        itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
        frames[seed] = (train, test, uf, itf)
        o = RankFM(factors=F, loss=loss, max_samples=ms, learning_rate=C2_TAG_LR, engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        o._init_all(train, uf, itf)
        for k, v in pending[("bpr_k32_tags", seed)].get(timeout=1500)["weights"].items():
            setattr(o, k, np.ascontiguousarray(v))
        o.is_fit = True
        oracle_hits.append(evaluation.hit_rate(o, test, k=10))
    want = float(np.mean(oracle_hits))
    every, got = None, {}
    for name in ("default", "half", "twice"):
        hits = []
        for seed, (train, test, uf, itf) in frames.items():
            for engine_seed in (100 + seed, 1100 + seed, 2100 + seed, 3100 + seed):
                tune = {} if name == "default" else {"table_every": max(1, every // 2) if name == "half" else every * 2}
                m = RankFM(factors=F, loss=loss, max_samples=ms, learning_rate=C2_TAG_LR, engine=EngineOptions(seed=engine_seed, tune=tune))
                np.random.seed(seed)
                m.fit(train, uf, itf, epochs=C2_TAG_EPOCHS)
                hits.append(evaluation.hit_rate(m, test, k=10))
                if every is None:
                    every = _every_of(m, len(train), C2_TAG_EPOCHS)
        got[name] = float(np.mean(hits))
    print("config-2 shape with tags, table quota sweep: default every %d-th row; hit_rate@10 %s, oracle %.4f" % (every, got, want))
    assert abs(got["default"] - want) <= 0.015, (got, want)
    # (twice the spacing sat at -1.5 in two sweeps of twelve runs while its FIRST epoch was left unpaced -- the exception made for the
    #  default quota, whose first-epoch log-likelihood at config 4 is better unpaced; a sparser quota is now paced from the first epoch on
    #  (rfm_api.hip): 24 runs per setting, default -0.60, twice -0.75, three times -0.99, half -1.06: profiles/r06_raw/r06aa, r06ab)
    assert abs(got["half"] - want) <= 0.015 and abs(got["twice"] - want) <= 0.015, (got, want)
