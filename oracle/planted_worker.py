"""Test infrastructure: one seed of a planted ranking problem and the sequential oracle's fit on it, as a picklable task for a
process pool (the quality tests farm their seeds out: data generation and the oracle are CPU work, the GPU side stays in the test
process).  Never imported by the product."""
import numpy as np


def fit_planted(args):
    """(seed, n_users, n_items, factors, epochs) -> dict(train, test, oracle weights); the oracle starts from the weights
    `np.random.seed(seed); RankFM(...)._init_all(train)` draws -- what the engine side of the test starts from as well -- and
    draws its negatives like the reference: uniformly over the whole catalogue (counter RNG, its own keyed visiting order)"""
    import pandas as pd
    from oracle import oracle as orc
    from rankfm_amd import EngineOptions, RankFM, synthetic
    seed, n_users, n_items, factors, epochs = args
    orc.build()
    d = synthetic.make_planted(n_users, n_items, seed=seed)
    train = pd.DataFrame(d["train"], columns=["u", "i"])
    m = RankFM(factors=factors, loss="bpr", engine=EngineOptions(seed=100 + seed))
    np.random.seed(seed)
    m._init_all(train)
    orc.fit(m.interactions, m.sample_weight, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i, m.v_uf,
            m.v_if, m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent, 1, epochs, perms=None,
            rng_mode=orc.RNG_COUNTER, seed=100 + seed, membership="binary")
    return dict(seed=seed, train=d["train"], test=d["test"], weights={k: getattr(m, k) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")})


def fit_pairs(args):
    """(tag, seed, train [n,2] int32, factors, epochs, loss, max_samples[, user_tags, item_tags, learning_rate[, seg_rows]]) -> dict(tag, seed, oracle
    weights): the sequential oracle on GIVEN
    training pairs (the config-2-shaped quality tests generate their data on the GPU and hand it over), from the weights
    `np.random.seed(seed); RankFM(...)._init_all(train)` draws, negatives drawn like the reference (uniformly over the catalogue)"""
    import pandas as pd
    from oracle import oracle as orc
    from rankfm_amd import EngineOptions, RankFM
    tag, seed, train, factors, epochs, loss, max_samples = args[:7]
    user_tags, item_tags, lr = (args[7], args[8], args[9]) if len(args) > 7 else (None, None, 0.1)
    seg_rows = args[10] if len(args) > 10 else 0       # > 0: visit the rows in the ENGINE'S order (user segments of <= seg_rows rows)
    orc.build()
    m = RankFM(factors=factors, loss=loss, max_samples=max_samples, learning_rate=lr, engine=EngineOptions(seed=100 + seed))
    np.random.seed(seed)
    uf = itf = None
    if user_tags is not None:      # (feature rows for exactly the users / items of the training data)
        us, its = np.unique(train[:, 0]), np.unique(train[:, 1])
        uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(user_tags[us])], axis=1)
        itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(item_tags[its])], axis=1)
    m._init_all(pd.DataFrame(train, columns=["u", "i"]), uf, itf)
    pairs, sw, perms = m.interactions, m.sample_weight, None
    if seg_rows:
        # the engine's visiting order (rankfm_amd.order.epoch_positions: user segments in a keyed order, rows inside a segment in a keyed
        # order) on the rows sorted like the CSR lists -- sequential, no asynchrony: what is left between this fit and the engine's is
        # asynchronous execution (+ the step damping)
        from rankfm_amd import order
        by = np.lexsort((pairs[:, 1], pairs[:, 0]))
        pairs, sw = np.ascontiguousarray(pairs[by]), np.ascontiguousarray(sw[by])
        perms = np.stack([order.epoch_positions(m.user_items.offsets, 100 + seed, e, seg_rows) for e in range(epochs)]).astype(np.int32)
    orc.fit(pairs, sw, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i, m.v_uf,
            m.v_if, m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent, 1 if loss == "bpr" else max_samples, epochs,
            perms=perms, rng_mode=orc.RNG_COUNTER, seed=100 + seed, membership="binary")
    return dict(tag=tag, seed=seed, weights={k: getattr(m, k) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")})
