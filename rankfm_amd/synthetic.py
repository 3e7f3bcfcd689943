"""Synthetic implicit-feedback workloads of the BASELINE.json configurations (SURVEY.md §8 d).

u ~ Uniform{0..U-1}; i ~ Zipf-like (popularity exponent `zipf_s`) over a random item permutation, or uniform when
zipf_s == 0; (u, i) pairs de-duplicated and topped up to exactly N; ids are already 0-based int32 indexes;
sample_weight = 1.  Everything is generated from seeds, so the GPU box needs no data files.
"""
import numpy as np

from ._rankfm import UserItemsCSR

# BASELINE.json configs (C1 is the reference's CPU-runnable MovieLens-1M-shaped case)
CONFIGS = {
    "C1": dict(n_users=6040, n_items=3706, n_interactions=1_000_000, factors=20, loss="bpr", max_samples=1),
    "C2": dict(n_users=100_000, n_items=50_000, n_interactions=5_000_000, factors=64, loss="bpr", max_samples=1),
    "C3": dict(n_users=100_000, n_items=50_000, n_interactions=5_000_000, factors=64, loss="warp", max_samples=50),
    "C4": dict(n_users=1_000_000, n_items=200_000, n_interactions=50_000_000, factors=64, loss="bpr", max_samples=1,
               n_user_features=32, n_item_features=32),
    # not in BASELINE.json: config 4's feature setting at config 2's size (single-GPU feature-path measurements)
    "C4S": dict(n_users=100_000, n_items=50_000, n_interactions=5_000_000, factors=64, loss="bpr", max_samples=1,
                n_user_features=32, n_item_features=32),
    "C5": dict(n_users=5_000_000, n_items=1_000_000, n_interactions=500_000_000, factors=128, loss="warp", max_samples=50),
}


def make_interactions(n_users, n_items, n_interactions, seed=0, zipf_s=1.0):
    """int32 [N,2] unique (user, item) pairs in random order, and the users' CSR item lists"""
    if n_interactions > n_users * (n_items - 1):
        raise ValueError("too dense: every user needs at least one unobserved item")
    rng = np.random.default_rng(seed)
    if zipf_s > 0:
        pop = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_s)
        cdf = np.cumsum(pop / pop.sum())
        item_of_rank = rng.permutation(n_items)
    keys = np.zeros(0, dtype=np.int64)
    while len(keys) < n_interactions:
        want = int((n_interactions - len(keys)) * 1.25) + 1024
        u = rng.integers(0, n_users, want, dtype=np.int64)
        if zipf_s > 0:
            i = item_of_rank[np.minimum(np.searchsorted(cdf, rng.random(want)), n_items - 1)].astype(np.int64)
        else:
            i = rng.integers(0, n_items, want, dtype=np.int64)
        keys = np.unique(np.concatenate([keys, u * n_items + i]))
        # no user may hold every item (the rejection sampler needs one unobserved item per user)
        deg = np.bincount(keys // n_items, minlength=n_users)
        if deg.max() >= n_items:
            full = np.flatnonzero(deg >= n_items)
            keys = keys[~(np.isin(keys // n_items, full) & (keys % n_items == 0))]
    keys = rng.permutation(keys)[:n_interactions]
    pairs = np.empty((n_interactions, 2), dtype=np.int32)
    pairs[:, 0] = keys // n_items
    pairs[:, 1] = keys % n_items
    csr = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], n_users)
    return pairs, csr


def make_features(n_rows, n_features, seed, density=0.25):
    """dense 0/1 tag features ~ Bernoulli(density), float32 [n_rows, n_features]"""
    rng = np.random.default_rng(seed)
    return (rng.random((n_rows, n_features)) < density).astype(np.float32)


def init_weights(n_users, n_items, factors, n_user_features=0, n_item_features=0, sigma=0.1, alpha=0.01, beta=0.1, seed=1492):
    """the reference's initialisation law (rankfm/rankfm.py:214-244) from a seeded generator"""
    rng = np.random.default_rng(seed)
    P, Q = max(n_user_features, 1), max(n_item_features, 1)
    scale = (alpha / beta) * sigma
    return dict(
        w_i=np.zeros(n_items, dtype=np.float32),
        w_if=np.zeros(Q, dtype=np.float32),
        v_u=rng.normal(0, sigma, (n_users, factors)).astype(np.float32),
        v_i=rng.normal(0, sigma, (n_items, factors)).astype(np.float32),
        v_uf=(rng.normal(0, scale, (P, factors)) if n_user_features else np.zeros((P, factors))).astype(np.float32),
        v_if=(rng.normal(0, scale, (Q, factors)) if n_item_features else np.zeros((Q, factors))).astype(np.float32),
    )


def make_planted(n_users=6040, n_items=3706, rank=16, seed=0, mean_degree=165.0, holdout=0.25, n_tags=0):
    """MovieLens-1M-shaped surrogate with planted low-rank structure: a discriminating ranking task
    on which the popularity baseline scores hit_rate@10 ~0.36 and the reference's BPR ~0.82.

    score(u, i) = 3/4 <A_u, B_i> + popularity(i) + Gumbel noise; user u observes her top-deg_u items,
    deg_u ~ clipped log-normal; observed pairs are shuffled and split train/test.  With n_tags > 0 also returns binary
    user/item tag features that carry signal (signs of the first planted dimensions).
    Returns dict(train [n,2] int32, test [m,2] int32, user_tags [U,n_tags] | None, item_tags [I,n_tags] | None).
    """
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n_users, rank)).astype(np.float32)
    B = rng.normal(size=(n_items, rank)).astype(np.float32)
    pop = np.empty(n_items, dtype=np.float32)
    pop[rng.permutation(n_items)] = -0.5 * np.log(np.arange(1, n_items + 1))
    deg = np.clip(rng.lognormal(np.log(mean_degree) - 0.5, 1.0, n_users), 20, n_items // 2).astype(np.int64)
    users, items = [], []
    step = 512
    for u0 in range(0, n_users, step):
        S = 0.75 * A[u0:u0 + step] @ B.T + pop + rng.gumbel(size=(min(step, n_users - u0), n_items)).astype(np.float32)
        order = np.argsort(-S, axis=1)
        for r in range(S.shape[0]):
            d = deg[u0 + r]
            users.append(np.full(d, u0 + r, dtype=np.int32))
            items.append(order[r, :d].astype(np.int32))
    pairs = np.stack([np.concatenate(users), np.concatenate(items)], 1)
    pairs = pairs[rng.permutation(len(pairs))]
    n_test = int(len(pairs) * holdout)
    out = dict(test=np.ascontiguousarray(pairs[:n_test]), train=np.ascontiguousarray(pairs[n_test:]), user_tags=None, item_tags=None)
    if n_tags:
        out["user_tags"] = (A[:, :n_tags] > 0).astype(np.float32)
        out["item_tags"] = (B[:, :n_tags] > 0).astype(np.float32)
    return out
