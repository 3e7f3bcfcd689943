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
