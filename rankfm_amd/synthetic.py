"""Synthetic implicit-feedback workloads of the BASELINE.json configurations (SURVEY.md §8 d).

u ~ Uniform{0..U-1}; i ~ Zipf-like (popularity exponent `zipf_s`) over a random item permutation, or uniform when
zipf_s == 0; (u, i) pairs de-duplicated and topped up to exactly N; ids are already 0-based int32 indexes;
sample_weight = 1.  Everything is generated from seeds, so the GPU box needs no data files.
"""
import os

import numpy as np

from ._rankfm import UserItemsCSR

# BASELINE.json configs (C1 is the reference's CPU-runnable MovieLens-1M-shaped case)
CONFIGS = {
    "C1": dict(n_users=6040, n_items=3706, n_interactions=1_000_000, factors=20, loss="bpr", max_samples=1),
    "C2": dict(n_users=100_000, n_items=50_000, n_interactions=5_000_000, factors=64, loss="bpr", max_samples=1),
    "C3": dict(n_users=100_000, n_items=50_000, n_interactions=5_000_000, factors=64, loss="warp", max_samples=50),
    # learning_rate 0.03: on 32 + 32 dense Bernoulli(0.25) tags the REFERENCE ALGORITHM ITSELF diverges at its default 0.1
    # ("[w_i] are not finite" from the sequential oracle within one epoch; stable at <= 0.05)
    "C4": dict(n_users=1_000_000, n_items=200_000, n_interactions=50_000_000, factors=64, loss="bpr", max_samples=1,
               n_user_features=32, n_item_features=32, learning_rate=0.03),
    # not in BASELINE.json: config 4's feature setting at config 2's size (single-GPU feature-path measurements)
    "C4S": dict(n_users=100_000, n_items=50_000, n_interactions=5_000_000, factors=64, loss="bpr", max_samples=1,
                n_user_features=32, n_item_features=32, learning_rate=0.03),
    "C5": dict(n_users=5_000_000, n_items=1_000_000, n_interactions=500_000_000, factors=128, loss="warp", max_samples=50),
}


def make_interactions(n_users, n_items, n_interactions, seed=0, zipf_s=1.0, item_seed=None):
    """int32 [N,2] unique (user, item) pairs in random order, and the users' CSR item lists.  `item_seed` fixes the
    popularity ranking of the items independently of `seed` (user blocks of one data set share the catalogue)."""
    if n_interactions > n_users * (n_items - 1):
        raise ValueError("too dense: every user needs at least one unobserved item")
    rng = np.random.default_rng(seed)
    if zipf_s > 0:
        pop = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_s)
        cdf = np.cumsum(pop / pop.sum())
        item_of_rank = (rng if item_seed is None else np.random.default_rng(item_seed)).permutation(n_items)
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


# ---- one data set per BASELINE config, defined block-wise so that a rank can generate exactly its own user shard ----
# A config's data set is the union of USER_BLOCKS user blocks: block b holds users [b U/64, (b+1) U/64) with N/64 interactions
# drawn like make_interactions over the SHARED item catalogue (one popularity ranking, `item_seed`).  Every block has the same
# number of interactions, so the interaction-balanced user split of distributed.shard_boundaries falls on block boundaries for
# world sizes dividing 64: rank r of W owns blocks [64 r / W, 64 (r+1) / W).  Config 5 (500 M interactions) can therefore be
# sharded over 8 GPUs without any process ever materialising the whole data set.
USER_BLOCKS = 64


def config_shard_blocks(rank, world, n_blocks=USER_BLOCKS):
    if n_blocks % world:
        raise ValueError("world size must divide %d" % n_blocks)
    per = n_blocks // world
    return range(rank * per, (rank + 1) * per)


def make_config_shard(name, rank=0, world=1, zipf_s=1.0, data_seed=0, init_seed=1492, blocks=None):
    """the user shard `rank` of `world` of BASELINE config `name` (user indexes rebased to the shard): dict with
    interactions, sample_weight, csr_offsets, csr_items, x_uf, x_if (whole catalogue), weights (v_u of the shard + the
    item-side tables, identical on every rank), user_lo, user_hi"""
    cfg = CONFIGS[name]
    U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
    P, Q = cfg.get("n_user_features", 0), cfg.get("n_item_features", 0)
    blocks = list(config_shard_blocks(rank, world) if blocks is None else blocks)
    # RFM_DATA_CACHE=<dir>: keep generated shards on disk (profiling scripts start bench.py many times on the same data)
    cache = os.environ.get("RFM_DATA_CACHE", "")
    cpath = os.path.join(cache, "%s_b%d-%d_z%g_d%d_i%d.npz" % (name, blocks[0], blocks[-1], zipf_s, data_seed, init_seed)) if cache else ""
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        out = {k: z[k] for k in z.files if not k.startswith("w__")}
        out["weights"] = {k[3:]: z[k] for k in z.files if k.startswith("w__")}
        out.update(user_lo=int(out["user_lo"]), user_hi=int(out["user_hi"]), config=cfg)
        return out
    ub, nb = U // USER_BLOCKS, N // USER_BLOCKS
    pairs, items, offs, vus, xus = [], [], [np.zeros(1, np.int64)], [], []
    for k, b in enumerate(blocks):
        p, csr = make_interactions(ub, I, nb, seed=[data_seed, 1 + b], zipf_s=zipf_s, item_seed=[data_seed, 0])
        p[:, 0] += k * ub
        pairs.append(p)
        items.append(csr.items)
        offs.append(csr.offsets[1:] + k * nb)
        vus.append(np.random.default_rng([init_seed, 1 + b]).normal(0, 0.1, (ub, F)).astype(np.float32))
        if P:
            xus.append(make_features(ub, P, [data_seed, 100 + b]))
    n_local = ub * len(blocks)
    w = init_weights(1, I, F, P, Q, seed=[init_seed, 0])
    w["v_u"] = np.concatenate(vus)
    pairs = np.concatenate(pairs)
    out = dict(interactions=pairs, sample_weight=np.ones(len(pairs), np.float32), csr_offsets=np.concatenate(offs),
               csr_items=np.concatenate(items), x_uf=np.concatenate(xus) if P else np.zeros((n_local, 1), np.float32),
               x_if=make_features(I, Q, [data_seed, 99]) if Q else np.zeros((I, 1), np.float32), weights=w,
               user_lo=blocks[0] * ub, user_hi=(blocks[-1] + 1) * ub, config=cfg)
    if cpath:
        os.makedirs(cache, exist_ok=True)
        np.savez(cpath, **{k: v for k, v in out.items() if k not in ("weights", "config")}, **{"w__" + k: v for k, v in w.items()})
    return out


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


def _planted_chunk(args):
    """top-deg items of a block of users under the planted score model (worker of make_planted_large)"""
    seed, c, u0, u1, n_items, rank, deg, B, pop = args[:9]
    rng = np.random.default_rng([seed, 1000 + c])
    A = rng.normal(size=(u1 - u0, rank)).astype(np.float32)
    try:        # one BLAS thread per worker: a pool of workers x a 256-thread BLAS each thrashes a many-core host
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            S = 0.75 * (A @ B.T) + pop
    except ImportError:
        S = 0.75 * (A @ B.T) + pop
    S += rng.gumbel(size=S.shape).astype(np.float32)
    d_max = int(deg.max())
    top = np.argpartition(-S, d_max - 1, axis=1)[:, :d_max]                     # the d_max best of every row, unordered
    top = np.take_along_axis(top, np.argsort(-np.take_along_axis(S, top, axis=1), axis=1), axis=1)     # ... best first
    keep = np.arange(d_max)[None, :] < deg[:, None]
    users = np.repeat(np.arange(u0, u1, dtype=np.int32), deg)
    return users, top[keep].astype(np.int32)


def make_planted_large(n_users, n_items, rank=16, seed=0, mean_degree=60.0, max_degree=1000, holdout=0.25, processes=None, chunk=2048,
                       pop_weight=0.5):
    """make_planted's score model (3/4 <A_u, B_i> + popularity + Gumbel noise, clipped log-normal degrees) for problems of
    BASELINE config 2's size: user blocks are generated independently (seeded per block, in a process pool) and the top items of
    a row are found by partial selection instead of a full sort.  NOT the same random stream as make_planted -- the reference-minted
    quality fixtures stay on make_planted.  Returns dict(train [n,2] int32, test [m,2] int32)."""
    import multiprocessing as mp
    rng = np.random.default_rng([seed, 0])
    B = rng.normal(size=(n_items, rank)).astype(np.float32)
    pop = np.empty(n_items, dtype=np.float32)
    # (pop_weight: the popularity term is -pop_weight log(rank); 1.0 gives item frequencies close to the Zipf(1) of the BASELINE configs)
    pop[rng.permutation(n_items)] = -pop_weight * np.log(np.arange(1, n_items + 1))
    deg = np.clip(rng.lognormal(np.log(mean_degree) - 0.5, 1.0, n_users), 10, min(max_degree, n_items // 2)).astype(np.int64)
    tasks = [(seed, c, u0, min(u0 + chunk, n_users), n_items, rank, deg[u0:u0 + chunk], B, pop)
             for c, u0 in enumerate(range(0, n_users, chunk))]
    processes = min(len(tasks), processes or min(32, os.cpu_count() or 1))
    if processes > 1:
        with mp.get_context("fork").Pool(processes) as pool:
            parts = pool.map(_planted_chunk, tasks)
    else:
        parts = [_planted_chunk(t) for t in tasks]
    pairs = np.stack([np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])], 1)
    pairs = pairs[rng.permutation(len(pairs))]
    n_test = int(len(pairs) * holdout)
    return dict(test=np.ascontiguousarray(pairs[:n_test]), train=np.ascontiguousarray(pairs[n_test:]), user_tags=None, item_tags=None)


def make_planted_large_device(n_users, n_items, rank=16, seed=0, mean_degree=60.0, max_degree=1000, holdout=0.25, pop_weight=0.5, device="cuda",
                              chunk=4096, n_tags=0):
    """make_planted_large's score model evaluated on the GPU (torch: the 5e9 scores of a config-2-sized problem take a second instead
    of the ten CPU-minutes a single process needs) -- test / measurement data only, a random stream of its own (torch generators seeded
    with `seed`).  Returns dict(train [n,2] int32, test [m,2] int32, user_tags / item_tags [*, n_tags] float32 | None) as numpy arrays;
    the pairs do not depend on `n_tags`."""
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(1_000_003 * int(seed) + 17)
    rng = np.random.default_rng([seed, 7])
    B = torch.randn((n_items, rank), generator=g, device=dev, dtype=torch.float32)
    pop = torch.empty(n_items, device=dev, dtype=torch.float32)
    pop[torch.as_tensor(rng.permutation(n_items), device=dev)] = -pop_weight * torch.log(torch.arange(1, n_items + 1, device=dev, dtype=torch.float32))
    deg = np.clip(rng.lognormal(np.log(mean_degree) - 0.5, 1.0, n_users), 10, min(max_degree, n_items // 2)).astype(np.int64)
    users, items, user_tags = [], [], []
    for u0 in range(0, n_users, chunk):
        u1 = min(u0 + chunk, n_users)
        A = torch.randn((u1 - u0, rank), generator=g, device=dev, dtype=torch.float32)
        if n_tags:
            user_tags.append((A[:, :n_tags] > 0).to(torch.float32).cpu().numpy())
        S = 0.75 * (A @ B.T) + pop
        U01 = torch.rand(S.shape, generator=g, device=dev, dtype=torch.float32).clamp_(1e-12, 1.0 - 1e-7)
        S -= torch.log(-torch.log(U01))                                          # + Gumbel(0, 1)
        d = torch.as_tensor(deg[u0:u1], device=dev)
        d_max = int(d.max().item())
        top = torch.topk(S, d_max, dim=1, sorted=True).indices                   # the d_max best of every row, best first
        keep = torch.arange(d_max, device=dev)[None, :] < d[:, None]
        users.append(np.repeat(np.arange(u0, u1, dtype=np.int32), deg[u0:u1]))
        items.append(top[keep].to(torch.int32).cpu().numpy())
    pairs = np.stack([np.concatenate(users), np.concatenate(items)], 1)
    pairs = pairs[rng.permutation(len(pairs))]
    n_test = int(len(pairs) * holdout)
    out = dict(test=np.ascontiguousarray(pairs[:n_test]), train=np.ascontiguousarray(pairs[n_test:]), user_tags=None, item_tags=None)
    if n_tags:      # binary tags that carry signal: the signs of the first planted dimensions (like make_planted)
        out["user_tags"] = np.concatenate(user_tags)
        out["item_tags"] = (B[:, :n_tags] > 0).to(torch.float32).cpu().numpy()
    return out
