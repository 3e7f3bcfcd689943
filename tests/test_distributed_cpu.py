"""The N > 1 path on CPU: world_size-2 gloo processes run the user-sharded trainer (rankfm_amd/distributed.py) with the
CPU oracle standing in for the HIP epoch (tests may use the oracle; the product wiring for the GPU is
make_device_trainer).  Checks the sharding, the flat-bucket all-reduce of item-side deltas, and that two shards
trained with one exchange per epoch track single-process sequential training."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

from rankfm_amd import synthetic
from rankfm_amd.distributed import SHARED_NAMES, SharedTables, ShardedTrainer, shard_boundaries, take_user_shard

U, I, N, F, EPOCHS, DAMPING = 300, 200, 12000, 8, 3, 32.0


def _problem():
    pairs, csr = synthetic.make_interactions(U, I, N, seed=4, zipf_s=0.8)
    w = synthetic.init_weights(U, I, F, seed=5)
    return pairs, csr, np.ones(N, np.float32), w


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_epoch_fn(shard, x_if, seed):
    """one sequential epoch of the oracle on this rank's shard, in place on the bucket views"""
    from oracle import oracle as orc
    x_uf = np.zeros((len(shard["v_u"]), 1), np.float32)

    def fn(views, epoch):
        t = {k: views[k].numpy() for k in SHARED_NAMES}          # numpy views share the bucket's memory
        return orc.fit(shard["interactions"], shard["sample_weight"], shard["csr_offsets"], shard["csr_items"], x_uf, x_if,
                       t["w_i"], t["w_if"], shard["v_u"], t["v_i"], t["v_uf"], t["v_if"], 0.01, 0.1, 0.1, "constant", 0.25,
                       1, 1, perms=None, rng_mode=orc.RNG_COUNTER, seed=seed, epoch_begin=epoch, membership="binary")
    return fn


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs, csr, sw, w = _problem()
    bounds = shard_boundaries(csr.offsets, world)
    shard = take_user_shard(pairs, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), w["v_u"], bounds[rank], bounds[rank + 1])
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    shared.set_merge_damping(np.bincount(pairs[:, 1], minlength=I), world, damping=DAMPING)
    trainer = ShardedTrainer(shared, _local_epoch_fn(shard, np.zeros((I, 1), np.float32), seed=100 + rank))
    lls = []
    for e in range(EPOCHS):
        lls.append(float(trainer.run_epoch(e)["ll"][0]))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), flat=shared.flat.numpy(), v_u=shard["v_u"], ll=np.array(lls),
             lo=bounds[rank], hi=bounds[rank + 1])
    dist.destroy_process_group()


def test_shard_boundaries_balance_interactions():
    _, csr, _, _ = _problem()
    for world in (1, 2, 3, 8):
        b = shard_boundaries(csr.offsets, world)
        assert b[0] == 0 and b[-1] == U and np.all(np.diff(b) >= 0) and len(b) == world + 1
        loads = np.diff(csr.offsets[b])
        assert loads.sum() == N and loads.max() <= N / world * 1.15 + csr.offsets[1:].max() * 0 + 200
    pairs, csr, sw, w = _problem()
    b = shard_boundaries(csr.offsets, 2)
    parts = [take_user_shard(pairs, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), w["v_u"], b[r], b[r + 1]) for r in range(2)]
    assert sum(len(p["interactions"]) for p in parts) == N
    assert all(p["interactions"][:, 0].max() < len(p["v_u"]) and p["csr_offsets"][0] == 0 for p in parts)
    assert parts[1]["csr_offsets"][-1] == len(parts[1]["csr_items"])


def test_two_gloo_ranks_exchange_item_deltas(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    # (1) replicas agree bit for bit after the exchange
    assert np.array_equal(r[0]["flat"], r[1]["flat"])

    # (2) the exchange is exactly "epoch start + scale * sum over ranks of local deltas": replay both ranks in this process
    pairs, csr, sw, w = _problem()
    bounds = shard_boundaries(csr.offsets, world)
    shards = [take_user_shard(pairs, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), w["v_u"], bounds[k], bounds[k + 1])
              for k in range(world)]
    ref = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    ref.set_merge_damping(np.bincount(pairs[:, 1], minlength=I), world, damping=DAMPING)
    assert ref.merge_scale.min() >= 0.5 and ref.merge_scale.max() <= 1.0
    for e in range(EPOCHS):
        start = ref.flat.clone()
        total = torch.zeros_like(start)
        for k in range(world):
            ref.flat.copy_(start)
            _local_epoch_fn(shards[k], np.zeros((I, 1), np.float32), seed=100 + k)(ref.views, e)
            total += ref.flat - start
        ref.flat.copy_(start + ref.merge_scale * total)
    np.testing.assert_allclose(r[0]["flat"], ref.flat.numpy(), rtol=0, atol=1e-6)
    for k in range(world):
        np.testing.assert_allclose(r[k]["v_u"], shards[k]["v_u"], rtol=0, atol=1e-6)

    # (3) and it learns like single-process sequential training on the whole data (statistical)
    o = {k: v.copy() for k, v in w.items()}
    out = oracle.fit(pairs, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), o["w_i"],
                     o["w_if"], o["v_u"], o["v_i"], o["v_uf"], o["v_if"], 0.01, 0.1, 0.1, "constant", 0.25, 1, EPOCHS,
                     perms=None, rng_mode=oracle.RNG_COUNTER, seed=1, membership="binary")
    ll = r[0]["ll"] + r[1]["ll"]
    assert ll[-1] > ll[0]
    # one exchange per epoch on a 12 k-row toy problem: each rank trains half an epoch blind to the other, so the first
    # epoch lags the sequential run by a few percent (at BASELINE sizes an epoch is millions of rows per rank)
    np.testing.assert_allclose(ll, out["ll"], rtol=0.08)
    w_i = ref.views["w_i"].numpy()
    assert abs(np.linalg.norm(w_i) - np.linalg.norm(o["w_i"])) < 0.25 * np.linalg.norm(o["w_i"])
    assert np.corrcoef(w_i, o["w_i"])[0, 1] > 0.9


def _fused_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, _, w = _problem()
    rng = np.random.default_rng(10 + rank)
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    counts = rng.integers(0, 400, I).astype(np.float64) * (rng.random(I) < 0.7)
    shared.set_merge_curvature(counts, world, learning_rate=0.1, mean_vu2=0.5 + rank, n_users=10 + rank)     # (everybody starts from the mean over ALL ranks' users)
    out = {}
    for x, vu2 in enumerate((3.0 + rank, 7.0 + 2 * rank)):                                       # two exchanges: the second uses the first's mean
        shared.begin_epoch()
        delta = torch.as_tensor(rng.normal(0, 0.01, shared._tail_at).astype(np.float32))
        shared.flat[:shared._tail_at] += delta
        flag = shared.exchange_fused(None, torch.tensor(vu2 * (10 + rank), dtype=torch.float64), 10 + rank, failed=False)
        out["flat%d" % x], out["delta%d" % x], out["flag%d" % x] = shared.flat.numpy().copy(), delta.numpy(), float(flag)
    np.savez(os.path.join(out_dir, "fused%d.npz" % rank), counts=counts, start=SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu")).flat.numpy(), **out)
    dist.destroy_process_group()


def test_fused_exchange_is_the_curvature_rule_in_one_all_reduce(tmp_path):
    """SharedTables.exchange_fused (round 4: deltas, curvature terms, |v_u|^2 sums and failure flags in ONE all-reduce of the bucket,
    no host round trip) against the rule computed by hand from curvature_log_rho / curvature_terms / curvature_scales: the first
    exchange uses the mean |v_u|^2 over all ranks' users from arming time (a sum / count all-reduce: rank 0 may own no users), the second
    the mean the ranks summed during the first."""
    from rankfm_amd.distributed import curvature_log_rho, curvature_scales, curvature_terms
    world = 2
    mp.spawn(_fused_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("fused%d.npz" % k)) for k in range(world)]
    ref = SharedTables({k: _problem()[3][k] for k in SHARED_NAMES}, torch.device("cpu"))
    T = ref._tail_at
    cur = r[0]["start"][:T].astype(np.float64)
    mean = (0.5 * 10 + 1.5 * 11) / 21.0
    for x, vu2 in enumerate(((3.0, 4.0), (7.0, 9.0))):
        assert np.array_equal(r[0]["flat%d" % x], r[1]["flat%d" % x])                             # replicas agree bit for bit
        log_rho = curvature_log_rho(0.1, SharedTables.CURVATURE_FACTORS, SharedTables.CURVATURE_BIASES, mean)
        sv, sb = curvature_scales(sum(curvature_terms(r[k]["counts"], log_rho) for k in range(world)), log_rho, world)
        scale = np.full(T, 1.0 / world)                   # (the alignment padding between the tables; the feature tables: the ranks' mean)
        a = ref._starts["v_i"]; scale[a:a + ref._sizes["v_i"]] = np.repeat(sv.numpy(), F)
        a = ref._starts["w_i"]; scale[a:a + ref._sizes["w_i"]] = sb.numpy()
        total = r[0]["delta%d" % x].astype(np.float64) + r[1]["delta%d" % x]
        cur = cur + scale * total
        np.testing.assert_allclose(r[0]["flat%d" % x][:T], cur, rtol=0, atol=2e-6)
        assert r[0]["flag%d" % x] == 0.0
        mean = (vu2[0] * 10 + vu2[1] * 11) / 21.0                                                 # what the ranks agreed on for the next exchange
        cur = r[0]["flat%d" % x][:T].astype(np.float64)


def test_shared_tables_bucket_layout():
    _, _, _, w = _problem()
    s = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    for k in SHARED_NAMES:
        assert tuple(s.views[k].shape) == w[k].shape and s.views[k].is_contiguous()
        assert (s.views[k].data_ptr() - s.flat.data_ptr()) % 256 == 0          # every table 256-byte aligned in the bucket
        assert np.array_equal(s.views[k].numpy(), w[k])
    s.views["w_i"][3] = 5.0
    assert s.flat[(s.views["w_i"].data_ptr() - s.flat.data_ptr()) // 4 + 3] == 5.0
    s.all_reduce_deltas()        # no process group: a no-op
    assert s.payload_bytes == s.flat.numel() * 4


def _fit_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pandas as pd
    from oracle import oracle as orc
    from rankfm_amd import RankFM
    from rankfm_amd.distributed import fit_distributed
    pairs, _, _, _ = _problem()
    ids = pd.DataFrame({"user_id": ["u%04d" % u for u in pairs[:, 0]], "item_id": pairs[:, 1] * 3 + 7})     # raw ids, not indexes

    def make_trainer(shard, tables, x_if, hyper, device, group):
        from rankfm_amd.distributed import agree_on_merge_damping
        shared = SharedTables(tables, torch.device("cpu"))
        agree_on_merge_damping(shared, shard, group, learning_rate=hyper["learning_rate"])      # like make_device_trainer
        x_uf = np.zeros((len(shard["v_u"]), 1), np.float32)

        def epoch_fn(views, epoch):
            t = {k: views[k].numpy() for k in SHARED_NAMES}
            return orc.fit(shard["interactions"], shard["sample_weight"], shard["csr_offsets"], shard["csr_items"], x_uf, x_if, t["w_i"],
                           t["w_if"], shard["v_u"], t["v_i"], t["v_uf"], t["v_if"], hyper["alpha"], hyper["beta"], hyper["learning_rate"],
                           hyper["learning_schedule"], hyper["learning_exponent"], hyper["max_samples"], 1, perms=None,
                           rng_mode=orc.RNG_COUNTER, seed=5 + dist.get_rank(), epoch_begin=epoch, membership="binary")
        return (ShardedTrainer(shared, epoch_fn, user_norms_fn=lambda: (float((shard["v_u"].astype(np.float64) ** 2).sum()), len(shard["v_u"]))),
                (lambda: shard["v_u"]))

    m = RankFM(factors=F)
    np.random.seed(3)
    fit_distributed(m, ids, epochs=2, make_trainer=make_trainer)
    np.savez(os.path.join(out_dir, "fit%d.npz" % rank), v_u=m.v_u, v_i=m.v_i, w_i=m.w_i, users=m.user_id.values.astype("U8"),
             csr_off=m.user_items.offsets, csr_items=m.user_items.items, interactions=m.interactions)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_fit_distributed_returns_the_full_model_on_every_rank(tmp_path, world):
    """the user-facing multi-GPU fit on 2 and on EIGHT gloo ranks (oracle as the epoch, rank-local identifier mapping, damped
    merge agreed over the ranks): identical complete models on every rank"""
    mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "fit0.npz"), np.load(tmp_path / ("fit%d.npz" % (world - 1)))
    for r in range(1, world - 1):
        c = np.load(tmp_path / ("fit%d.npz" % r))
        assert all(np.array_equal(a[k], c[k]) for k in ("v_u", "v_i", "w_i", "csr_items", "interactions"))
    for k in ("v_u", "v_i", "w_i"):
        assert np.array_equal(a[k], b[k]) and np.isfinite(a[k]).all(), k
    assert a["v_u"].shape == (U, F) and a["v_i"].shape == (I, F)
    w0 = synthetic.init_weights(U, I, F, seed=5)
    assert not np.allclose(a["v_u"][:5], 0) and list(a["users"][:2]) == ["u0000", "u0001"]
    # every user row was trained by exactly one rank: compare against the untouched init drawn with the same numpy seed
    np.random.seed(3)
    init_v_u = np.random.normal(0, 0.1, (U, F)).astype(np.float32)
    moved = np.abs(a["v_u"] - init_v_u).max(axis=1)
    assert (moved > 0).all()
    # the per-user item lists were built rank-locally and exchanged: every rank ends with the complete lists the single-process
    # front end builds
    from rankfm_amd import UserItemsCSR
    pairs, _, _, _ = _problem()
    want = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], U)
    for r in (a, b):
        assert np.array_equal(r["csr_off"], want.offsets) and np.array_equal(r["csr_items"], want.items)
        assert np.array_equal(r["interactions"], pairs)


def _failing_worker(rank, world, port, out_dir):
    """rank 1's local epoch raises; rank 0's is fine: both must come back with an exception instead of hanging in the all-reduce"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, _, w = _problem()
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))

    def fn(views, epoch):
        if rank == 1:
            raise AssertionError("item factors [v_i] are not finite")
        return dict(ll=np.zeros(1))
    trainer = ShardedTrainer(shared, fn)
    try:
        trainer.run_epoch(0)
        outcome = "no error"
    except AssertionError as e:
        outcome = "own: %s" % e
    except RuntimeError as e:
        outcome = "peer: %s" % e
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write(outcome)
    dist.destroy_process_group()


def test_a_failing_rank_stops_every_rank_instead_of_hanging_the_job(tmp_path):
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (tmp_path / "rank0.txt").read_text(), (tmp_path / "rank1.txt").read_text()
    assert r1.startswith("own: item factors") and r0.startswith("peer: another rank"), (r0, r1)


def test_more_ranks_than_users_leaves_idle_ranks_in_the_collectives():
    """shard_boundaries may hand a rank an empty user range; such a rank builds an idle trainer (no device session) that still
    takes part in every exchange"""
    from rankfm_amd.distributed import make_device_trainer
    off = np.array([0, 5, 9], dtype=np.int64)                    # two users, four ranks
    b = shard_boundaries(off, 4)
    assert b[0] == 0 and b[-1] == 2 and np.any(np.diff(b) == 0)
    _, _, _, w = _problem()
    empty = dict(interactions=np.zeros((0, 2), np.int32), sample_weight=np.zeros(0, np.float32), csr_offsets=np.zeros(1, np.int64),
                 csr_items=np.zeros(0, np.int32), x_uf=np.zeros((0, 1), np.float32), v_u=np.zeros((0, F), np.float32))
    trainer, sess = make_device_trainer(empty, {k: w[k] for k in SHARED_NAMES}, np.zeros((I, 1), np.float32), {}, torch.device("cpu"))
    assert sess is None
    before = trainer.shared.flat.clone()
    out = trainer.run_epoch(0)                                   # no process group: the exchange is a no-op, the epoch is empty
    assert out["n_draws"][0] == 0 and torch.equal(trainer.shared.flat, before)


def test_config_shards_are_the_interaction_balanced_user_split(monkeypatch):
    """bench.py --config C4 / C5 trains, on rank r of W, synthetic.make_config_shard(name, r, W): the blocks
    [64 r / W, 64 (r + 1) / W) of ONE data set.  Every block holds the same number of interactions, so that is exactly the split
    distributed.shard_boundaries makes of the whole data set -- checked at 8 ranks on a scaled-down config 4."""
    small = dict(n_users=6400, n_items=900, n_interactions=128_000, factors=8, loss="bpr", max_samples=1,
                 n_user_features=4, n_item_features=4, learning_rate=0.03)
    monkeypatch.setitem(synthetic.CONFIGS, "C4", small)
    whole = synthetic.make_config_shard("C4", 0, 1)
    assert len(whole["interactions"]) == small["n_interactions"] and whole["user_hi"] == small["n_users"]
    bounds = shard_boundaries(whole["csr_offsets"], 8)
    assert np.array_equal(bounds, np.arange(9) * (small["n_users"] // 8))
    for r in (0, 3, 7):
        sh = synthetic.make_config_shard("C4", r, 8)
        assert (sh["user_lo"], sh["user_hi"]) == (bounds[r], bounds[r + 1])
        part = take_user_shard(whole["interactions"], whole["sample_weight"], whole["csr_offsets"], whole["csr_items"], whole["x_uf"],
                               whole["weights"]["v_u"], bounds[r], bounds[r + 1])
        assert np.array_equal(np.sort(part["interactions"].view("i8").ravel()), np.sort(sh["interactions"].view("i8").ravel()))
        assert np.array_equal(part["csr_items"], sh["csr_items"]) and np.array_equal(part["x_uf"], sh["x_uf"])
        assert np.array_equal(part["v_u"], sh["weights"]["v_u"]) and np.array_equal(sh["x_if"], whole["x_if"])
        assert all(np.array_equal(sh["weights"][k], whole["weights"][k]) for k in SHARED_NAMES)


def _merge_emulation(oracle, world, epochs, lr, U=3000, I=2000, F=16, rule="clamp"):
    """`world` user shards trained by the oracle from the same epoch-start tables and merged with SharedTables' scale after
    every epoch -- what ShardedTrainer does across ranks, in one process -- next to sequential training of the whole data."""
    from rankfm_amd._rankfm import UserItemsCSR
    d = synthetic.make_planted(seed=1, n_users=U, n_items=I, mean_degree=100.0)
    pairs, test = d["train"], d["test"]
    n = len(pairs)
    csr = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], U)
    tcsr = UserItemsCSR.from_pairs(test[:, 0], test[:, 1], U)
    w = synthetic.init_weights(U, I, F, seed=3)
    sw, z_i = np.ones(n, np.float32), np.zeros((I, 1), np.float32)
    test_users = np.unique(test[:, 0])

    def hit_rate(v_u, v_i, w_i, k=10):
        s = v_u[test_users] @ v_i.T + w_i
        for r, u in enumerate(test_users):
            s[r, csr.items[csr.offsets[u]:csr.offsets[u + 1]]] = -np.inf
        top = np.argpartition(-s, k, axis=1)[:, :k]
        return float(np.mean([np.intersect1d(top[r], tcsr.items[tcsr.offsets[u]:tcsr.offsets[u + 1]]).size > 0
                              for r, u in enumerate(test_users)]))

    def fit(p, s_w, off, items, t, v_u, e, count, seed):
        return oracle.fit(p, s_w, off, items, np.zeros((len(v_u), 1), np.float32), z_i, t["w_i"], t["w_if"], v_u, t["v_i"], t["v_uf"],
                          t["v_if"], 0.01, 0.1, lr, "constant", 0.25, 1, count, perms=None, rng_mode=oracle.RNG_COUNTER, seed=seed,
                          epoch_begin=e, membership="binary")
    o = {k: v.copy() for k, v in w.items()}
    ll_seq = fit(pairs, sw, csr.offsets, csr.items, o, o["v_u"], 0, epochs, 1)["ll"]
    bounds = shard_boundaries(csr.offsets, world)
    shards = [take_user_shard(pairs, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), w["v_u"].copy(), bounds[r], bounds[r + 1])
              for r in range(world)]
    shared = SharedTables({k: w[k].copy() for k in SHARED_NAMES}, torch.device("cpu"))
    shared.set_merge_damping(np.bincount(pairs[:, 1], minlength=I), world, learning_rate=lr)
    n_rank = [np.bincount(s["interactions"][:, 1], minlength=I) for s in shards]
    ll = np.zeros(epochs)
    for e in range(epochs):
        if rule == "curvature":       # what ShardedTrainer._exchange does before every exchange, the ranks' terms summed by hand
            from rankfm_amd.distributed import curvature_log_rho, curvature_scales, curvature_terms
            mean_vu2 = float(np.mean(np.concatenate([np.sum(s["v_u"].astype(np.float64) ** 2, axis=1) for s in shards])))
            log_rho = curvature_log_rho(lr, SharedTables.CURVATURE_FACTORS, SharedTables.CURVATURE_BIASES, mean_vu2)
            sv, sb = curvature_scales(sum(curvature_terms(n, log_rho) for n in n_rank), log_rho, world)
            a = shared._starts["v_i"]; shared.merge_scale[a:a + shared._sizes["v_i"]] = sv.to(torch.float32).repeat_interleave(F)
            a = shared._starts["w_i"]; shared.merge_scale[a:a + shared._sizes["w_i"]] = sb.to(torch.float32)
        start, total = shared.flat.clone(), torch.zeros_like(shared.flat)
        for k, s in enumerate(shards):
            shared.flat.copy_(start)
            t = {name: shared.views[name].numpy() for name in SHARED_NAMES}
            ll[e] += fit(s["interactions"], s["sample_weight"], s["csr_offsets"], s["csr_items"], t, s["v_u"], e, 1, 100 + k)["ll"][0]
            total += shared.flat - start
        shared.flat.copy_(start + shared.merge_scale * total)
    v_u = np.concatenate([s["v_u"] for s in shards])
    v_i, w_i = shared.views["v_i"].numpy(), shared.views["w_i"].numpy()
    return dict(hit=hit_rate(v_u, v_i, w_i), hit_seq=hit_rate(o["v_u"], o["v_i"], o["w_i"]), ll=ll, ll_seq=ll_seq,
                w_i=float(np.linalg.norm(w_i) / np.linalg.norm(o["w_i"])), v_i=float(np.linalg.norm(v_i) / np.linalg.norm(o["v_i"])))


@pytest.mark.parametrize("lr, epochs", [(0.1, 10), (0.03, 20)])
def test_eight_user_shards_with_the_damped_merge_track_sequential_training(oracle, lr, epochs):
    """BASELINE configs 4 / 5 run on EIGHT ranks with one exchange per epoch.  Eight shards of a planted ranking problem (28 k rows
    per rank and epoch: far harsher than a BASELINE-sized epoch) must stay stable and rank like sequential training, at the
    reference's learning rate and at config 4's.  Measured: hit_rate@10 0.885 vs 0.892 (lr 0.1) and 0.896 vs 0.900 (lr 0.03),
    last-epoch log-likelihood +5.4 % / +2.1 % (the merged run converges a little later), |w_i| +11 %.  What the defaults of
    SharedTables.set_merge_damping guard against: ONE damping constant of 48 for factors and biases lets the biases diverge
    (0.51), and the lr-0.1 constant used at lr 0.03 learns three times too slowly (0.79 after 20 epochs)."""
    r = _merge_emulation(oracle, 8, epochs, lr)
    assert abs(r["hit"] - r["hit_seq"]) <= 0.015, r
    assert r["ll"][-1] > r["ll"][0] and abs(r["ll"][-1] / r["ll_seq"][-1] - 1.0) <= 0.10, r
    assert 0.8 <= r["w_i"] <= 1.2 and 0.8 <= r["v_i"] <= 1.1, r


@pytest.mark.parametrize("lr, epochs", [(0.1, 10), (0.03, 20)])
def test_eight_user_shards_with_the_curvature_merge_track_sequential_training(oracle, lr, epochs):
    """the same emulation with the merge rule fit_distributed uses by default since the end of round 3 (SharedTables.set_merge_curvature:
    the scale of the summed deltas follows the model's mean |v_u|^2, so the ranks' deltas are summed early and averaged late)."""
    r = _merge_emulation(oracle, 8, epochs, lr, rule="curvature")
    print("curvature merge, lr %g: hit_rate@10 %.4f (sequential %.4f)  last-epoch LL / sequential - 1 %+.3f  |w_i| %.3f |v_i| %.3f"
          % (lr, r["hit"], r["hit_seq"], r["ll"][-1] / r["ll_seq"][-1] - 1.0, r["w_i"], r["v_i"]))
    assert abs(r["hit"] - r["hit_seq"]) <= 0.015, r
    assert r["ll"][-1] > r["ll"][0] and abs(r["ll"][-1] / r["ll_seq"][-1] - 1.0) <= 0.10, r


def test_curvature_rule_limits():
    """few steps: the ranks' deltas add up (scale 1); many steps on every rank: they all say the same thing (scale 1 / ranks); an item
    only ONE rank steps keeps scale 1 however often; an item nobody steps: 1"""
    from rankfm_amd.distributed import curvature_log_rho, curvature_scales, curvature_terms
    log_rho = curvature_log_rho(0.1, 0.1, 0.3, 2.0)
    n = [np.array([1.0, 5000.0, 5000.0, 0.0]), np.array([1.0, 5000.0, 0.0, 0.0]), np.array([0.0, 5000.0, 0.0, 0.0]), np.array([1.0, 5000.0, 0.0, 0.0])]
    sv, sb = curvature_scales(sum(curvature_terms(x, log_rho) for x in n), log_rho, 4)
    np.testing.assert_allclose(sv.numpy(), [1.0, 0.25, 1.0, 1.0], atol=0.03)
    np.testing.assert_allclose(sb.numpy(), [1.0, 0.25, 1.0, 1.0], atol=0.05)
    mid = curvature_scales(sum(curvature_terms(np.array([40.0]), log_rho) for _ in range(4)), log_rho, 4)[0].item()
    assert 0.3 < mid < 0.9            # 40 steps per rank at kappa = 0.02: between the sum and the average


# ---- the one-window-late merge (round 5): SharedTables.exchange_late / ShardedTrainer(overlap=True) ----------------------------------
LATE_WINDOWS, LATE_EPOCHS = 3, 2


def _late_counts(rank, with_counts):
    """item update counts of a rank per epoch: zero (every item-side scale is exactly 1: the whole merge is dyadic arithmetic and can be
    compared bit for bit) or a histogram (the curvature rule's scales: compared to 1e-6)"""
    if not with_counts:
        return np.zeros(I, np.float64)
    rng = np.random.default_rng(50 + rank)
    return rng.integers(0, 300, I).astype(np.float64) * (rng.random(I) < 0.8)


def _late_delta(flat_tables, rank, step):
    """what a rank's local window adds to ITS tables: a term that depends on the tables it trains on (so that training on one's own
    un-merged result matters) and a rank / step pattern; every value a small multiple of 2^-12 -- sums and quarter-products stay exact"""
    T = flat_tables.numel()
    pattern = torch.as_tensor(((np.arange(T) * (rank + 3) + step * 7) % 17 - 8).astype(np.float32)) * 2.0 ** -9
    return flat_tables * 0.25 + pattern


def _late_initial(T):
    return torch.as_tensor(((np.arange(T) * 5) % 13 - 6).astype(np.float32)) * 2.0 ** -6


def _late_worker(rank, world, port, out_dir, with_counts):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, _, w = _problem()
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    T = shared._tail_at
    shared.flat[:T] = _late_initial(T)
    shared.set_merge_curvature(_late_counts(rank, with_counts), world, learning_rate=0.1, mean_vu2=0.4 + 0.1 * rank, n_users=7 + rank)
    step = [0]

    def fn(views, epoch, part=None):
        shared.flat[:T] += _late_delta(shared.flat[:T].clone(), rank, step[0])
        step[0] += 1
        return dict(ll=np.zeros(1))
    trainer = ShardedTrainer(shared, fn, syncs_per_epoch=LATE_WINDOWS, overlap=True,
                             user_norms_fn=lambda: (float(2.0 + rank + step[0]), 7 + rank))
    snaps = []
    for e in range(LATE_EPOCHS):
        trainer.run_epoch(e)
        snaps.append(shared.flat[:T].numpy().copy())          # (between exchanges the replicas differ, by design)
    trainer.finish()
    np.savez(os.path.join(out_dir, "late%d.npz" % rank), final=shared.flat[:T].numpy(), snaps=np.stack(snaps))
    dist.destroy_process_group()


def _late_by_hand(world, with_counts):
    """the blocking curvature rule with every merged delta applied ONE WINDOW LATE, rank by rank in one process"""
    from rankfm_amd.distributed import curvature_log_rho
    _, _, _, w = _problem()
    ref = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    T, n_items = ref._tail_at, ref._shapes["w_i"][0]
    counts = [torch.as_tensor(_late_counts(r, with_counts)) for r in range(world)]
    n_total = sum(counts)
    tables = [_late_initial(T).clone() for _ in range(world)]
    mean_vu2 = sum((0.4 + 0.1 * r) * (7 + r) for r in range(world)) / sum(7 + r for r in range(world))
    pending = None            # (scale [T], SUM [T], own deltas per rank, sum |v_u|^2, users) of the window whose reduction is in flight
    snaps, step = [[] for _ in range(world)], 0

    def scale_of(mean):
        log_rho = curvature_log_rho(0.1, SharedTables.CURVATURE_FACTORS, SharedTables.CURVATURE_BIASES, mean)
        window = 1.0 / LATE_WINDOWS
        scale = torch.full((T,), 1.0 / world, dtype=torch.float32)        # (the alignment padding between the tables)
        for name, lr_ in (("v_i", log_rho[0]), ("w_i", log_rho[1])):
            # (each rank's term is narrowed to float32 before the sum, like the bucket's tail)
            t = sum((-torch.expm1(lr_ * (c * window))).to(torch.float32).to(torch.float64) for c in counts)
            s = torch.where(t > 0, -torch.expm1(lr_ * (n_total * window)) / torch.clamp(t, min=1e-30), torch.ones_like(t)).clamp(1.0 / world, 1.0)
            a = ref._starts[name]
            per = ref._shapes["v_i"][1] if name == "v_i" else 1
            scale[a:a + ref._sizes[name]] = s.to(torch.float32).repeat_interleave(per)
        return scale

    def apply(p):
        nonlocal mean_vu2
        scale, total, own, s_vu2, users = p
        for r in range(world):
            tables[r] = tables[r] + scale * total - own[r]
        mean_vu2 = s_vu2 / users

    for e in range(LATE_EPOCHS):
        for k in range(LATE_WINDOWS):
            own = []
            for r in range(world):
                d = _late_delta(tables[r], r, step)
                tables[r] = tables[r] + d
                d = d.clone()
                own.append(d)
            scale = scale_of(mean_vu2)                       # rho from the mean agreed at the last COMPLETED reduction
            if pending is not None:
                apply(pending)
            total = own[0].clone()
            for r in range(1, world):
                total = total + own[r]
            pending = (scale, total, own, sum(2.0 + r + step + 1 for r in range(world)), sum(7 + r for r in range(world)))
            step += 1
        for r in range(world):
            snaps[r].append(tables[r].numpy().copy())
    apply(pending)
    for r in range(1, world):                            # the closing broadcast of rank 0's tables
        tables[r] = tables[0].clone()
    return tables, snaps


@pytest.mark.parametrize("world, with_counts", [(2, False), (8, False), (2, True)])
def test_late_merge_is_the_blocking_rule_applied_one_window_late(tmp_path, world, with_counts):
    """ShardedTrainer(overlap=True): the all-reduce of window k runs beside window k + 1 and is applied one window late
    (SharedTables.exchange_late); after finish() the replicas are identical.  Against the same rule computed by hand, rank by rank:
    bit for bit when every scale is a power of two (no item counts), to 2e-6 with the curvature rule's scales."""
    mp.spawn(_late_worker, args=(world, _free_port(), str(tmp_path), with_counts), nprocs=world, join=True)
    r = [np.load(tmp_path / ("late%d.npz" % k)) for k in range(world)]
    tables, snaps = _late_by_hand(world, with_counts)
    for k in range(1, world):
        assert np.array_equal(r[0]["final"], r[k]["final"])                       # identical replicas after the final exchange
    assert not np.array_equal(r[0]["snaps"][0], r[1]["snaps"][0])                 # ... and different ones in between, by design
    for k in range(world):
        for e in range(LATE_EPOCHS):
            if with_counts:
                np.testing.assert_allclose(r[k]["snaps"][e], snaps[k][e], rtol=0, atol=2e-6)
            else:
                assert np.array_equal(r[k]["snaps"][e], snaps[k][e]), (k, e)
        if with_counts:                          # (the replicas agree to rounding before the closing broadcast of rank 0's tables)
            np.testing.assert_allclose(r[k]["final"], tables[0].numpy(), rtol=0, atol=2e-6)
        else:
            assert np.array_equal(r[k]["final"], tables[k].numpy())


def _late_failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, _, w = _problem()
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    shared.set_merge_curvature(np.ones(I), world, learning_rate=0.1, mean_vu2=0.5, n_users=3)
    calls = [0]

    def fn(views, epoch, part=None):
        calls[0] += 1
        if rank == 1 and calls[0] == 2:
            raise AssertionError("item factors [v_i] are not finite")
        return dict(ll=np.zeros(1))
    trainer = ShardedTrainer(shared, fn, syncs_per_epoch=4, overlap=True, user_norms_fn=lambda: (1.0, 3))
    try:
        trainer.run_epoch(0)
        trainer.finish()
        outcome = "no error"
    except AssertionError as e:
        outcome = "own: %s" % e
    except RuntimeError as e:
        outcome = "peer: %s (after %d windows)" % (e, calls[0])
    with open(os.path.join(out_dir, "late_rank%d.txt" % rank), "w") as f:
        f.write(outcome)
    dist.destroy_process_group()


def test_late_merge_a_failing_rank_stops_its_peers_one_window_later(tmp_path):
    """the failure flag travels with the deltas: the peers see it when they wait for that window's reduction -- one window later --
    and stop before they launch another collective the failed rank would never join"""
    mp.spawn(_late_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (tmp_path / "late_rank0.txt").read_text(), (tmp_path / "late_rank1.txt").read_text()
    assert r1.startswith("own: item factors") and r0.startswith("peer: another rank") and "after 3 windows" in r0, (r0, r1)


def test_auto_overlap_takes_the_late_merge_only_where_it_is_faster():
    """ShardedTrainer(overlap="auto") decides after its first (blocking) epoch: blocking costs T + n x per epoch, the late merge
    max(T, 3 n x) -- config 5's share (246 ms of SGD, ~3 ms per exchange) overlaps, config 4's (3.9 ms, ~0.3 ms) does not"""
    _, _, _, w = _problem()
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    t = ShardedTrainer(shared, lambda views, epoch, part=None: dict(ll=np.zeros(1)), overlap="auto")
    t._decide_overlap(246.0, 3.0, 8)
    assert t._late_on and t.overlap_decision["late"]
    t._decide_overlap(3.9, 0.3, 8)
    assert not t._late_on
    t._decide_overlap(3.9, 0.3, 1)              # (one exchange per epoch: 0.9 ms of reductions fit behind 3.9 ms of SGD)
    assert t._late_on
    assert not t.late                           # (no curvature rule armed on one process: nothing to overlap)


def test_auto_overlap_refuses_the_late_merge_where_a_window_moves_an_item_most_of_the_way():
    """the one-window-late merge is a loop with a delay: with many updates per item per window its gain is near one and it rings
    (measured at configs 5 and 8 x 2: ShardedTrainer.LATE_MOVEMENT) -- `auto` stays blocking there even when late would be faster"""
    _, _, _, w = _problem()
    I = w["w_i"].shape[0]
    for per_epoch, want_late in ((90.0, True), (500.0, False), (720.0, False)):
        shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
        shared.set_merge_curvature(np.full(I, per_epoch), 1, learning_rate=0.1, mean_vu2=1.0, n_users=10)
        t = ShardedTrainer(shared, lambda views, epoch, part=None: dict(ll=np.zeros(1)), overlap="auto")
        t._decide_overlap(246.0, 3.0, 8)        # (faster by the clock in all three)
        d = t.overlap_decision
        assert d["faster"] and d["late"] == want_late and (d["window_movement"] <= t.LATE_MOVEMENT) == want_late, d
    assert abs(shared.window_movement(1.0 / 24) - (1.0 - 0.97 ** 30)) < 1e-9      # (720 / 24 = 30 updates at rho_w = 1 - 0.1 * 0.3)
    # ... and never for a model whose feature tables are trained: every row touches them (config 4 at its own size diverged with it)
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    shared.set_merge_curvature(np.full(I, 10.0), 1, learning_rate=0.1, mean_vu2=1.0, n_users=10)
    shared.has_feature_tables = True
    t = ShardedTrainer(shared, lambda views, epoch, part=None: dict(ll=np.zeros(1)), overlap="auto")
    t._decide_overlap(246.0, 3.0, 8)
    assert t.overlap_decision["faster"] and not t.overlap_decision["late"] and t.overlap_decision["window_movement"] == 1.0


def _bf16_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, _, w = _problem()
    rng = np.random.default_rng(20 + rank)
    shared = SharedTables({k: w[k] for k in SHARED_NAMES}, torch.device("cpu"))
    shared.exchange_dtype = "bf16"
    counts = rng.integers(0, 400, I).astype(np.float64)
    shared.set_merge_curvature(counts, world, learning_rate=0.1, mean_vu2=0.5, n_users=10)
    shared.begin_epoch()
    delta = torch.as_tensor(rng.normal(0, 0.01, shared._tail_at).astype(np.float32))
    shared.flat[:shared._tail_at] += delta
    flag = shared.exchange_fused(None, torch.tensor(5.0, dtype=torch.float64), 10, failed=rank == 1)
    np.savez(os.path.join(out_dir, "bf16_%d.npz" % rank), flat=shared.flat.numpy().copy(), delta=delta.numpy(), start=shared.start.numpy().copy(),
             scale=shared.merge_scale.numpy().copy(), flag=float(flag), payload=shared.payload_bytes)
    dist.destroy_process_group()


def test_bf16_exchange_rounds_the_deltas_only_and_keeps_the_tail_exact(tmp_path):
    """SharedTables.exchange_dtype = "bf16" (an option): the tables' deltas travel rounded to bfloat16 and are summed in bfloat16, the tail --
    curvature terms, |v_u|^2 sums, the failure flag -- in a second fp32 all-reduce, exact; `start + scale x sum` is fp32.  Two gloo ranks,
    rank 1 reports a failed slice (zero deltas, flag raised): the merged tables are start + scale x bf16(rank 0's delta), bit for bit, on both."""
    world = 2
    mp.spawn(_bf16_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("bf16_%d.npz" % k)) for k in range(world)]
    ref = SharedTables({k: _problem()[3][k] for k in SHARED_NAMES}, torch.device("cpu"))
    T = ref._tail_at
    assert np.array_equal(r[0]["flat"], r[1]["flat"]) and r[0]["flag"] == 1.0 and r[1]["flag"] == 1.0
    st = torch.as_tensor(r[0]["start"][:T])
    own = (st + torch.as_tensor(r[0]["delta"])) - st                                             # (the delta as the bucket holds it: flat - start)
    d16 = own.to(torch.bfloat16).to(torch.float32).numpy()                                       # (rank 1's deltas are zero: its slice failed)
    want = (torch.as_tensor(r[0]["start"][:T]) + torch.as_tensor(r[0]["scale"][:T]) * torch.as_tensor(d16)).numpy()
    np.testing.assert_allclose(r[0]["flat"][:T], want, rtol=0, atol=1e-7)
    assert float(np.abs(d16 - r[0]["delta"]).max()) > 0.0                                        # (the rounding is really there)
    assert int(r[0]["payload"]) == T * 2 + ref._tail_len * 4
