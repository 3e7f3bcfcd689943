#!/usr/bin/env python
"""bench.py -- (user, item, neg) pairwise updates/s of the RankFM SGD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE EPOCH of the hot path (BPR SGD, k=64) over the rank's resident interaction shard: the SGD
wavefront kernel over all rows + the epoch tail (finiteness check) + -- for N > 1 -- the RCCL all-reduce of the
item-side deltas.  Workload = BASELINE.json config 2 (synthetic 100k users x 50k items x 5M interactions,
factors=64, loss='bpr') per GPU; for N > 1 every rank owns its own 100k-user / 5M-interaction shard over the
shared 50k items (weak scaling; the 8-GPU total is config-4-shaped: 800k users, 40M interactions).  Inputs are
resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0) with the driver's contract fields plus:
  roofline     achieved = algorithmic bytes (24F+32 = 1568 B per update) x rows per launch / average
               HIP-event duration of the SGD kernel launch, against the 8 TB/s HBM3E peak; `peak_measured` = what a plain
               streaming kernel gets from this box's HBM (rfm_hbm_probe).  The timed kernel draws its negatives like the
               reference (uniformly over the whole catalogue).
  also         (N = 1, config 2 only) the other BASELINE configurations that fit one GPU, 5 timed epochs each after the config-2 timed
               region: config 3 (WARP on the same data), config 4 and config 5 as ONE GPU's share of 8 -- the real kernel name,
               kernel ms min / median, mean draws per update, algorithmic bytes, fraction of the HBM roofline (`--also ""` skips them)
  cpu_baseline the CPU restatement of the reference's `_fit` (oracle/, "port"; MT19937 + linear membership scan like
               the reference) timed on ONE host core (the reference is single-threaded) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable


def algorithmic_bytes_per_update(F, n_uf=0, n_if=0, draws=1.0):
    """reads (u,i) 8 + sw 4 + perm 4 + v_u 4F + v_i[i] 4F + w_i 4 + per draw (v_i[j] 4F + w_i[j] 4);
    writes v_u, v_i[i], v_i[j] 3*4F + 8  ->  BPR 24F + 32; each extra WARP draw adds 4F + 4; dense features add
    4P + 4Q(1 + draws)"""
    b = 24 * F + 32 + (draws - 1.0) * (4 * F + 4)
    if n_uf or n_if:
        b += 4 * n_uf + 4 * n_if * (1 + draws)
    return b


def cpu_baseline(shard, x_if, w, hyper, has_uf, has_if, seconds_budget=25.0):
    """time the CPU restatement (oracle = checker infrastructure, used here only as the reported baseline) on the same
    workload: same interactions, features, loss and max_samples, from the same initial weights"""
    from oracle import oracle as orc
    pairs, off, items, x_uf = shard["interactions"], shard["csr_offsets"], shard["csr_items"], shard["x_uf"]
    N = len(pairs)
    sw = np.ones(N, dtype=np.float32)
    rng = np.random.default_rng(0)

    def run(n_rows):
        perm = rng.permutation(N)[:n_rows].astype(np.int32)
        sub = np.ascontiguousarray(pairs[perm])                   # a random sample of the same workload's rows
        p = np.arange(n_rows, dtype=np.int32)[None, :]
        ww = {k: np.array(v, dtype=np.float32, copy=True) for k, v in w.items()}
        t0 = time.perf_counter()
        orc.fit(sub, sw[:n_rows], off, items, x_uf, x_if, ww["w_i"], ww["w_if"], ww["v_u"], ww["v_i"],
                ww["v_uf"], ww["v_if"], 0.01, 0.1, hyper["learning_rate"], "constant", 0.25, hyper["max_samples"], 1, perms=p,
                rng_mode=orc.RNG_MT19937, seed=1492, membership="linear", has_uf=has_uf, has_if=has_if)
        return time.perf_counter() - t0

    probe = min(N, 300_000)
    t = run(probe)
    rate = probe / t
    n_rows = int(min(N, max(probe, rate * seconds_budget * 0.6)))
    t = run(n_rows)
    return dict(value=n_rows / t, unit="updates/s", cores=1, kind="port",
                sample="%d randomly sampled rows of the same workload (same features / loss / max_samples, initial weights), 1 epoch, "
                       "oracle/rfm_oracle.c (gcc -O2 -ffast-math, MT19937 + linear membership scan like the reference; 1.19x the time of "
                       "the reference's Cython _fit at k=64 BPR per tools/calibrate_cpu.py), %.1f s on 1 core of %d"
                       % (n_rows, t, os.cpu_count() or 0))


def kernel_name(F, max_samples, n_uf, n_if):
    """the dominant kernel of a chip-filling Hogwild launch, as rocprofv3 --kernel-trace prints it (rfm_sgd_inst.inc picks the
    instantiation: 16-lane row groups, ceil(F / 16) dwords per lane)"""
    kpl = (F + 15) // 16
    if n_uf or n_if:
        return "rfm::sgd_features_fast_kernel<16, %d, false, 768> (+ rfm::feat_tables_kernel<16, %d, false> beside it on the engine's second stream)" % (kpl, kpl)
    if max_samples > 1:
        return "rfm::sgd_warp_kernel<16, %d, false, true, %s>" % (kpl, "true" if F == 16 * kpl else "false")
    # (BPR with hot-row accumulators; full factor rows run on segment-major item rows: the last template argument)
    return "rfm::sgd_segments_kernel<16, %d, false, true, false, %s>" % (kpl, "true" if F == 16 * kpl else "false")


def also_workload(name, device, c2_shard, c2_x_if, c2_weights, steps=5, warmup=2):
    """one more BASELINE configuration on this GPU: `steps` timed epochs of the resident shard, HIP-event kernel time per launch.
    C3 = config 2's data with WARP (max_samples 50); C4 / C5 = user shard 0 of 8 of the config's ONE data set."""
    import torch
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    cfg = synthetic.CONFIGS[name]
    F = cfg["factors"]
    n_uf, n_if = cfg.get("n_user_features", 0), cfg.get("n_item_features", 0)
    t0 = time.perf_counter()
    if name in ("C4", "C5"):
        sh = synthetic.make_config_shard(name, rank=0, world=8)
        data = (sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"])
        weights = sh["weights"]
        what = "%s: user shard 0 of 8 of ONE synthetic data set of %d users x %d items x %d interactions (this GPU: %d users, %d interactions, all items)" % (
            name, cfg["n_users"], cfg["n_items"], cfg["n_interactions"], sh["user_hi"] - sh["user_lo"], len(sh["interactions"]))
    else:
        data = (c2_shard["interactions"], c2_shard["sample_weight"], c2_shard["csr_offsets"], c2_shard["csr_items"], c2_shard["x_uf"], c2_x_if)
        weights = c2_weights
        what = "%s: config 2's synthetic %d users x %d items x %d interactions" % (name, cfg["n_users"], cfg["n_items"], cfg["n_interactions"])
    gen_s = time.perf_counter() - t0
    N = len(data[0])
    sess = DeviceSession(*data, {k: np.array(v, copy=True) for k, v in weights.items()}, seed=1492, device=device,
                         has_user_features=int(n_uf > 0), has_item_features=int(n_if > 0), keep_layout=True,
                         learning_rate=cfg.get("learning_rate", 0.1), max_samples=cfg["max_samples"])
    sess.run(epochs=warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = sess.run(epochs=steps, epoch_begin=warmup)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    launches = rep["launches_per_epoch"]
    k = np.array(rep["sgd_kernel_ms"], dtype=np.float64) / launches
    draws = float(np.sum(rep["n_draws"])) / (float(N) * steps)
    bpu = algorithmic_bytes_per_update(F, n_uf, n_if, draws)
    assert np.isfinite(float(rep["log_likelihood"][-1])), "%s diverged" % name
    out = {"workload": "%s, factors=%d, loss=%s%s%s, learning_rate=%g, the reference's sampler" % (
               what, F, cfg["loss"], " max_samples=%d" % cfg["max_samples"] if cfg["max_samples"] > 1 else "",
               ", %d + %d dense user / item features" % (n_uf, n_if) if n_uf or n_if else "", cfg.get("learning_rate", 0.1)),
           "kernel": kernel_name(F, cfg["max_samples"], n_uf, n_if), "rows": N, "steps": steps, "warmup": warmup,
           "kernel_ms_min": float(k.min()), "kernel_ms_median": float(np.median(k)), "sgd_launches_per_epoch": launches,
           "mean_draws": draws, "algorithmic_bytes_per_update": bpu,
           "frac": bpu * (N / launches) / (float(np.median(k)) * 1e-3) / 1e9 / HBM_PEAK_GBPS,       # (the SGD launch alone, median)
           "frac_of_step": N * steps / elapsed * bpu / 1e9 / HBM_PEAK_GBPS,                            # (from updates/s on the wall clock)
           "updates_per_s": N * steps / elapsed, "sampled_negatives_per_s": float(np.sum(rep["n_draws"])) / elapsed,
           "data_generation_s": gen_s}
    del sess
    torch.cuda.empty_cache()
    return out


def strong_scaling_record(world, rank, device, steps, warmup, barrier):
    """N > 1 only: BASELINE config 4 as ONE data set sharded by user over the ranks (strong scaling: 50 M interactions whatever N),
    timed like the main line -- barrier + synchronize on both sides, max over ranks.  A sub-record of the JSON line; the main line
    stays the weak-scaling config-2 workload the metric is quoted on."""
    import torch
    import torch.distributed as dist
    from rankfm_amd import synthetic
    from rankfm_amd.distributed import SHARED_NAMES, broadcast_from_rank0, make_device_trainer
    cfg = synthetic.CONFIGS["C4"]
    sh = synthetic.make_config_shard("C4", rank=rank, world=world)
    w = sh["weights"]
    shard = dict(interactions=sh["interactions"], sample_weight=sh["sample_weight"], csr_offsets=sh["csr_offsets"], csr_items=sh["csr_items"],
                 x_uf=sh["x_uf"], v_u=w["v_u"])
    hyper = dict(alpha=0.01, beta=0.1, learning_rate=cfg["learning_rate"], learning_schedule="constant", learning_exponent=0.25, max_samples=1)
    trainer, _ = make_device_trainer(shard, {k: w[k] for k in SHARED_NAMES}, sh["x_if"], hyper, device, seed=1492,
                                     has_user_features=1, has_item_features=1, overlap="auto")
    broadcast_from_rank0([trainer.shared.flat])
    epoch = 0
    for _ in range(warmup):
        trainer.run_epoch(epoch)
        epoch += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        rep = trainer.run_epoch(epoch)
        epoch += 1
    trainer.finish()
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert np.isfinite(float(rep["log_likelihood"][0]))
    return {"workload": "C4: ONE synthetic data set of %d users x %d items x %d interactions + %d + %d dense features, factors=%d, bpr, "
                        "sharded by user over %d GPUs, one RCCL all-reduce of the item-side deltas (%.1f MB) per epoch"
                        % (cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["n_user_features"], cfg["n_item_features"], cfg["factors"],
                           world, trainer.shared.payload_bytes / 1e6),
            "scaling": "strong", "value": float(cfg["n_interactions"]) * steps / elapsed, "unit": "updates/s", "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "n_gpus": world}


def main():
    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)      # `timeout -s USR1 ...` shows where a stuck run is waiting
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--rows-per-launch", type=int, default=0)
    ap.add_argument("--zipf", type=float, default=1.0)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--damping", type=float, default=0.0, help="hogwild damping M (0 default, <0 off)")
    ap.add_argument("--debug-flags", type=int, default=0)
    ap.add_argument("--syncs-per-epoch", default="auto", help="item-delta exchanges per epoch (N > 1): a number, or 'auto' = the production default "
                    "(8 per epoch during a fit's first 8 epochs, 1 afterwards: rankfm_amd.distributed.ShardedTrainer)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "late", "blocking"],
                    help="N > 1: 'late' = the one-window-late merge (the all-reduce runs beside the next window's SGD, three times the cadence), "
                         "'blocking' = every exchange blocks, 'auto' = decided after the first epoch: late only where it is faster AND stable at this "
                         "job's updates per item per window (ShardedTrainer.LATE_MOVEMENT; at the BASELINE configs' sizes that is 'blocking')")
    ap.add_argument("--exchange-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1, blocking exchanges: 'bf16' = the tables' deltas travel as bfloat16 (half the payload; SharedTables.exchange_dtype)")
    ap.add_argument("--factors", type=int, default=0, help="override the config's factor count (experiments)")
    ap.add_argument("--shape", type=int, default=0, help="experiment: 1-based index into the kernel shape table")
    ap.add_argument("--share", type=int, default=8, help="configs 4 / 5 on ONE GPU: run the user shard 0 of SHARE (1 = whole data set)")
    ap.add_argument("--weak", action="store_true", help="configs 4 / 5: weak scaling (every rank its own config-sized shard)")
    ap.add_argument("--learning-rate", type=float, default=0.0, help="override the config's learning rate")
    ap.add_argument("--also", default="C3,C4,C5", help="N = 1, config 2: the other configurations appended to the line as `also` (empty = none)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling sub-record (config 4 sharded over the ranks)")
    ap.add_argument("--tune", default="", help="geometry overrides (rfm_fit_tuning, experiments): 'hot_publications=24,segment_rows=32'")
    ap.add_argument("--per-epoch-calls", action="store_true", help="N = 1: one engine call per epoch through the multi-GPU trainer's path (what N > 1 runs between "
                    "exchanges) instead of ONE resident call for all timed epochs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rankfm_amd import synthetic
    from rankfm_amd.distributed import SHARED_NAMES, broadcast_from_rank0, make_device_trainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: rankfm_amd has no CPU fallback")
    # RFM_BENCH_SHARE_GPU=1 (functional testing only): all ranks share cuda:0 and talk over gloo, so the N > 1 code path
    # can be exercised on a one-GPU box; the driver's scaling runs use one GPU per rank over RCCL ("nccl").
    share_gpu = os.environ.get("RFM_BENCH_SHARE_GPU", "") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg = synthetic.CONFIGS[args.config]
    U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
    n_uf, n_if = cfg.get("n_user_features", 0), cfg.get("n_item_features", 0)
    lr = args.learning_rate if args.learning_rate > 0 else cfg.get("learning_rate", 0.1)
    # Configs 4 and 5 are ONE data set sharded by user (STRONG scaling, as BASELINE.json words them): rank r of W trains the
    # user shard r of W of the config's data (synthetic.make_config_shard; the interaction-balanced split of
    # distributed.shard_boundaries falls on its block boundaries).  On one GPU, `--share S` runs the shard 0 of S -- what one
    # of S GPUs holds (default 8, the configs' GPU count); `--share 1` the whole data set.  Configs 2 / 3 are single-GPU
    # configs: with N > 1 every rank gets its own config-sized shard over the shared catalogue (WEAK scaling), which is also
    # what `--weak` forces for 4 / 5.
    strong = args.config in ("C4", "C5") and not args.weak
    if strong:
        parts = world if world > 1 else max(args.share, 1)
        sh = synthetic.make_config_shard(args.config, rank=rank if world > 1 else 0, world=parts, zipf_s=args.zipf)
        if args.factors > 0:
            raise SystemExit("--factors needs --weak for configs 4 / 5")
        pairs, w, x_if = sh["interactions"], sh["weights"], sh["x_if"]
        shard = dict(interactions=pairs, sample_weight=sh["sample_weight"], csr_offsets=sh["csr_offsets"], csr_items=sh["csr_items"],
                     x_uf=sh["x_uf"], v_u=w["v_u"])
        from rankfm_amd._rankfm import UserItemsCSR
        csr = UserItemsCSR(sh["csr_offsets"], sh["csr_items"])
        n_local, u_local = len(pairs), sh["user_hi"] - sh["user_lo"]
        n_job = n_local * world                     # every shard holds exactly N / parts interactions
    else:
        if args.factors > 0:
            F = args.factors
        # each rank generates ITS OWN user shard (different seed) over the shared item catalogue
        pairs, csr = synthetic.make_interactions(U, I, N, seed=1000 * rank, zipf_s=args.zipf)
        w = synthetic.init_weights(U, I, F, n_uf, n_if, seed=1492 + rank)
        x_uf = synthetic.make_features(U, n_uf, 7 + rank) if n_uf else np.zeros((U, 1), np.float32)
        x_if = synthetic.make_features(I, n_if, 8) if n_if else np.zeros((I, 1), np.float32)
        shard = dict(interactions=pairs, sample_weight=np.ones(N, np.float32), csr_offsets=csr.offsets, csr_items=csr.items,
                     x_uf=x_uf, v_u=w["v_u"])
        n_local, u_local, n_job = N, U, N * world
    hyper = dict(alpha=0.01, beta=0.1, learning_rate=lr, learning_schedule="constant", learning_exponent=0.25,
                 max_samples=cfg["max_samples"])
    session_kw = dict(seed=1492, n_workgroups=args.workgroups, rows_per_launch=args.rows_per_launch,
                      has_user_features=int(n_uf > 0), has_item_features=int(n_if > 0),
                      shape_override=args.shape, hogwild_damping=args.damping, debug_flags=args.debug_flags, check_finite=not args.no_check,
                      tune={kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune.split(",") if kv})

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms, shader_mhz, draws = [], [], 0
    resident = world == 1 and not args.per_epoch_calls
    if resident:
        # ONE GPU: the resident training loop.  The session keeps the model, the interactions and -- between calls -- the engine's working
        # layout of the item-side weights in HBM (DeviceSession(keep_layout=True)); the K timed epochs are ONE call of the C ABI
        # (rfm_fit_device, epochs = K): per epoch one SGD launch and one epoch-tail launch (finiteness of all six arrays, pending hot-row
        # sums folded in), one report read back at the call's end.  Nothing of the reference's epoch is left out.
        from rankfm_amd.engine import DeviceSession
        trainer = None
        weights = {k: w[k] for k in SHARED_NAMES}
        weights["v_u"] = shard["v_u"]
        sess = DeviceSession(shard["interactions"], shard["sample_weight"], shard["csr_offsets"], shard["csr_items"], shard["x_uf"], x_if,
                             weights, device=device, keep_layout=True, **hyper, **session_kw)
        if args.warmup > 0:
            sess.run(epochs=args.warmup)
        barrier()
        t0 = time.perf_counter()
        rep = sess.run(epochs=args.steps, epoch_begin=args.warmup)
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = [float(x) for x in rep["sgd_kernel_ms"]]
        shader_mhz = [float(sess.geometry().get("shader_mhz", 0.0))]
        draws = int(np.sum(rep["n_draws"]))
        rep = dict(rep, log_likelihood=rep["log_likelihood"][-1:], n_draws=rep["n_draws"][-1:])
    else:
        trainer, sess = make_device_trainer(shard, {k: w[k] for k in SHARED_NAMES}, x_if, hyper, device,
                                            syncs_per_epoch=(args.syncs_per_epoch if args.syncs_per_epoch == "auto" else int(args.syncs_per_epoch)),
                                            overlap=False if world == 1 else {"auto": "auto", "late": True, "blocking": False}[args.exchange],
                                            exchange_dtype=args.exchange_dtype, **session_kw)
        broadcast_from_rank0([trainer.shared.flat])
        trainer.shared.record_waits = world > 1          # (time the waits for late reductions: `exposed_exchange_ms_per_epoch`)
        epoch = 0
        for _ in range(args.warmup):
            trainer.run_epoch(epoch)
            epoch += 1
        if world > 1 and hasattr(trainer.shared, "exposed_exchange_ms"):
            trainer.shared.exposed_exchange_ms()          # (forget the warm-up's waits)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rep = trainer.run_epoch(epoch)
            kernel_ms.append(float(rep["sgd_kernel_ms"][0]))
            if sess is not None and sess.geometry():
                shader_mhz.append(float(sess.geometry().get("shader_mhz", 0.0)))
            draws += int(rep["n_draws"][0])
            epoch += 1
        trainer.finish()                  # (late merge: the final blocking exchange belongs to the timed region)
        barrier()
        elapsed = time.perf_counter() - t0
    # N > 1: what an exchange costs by itself (one blocking all-reduce of a bucket-sized buffer, HIP events) and how much of the
    # exchanges the rank's stream actually waited for (SharedTables.exposed_exchange_ms: ~0 when the late merge hides them)
    exchange_ms = exposed_ms = None
    n_exchanges = 0
    if world > 1:
        sh_ = trainer.shared
        bf16 = args.exchange_dtype == "bf16"      # (what exchange_fused hands to RCCL: the deltas as bfloat16 + the fp32 tail beside them)
        scratch = torch.zeros(sh_._tail_at, dtype=torch.bfloat16, device=sh_.flat.device) if bf16 else torch.zeros_like(sh_.flat)
        scratch_tail = torch.zeros_like(sh_.tail) if bf16 else None

        def one_exchange():
            if bf16:
                work = dist.all_reduce(scratch, async_op=True)
                dist.all_reduce(scratch_tail)
                work.wait()
            else:
                dist.all_reduce(scratch)
        one_exchange()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        one_exchange()
        e1.record()
        torch.cuda.synchronize()
        exchange_ms = float(e0.elapsed_time(e1))
        del scratch
        if trainer.late and hasattr(trainer.shared, "exposed_exchange_ms"):
            exposed_ms, n_exchanges = trainer.shared.exposed_exchange_ms()
        else:
            # blocking exchanges are exposed by construction: every one of them, for its whole length
            n_exchanges = sum(trainer.exchanges_in_epoch(e) for e in range(args.warmup, args.warmup + args.steps))
            exposed_ms = exchange_ms * n_exchanges
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ll_last = float(rep["log_likelihood"][0])
    assert np.isfinite(ll_last), "training diverged"
    strong_rec = None
    if world > 1 and args.config == "C2" and not args.no_strong and not share_gpu:
        strong_rec = strong_scaling_record(world, rank, device, max(2, min(args.steps, 5)), 1, barrier)

    if rank == 0:
        N = n_local
        total_updates = float(n_job) * args.steps
        value = total_updates / elapsed
        launches = rep["launches_per_epoch"]
        mean_draws = draws / (float(N) * args.steps)
        bytes_per_update = algorithmic_bytes_per_update(F, n_uf, n_if, mean_draws)
        k_ms = float(np.mean(kernel_ms)) / launches                 # average duration of ONE SGD launch
        rows_per_launch = N / launches
        # achieved = updates/s x algorithmic bytes per update -- from `value` (the whole step: SGD launch, epoch
        # tail, host), per GPU; the SGD launch by itself (HIP events around it, what rocprofv3's per-kernel average is compared with)
        # is reported beside it as `kernel_only`
        achieved = value / world * bytes_per_update / 1e9
        achieved_kernel = bytes_per_update * rows_per_launch / (k_ms * 1e-3) / 1e9
        traffic = atomic_requests = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")      # PMC-derived HBM bytes / fabric requests per launch, when collected
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = args.config
                traffic = tj.get(key + "_hbm_bytes_per_launch")
                atomic_requests = tj.get(key + "_atomic_requests_per_launch")      # TCC_EA0_ATOMIC of the final tree's kernel (config 2)
            except Exception:
                traffic = None
        # run-to-run / box-to-box: the same binary differs by 10 - 25 % between boxes of one pool (profiles/r03_notes.md), so the line
        # carries the spread over its own steps, the shader clock the launches ran at, and the kernel time scaled to a 2.4 GHz clock
        k_all = np.array(kernel_ms) / launches
        mhz = float(np.median(shader_mhz)) if shader_mhz and np.median(shader_mhz) > 0 else None
        if strong:
            what = ("%s: user shard %d of %d of ONE synthetic data set of %d users x %d items x %d interactions (this GPU: %d users, %d "
                    "interactions, all items)" % (args.config, 0 if world == 1 else rank, world if world > 1 else max(args.share, 1), U, I, cfg["n_interactions"], u_local, n_local))
        else:
            what = "%s: synthetic %d users x %d items x %d interactions per GPU" % (args.config, U, I, N)
        geo = sess.geometry() if sess is not None else {}
        sampler = "negatives drawn uniformly over the whole catalogue (the reference's sampler, rankfm/_rankfm.pyx:250-253), user segments <= %d rows" % (geo.get("segment_rows") or 32)
        calls = ("ONE resident rfm_fit_device call for the %d timed epochs (per epoch: SGD launch + epoch tail), item-side weights kept in the engine's "
                 "layout between calls" % args.steps) if resident else "one rfm_fit_device call per epoch / exchange window (layouts converted per call)"
        peak_measured = None
        if world == 1:
            import ctypes as C
            from rankfm_amd import _hip
            rd, cp = C.c_double(0.0), C.c_double(0.0)
            if _hip.lib().rfm_hbm_probe(C.c_size_t(2 << 30), 5, C.byref(rd), C.byref(cp)) == 0:
                peak_measured = {"read": rd.value, "copy": cp.value, "unit": "GB/s",
                                 "how": "rfm_hbm_probe: 16 B/lane streaming kernel over 2 GiB, best of 5 launches (copy = read + write bytes)"}
        out = {
            "metric": "(user,item,neg) pairwise updates/sec at k=64; achieved HBM GB/s vs peak",
            "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, factors=%d, loss=%s%s%s, learning_rate=%g, zipf_s=%g, hogwild fp32 atomics, counter RNG, %s"
                                   % (what, F, cfg["loss"], " max_samples=%d" % cfg["max_samples"] if cfg["max_samples"] > 1 else "",
                                      ", %d + %d dense user/item features" % (n_uf, n_if) if n_uf or n_if else "", lr, args.zipf, sampler),
                       "calls": calls,
                       "geometry": {"segment_rows": geo.get("segment_rows", 0), "workgroups": geo.get("workgroups", 0)},
                       "n_users_total": u_local * world, "n_interactions_total": n_job, "parallelism": "user-shard dp%d" % world,
                       "rccl_ranks_seen": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else (1 if world == 1 else 0)),
                       "collective_backend": dist.get_backend() if world > 1 else None,
                       "merge_rule": ("curvature rule in ONE all-reduce of the bucket per exchange (SharedTables.exchange_fused); exchanges per epoch: %s%s"
                                      % (args.syncs_per_epoch, " = 8 during a fit's first 8 epochs, 1 afterwards (the timed steps are epochs %d .. %d)"
                                         % (args.warmup, args.warmup + args.steps - 1) if args.syncs_per_epoch == "auto" else "")) if world > 1 else None,
                       # (N > 1) one all-reduce of the bucket by itself, and the stream time per epoch spent WAITING for exchanges
                       "exchange": ({"payload_bytes": trainer.shared.payload_bytes, "exchange_ms": exchange_ms,
                                     "mode": args.exchange, "dtype": args.exchange_dtype, "late_merge_in_use": bool(trainer.late), "decision": getattr(trainer, "overlap_decision", None),
                                     "exposed_exchange_ms_per_epoch": (exposed_ms / args.steps) if exposed_ms is not None else None,
                                     "waits_timed": n_exchanges} if world > 1 else None),
                       "sgd_launches_per_epoch": launches, "waves_per_launch": rep["waves_per_launch"],
                       "mean_draws_per_update": mean_draws,
                       # WARP lines also carry the rate of sampled negatives (accepted draws, whole job)
                       "sampled_negatives_per_s": float(draws) * world / elapsed,
                       "final_mean_ll_per_update": ll_last / N},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "how": "achieved = value / n_gpus x algorithmic_bytes_per_update (SURVEY.md 8d: the whole step on the wall clock); kernel_only = the "
                                "SGD launch alone, algorithmic bytes of a launch / its HIP-event duration (compare with rocprofv3's average for `kernel`)",
                         "kernel_only": {"achieved": achieved_kernel, "frac": achieved_kernel / HBM_PEAK_GBPS, "ms_per_launch": k_ms,
                                         "step_minus_kernel_ms": elapsed / args.steps * 1e3 - k_ms * launches},
                         "peak_measured": peak_measured,
                         "kernel": kernel_name(F, cfg["max_samples"], n_uf, n_if),
                         "kernel_ms_per_launch": k_ms,
                         "kernel_ms_min": float(k_all.min()), "kernel_ms_median": float(np.median(k_all)), "kernel_ms_max": float(k_all.max()),
                         "frac_best_step": bytes_per_update * rows_per_launch / (float(k_all.min()) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "shader_mhz": mhz,
                         "kernel_ms_median_at_2400mhz": (float(np.median(k_all)) * mhz / 2400.0) if mhz else None,
                         # what bounds the BPR kernel (profiles/r05_notes.md): the memory-side fp32 atomic path retires 20.5 G 64-byte
                         # requests/s chip-wide on uniform targets, ~16 G/s at config 2's address mix; a BPR update issues 5 (negative) +
                         # 5 x the share of positives outside the 64 LDS-accumulated rows + the hot rows' publications and sweeps
                         "atomic_request_capacity_per_s": {"uniform_targets": 20.5e9, "config2_mix_row_major": 16.7e9, "config2_mix_segment_major": 18.3e9,
                                                           "source": "tools/microbench/pipe_model.hip, atomic_skew.hip (round 5)"},
                         # (PMC count of the profiled run / this run's kernel time: the kernel runs AT its mix's capacity)
                         "atomic_requests_per_update": (atomic_requests / rows_per_launch) if atomic_requests else None,
                         "atomic_requests_per_s": (atomic_requests / (k_ms * 1e-3)) if atomic_requests else None,
                         "algorithmic_bytes_per_update": bytes_per_update, "rows_per_launch": rows_per_launch},
        }
        if strong_rec is not None:
            out["strong_scaling"] = strong_rec
        if world == 1 and args.config == "C2" and not strong and args.also:
            out["also"] = [also_workload(name, device, shard, x_if, w) for name in args.also.split(",") if name]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shard, x_if, w, hyper, int(n_uf > 0), int(n_if > 0))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
