"""CPU experiment behind the table trainer's quota (profiles/r04_notes.md section 11): does RANKING QUALITY depend on how often the dense
feature tables are trained when nothing else differs?  The SEQUENTIAL oracle (reference sampler, keyed row shuffle) on planted problems
with tags that carry signal, the tables updated on every n-th visited row only (`table_every`, oracle/rfm_oracle.c: analysis option) --
optionally with `--tail K`: K extra table-only steps on random rows after every epoch, the rows frozen (what a tables kernel that
outlasts the row loops does), or `--tail -P`: the last P % of every epoch's rows do not train the tables (a trainer that finishes early);
`--batch B`: the tables' updates are scored on a snapshot of the tables taken every B table-training visits (the engine's trainer
applies batches of 64 staged steps scored on one table state), `--step s`: the tables' step length scaled by s.  hit_rate@10 on the
held-out pairs, evaluated on the CPU.  Analysis tooling (uses oracle/), not product.

    python tools/table_quota_standin.py [--users 30000 --items 12000 --seeds 3 --every 1,30,100,335,1000,3000] [--tail=0,2000] [--batch 1,64] [--step 1,0.25]
    (every combination of the four lists is run)"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WE = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")


def hit_rate_cpu(m, test_pairs, k=10, chunk=2048):
    """evaluation.hit_rate restated with numpy on the model's index space (train items of a user are not recommended; users without a
    test item do not count) -- rankfm/evaluation.py:9-33 through rankfm.py's recommend(filter_previous=True)"""
    u_idx = pd.Series(m.user_to_index)
    i_idx = pd.Series(m.item_to_index)
    tu, ti = u_idx.reindex(test_pairs[:, 0]).values, i_idx.reindex(test_pairs[:, 1]).values
    ok = ~(np.isnan(tu) | np.isnan(ti))
    tu, ti = tu[ok].astype(np.int64), ti[ok].astype(np.int64)
    order = np.argsort(tu, kind="stable")
    tu, ti = tu[order], ti[order]
    users, starts = np.unique(tu, return_index=True)
    ends = np.append(starts[1:], len(tu))
    Ueff = m.v_u + m.x_uf @ m.v_uf
    Veff = m.v_i + m.x_if @ m.v_if
    bias = m.w_i + m.x_if @ m.w_if
    off, items = m.user_items.offsets, m.user_items.items
    hits = 0
    for c0 in range(0, len(users), chunk):
        us = users[c0:c0 + chunk]
        S = Ueff[us] @ Veff.T + bias[None, :]
        for r, u in enumerate(us):
            S[r, items[off[u]:off[u + 1]]] = -np.inf
        top = np.argpartition(-S, k, axis=1)[:, :k]
        for r in range(len(us)):
            t = ti[starts[c0 + r]:ends[c0 + r]]
            hits += bool(np.intersect1d(top[r], t).size)
    return hits / len(users)


def job(spec):
    seed, every, tail, batch, tstep, a = spec
    from oracle import oracle as orc
    from rankfm_amd import EngineOptions, RankFM, synthetic
    orc.build()
    d = synthetic.make_planted(seed=seed, n_users=a["users"], n_items=a["items"], mean_degree=a["degree"], n_tags=a["tags"])
    train = pd.DataFrame(d["train"], columns=["u", "i"])
    us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
    uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
    itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
    m = RankFM(factors=a["factors"], loss="bpr", learning_rate=a["lr"], engine=EngineOptions(seed=100 + seed))
    np.random.seed(seed)
    m._init_all(train, uf, itf)
    t0 = time.time()
    kw = dict(table_every=every, table_batch=batch, table_step=(0.0 if tstep == 1.0 else tstep))
    if tail > 0:
        kw["table_tail"] = tail
    elif tail < 0:                      # (--tail -15 = the last 15 % of every epoch's rows do not train the tables)
        kw["table_quiet_rows"] = int(len(m.interactions) * (-tail) / 100.0)
    out = orc.fit(m.interactions, m.sample_weight, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i, m.v_uf, m.v_if,
                  m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent, 1, a["epochs"], perms=None, rng_mode=orc.RNG_COUNTER,
                  seed=100 + seed, membership="binary", **kw)
    m.is_fit = True
    hr = hit_rate_cpu(m, d["test"])
    return seed, (every, tail, batch, tstep), hr, float(out["ll64"][-1]) / len(m.interactions), {k: float(np.linalg.norm(getattr(m, k))) for k in WE}, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=30_000)
    ap.add_argument("--items", type=int, default=12_000)
    ap.add_argument("--degree", type=float, default=60.0)
    ap.add_argument("--tags", type=int, default=8)
    ap.add_argument("--factors", type=int, default=32)
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--every", default="1,30,100,335,1000,3000")
    ap.add_argument("--tail", default="0", help="comma list: K > 0 = K table-only visits after every epoch (table_tail); -P = the last P %% of every epoch's rows do not train the tables (table_quiet_rows)")
    ap.add_argument("--batch", default="1", help="comma list: table-training visits scored on one snapshot of the tables (1 = the current tables)")
    ap.add_argument("--step", default="1", help="comma list: scale of the tables' step length")
    ap.add_argument("--processes", type=int, default=min(8, os.cpu_count() or 1))
    a = vars(ap.parse_args())
    specs = [(s, int(e), int(t), int(b), float(ts), a) for t in a["tail"].split(",") for e in a["every"].split(",") for b in a["batch"].split(",")
             for ts in a["step"].split(",") for s in range(a["seeds"])]
    t0 = time.time()
    with mp.get_context("spawn").Pool(a["processes"]) as pool:
        res = pool.map(job, specs, chunksize=1)
    by = {}
    for seed, key, hr, ll, norms, dt in res:
        by.setdefault(key, []).append((seed, hr, ll, norms, dt))
    print("%d users x %d items, %d + %d tags, k = %d, lr %.3f, %d epochs, %d seeds (%.0f s)" % (a["users"], a["items"], a["tags"], a["tags"], a["factors"], a["lr"],
                                                                                           a["epochs"], a["seeds"], time.time() - t0))
    for (every, tail, batch, tstep), rows in sorted(by.items()):
        rows.sort()
        print("tables on every %5d-th row, tail %6d, batch %3d, step x %.3g: hit_rate@10 %s mean %.4f | LL/N %.4f | |w_i| %.2f |v_uf| %.3f |v_if| %.3f |w_if| %.3f  (%.0f s per run)"
              % (every, tail, batch, tstep, [round(r[1], 4) for r in rows], np.mean([r[1] for r in rows]), np.mean([r[2] for r in rows]), np.mean([r[3]["w_i"] for r in rows]),
                 np.mean([r[3]["v_uf"] for r in rows]), np.mean([r[3]["v_if"] for r in rows]), np.mean([r[3]["w_if"] for r in rows]), np.mean([r[4] for r in rows])), flush=True)


if __name__ == "__main__":
    main()
