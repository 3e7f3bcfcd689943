"""The REFERENCE'S OWN seed-to-seed spread of a feature model (build container only; see make_golden.py for how the reference is loaded):
what the tolerances on the engine's feature tables and ranking quality are anchored on (VERDICT r05, item 1 -- SURVEY section 8 row a6).

Two problems, EIGHT runs of the reference each, from ONE set of initial weights -- only the order of the rows differs from run to run
(`np.random.seed(1000 + s)` in front of the reference's `_fit`, whose `np.random.shuffle` is its only use of numpy's generator,
rankfm/_rankfm.pyx:197,227; its MT19937 negative stream restarts at 1492 in every call, :182).  That is exactly what separates the engine
from the reference in the parity tests: same data, same initial weights, another visiting order (and other draws).

  tags   the ranking-quality fixture of make_quality_tags_golden.py, data seeds 0 .. 4: 3,000 users x 2,000 items, 8 + 8 binary tags that
         carry signal, factors 20, BPR, learning rate 0.03, 5 epochs -> hit_rate@10 and the Frobenius norms of the six arrays per (data
         seed, run): [5, 8, 7].  (quality_planted_tags.npz holds ONE reference run per data seed; one run's hit rate moves by 1.6 points
         with the order alone, so the mean of those five carries +-0.7 point -- the forty runs here pin the reference's mean to +-0.25.)
         Plus, for data seed 0, eight runs in which the initial weights vary with the run as well (`np.random.seed(s)` before `fit`: the
         reference user's own run-to-run spread).
  c4r    BASELINE config 4 reduced 1 : 80 with its proportions kept (12,500 users x 20,000 items x 625,000 interactions: 50 per user,
         31 per item; 32 + 32 Bernoulli(0.25) tags without signal, factors 64, BPR, learning rate 0.03 -- BASELINE.md section 5) -> the
         norms of the six arrays after the first and after the second epoch per run.  (The tables' norms are set by the last ~1 / (2 beta
         eta) = 170 rows of the stream -- an exponential moving average of gradient noise -- so their RELATIVE spread does not depend on
         the size of the problem: this reduction anchors the full-size test of tests/test_gpu_configs.py.)

Only numbers are stored (`quality_tags_spread.npz`); the data are regenerated from the seeds."""
import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader  # noqa: E402
from rankfm_amd import synthetic  # noqa: E402

RankFM, ref_ext, ev = ref_loader.load_reference()
RUNS = 8
NAMES = ("v_u", "v_i", "w_i", "v_uf", "v_if", "w_if")
TAGS = dict(n_users=3000, n_items=2000, mean_degree=100.0, n_tags=8)
C4R = dict(n_users=12_500, n_items=20_000, n_interactions=625_000, factors=64, n_tags=32, learning_rate=0.03)


def norms(m):
    return [float(np.linalg.norm(np.asarray(m[k] if isinstance(m, dict) else getattr(m, k)).astype(np.float64))) for k in NAMES]


def tags_problem(data_seed=0):
    d = synthetic.make_planted(seed=data_seed, **TAGS)
    train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
    uf = pd.DataFrame(np.column_stack([np.arange(len(d["user_tags"])), d["user_tags"]]))
    itf = pd.DataFrame(np.column_stack([np.arange(len(d["item_tags"])), d["item_tags"]]))
    return train, test, uf, itf


def tags_runs():
    order_only, full = [], []
    for data_seed in range(5):
        train, test, uf, itf = tags_problem(data_seed)
        rows = []
        for s in range(RUNS):
            m = RankFM(factors=20, loss="bpr", learning_rate=0.03)
            np.random.seed(data_seed)
            m._init_all(train, uf, itf, None)             # ONE set of initial weights per data seed (rankfm/rankfm.py:214-244) -- the one
            #                                               `np.random.seed(data_seed); fit(...)` starts from -- ...
            np.random.seed(1000 + s)                      # ... and the run's own shuffle (rankfm/_rankfm.pyx:227)
            ref_ext._fit(m.interactions, m.sample_weight, m.user_items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i, m.v_uf, m.v_if,
                         m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent, 1, 5, False)
            m.is_fit = True
            rows.append([ev.hit_rate(m, test, k=10)] + norms(m))
            print("tags order-only", data_seed, s, np.round(rows[-1], 4), flush=True)
        order_only.append(rows)
    train, test, uf, itf = tags_problem(0)
    for s in range(RUNS):
        m = RankFM(factors=20, loss="bpr", learning_rate=0.03)
        np.random.seed(s)
        m.fit(train, user_features=uf, item_features=itf, epochs=5)
        full.append([ev.hit_rate(m, test, k=10)] + norms(m))
        print("tags full", s, np.round(full[-1], 4), flush=True)
    return np.array(order_only), np.array(full)


def c4r_problem():
    """the data of tests/test_gpu_configs.py's reduced config-4 problem (the test regenerates them from the same calls)"""
    c = C4R
    pairs, csr = synthetic.make_interactions(c["n_users"], c["n_items"], c["n_interactions"], seed=4)
    w = synthetic.init_weights(c["n_users"], c["n_items"], c["factors"], c["n_tags"], c["n_tags"], seed=1492)
    x_uf, x_if = synthetic.make_features(c["n_users"], c["n_tags"], 7), synthetic.make_features(c["n_items"], c["n_tags"], 8)
    return pairs, csr, w, x_uf, x_if


def c4r_runs():
    pairs, csr, w0, x_uf, x_if = c4r_problem()
    user_items = {u: csr.items[csr.offsets[u]:csr.offsets[u + 1]] for u in range(C4R["n_users"])}
    sw = np.ones(len(pairs), np.float32)
    out = []
    for s in range(RUNS):
        w = {k: v.copy() for k, v in w0.items()}
        np.random.seed(1000 + s)
        row = []
        t0 = time.time()
        # (two calls of one epoch each: the reference restarts its MT stream and its learning-rate schedule per call, and both are
        #  constant here -- the second call continues numpy's shuffle stream like a two-epoch call would)
        for _ in range(2):
            ref_ext._fit(pairs, sw, user_items, x_uf, x_if, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"],
                         0.01, 0.1, C4R["learning_rate"], "constant", 0.25, 1, 1, False)
            row += norms(w)
        out.append(row)
        print("c4r", s, np.round(row, 4), "%.0f s" % (time.time() - t0), flush=True)
    return np.array(out)


if __name__ == "__main__":
    tags_order, tags_full = tags_runs()
    c4r = c4r_runs()
    np.savez(os.path.join(HERE, "quality_tags_spread.npz"),
             tags_columns=np.array(["hit_rate"] + ["norm_" + k for k in NAMES]), tags_order_only=tags_order, tags_full=tags_full,
             c4r_columns=np.array(["e1_norm_" + k for k in NAMES] + ["e2_norm_" + k for k in NAMES]), c4r_order_only=c4r)
    print("tags, order only: mean over the 40 runs", np.round(tags_order.mean(axis=(0, 1)), 4), "sigma of a run around its data seed's mean",
          np.round(np.sqrt(((tags_order - tags_order.mean(axis=1, keepdims=True)) ** 2).sum(axis=(0, 1)) / (5 * (RUNS - 1))), 4))
    for name, a in (("tags, weights + order (data seed 0)", tags_full), ("c4r, order only", c4r)):
        print(name, "mean", np.round(a.mean(axis=0), 4), "relative sigma", np.round(a.std(axis=0, ddof=1) / np.abs(a.mean(axis=0)), 4))
