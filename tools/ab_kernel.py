"""GPU-box experiment: A/B timing of engine variants on ONE box inside ONE process (boxes differ by up to 10 % -- profiles/r03_notes.md
section 3 -- so variants are only ever compared inside one run of this tool).  Every variant gets a DeviceSession of its own on the
same data and the same initial weights; the measured epochs are INTERLEAVED (A B C A B C ...) so that clock drift hits all alike.

    python tools/ab_kernel.py --config C2 --variants "base;flags=128;flags=256;flags=384" [--epochs 6] [--rounds 4]

A variant is a ','-separated list of  flags=<debug_flags>  damping=<M>  workgroups=<n>  <tune_name>=<int>.
Prints per variant the SGD launch time (HIP events inside rfm_fit_device): min / median / mean over the measured epochs, the
updates/s of the median, the log-likelihood of the last epoch (a sanity check that the variant still trains the same model) and
the norms of v_u, v_i, w_i.  Measurement tooling, not product."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rankfm_amd import synthetic                       # noqa: E402


def parse_variant(text):
    kw = dict(debug_flags=0, hogwild_damping=0.0, n_workgroups=0, tune={})
    for part in [p for p in text.split(",") if p and p != "base"]:
        k, v = part.split("=")
        if k == "flags":
            kw["debug_flags"] = int(v)
        elif k == "damping":
            kw["hogwild_damping"] = float(v)
        elif k == "workgroups":
            kw["n_workgroups"] = int(v)
        else:
            kw["tune"][k] = int(v)
    return kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--variants", default="base;flags=128;flags=256;flags=384")
    ap.add_argument("--epochs", type=int, default=6, help="measured epochs per round and variant")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--zipf", type=float, default=1.0)
    ap.add_argument("--print-ll", action="store_true", help="also print every epoch's log-likelihood (sum over the rows)")
    a = ap.parse_args()
    cfg = synthetic.CONFIGS[a.config]
    t0 = time.time()
    if a.config in ("C4", "C5"):
        sh = synthetic.make_config_shard(a.config, rank=0, world=8, zipf_s=a.zipf)
        data = (sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"])
        weights = sh["weights"]
        n_uf, n_if = cfg.get("n_user_features", 0), cfg.get("n_item_features", 0)
    else:
        U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
        pairs, csr = synthetic.make_interactions(U, I, N, seed=0, zipf_s=a.zipf)
        data = (pairs, np.ones(N, np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32))
        weights = synthetic.init_weights(U, I, F, seed=1492)
        n_uf = n_if = 0
    N = len(data[0])
    print("data ready after %.1f s: %d rows" % (time.time() - t0, N), flush=True)
    import torch
    from rankfm_amd.engine import DeviceSession
    hyper = dict(learning_rate=cfg.get("learning_rate", 0.1), max_samples=cfg["max_samples"])
    names = a.variants.split(";")
    sessions, times, last, lls = [], [[] for _ in names], [None] * len(names), [[] for _ in names]
    for name in names:
        s = DeviceSession(*data, {k: np.array(v, copy=True) for k, v in weights.items()}, seed=1492, has_user_features=int(n_uf > 0),
                          has_item_features=int(n_if > 0), **hyper, **parse_variant(name))
        if a.warmup > 0:
            lls[len(sessions)].extend(float(x) for x in s.run(epochs=a.warmup)["log_likelihood"])
        sessions.append(s)
    epoch = a.warmup
    for r in range(a.rounds):
        for k, s in enumerate(sessions):
            rep = s.run(epochs=a.epochs, epoch_begin=epoch)
            times[k].extend(float(x) for x in rep["sgd_kernel_ms"])
            lls[k].extend(float(x) for x in rep["log_likelihood"])
            last[k] = rep
        epoch += a.epochs
    torch.cuda.synchronize()
    for k, name in enumerate(names):
        t = np.array(times[k])
        w = sessions[k].weights
        g = sessions[k].geometry()
        print("%-40s kernel ms min %.3f median %.3f mean %.3f  -> %.0f M updates/s  LL/N %.5f  draws/N %.2f  |v_u| %.3f |v_i| %.3f |w_i| %.4f  wg %d"
              % (name, t.min(), np.median(t), t.mean(), N / np.median(t) / 1e3, last[k]["log_likelihood"][-1] / N, last[k]["n_draws"][-1] / N,
                 float(w["v_u"].norm()), float(w["v_i"].norm()), float(w["w_i"].norm()), g["workgroups"]), flush=True)
        if g.get("table_producers", 0) > 0:
            print("    tables: %d producers, %d staged steps over the last call (every %.0f-th row), kernels overlapped %d us (tables %d us, rows %d us)  wg size %d"
                  % (g["table_producers"], g["table_steps"], N * len(last[k]["sgd_kernel_ms"]) / max(g["table_steps"], 1), g.get("table_overlap_us", -1),
                     g.get("table_span_us", [0, 0])[0], g.get("table_span_us", [0, 0])[1], g["groups_per_workgroup"]), flush=True)
        if a.print_ll:
            print("    LL per epoch (from the initial weights): " + " ".join("%.1f" % x for x in lls[k]), flush=True)


if __name__ == "__main__":
    main()
