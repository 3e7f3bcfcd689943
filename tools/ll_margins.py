"""GPU-box measurement behind the tolerances of the statistical parity tests (tests/test_gpu_parity.py, tests/test_gpu_configs.py):
every comparison of the Hogwild engine with the sequential oracle at BASELINE configs 2 / 3 / 4-share, REPEATED `--runs` times on
the GPU (Hogwild is not bit-reproducible), with the log-likelihood taken against the oracle's DOUBLE sum (`ll64`; the reference's
float accumulator `ll` is printed beside it: it is off by -0.5 % ... +0.6 % at these sizes, profiles/r02_notes.md).

    python tools/ll_margins.py [--runs 4] [--configs C2,C3,C4] [--procs 8]

`--damped` also runs the oracle under the engine's step damping (what is left between the two is asynchrony alone).
The oracle runs of all repetitions are farmed out to a process pool that is forked BEFORE the GPU is touched.
Test infrastructure (uses oracle/), not product."""
import argparse
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _gpulock import gpu                              # noqa: E402
from oracle import oracle as orc                      # noqa: E402
from rankfm_amd import order, synthetic               # noqa: E402

WEIGHTS = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")
DATA = {}          # name -> problem dict, filled before the pool is forked (workers inherit it)
TMP = tempfile.mkdtemp(prefix="llm_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)


def save_weights(w, tag):
    path = os.path.join(TMP, tag + ".npz")
    np.savez(path, **{k: np.asarray(w[k]) for k in WEIGHTS})
    return path


def oracle_task(spec):
    """one sequential oracle run in the engine's order: spec = dict(data, weights path, max_samples, epochs (list of epoch indexes),
    seed, lr, geometry | None (None = the reference's whole-catalogue sampler), pos_step / user_step paths | None)"""
    d = DATA[spec["data"]]
    z = np.load(spec["weights"])
    w = {k: np.array(z[k], copy=True) for k in WEIGHTS}
    geo = spec["geometry_order"]
    epochs = spec["epochs"]
    perms = np.stack([order.epoch_positions(d["csr_offsets"], spec["seed"], e, geo.get("segment_rows") or None) for e in epochs]).astype(np.int32)
    extra = {}
    if spec.get("steps"):
        s = np.load(spec["steps"])
        extra.update(pos_step=s["pos"], user_step=s["user"], neg_step=s["pos"])      # (an item's scale applies on either side of the pair)
    t0 = time.time()
    out = orc.fit(d["pairs_csr"], d["sw_csr"], d["csr_offsets"], d["csr_items"], d["x_uf"], d["x_if"], w["w_i"], w["w_if"], w["v_u"], w["v_i"],
                  w["v_uf"], w["v_if"], 0.01, 0.1, spec["lr"], "constant", 0.25, spec["max_samples"], len(epochs), perms=perms,
                  rng_mode=orc.RNG_COUNTER, seed=spec["seed"], membership="binary", epoch_begin=epochs[0],
                  want_negatives=spec["max_samples"] > 1, **extra)
    res = dict(tag=spec["tag"], ll=out["ll"], ll64=out["ll64"], norms={k: float(np.linalg.norm(w[k])) for k in WEIGHTS}, seconds=time.time() - t0)
    if out["nsamp"] is not None:
        res["draws"] = out["nsamp"].sum(axis=1)
    return res


def problem(name, pairs, sw, off, items, x_uf, x_if, weights):
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    DATA[name] = dict(pairs=pairs, sw=sw, csr_offsets=off, csr_items=items, x_uf=x_uf, x_if=x_if, weights=weights,
                      pairs_csr=np.ascontiguousarray(pairs[by_csr]), sw_csr=np.ascontiguousarray(sw[by_csr]))


def fmt(x):
    return "[" + ", ".join("%+.3f%%" % (100.0 * v) for v in np.atleast_1d(x)) + "]"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--configs", default="C2,C3,C4")
    ap.add_argument("--procs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--damped", action="store_true")
    ap.add_argument("--tune", default="", help="geometry overrides for the config-4 sessions: 'table_producers=6'")
    ap.add_argument("--debug-flags", type=int, default=0, help="rfm_fit_config.debug_flags of the config-4 sessions (64: no opening launch)")
    a = ap.parse_args()
    configs = a.configs.split(",")
    orc.build()
    t0 = time.time()
    if "C2" in configs or "C3" in configs:
        cfg = synthetic.CONFIGS["C2"]
        U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
        pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
        problem("C2", pairs, np.ones(N, np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32),
                synthetic.init_weights(U, I, F, seed=1492))
    if "C4" in configs:
        sh = synthetic.make_config_shard("C4", rank=0, world=8)
        problem("C4", sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"], sh["weights"])
    print("data ready after %.1f s; %d oracle processes" % (time.time() - t0, a.procs), flush=True)
    pool = mp.get_context("fork").Pool(a.procs)          # forked before anything touches the GPU

    import torch                                         # noqa: F401  (after the fork)
    from rankfm_amd.engine import DeviceSession

    def session(name, **kw):
        d = DATA[name]
        return DeviceSession(d["pairs"], d["sw"], d["csr_offsets"], d["csr_items"], d["x_uf"], d["x_if"], d["weights"], **kw)

    pending, runs = [], {}

    def submit(**spec):
        pending.append(pool.apply_async(oracle_task, (spec,)))

    # ---- config 2: two epochs from the initial weights ---------------------------------------------------------------------
    if "C2" in configs:
        w0p = save_weights(DATA["C2"]["weights"], "c2_init")
        for r in range(a.runs):
            s = session("C2", max_samples=1, seed=1492)
            with gpu():
                rep = s.run(epochs=2)
            g = s.weights_to_host()
            runs["C2:%d" % r] = dict(ll=rep["log_likelihood"].copy(), ms=rep["sgd_kernel_ms"].copy(),
                                     norms={k: float(np.linalg.norm(g[k])) for k in WEIGHTS})
            if r == 0:
                geo = s.geometry()
                print("C2 geometry %s" % {k: geo[k] for k in ("workgroups", "working_groups", "segment_rows")}, flush=True)
                submit(tag="C2:oracle", data="C2", weights=w0p, max_samples=1, epochs=[0, 1], seed=1492, lr=0.1, geometry_order=geo)
                if a.damped:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    pos, user = s.step_scales()
                    sp = os.path.join(TMP, "c2_steps.npz")
                    np.savez(sp, pos=pos, user=user)
                    submit(tag="C2:damped_oracle", data="C2", weights=w0p, max_samples=1, epochs=[0, 1], seed=1492, lr=0.1,
                           geometry_order=geo, steps=sp)
            del s
    # ---- config 3: three epochs of training, then one compared epoch from the same weights ------------------------------------
    if "C3" in configs:
        for r in range(a.runs):
            s = session("C2", max_samples=50, seed=1492)
            with gpu():
                s.run(epochs=3)
            wp = save_weights(s.weights_to_host(), "c3_w0_%d" % r)
            with gpu():
                rep = s.run(epochs=1, epoch_begin=3)
            g = s.weights_to_host()
            runs["C3:%d" % r] = dict(ll=rep["log_likelihood"].copy(), ms=rep["sgd_kernel_ms"].copy(), draws=rep["n_draws"].copy(),
                                    norms={k: float(np.linalg.norm(g[k])) for k in WEIGHTS})
            submit(tag="C3:%d:oracle" % r, data="C2", weights=wp, max_samples=50, epochs=[3], seed=1492, lr=0.1, geometry_order=s.geometry())
            if a.damped and r == 0:
                pos, user = s.step_scales()
                sp = os.path.join(TMP, "c3_steps.npz")
                np.savez(sp, pos=pos, user=user)
                submit(tag="C3:0:damped_oracle", data="C2", weights=wp, max_samples=50, epochs=[3], seed=1492, lr=0.1, geometry_order=s.geometry(), steps=sp)
            del s
    # ---- config 4, one GPU's share: first epoch from the initial weights, second epoch from the GPU's weights -----------------
    if "C4" in configs:
        lr = synthetic.CONFIGS["C4"]["learning_rate"]
        w0p = save_weights(DATA["C4"]["weights"], "c4_init")
        for r in range(a.runs):
            s = session("C4", max_samples=1, seed=1492, learning_rate=lr, debug_flags=a.debug_flags, tune={kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tune.split(",") if kv})
            with gpu():
                rep1 = s.run(epochs=1)
            g1 = s.weights_to_host()
            wp = save_weights(g1, "c4_w1_%d" % r)
            with gpu():
                rep2 = s.run(epochs=1, epoch_begin=1)
            g2 = s.weights_to_host()
            print("C4 run %d: table trainer: %d producers, %d staged steps in the last epoch = every %.0f-th row" % (
                r, s.geometry()["table_producers"], s.geometry()["table_steps"], len(DATA["C4"]["pairs"]) / max(s.geometry()["table_steps"], 1)),
                  "  us: trainer waited / ran, producers waited / ran (summed)", s.geometry()["feat_diag"], flush=True)
            runs["C4:e1:%d" % r] = dict(ll=rep1["log_likelihood"].copy(), ms=rep1["sgd_kernel_ms"].copy(), norms={k: float(np.linalg.norm(g1[k])) for k in WEIGHTS})
            runs["C4:e2:%d" % r] = dict(ll=rep2["log_likelihood"].copy(), ms=rep2["sgd_kernel_ms"].copy(), norms={k: float(np.linalg.norm(g2[k])) for k in WEIGHTS})
            if r == 0:
                submit(tag="C4:e1:oracle", data="C4", weights=w0p, max_samples=1, epochs=[0], seed=1492, lr=lr, geometry_order=s.geometry())
            submit(tag="C4:e2:%d:oracle" % r, data="C4", weights=wp, max_samples=1, epochs=[1], seed=1492, lr=lr, geometry_order=s.geometry())
            del s
    print("GPU runs done after %.1f s; waiting for %d oracle runs" % (time.time() - t0, len(pending)), flush=True)
    ora = {}
    for p in pending:
        res = p.get()
        ora[res["tag"]] = res
        print("  oracle %-28s %.1f s" % (res["tag"], res["seconds"]), flush=True)
    pool.close()

    def compare(gkey, okey, names=("v_u", "v_i", "w_i")):
        g, o = runs[gkey], ora[okey]
        line = "%-16s vs %-24s LL/ll64 - 1 %s  (vs float sum %s)  norms - 1 %s  kernel ms %s" % (
            gkey, okey, fmt(g["ll"] / o["ll64"] - 1.0), fmt(g["ll"] / o["ll"] - 1.0),
            " ".join("%s %+.2f%%" % (k, 100.0 * (g["norms"][k] / o["norms"][k] - 1.0)) for k in names), np.round(g["ms"], 3))
        if "draws" in g and "draws" in o:
            line += "  draws - 1 %s" % fmt(g["draws"] / o["draws"] - 1.0)
        print(line, flush=True)

    print("\n==== results (oracle float-sum error: " + ", ".join("%s %s" % (k, fmt(v["ll"] / v["ll64"] - 1.0)) for k, v in sorted(ora.items())) + ")")
    for r in range(a.runs):
        if "C2" in configs:
            compare("C2:%d" % r, "C2:oracle")
            if a.damped:
                compare("C2:%d" % r, "C2:damped_oracle")
        if "C3" in configs:
            compare("C3:%d" % r, "C3:%d:oracle" % r)
            if a.damped and r == 0:
                compare("C3:0", "C3:0:damped_oracle")
        if "C4" in configs:
            compare("C4:e1:%d" % r, "C4:e1:oracle", WEIGHTS)
            compare("C4:e2:%d" % r, "C4:e2:%d:oracle" % r, WEIGHTS)
    if a.damped and "C2" in configs:
        d, p = ora["C2:damped_oracle"], ora["C2:oracle"]
        print("C2: damped / plain sequential oracle - 1: LL %s  |w_i| %+.2f%%" % (fmt(d["ll64"] / p["ll64"] - 1.0),
                                                                                  100.0 * (d["norms"]["w_i"] / p["norms"]["w_i"] - 1.0)))


if __name__ == "__main__":
    main()
