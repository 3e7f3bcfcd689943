"""GPU-box experiment: BASELINE config 4 at its own size as EIGHT user shards on one GPU (distributed.emulate_ranks_on_one_device: the real
engine in every shard, merged like the ranks merge) against one session on the whole data -- norms and the correlation of the item biases
after E epochs, for variants of the exchange.  A variant is a ','-separated list of  late  nofeat  syncs=<n|auto>  tables=<mean|one>  bf16.

    python tools/merge_c4_scan.py "blocking;late;late,nofeat;late,syncs=8" [--epochs 2]

Measurement tooling, not product."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from rankfm_amd import synthetic                                  # noqa: E402
from rankfm_amd.distributed import emulate_ranks_on_one_device    # noqa: E402
from rankfm_amd.engine import DeviceSession                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--config", default="C4", help="C4 or C5 (the whole data set is generated on the host: C5 = 500 M rows, ~12 GB)")
    a = ap.parse_args()
    if a.config == "C2W":        # bench.py's weak-scaling workload at 8 GPUs as ONE data set: 8 x config 2's users and rows over its 50 k items
        c = synthetic.CONFIGS["C2"]
        pairs, csr = synthetic.make_interactions(c["n_users"] * a.world, c["n_items"], c["n_interactions"] * a.world, seed=0)
        w = synthetic.init_weights(c["n_users"] * a.world, c["n_items"], c["factors"], seed=1492)
        sh = dict(interactions=pairs, sample_weight=np.ones(len(pairs), np.float32), csr_offsets=csr.offsets, csr_items=csr.items,
                  x_uf=np.zeros((c["n_users"] * a.world, 1), np.float32), x_if=np.zeros((c["n_items"], 1), np.float32), weights=w, config=c)
    else:
        sh = synthetic.make_config_shard(a.config, rank=0, world=1)
    lr = sh["config"].get("learning_rate", 0.1)
    ms = sh["config"]["max_samples"]
    has_feat = bool(sh["config"].get("n_user_features", 0))
    hyper = dict(alpha=0.01, beta=0.1, learning_rate=lr, learning_schedule="constant", learning_exponent=0.25, max_samples=ms)
    single = {}
    first = None
    for variant in a.variants.split(";"):
        opts = [p for p in variant.split(",") if p]
        feat = has_feat and "nofeat" not in opts
        x_uf = sh["x_uf"] if feat else np.zeros((len(sh["x_uf"]), 1), np.float32)
        x_if = sh["x_if"] if feat else np.zeros((len(sh["x_if"]), 1), np.float32)
        w0 = {k: np.array(v, copy=True) for k, v in sh["weights"].items()}
        if not feat:
            F = w0["v_i"].shape[1]
            w0.update(v_uf=np.zeros((1, F), np.float32), v_if=np.zeros((1, F), np.float32), w_if=np.zeros(1, np.float32))
        if feat not in single:
            one = DeviceSession(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], x_uf, x_if,
                                {k: v.copy() for k, v in w0.items()}, max_samples=ms, seed=1492, learning_rate=lr)
            rep = one.run(epochs=a.epochs)
            single[feat] = (one.weights_to_host(), rep)
            del one
            torch.cuda.empty_cache()
        g, rep = single[feat]
        kw = dict(syncs_per_epoch="auto", late="late" in opts, table_merge="mean")
        for o in opts:
            if o.startswith("syncs="):
                kw["syncs_per_epoch"] = o[6:] if o[6:] == "auto" else int(o[6:])
            if o.startswith("tables="):
                kw["table_merge"] = o[7:]
            if o == "bf16":
                kw["exchange_dtype"] = "bf16"
        problem = dict(interactions=sh["interactions"], sample_weight=sh["sample_weight"], csr_offsets=sh["csr_offsets"], csr_items=sh["csr_items"],
                       x_uf=x_uf, x_if=x_if, weights=w0)
        m = emulate_ranks_on_one_device(problem, a.world, hyper, a.epochs, torch.device("cuda", 0), seed=1492,
                                        has_user_features=int(feat), has_item_features=int(feat), **kw)
        ratio = {k: round(float(np.linalg.norm(m[k].astype(np.float64)) / max(np.linalg.norm(g[k].astype(np.float64)), 1e-30)), 4) for k in g}
        print("%-32s norms merged / one GPU %s  corr(w_i) %.4f  (one GPU: LL per update %s)" % (
            variant or "blocking", ratio, float(np.corrcoef(m["w_i"], g["w_i"])[0, 1]), np.round(rep["log_likelihood"] / len(sh["interactions"]), 4)), flush=True)
        if first is None:
            first = m
        else:      # (two runs of the SAME variant differ too: Hogwild is not reproducible -- list a variant twice for the noise floor)
            print("%-32s   |x - first variant's| / |first variant's|: %s" % ("", {k: round(float(np.linalg.norm(m[k].astype(np.float64) - first[k]) / max(np.linalg.norm(first[k].astype(np.float64)), 1e-30)), 4) for k in g}), flush=True)


if __name__ == "__main__":
    main()
