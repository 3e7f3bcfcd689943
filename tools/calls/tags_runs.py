"""GPU-box probe: the tags variant of tests/test_gpu_quality.py (config-2 shape, k = 32, 8 + 8 tags, lr 0.05, 10 epochs), several engine
runs per seed -- how much of the engine's hit_rate@10 is run-to-run noise, and what a build changes.   python tags_runs.py <runs> [tune k=v ...]"""
import sys, os, numpy as np, pandas as pd
sys.path.insert(0, os.getcwd())
from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
opts = dict(users=100_000, items=50_000, tags=8, factors=32, lr=0.05, epochs=10, seeds=3)      # name:value arguments override these
tune = {}
for x in sys.argv[2:]:
    if ":" in x:
        k, v = x.split(":"); opts[k] = type(opts[k])(v)
    else:
        k, v = x.split("="); tune[k] = int(v)
for seed in range(opts['seeds']):
    d = synthetic.make_planted_large_device(opts['users'], opts['items'], seed=seed, n_tags=opts['tags'])
    train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
    us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
    uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
    itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
    out = []
    for r in range(runs):
        m = RankFM(factors=opts["factors"], loss="bpr", learning_rate=opts["lr"], engine=EngineOptions(seed=100 + seed, tune=tune))
        np.random.seed(seed)
        m.fit(train, uf, itf, epochs=opts['epochs'])
        g = m.last_fit_report["geometry"]
        out.append((evaluation.hit_rate(m, test, k=10), float(np.linalg.norm(m.w_i)), float(np.linalg.norm(m.v_uf)), float(np.linalg.norm(m.v_if)),
                    float(np.mean(m.last_fit_report["sgd_kernel_ms"])), g.get("table_steps", 0), g.get("table_overlap_us", 0), g.get("table_span_us", [0, 0])))
    print("seed %d: hit_rate@10 %s mean %.4f | |w_i| %s |v_uf| %s |v_if| %s | kernel ms %s table steps %s overlap/span us %s" % (
        seed, [round(o[0], 4) for o in out], np.mean([o[0] for o in out]), [round(o[1], 2) for o in out], [round(o[2], 3) for o in out],
        [round(o[3], 3) for o in out], [round(o[4], 3) for o in out], [o[5] for o in out], [(o[6], o[7]) for o in out][:2]), flush=True)
