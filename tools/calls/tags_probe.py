import sys, os, numpy as np, pandas as pd
sys.path.insert(0, os.getcwd())
from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic
d = synthetic.make_planted_large_device(100_000, 50_000, seed=0, n_tags=8)
train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
for lr in (0.03, 0.05, 0.1):
    for ep in (5, 10, 15):
        try:
            m = RankFM(factors=32, loss="bpr", learning_rate=lr, engine=EngineOptions(seed=100))
            np.random.seed(0)
            m.fit(train, uf, itf, epochs=ep)
            print("lr %.2f epochs %2d: hit_rate@10 %.4f  |w_i| %.2f |v_uf| %.3f |v_if| %.3f" % (lr, ep, evaluation.hit_rate(m, test, k=10), np.linalg.norm(m.w_i), np.linalg.norm(m.v_uf), np.linalg.norm(m.v_if)), flush=True)
        except Exception as e:
            print("lr %.2f epochs %d: %s" % (lr, ep, str(e)[:100]), flush=True)
