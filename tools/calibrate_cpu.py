"""BUILD-CONTAINER measurement: speed of the CPU restatement (oracle/rfm_oracle.c, the `cpu_baseline` of bench.py) relative to
the reference's own compiled Cython `_fit` on identical data, one core.  Needs /root/reference + oracle/_ref (build_ref.sh)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader
from oracle import oracle as orc
from rankfm_amd import synthetic
RankFM, ext, _ = ref_loader.load_reference()
for (U, I, N, F, ms) in ((20000, 10000, 1000000, 64, 1), (6040, 3706, 786000, 20, 1), (20000, 10000, 1000000, 64, 20)):
    pairs, csr = synthetic.make_interactions(U, I, N, seed=0, zipf_s=1.0)
    w = synthetic.init_weights(U, I, F, seed=1)
    sw = np.ones(N, np.float32); x_uf = np.zeros((U, 1), np.float32); x_if = np.zeros((I, 1), np.float32)
    user_items = {u: csr[u] for u in range(U)}
    def ref(epochs):
        g = {k: v.copy() for k, v in w.items()}
        np.random.seed(0); t0 = time.perf_counter()
        ext._fit(pairs, sw, user_items, x_uf, x_if, g["w_i"], g["w_if"], g["v_u"], g["v_i"], g["v_uf"], g["v_if"], 0.01, 0.1, 0.1,
                 "constant", 0.25, ms, epochs, False)
        return time.perf_counter() - t0
    def port(epochs):
        g = {k: v.copy() for k, v in w.items()}
        np.random.seed(0)
        idx = np.arange(N, dtype=np.int32); perms = []
        for _ in range(epochs):
            np.random.shuffle(idx); perms.append(idx.copy())
        t0 = time.perf_counter()
        orc.fit(pairs, sw, csr.offsets, csr.items, x_uf, x_if, g["w_i"], g["w_if"], g["v_u"], g["v_i"], g["v_uf"], g["v_if"], 0.01, 0.1, 0.1,
                "constant", 0.25, ms, epochs, perms=np.stack(perms), rng_mode=orc.RNG_MT19937, seed=1492, membership="linear")
        return time.perf_counter() - t0
    tr = (ref(3) - ref(1)) / 2      # per-epoch time without the reference's O(N) Python set-up
    tp = (port(3) - port(1)) / 2
    print("U=%d I=%d N=%d F=%d max_samples=%d: reference %.3f s/epoch (%.2f M upd/s)  restatement %.3f s/epoch (%.2f M upd/s)  ratio t_port/t_ref = %.2f"
          % (U, I, N, F, ms, tr, N / tr / 1e6, tp, N / tp / 1e6, tp / tr), flush=True)
