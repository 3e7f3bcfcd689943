"""CPU-only: the plain sequential oracle on config 2 / config 3 (same data, four epochs from the seeded initial weights, the engine's
keyed order and draws, seed 1492): absolute log-likelihoods (double sum), norms and draws per update, the figures
`tools/ab_kernel.py --print-ll --warmup 0 --epochs 4 --rounds 1` is compared with (profiles/r04_notes.md)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from rankfm_amd import synthetic, order
import multiprocessing as mp
cfg = synthetic.CONFIGS["C2"]
U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
w0 = synthetic.init_weights(U, I, F, seed=1492)
by = np.lexsort((pairs[:, 1], pairs[:, 0]))
pairs_csr = np.ascontiguousarray(pairs[by]); sw = np.ones(N, np.float32)
def run(ms):
    E = 4
    w = {k: np.array(v, copy=True) for k, v in w0.items()}
    perms = np.stack([order.epoch_positions(csr.offsets, 1492, e) for e in range(E)]).astype(np.int32)
    out = orc.fit(pairs_csr, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"],
                  0.01, 0.1, 0.1, "constant", 0.25, ms, E, perms=perms, rng_mode=orc.RNG_COUNTER, seed=1492, membership="binary", want_negatives=ms > 1)
    return ms, out["ll64"], {k: float(np.linalg.norm(w[k])) for k in ("v_u", "v_i", "w_i")}, (out["nsamp"].sum(axis=1) / N if ms > 1 else None)
with mp.get_context("fork").Pool(2) as pool:
    for ms, ll, norms, draws in pool.map(run, [1, 50]):
        print("plain oracle max_samples=%d: LL64 per epoch %s  norms %s  draws/N %s" % (ms, " ".join("%.1f" % x for x in ll), norms, draws), flush=True)
