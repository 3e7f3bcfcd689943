"""CPU-only (oracle = test infrastructure): the distribution of WARP draws per row on config 3 after four epochs, and what it costs a
wavefront whose four row groups wait for the slowest of their four rows -- the measurement behind sgd_warp_kernel's per-group state
machine (profiles/r04_notes.md): mean 22.4 draws, 33 % of the rows stop at the first, 36 % run to the cap of 50; batches of four
after the first draw: 5.8 per row on average, 11.6 for the maximum over four rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from rankfm_amd import synthetic, order
cfg = synthetic.CONFIGS["C2"]
U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
w = synthetic.init_weights(U, I, F, seed=1492)
by = np.lexsort((pairs[:, 1], pairs[:, 0]))
pairs_csr = np.ascontiguousarray(pairs[by]); sw = np.ones(N, np.float32)
E = 4
perms = np.stack([order.epoch_positions(csr.offsets, 1492, e) for e in range(E)]).astype(np.int32)
out = orc.fit(pairs_csr, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"],
              0.01, 0.1, 0.1, "constant", 0.25, 50, E, perms=perms, rng_mode=orc.RNG_COUNTER, seed=1492, membership="binary", want_negatives=True)
ns = out["nsamp"][E - 1].astype(np.int64)      # draws per visited position, last epoch
print("mean draws", ns.mean(), "hist", np.bincount(ns, minlength=51)[[1,2,3,4,5,8,12,16,24,32,49,50]] / len(ns))
# batches of 4 after the first draw: batches(n) = 0 if n == 1 else ceil((n - 1) / 4)
b4 = np.where(ns <= 1, 0, (ns - 1 + 3) // 4)
b8 = np.where(ns <= 1, 0, (ns - 1 + 7) // 8)
rng = np.random.default_rng(0)
for name, b in (("batches of 4", b4), ("batches of 8", b8)):
    # four independent rows per wavefront (different users): wave time per row step ~ max over its 4 groups
    idx = rng.permutation(len(b))[: (len(b) // 4) * 4].reshape(-1, 4)
    print(name, "mean per row %.2f   mean of max over 4 rows %.2f   ratio %.2f" % (b.mean(), b[idx].max(axis=1).mean(), b[idx].max(axis=1).mean() / b.mean()))
