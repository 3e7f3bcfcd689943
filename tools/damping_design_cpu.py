"""CPU-only analysis (oracle = test infrastructure): what does the engine's Hogwild step damping change BY ITSELF?  The sequential oracle
with the plan's step scales applied (several rules) against the plain oracle on config 2's data, `max_samples` draws (50 = config 3),
four epochs from the seeded initial weights, in the engine's keyed order and draws.  Result (profiles/r04_notes.md): scaling only the
POSITIVE item's step (rounds 1-3) moves the fixed point of a hot item's bias -- all of config 3's +1.9 % log-likelihood / +8 % |w_i| --
and only its BIAS part matters; scaling an item's step on BOTH sides of the pair leaves 0.01 % / 0.1 %.

    python tools/damping_design_cpu.py [max_samples] [epochs]"""
import os, sys, time
import numpy as np
import multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from rankfm_amd import synthetic, order
orc.build()
cfg = synthetic.CONFIGS["C2"]
U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
w0 = synthetic.init_weights(U, I, F, seed=1492)
by = np.lexsort((pairs[:, 1], pairs[:, 0]))
pairs_csr = np.ascontiguousarray(pairs[by]); sw = np.ones(N, np.float32)
count = np.bincount(pairs[:, 1], minlength=I).astype(np.float64)
in_flight, grid = 16384.0, 256.0
MS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
EPOCHS = int(sys.argv[2]) if len(sys.argv) > 2 else 4

def scales(M, pend_coef=0.5, hot=True):
    n = in_flight * count / N
    if hot:
        g0 = 16384.0
        hot_items = np.argsort(-count)[:64]
        hot_items = hot_items[count[hot_items] * g0 / N >= 16.0]
        period = np.clip(np.floor(count[hot_items] / (grid * 48.0) + 0.5), 1, 64)
        n[hot_items] += pend_coef * grid * period
    return np.where(n > M, M / n, 1.0).astype(np.float32)

def run(spec):
    name, ps, pb = spec[:3]
    ns = spec[3] if len(spec) > 3 else None
    w = {k: np.array(v, copy=True) for k, v in w0.items()}
    perms = np.stack([order.epoch_positions(csr.offsets, 1492, e) for e in range(EPOCHS)]).astype(np.int32)
    t0 = time.time()
    kw = {}
    if ps is not None:
        kw = dict(pos_step=ps, user_step=np.ones(U, np.float32), pos_step_bias=pb, neg_step=ns)
    out = orc.fit(pairs_csr, sw, csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"],
                  0.01, 0.1, 0.1, "constant", 0.25, MS, EPOCHS, perms=perms, rng_mode=orc.RNG_COUNTER, seed=1492, membership="binary", want_negatives=MS > 1, **kw)
    draws = out["nsamp"].sum(axis=1) if out["nsamp"] is not None else None
    return name, out["ll64"], {k: float(np.linalg.norm(w[k])) for k in ("v_u", "v_i", "w_i")}, draws, time.time() - t0

one = np.ones(I, np.float32)
specs = [("plain", None, None),
         ("M128 symmetric (pos and neg)", scales(128), None, scales(128)),
         ("M64 symmetric", scales(64), None, scales(64)),
         ("M32 symmetric", scales(32), None, scales(32)),
         ("M128 sym, no pending term", scales(128, 0.0), None, scales(128, 0.0)),
         ("M128 bias sym only", one, scales(128), None),   # placeholder replaced below
         ("M128 both (round 3)", scales(128), None),
         ("M256 both", scales(256), None)]
_unused = [("plain", None, None),
         ("M128 both (round 3)", scales(128), None),
         ("M256 both", scales(256), None),
         ("M512 both", scales(512), None),
         ("M128 bias only", one, scales(128)),
         ("M128 factors only", scales(128), one),
         ("M1024 factors, M128 bias", scales(1024), scales(128)),
         ("M128, pending coef 0.25", scales(128, 0.25), None)]
with mp.get_context("fork").Pool(8) as pool:
    res = pool.map(run, specs)
base = res[0]
for name, ll, norms, draws, dt in res:
    line = "%-28s LL/plain-1 %s  norms-1 %s" % (name, " ".join("%+.3f%%" % (100 * (a / b - 1)) for a, b in zip(ll, base[1])),
                                               " ".join("%s %+.2f%%" % (k, 100 * (norms[k] / base[2][k] - 1)) for k in norms))
    if draws is not None:
        line += "  draws-1 %s" % " ".join("%+.2f%%" % (100 * (a / b - 1)) for a, b in zip(draws, base[3]))
    print(line, " (%.0f s)" % dt, flush=True)
