"""CPU-only analysis: 2 / 4 / 8 user shards trained by the sequential oracle from the same epoch-start tables and merged like the
ranks of rankfm_amd/distributed.py do (SharedTables.merge_scale), against sequential training of the whole data, on the
MovieLens-1M-shaped planted problem.
    python tools/merge_emulation.py [epochs] [learning_rate] [zipf]
`zipf`: a larger problem (40,000 x 8,000, 1.8 M rows) whose item frequencies follow the Zipf(1) of BASELINE config 4 instead of the
MovieLens-shaped skew.  Fourth argument: settings "M:MW,..." in units of the clamp rule's default M, or "auto:C:CW[:ranks]" = the
curvature rule of SharedTables.set_merge_curvature with constants C (factors) and CW (biases).  Numbers in profiles/r02_notes.md /
r03_notes.md.
(test / analysis infrastructure: uses oracle/)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle as orc
from rankfm_amd import synthetic
from rankfm_amd._rankfm import UserItemsCSR
from rankfm_amd.distributed import SHARED_NAMES, SharedTables, shard_boundaries, take_user_shard
orc.build()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LR = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
ZIPF = len(sys.argv) > 3 and sys.argv[3] == "zipf"
SYNCS = int(sys.argv[5]) if len(sys.argv) > 5 else 1     # exchanges per epoch (every rank trains every SYNCS-th row of its shard between two of them)
TAGS = 8 if len(sys.argv) > 3 and sys.argv[3] == "tags" else 0     # MovieLens-shaped with 8 + 8 binary tags that carry signal (feature tables: averaged)
if ZIPF:
    U, I, F = 40000, 8000, 20
    d = synthetic.make_planted_large(U, I, seed=0, mean_degree=60.0, pop_weight=1.0)
else:
    U, I, F = 6040, 3706, 20
    d = synthetic.make_planted(seed=0, n_tags=TAGS)
pairs, test = d["train"], d["test"]
X_UF = d["user_tags"].astype(np.float32) if TAGS else np.zeros((U, 1), np.float32)
X_IF = d["item_tags"].astype(np.float32) if TAGS else np.zeros((I, 1), np.float32)
N = len(pairs)
csr = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], U)
w = synthetic.init_weights(U, I, F, seed=3, n_user_features=TAGS or 1, n_item_features=TAGS or 1) if TAGS else synthetic.init_weights(U, I, F, seed=3)
sw = np.ones(N, np.float32)
z_i = X_IF
test_users = np.unique(test[:, 0])
tcsr = UserItemsCSR.from_pairs(test[:, 0], test[:, 1], U)
def hit_rate(v_u, v_i, w_i, k=10, tabs=None):
    hits = 0
    if tabs is not None:        # (w_if, v_uf, v_if): the effective rows of a model with features (rankfm/_rankfm.pyx:48-89)
        v_u = v_u + X_UF @ tabs[1]
        w_i = w_i + X_IF @ tabs[0]
        v_i = v_i + X_IF @ tabs[2]
    for u0 in range(0, len(test_users), 512):
        us = test_users[u0:u0 + 512]
        S = v_u[us] @ v_i.T + w_i
        for r, u in enumerate(us):
            S[r, csr.items[csr.offsets[u]:csr.offsets[u + 1]]] = -np.inf
        top = np.argpartition(-S, k, axis=1)[:, :k]
        for r, u in enumerate(us):
            hits += bool(np.intersect1d(top[r], tcsr.items[tcsr.offsets[u]:tcsr.offsets[u + 1]]).size)
    return hits / len(test_users)
o = {k: v.copy() for k, v in w.items()}
out = orc.fit(pairs, sw, csr.offsets, csr.items, X_UF, z_i, o["w_i"], o["w_if"], o["v_u"], o["v_i"], o["v_uf"], o["v_if"],
              0.01, 0.1, LR, "constant", 0.25, 1, E, perms=None, rng_mode=orc.RNG_COUNTER, seed=1, membership="binary")
ll_seq = out["ll"]
print("sequential: hit_rate@10 %.4f" % hit_rate(o["v_u"], o["v_i"], o["w_i"], tabs=(o["w_if"], o["v_uf"], o["v_if"]) if TAGS else None), "LL/N", np.round(ll_seq / N, 4), flush=True)
counts = np.bincount(pairs[:, 1], minlength=I)
print("item frequency: top item %.2f %% of the rows, top 1 %% of the items %.1f %%" % (100.0 * counts.max() / N, 100.0 * np.sort(counts)[::-1][:max(I // 100, 1)].sum() / N))
RULE = (3.2 / LR, 3.2 / LR / 4.0)          # the committed rule of SharedTables.set_merge_damping at >= 4 ranks
SETTINGS = ((8,) + RULE, (8, 2 * RULE[0], RULE[1]), (8, RULE[0], RULE[0]), (8, 0.5 * RULE[0], 0.5 * RULE[1]))
if len(sys.argv) > 4:                      # "M:MW,M:MW,..." in units of the committed rule's M; "auto:C:CW" = the curvature rule
    SETTINGS = tuple((int(x.split(":")[3]) if x.count(":") >= 3 else 8, x) if x.startswith("auto") else (8, float(x.split(":")[0]) * RULE[0], float(x.split(":")[1]) * RULE[0])
                     for x in sys.argv[4].split(","))
for setting in SETTINGS:
    world = setting[0]
    auto = isinstance(setting[1], str)
    M, MW = (RULE if auto else setting[1:])
    if auto:
        c_v, c_w = float(setting[1].split(":")[1]), float(setting[1].split(":")[2])
    bounds = shard_boundaries(csr.offsets, world)
    shards = [take_user_shard(pairs, sw, csr.offsets, csr.items, X_UF, w["v_u"].copy(), bounds[r], bounds[r + 1]) for r in range(world)]
    ref = SharedTables({k: w[k].copy() for k in SHARED_NAMES}, torch.device("cpu"))
    ref.set_merge_damping(counts, world, damping=M)
    a = ref._starts["w_i"]
    nn = torch.as_tensor(counts.astype(np.float32))
    ref.merge_scale[a:a + ref._sizes["w_i"]] = torch.clamp(MW / torch.clamp(nn, min=1.0), min=1.0 / world, max=1.0)
    ll = np.zeros(E)
    n_rank = np.stack([np.bincount(s["interactions"][:, 1], minlength=I) for s in shards]).astype(np.float64) / SYNCS if auto else None     # [world, I]
    for e in range(E):
      for part in range(SYNCS):
        if auto:
            # curvature rule: an item row that rank r steps n_r times moves by (1 - rho^n_r) of the way to its optimum, rho = 1 - kappa,
            # kappa = eta x curvature ~ eta x c x mean |v_u|^2 (biases: eta x c_w); applied one after the other the ranks' steps
            # would move it (1 - rho^sum n_r): scale of the SUMMED deltas = (1 - rho^N) / sum_r (1 - rho^n_r)   (1 for few steps, 1 / ranks for many)
            mean_vu2 = float(np.mean([np.mean(np.sum(s["v_u"].astype(np.float64) ** 2, axis=1)) for s in shards]))
            def sat(kappa):
                lr_ = np.log1p(-min(kappa, 0.5))
                num = -np.expm1(lr_ * n_rank.sum(axis=0))
                den = (-np.expm1(lr_ * n_rank)).sum(axis=0)
                return np.where(den > 0, num / np.maximum(den, 1e-30), 1.0)
            sv, sb = sat(LR * c_v * mean_vu2), sat(LR * c_w)
            a_ = ref._starts["v_i"]; ref.merge_scale[a_:a_ + ref._sizes["v_i"]] = torch.as_tensor(np.repeat(sv, F).astype(np.float32))
            a_ = ref._starts["w_i"]; ref.merge_scale[a_:a_ + ref._sizes["w_i"]] = torch.as_tensor(sb.astype(np.float32))
            if e in (0, E - 1) and part == 0:
                print("   epoch %d: mean |v_u|^2 %.3f  kappa_v %.4f (M ~ %.0f)  scale of the busiest / median item %.3f / %.3f" % (
                    e, mean_vu2, LR * c_v * mean_vu2, 1.0 / (LR * c_v * mean_vu2), sv[np.argmax(counts)], np.median(sv)), flush=True)
        start = ref.flat.clone(); total = torch.zeros_like(start)
        for k in range(world):
            ref.flat.copy_(start)
            t = {n: ref.views[n].numpy() for n in SHARED_NAMES}
            s = shards[k]
            r = orc.fit(np.ascontiguousarray(s["interactions"][part::SYNCS]), np.ascontiguousarray(s["sample_weight"][part::SYNCS]), s["csr_offsets"], s["csr_items"], s["x_uf"], z_i,
                        t["w_i"], t["w_if"], s["v_u"], t["v_i"], t["v_uf"], t["v_if"], 0.01, 0.1, LR, "constant", 0.25, 1, 1, perms=None,
                        rng_mode=orc.RNG_COUNTER, seed=100 + k, epoch_begin=e, membership="binary")
            ll[e] += r["ll"][0]
            total += ref.flat - start
        ref.flat.copy_(start + ref.merge_scale * total)
    v_u = np.concatenate([s["v_u"] for s in shards])
    v_i, w_i = ref.views["v_i"].numpy(), ref.views["w_i"].numpy()
    nr = [float(np.linalg.norm(a) / np.linalg.norm(b) - 1) for a, b in ((v_u, o["v_u"]), (v_i, o["v_i"]), (w_i, o["w_i"]))]
    print("world %d %s" % (world, setting[1]) if auto else "world %d M %g MW " % (world, M) + str(MW), ": hit_rate@10 %.4f  LL/seq-1 %s  norms-1 %s" % (hit_rate(v_u, v_i, w_i, tabs=tuple(ref.views[n].numpy() for n in ("w_if", "v_uf", "v_if")) if TAGS else None), np.round(ll / ll_seq - 1, 3), np.round(nr, 3)), flush=True)
