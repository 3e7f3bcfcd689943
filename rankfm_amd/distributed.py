"""User-sharded data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-process (no counterpart to cite); the design follows SURVEY.md §8(e):

  * users are split into `world_size` contiguous index ranges balanced by interaction count; a rank owns its users'
    interactions, CSR slice, `v_u` rows and `x_uf` rows exclusively -- no traffic for them, ever;
  * the item-side / shared tables (`v_i, w_i, v_if, w_if, v_uf`) are replicated.  They live in ONE flat fp32
    buffer per rank (the weight tensors are views into it), so the exchange step is a single bucket:
        delta = flat - flat_at_epoch_start;  all_reduce(delta, SUM);  flat = flat_at_epoch_start + scale * delta
    One all-reduce per epoch: 13 MB at BASELINE config 2, 52 MB at config 4, 516 MB at config 5 -- a few ms on
    7 x 153 GB/s xGMI links against epochs of tens to hundreds of ms, so one large collective per epoch is the right
    granularity for point-to-point xGMI.
  * `scale` is the same stale-step damping the single-GPU engine applies to hot rows (DESIGN.md "staleness"): a row
    that received n updates across all ranks during the window gets min(1, M / n) of the summed delta (M = 32: an
    item's bias and factors settle within tens of updates, and two ranks' settled moves must not be added: measured
    +10 % on |w_i| with M = 128 in the 2-shard emulation test), never less than the plain average 1/world.  Rarely-touched rows therefore end the epoch with every rank's updates applied
    (sum of deltas), rows every rank hammered end at the ranks' average (summing K near-converged local moves
    would overshoot K-fold); the dense feature tables, which every row touches, are the ranks' average (SharedTables.table_merge).

Everything here works on CPU tensors with the gloo backend as well, which is how the N > 1 logic is tested
without GPUs (tests/test_distributed_cpu.py).
"""
import numpy as np
import pandas as pd
import torch
import torch.distributed as dist

from ._rankfm import UserItemsCSR

SHARED_NAMES = ("v_i", "w_i", "v_if", "w_if", "v_uf")     # replicated tables, in flat-buffer order


def shard_boundaries(csr_offsets, world_size):
    """user index boundaries [world_size + 1] of contiguous ranges with ~equal interaction counts"""
    off = np.asarray(csr_offsets, dtype=np.int64)
    n_users = len(off) - 1
    total = int(off[-1])
    targets = (np.arange(1, world_size, dtype=np.float64) * total / world_size)
    cuts = np.searchsorted(off, targets, side="left")
    bounds = np.concatenate([[0], np.clip(cuts, 0, n_users), [n_users]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def take_user_shard(interactions, sample_weight, csr_offsets, csr_items, x_uf, v_u, lo, hi):
    """the slice of the problem owned by users [lo, hi): user indexes are rebased to 0"""
    interactions = np.asarray(interactions)
    sel = (interactions[:, 0] >= lo) & (interactions[:, 0] < hi)
    local = interactions[sel].astype(np.int32, copy=True)
    local[:, 0] -= lo
    off = np.asarray(csr_offsets[lo:hi + 1], dtype=np.int64)
    return dict(interactions=np.ascontiguousarray(local), sample_weight=np.ascontiguousarray(np.asarray(sample_weight)[sel]),
                csr_offsets=off - off[0], csr_items=np.ascontiguousarray(csr_items[off[0]:off[-1]]),
                x_uf=np.ascontiguousarray(x_uf[lo:hi]), v_u=np.ascontiguousarray(v_u[lo:hi]), row_mask=sel)


class SharedTables:
    """the replicated tables packed into one flat buffer; `views[name]` are the tensors handed to the engine"""
    kMaxWaitEvents = 4096        # (exposed_exchange_ms: event pairs kept between two readings)
    # How the dense FEATURE tables (v_if, w_if, v_uf) are merged.  They are not sums of small steps: a rank's table trainer REPLACES them
    # within ~170 of its steps by an exponential moving average of recent gradient noise around a slowly moving signal, and most of a single
    # trajectory's NORM is that noise.  Measured at config 2's shape with 8 + 8 tags that carry signal, ten epochs, two data seeds, eight
    # engine shards on one GPU merged like the ranks, against ONE engine on the whole data (hit_rate@10 in points / norms of v_uf, v_if, w_if;
    # tools/merge_tags_scan.py, profiles/r06_notes.md section 8; the one engine itself is ~1 point under the sequential oracle here):
    #   "mean"  (default; rounds 1 - 5 and again from the end of round 6) the average of the ranks' deltas -- the saturated end of the
    #           curvature rule: every row touches these tables.  Robust over the cadence: +1.5 points with the default cadence (8 exchanges per
    #           epoch for 8 epochs, then 1), +2.9 at 8 throughout, +0.1 at 2, -1.7 at 1.  The noise averages out, so the merged tables are
    #           SMALLER than a single trajectory's: -45 / -33 / -75 % at eight ranks (-21 / -17 / -18 % at two; config 4 at its own size:
    #           |v_if| 0.38, |w_if| 0.42 of one GPU's) -- a stated deviation from a one-GPU fit's norms, on the side of the better ranking
    #           (tables frozen at their initial values rank +2.2 points on this problem: what the tables add here is mostly noise).
    # Two rules that keep a single trajectory's norms were built and measured in round 6 and are NOT in the product (commit 84dc127 has them;
    # the emulation below still takes "one"): the tables of ONE rank kept per exchange, in turns (+3 / +15 / -12 %): -4.7 points with the
    # default cadence, -0.1 at two ranks and 8 exchanges per epoch throughout; the ranks taking turns TRAINING them while the others run
    # without a table trainer: -6.9 (default cadence), -0.8 at 8 per epoch throughout, -5.9 at 2, -10.4 at 1, and divergence with the late
    # merge.  The other ranks' rows see such tables move only at the exchanges, which ranks within a point only with eight or more exchanges
    # per epoch for the whole fit; the mean keeps every rank's rows beside live tables.
    table_merge = "mean"
    TABLE_NAMES = ("v_if", "w_if", "v_uf")
    # What travels in a blocking exchange (exchange_fused).  "fp32": the bucket as it is, ONE all-reduce.  "bf16": the tables' DELTAS rounded
    # to bfloat16 (8 bits of mantissa: a relative error of 2^-9 per element and exchange, unbiased -- round to nearest even -- on deltas that
    # are themselves sums of noisy SGD steps; the tables stay fp32, `start + scale x sum` is fp32 arithmetic) in one all-reduce of half the
    # bytes, the tail (curvature terms, |v_u|^2 sums, flags: they must be exact) in a second small fp32 one issued beside it.  Not the default:
    # it changes what eight ranks compute and nothing here has met xGMI; measured in the one-GPU emulation (which rounds each shard's delta and
    # the running sum like a ring does): profiles/r06_notes.md section 9.  The late merge always exchanges fp32.
    exchange_dtype = "fp32"


    def _table_regions(self):
        return [(self._starts[k], self._starts[k] + self._sizes[k]) for k in self.TABLE_NAMES]

    def __init__(self, tables, device):
        shapes = {k: tuple(tables[k].shape) for k in SHARED_NAMES}
        sizes = {k: int(np.prod(shapes[k])) for k in SHARED_NAMES}
        # 64-element (256-byte) alignment of every table inside the bucket keeps 16-byte row loads aligned
        starts, total = {}, 0
        for k in SHARED_NAMES:
            starts[k] = total
            total += (sizes[k] + 63) // 64 * 64
        # ... and, behind the tables, what the ranks have to tell each other at an exchange BESIDES their deltas: per item the two
        # curvature terms of the merge rule (exchange_fused), the rank's sum of |v_u|^2 and user count, and a "my slice failed"
        # flag -- all summed by the SAME all-reduce as the deltas (start = 0 and scale = 1 over the tail: it comes back as the sum)
        n_items = shapes["w_i"][0]
        self._tail_at, self._tail_len = total, (2 * n_items + 3 + 63) // 64 * 64
        total += self._tail_len
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.tail = self.flat[self._tail_at:]
        self.views = {}
        for k in SHARED_NAMES:
            v = self.flat[starts[k]:starts[k] + sizes[k]].view(shapes[k])
            v.copy_(torch.as_tensor(np.asarray(tables[k]), dtype=torch.float32))
            self.views[k] = v
        self.start = self.flat.clone()
        self._starts, self._sizes, self._shapes = starts, sizes, shapes
        self.merge_scale = None          # per-element damping of the summed deltas (None = plain sum)

    def set_merge_damping(self, item_counts_all_ranks, world_size, damping=None, bias_damping=None, learning_rate=0.1):
        """per-element scale of the summed deltas: min(1, M / n_i) clipped below at 1/world_size for item i that all
        ranks together update n_i times per exchange window; 1/world_size (average) for the dense feature tables.

        M is the number of updates after which an item's movement stops being linear in their count (the logistic step
        saturates): summing the replicas' deltas is right below it, averaging them above.  It scales with 1 / learning_rate
        (default 3.2 / learning_rate: 32 at the reference's 0.1).  The item BIASES are the first to overshoot when many
        replicas' deltas are summed (a bias step moves the score by eta itself, a factor step by eta |v_u|^2): from four ranks
        on their M is a quarter of the factors'; two replicas overshoot by at most 2 (the floor of the scale is 1/2) and keep
        one M, which tracks sequential training best there (|w_i| -12 % after three epochs with the smaller one).  Measured on
        planted ranking problems with 8 shards (tests/test_distributed_cpu.py, profiles/r02_notes.md): one M = 32 for both is
        stable but M = 48 already lets the biases diverge (hit_rate@10 0.92 -> 0.21); with the biases at 8 the factors are
        stable up to M = 64 and diverge at 128; at learning_rate 0.03, M = 32 is three times too strong (0.78 after 14 epochs
        against 0.935 sequentially) and M = 107 / 27 converges."""
        if damping is None:
            damping = 3.2 / max(float(learning_rate), 1e-6)
        if bias_damping is None:
            bias_damping = float(damping) / (4.0 if world_size > 2 else 1.0)
        n = torch.as_tensor(np.asarray(item_counts_all_ranks, dtype=np.float64), dtype=torch.float32, device=self.flat.device)
        n = torch.clamp(n, min=1.0)
        scale = torch.full_like(self.flat, 1.0 / world_size)
        F = self._shapes["v_i"][1]
        a = self._starts["v_i"]
        scale[a:a + self._sizes["v_i"]] = torch.clamp(float(damping) / n, min=1.0 / world_size, max=1.0).repeat_interleave(F)
        a = self._starts["w_i"]
        scale[a:a + self._sizes["w_i"]] = torch.clamp(float(bias_damping) / n, min=1.0 / world_size, max=1.0)
        scale[self._tail_at:] = 1.0
        self.merge_scale = scale if world_size > 1 else None
        self._clamp = (np.asarray(item_counts_all_ranks, dtype=np.float64), int(world_size), damping, bias_damping, float(learning_rate))
        self._clamp_window = 1.0

    def set_clamp_window(self, window):
        """the clamp rule's item counts are per EPOCH; an exchange that closes the share `window` of an epoch sees window x counts"""
        c = getattr(self, "_clamp", None)
        if c is None or self.merge_scale is None or abs(float(window) - self._clamp_window) < 1e-12:
            return
        counts, world, damping, bias_damping, lr = c
        self.set_merge_damping(counts * float(window), world, damping, bias_damping, lr)
        self._clamp = c
        self._clamp_window = float(window)

    # The curvature rule (the default from round 3's end on).  An item row that a rank steps n times in an exchange window moves
    # about (1 - rho^n) of the way to where that rank's data would take it, rho = 1 - kappa, kappa = eta x (curvature of the loss
    # along the row) -- for a factor row ~ eta x c x mean |v_u|^2 (the logistic term's second derivative is sigma' <= 1/4 times the
    # squared projection of v_u), for a bias eta x c_w.  The same steps taken one rank after the other would move it
    # (1 - rho^(sum n_r)): the scale of the SUMMED deltas is
    #     s_i = (1 - rho^N_i) / sum_r (1 - rho^n_ri),        N_i = sum_r n_ri
    # -- 1 while the steps are few (the deltas add up), 1 / ranks once every rank has taken many (they all say the same thing) --
    # and it follows the model: |v_u|^2 grows by an order of magnitude over the first epochs, so summing is right early and
    # averaging later, which no constant M of the clamp rule (set_merge_damping) can say.  CPU emulation, 8 shards trained by the
    # sequential oracle (tools/merge_emulation.py, profiles/r03_notes.md section 5): hit_rate@10 against sequential training
    # MovieLens-shaped 0.943 / 0.932 (sequential 0.928 / 0.935 at learning rates 0.1 / 0.03; clamp rule 0.928 / 0.914),
    # Zipf(1) items 0.783 / 0.582 (0.790 / 0.606; clamp rule 0.714 / 0.512).  c = 0.1 for the factors (0.03 ... 0.12 within a
    # point of each other), c_w = 0.3 for the biases (0.25 holds everywhere, 0.12 lets them run away at learning rate 0.1).
    CURVATURE_FACTORS, CURVATURE_BIASES = 0.1, 0.3

    def set_merge_curvature(self, local_item_counts, world_size, learning_rate=0.1, c_factors=None, c_biases=None, group=None, mean_vu2=None, n_users=1):
        """arm the curvature rule: `local_item_counts` [I] = this rank's updates of every item per exchange window.  The scale
        itself is computed at every exchange (it needs the ranks' current mean |v_u|^2): by exchange_fused inside the one all-reduce
        (`mean_vu2` / `n_users` = this rank's mean |v_u|^2 and user count now; the mean over all ranks' users is what everybody starts from), or by refresh_merge_scale + two small
        collectives (the round-3 form, kept for callers that drive the exchange themselves)."""
        self._n_local = torch.as_tensor(np.asarray(local_item_counts, dtype=np.float64), dtype=torch.float64, device=self.flat.device)
        self._curvature = (float(learning_rate), float(self.CURVATURE_FACTORS if c_factors is None else c_factors),
                           float(self.CURVATURE_BIASES if c_biases is None else c_biases))
        self._world = int(world_size)
        self.merge_scale = torch.full_like(self.flat, 1.0 / world_size) if world_size > 1 else None      # (feature tables with table_merge "mean": the average)
        if self.merge_scale is not None:
            self.merge_scale[self._tail_at:] = 1.0                                                       # (the tail comes back as the plain sum)
        self._n_exchanges = 0
        # exchange_fused: per-item totals over all ranks (static: summed once, here) and the mean |v_u|^2 every rank starts from
        self._n_total = self._n_local.clone()
        if dist.is_available() and dist.is_initialized() and world_size > 1:
            dist.all_reduce(self._n_total, op=dist.ReduceOp.SUM, group=group)
        # (the mean over ALL ranks' users -- a sum / count all-reduce: rank 0 may own no users, and a zero here would turn the first
        #  exchange into an unscaled sum of the ranks' deltas; `mean_vu2` = this rank's mean, `n_users` its user count)
        m0 = torch.tensor([(float(mean_vu2) if mean_vu2 is not None else 0.0) * float(n_users), float(n_users)], dtype=torch.float64, device=self.flat.device)
        if dist.is_available() and dist.is_initialized() and world_size > 1:
            dist.all_reduce(m0, op=dist.ReduceOp.SUM, group=group)
        self._mean_vu2 = m0[0] / torch.clamp(m0[1], min=1.0)

    def refresh_merge_scale(self, sum_vu2, n_users, group=None, eta=None, window=1.0):
        """the curvature rule's scale for the coming exchange: `sum_vu2` / `n_users` = this rank's sum of |v_u|^2 and user count
        (two small all-reduces: the mean over all ranks, and the per-item terms); `eta` = the learning rate of the epoch just
        trained when it differs from the one the rule was armed with ('invscaling' schedule)"""
        if getattr(self, "_curvature", None) is None or self.merge_scale is None:
            return
        lr, c_v, c_w = self._curvature
        if eta is not None:
            lr = float(eta)
        dev = self.flat.device
        stat = torch.tensor([float(sum_vu2), float(n_users)], dtype=torch.float64, device=dev)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=group)
        if float(stat[1]) <= 0.0:
            return                       # (nobody reported its users: the scale stays what it was -- the average, initially)
        mean_vu2 = float(stat[0] / stat[1])
        log_rho = curvature_log_rho(lr, c_v, c_w, mean_vu2)
        terms = curvature_terms(self._n_local * window, log_rho)                                    # [3, I]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(terms, op=dist.ReduceOp.SUM, group=group)
        sv, sb = curvature_scales(terms, log_rho, self._world)
        F = self._shapes["v_i"][1]
        a = self._starts["v_i"]
        self.merge_scale[a:a + self._sizes["v_i"]] = sv.to(torch.float32).repeat_interleave(F)
        a = self._starts["w_i"]
        self.merge_scale[a:a + self._sizes["w_i"]] = sb.to(torch.float32)
        self.last_curvature = dict(mean_vu2=mean_vu2, kappa_factors=lr * c_v * mean_vu2, kappa_biases=lr * c_w)

    def window_movement(self, window=1.0, eta=None):
        """how far one exchange window carries an item's bias towards where its updates pull it, by the curvature rule's own model:
        the mean over the touched items of 1 - rho_w^(N_i window), N_i = all ranks' updates of item i per epoch.  The one-window-late
        merge is a delay of one window in that loop, and a delayed loop with a gain near one rings (ShardedTrainer.LATE_MOVEMENT)."""
        if getattr(self, "_curvature", None) is None or getattr(self, "_n_total", None) is None:
            return 0.0
        lr, c_v, c_w = self._curvature
        if eta is not None:
            lr = float(eta)
        n = self._n_total[self._n_total > 0]
        if n.numel() == 0:
            return 0.0
        log_rho_w = float(np.log(max(1e-12, 1.0 - min(lr * c_w, 1.0 - 1e-12))))
        return float((1.0 - torch.exp(log_rho_w * n * float(window))).mean())

    def begin_epoch(self):
        self.tail.zero_()
        self.start.copy_(self.flat)

    def exchange_fused(self, group, sum_vu2, n_users, failed=False, eta=None, window=1.0):
        """The exchange step of the curvature rule as ONE collective and without a host round trip: the ranks' deltas, their
        curvature terms, their |v_u|^2 sums and their failure flags travel in the same all-reduce of the flat bucket.
          * rho of this exchange comes from the mean |v_u|^2 the ranks agreed on at the PREVIOUS exchange (at arming time: rank 0's,
            broadcast with the tables) -- it moves by a few percent per exchange, and every rank must use the same value;
          * the per-item totals N_i = sum_r n_ri are static and were summed once when the rule was armed;
          * everything between the two fused passes over the bucket is device arithmetic on [I]-sized vectors.
        `sum_vu2` may be a device tensor (no .item() anywhere); `window` = the share of the rule's counting period (set_merge_curvature's
        item counts) this exchange closes -- 1 / k when the period is cut into k exchanges.  Returns the (device) sum of the ranks'
        failure flags."""
        lr, c_v, c_w = self._curvature
        if eta is not None:
            lr = float(eta)
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        n_items = self._shapes["w_i"][0]
        t = self.tail
        log_rho_v = torch.log1p(-torch.clamp(lr * c_v * self._mean_vu2, max=0.5))                  # device scalar (float64)
        log_rho_w = float(np.log1p(-min(lr * c_w, 0.5)))
        t[:n_items] = (-torch.expm1(log_rho_v * (self._n_local * window))).to(torch.float32)
        t[n_items:2 * n_items] = (-torch.expm1(log_rho_w * (self._n_local * window))).to(torch.float32)
        t[2 * n_items] = sum_vu2
        t[2 * n_items + 1] = float(n_users)
        t[2 * n_items + 2] = 1.0 if failed else 0.0
        if failed:
            self.flat[:self._tail_at].copy_(self.start[:self._tail_at])      # a failed slice contributes no deltas (they may be non-finite)
        use_kernels = self.flat.is_cuda
        if use_kernels:
            import ctypes as C
            from . import _hip
            stream = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
            n = self.flat.numel()
            with torch.cuda.device(self.flat.device):
                _hip.raise_for_status(_hip.lib().rfm_delta_begin(self.flat.data_ptr(), self.start.data_ptr(), n, stream))
        else:
            self.flat.sub_(self.start)
        self._n_exchanges += 1
        if world > 1 and self.exchange_dtype == "bf16":
            T = self._tail_at
            if getattr(self, "_x16", None) is None:
                self._x16 = torch.empty(T, dtype=torch.bfloat16, device=self.flat.device)
            self._x16.copy_(self.flat[:T])
            work = dist.all_reduce(self._x16, op=dist.ReduceOp.SUM, group=group, async_op=True)
            dist.all_reduce(self.tail, op=dist.ReduceOp.SUM, group=group)
            work.wait()
            self.flat[:T].copy_(self._x16)
        elif world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        t_v, t_w = t[:n_items].to(torch.float64), t[n_items:2 * n_items].to(torch.float64)
        lo = 1.0 / world
        sv = torch.where(t_v > 0, -torch.expm1(log_rho_v * (self._n_total * window)) / torch.clamp(t_v, min=1e-30), torch.ones_like(t_v)).clamp(lo, 1.0)
        sb = torch.where(t_w > 0, -torch.expm1(log_rho_w * (self._n_total * window)) / torch.clamp(t_w, min=1e-30), torch.ones_like(t_w)).clamp(lo, 1.0)
        F = self._shapes["v_i"][1]
        a = self._starts["v_i"]
        self.merge_scale[a:a + self._sizes["v_i"]].view(n_items, F).copy_(sv.to(torch.float32)[:, None].expand(n_items, F))
        a = self._starts["w_i"]
        self.merge_scale[a:a + self._sizes["w_i"]] = sb.to(torch.float32)
        if use_kernels:
            with torch.cuda.device(self.flat.device):
                _hip.raise_for_status(_hip.lib().rfm_delta_finish(self.flat.data_ptr(), self.start.data_ptr(), self.merge_scale.data_ptr(), 1.0, n, stream))
        else:
            self.flat.mul_(self.merge_scale).add_(self.start)
        # what the ranks agreed on, for the next exchange
        users = t[2 * n_items + 1].to(torch.float64)
        self._mean_vu2 = torch.where(users > 0, t[2 * n_items].to(torch.float64) / torch.clamp(users, min=1.0), self._mean_vu2)
        return t[2 * n_items + 2]

    # ---- the exchange OFF the critical path (round 5): one-window-late merge ------------------------------------------------------
    # The blocking rule makes every rank wait for the all-reduce of window k before it trains window k + 1: at BASELINE config 4 an
    # exchange moves 52 MB while a rank's share of an epoch is ~4 ms of SGD, and the default cadence is 8 exchanges per epoch for a
    # fit's first 8 epochs -- as long in RCCL as in SGD.  Late merge: a rank trains window k + 1 on ITS OWN result of window k while the
    # all-reduce of the window-k deltas runs beside it (torch's async collective: RCCL's own stream), and replaces its own window-k
    # delta by the merged one when window k + 1 is done:
    #     after window k (tables T = B_k + d_k, B_k = the snapshot the window started from):
    #         own_new = T - B_k                                   (this rank's delta of window k, tail = its curvature terms)
    #         if a reduction is pending (window k - 1):  wait;  T += scale_{k-1} * SUM_{k-1} - own_{k-1}          (the correction)
    #         own = own_new;  SUM = all_reduce(copy of own), asynchronously
    #     finish_late():  wait;  T += scale * SUM - own.
    # Summed over the windows every rank ends at  B_0 + sum_k scale_k * SUM_k : the replicas are identical again after finish_late(),
    # and equal to the blocking rule with every merged delta applied one window late.  Between exchanges a rank's tables differ from its
    # peers' by its own un-merged window -- by design.  rho of window k's scale comes from the mean |v_u|^2 agreed at the last COMPLETED
    # reduction (two windows back instead of one).
    def exchange_late(self, group, sum_vu2, n_users, failed=False, eta=None, window=1.0):
        """submit this window's deltas, apply the previous window's correction; returns the (device) sum of the failure flags that
        came back with the PREVIOUS window's reduction (0 when none was pending)"""
        lr, c_v, c_w = self._curvature
        if eta is not None:
            lr = float(eta)
        n_items = self._shapes["w_i"][0]
        T = self._tail_at
        if getattr(self, "_late_own", None) is None:
            self._late_own, self._late_sum = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
            self._late_work, self._late_meta = None, None
            self.late_wait_events = []                     # (CUDA) event pairs around the waits for reductions: the EXPOSED exchange time --
            #                                                recorded only while `record_waits` is set (bench.py), at most kMaxWaitEvents of them
        t = self.tail
        log_rho_v = torch.log1p(-torch.clamp(lr * c_v * self._mean_vu2, max=0.5))
        log_rho_w = float(np.log1p(-min(lr * c_w, 0.5)))
        t[:n_items] = (-torch.expm1(log_rho_v * (self._n_local * window))).to(torch.float32)
        t[n_items:2 * n_items] = (-torch.expm1(log_rho_w * (self._n_local * window))).to(torch.float32)
        t[2 * n_items] = sum_vu2
        t[2 * n_items + 1] = float(n_users)
        t[2 * n_items + 2] = 1.0 if failed else 0.0
        if failed:
            self.flat[:T].copy_(self.start[:T])            # a failed slice contributes no deltas (they may be non-finite)
        if getattr(self, "_late_tmp", None) is None:
            self._late_tmp = torch.zeros_like(self.flat)
        torch.sub(self.flat, self.start, out=self._late_tmp)                 # this window's own delta (tail: its curvature terms)
        flag = self._late_apply()                                            # the PREVIOUS window's correction (waits for its reduction)
        self.tail.zero_()                                                    # (the tail belongs to the exchange, not to the tables)
        # a peer failed in the previous window and has LEFT: no further collective may be launched (it would never complete), so the verdict
        # is needed on the host before the next all-reduce is submitted -- a 4-byte read-back that waits for this window's SGD (ADVICE r05:
        # it costs the launch-ahead of short windows; the alternative, a failed rank that stays for one more collective, was not built)
        if float(flag) > 0:
            return flag
        self._late_own, self._late_tmp = self._late_tmp, self._late_own
        self._late_sum.copy_(self._late_own)
        self._late_meta = (log_rho_v, log_rho_w, float(window))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self._late_work = dist.all_reduce(self._late_sum, op=dist.ReduceOp.SUM, group=group, async_op=True)
        else:
            self._late_work = True                         # (one rank: the "reduction" is the copy)
        return flag

    def _late_apply(self):
        """wait for the pending reduction and replace this rank's own delta of that window by the merged one"""
        zero = torch.zeros((), dtype=torch.float32, device=self.flat.device)
        if getattr(self, "_late_work", None) is None:
            return zero
        if self._late_work is not True:
            if self.flat.is_cuda and getattr(self, "record_waits", False) and len(self.late_wait_events) < self.kMaxWaitEvents:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._late_work.wait()
                e1.record()
                self.late_wait_events.append((e0, e1))
            else:
                self._late_work.wait()
        self._late_work = None
        log_rho_v, log_rho_w, window = self._late_meta
        world = self._world
        n_items, T, F = self._shapes["w_i"][0], self._tail_at, self._shapes["v_i"][1]
        ts = self._late_sum[T:]
        t_v, t_w = ts[:n_items].to(torch.float64), ts[n_items:2 * n_items].to(torch.float64)
        lo = 1.0 / world
        sv = torch.where(t_v > 0, -torch.expm1(log_rho_v * (self._n_total * window)) / torch.clamp(t_v, min=1e-30), torch.ones_like(t_v)).clamp(lo, 1.0)
        sb = torch.where(t_w > 0, -torch.expm1(log_rho_w * (self._n_total * window)) / torch.clamp(t_w, min=1e-30), torch.ones_like(t_w)).clamp(lo, 1.0)
        a = self._starts["v_i"]
        self.merge_scale[a:a + self._sizes["v_i"]].view(n_items, F).copy_(sv.to(torch.float32)[:, None].expand(n_items, F))
        a = self._starts["w_i"]
        self.merge_scale[a:a + self._sizes["w_i"]] = sb.to(torch.float32)
        # T += scale * SUM - own   (one fused pass on the tables part of the bucket)
        self.flat[:T].addcmul_(self.merge_scale[:T], self._late_sum[:T]).sub_(self._late_own[:T])
        users = ts[2 * n_items + 1].to(torch.float64)
        self._mean_vu2 = torch.where(users > 0, ts[2 * n_items].to(torch.float64) / torch.clamp(users, min=1.0), self._mean_vu2)
        return ts[2 * n_items + 2].clone()

    def finish_late(self, group=None):
        """the final, blocking exchange of the late merge: afterwards every rank holds the same tables.  Each rank has summed the same
        merged deltas onto the same start, but in its own order (its own deltas first, the corrections later): the replicas agree to
        rounding (1e-7), and one broadcast of rank 0's tables at the END OF THE FIT makes them agree bit for bit, which is what the
        blocking rule guarantees and what the callers of fit_distributed rely on."""
        flag = self._late_apply()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat[:self._tail_at], src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        return flag

    def exposed_exchange_ms(self, reset=True):
        """(CUDA) milliseconds the rank's stream waited for reductions since the last call -- the part of the exchanges that was NOT hidden"""
        ev = getattr(self, "late_wait_events", None) or []
        if ev:
            torch.cuda.synchronize(self.flat.device)
        ms = [a.elapsed_time(b) for a, b in ev]
        if reset and ev:
            self.late_wait_events = []
        return float(sum(ms)), len(ms)

    def all_reduce_deltas(self, group=None, average=False):
        """the exchange step: after it every rank holds epoch_start + sum (or mean) over ranks of its epoch's deltas"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        world = dist.get_world_size(group)
        if self.flat.is_cuda:
            # one fused pass over the bucket on each side of the collective (rfm_delta_begin / rfm_delta_finish) instead of four
            # elementwise passes: the bucket is 52 MB at BASELINE config 4 and 516 MB at config 5
            import ctypes as C
            from . import _hip
            with torch.cuda.device(self.flat.device):
                stream = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
                n = self.flat.numel()
                _hip.raise_for_status(_hip.lib().rfm_delta_begin(self.flat.data_ptr(), self.start.data_ptr(), n, stream))
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                scale = None if (average or self.merge_scale is None) else self.merge_scale.data_ptr()
                _hip.raise_for_status(_hip.lib().rfm_delta_finish(self.flat.data_ptr(), self.start.data_ptr(), scale,
                                                                  1.0 / world if average else 1.0, n, stream))
            return
        # CPU tensors (the gloo tests of the N > 1 logic): the same arithmetic with torch ops
        self.flat.sub_(self.start)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat.div_(world)
        elif self.merge_scale is not None:
            self.flat.mul_(self.merge_scale)
        self.flat.add_(self.start)

    @property
    def payload_bytes(self):
        """bytes a rank hands to the collective(s) of one blocking exchange"""
        if self.exchange_dtype == "bf16":
            return self._tail_at * 2 + self._tail_len * 4
        return self.flat.numel() * 4


def curvature_log_rho(learning_rate, c_factors, c_biases, mean_vu2):
    """log rho of the curvature rule (SharedTables.set_merge_curvature) for the factor rows and for the biases"""
    return (float(np.log1p(-min(learning_rate * c_factors * mean_vu2, 0.5))), float(np.log1p(-min(learning_rate * c_biases, 0.5))))


def curvature_terms(n_local, log_rho):
    """what ONE rank contributes to the curvature rule: [1 - rho_v^n | 1 - rho_w^n | n] per item (float64 [3, I]); the ranks' terms
    are summed (one all-reduce) before curvature_scales"""
    n = torch.as_tensor(n_local, dtype=torch.float64)
    return torch.stack([-torch.expm1(log_rho[0] * n), -torch.expm1(log_rho[1] * n), n])


def curvature_scales(terms_sum, log_rho, world_size):
    """(scale of the summed factor-row deltas, scale of the summed bias deltas) per item:  (1 - rho^N) / sum_r (1 - rho^n_r),
    clipped to [1 / ranks, 1]; 1 for an item nobody stepped"""
    N, lo = terms_sum[2], 1.0 / world_size
    out = []
    for k in (0, 1):
        s = torch.where(terms_sum[k] > 0, -torch.expm1(log_rho[k] * N) / torch.clamp(terms_sum[k], min=1e-300), torch.ones_like(N))
        out.append(s.clamp(lo, 1.0))
    return out[0], out[1]


def broadcast_from_rank0(tensors, group=None):
    """make the replicated tables identical on every rank before training"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=0, group=group)


class ShardedTrainer:
    """epoch loop of one rank: local SGD epoch on the rank's user shard, then the item-table all-reduce.

    `epoch_fn(shared_views, epoch)` runs one local epoch in place on the rank's tensors; on the GPU it is
    `DeviceSession.run` (see make_device_trainer), in the CPU tests it is any stand-in with the same contract.
    """

    def __init__(self, shared, epoch_fn, group=None, average=False, syncs_per_epoch=1, user_norms_fn=None, eta_fn=None, overlap=False):
        self.shared, self.epoch_fn, self.group, self.average = shared, epoch_fn, group, average
        # overlap: the one-window-late merge (SharedTables.exchange_late) -- the all-reduce of a window's deltas runs beside the next
        # window's SGD; needs the fused curvature rule, and finish() before the tables are read.  True / False / "auto" (decided after
        # the first epoch from what an exchange and an epoch's SGD cost: _decide_overlap)
        self.overlap = overlap if overlap == "auto" else bool(overlap)
        # exchanges per epoch: a number, or "auto" (the default of fit_distributed / bench.py) = AUTO_EXCHANGES per epoch during a fit's
        # first AUTO_EPOCHS epochs, one per epoch afterwards (exchanges_in_epoch).  Measured with the REAL engine in every shard
        # (tools/merge_engine_scan.py: eight shards of a config-2-shaped planted problem on one GPU, profiles/r04_notes.md): with one
        # exchange per epoch the merged model starts slowly -- hit_rate@10 -13.9 points after 5 epochs against one GPU on the whole
        # data (2 / 4 / 8 exchanges per epoch: -7.4 / -2.9 / +0.2; the schedule 8, 4, 2, 1, 1: -5.0) -- and overtakes it later (+2.2
        # after 15 epochs, +4.0 after 40: the merge is a regulariser).  Shards that train a whole epoch blind to each other while
        # the model moves fastest drift apart (a factor model may turn its latent space freely); once it moves slowly, one exchange
        # per epoch is enough.  Every extra exchange is one more all-reduce of the bucket (13 MB at config 2, 52 MB at config 4).
        self.syncs_per_epoch = syncs_per_epoch
        self.eta_fn = eta_fn             # epoch -> learning rate of that epoch (None: the constant the merge rule was armed with)
        # () -> (sum over this rank's users of |v_u|^2, number of users): what the curvature rule of the merge needs before
        # every exchange (SharedTables.refresh_merge_scale); None = the rank has no users
        self.user_norms_fn = user_norms_fn

    @property
    def fused(self):
        """the curvature rule with everything in one all-reduce (SharedTables.exchange_fused): the default of a multi-rank job"""
        return (getattr(self.shared, "_curvature", None) is not None and self.shared.merge_scale is not None and not self.average
                and getattr(self.shared, "_n_total", None) is not None)

    AUTO_EXCHANGES, AUTO_EPOCHS = 8, 8
    # The late merge hears from the peers one window later, so it needs shorter windows for the same model: measured with the real engine
    # in eight shards of a config-2-shaped problem (profiles/r05_notes.md; hit_rate@10 against the sequential oracle after 5 epochs):
    # blocking, 8 windows per epoch -0.36 point; late, 8 / 12 / 16 / 24 windows -3.2 / -1.6 / -0.87 / -0.12.  Three times the cadence.
    LATE_FACTOR = 3
    # ... and it is a feedback loop with a delay of one window: a rank keeps pushing an item for a whole window on a state that lacks what
    # its peers pushed in the window before.  Where one window already carries an item most of the way (SharedTables.window_movement
    # near 1: many updates per item per window) the loop's gain is near one and the delayed loop rings.  Measured at the configurations'
    # own sizes, eight engine shards on one GPU against one GPU on the whole data (tools/merge_c4_scan.py, profiles/r06_raw/r06k, r06l):
    # config 5 (1 M items, 500 M rows, 24 windows / epoch, 2 epochs; movement 0.47): late |w_i| x2.20, |v_u| x3.66, |v_i| x1.57 -- blocking 0.999 /
    # 1.004 / 1.005; eight times config 2 (the weak-scaling bench load, 3 epochs; movement 0.60): late |w_i| +7.4 %, |v_i| +4.2 % and growing --
    # blocking +0.5 / +1.7 %.  The config-2-shaped planted problem the cadence above was measured on has a movement of 0.11 per late
    # window and is stable.  "auto" therefore takes the late merge only when it is faster AND the movement per late window is small;
    # overlap=True forces it (the caller's risk: check the norms against a blocking run).
    LATE_MOVEMENT = 0.3

    @property
    def late(self):
        """is the one-window-late merge in use right now?  overlap True: always (with the fused curvature rule); "auto": after the first
        epoch has measured what an exchange and an epoch's SGD cost on this job -- see _decide_overlap"""
        return self.fused and (self.overlap is True or (self.overlap == "auto" and getattr(self, "_late_on", False)))

    def exchanges_in_epoch(self, epoch):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return 1                              # (one rank: nothing to exchange, the epoch is one launch)
        k = self.LATE_FACTOR if self.late else 1
        if self.syncs_per_epoch == "auto":
            return k * (self.AUTO_EXCHANGES if int(epoch) < self.AUTO_EPOCHS else 1)
        return max(1, int(self.syncs_per_epoch))          # (an explicit number is taken as it is, in either mode)

    def _decide_overlap(self, sgd_ms, exchange_ms, n_blocking):
        """overlap == "auto", after the first (blocking) epoch: the late merge hides its exchanges behind the SGD but needs LATE_FACTOR
        times as many; with T = an epoch's SGD time, x = one exchange, n = the blocking cadence:
            blocking  T + n x          late  max(T, LATE_FACTOR n x)
        -- late wins when the exchanges fit behind the SGD, or still when (LATE_FACTOR - 1) n x < T.  A rank's share of config 5 (WARP,
        k = 128: T = 246 ms, x ~ 3 ms over xGMI) overlaps; config 4's (T = 3.9 ms, x ~ 0.3 ms for 52 MB) does not.  Every rank must
        decide alike: the maximum of the ranks' measurements is used."""
        n = max(1, int(n_blocking))
        # (a model with trained FEATURE TABLES never takes the late merge: every row touches them, one window replaces them -- a movement of
        #  one -- and a late correction of them destabilised config 4 at its own size, every norm x 10^3 after two epochs: DESIGN.md section 8)
        move = self.shared.window_movement(1.0 / (self.LATE_FACTOR * n)) if hasattr(self.shared, "window_movement") else 0.0
        if getattr(self.shared, "has_feature_tables", False):
            move = 1.0
        m = torch.tensor([float(sgd_ms), float(exchange_ms), float(move)], dtype=torch.float64, device=self.shared.flat.device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        T, x, move = float(m[0]), float(m[1]), float(m[2])
        faster = max(T, self.LATE_FACTOR * n * x) < T + n * x
        self._late_on = bool(faster and move <= self.LATE_MOVEMENT)
        self.overlap_decision = dict(sgd_ms_per_epoch=T, exchange_ms=x, blocking_windows=n, faster=bool(faster), window_movement=move,
                                     late=self._late_on)

    def _exchange(self, epoch=None, err=None, window=1.0):
        if self.fused:
            s, n = self.user_norms_fn() if self.user_norms_fn is not None else (0.0, 0)
            eta = self.eta_fn(epoch) if (self.eta_fn is not None and epoch is not None) else None
            if self.late:
                flag = self.shared.exchange_late(self.group, s, n, failed=err is not None, eta=eta, window=window)
                if err is not None:
                    raise err                   # (its zero deltas and its flag are on their way: the peers' collective completes)
                if float(flag) > 0:             # (read inside exchange_late already: no collective was launched behind it)
                    raise RuntimeError("another rank's local epoch failed; stopping on every rank")
                return
            timing = self.overlap == "auto" and not hasattr(self, "_late_on")
            if timing:                          # (the first epoch of an "auto" job measures what one exchange costs)
                import time
                if self.shared.flat.is_cuda:
                    torch.cuda.synchronize(self.shared.flat.device)
                t0 = time.perf_counter()
            flag = self.shared.exchange_fused(self.group, s, n, failed=err is not None, eta=eta, window=window)
            if timing:
                if self.shared.flat.is_cuda:
                    torch.cuda.synchronize(self.shared.flat.device)
                self._exchange_ms = getattr(self, "_exchange_ms", []) + [(time.perf_counter() - t0) * 1e3]
            if err is not None:
                raise err                       # (after the collective: the peers are not left waiting in it)
            # a PEER's failure: on CPU tensors the flag is read at once; on the GPU it is copied to pinned memory behind the exchange
            # and looked at after the next local slice, whose engine call synchronises the stream anyway (check_peers) -- no host
            # round trip is added to the exchange
            if flag.is_cuda:
                if getattr(self, "_peer_flag", None) is None:
                    self._peer_flag = torch.zeros(1, dtype=torch.float32).pin_memory()
                self._peer_flag.copy_(flag.reshape(1), non_blocking=True)
                self._peer_pending = True
            elif float(flag) > 0:
                raise RuntimeError("another rank's local epoch failed; stopping on every rank")
            return
        if getattr(self.shared, "_curvature", None) is not None and not self.average:
            s, n = self.user_norms_fn() if self.user_norms_fn is not None else (0.0, 0)
            eta = self.eta_fn(epoch) if (self.eta_fn is not None and epoch is not None) else None
            self.shared.refresh_merge_scale(float(s), n, self.group, eta=eta, window=window)
        elif not self.average and hasattr(self.shared, "set_clamp_window"):
            self.shared.set_clamp_window(window)
        self.shared.all_reduce_deltas(self.group, self.average)

    def finish(self):
        """end of the fit: with the late merge, the final blocking exchange (the replicas are identical afterwards); a peer's failure
        flagged in the last exchange is raised here"""
        if self.fused and self.overlap and getattr(self.shared, "_late_own", None) is not None:
            flag = self.shared.finish_late(self.group)
            if flag.is_cuda:
                torch.cuda.synchronize(flag.device)
            if float(flag) > 0:
                raise RuntimeError("another rank's local epoch failed; stopping on every rank")
        self.check_peers(synchronize=True)

    def check_peers(self, synchronize=False):
        """raise if the last fused exchange carried a peer's failure flag (GPU path; `synchronize` for the call after the LAST exchange)"""
        if getattr(self, "_peer_pending", False):
            if synchronize:
                torch.cuda.synchronize(self.shared.flat.device)
            self._peer_pending = False
            if float(self._peer_flag[0]) > 0:
                raise RuntimeError("another rank's local epoch failed; stopping on every rank")

    def _local(self, epoch, **kw):
        """one local slice.  A rank whose slice failed (saturated user, non-finite weights, a HIP error) must not leave its peers
        waiting in the all-reduce: with the fused exchange it still joins the collective -- with zero deltas and its failure flag
        raised -- and raises afterwards (_exchange); otherwise the ranks agree on the outcome BEFORE the collective."""
        err, out = None, None
        try:
            out = self.epoch_fn(self.shared.views, epoch, **kw)
        except Exception as e:      # noqa: BLE001 -- re-raised, on every rank
            err = e
        if self.fused:
            if err is None:
                self.check_peers()              # (the engine call above has synchronised the stream)
            return out, err
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            flag = torch.tensor([0.0 if err is None else 1.0], device=self.shared.flat.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if err is None and flag.item() > 0:
                err = RuntimeError("another rank's local epoch failed; stopping on every rank")
        if err is not None:
            raise err
        return out, None

    def run_epoch(self, epoch):
        out = self._run_epoch(epoch)
        if self.overlap == "auto" and self.fused and not hasattr(self, "_late_on") and getattr(self, "_exchange_ms", None):
            ms = out.get("sgd_kernel_ms") if isinstance(out, dict) else None
            self._decide_overlap(float(np.sum(ms)) if ms is not None else 0.0, float(np.median(self._exchange_ms)), len(self._exchange_ms))
        return out

    def _run_epoch(self, epoch):
        # one epoch = `syncs_per_epoch` slices of the visiting order, each followed by the delta exchange
        n_x = self.exchanges_in_epoch(epoch)
        if n_x <= 1:
            self.shared.begin_epoch()
            out, err = self._local(epoch)
            self._exchange(epoch, err)
            return out
        total = None
        for k in range(n_x):
            self.shared.begin_epoch()
            out, err = self._local(epoch, part=(k, n_x))
            self._exchange(epoch, err, window=1.0 / n_x)
            if total is None:
                total = {key: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for key, v in out.items()}
            else:
                for key in ("log_likelihood", "ll", "sgd_kernel_ms", "n_draws"):
                    if key in out and out[key] is not None:
                        total[key] = total[key] + out[key]
        return total


def agree_on_merge_damping(shared, shard, group=None, merge_damping=None, syncs_per_epoch=1, learning_rate=0.1):
    """arm the damped merge (a no-op without a process group / on one rank).  merge_damping None: the curvature rule
    (SharedTables.set_merge_curvature: every rank keeps its OWN item histogram per exchange window, the scale is refreshed
    before every exchange by ShardedTrainer).  A number M: the clamp rule min(1, M / n_i) of rounds 1-3, which needs every item's
    update count over ALL ranks: one all-reduce of the ranks' item histograms, then SharedTables.set_merge_damping."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    device = shared.flat.device
    counts = torch.bincount(torch.as_tensor(np.asarray(shard["interactions"])[:, 1].astype(np.int64)),
                            minlength=shared.views["w_i"].shape[0]).to(device=device, dtype=torch.float32)
    if merge_damping is None:
        v_u = np.asarray(shard.get("v_u", np.zeros((0, 1), np.float32)))
        mean_vu2 = float((v_u.astype(np.float64) ** 2).sum() / max(len(v_u), 1))
        # (counts per EPOCH: the share of an epoch an exchange closes is passed at the exchange, ShardedTrainer._exchange)
        shared.set_merge_curvature(counts.cpu().numpy(), dist.get_world_size(group), learning_rate=learning_rate, group=group, mean_vu2=mean_vu2,
                                   n_users=len(v_u))
        return
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    # (counts per EPOCH: the share of an epoch an exchange closes is applied at the exchange, SharedTables.set_clamp_window -- with the
    #  "auto" cadence it changes during the fit)
    shared.set_merge_damping(counts.cpu().numpy(), dist.get_world_size(group), merge_damping, learning_rate=learning_rate)


def make_device_trainer(shard, shared_tables, x_if, hyper, device, group=None, average=False, merge_damping=None,
                        syncs_per_epoch=1, overlap=False, exchange_dtype="fp32", **session_kw):
    """wire a rank's shard to the HIP engine: weights are views into the flat bucket, so the engine's in-place
    atomics and the all-reduce act on the same memory"""
    from .engine import DeviceSession
    shared = SharedTables(shared_tables, device)
    assert exchange_dtype in ("fp32", "bf16"), exchange_dtype
    shared.exchange_dtype = exchange_dtype
    agree_on_merge_damping(shared, shard, group, merge_damping, syncs_per_epoch, hyper.get("learning_rate", 0.1))
    if len(shard["csr_offsets"]) <= 1 or len(shard["interactions"]) == 0:
        # a rank without users (more ranks than users, or a few heavy users): it trains nothing but still joins every collective
        def idle_epoch(_views, epoch, part=None):
            return dict(status=0, log_likelihood=np.zeros(1), reg_penalty=np.zeros(1), sgd_kernel_ms=np.zeros(1, np.float32),
                        n_draws=np.zeros(1, np.int64), epochs_done=1, launches_per_epoch=0, waves_per_launch=0)
        return ShardedTrainer(shared, idle_epoch, group=group, average=average, syncs_per_epoch=syncs_per_epoch, overlap=overlap), None
    weights = dict(shared.views)
    weights["v_u"] = torch.as_tensor(shard["v_u"]).to(device)
    sess = DeviceSession(shard["interactions"], shard["sample_weight"], shard["csr_offsets"], shard["csr_items"],
                         shard["x_uf"], x_if, weights, device=device, **hyper, **session_kw)
    # DeviceSession.up() keeps tensors that are already resident float32 contiguous -> still the bucket views
    for k in SHARED_NAMES:
        assert sess.weights[k].data_ptr() == shared.views[k].data_ptr(), "shared table was copied out of the bucket"
    shared.has_feature_tables = bool(sess.has_uf or sess.has_if)       # (ShardedTrainer._decide_overlap: no late merge for such a model)

    def epoch_fn(_views, epoch, part=None):
        return sess.run(epochs=1, epoch_begin=epoch, part=part)

    def user_norms():           # (a device scalar: the fused exchange never reads it on the host)
        v = sess.weights["v_u"]
        return torch.linalg.vector_norm(v, dtype=torch.float64) ** 2, int(v.shape[0])

    def eta_of(epoch):          # rankfm/_rankfm.pyx:220-223
        lr = float(hyper.get("learning_rate", 0.1))
        if hyper.get("learning_schedule", "constant") == "invscaling":
            return lr / float(epoch + 1) ** float(hyper.get("learning_exponent", 0.25))
        return lr

    return ShardedTrainer(shared, epoch_fn, group=group, average=average, syncs_per_epoch=syncs_per_epoch, user_norms_fn=user_norms,
                          eta_fn=eta_of, overlap=overlap), sess


def emulate_ranks_on_one_device(problem, world, hyper, epochs, device, syncs_per_epoch=1, seed=1492, c_factors=None, c_biases=None, late=False,
                                table_merge="mean", exchange_dtype="fp32", **session_kw):
    """What `world` ranks would compute, on ONE GPU and in one process: `world` user shards, each trained by the REAL engine (its own
    DeviceSession, its own copy of the item-side tables, the concurrency plan a rank of that size gets), merged after every exchange
    window exactly like ShardedTrainer / SharedTables.exchange_fused merge the ranks -- curvature rule, rho from the mean |v_u|^2 of
    the previous exchange, per-item totals -- only with the all-reduce replaced by a loop.  A development and test tool (no 8-GPU node
    is needed to see what the merge rule does to the model when the shards are trained asynchronously); the shards run one after the
    other, so it says nothing about time.
    `late`: the one-window-late merge (SharedTables.exchange_late) -- every shard trains window k + 1 on its OWN result of window k and
    has its window-k delta replaced by the merged one afterwards; a final exchange makes the shards identical.
    problem: dict(interactions, sample_weight, csr_offsets, csr_items, x_uf, x_if, weights); returns the merged weights (numpy)."""
    U = len(problem["csr_offsets"]) - 1
    w = problem["weights"]
    bounds = shard_boundaries(problem["csr_offsets"], world)
    trainers, sessions, counts = [], [], []
    for r in range(world):
        sh = take_user_shard(problem["interactions"], problem["sample_weight"], problem["csr_offsets"], problem["csr_items"], problem["x_uf"],
                             w["v_u"], int(bounds[r]), int(bounds[r + 1]))
        t, sess = make_device_trainer(sh, {k: w[k] for k in SHARED_NAMES}, problem["x_if"], hyper, device, seed=seed + r, **session_kw)
        trainers.append(t)
        sessions.append(sess)
        counts.append(torch.bincount(torch.as_tensor(sh["interactions"][:, 1].astype(np.int64)), minlength=len(w["w_i"])).to(device=device, dtype=torch.float64))
    ref = trainers[0].shared
    T, n_items, F = ref._tail_at, ref._shapes["w_i"][0], ref._shapes["v_i"][1]
    lr = float(hyper.get("learning_rate", 0.1))
    c_v = SharedTables.CURVATURE_FACTORS if c_factors is None else c_factors
    c_w = SharedTables.CURVATURE_BIASES if c_biases is None else c_biases
    n_total = sum(counts)
    v0 = sessions[0].weights["v_u"]
    mean_vu2 = (torch.linalg.vector_norm(v0, dtype=torch.float64) ** 2 / max(v0.shape[0], 1))
    master = ref.flat[:T].clone()
    scale = torch.full((T,), 1.0 / world, dtype=torch.float32, device=device)
    pending = None                # (late) the window whose "reduction" is in flight: (scale, total, own deltas, mean |v_u|^2 it reports)
    n_windows_done = 0

    def apply_pending():
        nonlocal mean_vu2
        p_scale, p_total, p_own, p_mean = pending
        for r in range(world):
            trainers[r].shared.flat[:T].add_(p_scale * p_total - p_own[r])
        mean_vu2 = p_mean

    for e in range(epochs):
        n_x = ((ShardedTrainer.LATE_FACTOR if late else 1) * (ShardedTrainer.AUTO_EXCHANGES if e < ShardedTrainer.AUTO_EPOCHS else 1)) if syncs_per_epoch == "auto" else max(int(syncs_per_epoch), 1)
        for k in range(n_x):
            log_rho_v = torch.log1p(-torch.clamp(lr * c_v * mean_vu2, max=0.5))
            log_rho_w = float(np.log1p(-min(lr * c_w, 0.5)))
            total = torch.zeros_like(master)
            t_v = torch.zeros(n_items, dtype=torch.float64, device=device)
            t_w = torch.zeros_like(t_v)
            sum_vu2, users = 0.0, 0
            own = []
            for r in range(world):
                tr, sess = trainers[r], sessions[r]
                if not late:
                    tr.shared.flat[:T].copy_(master)
                before = tr.shared.flat[:T].clone()
                if sess is not None:
                    sess.run(epochs=1, epoch_begin=e, part=(k, n_x) if n_x > 1 else None)
                    sum_vu2 = sum_vu2 + torch.linalg.vector_norm(sess.weights["v_u"], dtype=torch.float64) ** 2
                    users += int(sess.weights["v_u"].shape[0])
                d = tr.shared.flat[:T] - before
                if table_merge == "one":          # (an experiment of this tool only -- the product merges the tables as the mean: the feature tables
                    for name in ("v_if", "w_if", "v_uf"):     #  of ONE shard kept per window, in turns; late: every shard keeps its own until the end)
                        a0 = ref._starts[name]
                        if late or r != n_windows_done % world:
                            d[a0:a0 + ref._sizes[name]] = 0.0
                if exchange_dtype == "bf16" and not late:      # (SharedTables.exchange_dtype: each delta and the running sum rounded like a ring does)
                    total = (total + d.to(torch.bfloat16).to(torch.float32)).to(torch.bfloat16).to(torch.float32)
                else:
                    total += d
                if late:
                    own.append(d)
                t_v += (-torch.expm1(log_rho_v * (counts[r] / n_x))).to(torch.float32).to(torch.float64)
                t_w += (-torch.expm1(log_rho_w * (counts[r] / n_x))).to(torch.float32).to(torch.float64)
            lo = 1.0 / world
            sv = torch.where(t_v > 0, -torch.expm1(log_rho_v * (n_total / n_x)) / torch.clamp(t_v, min=1e-30), torch.ones_like(t_v)).clamp(lo, 1.0)
            sb = torch.where(t_w > 0, -torch.expm1(log_rho_w * (n_total / n_x)) / torch.clamp(t_w, min=1e-30), torch.ones_like(t_w)).clamp(lo, 1.0)
            a = ref._starts["v_i"]
            scale[a:a + ref._sizes["v_i"]].view(n_items, F).copy_(sv.to(torch.float32)[:, None].expand(n_items, F))
            a = ref._starts["w_i"]
            scale[a:a + ref._sizes["w_i"]] = sb.to(torch.float32)
            if table_merge == "one":
                for name in ("v_if", "w_if", "v_uf"):
                    a = ref._starts[name]
                    scale[a:a + ref._sizes[name]] = 1.0
            n_windows_done += 1
            if late:
                if pending is not None:
                    apply_pending()
                pending = (scale.clone(), total, own, sum_vu2 / max(users, 1))
            else:
                master = master + scale * total
                mean_vu2 = sum_vu2 / max(users, 1)
    if late and pending is not None:
        apply_pending()
        master = trainers[0].shared.flat[:T].clone()
    out = {}
    ref.flat[:T].copy_(master)
    for name in SHARED_NAMES:
        out[name] = ref.views[name].detach().cpu().numpy().copy()
    v_u = np.array(w["v_u"], dtype=np.float32, copy=True)
    for r in range(world):
        if sessions[r] is not None:
            v_u[int(bounds[r]):int(bounds[r + 1])] = sessions[r].weights["v_u"].detach().cpu().numpy()
    out["v_u"] = v_u
    return out


def fit_distributed(model, interactions, user_features=None, item_features=None, sample_weight=None, epochs=1, verbose=False,
                    group=None, device=None, merge_damping=None, syncs_per_epoch="auto", make_trainer=None, overlap="auto",
                    exchange_dtype="fp32"):
    """`RankFM.fit` across the ranks of a torch.distributed job (one process per GPU, `torchrun`): every rank calls it with
    the SAME arguments and the same numpy seed.

    Every rank maps the identifiers of ITS slice of the rows (the maps themselves are agreed on first, the mapped pairs are
    all-gathered), builds the sorted item lists of ITS users only, trains its user shard on its GPU and exchanges the item-side
    deltas once per epoch (ShardedTrainer).  At the end the user factors and the item lists are all-gathered, so every rank returns the
    complete fitted model with the reference's attribute layout.  With world size 1 this is `model.fit(...)` on the resident-session path.

    `merge_damping`: None = the curvature rule for scaling the summed deltas (SharedTables.set_merge_curvature: follows the model's
    mean |v_u|^2, no constant to choose); a number M = the clamp rule min(1, M / n_i) of earlier rounds.  `syncs_per_epoch`: exchanges
    of the item-side deltas per epoch -- "auto" (default): eight per epoch during the first eight epochs, when the model moves fastest
    and shards that do not hear from each other drift apart, one per epoch afterwards (ShardedTrainer); a number: that many, always.

    `overlap`: True = the one-window-late merge -- the all-reduce of a window's deltas runs beside the next window's SGD and is applied
    one window late, a final blocking exchange makes the replicas identical (SharedTables.exchange_late); it hears from the peers a
    window later and therefore runs three times the cadence (ShardedTrainer.LATE_FACTOR, measured).  False = every exchange blocks
    (rounds 2-4).  "auto" (default): the first epoch blocks and measures; the late merge is taken from the second epoch on when it
    is the faster of the two on this job AND one of its windows moves an item only a little (ShardedTrainer._decide_overlap,
    LATE_MOVEMENT: at configs 2 - 5's own sizes the late merge rings -- measured round 6, DESIGN.md section 8.1 -- and "auto" keeps
    the blocking merge there; True forces it at the caller's risk).  Only the curvature rule overlaps.

    `exchange_dtype`: "fp32" (default) or "bf16" = the tables' deltas travel rounded to bfloat16, half the bytes (SharedTables.exchange_dtype;
    blocking exchanges of the curvature rule only; free in the one-GPU emulation, never run over xGMI).

    `make_trainer(shard, shared_tables, x_if, hyper, device, group)` -> (ShardedTrainer, finish) replaces the HIP engine in the
    CPU tests; `finish()` must return the shard's trained v_u as a numpy array.
    """
    assert isinstance(epochs, int) and epochs >= 1, "[epochs] must be a positive integer"
    assert isinstance(verbose, bool), "[verbose] must be a boolean value"
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    model._reset_state()
    # Rank-local front end.  Identifier hashing is the expensive host step (one hash lookup per row and column): every rank maps only
    # ITS slice of the rows -- rows [r N / W, (r + 1) N / W) -- after the ranks have agreed on the identifier <-> index maps (union
    # of the slices' unique identifiers: U + I values exchanged, not N), and the mapped int32 pairs are all-gathered so that every
    # rank ends with the reference's complete `interactions` attribute in the original row order.  The per-user sorted item lists
    # -- the one O(N log N) step -- are built by every rank for ITS users only and exchanged at the end.  The weights are drawn in
    # full on every rank (numpy's stream must advance identically).
    assert isinstance(interactions, (np.ndarray, pd.DataFrame)), "[interactions] must be np.ndarray or pd.dataframe"
    assert interactions.shape[1] == 2, "[interactions] should be: [user_id, item_id]"
    # the device is resolved BEFORE the first collective: under RCCL every rank must communicate on ITS GPU, also for a caller who
    # passes device=cuda:local_rank without having called torch.cuda.set_device (all ranks on cuda:0 = duplicate-GPU error or a hang)
    if make_trainer is None:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        if device.type == "cuda":
            torch.cuda.set_device(device)
    if world > 1:
        from .utils import get_data
        data = get_data(interactions)
        n_rows = len(data)
        r_lo, r_hi = n_rows * rank // world, n_rows * (rank + 1) // world
        uniq = [None] * world
        dist.all_gather_object(uniq, (pd.unique(data[r_lo:r_hi, 0]), pd.unique(data[r_lo:r_hi, 1])), group=group)
        model._set_ids(np.sort(pd.unique(np.concatenate([u for u, _ in uniq]))), np.sort(pd.unique(np.concatenate([i for _, i in uniq]))))
        mine = model._index_pairs(data[r_lo:r_hi], None)
        comm_dev = (device if isinstance(device, torch.device) and device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())) \
            if dist.get_backend(group) == "nccl" else torch.device("cpu")
        longest = max(n_rows * (r + 1) // world - n_rows * r // world for r in range(world))
        buf = torch.zeros((max(longest, 1), 2), dtype=torch.int32, device=comm_dev)
        buf[:len(mine)] = torch.as_tensor(mine).to(comm_dev)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        pairs = np.concatenate([parts[r][:n_rows * (r + 1) // world - n_rows * r // world].cpu().numpy() for r in range(world)])
        if sample_weight is not None:
            # (the same three checks, with the same messages, as RankFM._index_pairs makes on the single-process path)
            assert isinstance(sample_weight, (np.ndarray, pd.Series)), "[sample_weight] must be np.ndarray or pd.series"
            assert sample_weight.ndim == 1, "[sample_weight] must a vector (ndim=1)"
            assert len(sample_weight) == n_rows, "[sample_weight] must have the same length as [interactions]"
            model.sample_weight = np.ascontiguousarray(get_data(sample_weight), dtype=np.float32)
        else:
            model.sample_weight = np.ones(n_rows, dtype=np.float32)
    else:
        model._init_ids(interactions)
        pairs = model._index_pairs(interactions, sample_weight)
    model.interactions = np.ascontiguousarray(pairs, dtype=np.int32)
    n_users = len(model.user_idx)
    offsets = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(np.bincount(pairs[:, 0], minlength=n_users), out=offsets[1:])
    model._init_features(user_features, item_features)
    model._init_weights(user_features, item_features)
    max_samples = 1 if model.loss == "bpr" else model.max_samples            # rankfm/rankfm.py:294-297
    bounds = shard_boundaries(offsets, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sel = (pairs[:, 0] >= lo) & (pairs[:, 0] < hi)
    local = pairs[sel].astype(np.int32, copy=True)
    local[:, 0] -= lo
    local_csr = UserItemsCSR.from_pairs(local[:, 0], local[:, 1], hi - lo)
    shard = dict(interactions=np.ascontiguousarray(local), sample_weight=np.ascontiguousarray(model.sample_weight[sel]),
                 csr_offsets=local_csr.offsets, csr_items=local_csr.items, x_uf=np.ascontiguousarray(model.x_uf[lo:hi]),
                 v_u=np.ascontiguousarray(model.v_u[lo:hi]), row_mask=sel)
    hyper = dict(alpha=model.alpha, beta=model.beta, learning_rate=model.learning_rate, learning_schedule=model.learning_schedule,
                 learning_exponent=model.learning_exponent, max_samples=max_samples)
    tables = {k: getattr(model, k) for k in SHARED_NAMES}
    if make_trainer is None:
        seed = int(np.random.randint(0, 2**31 - 1)) + rank if model.engine.seed is None else int(model.engine.seed) + rank
        trainer, sess = make_device_trainer(shard, tables, model.x_if, hyper, device, group=group, merge_damping=merge_damping,
                                            syncs_per_epoch=syncs_per_epoch, overlap=overlap if merge_damping is None else False,
                                            exchange_dtype=exchange_dtype, seed=seed, has_user_features=int(model.x_uf.any()), has_item_features=int(model.x_if.any()),
                                            want_penalty=verbose, hogwild_damping=model.engine.damping,
                                            # every engine option of the single-GPU path applies to the shards as well
                                            debug_flags=int(model.engine.debug_flags),
                                            tune=model.engine.tune, n_workgroups=model.engine.n_workgroups,
                                            rows_per_launch=model.engine.rows_per_launch, check_finite=model.engine.check_finite)
        finish = (lambda: sess.weights["v_u"].detach().cpu().numpy()) if sess is not None else (lambda: np.zeros((0, model.factors), np.float32))   # noqa: E731
    else:
        trainer, finish = make_trainer(shard, tables, model.x_if, hyper, device, group)
    broadcast_from_rank0([trainer.shared.flat], group)
    for e in range(epochs):
        out = trainer.run_epoch(e)
        if verbose:
            # a peer whose slice failed raised right behind the exchange and will not join the collective below: look at the failure
            # flag of that exchange FIRST (the healthy ranks used to wait in the all-reduce for the NCCL watchdog)
            trainer.check_peers(synchronize=True)
            ll = torch.tensor([float(np.sum(out.get("log_likelihood", out.get("ll", [0.0]))))], dtype=torch.float64,
                              device=trainer.shared.flat.device)
            if world > 1:
                dist.all_reduce(ll, group=group)
            if rank == 0:
                print("\ntraining epoch:", e)
                print("log likelihood (un-penalised, all ranks):", round(float(ll.item()), 2))
    trainer.finish()                          # (late merge: the final blocking exchange; a peer's failure flagged in the LAST exchange)
    # assemble the full model on every rank: item-side tables are already identical, user factors are all-gathered
    for k in SHARED_NAMES:
        getattr(model, k)[...] = trainer.shared.views[k].detach().cpu().numpy()
    v_u_local = np.ascontiguousarray(finish(), dtype=np.float32)
    if world > 1:
        # tensors, not pickled objects: every rank contributes its shard padded to the largest shard (2.56 GB of user factors at
        # BASELINE config 5 would not survive all_gather_object)
        dev = trainer.shared.flat.device
        rows = int(np.max(np.diff(bounds)))
        mine = torch.zeros((max(rows, 1), model.v_u.shape[1]), dtype=torch.float32, device=dev)
        if hi > lo:
            mine[:hi - lo] = torch.as_tensor(v_u_local).to(dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        for r in range(world):
            plo, phi = int(bounds[r]), int(bounds[r + 1])
            if phi > plo:
                model.v_u[plo:phi] = parts[r][:phi - plo].cpu().numpy()
    else:
        model.v_u[lo:hi] = v_u_local
    # the complete per-user item lists on every rank: each rank contributes its users' sorted lists (padded to the longest)
    if world > 1:
        dev = trainer.shared.flat.device
        longest = int(np.max(offsets[bounds[1:]] - offsets[bounds[:-1]]))
        mine = torch.zeros(max(longest, 1), dtype=torch.int32, device=dev)
        mine[:len(local_csr.items)] = torch.as_tensor(local_csr.items).to(dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        items = np.concatenate([parts[r][:int(offsets[bounds[r + 1]] - offsets[bounds[r]])].cpu().numpy() for r in range(world)])
    else:
        items = local_csr.items
    model.user_items = UserItemsCSR(offsets, items)
    assert np.isfinite(model.v_u).all() and np.isfinite(model.v_i).all(), "model weights are not finite"
    model.epochs_trained += epochs
    model.is_fit = True
    return model
