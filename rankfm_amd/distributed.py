"""User-sharded data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-process (no counterpart to cite); the design follows SURVEY.md §8(e):

  * users are split into `world_size` contiguous index ranges balanced by interaction count; a rank owns its users'
    interactions, CSR slice, `v_u` rows and `x_uf` rows exclusively -- no traffic for them, ever;
  * the item-side / shared tables (`v_i, w_i, v_if, w_if, v_uf`) are replicated.  They live in ONE flat fp32
    buffer per rank (the weight tensors are views into it), so the exchange step is a single bucket:
        delta = flat - flat_at_epoch_start;  all_reduce(delta, SUM);  flat = flat_at_epoch_start + delta
    i.e. every rank ends the epoch with all ranks' updates applied (sum of deltas; `average=True` gives the
    conservative mean).  One all-reduce per epoch (or per `syncs_per_epoch` slice of it): 13 MB at BASELINE
    config 2, 52 MB at config 4, 516 MB at config 5 -- a few ms on 7 x 153 GB/s xGMI links against epochs of
    tens to hundreds of ms, so one large collective per epoch is the right granularity for point-to-point xGMI.

Everything here works on CPU tensors with the gloo backend as well, which is how the N > 1 logic is tested
without GPUs (tests/test_distributed_cpu.py).
"""
import numpy as np
import torch
import torch.distributed as dist

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

    def __init__(self, tables, device):
        shapes = {k: tuple(tables[k].shape) for k in SHARED_NAMES}
        sizes = {k: int(np.prod(shapes[k])) for k in SHARED_NAMES}
        # 64-element (256-byte) alignment of every table inside the bucket keeps 16-byte row loads aligned
        starts, total = {}, 0
        for k in SHARED_NAMES:
            starts[k] = total
            total += (sizes[k] + 63) // 64 * 64
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.views = {}
        for k in SHARED_NAMES:
            v = self.flat[starts[k]:starts[k] + sizes[k]].view(shapes[k])
            v.copy_(torch.as_tensor(np.asarray(tables[k]), dtype=torch.float32))
            self.views[k] = v
        self.start = self.flat.clone()

    def begin_epoch(self):
        self.start.copy_(self.flat)

    def all_reduce_deltas(self, group=None, average=False):
        """the exchange step: after it every rank holds epoch_start + sum (or mean) over ranks of its epoch's deltas"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        self.flat.sub_(self.start)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat.div_(dist.get_world_size(group))
        self.flat.add_(self.start)

    @property
    def payload_bytes(self):
        return self.flat.numel() * 4


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

    def __init__(self, shared, epoch_fn, group=None, average=False, syncs_per_epoch=1):
        self.shared, self.epoch_fn, self.group, self.average = shared, epoch_fn, group, average
        self.syncs_per_epoch = syncs_per_epoch

    def run_epoch(self, epoch):
        self.shared.begin_epoch()
        out = self.epoch_fn(self.shared.views, epoch)
        self.shared.all_reduce_deltas(self.group, self.average)
        return out


def make_device_trainer(shard, shared_tables, x_if, hyper, device, group=None, average=False, **session_kw):
    """wire a rank's shard to the HIP engine: weights are views into the flat bucket, so the engine's in-place
    atomics and the all-reduce act on the same memory"""
    from .engine import DeviceSession
    shared = SharedTables(shared_tables, device)
    weights = dict(shared.views)
    weights["v_u"] = torch.as_tensor(shard["v_u"]).to(device)
    sess = DeviceSession(shard["interactions"], shard["sample_weight"], shard["csr_offsets"], shard["csr_items"],
                         shard["x_uf"], x_if, weights, device=device, **hyper, **session_kw)
    # DeviceSession.up() keeps tensors that are already resident float32 contiguous -> still the bucket views
    for k in SHARED_NAMES:
        assert sess.weights[k].data_ptr() == shared.views[k].data_ptr(), "shared table was copied out of the bucket"

    def epoch_fn(_views, epoch):
        return sess.run(epochs=1, epoch_begin=epoch)

    return ShardedTrainer(shared, epoch_fn, group=group, average=average), sess
