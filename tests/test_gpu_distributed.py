"""The multi-GPU fit on the real engine with TWO ranks sharing the one GPU of the test box (gloo carries the exchange; the
driver's scaling runs use one GPU per rank over RCCL).  Exercises make_device_trainer, the bucket all-reduce on device
tensors, the damped merge and the final assembly of the model."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

U, I, N, F = 6000, 3000, 300_000, 32


def _worker(rank, world, port, out_dir, exchange_dtype="fp32"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from rankfm_amd import EngineOptions, RankFM, synthetic
    from rankfm_amd.distributed import fit_distributed
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs, _ = synthetic.make_interactions(U, I, N, seed=1)
    m = RankFM(factors=F, engine=EngineOptions(seed=9))
    np.random.seed(4)
    fit_distributed(m, pairs, epochs=3, device=torch.device("cuda", 0), exchange_dtype=exchange_dtype)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), v_u=m.v_u, v_i=m.v_i, w_i=m.w_i)
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange_dtype", ["fp32", "bf16"])
def test_fit_distributed_two_ranks_one_gpu(tmp_path, oracle, exchange_dtype):
    """(`bf16`: the tables' deltas travel as bfloat16, SharedTables.exchange_dtype -- the same bars)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), exchange_dtype), nprocs=2, join=True)
    a, b = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for k in ("v_u", "v_i", "w_i"):
        assert np.array_equal(a[k], b[k]) and np.isfinite(a[k]).all(), k
    # and it learned what single-process sequential training learns (norms within 5 %, see test_two_user_shards_...)
    from rankfm_amd import RankFM, synthetic
    pairs, _ = synthetic.make_interactions(U, I, N, seed=1)
    m = RankFM(factors=F)
    np.random.seed(4)
    m._init_all(pairs)
    oracle.fit(m.interactions, m.sample_weight, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i,
               m.v_uf, m.v_if, 0.01, 0.1, 0.1, "constant", 0.25, 1, 3, perms=None, rng_mode=oracle.RNG_COUNTER, seed=9, membership="binary")
    for k in ("v_u", "v_i", "w_i"):
        got, want = np.linalg.norm(a[k]), np.linalg.norm(getattr(m, k))
        assert abs(got - want) <= 0.05 * want, (k, got, want)


def _nccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from rankfm_amd import EngineOptions, RankFM, synthetic
    from rankfm_amd.distributed import fit_distributed
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)          # "nccl" IS RCCL on ROCm
    assert dist.get_backend() == "nccl" and dist.get_world_size() == world
    pairs, _ = synthetic.make_interactions(U, I, N, seed=1)
    m = RankFM(factors=F, engine=EngineOptions(seed=9))
    np.random.seed(4)
    fit_distributed(m, pairs, epochs=2, device=dev)
    np.savez(os.path.join(out_dir, "n%d.npz" % rank), v_u=m.v_u, v_i=m.v_i, w_i=m.w_i)
    dist.destroy_process_group()


def test_fit_distributed_over_rccl():
    """the same fit over the production backend: one rank per visible GPU (two when the box has them, else a world of one) with
    torch.distributed's "nccl" backend, which is RCCL on ROCm.  With two ranks the replicas must agree bit for bit."""
    import tempfile
    import torch
    world = 2 if torch.cuda.device_count() >= 2 else 1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_nccl_worker, args=(world, port, d), nprocs=world, join=True)
        r = [np.load(os.path.join(d, "n%d.npz" % k)) for k in range(world)]
    for k in ("v_u", "v_i", "w_i"):
        assert np.isfinite(r[0][k]).all() and all(np.array_equal(r[0][k], x[k]) for x in r[1:]), k


def test_fused_delta_kernels_match_the_elementwise_form():
    """rfm_delta_begin / rfm_delta_finish (one pass over the bucket on each side of the all-reduce) against the four elementwise
    torch passes they replace; odd length and a per-element scale, a uniform scale, and the plain sum"""
    import ctypes as C
    import torch
    from rankfm_amd import _hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    n = 1_000_003
    flat = torch.randn(n, generator=g).to(dev)
    start = torch.randn(n, generator=g).to(dev)
    scale = torch.rand(n, generator=g).to(dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    want_delta = flat - start
    f = flat.clone()
    assert _hip.lib().rfm_delta_begin(f.data_ptr(), start.data_ptr(), n, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(f, want_delta)
    for sc, uni in ((scale, 1.0), (None, 0.125), (None, 1.0)):
        h = want_delta.clone()
        assert _hip.lib().rfm_delta_finish(h.data_ptr(), start.data_ptr(), None if sc is None else sc.data_ptr(), uni, n, stream) == 0
        torch.cuda.synchronize()
        want = start + (sc if sc is not None else uni) * want_delta
        assert torch.allclose(h, want, rtol=0, atol=1e-6)
    # unaligned views (tables inside a bucket start at 256-byte boundaries, but the entry points must not assume it)
    f2 = flat.clone()
    assert _hip.lib().rfm_delta_begin(f2[1:].data_ptr(), start[1:].data_ptr(), n - 1, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(f2[1:], want_delta[1:]) and f2[0] == flat[0]
