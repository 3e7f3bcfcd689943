"""Property tests (hypothesis) of the host-side logic the device plan relies on: the keyed permutation is a bijection on
every domain size, user segments tile the CSR rows, user shards tile the users, CSR construction is order-insensitive.
CPU only; the kernels use the same rules (include/rfm_rng.h, rfm_api.hip) and tests/test_synthetic.py ties the numpy
mirror to the C header."""
import numpy as np
from hypothesis import given, settings, strategies as st

from rankfm_amd import order
from rankfm_amd._rankfm import UserItemsCSR
from rankfm_amd.distributed import shard_boundaries


@settings(max_examples=60, deadline=None)
@given(n=st.integers(1, 3000), key=st.integers(0, 2 ** 32 - 1))
def test_keyed_permutation_is_a_bijection(n, key):
    out = order.perm(np.arange(n), n, order.perm_bits(n), key)
    assert out.min() >= 0 and out.max() < n and len(np.unique(out)) == n


@settings(max_examples=40, deadline=None)
@given(degrees=st.lists(st.integers(0, 200), min_size=1, max_size=60), seed=st.integers(0, 2 ** 31 - 1), epoch=st.integers(0, 50))
def test_segments_tile_the_rows_and_the_epoch_order_visits_each_once(degrees, seed, epoch):
    off = np.concatenate([[0], np.cumsum(degrees)]).astype(np.int64)
    users, begin, length = order.segments(off)
    assert length.sum() == off[-1] and (length <= order.SEGMENT_ROWS).all() and (length >= 1).all()
    # segments of a user are consecutive, disjoint and inside the user's range
    for u in np.unique(users):
        b, l = begin[users == u], length[users == u]
        assert b[0] == off[u] and (b[1:] == b[:-1] + l[:-1]).all() and b[-1] + l[-1] == off[u + 1]
        assert l.max() - l.min() <= 1                                  # near-equal runs
    if off[-1] > 0:
        pos = order.epoch_positions(off, seed, epoch)
        assert np.array_equal(np.sort(pos), np.arange(off[-1]))


@settings(max_examples=40, deadline=None)
@given(degrees=st.lists(st.integers(0, 500), min_size=1, max_size=80), world=st.integers(1, 9))
def test_user_shards_tile_the_users(degrees, world):
    off = np.concatenate([[0], np.cumsum(degrees)]).astype(np.int64)
    bounds = shard_boundaries(off, world)
    assert len(bounds) == world + 1 and bounds[0] == 0 and bounds[-1] == len(degrees)
    assert (np.diff(bounds) >= 0).all()


@settings(max_examples=40, deadline=None)
@given(pairs=st.lists(st.tuples(st.integers(0, 20), st.integers(0, 15)), min_size=0, max_size=300), seed=st.integers(0, 1000))
def test_csr_construction_does_not_depend_on_the_row_order(pairs, seed):
    p = np.array(pairs, dtype=np.int64).reshape(-1, 2)
    a = UserItemsCSR.from_pairs(p[:, 0], p[:, 1], 21)
    q = p[np.random.default_rng(seed).permutation(len(p))]
    b = UserItemsCSR.from_pairs(q[:, 0], q[:, 1], 21)
    assert np.array_equal(a.offsets, b.offsets) and np.array_equal(a.items, b.items)
    for u in range(21):
        assert np.array_equal(a[u], np.sort(p[p[:, 0] == u, 1]))
