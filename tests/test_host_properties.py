"""Property tests (hypothesis) of the host-side integer logic: RePaint schedule, molecule sharding, slice cuts.  CPU only."""
import importlib

import torch
from hypothesis import given, settings, strategies as st

from oracle import gcdm_oracle as O

vd = importlib.import_module("bio-diffusion_amd.variational_diffusion")
par = importlib.import_module("bio-diffusion_amd.parallel")


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 12), st.integers(1, 60), st.integers(1, 1200))
def test_repaint_schedule_closed_form_equals_the_restated_loop(r, j, T):
    """The product's closed form vs the oracle's restatement of the reference loop (itself pinned to 192 reference outputs and live), and the
    invariant the inpainting loop relies on: following the schedule with jumps of j never leaves [0, T-1] and ends exactly at s = -1."""
    sched = vd.repaint_schedule(r, j, T)
    assert sched == O.get_repaint_schedule(r, j, T)
    s = T - 1
    for i, n in enumerate(sched):
        assert n >= 1
        s -= n
        assert s >= -1
        if i < len(sched) - 1:
            s += j
            assert s <= T - 1
    assert s == -1
    # every stretch but the remainder is denoised exactly r times
    assert sum(sched) == T + (len(sched) - 1) * j


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 5000), st.integers(1, 16))
def test_shard_ranges_partition_the_batch(B, world):
    cover = []
    sizes = []
    for rank in range(world):
        lo, hi = par.shard_range(B, rank, world)
        assert 0 <= lo <= hi <= B
        cover += list(range(lo, hi))
        sizes.append(hi - lo)
    assert cover == list(range(B)) and max(sizes) - min(sizes) <= 1


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(1, 181), min_size=1, max_size=300), st.integers(1, 6))
def test_slice_cuts_are_a_contiguous_partition(sizes, K):
    nn_ = torch.tensor(sizes)
    if K > len(sizes):
        try:
            vd.slice_cuts(nn_, K)
            assert False, "expected ValueError"
        except ValueError:
            return
    cuts = vd.slice_cuts(nn_, K)
    assert cuts[0] == 0 and cuts[-1] == len(sizes) and len(cuts) == K + 1
    assert all(b > a for a, b in zip(cuts[:-1], cuts[1:]))
