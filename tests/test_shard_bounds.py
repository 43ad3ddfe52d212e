"""Property test of the batch partition every rank computes for itself (speakerguard_amd/shard.py)."""
from hypothesis import given, settings
from hypothesis import strategies as st

from speakerguard_amd.shard import shard_bounds


@settings(max_examples=300, deadline=None)
@given(n=st.integers(0, 5000), world=st.integers(1, 64), granule=st.integers(1, 64))
def test_shard_bounds_partition(n, world, granule):
    b = shard_bounds(n, world, granule)
    assert len(b) == world
    assert b[0][0] == 0 and b[-1][1] == n
    sizes = []
    for r, (s, e) in enumerate(b):
        assert 0 <= s <= e <= n
        if r:
            assert s == b[r - 1][1]  # contiguous, ordered, no overlap
        sizes.append(e - s)
    # every shard but the one holding the ragged tail is a whole number of granules, and the load is balanced to
    # within one granule
    ragged = [z for z in sizes if z % granule]
    assert len(ragged) <= 1
    full = [z for z in sizes if z and z % granule == 0]
    if full:
        assert max(full) - min(full) <= granule
    assert sum(sizes) == n


def _single_calls(ranges, batch_size):
    """utterances that end up alone in a model call when every range is attacked in chunks of min(batch_size, range)"""
    singles = set()
    for lo, hi in ranges:
        n = hi - lo
        if n <= 0:
            continue
        bs = min(batch_size, n)
        for s in range(lo, hi, bs):
            if min(hi, s + bs) - s == 1:
                singles.add(s)
    return singles


@settings(max_examples=500, deadline=None)
@given(n=st.integers(1, 700), world=st.integers(1, 16), batch_size=st.integers(1, 80))
def test_coupled_plan_never_changes_which_utterances_sit_alone_in_a_model_call(n, world, batch_size):
    """a batch-coupled defense (FeCo: reference defense/feature_level.py:33, force = feat.shape[0] > 1) treats a
    one-utterance model call differently; the cut over the ranks must leave alone exactly the utterances the unsharded run
    leaves alone."""
    from speakerguard_amd.shard import coupled_plan
    plan = coupled_plan(n, world, batch_size)
    assert len(plan) == world
    flat = [r for ranges in plan for r in ranges]
    assert flat[0][0] == 0 and flat[-1][1] == n
    for a, b in zip(flat, flat[1:]):
        assert a[1] == b[0] and a[0] < a[1]  # contiguous, ascending over the ranks, no empty range
    assert all(len(ranges) <= 2 for ranges in plan)
    assert _single_calls(flat, batch_size) == _single_calls([(0, n)], batch_size)
