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
