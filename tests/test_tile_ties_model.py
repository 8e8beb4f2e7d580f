"""Model of how k_gather_dedup_build orders the small tie groups inside its tiles (biograph_amd/csrc/build_seqset.hip,
`sort_small`): the halo arithmetic and the classification rule restated in Python and checked by brute force — every
tile sees a window of its own positions plus a halo of K + 2 on each side, a member looks at most K + 1 positions to
each side for the ends of its group, a group with more than K members in sight is left alone (it arrives in order),
the others are ranked; a tile only uses its own positions and their two neighbours.  What must hold: whatever the group
sizes and wherever the tile edges fall, the positions every tile uses come out as in the full sort.  (No GPU: this pins
the argument, tests/test_gpu_parity.py pins the kernel.  The kernel's halo of K + 2 is generous: the model also holds
with K and breaks with K - 1.)"""
import random

import pytest


def tile_view(keys, n, b0, T, K):
    """What the tile at b0 computes: the record at each slot of [b0 - H, b0 + T + H) after ranking the small groups.
    keys[p] = (leading half, rest): groups = runs of equal leading halves."""
    H = K + 2
    NS = T + 2 * H
    slot_pos = [b0 + i - H for i in range(NS)]
    valid = [0 <= p < n for p in slot_pos]

    def same(q, k):
        return 0 <= q < NS and valid[q] and keys[slot_pos[q]][0] == k

    pmap = list(range(NS))
    for j in range(NS):
        at = j
        if valid[j]:
            k = keys[slot_pos[j]][0]
            if same(j - 1, k) or same(j + 1, k):
                s, e = j, j + 1
                while j - s <= K and same(s - 1, k):
                    s -= 1
                while e - s <= K and same(e, k):
                    e += 1
                if e - s <= K:
                    mine = keys[slot_pos[j]]
                    rank = 0
                    for q in range(s, e):
                        if q == j:
                            continue
                        other = keys[slot_pos[q]]
                        if other < mine or (other == mine and q < j):
                            rank += 1
                    at = s + rank
        pmap[at] = j
    assert sorted(pmap) == list(range(NS))  # a bijection: every slot is written once
    return {slot_pos[i]: keys[slot_pos[pmap[i]]] for i in range(NS) if valid[i]}, H


@pytest.mark.parametrize("K,T", [(3, 16), (8, 32), (1, 8), (5, 7)])
def test_tiles_order_small_groups_like_the_full_sort(K, T):
    rng = random.Random(1000 * K + T)
    for trial in range(60):
        # groups of 1 .. 3K members; the ones above K arrive ordered (the other kernels' work), the rest shuffled
        keys, want = [], []
        lead = 0
        while len(keys) < 5 * T + rng.randrange(T):
            g = rng.choice([1, 1, 2, 2, 3, K, K + 1, rng.randrange(1, 3 * K + 1)])
            members = sorted((lead, rng.randrange(4)) for _ in range(g))
            want += members
            if g <= K:
                rng.shuffle(members)
            keys += members
            lead += 1
        n = len(keys)
        for b0 in range(0, n, T):
            view, H = tile_view(keys, n, b0, T, K)
            # the tile's own positions and their two neighbours are what it reads afterwards
            for p in range(max(0, b0 - 1), min(n, b0 + T + 1)):
                assert view[p] == want[p], (trial, b0, p)
