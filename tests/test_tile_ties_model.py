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


def small_groups_pass(keys, n, B, K_small, K_tile, defer):
    """k_sort_small_groups restated: workgroups of B positions; a member looks K_small + 1 positions to each side for
    the ends of its group; groups of at most K_small members are ordered by their head's workgroup (all of them, or with
    `defer` only those above K_tile), which also writes the members that lie past its own positions (at most one group
    per workgroup reaches past the end); everything else is written by its own thread.  Returns (sidx order as keys,
    flags of the big groups)."""
    out = [None] * n
    flags = [0] * n
    written = [0] * n
    for b0 in range(0, n, B):
        lds = {p: keys[p] for p in range(b0, min(n, b0 + B + K_small))}
        tail = 0
        info = {}
        for i in range(b0, min(n, b0 + B)):
            k = keys[i][0]
            s, e = i, i + 1
            small = False
            big = 0
            tied = (i > 0 and keys[i - 1][0] == k) or (i + 1 < n and keys[i + 1][0] == k)
            if tied:
                while s > 0 and i - s <= K_small and keys[s - 1][0] == k:
                    s -= 1
                while e < n and e - s <= K_small and keys[e][0] == k:
                    e += 1
                if e - s > K_small:
                    big = 1
                else:
                    small = True
            here = small and (not defer or e - s > K_tile)
            info[i] = (s, e, here, big)
        for i, (s, e, here, big) in info.items():
            if here and s == i:  # head: order the group where it lies
                lds.update(zip(range(s, e), sorted(lds[p] for p in range(s, e))))
                if e > b0 + B:
                    assert tail == 0
                    tail = e - (b0 + B)
        for i, (s, e, here, big) in info.items():
            if not (here and s < b0):
                out[i] = lds[i]
                written[i] += 1
            flags[i] = big
        for t in range(tail):
            out[b0 + B + t] = lds[b0 + B + t]
            written[b0 + B + t] += 1
    assert written == [1] * n  # every position is written exactly once
    return out, flags


@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("B,K_small,K_tile", [(16, 5, 2), (8, 6, 3), (32, 4, 1)])
def test_small_groups_pass_then_tiles_give_the_full_sort(B, K_small, K_tile, defer):
    """The pass over the sorted keys followed (with defer) by the tiles' ranking: groups up to K_small come out ordered,
    bigger ones are flagged and left alone."""
    rng = random.Random(B * 100 + K_small * 10 + K_tile + (1000 if defer else 0))
    for trial in range(40):
        keys, want, size_of = [], [], []
        lead = 0
        while len(keys) < 6 * B + rng.randrange(B):
            g = rng.choice([1, 1, 2, 3, K_tile, K_tile + 1, K_small, K_small + 1, rng.randrange(1, 2 * K_small + 2)])
            members = [(lead, rng.randrange(5)) for _ in range(g)]
            keys += members
            want += sorted(members) if g <= K_small else members  # big groups: someone else's work
            size_of += [g] * g
            lead += 1
        n = len(keys)
        out, flags = small_groups_pass(keys, n, B, K_small, K_tile, defer)
        assert flags == [1 if g > K_small else 0 for g in size_of]
        if defer:
            # the tiles see the big groups as the later kernels leave them: ordered
            fixed, p = [], 0
            while p < n:
                g = size_of[p]
                fixed += sorted(out[p:p + g]) if g > K_small else out[p:p + g]
                p += g
            want_all = []
            p = 0
            while p < n:
                g = size_of[p]
                want_all += sorted(keys[p:p + g])
                p += g
            T = 2 * B
            for b0 in range(0, n, T):
                view, _ = tile_view(fixed, n, b0, T, K_tile)
                for q in range(max(0, b0 - 1), min(n, b0 + T + 1)):
                    assert view[q] == want_all[q], (trial, b0, q)
        else:
            assert out == want
