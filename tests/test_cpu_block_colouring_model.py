"""The ALGORITHM of the per-block colouring of the blocked solver (nudge_amd/csrc/nh_blocks.h, k_blk_prepare_local), restated in numpy -- no GPU.
A block's contacts are coloured so that no two contacts of one colour share a dynamic body (local index 0 = the static world, which constrains nobody):
  * a contact offers the colour it had last step to both bodies; a colour offered twice to one body is given up by BOTH contacts (the bit stays marked);
  * the rest go through Jones-Plassmann rounds: every uncoloured contact bids for its two bodies with a hashed priority made of its index in the caller's
    contact list, a contact that holds both takes the lowest colour neither body uses.
What must hold, and what the device-side structure check (NH_BLK_CHECK=1) and the determinism tests rely on: the colouring is valid, and it does not depend on
the order in which the contacts of the block are visited (on the device: the arrival order of atomics)."""
import numpy as np

LL = 32


def colour_block(pairs, ids, seeds, visit):
    """pairs[k] = (la, lb) local bodies, ids[k] = index in the caller's list, seeds[k] in 0..LL; `visit` = the order the contacts are walked in."""
    n = len(pairs)
    nb = int(pairs.max()) + 1
    mask = np.zeros(nb, dtype=np.uint64)
    clash = np.zeros(nb, dtype=np.uint64)
    col = np.array(seeds, dtype=np.int64)
    for k in visit:                                                   # seeds offered (atomicOr; a bit found set: the clash mask)
        if col[k]:
            bit = np.uint64(1) << np.uint64(col[k] - 1)
            for b in pairs[k]:
                if b:
                    if mask[b] & bit:
                        clash[b] |= bit
                    mask[b] |= bit
    for k in visit:                                                   # both contacts of a clash give the colour up
        if col[k]:
            bit = np.uint64(1) << np.uint64(col[k] - 1)
            if any(b and (clash[b] & bit) for b in pairs[k]):
                col[k] = 0
    bid = ((ids.astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) << np.uint64(32) | ids.astype(np.uint64)
    rounds = 0
    while (col == 0).any():
        rounds += 1
        best = np.full(nb, np.iinfo(np.uint64).max, dtype=np.uint64)
        for k in visit:
            if not col[k]:
                for b in pairs[k]:
                    if b:
                        best[b] = min(best[b], bid[k])
        winners = [k for k in visit if not col[k] and all((not b) or best[b] == bid[k] for b in pairs[k])]
        assert winners, "no winner in a round: the bids are not a total order"
        for k in winners:                                             # (winners share no body: their masks are disjoint)
            used = np.uint64(0)
            for b in pairs[k]:
                if b:
                    used |= mask[b]
            q = 0
            while (used >> np.uint64(q)) & np.uint64(1):
                q += 1
            assert q < LL
            col[k] = q + 1
            for b in pairs[k]:
                if b:
                    mask[b] |= np.uint64(1) << np.uint64(q)
    return col, rounds


def _block(rng, n_bodies=300, radius=0.16):
    """Bodies scattered in a unit cube, a contact between every two closer than `radius`, some contacts with the static world."""
    p = rng.uniform(0.0, 1.0, size=(n_bodies, 3))
    pairs = []
    for i in range(n_bodies):
        d = np.linalg.norm(p[i + 1:] - p[i], axis=1)
        for j in np.flatnonzero(d < radius):
            pairs.append((i + 1, i + 2 + j))
        if p[i, 1] < 0.08:
            pairs.append((0, i + 1))                                 # on the floor: the static world is local body 0
    return np.array(pairs, dtype=np.int64)


def _valid(pairs, col):
    seen = set()
    for (a, b), c in zip(pairs, col):
        for x in (a, b):
            if x:
                assert (x, c) not in seen, "two contacts of one colour on one body"
                seen.add((x, c))


def test_block_colouring_is_valid_and_independent_of_the_visiting_order():
    rng = np.random.default_rng(4)
    for trial in range(4):
        pairs = _block(rng)
        n = len(pairs)
        ids = rng.permutation(100000)[:n]
        # no seeds (a first step)
        col0, rounds0 = colour_block(pairs, ids, np.zeros(n, dtype=np.int64), np.arange(n))
        _valid(pairs, col0)
        assert col0.max() <= LL
        # last step's colours as seeds, a fifth of them lost (new contacts), a few replaced by garbage that clashes
        seeds = col0.copy()
        seeds[rng.random(n) < 0.2] = 0
        bad = rng.random(n) < 0.05
        seeds[bad] = rng.integers(1, 6, size=int(bad.sum()))
        a, ra = colour_block(pairs, ids, seeds, np.arange(n))
        b, rb = colour_block(pairs, ids, seeds, rng.permutation(n))
        c, rc = colour_block(pairs, ids, seeds, np.arange(n)[::-1])
        _valid(pairs, a)
        assert np.array_equal(a, b) and np.array_equal(a, c), "the outcome depends on the visiting order"
        kept = int(((a == seeds) & (seeds > 0)).sum())
        assert kept > 0.6 * int((seeds > 0).sum())                   # most seeds survive: the rounds only have the rest to colour
        assert ra <= rounds0
