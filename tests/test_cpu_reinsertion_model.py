"""The ALGORITHM of the kept pair list with re-insertion (nudge_amd/csrc/nh_collide.hip, "6. kept pairs", "7. re-insertion"), restated in numpy and checked
against a brute-force overlap test on random motion -- no GPU.  The kernels are tested on the device (tests/test_gpu_parity.py, test_gpu_atsize.py); this file
pins the reasoning they rest on:
  * boxes are inflated by a margin and sorted into a uniform grid by the cell of their min corner; a box is "small" iff it spans at most two cells per axis,
    so a partner of box A has its min-corner cell in [c_min(A) - 1, c_max(A)] per axis; large boxes live on a list;
  * the kept list holds the pairs of overlapping INFLATED boxes; while every box stays inside its inflated box, filtering the list with the exact boxes gives
    exactly the overlapping pairs;
  * a small box that leaves its inflated box gets a new one and the next generation stamp; kept pairs carry the stamps of both members and die when a stamp
    is stale; the leaver's new pairs come from the OLD grid for everybody who has not moved since the rebuild (stamp 0; stale entries skipped), from the large
    list, and from the list of everybody who has moved (current boxes), two leavers of one step meeting once."""
import numpy as np


def overlap(amin, amax, bmin, bmax):
    return bool(np.all(bmax > amin) and np.all(amax > bmin))          # strict, nudge.cpp:3306-3308


class KeptList:
    def __init__(self, cell, margin):
        self.cell, self.margin = cell, margin
        self.valid = False

    def cells(self, x):
        return np.floor(x / self.cell).astype(np.int64)

    def rebuild(self, mn, mx):
        n = len(mn)
        self.fmin, self.fmax = mn - self.margin, mx + self.margin
        self.gen = np.zeros(n, dtype=np.int64)
        c0, c1 = self.cells(self.fmin), self.cells(self.fmax)
        self.large = np.any(c1 - c0 > 1, axis=1)
        self.grid = {}
        for i in np.flatnonzero(~self.large):
            self.grid.setdefault(tuple(c0[i]), []).append((i, self.fmin[i].copy(), self.fmax[i].copy()))       # a COPY of the box, like the cell-sorted array
        self.large_list = list(np.flatnonzero(self.large))
        self.moved = []
        self.kept = []
        for i in range(n):                                            # (the grid search; order and method do not matter for the model)
            for j in range(i + 1, n):
                if overlap(self.fmin[i], self.fmax[i], self.fmin[j], self.fmax[j]):
                    self.kept.append((i, 0, j, 0))
        self.valid = True
        self.rebuilds = getattr(self, "rebuilds", 0) + 1

    def query(self, i):
        """Partners of box i's CURRENT inflated box among everybody else: old grid (stamp 0 only), large list, moved list."""
        out = []
        c0, c1 = self.cells(self.fmin[i]), self.cells(self.fmax[i])
        for cx in range(c0[0] - 1, c1[0] + 1):
            for cy in range(c0[1] - 1, c1[1] + 1):
                for cz in range(c0[2] - 1, c1[2] + 1):
                    for (j, bmin, bmax) in self.grid.get((cx, cy, cz), []):
                        if self.gen[j] == 0 and j != i and overlap(self.fmin[i], self.fmax[i], bmin, bmax):
                            out.append(j)
        for j in self.large_list:
            if overlap(self.fmin[i], self.fmax[i], self.fmin[j], self.fmax[j]):
                out.append(j)
        return out

    def step(self, mn, mx):
        if not self.valid:
            self.rebuild(mn, mx)
        else:
            out = ~(np.all(mn >= self.fmin, axis=1) & np.all(mx <= self.fmax, axis=1))
            leavers = list(np.flatnonzero(out))
            if any(self.large[i] for i in leavers):
                self.rebuild(mn, mx)                                  # a large collider on the move: the rebuild
            else:
                for i in leavers:                                     # k_xform: new box, next stamp, on the lists
                    self.fmin[i], self.fmax[i] = mn[i] - self.margin, mx[i] + self.margin
                    if self.gen[i] == 0:
                        self.moved.append(i)
                    self.gen[i] += 1
                    c0, c1 = self.cells(self.fmin[i]), self.cells(self.fmax[i])
                    assert np.all(c1 - c0 <= 2)                       # same extent as when it was small: at most three cells in this arithmetic
                now = set(leavers)
                for i in leavers:                                     # k_reinsert
                    for j in self.query(i):
                        self.kept.append((i, self.gen[i], j, self.gen[j]))
                    for j in self.moved:
                        if j == i or (j in now and j > i):
                            continue
                        if overlap(self.fmin[i], self.fmax[i], self.fmin[j], self.fmax[j]):
                            self.kept.append((i, self.gen[i], j, self.gen[j]))
        # k_kept_filter: exact boxes, current stamps
        pairs = set()
        for (i, gi, j, gj) in self.kept:
            if self.gen[i] == gi and self.gen[j] == gj and overlap(mn[i], mx[i], mn[j], mx[j]):
                p = (min(i, j), max(i, j))
                assert p not in pairs, "a pair twice in the kept list"
                pairs.add(p)
        return pairs


def brute(mn, mx):
    n = len(mn)
    return {(i, j) for i in range(n) for j in range(i + 1, n) if overlap(mn[i], mx[i], mn[j], mx[j])}


def test_kept_list_with_reinsertion_gives_the_pairs_of_a_full_search():
    rng = np.random.default_rng(7)
    n = 70
    pos = rng.uniform(0.0, 6.0, size=(n, 3))
    half = rng.uniform(0.2, 0.45, size=(n, 3))
    half[0] = (4.0, 0.1, 4.0); pos[0] = (3.0, -0.1, 3.0)            # a slab: large for the grid
    vel = np.zeros((n, 3))
    k = KeptList(cell=1.0, margin=0.03)
    inserts = 0
    for step in range(160):
        if step % 7 == 0:                                            # a few boxes get a push; now and then one is teleported, once the slab itself moves
            who = rng.choice(np.arange(1, n), size=4, replace=False)
            vel[who] = rng.uniform(-0.08, 0.08, size=(4, 3))
        if step % 31 == 30:
            pos[rng.integers(1, n)] = rng.uniform(0.0, 6.0, size=3)
        if step == 100:
            pos[0, 0] += 0.5
        vel *= 0.93
        pos += vel
        mn, mx = pos - half, pos + half
        before = sum(k.gen) if k.valid else 0
        got = k.step(mn, mx)
        inserts += max(0, int(sum(k.gen) - before)) if k.valid else 0
        assert got == brute(mn, mx), step
    assert inserts > 100 and k.rebuilds <= 3, (inserts, k.rebuilds)          # (the first step, the slab's move -- not one per leaver)
