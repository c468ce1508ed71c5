"""The chained sweep of the blocked solver (nudge_amd/csrc/nh_blocks.h: k_blk_sweep<.., CHAIN>, k_blk_chain_nbr) restated in numpy / plain Python and checked on the
CPU: the eight block colours of a pass run as ONE launch; a workgroup draws its task from a ticket counter in (colour, block) order and waits only for the tasks of
EARLIER colours among its 26 grid neighbours.  What has to hold, whatever the hardware does with the workgroups:

  * no deadlock: with ANY number of workgroup slots and ANY order in which running workgroups make progress, every task finishes -- because a task only ever waits
    for tasks with a LOWER ticket, and tickets go out in the order workgroups start;
  * no two tasks that touch a common body run at the same time (two blocks touch a common body only if they are grid neighbours: a block borrows ghosts from
    neighbours only, and only neighbours borrow from it);
  * every pair of neighbouring blocks runs in the order of the eight launches (lower colour first) -- so each task reads exactly what it reads there, and the
    results are the bits of the eight launches (checked on the GPU by tests/test_gpu_blocks.py::test_chained_sweeps_are_the_bits_of_the_eight_launches)."""
import numpy as np
import pytest

NONE = 0xFFFFFFFF


def build_tasks(dims, has_work):
    """Tasks in launch order (colour = parity of the block coordinates, blocks of one colour in arbitrary order -- the device fills its lists by atomics) and the
    wait lists of k_blk_chain_nbr: the neighbours with work whose task comes in an earlier colour."""
    d0, d1, d2 = dims
    rng = np.random.default_rng(d0 * 100 + d1 * 10 + d2)
    blocks = [(x, y, z) for z in range(d2) for y in range(d1) for x in range(d0) if has_work[x, y, z]]
    colour = lambda b: (b[0] & 1) | (b[1] & 1) << 1 | (b[2] & 1) << 2          # noqa: E731
    phases = [[b for b in blocks if colour(b) == c] for c in range(8)]
    for p in phases:
        rng.shuffle(p)
    tasks = [b for p in phases for b in p]
    start = np.cumsum([0] + [len(p) for p in phases])
    slot = {b: t for t, b in enumerate(tasks)}
    waits = []
    for t, b in enumerate(tasks):
        p = colour(b)
        w = []
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    n = (b[0] + dx, b[1] + dy, b[2] + dz)
                    if n != b and n in slot and slot[n] < start[p]:
                        w.append(slot[n])
        waits.append(w)
    return tasks, waits, colour


def neighbours(a, b):
    return a != b and all(abs(a[k] - b[k]) <= 1 for k in range(3))


@pytest.mark.parametrize("dims,fill,slots", [((4, 4, 4), 1.0, 3), ((5, 3, 4), 0.7, 8), ((8, 8, 2), 0.9, 64), ((2, 2, 2), 1.0, 1), ((6, 5, 4), 0.5, 5), ((9, 1, 1), 1.0, 2)])
def test_chained_pass_never_deadlocks_never_overlaps_neighbours_and_keeps_the_launch_order(dims, fill, slots):
    rng = np.random.default_rng(sum(dims) * 7 + slots)
    has_work = rng.random(dims) < fill
    tasks, waits, colour = build_tasks(dims, has_work)
    n = len(tasks)
    for trial in range(20):
        # `slots` workgroups are resident; a free slot is taken by a new workgroup, which draws the next ticket when it STARTS; a resident workgroup is waiting
        # (some task it waits for is not done), or running; the scheduler picks at random what happens next
        ticket, done, finish_order = 0, [False] * n, []
        resident = []                       # [task, state] with state "waiting" / "running"
        running_blocks = set()
        steps = 0
        while len(finish_order) < n:
            steps += 1
            assert steps < 100 * n + 1000, "no progress: deadlock"
            events = []
            if len(resident) < slots and ticket < n:
                events.append("start")
            for i, (t, state) in enumerate(resident):
                if state == "waiting" and all(done[w] for w in waits[t]):
                    events.append(("go", i))
                if state == "running":
                    events.append(("finish", i))
            assert events, "nothing can happen: deadlock"
            ev = events[rng.integers(len(events))]
            if ev == "start":
                resident.append([ticket, "waiting"]); ticket += 1
            elif ev[0] == "go":
                t = resident[ev[1]][0]
                for other in running_blocks:
                    assert not neighbours(tasks[t], tasks[other]), "two neighbouring blocks run at the same time"
                running_blocks.add(t)
                resident[ev[1]][1] = "running"
            else:
                t = resident[ev[1]][0]
                running_blocks.discard(t); done[t] = True; finish_order.append(t)
                resident.pop(ev[1])
        # every pair of neighbours in the order of the eight launches: lower colour first
        pos = {t: i for i, t in enumerate(finish_order)}
        for a in range(n):
            for b in range(a + 1, n):
                if neighbours(tasks[a], tasks[b]):
                    ca, cb = colour(tasks[a]), colour(tasks[b])
                    assert ca != cb, "neighbouring blocks never share a colour"
                    assert (pos[a] < pos[b]) == (ca < cb), "a pair of neighbouring blocks ran in another order than the eight launches run them"
