"""The closed form the pipelined walk uses for one hop's admissions (vector_store_amd/csrc/pipe_device.hpp, TopOps::accept / merge),
restated in Python and checked against the loop it replaces -- usearch's `search_to_find_in_base` inner loop as the oracle restates
it (oracle/cpu_hnsw.cpp: `if (top.size() < top_limit || d < radius) { next.insert; if (ok) top.insert; radius = top.top() }`), one
neighbour at a time against a radius that moves with every admission.

No GPU: this pins the ARITHMETIC (who passes, where every entry lands); that the kernel computes it is what tests/test_gpu_pipe.py and
tests/test_gpu_parity.py check against the oracle bit for bit."""
import random

import pytest


def literal_hop(top, limit, row, admissible):
    """The CPU loop: returns (`top` afterwards, ascending; the neighbours that entered `next`)."""
    top = sorted(top)
    pushed = []
    for d, ok in zip(row, admissible):
        if len(top) == limit and not d < top[-1]:  # `top.size() < top_limit || d < radius`
            continue
        pushed.append(d)
        if ok:
            top.append(d)
            top.sort()
            if len(top) > limit:
                top.pop()
    return top, pushed


def closed_form_hop(top, limit, row, admissible, rows_cap=256):
    """TopOps::accept + merge: counts instead of a moving radius, destinations instead of insertions.
    `rows_cap`: the register rows' capacity (unused positions hold +inf, as in the kernel)."""
    top = sorted(top)
    inf = float("inf")
    regs = top + [inf] * (rows_cap - len(top))
    passed, le = [], []
    # the shortcut (TopOps::accept, `sure`): with B admissible neighbours in the row, one closer than the member at position
    # limit - 1 - B passes whatever the others do -- no count needed for it
    b_all = sum(1 for ok in admissible if ok)
    threshold = regs[limit - 1 - b_all] if b_all < limit else -inf
    for j, dj in enumerate(row):
        gt = sum(1 for x in regs if x > dj)  # ballots over the rows (the +inf padding counted, taken off next)
        le_j = rows_cap - gt  # members of `top` not farther than neighbour j
        before = sum(1 for i in range(j) if admissible[i] and row[i] <= dj)  # admissible neighbours earlier in the row, not farther
        exact = le_j + before < limit
        if dj < threshold:
            assert exact, "the shortcut passed a neighbour the exact count turns away"
        passed.append(exact)
        le.append(le_j)
    new = [j for j in range(len(row)) if passed[j] and admissible[j]]
    merged = {}
    for p, x in enumerate(top):  # an old member moves up by the number of new ones closer than it
        merged[p + sum(1 for j in new if row[j] < x)] = x
    for j in new:  # a new one: old members not farther + new ones in front of it (closer, or as close and earlier in the row)
        in_front = sum(1 for i in new if row[i] < row[j] or (row[i] == row[j] and i < j))
        dest = le[j] + in_front
        assert dest not in merged, "two entries with one destination"
        merged[dest] = row[j]
    total = len(top) + len(new)
    assert sorted(merged) == list(range(total)), "holes in the merged list"
    out = [merged[i] for i in range(total)]
    assert out == sorted(out)
    return out[:limit], [row[j] for j in range(len(row)) if passed[j]]


def _case(rng, limit, ties):
    size = rng.choice([0, limit, limit, rng.randint(0, limit)])
    universe = 40 if ties else 100_000
    top = [float(rng.randrange(universe)) for _ in range(size)]
    row = [float(rng.randrange(universe)) for _ in range(rng.randint(0, 32))]
    if top and row and rng.random() < 0.3:
        row[0] = max(top)  # a neighbour exactly AT the radius: turned away (`d < radius` is strict)
    admissible = [rng.random() < rng.choice([1.0, 0.5]) for _ in row]
    if size == limit and top:  # the kernel only asks about neighbours below the hop's first radius once `top` is full
        r0 = max(top)
        keep = [i for i in range(len(row)) if row[i] < r0]
        row, admissible = [row[i] for i in keep], [admissible[i] for i in keep]
    return top, row, admissible


@pytest.mark.parametrize("ties", [False, True], ids=["distinct", "many_equal_distances"])
def test_closed_form_admissions_equal_the_literal_loop(ties):
    rng = random.Random(7 + ties)
    for _ in range(20_000):
        limit = rng.choice([1, 2, 5, 10, 64, 200])
        top, row, admissible = _case(rng, limit, ties)
        assert closed_form_hop(top, limit, row, admissible) == literal_hop(top, limit, row, admissible), (top, limit, row, admissible)


def test_a_row_that_fills_top_midway_meets_the_radius_behind_it():
    # three members, limit 4: the first admissible neighbour fills `top`, the ones behind it meet a finite radius
    top, limit = [1.0, 2.0, 3.0], 4
    row, ok = [9.0, 8.0, 2.5, 9.5], [True, True, True, True]
    assert literal_hop(top, limit, row, ok) == ([1.0, 2.0, 2.5, 3.0], [9.0, 8.0, 2.5])
    assert closed_form_hop(top, limit, row, ok) == literal_hop(top, limit, row, ok)
    # ... and a neighbour that is not admissible (a removed member) is pushed but does not move the radius
    ok = [False, True, True, True]
    assert closed_form_hop(top, limit, row, ok) == literal_hop(top, limit, row, ok)


def front_merge(front, new):
    """pipe_device.hpp `front_merge`: the sorted front of `next` (<= 64 entries, one per lane) and a hop's pushes below its reach, by
    destination: an old entry moves up by the number of new ones strictly closer, a new one lands behind the old ones not farther than
    it and behind the new ones in front of it (closer, or as close and earlier in the row).  Returns (front afterwards, what went to
    the pool)."""
    dest = {}
    for p, x in enumerate(front):
        dest[p + sum(1 for d in new if d < x)] = x
    for i, d in enumerate(new):
        old_le = sum(1 for x in front if x <= d)
        before = sum(1 for j, e in enumerate(new) if e < d or (e == d and j < i))
        assert old_le + before not in dest, "two entries with one destination"
        dest[old_le + before] = d
    total = len(front) + len(new)
    assert sorted(dest) == list(range(total)), "holes in the merged front"
    merged = [dest[i] for i in range(total)]
    return merged[:64], merged[64:]


def test_front_merge_is_a_sorted_merge_and_the_overflow_is_the_farthest():
    rng = random.Random(11)
    for _ in range(20_000):
        universe = rng.choice([30, 100_000])  # (30: many equal distances)
        front = sorted(float(rng.randrange(universe)) for _ in range(rng.randint(0, 64)))
        new = [float(rng.randrange(universe)) for _ in range(rng.randint(2, 32))]
        kept, pool = front_merge(front, new)
        everything = sorted(front + new)
        assert kept == everything[:64] and sorted(pool) == everything[64:]
