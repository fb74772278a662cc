"""The rule by which pass 2 of the IDW search decides a voxel whose 4th and 5th distances tie (csrc/idw.hip, IDW_CONSIDER_FIXED),
checked on the CPU against a literal simulation of the reference's selection: torch.topk(k=4, largest=False) on >= 256 points is
std::partial_sort = libstdc++ __heap_select (max-heap of 4, root replaced on STRICTLY smaller value) + __sort_heap (oracle/idw_knn.c).

Rule: with D the 4th smallest distance and S the points with d <= D in index order, the selected set is the first four members of S
unless a member with d < D follows them (then the outcome depends on the whole history, far points included: the GPU replays those)."""
import random


def _adjust_heap(f, hole, ln, value):
    top, child = hole, hole
    while child < (ln - 1) // 2:
        child = 2 * (child + 1)
        if f[child][0] < f[child - 1][0]:
            child -= 1
        f[hole] = f[child]
        hole = child
    if (ln & 1) == 0 and child == (ln - 2) // 2:
        child = 2 * (child + 1)
        f[hole] = f[child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > top and f[parent][0] < value[0]:
        f[hole] = f[parent]
        hole = parent
        parent = (hole - 1) // 2
    f[hole] = value


def reference_selection(ds):
    h = [(ds[j], j) for j in range(4)]
    for parent in (1, 0):
        _adjust_heap(h, parent, 4, h[parent])
    for j in range(4, len(ds)):
        if ds[j] < h[0][0]:
            _adjust_heap(h, 0, 4, (ds[j], j))
    return sorted(i for _, i in h)


def fixed_bound_decision(ds):
    """None = left to the replay."""
    D = sorted(ds)[3]
    s = []
    for j, d in enumerate(ds):
        if d <= D:
            if len(s) < 4:
                s.append(j)
            elif d < D:
                return None
    return sorted(s)


def test_decided_voxels_equal_the_heap_select():
    rng = random.Random(1)
    decided = 0
    for _ in range(40000):
        ds = [rng.randint(0, 6) for _ in range(rng.randint(5, 14))]        # small integers: ties everywhere
        got = fixed_bound_decision(ds)
        if got is not None:
            decided += 1
            assert got == reference_selection(ds), ds
    assert decided > 20000


def test_undecided_voxels_do_depend_on_the_far_points():
    """The cases the rule leaves to the replay are not decidable from the points within D: interleaving far points changes them."""
    near = [1, 3, 3, 3, 2]                                                  # D = 3, a point with d < D arrives fifth
    assert fixed_bound_decision(near) is None
    outs = set()
    rng = random.Random(2)
    for _ in range(200):
        seq = [(v, i) for i, v in enumerate(near)]
        for _ in range(rng.randint(0, 10)):
            seq.insert(rng.randint(0, len(seq)), (3 + rng.randint(1, 5), -1))
        sel = reference_selection([v for v, _ in seq])
        outs.add(tuple(sorted(seq[i][1] for i in sel)))
    assert len(outs) > 1, outs
