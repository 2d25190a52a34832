"""The multi-cascade marcher's walk through one 64-point chunk (csrc/nerf_kernels.hip, k1_count phase B) as a host model: the reference's skip rule
(testbed_nerf.cu:798-807: an occupied lattice point goes to its successor, an empty one jumps by advance_to_next_voxel's length, a point outside the box ends the march)
followed ONE VISITED POINT AT A TIME -- what the kernel did up to round 6 -- against the orbit marked by POINTER DOUBLING over the 64 lanes (round k joins the visited
prefix of length 2^k with its image under next^(2^k); ends when a round adds nothing).  Same visited set, same emitted samples, same exit flag, same landing point, for
random chunks and the corner cases (entry behind the chunk, jump past the lattice, box boundary inside the chunk, all occupied, all empty with unit / long jumps).
CPU only: the device kernel itself is held against the oracle's marcher by tests/test_gpu_nerf.py / test_k1_lattice_model.py on the GPU."""
import numpy as np


def serial_walk(m, inside, skip, j0):
    """the scalar loop: returns (emitted mask, left_box, j_after)"""
    sm, j, done = 0, j0, False
    while j < 64:
        if not (inside >> j) & 1:
            done = True
            break
        rest = ~(m >> j) & ((1 << (64 - j)) - 1)          # bit 0 = 1 <=> point j is empty (bits past the chunk's end cleared)
        run = (rest & -rest).bit_length() - 1 if rest else 64 - j
        if run:
            ln = min(run, 64 - j)
            sm |= ((1 << ln) - 1) << j
            j += ln
        else:
            j += int(skip[j])
    return sm, done, j


def doubling_walk(m, inside, skip, j0):
    """the kernel's phase B: nxt per lane, visited set V by pointer doubling, <= 6 rounds"""
    lanes = np.arange(64)
    occ = np.array([(m >> int(l)) & 1 for l in lanes], bool)
    ins = np.array([(inside >> int(l)) & 1 for l in lanes], bool)
    nxt = np.where(occ, lanes + 1, np.where(ins, lanes + skip.astype(np.int64), 0x10000))
    if j0 >= 64:
        return 0, False, j0, 0
    V = np.zeros(64, bool); V[j0] = True
    J = nxt.copy()
    rounds = 0
    for _ in range(6):
        rounds += 1
        mark = V.copy()
        for l in np.nonzero(V)[0]:
            if J[l] < 64:
                mark[J[l]] = True
        if (mark == V).all():
            break
        V = mark
        J = np.where(J < 64, J[np.minimum(J, 63)], J)
    vmask = sum(1 << int(l) for l in np.nonzero(V)[0])
    sm = vmask & m
    left_box = (vmask & ~inside & ((1 << 64) - 1)) != 0
    jlast = int(np.nonzero(V)[0].max())
    return sm, left_box, int(nxt[jlast]), rounds


def _check(m, inside, skip, j0):
    sm_s, done_s, j_s = serial_walk(m, inside, skip, j0)
    sm_d, done_d, j_d, _ = doubling_walk(m, inside, skip, j0)
    assert sm_s == sm_d and done_s == done_d, (hex(m), hex(inside), j0)
    if not done_s:  # (after the march has left the box the landing point is never read)
        assert j_s == j_d, (hex(m), hex(inside), j0, j_s, j_d)


def test_orbit_by_pointer_doubling_is_the_serial_walk():
    rng = np.random.default_rng(11)
    full = (1 << 64) - 1
    for trial in range(4000):
        kind = trial % 8
        p_occ = (0.0, 0.03, 0.2, 0.5, 0.9, 1.0, 0.1, 0.3)[kind]
        m = sum(1 << i for i in range(64) if rng.random() < p_occ)
        n_in = 64 if kind % 3 else int(rng.integers(0, 65))      # the in-box points are a prefix of the lattice
        inside = (1 << n_in) - 1
        m &= inside                                              # occupied implies inside
        hi = (2, 2, 4, 9, 40, 3, 300, 70000)[kind]
        skip = rng.integers(1, hi, 64).astype(np.uint32)
        j0 = int(rng.integers(0, 64)) if trial % 5 else int(rng.integers(0, 200))
        _check(m, inside & full, skip, j0)


def test_orbit_corner_cases_and_round_count():
    ones = np.ones(64, np.uint32)
    full = (1 << 64) - 1
    # far field: every point empty, every jump one point long -- 64 scalar steps then, six doubling rounds now
    sm, done, j, rounds = doubling_walk(0, full, ones, 0)
    assert (sm, done, j, rounds) == (0, False, 64, 6) and serial_walk(0, full, ones, 0) == (0, False, 64)
    # all occupied: one run
    _check(full, full, ones, 0); _check(full, full, ones, 17)
    # a jump that lands exactly on the first point outside the box ends the march there
    skip = ones.copy(); skip[3] = 7
    _check(0b111, (1 << 10) - 1, skip, 0)
    # entry point behind the chunk: nothing visited, the landing point is the entry point
    assert doubling_walk(full, full, ones, 80)[:3] == (0, False, 80) and serial_walk(full, full, ones, 80) == (0, False, 80)
    # a jump far past the lattice (the kernel stores jump lengths clamped to 16 bits: any value >= 2048 ends the march alike)
    skip = ones.copy(); skip[5] = 65535
    _check(0b11111, full, skip, 0)
    # short orbits end early
    skip = np.full(64, 40, np.uint32)
    assert doubling_walk(0, full, skip, 0)[3] <= 2
