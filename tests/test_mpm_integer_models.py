"""Integer arithmetic the fused MPM kernel relies on, restated in Python integers / numpy and checked exhaustively enough to
pin the reasoning in pixie_amd/csrc/mpm.hip (no GPU needed):

  * pack_pair / unpack_pair (:533-539): two 32-bit two's-complement sums carried by ONE 64-bit integer add or subtract per
    contribution -- exact whatever the order and the mix of ds_add_u64 / ds_sub_u64, as long as each half's total fits 32 bits;
  * the tie rule of the packed scatter (:702-717): even nodes add rpi(x) (ties up), odd nodes subtract rpi(-x) (ties down);
  * the (mass, momentum_z) word, whose low half needs no borrow correction because masses are never negative;
  * the slot of a particle inside its block after re-binning (bin_local_order_kernel): round-robin over the 64 cells, a pure
    function of (cell, rank within the cell) -- a permutation, ordered by (rank, cell)."""
import math

import numpy as np

M64 = (1 << 64) - 1


def s32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def pack_pair(low, high):
    """(u64)(u32)low | ((u64)(u32)(high + (low >> 31)) << 32)  -- `low >> 31` is the arithmetic shift: 0 or -1"""
    return (low & 0xFFFFFFFF) | (((high + (-1 if low < 0 else 0)) & 0xFFFFFFFF) << 32)


def unpack_pair(w):
    low = s32(w)
    high = s32((w >> 32) - (-1 if low < 0 else 0))
    return low, high


def rpi(x):
    """v_cvt_rpi_i32_f32: floor(x + 0.5)"""
    return math.floor(x + 0.5)


def test_pack_pair_is_low_plus_high_shifted():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        lo, hi = (int(v) for v in rng.integers(-2 ** 31, 2 ** 31, 2))
        assert pack_pair(lo, hi) == (lo + (hi << 32)) & M64
        assert unpack_pair(pack_pair(lo, hi)) == (lo, hi)
    for lo, hi in ((0, 0), (-1, 0), (0, -1), (-1, -1), (-2 ** 31, 2 ** 31 - 1), (2 ** 31 - 1, -2 ** 31)):
        assert unpack_pair(pack_pair(lo, hi)) == (lo, hi)


def test_packed_sums_are_exact_under_any_mix_of_add_and_sub():
    rng = np.random.default_rng(1)
    for trial in range(200):
        n = int(rng.integers(1, 257))
        mag = int(rng.choice([1, 2 ** 8, 2 ** 22]))          # 256 contributions below 2^22 stay below 2^30
        lows = rng.integers(-mag, mag + 1, n)
        highs = rng.integers(-mag, mag + 1, n)
        sub = rng.integers(0, 2, n).astype(bool)              # odd nodes subtract the packed NEGATED contribution
        word = 0
        for lo, hi, s in zip(lows.tolist(), highs.tolist(), sub.tolist()):
            if s:
                word = (word - pack_pair(-lo, -hi)) & M64
            else:
                word = (word + pack_pair(lo, hi)) & M64
        assert unpack_pair(word) == (int(lows.sum()), int(highs.sum()))


def test_mass_word_needs_no_borrow_correction():
    """even nodes: word += (u32)m | (u32)z << 32 ; odd nodes: word -= pack_pair(rpi(-m), rpi(-z)); decode: plain halves.
    Valid because the mass total is never negative (the low half never borrows in the end)."""
    rng = np.random.default_rng(2)
    for trial in range(200):
        n = int(rng.integers(1, 257))
        m = rng.integers(0, 2 ** 22, n)                       # scaled masses: non-negative
        z = rng.integers(-2 ** 22, 2 ** 22, n)
        odd = rng.integers(0, 2, n).astype(bool)
        word = 0
        for mi, zi, o in zip(m.tolist(), z.tolist(), odd.tolist()):
            if o:
                word = (word - pack_pair(-mi, -zi)) & M64
            else:
                word = (word + ((mi & 0xFFFFFFFF) | ((zi & 0xFFFFFFFF) << 32))) & M64
        assert s32(word) == int(m.sum()) and s32(word >> 32) == int(z.sum())


def test_tie_rule_alternates_direction_and_is_unbiased():
    # exact ties: contributions with one fractional bit
    for k in range(-6, 7):
        x = k + 0.5
        up, down = rpi(x), -rpi(-x)
        assert up == k + 1 and down == k                      # even nodes round ties up, odd nodes down
    rng = np.random.default_rng(3)
    x = rng.integers(-2 ** 12, 2 ** 12, 100000) + 0.5         # nothing but ties
    even = np.floor(x + 0.5)
    odd = -np.floor(-x + 0.5)
    assert abs(float((even - x).mean()) - 0.5) < 1e-12 and abs(float((odd - x).mean()) + 0.5) < 1e-12
    # non-ties: both forms are round-to-nearest
    y = rng.normal(size=100000) * 1000.0
    y = y[np.abs(y - np.floor(y) - 0.5) > 1e-9]
    assert np.array_equal(np.floor(y + 0.5), -np.floor(-y + 0.5))


def test_round_robin_slot_is_a_permutation_ordered_by_rank_then_cell():
    """pos(c, r) = sum_k [ min(n_k, r) + (k < c and n_k > r) ]  over the 64 cells of a block (mpm.hip: bin_local_order_kernel)"""
    rng = np.random.default_rng(4)
    for trial in range(50):
        n = rng.integers(0, 12, 64)
        n[rng.integers(0, 64, 8)] = 0                         # some empty cells
        entries = [(c, r) for c in range(64) for r in range(int(n[c]))]
        pos = {}
        for c, r in entries:
            p = int(np.minimum(n, r).sum() + np.count_nonzero((np.arange(64) < c) & (n > r)))
            pos[(c, r)] = p
        assert sorted(pos.values()) == list(range(len(entries)))
        by_pos = sorted(entries, key=lambda e: pos[e])
        assert by_pos == sorted(entries, key=lambda e: (e[1], e[0]))   # round-robin: rank-major, cell-minor
        # consequence the kernels rely on: 64 consecutive slots of the first round sit in 64 different cells
        first_round = [c for c, r in by_pos[: int(np.count_nonzero(n))]]
        assert len(set(first_round)) == len(first_round)


def test_material_classes_come_first_and_a_single_class_keeps_the_old_order():
    """Round 5: slot = (particles of earlier material classes) + the round-robin position inside the class, counts per
    (class, cell) -- mpm.hip: bin_local_order_kernel.  A permutation ordered by (class, rank, cell); with one class present it is
    the formula above unchanged, so single-material scenes keep their layout (and their bits)."""
    rng = np.random.default_rng(7)
    for trial in range(30):
        classes = 1 if trial < 5 else int(rng.integers(2, 8))
        n = rng.integers(0, 6, (7, 64))
        n[classes:] = 0
        base = np.concatenate([[0], np.cumsum(n.sum(1))])[:7]
        entries = [(m, c, r) for m in range(7) for c in range(64) for r in range(int(n[m, c]))]
        pos = {(m, c, r): int(base[m] + np.minimum(n[m], r).sum() + np.count_nonzero((np.arange(64) < c) & (n[m] > r))) for m, c, r in entries}
        assert sorted(pos.values()) == list(range(len(entries)))
        by_pos = sorted(entries, key=lambda e: pos[e])
        assert by_pos == sorted(entries, key=lambda e: (e[0], e[2], e[1]))       # class-major, then rank, then cell
        if classes == 1:
            old = {(c, r): int(np.minimum(n[0], r).sum() + np.count_nonzero((np.arange(64) < c) & (n[0] > r))) for _, c, r in entries}
            assert all(pos[(0, c, r)] == old[(c, r)] for _, c, r in entries)


# ---- exact mode: 64-bit fixed point through the double adder (mpm.hip: to_fixed / from_fixed / scale_for) ----
K_MAGIC = 6755399441055744.0            # 1.5 * 2^52


def to_fixed(x32):
    return int(np.array([np.float64(np.float32(x32)) + K_MAGIC], dtype=np.float64).view(np.uint64)[0])


def from_fixed(v, inv_scale):
    hi = ((((v >> 32) & 0x7FFFF) ^ 0x40000) | 0x43300000) & 0xFFFFFFFF
    bits = np.array([(hi << 32) | (v & 0xFFFFFFFF)], dtype=np.uint64)
    d = float(bits.view(np.float64)[0]) - 5629499534213120.0        # 2^52 + 2^50
    return np.float32(d * float(inv_scale))


def test_double_magic_accumulates_exact_integers():
    rng = np.random.default_rng(5)
    for trial in range(100):
        n = int(rng.integers(1, 257))
        x = (rng.normal(size=n) * 2.0 ** rng.integers(10, 41)).astype(np.float32)
        x = np.clip(x, -(2.0 ** 42 - 2 ** 20), 2.0 ** 42 - 2 ** 20).astype(np.float32)   # every contribution scaled below 2^42
        word = 0
        for xi in x.tolist():
            word = (word + to_fixed(xi)) & M64          # ds_add_u64 of the raw bit patterns, any order
        exact = sum(int(np.rint(np.float64(xi))) for xi in x.tolist())   # the adder rounds to nearest-even, as rint does
        assert abs(exact) < 2 ** 50
        assert float(from_fixed(word, 1.0)) == float(np.float32(exact))
    # order independence is integer addition's; spot-check a permutation anyway
    x = (rng.normal(size=64) * 2.0 ** 30).astype(np.float32).tolist()
    a = 0
    for xi in x:
        a = (a + to_fixed(xi)) & M64
    b = 0
    for xi in reversed(x):
        b = (b + to_fixed(xi)) & M64
    assert a == b


def scale_for(bound, top):
    bits = int(np.array([bound], dtype=np.float32).view(np.uint32)[0])
    eb = ((bits >> 23) & 0xFF) - 127
    if not (bound > 0.0) or eb > 120:
        return np.float32(1.0)
    e = max(-80, min(120, top - eb))
    return np.array([(e + 127) << 23], dtype=np.uint32).view(np.float32)[0]


def pow2_reciprocal(s):
    bits = int(np.array([s], dtype=np.float32).view(np.uint32)[0])
    return np.array([(0x7F000000 - bits) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]


def test_scale_puts_the_bound_in_the_top_octave_and_inverts_exactly():
    rng = np.random.default_rng(6)
    for top in (29, 41):
        for _ in range(500):
            bound = np.float32(abs(rng.normal()) * 10.0 ** rng.integers(-20, 15) + 1e-30)
            s = scale_for(bound, top)
            m, e = math.frexp(float(s))
            assert m == 0.5                                       # a power of two
            if -80 < top - (math.frexp(float(bound))[1] - 1) < 120:
                assert 2.0 ** top <= float(bound) * float(s) < 2.0 ** (top + 1)
            assert float(pow2_reciprocal(s)) * float(s) == 1.0
    assert float(scale_for(np.float32(0.0), 29)) == 1.0 and float(scale_for(np.float32(np.inf), 29)) == 1.0
    assert float(scale_for(np.float32(np.nan), 41)) == 1.0
