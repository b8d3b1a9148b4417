"""A CPU model of the 64-rows-per-wave forward's key walk (fa_fwd_w64.hip: block tile range, per-wave idle | active | idle iteration ranges,
plain vs masked iterations, the per-lane visibility bitmap of set_mask) in both directions -- ascending (plain attention) and descending (the
causal-ALiBi variant, which starts at the diagonal and whose drain iteration scores the tile BELOW the lowest one, the zero tile "left of key 0"
when the block's range starts at key 0).  Checked against the mask definition the oracle uses: every visible (query, key) pair is scored as
visible exactly once, no invisible pair ever is -- keys < 0 and >= seqlen_k included.  The model restates the kernel's integer arithmetic line by
line (names as in the source), so a change there has to be made here too; it exists because the descending walk's first version passed every
causal test and counted phantom keys under a left window (the per-lane lower limit went negative; profiles/r03_fwd_schedules.txt)."""
import numpy as np
import pytest

BM, BN, NW, QB = 256, 64, 4, 2


def acc_row(r, hi):   # fa_device.h: row of accumulator element r in the 32x32 MFMA layout
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def walk_counts(sq, sk, wl, wr, desc, peel=False):
    """Times each (query, key) pair is scored as visible; keys are offset by PAD so that keys < 0 and >= sk have a slot.
    peel (round 5, plain attention, ascending): the wave's LAST iteration issues no score chain at all (fast_step MODE 3 / 4) and its FIRST one no P.V -- the
    model then scores nothing in iteration u_last and checks that the two are different iterations."""
    PAD = 2 * BN
    cnt = np.zeros((sq, sk + 2 * PAD), dtype=np.int32)
    shift = sk - sq
    for m0 in range(0, sq, BM):
        blk_last = min(m0 + BM, sq) - 1
        kmax, kmin = sk - 1, 0
        if wr >= 0: kmax = min(kmax, blk_last + shift + wr)
        if wl >= 0: kmin = max(0, m0 + shift - wl)
        n_min = kmin // BN
        n_max = (kmax // BN + 1) if kmax >= kmin else n_min
        n_tiles = n_max - n_min
        n_steps = 2 * n_tiles
        key_base = n_min * BN
        tile_of = (lambda u: n_tiles - 1 - u) if desc else (lambda u: u)
        step_key = lambda i: key_base + BN * tile_of(i >> 1) + 32 * (i & 1)
        if n_tiles <= 0:
            continue
        for wave in range(NW):
            w_row0 = m0 + wave * 64
            w_row1 = min(w_row0 + 63, sq - 1)
            if not (w_row0 < sq):
                continue
            w_kmax = min(sk - 1, w_row1 + shift + wr) if wr >= 0 else sk - 1
            w_kmin = max(0, w_row0 + shift - wl) if wl >= 0 else 0
            w_full_hi = min(sk - 1, w_row0 + shift + wr) if wr >= 0 else sk - 1
            w_full_lo = (w_row1 + shift - wl) if wl >= 0 else 0
            a_lo = max(0, (w_kmin - key_base) >> 5)
            a_hi = min(n_steps - 1, (w_kmax - key_base) >> 5)
            if a_hi < a_lo:
                continue
            u_first, u_last = a_lo >> 1, min(n_tiles, (a_hi + 2) >> 1)
            f_lo = max(0, (w_full_lo - key_base + 31) >> 5)
            f_hi = (w_full_hi - 31 - key_base) >> 5
            p_lo, p_hi = max(u_first, (f_lo + 1) >> 1), min(u_last, (f_hi - 1) >> 1)
            if desc:
                t_lo, t_hi, pl_t, ph_t = a_lo >> 1, a_hi >> 1, (f_lo + 1) >> 1, (f_hi - 1) >> 1
                u_first, u_last = n_tiles - 1 - t_hi, n_tiles - t_lo
                p_lo, p_hi = max(u_first, n_tiles - 1 - ph_t), min(u_last, n_tiles - 1 - pl_t)
            if p_hi < p_lo:
                p_lo, p_hi = u_last + 1, u_last
            if peel:
                assert not desc and u_last > (u_first & ~1), (u_first, u_last)   # the peeled first and last iteration are two iterations
            for u in range(u_first & ~1, u_last + 1):
                if peel and u == u_last:
                    continue   # (no score chain in the last iteration: whatever it would have scored must have been invisible)
                masked = not (p_lo <= u <= p_hi)
                for i_step in (2 * u, 2 * u + 1):
                    k0m = step_key(i_step)
                    for qb in range(QB):
                        for qi in range(32):
                            my_row = w_row0 + 32 * qb + qi
                            if my_row >= sq:
                                continue   # (rows past the end are computed and never stored)
                            lim_hi = min(sk - 1, my_row + shift + wr) if wr >= 0 else sk - 1
                            lim_lo = max(0, my_row + shift - wl) if wl >= 0 else 0
                            for hi in range(2):
                                if masked:
                                    rel_hi, rel_lo = min(lim_hi - k0m - 4 * hi, 31), max(lim_lo - k0m - 4 * hi, 0)
                                    ones = 0xffffffff if rel_hi - rel_lo >= 31 else ((2 << ((rel_hi - rel_lo) & 31)) - 1) & 0xffffffff
                                    bits = ((ones << (rel_lo & 31)) & 0xffffffff) if rel_hi >= rel_lo else 0
                                else:
                                    bits = 0xffffffff
                                for r in range(16):
                                    if (bits >> acc_row(r, 0)) & 1:
                                        cnt[my_row, k0m + 4 * hi + acc_row(r, 0) + PAD] += 1
    return cnt, PAD


def visible(sq, sk, wl, wr, pad):
    i = np.arange(sq)[:, None] + (sk - sq)
    j = np.arange(-pad, sk + pad)[None, :]
    ok = np.broadcast_to((j >= 0) & (j < sk), (sq, j.shape[1])).copy()
    if wr >= 0: ok &= j <= i + wr
    if wl >= 0: ok &= j >= i - wl
    return ok.astype(np.int32)


SHAPES = [(256, 256), (300, 300), (64, 200), (257, 513), (512, 384), (70, 70), (1, 130), (320, 1)]


@pytest.mark.parametrize("sq,sk", SHAPES)
@pytest.mark.parametrize("wl,wr", [(-1, -1), (-1, 0), (100, 0), (300, 0), (0, 0), (-1, 17), (40, 25), (64, -1), (5, 200)])
def test_ascending_walk_scores_each_visible_pair_once(sq, sk, wl, wr):
    cnt, pad = walk_counts(sq, sk, wl, wr, desc=False)
    assert np.array_equal(cnt, visible(sq, sk, wl, wr, pad))
    cnt, pad = walk_counts(sq, sk, wl, wr, desc=False, peel=True)   # ... and with the last iteration's chains not issued (round 5)
    assert np.array_equal(cnt, visible(sq, sk, wl, wr, pad))


@pytest.mark.parametrize("sq,sk", SHAPES)
@pytest.mark.parametrize("wl", [-1, 0, 31, 100, 300, 1000])
def test_descending_walk_scores_each_visible_pair_once(sq, sk, wl):
    """The causal-ALiBi variant's domain: right bound on the diagonal (wr = 0), any left bound."""
    cnt, pad = walk_counts(sq, sk, wl, 0, desc=True)
    assert np.array_equal(cnt, visible(sq, sk, wl, 0, pad))


def test_random_shapes_and_windows_both_directions():
    """Seeded sweep: lengths around the 256-row block / 64-key tile edges, windows from 0 to beyond the sequence."""
    rng = np.random.default_rng(7)
    edge = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 700]
    for _ in range(80):
        sq, sk = int(rng.choice(edge)), int(rng.choice(edge))
        wl = int(rng.choice([-1, 0, 1, 31, 32, 63, 64, 65, 200, 1000]))
        wr = int(rng.choice([-1, 0, 1, 31, 64, 200]))
        cnt, pad = walk_counts(sq, sk, wl, wr, desc=False)
        assert np.array_equal(cnt, visible(sq, sk, wl, wr, pad)), (sq, sk, wl, wr, "ascending")
        cnt, pad = walk_counts(sq, sk, wl, wr, desc=False, peel=True)
        assert np.array_equal(cnt, visible(sq, sk, wl, wr, pad)), (sq, sk, wl, wr, "ascending, peeled")
        cnt, pad = walk_counts(sq, sk, wl, 0, desc=True)
        assert np.array_equal(cnt, visible(sq, sk, wl, 0, pad)), (sq, sk, wl, 0, "descending")
