"""Brute-force bounds on the oracle's heuristics (the upstream encoders are absent, so "how far
from the best possible block" is the independent yardstick available):

* ETC1: the TRUE optimum of a block -- every RGB444 pair (individual mode) and every RGB555 base
  with every legal delta (differential mode), both flips, all eight tables, best modifier per
  texel -- by exhaustive enumeration in numpy.  The oracle must never beat it (that would mean
  encoder and decoder disagree) and must stay within a stated distance of it per quality level.
* BC7 mode 6: every single-coordinate move of the emitted endpoints (+-1, +-2 on each 7-bit
  component, p-bit flips) with optimal re-indexing: the emitted block must be (nearly) a local
  optimum of the exact error.
* BC4: exhaustive 256 x 256 endpoint pairs lives in tests/test_oracle_bc15.py.
"""
import os

import numpy as np
import pytest
from scipy.ndimage import minimum_filter

import oracle_lib as O
from cuttlefish_amd import synth

MOD = np.array([[2, 8, -2, -8], [5, 17, -5, -17], [9, 29, -9, -29], [13, 42, -13, -42],
                [18, 60, -18, -60], [24, 80, -24, -80], [33, 106, -33, -106], [47, 183, -47, -183]])


def _half_errors(tex, bits):
    """tex (8, 3) -> err[n, n, n] (n = 2^bits): min over tables of sum over texels of min over
    the four modifiers, for every base colour of that precision (ETC1 specification)."""
    n = 1 << bits
    v = np.arange(n)
    ex = (v << 3 | v >> 2) if bits == 5 else (v << 4 | v)
    best = None
    for t in range(8):
        per = []
        for c in range(3):
            val = np.clip(ex[:, None] + MOD[t][None, :], 0, 255)
            per.append((val[:, :, None] - tex[None, None, :, c])**2)            # (n, 4, 8)
        tot = per[0][:, None, None] + per[1][None, :, None] + per[2][None, None, :]
        e = tot.min(axis=3).sum(axis=3)
        best = e if best is None else np.minimum(best, e)
    return best


def _etc1_optimum(blk):
    best = None
    for flip in (0, 1):
        h1, h2 = (blk[:, :2], blk[:, 2:]) if flip == 0 else (blk[:2], blk[2:])
        h1, h2 = h1.reshape(-1, 3), h2.reshape(-1, 3)
        e = int(_half_errors(h1, 4).min() + _half_errors(h2, 4).min())           # individual
        best = e if best is None else min(best, e)
        e1, e2 = _half_errors(h1, 5), _half_errors(h2, 5)                        # differential
        m = minimum_filter(e2, size=8, origin=0, mode="constant", cval=2**30)    # deltas -4 .. +3
        best = min(best, int((e1 + m).min()))
    return best


def test_etc1_against_the_exhaustive_optimum():
    img = synth.photo(48, 16, seed=21)
    img[..., 3] = 255
    opt = np.array([_etc1_optimum(img[by:by + 4, bx:bx + 4, :3].astype(np.int32))
                    for by in range(0, 16, 4) for bx in range(0, 48, 4)])
    gaps = []
    for q in range(5):
        dec = O.decode_etc(O.encode(img, 37, quality=q, threads=4), 37, 48, 16)
        e = ((dec[..., :3].astype(int) - img[..., :3])**2).reshape(4, 4, 12, 4, 3).sum(axis=(1, 3, 4)).reshape(-1)
        assert (e >= opt).all()                       # nothing decodes better than the optimum
        gaps.append(10*np.log10(e.sum()/opt.sum()))
    # measured on this strip: 0.55 / 0.55 / 0.10 / 0.09 / 0.09 dB
    assert gaps[0] < 1.0 and gaps[2] < 0.25 and gaps[3] < 0.2 and gaps[4] <= gaps[2] + 1e-9, gaps


W6 = np.array([0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64])


def _mode6_error(px, e0, e1):
    """exact SSE of a mode-6 endpoint pair (8-bit incl. p-bit) with the best index per texel"""
    pal = (e0[None, :]*(64 - W6[:, None]) + e1[None, :]*W6[:, None] + 32) >> 6          # (16, 4)
    return int(((px[:, None, :] - pal[None, :, :])**2).sum(-1).min(axis=1).sum())


def test_bc7_mode6_blocks_are_local_optima_of_the_exact_error():
    yy, xx = np.mgrid[0:64, 0:96].astype(np.float64)
    rgb = [128 + 100*np.sin(xx/17 + yy/31), 128 + 90*np.cos(xx/23 - yy/13), 128 + 80*np.sin(xx/11 + yy/19 + 1)]
    tot_e = tot_b = n6 = 0
    for alpha in (0*xx + 255, 128 + 100*np.sin(xx/29 + yy/9)):       # smooth content: where mode 6 wins
        img = np.stack(rgb + [alpha], -1).round().clip(0, 255).astype(np.uint8)
        blk = O.encode(img, 36, quality=3, threads=8).reshape(-1, 16)
        for i, b in enumerate(blk):
            v = int.from_bytes(bytes(b), "little")
            if (v & 0x7F) != 0x40:
                continue
            n6 += 1
            f = [(v >> (7 + 7*k)) & 0x7F for k in range(8)]            # r0 r1 g0 g1 b0 b1 a0 a1
            p0, p1 = (v >> 63) & 1, (v >> 64) & 1
            by, bx = divmod(i, 24)
            px = img[by*4:by*4 + 4, bx*4:bx*4 + 4].reshape(16, 4).astype(np.int64)

            def err(ff, q0, q1):
                e0 = np.array([(ff[2*c] << 1) | q0 for c in range(4)])
                e1 = np.array([(ff[2*c + 1] << 1) | q1 for c in range(4)])
                return _mode6_error(px, e0, e1)
            base = err(f, p0, p1)
            best = base
            for k in range(8):
                for d in (-2, -1, 1, 2):
                    g = list(f)
                    g[k] = min(127, max(0, g[k] + d))
                    best = min(best, err(g, p0, p1))
            best = min(best, err(f, p0 ^ 1, p1), err(f, p0, p1 ^ 1), err(f, p0 ^ 1, p1 ^ 1))
            tot_e += base
            tot_b += best
    assert n6 >= 40
    gap = 10*np.log10(max(tot_e, 1)/max(tot_b, 1))
    print("mode-6 blocks %d, gap to the best single-coordinate move %.3f dB" % (n6, gap))
    assert gap < 0.3, gap      # measured 0.086 dB: one more endpoint move would gain that little


def _quality_tables():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "quality_tables", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "quality_tables.py"))
    qt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(qt)
    return qt


def test_bc7_ladder_on_real_photograph_blocks():
    """The ladder is held to blocks of REAL photographs (tests/golden/real_blocks.npz; round-4 VERDICT: the synthetic
    tile never visits BC7 modes 1 / 6 / 0 / 2 / 4, real pictures live in them).  Gap to cfo_bc7_wide_search on the first
    384 opaque / 384 alpha-carrying blocks -- the whole set (4 096 / 1 024 blocks) is profiles/r05_quality_tables.md:
    Normal 0.18 / 0.08 dB, High 0.08 / 0.04, Highest 0.06 / 0.02; round 4's ladder on the same blocks: 0.60 / 1.00,
    0.55 / 0.97, 0.11 / 0.02."""
    qt = _quality_tables()
    rows = qt.bc7_gap(384, np.random.default_rng(7), kind="real")
    for label, ps in rows:
        assert ps[5] >= max(ps[:5]) - 1e-9, (label, ps)        # the wide search is a bound
        assert ps[5] - ps[2] <= 0.25, (label, ps)              # Normal
        assert ps[5] - ps[3] <= 0.12, (label, ps)              # High
        assert ps[5] - ps[4] <= 0.10, (label, ps)              # Highest
        assert all(ps[q + 1] >= ps[q] - 1e-9 for q in range(4)), (label, ps)
    # High is a different, deeper search than Normal (bc7enc: uber level 4 against 1, S3tcConverter.cpp:193,204)
    assert rows[0][1][3] - rows[0][1][2] >= 0.05, rows[0]


def test_bc7_and_bc6h_ladders_stay_close_to_the_wide_search():
    """Gap of the quality ladder to the wide search (cfo_bc7_wide_search: every mode x partition x rotation x
    selector, every fit solved by a steepest descent on the quantised endpoint grid under the exact error from
    several starts -- an endpoint solver that is not the encoder's; cfo_bc6h_wide_search: all 33 candidates with
    12 refit rounds) on the SYNTHETIC tile (the real-photograph blocks: the test above).  2 048 blocks: Normal
    0.09 / 0.05 dB, High 0.03 / 0.02, Highest 0.03 / 0.01 on opaque / alpha-carrying content; here 192 blocks keep
    the CPU suite fast, and the thresholds leave room for what a sample of that size moves (+- 0.02 dB)."""
    qt = _quality_tables()
    rng = np.random.default_rng(7)
    for label, ps in qt.bc7_gap(192, rng):
        assert ps[5] >= max(ps[:5]) - 1e-9, (label, ps)        # the wide search is a bound
        assert ps[5] - ps[2] <= 0.10, (label, ps)              # Normal within 0.10 dB of it (north_star's tolerance)
        assert ps[5] - ps[3] <= 0.06, (label, ps)
        assert ps[5] - ps[4] <= 0.06, (label, ps)
        assert all(ps[q + 1] >= ps[q] - 1e-9 for q in range(3)) and ps[4] >= ps[3] - 0.01, (label, ps)
    ps = qt.bc6h_gap(96, rng)
    assert ps[5] >= ps[4] - 0.02 and ps[5] - ps[2] <= 0.45, ps


def test_etc_ladders_against_the_true_optimum_of_every_mode():
    """cfo_etc_true_optimum: individual / differential by enumeration of every base colour, planar channel by
    channel over every (O, H, V), T and H over every distance and every base colour of the box that must hold an
    optimum.  Checked here against the numpy enumerator above (ETC1) and against the decoder (the block it
    returns decodes to the error it claims); then the ladders: nothing the encoder emits beats the optimum, and
    ETC2 RGB Normal stays within 0.3 dB of it (192 sampled blocks: 0.26 dB on 1 024, profiles/r04_quality_tables.md;
    nine tenths of what is left sits in the few T / H blocks of a sample, so the sample size moves it)."""
    import ctypes
    import importlib.util
    L = O.lib()
    L.cfo_etc_true_optimum.restype = ctypes.c_uint32
    L.cfo_etc_true_optimum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    img = synth.photo(48, 8, seed=21)
    img[..., 3] = 255
    for by in range(0, 8, 4):
        for bx in range(0, 48, 4):
            blk = np.ascontiguousarray(img[by:by + 4, bx:bx + 4])
            out = np.zeros(8, np.uint8)
            e1 = L.cfo_etc_true_optimum(blk.ctypes.data, 0, out.ctypes.data)
            assert e1 == _etc1_optimum(blk[..., :3].astype(np.int32))
            assert e1 == int(((O.decode_etc(out, 37, 4, 4)[..., :3].astype(int) - blk[..., :3])**2).sum())
            e2 = L.cfo_etc_true_optimum(blk.ctypes.data, 1, out.ctypes.data)
            assert e2 <= e1
            assert e2 == int(((O.decode_etc(out, 38, 4, 4)[..., :3].astype(int) - blk[..., :3])**2).sum())
    spec = importlib.util.spec_from_file_location(
        "quality_tables", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "quality_tables.py"))
    qt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(qt)
    ps, opt = qt.etc_gap(192, np.random.default_rng(11), 38)          # asserts e >= optimum per block itself
    assert opt - ps[2] <= 0.30 and opt - ps[4] <= 0.20, (ps, opt)
    assert all(ps[q + 1] >= ps[q] - 1e-9 for q in range(4)), ps


def test_etc_and_astc_ladders_on_real_photograph_blocks():
    """Round 5: the ETC and ASTC ladders are held to blocks of REAL photographs (tests/golden/real_blocks.npz) like
    BC7's.  ETC2 RGB against the TRUE optimum: round 4's ladder sat 0.51 / 0.42 / 0.34 dB under it on the 4 096 blocks,
    this round's 0.24 / 0.18 / 0.17 (tools/etc_lab.py; here the first 384).  ASTC 6x6 against cfo_astc_wide_search:
    round 4 1.03 / 0.95 / 0.79 on 256 blocks, mid-round 0.85 / 0.59 / 0.55, now 0.58 / 0.34 / 0.29 -- and High is a deeper search than Normal
    (AstcConverter.cpp:184,187: MEDIUM against THOROUGH), not the same one with more seeds ranked."""
    qt = _quality_tables()
    ps, opt = qt.etc_gap(384, np.random.default_rng(1), 38, kind="real")
    assert opt - ps[2] <= 0.25 and opt - ps[3] <= 0.20 and opt - ps[4] <= 0.20, (ps, opt)
    assert all(ps[q + 1] >= ps[q] - 1e-9 for q in range(4)), ps
    ps, wide = qt.astc_gap(256, np.random.default_rng(1), 6, 6, kind="real")
    assert wide >= max(ps) - 1e-9, (ps, wide)
    # (late round 5: partition seeds ranked by line-fit error, refinement rounds on the 16 best results:
    #  0.58 / 0.34 / 0.29 on these 256 blocks, 0.54 / 0.33 / 0.28 on the 768 of profiles/r05_quality_tables.md)
    # (round 6: a refinement round takes a step towards the least-squares grid (oracle phase_b), and so does every evaluation
    #  of the wide search: the BOUND rose from 36.24 to 36.92 dB on these blocks, Normal / High from 35.84 / 35.90 to
    #  36.25 / 36.48 with ONE round where they had two -- both above round 5's bound -- 0.67 / 0.45 / 0.29 under the new one)
    assert wide - ps[3] <= 0.50 and wide - ps[2] <= 0.72 and wide - ps[4] <= 0.33, (ps, wide)
    assert ps[2] >= 36.20 and ps[3] >= 36.40 and wide >= 36.85, (ps, wide)
    assert ps[3] - ps[2] >= 0.10, ps                       # High above Normal by a measurable step (round 6: Normal's config ranking gained 0.06 dB)
    assert all(ps[q + 1] >= ps[q] - 1e-9 for q in range(4)), ps
    # 4x4: the wide search takes two partitions in a third of these blocks, and the seed it takes is seldom the one
    # a clustering of the texels points at (0.55 dB at High with the cluster-overlap ranking; 0.24 here, 0.18 on 768)
    ps, wide = qt.astc_gap(256, np.random.default_rng(1), 4, 4, kind="real")
    assert wide >= max(ps) - 1e-9, (ps, wide)
    assert wide - ps[3] <= 0.35 and wide - ps[2] <= 0.60, (ps, wide)      # (round 6: against the tighter bound, one round)
    assert all(ps[q + 1] >= ps[q] - 1e-9 for q in range(4)), ps


def test_astc_ladder_against_the_wide_search():
    """cfo_astc_wide_search (LDR): one partition, a second weight plane on every component, every canonical seed of
    the 2 / 3 / 4-partition tables, EVERY legal block mode of the class (the census tables: no cut at the 200
    best-scored, no kernel column limit), every endpoint-mode family forced and measured exactly, the best triples
    iterated (weights re-projected on the decoded endpoints); the better of a run over the census tables and one
    over the encoder's own lists.  The block it writes decodes (through the decoder pinned to Mesa's) to the error
    it claims, nothing the ladder emits beats it, and the gap of High stays inside what round 4 measured
    (profiles/r04_quality_tables.md: 6x6 High 0.54 / 0.46 dB on 512 opaque / alpha-carrying blocks, 4x4 0.20 /
    0.46, 8x8 0.51 / 0.57 -- the weights <-> endpoints iteration and the partition seeds of the large footprints
    are where the ladder leaves most: DESIGN section 7)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "quality_tables", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "quality_tables.py"))
    qt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(qt)
    for alpha in (False, True):
        ps, wide = qt.astc_gap(96, np.random.default_rng(4), 6, 6, alpha)
        assert wide >= max(ps) - 1e-9, (alpha, ps, wide)
        assert wide - ps[3] <= 0.60, (alpha, ps, wide)
    ps, wide = qt.astc_gap(48, np.random.default_rng(5), 4, 4, False)
    assert wide >= max(ps) - 1e-9 and wide - ps[3] <= 0.40, (ps, wide)
    # the full-resolution grid: a two-colour 8x8 block with a ragged edge is what the 8x8 x 2-level config is for;
    # the encoder lists it since round 4 (76-row lane column) and must come close to the bound on such a block
    rng = np.random.default_rng(6)
    mask = rng.random((8, 8)) < 0.5
    blk = np.where(mask[..., None], np.array([230, 40, 30, 255], np.uint8), np.array([20, 60, 200, 255], np.uint8))
    blk = np.ascontiguousarray(blk.astype(np.uint8))
    import ctypes
    import oracle_lib as O
    from cuttlefish_amd import Format
    L = O.lib()
    L.cfo_astc_wide_search.restype = ctypes.c_uint64
    L.cfo_astc_wide_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = np.zeros(16, np.uint8)
    L.cfo_astc_wide_search(blk.ctypes.data, 8, 8, 0, out.ctypes.data)
    dec_w, _ = O.decode_astc(out, int(Format.ASTC_8x8), 8, 8)
    dec_e, _ = O.decode_astc(O.encode(blk, int(Format.ASTC_8x8), quality=2, threads=1), int(Format.ASTC_8x8), 8, 8)
    ew = float(((dec_w.astype(np.int64) - blk)[..., :3]**2).sum())
    ee = float(((dec_e.astype(np.int64) - blk)[..., :3]**2).sum())
    assert ew <= 64*3*4 and ee <= 64*3*4, (ew, ee)      # both within +-2 per channel on average: the exact grid


def test_bc1_and_bc3_colour_blocks_against_the_true_optimum():
    """tests/golden/bc1_optimum.npz: per block the smallest error ANY BC1 colour block can reach -- all 2^32 RGB565
    endpoint pairs walked on the GPU by tools/bounds/bc1_optimum.hip (tools/bc1_bound.py, round 5; 8 s per 512
    blocks on an MI355X): E4 over the four-colour palettes (BC2 / BC3 colour, BC1 in c0 > c1 order), E3 over the
    three-colour + black ones.  First 512 opaque blocks of the real-photograph fixture and 512 sampled blocks of the
    synthetic tile.  Nothing the encoder emits beats it (encoder and decoder agree), and the ladder: real BC1 0.054 /
    0.019 / 0.011 dB at Normal / High / Highest, BC3 colour 0.098 / 0.019 / 0.011; synthetic 0.23 / 0.09 / 0.09."""
    import real_lib as R
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bc1_optimum.npz"))
    n = int(fx["blocks"])
    qt = _quality_tables()
    img = synth.photo(512, 512, seed=21)
    img[..., 3] = 255
    rng = np.random.default_rng(20260929)
    ys = rng.integers(0, 128, n) * 4
    xs = rng.integers(0, 128, n) * 4
    sets = {"real": R.blocks4(n), "synth": np.stack([img[y:y + 4, x:x + 4] for y, x in zip(ys, xs)])}
    if "real_b_e4" in fx.files:          # round 6: the held-out photograph group (BC1 0.04 / 0.01 / 0.01 dB, BC3 0.06 / 0.01 / 0.01)
        sets["real_b"] = R.blocks4(n, group="b")
    for name, blocks in sets.items():
        strip = R.strip(np.ascontiguousarray(blocks))
        e4, e3 = fx[name + "_e4"].astype(np.float64), fx[name + "_e3"].astype(np.float64)
        for fmt, bound in ((29, np.minimum(e4, e3)), (32, e4)):
            gaps = []
            for q in range(5):
                dec = O.decode(O.encode(strip, fmt, quality=q, threads=8), fmt, 4 * n, 4)
                e = ((dec[..., :3].astype(np.int64) - strip[..., :3]) ** 2).reshape(4, n, 4, 3).sum(axis=(0, 2, 3))
                assert (e >= bound).all(), (name, fmt, q)
                gaps.append(10 * np.log10(e.sum() / bound.sum()))
            assert gaps[2] <= 0.30 and gaps[3] <= 0.15 and gaps[4] <= 0.15, (name, fmt, gaps)
            assert all(gaps[q + 1] <= gaps[q] + 1e-9 for q in range(4)), (name, fmt, gaps)
    assert qt is not None


def test_ladders_on_both_photograph_groups_pooled_and_worst_image():
    """Round-5 VERDICT item 2: a pooled figure over one set of photographs hid a six-fold spread between pictures, and the
    ladders had only ever been measured on the pictures they were balanced on.  tests/golden/real_blocks.npz now holds a
    second, HELD-OUT group (b: hubble_deep_field, ihc, retina, motorcycle_right, color -- no tool reads it), and
    tools/quality_real.py measures every family per image.  Here: a sample of both groups, pooled AND worst image, and
    the alpha-carrying rows (ETC2 RGBA8 against the true optimum of its two halves, ASTC 6x6 with alpha).  The whole
    table is profiles/r06_quality_real.md.  What the numbers say plainly: BC7 generalises (High 0.06 .. 0.09 dB pooled on
    either group); ETC2 RGB held its group-a figure less well on group b (0.37 against 0.22 dB at Normal on 1 024 blocks,
    hubble_deep_field 0.50) until the T / H seed got two Lloyd steps (now 0.30 against 0.20; this sample of 256: 0.36 / 0.11); ASTC 6x6 High is 0.33 / 0.39 dB pooled and 0.6 .. 0.9 (a: motorcycle; 32 .. 96 blocks of it) / 1.6 dB (b: "color", a near-flat
    graphic at 53 dB) on the worst picture -- smooth pictures want small weight grids with many levels, which the
    config ranking seldom puts among its eight; ASTC with alpha was 1.0 .. 1.4 dB out at every level with round 5's lists and
    is 0.5 .. 0.7 with lists ranked on textured alpha."""
    import importlib.util
    import real_lib as R
    spec = importlib.util.spec_from_file_location(
        "quality_real", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "quality_real.py"))
    Q = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(Q)

    def gaps(rows):
        pooled = rows[0][2]
        worst = max(rows[1:], key=lambda r: r[2][5] - r[2][3]) if len(rows) > 1 else None
        g = lambda ps: [ps[5] - ps[q] for q in (2, 3, 4)]
        return g(pooled), (worst[0], g(worst[2])) if worst else None
    lim = {   # (pooled Normal, pooled High, pooled Highest, worst-image High)
        ("bc7", "a"): (0.25, 0.12, 0.10, 0.30), ("bc7", "b"): (0.25, 0.12, 0.10, 0.35),
        ("etc2", "a"): (0.25, 0.20, 0.20, 0.35), ("etc2", "b"): (0.42, 0.38, 0.33, 0.50),
        ("astc6", "a"): (0.72, 0.50, 0.36, 1.10), ("astc6", "b"): (0.85, 0.70, 0.58, 1.75)}      # (astc6: against the round-6 bound, which rose 0.65 dB with the least-squares step)
    for group in ("a", "b"):
        names = R.image_names(group)
        b4 = R.blocks4(256, group=group)
        img4 = np.arange(len(b4)) % len(names)
        b6 = R.blocks(6, 6, 256, group=group)
        img6 = np.arange(len(b6)) % len(names)
        fam = {"bc7": Q.rows_of(*Q.bc7_sse(b4), img4, names), "etc2": Q.rows_of(*Q.etc_sse(b4, 38), img4, names),
               "astc6": Q.rows_of(*Q.astc_sse(b6, 6, 6), img6, names)}
        for key, rows in fam.items():
            pooled, worst = gaps(rows)
            ln, lh, lhh, lw = lim[(key, group)]
            assert pooled[0] <= ln and pooled[1] <= lh and pooled[2] <= lhh, (key, group, pooled)
            assert worst[1][1] <= lw, (key, group, worst)
            assert pooled[1] <= pooled[0] + 1e-9 and pooled[2] <= pooled[1] + 0.01, (key, group, pooled)
    ba = R.blocks4(256, alpha=True)
    pooled, _ = gaps(Q.rows_of(*Q.etc2_rgba8_sse(ba), None, None))          # asserts e >= optimum per block itself
    assert pooled[0] <= 0.25 and pooled[2] <= 0.15, pooled
    pooled, _ = gaps(Q.rows_of(*Q.astc_sse(R.blocks_alpha(6, 6, 128), 6, 6, True), None, None))
    assert pooled[1] <= 0.80 and pooled[1] <= pooled[0] + 1e-9, pooled      # (0.50 against round 5's bound; 0.70 against round 6's, which takes the least-squares step in every evaluation)


def test_eac_true_optimum_is_a_bound_and_decodes_to_what_it_claims():
    """cfo_eac_true_optimum (round 6): every base x multiplier x table of an EAC block -- the bound of the EAC alpha rows.
    The block it writes decodes to the error it returns, and nothing the search emits at any level beats it."""
    import ctypes
    L = O.lib()
    L.cfo_eac_true_optimum.restype = ctypes.c_uint32
    L.cfo_eac_true_optimum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.cfo_decode_eac.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.cfo_eac_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(3)
    for case in range(24):
        if case % 3 == 0:
            v = rng.integers(0, 256, 16)
        elif case % 3 == 1:
            v = np.clip(128 + (np.arange(16) - 8) * int(rng.integers(1, 12)) + rng.integers(-3, 4, 16), 0, 255)
        else:
            v = np.where(rng.random(16) < 0.5, int(rng.integers(0, 80)), int(rng.integers(170, 256)))
        v = np.ascontiguousarray(v.astype(np.int32))
        out, dec = np.zeros(8, np.uint8), np.zeros(16, np.int32)
        e = L.cfo_eac_true_optimum(v.ctypes.data, 0, out.ctypes.data)
        L.cfo_decode_eac(out.ctypes.data, 0, dec.ctypes.data)
        assert e == int(((dec - v) ** 2).sum())
        for R_ in (1, 2, 4):
            L.cfo_eac_search(v.ctypes.data, 0, 0xFFFF, R_, out.ctypes.data)
            L.cfo_decode_eac(out.ctypes.data, 0, dec.ctypes.data)
            assert int(((dec - v) ** 2).sum()) >= e
