"""The PAIRS image (pgx_image.h: two-step FM index next to the dense2 image) against a brute-force construction from the
BWT: c2(p) = BWT[LF(p)], pair counts per 128 positions, special runs, the 2-step base table.  CPU tier: the host builder only."""
import os

import numpy as np
import pytest

import pgx_ffi as P
import pgx_workload as W
from image_emu import Consts, PairsLayout

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NUC = {10: 0, ord("A"): 1, ord("C"): 2, ord("G"): 3, ord("N"): 4, ord("T"): 5}
TWO = np.array([-1, 0, 1, 2, -1, 3])


@pytest.fixture(params=["96", "64"])
def psyms(request, monkeypatch):
    """both strides of the PAIRS image (pgx_image.h: blocks of 96 positions that tile the BWT, or one every 64 positions): the host builder
    takes PGX_PAIRS_STRIDE"""
    monkeypatch.setenv("PGX_PAIRS_STRIDE", request.param)
    return int(request.param)


def _bwt_codes(rl_path):
    sym, ln = W.read_rlbwt_runs(rl_path)
    return np.repeat(np.array([NUC[int(s)] for s in sym], dtype=np.int64), ln.astype(np.int64))


def _check(idx, bw, syms=None, ext=True):
    c = Consts(idx.image_view(6))
    assert c.has_pairs == 1 and idx.info().image_pairs == 1 and idx.info().image_kind == P.IMAGE_DENSE2
    n = len(bw)
    tot = np.bincount(bw, minlength=6)
    trueC = np.concatenate([[0], np.cumsum(tot)])
    # LF and the second symbol
    occ = np.zeros(n, dtype=np.int64)
    for code in range(6):
        m = bw == code
        occ[m] = np.arange(int(m.sum()))
    lf = trueC[bw] + occ
    y = TWO[bw]
    x = np.where(y >= 0, TWO[bw[np.minimum(lf, n - 1)]], -1)
    special = (y < 0) | (x < 0)
    pair = np.where(special, -1, 4 * y + x)
    blocks = idx.image_view(20).reshape(-1, 32)
    L = PairsLayout(c)
    B, S = L.syms, L.stride
    assert idx.info().pairs_stride == S and (syms is None or S == syms)
    nb = n // S + 1
    assert len(blocks) == nb
    # special runs (statistics) and, per first symbol, the positions whose second symbol is special: what the pair counts do not see of it
    starts = np.flatnonzero(special & ~np.concatenate([[False], special[:-1]]))
    assert c.pair_runs == len(starts)
    half = [np.concatenate([[0], np.cumsum((y == yy) & (x < 0))]) for yy in range(4)]  # half[yy][p] = such positions before p
    n_special = np.concatenate([[0], np.cumsum(special)])
    cum = np.zeros(16, dtype=np.int64)
    for b in range(nb):
        s0, s1 = S * b, min(S * b + B, n)
        h = blocks[b]
        assert [L.pair_before(h, i >> 2, i & 3) for i in range(16)] == list(cum), b
        flag = bool(special[s0:s1].any())
        assert L.flag(h) == int(h[16]) >> 31 == (1 if flag else 0), (b, hex(int(h[16])), flag)
        assert [L.half_before(h, yy) for yy in range(4)] == [int(half[yy][s0]) for yy in range(4)], b
        assert [L.rank_before(h, yy) for yy in range(4)] == [int((y[:s0] == yy).sum()) for yy in range(4)], b
        # every position before a block is a regular pair or a special position; rank of symbol y = its row sum + its half-special count
        assert s0 - int(cum.sum()) == int(n_special[s0])
        for yy in range(4):
            assert int(cum[4 * yy:4 * yy + 4].sum()) + int(half[yy][s0]) == int((y[:s0] == yy).sum())
        for i in range(s1 - s0):
            pv = int(pair[s0 + i])
            bits = [(L.plane_word(h, pl, i >> 5) >> (i & 31)) & 1 for pl in range(4)]
            if pv >= 0:
                assert bits == [(pv >> 2) & 1, (pv >> 3) & 1, pv & 1, (pv >> 1) & 1], (b, i)
            else:
                assert bits == [0, 0, 0, 0]
        for i in range(s1 - s0, B):
            assert all(((L.plane_word(h, pl, i >> 5) >> (i & 31)) & 1) == 0 for pl in range(4))
        # run continuation: how far the pair / the first symbol of the block's last position goes on behind the block (at most 255)
        e0 = s0 + B
        xp = x1 = 0
        if ext and e0 < n and pair[e0 - 1] >= 0:
            while xp < 255 and e0 + xp < n and pair[e0 + xp] == pair[e0 - 1]:
                xp += 1
            while x1 < 255 and e0 + x1 < n and y[e0 + x1] == y[e0 - 1]:
                x1 += 1
        assert (L.ext_pair(h), L.ext_first(h)) == (xp, x1), (b, L.ext_pair(h), L.ext_first(h), xp, x1)
        assert int(h[16]) & 0x7F000000 == 0 and int(h[19]) >> 24 == 0
        adv = pair[s0:min(s0 + S, n)]  # the counts move on by the stride
        cum += np.bincount(adv[adv >= 0], minlength=16)
    for yy, code in enumerate((1, 2, 3, 5)):
        exp = np.bincount(bw[:trueC[code]], minlength=6)
        assert [int(v) for v in c.pair_t2[8 * yy:8 * yy + 6]] == [int(v) for v in exp]


@pytest.mark.parametrize("mode", [P.MODE_COMPAT, P.MODE_STRICT])
def test_pairs_image_of_a_text_with_N_runs(workdir, mode, psyms):
    rng = np.random.default_rng(5)
    seqs = []
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    for h in range(5):
        s = list(base)
        for i in rng.integers(0, len(s), 30):
            s[i] = "ACGT"[rng.integers(0, 4)]
        if h % 2:
            s[700:700 + 90] = "N" * 90
        if h == 4:
            s[-3:] = "NNN"
        seqs.append("".join(s))
    text = os.path.join(workdir, "pairs_n.txt")
    with open(text, "w") as f:
        for s in seqs:
            f.write(s + "\n")
    ri, _, rl = (W.build_index_from_text(text, workdir, "pairs_n", with_tags=False) + (None,))[:3]
    rl = os.path.join(workdir, "pairs_n.rl_bwt")
    idx = P.Index(ri, mode=mode | P.MODE_IMAGE_PAIRS)
    _check(idx, _bwt_codes(rl), psyms)


def test_pairs_image_of_the_reference_fixtures(workdir, psyms):
    for name in ("x.rl_bwt", "med_test.rl_bwt"):
        ri, _ = W.build_index_from_rlbwt(os.path.join(G, name), workdir, "pairs_" + name, with_tags=False)
        idx = P.Index(ri, mode=P.MODE_STRICT | P.MODE_IMAGE_PAIRS)
        _check(idx, _bwt_codes(os.path.join(G, name)), psyms)


def test_pairs_image_refused_where_the_tables_are_not_the_textbook_ones(workdir, built):
    """an index without N in COMPAT mode carries the reference's quirk tables: no PAIRS image (and none by default)"""
    ri, _ = W.build_index_from_rlbwt(os.path.join(G, "x.rl_bwt"), workdir, "pairs_x_compat", with_tags=False)
    from oracle_ffi import RIndex
    if RIndex(ri).has_N:
        pytest.skip("fixture has N")
    try:
        idx = P.Index(ri, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    except P.PgxError as e:
        assert e.code == P.ERR_UNSUPPORTED
    else:  # the tables happen to be the textbook ones for this index: then the image must be right
        _check(idx, _bwt_codes(os.path.join(G, "x.rl_bwt")))
    assert P.Index(ri).info().image_pairs == 0  # small index: dense image, no pairs


class PairsEmu:
    """one trip of pgx_find_mems_pairs_kernel in Python, from the image views: the two-step arithmetic of pgx_kernels.hip (counts below relA and
    in [relA, relB) of the 96-position block of p0, the neighbour block when the interval runs on, the bail conditions), TEST INFRASTRUCTURE ONLY"""

    def __init__(self, idx):
        self.c = Consts(idx.image_view(6))
        self.blocks = idx.image_view(20).reshape(-1, 32)
        self.L = PairsLayout(self.c)

    def _counts(self, b, rel_a, rel_b, t1, t2):
        h = self.blocks[b]
        pl = [[self.L.plane_word(h, p, w) for w in range(self.L.words)] for p in range(4)]
        bits = lambda p, i: (pl[p][i >> 5] >> (i & 31)) & 1
        c1 = lambda i: bits(0, i) | (bits(1, i) << 1)
        c2 = lambda i: bits(2, i) | (bits(3, i) << 1)
        e1p = sum(1 for i in range(rel_a) if c1(i) == t1)
        e2p = sum(1 for i in range(rel_a) if c1(i) == t1 and c2(i) == t2)
        rng = range(rel_a, rel_b)
        e1r = sum(1 for i in rng if c1(i) == t1)
        g1r = sum(1 for i in rng if c1(i) > t1)
        e2r = sum(1 for i in rng if c1(i) == t1 and c2(i) == t2)
        g2r = sum(1 for i in rng if c1(i) == t1 and c2(i) > t2)
        return e1p, e2p, e1r, g1r, e2r, g2r

    def two_step(self, tri, b1, b2, fwd):
        """(after the first extension, after both) or None where the kernel hands the read on"""
        k, kp, s = tri
        e1, e2 = self.c.ext_tab[(256 if fwd else 0) + b1], self.c.ext_tab[(256 if fwd else 0) + b2]
        cv1, cv2 = e1 & 7, e2 & 7
        reg = lambda e: not ((e >> 24) & 1) and (e & 7) in (1, 2, 3, 5)
        assert reg(e1) and reg(e2)
        t1, t2 = cv1 - 1 - (cv1 >> 2), cv2 - 1 - (cv2 >> 2)
        kk, kq = (kp, k) if fwd else (k, kp)
        p0, p1 = kk, kk + s
        B, S, L = self.L.syms, self.L.stride, self.L
        bf = p0 // S
        endrel = p1 - S * bf
        h = self.blocks[bf]
        over = endrel - B
        cont2 = over > 0 and over <= L.ext_pair(h)  # the interval ends inside the stretch that continues the block's last pair: one line answers both
        cont1 = over > 0 and not cont2 and over <= L.ext_first(h)  # ... or its first symbol: one line answers the first extension
        if endrel > S + B and not (cont1 or cont2):
            return None
        if L.flag(h):
            return None
        e1p, e2p, e1r, g1r, e2r, g2r = self._counts(bf, p0 - S * bf, min(endrel, B), t1, t2)
        a01 = L.rank_before(h, t1) + e1p
        a02 = L.pair_before(h, t1, t2) + e2p
        if cont1 or cont2:
            pl = lambda p: (L.plane_word(h, p, 2) >> 31) & 1
            l1, l2 = pl(0) | (pl(1) << 1), pl(2) | (pl(3) << 1)
            e1r += over if l1 == t1 else 0
            g1r += over if l1 > t1 else 0
            if cont2:
                e2r += over if (l1 == t1 and l2 == t2) else 0
                g2r += over if (l1 == t1 and l2 > t2) else 0
            self.cont_trips = getattr(self, "cont_trips", 0) + 1
        elif endrel > B:  # the next block starts S positions on; the first has answered up to its position B
            h2 = self.blocks[bf + 1]
            if L.flag(h2):
                return None
            _, _, a, b, c, d = self._counts(bf + 1, B - S, endrel - S, t1, t2)
            e1r, g1r, e2r, g2r = e1r + a, g1r + b, e2r + c, g2r + d
        s1, k1, q1 = e1r, a01 + self.c.C[(e1 >> 3) & 7], kq + g1r
        s2, k2, q2 = e2r, a02 + self.c.C[(e2 >> 3) & 7] + self.c.pair_t2[8 * t1 + cv2], q1 + g2r
        first = (0, 0, 0) if s1 == 0 else ((q1, k1, s1) if fwd else (k1, q1, s1))
        both = (0, 0, 0) if s2 == 0 else ((q2, k2, s2) if fwd else (k2, q2, s2))
        return first, (None if cont1 else both)  # (the first symbol's run alone says nothing about the second symbols behind the block)


@pytest.mark.parametrize("haps", [4, 48])
@pytest.mark.parametrize("mode,omode", [(P.MODE_COMPAT, 0), (P.MODE_STRICT, 1)])
def test_two_step_arithmetic_equals_two_oracle_extensions(workdir, mode, omode, psyms, haps):
    """walks of random reads through the oracle, every interval narrower than two blocks extended by the next two symbols through the PAIRS
    image: the first result and the result of both equal the oracle's stepwise bi-intervals, backward and forward.  48 haplotypes: intervals ~48
    wide, a quarter of them run on behind their block -- mostly inside the run of its last position (the run continuation of pgx_image.h)"""
    import oracle_ffi as O
    rng = np.random.default_rng(17)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 6000))
    seqs = []
    for h in range(haps):
        s = list(base)
        for i in rng.integers(0, len(s), 40):
            s[i] = "ACGT"[rng.integers(0, 4)]
        if h == 1:
            s[2000:2040] = "N" * 40
        seqs.append("".join(s))
    rc = lambda t: t[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
    text = os.path.join(workdir, "pairs_walk%d.txt" % haps)
    with open(text, "w") as f:
        for s in seqs:
            f.write(s + "\n" + rc(s) + "\n")
    ri_path = W.build_index_from_text(text, workdir, "pairs_walk%d" % haps, with_tags=False)[0]
    idx, ri = P.Index(ri_path, mode=mode | P.MODE_IMAGE_PAIRS), O.RIndex(ri_path)
    emu = PairsEmu(idx)
    checked = handed_on = 0
    for _ in range(300):
        s = seqs[int(rng.integers(0, len(seqs)))]
        a = int(rng.integers(0, len(s) - 60))
        read = s[a:a + 60].replace("N", "A").encode()
        for fwd in (False, True):
            tri = ri.full()
            order = range(len(read)) if fwd else range(len(read) - 1, -1, -1)
            step = (lambda t, ch: ri.fwd(t, ch, omode)) if fwd else (lambda t, ch: ri.bwd(t, ch, omode))
            order = list(order)
            for q in range(len(order) - 1):
                b1, b2 = read[order[q]], read[order[q + 1]]
                if tri[2] and tri[2] <= 400:
                    got = emu.two_step(tri, b1, b2, fwd)
                    exp1 = step(tri, chr(b1))
                    exp2 = step(exp1, chr(b2)) if exp1[2] else (0, 0, 0)
                    if got is None:
                        handed_on += 1
                    else:
                        assert tuple(int(v) for v in got[0]) == tuple(int(v) for v in exp1), (tri, chr(b1), fwd)
                        if exp1[2] and got[1] is not None:
                            assert tuple(int(v) for v in got[1]) == tuple(int(v) for v in exp2), (tri, chr(b1), chr(b2), fwd)
                        checked += 1
                tri = step(tri, chr(b1))
                if tri[2] == 0:
                    break
    assert checked > 5000 and handed_on < checked // 5, (checked, handed_on)
    assert getattr(emu, "cont_trips", 0) > (1000 if haps == 48 else 10)  # intervals answered through the run continuation of their block


def test_pairs_image_does_not_depend_on_the_builder_threads(workdir, monkeypatch, psyms):
    """the builder splits the BWT into chunks of blocks, one thread each (LF offsets, pair counts and special runs are stitched at the
    chunk borders): 1, 3 and 16 threads give the same bytes, and they are the brute-force image"""
    rng = np.random.default_rng(23)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 40000))
    seqs = []
    for h in range(3):
        s = list(base)
        for i in rng.integers(0, len(s), 300):
            s[i] = "ACGT"[rng.integers(0, 4)]
        if h:
            a = 5000 * h
            s[a:a + 700] = "N" * 700
        seqs.append("".join(s))
    text = os.path.join(workdir, "pairs_thr.txt")
    with open(text, "w") as f:
        for s in seqs:
            f.write(s + "\n")
    ri = W.build_index_from_text(text, workdir, "pairs_thr", with_tags=False)[0]
    images = []
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("PGX_BUILD_THREADS", threads)
        idx = P.Index(ri, mode=P.MODE_STRICT | P.MODE_IMAGE_PAIRS)
        images.append((idx.image_view(20).tobytes(), bytes(idx.image_view(6))))
        if threads == "16":
            _check(idx, _bwt_codes(os.path.join(workdir, "pairs_thr.rl_bwt")), psyms)
    assert images[0] == images[1] == images[2]


class PairsKernelEmu(PairsEmu):
    """pgx_find_mems_pairs_kernel for one read, from the image views: the stage machine with seeds of depth K (computed here by K stepwise
    extensions of the dense2 image, which is what the table holds), the first-extension table, two-step trips with the "first result decides"
    rule, and -- where the PAIRS blocks cannot answer -- that one extension through the image they accompany (ImageEmu = the stepwise arithmetic).
    TEST INFRASTRUCTURE ONLY."""

    def __init__(self, idx, K):
        super().__init__(idx)
        from image_emu import ImageEmu
        self.base = ImageEmu(idx)
        self.K = K
        self.two_step_trips = self.single_trips = self.other_steps = 0

    def _reg(self, byte, fwd):
        e = self.c.ext_tab[(256 if fwd else 0) + byte]
        return not ((e >> 24) & 1) and (e & 7) in (1, 2, 3, 5)

    def _seed(self, b, j, K):
        """(alive tri | None, depth): the K extensions from the full interval over b[j-K+1..j], last byte first"""
        tri = (0, 0, self.c.n)
        for d in range(K):
            tri = self.base.extend(tri, b[j - d], False)
            if tri[2] == 0:
                return None, d + 1
        return tri, K

    def find_all_mems(self, read, min_len, min_occ):
        b = read.encode() if isinstance(read, str) else bytes(read)
        ln, n, K = len(b), self.c.n, self.K
        assert min_len >= K
        out, x, next_ = [], 0, 0
        acgt = lambda lo, hi: all(ch in b"ACGT" for ch in b[lo:hi + 1])

        def stage(tri, j, x, ph, fresh):
            """runs one stage; returns (tri, j, small, extensions)"""
            ne = 0
            while True:
                fwd = ph == 2
                if fresh:
                    fresh = False
                    avail = (j - x + 1) if ph == 1 else (j - x)
                    byte = b[j] if j < ln else 0
                    if j < ln and avail >= K and acgt(j - K + 1, j):  # seed table
                        t, depth = self._seed(b, j, K)
                        if t is not None and t[2] >= min_occ:
                            tri, j, ne, small = t, j - (K - 1), ne + K, False
                        elif t is None and min_occ <= 1:
                            tri, j, ne, small = (0, 0, 0), j - (depth - 1), ne + depth, True
                        else:  # the ordinary first extension stands (first_ext)
                            tri = self.base.extend((0, 0, n), byte, False); ne += 1
                            small = tri[2] < min_occ or tri[2] == 0
                    else:
                        tri = self.base.extend((0, 0, n), byte, False); ne += 1
                        small = tri[2] < min_occ or tri[2] == 0
                        # (the end table and first_ext[256 + byte] only replace further stepwise extensions by their results)
                else:
                    byte = b[j]
                    j2 = j + 1 if fwd else j - 1
                    rem2 = (j - 1 >= x) if ph == 1 else ((j + 1 < ln) if fwd else (j - 1 > x))
                    if not self._reg(byte, fwd):
                        # a symbol outside A C G T has no occurrence in a range free of special positions; where the probe's blocks are flagged (or
                        # the interval is wider than two of them) the extension goes through the other image like any other
                        got = self.two_step(tri, ord("A"), ord("A"), fwd)
                        if got is None:
                            self.other_steps += 1
                            tri = self.base.extend(tri, byte, fwd)
                        else:
                            tri = (0, 0, 0)
                        ne += 1
                        small = tri[2] < min_occ or tri[2] == 0
                    else:
                        two = rem2 and self._reg(b[j2], fwd)
                        got = self.two_step(tri, byte, b[j2] if two else ord("A"), fwd)
                        if got is None:  # this ONE extension through the image the PAIRS image accompanies, then on with pairs (pgx_kernels.hip "bail")
                            self.other_steps += 1
                            got = (self.base.extend(tri, byte, fwd), None)
                            two = False
                        first, both = got
                        two = two and both is not None  # (an interval that ends in the continuation of the first symbol alone: one extension)
                        small1 = first[2] < min_occ or first[2] == 0
                        if two and not small1:
                            self.two_step_trips += 1
                            if fwd:
                                self.J = first
                            tri, j, ne = both, j2, ne + 2
                        else:
                            self.single_trips += 1
                            tri, ne = first, ne + 1
                        small = tri[2] < min_occ or tri[2] == 0
                if small:
                    return tri, j, True, ne
                if ph == 1:
                    if j == x:
                        return tri, j, False, ne
                    j -= 1
                elif ph == 2:
                    self.J = tri
                    j += 1
                    if j >= ln:
                        return tri, j, False, ne
                else:
                    j -= 1
                    if j <= x:
                        return tri, j, False, ne

        while True:
            if x >= ln or (ln - x) < min_len:
                break
            tri, j, small, ne = stage((0, 0, n), x + min_len - 1, x, 1, True); next_ += ne
            if small:
                x = j + 1; continue
            self.J = tri
            j = x + min_len
            if j < ln:
                tri, j, small, ne = stage(tri, j, x, 2, False); next_ += ne
            e = j
            out.append((x, e, self.J[0], self.J[2]))
            nxt = x + 1
            if e > x:
                tri, j, small, ne = stage((0, 0, n), e, x, 3, True); next_ += ne
                if small:
                    nxt = j + 1
            x = nxt
        return out, next_


@pytest.mark.parametrize("mode,omode", [(P.MODE_COMPAT, 0), (P.MODE_STRICT, 1)])
def test_pairs_kernel_state_machine_emulated(workdir, mode, omode, psyms):
    """the whole two-step search of reads, emulated from the image views, against the oracle's find_all_mems: MEMs and extension counts"""
    import oracle_ffi as O
    rng = np.random.default_rng(29)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 20000))
    seqs = []
    for h in range(3):
        s = list(base)
        for i in rng.integers(0, len(s), 150):
            s[i] = "ACGT"[rng.integers(0, 4)]
        if h == 2:
            s[7000:7050] = "N" * 50
        seqs.append("".join(s))
    text = os.path.join(workdir, "pairs_sm.txt")
    with open(text, "w") as f:
        for s in seqs:
            f.write(s + "\n")
    ri_path = W.build_index_from_text(text, workdir, "pairs_sm", with_tags=False)[0]
    idx, ri = P.Index(ri_path, mode=mode | P.MODE_IMAGE_PAIRS), O.RIndex(ri_path)
    emu = PairsKernelEmu(idx, K=6)
    for i in range(120):
        s = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(12, 90))
        a = int(rng.choice([0, len(s) - ln, int(rng.integers(0, len(s) - ln))]))
        r = bytearray(s[a:a + ln].encode())
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, ln))] = int(rng.choice(np.frombuffer(b"ACGTNa", dtype=np.uint8)))
        for min_len, min_occ in ((8, 1), (11, 2)):
            exp = ri.find_all_mems(bytes(r), min_len, min_occ, omode, with_ext=True)
            assert emu.find_all_mems(bytes(r), min_len, min_occ) == exp, (bytes(r), min_len, min_occ)
    assert emu.two_step_trips > 2000 and emu.two_step_trips > emu.single_trips and emu.other_steps > 0  # (the N run and the sequence ends flag blocks)
