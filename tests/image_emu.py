"""Pure-Python walk of the flat device image (pgx_image.h) -- TEST INFRASTRUCTURE ONLY.

Lets the CPU-only test tier check the *image* (blocks, directory, extension tables) that
pgx_index_open builds against the oracle, without a GPU.  It mirrors pgx_rank_ab / pgx_extend of
pgx_kernels.hip line for line; it is never used by the product.
"""
import struct

import numpy as np

RUN_LEN_BITS, RUN_LEN_MAX, BLOCK_RUNS = 12, 4095, 16
M64 = (1 << 64) - 1


class Consts:
    """PgxConsts of pgx_image.h"""

    def __init__(self, raw):
        raw = bytes(raw)
        o = 0
        self.n, = struct.unpack_from("<Q", raw, o); o += 8
        self.C = struct.unpack_from("<8Q", raw, o); o += 64
        self.ext_tab = struct.unpack_from("<512I", raw, o); o += 2048
        self.slot_code = struct.unpack_from("<8I", raw, o); o += 32
        self.sigma, self.excl_mask, self.dir_shift, self.n_blocks = struct.unpack_from("<4I", raw, o); o += 16
        self.dir_entries, = struct.unpack_from("<Q", raw, o); o += 8
        self.n_tag_runs, self.tag_dir_entries = struct.unpack_from("<2Q", raw, o); o += 16
        self.tag_dir_shift, self.has_tags, self.mode, self.count_supported = struct.unpack_from("<4I", raw, o); o += 16
        self.cnt_tab = struct.unpack_from("<256I", raw, o); o += 1024
        self.image_kind, self.has_pairs = struct.unpack_from("<2I", raw, o); o += 8
        self.pair_t2 = struct.unpack_from("<32I", raw, o); o += 128
        self.pair_runs, _ = struct.unpack_from("<2I", raw, o); o += 8
        self.wide, self.d2_sb_shift, self.pairs_sb_shift, self.n_sb2, self.n_sbp, self.pairs_stride = struct.unpack_from("<6I", raw, o); o += 24
        self.pair_t2w = struct.unpack_from("<32Q", raw, o); o += 256
        assert o == len(raw), (o, len(raw))


class PairsLayout:
    """where the fields of a PAIRS block (32 dwords, 96 positions) are, and where the blocks start (pgx_image.h): pair counts dw 0..15, c2-special
    counts + flag dw 16..19, planes dw 20..31 of three words each; block b starts at position stride * b (stride 96: the blocks tile the BWT;
    stride 64: they overlap by 32 positions)"""

    def __init__(self, consts):
        self.syms = 96
        self.stride = consts.pairs_stride or 96
        assert self.stride in (96, 64)
        self.words = 3

    def flag(self, h):
        return int(h[16]) >> 31

    def rank_before(self, h, y):
        """positions before the block whose first symbol is y (second symbol regular or special)"""
        return sum(int(h[4 * y + x]) for x in range(4)) + self.half_before(h, y)

    def pair_before(self, h, y, x):
        return int(h[4 * y + x])

    def half_before(self, h, y):
        return int(h[16 + y]) & 0xFFFFFF  # 24-bit counts; the bits above hold the flag (dw 16) and the run continuation (dw 17, 18)

    def ext_pair(self, h):
        """positions right behind the block (<= 255) that carry the same regular pair as the block's last position"""
        return int(h[17]) >> 24

    def ext_first(self, h):
        """the same for the first symbol alone"""
        return int(h[18]) >> 24

    def plane_word(self, h, plane, w):
        return int(h[20 + 3 * plane + w])


class ImageEmu:
    def __init__(self, index):
        self.c = Consts(index.image_view(6))
        self.blocks = index.image_view(0).view(np.uint32).reshape(-1, 32 if self.c.image_kind == 2 else 16)
        self.exc = index.image_view(15) if self.c.image_kind == 2 else None
        self.dir = index.image_view(1)
        self.bstart = index.image_view(2)
        self.blow = index.image_view(7)
        self.tstart = index.image_view(3)
        self.tvals = index.image_view(4)
        self.tdir = index.image_view(5)

    def block_counts(self, b):
        dw = [int(x) for x in self.blocks[b]]
        c = [dw[i] for i in range(6)]
        for i in range(4):
            c[i] |= ((dw[6] >> (8 * i)) & 0xFF) << 32
        c[4] |= (dw[7] & 0xFF) << 32
        c[5] |= ((dw[7] >> 8) & 0xFF) << 32
        ents = []
        for e in range(BLOCK_RUNS):
            v = (dw[8 + e // 2] >> (16 * (e & 1))) & 0xFFFF
            assert (v >> RUN_LEN_BITS) % 3 == 0 and (v >> RUN_LEN_BITS) <= 15
            ents.append(((v >> RUN_LEN_BITS) // 3, v & RUN_LEN_MAX))
        return c, ents, (dw[7] >> 16) & 0x1F

    def find_block(self, pos):
        c = self.c
        di = pos >> c.dir_shift
        e = int(self.dir[di])
        lo, cnt, l0, l1 = e & 0xFFFFFFFF, (e >> 32) & 0xFF, (e >> 40) & 0xFFF, e >> 52
        lowp = pos & ((1 << c.dir_shift) - 1)
        hi = lo + cnt if cnt < 255 else int(self.dir[di + 1]) & 0xFFFFFFFF
        if cnt <= 2:
            hi = lo = lo + (1 if cnt >= 1 and l0 <= lowp else 0) + (1 if cnt >= 2 and l1 <= lowp else 0)
        while lo < hi:
            mid = (lo + hi) >> 1
            if int(self.blow[mid]) <= lowp:
                lo = mid + 1
            else:
                hi = mid
        b = lo - 1
        assert int(self.bstart[b]) <= pos and (b + 1 == c.n_blocks or pos < int(self.bstart[b + 1])), (pos, b)
        return b

    def dense_rank(self, pos, cv, mrow):
        """pgx_dense_rank: header counts + popcounts of plane combinations under a prefix mask"""
        dw = [int(x) for x in self.blocks[pos >> 6]]
        cnt = [dw[i] for i in range(6)]
        for i in range(4):
            cnt[i] |= ((dw[6] >> (8 * i)) & 0xFF) << 32
        cnt[4] |= (dw[7] & 0xFF) << 32
        cnt[5] |= ((dw[7] >> 8) & 0xFF) << 32
        rel = pos & 63
        m = (1 << rel) - 1
        p = [(dw[8 + 2 * i] | (dw[9 + 2 * i] << 32)) & m for i in range(3)]
        pc = lambda v: bin(v).count("1")
        n1, n2, n4, n3, n5 = pc(p[0]), pc(p[1]), pc(p[2]), pc(p[0] & p[1]), pc(p[0] & p[2])
        assert pc(p[1] & p[2]) == 0  # codes 6, 7 never occur
        t = [rel - (n1 + n2 + n4 - n3 - n5), n1 - n3 - n5, n2 - n3, n3, n4 - n5, n5]
        A = cnt[cv] + t[cv]
        B = sum((cnt[i] + t[i]) * ((mrow >> (3 * i)) & 7) for i in range(6))
        return A & M64, B & M64

    def dense2_rank(self, pos, cv, mrow):
        """pgx_dense2_rank: 384 symbols per 128-byte block: header, three sub-blocks of two planes (A C G T = 0 1 2 3), exception
        runs for \\n and N; a probe reads the header and the sub-block of its position"""
        blk = (pos * 0xAAAAAAAB) >> 40
        assert blk == pos // 384 and pos < (1 << 32)
        dw = [int(x) for x in self.blocks[blk]]
        rel = pos - blk * 384
        sub, r = rel >> 7, rel & 127
        sc = (dw[6] | (dw[7] << 32)) >> (27 if sub == 2 else 0)
        n0, n1, n3 = (sc & 511, (sc >> 9) & 511, (sc >> 18) & 511) if sub else (0, 0, 0)
        pc = lambda v: bin(v).count("1")
        # the stored sub-block counts are the popcounts of the sub-blocks before this one
        chk = [0, 0, 0]
        for s in range(sub):
            for k in range(4):
                a, d = dw[8 + 8 * s + k], dw[12 + 8 * s + k]
                chk[0] += pc(a); chk[1] += pc(d); chk[2] += pc(a & d)
        assert chk == [n0, n1, n3], (blk, sub)
        for h in range(4):
            t = r - 32 * h
            m = 0xFFFFFFFF if t >= 32 else ((1 << t) - 1 if t > 0 else 0)
            a, d = dw[8 + 8 * sub + h] & m, dw[12 + 8 * sub + h] & m
            n0 += pc(a); n1 += pc(d); n3 += pc(a & d)
        e0 = e4 = 0
        prev_end = 0
        for i in range(dw[5] >> 24):
            u = int(self.exc[(dw[5] & 0xFFFFFF) + i])
            st, ln, kind = u & 511, (u >> 9) & 511, (u >> 18) & 1
            assert ln >= 1 and st >= prev_end and st + ln <= 384 and u >> 19 == 0
            prev_end = st + ln
            for q in range(st, st + ln):  # exception positions are stored as code 0 in both planes
                sb, rr = 8 + 8 * (q >> 7), q & 127
                assert not (dw[sb + (rr >> 5)] >> (rr & 31)) & 1 and not (dw[sb + 4 + (rr >> 5)] >> (rr & 31)) & 1
            take = min(rel - st, ln) if rel > st else 0
            if kind:
                e4 += take
            else:
                e0 += take
        hsum = sum(dw[:5])
        c = [(pos - rel) - hsum + e0, dw[0] + rel - (n0 + n1 - n3) - e0 - e4, dw[1] + n0 - n3, dw[2] + n1 - n3, dw[4] + e4, dw[3] + n3]
        assert all(v >= 0 for v in c) and sum(c) == pos
        A = c[cv]
        B = sum(c[i] * ((mrow >> (3 * i)) & 7) for i in range(6))
        return A & M64, B & M64

    def rank_ab(self, pos, cv, mrow):
        c = self.c
        pos = min(pos, c.n)
        if c.image_kind == 2:
            return self.dense2_rank(pos, cv, mrow)
        if c.image_kind == 1:
            return self.dense_rank(pos, cv, mrow)
        b = self.find_block(pos)
        cnt, ents, _ = self.block_counts(b)
        start = sum(cnt[i] for i in range(6) if not (c.excl_mask >> i) & 1)
        assert start == int(self.bstart[b]), (b, start, int(self.bstart[b]))
        A = cnt[cv]
        B = sum(cnt[i] * ((mrow >> (3 * i)) & 7) for i in range(6))
        rel = pos - start
        assert 0 <= rel <= BLOCK_RUNS * RUN_LEN_MAX
        for code, ln in ents:
            take = min(ln, rel)
            rel -= take
            if code == cv:
                A += take
            B += take * ((mrow >> (3 * code)) & 7)
        return A & M64, B & M64

    def rank_pair(self, pos0, pos1, cv, mrow):
        """pgx_rank_pair: a two-trip loop; trip 0 decodes pos0's block and serves pos1 too when it covers it"""
        c = self.c
        p0, p1 = min(pos0, c.n), min(pos1, c.n)
        if c.image_kind == 2:
            (A0, B0), (A1, B1) = self.dense2_rank(p0, cv, mrow), self.dense2_rank(p1, cv, mrow)
            return A0, A1, (B1 - B0) & M64
        if c.image_kind == 1:
            (A0, B0), (A1, B1) = self.dense_rank(p0, cv, mrow), self.dense_rank(p1, cv, mrow)
            return A0, A1, (B1 - B0) & M64
        A0 = A1 = B0 = B1 = 0
        done = False
        for it in (0, 1):
            if done:
                continue
            p = p1 if it else p0
            b = self.find_block(p)
            cnt, ents, _ = self.block_counts(b)
            start = sum(cnt[i] for i in range(6) if not (c.excl_mask >> i) & 1)
            a = cnt[cv]
            bw = sum(cnt[i] * ((mrow >> (3 * i)) & 7) for i in range(6))
            relp = p - start
            d1 = (p1 - start) & M64
            rels = 0 if it else min(d1, 0xFFFFFFFF)
            iap = ibp = ias = ibs = total = 0
            for code, ln in ents:
                tp, ts = min(ln, relp), min(ln, rels)
                m = (mrow >> (3 * code)) & 7
                total += ln
                relp -= tp
                rels -= ts
                if code == cv:
                    iap += tp
                    ias += ts
                ibp += tp * m
                ibs += ts * m
            if it == 0:
                A0, B0 = a + iap, bw + ibp
                if d1 < total or (rels == 0 and b + 1 == c.n_blocks):
                    A1, B1, done = a + ias, bw + ibs, True
            else:
                A1, B1 = a + iap, bw + ibp
        out = (A0 & M64, A1 & M64, (B1 - B0) & M64)
        A0f, B0f = self.rank_ab(pos0, cv, mrow)
        A1f, B1f = self.rank_ab(pos1, cv, mrow)
        assert out == (A0f, A1f, (B1f - B0f) & M64), (pos0, pos1, cv, mrow)
        return out

    def rank6_true(self, pos):
        return [self.rank_ab(pos, code, 0)[0] for code in range(6)]

    def rank_cache(self, pos):
        return [self.rank_ab(pos, self.c.slot_code[i], 0)[0] for i in range(self.c.sigma)]

    def extend(self, tri, byte, fwd):
        k, kp, s = tri
        e = self.c.ext_tab[(256 if fwd else 0) + byte]
        cv, v, mrow, kill = e & 7, (e >> 3) & 7, (e >> 6) & 0x3FFFF, (e >> 24) & 1
        kk, kq = (kp, k) if fwd else (k, kp)
        A0, A1, dB = self.rank_pair(kk, (kk + s) & M64, cv, mrow)
        if kill or A0 >= A1:
            return (0, 0, 0)
        nk = (A0 + self.c.C[v]) & M64
        nq = (kq + dB) & M64
        ns = A1 - A0
        return (nq, nk, ns) if fwd else (nk, nq, ns)

    def find_all_mems(self, read, min_len, min_occ):
        """state machine of pgx_find_mems_kernel, one read"""
        return self.find_all_mems_from(read, min_len, min_occ, 0)

    def find_all_mems_from(self, read, min_len, min_occ, x0):
        """... from start position x0 on (a read the pairs kernel hands on resumes like this)"""
        b = read.encode() if isinstance(read, str) else bytes(read)
        ln, n = len(b), self.c.n
        out, x, next_ = [], x0, 0
        while True:
            if x >= ln or (ln - x) < min_len:
                break
            tri = (0, 0, n)
            restart = None
            if min_len > 0:
                j = x + min_len - 1
                while True:
                    tri = self.extend(tri, b[j] if j < ln else 0, False); next_ += 1
                    if tri[2] < min_occ or tri[2] == 0:
                        restart = j + 1; break
                    if j == x:
                        break
                    j -= 1
            if restart is not None:
                x = restart; continue
            J = tri
            j = x + min_len
            while j < ln:
                tri = self.extend(tri, b[j], True); next_ += 1
                if tri[2] < min_occ or tri[2] == 0:
                    break
                J = tri; j += 1
            e = j
            out.append((x, e, J[0], J[2]))
            tri = (0, 0, n)
            j = e
            nxt = x + 1
            while j > x:
                tri = self.extend(tri, b[j] if j < ln else 0, False); next_ += 1
                if tri[2] < min_occ or tri[2] == 0:
                    nxt = j + 1; break
                j -= 1
            x = nxt
        return out, next_

    def count(self, read):
        """pgx_count_kernel, one read"""
        b = read.encode() if isinstance(read, str) else bytes(read)
        lo, hi = 0, self.c.n - 1
        for ch in reversed(b):
            if lo > hi:
                break
            e = self.c.cnt_tab[ch]
            if (e >> 24) & 1:
                return (1, 0)
            A0, A1, _ = self.rank_pair(lo, hi + 1, e & 7, 0)
            if A1 == A0:
                return (1, 0)
            lo = A0 + self.c.C[(e >> 3) & 7]
            hi = lo + (A1 - A0) - 1
        return (lo, hi)

    # tags
    def tag_rank(self, x):
        c = self.c
        di = x >> c.tag_dir_shift
        if di + 1 >= c.tag_dir_entries:
            return c.n_tag_runs
        lo, hi = int(self.tdir[di]), int(self.tdir[di + 1])
        while lo < hi:
            mid = (lo + hi) >> 1
            if int(self.tstart[mid]) <= x:
                lo = mid + 1
            else:
                hi = mid
        return lo

    def tag_query(self, start, end):
        f, g = self.tag_rank(start), self.tag_rank(end)
        cnt = g - f + 1
        first = f - 1 if f % 10 else f
        vals, over = [], False
        for t in range(cnt):
            it = first + t
            if it < len(self.tvals):
                vals.append(int(self.tvals[it]))
            else:
                vals.append(0); over = True
        return cnt, sorted(set(vals)), over


class LocateEmu:
    """Walk of the locate image (PgxLocConsts + views 8..13) exactly as pgx_locate_kernels.hip walks it."""

    NO = M64

    def __init__(self, index):
        raw = bytes(index.image_view(14))
        (self.n, self.n_runs, self.n_last, self.max_length, self.rdir_entries, self.ldir_entries,
         self.rdir_shift, self.ldir_shift) = struct.unpack("<6Q2I", raw)
        self.rstart = index.image_view(8)
        self.rsamp = index.image_view(9)
        self.rdir = index.image_view(10)
        self.lpos = index.image_view(11)
        self.lnext = index.image_view(12)
        self.ldir = index.image_view(13)
        assert len(self.rstart) == self.n_runs + 1 and len(self.rsamp) == self.n_runs
        assert len(self.lpos) == self.n_last == len(self.lnext)
        assert len(self.rdir) == self.rdir_entries and len(self.ldir) == self.ldir_entries

    @staticmethod
    def _upper(arr, dirv, shift, entries, cnt, x):
        di = x >> shift
        if di + 1 >= entries:
            return cnt
        lo, hi = int(dirv[di]), int(dirv[di + 1])
        while lo < hi:
            mid = (lo + hi) >> 1
            if int(arr[mid]) <= x:
                lo = mid + 1
            else:
                hi = mid
        return lo

    def locate_next(self, prev):
        if prev == self.NO:
            return self.NO
        c = self._upper(self.lpos, self.ldir, self.ldir_shift, self.ldir_entries, self.n_last, prev)
        if c == 0:
            return self.NO
        nx = int(self.lnext[c - 1])
        return self.NO if nx == self.NO else nx + (prev - int(self.lpos[c - 1]))

    def locate(self, first, last):
        """values of BWT[first..last]: one chain per (range, run) piece from the run's head sample"""
        if last < first:
            return []
        r0 = self._upper(self.rstart, self.rdir, self.rdir_shift, self.rdir_entries, self.n_runs, first) - 1
        r1 = self._upper(self.rstart, self.rdir, self.rdir_shift, self.rdir_entries, self.n_runs, last) - 1
        out = []
        for run in range(r0, r1 + 1):
            rs, re = int(self.rstart[run]), int(self.rstart[run + 1])
            a, b = max(first, rs), min(last, re - 1)
            v = int(self.rsamp[run])
            for _ in range(rs, a):
                v = self.locate_next(v)
            for p in range(a, b + 1):
                out.append(v)
                if p < b:
                    v = self.locate_next(v)
        return out
