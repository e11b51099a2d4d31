"""query_tags path: FastLocate::count / count_encoded.
CPU tier: the count tables of the device image, walked like pgx_count_kernel, vs the oracle.
GPU tier: pgx_count_batch and the query_tags CLI vs the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W
from image_emu import ImageEmu

G = O.GOLDEN
BT = os.path.join(G, "bidirectional_test")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _patterns(text, rng, n=300):
    pats = ["", "A", "T", "ACG", "GAT", "GATTACA", "ACGN", "acg", "N", "TTTT", "A" * 40]
    for _ in range(n):
        L = int(rng.integers(1, 30))
        s = int(rng.integers(0, len(text) - L))
        p = text[s:s + L]
        if "\n" in p:
            continue
        if rng.random() < 0.3:  # one substitution
            k = int(rng.integers(0, L))
            p = p[:k] + "ACGTN"[int(rng.integers(0, 5))] + p[k + 1:]
        pats.append(p)
    return pats


def _count_truth(text, p):
    c, i = 0, text.find(p)
    while i >= 0:
        c, i = c + 1, text.find(p, i + 1)
    return c


def _cases(workdir):
    """(ri path, text path, oracle mode/P mode pairs that must be supported)"""
    out = [(os.path.join(BT, "xy.ri"), os.path.join(BT, "contigs_xy"), [0, 1])]  # legacy fixture: COMPAT + STRICT
    enc, _ = W.build_index_from_rlbwt(os.path.join(BT, "contigs_xy.rl_bwt"), workdir, "cnt_xy_enc", with_tags=False)
    out.append((enc, os.path.join(BT, "contigs_xy"), [1]))  # encoded without N: COMPAT goes through the literal image (quirk 3), see below
    text = os.path.join(workdir, "cnt_toy.txt")
    W.synth_pangenome_text(text, base_len=5000, n_hap=2, seed=3, n_runs=2, n_run_len=(20, 100))
    ri6, _, _ = W.build_index_from_text(text, workdir, "cnt_toy", with_tags=False)
    out.append((ri6, text, [0, 1]))  # encoded, sigma = 6
    return out


def test_count_tables_cpu(workdir):
    rng = np.random.default_rng(17)
    for ri_path, text_path, modes in _cases(workdir):
        text = open(text_path).read()
        ri = O.RIndex(ri_path)
        for mode, force in ((0, 0), (1, 0), (0, P.MODE_IMAGE_RL), (1, P.MODE_IMAGE_RL)):  # automatic layout (dense where possible), and RL
            idx = P.Index(ri_path, mode=mode | force)
            emu = ImageEmu(idx)
            assert bool(emu.c.count_supported) == (mode in modes)
            if mode not in modes:
                continue
            for p in _patterns(text, rng, 60):
                exp = ri.count(p, mode)
                assert emu.count(p) == exp, (ri_path, mode, p)
                if mode == 1 or ri.sigma == 6 or not ri.encoded:
                    n = exp[1] - exp[0] + 1 if exp[0] <= exp[1] else 0
                    if p and all(ch in "ACGTN" for ch in p):
                        assert n == _count_truth(text, p), p


def _lit_rank(bstart, cum, runs, roff, pos, target):
    """pgx_lit_rank (pgx_kernels.hip): the reference's rankAt_encoded on an encoded index without N"""
    b = int(np.searchsorted(bstart, pos, side="right")) - 1
    rel = pos - int(bstart[b])
    rank = cur = 0
    for e in range(int(roff[b]), int(roff[b + 1])):
        u = int(runs[e])
        code, ln = u >> 56, u & ((1 << 56) - 1)
        if code == target:
            if cur + ln > rel:
                rank += rel - cur
                break
            rank += ln
        cur += ln
        if cur > rel:
            break
    return rank + int(cum[b * 6 + target])


def test_quirk3_literal_image(workdir):
    """on an encoded index without N the reference's count_encoded is wrong-but-deterministic (rankAt_encoded reads six
    cumulative varints where five were written, so its run scan starts one varint late): the oracle restates it literally
    and the product reproduces it from an image of the reference's blocks as that scan sees them (CPU tier: the image walked
    like pgx_lit_count_kernel)"""
    rng = np.random.default_rng(23)
    for name, text_name in (("bidirectional_test/contigs_xy.rl_bwt", "bidirectional_test/contigs_xy"), ("x.rl_bwt", "x.newline_separated")):
        enc, _ = W.build_index_from_rlbwt(os.path.join(G, name), workdir, "q3_" + os.path.basename(name), with_tags=False)
        e = O.RIndex(enc)
        if "contigs_xy" in name:
            legacy = O.RIndex(os.path.join(BT, "xy.ri"))
            assert legacy.count("ACG") == (969, 988) and e.count("ACG", O.MODE_STRICT) == (969, 988)
            assert e.count("ACG") != (969, 988)
        idx = P.Index(enc)
        bstart, cum, runs, roff = (idx.image_view(w) for w in (16, 17, 18, 19))
        assert len(cum) == 6 * len(bstart) and len(roff) == len(bstart) + 1 and bstart[0] == 0
        C, sm, n = e.C_array(), e.sym_map(), e.n
        text = open(os.path.join(G, text_name)).read()
        for p in _patterns(text, rng, 150) + ["T", "TT", "N", "\x00A"]:
            lo, hi = 0, n - 1
            for ch in reversed(p.encode()):
                if lo > hi:
                    lo, hi = 1, 0
                    continue
                target = b"\nACGNT".find(bytes([ch]))
                target = 0 if target < 0 else target
                f = _lit_rank(bstart, cum, runs, roff, lo, target)
                inside = _lit_rank(bstart, cum, runs, roff, hi + 1, target) - f
                if inside == 0:
                    lo, hi = 1, 0
                    continue
                lo = f + C[sm[ch]]
                hi = lo + inside - 1
            assert (lo, hi) == e.count(p), (name, p)
    with pytest.raises(P.PgxError):  # the literal image exists for that one shape only
        P.Index(os.path.join(BT, "xy.ri")).image_view(16)


@pytest.mark.gpu
def test_count_batch_gpu(workdir):
    rng = np.random.default_rng(18)
    for ri_path, text_path, modes in _cases(workdir):
        text = open(text_path).read()
        ri = O.RIndex(ri_path)
        pats = _patterns(text, rng, 2000)
        cat, offs = O.pack_reads(pats)
        for mode, force in ((0, 0), (1, 0), (0, P.MODE_IMAGE_RL), (1, P.MODE_IMAGE_RL)):
            idx = P.Index(ri_path, mode=mode | force)
            # (COMPAT on the encoded index without N: the literal image of quirk 3 answers, wrong-but-deterministic like the reference)
            got = idx.count_batch(cat, offs)
            for i, p in enumerate(pats):
                assert (int(got[i][0]), int(got[i][1])) == ri.count(p, mode), (ri_path, mode, p)
            # one LF step from random ranges
            rg = np.sort(rng.integers(0, ri.n, (500, 2)), axis=1).astype(np.uint64)
            sy = np.frombuffer(b"ACGTN\n$a", dtype=np.uint8)[rng.integers(0, 8, 500)]
            lf = idx.lf_batch(rg, sy)
            for q in range(500):
                assert (int(lf[q][0]), int(lf[q][1])) == ri.LF((int(rg[q][0]), int(rg[q][1])), int(sy[q]), mode), (ri_path, mode, q)


@pytest.mark.gpu
def test_query_tags_cli(built, tmp_path):
    """stdout grammar of src/query_tags.cpp:103-108 + tag_arrays.cpp:885-889 on the reference's fixture pair"""
    ri_path, tags_path = os.path.join(BT, "xy.ri"), os.path.join(BT, "xy_bidirectional_compressed.tags")
    reads = [l for l in open(os.path.join(BT, "reads.txt")).read().split("\n") if l] + ["ACG", "GAT", "TTTTGG"]
    path = str(tmp_path / "qt_reads.txt")
    open(path, "w").write("\n".join(reads) + "\n\n")
    r = subprocess.run([os.path.join(ROOT, "pangenome-index_amd", "query_tags"), ri_path, tags_path, path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ri, tags = O.RIndex(ri_path), O.Tags(tags_path, O.TAGS_BYTECODE)
    exp = ""
    for i, rd in enumerate(reads):
        lo, hi = ri.count(rd)
        if lo > hi:
            assert ("Read %d has no matches" % i) in r.stderr
            continue
        rn, pos, _ = tags.query(lo, hi)
        exp += "Number of unique positions: %d\n" % len(pos) + "".join("%d, " % p for p in pos) + "\n"
        exp += "read_index=%d\tlen=%d\tbwt_start=%d\tbwt_end=%d\truns=%d\n" % (i, len(rd), lo, hi, rn)
    assert r.stdout == exp and "read_index=" in exp
