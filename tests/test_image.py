"""CPU tier: the flat device image built by pgx_index_open (blocks, directory, extension tables, tag
arrays) checked against the oracle by walking it exactly as the kernels do (tests/image_emu.py)."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W
from image_emu import ImageEmu

G = O.GOLDEN
BT = os.path.join(G, "bidirectional_test")
BYTES = b"ACGTNacgtn\x00\n$XR\r"


def _check_rank_and_extend(idx, ri, mode, n_ext=600, step=1):
    emu = ImageEmu(idx)
    c = emu.c
    assert c.n == ri.n and c.sigma == ri.sigma and list(c.C[: ri.sigma]) == ri.C_array()
    # every block: start position = sum of header counts (minus the excluded quirk slots)
    if c.image_kind == P.IMAGE_RL:
        assert len(emu.bstart) == c.n_blocks and emu.bstart[0] == 0
    elif c.image_kind == P.IMAGE_DENSE2:
        assert c.n_blocks == c.n // 384 + 1 and len(emu.blocks) == c.n_blocks
    else:
        assert c.n_blocks == (c.n >> 6) + 1 and len(emu.blocks) == c.n_blocks
    for pos in list(range(0, ri.n + 1, step)) + [ri.n, ri.n + 5, 1 << 62]:
        p = min(pos, ri.n)
        if mode == O.MODE_COMPAT:
            assert emu.rank_cache(pos) == ri.rank_at_cached(p), pos
        else:
            assert emu.rank6_true(pos) == ri.rank6_true(p), pos
    rng = np.random.default_rng(5)
    for _ in range(n_ext):
        k = int(rng.integers(0, ri.n))
        s = int(rng.integers(1, ri.n - k + 1))
        kp = int(rng.integers(0, ri.n))
        a = BYTES[int(rng.integers(0, len(BYTES)))]
        for fwd in (False, True):
            exp = (ri.fwd if fwd else ri.bwd)((k, kp, s), a, mode)
            assert emu.extend((k, kp, s), a, fwd) == exp, (k, kp, s, a, fwd)
    return emu


@pytest.mark.parametrize("mode", [O.MODE_COMPAT, O.MODE_STRICT])
def test_image_xy_legacy_fixture(built, mode):
    ri_path = os.path.join(BT, "xy.ri")
    idx = P.Index(ri_path, os.path.join(BT, "xy_bidirectional_compressed.tags"), mode=mode)
    ri = O.RIndex(ri_path)
    inf = idx.info()
    assert (inf.bwt_size, inf.sigma, inf.n_sequences, inf.n_ref_blocks, inf.n_runs) == (8022, 5, 8, 163, 1620)
    assert inf.is_encoded == 0 and inf.has_tags == 1 and inf.tag_format == P.TAGS_BYTECODE and inf.n_tag_runs == 6031
    assert abs(inf.ref_block_mean_bytes - 18.7) < 0.1  # "measured on xy: mean 18.7 B"
    emu = _check_rank_and_extend(idx, ri, mode, step=3)
    # COMPAT on a legacy no-N index: slot 4 carries the reference-block cumulative endmarker count
    assert emu.c.excl_mask == (0x10 if mode == O.MODE_COMPAT else 0)
    for rd in ["ACCCTAGAGTAT", "GGTAGCCATGCT", "TTTTGGAGGAGT", "CCCATAGTCGAA", "ATATATATATAT", "", "A", "NNNNNN"]:
        for ml in (0, 3, 5):
            mems, ne = ri.find_all_mems(rd, ml, 1, mode, with_ext=True)
            assert emu.find_all_mems(rd, ml, 1) == (mems, ne), (rd, ml)


@pytest.mark.parametrize("mode", [O.MODE_COMPAT, O.MODE_STRICT])
@pytest.mark.parametrize("name,encoded", [("x.rl_bwt", True), ("x.rl_bwt", False), ("med_test.rl_bwt", True),
                                          ("bidirectional_test/small_test/test.rl_bwt", True),
                                          ("bidirectional_test/small_test/test.rl_bwt", False),
                                          ("two_contig_graph/contigs_XY.rl_bwt", True)])
@pytest.mark.parametrize("image", [P.IMAGE_RL, P.IMAGE_DENSE, P.IMAGE_DENSE2])
def test_image_built_indexes(workdir, name, encoded, mode, image):
    ri_path, _ = W.build_index_from_rlbwt(os.path.join(G, name), workdir, "img_" + os.path.basename(name), encoded=encoded,
                                          with_tags=False)
    ri = O.RIndex(ri_path)
    force = {P.IMAGE_DENSE: P.MODE_IMAGE_DENSE, P.IMAGE_DENSE2: P.MODE_IMAGE_DENSE2, P.IMAGE_RL: P.MODE_IMAGE_RL}[image]
    if image != P.IMAGE_RL and mode == O.MODE_COMPAT and not encoded and not ri.has_N and ri.sigma == 5:
        # legacy layout without N in COMPAT: a header slot carries the reference-block quirk value, so the device blocks
        # must refine the reference's blocks -- no dense image there
        with pytest.raises(P.PgxError) as e:
            P.Index(ri_path, mode=mode | force)
        assert e.value.code == P.ERR_UNSUPPORTED
        return
    idx = P.Index(ri_path, mode=mode | force)
    assert bool(idx.info().is_encoded) == encoded and idx.info().image_kind == image and idx.info().mode == mode
    emu = _check_rank_and_extend(idx, ri, mode, n_ext=300, step=2)
    for rd in ["ACCCTAGAGTAT", "GATTAGATACAT", "TTTTGGAGGAGTNNA", ""]:
        assert emu.find_all_mems(rd, 3, 1) == ri.find_all_mems(rd, 3, 1, mode, with_ext=True), rd


def test_image_reference_two_contig_fixture(built):
    """the reference's second real FastLocate file (two_contig_graph/r-index/xy.ri)"""
    ri_path = os.path.join(G, "two_contig_graph", "xy.ri")
    idx, ri = P.Index(ri_path), O.RIndex(ri_path)
    inf = idx.info()
    assert (inf.bwt_size, inf.n_runs, inf.n_ref_blocks) == (4011, 810, 82)
    _check_rank_and_extend(idx, ri, O.MODE_COMPAT, n_ext=300, step=2)


def test_long_runs_split_and_merge(workdir):
    """runs longer than the 13-bit entry limit are split; adjacent equal symbols are merged"""
    text = os.path.join(workdir, "long.txt")
    with open(text, "w") as f:
        f.write("A" * 70000 + "C" * 9001 + "G" * 16383 + "T" * 5 + "N" * 40000 + "A\n")
        f.write("T" * 30000 + "G" * 8192 + "ACGT" * 10 + "C" * 8191 + "\n")
    rl = os.path.join(workdir, "long.rl_bwt")
    P.build_rlbwt(text, rl)
    _, lens = W.read_rlbwt_runs(rl)
    assert int(lens.max()) > 60000
    ri_path = os.path.join(workdir, "long.ri")
    P.build_rindex(rl, ri_path, True)
    idx, ri = P.Index(ri_path, mode=P.MODE_IMAGE_RL), O.RIndex(ri_path)
    assert ri.sigma == 6 and ri.has_N and idx.info().image_kind == P.IMAGE_RL
    emu = ImageEmu(idx)
    n = ri.n
    rng = np.random.default_rng(9)
    for pos in [0, 1, 2, 69999, 70000, 70001, n - 1, n] + [int(v) for v in rng.integers(0, n, 400)]:
        assert emu.rank6_true(pos) == ri.rank6_true(pos)
        assert emu.rank_cache(pos) == ri.rank_at_cached(pos)
    raw = open(text).read()
    for p in ["A" * 100, "N" * 39999, "GACGT", "CCCC", "TTTTTG"]:
        exp = sum(1 for i in range(len(raw) - len(p) + 1) if raw.startswith(p, i)) if len(p) < 200 else None
        tri = ri.bwd_pattern(p, O.MODE_STRICT)
        if exp is not None:
            assert tri[2] == exp, p
        assert emu.count(p) == ri.count(p)
    # the same index as bit planes: dense (64-byte blocks) is the automatic choice while it fits the memory-side cache, dense2 forced
    for force, kind in ((P.MODE_IMAGE_DENSE2, P.IMAGE_DENSE2), (0, P.IMAGE_DENSE)):
        idx2 = P.Index(ri_path, mode=force)
        assert idx2.info().image_kind == kind and not idx2.info().image_in_lds
        emu2 = ImageEmu(idx2)
        for pos in [0, 1, 63, 64, 65, 383, 384, 385, 69999, 70000, n - 1, n, n + 7] + [int(v) for v in rng.integers(0, n, 300)]:
            assert emu2.rank6_true(pos) == ri.rank6_true(min(pos, n))
        for p in ["A" * 100, "GACGT", "TTTTTG", "N" * 300, "NNNNA"]:
            assert emu2.count(p) == ri.count(p)
    # the N run (40 000 symbols) is a handful of exception runs per 384-symbol block, not one per symbol
    assert len(idx2.image_view(0)) // 64 > 100 and len(P.Index(ri_path, mode=P.MODE_IMAGE_DENSE2).image_view(15)) < 2 * (n // 384 + 1)


def test_image_layout_choice(workdir, built):
    """automatic layout: dense while the BWT is small, never on the legacy-quirk header, overridable"""
    xy = os.path.join(BT, "xy.ri")
    assert P.Index(xy).info().image_kind == P.IMAGE_RL            # legacy, no N, COMPAT: quirk slot in the header
    assert P.Index(xy, mode=P.MODE_STRICT).info().image_kind == P.IMAGE_DENSE
    assert P.Index(xy, mode=P.MODE_STRICT | P.MODE_IMAGE_RL).info().image_kind == P.IMAGE_RL
    with pytest.raises(P.PgxError) as e:
        P.Index(xy, mode=P.MODE_IMAGE_RL | P.MODE_IMAGE_DENSE)
    assert e.value.code == P.ERR_ARG
    with pytest.raises(P.PgxError) as e:
        P.Index(xy, mode=7)
    assert e.value.code == P.ERR_ARG


def test_tag_image_both_formats(workdir):
    bc = os.path.join(BT, "xy_bidirectional_compressed.tags")
    idx = P.Index(os.path.join(BT, "xy.ri"), bc)
    emu, t = ImageEmu(idx), O.Tags(bc, O.TAGS_BYTECODE)
    rng = np.random.default_rng(2)
    n = 8022
    qs = [(0, n - 1), (35, 35), (0, 0), (n - 1, n - 1), (8036, 8037)] + [
        (int(s), int(min(n - 1, s + l))) for s, l in zip(rng.integers(0, n, 300), rng.integers(0, 200, 300))]
    for st, en in qs:
        assert emu.tag_query(st, en) == t.query(st, en), (st, en)
    # compact sdsl format written by pgx_write_compact_tags
    ri_path, tags_path = W.build_index_from_rlbwt(os.path.join(G, "x.rl_bwt"), workdir, "x_tags")
    idx2 = P.Index(ri_path, tags_path)
    assert idx2.info().tag_format == P.TAGS_COMPACT
    emu2, t2 = ImageEmu(idx2), O.Tags(tags_path, O.TAGS_COMPACT)
    assert t2.n_runs == 1089 and t2.n_starts == 109
    for st, en in [(0, 3011), (0, 0), (3011, 3011)] + [(int(s), int(min(3011, s + l))) for s, l in
                                                        zip(rng.integers(0, 3012, 300), rng.integers(0, 300, 300))]:
        assert emu2.tag_query(st, en) == t2.query(st, en), (st, en)


def test_tag_overflow_is_defined(workdir):
    """queries whose first_bit_index is a multiple of 10 at the very end read one item past the
    stored runs in the reference (UB); both sides define that item as 0 and flag it"""
    vals = (np.arange(20, dtype=np.uint64) + 1) << np.uint64(11)
    lens = np.full(20, 3, dtype=np.uint64)
    path = os.path.join(workdir, "t20.tags")
    P.write_compact_tags(path, vals, lens)
    t = O.Tags(path, O.TAGS_COMPACT)
    assert t.n_runs == 20 and t.n_starts == 2
    rn, pos, over = t.query(59, 59)  # run number 20 -> 20 % 10 == 0 -> reads item 20 (absent)
    assert (rn, pos, over) == (1, [0], True)


def test_rejects_bad_files(workdir, built):
    bad = os.path.join(workdir, "bad.ri")
    open(bad, "wb").write(b"\x00\x0a\x03\x00" + b"\x00" * 64)  # like test_data/x.giraffe.ri: foreign r-index
    with pytest.raises(P.PgxError) as e:
        P.Index(bad)
    assert e.value.code == P.ERR_FORMAT and "Invalid tag" in str(e.value)
    with pytest.raises(P.PgxError) as e:
        P.Index(os.path.join(workdir, "does_not_exist.ri"))
    assert e.value.code == P.ERR_IO and "Cannot open r-index" in str(e.value)
    raw = open(os.path.join(BT, "xy.ri"), "rb").read()
    open(bad, "wb").write(raw[:20000])
    with pytest.raises(P.PgxError) as e:
        P.Index(bad)
    assert e.value.code == P.ERR_FORMAT
    with pytest.raises(P.PgxError) as e:
        P.Index(os.path.join(BT, "xy.ri"), os.path.join(BT, "xy_bidirectional.tags"))  # "algorithm format": not a query format
    assert e.value.code in (P.ERR_FORMAT, P.ERR_UNSUPPORTED)
