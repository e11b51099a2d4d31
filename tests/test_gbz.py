"""GBZ reader for merge_tags (reference src/merge_tags.cpp:443-445,478-515): first node of every path and
weakly connected components, read from the GBWT records inside the reference's GBZ fixtures (test_data/**/*.gbz), then the
whole merge driven by the graph: `merge_tags <graph.gbz> <whole.ri> <tag_dir>` with the reference's own graph, texts and r-index."""
import os
import subprocess

import numpy as np
import pytest

import gbz_emu
import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W

G = O.GOLDEN
BT = os.path.join(G, "bidirectional_test")
FIXTURES = [os.path.join(BT, "xy.gbz"), os.path.join(BT, "x.gbz"), os.path.join(BT, "y.gbz"), os.path.join(G, "x.giraffe.gbz")]


@pytest.mark.parametrize("path", FIXTURES)
def test_gbwt_records_walk_every_path(path):
    """the restated record format is self-consistent on bytes the reference's toolchain wrote: every sequence walked with LF
    through the decoded records reaches the endmarker, the walks add up to header.size, and sequence 2k+1 is sequence 2k
    reversed with flipped orientations (bidirectional GBWT)"""
    g = gbz_emu.parse_gbwt(path)
    paths = [gbz_emu.walk(g, s) for s in range(g["nseq"])]
    assert sum(len(p) + 1 for p in paths) == g["size"] and g["nseq"] % 2 == 0 and all(paths)
    for k in range(g["nseq"] // 2):
        assert [x ^ 1 for x in reversed(paths[2 * k])] == paths[2 * k + 1]


@pytest.mark.parametrize("path", FIXTURES)
def test_library_reader_matches_restatement(built, path):
    g = gbz_emu.parse_gbwt(path)
    comp = gbz_emu.components(g)
    first, component, max_node, n_comp = P.gbz_paths(path)
    assert len(first) == g["nseq"] and n_comp == len(set(comp.values())) and max_node == max(comp)
    for s in range(g["nseq"]):
        node = gbz_emu.walk(g, s)[0] // 2
        assert int(first[s]) == node and int(component[s]) == comp[node]


def test_known_graphs(built, tmp_path):
    first, component, max_node, n_comp = P.gbz_paths(os.path.join(BT, "xy.gbz"))
    # two contigs: x = nodes 1..69, y = nodes 70..138; forward paths start at the first node, reverse paths at the last
    assert list(first) == [1, 69, 1, 69, 70, 138, 70, 138] and list(component) == [0, 0, 0, 0, 1, 1, 1, 1]
    assert (max_node, n_comp) == (138, 2)
    assert O.RIndex(os.path.join(BT, "xy.ri")).C_array()[1] == len(first)  # tot_strings == GBWT sequences (merge_tags.cpp:500-515)
    first, component, max_node, n_comp = P.gbz_paths(os.path.join(BT, "y.gbz"))  # GBWT with a node offset (139)
    assert list(first) == [70, 138, 70, 138] and n_comp == 1 and max_node == 138
    with pytest.raises(P.PgxError) as e:
        P.gbz_paths(os.path.join(BT, "xy.ri"))
    assert e.value.code == P.ERR_FORMAT
    raw = open(os.path.join(BT, "xy.gbz"), "rb").read()
    for cut in (10, 200, 430, 1000, len(raw) - 3000):  # truncations must be rejected, never crash
        bad = str(tmp_path / "trunc.gbz")
        open(bad, "wb").write(raw[:cut])
        with pytest.raises(P.PgxError):
            P.gbz_paths(bad)


def _g(seq, off, node_base):
    """synthetic tag of suffix (seq, off): a node of the sequence's own graph component"""
    return ((node_base + 1 + off // 16) << 11) | ((seq & 1) << 10) | (off % 16)


@pytest.mark.gpu
def test_merge_tags_with_the_reference_argv(workdir):
    """chromosomes x and y of the reference's two-contig graph: per-chromosome tag streams along each chromosome's own suffix
    array (indexes built from the reference's contigs_x / contigs_y .rl_bwt), merged along the reference's whole-genome xy.ri
    with the sequence -> file map taken from the reference's xy.gbz"""
    from test_merge_tags import _write_algorithm_tags

    tag_dir = os.path.join(workdir, "gbz_tags")
    os.makedirs(tag_dir, exist_ok=True)
    whole = O.RIndex(os.path.join(BT, "xy.ri"))
    n_seq = whole.C_array()[1]
    for name, seq_base, node_base in (("contigs_y", 4, 69), ("contigs_x", 0, 0)):  # (file order must not matter)
        ri_c, _ = W.build_index_from_rlbwt(os.path.join(BT, name + ".rl_bwt"), workdir, "gbz_" + name, with_tags=False)
        r = O.RIndex(ri_c)
        sa, ml = r.decompress_sa(), r.max_length
        tags = [_g(seq_base + int(v) // ml, int(v) % ml, node_base) for v in sa[4:]]  # four sequences per contig
        _write_algorithm_tags(os.path.join(tag_dir, name + ".tags"), tags, header=(name == "contigs_x"))
    sa, ml = whole.decompress_sa(), whole.max_length
    expected = [0] * n_seq + [_g(int(v) // ml, int(v) % ml, 0 if int(v) // ml < 4 else 69) for v in sa[n_seq:]]
    exe = os.path.join(os.path.dirname(os.path.abspath(P.__file__)), "merge_tags")
    out = os.path.join(workdir, "gbz_whole.tags")
    r = subprocess.run([exe, os.path.join(BT, "xy.gbz"), os.path.join(BT, "xy.ri"), tag_dir, "--out", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Index files merged and ready to use!" in r.stderr, r.stderr
    t = O.Tags(out, O.TAGS_COMPACT)
    got = []
    for k in range(t.n_runs):
        a = t.L.orc_tags_interval(t.h, k)
        b = t.L.orc_tags_interval(t.h, k + 1) if k + 1 < t.n_runs else whole.n
        got += [t.L.orc_tags_item(t.h, k)] * (b - a)
    assert got == expected
    # the item width comes from the graph's largest node id (138 -> 8 bits + 11), merge_tags.cpp:627-638
    width = open(out, "rb").read()[8]
    assert width == 19
    # default output name in the working directory, like the reference (:538)
    r = subprocess.run([exe, os.path.join(BT, "xy.gbz"), os.path.join(BT, "xy.ri"), tag_dir], capture_output=True, text=True, timeout=300, cwd=workdir)
    assert r.returncode == 0 and open(os.path.join(workdir, "whole_genome_tag_array_compressed.tags"), "rb").read() == open(out, "rb").read()
    # a graph whose paths are not those of the r-index is refused
    r = subprocess.run([exe, os.path.join(BT, "x.gbz"), os.path.join(BT, "xy.ri"), tag_dir, "--out", out + ".bad"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
