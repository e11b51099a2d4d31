"""The LCE image as the device builds it (pgx_runtime.hip ensure_lce: pgx_lce_scatter / pack / lcp kernels; pgx_image.h "LCE image") against the same arrays made
on the CPU from the oracle's suffix array and the text file: suffix array in text coordinates, text at two bits per symbol, line flags, common prefixes of
neighbouring suffixes -- bit for bit (what find_all_mems' forward stages, algorithm.hpp:676-700, are answered from on the text path)."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_workload as W
import pgx_ffi as P
from test_lce_math import _lcp_table

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("haps", [5, 24])
def test_lce_image_equals_the_cpu_construction(workdir, haps):
    text = os.path.join(workdir, "lce_img_%d.txt" % haps)
    W.synth_pangenome_text(text, base_len=400_000 // haps, n_hap=haps, seed=21 + haps, n_runs=3, n_run_len=(30, 700))
    ri_path, tags_path = W.build_index_from_text(text, workdir, "lce_img_%d" % haps)[:2]
    ri = O.RIndex(ri_path)
    seqs = W.load_sequences(text)
    ml, n = ri.max_length, ri.n
    sa = ri.decompress_sa()
    seq_start = np.concatenate([[0], np.cumsum([len(s) + 1 for s in seqs])])
    T = np.concatenate([np.concatenate([s, [10]]) for s in seqs]).astype(np.uint8)
    assert len(T) == n
    gpos = (seq_start[(sa // ml).astype(np.int64)] + (sa % ml).astype(np.int64)).astype(np.int64)
    idx = P.Index(ri_path, tags_path, mode=P.MODE_COMPAT | P.MODE_IMAGE_PAIRS)
    n_words = (n + 15) // 16 + 64
    d_sa = idx.lce_view(30, 4 * n).view(np.uint32)
    assert np.array_equal(d_sa.astype(np.int64), gpos)
    # text: A C T G = 0 1 2 3, anything else (and what lies behind the text) reads as 3 and flags its line
    code = np.full(256, 3, dtype=np.uint32)
    for c, v in zip(b"ACTG", range(4)):
        code[c] = v
    sym = np.full(16 * n_words, 3, dtype=np.uint32)
    sym[:n] = code[T]
    words = (sym.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
    assert np.array_equal(idx.lce_view(31, 4 * n_words).view(np.uint32), words)
    special = np.ones(16 * n_words, dtype=bool)
    special[:n] = ~np.isin(T, np.frombuffer(b"ACGT", dtype=np.uint8))
    n_lines = (n_words + 31) // 32
    line_bad = np.zeros(32 * ((n_words // 1024) + 2), dtype=bool)
    lb = np.add.reduceat(special.astype(np.int64), np.arange(0, 16 * n_words, 512)) > 0
    line_bad[:len(lb)] = lb
    assert len(lb) == n_lines
    flags = np.packbits(line_bad.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
    assert np.array_equal(idx.lce_view(32, 4 * len(flags)).view(np.uint32), flags)
    lcp = _lcp_table(T, gpos)
    d_lcp = idx.lce_view(33, n)
    bad = np.flatnonzero(d_lcp.astype(np.int64) != lcp)
    assert len(bad) == 0, (len(bad), bad[:5], d_lcp[bad[:5]], lcp[bad[:5]])
    assert (lcp == 254).any() and (lcp == 255).any() and (lcp < 20).any()
    idx.close()
