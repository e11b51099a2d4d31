"""GPU parity of the tag path on segments of every size class, incl. > 16384 runs (global scratch)."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P

pytestmark = pytest.mark.gpu


def test_tag_query_all_size_classes(workdir, x_index):
    rng = np.random.default_rng(21)
    n_runs = 70000
    vals = (rng.integers(1, 3000, n_runs).astype(np.uint64) << np.uint64(11)) | rng.integers(0, 1024, n_runs).astype(np.uint64)
    lens = rng.integers(1, 4, n_runs).astype(np.uint64)
    path = os.path.join(workdir, "huge.tags")
    P.write_compact_tags(path, vals, lens)
    total = int(lens.sum())
    idx = P.Index(x_index[0], path)
    t = O.Tags(path, O.TAGS_COMPACT)
    st = np.array([0, 0, 5, 100, 1000, 17, 0, 3, 40000, total - 1, total - 50], dtype=np.uint64)
    en = np.array([total - 1, 60000, 5, 130, 9000, 60, 20, 3, 40100, total - 1, total - 1], dtype=np.uint64)
    extra_s = rng.integers(0, total - 1, 200).astype(np.uint64)
    extra_l = np.concatenate([rng.integers(0, 40, 100), rng.integers(40, 5000, 60), rng.integers(5000, total, 40)]).astype(np.uint64)
    st = np.concatenate([st, extra_s])
    en = np.concatenate([en, np.minimum(extra_s + extra_l, np.uint64(total - 1))])
    # identical large queries are answered once and copied (dedup path): repeat some of them
    st = np.concatenate([st, st[:2], st[:2], st[-40:], st[-40:]])
    en = np.concatenate([en, en[:2], en[:2], en[-40:], en[-40:]])
    rn, po, pos, nover = idx.tag_query_batch(st, en)
    classes = set()
    for i in range(len(st)):
        ern, epos, eover = t.query(int(st[i]), int(en[i]))
        assert int(rn[i]) == ern, i
        assert np.array_equal(pos[po[i]:po[i + 1]], np.array(epos, dtype=np.uint64)), (i, ern)
        classes.add(0 if ern <= 16 else 1 if ern <= 64 else 2 if ern <= 2048 else 3 if ern <= 16384 else 4)
    assert classes == {0, 1, 2, 3, 4}


def test_tag_buckets_sparse_and_crowded(workdir, x_index):
    """the bucket lines of the locate kernel (pgx_tag_bucket_kernel: ~4 runs per 128-byte line): a tag array of long runs with a cluster of 2 000
    one-position runs in the middle, so that the cluster's buckets hold far more than the ten runs a line takes (flagged: answered through
    tdir / tpair) while their neighbours hold one or none; one-position queries at every position around the cluster, intervals that start in
    one kind of bucket and end in the other, intervals that reach the end of the array"""
    rng = np.random.default_rng(31)
    lens = np.concatenate([np.full(500, 1000), np.ones(2000, dtype=np.int64), np.full(500, 1000), rng.integers(1, 30, 3000)]).astype(np.uint64)
    vals = (rng.integers(1, 5000, len(lens)).astype(np.uint64) << np.uint64(11)) | rng.integers(0, 1024, len(lens)).astype(np.uint64)
    path = os.path.join(workdir, "buckets.tags")
    P.write_compact_tags(path, vals, lens)
    total = int(lens.sum())
    idx = P.Index(x_index[0], path)
    t = O.Tags(path, O.TAGS_COMPACT)
    c0 = 500 * 1000  # first position of the cluster
    around = np.arange(c0 - 1200, c0 + 3300, dtype=np.uint64)
    st = np.concatenate([around, around, rng.integers(0, total - 1, 1500).astype(np.uint64),
                         np.array([0, total - 1, total - 1, c0 - 5000, c0 + 1990, 1000 * 1000 + 2000 - 1], dtype=np.uint64)])
    ln = np.concatenate([np.zeros(len(around), dtype=np.uint64), rng.integers(0, 2600, len(around)).astype(np.uint64),
                         np.concatenate([rng.integers(0, 12, 700), rng.integers(12, 3000, 800)]).astype(np.uint64),
                         np.array([0, 0, 50, 9000, 4000, 40], dtype=np.uint64)])
    en = np.minimum(st + ln, np.uint64(total - 1))
    rn, po, pos, nover = idx.tag_query_batch(st, en)
    for i in range(len(st)):
        ern, epos, eover = t.query(int(st[i]), int(en[i]))
        assert int(rn[i]) == ern, (i, int(st[i]), int(en[i]))
        assert np.array_equal(pos[po[i]:po[i + 1]], np.array(epos, dtype=np.uint64)), (i, int(st[i]), int(en[i]), ern)
