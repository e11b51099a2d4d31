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
