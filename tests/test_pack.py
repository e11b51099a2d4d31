"""pgx_pack_reads (host side of pgx_batch_upload_packed; the find_mems CLI's parse threads call it): the batch as two bits per symbol, the order
of the seed index ((byte >> 1) & 3: A C T G = 0 1 2 3), and the reads with any other byte listed with their bytes.  CPU tier: against a numpy
restatement of the layout include/pgx.h states; the device side (same results as a byte upload) is tests/test_gpu_parity.py."""
import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P


def _reference(cat, offs):
    n_words = (len(cat) + 15) // 16
    codes = np.zeros(n_words * 16, dtype=np.uint32)
    codes[: len(cat)] = (cat.astype(np.uint32) >> 1) & 3
    packed = np.zeros(n_words, dtype=np.uint32)
    for k in range(16):
        packed |= codes[k::16] << np.uint32(2 * k)
    ok = np.isin(cat, np.frombuffer(b"ACGT", dtype=np.uint8))
    bad_before = np.concatenate(([0], np.cumsum(~ok)))
    ids = [i for i in range(len(offs) - 1) if bad_before[int(offs[i + 1])] - bad_before[int(offs[i])] > 0]
    side = b"".join(bytes(cat[int(offs[i]):int(offs[i + 1])]) for i in ids)
    return packed, np.array(ids, dtype=np.uint64), np.frombuffer(side, dtype=np.uint8), ok


def _pack(cat, offs, threads, id_cap=None, byte_cap=None):
    packed = np.zeros(max((len(cat) + 15) // 16, 1), dtype=np.uint32)
    ids = np.zeros(len(offs) if id_cap is None else id_cap, dtype=np.uint64)
    side = np.zeros(len(cat) + 1 if byte_cap is None else byte_cap, dtype=np.uint8)
    ns, nb = P.pack_reads(cat, offs, packed, ids, side, threads=threads)
    return packed, ids[:ns], side[:nb]


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_pack_reads_layout_and_side_list(built, threads):
    rng = np.random.default_rng(5)
    reads = []
    for i in range(5000):
        ln = int(rng.integers(0, 400))
        r = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, ln)].copy()
        if ln and i % 7 == 0:
            for _ in range(int(rng.integers(1, 4))):
                r[int(rng.integers(0, ln))] = int(rng.choice(np.frombuffer(b"Nacgt\x00\n$\xff", dtype=np.uint8)))
        reads.append(bytes(r))
    reads += [b"", b"N", b"A", b"", b"ACGTACGTACGTACGT", b"n" * 33]
    cat, offs = O.pack_reads(reads)
    want_p, want_ids, want_side, ok = _reference(cat, offs)
    packed, ids, side = _pack(cat, offs, threads)
    assert np.array_equal(ids, want_ids) and np.array_equal(side, want_side) and len(ids) > 500
    # words that hold only A C G T symbols (and padding) are defined; the others belong to listed reads and mean nothing
    word_ok = np.ones(len(want_p) * 16, dtype=bool)
    word_ok[: len(cat)] = ok
    word_ok = word_ok.reshape(-1, 16).all(axis=1)
    assert np.array_equal(packed[word_ok], want_p[word_ok]) and word_ok.sum() > 1000
    assert np.array_equal(packed, want_p)  # (in fact every symbol is packed as (byte >> 1) & 3, listed or not)


def test_pack_reads_edges(built):
    # empty batch, a batch of empty reads, offsets that do not start at 0 (the packed stream is relative to offsets[0])
    z = np.zeros(0, dtype=np.uint8)
    packed, ids, side = _pack(z, np.zeros(1, dtype=np.uint64), 1)
    assert len(ids) == 0 and len(side) == 0
    packed, ids, side = _pack(z, np.zeros(6, dtype=np.uint64), 1)
    assert len(ids) == 0
    cat = np.frombuffer(b"TTTTTTACGTNACGTACGTACGTACGTACGTAAC", dtype=np.uint8)
    offs = np.array([6, 11, 11, 34], dtype=np.uint64)
    want_p, want_ids, want_side, _ = _reference(cat[6:], offs - 6)
    packed, ids, side = _pack(cat, offs, 1)
    assert np.array_equal(packed[: len(want_p)], want_p) and list(ids) == [0] and bytes(side) == b"ACGTN"
    # a side list that does not fit: PGX_ERR_NOMEM, and the counts say what the batch needs
    cat2, offs2 = O.pack_reads([b"ACGN", b"ACGT", b"NNNN", b"acgt"])
    with pytest.raises(P.PgxError) as e:
        _pack(cat2, offs2, 1, id_cap=2)
    assert e.value.code == P.ERR_NOMEM
    with pytest.raises(P.PgxError) as e:
        _pack(cat2, offs2, 1, byte_cap=7)
    assert e.value.code == P.ERR_NOMEM
    _, ids, side = _pack(cat2, offs2, 1, id_cap=3, byte_cap=12)
    assert list(ids) == [0, 2, 3] and bytes(side) == b"ACGNNNNNacgt"
