"""The reference's own hot-path property tests, run as written (reference tests/test_rindex.cpp:288-486):

  FMDINDEX_Test.BackwardExtensionMatchesLF[_Encoded]                      :288-373
      walking a 72-mer right to left, backward_extend[_encoded](bint, c).{forward, size} must equal the range
      LF[_encoded]({bint.forward, bint.forward + bint.size - 1}, c) whenever that range is non-empty
  FMDINDEX_Test.CompareSampledKmersWithReverseComplementsBIGTEST[_Encoded] :376-486
      100 sampled 12-mers over ACGTN (std::mt19937(42)): I(kmer).forward == I(revcomp).reverse,
      I(kmer).reverse == I(revcomp).forward, sizes equal

The reference runs them on test_data/big_test/merged_info{,.rl_bwt}, which is not in its repository;
here the same procedures run in COMPAT mode on a sigma = 6 both-strands index (the shape the reference's fixture has:
"ACGTN" k-mers, reverse complements present), legacy and encoded layout, once against the oracle (CPU tier) and once
through the device primitives pgx_extend_batch / pgx_lf_batch / pgx_count_batch (GPU tier).  The sampler is numpy's
MT19937 seeded with 42 (the same generator and seed; libstdc++'s uniform_int_distribution mapping is not reproduced).
"""
import os

import numpy as np
import pytest

import oracle_ffi as O
import pgx_ffi as P
import pgx_workload as W

# the reference's literal 72-mer (test_rindex.cpp:293); it does not occur in the synthetic text, so -- like every range the
# reference test meets after its k-mer leaves the index -- the comparison is skipped once LF reports an empty range
REF_KMER = "ATCAAAGAAAAAAGCCCAACATATCCATTACCATTACTAGTTACACATAGCATCAGGAACCAGAGAGTTGGA"
_COMP = {ord("A"): ord("T"), ord("C"): ord("G"), ord("G"): ord("C"), ord("T"): ord("A")}


@pytest.fixture(scope="module")
def fmd(workdir):
    """sigma = 6 pangenome text, both strands; encoded and legacy .ri built from the same BWT"""
    text = os.path.join(workdir, "fmdprop.txt")
    W.synth_pangenome_text(text, base_len=80000, n_hap=4, seed=5, n_runs=3, n_run_len=(50, 1500))
    enc, _, rl = W.build_index_from_text(text, workdir, "fmdprop", with_tags=False)
    leg, _ = W.build_index_from_rlbwt(rl, workdir, "fmdprop", encoded=False, with_tags=False)
    full_text = "".join(l for l in open(text).read().split("\n") if l)  # test_rindex.cpp:386-390
    return dict(enc=enc, leg=leg, text=full_text)


def _kmers72(full_text):
    rng = np.random.RandomState(7)
    out = [REF_KMER]
    while len(out) < 6:
        p = int(rng.randint(0, len(full_text) - 72))
        out.append(full_text[p:p + 72])
    return out


def _sample_12mers(full_text, k=12, num=100):
    """test_rindex.cpp:392-407: positions uniform in [0, size - k], k-mers with symbols outside ACGTN are re-drawn"""
    rng = np.random.RandomState(42)  # MT19937, seed 42
    out = []
    while len(out) < num:
        pos = int(rng.randint(0, len(full_text) - k + 1))
        kmer = full_text[pos:pos + k]
        if any(c not in "ACGTN" for c in kmer):
            continue
        out.append(kmer)
    return out


def _revcomp(kmer):
    return "".join(chr(_COMP.get(ord(c), ord(c))) for c in reversed(kmer))  # FastLocate::complement: identity outside ACGT


# ---------------------------------------------------------------------------------------------------------------------
# CPU tier: the oracle's restated backward_extend[_encoded] / LF[_encoded]
@pytest.mark.parametrize("layout", ["enc", "leg"])
def test_backward_extension_matches_lf_oracle(fmd, layout):
    ri = O.RIndex(fmd[layout])
    assert ri.sigma == 6 and ri.encoded == (layout == "enc")
    checked = 0
    for kmer in _kmers72(fmd["text"]):
        bint = (0, 0, ri.n)  # test_rindex.cpp:304
        for c in reversed(kmer):
            fwd_expected = ri.LF((bint[0], (bint[0] + bint[2] - 1) % 2**64), c)  # :311 / :354
            expected_size = fwd_expected[1] - fwd_expected[0] + 1 if fwd_expected[0] <= fwd_expected[1] else 0
            extended = ri.bwd(bint, c)
            if expected_size > 0:  # :328-331
                assert extended[2] == expected_size and extended[0] == fwd_expected[0], (kmer, c)
                checked += 1
            bint = extended
    assert checked > 5 * 72  # the five sampled 72-mers stay in the index to their last symbol


@pytest.mark.parametrize("layout", ["enc", "leg"])
def test_sampled_kmers_fmd_symmetry_oracle(fmd, layout):
    ri = O.RIndex(fmd[layout])
    with_n = 0
    for kmer in _sample_12mers(fmd["text"]):
        rc = _revcomp(kmer)
        a, b = ri.bwd_pattern(kmer), ri.bwd_pattern(rc)  # :415-425
        assert a[0] == b[1] and a[1] == b[0] and a[2] == b[2] and a[2] > 0, (kmer, rc, a, b)  # :430-432
        with_n += "N" in kmer
    assert with_n > 0  # the N runs are sampled too


# ---------------------------------------------------------------------------------------------------------------------
# GPU tier: the same procedures through the C ABI
def _chain(idx, pats):
    """backward-extend every pattern right to left from the full interval; returns the interval after every step"""
    n = idx.info().bwt_size
    iv = np.zeros(len(pats), dtype=P.BIINT_DTYPE)
    iv["size"] = n
    steps = []
    L = len(pats[0])
    for t in range(L):
        syms = np.array([ord(p[L - 1 - t]) for p in pats], dtype=np.uint8)
        steps.append((iv.copy(), syms))
        iv = idx.extend_batch(iv, syms, np.zeros(len(pats), dtype=np.uint8))
    return steps, iv


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["enc", "leg"])
def test_backward_extension_matches_lf_device(fmd, layout):
    ri = O.RIndex(fmd[layout])
    kmers = _kmers72(fmd["text"])
    for force in (P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL, P.MODE_IMAGE_DENSE2):
        idx = P.Index(fmd[layout], mode=P.MODE_COMPAT | force)
        steps, last = _chain(idx, kmers)
        checked = 0
        for t, (iv, syms) in enumerate(steps):
            rng = np.stack([iv["forward"], iv["forward"] + iv["size"].astype(np.uint64) - np.uint64(1)], axis=1)
            lf = idx.lf_batch(rng, syms)  # LF[_encoded] on the device
            ext = steps[t + 1][0] if t + 1 < len(steps) else last
            for q in range(len(kmers)):
                # device primitives == oracle
                tri = (int(iv["forward"][q]), int(iv["reverse"][q]), int(iv["size"][q]))
                assert (int(ext["forward"][q]), int(ext["reverse"][q]), int(ext["size"][q])) == ri.bwd(tri, int(syms[q]))
                assert (int(lf[q][0]), int(lf[q][1])) == ri.LF((int(rng[q][0]), int(rng[q][1])), int(syms[q]))
                # the reference's assertion
                if lf[q][0] <= lf[q][1]:
                    assert int(ext["size"][q]) == int(lf[q][1] - lf[q][0] + 1) and int(ext["forward"][q]) == int(lf[q][0])
                    checked += 1
        assert checked > 5 * 72
        # count[_encoded] of every suffix of the k-mer is the same chain of LF steps (r-index.hpp:540-556)
        sufs = [k[i:] for k in kmers[1:] for i in range(0, 72, 7)]
        cat, offs = O.pack_reads(sufs)
        got = idx.count_batch(cat, offs)
        for s, g in zip(sufs, got):
            assert (int(g[0]), int(g[1])) == ri.count(s)
            tri = ri.bwd_pattern(s)
            assert tri[2] == int(g[1]) - int(g[0]) + 1 and tri[0] == int(g[0])
        idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["enc", "leg"])
def test_sampled_kmers_fmd_symmetry_device(fmd, layout):
    kmers = _sample_12mers(fmd["text"])
    rcs = [_revcomp(k) for k in kmers]
    for force in (P.MODE_IMAGE_DENSE, P.MODE_IMAGE_RL, P.MODE_IMAGE_DENSE2):
        idx = P.Index(fmd[layout], mode=P.MODE_COMPAT | force)
        _, a = _chain(idx, kmers)
        _, b = _chain(idx, rcs)
        assert np.array_equal(a["forward"], b["reverse"]) and np.array_equal(a["reverse"], b["forward"])
        assert np.array_equal(a["size"], b["size"]) and (a["size"] > 0).all()
        idx.close()
