"""GPU parity of the semi-global search (ScatterProfile strip kernel) with the oracle and the reference's two
known answers (search.rs:30-31, pa_python/readme.md:13-16)."""
import numpy as np
import pytest

from tests.util_seq import rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


def test_search_known_answers(pa):
    assert pa.search(b"AC", b"CTTACTTA", 0.0) == [0, 0, 1, 2, 1, 0, 1, 2, 1, 0, 0]
    assert pa.search(b"CT", b"ACTG", 1.0) == [2, 2, 1, 0, 1, 2, 2]


def test_search_matches_oracle_random(pa, oracle):
    rng = np.random.default_rng(11)
    alphabet_p = b"ACGTNYR*acgtnyr"
    for _ in range(60):
        plen = int(rng.integers(1, 200))
        tlen = int(rng.integers(0, 1500))
        text = rand_seq(tlen, seed=int(rng.integers(1 << 30)))
        if rng.integers(0, 2):
            text = text.lower()
        pattern = bytes(alphabet_p[i] for i in rng.integers(0, len(alphabet_p), plen))
        uc = float(rng.choice([0.0, 0.25, 0.5, 1.0, 0.3]))
        assert pa.search(pattern, text, uc) == oracle.search(pattern, text, uc)


def test_search_multi_strip_pattern(pa, oracle):
    """pattern longer than one strip (2048 rows) => chained scatter strips."""
    pattern = rand_seq(5000, seed=3)
    text = rand_seq(300, seed=4) + pattern[100:4900] + rand_seq(200, seed=5)
    got = pa.search(pattern, text, 0.5)
    assert got == oracle.search(pattern, text, 0.5)


def test_search_rejects_bad_bases(pa):
    with pytest.raises(ValueError):
        pa.search(b"ACX", b"ACGT", 0.0)
    with pytest.raises(ValueError):
        pa.search(b"AC", b"ACGN", 0.0)


def _check_trace(pa, oracle, pattern, text, uc, idx):
    got = pa.search_trace(pattern, text, uc, idx)
    want = oracle.search_trace(pattern, text, uc, idx)
    assert got == want, (len(pattern), len(text), uc, idx)
    cigar, path = got
    # the path is a monotone walk that ends at idx's position, and the CIGAR spells its steps
    steps = sum(int(n or 1) for n, _ in __import__("re").findall(r"(\d*)([=XID])", cigar))
    assert len(path) == steps + 1
    for (i0, j0), (i1, j1) in zip(path, path[1:]):
        assert (i1 - i0, j1 - j0) in ((1, 1), (1, 0), (0, 1))


def test_search_trace_doc_example(pa, oracle):
    """search.rs:28-45: every output index of the documented example, against the restated trace (search.rs:125-228)."""
    for idx in range(11):
        _check_trace(pa, oracle, b"AC", b"CTTACTTA", 0.0, idx)
    assert pa.search_trace(b"AC", b"CTTACTTA", 0.0, 5) == ("2=", [(3, 0), (4, 1), (5, 2)])


def test_search_trace_matches_oracle_random(pa, oracle):
    rng = np.random.default_rng(23)
    alphabet_p = b"ACGTNYR*acgt"
    for _ in range(25):
        plen = int(rng.integers(1, 150))
        tlen = int(rng.integers(1, 1200))
        text = rand_seq(tlen, seed=int(rng.integers(1 << 30)))
        pattern = bytes(alphabet_p[i] for i in rng.integers(0, len(alphabet_p), plen))
        if rng.integers(0, 2) and tlen > plen + 10:  # plant a noisy copy so that good hits exist
            at = int(rng.integers(0, tlen - plen))
            core = bytes(c if c in b"ACGT" else b"ACGT"[k % 4] for k, c in enumerate(pattern.upper()))
            text = text[:at] + core + text[at + plen:]
        uc = float(rng.choice([0.0, 0.5, 1.0]))
        out = pa.search(pattern, text, uc)
        best = int(np.argmin(out[: tlen + 1]))
        for idx in {best, 0, tlen, tlen + plen, int(rng.integers(0, tlen + plen + 1))}:
            _check_trace(pa, oracle, pattern, text, uc, idx)


def test_search_trace_needs_wider_refill(pa, oracle):
    """A hit whose alignment is longer than 2|pattern| text characters (a long deletion) forces the width doubling."""
    pattern = rand_seq(40, seed=9)
    text = rand_seq(500, seed=10) + pattern[:20] + rand_seq(70, seed=12) + pattern[20:] + rand_seq(100, seed=11)
    out = pa.search(pattern, text, 1.0)
    idx = 500 + 20 + 70 + 20
    _check_trace(pa, oracle, pattern, text, 1.0, idx)
    _check_trace(pa, oracle, pattern, text, 1.0, int(np.argmin(out[: len(text) + 1])))
