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
