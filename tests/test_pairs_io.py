"""pa-bin's input/output formats (pa-bin/src/lib.rs:67-114, main.rs:24-35) behind the C ABI: parsing is host code and
runs without a GPU; the end-to-end `align_file` needs one."""
import pytest


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.capi.load()
    return pa


def test_seq_format(pa, tmp_path):
    f = tmp_path / "x.seq"
    f.write_text(">ACGT\n<ACGA\n>TT\r\n<T\r\n>dangling\n")
    assert pa.read_pairs(f) == [(b"ACGT", b"ACGA"), (b"TT", b"T")]  # CRLF stripped, odd last line dropped


def test_seq_format_requires_markers(pa, tmp_path):
    f = tmp_path / "bad.seq"
    f.write_text("ACGT\n<ACGA\n")
    with pytest.raises(pa.PaError):
        pa.read_pairs(f)


def test_txt_format(pa, tmp_path):
    f = tmp_path / "x.txt"
    f.write_text("ACGT\nACGA\n\nGG\n")
    assert pa.read_pairs(f) == [(b"ACGT", b"ACGA"), (b"", b"GG")]


def test_fasta_format(pa, tmp_path):
    f = tmp_path / "x.fa"
    f.write_text(">r1 first\nACGT\nAC\n>r2\nACGA\n>r3\nTTT\n>r4\n\nTT\nT\n>odd\nA\n")
    assert pa.read_pairs(f) == [(b"ACGTAC", b"ACGA"), (b"TTT", b"TTT")]
    for ext in ("fna", "fasta"):
        g = tmp_path / f"y.{ext}"
        g.write_text(">a\nAC\n>b\nAG\n")
        assert pa.read_pairs(g) == [(b"AC", b"AG")]


def test_reader_edge_cases(pa, tmp_path):
    """The reader works on views into the mapped file: no final newline, empty files, CRLF everywhere, sequence data before the first
    FASTA header, FASTA records compacted in place."""
    f = tmp_path / "nonl.txt"
    f.write_bytes(b"ACGT\nAC")  # no newline at the end: still a line
    assert pa.read_pairs(f) == [(b"ACGT", b"AC")]
    f = tmp_path / "empty.seq"
    f.write_bytes(b"")
    assert pa.read_pairs(f) == []
    f = tmp_path / "one.seq"
    f.write_bytes(b">ACGT")  # a lone first line is dropped unchecked
    assert pa.read_pairs(f) == []
    f = tmp_path / "crlf.fa"
    f.write_bytes(b">h1\r\nAC\r\nGT\r\n\r\n>h2\r\nA\r\n>h3\r\n>h4\r\nTTTT")
    assert pa.read_pairs(f) == [(b"ACGT", b"A"), (b"", b"TTTT")]
    f = tmp_path / "early.fa"
    f.write_bytes(b"ACGT\n>h\nAC\n")
    with pytest.raises(pa.PaError):
        pa.read_pairs(f)
    f = tmp_path / "blank_first.fa"
    f.write_bytes(b"\n\n>h\nAC\n>g\nAG\n")  # empty lines before the first header are no data
    assert pa.read_pairs(f) == [(b"AC", b"AG")]
    f = tmp_path / "bad2.seq"
    f.write_bytes(b">AC\n<AG\n>AC\nAG\n")
    with pytest.raises(pa.PaError, match="line 3"):
        pa.read_pairs(f)


def test_reader_large_file_equals_a_plain_parse(pa, tmp_path):
    import random

    rng = random.Random(7)
    lines = []
    want = []
    for i in range(3000):
        a = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 400)))
        b = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 400)))
        want.append((a, b))
        lines.append(b">" + a + (b"\r\n" if i % 7 == 0 else b"\n") + b"<" + b + b"\n")
    f = tmp_path / "big.seq"
    f.write_bytes(b"".join(lines))
    assert pa.read_pairs(f) == want
    fa = tmp_path / "big.fa"
    with fa.open("wb") as out:
        for i, (a, b) in enumerate(want):
            for r, s in (("a", a), ("b", b)):
                out.write(f">rec{i}{r}\n".encode())
                for k in range(0, len(s), 60):
                    out.write(s[k:k + 60] + b"\n")
    assert pa.read_pairs(fa) == want


def test_directory_and_unknown_extension(pa, tmp_path):
    d = tmp_path / "in"
    d.mkdir()
    (d / "b.txt").write_text("C\nG\n")
    (d / "a.seq").write_text(">A\n<T\n")
    assert pa.read_pairs(d) == [(b"A", b"T"), (b"C", b"G")]  # files in name order
    bad = tmp_path / "x.csv"
    bad.write_text("A\nC\n")
    with pytest.raises(pa.PaError):
        pa.read_pairs(bad)
    with pytest.raises(pa.PaError):
        pa.read_pairs(tmp_path / "missing.seq")


@pytest.mark.gpu
def test_align_file_writes_cost_cigar_lines(pa, oracle, tmp_path):
    from tests.util_seq import gen_pair

    pa.require_gpu()
    pairs = [gen_pair(n, 0.08, seed=n) for n in (50, 300, 1000, 2600)] + [(b"ACTCGCT", b"AACTCGTT")]
    f = tmp_path / "in.seq"
    f.write_text("".join(f">{a.decode()}\n<{b.decode()}\n" for a, b in pairs))
    out = tmp_path / "out.csv"
    assert pa.align_file(f, out) == len(pairs)
    lines = out.read_text().splitlines()
    assert len(lines) == len(pairs)
    for (a, b), line in zip(pairs, lines):
        cost, cigar = line.split(",")
        assert int(cost) == oracle.levenshtein(a, b)
        assert oracle.cigar_verify(cigar, a, b) == int(cost)
    assert lines[-1].startswith("2,")  # astarpa-c/example.c:8-29


@pytest.mark.gpu
def test_align_file_with_aligner_params(pa, oracle, tmp_path):
    """pa_align_file_params: pa-bin's loop with an aligner's parameters -- `simple` through the batched A*PA2, `full` (outside the
    batched family) through a loop over pa_align; every line is what the CPU-kernel engine returns for those parameters."""
    from tests.util_seq import gen_pair

    pa.require_gpu()
    pairs = [gen_pair(n, 0.08, seed=n) for n in (50, 300, 1000, 2600, 7000)] + [(b"ACTCGCT", b"AACTCGTT")]
    f = tmp_path / "in.seq"
    f.write_text("".join(f">{a.decode()}\n<{b.decode()}\n" for a, b in pairs))
    for prm, oprm in ((pa.AstarPa2Params.simple(), oracle.params_simple()), (pa.AstarPa2Params.full(), oracle.params_full())):
        out = tmp_path / "out.csv"
        assert pa.align_file(f, out, params=prm) == len(pairs)
        lines = out.read_text().splitlines()
        assert len(lines) == len(pairs)
        for (a, b), line in zip(pairs, lines):
            cost, cigar = line.split(",")
            want = oracle.cpu_align(a, b, oprm)
            assert (int(cost), cigar) == (want[0], want[1])
