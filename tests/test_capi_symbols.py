"""CPU-side checks: the C-ABI library builds for gfx950, loads, and exports every declared symbol."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared(header: Path):
    txt = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return set(re.findall(r"\b([a-z_0-9]+)\s*\(", txt)) - {"defined"}


def test_library_exports_every_declared_symbol():
    import astar_pairwise_aligner_amd as pa

    lib = pa.capi.load()
    declared = set()
    for h in (ROOT / "include").glob("*.h"):
        declared |= {s for s in _declared(h) if s.startswith(("pa_", "astarpa"))}
    assert declared, "no declarations found"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"symbols declared in include/*.h but not exported: {missing}"
    assert set(pa.capi.EXPORTED_SYMBOLS) <= declared | {"pa_align"}


def test_no_gpu_fails_loudly():
    import astar_pairwise_aligner_amd as pa

    if pa.capi.load().pa_device_count() > 0:
        return
    import pytest

    with pytest.raises(pa.PaError):
        pa.require_gpu()
    with pytest.raises(pa.PaError):
        pa.Batch([(b"ACGT", b"ACGT")])


# ---- the Rust mirrors in rust-shim/ (source only: no Rust toolchain here) against the C headers -------------------------
_C2RUST = {"int32_t": "i32", "uint64_t": "u64", "double": "f64", "float": "f32", "pa_block_params": "PaBlockParams"}


def _c_struct_fields(header: Path, name: str):
    txt = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", txt, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        fields += [(n.strip(), _C2RUST[ctype]) for n in names.split(",")]
    return fields


def _rust_struct_fields(src: Path, name: str):
    body = re.search(r"pub struct " + name + r"\s*\{(.*?)\n\}", src.read_text(), flags=re.S).group(1)
    return [(m.group(1), m.group(2)) for m in re.finditer(r"pub (\w+):\s*(\w+),", body)]


def test_rust_mirror_matches_header():
    """Every #[repr(C)] struct of rust-shim/astarpa2-hip mirrors include/pa_astarpa2.h field for field (names, order, widths)."""
    hdr = ROOT / "include" / "pa_astarpa2.h"
    rs = ROOT / "rust-shim" / "astarpa2-hip" / "src" / "lib.rs"
    for c_name, r_name in (("pa_block_params", "PaBlockParams"), ("pa_astarpa2_params", "PaAstarPa2Params"), ("pa_astarpa2_stats", "PaAstarPa2Stats")):
        assert _rust_struct_fields(rs, r_name) == _c_struct_fields(hdr, c_name), c_name
        assert "#[repr(C)]" in rs.read_text().split("pub struct " + r_name)[0].rsplit("///", 1)[-1]


def test_rust_extern_functions_are_exported():
    import astar_pairwise_aligner_amd as pa

    lib = pa.capi.load()
    for crate in ("astarpa2-hip", "pa-bitpacking-hip"):
        src = (ROOT / "rust-shim" / crate / "src" / "lib.rs").read_text()
        block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
        names = re.findall(r"pub fn (\w+)\(", block)
        assert names
        for n in names:
            assert hasattr(lib, n), (crate, n)


def test_c_layout_program_compiles():
    """tests/c_abi/layout_check.c: _Static_assert(offsetof(..)) for every field the Rust mirrors declare."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None:
        import pytest

        pytest.skip("no C compiler")
    for prog in ("layout_check.c", "link_check.c", "ctx_check.c", "batch_view_check.c", "dropin_threads.c"):
        subprocess.run([gcc, "-fsyntax-only", "-Wall", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_abi" / prog)], check=True)


def test_text_helpers_of_the_binding():
    """The two ways the Python layer turns the library's CIGAR texts into str: NUL-terminated strings behind an array of pointers
    (pa_batch_align; NULL -> "") and pointer + length pairs without a terminator (pa_batch_align_view)."""
    import ctypes as C

    from astar_pairwise_aligner_amd import capi

    texts = [b"12=X3I=", b"", b"=" * 5000, b"4=2D1X"]
    bufs = [C.create_string_buffer(t) for t in texts]
    ptrs = (C.c_void_p * 6)(*[C.addressof(b) for b in bufs], None, None)
    assert capi._c_strings(ptrs, 5) == [t.decode() for t in texts] + [""]
    packed = C.create_string_buffer(b"".join(texts), sum(len(t) for t in texts) + 1)  # back to back, no terminators in between
    off = 0
    for t in texts:
        assert capi._str_from_c_n(C.addressof(packed) + off, len(t)) == t.decode()
        off += len(t)
