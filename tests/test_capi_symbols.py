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
