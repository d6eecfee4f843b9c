"""Build the in-tree HIP shared library (gfx950 only; hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libastarpa_c_hip.so"

HIP_SOURCES = ["pa_hip.hip", "engine_hip.hip", "astarpa_c.hip", "pairs_io.hip", "apa2_simple_unit.hip", "apa2_full_unit.hip", "gcsh_build_unit.hip", "sketch_unit.hip", "slice_unit.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built")


def sources() -> list[Path]:
    return [CSRC / s for s in HIP_SOURCES if (CSRC / s).exists()]


STAMP_PATH = PKG_DIR / ".libastarpa_c_hip.hash"


def source_hash() -> str:
    """Content hash of everything the library is built from.  File times do not survive the copy to a GPU box in any useful order (a
    header edited after the last build made the box spend 50 s in hipcc before its first kernel), contents do."""
    import hashlib

    h = hashlib.sha256()
    for d in sorted(list(CSRC.glob("*")) + list((PKG_DIR.parent / "include").glob("*.h"))):
        if d.is_file():
            h.update(d.name.encode())
            h.update(d.read_bytes())
    h.update(os.environ.get("PA_HIPCC_EXTRA", "").encode())
    return h.hexdigest()


def kernel_hash() -> str:
    """Content hash of the headline kernels' sources (csrc/strip_kernel.hpp, csrc/slice_kernel.hpp): profiles/pmc_latest.json is stamped with it, and bench.py
    prints PMC-derived figures only when the counters were collected over this very code."""
    import hashlib

    return hashlib.sha256((CSRC / "strip_kernel.hpp").read_bytes() + (CSRC / "slice_kernel.hpp").read_bytes()).hexdigest()


def is_stale() -> bool:
    if not LIB_PATH.exists() or not STAMP_PATH.exists():
        return True
    return STAMP_PATH.read_text().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> libastarpa_c_hip.so next to this file."""
    if not force and not is_stale():
        return LIB_PATH
    # One hipcc per translation unit, side by side (they share no device symbols: no -fgpu-rdc), then one link: the four units take
    # 2.5 minutes one after the other and as long as the slowest (pa_hip.hip) in parallel.
    from concurrent.futures import ThreadPoolExecutor

    extra = os.environ.get("PA_HIPCC_EXTRA", "").split()  # experiments only (e.g. -DPA_PHASE_BARRIERS)
    objdir = PKG_DIR / "build"
    objdir.mkdir(exist_ok=True)
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", str(PKG_DIR.parent / "include")]

    import hashlib

    def unit_hash(depfile: Path) -> str | None:
        """Content hash of every file the unit was compiled from last time (its -MMD list), None if unknown."""
        if not depfile.exists():
            return None
        words = depfile.read_text().replace("\\\n", " ").split()
        deps = sorted({w for w in words[1:] if not w.endswith(":")})
        h = hashlib.sha256((" ".join(base + extra)).encode())
        for d in deps:
            try:
                h.update(d.encode())
                h.update(Path(d).read_bytes())
            except OSError:
                return None
        return h.hexdigest()

    def compile_one(src: Path) -> Path:
        # a unit whose sources (the files its last compilation read) did not change keeps its object: editing one kernel header
        # recompiles the units that include it, not the 100 s of pa_hip.hip
        obj = objdir / (src.stem + ".o")
        dep, stamp = objdir / (src.stem + ".d"), objdir / (src.stem + ".hash")
        if not force_all and obj.exists() and stamp.exists() and stamp.read_text().strip() == (unit_hash(dep) or "-"):
            return obj
        cmd = base + ["-MMD", "-MF", str(dep), "-c", str(src), "-o", str(obj)] + extra
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        stamp.write_text((unit_hash(dep) or "-") + "\n")
        return obj

    force_all = force and os.environ.get("PA_BUILD_INCREMENTAL", "") == ""
    with ThreadPoolExecutor(max_workers=len(HIP_SOURCES)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(o) for o in objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    STAMP_PATH.write_text(source_hash() + "\n")
    return LIB_PATH
