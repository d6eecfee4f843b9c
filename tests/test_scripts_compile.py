"""Every Python script of the repository parses (bench.py and the measurement tools only run on the GPU box: a syntax slip there would
otherwise show up at the end of a round)."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) +
                 glob.glob(os.path.join(ROOT, "tests", "tools", "*.py")) + glob.glob(os.path.join(ROOT, "astar-pairwise-aligner_amd", "*.py")))


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_script_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


def test_bench_flags_without_a_gpu():
    """bench.py --help works on a box without a GPU (argument parsing comes before any device work)."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
