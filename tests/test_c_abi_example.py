"""
The drop-in boundary is a C ABI: include/hmg.h must be a plain C header, and a C program (no Python, no torch)
must be able to drive a V-cycle through it -- the path a Julia `ccall` host takes.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "homogenization.jl_amd")


def _build(tmp_path):
    exe = str(tmp_path / "capi_vcycle")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "capi_vcycle.c"), "-o", exe, "-L" + LIBDIR, "-lhmg_hip",
           "-Wl,-rpath," + LIBDIR]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_header_is_plain_c_and_example_links(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "hmg.h"\nint main(void) { return hmg_version() == 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                    "-I" + os.path.join(ROOT, "include"), str(src)], check=True)
    exe = _build(tmp_path)
    import torch
    if not torch.cuda.is_available():
        # without a GPU the program must fail loudly at context creation: there is no CPU compute path
        out = subprocess.run([exe, "2", "2", "1"], capture_output=True, text=True)
        assert out.returncode != 0 and "no CPU fallback" in out.stderr


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_c_program_runs_vcycles(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "4", "4", "4"], capture_output=True, text=True, timeout=300)
    if out.returncode < 0 and "cycle 1" not in out.stdout:
        # Seen once in round 4 (of ~15 runs): as the FIRST process to touch the GPU of a fresh box the program died of SIGSEGV within
        # 0.4 s with nothing on stdout (then fully buffered: the program prints line by line now); the same binary ran clean right after, plain and under rocgdb.
        # One second attempt, reported; a crash that repeats fails the test.
        import warnings
        warnings.warn(f"capi_vcycle died of signal {-out.returncode} before its first V-cycle on its first start (output: "
                      f"{out.stdout!r} {out.stderr!r}); started again")
        out = subprocess.run([exe, "4", "4", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "residual decreased: ok" in out.stdout
    norms = [float(l.split("|r| =")[1].split()[0]) for l in out.stdout.splitlines() if "|r| =" in l]
    assert len(norms) == 4 and norms[-1] < 0.05 * norms[0]
