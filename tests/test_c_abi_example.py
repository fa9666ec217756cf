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
    cmd = ["gcc", "-std=c99", "-O2", "-g", "-rdynamic", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
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


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_example_reports_where_it_crashed(tmp_path):
    """The example is the library on the SYSTEM HIP runtime with no Python around it -- what a Julia `ccall` host gets.  If it
    ever dies of a signal the test below must fail with the place in its output, so the evidence path itself is tested here."""
    exe = _build(tmp_path)
    out = subprocess.run([exe, "--crash-selftest"], capture_output=True, text=True, timeout=60)
    assert out.returncode == -11
    assert "fatal signal 11" in out.stderr and "last ABI call entered: (crash self-test)" in out.stderr
    assert "main" in out.stderr.split("backtrace of the faulting thread:")[1]


def _run(exe, *args):
    env = dict(os.environ, HMG_EXAMPLE_TRACE="1")      # a breadcrumb per ABI call on stderr
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300, env=env)
    # no second start: a crash fails the test with what the program said about it (fatal-signal handler: fault address, last ABI
    # call entered, backtrace of the faulting thread)
    assert out.returncode == 0, f"capi_vcycle {' '.join(args)} -> {out.returncode}\n--- stdout\n{out.stdout}\n--- stderr\n{out.stderr}"
    assert "residual decreased: ok" in out.stdout
    return [float(l.split("|r| =")[1].split()[0]) for l in out.stdout.splitlines() if "|r| =" in l]


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_c_program_runs_vcycles(tmp_path):
    exe = _build(tmp_path)
    norms = _run(exe, "4", "4", "4")
    assert len(norms) == 4 and norms[-1] < 0.05 * norms[0]


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
@pytest.mark.parametrize("n,levels", [(6, 5), (3, 6)])
def test_c_program_runs_the_large_cell_kernels(tmp_path, n, levels):
    """Levels 5 (one wave per cell) and 6 (register-blocked, spare direction vector) through the plain C boundary as well."""
    exe = _build(tmp_path)
    norms = _run(exe, str(n), str(levels), "3")
    assert len(norms) == 3 and norms[-1] < 0.2 * norms[0]
