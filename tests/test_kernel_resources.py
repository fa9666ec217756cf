"""Register budget of the benchmarked apply instantiations (cross-compiled for gfx950, no GPU needed).  Three 512-thread
workgroups per CU leave 80 VGPRs per lane (MI355X_MICROARCH.md, register table: 6 waves per SIMD); the hot instantiations
must fit without scratch -- one more live value in a shared code path once cost the main fused kernel three spilled
registers without any test noticing (round 3)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

# template arguments <DIM, NT, SPT, FUSED, RB, WD, CG, RS> as they appear in the mangled names
BUDGET = {
    "Li3ELi512ELi13ELb0ELi6ELb0ELb0ELb0E": (80, 0),     # level 6, plain apply / residual
    "Li3ELi512ELi13ELb1ELi6ELb0ELb0ELb0E": (80, 0),     # level 6, fused CG passes (6 of the 9 finest-level launches)
    "Li3ELi512ELi13ELb1ELi6ELb0ELb1ELb0E": (80, 0),     # level 6, residual with the coarse-grid correction staged in the image
    "Li3ELi512ELi13ELb1ELi6ELb0ELb0ELb1E": (80, 16),    # level 6, local residual with the restriction in its epilogue
    "Li3ELi256ELi4ELb1ELi4ELb0ELb0ELb0E": (80, 0),      # level 5, fused
    "Li3ELi256ELi4ELb1ELi4ELb0ELb0ELb1E": (80, 0),      # level 5, restriction in the epilogue
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_instantiations_fit_their_register_budget(tmp_path):
    src = os.path.join(ROOT, "homogenization.jl_amd", "csrc", "hmg_kernels.hip")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", src,
                          "-o", str(tmp_path / "k.o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    found = {}
    for m in re.finditer(r"Function Name: _ZN3hmg7k_applyI(\w+?)EEvNS_8LevelDev.*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)",
                         out.stderr, re.S):
        found[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    for name, (vgprs, scratch) in BUDGET.items():
        assert name in found, f"instantiation {name} not compiled; have {sorted(found)[:5]} ..."
        assert found[name][0] <= vgprs, (name, found[name])
        assert found[name][1] <= scratch, (name, found[name])
