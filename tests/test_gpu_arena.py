"""The library's device arena (csrc/hip_util.hpp: DeviceArena) under a random allocation / free load, compiled from
tests/tools/arena_test.hip with hipcc on the GPU box: ranges never overlap, keep their content, coalesce back into whole chunks."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_arena_random_load(product_lib, tmp_path, seed):
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "arena_test"
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", str(ROOT / "matchtigs_amd" / "csrc"),
                        str(ROOT / "tests" / "tools" / "arena_test.hip"), "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe), str(seed), "3000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "arena_test ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
