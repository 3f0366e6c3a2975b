"""The C ABI is usable without torch: build and run tools/c_abi_demo.cpp against libmonoforce_hip.so."""
import os
import subprocess

import pytest

from tests.conftest import REPO

pytestmark = pytest.mark.gpu


def test_torch_free_c_client(tmp_path):
    exe = str(tmp_path / 'c_abi_demo')
    lib_dir = os.path.join(REPO, 'monoforce_amd', 'csrc')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', os.path.join(REPO, 'tools', 'c_abi_demo.cpp'),
                    '-I' + os.path.join(REPO, 'include'), '-L' + lib_dir, '-lmonoforce_hip', '-Wl,-rpath,' + lib_dir, '-o', exe],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'C ABI demo ok' in out.stdout, out.stdout + out.stderr
