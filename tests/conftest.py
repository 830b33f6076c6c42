import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running emulator test")


def hdu_pkg():
    return importlib.import_module("h-denseunet_amd")


def _ensure_emulator():
    import emu_bind
    path = emu_bind.emulator_library_path()
    srcs = [os.path.join(ROOT, "h-denseunet_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "h-denseunet_amd", "csrc"))]
    srcs += [os.path.join(ROOT, "tests", "hipemu", f) for f in ("hipemu.h", "hipemu.cpp")]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call([os.path.join(ROOT, "build.sh"), "emu"])
    return path


@pytest.fixture
def emu_lib():
    """Bind the x86 emulator build of the kernel sources (test infrastructure)."""
    _ensure_emulator()
    import emu_bind
    pkg = hdu_pkg()
    emu_bind.use_emulator()
    return pkg


@pytest.fixture
def hip_lib():
    """Bind the gfx950 product library; fails loudly if it is missing or no GPU is visible."""
    import torch
    pkg = hdu_pkg()
    pkg.lib.load()
    assert pkg.lib.backend() == "hip-gfx950"
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    return pkg


def backend_params():
    return [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=backend_params())
def hdu(request):
    if request.param == "emu":
        return request.getfixturevalue("emu_lib")
    return request.getfixturevalue("hip_lib")


@pytest.fixture(params=[2, 6], ids=["dma2", "ring6"])
def dma_stages(request):
    """run a test under both LDS ring depths of the DMA implicit GEMM"""
    lib = hdu_pkg().lib
    yield request.param
