"""The C-ABI shared library loads and exports every symbol include/hdu.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "hdu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(hdu_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_what_python_binds():
    import importlib
    lib = importlib.import_module("h-denseunet_amd.lib")
    assert set(lib.EXPORTS) == set(declared_symbols())


@pytest.mark.parametrize("which", ["hip", "emu"])
def test_library_exports_every_declared_symbol(which):
    import importlib
    lib = importlib.import_module("h-denseunet_amd.lib")
    import emu_bind
    path = lib.product_library_path() if which == "hip" else emu_bind.emulator_library_path()
    if not os.path.exists(path):
        subprocess.check_call([os.path.join(ROOT, "build.sh"), which])
    import torch  # noqa: F401  (one shared HIP runtime, see lib._bind)
    so = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(so, name), name
    so.hdu_backend.restype = ctypes.c_char_p
    assert so.hdu_backend().decode() == ("hip-gfx950" if which == "hip" else "emu-x86")


def test_product_path_fails_loudly_without_library(tmp_path):
    import importlib
    lib = importlib.import_module("h-denseunet_amd.lib")
    with pytest.raises(lib.HduError, match="no CPU fallback"):
        lib.load(str(tmp_path / "missing_libhdu.so"))


def test_product_loader_refuses_the_emulator_library(emu_lib):
    """lib.load() binds the gfx950 build only: handed the x86 emulator build of the same ABI it raises instead of becoming a CPU path"""
    import importlib
    import emu_bind
    lib = importlib.import_module("h-denseunet_amd.lib")
    with pytest.raises(lib.HduError, match="not the gfx950 product library"):
        lib.load(emu_bind.emulator_library_path())
    assert lib.backend() == "emu-x86"          # (the failed load left the test binding in place)


def test_product_library_refuses_cpu_storage():
    """with the gfx950 library bound and no GPU visible the ops refuse to run (no silent CPU path)"""
    import importlib
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = importlib.import_module("h-denseunet_amd.lib")
    ops = importlib.import_module("h-denseunet_amd.ops")
    lib.load()
    try:
        with pytest.raises(lib.HduError, match="no CPU fallback"):
            ops.device()
    finally:
        import emu_bind
        emu_bind.use_emulator()


def test_binding_refuses_a_library_of_another_abi(monkeypatch):
    """hdu_abi_version() / hdu_sizeof_conv_desc() must match the ctypes mirror: a stale libhdu.so (built before a struct
    gained a field) is refused at load time instead of being driven through a descriptor of another layout"""
    import importlib
    lib = importlib.import_module("h-denseunet_amd.lib")
    import emu_bind
    so = ctypes.CDLL(emu_bind.emulator_library_path())
    so.hdu_sizeof_conv_desc.restype = ctypes.c_size_t
    assert so.hdu_abi_version() == lib.ABI_VERSION and so.hdu_sizeof_conv_desc() == ctypes.sizeof(lib.ConvDesc)
    hdr = open(os.path.join(ROOT, "include", "hdu.h")).read()
    assert int(re.search(r"#define HDU_ABI_VERSION (\d+)", hdr).group(1)) == lib.ABI_VERSION
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION + 1)
    with pytest.raises(lib.HduError, match="rebuild"):
        lib._bind(emu_bind.emulator_library_path())
