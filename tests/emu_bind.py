"""TEST INFRASTRUCTURE ONLY: binds the x86 emulator build of the kernel sources (tests/hipemu/libhdu_emu.so) under the
package's ctypes mirror, so that the CPU-only test tier can execute kernel logic.  Nothing under h-denseunet_amd/ knows this
library exists (the binder used to live in the product package); `lib.load()` itself refuses any library
whose hdu_backend() is not "hip-gfx950"."""
import importlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emulator_library_path():
    return os.path.join(ROOT, "tests", "hipemu", "libhdu_emu.so")


def use_emulator():
    lib = importlib.import_module("h-denseunet_amd.lib")
    path = emulator_library_path()
    if not os.path.exists(path):
        raise lib.HduError("emulator library not built: run ./build.sh emu")
    bound = lib._bind(path)
    backend = bound.hdu_backend().decode()
    assert backend == "emu-x86", backend
    lib._lib, lib._backend = bound, backend
    lib._apply_env_tuning(bound)
    return bound
