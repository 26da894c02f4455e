"""Loads the TEST-ONLY fibre-emulated build of libzkhip (tests/_emu).  Used only by `-m "not gpu"` tests to
exercise kernel indexing / LDS / barrier logic on tiny inputs; the product library is never built this way."""
import os
import subprocess

from zokrates_amd import native

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "_emu")
EMU_LIB = os.path.join(EMU_DIR, "libzkhip_emu.so")
CSRC = os.path.join(HERE, "..", "zokrates_amd", "csrc")
_lib = None


def emu_library():
    global _lib
    if _lib is None and os.environ.get("ZKHIP_EMU_LIBRARY"):      # another build of the emulator (e.g. -DZK_CHECKED: index assertions)
        _lib = native.Library(os.environ["ZKHIP_EMU_LIBRARY"])
    if _lib is None:
        host = os.path.join(CSRC, "host")
        srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(host, f) for f in os.listdir(host)]
        srcs += [os.path.join(HERE, "..", "include", h) for h in ("zkhip.h", "zkhip_backend.hpp")]
        built = [EMU_LIB, os.path.join(EMU_DIR, "zkhip-cli-emu")]        # the library and the compiled host layer linked against it
        if not all(os.path.exists(b) for b in built) or any(os.path.getmtime(s) > min(os.path.getmtime(b) for b in built) for s in srcs):
            subprocess.check_call([os.path.join(EMU_DIR, "build_emu.sh")])
        _lib = native.Library(EMU_LIB)
    return _lib
