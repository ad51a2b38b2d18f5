"""Builds / loads tests/emul/libwsnark_emul.so: the product's kernel SOURCES compiled by g++
against a CPU thread emulator (tests/emul/hip_emul.h).  CPU tests use it to check kernel index
math without a GPU; it is not the product and GPU tests never touch it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wasmsnark_amd", "csrc")
# WSNARK_EMUL_SAN=asan | tsan: the sanitizer build of the same library (make -C wasmsnark_amd/csrc emul-san SAN=...), for the runs
# tests/test_sanitizers.py starts with the sanitizer's runtime preloaded
_SAN = os.environ.get("WSNARK_EMUL_SAN", "")
SO = SO_PATH = os.path.join(ROOT, "tests", "emul", "libwsnark_emul%s.so" % ("_" + _SAN if _SAN else ""))

_bn = None


def emul_bn128():
    global _bn
    if _bn is None:
        if _SAN:
            subprocess.check_call(["make", "-C", CSRC, "-s", "-j8", "emul-san", "SAN=" + {"asan": "address,undefined", "tsan": "thread"}[_SAN]])
        else:
            subprocess.check_call(["make", "-C", CSRC, "-s", "-j8", "emul"])
        from wasmsnark_amd import _lib, bn128

        class EmulLib(_lib.Lib):          # tests only: the same ctypes binding over the emulator build of the same ABI
            SO = SO_PATH

        _bn = bn128.Bn128(lib=EmulLib())
    return _bn
