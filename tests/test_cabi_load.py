"""CPU: the product library (hipcc-built for gfx950) loads and exports every symbol that
include/wsnark.h declares; no compute is attempted without a GPU, and there is no CPU fallback."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "wsnark.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wsnark_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_bound():
    from wasmsnark_amd import _lib
    assert sorted(_lib.SYMBOLS) == _declared()


def test_library_loads_and_exports_all():
    so = os.path.join(ROOT, "wasmsnark_amd", "libwsnark.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    from wasmsnark_amd import _lib
    lib = _lib.Lib()
    assert lib.path == so
    for s in _declared():
        assert hasattr(lib.c, s)


def test_no_silent_fallback(tmp_path):
    """A missing library is an ImportError, not a CPU path; and the product's binding takes no path and reads no environment."""
    import inspect
    from wasmsnark_amd import _lib

    class Missing(_lib.Lib):
        SO = str(tmp_path / "missing.so")
    with pytest.raises(ImportError):
        Missing()
    assert list(inspect.signature(_lib.Lib.__init__).parameters) == ["self"] and not inspect.signature(_lib.load).parameters
    assert "environ" not in inspect.getsource(_lib.Lib.__init__)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "wasmsnark_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c", ".js")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in src and "liboracle" not in src and "bn128_oracle" not in src, f


def test_uninitialised_calls_fail_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    so = os.path.join(ROOT, "wasmsnark_amd", "libwsnark.so")
    from wasmsnark_amd import _lib, WsnarkError
    lib = _lib.Lib()
    import ctypes as C
    buf = (C.c_uint8 * 64)()
    assert lib.c.wsnark_fr_ntt(buf, 2, 0, 0) == 5          # WSNARK_ERR_NOINIT
    with pytest.raises(WsnarkError):
        lib.init(0)                                         # no HIP device here -> error, not fallback
