"""CPU-only: the C-ABI library builds for gfx950, loads, and exports exactly what include/fmhip.h declares.
No compute call is made (there is no GPU here); calls that need a device must fail loudly, not fall back."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fmhip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fmhip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(fm):
    assert declared_functions() == sorted(fm._native.SYMBOLS)


def test_library_exports_every_declared_symbol(fm):
    lib = fm.lib()
    for name in declared_functions():
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", fm._native.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(declared_functions()) <= exported
    # nothing but the C-ABI is exported with C linkage under the fmhip_ prefix
    assert {s for s in exported if s.startswith("fmhip_")} == set(declared_functions())


def test_library_contains_gfx950_code_object(fm):
    blob = open(fm._native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert fm.lib().fmhip_abi_version() == 1


def test_no_cpu_fallback_without_device(fm):
    """Without a GPU every compute entry point reports an error — there is no silent CPU path."""
    import ctypes as C
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    lib = fm.lib()
    if lib.fmhip_is_initialized():
        pytest.skip("runtime already initialised")
    assert lib.fmhip_init(0) != 0
    out = C.c_int64(0)
    assert lib.fmhip_vec_create_filled(16, 1.0, C.byref(out)) == fm._native.ERR_NOT_INITIALIZED
    assert b"fmhip_init" in lib.fmhip_last_error()
    with pytest.raises(fm.FmhipError):
        fm.RandomVariableHipFactory().createRandomVariable(0.0, [1.0, 2.0])


def test_product_never_touches_the_oracle():
    """The product tree must not import, link or mention the oracle (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd")
    for base, _, files in os.walk(pkg):
        if os.sep + "build" in base or os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    assert "import oracle" not in code and "from oracle" not in code and "fm_oracle" not in code, (f, line)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "lib", "libfmhip.so")], text=True)
    assert "oracle" not in out
