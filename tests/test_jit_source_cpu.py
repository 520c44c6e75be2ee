"""CPU-side checks of the specialised-kernel tier: the source generator (needs no device) and, when hiprtc is usable in
this container, that the generated source compiles for gfx950 with the library's own embedded headers."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(fm):
    p = fm.Program(3)
    t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", 0, s=4.0), s=2.0), 1), 2)
    u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
    w = p.op("CHOOSE", t, u, 0)
    p.output(w)
    p.reduce(w)
    return p


def test_generated_source_shape():
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    src = build(fm).source()
    assert src.count('extern "C" __global__') == 2                      # row block inline (batch 1) and row table flavours
    assert "fm_jit_inline" in src and "fm_jit_table" in src
    assert '#include "fm_kernel_parts.hpp"' in src
    assert "red_accumulate<E>" in src and "block_combine<1>" in src     # the interpreter's own reduction code
    assert src.count("ueval<") == 2 * 9                                 # LDA + 8 micro-ops (DIV_S by 2 became MULT_S; CHOOSE reuses t) …
    assert src.count("sqrt_all<E>(a)") == 2                             # … and sqrt, whose E elements share one special-case branch
    # deterministic: the text is the cache key of the compiled kernel
    assert src == build(fm).source()


def test_source_of_a_bad_description_is_an_error():
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    import pytest
    p = fm.Program(1)
    p.op("ADD", 0, 5)                   # operand 5 does not exist
    p.output(1)
    with pytest.raises(fm.FmhipError):
        p.source()
