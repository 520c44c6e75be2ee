"""Error behaviour of the C-ABI (include/fmhip.h): every misuse returns a negative status and a message, never a crash and
never a silent null (SURVEY.md §8b "Errors": the reference throws CudaException / returns null for unsupported methods)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OK, INVALID_HANDLE, SIZE_MISMATCH, INVALID_ARGUMENT, PROGRAM_LIMIT = 0, -1, -2, -5, -8


@pytest.fixture()
def lib(gpu):
    return gpu.lib()


def vec(gpu, n, fill=1.0):
    return gpu.DeviceVector.filled(n, fill)


def test_handles(gpu, lib):
    h = C.c_int64(0)
    assert lib.fmhip_vec_create_filled(16, 1.0, C.byref(h)) == OK
    assert lib.fmhip_vec_retain(h) == OK and lib.fmhip_vec_release(h) == OK and lib.fmhip_vec_release(h) == OK
    n = C.c_int64(0)
    for call in (lambda: lib.fmhip_vec_release(h), lambda: lib.fmhip_vec_retain(h), lambda: lib.fmhip_vec_size(h, C.byref(n)),
                 lambda: lib.fmhip_call_v1s0(10, h, C.byref(C.c_int64(0))), lambda: lib.fmhip_reduce_moments(h, 0.0, C.byref(gpu.Moments()))):
        assert call() == INVALID_HANDLE                         # use after the last release
    assert b"handle" in lib.fmhip_last_error()
    assert lib.fmhip_vec_release(C.c_int64(0)) == INVALID_HANDLE and lib.fmhip_vec_release(C.c_int64(-7)) == INVALID_HANDLE


def test_null_pointers_and_sizes(gpu, lib):
    assert lib.fmhip_vec_create_filled(16, 1.0, None) == INVALID_ARGUMENT
    assert lib.fmhip_vec_create_from_double(None, 4, C.byref(C.c_int64(0))) == INVALID_ARGUMENT
    assert lib.fmhip_vec_create_filled(-1, 1.0, C.byref(C.c_int64(0))) == INVALID_ARGUMENT
    assert lib.fmhip_vec_create_uninitialized((1 << 31) + 1, C.byref(C.c_int64(0))) == INVALID_ARGUMENT
    a, b = vec(gpu, 8), vec(gpu, 9)
    out = C.c_int64(0)
    assert lib.fmhip_call_v2s0(21, a.handle, b.handle, C.byref(out)) == SIZE_MISMATCH
    assert lib.fmhip_call_v3s0(31, a.handle, a.handle, b.handle, C.byref(out)) == SIZE_MISMATCH
    assert lib.fmhip_call_v2s0(21, a.handle, a.handle, None) == INVALID_ARGUMENT
    buf = (C.c_double * 4)()
    assert lib.fmhip_vec_read_double(a.handle, buf, 4) != OK     # count must equal the vector size
    assert lib.fmhip_vec_read_double(a.handle, None, 8) == INVALID_ARGUMENT
    assert lib.fmhip_reduce_moments(a.handle, 0.0, None) == INVALID_ARGUMENT
    assert lib.fmhip_reduce_moments_batch(None, 1, None, (gpu.Moments * 1)()) == INVALID_ARGUMENT
    hs = (C.c_int64 * 2)(a.handle, b.handle)
    assert lib.fmhip_reduce_moments_batch(hs, 2, None, (gpu.Moments * 2)()) == SIZE_MISMATCH
    assert lib.fmhip_reduce_moments_batch(hs, 0, None, (gpu.Moments * 2)()) == INVALID_ARGUMENT


@pytest.mark.parametrize("opcode", [0, -3, 32, 1000])
def test_unknown_opcodes(gpu, lib, opcode):
    a = vec(gpu, 8)
    out = C.c_int64(0)
    assert lib.fmhip_call_v1s0(opcode, a.handle, C.byref(out)) == INVALID_ARGUMENT
    assert lib.fmhip_call_v1s1(opcode, a.handle, 1.0, C.byref(out)) == INVALID_ARGUMENT
    assert lib.fmhip_call_v2s0(opcode, a.handle, a.handle, C.byref(out)) == INVALID_ARGUMENT


def test_call_shape_must_match_the_opcode(gpu, lib):
    a = vec(gpu, 8)
    out = C.c_int64(0)
    assert lib.fmhip_call_v1s0(21, a.handle, C.byref(out)) == INVALID_ARGUMENT             # ADD is binary
    assert lib.fmhip_call_v1s1(12, a.handle, 1.0, C.byref(out)) == INVALID_ARGUMENT        # EXP takes no scalar
    assert lib.fmhip_call_v2s0(3, a.handle, a.handle, C.byref(out)) == INVALID_ARGUMENT    # ADD_S is unary + scalar
    assert lib.fmhip_call_v2s1(21, a.handle, a.handle, 1.0, C.byref(out)) == INVALID_ARGUMENT
    assert lib.fmhip_call_v3s0(21, a.handle, a.handle, a.handle, C.byref(out)) == INVALID_ARGUMENT


def test_program_descriptions(gpu, lib):
    P = gpu.ProgOp
    ops = (P * 2)(P(21, 0, 1, -1, 0.0), P(12, 2, -1, -1, 0.0))          # t = x + y; u = exp(t)
    outs = (C.c_int32 * 1)(3)
    h = C.c_int64(0)
    assert lib.fmhip_program_create(ops, 2, 2, outs, 1, None, 0, C.byref(h)) == OK
    assert lib.fmhip_program_release(h) == OK and lib.fmhip_program_release(h) == INVALID_HANDLE
    bad_ref = (P * 1)(P(21, 0, 7, -1, 0.0))                              # operand 7 does not exist
    assert lib.fmhip_program_create(bad_ref, 1, 2, (C.c_int32 * 1)(2), 1, None, 0, C.byref(h)) != OK
    forward = (P * 2)(P(21, 0, 3, -1, 0.0), P(12, 0, -1, -1, 0.0))       # uses a value defined later
    assert lib.fmhip_program_create(forward, 2, 2, (C.c_int32 * 1)(3), 1, None, 0, C.byref(h)) != OK
    assert lib.fmhip_program_create(ops, 2, 2, None, 0, None, 0, C.byref(h)) == INVALID_ARGUMENT       # neither output nor reduction
    assert lib.fmhip_program_create(ops, 2, 2, (C.c_int32 * 1)(9), 1, None, 0, C.byref(h)) != OK      # output id out of range
    too_many = (C.c_int32 * 9)(*([2] * 9))
    assert lib.fmhip_program_create(ops, 2, 2, too_many, 9, None, 0, C.byref(h)) == PROGRAM_LIMIT
    assert lib.fmhip_program_create(ops, 2, 13, outs, 1, None, 0, C.byref(h)) == PROGRAM_LIMIT         # more than 12 inputs
    assert lib.fmhip_program_create(ops, 2, 2, outs, 1, (C.c_int32 * 3)(2, 2, 2), 3, C.byref(h)) == PROGRAM_LIMIT   # 3 reductions


def test_program_run_arguments(gpu, lib):
    p = gpu.Program(2)
    p.output(p.op("ADD", 0, 1))
    p.compile()
    a, b, c = vec(gpu, 8), vec(gpu, 8), vec(gpu, 9)
    ins = (C.c_int64 * 2)(a.handle, c.handle)
    outs = (C.c_int64 * 1)()
    assert lib.fmhip_program_run(p.handle, 1, ins, outs, None, None, None) == SIZE_MISMATCH
    assert lib.fmhip_program_run(p.handle, 0, ins, outs, None, None, None) != OK
    assert lib.fmhip_program_run(p.handle, 1, None, outs, None, None, None) == INVALID_ARGUMENT
    assert lib.fmhip_program_run(987654, 1, ins, outs, None, None, None) == INVALID_HANDLE
    good = (C.c_int64 * 2)(a.handle, b.handle)
    small = (C.c_int64 * 1)(c.handle)
    assert lib.fmhip_program_run_into(p.handle, 1, good, small, None, None, None) == SIZE_MISMATCH      # output vector of another size
    assert lib.fmhip_program_run(p.handle, 1, good, outs, None, None, None) == OK
    assert lib.fmhip_vec_release(outs[0]) == OK


def test_brownian_and_modes(gpu, lib):
    dt = (C.c_double * 2)(0.1, 0.2)
    out = (C.c_int64 * 4)()
    assert lib.fmhip_bm_generate(1, 2, 2, 0, 0, dt, out) != OK or all(h != 0 for h in out)             # empty paths: error or valid handles
    assert lib.fmhip_bm_generate(1, 0, 2, 16, 0, dt, out) != OK
    assert lib.fmhip_bm_generate(1, 2, 0, 16, 0, dt, out) != OK
    assert lib.fmhip_bm_generate(1, 2, 2, 16, 0, None, out) == INVALID_ARGUMENT
    assert lib.fmhip_bm_generate(1, 2, 2, 16, 0, (C.c_double * 2)(0.1, -0.2), out) != OK               # negative time step
    assert lib.fmhip_set_math_mode(7, None) == INVALID_ARGUMENT and lib.fmhip_set_jit(9, None) == INVALID_ARGUMENT
    assert lib.fmhip_program_source(None, 1, 1, None, 0, None, 0, None, 0, None) == INVALID_ARGUMENT
