"""Compiled fused op streams (fmhip_program_*): build once, run over a batch of vector tuples in ONE launch."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .random_variable import OP, DeviceVector


class Program:
    """SSA builder + handle of a compiled program.

    >>> p = Program(n_inputs=2)
    >>> t = p.op("ADD_S", p.input(0), s=4.0); u = p.op("MULT", t, p.input(1))
    >>> p.output(u); p.reduce(u); p.compile()
    >>> outs, moments = p.run([[x, y]])
    """

    def __init__(self, n_inputs: int):
        self.n_inputs = n_inputs
        self.ops = []
        self.outputs = []
        self.reduces = []
        self.handle = 0

    def input(self, k: int) -> int:
        assert 0 <= k < self.n_inputs
        return k

    def op(self, name: str, a: int, b: int = -1, c: int = -1, s: float = 0.0) -> int:
        self.ops.append((OP[name], a, b, c, float(s)))
        return self.n_inputs + len(self.ops) - 1

    def output(self, value: int) -> None:
        self.outputs.append(value)

    def reduce(self, value: int) -> None:
        self.reduces.append(value)

    def _description(self):
        arr = (N.ProgOp * max(1, len(self.ops)))()
        for i, (code, a, b, c, s) in enumerate(self.ops):
            arr[i] = N.ProgOp(code, a, b, c, s)
        outs = (C.c_int32 * max(1, len(self.outputs)))(*self.outputs)
        reds = (C.c_int32 * max(1, len(self.reduces)))(*self.reduces)
        return arr, outs, reds

    def source(self) -> str:
        """Generated HIP source of the specialised (JIT tier) kernel pair of this op stream; needs no device."""
        arr, outs, reds = self._description()
        need = C.c_int64(0)
        args = (arr, len(self.ops), self.n_inputs, outs, len(self.outputs), reds, len(self.reduces))
        N.check(N.lib().fmhip_program_source(*args, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        N.check(N.lib().fmhip_program_source(*args, buf, need.value + 1, C.byref(need)))
        return buf.value.decode()

    def tier(self):
        """(tier, vgprs): tier 0 = interpreter, 1 = specialised kernel ready."""
        t, v = C.c_int32(0), C.c_int32(0)
        N.check(N.lib().fmhip_program_tier(self.handle, C.byref(t), C.byref(v)))
        return t.value, v.value

    def compile(self) -> "Program":
        arr, outs, reds = self._description()
        h = C.c_int64(0)
        N.check(N.lib().fmhip_program_create(arr, len(self.ops), self.n_inputs, outs, len(self.outputs),
                                             reds, len(self.reduces), C.byref(h)))
        self.handle = h.value
        return self

    def __del__(self):
        h, self.handle = self.handle, 0
        if h and N is not None and N._lib is not None:
            try:
                N._lib.fmhip_program_release(h)
            except Exception:
                pass

    def _marshal(self, rows):
        batch = len(rows)
        ins = (C.c_int64 * (batch * self.n_inputs))()
        for b, row in enumerate(rows):
            assert len(row) == self.n_inputs
            for k, v in enumerate(row):
                ins[b * self.n_inputs + k] = v.handle
        return batch, ins

    def run(self, rows, shifts=None, want_moments=True, device_moments: int = 0):
        """rows: list of input tuples (DeviceVector). Returns (outputs[batch][n_out], moments float64[batch][n_red][4] | None)."""
        batch, ins = self._marshal(rows)
        n = rows[0][0].n
        n_out, n_red = len(self.outputs), len(self.reduces)
        outs = (C.c_int64 * max(1, batch * n_out))()
        sh = None
        if shifts is not None:
            sh = (C.c_double * n_red)(*shifts)
        mom = None
        if n_red and want_moments:
            mom = (N.Moments * (batch * n_red))()
        N.check(N.lib().fmhip_program_run(self.handle, batch, ins, outs, sh, mom, C.c_void_p(device_moments or None)))
        out_vecs = [[DeviceVector(outs[b * n_out + k], n) for k in range(n_out)] for b in range(batch)]
        m = None
        if mom is not None:
            m = np.frombuffer(mom, dtype=np.float64).reshape(batch, n_red, 4).copy()
        return out_vecs, m

    def run_into(self, rows, out_rows, shifts=None, want_moments=True, device_moments: int = 0):
        """Like run, but writes into existing vectors (steady-state loops: no allocation, no handle churn)."""
        batch, ins = self._marshal(rows)
        n_out, n_red = len(self.outputs), len(self.reduces)
        outs = (C.c_int64 * max(1, batch * n_out))()
        for b, row in enumerate(out_rows):
            for k, v in enumerate(row):
                outs[b * n_out + k] = v.handle
        sh = (C.c_double * n_red)(*shifts) if shifts is not None else None
        mom = (N.Moments * (batch * n_red))() if (n_red and want_moments) else None
        N.check(N.lib().fmhip_program_run_into(self.handle, batch, ins, outs, sh, mom, C.c_void_p(device_moments or None)))
        if mom is None:
            return None
        return np.frombuffer(mom, dtype=np.float64).reshape(batch, n_red, 4).copy()
