"""Engine life cycle with the background compiler thread: shutdown → init again, and process exit while a compilation is
still queued (the reference's tests call RandomVariableCuda.purge() in @After and rely on JVM exit for the rest)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = (
    "import importlib, sys, json, numpy as np\n"
    f"sys.path.insert(0, {ROOT!r})\n"
    "fm = importlib.import_module('finmath-lib-cuda-extensions_amd')\n"
    "def run():\n"
    "    x = fm.DeviceVector.from_host(np.linspace(0.1, 2.0, 6007, dtype=np.float32))\n"
    "    p = fm.Program(1); w = p.op('SQRT', p.op('EXP', p.op('ADD_S', 0, s=0.5))); p.output(w); p.reduce(w); p.compile()\n"
    "    outs, m = p.run([[x]])\n"
    "    return int(outs[0][0].to_float32().view(np.uint32).sum()), float(m[0, 0, 0]), p.tier()[0]\n")


def run_script(body, tmp_path, **env):
    e = dict(os.environ, FMHIP_JIT_CACHE_DIR=str(tmp_path), **env)
    out = subprocess.run([sys.executable, "-c", PRELUDE + body], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout.strip().splitlines()


def test_shutdown_and_reinitialise(tmp_path):
    lines = run_script(
        "fm.init(0); fm.set_jit(fm.JIT_SYNC); a = run(); fm.shutdown()\n"
        "fm.init(0); fm.set_jit(fm.JIT_SYNC); b = run(); s = fm.jit_stats(); fm.shutdown()\n"
        "fm.init(0); fm.set_jit(fm.JIT_OFF); c = run(); fm.shutdown()\n"
        "print(json.dumps([a, b, c, s]))\n", tmp_path)
    a, b, c, stats = json.loads(lines[-1])
    assert a[:2] == b[:2] == c[:2]                      # same bits from every incarnation and from both tiers
    assert a[2] == 1 and b[2] == 1 and c[2] == 0
    assert stats["compiled"] == 1 and stats["disk_cache_hits"] == 1 and stats["failed"] == 0


def test_exit_with_pending_compilations(tmp_path):
    """Queue several asynchronous compilations and leave without shutdown: the process must end cleanly and promptly."""
    lines = run_script(
        "fm.init(0); fm.set_jit(fm.JIT_AUTO)\n"
        "ps = []\n"
        "for k in range(12):\n"
        "    p = fm.Program(1); w = 0\n"
        "    for j in range(k + 2): w = p.op('EXP' if j % 2 else 'LOG', w)\n"
        "    p.output(w); ps.append(p.compile())\n"
        "print('queued', fm.jit_stats()['pending'] > 0)\n", tmp_path)
    assert lines[-1].startswith("queued")


def test_native_exit_without_shutdown(tmp_path):
    """A C host that queues compilations and returns from main without fmhip_shutdown(): the exit handler registered by
    the library joins the compiler thread before the compiler library's own exit-time destructors run."""
    src = tmp_path / "exit_test.c"
    src.write_text(
        '#include <stdio.h>\n#include "fmhip.h"\n'
        "int main(void) {\n"
        "  if (fmhip_init(0)) return 2;\n"
        "  for (int k = 0; k < 10; ++k) {\n"
        "    fmhip_prog_op ops[12]; int n = k + 2;\n"
        "    for (int j = 0; j < n; ++j) { ops[j].opcode = (j % 2) ? FMHIP_OP_EXP : FMHIP_OP_LOG; ops[j].a = j; ops[j].b = -1; ops[j].c = -1; ops[j].scalar = 0; }\n"
        "    int32_t out = n; fmhip_program p;\n"
        "    if (fmhip_program_create(ops, n, 1, &out, 1, NULL, 0, &p)) { printf(\"%s\\n\", fmhip_last_error()); return 3; }\n"
        "  }\n"
        "  int64_t pending = 0; fmhip_jit_stats(NULL, NULL, &pending, NULL, NULL);\n"
        "  printf(\"pending %lld\\n\", (long long)pending);\n"
        "  return 0;\n}\n")
    exe = tmp_path / "exit_test"
    libdir = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-o", str(exe), str(src), f"-I{os.path.join(ROOT, 'include')}", f"-L{libdir}", "-lfmhip", f"-Wl,-rpath,{libdir}"])
    for _ in range(3):
        out = subprocess.run([str(exe)], env=dict(os.environ, FMHIP_JIT_CACHE_DIR="off"), capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.startswith("pending")
