"""CPU side of the table-driven log: the committed table header is what the generator produces, and the device algorithm,
restated operation by operation in C (tools/check_log_table.cpp), agrees with `(float)log((double)x)` on a strided sample
of all positive fp32 arguments (the full sweep is the tool's default mode: 8 s on 8 cores)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "csrc")


def test_table_header_is_current(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import minimax_coefficients as mc
    from decimal import Decimal as D
    A = D(2) ** -9 * (1 + D(2) ** -5)
    q = mc.cheb_fit(mc.q_log1p, -A, A, 3)
    out = tmp_path / "fm_log_table.hpp"
    mc.write_table(str(out), [float(c) for c in q])
    assert out.read_text() == open(os.path.join(CSRC, "fm_log_table.hpp")).read()


def test_c_restatement_matches_libm_on_a_sample(tmp_path):
    exe = str(tmp_path / "check_log_table")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-DCHECK_STRIDE=257", "-o", exe, os.path.join(ROOT, "tools", "check_log_table.cpp")],
                   check=True, cwd=ROOT)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differences" in r.stdout
