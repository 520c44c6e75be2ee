"""bench.py's contract, checked on the GPU box: ONE JSON line with the keys the driver reads, the roofline and cpu_baseline
objects, and the in-run parity of the timed kernel's expectations against the CPU twin (stream half only; short)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_half_of_the_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stream", "--gpus", "1", "--steps", "6", "--warmup", "2",
                          "--sustained-seconds", "0.3"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                               # ONE JSON line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "path-ops/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.3 < r["frac"] < 0.78                                        # below the streaming ceiling of this data, above a broken kernel
    assert r["tier"] == ("interpreter" if os.environ.get("FMHIP_JIT") == "off" else "specialised")
    # value = path-ops of the whole job / step time; consistent with ms_per_step
    assert abs(d["value"] - 12 * 1_000_000 * 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 1e8 and "sample" in c
    assert d["parity_triple0"]["ok"] and d["parity_triple0"]["min_max_identical"]
    assert d["value"] > 1000 * c["value"]
