"""What re-reading from the Infinity Cache is worth: stream S over 64 rows whose (x, y, z) are 64 DISTINCT triples (768 MB read from
HBM per launch) against 64 rows that all read the SAME triple (12 MB, resident in the 256 MB memory-side cache after the first
row) — outputs distinct either way.  Also the no-math variant (w = x + y + z).  Sustained µs per launch."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
import bench
fm.init(0); fm.set_jit(fm.JIT_SYNC)
n, B = 1_000_000, 64
rows = bench.synthetic_inputs(fm, B, n, 0)
shared = [rows[0] for _ in range(B)]
outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
partial = torch.zeros(B * 4, dtype=torch.float64, device="cuda:0")
ext = torch.cuda.ExternalStream(fm.stream_ptr(), device=torch.device("cuda", 0))

def sustained(p, r, moments):
    def run(k):
        for _ in range(k):
            p.run_into(r, outs, want_moments=False, device_moments=partial.data_ptr() if moments else None)
    run(100); fm.synchronize()
    chunks = 40
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    evs[0].record(ext)
    for c in range(chunks):
        run(50); evs[c + 1].record(ext)
    fm.synchronize()
    us = [evs[c].elapsed_time(evs[c + 1]) * 20.0 for c in range(chunks)]
    return round(sum(us[chunks // 2:]) / (chunks - chunks // 2), 1)

S = bench.build_stream_s(fm)
p = fm.Program(3); p.output(p.op("ADD", p.op("ADD", 0, 1), 2)); T = p.compile()
print(json.dumps({"stream_S_distinct_us": sustained(S, rows, True), "stream_S_shared_us": sustained(S, shared, True),
                  "triad_distinct_us": sustained(T, rows, False), "triad_shared_us": sustained(T, shared, False)}))
