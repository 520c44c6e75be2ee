"""How to time the bench kernel: HIP events around a whole region of back-to-back launches (duration / launches) versus one
event pair per launch (fmhip_profile_*; also what a tracing profiler does).  Alternates the two methods several times in
one process so that chip temperature / clock state is the same for both."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
fm.set_jit(fm.JIT_SYNC)
n, B = 1_000_000, 64
bm = fm.BrownianMotionHip(fm.TimeDiscretization(0.0, B, 1.0), 3, n, 31415)
rows = []
for b in range(B):
    g = [bm.getBrownianIncrement(b, f) for f in range(3)]
    rows.append([g[0].mult(0.25).add(0.5).cap(1.0).floor(0.0).realizations, g[1].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations,
                 g[2].mult(0.25).add(1.0).cap(1.5).floor(0.5).realizations])
del bm
outs = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
p = fm.Program(3)
x, y, z = 0, 1, 2
t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
w = p.op("CHOOSE", t, v, x)
p.output(w); p.reduce(w); p.compile()
partial = torch.zeros(B * 4, dtype=torch.float64, device="cuda:0")
stream = torch.cuda.ExternalStream(fm.stream_ptr(), device=torch.device("cuda", 0))


def run(k):
    for _ in range(k):
        p.run_into(rows, outs, want_moments=False, device_moments=partial.data_ptr())


run(20); fm.synchronize()
for rep in range(4):
    K = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); run(K); e1.record(stream); fm.synchronize(); torch.cuda.synchronize()
    region = e0.elapsed_time(e1) / K * 1e3
    fm.profile_enable(True); run(K); ms, k = fm.profile_read(); fm.profile_enable(False)
    per_launch = ms / k * 1e3
    print(f"round {rep}: region events {region:7.1f} us/launch   per-launch events {per_launch:7.1f} us/launch", flush=True)
