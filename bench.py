#!/usr/bin/env python3
"""bench.py — headline benchmark of the RandomVariable hot path on MI355X (BASELINE.json configs[1]).

A step = ONE pass of the canonical fused RandomVariable stream S (SURVEY.md §8d config 2:
    t = x.add(4).div(2).mult(y).sub(z);  u = t.exp().log().abs().sqrt();
    v = u.cap(1.5).floor(0.25).addProduct(y,z);  w = t.choose(v,x);  {avg,var,min,max}(w)
 — 12 path-ops, 3 input vectors, 1 escaping output, 1 fused reduction) over a batch of B independent
(x,y,z) triples of N = 1 000 000 paths, executed as ONE horizontally batched launch through the C-ABI
(fmhip_program_run_into).  B·16 MB ≫ 256 MB Infinity Cache, so the traffic is HBM traffic.
Inputs: x, y, z = java.util.Random(31415 / 27182 / 16180).nextDouble() (+0.5 for y and z) as config 2 specifies, narrowed to
fp32 and resident in HBM before the timed region starts; the expectations of triple 0 are checked against the CPU twin on the
same inputs in the run itself ("parity_triple0").

metric  path-ops/s = 12 · N · B · n_gpus / step time        (whole job, all ranks)
roofline achieved = algorithmic bytes per launch (4 B · (3 in + 1 out) · N · B) / average device duration of the
         fused-program kernel, measured live with HIP events on the runtime's own stream (fmhip_profile_*).
cpu_baseline: the oracle (C restatement of the reference's CPU class, one single-threaded loop and one fresh
         array per method call — the reference's cost model) timed on this host, rank 0, N=1 only.

Multi-GPU (torchrun, one rank per GPU): independent Monte-Carlo path blocks per rank (weak scaling, no data-path
collective); the only exchange is ONE small RCCL all-gather of the per-rank expectation partials per 8 steps
(--exchange-every; DESIGN.md §7 says why not per step).

The default run carries BOTH halves of BASELINE.json's metric in ONE JSON line: the stream line above (top level,
contract keys unchanged) and, under "lmm", the LIBOR-Market-Model ATM calibration at 1 M paths per GPU (configs[3]; with
N ranks configs[4]: N x 1 M paths, one RCCL all-gather of the 144 expectation partials per objective evaluation) — wall
seconds, LM iterations, evaluations, mean deviation against the reference's 2e-4 acceptance, its own roofline (algorithmic
bytes of ALL program launches of the real calibration / their summed device time) and its own cpu_baseline.  The
calibration runs in the native driver bin/lmm_hip (C++ over the C-ABI), started as a child process BEFORE this process
touches the GPU.  `--workload stream` / `--workload lmm` run one half only.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_PATHS = 1_000_000
N_OPS = 12
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured float4-copy ceiling ≈ 6290 GB/s


def build_stream_s(fm):
    p = fm.Program(3)
    x, y, z = 0, 1, 2
    t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", x, s=4.0), s=2.0), y), z)
    u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
    v = p.op("ADDPRODUCT", p.op("FLOOR_S", p.op("CAP_S", u, s=1.5), s=0.25), y, z)
    w = p.op("CHOOSE", t, v, x)
    p.output(w)
    p.reduce(w)
    return p.compile()


_LCG_A, _LCG_C, _LCG_MASK = 0x5DEECE66D, 0xB, (1 << 48) - 1
_LCG_TABLE = {}


def java_random_doubles(seed, count, skip=0):
    """`new java.util.Random(seed)`: the doubles number skip … skip+count-1 of its nextDouble() stream (published specification:
    48-bit LCG s ← s·0x5DEECE66D + 0xB, nextDouble = ((next(26) << 27) + next(27))·2⁻⁵³).  Vectorised: the state k
    steps ahead is A_k·s + C_k (mod 2⁴⁸); uint64 arithmetic wraps mod 2⁶⁴, of which 2⁴⁸ is a divisor."""
    import numpy as np
    def jump(k):                                   # (A_k, C_k) by square-and-multiply on the affine map
        a, c, A, C = _LCG_A, _LCG_C, 1, 0
        while k:
            if k & 1:
                A, C = (A * a) & _LCG_MASK, (C * a + c) & _LCG_MASK
            a, c = (a * a) & _LCG_MASK, (c * a + c) & _LCG_MASK
            k >>= 1
        return A, C
    m = 2 * count                                  # two draws per double
    if _LCG_TABLE.get("m") != m:
        A = np.empty(m, dtype=np.uint64); C = np.empty(m, dtype=np.uint64)
        A[0], C[0] = _LCG_A, _LCG_C                # entry i: i+1 steps ahead
        filled = 1
        while filled < m:                          # doubling: steps filled+1 … 2·filled = (steps 1 … filled) after `filled` steps
            k = min(filled, m - filled)
            Ak, Ck = jump(filled)
            A[filled:filled + k] = A[:k] * np.uint64(Ak)
            C[filled:filled + k] = C[:k] * np.uint64(Ak) + np.uint64(Ck)
            filled += k
        _LCG_TABLE.update(m=m, A=A, C=C)
    A, C = _LCG_TABLE["A"], _LCG_TABLE["C"]
    s0 = (seed ^ _LCG_A) & _LCG_MASK
    As, Cs = jump(2 * skip)
    start = np.uint64((As * s0 + Cs) & _LCG_MASK)
    states = (A * start + C) & np.uint64(_LCG_MASK)
    hi = (states[0::2] >> np.uint64(22)).astype(np.float64)      # next(26)
    lo = (states[1::2] >> np.uint64(21)).astype(np.float64)      # next(27)
    return (hi * float(1 << 27) + lo) * (1.0 / float(1 << 53))


def synthetic_inputs(fm, batch, n, rank):
    """x, y, z from java.util.Random seeds 31415 / 27182 / 16180, uniform [0,1) (+0.5 for y and z),
    narrowed to fp32 as the factory does and uploaded once; triple b of rank r is the block [(r·batch + b)·n, +n) of each stream."""
    import numpy as np
    rows = []
    for b in range(batch):
        skip = (rank * batch + b) * n
        x = java_random_doubles(31415, n, skip).astype(np.float32)
        y = (java_random_doubles(27182, n, skip) + 0.5).astype(np.float32)
        z = (java_random_doubles(16180, n, skip) + 0.5).astype(np.float32)
        rows.append([fm.DeviceVector.from_host(x), fm.DeviceVector.from_host(y), fm.DeviceVector.from_host(z)])
    return rows


def cpu_baseline(target_seconds=8.0, all_cores_seconds=6.0):
    """Stream S on the CPU oracle, one 1M-path triple per pass (the reference's cost model: single-threaded loops, one fresh
    array per method), repeated for ~target_seconds per class:
      * headline: the DOUBLE class the north star names (RandomVariableFromArrayFactory → RandomVariableFromDoubleArray,
        finmath-lib 5.1.3, not vendored: timed is the stand-in oracle/rv_double.c);
      * `float_twin`: the reference's in-tree fp32 twin RandomVariableFromFloatArray (oracle/rv_float.c);
      * `all_cores` (information): the fp32 loop in one fresh process per host core over independent triples."""
    import oracle as o
    n = N_PATHS
    xd, yd, zd = o.java_random_doubles(31415, n), o.java_random_doubles(27182, n) + 0.5, o.java_random_doubles(16180, n) + 0.5
    x, y, z = o.f_from_double(xd), o.f_from_double(yd), o.f_from_double(zd)

    def pass_float():
        t = o.f_v2s0("SUB", o.f_v2s0("MULT", o.f_v1s1("DIV_S", o.f_v1s1("ADD_S", x, 4.0), 2.0), y), z)
        u = o.f_v1s0("SQRT", o.f_v1s0("ABS", o.f_v1s0("LOG", o.f_v1s0("EXP", t))))
        v = o.f_v3s0("ADDPRODUCT", o.f_v1s1("FLOOR_S", o.f_v1s1("CAP_S", u, 1.5), 0.25), y, z)
        w = o.f_v3s0("CHOOSE", t, v, x)
        return o.f_average(w), o.f_variance(w), o.f_min(w), o.f_max(w)

    def pass_double():
        d = o.d_apply
        t = d("SUB", d("MULT", d("DIV_S", d("ADD_S", xd, 4.0), 2.0), yd), zd)
        u = d("SQRT", d("ABS", d("LOG", d("EXP", t))))
        v = d("ADDPRODUCT", d("FLOOR_S", d("CAP_S", u, 1.5), 0.25), yd, zd)
        w = d("CHOOSE", t, v, xd)
        return o.d_average(w), o.d_variance(w), o.d_min(w), o.d_max(w)

    def timed(one_pass, seconds):
        one_pass()
        passes, t0 = 0, time.perf_counter()
        while True:
            one_pass()
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or passes >= 2000:
                return passes, dt

    pd, td = timed(pass_double, target_seconds)
    result = {"value": N_OPS * n * pd / td, "unit": "path-ops/s", "cores": 1, "kind": "port",
              "sample": f"{pd} passes of stream S over one 1M-path (x,y,z) triple in {td:.1f} s on the double-precision stand-in for "
                        f"RandomVariableFromDoubleArray (oracle/rv_double.c: one loop + one fresh array per method)"}
    if all_cores_seconds < 0:              # internal: one worker of the all-cores leg times the fp32 loop only
        pf, tf = timed(pass_float, target_seconds)
        return {"value": N_OPS * n * pf / tf}
    pf, tf = timed(pass_float, target_seconds)
    avg, var, mn, mx = pass_float()                # the checker's moments of w for triple 0 = the GPU's first triple (same LCG blocks)
    result["twin_moments_triple0"] = {"average": avg, "variance": var, "min": mn, "max": mx}
    result["float_twin"] = {"value": N_OPS * n * pf / tf, "unit": "path-ops/s", "cores": 1, "kind": "port",
                            "sample": f"{pf} passes in {tf:.1f} s on the C restatement of the reference's RandomVariableFromFloatArray"}
    if all_cores_seconds > 0:
        import subprocess
        cores = max(1, min(len(os.sched_getaffinity(0)), 64))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(all_cores_seconds)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
        total = 0.0
        for pr in procs:
            out, _ = pr.communicate(timeout=all_cores_seconds * 6 + 60)
            if pr.returncode == 0 and out.strip():
                total += json.loads(out.strip().splitlines()[-1])["value"]
        result["all_cores"] = {"value": total, "unit": "path-ops/s", "cores": cores, "kind": "port",
                               "sample": f"{cores} processes x ~{all_cores_seconds:.0f} s of the fp32 twin's loop over independent triples"}
    return result


LMM_HIP = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "bin", "lmm_hip")
LMM_CPU = os.path.join(ROOT, "oracle", "host", "lmm_cpu")


def lmm_traffic_ratio():
    """HBM bytes of the LMM op stream (FETCH_SIZE x 2 + WRITE_SIZE over every launch of 24 objective evaluations, separate rocprofv3 --pmc passes)
    over the engine's algorithmic bytes for the same launches: measured OFFLINE (counters need profiler passes of their own), committed under
    profiles/ (benchmarks/lmm_hbm_traffic.py); null if the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "round04_lmm_hbm_traffic.json")) as fh:
            t = json.load(fh)
        return {"ratio": t["traffic_over_algorithmic"], "source": "profiles/round04_lmm_hbm_traffic.json"}
    except Exception:
        return None


def lmm_leg(args, world, rank, nonce, cpu_base=True, store=None):
    """BASELINE.json configs[3] / [4]: LMM ATM swaption calibration (LIBORMarketModelCalibrationATMTest.java:186-340 inputs,
    acceptance :466) at `--paths` paths per GPU in the native driver host/lmm.hpp over the C-ABI.  Runs in child processes
    (one lmm_hip per rank; LOCAL_RANK picks the device) and must be called BEFORE this process touches the GPU.
      run 1  the calibration, unprofiled: wall seconds = the metric's second half (cold code-object cache on a fresh box);
      run 2  the same calibration with every program launch bracketed by HIP events on the runtime stream (--profile):
             roofline.achieved = algorithmic bytes of all launches (fmhip_traffic_stats) / their summed device time;
      N = 1 also: profiled replay of objective evaluations one at a time and 8 in lock-step (what a launch shape is worth),
             and the cpu_baseline: ONE objective evaluation of the same model at the same path count on the CPU twin
             (oracle/host/lmm_cpu, ~15 s on one core), scaled by the number of evaluations the calibration needed.
    Returns the dict that goes under "lmm" (rank 0) or None."""
    import subprocess
    if not os.path.exists(LMM_HIP):          # build products of csrc/Makefile and oracle/Makefile
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "csrc")], stdout=subprocess.DEVNULL)
    if not os.path.exists(LMM_CPU):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    paths = args.paths                      # per GPU; configs[4] = 1M paths on each of 8 GPUs (path sharding, weak scaling)
    base = [LMM_HIP, "--paths", str(paths), "--mode", "calibrate", "--max-iterations", str(args.lmm_iterations)]

    def dist_args(tag):
        if nonce is None:
            return []
        # one native process per GPU; they find each other through an RCCL unique-id file whose name and content carry the
        # launch's nonce (handed to every rank by the launcher's store), so an id left by another run is never picked up
        return ["--world", str(world), "--rank", str(rank), "--nccl-id-file", f"/tmp/fmhip_nccl_id_{nonce}_{tag}", "--nccl-nonce", str(nonce)]

    def run(cmd, env=None):
        t0 = time.perf_counter()
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)      # a child that hangs (a peer died inside a collective) is killed
        if out.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)} (rank {rank}) failed: {out.stderr[-2000:]}")
        return json.loads(out.stdout.strip().splitlines()[-1]), time.perf_counter() - t0

    # what THIS box's memory system and clocks give (bin/box_speed: non-temporal copy and read-only sweep of 2 GiB, clock under vector
    # load): the same binary reads 0.65 of the HBM peak on one MI355X box and 0.71 on another — context for every fraction below
    box = None
    if rank == 0:
        try:
            box = json.loads(subprocess.run([os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "bin", "box_speed")], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
        except Exception as e:
            box = {"error": str(e)[-200:]}
    r, wall = run(base + dist_args("a"))
    rp, _ = run(base + ["--profile"] + dist_args("b"))
    per_rank = None
    if store is not None and world > 1:
        # every rank ran the same optimiser on the same (all-gathered, rank-ordered) expectations: the calibrated parameter vectors
        # must be equal BIT FOR BIT on all ranks — checked here, through the launcher's store, before anything is reported
        store.set(f"fmhip_lmm_{nonce}_{rank}", json.dumps({"seconds": r["seconds"], "parameters": r["parameters"], "evaluations": r["evaluations"],
                                                          "kernel_s": rp["kernel_ms_total"] / 1e3, "achieved_GBps": rp["achieved_GBps"]}))
        if rank == 0:
            per_rank = [json.loads(store.get(f"fmhip_lmm_{nonce}_{k}").decode()) for k in range(world)]
            for k in range(1, world):
                if per_rank[k]["parameters"] != per_rank[0]["parameters"] or per_rank[k]["evaluations"] != per_rank[0]["evaluations"]:
                    raise RuntimeError(f"rank {k} calibrated other parameters than rank 0: the ranks did not see the same expectations")
    if rank != 0:
        return None
    kernel_s = rp["kernel_ms_total"] / 1e3
    # Where the host's own time goes (the engine's FMHIP_HOST_PROFILE table of a third run, N = 1 only): the slots that do not contain
    # each other, largest first.  host_idle_s = wall time of the profiled run the device spent without a kernel.
    host_top = None
    if world == 1:
        try:
            hp = subprocess.run(base, capture_output=True, text=True, env=dict(os.environ, FMHIP_HOST_PROFILE="1"), timeout=240)
            slots = {}
            for ln in hp.stderr.splitlines():
                parts = ln.split()
                if "calls" in parts and parts[-1] == "us/call":
                    slots[" ".join(parts[:parts.index("calls") - 1])] = float(parts[parts.index("calls") + 1])
            leaf = {k: v for k, v in slots.items() if not k.endswith("(total)") or k.startswith("graph_clone")}
            host_top = [{"what": k, "seconds": v} for k, v in sorted(leaf.items(), key=lambda kv: -kv[1])[:3]]
        except Exception as e:
            host_top = [{"error": str(e)[-200:]}]
    lmm = {"metric": "LMM calib wall-time, 1M paths" if world == 1 else f"LMM calib wall-time, {world}x1M paths (path-sharded)",
           "value": r["seconds"], "unit": "s", "higher_is_better": False, "n_gpus": world, "paths_per_gpu": paths, "paths": paths * world,
           "lm_iterations": r["iterations"], "objective_evaluations": r["evaluations"],
           "seconds_per_objective_evaluation": r["seconds"] / r["evaluations"],
           "mean_deviation": r["mean_deviation"], "rms_deviation": r["rms_deviation"], "initial_rms": r["initial_rms"],
           "acceptance": "abs(mean_deviation) < 2e-4 (LIBORMarketModelCalibrationATMTest.java:466)",
           "accepted": abs(r["mean_deviation"]) < 2e-4,
           "workload": "LIBORMarketModelCalibrationATMTest inputs: 80 forward rates, 1 factor, spot measure, normal state space, 144 ATM swaptions, "
                       "50 volatility parameters, Levenberg-Marquardt with finite differences; native driver over the C-ABI",
           "kernel_launches": r["kernel_launches"], "path_ops_per_s": r["path_ops"] / r["seconds"], "box_speed": box,
           "process_wall_s": wall, "seconds_second_run_warm_code_object_cache_profiled": rp["seconds"],
           "specialised_kernels": r.get("specialised_kernels"), "specialisations_from_disk_cache": r.get("specialisations_from_disk_cache"),
           "per_rank_seconds": None if per_rank is None else [x["seconds"] for x in per_rank],
           "per_rank_roofline_frac": None if per_rank is None else [x["achieved_GBps"] / HBM_PEAK_GBS for x in per_rank],
           "ranks_calibrated_identical_parameters": None if per_rank is None else True,
           "rccl": {"calls": r.get("rccl_collectives", 0), "summed_latency_s": r.get("rccl_collective_seconds", 0.0),
                    "what": "one all-gather of 144 x {sum, sumsq, min, max} fp64 partials per objective evaluation; latency = enqueue to result on the host"},
           "roofline": {"bound": "hbm", "achieved": rp["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rp["achieved_GBps"] / HBM_PEAK_GBS,
                        "traffic": None, "traffic_over_algorithmic_offline": lmm_traffic_ratio(),
                        "kernel": "all fused-program launches of the calibration itself (rank 0)",
                        "launches": rp["profiled_launches"], "specialised_launches": rp["specialised_launches"],
                        "algorithmic_bytes": rp["algorithmic_bytes"], "summed_kernel_s": kernel_s,
                        "device_busy_fraction_of_wall": kernel_s / rp["seconds"],
                        "host_idle_s": rp["seconds"] - kernel_s, "host_time_top3": host_top,
                        "timing": "one HIP event pair per launch on the runtime stream, summed"}}
    eng = rp.get("engine") or {}
    lmm["roofline"]["merged_launches"] = eng.get("merged_launches")
    lmm["roofline"]["merged_chains"] = eng.get("merged_chains")
    lmm["roofline"]["common_rows"] = eng.get("common_rows")        # rows that were not computed: an earlier row of the launch read the same vectors with the same scalars
    if world == 1:
        # Round 5: the swaptions of one exercise date run as ONE launch that loads every forward rate once (runtime.cpp: merge_families) — a third
        # of the bytes of one launch per tenor, and kernels that are bound by their arithmetic (the correctly rounded division per chain and
        # period), no longer by HBM; and rows of a launch that read the same vectors with the same scalars (the bumped parameter sets of a
        # Jacobian batch before their bump matters) are computed once (common rows): fewer path-ops, not faster ones.  `frac` above counts, as always, the bytes the launches that RAN have to move; the same calibration with
        # one launch per shape (FMHIP_MERGE_CHAINS=0: rounds 1-4) is measured beside it, and `frac_at_the_bytes_of_one_launch_per_shape`
        # sets ITS bytes against this run's kernel time: what the merged launches are worth in the old currency.
        try:
            u, _ = run(base, env=dict(os.environ, FMHIP_MERGE_CHAINS="0", FMHIP_COMMON_ROWS="0"))
            up, _ = run(base + ["--profile"], env=dict(os.environ, FMHIP_MERGE_CHAINS="0", FMHIP_COMMON_ROWS="0"))
            lmm["roofline"]["one_launch_per_shape"] = {
                "seconds": u["seconds"], "kernel_launches": u["kernel_launches"], "algorithmic_bytes": up["algorithmic_bytes"], "summed_kernel_s": up["kernel_ms_total"] / 1e3,
                "achieved": up["achieved_GBps"], "frac": up["achieved_GBps"] / HBM_PEAK_GBS, "mean_deviation": u["mean_deviation"],
                "identical_to_the_merged_run": u["mean_deviation"] == r["mean_deviation"] and u["rms_deviation"] == r["rms_deviation"] and u["parameters"] == r["parameters"],
                "what": "FMHIP_MERGE_CHAINS=0 FMHIP_COMMON_ROWS=0: every component shape its own launch, every row of a launch computed, as in rounds 1-4 (HBM-bound throughout)"}
            lmm["roofline"]["frac_at_the_bytes_of_one_launch_per_shape"] = up["algorithmic_bytes"] / kernel_s / 1e9 / HBM_PEAK_GBS
            lmm["roofline"]["bytes_over_one_launch_per_shape"] = rp["algorithmic_bytes"] / up["algorithmic_bytes"]
            lmm["roofline"]["bound_note"] = ("simulation launches (two thirds of the kernel time): HBM / VALU issue as in round 4; merged valuation launches: VALU-bound "
                                             "(8 issue slots per chain, period and pair of paths, 3 of them the division's; DESIGN.md 4.6) at about 0.4 of the HBM peak on their own, smaller, byte count")
        except Exception as e:
            lmm["roofline"]["one_launch_per_shape"] = {"error": str(e)[-500:]}
        # (the replays value ONE parameter set eight times: with common rows on, eight identical rows would be one — what a launch SHAPE is worth is measured with every row computed)
        env = dict(os.environ, FMHIP_JIT="sync", FMHIP_COMMON_ROWS="0")
        for key, batch in (("replay_one_at_a_time", 1), ("replay_8_in_lock_step", 8)):
            pj, _ = run([LMM_HIP, "--paths", str(paths), "--mode", "evaluate", "--evaluations", "8", "--jacobian-batch", str(batch), "--warmup-evaluations", str(batch), "--profile"], env=env)
            lmm["roofline"][key] = {"achieved": pj["achieved_GBps"], "frac": pj["achieved_GBps"] / HBM_PEAK_GBS, "launches_per_evaluation": pj["profiled_launches"] / 8,
                                    "kernel_ms_per_evaluation": pj["kernel_ms_total"] / 8}
        # Context, not the metric: the reference's swaption SMILE calibration (LIBORMarketModelCalibrationTest.java; 5 factors,
        # blended local + stochastic volatility, 8 parameters) at the larger of the two path counts the reference publishes wall
        # times for (README.md:242-255) — the only published numbers this path has (other hardware).
        smile_hip = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "bin", "lmm_smile_hip")
        try:
            s1, _ = run([smile_hip, "--paths", "163840"])
            s2, _ = run([smile_hip, "--paths", "163840", "--profile"])
        except Exception as e:                      # context only: never in the way of the metric
            s1 = None
            lmm["smile_calibration"] = {"error": str(e)[-500:]}
        if s1 is not None:
            lmm["smile_calibration"] = {
                "workload": "LIBORMarketModelCalibrationTest inputs: 40 forward rates, 5 factors + stochastic volatility, blended local volatility, "
                            "19 swaptions in log-normal volatility (15 within the 20y horizon), 8 parameters, at most 30 LM iterations",
                "paths": 163840, "seconds": s1["seconds"], "seconds_warm_code_object_cache_profiled": s2["seconds"],
                "objective_evaluations": s1["evaluations"], "lm_iterations": s1["iterations"], "kernel_launches": s1["kernel_launches"],
                "speculative_evaluations_discarded": s1.get("speculative_evaluations_discarded"),
                "speculation": "the eight bumped parameter sets around every trial point are valued beside it (rows of the same launches): an accepted point has its "
                               "Jacobian already, a rejected one wastes eight rows - one recording of the simulation per iteration instead of two; the optimiser's path is unchanged",
                "mean_deviation": s1["mean_deviation"], "rms_deviation": s1["rms_deviation"],
                "acceptance": "abs(mean_deviation) < 1e-2 (LIBORMarketModelCalibrationTest.java:358)", "accepted": abs(s1["mean_deviation"]) < 1e-2,
                "achieved_GBps_all_launches": s2.get("achieved_GBps"),
                "published_reference": {"gpu_seconds": 51.70, "gpu": "GeForce GTX 1080", "cpu_seconds": 719.33, "cpu": "i7-7800X, multi-threaded",
                                        "rms_error": 0.00480, "source": "README.md:254-255"},
                "note": "model and optimiser restated from finmath-lib's documentation (the jar is not vendored): same inputs, same acceptance "
                        "test, not the same optimiser path; other hardware - context only, vs_baseline stays null"}
        # What finmath-lib's own Euler scheme and optimizer would get through the interface: RandomVariable methods and one getAverage() per
        # product, no hold / flush / replication / lock-step batches, every state kept (lmm_hip --finmath-like).
        try:
            fl, _ = run(base + ["--finmath-like"])
            lmm["finmath_like_seconds"] = fl["seconds"]
            lmm["finmath_like"] = {"seconds": fl["seconds"], "kernel_launches": fl["kernel_launches"], "mean_deviation": fl["mean_deviation"], "engine": fl.get("engine"),
                                   "identical_to_the_native_driver": fl["mean_deviation"] == r["mean_deviation"] and fl["rms_deviation"] == r["rms_deviation"],
                                   "what": "the same calibration through methods + getAverage() only; the engine groups time steps itself (fmhip_set_step_grouping)"}
        except Exception as e:
            lmm["finmath_like"] = {"error": str(e)[-500:]}
        # … and with the optimiser's thread pool (LIBORMarketModelCalibrationATMTest.java:319: finmath-lib evaluates the columns of a Jacobian on
        # several threads): the same caller on four threads, an engine each (fmhip_set_thread_engines).  At 1 M paths the calibration is bound by
        # the device and the threads gain nothing; where it is bound by the host (100 k paths) they do.
        try:
            th = {}
            for label, p_, t_ in (("paths_1000000_threads_4", paths, 4), ("paths_100000_threads_1", 100000, 1), ("paths_100000_threads_4", 100000, 4)):
                tj, _ = run([LMM_HIP, "--paths", str(p_), "--mode", "calibrate", "--max-iterations", str(args.lmm_iterations), "--finmath-like", "--threads", str(t_)])
                th[label] = {"seconds": tj["seconds"], "mean_deviation": tj["mean_deviation"], "evaluations": tj["evaluations"]}
            th["identical_to_one_thread_at_1000000_paths"] = th["paths_1000000_threads_4"]["mean_deviation"] == lmm.get("finmath_like", {}).get("mean_deviation")
            th["what"] = "lmm_hip --finmath-like --threads T: Jacobian columns on T caller threads, an engine per thread on the one GPU"
            lmm["finmath_like_threads"] = th
        except Exception as e:
            lmm["finmath_like_threads"] = {"error": str(e)[-500:]}
        # ONE process, a device LIST (fmhip_init_devices: an engine and a worker thread per shard behind the same handles).  A gpurun / driver
        # box lends this rank one GPU, so the list names it twice: two shards of ONE GPU — what the front costs and that it calibrates the
        # same parameters, NOT a scaling figure.  Unmeasured on more than one physical GPU.
        try:
            dl, _ = run(base + ["--devices", "0,0"])
            lmm["device_list_rehearsal"] = {"devices": [0, 0], "seconds": dl["seconds"], "mean_deviation": dl["mean_deviation"],
                                            "same_acceptance": abs(dl["mean_deviation"]) < 2e-4,
                                            "what": "the same calibration in one process behind fmhip_init_devices({0, 0}): two path shards on separate streams of ONE GPU "
                                                    "(expectations = the shards' moments combined in shard order); the cost of the front, not a scaling figure - "
                                                    "unmeasured on more than one physical GPU"}
        except Exception as e:
            lmm["device_list_rehearsal"] = {"error": str(e)[-500:]}
        # … and the same caller with a garbage collector's idea of lifetime: every handle is released LATE, in bursts, by another thread —
        # what a JVM does with the reference's / the Java binding's wrappers (RandomVariableCuda.java:293-305; java/…/DeviceVector.java: a
        # Cleaner action per handle).  lmm_hip --release-lag 100: a collection every 100 ms ("never, until 256 MB of dead wrappers" —
        # --release-lag-bytes — fills the device and is measured by benchmarks/round5/release_lag.sh, not here: the process after it meets a device
        # still busy giving 300 GB back).  The engine must not decide what to store by live handles (escape policy, runtime.hpp): bytes written, launches and
        # the tier they run on are set against the run whose temporaries die at once (RAII, above).
        # (last of the driver runs: a process that held most of the device leaves the next one a device still busy giving the memory back)
        try:
            def gc_block(extra):
                # (a process that held ~100 GB — the vectors of dead wrappers a collector has not released yet — leaves the NEXT process a device
                # that is still giving that memory back: its allocations crawl, 4.4 s become 6.8 s.  Measured, benchmarks/round5/lag_window.sh and
                # profiles/round05b_merged_chains.txt; eight seconds between the processes and the figures are those of a process on its own.)
                time.sleep(8.0)
                g, _ = run(base + ["--finmath-like"] + extra)
                e, e0 = g.get("engine", {}), fl.get("engine", {})
                return {"seconds": g["seconds"], "seconds_over_raii": round(g["seconds"] / fl["seconds"], 3), "kernel_launches": g["kernel_launches"], "mean_deviation": g["mean_deviation"],
                        "identical_to_raii": g["mean_deviation"] == fl["mean_deviation"] and g["rms_deviation"] == fl["rms_deviation"],
                        "algorithmic_bytes_written": e.get("algorithmic_bytes_written"),
                        "bytes_written_over_raii": round(e.get("algorithmic_bytes_written", 0) / max(1, e0.get("algorithmic_bytes_written", 1)), 3),
                        "specialised_launches": g.get("specialised_launches"), "interpreter_launches": e.get("interpreter_launches"),
                        "interpreter_fraction": round(e.get("interpreter_launches", 0) / max(1, g["kernel_launches"]), 4),
                        "hiprtc_compilations": g.get("specialised_kernels", 0) - g.get("specialisations_from_disk_cache", 0),
                        "values_deferred": e.get("values_deferred"), "values_demanded": e.get("values_demanded"),
                        "peak_bytes_reserved": e.get("peak_bytes_reserved"), "release_lag": g.get("release_lag")}
            gc = {"raii": {"seconds": fl["seconds"], "kernel_launches": fl["kernel_launches"], "algorithmic_bytes_written": fl.get("engine", {}).get("algorithmic_bytes_written"),
                           "interpreter_launches": fl.get("engine", {}).get("interpreter_launches"), "peak_bytes_reserved": fl.get("engine", {}).get("peak_bytes_reserved")},
                  "collect_every_100_ms": gc_block(["--release-lag", "100"]),
                  "collect_every_20_ms": gc_block(["--release-lag", "20"]),
                  "what": "lmm_hip --finmath-like --release-lag MS | --release-lag-bytes B: the C++ mirror's handle releases are queued and performed by a collector "
                          "thread, as a JVM's Cleaner would (host/random_variable.hpp: ReleaseLag); a device allocation that fails runs a collection and is retried"}
            lmm["finmath_like_gc"] = gc
        except Exception as e:
            lmm["finmath_like_gc"] = {"error": str(e)[-500:]}
        if cpu_base:
            cj, _ = run([LMM_CPU, "--paths", str(paths), "--mode", "evaluate", "--evaluations", "1"])
            per_eval = cj["seconds_simulation_per_evaluation"] + cj["seconds_valuation_per_evaluation"]
            lmm["cpu_baseline"] = {"value": per_eval * r["evaluations"], "unit": "s", "cores": 1, "kind": "port",
                                   "sample": f"1 objective evaluation of the same model at {paths} paths on the CPU twin = {per_eval:.2f} s, "
                                             f"scaled by the {r['evaluations']} evaluations the calibration needed"}
    return lmm


def brownian_block(fm, launches=8):
    """BASELINE.json configs[2]: BrownianMotionHip 1 M paths x 200 steps x 5 factors (seed 31415 + k) = 4.0 GB of N(0, dt) increments per
    generation, one launch of fm_bm_kernel into a slab the pool hands back from the previous generation (the first one allocates it
    and is not counted); device time of that kernel alone from HIP events on the runtime stream (fmhip_profile_*).  Replaces
    BrownianMotionCudaWithRandomVariableCuda.java:168-178 (one curandGenerateNormal per step and factor).  Then the Heston Monte-Carlo
    of config 3 with vol-of-vol 0 on the last generation: must reproduce the Black-Scholes value 0.18994 within 0.005
    (MonteCarloBlackScholesModelTest.java:156)."""
    mc = importlib.import_module("finmath-lib-cuda-extensions_amd.montecarlo")
    n, steps, factors, dt = N_PATHS, 200, 5, 0.01
    td = fm.TimeDiscretization(0.0, steps, dt)
    bm = fm.BrownianMotionHip(td, factors, n, 31415)
    bm.getBrownianIncrement(0, 0)                      # allocates the 4 GB slab (first touch): not timed
    del bm
    us = []
    fm.profile_enable(True)
    for k in range(launches):
        bm = fm.BrownianMotionHip(td, factors, n, 31416 + k)
        bm.getBrownianIncrement(0, 0)
        ms, count = fm.profile_read()
        us.append(ms * 1e3 / max(1, count))
        if k + 1 < launches:
            del bm
    fm.profile_enable(False)
    nbytes = 4.0 * n * steps * factors
    us = us[2:]                                         # the first generations after the allocation still ramp the clock: min / median / mean of the rest
    avg, med = sum(us) / len(us), sorted(us)[len(us) // 2]
    analytic = mc.black_scholes_call_analytic(1.0, 0.05, 0.30, 2.0, 1.05)
    # The Heston Monte-Carlo of config 3 (Euler, full truncation, factors 0 and 1; xi = 0 is the Black-Scholes limit of
    # MonteCarloBlackScholesModelTest.java:62-85) on the last generation: recorded through the Python mirror under a hold — the engine sees
    # the whole time loop, finds it periodic in the time index and runs it as ONE rolled-loop launch — every launch bracketed by HIP
    # events: device time, algorithmic bytes (4 B x paths x (vectors read + stored) per launch: the two increments of every step are
    # read once, the state lives in registers) and the fraction of the HBM peak.  wall_s is this process recording 200 x 8 methods in
    # Python and is NOT a device figure.  The first pass of each model meets its graph shapes (untimed).
    prev = fm.set_fusion(True)
    heston = {}
    for xi in (0.0, 0.3):
        for rep in range(2):
            fm.synchronize()
            fm.profile_enable(True)
            b0, _ = fm.traffic_stats()
            l0 = fm.pool_stats().n_kernel_launches
            t0 = time.perf_counter()
            with fm.holding():
                price, value = mc.heston_call_mc(bm, 1.0, 0.05, 0.09, 1.0, 0.09, xi, -0.5, 2.0, 1.05)
            wall = time.perf_counter() - t0
            ms, count = fm.profile_read()
            fm.profile_enable(False)
            b1, _ = fm.traffic_stats()
            del value
            fm.jit_wait()
        heston[xi] = {"price": price, "kernel_ms": ms, "launches": fm.pool_stats().n_kernel_launches - l0, "algorithmic_bytes": b1 - b0,
                      "achieved_GBps": (b1 - b0) / (ms * 1e-3) / 1e9 if ms > 0 else None, "frac": (b1 - b0) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
                      "wall_s_python_recording": wall}
    fm.set_fusion(prev)
    del bm
    fm.purge()
    h0 = heston[0.0]
    return {"workload": "BrownianMotionHip 1M paths x 200 steps x 5 factors (BASELINE.json configs[2])", "kernel": "fm::fm_bm_kernel (Philox4x32-10 + LDS-table inverse normal CDF)",
            "bytes_written_per_launch": nbytes, "launches": len(us), "avg_kernel_us": avg, "median_kernel_us": med, "min_kernel_us": min(us), "max_kernel_us": max(us),
            "roofline": {"bound": "hbm", "achieved": nbytes / (avg * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "frac_best_launch": nbytes / (min(us) * 1e-6) / 1e9 / HBM_PEAK_GBS, "frac_median_launch": nbytes / (med * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "timing": f"one HIP event pair per launch on the runtime stream (fmhip_profile_read); {launches} generations, the first two dropped"},
            "normals_per_s": n * steps * factors / (avg * 1e-6),
            "heston_xi0": {"price": h0["price"], "black_scholes_analytic": analytic, "abs_error": abs(h0["price"] - analytic),
                           "acceptance": "abs error < 0.005 (MonteCarloBlackScholesModelTest.java:156)", "accepted": abs(h0["price"] - analytic) < 0.005,
                           "kernel_ms": h0["kernel_ms"], "launches": h0["launches"], "algorithmic_bytes": h0["algorithmic_bytes"],
                           "roofline": {"bound": "hbm", "achieved": h0["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": h0["frac"],
                                        "timing": "HIP event pair per launch, summed over the simulation's launches (time loop as one rolled-loop launch)"},
                           "wall_s_python_recording": h0["wall_s_python_recording"]},
            "heston_xi03": {"price": heston[0.3]["price"], "kernel_ms": heston[0.3]["kernel_ms"], "launches": heston[0.3]["launches"], "algorithmic_bytes": heston[0.3]["algorithmic_bytes"],
                            "roofline": {"bound": "hbm", "achieved": heston[0.3]["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": heston[0.3]["frac"]},
                            "wall_s_python_recording": heston[0.3]["wall_s_python_recording"]}}


def rendezvous_nonce(world, rank):
    """Rendezvous WITHOUT touching the GPU: a TCP store on MASTER_ADDR:MASTER_PORT hands every rank the launch's nonce now and
    carries the process group of the stream leg later.  Under torchrun the store is hosted by the launcher's agent
    (TORCHELASTIC_USE_AGENT_STORE) and every rank is a client; launched by hand, rank 0 hosts it."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    hosted_by_agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, rank == 0 and not hosted_by_agent,
                          timeout=datetime.timedelta(seconds=900))
    key = "fmhip_nonce_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    if rank == 0:
        store.set(key, str(int.from_bytes(os.urandom(6), "little")))
    return store, int(store.get(key).decode())


def self_launch(n_gpus):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a child process on a free
    local port, pass its stderr through, print the one JSON line of its rank 0 and return its exit code.  Called before anything
    in this process has touched the GPU (no torch import, no libfmhip)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    if lines:
        print(lines[-1], flush=True)
    elif child.returncode == 0:
        print(child.stdout[-2000:], file=sys.stderr)
        return 1
    return child.returncode


def dry_run(args, world, rank):
    """FMHIP_BENCH_DRY=1 (tests/test_parallel_gloo.py, no GPU): the launch path of the real run — rendezvous through the launcher's
    store, a process group on that store, max-over-ranks of a timed region, ONE line from rank 0 — with the GPU legs left out and gloo
    in place of RCCL.  Never a measurement: the line says so and carries no value."""
    import torch
    import torch.distributed as dist
    store, nonce = rendezvous_nonce(world, rank) if world > 1 else (None, 0)
    if world > 1:
        dist.init_process_group(backend="gloo", store=store, rank=rank, world_size=world)
    t0 = time.perf_counter()
    mine = torch.tensor([float(nonce), float(rank)], dtype=torch.float64)
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(everyone, mine)
        dist.barrier()
    else:
        everyone = [mine]
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    same = all(float(v[0]) == float(nonce) for v in everyone) and [int(v[1]) for v in everyone] == list(range(world))
    if rank == 0:
        print(json.dumps({"metric": "dry run of the launch path (no GPU leg ran)", "value": None, "unit": "path-ops/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "dry": True, "ranks_agree_on_nonce": same,
                          "region_s_max_over_ranks": float(elapsed.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if same else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["both", "stream", "lmm"], default="both")
    ap.add_argument("--sustained-seconds", type=float, default=2.5, help="length of the sustained leg (back-to-back launches); 0 = skip")
    ap.add_argument("--lmm-iterations", type=int, default=12)
    ap.add_argument("--exchange-every", type=int, default=8, help="N > 1: steps whose expectation partials travel in one all-gather")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="independent (x,y,z) triples per launch")
    ap.add_argument("--paths", type=int, default=N_PATHS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)      # internal: one process of the all-cores CPU baseline
    args = ap.parse_args()
    if args.cpu_worker > 0:
        print(json.dumps(cpu_baseline(args.cpu_worker, -1.0)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` typed without a launcher: this process has imported neither torch nor the library and has made
        # no GPU call, so it starts the N ranks itself — torchrun as a CHILD process (never an exec), one rank per GPU — and relays
        # rank 0's one JSON line and the launcher's return code.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:                     # a launcher's WORLD_SIZE is what actually runs: report that, never exit without a line
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; running and reporting n_gpus = {world}", file=sys.stderr)
        args.gpus = world
    if os.environ.get("FMHIP_BENCH_DRY") == "1":
        sys.exit(dry_run(args, world, rank))
    # FMHIP_BENCH_FORCE_DIST=1 exercises the collective paths with a single rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("FMHIP_BENCH_FORCE_DIST") == "1"
    store, nonce = rendezvous_nonce(world, rank) if use_dist else (None, None)
    lmm = None
    if args.workload in ("both", "lmm"):
        try:
            lmm = lmm_leg(args, world, rank, nonce, cpu_base=not args.no_cpu_baseline, store=store)     # child processes; this process has not touched the GPU yet
        except Exception as e:
            if args.workload == "lmm":
                raise
            lmm = {"error": str(e)[-1500:]} if rank == 0 else None                    # the stream half is still measured and reported
    if args.workload == "lmm":
        if rank == 0:
            line = dict(lmm)
            line.update({"steps": lmm["lm_iterations"], "warmup": 0, "ms_per_step": lmm["value"] / max(1, lmm["lm_iterations"]) * 1e3,
                         "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": lmm.pop("workload")}})
            print(json.dumps(line), flush=True)
        return
    # stdout carries the ONE JSON line and nothing else: native libraries (RCCL prints a version banner on the first
    # communicator) write to file descriptor 1 directly, so fd 1 is pointed at stderr and the line goes to the saved fd.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch                         # before libfmhip: one HIP runtime in the process (see _native.lib)
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if use_dist:
        dist.init_process_group(backend="nccl", store=store, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    fm.init(local_rank)
    dev_name, cus, hbm = fm.device_info()
    n, B = args.paths, args.batch

    prog = build_stream_s(fm)            # explicit program: queued for the specialised-kernel tier (hiprtc) at creation
    rows = synthetic_inputs(fm, B, n, rank)
    out_rows = [[fm.DeviceVector.filled(n, 0.0)] for _ in range(B)]
    fm.synchronize()

    ext_stream = torch.cuda.ExternalStream(fm.stream_ptr(), device=torch.device("cuda", local_rank))
    dev = f"cuda:{local_rank}"
    # Expectation buffers: G consecutive steps write their B x {Σ, Σ², min, max} partials into the G slots of a bank; a full bank
    # travels in ONE all-gather (RCCL, on its own stream) while the steps of the other bank compute.  Why G > 1: the exchange needs an
    # event on the runtime stream, and on this platform an event between two launches costs 13-19 µs of device time (compare
    # roofline.avg_kernel_us_one_event_pair_per_launch) — per step that is 7-10 % of a 180 µs launch, per 8 steps 1 %.
    G = max(1, args.exchange_every) if use_dist else 1
    partials = [torch.zeros(G * B * 4, dtype=torch.float64, device=dev) for _ in range(2)]
    gathers = [torch.zeros(world * G * B * 4, dtype=torch.float64, device=dev) for _ in range(2)] if use_dist else None
    partial = partials[0]
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None
    computed = [torch.cuda.Event() for _ in range(2)] if use_dist else None
    gathered_ev = [None, None]
    step_no = [0]
    rccl_events = []                     # (begin, end) on the collective stream, one pair per all-gather of the timed region
    timing_rccl = [False]

    def exchange(bank):
        # the single exchange of the path: expectation partials of all ranks, overlapped with the steps of the other bank
        computed[bank].record(ext_stream)
        comm_stream.wait_event(computed[bank])
        with torch.cuda.stream(comm_stream):
            if timing_rccl[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(comm_stream)
            dist.all_gather_into_tensor(gathers[bank], partials[bank])
            if timing_rccl[0]:
                e1.record(comm_stream)
                rccl_events.append((e0, e1))
            gathered_ev[bank] = torch.cuda.Event()
            gathered_ev[bank].record(comm_stream)

    def step():
        # one launch: 12 ops over B triples + fused reductions; moments stay on the device
        bank, slot = (step_no[0] // G) & 1, step_no[0] % G
        step_no[0] += 1
        if use_dist and slot == 0 and gathered_ev[bank] is not None and not gathered_ev[bank].query():
            ext_stream.wait_event(gathered_ev[bank])         # this bank's previous content has not been sent yet (rare): order after it
        prog.run_into(rows, out_rows, want_moments=False, device_moments=partials[bank].data_ptr() + slot * B * 32)
        if use_dist and slot == G - 1:
            exchange(bank)

    def finish_exchange():
        # a bank that the last steps filled only partly travels too: every timed step's expectations are exchanged inside the timed region
        if use_dist and step_no[0] % G != 0:
            exchange((step_no[0] // G) & 1)
            step_no[0] += G - step_no[0] % G

    def barrier_sync():
        fm.synchronize()                 # this rank's work (runtime stream + collective stream) is done …
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()               # … and so is everybody else's
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    finish_exchange()
    fm.jit_wait()                        # steady state: the background compilation of the specialised kernel has finished
    tier, jit_vgprs = prog.tier()
    # Sustained leg, BEFORE the timed region: >= 2 s of back-to-back launches of the headline program, timed in chunks of 50
    # launches (one HIP event pair per chunk on the runtime stream).  It is a measurement of its own (the clock the chip holds under
    # this load settles within the first ~0.1-0.4 s) and it leaves the chip in its steady state for the K timed steps that
    # follow without a pause: a 20-step region entered from an idle chip measures the clock ramp (first box of round 2: 221 µs
    # per launch from idle against 179 µs sustained), not the kernel.
    alg_bytes = 4.0 * (3 + 1) * n * B
    sustained = None
    if args.sustained_seconds > 0:
        chunk = 50
        n_chunks = max(4, int(args.sustained_seconds / (chunk * 200e-6)))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_chunks + 1)]
        evs[0].record(ext_stream)
        for c in range(n_chunks):
            for _ in range(chunk):
                prog.run_into(rows, out_rows, want_moments=False, device_moments=partial.data_ptr())
            evs[c + 1].record(ext_stream)
        fm.synchronize()
        us = [evs[c].elapsed_time(evs[c + 1]) * 1e3 / chunk for c in range(n_chunks)]
        tail = us[len(us) // 2:]
        sustained = {"seconds": sum(us) * chunk / 1e6, "launches": n_chunks * chunk, "avg_kernel_us": sum(us) / len(us), "min_chunk_avg_us": min(us),
                     "max_chunk_avg_us": max(us), "first_chunk_avg_us": us[0], "second_half_avg_us": sum(tail) / len(tail),
                     "frac": alg_bytes / (sum(us) / len(us) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "second_half_frac": alg_bytes / (sum(tail) / len(tail) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "timing": f"HIP events on the runtime stream around chunks of {chunk} back-to-back launches, run right before the timed region"}
        for _ in range(args.warmup):     # the W warm-up steps of the contract, directly in front of the timed region
            step()
    step()
    finish_exchange()
    barrier_sync()
    # HIP events on the RUNTIME stream (the stream the kernel is launched on) around the timed region: device time of the K
    # back-to-back launches, gaps included — the live figure behind roofline.achieved
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timing_rccl[0] = use_dist
    t0 = time.perf_counter()
    ev_begin.record(ext_stream)
    for _ in range(args.steps):
        step()
    finish_exchange()
    ev_end.record(ext_stream)
    barrier_sync()
    elapsed = time.perf_counter() - t0
    timing_rccl[0] = False
    region_kernel_s = ev_begin.elapsed_time(ev_end) / 1e3 / args.steps
    per_gpu_kernel_s = [region_kernel_s]
    rccl = None
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([region_kernel_s], dtype=torch.float64, device=f"cuda:{local_rank}")
        everyone = torch.zeros(world, dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_gather_into_tensor(everyone, mine)
        per_gpu_kernel_s = [float(v) for v in everyone.tolist()]
        lat_ms = [a.elapsed_time(b) for a, b in rccl_events]
        rccl = {"calls": len(lat_ms), "summed_latency_ms": float(sum(lat_ms)), "max_latency_ms": float(max(lat_ms)) if lat_ms else 0.0,
                "bytes_per_rank_and_call": G * B * 32, "steps_per_call": G,
                "what": f"one all-gather of the B x {{sum, sumsq, min, max}} fp64 expectation partials of {G} consecutive steps, on its own stream, "
                        "overlapped with the following steps; latency = device time of the collective (HIP events on that stream, rank 0)"}

    # combined expectations (sanity: finite, and identical on every rank by construction)
    par = importlib.import_module("finmath-lib-cuda-extensions_amd.parallel")
    last_bank = ((step_no[0] - 1) // G) & 1       # the bank of the last exchange; slot 0 of it was written by a timed step
    if use_dist:
        comb = par.combine_moments(gathers[last_bank].view(world, G, B, 4)[:, 0].contiguous())
    else:
        comb = partials[last_bank].view(G, B, 4)[0]
    mean_w = float((comb[:, 0] / (world * n)).mean().item())
    assert np.isfinite(mean_w)
    m0 = [float(v) for v in comb[0].tolist()]      # {sum, sumsq about 0, min, max} of w for triple 0

    # cross-check: one event pair per launch (what a tracing profiler sees), separate short pass right after the timed region
    fm.profile_enable(True)
    for _ in range(max(5, min(args.steps, 20))):
        prog.run_into(rows, out_rows, want_moments=False, device_moments=partial.data_ptr())
    kernel_ms, n_launch = fm.profile_read()
    fm.profile_enable(False)
    per_launch_kernel_s = kernel_ms / 1e3 / max(1, n_launch)
    avg_kernel_s = region_kernel_s
    achieved = alg_bytes / avg_kernel_s / 1e9

    def measure(pr, warm=150, launches=300):
        """Device time per launch of `pr`: `launches` back-to-back launches between two HIP events on the runtime stream, entered
        from `warm` launches of the same program (same conditions as the headline: no idle chip in front of the region)."""
        for _ in range(warm):
            pr.run_into(rows, out_rows, want_moments=False, device_moments=partial.data_ptr())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ext_stream)
        for _ in range(launches):
            pr.run_into(rows, out_rows, want_moments=False, device_moments=partial.data_ptr())
        b.record(ext_stream)
        fm.synchronize()
        return a.elapsed_time(b) / 1e3 / launches

    # additional information: the same program on the interpreter tier (what a program runs on until its kernel is compiled)
    interp_kernel_s = None
    if tier == 1:
        prev_jit = fm.set_jit(fm.JIT_OFF)
        interp_kernel_s = measure(prog)
        fm.set_jit(prev_jit)

    # additional information (not the headline): the same stream with FMHIP_MATH_FAST (hardware exp/log, <= 2 ulp), and the
    # headline program again right after it under the same conditions (exact vs fast on the same box, same minute)
    fm.set_math_mode(fm.MATH_FAST)
    prog_fast = build_stream_s(fm)
    fm.set_math_mode(fm.MATH_EXACT)
    fm.jit_wait()
    fast_kernel_s = measure(prog_fast)
    exact_again_s = measure(prog)

    # HBM traffic of the same kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950
    # correction applied): NOT measured in this run — counters need a profiler pass of their own — but OFFLINE on this exact
    # workload, committed under profiles/ (newest round first); null for other shapes.
    traffic, traffic_source = None, None
    for name in ("round05_hbm_traffic.json", "round04_hbm_traffic.json", "round03_hbm_traffic.json", "round02_hbm_traffic.json", "round01_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                prof = json.load(fh)
            if prof["workload"]["paths"] == n and prof["workload"]["batch"] == B:
                traffic, traffic_source = prof["hbm_bytes_per_launch"], f"offline rocprofv3 --pmc passes of this workload: profiles/{name}"
                break
        except Exception:
            continue

    brownian = None
    if world == 1 and args.workload == "both":
        del rows, out_rows                      # the stream half's 1.5 GB of inputs are no longer needed
        rows = out_rows = None
        try:
            brownian = brownian_block(fm)
        except Exception as e:                  # never in the way of the headline
            brownian = {"error": str(e)[-500:]}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = N_OPS * n * B * world / (elapsed / args.steps)
        line = {
            "metric": "Monte-Carlo path-ops/sec (fused RandomVariable stream)",
            "value": value, "unit": "path-ops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RandomVariableHipFactory 1M-path elementwise+reduction microbench: stream S "
                                   "(12 fused path-ops, 3 inputs, 1 output) + fused {sum,sumsq,min,max}, "
                                   f"batch of {B} independent (x,y,z) triples per launch",
                       "inputs": "x, y, z = java.util.Random(31415 / 27182 / 16180).nextDouble() (+0.5 for y, z), narrowed to fp32 "
                                 "(SURVEY.md 8d config 2); triple b of rank r = block (r*batch + b) of each stream",
                       "paths_per_gpu": n, "batch": B, "ops_per_path": N_OPS,
                       "parallelism": f"path-shard x{world}" if world > 1 else "single GPU",
                       "device": dev_name, "compute_units": cus},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": ("fm_jit_<hash>_t (specialised kernel of stream S, generated + compiled at run time)" if tier == 1
                                    else "fm::fm_program_kernel<1, false, 8, 9, 3, float __vector(9)>"),
                         "tier": "specialised" if tier == 1 else "interpreter", "vgprs": jit_vgprs,
                         "avg_kernel_us": avg_kernel_s * 1e6, "avg_kernel_us_one_event_pair_per_launch": per_launch_kernel_s * 1e6,
                         "timing": "HIP events on the runtime stream around the K timed launches, duration / K (launch gaps included)",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "per_gpu_frac": [alg_bytes / t / 1e9 / HBM_PEAK_GBS for t in per_gpu_kernel_s]},
            "sustained": sustained,
            "rccl": rccl,
            "mean_w": mean_w,
            "interpreter_tier": None if interp_kernel_s is None else {
                "note": "same program on the bytecode interpreter kernel (tier 0, no compilation)",
                "avg_kernel_us": interp_kernel_s * 1e6, "achieved_GBps": alg_bytes / interp_kernel_s / 1e9,
                "frac": alg_bytes / interp_kernel_s / 1e9 / HBM_PEAK_GBS},
            "jit": fm.jit_stats(),
            "fast_math": {"note": "same workload with fmhip_set_math_mode(FMHIP_MATH_FAST): exp/log on v_exp_f32/v_log_f32, "
                                  "within 2 fp32 ulp (accuracy class of the reference kernels' CUDA expf/logf); not the headline",
                          "avg_kernel_us": fast_kernel_s * 1e6, "achieved_GBps": alg_bytes / fast_kernel_s / 1e9,
                          "frac": alg_bytes / fast_kernel_s / 1e9 / HBM_PEAK_GBS,
                          "path_ops_per_s_per_gpu": N_OPS * n * B / fast_kernel_s,
                          "exact_measured_the_same_way_us": exact_again_s * 1e6, "exact_over_fast": exact_again_s / fast_kernel_s,
                          "timing": "300 back-to-back launches between two HIP events, entered from 150 launches of the same program"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            # parity in the run itself: the timed kernel's expectations for triple 0 against the CPU twin on the same inputs
            tw = line["cpu_baseline"].pop("twin_moments_triple0")
            gpu_avg = m0[0] / n
            line["parity_triple0"] = {"gpu": {"average": gpu_avg, "min": m0[2], "max": m0[3]}, "cpu_twin": tw,
                                      "average_rel_diff": abs(gpu_avg - tw["average"]) / abs(tw["average"]),
                                      "min_max_identical": m0[2] == tw["min"] and m0[3] == tw["max"],
                                      "ok": abs(gpu_avg - tw["average"]) <= 1e-12 * abs(tw["average"]) and m0[2] == tw["min"] and m0[3] == tw["max"]}
        if brownian is not None:
            line["brownian"] = brownian
        if lmm is not None:
            line["lmm"] = lmm
        print(json.dumps(line), file=json_out, flush=True)

    rows = out_rows = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
