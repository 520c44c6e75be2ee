"""The engine's host runtime (csrc/runtime.cpp, abi.cpp, jit.cpp: node pools, intrusive reference counts, replica descriptions, tickets, the
pinned arena, plan caches) under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer, on a TEST-ONLY null device
(tests/nulldev/null_hip.cpp: device memory = host memory, launches compute nothing) — no GPU needed; GPU sanitizers do not exist on this
pool.  tests/nulldev/drive.cpp replays the graph shapes of the GPU tests (replicas in the ten orders of test_gpu_replicas.py, long
chains through segments / rolled / peeled plans, expectations taken along, values given up, tickets out of order, the row-table ring
and the moments arena wrapping at tiny sizes, eight threads, shutdown and re-initialisation) and, in a sweep of its own, a failing
allocation in the middle of a replicated launch of 1100 members; `lagging`: a caller whose handles are released late, in bursts, by a
collector thread — the lifetime contract of a JVM (values left unstored, computed from their recipes on demand); `oom`: a device that is
full.  The null device is not a back end and never ships."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
NULLDEV = os.path.join(HERE, "nulldev")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", NULLDEV, "-j8", "asan", "tsan", "lmm_asan", "lmm_tsan"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return os.path.join(NULLDEV, "build")


def run(binary, tmp_path, *scenarios, **extra_env):
    env = dict(os.environ, FMHIP_JIT_CACHE_DIR=str(tmp_path / "code_objects"), FMHIP_JIT_PACK_DIR="off", FMHIP_RING_BYTES="16384", FMHIP_ARENA_BYTES="4096",
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    env.update(extra_env)
    return subprocess.run([binary, *scenarios], capture_output=True, text=True, timeout=600, env=env)


def test_host_runtime_is_clean_under_address_and_undefined_behaviour_sanitizers(built, tmp_path):
    r = run(os.path.join(built, "drive_asan"), tmp_path)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stdout[-1500:] + r.stderr[-6000:]
    assert r.stdout.count("done") == 16          # eight scenarios, twice (shutdown and re-initialisation in between)
    again = run(os.path.join(built, "drive_asan"), tmp_path)      # once more: the code objects of the first run come from the cache directory
    assert again.returncode == 0 and "Sanitizer" not in again.stderr, again.stderr[-6000:]


def test_host_runtime_is_clean_under_thread_sanitizer(built, tmp_path):
    r = run(os.path.join(built, "drive_tsan"), tmp_path)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout[-1500:] + r.stderr[-6000:]
    assert r.stdout.count("done") == 16


def test_a_failing_allocation_inside_a_replicated_launch_leaves_nothing_behind(built, tmp_path):
    """1101 vectors + 1 are allocated before the flush; the 2 x 1101 buffers of the launch follow in two parts of 1024 and 77 members: the
    hook is swept over that stretch.  Every position must leave a process that ends cleanly — no leak, no use after free, no bad status."""
    for at in (1103, 1500, 2125, 2127, 2400, 3150, 3151, 3300, 10**9):
        r = run(os.path.join(built, "drive_asan"), tmp_path, "failure", FMHIP_TEST_FAIL_ALLOC_AT=str(at))
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (at, r.stdout[-500:] + r.stderr[-6000:])
        assert "failure done" in r.stdout


def test_a_full_device(built, tmp_path):
    """The pool's last resort (drop every cached slab, try once more) on the first allocation of a new size class, the out-of-memory
    status, and a fused launch that cannot get its output — on a null device of 800 MB.  Round 5: heap corruption on the real device
    (a reference into the pool's size-class table held across the purge that erases the entry); the native driver under a lagging
    collector and a device that fills up within a few evaluations (collections forced by out-of-memory statuses, retries)."""
    r = run(os.path.join(built, "drive_asan"), tmp_path, "oom", FMNULL_DEVICE_BYTES="800000000", FMHIP_POOL_HEADROOM_BYTES="1000000")
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout[-500:] + r.stderr[-6000:]
    assert r.stdout.count("oom done") == 2
    for policy in ("0", "1"):
        d = run(os.path.join(built, "lmm_asan"), tmp_path, "--paths", "3000", "--mode", "calibrate", "--max-iterations", "1", "--finmath-like", "--release-lag", "50",
                FMNULL_DEVICE_BYTES="400000000", FMHIP_POOL_HEADROOM_BYTES="20000000", FMHIP_ESCAPE_POLICY=policy, FMHIP_RING_BYTES="1048576", FMHIP_ARENA_BYTES="65536")
        assert d.returncode == 0 and "Sanitizer" not in d.stderr and "runtime error" not in d.stderr, (policy, d.stdout[-500:] + d.stderr[-6000:])
        assert '"collections_forced_by_out_of_memory": 0' not in d.stdout or policy == "1"


@pytest.mark.parametrize("shards", [2, 3])
def test_a_device_list_is_clean_under_both_sanitizer_builds(built, tmp_path, shards):
    """The same scenarios behind fmhip_init_devices (csrc/sharded.cpp: one engine and one worker thread per shard, the caller's calls replayed
    through a single-producer ring per worker, reads and reductions gathered): ASan + UBSan, then ThreadSanitizer."""
    r = run(os.path.join(built, "drive_asan"), tmp_path, FMNULL_DEVICES=str(shards))
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout[-1500:] + r.stderr[-6000:]
    assert r.stdout.count("done") == 16
    t = run(os.path.join(built, "drive_tsan"), tmp_path, FMNULL_DEVICES=str(shards))
    assert t.returncode == 0 and "ThreadSanitizer" not in t.stderr, t.stdout[-1500:] + t.stderr[-6000:]
    assert t.stdout.count("done") == 16


def test_the_expectation_collective_of_a_device_list(built, tmp_path):
    """Expectations wanted ON the devices of a device list: with distinct device indices the front asks RCCL (here: the stand-ins of
    tests/nulldev/null_rccl.cpp, found by dlsym as librccl.so's would be) for a communicator per device and issues ONE all-gather for
    all of them inside ncclGroupStart / ncclGroupEnd, a combine kernel per device follows; with a repeated index the host combines.
    Every device must hold the bits the host-side reduction returns (scenario `collective`), under ASan and TSan, 4 and 8 devices."""
    for env in ({"FMNULL_DEVICES": "4", "FMNULL_DISTINCT": "1"}, {"FMNULL_DEVICES": "8", "FMNULL_DISTINCT": "1"}, {"FMNULL_DEVICES": "3"}):
        r = run(os.path.join(built, "drive_asan"), tmp_path, "collective", "lagging", **env)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (env, r.stdout[-500:] + r.stderr[-6000:])
        assert r.stdout.count("done") == 4
    t = run(os.path.join(built, "drive_tsan"), tmp_path, "collective", "lagging", "basic", FMNULL_DEVICES="4", FMNULL_DISTINCT="1")
    assert t.returncode == 0 and "ThreadSanitizer" not in t.stderr, t.stdout[-500:] + t.stderr[-6000:]


def test_the_native_lmm_driver_is_clean_under_both_sanitizer_builds(built, tmp_path):
    """The real workload instead of hand-made scenarios: host/lmm_hip_main.cpp (the LMM calibration driver over the C-ABI) linked against the
    null device — Jacobian batches as replica descriptions, payoff values given up, expectation tickets collected one batch late, rolled and
    peeled plans of the 80-rate simulation, the caller without hints (every method a node, time steps grouped by the engine), a device list
    of two shards, and the path-sharded form whose expectation partials are gathered into a device buffer (RCCL replaced by stand-ins).
    The numbers are meaningless (the null device computes nothing): what is checked is that every one of these runs ends with status 0 and
    nothing for ASan / UBSan / LeakSanitizer, resp. ThreadSanitizer, to report."""
    asan, tsan = os.path.join(built, "lmm_asan"), os.path.join(built, "lmm_tsan")
    base = ["--paths", "3000"]
    cases = [["--mode", "calibrate", "--max-iterations", "1"],
             ["--mode", "evaluate", "--evaluations", "3", "--finmath-like"],
             ["--mode", "calibrate", "--max-iterations", "1", "--devices", "0,0"],
             ["--mode", "calibrate", "--max-iterations", "1", "--world", "1", "--rank", "0", "--nccl-id-file", str(tmp_path / "id"), "--nccl-nonce", "5"],
             # an engine per caller thread (fmhip_set_thread_engines): the Jacobian's columns on four threads, the Brownian increments and the
             # initial curve owned by the first engine and imported by the others, results read and released across threads; two iterations,
             # so that the second pool of threads takes over the engines the first one left behind
             ["--mode", "calibrate", "--max-iterations", "2", "--finmath-like", "--threads", "4"],
             # the caller without hints whose handles die when a collector thread says so (host/random_variable.hpp: ReleaseLag): every 5 ms,
             # and "never, until 2 MB of dead wrappers have piled up"
             ["--mode", "calibrate", "--max-iterations", "1", "--finmath-like", "--release-lag", "5"],
             ["--mode", "evaluate", "--evaluations", "12", "--finmath-like", "--release-lag-bytes", "2000000"]]
    for args in cases:
        r = run(asan, tmp_path, *base, *args, FMHIP_RING_BYTES="1048576", FMHIP_ARENA_BYTES="65536")
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (args, r.stdout[-500:] + r.stderr[-6000:])
        assert '"evaluations"' in r.stdout
        # the swaptions of an exercise date run as merged launches (runtime.cpp: merge_families) in every one of these forms
        assert int(r.stdout.split('"merged_launches": ')[1].split(",")[0]) > 0, (args, r.stdout[-800:])
    for args in (cases[0], cases[2], cases[4], cases[5]):
        t = run(tsan, tmp_path, *base, *args, FMHIP_RING_BYTES="1048576", FMHIP_ARENA_BYTES="65536")
        assert t.returncode == 0 and "ThreadSanitizer" not in t.stderr, (args, t.stdout[-500:] + t.stderr[-6000:])


def test_a_kernel_another_caller_thread_is_compiling_is_waited_for_and_the_waiter_is_woken(built, tmp_path):
    """Thread engines share one compiler.  With FMHIP_JIT=sync a caller compiles on its own thread; a second thread that asks for the same
    kernel meanwhile waits for it — and must be woken when the first is done (until round 5 only the compiler's worker thread woke such
    waiters: four thread engines that met the merged kernels, which take tens of seconds each, waited for good: the kernel pack recording
    was killed after seven silent minutes).  The null device's hiprtc takes FMNULL_COMPILE_MS per kernel here; no code-object cache."""
    t = run(os.path.join(built, "lmm_tsan"), tmp_path, "--paths", "3000", "--mode", "calibrate", "--max-iterations", "2", "--finmath-like", "--threads", "4",
            FMHIP_JIT="sync", FMHIP_JIT_CACHE_DIR="off", FMNULL_COMPILE_MS="15", FMHIP_RING_BYTES="1048576", FMHIP_ARENA_BYTES="65536")
    assert t.returncode == 0 and "ThreadSanitizer" not in t.stderr, t.stdout[-500:] + t.stderr[-4000:]
    assert '"evaluations"' in t.stdout


def test_thread_engines_are_clean_under_both_sanitizer_builds(built, tmp_path):
    """Every scenario again with an engine per caller thread (fmhip_set_thread_engines; csrc/abi.cpp, namespace te): the eight threads of
    `threads` record into engines of their own; `shared` crosses them — a pending vector of the main thread as an operand of six threads
    (exported by its owner, imported as an aliasing leaf), the main thread's program run by the others with mixed operands, tickets begun
    on one thread and ended on another, vectors handed over and released across engines; then shutdown (engines retired) and the same again."""
    r = run(os.path.join(built, "drive_asan"), tmp_path, FMNULL_THREAD_ENGINES="1")
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout[-1500:] + r.stderr[-6000:]
    assert r.stdout.count("done") == 16
    t = run(os.path.join(built, "drive_tsan"), tmp_path, FMNULL_THREAD_ENGINES="1")
    assert t.returncode == 0 and "ThreadSanitizer" not in t.stderr, t.stdout[-1500:] + t.stderr[-6000:]
    assert t.stdout.count("done") == 16

