// drive.cpp — drives the engine's host runtime through its C-ABI on the TEST-ONLY null device (null_hip.cpp) under the sanitizers:
// the graph shapes of the GPU tests — record / flush / clone / expand / release in the orders of tests/test_gpu_replicas.py, long
// chains through segments, rolled and peeled plans, expectations taken along, values given up, tickets begun and ended out of order,
// the row-table ring and the moments arena wrapping at tiny sizes, a failing allocation in the middle of a replicated launch, eight
// threads as tests/test_gpu_threads.py, shutdown and re-initialisation.  Nothing is computed: statuses are checked, the sanitizers do
// the rest.   usage: drive [scenario …]   (none = all)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fmhip.h"

#define OK(x) do { const int st_ = (x); if (st_ != FMHIP_OK) { std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, st_, fmhip_last_error()); std::abort(); } } while (0)
#define EXPECT(x, code) do { const int st_ = (x); if (st_ != (code)) { std::fprintf(stderr, "%s:%d: %s -> %d, expected %d (%s)\n", __FILE__, __LINE__, #x, st_, (int)(code), fmhip_last_error()); std::abort(); } } while (0)

typedef fmhip_vec V;
static V filled(int64_t n, double v) { V h = 0; OK(fmhip_vec_create_filled(n, v, &h)); return h; }
static V u1(int op, V a) { V o = 0; OK(fmhip_call_v1s0(op, a, &o)); return o; }
static V s1(int op, V a, double s) { V o = 0; OK(fmhip_call_v1s1(op, a, s, &o)); return o; }
static V b2(int op, V a, V b) { V o = 0; OK(fmhip_call_v2s0(op, a, b, &o)); return o; }
static V s2(int op, V a, V b, double s) { V o = 0; OK(fmhip_call_v2s1(op, a, b, s, &o)); return o; }
static void rel(V h) { OK(fmhip_vec_release(h)); }
static void rel(std::vector<V>& v) { for (V h : v) rel(h); v.clear(); }

// tests/test_gpu_replicas.py: record() — two roots sharing an inner value; long_chain() — a backward induction of `periods` periods
static std::vector<V> record(const V* vecs, V shared, const double* s, std::vector<V>* inner = nullptr) {
    const V d = s2(FMHIP_OP_DISCOUNT, vecs[0], vecs[1], s[0]), t = s1(FMHIP_OP_MULT_S, d, s[1]);
    rel(d);
    const V ap = s2(FMHIP_OP_ADDPRODUCT_VS, t, shared, s[2]), u = b2(FMHIP_OP_ADD, ap, vecs[2]);
    rel(ap);
    if (inner) inner->push_back(t); else rel(t);
    const V f = s1(FMHIP_OP_FLOOR_S, u, s[3]), w = u1(FMHIP_OP_SQUARED, f);
    rel(f);
    return { u, w };
}
static std::vector<V> long_chain(const V* vecs, V shared, const double* s, int periods = 30) {
    V value = 0;
    for (int p = 0; p < periods; ++p) {
        V libor = vecs[p % 3];
        bool own = false;
        if (p % 5 == 0) { libor = s1(FMHIP_OP_MULT_S, vecs[p % 3], 1.0 + 0.013 * p); own = true; }
        const V a = s1(FMHIP_OP_SUB_S, libor, s[0]), payoff = s1(FMHIP_OP_MULT_S, a, s[1]);
        rel(a);
        V next = payoff;
        if (value) { next = b2(FMHIP_OP_ADD, value, payoff); rel(value); rel(payoff); }
        value = s2(FMHIP_OP_DISCOUNT, next, libor, s[2]);
        rel(next);
        if (own) rel(libor);
    }
    const V fl = s1(FMHIP_OP_FLOOR_S, value, 0.0), ex = u1(FMHIP_OP_EXP, shared), out = b2(FMHIP_OP_DIV, fl, ex);
    rel(value); rel(fl); rel(ex);
    return { out };
}

struct Inputs { std::vector<std::vector<V>> dev; V shared = 0; std::vector<std::vector<double>> scal; int copies; int64_t n; };
static Inputs make_inputs(int64_t n, int copies) {
    Inputs in; in.copies = copies; in.n = n;
    for (int j = 0; j <= copies; ++j) { in.dev.push_back({ filled(n, 0.5 + j), filled(n, 1.0 + 0.1 * j), filled(n, 0.75) }); in.scal.push_back({ 0.5 + 0.1 * j, 1.25 - 0.05 * j, 0.25, 0.3 + j }); }
    in.shared = filled(n, 0.1);
    return in;
}
static void free_inputs(Inputs& in) { for (auto& r : in.dev) rel(r); rel(in.shared); in.dev.clear(); }
static void read_all(const std::vector<V>& vs, int64_t n) { std::vector<float> h((size_t)n); for (V v : vs) OK(fmhip_vec_read_float(v, h.data(), n)); }

typedef std::function<std::vector<V>(const V*, V, const double*)> Chain;
// the original recorded, `copies` copies as a description; what happens before the flush is the case
static void replica_case(const Chain& chain, int which, bool jit_sync) {
    Inputs in = make_inputs(/*n*/ 1031, /*copies*/ 4);
    int prev = 0;
    OK(fmhip_set_jit(jit_sync ? FMHIP_JIT_SYNC : FMHIP_JIT_OFF, &prev));
    OK(fmhip_fusion_hold(1, nullptr));
    std::vector<V> roots = chain(in.dev[0].data(), in.shared, in.scal[0].data());
    const int n_roots = (int)roots.size(), n_map = 3;
    std::vector<V> from(in.dev[0]), to, out((size_t)n_roots * in.copies);
    std::vector<double> sc;
    int n_scal = 0;
    OK(fmhip_graph_scalars(roots.data(), n_roots, nullptr, 0, &n_scal));
    std::vector<double> recorded((size_t)n_scal);
    OK(fmhip_graph_scalars(roots.data(), n_roots, recorded.data(), n_scal, &n_scal));
    for (int j = 1; j <= in.copies; ++j) { for (V v : in.dev[(size_t)j]) to.push_back(v); for (int k = 0; k < n_scal; ++k) sc.push_back(recorded[(size_t)k] * (1.0 + 0.01 * j)); }
    OK(fmhip_graph_clone(roots.data(), n_roots, in.copies, from.data(), to.data(), n_map, sc.data(), n_scal, out.data()));
    std::vector<V> extra;
    switch (which) {
    case 0: break;                                                                  // flushed as recorded: the copies run as rows of the original's launches
    case 1: extra.push_back(s1(FMHIP_OP_ADD_S, roots[0], 1.0)); break;                // an operation on top of the original
    case 2: extra.push_back(s1(FMHIP_OP_MULT_S, out[0], 2.0)); break;                 // a copy used before the flush
    case 3: { std::vector<V> again((size_t)n_roots * 2); std::vector<V> to2;         // a copy of a copy
              for (int j = 0; j < 2; ++j) for (V v : in.dev[(size_t)j + 1]) to2.push_back(v);
              OK(fmhip_graph_clone(out.data(), n_roots, 2, in.dev[1].data(), to2.data(), n_map, nullptr, 0, again.data()));
              for (V v : again) extra.push_back(v); break; }
    case 4: rel(roots); break;                                                        // the original's handles released before the flush
    case 5: for (V& v : out) { rel(v); v = 0; } break;                                // all copies released before the flush
    case 6: { std::vector<float> h(1031); OK(fmhip_vec_read_float(out.back(), h.data(), 1031)); break; }   // one copy read (materialised on its own)
    case 7: rel(out[1]); out[1] = 0; break;                                           // one copy released
    case 8: { fmhip_moments m; OK(fmhip_reduce_moments(out[0], 0.0, &m)); break; }   // the expectation of a copy asked for under the hold
    case 9: OK(fmhip_vec_retain(roots[0])); extra.push_back(roots[0]); break;         // a second handle on an original root
    }
    OK(fmhip_fusion_hold(0, nullptr));
    OK(fmhip_flush());
    read_all(roots, in.n);
    for (V v : out) if (v) { std::vector<float> h((size_t)in.n); OK(fmhip_vec_read_float(v, h.data(), in.n)); }
    read_all(extra, in.n);
    rel(roots); for (V v : out) if (v) rel(v);
    rel(extra);
    free_inputs(in);
    OK(fmhip_set_jit(prev, nullptr));
}

static void scenario_basic() {
    const int64_t n = 777;
    V a = filled(n, 1.5), b = filled(n, 0.5);
    std::vector<double> host((size_t)n, 0.25);
    V c = 0; OK(fmhip_vec_create_from_double(host.data(), n, &c));
    for (int fusion = 0; fusion < 2; ++fusion) {
        OK(fmhip_set_fusion(fusion, nullptr));
        V t = b2(FMHIP_OP_ADD, a, b), u = s1(FMHIP_OP_DIV_S, t, 2.0), w = u1(FMHIP_OP_EXP, u), x = s2(FMHIP_OP_ACCRUE, w, c, 0.5);
        fmhip_moments m; OK(fmhip_reduce_moments(x, 0.0, &m)); OK(fmhip_reduce_moments(x, 0.5, &m));
        std::vector<double> out((size_t)n); OK(fmhip_vec_read_double(x, out.data(), n));
        V both[2] = { x, w }; fmhip_moments mm[2]; OK(fmhip_reduce_moments_batch(both, 2, nullptr, mm));
        rel(t); rel(u); rel(w); rel(x);
    }
    EXPECT(fmhip_vec_release(123456789), FMHIP_ERR_INVALID_HANDLE);
    V bad = 0; EXPECT(fmhip_call_v2s0(FMHIP_OP_ADD, a, filled(3, 1.0), &bad), FMHIP_ERR_SIZE_MISMATCH);
    rel(a); rel(b); rel(c);
    OK(fmhip_flush()); OK(fmhip_pool_clean()); OK(fmhip_pool_purge());
}

static void scenario_replicas() {
    OK(fmhip_set_fusion(1, nullptr));
    for (int jit = 0; jit < 2; ++jit)
        for (int which = 0; which < 10; ++which) {
            replica_case([](const V* v, V sh, const double* s) { return record(v, sh, s); }, which, jit != 0);
            replica_case([](const V* v, V sh, const double* s) { return long_chain(v, sh, s); }, which, jit != 0);
            replica_case([](const V* v, V sh, const double* s) { return long_chain(v, sh, s); }, which, jit != 0);     // the shape's plan exists now
        }
    OK(fmhip_pool_purge());
}

// expectations: taken along by the flush, values given up, tickets out of order, the arena wrapping (FMHIP_ARENA_BYTES is tiny here)
static void scenario_expectations() {
    OK(fmhip_set_fusion(1, nullptr));
    int prev = 0; OK(fmhip_set_jit(FMHIP_JIT_SYNC, &prev));
    Inputs in = make_inputs(2049, 2);
    std::vector<fmhip_ticket> tickets;
    std::vector<int> counts;
    for (int round = 0; round < 12; ++round) {
        OK(fmhip_fusion_hold(2, nullptr));
        std::vector<V> payoffs;
        for (int k = 0; k < 9; ++k) {
            std::vector<double> s = { 0.02 + 0.001 * k, 0.5, 0.5, 0.0 };
            std::vector<V> r = (k % 3 == 2) ? record(in.dev[(size_t)k % 3].data(), in.shared, s.data()) : long_chain(in.dev[(size_t)k % 3].data(), in.shared, s.data(), k % 3 == 0 ? 24 : 3);
            for (V v : r) payoffs.push_back(v);
        }
        OK(fmhip_fusion_hold(0, nullptr));
        if (round % 2 == 1) OK(fmhip_vec_give_up_values(payoffs.data(), (int)payoffs.size()));
        if (round % 4 == 3) { V keep = s1(FMHIP_OP_ADD_S, payoffs[1], 1.0); rel(keep); }       // a consumer of a given-up value that goes away again
        int shards = 1; OK(fmhip_device_count(&shards));
        if (round % 3 == 0 && shards == 1) {        // the same moments into a device buffer (a caller with its own exchange): collected from the arena by a kernel
            V send = 0; OK(fmhip_vec_create_uninitialized((int64_t)payoffs.size() * 8, &send));
            void* dev = nullptr; OK(fmhip_vec_device_ptr(send, &dev));
            OK(fmhip_reduce_moments_batch_device(payoffs.data(), (int)payoffs.size(), nullptr, dev));
            rel(send);
        }
        fmhip_ticket t = 0;
        OK(fmhip_reduce_moments_batch_begin(payoffs.data(), (int)payoffs.size(), nullptr, &t));
        tickets.push_back(t); counts.push_back((int)payoffs.size());
        if (round % 2 == 1) {                       // given up: reading is an error where nothing was kept, the moments stay
            std::vector<float> h(2049);
            const int st = fmhip_vec_read_float(payoffs[0], h.data(), 2049);
            if (st != FMHIP_OK && st != FMHIP_ERR_INVALID_ARGUMENT) std::abort();
            fmhip_moments m; OK(fmhip_reduce_moments(payoffs[0], 0.0, &m));
        }
        rel(payoffs);
        if (tickets.size() == 3) {                  // ended out of order
            for (int i : { 1, 0, 2 }) { std::vector<fmhip_moments> m((size_t)counts[(size_t)i]); OK(fmhip_reduce_moments_batch_end(tickets[(size_t)i], m.data(), counts[(size_t)i])); }
            fmhip_moments one; EXPECT(fmhip_reduce_moments_batch_end(tickets[0], &one, 1), FMHIP_ERR_INVALID_HANDLE);
            tickets.clear(); counts.clear();
        }
    }
    std::vector<double> shifts = { 0.5, 0.25 };
    V pair[2] = { in.dev[0][0], in.dev[0][1] };
    fmhip_ticket t = 0; OK(fmhip_reduce_moments_batch_begin(pair, 2, shifts.data(), &t));
    fmhip_moments m2[2]; EXPECT(fmhip_reduce_moments_batch_end(t, m2, 3), FMHIP_ERR_SIZE_MISMATCH);
    free_inputs(in);
    OK(fmhip_set_jit(prev, nullptr));
}

// explicit programs, the row-table ring wrapping (FMHIP_RING_BYTES is tiny here), Brownian increments and the engine's own time-step grouping
static void scenario_programs_and_steps() {
    const int64_t n = 1500;
    fmhip_prog_op ops[3] = { { FMHIP_OP_ADD_S, 0, -1, -1, 4.0 }, { FMHIP_OP_MULT, 3, 1, -1, 0.0 }, { FMHIP_OP_CHOOSE, 4, 2, 0, 0.0 } };
    const int32_t outs[1] = { 5 }, reds[1] = { 5 };
    fmhip_program p = 0; OK(fmhip_program_create(ops, 3, 3, outs, 1, reds, 1, &p));
    const int batch = 40;
    std::vector<V> in, out((size_t)batch);
    for (int b = 0; b < batch; ++b) for (int k = 0; k < 3; ++k) in.push_back(filled(n, 0.5 + k));
    std::vector<fmhip_moments> m((size_t)batch);
    for (int rep = 0; rep < 30; ++rep) { OK(fmhip_program_run(p, batch, in.data(), out.data(), nullptr, m.data(), nullptr)); rel(out); out.assign((size_t)batch, 0); }
    for (int b = 0; b < batch; ++b) out[(size_t)b] = filled(n, 0.0);
    for (int rep = 0; rep < 30; ++rep) OK(fmhip_program_run_into(p, batch, in.data(), out.data(), nullptr, nullptr, nullptr));
    OK(fmhip_program_release(p));
    rel(in); rel(out);
    // an Euler scheme through methods only: the first use of an increment with a new time index marks a step, four steps run together
    OK(fmhip_set_fusion(1, nullptr));
    const int steps = 24;
    std::vector<double> dt((size_t)steps, 0.1);
    for (int sim = 0; sim < 3; ++sim) {
        std::vector<V> inc((size_t)steps * 2), inc2((size_t)steps);
        OK(fmhip_bm_generate(31415 + sim, steps, 2, n, 0, dt.data(), inc.data()));
        OK(fmhip_bm_generate(27182 + sim, steps, 1, n, 0, dt.data(), inc2.data()));
        V x = filled(n, 0.0), v = filled(n, 0.09);
        for (int i = 0; i < steps; ++i) {
            const V vp = s1(FMHIP_OP_FLOOR_S, v, 0.0), sq = u1(FMHIP_OP_SQRT, vp), d = b2(FMHIP_OP_MULT, sq, inc[(size_t)i * 2]), x2 = b2(FMHIP_OP_ADD, x, d);
            const V e = s2(FMHIP_OP_ADDPRODUCT_VS, v, inc[(size_t)i * 2 + 1], 0.3), v2 = s2(FMHIP_OP_ADDPRODUCT_VS, e, inc2[(size_t)i], 0.1);      // (a second generation inside the step)
            rel(vp); rel(sq); rel(d); rel(e); rel(x); rel(v);
            x = x2; v = v2;
            if (sim == 2 && i == 13) { fmhip_moments mm; OK(fmhip_reduce_moments(x, 0.0, &mm)); }    // a value read in the middle of a group
        }
        fmhip_moments mm; OK(fmhip_reduce_moments(x, 0.0, &mm));
        rel(x); rel(v); rel(inc); rel(inc2);
    }
}

static void scenario_threads() {
    OK(fmhip_set_fusion(1, nullptr));
    std::atomic<int> failures{ 0 };
    std::vector<std::thread> ts;
    for (int t = 0; t < 8; ++t)
        ts.emplace_back([t, &failures] {
            const int64_t n = 600 + t;
            for (int rep = 0; rep < 25; ++rep) {
                V a = 0, b = 0;
                if (fmhip_vec_create_filled(n, 1.0 + t, &a) != FMHIP_OK || fmhip_vec_create_filled(n, 0.5, &b) != FMHIP_OK) { ++failures; return; }
                V cur = a;
                for (int k = 0; k < 12 + (rep % 5) * 10; ++k) {
                    V nx = 0;
                    if (fmhip_call_v2s1(FMHIP_OP_DISCOUNT, cur, b, 0.5, &nx) != FMHIP_OK) { ++failures; return; }
                    if (cur != a) fmhip_vec_release(cur);
                    cur = nx;
                }
                fmhip_moments m;
                if (fmhip_reduce_moments(cur, 0.0, &m) != FMHIP_OK) ++failures;
                if (rep % 3 == 0) { fmhip_ticket tk = 0; V one[1] = { cur }; fmhip_moments mm[1];
                                    if (fmhip_reduce_moments_batch_begin(one, 1, nullptr, &tk) != FMHIP_OK || fmhip_reduce_moments_batch_end(tk, mm, 1) != FMHIP_OK) ++failures; }
                if (cur != a) fmhip_vec_release(cur);
                fmhip_vec_release(a); fmhip_vec_release(b);
            }
        });
    for (auto& th : ts) th.join();
    if (failures.load() != 0) { std::fprintf(stderr, "threads: %d failed calls (%s)\n", failures.load(), fmhip_last_error()); std::abort(); }
}

// Vectors, a pending expression, a program and tickets that cross threads (with FMNULL_THREAD_ENGINES=1: engines — a foreign operand is
// imported, everything else runs on the owner's engine; without: the one engine's lock)
static void scenario_shared() {
    OK(fmhip_set_fusion(1, nullptr));
    const int64_t n = 777;
    V x = filled(n, 1.25), y = filled(n, 0.5);
    V pending = s1(FMHIP_OP_MULT_S, x, 1.5);                       // not computed when the threads start
    const fmhip_prog_op ops[] = { { FMHIP_OP_MULT, 0, 1, -1, 0.0 }, { FMHIP_OP_ADD_S, 2, -1, -1, 1.0 } };
    const int32_t outv = 3;
    fmhip_program prog = 0; OK(fmhip_program_create(ops, 2, 2, &outv, 1, &outv, 1, &prog));
    std::vector<V> handed(6, 0);
    std::vector<fmhip_ticket> tickets(6, 0);
    std::atomic<int> failures{ 0 };
    std::vector<std::thread> ts;
    for (int t = 0; t < 6; ++t)
        ts.emplace_back([&, t] {
            V cur = 0;
            if (fmhip_call_v2s0(FMHIP_OP_ADD, pending, y, &cur) != FMHIP_OK) { ++failures; return; }          // a foreign, pending operand
            for (int k = 0; k < 20; ++k) { V nx = 0; if (fmhip_call_v2s1(FMHIP_OP_DISCOUNT, cur, x, 0.25, &nx) != FMHIP_OK) { ++failures; return; } fmhip_vec_release(cur); cur = nx; }
            fmhip_moments m;
            if (fmhip_reduce_moments(cur, 0.0, &m) != FMHIP_OK) ++failures;
            if (fmhip_reduce_moments(x, 0.0, &m) != FMHIP_OK) ++failures;                                       // the moments of another thread's vector
            V in[2] = { cur, y }, out[1] = { 0 }; fmhip_moments pm[1];
            if (fmhip_program_run(prog, 1, in, out, nullptr, pm, nullptr) != FMHIP_OK) ++failures;              // another thread's program, mixed operands
            else fmhip_vec_release(out[0]);                                                                     // … its output belongs to the program's engine
            V both[2] = { cur, pending };
            if (fmhip_reduce_moments_batch_begin(both, 2, nullptr, &tickets[(size_t)t]) != FMHIP_OK) ++failures; // ended by the main thread
            if (fmhip_vec_retain(y) != FMHIP_OK || fmhip_vec_release(y) != FMHIP_OK) ++failures;
            int64_t size = 0; if (fmhip_vec_size(pending, &size) != FMHIP_OK || size != n) ++failures;
            handed[(size_t)t] = cur;                                                                            // read and released by the main thread
        });
    for (auto& th : ts) th.join();
    if (failures.load() != 0) { std::fprintf(stderr, "shared: %d failed calls (%s)\n", failures.load(), fmhip_last_error()); std::abort(); }
    for (int t = 0; t < 6; ++t) {
        fmhip_moments mm[2]; OK(fmhip_reduce_moments_batch_end(tickets[(size_t)t], mm, 2));
        std::vector<float> h((size_t)n); OK(fmhip_vec_read_float(handed[(size_t)t], h.data(), n));
        rel(handed[(size_t)t]);
    }
    OK(fmhip_synchronize());
    fmhip_pool_stats_t st; OK(fmhip_pool_stats(&st));
    OK(fmhip_program_release(prog));
    rel(pending); rel(x); rel(y);
}

// FMHIP_TEST_FAIL_ALLOC_AT is set by the caller for this scenario: an allocation fails somewhere inside a replicated launch of 1100 members
static void scenario_failure() {
    OK(fmhip_set_fusion(1, nullptr));
    OK(fmhip_set_jit(FMHIP_JIT_OFF, nullptr));
    const int64_t n = 64; const int copies = 1100;
    std::vector<V> xs; for (int j = 0; j <= copies; ++j) xs.push_back(filled(n, 1.0 + 0.001 * j));
    V y = filled(n, 0.5);
    OK(fmhip_fusion_hold(1, nullptr));
    V t = s2(FMHIP_OP_ACCRUE, xs[0], y, 0.5), u = s1(FMHIP_OP_ADD_S, t, 1.0), w = b2(FMHIP_OP_MULT, u, y);
    rel(u);
    V roots[2] = { t, w };
    std::vector<V> out((size_t)copies * 2);
    OK(fmhip_graph_clone(roots, 2, copies, &xs[0], &xs[1], 1, nullptr, 0, out.data()));
    OK(fmhip_fusion_hold(0, nullptr));
    const int st = fmhip_flush();
    if (st != FMHIP_OK && st != FMHIP_ERR_OUT_OF_MEMORY) std::abort();
    int lost = 0;
    std::vector<float> h((size_t)n);
    for (V v : out) { const int r = fmhip_vec_read_float(v, h.data(), n); if (r != FMHIP_OK && r != FMHIP_ERR_INVALID_ARGUMENT && r != FMHIP_ERR_OUT_OF_MEMORY) std::abort(); lost += r == FMHIP_ERR_INVALID_ARGUMENT; }
    std::printf("failure: flush status %d, %d copies lost\n", st, lost);
    for (V v : roots) { const int r = fmhip_vec_read_float(v, h.data(), n); if (r != FMHIP_OK && r != FMHIP_ERR_OUT_OF_MEMORY) std::abort(); }
    rel(out); rel(t); rel(w); rel(xs); rel(y);
    OK(fmhip_flush());
}

// cost of a recorded method on the caller's side (and of its replay on the shards): not a sanitizer scenario — build with -O2, no sanitizer
static void scenario_speed() {
    OK(fmhip_set_fusion(1, nullptr));
    OK(fmhip_fusion_hold(1, nullptr));
    const int64_t n = 1024;
    V a = filled(n, 1.0), b = filled(n, 0.5);
    const int reps = 2000000;
    const auto t0 = std::chrono::steady_clock::now();
    V cur = a;
    for (int k = 0; k < reps; ++k) {
        V nx = 0;
        OK(fmhip_call_v2s1(FMHIP_OP_DISCOUNT, cur, b, 0.5, &nx));
        if (cur != a) OK(fmhip_vec_release(cur));
        cur = nx;
        if (k % 200 == 199) { rel(cur); cur = a; }         // chains of 200: released unexecuted (under the hold nothing runs)
    }
    const auto t1 = std::chrono::steady_clock::now();
    OK(fmhip_synchronize());
    const auto t2 = std::chrono::steady_clock::now();
    std::printf("speed: %.0f ns per recorded method + release on the caller's thread, %.0f ns including the shards' replay\n",
                std::chrono::duration<double>(t1 - t0).count() / reps * 1e9, std::chrono::duration<double>(t2 - t0).count() / reps * 1e9);
    if (cur != a) rel(cur);
    // … and the FRONT's own cost (a device list: what the caller's thread pays to hand a method and a release to the shards): bursts that
    // fit the command ring, so that the caller never waits for a slot; the shards' replay (waited for between the bursts) is not timed
    {
        const int burst = 12000, bursts = 40;
        double seconds = 0.0;
        for (int r = 0; r < bursts; ++r) {
            OK(fmhip_synchronize());
            const auto b0 = std::chrono::steady_clock::now();
            V c2 = a;
            for (int k = 0; k < burst; ++k) {
                V nx = 0;
                OK(fmhip_call_v2s1(FMHIP_OP_DISCOUNT, c2, b, 0.5, &nx));
                if (c2 != a) OK(fmhip_vec_release(c2));
                c2 = nx;
                if (k % 200 == 199) { rel(c2); c2 = a; }
            }
            seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - b0).count();
            if (c2 != a) rel(c2);
        }
        std::printf("speed: %.0f ns per recorded method + release in bursts of %d (the caller never waits for the command ring)\n", seconds / (burst * bursts) * 1e9, burst);
    }
    OK(fmhip_fusion_hold(0, nullptr));
    rel(a); rel(b);
}

// A caller whose handles die when a garbage collector says so (the Java binding: a Cleaner action per handle; the reference:
// RandomVariableCuda.java:293-305): every temporary keeps its handle until a collector THREAD releases it, in bursts, a few milliseconds
// late.  The engine leaves such values unstored once it has learnt that nobody comes for them (Node::deferred, runtime.hpp: escape policy) —
// and computes them from their recipes when somebody does: read, reduced, an operand of a later method, the vector they read overwritten
// in place, fmhip_pool_clean.  Long chains (segments), a little Euler scheme over Brownian increments (time-step grouping, rolled loops).
static void scenario_lagging() {
    const int64_t n = 2049;
    int was_fusion = 0;
    OK(fmhip_set_fusion(1, &was_fusion));
    std::mutex mu;
    std::vector<V> dead;
    std::atomic<bool> quit{ false };
    std::thread collector([&] {
        while (!quit.load()) {
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            std::vector<V> batch;
            { std::lock_guard<std::mutex> lock(mu); batch.swap(dead); }
            for (V h : batch) rel(h);
        }
    });
    auto die = [&](V h) { std::lock_guard<std::mutex> lock(mu); dead.push_back(h); };
    std::vector<float> host((size_t)n);
    V x = filled(n, 0.5), y = filled(n, 1.5), z = filled(n, 0.75);
    for (int round = 0; round < 12; ++round) {
        std::vector<V> temps;
        V cur = s1(FMHIP_OP_ADD_S, x, 1.0 + round);
        for (int k = 0; k < 60; ++k) {
            const V nx = k % 3 == 0 ? b2(FMHIP_OP_MULT, cur, y) : k % 3 == 1 ? s2(FMHIP_OP_ADDPRODUCT_VS, cur, z, 0.5) : u1(FMHIP_OP_ABS, cur);
            temps.push_back(cur);
            cur = nx;
        }
        OK(fmhip_flush());
        fmhip_moments m; OK(fmhip_reduce_moments(cur, 0.0, &m));
        if (round % 4 == 3) {                                        // values left unstored are used after all
            OK(fmhip_vec_read_float(temps[17], host.data(), n));
            const V late = b2(FMHIP_OP_ADD, temps[5], temps[40]);
            OK(fmhip_vec_read_float(late, host.data(), n));
            OK(fmhip_reduce_moments(temps[33], 0.0, &m));
            die(late);
        }
        if (round == 7 && !std::getenv("FMNULL_DEVICES")) { void* ptr = nullptr; OK(fmhip_vec_device_ptr(x, &ptr)); }     // may be written in place: whoever reads x is computed first
        if (round == 9) OK(fmhip_pool_clean());
        for (V t : temps) die(t);
        die(cur);
    }
    // an Euler scheme: 6 components, 16 time steps, every state and every temporary keeps its handle; expectations of the final state
    const int steps = 16, comps = 6;
    std::vector<double> dt((size_t)steps, 0.25);
    for (int evaluation = 0; evaluation < 5; ++evaluation) {
        std::vector<V> dW((size_t)steps);
        OK(fmhip_bm_generate(1234 + evaluation, steps, 1, n, 0, dt.data(), dW.data()));
        std::vector<V> state((size_t)comps), all;
        for (int j = 0; j < comps; ++j) { state[(size_t)j] = s1(FMHIP_OP_MULT_S, x, 0.1 * (j + 1)); }
        for (int i = 0; i < steps; ++i) {
            V sum = 0;
            for (int j = 0; j < comps; ++j) {
                const V t1 = s1(FMHIP_OP_MULT_S, state[(size_t)j], 0.5), t2 = s1(FMHIP_OP_ADD_S, t1, 1.0), tr = s1(FMHIP_OP_VID_S, t2, 0.02);
                const V ns = sum ? b2(FMHIP_OP_ADD, sum, tr) : tr;
                const V drift = s1(FMHIP_OP_MULT_S, ns, 0.3), a = s2(FMHIP_OP_ADDPRODUCT_VS, state[(size_t)j], drift, 0.25), nx = s2(FMHIP_OP_ADDPRODUCT_VS, a, dW[(size_t)i], 0.3);
                all.push_back(t1); all.push_back(t2); if (sum) all.push_back(tr); if (sum) all.push_back(sum);
                all.push_back(drift); all.push_back(a); all.push_back(state[(size_t)j]);
                sum = ns; state[(size_t)j] = nx;
            }
            all.push_back(sum);
        }
        for (int j = 0; j < comps; ++j) { fmhip_moments m; OK(fmhip_reduce_moments(state[(size_t)j], 0.0, &m)); }
        if (evaluation == 3) { OK(fmhip_vec_read_float(all[all.size() / 2], host.data(), n)); OK(fmhip_vec_read_float(all[7], host.data(), n)); }     // states of the middle of the simulation, wanted after all
        for (V h : all) die(h);
        for (V h : state) die(h);
        for (V h : dW) die(h);
    }
    quit.store(true);
    collector.join();
    rel(dead);
    rel(x); rel(y); rel(z);
    fmhip_engine_stats_t es; OK(fmhip_engine_stats(&es));
    if (es.values_deferred <= 0 || es.values_demanded <= 0 || es.values_deferred_now != 0) { std::fprintf(stderr, "lagging: deferred %lld, demanded %lld, deferred now %lld\n", (long long)es.values_deferred, (long long)es.values_demanded, (long long)es.values_deferred_now); std::abort(); }
    OK(fmhip_set_fusion(was_fusion, nullptr));
}

// A device that is FULL (FMNULL_DEVICE_BYTES: the null device refuses allocations beyond it): the pool's last resort — drop every cached
// slab, try once more (RandomVariableCuda.java:340) — on the first allocation of a size class it has never seen, with and without success,
// then the error a garbage-collected caller answers with a collection.  Round 5 found heap corruption here on a real device (a reference
// into the pool's size-class table held across the purge that erases its entry).
static void scenario_oom() {
    if (!std::getenv("FMNULL_DEVICE_BYTES")) { std::printf("oom: FMNULL_DEVICE_BYTES not set, skipped\n"); return; }
    const int64_t n = 1 << 20;                                   // 4 MiB vectors
    std::vector<V> small;
    for (;;) { V h = 0; const int st = fmhip_vec_create_filled(n, 1.0, &h); if (st == FMHIP_ERR_OUT_OF_MEMORY) break; OK(st); small.push_back(h); if (small.size() > 100000) std::abort(); }
    if (small.size() < 8) std::abort();
    rel(small);                                                   // all cached in the pool now: the device is full of free blocks
    V big = 0;
    OK(fmhip_vec_create_filled(5 * n + 123, 2.0, &big));          // a new size class: hipMalloc fails, the purge makes room
    V big2 = 0;
    OK(fmhip_vec_create_filled(3 * n + 77, 2.0, &big2));          // … and another one
    std::vector<V> more;
    for (;;) { V h = 0; const int st = fmhip_vec_create_filled(n, 1.0, &h); if (st == FMHIP_ERR_OUT_OF_MEMORY) break; OK(st); more.push_back(h); if (more.size() > 100000) std::abort(); }
    // every second vector released: cached blocks, but no slab is free as a whole — a small allocation is carved out of a cached block
    std::vector<V> odd;
    for (size_t i = 0; i < more.size(); ++i) if (i % 2) rel(more[i]); else odd.push_back(more[i]);
    more.swap(odd);
    std::vector<V> tiny;
    for (int i = 0; i < 300; ++i) tiny.push_back(filled(1000 + 7 * i, 0.5));            // 300 size classes, none seen before
    fmhip_moments mm; OK(fmhip_reduce_moments(tiny[5], 0.0, &mm));
    rel(tiny);
    for (;;) { V h = 0; const int st = fmhip_vec_create_filled(n, 1.0, &h); if (st == FMHIP_ERR_OUT_OF_MEMORY) break; OK(st); more.push_back(h); if (more.size() > 200000) std::abort(); }
    V huge = 0;
    EXPECT(fmhip_vec_create_filled(7 * n + 5, 2.0, &huge), FMHIP_ERR_OUT_OF_MEMORY);      // nothing cached to drop: the error, and an engine that carries on
    int was = 0; OK(fmhip_set_fusion(1, &was));
    const V t = s1(FMHIP_OP_ADD_S, big, 1.0), u = b2(FMHIP_OP_MULT, t, big);
    EXPECT(fmhip_flush(), FMHIP_ERR_OUT_OF_MEMORY);               // a fused launch that cannot get its output
    rel(more);                                                    // the caller's collection
    OK(fmhip_flush());
    std::vector<float> host((size_t)(5 * n + 123));
    OK(fmhip_vec_read_float(u, host.data(), 5 * n + 123));
    rel(t); rel(u); rel(big); rel(big2);
    OK(fmhip_set_fusion(was, nullptr));
}

// Expectations wanted ON the devices of a device list (fmhip_reduce_moments_batch_devices; the *_device variants: the first listed device):
// the one exchange between devices — a grouped all-gather over the listed devices and a combine kernel per device when the devices are
// distinct (FMNULL_DISTINCT=1: the list {0, 1, 2 …}; RCCL = the stand-ins of null_rccl.cpp), a host combine when an index repeats.  Every
// device must end up with the bits fmhip_reduce_moments_batch returns on the host.
static void scenario_collective() {
    int shards = 1, kind = -1;
    OK(fmhip_device_count(&shards));
    char why[256];
    OK(fmhip_expectation_collective(&kind, why, sizeof why));
    const bool distinct = std::getenv("FMNULL_DISTINCT") != nullptr;
    if (kind != (shards == 1 ? 0 : distinct ? 1 : 2)) { std::fprintf(stderr, "collective: kind %d with %d shards (%s)\n", kind, shards, why); std::abort(); }
    int was = 0; OK(fmhip_set_fusion(1, &was));
    const int64_t n = 40003;
    V x = filled(n, 0.5), y = filled(n, 1.5);
    for (int round = 0; round < 3; ++round) {
        const int count = 5 + 60 * round;
        std::vector<V> vs;
        for (int k = 0; k < count; ++k) { const V t = s1(FMHIP_OP_MULT_S, x, 1.0 + k), u = b2(FMHIP_OP_ADD, t, y); rel(t); vs.push_back(u); }
        std::vector<std::vector<double>> dev((size_t)shards, std::vector<double>((size_t)count * 4, -1.0));       // ("device" memory of the null device is host memory)
        std::vector<void*> out;
        for (int d = 0; d < shards; ++d) out.push_back(d == 1 && shards > 2 ? nullptr : dev[(size_t)d].data());     // (one device that does not want them)
        OK(fmhip_reduce_moments_batch_devices(vs.data(), count, nullptr, out.data(), shards));
        OK(fmhip_synchronize());
        std::vector<fmhip_moments> host((size_t)count);
        OK(fmhip_reduce_moments_batch(vs.data(), count, nullptr, host.data()));
        for (int d = 0; d < shards; ++d) {
            if (!out[(size_t)d]) continue;
            if (std::memcmp(dev[(size_t)d].data(), host.data(), (size_t)count * 32) != 0) { std::fprintf(stderr, "collective: device %d of %d disagrees with the host-side moments (round %d)\n", d, shards, round); std::abort(); }
        }
        std::vector<double> first((size_t)count * 4, -1.0);
        OK(fmhip_reduce_moments_batch_device(vs.data(), count, nullptr, first.data()));                          // the first listed device
        OK(fmhip_synchronize());
        if (std::memcmp(first.data(), host.data(), (size_t)count * 32) != 0) std::abort();
        double one[4] = { -1, -1, -1, -1 };
        OK(fmhip_reduce_moments_device(vs[0], 0.0, one));
        OK(fmhip_synchronize());
        if (std::memcmp(one, &host[0], 32) != 0) std::abort();
        void* stream = nullptr;
        OK(fmhip_get_stream_of(shards - 1, &stream));
        EXPECT(fmhip_get_stream_of(shards, &stream), FMHIP_ERR_INVALID_ARGUMENT);
        rel(vs);
    }
    rel(x); rel(y);
    OK(fmhip_set_fusion(was, nullptr));
}

int main(int argc, char** argv) {
    struct Scenario { const char* name; void (*run)(); };
    const Scenario all[] = { { "basic", scenario_basic }, { "replicas", scenario_replicas }, { "expectations", scenario_expectations },
                             { "programs", scenario_programs_and_steps }, { "threads", scenario_threads }, { "shared", scenario_shared }, { "failure", scenario_failure }, { "speed", scenario_speed }, { "lagging", scenario_lagging }, { "oom", scenario_oom }, { "collective", scenario_collective } };
    std::vector<std::string> wanted;
    for (int i = 1; i < argc; ++i) wanted.push_back(argv[i]);
    const bool only_failure = wanted.size() == 1 && wanted[0] == "failure";      // (the hook counts the allocations of the whole process: one cycle)
    for (int cycle = 0; cycle < (only_failure ? 1 : 2); ++cycle) {                  // twice: shutdown and re-initialisation in between
        // FMNULL_DEVICES=N: the same scenarios behind a device list of N shards (sharded.cpp: one engine and one worker thread per shard)
        const int n_devices = std::getenv("FMNULL_DEVICES") ? std::atoi(std::getenv("FMNULL_DEVICES")) : 1;
        // (FMHIP_WORKER_THREAD=1 with one device: the front with ONE shard)
        if (n_devices > 1 || (n_devices == 1 && std::getenv("FMHIP_WORKER_THREAD"))) { std::vector<int> devices((size_t)n_devices, 0);
            // (FMNULL_DISTINCT=1: distinct indices — the front asks RCCL, i.e. null_rccl.cpp, for a communicator per device)
            if (std::getenv("FMNULL_DISTINCT")) for (int k = 0; k < n_devices; ++k) devices[(size_t)k] = k;
            OK(fmhip_init_devices(devices.data(), n_devices)); int c = 0; OK(fmhip_device_count(&c)); if (c != n_devices) std::abort(); }
        else OK(fmhip_init(0));
        // FMNULL_THREAD_ENGINES=1: an engine per caller thread (fmhip_set_thread_engines) — the scenarios' threads record side by side
        if (n_devices <= 1 && std::getenv("FMNULL_THREAD_ENGINES")) { int was = -1; OK(fmhip_set_thread_engines(1, &was)); if (was != 0) std::abort(); }
        for (const Scenario& s : all) {
            bool run = wanted.empty() ? (std::strcmp(s.name, "failure") != 0 && std::strcmp(s.name, "speed") != 0 && std::strcmp(s.name, "oom") != 0) : false;
            for (const std::string& w : wanted) run |= w == s.name;
            if (!run) continue;
            s.run();
            std::printf("cycle %d: %s done\n", cycle, s.name);
            std::fflush(stdout);
        }
        OK(fmhip_shutdown());
    }
    return 0;
}
