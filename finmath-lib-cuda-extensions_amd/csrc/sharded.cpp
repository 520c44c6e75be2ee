// sharded.cpp — see sharded.hpp: one engine and one worker thread per device of a device list, the caller's C-ABI calls replayed on all.
#include "sharded.hpp"
#include "runtime.hpp"
#include "kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace fm {

void set_last_error(const std::string& message);              // abi.cpp: the calling thread's fmhip_last_error()

// Contiguous block of shard `shard`: boundaries at multiples of four paths (one Philox call covers four consecutive paths) except the
// very end, sizes differing by at most four — the rule of parallel.py: path_shard, which the process-per-GPU path uses.
void shard_range(int64_t n, int shards, int shard, int64_t* offset, int64_t* count) {
    const int64_t groups = (n + 3) / 4, base = groups / shards, extra = groups % shards;
    const int64_t g0 = shard * base + std::min<int64_t>(shard, extra), g1 = g0 + base + (shard < extra ? 1 : 0);
    const int64_t off = std::min(g0 * 4, n), end = std::min(g1 * 4, n);
    *offset = off; *count = end - off;
}

namespace {

struct Worker;
typedef std::function<void(Worker&)> Command;

// Front handle → entry.  Handles are consecutive numbers, most of them short-lived (a calibration hands out tens of millions): pages of
// 4096 entries, allocated when the first handle of a page is issued and freed when its last one is gone (as the engine's own HandleTable).
template <typename T>
class PagedTable {
public:
    T* get(int64_t id) {
        if (id <= 0) return nullptr;
        const size_t page = (size_t)(id >> BITS);
        if (page >= pages_.size() || !pages_[page].slot || !pages_[page].used[(size_t)id & MASK]) return nullptr;
        return &pages_[page].slot[(size_t)id & MASK];
    }
    T& put(int64_t id, const T& value) {
        const size_t page = (size_t)(id >> BITS);
        if (page >= pages_.size()) pages_.resize(page + 1);
        Page& pg = pages_[page];
        if (!pg.slot) { pg.slot.reset(new T[(size_t)1 << BITS]()); pg.used.reset(new char[(size_t)1 << BITS]()); pg.live = 0; }
        if (!pg.used[(size_t)id & MASK]) { pg.used[(size_t)id & MASK] = 1; ++pg.live; ++size_; }
        return pg.slot[(size_t)id & MASK] = value;
    }
    void erase(int64_t id) {
        if (id <= 0) return;
        const size_t page = (size_t)(id >> BITS);
        if (page >= pages_.size() || !pages_[page].slot || !pages_[page].used[(size_t)id & MASK]) return;
        Page& pg = pages_[page];
        pg.used[(size_t)id & MASK] = 0; --size_;
        if (--pg.live == 0) { pg.slot.reset(); pg.used.reset(); }
    }
    size_t size() const { return size_; }
private:
    static constexpr int BITS = 12;
    static constexpr size_t MASK = ((size_t)1 << BITS) - 1;
    struct Page { std::unique_ptr<T[]> slot; std::unique_ptr<char[]> used; uint32_t live = 0; };
    std::vector<Page> pages_;
    size_t size_ = 0;
};

// A queued call: plain data.  The calls a Monte-Carlo caller makes by the million — a method, a release — carry their operands; everything
// else names a closure kept beside the ring (Shards::fns).  `target`: the shard the command is for, or -1 for all of them.
struct Cmd {
    enum Kind : uint8_t { FN, CALL, RELEASE, RETAIN } kind = FN;
    bool has_scalar = false, last = false;
    int8_t target = -1;
    int32_t opcode = 0, n_in = 0;
    fmhip_vec a0 = 0, a1 = 0, a2 = 0, id = 0;
    double scalar = 0.0;
};

struct Shards;

struct Worker {
    int shard = 0, shards = 1, device = 0;
    Engine* engine = nullptr;
    Shards* front = nullptr;
    std::thread thread;
    alignas(64) std::atomic<uint64_t> head{ 0 };        // next slot of the front's ring this worker takes (its own line: the producer reads it when the ring looks full)
    alignas(64) std::atomic<bool> quit{ false };
    // the expectation collective (front::reduce_moments_batch_devices): this shard's device buffer [shard][vector][Σ, Σ², min, max] — its own
    // block filled by its launches, the others' by the all-gather — and the stream everything of this shard is ordered on
    fmhip_vec gather_vec = 0; double* gather_ptr = nullptr; size_t gather_doubles = 0; void* stream = nullptr;
    // worker-side tables: the front's numbers → this engine's handles
    PagedTable<fmhip_vec> local;
    std::unordered_map<int64_t, fmhip_program> programs;
    std::unordered_map<int64_t, fmhip_ticket> tickets;
    // the first thing that went wrong on this shard since the front last looked
    std::mutex error_mu;
    int error_code = FMHIP_OK;
    std::string error_message;

    fmhip_vec at(fmhip_vec front_handle) { const fmhip_vec* h = local.get(front_handle); return h ? *h : 0; }
    void bind(fmhip_vec front_handle, fmhip_vec mine) { local.put(front_handle, mine); }
    bool ok(int status) {
        if (status == FMHIP_OK) return true;
        std::lock_guard<std::mutex> lk(error_mu);
        if (error_code == FMHIP_OK) { error_code = status; error_message = "device shard " + std::to_string(shard) + ": " + fmhip_last_error(); }
        return false;
    }
    void execute(const Cmd& x, uint64_t position);
    void run();
};

struct Meta { int64_t n = 0; int refs = 0; };
struct ProgramMeta { int n_in = 0, n_out = 0, n_red = 0; };

struct Shards {
    std::mutex mu;                                     // the front: one caller at a time
    std::vector<std::unique_ptr<Worker>> workers;
    PagedTable<Meta> meta;                             // [front handle]
    int64_t next_vec = 1;
    std::unordered_map<int64_t, ProgramMeta> programs;
    std::unordered_map<int64_t, int> tickets;          // front ticket → count
    int64_t next_other = 1;                            // program and ticket numbers
    int fusion = 0, hold = 0, group_steps = 4, math_mode = FMHIP_MATH_EXACT, jit_mode = FMHIP_JIT_AUTO;
    // The one exchange between devices ("single-process ncclCommInitAll … inside ncclGroupStart/End"): a communicator per
    // listed device when the devices are DISTINCT and RCCL is there (looked up at run time: the library does not link it — a process with
    // one device never needs it); otherwise (an index repeats: shards of one device, the one-GPU test rig) expectations wanted on the
    // devices are combined on the host and copied back.  collective: 1 = RCCL, 2 = host combine.
    std::vector<ncclComm_t> comms;
    int collective = 2;
    std::string collective_why;
    struct Rccl {
        void* lib = nullptr;
        decltype(&ncclCommInitAll) CommInitAll = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr; decltype(&ncclGroupStart) GroupStart = nullptr;
        decltype(&ncclGroupEnd) GroupEnd = nullptr; decltype(&ncclAllGather) AllGather = nullptr; decltype(&ncclGetErrorString) GetErrorString = nullptr;
        bool load() {
            auto sym = [&](const char* name) -> void* { void* p = dlsym(RTLD_DEFAULT, name); if (!p && lib) p = dlsym(lib, name); return p; };
            if (!dlsym(RTLD_DEFAULT, "ncclCommInitAll")) { lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); }
            CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy"); GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
            GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd"); AllGather = (decltype(AllGather))sym("ncclAllGather"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
            return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather && GetErrorString;
        }
    } rccl;

    // The command queue: ONE ring for all shards — the producer (the front, one caller at a time under its mutex) writes a command once,
    // every worker reads it with a head of its own; a slot is written again when the slowest head has passed it.  A recorded method is
    // ≈ 50 ns of work for an engine, so the hand-over must not cost much more, and must not grow with the number of shards (until round 5:
    // one ring per worker and one copy of every command per shard — the caller's thread was the bound from a handful of shards on): plain
    // data, no lock and no system call while the workers are awake; the producer makes its commands visible (one store to a line the
    // consumers poll) every eighth command and whenever somebody is about to wait.  A worker that finds nothing spins for ≈ 50 µs, then
    // sleeps with a timeout: a tail the producer has not announced to a sleeper is found within a millisecond.
    static constexpr size_t RING = size_t(1) << 15;
    std::vector<Cmd> ring = std::vector<Cmd>(RING);
    std::vector<Command> fns = std::vector<Command>(RING);                  // the closure of slot i (Cmd::FN), dropped by the last worker that has run it
    std::unique_ptr<std::atomic<int>[]> fn_left{ new std::atomic<int>[RING]() };
    alignas(64) std::atomic<uint64_t> tail{ 0 };        // slots below this one are filled (published)
    alignas(64) uint64_t tail_local = 0;                // producer's own count (ahead of `tail` by at most 7)
    uint64_t head_seen = 0;                             // producer's last look at the slowest head
    std::mutex sleep_mu;
    std::condition_variable sleep_cv;
    std::atomic<int> sleeping{ 0 };

    int D() const { return (int)workers.size(); }
    uint64_t slowest_head() const { uint64_t h = ~uint64_t(0); for (const auto& w : workers) h = std::min(h, w->head.load(std::memory_order_acquire)); return h; }
    size_t next_slot() {                                // the index of the next slot to fill (producer)
        if (tail_local - head_seen >= RING) {           // looks full: look again, wait if it is
            publish();
            while (tail_local - (head_seen = slowest_head()) >= RING) std::this_thread::yield();
        }
        return (size_t)(tail_local & (RING - 1));
    }
    Cmd& slot() { return ring[next_slot()]; }
    void pushed() { if ((++tail_local & 7u) == 0) publish(); }
    void publish() {
        if (tail.load(std::memory_order_relaxed) == tail_local) return;
        tail.store(tail_local, std::memory_order_seq_cst);
        if (sleeping.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lk(sleep_mu); sleep_cv.notify_all(); }
    }
    // a closure for one shard (target >= 0) or for all: stored once, visible at once (closures are the rare calls)
    void post(Command c, int target = -1) {
        const size_t i = next_slot();
        Cmd& x = ring[i];
        x = Cmd(); x.kind = Cmd::FN; x.target = (int8_t)target;
        fns[i] = std::move(c);
        fn_left[i].store(D(), std::memory_order_release);
        pushed(); publish();
    }
    void stop_workers() { publish(); for (auto& w : workers) w->quit.store(true, std::memory_order_seq_cst); std::lock_guard<std::mutex> lk(sleep_mu); sleep_cv.notify_all(); }
    // waits until every worker has run everything posted so far; then the first error any shard has met (shard order) is thrown
    void wait() {
        std::mutex m; std::condition_variable cv; int left = D();
        post([&](Worker&) { std::lock_guard<std::mutex> lk(m); if (--left == 0) cv.notify_one(); });
        { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return left == 0; }); }
        check();
    }
    void check() {
        int code = FMHIP_OK; std::string message;
        for (auto& w : workers) {
            std::lock_guard<std::mutex> lk(w->error_mu);
            if (w->error_code != FMHIP_OK && code == FMHIP_OK) { code = w->error_code; message = w->error_message; }
            w->error_code = FMHIP_OK; w->error_message.clear();
        }
        if (code != FMHIP_OK) throw Error(code, message);
    }
    Meta& vec(fmhip_vec h) {
        Meta* m = meta.get(h);
        if (!m || m->refs <= 0) throw Error(FMHIP_ERR_INVALID_HANDLE, "invalid vector handle " + std::to_string(h));
        return *m;
    }
    fmhip_vec fresh(int64_t n) { const fmhip_vec id = next_vec++; meta.put(id, Meta{ n, 1 }); return id; }
};

void Worker::execute(const Cmd& x, uint64_t position) {
    switch (x.kind) {
    case Cmd::CALL: {
        fmhip_vec h = 0;
        int st;
        if (x.n_in == 1) st = x.has_scalar ? fmhip_call_v1s1(x.opcode, at(x.a0), x.scalar, &h) : fmhip_call_v1s0(x.opcode, at(x.a0), &h);
        else if (x.n_in == 2) st = x.has_scalar ? fmhip_call_v2s1(x.opcode, at(x.a0), at(x.a1), x.scalar, &h) : fmhip_call_v2s0(x.opcode, at(x.a0), at(x.a1), &h);
        else st = fmhip_call_v3s0(x.opcode, at(x.a0), at(x.a1), at(x.a2), &h);
        if (ok(st)) bind(x.id, h);
        break; }
    case Cmd::RELEASE: ok(fmhip_vec_release(at(x.a0))); if (x.last) local.erase(x.a0); break;
    case Cmd::RETAIN: ok(fmhip_vec_retain(at(x.a0))); break;
    case Cmd::FN: {
        const size_t i = (size_t)(position & (Shards::RING - 1));
        struct Last { Shards* f; size_t i; ~Last() { if (f->fn_left[i].fetch_sub(1, std::memory_order_acq_rel) == 1) f->fns[i] = nullptr; } } last{ front, i };      // (whoever runs it last lets the closure go)
        if (x.target < 0 || x.target == shard) front->fns[i](*this);
        break; }
    }
}

void Worker::run() {
    Engine::bind_thread(engine);
    Shards& f = *front;
    for (;;) {
        const uint64_t h = head.load(std::memory_order_relaxed);
        if (f.tail.load(std::memory_order_acquire) == h) {
            bool found = false;
            for (int spin = 0; spin < 2000 && !found; ++spin) {           // ≈ 50 µs
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                found = f.tail.load(std::memory_order_acquire) != h;
            }
            if (!found) {
                std::unique_lock<std::mutex> lk(f.sleep_mu);
                f.sleeping.fetch_add(1, std::memory_order_seq_cst);
                // (wait_until on the system clock = pthread_cond_timedwait; wait_for would be pthread_cond_clockwait, which this toolchain's ThreadSanitizer does not model)
                f.sleep_cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(1), [&] { return f.tail.load(std::memory_order_seq_cst) != h || quit.load(std::memory_order_seq_cst); });
                f.sleeping.fetch_sub(1, std::memory_order_seq_cst);
                if (f.tail.load(std::memory_order_seq_cst) == h && quit.load()) break;
            }
            continue;
        }
        const Cmd x = f.ring[(size_t)(h & (Shards::RING - 1))];
        try { execute(x, h); }
        catch (const Error& e) { std::lock_guard<std::mutex> lk(error_mu); if (error_code == FMHIP_OK) { error_code = e.code; error_message = "device shard " + std::to_string(shard) + ": " + e.what(); } }
        catch (const std::exception& e) { std::lock_guard<std::mutex> lk(error_mu); if (error_code == FMHIP_OK) { error_code = FMHIP_ERR_HIP; error_message = "device shard " + std::to_string(shard) + ": " + e.what(); } }
        head.store(h + 1, std::memory_order_release);
    }
    Engine::bind_thread(nullptr);
}

Shards* g_shards = nullptr;                            // non-null while a device list is active
std::mutex g_shards_mu;

template <typename F>
int fronted(F&& f) {
    try {
        Shards* s = g_shards;
        if (!s) throw Error(FMHIP_ERR_NOT_INITIALIZED, "fmhip_init_devices has not been called");
        std::lock_guard<std::mutex> lk(s->mu);
        f(*s);
        return FMHIP_OK;
    } catch (const Error& e) { set_last_error(e.what()); return e.code; }
    catch (const std::bad_alloc&) { set_last_error("host allocation failed"); return FMHIP_ERR_OUT_OF_MEMORY; }
    catch (const std::exception& e) { set_last_error(e.what()); return FMHIP_ERR_HIP; }
}

void need(const void* p, const char* what) { if (!p) throw Error(FMHIP_ERR_INVALID_ARGUMENT, std::string("null pointer: ") + what); }

// sums in shard order, java.lang.Math.min / max (NaN-propagating, -0.0 < +0.0): fmhip_expectation_combine's rule
void combine(const std::vector<std::vector<fmhip_moments>>& per_shard, int count, fmhip_moments* out) {
    std::vector<fmhip_moments> flat;
    for (const auto& v : per_shard) flat.insert(flat.end(), v.begin(), v.end());
    const int st = fmhip_expectation_combine(flat.data(), (int)per_shard.size(), count, out);
    if (st != FMHIP_OK) throw Error(st, fmhip_last_error());
}

struct OpShape { int n_vec; bool scalar; };
OpShape op_shape(int opcode) {
    if (opcode >= FMHIP_OP_CAP_S && opcode <= FMHIP_OP_POW_S) return { 1, true };
    if (opcode >= FMHIP_OP_SQUARED && opcode <= FMHIP_OP_ISNAN) return { 1, false };
    if (opcode >= FMHIP_OP_CAP && opcode <= FMHIP_OP_DIV) return { 2, false };
    if (opcode >= FMHIP_OP_ACCRUE && opcode <= FMHIP_OP_ADDPRODUCT_VS) return { 2, true };
    if (opcode >= FMHIP_OP_ADDPRODUCT && opcode <= FMHIP_OP_CHOOSE) return { 3, false };
    return { 0, false };
}

} // namespace

bool front_active() { return g_shards != nullptr && !Engine::thread_is_bound(); }

namespace front {

int init_devices(const int* devices, int count) {
    try {
        if (!devices || count < 1) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a device list needs at least one entry");
        if (count > 64) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "more than 64 device shards");
        std::lock_guard<std::mutex> lk(g_shards_mu);
        if (g_shards) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a device list is active already");
        if (Engine::get().initialized()) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "already initialised on one device: fmhip_shutdown first");
        std::unique_ptr<Shards> s(new Shards());
        for (int d = 0; d < count; ++d) {
            std::unique_ptr<Worker> w(new Worker());
            w->shard = d; w->shards = count; w->device = devices[d];
            w->engine = Engine::create();
            w->front = s.get();
            s->workers.push_back(std::move(w));
        }
        for (auto& w : s->workers) { Worker* p = w.get(); p->thread = std::thread([p] { p->run(); }); }
        s->post([](Worker& me) { me.ok(fmhip_init(me.device)); });
        auto tear_down = [&]() {
            s->post([](Worker&) { (void)fmhip_shutdown(); });
            s->stop_workers();
            for (auto& w : s->workers) { if (w->thread.joinable()) w->thread.join(); delete w->engine; w->engine = nullptr; }
        };
        try { s->wait(); } catch (...) { tear_down(); throw; }
        // what the engines read from the environment at initialisation
        s->post([&s](Worker&) { Engine& e = Engine::get(); s->group_steps = e.group_steps; s->jit_mode = e.jit_mode; }, 0);
        s->wait();
        {   // the expectation collective: RCCL over the listed devices when they are distinct
            bool distinct = true;
            for (int d = 0; d < count; ++d) for (int e = 0; e < d; ++e) distinct &= devices[d] != devices[e];
            static const bool WANT = [] { const char* e = std::getenv("FMHIP_DEVICE_LIST_RCCL"); return !(e && e[0] == '0'); }();
            if (!distinct) s->collective_why = "a device index repeats (shards of one device): expectations wanted on the devices are combined on the host";
            else if (!WANT) s->collective_why = "FMHIP_DEVICE_LIST_RCCL=0";
            else if (!s->rccl.load()) s->collective_why = "librccl.so not found";
            else {
                s->comms.assign((size_t)count, nullptr);
                const ncclResult_t r = s->rccl.CommInitAll(s->comms.data(), count, devices);
                if (r != ncclSuccess) { s->comms.clear(); s->collective_why = std::string("ncclCommInitAll: ") + s->rccl.GetErrorString(r); }
                else s->collective = 1;
            }
        }
        g_shards = s.release();
        return FMHIP_OK;
    } catch (const Error& e) { set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { set_last_error(e.what()); return FMHIP_ERR_HIP; }
}

int shutdown() {
    std::lock_guard<std::mutex> lk(g_shards_mu);
    Shards* s = g_shards;
    if (!s) return FMHIP_OK;
    {
        std::lock_guard<std::mutex> lk2(s->mu);
        s->post([](Worker& w) { (void)fmhip_synchronize(); if (w.gather_vec) { (void)fmhip_vec_release(w.gather_vec); w.gather_vec = 0; } });
        try { s->wait(); } catch (...) {}
        for (ncclComm_t c : s->comms) if (c) (void)s->rccl.CommDestroy(c);
        s->comms.clear();
        s->post([](Worker&) { (void)fmhip_shutdown(); });
        s->stop_workers();
        for (auto& w : s->workers) { if (w->thread.joinable()) w->thread.join(); delete w->engine; w->engine = nullptr; }
        g_shards = nullptr;
    }
    delete s;
    return FMHIP_OK;
}

int device_info(char* name_buf, int name_buf_len, int* n_compute_units, int64_t* hbm_bytes) {
    return fronted([&](Shards& s) {
        std::vector<int64_t> hbm((size_t)s.D(), 0);
        s.post([=, &hbm](Worker& me) { const bool first = me.shard == 0; me.ok(fmhip_device_info(first ? name_buf : nullptr, first ? name_buf_len : 0, first ? n_compute_units : nullptr, &hbm[(size_t)me.shard])); });
        s.wait();
        int64_t total = 0;                                     // the memory of every DISTINCT device of the list (an index may repeat: shards of one device)
        for (int d = 0; d < s.D(); ++d) { bool seen = false; for (int e = 0; e < d; ++e) seen |= s.workers[(size_t)e]->device == s.workers[(size_t)d]->device; if (!seen) total += hbm[(size_t)d]; }
        if (hbm_bytes) *hbm_bytes = total;
    });
}
int device_count(int* count) { return fronted([&](Shards& s) { need(count, "count"); *count = s.D(); }); }

int synchronize() { return fronted([&](Shards& s) { s.post([](Worker& w) { w.ok(fmhip_synchronize()); }); s.wait(); }); }

int vec_create_from_host(const void* host, bool is_double, int64_t n, fmhip_vec* out) {
    return fronted([&](Shards& s) {
        need(out, "out");
        if (n < 0 || n > (int64_t(1) << 31)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "invalid vector size " + std::to_string(n));
        if (n > 0 && !host) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null host pointer");
        const fmhip_vec id = s.fresh(n);
        s.post([=](Worker& w) {                                // straight from the caller's buffer: the call waits for the uploads
            int64_t off, cnt; shard_range(n, w.shards, w.shard, &off, &cnt);
            fmhip_vec h = 0;
            const int st = is_double ? fmhip_vec_create_from_double((const double*)host + off, cnt, &h) : fmhip_vec_create_from_float((const float*)host + off, cnt, &h);
            if (w.ok(st)) w.bind(id, h);
        });
        // (a shard that failed leaves the others with a block bound under this number: they give it back)
        try { s.wait(); } catch (...) { s.post([=](Worker& w) { if (const fmhip_vec h = w.at(id)) { (void)fmhip_vec_release(h); w.local.erase(id); } }); s.meta.erase(id); throw; }
        *out = id;
    });
}

int vec_create_filled(int64_t n, double value, bool initialised, fmhip_vec* out) {
    return fronted([&](Shards& s) {
        need(out, "out");
        if (n < 0 || n > (int64_t(1) << 31)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "invalid vector size " + std::to_string(n));
        const fmhip_vec id = s.fresh(n);
        s.post([=](Worker& w) {
            int64_t off, cnt; shard_range(n, w.shards, w.shard, &off, &cnt);
            fmhip_vec h = 0;
            if (w.ok(initialised ? fmhip_vec_create_filled(cnt, value, &h) : fmhip_vec_create_uninitialized(cnt, &h))) w.bind(id, h);
        });
        *out = id;
    });
}

int vec_retain(fmhip_vec v) {
    return fronted([&](Shards& s) {
        s.vec(v).refs++;
        Cmd& x = s.slot(); x = Cmd(); x.kind = Cmd::RETAIN; x.a0 = v; s.pushed();
    });
}
int vec_release(fmhip_vec v) {
    return fronted([&](Shards& s) {
        const bool last = --s.vec(v).refs == 0;
        if (last) s.meta.erase(v);
        Cmd& x = s.slot(); x = Cmd(); x.kind = Cmd::RELEASE; x.a0 = v; x.last = last; s.pushed();
    });
}
int vec_size(fmhip_vec v, int64_t* n_out) { return fronted([&](Shards& s) { need(n_out, "n_out"); *n_out = s.vec(v).n; }); }

int vec_read(fmhip_vec v, void* host_out, bool as_double, int64_t n) {
    return fronted([&](Shards& s) {
        const Meta& m = s.vec(v);
        if (n != m.n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "read of " + std::to_string(n) + " elements from a vector of " + std::to_string(m.n));
        if (n > 0 && !host_out) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null host pointer");
        s.post([=](Worker& w) {                                // every shard copies its block straight into the caller's buffer
            int64_t off, cnt; shard_range(n, w.shards, w.shard, &off, &cnt);
            w.ok(as_double ? fmhip_vec_read_double(w.at(v), (double*)host_out + off, cnt) : fmhip_vec_read_float(w.at(v), (float*)host_out + off, cnt));
        });
        s.wait();
    });
}

int call(int opcode, int n_in, const fmhip_vec* in, double scalar, bool has_scalar, fmhip_vec* out) {
    return fronted([&](Shards& s) {
        need(out, "out");
        const OpShape shape = op_shape(opcode);
        if (shape.n_vec == 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown opcode " + std::to_string(opcode));
        if (shape.n_vec != n_in || shape.scalar != has_scalar) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "opcode " + std::to_string(opcode) + " does not match this call shape");
        fmhip_vec a[3] = { 0, 0, 0 };
        int64_t n = 0;
        for (int i = 0; i < n_in; ++i) {
            const Meta& m = s.vec(in[i]);
            if (i == 0) n = m.n;
            else if (m.n != n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "operand sizes differ: " + std::to_string(n) + " vs " + std::to_string(m.n));
            a[i] = in[i];
        }
        const fmhip_vec id = s.fresh(n);
        Cmd& x = s.slot();                                     // written ONCE, whatever the number of shards
        x.kind = Cmd::CALL; x.has_scalar = has_scalar; x.last = false; x.target = -1; x.opcode = opcode; x.n_in = n_in; x.scalar = scalar;
        x.a0 = a[0]; x.a1 = a[1]; x.a2 = a[2]; x.id = id;
        s.pushed();
        *out = id;
    });
}

int set_int(int what, int value, int* previous) {
    return fronted([&](Shards& s) {
        int* mirror = what == 0 ? &s.fusion : what == 1 ? &s.hold : what == 2 ? &s.group_steps : what == 3 ? &s.math_mode : &s.jit_mode;
        if (what == 2 && value < 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "negative number of time steps");
        if (what == 3 && value != FMHIP_MATH_EXACT && value != FMHIP_MATH_FAST) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown math mode");
        if (what == 4 && value != FMHIP_JIT_OFF && value != FMHIP_JIT_AUTO && value != FMHIP_JIT_SYNC) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown JIT mode");
        if (previous) *previous = *mirror;
        *mirror = what == 0 ? (value != 0) : what == 1 ? (value == 2 ? 2 : (value != 0 ? 1 : 0)) : value;
        s.post([=](Worker& w) {
            w.ok(what == 0 ? fmhip_set_fusion(value, nullptr) : what == 1 ? fmhip_fusion_hold(value, nullptr) : what == 2 ? fmhip_set_step_grouping(value, nullptr)
                 : what == 3 ? fmhip_set_math_mode(value, nullptr) : fmhip_set_jit(value, nullptr));
        });
    });
}

int flush() { return fronted([&](Shards& s) { s.post([](Worker& w) { w.ok(fmhip_flush()); }); s.check(); }); }

int graph_clone(const fmhip_vec* roots, int n_roots, int n_copies, const fmhip_vec* leaf_from, const fmhip_vec* leaf_to, int n_map, const double* scalars, int n_scalars, fmhip_vec* out) {
    return fronted([&](Shards& s) {
        if (n_roots <= 0 || !roots) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "no roots");
        if (n_copies < 0 || n_map < 0 || !out || (n_map > 0 && (!leaf_from || !leaf_to))) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad graph replication request");
        std::vector<int64_t> root_n;
        for (int r = 0; r < n_roots; ++r) root_n.push_back(s.vec(roots[r]).n);
        for (int i = 0; i < n_map; ++i) { const int64_t n = s.vec(leaf_from[i]).n; for (int j = 0; j < n_copies; ++j) if (s.vec(leaf_to[(size_t)j * n_map + i]).n != n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "a substituted operand differs in size"); }
        auto r = std::make_shared<std::vector<fmhip_vec>>(roots, roots + n_roots);
        auto lf = std::make_shared<std::vector<fmhip_vec>>(leaf_from, leaf_from + n_map);
        auto lt = std::make_shared<std::vector<fmhip_vec>>(leaf_to, leaf_to + (size_t)n_map * n_copies);
        auto sc = std::make_shared<std::vector<double>>();
        if (scalars) sc->assign(scalars, scalars + (size_t)n_scalars * n_copies);
        const bool with_scalars = scalars != nullptr;
        std::vector<fmhip_vec> ids((size_t)n_roots * n_copies);
        for (int j = 0; j < n_copies; ++j) for (int k = 0; k < n_roots; ++k) ids[(size_t)j * n_roots + k] = s.fresh(root_n[(size_t)k]);
        auto idv = std::make_shared<std::vector<fmhip_vec>>(ids);
        s.post([=](Worker& w) {
            std::vector<fmhip_vec> lr, lfrom, lto, got(idv->size(), 0);
            for (fmhip_vec h : *r) lr.push_back(w.at(h));
            for (fmhip_vec h : *lf) lfrom.push_back(w.at(h));
            for (fmhip_vec h : *lt) lto.push_back(w.at(h));
            if (w.ok(fmhip_graph_clone(lr.data(), n_roots, n_copies, lfrom.data(), lto.data(), n_map, with_scalars ? sc->data() : nullptr, n_scalars, got.data())))
                for (size_t i = 0; i < got.size(); ++i) w.bind((*idv)[i], got[i]);
        });
        // (the count of scalar operands is checked by the engines: an error there surfaces at the next call that waits)
        std::memcpy(out, ids.data(), ids.size() * sizeof(fmhip_vec));
    });
}

int graph_scalars(const fmhip_vec* roots, int n_roots, double* scalars_out, int capacity, int* n_scalars) {
    return fronted([&](Shards& s) {
        if (!n_scalars) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null count pointer");
        if (n_roots <= 0 || !roots) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "no roots");
        for (int r = 0; r < n_roots; ++r) (void)s.vec(roots[r]);
        std::vector<fmhip_vec> r(roots, roots + n_roots);
        s.post([&](Worker& w) {                                // every shard holds the same graph: shard 0 answers
            std::vector<fmhip_vec> lr; for (fmhip_vec h : r) lr.push_back(w.at(h));
            w.ok(fmhip_graph_scalars(lr.data(), n_roots, scalars_out, capacity, n_scalars));
        }, 0);
        s.wait();
    });
}

int reduce_moments_batch(const fmhip_vec* vectors, int count, const double* shifts, fmhip_moments* out) {
    return fronted([&](Shards& s) {
        need(vectors, "vectors"); need(out, "out");
        if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
        for (int i = 0; i < count; ++i) if (s.vec(vectors[i]).n != s.vec(vectors[0]).n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "batched reduction over vectors of different size");
        std::vector<std::vector<fmhip_moments>> per((size_t)s.D(), std::vector<fmhip_moments>((size_t)count));
        s.post([&](Worker& w) {
            std::vector<fmhip_vec> l; for (int i = 0; i < count; ++i) l.push_back(w.at(vectors[i]));
            w.ok(count == 1 ? fmhip_reduce_moments(l[0], shifts ? shifts[0] : 0.0, per[(size_t)w.shard].data()) : fmhip_reduce_moments_batch(l.data(), count, shifts, per[(size_t)w.shard].data()));
        });
        s.wait();
        combine(per, count, out);
    });
}

int reduce_moments_batch_begin(const fmhip_vec* vectors, int count, const double* shifts, fmhip_ticket* ticket_out) {
    return fronted([&](Shards& s) {
        need(vectors, "vectors"); need(ticket_out, "ticket_out");
        if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
        for (int i = 0; i < count; ++i) if (s.vec(vectors[i]).n != s.vec(vectors[0]).n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "batched reduction over vectors of different size");
        auto v = std::make_shared<std::vector<fmhip_vec>>(vectors, vectors + count);
        auto sh = std::make_shared<std::vector<double>>();
        if (shifts) sh->assign(shifts, shifts + count);
        const bool shifted = shifts != nullptr;
        const int64_t id = s.next_other++;
        s.tickets[id] = count;
        s.post([=](Worker& w) {
            std::vector<fmhip_vec> l; for (fmhip_vec h : *v) l.push_back(w.at(h));
            fmhip_ticket t = 0;
            if (w.ok(fmhip_reduce_moments_batch_begin(l.data(), count, shifted ? sh->data() : nullptr, &t))) w.tickets[id] = t;
        });
        *ticket_out = id;
    });
}

int reduce_moments_batch_end(fmhip_ticket ticket, fmhip_moments* out, int count) {
    return fronted([&](Shards& s) {
        need(out, "out");
        auto it = s.tickets.find(ticket);
        if (it == s.tickets.end()) throw Error(FMHIP_ERR_INVALID_HANDLE, "unknown (or already ended) expectation ticket");
        if (it->second != count) throw Error(FMHIP_ERR_SIZE_MISMATCH, "the ticket holds " + std::to_string(it->second) + " expectations, the caller asks for " + std::to_string(count));
        s.tickets.erase(it);
        std::vector<std::vector<fmhip_moments>> per((size_t)s.D(), std::vector<fmhip_moments>((size_t)count));
        s.post([&, ticket](Worker& w) {
            auto mine = w.tickets.find(ticket);
            if (mine == w.tickets.end()) { w.ok(FMHIP_ERR_INVALID_HANDLE); return; }
            const fmhip_ticket t = mine->second;
            w.tickets.erase(mine);
            w.ok(fmhip_reduce_moments_batch_end(t, per[(size_t)w.shard].data(), count));
        });
        s.wait();
        combine(per, count, out);
    });
}

// The expectations of `count` vectors ON THE DEVICES: device_out[d] (a buffer of count x 32 bytes on the d-th listed device; nullptr: not
// wanted there) receives the moments of the WHOLE vectors — all shards combined in shard order, the same bits on every device.
// Every shard's launches leave its moments in its block of its own gather buffer; ONE all-gather, issued for all devices from this thread
// inside ncclGroupStart / ncclGroupEnd on the shards' streams, makes every device hold all blocks; fm_combine_moments_kernel combines
// them per device.  Nothing waits for the devices here: the results are ordered on the shards' streams (fmhip_get_stream_of).  With a
// repeated device index (or without RCCL) the shards' blocks are read back, combined on the host and copied to the devices.
int reduce_moments_batch_devices(const fmhip_vec* vectors, int count, const double* shifts, void* const* device_out, int n_devices) {
    return fronted([&](Shards& s) {
        need(vectors, "vectors"); need(device_out, "device_out");
        if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
        const int D = s.D();
        if (n_devices != D) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "one output pointer per listed device: " + std::to_string(D) + " expected, " + std::to_string(n_devices) + " given");
        for (int i = 0; i < count; ++i) if (s.vec(vectors[i]).n != s.vec(vectors[0]).n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "batched reduction over vectors of different size");
        const size_t block = (size_t)count * 4;                // doubles per shard
        s.post([&](Worker& w) {
            if (w.gather_doubles < block * (size_t)w.shards) {
                if (w.gather_vec) { w.ok(fmhip_vec_release(w.gather_vec)); w.gather_vec = 0; w.gather_ptr = nullptr; w.gather_doubles = 0; }
                const size_t want = std::max(block * (size_t)w.shards, (size_t)4096);
                fmhip_vec h = 0; void* p = nullptr;
                if (!w.ok(fmhip_vec_create_uninitialized((int64_t)want * 2, &h)) || !w.ok(fmhip_vec_device_ptr(h, &p))) { if (h) (void)fmhip_vec_release(h); return; }
                w.gather_vec = h; w.gather_ptr = (double*)p; w.gather_doubles = want;
            }
            std::vector<fmhip_vec> l; for (int i = 0; i < count; ++i) l.push_back(w.at(vectors[i]));
            w.ok(fmhip_reduce_moments_batch_device(l.data(), count, shifts, w.gather_ptr + block * (size_t)w.shard));
            w.ok(fmhip_get_stream(&w.stream));
        });
        s.wait();
        if (s.collective == 1) {
            auto nccl = [&](ncclResult_t r, const char* what) { if (r != ncclSuccess) throw Error(FMHIP_ERR_HIP, std::string(what) + ": " + s.rccl.GetErrorString(r)); };
            nccl(s.rccl.GroupStart(), "ncclGroupStart");
            ncclResult_t first = ncclSuccess;
            for (int d = 0; d < D; ++d) {
                Worker& w = *s.workers[(size_t)d];
                const ncclResult_t r = s.rccl.AllGather(w.gather_ptr + block * (size_t)d, w.gather_ptr, block, ncclDouble, s.comms[(size_t)d], (hipStream_t)w.stream);
                if (first == ncclSuccess) first = r;
            }
            nccl(s.rccl.GroupEnd(), "ncclGroupEnd");
            nccl(first, "ncclAllGather");
            for (int d = 0; d < D; ++d) {
                if (!device_out[d]) continue;
                Worker& w = *s.workers[(size_t)d];
                hip_check(hipSetDevice(w.device), "hipSetDevice");
                hip_check(launch_combine_moments(w.gather_ptr, (uint32_t)D, (uint32_t)count, (double*)device_out[d], (hipStream_t)w.stream), "launch fm_combine_moments_kernel");
            }
            return;
        }
        // host combine: every shard's block read back, combined, handed to the devices that want it
        std::vector<std::vector<fmhip_moments>> per((size_t)D, std::vector<fmhip_moments>((size_t)count));
        s.post([&](Worker& w) {
            hipError_t e = hipMemcpyAsync(per[(size_t)w.shard].data(), w.gather_ptr + block * (size_t)w.shard, block * 8, hipMemcpyDeviceToHost, (hipStream_t)w.stream);
            if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)w.stream);
            if (e != hipSuccess) { (void)hipGetLastError(); w.ok(FMHIP_ERR_HIP); }
        });
        s.wait();
        std::vector<fmhip_moments> all((size_t)count);
        combine(per, count, all.data());
        s.post([&](Worker& w) {
            if (!device_out[w.shard]) return;
            hipError_t e = hipMemcpyAsync(device_out[w.shard], all.data(), block * 8, hipMemcpyHostToDevice, (hipStream_t)w.stream);
            if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)w.stream);      // (`all` is this call's)
            if (e != hipSuccess) { (void)hipGetLastError(); w.ok(FMHIP_ERR_HIP); }
        });
        s.wait();
    });
}
int get_stream_of(int shard, void** stream_out) {
    return fronted([&](Shards& s) {
        need(stream_out, "stream_out");
        if (shard < 0 || shard >= s.D()) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "no such device shard: " + std::to_string(shard));
        s.post([=](Worker& w) { w.ok(fmhip_get_stream(stream_out)); }, shard);
        s.wait();
    });
}
int expectation_collective(int* kind, char* why, int why_len) {
    return fronted([&](Shards& s) {
        if (kind) *kind = s.collective;
        if (why && why_len > 0) { std::strncpy(why, s.collective_why.c_str(), (size_t)why_len - 1); why[why_len - 1] = 0; }
    });
}

int vec_give_up_values(const fmhip_vec* vectors, int count) {
    return fronted([&](Shards& s) {
        need(vectors, "vectors");
        if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
        for (int i = 0; i < count; ++i) (void)s.vec(vectors[i]);
        auto v = std::make_shared<std::vector<fmhip_vec>>(vectors, vectors + count);
        s.post([=](Worker& w) { std::vector<fmhip_vec> l; for (fmhip_vec h : *v) l.push_back(w.at(h)); w.ok(fmhip_vec_give_up_values(l.data(), count)); });
    });
}

int program_create(const fmhip_prog_op* ops, int n_ops, int n_inputs, const int32_t* out_values, int n_outputs, const int32_t* reduce_values, int n_reduce, fmhip_program* out) {
    return fronted([&](Shards& s) {
        need(out, "out");
        const int64_t id = s.next_other++;
        s.post([&, id](Worker& w) { fmhip_program p = 0; if (w.ok(fmhip_program_create(ops, n_ops, n_inputs, out_values, n_outputs, reduce_values, n_reduce, &p))) w.programs[id] = p; });
        s.wait();                                              // (a program that does not compile is refused here, as on one device)
        s.programs[id] = { n_inputs, n_outputs, n_reduce };
        *out = id;
    });
}
int program_release(fmhip_program p) {
    return fronted([&](Shards& s) {
        if (!s.programs.erase(p)) throw Error(FMHIP_ERR_INVALID_HANDLE, "invalid program handle " + std::to_string(p));
        s.post([=](Worker& w) { auto it = w.programs.find(p); if (it != w.programs.end()) { w.ok(fmhip_program_release(it->second)); w.programs.erase(it); } });
    });
}
int program_shape(fmhip_program p, int* n_inputs, int* n_outputs, int* n_reduce) {
    return fronted([&](Shards& s) {
        auto it = s.programs.find(p);
        if (it == s.programs.end()) throw Error(FMHIP_ERR_INVALID_HANDLE, "invalid program handle " + std::to_string(p));
        if (n_inputs) *n_inputs = it->second.n_in;
        if (n_outputs) *n_outputs = it->second.n_out;
        if (n_reduce) *n_reduce = it->second.n_red;
    });
}
int program_tier(fmhip_program p, int* tier, int* vgprs) {
    return fronted([&](Shards& s) {
        if (!s.programs.count(p)) throw Error(FMHIP_ERR_INVALID_HANDLE, "invalid program handle " + std::to_string(p));
        s.post([=](Worker& w) { w.ok(fmhip_program_tier(w.programs[p], tier, vgprs)); }, 0);
        s.wait();
    });
}

int program_run(fmhip_program p, int batch, const fmhip_vec* inputs, fmhip_vec* outputs, bool into, const double* reduce_shift, fmhip_moments* moments) {
    return fronted([&](Shards& s) {
        auto it = s.programs.find(p);
        if (it == s.programs.end()) throw Error(FMHIP_ERR_INVALID_HANDLE, "invalid program handle " + std::to_string(p));
        const ProgramMeta pm = it->second;
        if (batch <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "batch must be positive");
        if (!inputs || (pm.n_out > 0 && !outputs)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null handle array");
        int64_t n = -1;
        for (int i = 0; i < batch * pm.n_in; ++i) { const int64_t m = s.vec(inputs[i]).n; if (n < 0) n = m; else if (m != n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "program inputs differ in size"); }
        auto in = std::make_shared<std::vector<fmhip_vec>>(inputs, inputs + (size_t)batch * pm.n_in);
        auto outs = std::make_shared<std::vector<fmhip_vec>>((size_t)batch * pm.n_out);
        for (size_t i = 0; i < outs->size(); ++i) {
            if (into) { if (s.vec(outputs[i]).n != n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "program output differs in size"); (*outs)[i] = outputs[i]; }
            else (*outs)[i] = s.fresh(n);
        }
        auto sh = std::make_shared<std::vector<double>>();
        if (reduce_shift) sh->assign(reduce_shift, reduce_shift + pm.n_red);
        const bool shifted = reduce_shift != nullptr, want = moments != nullptr;
        const int n_mom = batch * pm.n_red;
        auto per = std::make_shared<std::vector<std::vector<fmhip_moments>>>((size_t)s.D(), std::vector<fmhip_moments>((size_t)std::max(1, n_mom)));
        s.post([=](Worker& w) {
            std::vector<fmhip_vec> li, lo(outs->size(), 0);
            for (fmhip_vec h : *in) li.push_back(w.at(h));
            if (into) for (size_t i = 0; i < outs->size(); ++i) lo[i] = w.at((*outs)[i]);
            fmhip_moments* m = want ? (*per)[(size_t)w.shard].data() : nullptr;
            const int st = into ? fmhip_program_run_into(w.programs[p], batch, li.data(), lo.data(), shifted ? sh->data() : nullptr, m, nullptr)
                                : fmhip_program_run(w.programs[p], batch, li.data(), lo.data(), shifted ? sh->data() : nullptr, m, nullptr);
            if (w.ok(st) && !into) for (size_t i = 0; i < lo.size(); ++i) w.bind((*outs)[i], lo[i]);
        });
        if (want) { s.wait(); if (n_mom > 0) combine(*per, n_mom, moments); }
        if (!into) std::memcpy(outputs, outs->data(), outs->size() * sizeof(fmhip_vec));
    });
}

int jit_wait() { return fronted([&](Shards& s) { s.post([](Worker& w) { w.ok(fmhip_jit_wait()); }); s.wait(); }); }
int jit_stats(int64_t* compiled, int64_t* failed, int64_t* pending, double* compile_seconds, int64_t* disk_cache_hits) {
    return fronted([&](Shards& s) {
        struct One { int64_t c = 0, f = 0, p = 0, d = 0; double sec = 0; };
        std::vector<One> per((size_t)s.D());
        s.post([&](Worker& w) { One& o = per[(size_t)w.shard]; w.ok(fmhip_jit_stats(&o.c, &o.f, &o.p, &o.sec, &o.d)); });
        s.wait();
        One t; for (const One& o : per) { t.c += o.c; t.f += o.f; t.p += o.p; t.d += o.d; t.sec += o.sec; }
        if (compiled) *compiled = t.c;
        if (failed) *failed = t.f;
        if (pending) *pending = t.p;
        if (compile_seconds) *compile_seconds = t.sec;
        if (disk_cache_hits) *disk_cache_hits = t.d;
    });
}

int bm_generate(int64_t seed, int n_steps, int n_factors, int64_t n_paths, int64_t path_offset, const double* dt, fmhip_vec* out) {
    return fronted([&](Shards& s) {
        if (n_steps <= 0 || n_factors <= 0 || !dt || !out || path_offset < 0 || n_paths < 0 || n_paths > (int64_t(1) << 31)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad Brownian motion description");
        if (path_offset % 4 != 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "with a device list the path offset of a Brownian motion must be a multiple of four");
        for (int i = 0; i < n_steps; ++i) if (!(dt[i] >= 0.0)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "negative time step");
        const size_t streams = (size_t)n_steps * n_factors;
        auto ids = std::make_shared<std::vector<fmhip_vec>>(streams);
        for (size_t k = 0; k < streams; ++k) (*ids)[k] = s.fresh(n_paths);
        auto steps = std::make_shared<std::vector<double>>(dt, dt + n_steps);
        s.post([=](Worker& w) {                                // shard d generates ITS block of every increment: the counter is the global path index
            int64_t off, cnt; shard_range(n_paths, w.shards, w.shard, &off, &cnt);
            std::vector<fmhip_vec> got(ids->size(), 0);
            if (w.ok(fmhip_bm_generate(seed, n_steps, n_factors, cnt, path_offset + off, steps->data(), got.data())))
                for (size_t k = 0; k < got.size(); ++k) w.bind((*ids)[k], got[k]);
        });
        std::memcpy(out, ids->data(), streams * sizeof(fmhip_vec));
    });
}

int pool(int what) { return fronted([&](Shards& s) { s.post([=](Worker& w) { w.ok(what == 0 ? fmhip_pool_clean() : fmhip_pool_purge()); }); s.wait(); }); }
int pool_stats(fmhip_pool_stats_t* out) {
    return fronted([&](Shards& s) {
        need(out, "out");
        std::vector<fmhip_pool_stats_t> per((size_t)s.D());
        s.post([&](Worker& w) { w.ok(fmhip_pool_stats(&per[(size_t)w.shard])); });
        s.wait();
        fmhip_pool_stats_t t{};
        for (const fmhip_pool_stats_t& p : per) {
            t.bytes_reserved += p.bytes_reserved; t.bytes_in_use += p.bytes_in_use; t.bytes_cached += p.bytes_cached;
            t.device_bytes_free += p.device_bytes_free; t.device_bytes_total += p.device_bytes_total;
            t.n_alloc_hits += p.n_alloc_hits; t.n_alloc_misses += p.n_alloc_misses;
            t.n_kernel_launches += p.n_kernel_launches; t.n_ops_executed += p.n_ops_executed;
        }
        t.n_live_vectors = (int64_t)s.meta.size();             // the caller's vectors (every shard holds a block of each)
        *out = t;
    });
}
int traffic_stats(int64_t* algorithmic_bytes, int64_t* specialised_launches) {
    return fronted([&](Shards& s) {
        std::vector<int64_t> b((size_t)s.D(), 0), l((size_t)s.D(), 0);
        s.post([&](Worker& w) { w.ok(fmhip_traffic_stats(&b[(size_t)w.shard], &l[(size_t)w.shard])); });
        s.wait();
        int64_t tb = 0, tl = 0; for (int d = 0; d < s.D(); ++d) { tb += b[(size_t)d]; tl += l[(size_t)d]; }
        if (algorithmic_bytes) *algorithmic_bytes = tb;
        if (specialised_launches) *specialised_launches = tl;
    });
}
int engine_stats(fmhip_engine_stats_t* out) {
    return fronted([&](Shards& s) {
        need(out, "out");
        std::vector<fmhip_engine_stats_t> per((size_t)s.D());
        s.post([&](Worker& w) { w.ok(fmhip_engine_stats(&per[(size_t)w.shard])); });
        s.wait();
        fmhip_engine_stats_t t; std::memset(&t, 0, sizeof t);
        for (const fmhip_engine_stats_t& p : per) { int64_t* a = &t.size; const int64_t* b = &p.size; for (size_t i = 1; i < sizeof t / sizeof(int64_t); ++i) a[i] += b[i]; }
        t.size = (int64_t)sizeof t;
        *out = t;
    });
}
int profile_enable(int enabled) { return fronted([&](Shards& s) { s.post([=](Worker& w) { w.ok(fmhip_profile_enable(enabled)); }); s.wait(); }); }
int profile_read(double* kernel_ms_total, int64_t* n_launches) {
    return fronted([&](Shards& s) {                            // the shards run side by side: the slowest one's device time, shard 0's launches
        std::vector<double> ms((size_t)s.D(), 0.0); std::vector<int64_t> n((size_t)s.D(), 0);
        s.post([&](Worker& w) { w.ok(fmhip_profile_read(&ms[(size_t)w.shard], &n[(size_t)w.shard])); });
        s.wait();
        double worst = 0.0; for (double v : ms) worst = std::max(worst, v);
        if (kernel_ms_total) *kernel_ms_total = worst;
        if (n_launches) *n_launches = n[0];
    });
}
int unsupported(const char* what) {
    set_last_error(std::string(what) + " is not available with a device list (fmhip_init_devices): it names one device");
    return FMHIP_ERR_INVALID_ARGUMENT;
}

} // namespace front
} // namespace fm
