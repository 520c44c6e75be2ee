// runtime.hpp — process-wide engine behind the C-ABI: device binding, stream, pooling allocator,
// vector handles, program compiler (SSA → register bytecode), lazy fusion front-end.
//
// Replaces the reference's DeviceMemoryPool inner class (RandomVariableCuda.java:119-558):
//   reference                                         here
//   ------------------------------------------------------------------------------------------
//   pool keyed by GC reachability (WeakReference +    explicit retain/release, size-class free lists,
//   ReferenceQueue, System.gc(), cudaMemGetInfo on    no GC polling, no hipMemGetInfo on the hot path
//   every miss: :280-390)
//   single-thread executor serialising all CUDA       any thread may call; one in-order HIP stream,
//   calls (:155)                                      one mutex around the (short) enqueue paths
//   one launch per method (:539-557)                  op streams compiled to one launch (Program)
#pragma once
#include <hip/hip_runtime_api.h>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/fmhip.h"
#include "fm_program.h"
#include "jit.hpp"

namespace fm {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---------------------------------------------------------------- device memory

class Pool {
public:
    void* alloc(size_t bytes, size_t* cap_out);       // throws Error(OOM)
    void  release(void* p, size_t cap);               // back to the free list (stream-ordered reuse)
    void  purge(size_t need = ~size_t(0));            // hipFree what is cached: everything, or slabs until `need` bytes have gone back to the driver
    int64_t reserved = 0, in_use = 0, cached = 0, hits = 0, misses = 0, peak_reserved = 0;
private:
    struct Slab { void* base; size_t cap; size_t blocks; size_t parent_cap; };       // parent_cap != 0: carved out of a cached block of that size class (borrow)
    void* borrow(size_t cap);
    std::unordered_map<size_t, std::vector<void*>> free_;
    std::vector<Slab> slabs_;
    std::unordered_map<size_t, size_t> slab_blocks_;       // blocks of the last slab per size class (geometric growth)
};

struct Buffer {                 // refcounted device storage; views keep their parent alive
    float*  ptr = nullptr;
    size_t  cap = 0;            // bytes owned (0 for views)
    int     refs = 0;
    Buffer* parent = nullptr;
    int32_t foreign = -1;       // >= 0: storage of ANOTHER engine's vector (thread engines: Engine::import_vector); index into foreign_
};

// ---------------------------------------------------------------- vectors / lazy nodes

struct Node {
    int64_t id = 0;
    int64_t n = 0;
    int     refs_ext = 0;       // handles held by the caller
    int     refs_int = 0;       // references from pending consumer nodes
    Buffer* buf = nullptr;      // non-null ⇔ materialised
    // pending expression (valid while buf == nullptr)
    int     opcode = 0;
    int     n_in = 0;
    Node*   in[3] = { nullptr, nullptr, nullptr };
    double  scalar = 0.0;
    int     weight = 0;         // upper bound of pending ops below this node (fusion budget)
    // {Σ, Σ², min, max} of the vector when somebody has computed them already (a batched flush that took the expectations of all
    // pending roots along: Engine::reduce); vectors are immutable — the two ways to write into one (fmhip_program_run_into, a raw device
    // pointer handed out) clear and block this
    bool    has_moments = false, moments_blocked = false;
    // fmhip_vec_give_up_values: the caller wants this pending value's moments and will never read the value itself.  `discard`: marked
    // (a root nobody else references); `discarded`: a flush took the moments in the launch that computed the value and did NOT store
    // it — the node stays without storage, is not a root of later flushes, and reading it is an error.
    bool    discard = false, discarded = false;
    // DEFERRED: a launch computed this value in registers for its consumers and did not store it, although the caller still holds a
    // handle (Engine: escape policy).  The node keeps its expression — the recipe: references on its operands —, is on the engine's
    // deferred list instead of the pending list (no flush executes it on its own account), and is computed and stored once if the
    // handle is ever used again (an operand of a later call, a read, an expectation).  The release that eventually arrives frees the recipe.
    bool    deferred = false;
    // escape policy: the slot (index + 1 into Engine::policy_state_, of generation watch_gen) that learns from what happens to this handle
    uint32_t watch = 0, watch_gen = 0;
    // scratch fields of the DAG builder (valid when mark == the builder's current epoch): no hash maps on the hot path
    uint64_t mark = 0;
    int     tmp_id = 0, tmp_uses = 0;
    // membership in the engine's list of pending nodes (intrusive, O(1) insert / erase without hashing: recording a method is
    // ≈ 60 ns in all, a hash-set insert was a third of it)
    Node*   pend_prev = nullptr;
    Node*   pend_next = nullptr;
    // replica groups (fmhip_graph_clone, see ReplicaGroup): rep_id != 0 marks an operation of a replicated pending graph
    // (rep_copy == 0; rep_index = its position in recording order, rep_root = its number among the replicated roots or -1) or
    // the root of a copy that exists as a description only (rep_copy = copy number + 1, rep_root = which root)
    uint32_t rep_id = 0;
    int32_t  rep_index = -1, rep_root = -1;
    uint32_t rep_copy = 0;
    uint32_t leaf_rep_id = 0;   // a substituted operand (leaf_from[i]) of that live group …
    int32_t  leaf_rep_index = -1;   // … i
    // a Brownian increment (fmhip_bm_generate): which generation it belongs to, its time index, the number of time steps
    uint32_t bm_id = 0;
    int32_t  bm_step = -1, bm_steps = 0;
    // (behind everything that recording, releasing and walking a pending graph touch: the moments themselves)
    double  moments[4] = { 0.0, 0.0, 0.0, 0.0 };
    volatile uint64_t* moments_slot = nullptr;   // the same four values still on their way from a launch (pinned memory, Engine::moments_arena_)
};

// Copies of a pending graph that exist as a DESCRIPTION until the graph runs (fmhip_graph_clone).  A copy differs from the
// original in the vectors it reads (leaf_from[i] → leaf_to[copy][i]) and in its scalar operands, never in structure: instead of
// duplicating every operation node — and walking, signing, scheduling and dismantling the duplicates at the next flush, which is
// where a launch-bound caller's host time went (LMM calibration: 17.7 of 21 M nodes per run were such duplicates) — the engine
// creates only the copies' ROOT nodes and, when the original's components are executed, appends one batch row per copy to the
// original's launches: inputs by substitution, scalars from the copy's list, outputs into the copy's root nodes.  Whenever the
// original's graph is no longer what was replicated (operations recorded on top of it, other handles inside it, a copy's root
// used before the flush), the description is expanded into ordinary nodes first (expand_replicas): same results, old cost.
struct ReplicaGroup {
    uint32_t id = 0;
    int n_copies = 0, n_roots = 0, n_scalars = 0;
    int64_t id_base = 0;                        // copy j owns the ids [id_base + j·graph_size, +graph_size): its operation with recording index i
    int graph_size = 0;                         //   has the id id_base + j·graph_size + i, whether it ever becomes a node or not
    std::vector<Node*> roots;                   // the original's roots, or nullptr where the root was a vector already (each pending one holds one refs_ext of the group)
    std::vector<char>  root_done;               // per root: executed (its copies have their buffers)
    std::vector<Node*> leaf_from;               // substituted operands (each holds one refs_int)
    std::vector<Node*> leaf_to;                 // [copy][i] (each holds one refs_int)
    std::vector<int32_t> scalar_slot;           // per operation (recording index): index into a copy's scalar list, or -1
    std::vector<double> scalars;                // [copy][n_scalars]; empty = the original's
    std::vector<Node*> copy_roots;              // [copy][root]: pending nodes without an expression (each holds one refs_int), nullptr for a shared vector
    int remaining = 0;                          // roots not executed yet
};

// ---------------------------------------------------------------- compiled programs

struct SsaOp { int opcode; int a, b, c; double scalar; };

struct Program {
    int n_in = 0, n_out = 0, n_red = 0, n_ops = 0, n_scal = 1;
    DevProgramArgs proto{};                 // ops / out_reg / red_reg / counts filled in
    std::vector<float> scalars;             // default scalar operands (explicit API)
    int refs = 1;
    // execution tier (jit.hpp): work done on the interpreter so far, and the specialised kernel once requested
    // steady-state loops (run_into over the same vectors): the row table of the previous launch is still in the ring
    std::vector<uint64_t> last_table;       // its bytes
    const uint64_t* last_dev_rows = nullptr;
    uint64_t last_ring_generation = 0;
    double interpreted_work = 0.0;          // elements x micro-ops
    bool jit_probed = false;                // the caches have been asked for this program's kernel (once, at its first launch)
    std::shared_ptr<JitSlot> jit;
};


// Handle → node.  Handles are consecutive integers, most of them short-lived (a Monte-Carlo calibration hands out tens of millions):
// pages of 4096 entries indexed directly, allocated when the first handle of a page is issued and freed when its last one is gone —
// no hashing and no allocation per handle on the path of every recorded method (a hash map was a third of its 80 ns).
class HandleTable {
public:
    // (the top 16 bits of a handle name the engine that owns it — thread engines, FM_OWNER_SHIFT; the table of an engine is indexed by the rest)
    Node* get(int64_t id) const {
        if (id <= 0) return nullptr;
        id &= LOCAL;
        const size_t page = (size_t)(id >> BITS);
        if (page >= pages_.size() || !pages_[page].slot) return nullptr;
        return pages_[page].slot[(size_t)id & MASK];
    }
    void put(int64_t id, Node* nd) {
        id &= LOCAL;
        const size_t page = (size_t)(id >> BITS);
        if (page >= pages_.size()) pages_.resize(page + 1);
        Page& pg = pages_[page];
        if (!pg.slot) { pg.slot = new Node*[(size_t)1 << BITS](); pg.live = 0; }
        Node*& s = pg.slot[(size_t)id & MASK];
        if (!s) { ++pg.live; ++size_; }
        s = nd;
    }
    void erase(int64_t id) {
        if (id <= 0) return;
        id &= LOCAL;
        const size_t page = (size_t)(id >> BITS);
        if (page >= pages_.size() || !pages_[page].slot) return;
        Page& pg = pages_[page];
        Node*& s = pg.slot[(size_t)id & MASK];
        if (!s) return;
        s = nullptr; --size_;
        if (--pg.live == 0) { delete[] pg.slot; pg.slot = nullptr; }
    }
    size_t size() const { return size_; }
    template <typename F> void for_each(F&& f) const {
        for (const Page& pg : pages_) if (pg.slot) for (size_t i = 0; i < ((size_t)1 << BITS); ++i) if (pg.slot[i]) f(pg.slot[i]);
    }
    void clear() { for (Page& pg : pages_) { delete[] pg.slot; pg.slot = nullptr; pg.live = 0; } pages_.clear(); size_ = 0; }
    ~HandleTable() { clear(); }
    HandleTable() = default;
    HandleTable(const HandleTable&) = delete;
    HandleTable& operator=(const HandleTable&) = delete;
private:
    static constexpr int64_t LOCAL = (int64_t(1) << 48) - 1;
    static constexpr int BITS = 12;
    static constexpr size_t MASK = ((size_t)1 << BITS) - 1;
    struct Page { Node** slot = nullptr; uint32_t live = 0; };
    std::vector<Page> pages_;
    size_t size_ = 0;
};

// The engine's lock: recursive, and taken and released once per recorded method and once per release — forty million times per calibration of a
// caller without hints, practically never contended (waits for the device happen outside it).  An owner tag and a depth instead of a pthread
// recursive mutex; a thread that finds it taken spins briefly, then yields, then naps (a holder keeps it for microseconds, a flush for up to a
// millisecond).  FMHIP_ENGINE_MUTEX=pthread builds nothing else: the type is chosen at compile time below.
class EngineMutex {
public:
    void lock() {
        const uint64_t me = self();
        if (owner_.load(std::memory_order_relaxed) == me) { ++depth_; return; }
        uint64_t expected = 0;
        if (!owner_.compare_exchange_strong(expected, me, std::memory_order_acquire, std::memory_order_relaxed)) lock_slow(me);
        depth_ = 1;
    }
    bool try_lock() {
        const uint64_t me = self();
        if (owner_.load(std::memory_order_relaxed) == me) { ++depth_; return true; }
        uint64_t expected = 0;
        if (!owner_.compare_exchange_strong(expected, me, std::memory_order_acquire, std::memory_order_relaxed)) return false;
        depth_ = 1;
        return true;
    }
    void unlock() { if (--depth_ == 0) owner_.store(0, std::memory_order_release); }
private:
    static uint64_t self() { static std::atomic<uint64_t> next{ 1 }; static thread_local const uint64_t id = next.fetch_add(1); return id; }
    void lock_slow(uint64_t me);
    alignas(64) std::atomic<uint64_t> owner_{ 0 };
    int depth_ = 0;                              // (the owner's)
    alignas(64) char pad_[8] = { 0 };
public:
    EngineMutex() { (void)pad_; }
};

class Engine {
public:
    static Engine& get();
    static Engine* create();                 // a further engine (sharded.cpp: one per device of a device list); the caller owns it
    // Thread engines (fmhip_set_thread_engines): an engine per caller thread on one device.  index() is in the top 16 bits of every handle
    // this engine hands out; a vector of another engine enters this one as a LEAF that aliases the owner's storage (the owner keeps a
    // reference for it; the two streams are ordered by events in both directions).
    static constexpr int OWNER_SHIFT = 48;
    static int owner_of(int64_t handle) { return (int)((uint64_t)handle >> OWNER_SHIFT); }
    int index() const { return index_; }
    std::atomic<bool> retired{ false };                        // a thread engine after fmhip_shutdown: threads bound to it fall back to the process-wide engine
    void set_index(int i);                                     // before init(): the engine's number among the thread engines
    struct Exported { float* ptr = nullptr; int64_t n = 0; uint32_t bm_id = 0; int32_t bm_step = -1, bm_steps = 0; };
    Exported export_vector(fmhip_vec h, hipEvent_t ready);     // owner side: computed, one more reference, `ready` recorded on this stream
    fmhip_vec import_vector(int owner, fmhip_vec owner_handle, const Exported& x, hipEvent_t ready);      // this stream waits for `ready`; returns a handle of THIS engine (one reference)
    struct ForeignDone { int owner; fmhip_vec handle; hipEvent_t done; };
    fmhip_vec find_import(fmhip_vec foreign_handle);           // the import of that vector if one is alive here (one more reference), else 0
    bool has_foreign_done() const { return !foreign_done_.empty(); }
    std::vector<ForeignDone> take_foreign_done();              // imports that have died: their owners' references are to be given back (outside this engine's lock)
    void release_exported(fmhip_vec h, hipEvent_t done);       // owner side: this stream waits for `done`, then the reference goes
    static void bind_thread(Engine* e);      // Engine::get() of THIS thread returns e from now on (nullptr: the process-wide engine again)
    static bool thread_is_bound();
    EngineMutex mu;
    // Releases from a thread that is NOT driving the engine — a garbage collector's cleaner thread handing back, in bursts, the handles of
    // objects that died a while ago (java/net/finmath/hip/DeviceVector.java; the reference's ReferenceQueue, RandomVariableCuda.java:293-305)
    // — never take the engine lock one by one: a caller thread that records two million methods a second would meet the releaser on the
    // lock at every other call and sleep on it (measured, lmm_hip --finmath-like --release-lag 100, 1 M paths: 14.1 s against 4.3 s with
    // the same launches).  They are queued (a tiny lock of their own; abi.cpp collects 256 per releasing thread first) and PERFORMED by the
    // driving thread — preferably while it has nothing else to do: in small portions while it waits for the device (the spin loops of
    // slot_wait / ticket_take / red_wait), all of them before a flush looks for memory in vain (new_buffer), at the latest when LATE_EAGER
    // wait.  A node that was released 100 ms after its creation is cold: performing 21 M of them between the recorded methods cost the
    // hint-free calibration 1.5 s; the device gives that time away for nothing while it is waited for.  The releasing thread performs the
    // queue itself only when no thread has entered the engine for 10 ms (the caller has gone quiet).
    static size_t late_eager() { static const size_t v = [] { const char* e = std::getenv("FMHIP_LATE_EAGER"); return e ? (size_t)std::atoll(e) : (size_t(1) << 17); }(); return v; }      // queued releases at which the driving thread performs them at once
    static size_t late_portion() { static const size_t v = [] { const char* e = std::getenv("FMHIP_LATE_PORTION"); return e ? (size_t)std::atoll(e) : (size_t)48; }(); return v; }       // … and how many it performs per look while it waits for the device
    // (a count of entries, on a cache line of its own, written by the driving threads and read by a releasing thread once per 256 releases:
    // until it was — and while every release read the driving thread's tag, which every entry wrote — the line went back and forth between the
    // two cores with every recorded method: 0.10 → 0.25 µs per method for as long as a collector's burst lasted)
    void note_driver() { driver_seq_.store(driver_seq_.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed); }
    uint64_t driver_seq() const { return driver_seq_.load(std::memory_order_relaxed); }
    void release_later(const fmhip_vec* hs, size_t count) {
        std::lock_guard<std::mutex> lock(late_mu_);
        late_.insert(late_.end(), hs, hs + count);
        late_count_.store(late_.size() - late_pos_, std::memory_order_release);
    }
    bool has_late() const { return late_count_.load(std::memory_order_acquire) != 0; }
    size_t late_count() const { return late_count_.load(std::memory_order_relaxed); }
    void drain_late(size_t at_most = ~size_t(0));              // under `mu`
    // Callers that wait for moments WITHOUT the engine lock (abi.cpp: fmhip_reduce_moments, fmhip_reduce_moments_batch_end) hold slots,
    // pinned blocks and events of this engine meanwhile: counted here (under the lock, before it is dropped); fmhip_shutdown waits for zero.
    std::atomic<int> waits_in_flight{ 0 };

    void init(int device_index);
    void shutdown();
    bool initialized() const { return initialized_; }
    int device_index() const { return device_; }
    void require_init() const;
    void synchronize();
    hipStream_t stream() const { return stream_; }
    void device_info(char* name, int len, int* cus, int64_t* hbm);

    // vectors
    Node* node(fmhip_vec h);                                   // throws INVALID_HANDLE
    fmhip_vec create_from_host(const void* src, bool is_double, int64_t n);
    fmhip_vec create_filled(int64_t n, float v);
    fmhip_vec create_uninitialized(int64_t n);
    void retain(fmhip_vec h);
    void release(fmhip_vec h);
    void read(fmhip_vec h, void* dst, bool as_double, int64_t n);
    void* device_ptr(fmhip_vec h);

    // ops
    fmhip_vec call(int opcode, int n_in, const fmhip_vec* in, double scalar, bool has_scalar);
    bool fusion = false;
    int fusion_hold = 0;                    // fmhip_fusion_hold: 1 = no execution on the engine's own accord; 2 = the same, but everything pending is
                                            // executed once more than FUSION_SOFT_CAP operations wait (a hold somebody may forget to lift)
    int math_mode = FMHIP_MATH_EXACT;
    // Time-step grouping (fmhip_set_step_grouping): a discretisation scheme reads the Brownian increments of time index i exactly
    // while it computes step i — the one place where a caller that knows nothing about this engine (finmath-lib's Euler scheme)
    // shows where a time step ends.  With group_steps = S > 0 the methods recorded between S such boundaries stay pending (like
    // a soft hold) and are executed together: the engine sees S whole time steps at once, schedules them component by component
    // and runs the periodic stretch as one rolled-loop launch, instead of cutting the stream every ≈ 40 methods.
    // expectation communicator (fmhip_set_expectation_comm): applied to host-side moments in abi.cpp
    int comm_world = 1, comm_rank = 0;
    fmhip_gather_fn comm_gather = nullptr;
    void* comm_context = nullptr;
    int group_steps = 4;
    // Launches with a fused reduction of at most this many spans in all give every workgroup one UNIT of the reduction tree instead of a
    // span (runtime.cpp: launch); FMHIP_UNIT_WORKGROUPS.
    // (512 since round 5: two or three rows of 1 M paths — 246 / 369 spans — used to fall between "one row: a unit per workgroup" and "four rows and
    // more: enough spans", took no moments along and cost a reduction launch each: 6 210 of the hint-free calibration's 43 471 launches)
    int64_t unit_workgroups_ = 512;
    bool unit_launch(int64_t n, int64_t batch) const;
    void end_step_group() { group_hold_ = false; group_steps_pending_ = 0; }      // a value is read, or the caller flushes: whatever was being grouped has run
    void flush_all();
    void materialize(const std::vector<Node*>& targets);
    void graph_clone(const fmhip_vec* roots, int n_roots, int n_copies, const fmhip_vec* leaf_from, const fmhip_vec* leaf_to, int n_map,
                     const double* scalars, int n_scalars, fmhip_vec* out);
    int graph_scalars(const fmhip_vec* roots, int n_roots, double* out, int capacity);

    // reductions
    // what a launch with fused reductions holds while it runs (runtime.cpp: red_begin / red_wait / red_release)
    struct RedLaunch { void* partials = nullptr; size_t partials_cap = 0; void* results = nullptr; size_t results_cap = 0; void* dev_moments = nullptr;
                       bool on_host = false; volatile uint64_t* poll_flag = nullptr; uint64_t done_value = 0; int slot = -1;
                       bool pending = false; int batch = 0, n_red = 0; fmhip_moments* host = nullptr; };       // pending: launched, not yet waited for (defer_red_)
    // hand_over: a caller that can wait WITHOUT the engine lock (abi.cpp) receives the launch whose moments are still on their way
    // (pending, slot >= 0: results and completion flag in a slot of their own in pinned memory) instead of having reduce() wait for
    // it: red_poll() without the lock, red_complete() with it.
    void reduce(fmhip_vec h, double shift, fmhip_moments* host_out, void* dev_out, RedLaunch* hand_over = nullptr);
    static bool red_poll(const RedLaunch& red);              // spins on the flag (≤ 2 ms); true = the moments have arrived
    void red_complete(RedLaunch& red, bool arrived);         // copies the moments out (waits for the stream first if they have not arrived), releases
    void reduce_batch(const fmhip_vec* hs, int count, const double* shifts, fmhip_moments* host_out, void* dev_out);
    // fmhip_reduce_moments_batch_begin / _end: the moments land in a block of pinned memory of the ticket's own, an event behind the
    // launch tells when (waited for WITHOUT the engine lock: abi.cpp).
    struct MomentsTicket { void* host = nullptr; size_t cap = 0; hipEvent_t event = nullptr; int count = 0;
                           // … or, when the launches that computed the vectors took their moments along (reduce_batch_begin on pending vectors):
                           // the slots of the pinned arena they arrive in (nullptr: in `ready` already) — no launch, no block, no event
                           std::vector<volatile uint64_t*> slots; std::vector<fmhip_moments> ready; };
    int64_t reduce_batch_begin(const fmhip_vec* hs, int count, const double* shifts);
    int64_t reduce_batch_begin_from_launches(const fmhip_vec* hs, int count);
    void reduce_batch_device(const fmhip_vec* hs, int count, const double* shifts, void* dev_out);
    void reduce_batch_device_from_launches(const fmhip_vec* hs, int count, void* dev_out);   // the same moments into a device buffer (an RCCL send buffer): gathered in-stream from their arena slots
    void give_up_values(const fmhip_vec* hs, int count);
    MomentsTicket ticket_take(int64_t id);                   // removes it from the table (under the lock)
    void ticket_retire(MomentsTicket& t);                    // block and event back to their free lists (under the lock)

    // programs
    fmhip_program program_create(const fmhip_prog_op* ops, int n_ops, int n_in, const int32_t* outs, int n_out,
                                 const int32_t* reds, int n_red);
    void program_release(fmhip_program p);
    std::string program_source(const fmhip_prog_op* ops, int n_ops, int n_in, const int32_t* outs, int n_out, const int32_t* reds, int n_red);
    int jit_mode = FMHIP_JIT_AUTO;
    void jit_wait() { jit().wait_idle(); }
    JitStats jit_stats() { return jit_shared_ ? JitStats() : jit_.stats(); }      // (a thread engine's kernels are counted where they live: in the first engine)
    void share_jit_of(Engine& first) { jit_shared_ = &first.jit_; }     // before init(): thread engines on one device load every code object once
    Jit& jit() { return jit_shared_ ? *jit_shared_ : jit_; }
    int64_t jit_launches() const { return n_jit_launches_; }
    int64_t rolled_launches() const { return n_rolled_launches_; }
    int64_t algorithmic_bytes() const { return algorithmic_bytes_; }
    void engine_stats(fmhip_engine_stats_t* out);
    Program* program(fmhip_program p);
    void program_run(fmhip_program p, int batch, const fmhip_vec* inputs, fmhip_vec* outputs, bool into,
                     const double* shifts, fmhip_moments* moments, void* dev_moments);

    // brownian increments
    void bm_generate(int64_t seed, int n_steps, int n_factors, int64_t n_paths, int64_t path_offset,
                     const double* dt, fmhip_vec* out);

    // pool
    void pool_clean();
    void pool_purge();
    void pool_stats(fmhip_pool_stats_t* out);

    // measurement
    void profile_enable(bool on);
    bool profiling() const { return profiling_; }
    void profile_read(double* ms_total, int64_t* n);

private:
    Engine() { pend_clear(); }
    bool initialized_ = false;
    int device_ = -1;
    hipStream_t stream_ = nullptr;
    Pool pool_;
    Jit jit_;
    Jit* jit_shared_ = nullptr;
    int64_t next_id_ = 1;
    int index_ = 0;
    struct Foreign { int owner = 0; fmhip_vec handle = 0; bool live = false; };
    std::vector<Foreign> foreign_;                               // imports alive in this engine (Buffer::foreign indexes it)
    std::vector<int32_t> foreign_free_;
    std::unordered_map<fmhip_vec, Node*> import_of_;            // foreign handle → its live import in this engine
    std::vector<ForeignDone> foreign_done_;
    HandleTable nodes_;
    Node pending_head_;                                          // circular list of the nodes without storage (lazy expressions)
    void pend_insert(Node* nd) { nd->pend_prev = pending_head_.pend_prev; nd->pend_next = &pending_head_; pending_head_.pend_prev->pend_next = nd; pending_head_.pend_prev = nd; ++n_pending_; }
    // (off whichever list it is on: the pending list or — Node::deferred — the deferred list)
    void pend_erase(Node* nd) { if (!nd->pend_next) return; nd->pend_prev->pend_next = nd->pend_next; nd->pend_next->pend_prev = nd->pend_prev; nd->pend_prev = nd->pend_next = nullptr;
                                if (nd->deferred) { nd->deferred = false; --n_deferred_; } else --n_pending_; }
    void pend_clear() { pending_head_.pend_prev = pending_head_.pend_next = &pending_head_; n_pending_ = 0; deferred_head_.pend_prev = deferred_head_.pend_next = &deferred_head_; n_deferred_ = 0; }
    size_t n_pending_ = 0;                      // nodes on the pending list
    std::vector<Node*> node_pool_;                               // recycled Node objects
    std::unordered_map<int64_t, Program*> programs_;
    std::unordered_map<std::string, Program*> program_cache_;    // lazy front-end, keyed by structure
    int64_t n_launches_ = 0, n_ops_executed_ = 0, n_jit_launches_ = 0, algorithmic_bytes_ = 0;
    uint64_t epoch_ = 0;
    bool profiling_ = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> profile_events_;
    struct ProfileTag { int n_ops, n_in, n_out, n_red, batch, tier; int64_t n; };     // what each recorded launch was
    std::vector<ProfileTag> profile_tags_;

    // pinned staging for H2D/D2H and the row-table ring
    void*  stage_ = nullptr;  size_t stage_cap_ = 0;
    // results + completion flag of single-row reductions wanted on the host: 128-byte slots in pinned memory, one per launch in flight
    char* result_slots_ = nullptr; std::vector<int> free_slots_;
    std::unordered_map<int64_t, MomentsTicket> tickets_; int64_t next_ticket_ = 1;
    std::vector<MomentsTicket> free_tickets_;                // retired tickets: their pinned block and event are used again
    static constexpr int RESULT_SLOTS = 64;
    void*  ring_host_ = nullptr; void* ring_dev_ = nullptr; size_t ring_cap_ = 0, ring_off_ = 0;
    uint64_t ring_generation_ = 1;          // bumped on every wrap: device copies of older tables may be overwritten
    uint32_t* counters_dev_ = nullptr;      // arrival counters of the fused final combine (65536 rows, zero between launches)
    uint64_t poll_sequence_ = 0;            // value of the completion flag of the last polled reduction (see launch)
    void*  dump_dev_ = nullptr;             // FM_DUMP_BYTES nobody reads: target of the stores of lanes past the end of a vector (rolled kernels)
    void*  ensure_stage(size_t bytes);
    size_t ring_reserve(size_t bytes);

    Node* new_node(int64_t n);
    void collect_pending(const fmhip_vec* roots, int n_roots, std::vector<Node*>& graph);      // pending nodes below the roots, in recording order
    Buffer* new_buffer(int64_t n_floats);
    void buffer_unref(Buffer* b);
    void node_unref_int(Node* nd);
    void node_maybe_free(Node* nd);
    void drop_expression(Node* nd);

    Program* compile(const std::vector<SsaOp>& ops, int n_in, const std::vector<int>& outs, const std::vector<int>& reds,
                     std::vector<float>* scalars_out, bool fixed_scalars);
    Program* compile_variant(const std::vector<SsaOp>& ops, int n_in, const std::vector<int>& outs, const std::vector<int>& reds,
                             std::vector<float>* scalars_out, int variant, bool fixed_scalars);
    // rows: per batch row, n_in input buffers + n_out output buffers (+ per-row scalars, shifts)
    struct RowSpec { std::vector<const float*> in; std::vector<float*> out; const float* scalars; const double* shifts; };
    void launch(Program* p, int64_t n, const std::vector<RowSpec>& rows, fmhip_moments* host_moments, void* dev_moments);
    Program* reduce_program();
    bool try_fused(const std::vector<Node*>& roots);
    struct Dag;
    struct BigDag;
    // What it takes to run a component of a known shape again: per segment the program and where its row block comes from.
    // Indices >= 0 address BigDag::order, indices < 0 the leaves (-1 - index).  Built the first time a shape is seen (by the
    // general path: segment search, Dag per segment, signature strings), then every later component of that shape — the next
    // Euler steps, the other parameter sets of a Jacobian batch, the next objective evaluation — is executed straight from
    // the plan: no Dag, no strings, no hash lookups per segment (the general path cost ≈ 16 µs of host time per segment and
    // member, which made two-step groups of the LMM simulation host-bound).
    struct BigPlan {
        struct Seg { Program* prog = nullptr; std::vector<int32_t> in, out, scal; int zone = 0;       // zone: 0 before a rolled loop, 1 inside (its fallback), 2 behind
                     std::vector<int32_t> free_after;                                                 // values no later segment reads (members without nodes release them here)
                     // the LAST segment only: what it takes to compile the variant that also reduces the component's root (reduce())
                     std::vector<SsaOp> ssa; std::vector<int> out_ids; int n_in = 0; Program* prog_red = nullptr; bool no_red = false; };
        std::vector<Seg> segs;
        std::string sig;            // the shape the plan was made for (the cache is keyed by its hash)
        bool segs_missing = false;  // the shape has only run as ONE launch of its peeled kernel so far: `segs` is cut when that launch is not available (run_plan)
        bool discards_root = false; // the component's root is wanted for its moments only (Node::discard): the peeled kernels of this plan do not store it
        // A periodic stretch of the scheduled order — the same few operations over one component after another, each iteration
        // feeding the next (a running sum) — as ONE launch of a kernel that loops over the iterations (runtime.cpp: rolled loops).
        // Positions are offsets inside one iteration; the loop covers order[begin + r·period + q], r < iterations.
        struct Rolled {
            bool present = false;
            std::shared_ptr<JitSlot> jit;                               // requested when the JIT tier is on (at discovery, or at the first later use)
            std::string source; int elems = 0;                          // the kernel's source text (kept until it has been handed to the compiler)
            uint32_t begin = 0, period = 0, iterations = 0, row_words = 0;
            std::vector<int32_t> global_leaf;                           // indices into BigDag::leaves: inputs every iteration reads
            std::vector<uint32_t> carried;                              // positions whose value of the PREVIOUS iteration is read
            std::vector<std::pair<uint32_t, uint32_t>> leaf_in;         // (position, operand) of the first use of each per-iteration input
            std::vector<int32_t> iter_leaf;                             // [iteration][input]: which leaf (index into BigDag::leaves) that is
            // PEELED form (jit.hpp: RolledBody::Peel): the whole component — the operations in front of the loop, the loop, the ones
            // behind it — as ONE launch; used for launches of few workgroups, where a launch costs more than the bytes it moves
            struct Peeled {
                bool present = false;
                std::shared_ptr<JitSlot> jit;
                std::string source; int elems = 0;
                uint32_t row_words = 0, n_pre_scal = 0, n_post_scal = 0, n_ops = 0;
                std::vector<int32_t> extra_leaf;                        // leaves read by the operations outside the loop (x0, x1, …)
                std::vector<uint32_t> pre_out, post_out;                // order positions stored from in front of / behind the loop
                std::vector<uint32_t> pre_scal, post_scal;              // order positions of the scalar-carrying operations there
                std::vector<char> final_store;                          // per final value: stored?
                // the variant that also takes the moments of the component's root (RolledBody::Peel::reduce); compiled when first asked for
                std::string source_red; std::shared_ptr<JitSlot> jit_red;
                std::string desc_red;                                   // its one-line description (jit.hpp): what the merged form (merge_families) is derived from
                int mergeable = -1;                                     // -1: not looked at yet; 0 / 1: this shape may be a chain of a merged launch
            } peeled;
            std::vector<uint32_t> out_pos, scal_pos;                    // positions stored per iteration; positions with a scalar operand
            std::vector<uint32_t> final_pos;                            // positions whose value of the LAST iteration is stored behind the loop
        } rolled;
    };
    bool build_big(const std::vector<Node*>& roots, BigDag& big);
    // build_big: how a shape (the signature of the walk) was scheduled when it was first seen
    // (signatures here are STRUCTURAL — opcodes and operands, never who holds a handle: the same memo serves a caller that frees its
    // temporaries at the end of the statement and one whose garbage collector frees them a hundred milliseconds later.)  The flags — which
    // values are stored — are decided per occurrence (escape policy); the signature and hash that go with the flags seen last are kept.
    struct ShapePolicy { uint32_t base = 0, size = 0, gen = 0; uint64_t flush = 0; std::vector<uint8_t> decided; };
    struct ScheduleMemo {
        std::string walk_sig, sched_sig; std::vector<uint32_t> perm; uint64_t sched_hash = 0; ShapePolicy policy;
        struct Variant { std::vector<char> flags; std::string sig; uint64_t hash = 0; };       // the same structure with these values stored: signature and hash of the component shape
        std::vector<Variant> variants; size_t last_variant = 0;
    };
    std::unordered_map<uint64_t, ScheduleMemo> schedule_cache_;
    size_t schedule_cache_bytes_ = 0;
    std::string walk_sig_;
    // ---- escape policy: which values of a fused launch are STORED.
    // A value somebody outside the component needs (a pending consumer elsewhere, a root of the flush) is stored — that is a fact.  A value
    // whose only claim to storage is a live HANDLE is a guess about the caller's future: under a garbage-collected caller (the JVM the
    // reference lives in: RandomVariableCuda.java:96-106, 293-305) the temporary of `x.add(y).mult(z)` still has its handle when the flush
    // comes, and storing by handles means storing every operation's result.  So the engine LEARNS, per position of a component shape,
    // from what happens to such handles: stored for observation the first time (a caller that frees temporaries promptly: what it still
    // holds, it probably wants), found untouched when the shape comes round again → not stored from then on (deferred: Node::deferred),
    // touched after all → computed from its recipe, stored, and that position is stored for good (each position errs at most once).  A
    // component in which most internally consumed values still carry handles comes from a caller that does not free temporaries at all:
    // there the first guess is "not needed".  FMHIP_ESCAPE_POLICY=0: every handle is stored (rounds 1–4).
    enum : uint8_t { POLICY_NEW = 0, POLICY_OBSERVING = 1, POLICY_STORE = 2, POLICY_DEFER = 3 };
    std::vector<uint8_t> policy_state_;
    uint32_t policy_gen_ = 1;
    uint64_t flush_seq_ = 0;                                      // one number per flush / materialisation: decisions are taken once per position and flush
    std::unordered_map<std::string, ShapePolicy> dag_policies_;   // single-launch components, by structural signature
    void policy_bind(ShapePolicy& p, size_t size);                // (re)allocates the slots of a shape for the current generation and flush
    bool policy_store(ShapePolicy& p, size_t pos, bool optimistic);   // the decision for a handle-only value at that position (once per flush)
    void policy_reset();                                          // forgets everything learnt (caches dropped: pool_purge, a full schedule cache)
    // the caller uses this handle again (an operand, a read, an expectation): its position is stored from now on; a value that was left unstored is wanted after all
    void touch(Node* nd) { if (nd->watch) { if (nd->watch_gen == policy_gen_) policy_state_[nd->watch - 1] = POLICY_STORE; nd->watch = 0; } if (nd->deferred && !nd->buf) ++n_demanded_; }
    void defer_node(Node* nd, const ShapePolicy* p, size_t pos);  // off the pending list, onto the deferred list; watched
    void watch_node(Node* nd, const ShapePolicy& p, size_t pos) { nd->watch = p.base + (uint32_t)pos + 1; nd->watch_gen = p.gen; }
    void demand(Node* nd);                                        // a deferred value is wanted after all: computed from its recipe and stored
    void materialize_deferred();                                  // every deferred value that still has a handle (fmhip_pool_clean; before a vector is written in place)
    std::mutex late_mu_;
    std::vector<fmhip_vec> late_;
    size_t late_pos_ = 0;                                      // late_[late_pos_ …) wait
    std::atomic<size_t> late_count_{ 0 };
    alignas(64) std::atomic<uint64_t> driver_seq_{ 0 };
    alignas(64) std::atomic<uint64_t> after_driver_seq_{ 0 };      // (keeps what follows off that line)
    Node deferred_head_;                                          // circular list of the deferred nodes (Node::pend_prev / pend_next)
    size_t n_deferred_ = 0;
    int64_t n_deferred_total_ = 0, n_demanded_ = 0, bytes_written_ = 0, n_interpreter_launches_ = 0;
    int64_t n_late_waiting_ = 0, n_late_at_once_ = 0, late_ns_ = 0;
    void commit_node(Node* nd, Buffer* b) { nd->buf = b; pend_erase(nd); }       // a launch has stored this value
    bool segment_dag(const BigDag& big, size_t s, size_t e, Dag& dag, const std::vector<int32_t>& uses);
    // An expectation asked of a large pending expression: taken by the launch that computes its root (the last segment of its plan)
    struct ReduceRequest { double shift; fmhip_moments* host_out; void* dev_out; bool done; };
    // reduce() of a pending expression: the launch that takes the moments leaves its wait to reduce() — the bookkeeping behind the launch
    // (outputs become vectors, expressions are dismantled) happens while the device works, not after it
    RedLaunch* defer_red_ = nullptr;
    // reduce() with much other work pending: the flush that follows takes the moments of EVERY pending root in the launches that compute
    // them (rows of the same launches) and leaves them with the nodes — the next getAverage() calls find them there
    bool want_root_moments_ = false;
    // … and a flush the engine makes on its own while a caller is still recording (call(): many operations pending under the engine's
    // step-group hold) takes them along WITHOUT waiting: the launches write into slots of a pinned arena, the nodes remember their slot,
    // reduce() waits for a slot when its vector is asked for.  A slot is {Σ, Σ², min, max} as bit patterns, preset to a NaN payload
    // no reduction produces (results are canonicalised); the arena is reused after a stream synchronisation that first collects what
    // is still outstanding.
    bool async_moments_ = false;
    size_t ops_since_boundary_ = 0;                              // methods recorded since the last time-step boundary (step_boundary)
    char* moments_arena_ = nullptr; size_t arena_off_ = 0;
    size_t ARENA_BYTES = size_t(16) << 20;                        // 512 k slots: a wrap waits for the stream, so it should be rare (a 1 M-path calibration uses 89 k); FMHIP_ARENA_BYTES: tests shrink it
    static constexpr uint64_t MOMENTS_SENTINEL = 0x7ff8dead0000beefull;
    std::vector<std::pair<int64_t, volatile uint64_t*>> arena_outstanding_;
    double* arena_alloc(size_t count);                           // count slots, preset; may synchronise the stream (arena full)
    void arena_collect();                                        // stream synchronised: every outstanding slot goes to its node (if it still exists)
    void arena_assign(Node* nd, double* slot);
    void wait_for_stream(const char* what);                      // hipStreamSynchronize, with queued releases performed meanwhile
    bool slot_wait(Node* nd);                                    // the node's slot has arrived (or the stream is waited for): moments into the node
    void red_begin(RedLaunch& red, int batch, int n_red, size_t blocks_per_row, fmhip_moments* host_moments, void* dev_moments);
    void red_wait(RedLaunch& red, int batch, int n_red, fmhip_moments* host_moments);
    void red_release(RedLaunch& red);
    void run_big_group(std::vector<BigDag>& group, ReduceRequest* rr = nullptr);
    void plan_segments(BigPlan& plan, std::vector<BigDag>& group);
    void plan_loop(BigPlan& plan, const BigDag& g);
    void run_planned_segment(const BigPlan::Seg& seg, std::vector<BigDag>& group, size_t first, size_t count, ReduceRequest* rr = nullptr, Program* prog_red = nullptr);
    void run_plan(BigPlan& plan, std::vector<BigDag>& group, ReduceRequest* rr = nullptr);
    Node* single_root(const BigDag& b, const BigDag& g0);
    void commit_described(BigDag& big, size_t pos, Buffer* b);
    bool detect_loop(const BigDag& g, const std::vector<std::array<int32_t, 3>>& operand, BigPlan::Rolled& out, std::string* source, int* elems, RolledBody* body_out = nullptr);
    void run_rolled(const BigPlan::Rolled& ro, std::vector<BigDag>& group, size_t first, size_t count);
    bool plan_peel(const BigDag& g, const std::vector<std::array<int32_t, 3>>& operand, BigPlan::Rolled& ro, const RolledBody& body);
    // row_of (optional): per member, the row of the launch that computed it.  Members whose row would be IDENTICAL — the same input
    // vectors and the same scalars — are computed once and share the stored vectors (common rows: the parameter sets of a Jacobian batch
    // before the time their bumped parameter first matters); with a fused reduction only when the caller can take the mapping.
    void run_peeled(const BigPlan::Rolled& ro, std::vector<BigDag>& group, size_t first, size_t count, ReduceRequest* rr = nullptr, std::vector<uint32_t>* row_of = nullptr);
    void make_private(Node* nd);                              // a vector about to be written in place: storage of its own if it shares it (common rows)
    int64_t n_common_rows_ = 0;
    int64_t n_rolled_launches_ = 0;
    // Components of ONE loop shape whose vectors are the same sequence, the shorter ones reading a suffix of the longest one's (the swaptions
    // of one exercise date: every tenor reads the forward rates from its last period back to the exercise date), as ONE launch that loads
    // every vector once (jit.hpp: RolledBody::chains).  Looked for among the large components of a flush that takes the moments of all
    // pending roots along; whatever does not fit — a kernel not compiled yet, a family of one — runs as before.  FMHIP_MERGE_CHAINS=0: off.
    // A component small enough for one launch of its own (a Dag: the swaptions of few periods) may be a chain of such a family too: its
    // operations are, position by position, the head of a mergeable shape, R >= 0 iterations of its body and its tail (match_small;
    // remembered per Dag signature).  Groups of such components wait for the families (SmallGroup) instead of running at once.
    struct SmallMatch { bool ok = false; int shape = -1; uint32_t R = 0; std::vector<uint16_t> seq_leaf, post_leaf; };
    struct SmallGroup;
    void merge_families(std::vector<std::vector<BigDag>>& groups, std::vector<SmallGroup>& small);
    const SmallMatch* match_small(const Dag& d);
    int merge_shape_index(const std::string& desc);
    std::vector<std::string> merge_shapes_;                                          // descriptions (desc_red) of the mergeable loop shapes met so far …
    std::vector<RolledBody> merge_shape_bodies_;                                     // … and their bodies
    std::unordered_map<std::string, SmallMatch> small_match_;
    std::unordered_map<std::string, std::shared_ptr<JitSlot>> merged_kernels_;        // by description of the merged body
    int64_t n_merged_launches_ = 0, n_merged_chains_ = 0;
    // replica groups: live descriptions by id (ids are never reused: a stale stamp on a recycled node finds nothing)
    std::unordered_map<uint32_t, ReplicaGroup*> replicas_;
    uint32_t next_replica_id_ = 1;
    // time-step grouping: the generation and time index of the increment seen last, steps recorded since the last flush, whether
    // the engine itself is holding pending work back for a group
    uint32_t next_bm_id_ = 1, group_bm_id_ = 0;
    int32_t  group_last_step_ = -1;
    std::unordered_map<uint32_t, int32_t> group_last_by_bm_;     // per generation: the time index of its increment seen last (step_boundary)
    int      group_steps_pending_ = 0;
    bool     group_hold_ = false;
    void step_boundary(const Node* increment);
    ReplicaGroup* replica_of(const Node* nd) const { if (!nd->rep_id) return nullptr; auto it = replicas_.find(nd->rep_id); return it == replicas_.end() ? nullptr : it->second; }
    void expand_replicas(ReplicaGroup* g);                       // the description becomes ordinary pending nodes (fallback; also frees the group)
    void expand_replicas_below(const std::vector<Node*>& targets);   // every group the pending graph below `targets` touches
    void replica_roots_done(ReplicaGroup* g, const std::vector<int>& roots);   // those roots have been executed with all their copies
    void destroy_replica_group(ReplicaGroup* g);
    void replicas_after_failure(const std::vector<std::pair<ReplicaGroup*, std::vector<int>>>& done);   // a launch sequence over these roots threw midway
    struct ReplicaView;
    std::unordered_map<uint64_t, BigPlan> plan_cache_;                        // component shape -> segments, programs and row-block sources
    bool build_dag(const std::vector<Node*>& roots, Dag& dag);
    // reduce_shift != nullptr (one DAG only): the root is ALSO reduced in the same launch — {Σ, Σ(x-shift)², min, max} into host_moments / dev_moments
    bool run_dags_plain(std::vector<Dag>& dags, const Dag* proto);
    bool run_dags(std::vector<Dag>& dags, const double* reduce_shift = nullptr, fmhip_moments* host_moments = nullptr, void* dev_moments = nullptr, const Dag* proto = nullptr);
    Dag replica_dag(const Dag& d, ReplicaGroup* g, int copy);
    ReplicaGroup* clean_replica_group(uint32_t rep_id, bool uniform) const { if (!rep_id || !uniform) return nullptr; auto it = replicas_.find(rep_id); return it == replicas_.end() ? nullptr : it->second; }
};

void hip_check(hipError_t e, const char* what);

// Host-side time accounting of the front-end (FMHIP_HOST_PROFILE=1 prints the table at shutdown): where the wall time of
// a launch-bound caller (the LMM calibration: 31 000 method calls and 700 launches per objective evaluation) goes.
struct HostProfile {
    enum Slot { CALL, RELEASE, FLUSH, BUILD_DAG, RUN_DAGS, LAUNCH, LAUNCH_API, ROW_UPLOAD, REDUCE, FLUSH_COMPONENTS, BUILD_BIG, RUN_BIG, CLONE, N_SLOTS };
    bool on = false;
    double seconds[N_SLOTS] = { 0 };
    long long count[N_SLOTS] = { 0 };
    void report() const;
};
extern HostProfile g_host_profile;
struct HostTimer {
    HostProfile::Slot slot; bool on; std::chrono::steady_clock::time_point t0;
    explicit HostTimer(HostProfile::Slot s) : slot(s), on(g_host_profile.on) { if (on) t0 = std::chrono::steady_clock::now(); }
    ~HostTimer() { if (on) { g_host_profile.seconds[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); g_host_profile.count[slot]++; } }
};

} // namespace fm
