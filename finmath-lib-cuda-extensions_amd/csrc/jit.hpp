// jit.hpp — second execution tier of compiled programs: a straight-line gfx950 kernel per program, generated from the
// micro-op stream and compiled at run time with hiprtc on a background thread.
//
// Why a second tier.  The interpreter (kernels.hip) starts instantly and reaches the HBM streaming ceiling for light
// programs, but for arithmetic-heavy streams it pays for its generality: the virtual register file pins 72-120 VGPRs
// (3 waves per SIMD with a fused reduction), every micro-op costs a scalar decode + s_set_gpr_idx window, and nothing can
// be scheduled across micro-op boundaries.  A specialised kernel has none of that.  Both tiers are built from the SAME
// device functions (fm_device_math.hpp, fm_kernel_parts.hpp) with the same floating-point flags, so they are bit-identical
// and a program can move from one to the other between two launches.
//
// The reference has no counterpart: it launches one precompiled PTX kernel per method (RandomVariableCuda.java:539-557).
#pragma once
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "fm_program.h"

namespace fm {

struct JitSlot {
    enum State : int { QUEUED = 1, READY = 2, FAILED = 3 };
    std::atomic<int> state{ QUEUED };
    std::string name, source, log;
    hipModule_t module = nullptr;
    hipFunction_t fn_inline = nullptr, fn_table = nullptr;      // row block in the kernarg segment / in the row table
    int vgprs = 0;
    int elems = 0;                                              // elements per lane and pass of this kernel (the host tiles the launch with it)
    bool from_disk = false;                                     // code object came from the persistent cache
    double compile_seconds = 0.0;
};

struct JitStats { int64_t compiled = 0, failed = 0, pending = 0, disk_hits = 0; double seconds = 0.0; };

// Shape of a specialised kernel: elements per lane and pass, how many elements of an exp / log body are interleaved,
// an occupancy hint for the register allocator, software prefetch of the next pass.
struct JitShape { int elems = 8, group = 4, waves = 0; bool prefetch = true; bool deep = false; };      // deep: the loads of ALL passes of a workgroup issued at once
JitShape jit_shape(const DevProgramArgs& proto);

// Source text of the specialised kernel pair of a program (deterministic: it doubles as the cache key).
std::string jit_generate_source(const DevProgramArgs& proto);

// Kernel pack: programs of known workloads, compiled at BUILD time (hiprtc needs no device) into <library directory>/jit_pack, a
// read-only second level behind the user's code-object cache — a fresh machine starts those workloads on specialised kernels
// instead of spending its first seconds on the interpreter tier while ~25 kernels compile one after the other (hiprtc
// serialises concurrent compilations).  A pack entry is the one-line description of a program (everything the source generator
// reads); FMHIP_JIT_RECORD=<file> appends the description of every program a run asks the tier for (csrc/kernel_pack.txt is
// such a recording, benchmarks/record_kernel_pack.sh makes it).  An entry that no longer matches what the engine generates is
// simply never looked up.
std::string jit_describe(const DevProgramArgs& proto);
bool jit_parse_description(const std::string& line, DevProgramArgs& proto);

// Body of a rolled loop (runtime.cpp: detect_loop): everything the source of its kernel depends on.  Operand names: v<q> = value of
// position q of this iteration, c<k> = k-th value carried over from the previous iteration, g<k> = k-th loop-invariant input vector,
// l<m> = m-th input vector of this iteration.  Iteration count, pointers and scalar operands are run-time arguments (fm_program.h).
struct RolledBody {
    int elems = 8; bool uses_log = false;
    uint32_t globals = 0, inputs = 0;
    std::vector<uint32_t> carried, final_pos, out_pos;                 // positions whose value is carried / stored once behind the loop / stored every iteration
    struct Op { uint32_t uop; std::string x0, x1, x2; bool scalar; };  // x1 / x2 empty: the micro-op does not read them
    std::vector<Op> ops;
    // PEELED form: the operations in front of the loop and behind it run in the SAME launch (runtime.cpp: plan_peel) — a component
    // that is a short head, a periodic stretch and a short tail (a swaption's backward induction with its payoff) is ONE launch
    // instead of three, and the values between the parts stay in registers.  Further operand names: p<i> = result of operation i
    // in front of the loop, q<i> = of operation i behind it, x<k> = k-th extra input vector (the first extra_pre of them are loaded
    // in front of the loop, the others behind it), F<k> = k-th final value of the loop.  The carried values start from
    // carried_init[k] (a p-name) instead of being loaded.
    struct Peel {
        bool present = false;
        std::vector<Op> pre, post;
        uint32_t extra_pre = 0, extra_post = 0;
        std::vector<std::string> carried_init;
        std::vector<uint32_t> pre_out, post_out;                        // operations (indices into pre / post) whose result is stored
        std::vector<uint32_t> final_store;                              // per final value: 1 = stored (somebody outside the component reads it)
        // The variant that also takes the moments of one of its values (the component's root: `chain.getAverage()`): "q<i>" = result of
        // post[i], "F<k>" = final value k; empty = none.  8 elements per lane only: a workgroup's tile is then one unit of the reduction
        // tree (fm_kernel_parts.hpp), so the moments are those of the stand-alone reduction to the last bit.
        std::string reduce;
    } peel;
    // MERGED form (chains >= 2; runtime.cpp: merge_families): ONE launch serves `chains` components of this very shape whose inputs —
    // the head's own vectors, then the loop's, one per step — are the same vectors, the shorter components reading a SUFFIX of the longest
    // one's sequence (the swaptions of one exercise date, tenor by tenor: each reads the forward rates from its last period back to the
    // exercise date).  A step loads its vector once and every chain that has started takes its turn on it: the head's operations in the
    // chain's first steps (one stage per head input), the loop body afterwards; the tail and the moments of every chain behind the loop.
    // Preconditions (the host checks them, the generator returns an empty string otherwise): peeled with a reduction, 8 elements per
    // lane, no loop-invariant inputs, one input and no output per iteration, nothing stored by the head, final values = carried values.
    // shared_den: every `discount(·, <the step's vector>, s)` of head and body carries the same scalar in all chains (the host compares
    // them launch by launch): denominator, reciprocal and Newton step are computed once per step (fm_device_math.hpp: DivPrepared).
    uint32_t chains = 0;
    bool shared_den = false;
};
// Where the scalars that shared_den needs to be equal sit: indices into the head's scalar list and into one iteration's.
void jit_merged_shared_scalars(const RolledBody& body, std::vector<uint32_t>& pre_slots, std::vector<uint32_t>& body_slots);
std::string jit_generate_rolled_source(const RolledBody& body);
std::string jit_describe(const RolledBody& body);
bool jit_parse_description(const std::string& line, RolledBody& body);
// Compiles `source` for gfx950 WITHOUT a device and stores the code object under `dir` with the key the run-time lookup uses.
bool jit_precompile(const std::string& source, const std::string& dir, std::string* log);

class Jit {
public:
    ~Jit();                                                     // static destruction: join only, no HIP calls (fmhip_shutdown unloads)
    void start(int device);
    void stop();                                               // joins the worker, unloads every module
    void quiesce();                                            // joins the worker only (process exit: no HIP calls)
    // Returns the (shared) slot of this program; compiles synchronously when `sync`, else queues it for the worker.
    std::shared_ptr<JitSlot> request(const DevProgramArgs& proto, bool sync);
    // The same for a kernel pair given as source text (rolled loops, runtime.cpp): `elems` = elements per lane and pass.
    std::shared_ptr<JitSlot> request_source(std::string source, int elems, bool sync, bool cached_only = false);
    // The slot of a program only if its kernel exists already (loaded in this process, in the user's cache or in the pack): looked up
    // on the calling thread at a lazily built program's FIRST launch; nullptr = it stays on the interpreter until it has earned a compilation.
    std::shared_ptr<JitSlot> request_cached(const DevProgramArgs& proto);
    void record(const std::string& description);              // FMHIP_JIT_RECORD: one line per kernel asked for (kernel pack)
    void wait_idle();                                          // until the queue is drained
    JitStats stats();
private:
    void worker();
    void compile(JitSlot& s);
    bool load_cached(JitSlot& s);
    bool finish(JitSlot& s, std::chrono::steady_clock::time_point t0);
    int device_ = 0;
    bool running_ = false, stopping_ = false;
    std::thread thread_;
    std::mutex mu_;
    std::condition_variable cv_, idle_cv_;
    std::deque<std::shared_ptr<JitSlot>> queue_;
    int in_flight_ = 0;
    std::unordered_map<std::string, std::shared_ptr<JitSlot>> cache_;     // by source text
    JitStats stats_;
};

} // namespace fm
