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
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

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
struct JitShape { int elems = 8, group = 4, waves = 0; bool prefetch = true; };
JitShape jit_shape(const DevProgramArgs& proto);

// Source text of the specialised kernel pair of a program (deterministic: it doubles as the cache key).
std::string jit_generate_source(const DevProgramArgs& proto);

class Jit {
public:
    ~Jit() { quiesce(); }                                       // static destruction: join only, no HIP calls (fmhip_shutdown unloads)
    void start(int device);
    void stop();                                               // joins the worker, unloads every module
    void quiesce();                                            // joins the worker only (process exit: no HIP calls)
    // Returns the (shared) slot of this program; compiles synchronously when `sync`, else queues it for the worker.
    std::shared_ptr<JitSlot> request(const DevProgramArgs& proto, bool sync);
    // The same for a kernel pair given as source text (rolled loops, runtime.cpp): `elems` = elements per lane and pass.
    std::shared_ptr<JitSlot> request_source(std::string source, int elems, bool sync);
    void wait_idle();                                          // until the queue is drained
    JitStats stats();
private:
    void worker();
    void compile(JitSlot& s);
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
