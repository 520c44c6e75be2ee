/*
 * fmhip.h — C-ABI of the MI355X-native RandomVariable / BrownianMotion engine.
 *
 * This is the drop-in boundary.  It replaces the JCuda/JCurand JNI surface
 * that the reference binds for this path:
 *
 *   reference (Java, JCuda)                                   this header
 *   ---------------------------------------------------------------------------------------------
 *   cuInit/cuDeviceGet/cuCtxCreate/cuModuleLoad               fmhip_init / fmhip_shutdown
 *     (RandomVariableCuda.java:159-248)
 *   DeviceMemoryPool.getDevicePointer(long)  (:280)           fmhip_vec_create_uninitialized
 *   DeviceMemoryPool.getDevicePointer(float[]) (:457)         fmhip_vec_create_from_float / _from_double
 *   DeviceMemoryPool.getValuesAsFloat        (:469)           fmhip_vec_read_float / _read_double
 *   DeviceMemoryPool.callFunctionv1s0 … v2s1 (:483-557)       fmhip_call_v1s0 … fmhip_call_v3s0
 *   DeviceMemoryPool.clean / purge           (:393,:424)      fmhip_pool_clean / fmhip_pool_purge
 *   getDeviceFreeMemPercentage               (:435)           fmhip_pool_stats
 *   getAverage/getVariance/getMin/getMax     (:830-901)       fmhip_reduce_moments (on device, 32 B back)
 *   curandCreateGenerator/SetSeed/GenerateNormal/Destroy      fmhip_bm_generate
 *     (BrownianMotionCudaWithRandomVariableCuda.java:141-182)
 *   — (no equivalent: one launch per method call)             fmhip_program_* (fused op streams, batched)
 *
 * Conventions
 *   - plain C types only; every function returns an int status (0 = FMHIP_OK, <0 = error class);
 *     a human-readable message for the calling thread's last error is returned by fmhip_last_error().
 *   - a vector handle (fmhip_vec) is an opaque non-zero int64 owned by the caller; release it with
 *     fmhip_vec_release.  Storage is fp32 on the device (README.md:100 of the reference), values cross
 *     the boundary as double or float.
 *   - scalars are passed as double and narrowed with (float) before use, as the reference does
 *     (RandomVariableCuda.java:521,533).
 *   - all entry points are thread-safe; device work is enqueued on one in-order stream per engine — one engine per process
 *     (fmhip_init), per listed device (fmhip_init_devices) or per caller thread (fmhip_set_thread_engines); only the read /
 *     reduce entry points block.
 *   - a process drives one GPU (fmhip_init: one rank per GPU under torch.distributed / RCCL) or several (fmhip_init_devices).
 *   - a handle may be released late and from another thread (a garbage collector's cleaner: what the reference's WeakReference /
 *     ReferenceQueue pool and the Java binding's Cleaner do): the engine does not decide what to store by live handles, see
 *     fmhip_engine_stats.
 */
#ifndef FMHIP_H
#define FMHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMHIP_ABI_VERSION 1

typedef int64_t fmhip_vec;      /* 0 is never a valid handle */
typedef int64_t fmhip_program;  /* 0 is never a valid handle */

typedef enum fmhip_status {
    FMHIP_OK                  =  0,
    FMHIP_ERR_INVALID_HANDLE  = -1,
    FMHIP_ERR_SIZE_MISMATCH   = -2,
    FMHIP_ERR_OUT_OF_MEMORY   = -3,   /* maps to java.lang.OutOfMemoryError (RandomVariableCuda.java:375) */
    FMHIP_ERR_HIP             = -4,   /* a HIP runtime call failed (CudaException in the reference, :167) */
    FMHIP_ERR_INVALID_ARGUMENT= -5,
    FMHIP_ERR_NOT_INITIALIZED = -6,
    FMHIP_ERR_UNSUPPORTED     = -7,   /* UnsupportedOperationException in the reference */
    FMHIP_ERR_PROGRAM_LIMIT   = -8    /* a fused program exceeds the register / input budget */
} fmhip_status;

/*
 * Opcodes.  One per kernel of RandomVariableCudaKernel.cu (line numbers of the .cu given), plus the
 * operations the reference defines only in its CPU twin RandomVariableFromFloatArray.java
 * (choose :1264, isNaN :1441, sin :927, cos :942, addRatio :1395, subRatio :1418).
 * Arithmetic contract for every opcode: each elementary operation rounds to fp32 separately
 * (no FMA contraction — the reference compiles with `nvcc -fmad false`, JCudaUtils.java:69-70);
 * exp/log/pow/sin/cos are evaluated in fp64 and narrowed, sqrt and division are correctly rounded.
 */
typedef enum fmhip_opcode {
    /* vector (a) ∘ scalar (s):  fmhip_call_v1s1 */
    FMHIP_OP_CAP_S      =  1,  /* min(a,s)        capByScalar   .cu:2   */
    FMHIP_OP_FLOOR_S    =  2,  /* max(a,s)        floorByScalar .cu:13  */
    FMHIP_OP_ADD_S      =  3,  /* a + s           addScalar     .cu:24  */
    FMHIP_OP_SUB_S      =  4,  /* a - s           subScalar     .cu:34  */
    FMHIP_OP_BUS_S      =  5,  /* -a + s          busScalar     .cu:44  */
    FMHIP_OP_MULT_S     =  6,  /* a * s           multScalar    .cu:54  */
    FMHIP_OP_DIV_S      =  7,  /* a / s           divScalar     .cu:65  */
    FMHIP_OP_VID_S      =  8,  /* s / a           vidScalar     .cu:76  */
    FMHIP_OP_POW_S      =  9,  /* pow(a,s)        cuPow         .cu:98  */
    /* unary:  fmhip_call_v1s0 */
    FMHIP_OP_SQUARED    = 10,  /* a * a           squared       .cu:87  */
    FMHIP_OP_SQRT       = 11,  /* sqrt(a)         cuSqrt        .cu:109 */
    FMHIP_OP_EXP        = 12,  /* exp(a)          cuExp         .cu:119 */
    FMHIP_OP_LOG        = 13,  /* log(a)          cuLog         .cu:129 */
    FMHIP_OP_INVERT     = 14,  /* 1.0f / a        invert        .cu:139 */
    FMHIP_OP_ABS        = 15,  /* |a|             cuAbs         .cu:149 */
    FMHIP_OP_SIN        = 16,  /* sin(a)          twin :927     */
    FMHIP_OP_COS        = 17,  /* cos(a)          twin :942     */
    FMHIP_OP_ISNAN      = 18,  /* a!=a ? 1 : 0    twin :1441    */
    /* vector ∘ vector:  fmhip_call_v2s0 */
    FMHIP_OP_CAP        = 19,  /* min(a,b)        cap           .cu:160 */
    FMHIP_OP_FLOOR      = 20,  /* max(a,b)        cuFloor       .cu:170 */
    FMHIP_OP_ADD        = 21,  /* a + b           add           .cu:180 */
    FMHIP_OP_SUB        = 22,  /* a - b           sub           .cu:191 */
    FMHIP_OP_MULT       = 23,  /* a * b           mult          .cu:202 */
    FMHIP_OP_DIV        = 24,  /* a / b           cuDiv         .cu:213 */
    /* two vectors and a scalar:  fmhip_call_v2s1 */
    FMHIP_OP_ACCRUE     = 25,  /* a * (1 + b*s)   accrue        .cu:224 */
    FMHIP_OP_DISCOUNT   = 26,  /* a / (1 + b*s)   discount      .cu:234 */
    FMHIP_OP_ADDPRODUCT_VS = 27, /* a + b*s       addProduct_vs .cu:257 */
    /* three vectors:  fmhip_call_v3s0 */
    FMHIP_OP_ADDPRODUCT = 28,  /* a + b*c         addProduct    .cu:247 */
    FMHIP_OP_ADDRATIO   = 29,  /* a + b/c         addRatio      .cu:267 */
    FMHIP_OP_SUBRATIO   = 30,  /* a - b/c         subRatio      .cu:277 */
    FMHIP_OP_CHOOSE     = 31,  /* a>=0 ? b : c    twin :1264    */
    FMHIP_OP__COUNT     = 32
} fmhip_opcode;

/* Result of an on-device reduction over one vector (all fp64). */
typedef struct fmhip_moments {
    double sum;     /* Σ (x_i - shift)            */
    double sumsq;   /* Σ (x_i - shift)^2          */
    double min;     /* min x_i (NaN if any NaN, Java Math.min semantics) */
    double max;     /* max x_i (NaN if any NaN)   */
} fmhip_moments;

typedef struct fmhip_pool_stats_t {
    int64_t bytes_reserved;     /* device bytes obtained from hipMalloc and still held      */
    int64_t bytes_in_use;       /* bytes handed out to live vectors                         */
    int64_t bytes_cached;       /* bytes sitting in free lists                              */
    int64_t device_bytes_free;  /* hipMemGetInfo free (queried here only, never on the hot path) */
    int64_t device_bytes_total;
    int64_t n_alloc_hits;       /* allocations served from a free list                      */
    int64_t n_alloc_misses;     /* allocations that had to call hipMalloc                   */
    int64_t n_live_vectors;
    int64_t n_kernel_launches;  /* launches since init (fusion statistics)                  */
    int64_t n_ops_executed;     /* element-wise ops executed inside those launches          */
} fmhip_pool_stats_t;

/* ---------------------------------------------------------------- lifecycle */

/* Bind this process to one device and create the runtime (stream, pool, kernels).
 * device_index < 0: use env FMHIP_DEVICE_INDEX, else LOCAL_RANK, else 0.
 * Idempotent for the same device.  (RandomVariableCuda.java:159-248) */
int fmhip_init(int device_index);
/* ONE process, SEVERAL devices (replaces the single device index of RandomVariableCuda.java:161,177): every
 * vector is cut into contiguous blocks of paths (boundaries at multiples of four), block d lives on devices[d], every method runs on
 * every block, a read gathers the blocks by one device-to-host copy per device, and host-side moments are the per-device moments
 * added in device order (the rule of fmhip_expectation_combine) — no collective, nothing travels between devices.  Everything else of
 * this header works unchanged on the caller's side; handles are the library's own numbers.  An index may repeat (shards on separate
 * streams of one device: how this is tested on a one-GPU box).  Calls that only return handles are queued to one worker thread per
 * device and return at once: an error a device meets later (an allocation that fails) is returned by the next call that waits (a read,
 * a reduction, fmhip_synchronize).  Expectations wanted ON the devices — fmhip_reduce_moments_batch_devices: a buffer per listed
 * device; the *_device variants: a buffer on the first listed device — are the ONE exchange between the devices: every shard's launches
 * leave its moments in a device buffer of its own, one RCCL all-gather issued for all listed devices inside ncclGroupStart /
 * ncclGroupEnd makes every device hold all of them, a kernel per device combines them in device order (fmhip_expectation_combine's
 * rule: the same bits everywhere); with a repeated index — or without librccl.so, which is looked up at run time — they are combined
 * on the host instead (fmhip_expectation_collective tells which).  Not available with a device list: fmhip_get_stream (use
 * fmhip_get_stream_of), fmhip_vec_device_ptr, device_moments of fmhip_program_run, fmhip_set_expectation_comm with more than one
 * rank.  count == 1 is fmhip_init(devices[0]).  UNMEASURED on more than one physical GPU. */
int fmhip_init_devices(const int* devices, int count);
int fmhip_device_count(int* count);           /* devices (shards) behind the handles: 1 after fmhip_init */
/* An ENGINE PER CALLER THREAD on the one device of fmhip_init: every thread that calls into the library gets a pending graph, a stream,
 * a memory pool, a time-step grouping and a lock of its own (the thread that switches this on keeps the engine it initialised).  Threads
 * that simulate side by side — finmath-lib's optimiser evaluates the columns of a Jacobian on a thread pool
 * (LIBORMarketModelCalibrationATMTest.java:319) — record without meeting each other and their launches share the device.  Handles stay
 * valid everywhere: a vector of another thread is computed by ITS engine if it is still pending and enters a method as an operand that
 * aliases that storage (streams ordered by events); release, retain, read, moments, programs and tickets of another thread run on the
 * owner's engine.  Settings (fusion, math mode, JIT mode, step grouping) apply to every engine; fmhip_flush and fmhip_fusion_hold to the
 * calling thread's; statistics are summed; fmhip_synchronize waits for all.  Not together with a device list.  Ends with fmhip_shutdown.
 * Vectors are immutable, which is what makes sharing them safe; the two ways to WRITE into one (fmhip_program_run_into, a raw device pointer)
 * are the caller's to order against other threads' use of that vector (fmhip_synchronize).
 * The reference funnels every thread through one executor thread (RandomVariableCuda.java:155). */
int fmhip_set_thread_engines(int enabled, int* previous);
int fmhip_shutdown(void);
int fmhip_is_initialized(void);
int fmhip_abi_version(void);
/* Message of the calling thread's most recent failing call ("" if none). Never NULL. */
const char* fmhip_last_error(void);
/* Device name into buf (NUL-terminated, truncated), CU count, HBM bytes. Any pointer may be NULL. */
int fmhip_device_info(char* name_buf, int name_buf_len, int* n_compute_units, int64_t* hbm_bytes);
/* Block until all enqueued device work is complete (cuCtxSynchronize in the reference, :474). */
int fmhip_synchronize(void);
/* The HIP stream (hipStream_t as void*) all work is enqueued on — for interop with torch / RCCL. */
int fmhip_get_stream(void** stream_out);

/* ---------------------------------------------------------------- vectors */

/* Narrow double→float on the host (RandomVariableCuda.java:768-774) and upload. n >= 0. */
int fmhip_vec_create_from_double(const double* host_values, int64_t n, fmhip_vec* out);
int fmhip_vec_create_from_float(const float* host_values, int64_t n, fmhip_vec* out);
/* Device vector with every element = (float)value (twin ctor RandomVariableFromFloatArray.java:139-146). */
int fmhip_vec_create_filled(int64_t n, double value, fmhip_vec* out);
/* Pool allocation without initialisation (RandomVariableCuda.getDevicePointer(long), :737). */
int fmhip_vec_create_uninitialized(int64_t n, fmhip_vec* out);
int fmhip_vec_retain(fmhip_vec v);
int fmhip_vec_release(fmhip_vec v);
int fmhip_vec_size(fmhip_vec v, int64_t* n_out);
/* getRealizations(): materialise, D2H, widen float→double (RandomVariableCuda.java:1116-1123). */
int fmhip_vec_read_double(fmhip_vec v, double* host_out, int64_t n);
int fmhip_vec_read_float(fmhip_vec v, float* host_out, int64_t n);
/* Raw device pointer of the (materialised) fp32 storage; valid until the handle is released.  Vectors that were computed as identical
 * rows of one launch share their storage (the same inputs, the same scalars: computed once); a vector that shares it receives storage
 * of its own here, and in fmhip_program_run_into, before the pointer is handed out — writing through it never changes another vector. */
int fmhip_vec_device_ptr(fmhip_vec v, void** device_ptr_out);

/* ---------------------------------------------------------------- eager ops, one per reference launch helper */

/* result = f(a)            callFunctionv1s0 (RandomVariableCuda.java:483) */
int fmhip_call_v1s0(int opcode, fmhip_vec a, fmhip_vec* out);
/* result = f(a, (float)s)  callFunctionv1s1 (:515) */
int fmhip_call_v1s1(int opcode, fmhip_vec a, double s, fmhip_vec* out);
/* result = f(a, b)         callFunctionv2s0 (:493) */
int fmhip_call_v2s0(int opcode, fmhip_vec a, fmhip_vec b, fmhip_vec* out);
/* result = f(a, b, (float)s) callFunctionv2s1 (:527) */
int fmhip_call_v2s1(int opcode, fmhip_vec a, fmhip_vec b, double s, fmhip_vec* out);
/* result = f(a, b, c)      callFunctionv3s0 (:504) */
int fmhip_call_v3s0(int opcode, fmhip_vec a, fmhip_vec b, fmhip_vec c, fmhip_vec* out);

/* Deferred execution of the five calls above: when enabled they only record a node and return a
 * pending handle; chains are executed as ONE fused launch when a value is needed (read, reduce,
 * device_ptr, fmhip_flush) — results are bit-identical to eager execution. Default: disabled (eager).
 * Returns the previous setting through *previous (may be NULL). */
int fmhip_set_fusion(int enabled, int* previous);
/* Execute every pending node that is still referenced by a live handle (identical programs over
 * different vectors are batched into one launch). */
int fmhip_flush(void);
/* While held (hold != 0), a pending chain is never executed on the engine's own accord (normally it is once ≈ 40 methods
 * have accumulated below one handle, so that the device starts early); it runs when a value is needed or at fmhip_flush.
 * For callers that record many independent chains of identical structure — all products of a valuation, all bumped
 * parameter sets of a Jacobian — and want them batched as rows of the same launches.  Releasing the hold executes nothing
 * by itself.  hold == 2 is a SOFT hold: the same, except that everything pending is executed once more than 32768
 * operations wait — for helpers that group work on a caller's behalf and cannot know when the caller is done
 * (BrownianMotionHip groups the time steps of an Euler scheme this way).  Returns the previous setting (0, 1 or 2)
 * through *previous (may be NULL). */
int fmhip_fusion_hold(int hold, int* previous);
/* Time-step grouping on behalf of callers that give no hints (finmath-lib's Euler scheme through the RandomVariable interface).
 * A discretisation scheme reads the Brownian increments of time index i exactly while it computes step i; the engine watches for
 * the first use of an increment of fmhip_bm_generate with a new time index and keeps the methods recorded between `steps` such
 * boundaries pending (as under a soft hold), then executes them together — whole time steps, scheduled component by component,
 * their periodic stretch as one rolled-loop launch — instead of cutting the stream every ≈ 40 methods.  The last time index of a
 * generation ends the grouping.  Only while the fusion front-end is on and the caller holds nothing itself (fmhip_fusion_hold).
 * steps = 0 switches it off; default 4 (environment: FMHIP_GROUP_STEPS).  Results never depend on it.
 * Replaces what BrownianMotionCudaWithRandomVariableCuda's callers get from the reference: nothing (one launch per method).
 * Returns the previous setting through *previous (may be NULL). */
int fmhip_set_step_grouping(int steps, int* previous);
/* Replicate PENDING expressions: a caller that is about to record the same chain of methods again with other vectors and other
 * scalar operands — the next scenario, the next bumped parameter set of a Jacobian — records it ONCE and asks for copies.
 * The graph = every pending (not yet executed) operation below `roots`.  Copy j reads leaf_to[j*n_map + i] wherever the
 * original reads leaf_from[i] (operands outside the graph; operands not listed are shared by all copies), and takes its scalar
 * operands from scalars[j*n_scalars + k], k counting the graph's scalar-carrying operations in the order they were recorded
 * (scalars == NULL: the original's; otherwise n_scalars must equal the graph's count — see fmhip_graph_scalars).
 * out[j*n_roots + r] receives the copy of roots[r] (a handle the caller releases); the inner values of a copy have no handle.
 * Cost ≈ 50 ns per operation and copy, against one C call, one handle and one release per operation when recorded by hand.
 * The copies are ordinary pending expressions: the next flush batches them with the original as rows of the same launches.
 * Requires the fusion front-end (fmhip_set_fusion); results are bit-identical to recording each copy by hand. */
int fmhip_graph_clone(const fmhip_vec* roots, int n_roots, int n_copies,
                      const fmhip_vec* leaf_from, const fmhip_vec* leaf_to, int n_map,
                      const double* scalars, int n_scalars, fmhip_vec* out);
/* The scalar operands of the pending graph below `roots`, in recording order: *n_scalars receives their number, the first
 * min(capacity, number) go to scalars_out (may be NULL).  Lets a caller check its scalar list against what it recorded. */
int fmhip_graph_scalars(const fmhip_vec* roots, int n_roots, double* scalars_out, int capacity, int* n_scalars);

/* Arithmetic mode of exp and log (everything else is identical in both modes):
 *   FMHIP_MATH_EXACT (default): evaluated in fp64 and narrowed once — bit-identical to the reference's CPU twin
 *                               `(float)Math.exp(x)` except where two fp64 libms differ in the last fp64 ulp;
 *   FMHIP_MATH_FAST:            hardware v_exp_f32 / v_log_f32 with fp32 range reduction, within 2 fp32 ulp — the accuracy
 *                               class of the CUDA expf/logf used by the reference's kernels; ~4x fewer instructions.
 * Applies to programs compiled after the call (eager ops, fused chains, fmhip_program_create). */
#define FMHIP_MATH_EXACT 0
#define FMHIP_MATH_FAST  1
int fmhip_set_math_mode(int mode, int* previous);

/* ---------------------------------------------------------------- reductions */

/* One pass over v on the device: Σ(x-shift), Σ(x-shift)², min, max with fp64 accumulation; 32 bytes
 * come back.  Replaces the reference's D2H of the whole vector + host loops
 * (RandomVariableCuda.java:830-901 → RandomVariableFromFloatArray.java:284-382).
 * The moments of a vector are a function of its elements and its length alone (one reduction tree per vector): the same bits
 * whether a launch of their own computes them, the launch that computes the vector, or a launch over many vectors.  With fusion
 * on (fmhip_set_fusion) and v pending among much other pending work, the call may execute ALL pending work — expressions of equal
 * shape as rows of the same launches — and keep the unshifted moments of every vector those launches produce: later calls for
 * those vectors return them without a launch (vectors are immutable; fmhip_program_run_into and fmhip_vec_device_ptr make the
 * engine forget).  Never under the caller's hard hold (fmhip_fusion_hold(1)).  The calling thread waits for the device without
 * holding the engine: other threads keep recording and launching. */
int fmhip_reduce_moments(fmhip_vec v, double shift, fmhip_moments* out);
/* Same, but the 4 doubles are written to caller-owned DEVICE memory (e.g. a torch tensor that is then
 * all-reduced with RCCL); asynchronous on the runtime stream. */
int fmhip_reduce_moments_device(fmhip_vec v, double shift, void* device_out_4_doubles);

/* `count` vectors of equal size reduced by ONE launch and one 32·count-byte read-back (all expectations of one objective
 * evaluation at once).  shifts may be NULL (= 0 for all). */
int fmhip_reduce_moments_batch(const fmhip_vec* vectors, int count, const double* shifts, fmhip_moments* out);
/* Same, results (count x 4 doubles) left in caller-owned DEVICE memory, asynchronous on the runtime stream — the send buffer
 * of the one RCCL all-reduce per objective evaluation when paths are sharded over GPUs. */
int fmhip_reduce_moments_batch_device(const fmhip_vec* vectors, int count, const double* shifts, void* device_out);
/* The same ON EVERY DEVICE of a device list (fmhip_init_devices): device_out[d] — a buffer of count x 32 bytes on the d-th listed device,
 * the caller's own allocation; NULL: not wanted there — receives the moments of the WHOLE vectors, all path shards combined in device
 * order (bit-identical on every device).  Nothing is waited for: the result is ordered on the shards' streams (fmhip_get_stream_of).
 * n_devices = fmhip_device_count (with one device: 1, and this is fmhip_reduce_moments_batch_device).  The reference has no counterpart
 * (one device index: RandomVariableCuda.java:161,177); SURVEY.md §8e. */
int fmhip_reduce_moments_batch_devices(const fmhip_vec* vectors, int count, const double* shifts, void* const* device_out, int n_devices);
/* The stream the work of device shard `shard` is ordered on (shard 0 of one device: fmhip_get_stream). */
int fmhip_get_stream_of(int shard, void** stream_out);
/* How expectations wanted on the devices of a device list travel: kind 0 = one device, nothing to exchange; 1 = a grouped RCCL all-gather
 * over the listed devices; 2 = combined on the host (`why`, if given, says why: a repeated device index, librccl.so not found …). */
int fmhip_expectation_collective(int* kind, char* why, int why_len);
/* The same reduction in two halves: _begin enqueues it and returns a ticket at once; _end waits for THAT reduction — not for work
 * enqueued after it — writes the `count` moments (the expectation communicator applied, as in fmhip_reduce_moments_batch) and
 * retires the ticket.  A caller that evaluates one parameter set after the other records and enqueues set k+1 between the two
 * halves of set k: the device never waits for the host at the boundary (one in-order stream: a blocking read of set k's results
 * AFTER set k+1 was enqueued would wait for set k+1 as well).  The reference has no counterpart: its getAverage() synchronises
 * the context (RandomVariableCuda.java:869-878).  A ticket must be ended exactly once. */
typedef int64_t fmhip_ticket;
int fmhip_reduce_moments_batch_begin(const fmhip_vec* vectors, int count, const double* shifts, fmhip_ticket* ticket_out);
int fmhip_reduce_moments_batch_end(fmhip_ticket ticket, fmhip_moments* out, int count);
/* For a caller that wants the EXPECTATIONS of vectors and will never read their values — a Monte-Carlo product's payoff, of which
 * getValue() returns the average.  A vector that is still a pending expression and that only the caller references need then not be
 * written to memory at all: when a later fmhip_reduce_moments / _batch / _batch_begin runs what is pending, the launch that computes
 * such a vector takes its moments and drops it (the one store per workgroup behind a chain of reads costs that launch 8-10 % of its
 * rate).  After this call the values of `vectors` must not be read or used as operands again
 * (FMHIP_ERR_INVALID_ARGUMENT where the engine has in fact not kept them); their moments stay available; the handles are released
 * as usual.  Vectors computed already, or referenced by somebody else, are simply left as they are.  The reference has no
 * counterpart: its getAverage() copies the whole vector to the host (RandomVariableCuda.java:869-878). */
int fmhip_vec_give_up_values(const fmhip_vec* vectors, int count);

/* Expectation communicator: Monte-Carlo paths sharded over processes (one GPU each) behind an UNCHANGED caller.
 * Every vector of this process holds the paths [rank·n, (rank+1)·n) of a global vector of world·n paths; all element-wise
 * work is local; the one thing that couples paths is an expectation.  With a communicator set, fmhip_reduce_moments and
 * fmhip_reduce_moments_batch return the moments of the GLOBAL vector on every rank: the local {Σ, Σ², min, max} are handed to
 * `gather` (called on the host, by the thread that asked; it must deliver every rank's `count` doubles to every rank, in rank
 * order — an all-gather over RCCL / MPI / gloo) and combined by fmhip_expectation_combine: sums added in RANK ORDER (bitwise
 * equal on every rank, whatever the transport), min / max over ranks.  The drop-in classes divide by world·n
 * (fmhip_expectation_world), so getAverage() / getVariance() of a finmath-lib product mean the same as on one GPU and every
 * rank's optimiser takes the same step.  The *_device variants stay local partials (for callers that run their own collective
 * on the device).  Every rank must ask for the same expectations in the same order.  world = 1 or gather = NULL removes it.
 * The reference has nothing here: one process, one device (RandomVariableCuda.java:161,177). */
typedef int (*fmhip_gather_fn)(void* context, const double* local, int count_doubles, double* gathered /* [world][count_doubles] */);
int fmhip_set_expectation_comm(int world, int rank, fmhip_gather_fn gather, void* context);
/* The communicator's size and this process's rank (1 and 0 without one). */
int fmhip_expectation_world(int* world, int* rank);
/* The combination rule on its own (host only, needs no device): gathered[r·count + k] = rank r's moments of vector k. */
int fmhip_expectation_combine(const fmhip_moments* gathered, int world, int count, fmhip_moments* out);

/* ---------------------------------------------------------------- fused programs */

/* One SSA instruction. Values are numbered: 0 … n_inputs-1 are the program inputs, n_inputs+i is the
 * result of ops[i]. Unused operands are -1. */
typedef struct fmhip_prog_op {
    int32_t opcode;     /* fmhip_opcode */
    int32_t a, b, c;    /* operand value ids */
    double  scalar;     /* narrowed to float, used by the *_S / v2s1 opcodes */
} fmhip_prog_op;

/* Compile an op stream once. out_values[n_outputs]: value ids materialised as new vectors;
 * reduce_values[n_reduce] (n_reduce <= 2): value ids reduced to fmhip_moments inside the same launch. */
int fmhip_program_create(const fmhip_prog_op* ops, int n_ops, int n_inputs,
                         const int32_t* out_values, int n_outputs,
                         const int32_t* reduce_values, int n_reduce,
                         fmhip_program* out);
int fmhip_program_release(fmhip_program p);
/* Number of kernel launches one run of p takes (1 unless the program had to be split). */
int fmhip_program_launch_count(fmhip_program p, int* n_launches);
/* Inputs, outputs and fused reductions per batch row of p: the lengths fmhip_program_run expects of its arrays. */
int fmhip_program_shape(fmhip_program p, int* n_inputs, int* n_outputs, int* n_reduce);
/* Run p over `batch` independent input tuples in ONE launch (horizontal batching).
 *   inputs  [batch * n_inputs ]  all of equal size
 *   outputs [batch * n_outputs]  receives new handles (caller releases)
 *   reduce_shift [n_reduce] or NULL (=0)
 *   moments [batch * n_reduce] host results, or NULL to skip the host read-back
 *   device_moments: optional device buffer of batch*n_reduce*4 doubles, or NULL */
int fmhip_program_run(fmhip_program p, int batch,
                      const fmhip_vec* inputs, fmhip_vec* outputs,
                      const double* reduce_shift, fmhip_moments* moments, void* device_moments);
/* As fmhip_program_run, but writes into caller-provided existing output vectors (no allocation);
 * an output may alias an input of the same row. */
int fmhip_program_run_into(fmhip_program p, int batch,
                           const fmhip_vec* inputs, const fmhip_vec* outputs,
                           const double* reduce_shift, fmhip_moments* moments, void* device_moments);

/* ---------------------------------------------------------------- execution tiers
 * A compiled program (explicit, or built by the lazy front-end) starts on the bytecode interpreter kernel and may be
 * promoted to a SPECIALISED kernel: straight-line gfx950 code generated from its op stream and compiled at run time
 * (hiprtc) on a background thread.  Both tiers are built from the same device functions with the same floating-point
 * flags and give bit-identical results.  The reference has no counterpart (one precompiled PTX kernel per method,
 * RandomVariableCuda.java:78-117, :539-557). */
#define FMHIP_JIT_OFF  0     /* interpreter only */
#define FMHIP_JIT_AUTO 1     /* default: explicit programs at creation, lazy programs once they are hot; asynchronous */
#define FMHIP_JIT_SYNC 2     /* compile before the first launch (deterministic tier: tests, benchmarks) */
/* Default from the environment variable FMHIP_JIT = off | auto | sync. */
int fmhip_set_jit(int mode, int* previous);
/* Blocks until every queued compilation has finished. */
int fmhip_jit_wait(void);
/* compiled = kernels ready (of which disk_cache_hits were loaded from the persistent code-object cache,
 * $FMHIP_JIT_CACHE_DIR, default ~/.cache/fmhip-jit, "off" disables), compile_seconds = host time spent on them. */
int fmhip_jit_stats(int64_t* compiled, int64_t* failed, int64_t* pending, double* compile_seconds, int64_t* disk_cache_hits);
/* *tier: 0 = interpreter, 1 = specialised kernel ready (the next launch uses it); *vgprs = its register count. */
int fmhip_program_tier(fmhip_program program, int* tier, int* vgprs);
/* The generated source of a program's specialised kernel (needs no device).  Copies at most `capacity` bytes
 * including the terminating 0 and stores the full length (without the 0) in *needed. */
int fmhip_program_source(const fmhip_prog_op* ops, int n_ops, int n_inputs, const int32_t* out_values, int n_outputs,
                         const int32_t* reduce_values, int n_reduce, char* buffer, int64_t capacity, int64_t* needed);

/* ---------------------------------------------------------------- Brownian increments */

/* Fill n_steps*n_factors new vectors of n_paths N(0, dt_step) increments:
 *   out[step*n_factors + factor][p] = (float)sqrt(dt[step]) * Z(seed, step, factor, path_offset + p)
 * Z is a counter-based normal — Philox4x32-10 (counter = (global path index / 4, step * n_factors + factor), key = seed), each 32-bit
 * word mapped through the inverse normal CDF (a segment table in LDS and a cubic per segment: csrc/fm_normal_table.hpp) — a pure
 * function of (seed, step, factor, global path index), hence independent of launch geometry and of how paths are sharded over GPUs.
 * Replaces curandGenerateNormal per (step,factor) (BrownianMotionCudaWithRandomVariableCuda.java:168-178). */
int fmhip_bm_generate(int64_t seed, int n_steps, int n_factors, int64_t n_paths, int64_t path_offset,
                      const double* dt, fmhip_vec* out);

/* Host-side Mersenne-Twister Brownian motion (restatement of finmath-lib's BrownianMotionFromMersenneRandomNumbers, the
 * generator the reference's tests feed to every factory: LIBORMarketModelCalibrationATMTest.java:283): MT19937 seeded like
 * commons-math3 MersenneTwister(int), nextDouble(), inverse normal CDF (AS 241), times sqrt(dt); draw order path-major.
 * fmhip_mersenne_increments fills host doubles out[(step*n_factors+factor)*n_paths + path] and needs no device;
 * fmhip_bm_generate_mersenne uploads them as n_steps*n_factors vectors (narrowed to fp32 like any double[] upload). */
int fmhip_mersenne_increments(int32_t seed, int n_steps, int n_factors, int64_t n_paths, const double* dt, double* host_out);
int fmhip_bm_generate_mersenne(int32_t seed, int n_steps, int n_factors, int64_t n_paths, const double* dt, fmhip_vec* out);
/* Inverse of the standard normal CDF (Wichura AS 241 / PPND16), exposed for tests. Returns the value (no status). */
double fmhip_inverse_normal_cdf(double p);

/* ---------------------------------------------------------------- pool */

/* Return cached (unused) device buffers to the driver (DeviceMemoryPool.clean, :393). */
int fmhip_pool_clean(void);
/* clean + drop every cached buffer; live vectors stay valid (DeviceMemoryPool.purge, :424). */
int fmhip_pool_purge(void);
int fmhip_pool_stats(fmhip_pool_stats_t* out);

/* ---------------------------------------------------------------- measurement */

/* While enabled, every launch of the fused-program kernel is bracketed by a pair of HIP events on the
 * runtime stream.  fmhip_profile_read blocks until the recorded launches have finished, returns the sum
 * of their device durations (ms) and their count, and clears the record.  Used by bench.py for the live
 * roofline figure; adds two event records per launch, so leave it off in production. */
int fmhip_profile_enable(int enabled);
/* Algorithmic bytes of all program launches so far (4 B x N x (inputs read + outputs written) per batch row;
 * reductions add nothing) and how many of those launches ran on the specialised tier. */
int fmhip_traffic_stats(int64_t* algorithmic_bytes, int64_t* specialised_launches);
/* Counters of the engine's work so far (all engines of the process added up; every field an int64, `size` = bytes filled in).
 * The reference keeps no such record; the nearest are the log lines of DeviceMemoryPool (RandomVariableCuda.java:287-296).
 *   values_deferred / _now / values_demanded: results of recorded methods that a fused launch computed for its consumers and did NOT
 *   store although the caller still held a handle (the engine has learnt that handles at such positions are never used again — a
 *   garbage-collected caller holds one for EVERY temporary until its next collection), how many of them exist now, and how many were
 *   wanted after all (computed again from their recipe and stored: each costs a launch, and teaches the engine to store that position). */
typedef struct fmhip_engine_stats_t {
    int64_t size;
    int64_t kernel_launches, specialised_launches, interpreter_launches;
    int64_t algorithmic_bytes, algorithmic_bytes_written;
    int64_t values_deferred, values_deferred_now, values_demanded;
    int64_t pending_operations;
    int64_t peak_bytes_reserved;
    /* releases that arrived from a thread which only ever releases (a collector's cleaner): queued, and performed by a driving thread
     * while it waited for the device / at once because too many waited; the time the driving threads spent on them */
    int64_t late_releases_while_waiting, late_releases_at_once, late_release_nanoseconds;
    /* launches that served several components of one loop shape reading the same sequence of vectors (each vector loaded once for all of
     * them: the swaptions of one exercise date), and how many components they served */
    int64_t merged_launches, merged_chains;
    /* rows of batched launches that were not computed because an earlier row of the same launch read the same vectors with the same
     * scalars (the parameter sets of a finite-difference batch before their bumped parameter matters): they share its vectors */
    int64_t common_rows;
} fmhip_engine_stats_t;
int fmhip_engine_stats(fmhip_engine_stats_t* out);
int fmhip_profile_read(double* kernel_ms_total, int64_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* FMHIP_H */
