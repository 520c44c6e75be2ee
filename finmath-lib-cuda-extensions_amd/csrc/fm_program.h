// fm_program.h — the device-visible form of a fused RandomVariable op stream (shared host/device).
//
// A program is a short ACCUMULATOR-machine bytecode executed per path by ONE kernel (kernels.hip):
//   - an accumulator A and FM_NREG virtual fp32 registers R[] per element, all in VGPRs (R is indexed
//     through s_set_gpr_idx; no scratch, no LDS);
//   - inputs are preloaded into R[0 … n_in-1] with 128-bit loads, all in flight together;
//   - each micro-op is wave-uniform: A = f(A [, R[r1] [, R[r2]]] [, scalar]), optionally followed by
//     R[st] = A.  A chain x.add(4).div(2).exp() never touches R: operands are fetched from R only where an
//     instruction really has a second/third vector operand (this is what makes interpretation cheap: the
//     first design, a 3-address register machine, spent 16 v_mov per op and ran at 1.5 TB/s);
//   - up to FM_MAX_OUT registers are stored as new vectors, up to FM_MAX_RED registers are reduced
//     (Σ, Σ², min, max in fp64) inside the same launch.
// Everything that differs between the rows of a horizontally batched launch (vector pointers, scalar
// operands, reduction shifts) lives in a per-row "row block", so identical op streams over different
// vectors/scalars (e.g. all LIBOR components of one Euler step) run as ONE launch.
#pragma once
#ifdef __HIPCC_RTC__            // hiprtc (JIT tier): no libc headers; fixed-width types come from its built-in runtime header
using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;
using size_t = decltype(sizeof(0));
#else
#include <stdint.h>
#endif

namespace fm {

constexpr int FM_NREG     = 16;    // virtual registers per element lane
constexpr int FM_MAX_OPS  = 128;   // micro-ops per launch (one extra slot is kept for the prefetch of ops[pc+1])
constexpr int FM_MAX_IN   = 12;    // preloaded input vectors per launch
constexpr int FM_MAX_OUT  = 8;     // materialised output vectors per launch
constexpr int FM_MAX_RED  = 2;     // fused reductions per launch
constexpr int FM_MAX_SCAL = 64;    // scalar operand slots per launch
constexpr int FM_BLOCK    = 256;   // threads per workgroup (4 waves)
constexpr int FM_VEC      = 4;     // elements per thread per tile (one 128-bit access per vector)
constexpr int FM_UNIT_ELEMS = 2048; // a unit of the reduction tree: FM_BLOCK lanes x 8 elements (fm_kernel_parts.hpp)
constexpr int FM_SPAN_UNITS = 4;    // units per span of the reduction tree

// Micro-ops.  "_A/_B/_T/_P/_N" name which operand of the public opcode sits in the accumulator.
enum UOp : uint32_t {
    U_NOP = 0, U_LDA,                                                       // A = R[r1]
    // A = f(A)
    U_SQUARED, U_SQRT, U_EXP, U_LOG, U_INVERT, U_ABS, U_SIN, U_COS, U_ISNAN,
    // A = f(A, s)
    U_CAP_S, U_FLOOR_S, U_ADD_S, U_SUB_S, U_BUS_S, U_MULT_S, U_DIV_S, U_VID_S, U_POW_S,
    // A = f(A, R[r1])
    U_CAP, U_FLOOR, U_ADD, U_MULT, U_SUB /*A-R*/, U_BUS /*R-A*/, U_DIV /*A/R*/, U_VID /*R/A*/,
    // A = f(A, R[r1], s)
    U_ACCRUE_A /*A*(1+R*s)*/, U_ACCRUE_B /*R*(1+A*s)*/, U_DISCOUNT_A /*A/(1+R*s)*/, U_DISCOUNT_B /*R/(1+A*s)*/,
    U_ADDPRODUCT_VS_A /*A+R*s*/, U_ADDPRODUCT_VS_B /*R+A*s*/,
    // A = f(A, R[r1], R[r2])
    U_ADDPRODUCT_A /*A+R1*R2*/, U_ADDPRODUCT_B /*R1+A*R2*/, U_ADDRATIO_A /*A+R1/R2*/, U_SUBRATIO_A /*A-R1/R2*/,
    U_CHOOSE_T /*A>=0?R1:R2*/, U_CHOOSE_P /*R1>=0?A:R2*/, U_CHOOSE_N /*R1>=0?R2:A*/,
    // fast-math variants (hardware transcendentals, ≤ 2 ulp)
    U_EXP_FAST, U_LOG_FAST,
    U__COUNT
};

// Number of register operands (R[r1], R[r2]) a micro-op reads besides the accumulator.
constexpr int fm_uop_operands(uint32_t code) {
    return (code == U_LDA || (code >= U_CAP && code <= U_ADDPRODUCT_VS_B)) ? 1
         : (code >= U_ADDPRODUCT_A && code <= U_CHOOSE_N) ? 2 : 0;
}

// The micro-ops that contain a division: the specialised kernels evaluate them two elements at a time (fm_device_math.hpp: ueval_div_all).
constexpr bool fm_uop_divides(uint32_t code) {
    return code == U_INVERT || code == U_DIV_S || code == U_VID_S || code == U_DIV || code == U_VID || code == U_DISCOUNT_A || code == U_DISCOUNT_B ||
           code == U_ADDRATIO_A || code == U_SUBRATIO_A;
}

// Instruction word: code[0:7] r1[8:11] r2[12:15] store[16:19] scalar_slot[24:31].
// Every micro-op writes A to R[store]; the LAST register of the file is a dummy that is never allocated ("no
// store") — an unconditional write is cheaper for the compiler than a conditional one (see the kernel).
// Two kernel variants: variant 0 = 4 elements per lane, 16 registers; variant 1 = 8 elements per lane, 9 registers.
struct DevOp { uint32_t w; };
constexpr int FM_VARIANT_NREG[2] = { 16, 9 };
constexpr int FM_VARIANT_ELEMS[2] = { 4, 8 };
static inline uint32_t fm_pack_op(unsigned code, unsigned r1, unsigned r2, unsigned st, unsigned sslot) {
    return (code & 0xffu) | ((r1 & 15u) << 8) | ((r2 & 15u) << 12) | ((st & 15u) << 16) | ((sslot & 0xffu) << 24);
}

// Row block layout (one per batch row), in units of 8 bytes:
//   [0, n_in)                      const float*  input pointers
//   [n_in, n_in+n_out)             float*        output pointers
//   [.., +n_red)                   double        reduction shifts
//   then n_scal floats (padded to a multiple of 2)
constexpr int FM_ROW_WORDS_MAX = FM_MAX_IN + FM_MAX_OUT + FM_MAX_RED + FM_MAX_SCAL / 2;   // 54 x 8 B
constexpr int FM_INLINE_WORDS = 368;    // row blocks carried in the kernel arguments (the argument segment is limited to 4 KB)

// Arrival counters of the fused final combine: 8 planes (up to 7 group counters + one second-level counter per row:
// fm_kernel_parts.hpp block_combine) of one 256-byte block per row — 128 MB of the 288 GB.  Device-scope atomics are executed at the memory side, ≈ 11-13 ns apiece
// on one address; packed into one or two cache lines the counters of all rows of a launch serialise there (measured on 64
// rows x 123 workgroups: stand-alone reduction 95 µs packed, 44 µs apart; the bench launch with 4x the workgroups 363 → 223 µs).
constexpr uint32_t FM_COUNTER_STRIDE = 64;      // in uint32_t
constexpr uint32_t FM_COUNT_IN_GROUPS_FROM = 256;  // workgroups of a row from which a row of ONE combine group counts its arrivals in seven (block_combine)
constexpr uint32_t FM_MAX_ROWS = 65536;
constexpr uint32_t FM_COUNTER_PLANES = 8;
constexpr size_t   FM_COUNTER_PLANE = (size_t)FM_MAX_ROWS * FM_COUNTER_STRIDE;      // in uint32_t

constexpr uint32_t FM_ARGS_LOG_TABLE = 1u;      // the program evaluates log_f: the kernel copies the log table into LDS first

struct DevProgramArgs {
    uint32_t n_ops, n_in, n_out, n_red;
    uint32_t n_scal, row_words;          // row stride in 8-byte words
    uint32_t tiles_per_row, use_inline;  // passes of FM_BLOCK*E elements; use_inline: the row blocks of the whole batch are in inline_row
    uint32_t variant, flags;             // kernel variant (FM_VARIANT_*); FM_ARGS_* bits
    uint32_t block_tiles, span_blocks;   // passes a workgroup takes (workgroup b: tiles [b·block_tiles, (b+1)·block_tiles)); with reductions: workgroups per
                                         //   span of the reduction tree (fm_kernel_parts.hpp: 1 = a workgroup takes a whole span, 4 = one unit each)
    int64_t  n;                          // elements per vector
    double*   results;                   // [batch][n_red][4] final {Σ, Σ², min, max} (written by the last workgroup of a row)
    uint32_t* counters;                  // [batch] arrival counters of the fused final combine, zero between launches
    uint64_t* done_flag;                 // batch == 1 with the results wanted on the host: the wave that writes the final moments stores done_value here
    uint64_t  done_value;                //   afterwards (system-scope release) — the host polls pinned memory instead of synchronising the stream; else nullptr
    uint32_t out_reg[FM_MAX_OUT];        // 32-bit so that they are fetched with scalar loads (gfx9 has no s_load_u8)
    uint32_t red_reg[FM_MAX_RED];
    DevOp    ops[FM_MAX_OPS + 2];            // two slack entries: the kernel prefetches ops[pc+1], ops[pc+2]
    uint64_t inline_row[FM_INLINE_WORDS];    // [batch][row_words] when batch·row_words fits: no table upload (an in-stream copy kernel, ≈ 5 µs) for that launch
};

static_assert(sizeof(DevProgramArgs) <= 4096, "kernel arguments are limited to 4 KB");

// Arguments of a ROLLED-LOOP kernel (runtime.cpp: rolled components): the body of one iteration is compiled into the kernel, the
// iteration count and everything that differs between iterations (vector pointers, scalar operands) come from the row table:
//   row = [G global input ptrs][CI carried-in ptrs][iterations x {LI input ptrs, LO output ptrs}][iterations x LS scalars (float)]
struct DevRolledArgs {
    int64_t  n;                          // elements per vector
    uint32_t tiles_per_row;              // passes of FM_BLOCK*E elements
    uint32_t row_words;                  // row stride in 8-byte words
    uint32_t iterations;
    uint32_t pad;                        // merged launches: chains per row (results: [rows][chains][4]); else 0
    uint64_t dump;                       // FM_DUMP_BYTES of device memory nobody reads: where lanes past the end of a vector put their stores
    // kernels with a fused reduction of one of their values (one unit of the reduction tree per workgroup: fm_kernel_parts.hpp), else unused
    double    shift;                     // subtracted from every element before it is added (getVariance's second pass)
    double*   partials;                  // [rows][tiles + 8][4]
    double*   results;                   // [rows][4] final {Σ, Σ², min, max}
    uint32_t* counters;                  // arrival counters, as DevProgramArgs::counters
    uint64_t* done_flag;                 // as DevProgramArgs::done_flag
    uint64_t  done_value;
    // peeled kernels (launches of few rows): the row table travels here when rows x row_words fits — no in-stream copy in front of the launch
    uint64_t  inline_row[FM_INLINE_WORDS];
};
static_assert(sizeof(DevRolledArgs) <= 4096, "kernel arguments are limited to 4 KB");
constexpr size_t FM_DUMP_BYTES = (size_t)FM_BLOCK * 16 * 4;      // one float4 per lane and per float4-per-lane of the widest kernel (E = 16)
static_assert(FM_INLINE_WORDS >= FM_ROW_WORDS_MAX, "one row always fits");

} // namespace fm
