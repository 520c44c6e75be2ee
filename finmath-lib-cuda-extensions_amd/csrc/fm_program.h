// fm_program.h — the device-visible form of a fused RandomVariable op stream (shared host/device).
//
// A program is a short register-machine bytecode executed per path by ONE kernel (kernels.hip):
//   - FM_NREG virtual fp32 registers per element, held in VGPRs (indexed through s_set_gpr_idx);
//   - inputs are preloaded into registers 0 … n_in-1 with 128-bit loads, all in flight together;
//   - each instruction is wave-uniform: {opcode, dst, a, b, c, scalar slot};
//   - up to FM_MAX_OUT registers are stored as new vectors, up to FM_MAX_RED registers are reduced
//     (Σ, Σ², min, max in fp64) inside the same launch.
// Everything that differs between the rows of a horizontally batched launch (vector pointers, scalar
// operands, reduction shifts) lives in a per-row "row block", so identical op streams over different
// vectors/scalars (e.g. all LIBOR components of one Euler step) run as ONE launch.
#pragma once
#include <stdint.h>

namespace fm {

constexpr int FM_NREG     = 16;    // virtual registers per element lane
constexpr int FM_MAX_OPS  = 96;    // instructions per launch
constexpr int FM_MAX_IN   = 12;    // preloaded input vectors per launch
constexpr int FM_MAX_OUT  = 8;     // materialised output vectors per launch
constexpr int FM_MAX_RED  = 2;     // fused reductions per launch
constexpr int FM_MAX_SCAL = 64;    // scalar operand slots per launch
constexpr int FM_BLOCK    = 256;   // threads per workgroup (4 waves)
constexpr int FM_VEC      = 4;     // elements per thread per tile (one 128-bit access per vector)

// Instruction word: code[0:7] dst[8:11] a[12:15] b[16:19] c[20:23] scalar_slot[24:31]
struct DevOp { uint32_t w; };
static inline uint32_t fm_pack_op(unsigned code, unsigned d, unsigned a, unsigned b, unsigned c, unsigned sslot) {
    return (code & 0xffu) | ((d & 15u) << 8) | ((a & 15u) << 12) | ((b & 15u) << 16) | ((c & 15u) << 20) | ((sslot & 0xffu) << 24);
}

// Row block layout (one per batch row), in units of 8 bytes:
//   [0, n_in)                      const float*  input pointers
//   [n_in, n_in+n_out)             float*        output pointers
//   [.., +n_red)                   double        reduction shifts
//   then n_scal floats (padded to a multiple of 2)
constexpr int FM_ROW_WORDS_MAX = FM_MAX_IN + FM_MAX_OUT + FM_MAX_RED + FM_MAX_SCAL / 2;   // 54 x 8 B

struct DevProgramArgs {
    uint32_t n_ops, n_in, n_out, n_red;
    uint32_t n_scal, row_words;          // row stride in 8-byte words
    uint32_t tiles_per_row, use_inline;  // tiles of FM_BLOCK*FM_VEC elements; batch==1 → row block inline
    int64_t  n;                          // elements per vector
    const uint64_t* rows;                // device table [batch][row_words] (when !use_inline)
    double*  partials;                   // [batch][n_red][gridDim.x][4] block partials of the fused reductions
    uint8_t  out_reg[FM_MAX_OUT];
    uint8_t  red_reg[FM_MAX_RED];
    uint8_t  pad_[6];
    DevOp    ops[FM_MAX_OPS];
    uint64_t inline_row[FM_ROW_WORDS_MAX];
};

struct DevFinalizeArgs {
    const double* partials;   // [batch*n_red][n_blocks][4]
    double*       out;        // [batch*n_red][4]
    uint32_t      n_blocks;
};

} // namespace fm
