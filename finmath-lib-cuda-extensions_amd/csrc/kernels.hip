// kernels.hip — the hand-written gfx950 (CDNA4) kernels of the RandomVariable / BrownianMotion hot path.
//
//   fm_program_kernel   tier 0 of a compiled program (tier 1 = the per-program kernels jit.cpp generates from the same
//                       building blocks, fm_kernel_parts.hpp / fm_device_math.hpp):
//                       one launch = a whole chain of RandomVariable methods over `batch` independent
//                       vector tuples (replaces the reference's one-launch-per-method scheme,
//                       RandomVariableCuda.java:483-557 + the 27 kernels of RandomVariableCudaKernel.cu),
//                       optionally ending in fused {Σ, Σ², min, max} reductions (replaces the
//                       D2H-and-host-loop reductions, RandomVariableCuda.java:830-901).
//   fm_bm_kernel        counter-based Philox4x32-10 + LDS-table-driven inverse-normal-CDF increments (replaces
//                       curandGenerateNormal, BrownianMotionCudaWithRandomVariableCuda.java:168-178).
//   fm_fill_kernel      constant fill.
//
// Byte movers by construction: nothing here is a contraction, MFMA is not used.  Measured: method chains of
// arithmetic ops stream at the HBM ceiling; chains with fp64-evaluated exp/log and the IEEE-only normal generator are
// VALU-bound.
// Design for CDNA4: 64-wide waves, 256-thread workgroups, one 128-bit access per lane per vector
// (1 KiB per wave-instruction, fully coalesced), every input load of a tile issued before the first
// use so ≥ n_in KiB per wave are in flight, virtual registers in VGPRs (s_set_gpr_idx — no scratch,
// no LDS traffic), wave64 shuffles + LDS only for the 4-wave reduction epilogue.
#include <hip/hip_runtime.h>
#include "fm_program.h"
#include "fm_device_math.hpp"
#include "fm_kernel_parts.hpp"
#include "fm_normal_table.hpp"
#include "kernels.h"

namespace fm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x9 __attribute__((ext_vector_type(9)));   // > 8 elements: dynamic indexing stays on s_set_gpr_idx (<= 8 would be expanded into v_cndmask chains)
// ---------------------------------------------------------------------------------------------
// Fused program interpreter
// ---------------------------------------------------------------------------------------------

// One case of the dispatch switch: number of register operands fetched from R = 0, 1 or 2.  E elements per lane.
#define FM_U0(CODE) case CODE:                                                                          \
        _Pragma("unroll") for (int j = 0; j < E; ++j) a[j] = ueval<CODE>(a[j], 0.f, 0.f, s);           \
        break;
// Register-hungry fp64 bodies (exact exp/log): evaluate four elements at a time so that the live fp64 temporaries of
// eight interleaved evaluations do not push the kernel past 128 VGPRs (= below 4 waves per SIMD).
#define FM_U0H(CODE) case CODE:                                                                         \
        _Pragma("unroll") for (int g = 0; g < E; g += 4) {                                              \
            _Pragma("unroll") for (int j = g; j < g + 4; ++j) a[j] = ueval<CODE>(a[j], 0.f, 0.f, s);    \
            __builtin_amdgcn_sched_barrier(0);                                                          \
        }                                                                                               \
        break;
// sqrt: all E elements share one special-case branch (fm_device_math.hpp: sqrt_all)
#define FM_USQRT case U_SQRT: sqrt_all<E>(a); break;
// log: four elements at a time like the other fp64 bodies, their special arguments behind one wave-uniform branch (log_all)
#define FM_ULOG case U_LOG:                                                                             \
        _Pragma("unroll") for (int g = 0; g < E; g += 4) { log_all<4>(a + g); __builtin_amdgcn_sched_barrier(0); }   \
        break;
// (micro-ops with a division stay on `a / b` here: the paired expansion of the specialised tier, ueval_div_all, costs this kernel
// 8 VGPRs — 129 with a fused reduction, i.e. 3 instead of 4 waves per SIMD; the quotients are the same either way)
#define FM_U1(CODE) case CODE: {                                                                        \
        float p[E];                                                                                     \
        _Pragma("unroll") for (int j = 0; j < E; ++j) p[j] = R[j][r1];                                  \
        _Pragma("unroll") for (int j = 0; j < E; ++j) a[j] = ueval<CODE>(a[j], p[j], 0.f, s);           \
        } break;
#define FM_U2(CODE) case CODE: {                                                                        \
        float p[E], q[E];                                                                               \
        _Pragma("unroll") for (int j = 0; j < E; ++j) { p[j] = R[j][r1]; q[j] = R[j][r2]; }             \
        _Pragma("unroll") for (int j = 0; j < E; ++j) a[j] = ueval<CODE>(a[j], p[j], q[j], s);          \
        } break;

// E elements per lane (E/4 tiles of 1024 elements per pass), RegVec = register-file vector (NREG floats).
//   variant 0: E = 4, 16 registers  — programs with many live values
//   variant 1: E = 8,  9 registers  — short programs: twice the work per instruction dispatch and twice the bytes in flight
// NIN_T = compile-time bound of the number of preloaded inputs (the preload loop is unrolled NIN_T times).
template <int NRED, bool INLINE_ROW, int E, int NREG, int NIN_T, typename RegVec>
__global__ void __launch_bounds__(FM_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) fm_program_kernel(const DevProgramArgs A,
                                                               const uint64_t* __restrict__ rows,     // [batch][row_words]
                                                               double* __restrict__ partials)         // [batch][NRED][grid.x][4]
{
    constexpr int T = E / FM_VEC;                   // float4 per lane per pass
    const uint32_t row = blockIdx.y;
    // Row block: wave-uniform, read with SCALAR loads — kernarg segment when INLINE_ROW (small batches), else the row table, which
    // must be a `const __restrict__` kernel parameter of its own: fetched through a pointer stored inside the
    // argument struct the compiler cannot prove it read-only and falls back to vector loads + v_readfirstlane.
    const uint64_t* __restrict__ rowp = (INLINE_ROW ? A.inline_row : rows) + (size_t)row * A.row_words;
    const uint32_t n_in = A.n_in, n_out = A.n_out, n_ops = A.n_ops;
    const int64_t n = A.n;
    const float* __restrict__ scal = reinterpret_cast<const float*>(rowp + n_in + n_out + A.n_red);
    if (A.flags & FM_ARGS_LOG_TABLE) log_table_init();          // wave- and workgroup-uniform

    constexpr int NR = NRED > 0 ? NRED : 1;
    double acc_sum[NR], acc_sq[NR], shift[NR];
    float  acc_min[NR], acc_max[NR];
    unsigned long long nan_mask[NR];
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
        acc_sum[r] = 0.0; acc_sq[r] = 0.0; nan_mask[r] = 0ull;
        acc_min[r] = __builtin_huge_valf(); acc_max[r] = -__builtin_huge_valf();
        shift[r] = reinterpret_cast<const double*>(rowp)[n_in + n_out + r];
    }

    // Virtual register file, one vector per element lane (static indices only).  Deliberately NOT zero-initialised:
    // a register is always written (preload / store) before the program reads it, and clearing 80 VGPRs per workgroup
    // cost 10 VALU instructions per element.  The empty asm gives the vectors a defined (arbitrary) value.
    RegVec R[E];
#pragma unroll
    for (int j = 0; j < E; ++j) asm volatile("" : "=v"(R[j]));

    // A.tiles_per_row counts passes of FM_BLOCK*E elements.
    // workgroup b takes the tiles [b·P, (b+1)·P) — the same assignment as the specialised kernels (jit.cpp)
    const uint32_t tiles_per_block = A.block_tiles;
    const uint32_t tile_begin = blockIdx.x * tiles_per_block;
    const uint32_t tile_end = tile_begin + tiles_per_block < A.tiles_per_row ? tile_begin + tiles_per_block : A.tiles_per_row;
    for (uint32_t tile = tile_begin; tile < tile_end; ++tile) {
        // No divergent branch around the interpreter (it would make the compiler fetch instruction words and scalar
        // operands with VECTOR loads — measured: one global_load + 500 cycles per micro-op).  Lanes past the end read
        // element 0 and are masked at the stores / reductions.  Vectors are padded to 256 B, so a partially valid
        // float4 is in bounds.
        // 32-bit float4 indices (the host limits a vector to 2^31 elements): half the address registers of int64.
        uint32_t i4[T], i4c[T];
        bool lane_valid[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            i4[t] = (tile * T + t) * FM_BLOCK + threadIdx.x;            // float4 index inside the vector, unit stride across lanes
            lane_valid[t] = (int64_t)i4[t] * FM_VEC < n;
            i4c[t] = lane_valid[t] ? i4[t] : 0u;
        }

        // ---- preload: every input vector, 16 B per lane and tile, all loads in flight before the first use.
        // The conditional part is confined to the loaded floats; the insertion into the register-file vectors is
        // unconditional (a conditional insert makes every file register a PHI at each branch and wrecks register
        // allocation: 253 VGPRs + scratch instead of ~110).
#pragma unroll
        for (int k = 0; k < NIN_T; ++k) {
            {
                f32x4 v[T];
#pragma unroll
                for (int t = 0; t < T; ++t) v[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
                if (k < (int)n_in) {
                    const gfloat4* __restrict__ p = reinterpret_cast<const gfloat4*>(rowp[k]);
#pragma unroll
                    for (int t = 0; t < T; ++t) v[t] = load_stream(p, i4c[t]);
                }
#pragma unroll
                for (int t = 0; t < T; ++t) { R[4 * t + 0][k] = v[t].x; R[4 * t + 1][k] = v[t].y; R[4 * t + 2][k] = v[t].z; R[4 * t + 3][k] = v[t].w; }
            }
        }

        // ---- interpret: one wave-uniform decode per micro-op, E elements per lane, accumulator in a[].
        float a[E];
#pragma unroll
        for (int j = 0; j < E; ++j) asm volatile("" : "=v"(a[j]));     // the first micro-op of every program is U_LDA
        // Two-deep software pipeline of the scalar fetches: while micro-op pc executes, the scalar operand of pc+1 and
        // the instruction word of pc+2 are in flight (ops[] carries two slack entries).
        uint32_t w = A.ops[0].w;
        uint32_t w1 = A.ops[1].w;
        float s = scal[w >> 24];
        for (uint32_t pc = 0; pc < n_ops; ++pc) {
            const float s_next = scal[w1 >> 24];
            const uint32_t w2 = A.ops[pc + 2].w;
            const uint32_t code = w & 0xffu;
            const uint32_t r1 = (w >> 8) & 15u, r2 = (w >> 12) & 15u, st = (w >> 16) & 15u;
            switch (code) {
                FM_U1(U_LDA)
                FM_U0(U_SQUARED) FM_USQRT FM_U0H(U_EXP) FM_ULOG FM_U0(U_INVERT) FM_U0(U_ABS)
                FM_U0(U_ISNAN) FM_U0(U_EXP_FAST) FM_U0(U_LOG_FAST)
                FM_U0(U_CAP_S) FM_U0(U_FLOOR_S) FM_U0(U_ADD_S) FM_U0(U_SUB_S) FM_U0(U_BUS_S) FM_U0(U_MULT_S)
                FM_U0(U_DIV_S) FM_U0(U_VID_S)
                FM_U1(U_CAP) FM_U1(U_FLOOR) FM_U1(U_ADD) FM_U1(U_MULT) FM_U1(U_SUB) FM_U1(U_BUS) FM_U1(U_DIV) FM_U1(U_VID)
                FM_U1(U_ACCRUE_A) FM_U1(U_ACCRUE_B) FM_U1(U_DISCOUNT_A) FM_U1(U_DISCOUNT_B)
                FM_U1(U_ADDPRODUCT_VS_A) FM_U1(U_ADDPRODUCT_VS_B)
                FM_U2(U_ADDPRODUCT_A) FM_U2(U_ADDPRODUCT_B) FM_U2(U_ADDRATIO_A) FM_U2(U_SUBRATIO_A)
                FM_U2(U_CHOOSE_T) FM_U2(U_CHOOSE_P) FM_U2(U_CHOOSE_N)
                default:
                    // pow / sin / cos call out-of-line fp64 library code; a call site inside the 8-element kernel would
                    // add the callee's ~40 VGPRs to 96 live ones (→ 3 waves/SIMD for everybody).  Programs that use
                    // them are compiled for the 4-element / 16-register variant (runtime.cpp: compile()).
                    if constexpr (E == 4) {
                        switch (code) {
                            FM_U0(U_SIN) FM_U0(U_COS)
                            case U_POW_S: pow_all<E>(a, s); break;       // wave-uniform exponent: the common ones take code of their own (fm_device_math.hpp)
                            default: break;
                        }
                    }
                    break;
            }
            if (st != (uint32_t)(NREG - 1)) {       // the last register of the file means "no store"
#pragma unroll
                for (int j = 0; j < E; ++j) R[j][st] = a[j];
            }
            w = w1; w1 = w2; s = s_next;
        }

        // ---- materialise the escaping values
        for (uint32_t k = 0; k < n_out; ++k) {
            const uint32_t reg = A.out_reg[k];
            gfloat4* __restrict__ q = reinterpret_cast<gfloat4*>(rowp[n_in + k]);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 v = { R[4 * t + 0][reg], R[4 * t + 1][reg], R[4 * t + 2][reg], R[4 * t + 3][reg] };
                if (lane_valid[t]) store_stream(q, i4[t], v);
            }
        }

        // ---- fused reductions (fp64 accumulation of fp32 values, as the twin does: :325-333, :373-381).
        // min/max use the hardware v_min_f32/v_max_f32 (which order -0 < +0 like java.lang.Math.min/max) and track NaN
        // separately in a wave-level ballot mask (scalar registers): NaN anywhere ⇒ the reduction result is NaN.
        const bool pass_full = ((int64_t)tile * T + T) * (FM_BLOCK * FM_VEC) <= n;      // wave-uniform: no masking needed
#pragma unroll
        for (int r = 0; r < NRED; ++r) {
            const uint32_t reg = A.red_reg[r];
            float x[E];
#pragma unroll
            for (int j = 0; j < E; ++j) x[j] = R[j][reg];
            red_accumulate<E>(x, shift[r], pass_full, i4, n, acc_sum[r], acc_sq[r], acc_min[r], acc_max[r], nan_mask[r]);
        }
        // ---- the reduction tree's unit / span bookkeeping (fm_kernel_parts.hpp): lane values through LDS, one wave per unit
        if constexpr (NRED > 0) red_tile_end<NRED, E>(tile - tile_begin, tile + 1u == tile_end, acc_sum, acc_sq, acc_min, acc_max, nan_mask, shift);
    }

    // ---- one partial per workgroup; the last workgroup of the row adds them
    if constexpr (NRED > 0) block_combine<NRED>(partials, row, A.results, A.counters + (size_t)row * FM_COUNTER_STRIDE, A.done_flag, A.done_value, A.span_blocks);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + inverse-normal-CDF increments.  Normative definition: oracle/philox_normal.c — the operation order
// below must not be changed without changing that file (bit-for-bit parity).
//
// Write-only: 4 B per normal, nothing read.  The budget of a byte mover at the 6.2–6.4 TB/s streaming ceiling is ≈ 0.65 ns
// per normal and SIMD; measured without memory traffic (benchmarks/valu_cost.hip): Philox4x32-10 alone 0.32 ns (10 rounds of
// two v_mad_u64_u32), round 1's IEEE-only Box–Muller on top 1.17 ns in all (2.3–2.8 TB/s written, VALU-bound).  The
// transform is therefore an LDS-table-driven inverse CDF by hierarchical segmentation: leading-zero count (the octave of the
// tail) and three mantissa bits pick one of 256 cubics — one v_ffbh_u32, four integer instructions, one int→float conversion,
// one ds_read_b128, three FMAs, one v_bfi for the sign: ≈ 12 VALU instructions per normal instead of ≈ 45.
// ---------------------------------------------------------------------------------------------

// gfx950: any three-input boolean function is ONE instruction (v_bitop3_b32); 0x96 = a ^ b ^ c
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0x96); }

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        // one 32x32→64 multiply per pair (v_mad_u64_u32) instead of a quarter-rate v_mul_hi_u32 + v_mul_lo_u32 each
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = xor3(hi1, c1, k0), n2 = xor3(hi0, c3, k1);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One fp32 FMA that the SLP vectoriser cannot pair up (it packed the cubics of two normals into v_pk_fma_f32 at the price of
// five v_mov per pair — more instructions than the three plain FMAs it replaced).
__device__ __forceinline__ float fma1(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

// One standard normal from one 32-bit word; `table` = the 256 x {c0, c1, c2, c3} cubics in LDS (fm_normal_table.hpp).
// 12 VALU instructions + one ds_read_b128.
__device__ __forceinline__ float spec_normal(uint32_t w, const f32x4* table)
{
    const uint32_t k = (w << 1) | 1u;                               // odd: p = k·2^-33 in (0, 1/2)            v_lshl_or_b32
    const uint32_t lz = (uint32_t)__builtin_clz(k);                 // k != 0                                   v_ffbh_u32
    const uint32_t norm = k << lz;
    const uint32_t idx = (norm >> 28) & 7u;
    const float tf = (float)(norm & 0x0FFFFFFFu);                   // v_cvt_f32_u32: round to nearest even, like the oracle's cast
    const f32x4 c = table[lz * 8u + idx];                           // one ds_read_b128
    float m = fma1(c.w, tf, c.z);
    m = fma1(m, tf, c.y);
    m = fma1(m, tf, c.x);
    uint32_t z;                                                     // copysign(|z|, bit 31 of w): magnitude bits of m, sign bit of w
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(z) : "s"(0x7FFFFFFFu), "v"(m), "v"(w));     // (VOP3 on gfx9 takes no literal: the mask sits in an SGPR)
    return __uint_as_float(z);
}

__device__ __forceinline__ void normal4(uint32_t k0, uint32_t k1, uint64_t pb, uint32_t stream, const f32x4* table, float z[4])
{
    uint32_t r[4];
    philox4x32_10((uint32_t)pb, (uint32_t)(pb >> 32), stream, 0x464D4850u, k0, k1, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = spec_normal(r[j], table);
}

// grid = (path tiles of FM_BM_TILE float4, groups of FM_BM_STREAMS streams).  Vector `stream` lives at slab + stream * stride_floats.
// A workgroup copies the 4 KB table into LDS once and then writes FM_BM_STREAMS x FM_BM_TILE float4 = 64 KB of increments.
constexpr int FM_BM_PASSES = 2;                                     // float4 per lane and stream
constexpr int FM_BM_TILE = FM_BLOCK * FM_BM_PASSES;                 // float4 per workgroup and stream (2048 paths)
constexpr int FM_BM_STREAMS = 8;                                    // streams (step x factor vectors) per workgroup
static_assert(FM_NORMAL_TABLE_ENTRIES == FM_BLOCK, "one table entry per thread in the LDS copy");

// sq_stream[s] = (float)sqrt(dt) of local stream s — a `const __restrict__` parameter of its own so that it is read with SCALAR loads:
// fetched through the argument struct it became one global_load per stream, and waiting for a vector load waits for every older
// STORE of the wave as well (vmcnt counts them together, in order) — the write stream drained once per stream (4.97 TB/s).
__global__ void __launch_bounds__(FM_BLOCK) fm_bm_kernel(const DevBmArgs A, const float* __restrict__ sq_stream, const uint32_t n_streams)
{
    __shared__ f32x4 table[FM_NORMAL_TABLE_ENTRIES];
    table[threadIdx.x] = reinterpret_cast<const f32x4*>(FM_NORMAL_TABLE)[threadIdx.x];
    __syncthreads();
    const int64_t n4 = (A.n_paths + 3) >> 2;
    const uint32_t s_begin = blockIdx.y * FM_BM_STREAMS;
    const uint32_t s_end = s_begin + FM_BM_STREAMS < n_streams ? s_begin + FM_BM_STREAMS : n_streams;
    int64_t i4[FM_BM_PASSES];
    uint64_t g0[FM_BM_PASSES];
#pragma unroll
    for (int t = 0; t < FM_BM_PASSES; ++t) {
        i4[t] = (int64_t)blockIdx.x * FM_BM_TILE + t * FM_BLOCK + threadIdx.x;
        g0[t] = (uint64_t)(A.path_offset + i4[t] * 4);             // global index of this lane's first path
    }
    const bool aligned = (A.path_offset & 3) == 0;                  // wave-uniform
    for (uint32_t stream = s_begin; stream < s_end; ++stream) {
        const float sq = sq_stream[stream];
        f32x4* __restrict__ out = reinterpret_cast<f32x4*>(A.slab + (size_t)stream * A.stride_floats);
        const uint32_t gstream = A.stream0 + stream;
#pragma unroll
        for (int t = 0; t < FM_BM_PASSES; ++t) {
            float z[4];
            if (aligned) {
                normal4(A.key0, A.key1, g0[t] >> 2, gstream, table, z);
            } else {                                               // shard offset not a multiple of 4: two blocks
                float za[4], zb[4];
                normal4(A.key0, A.key1, g0[t] >> 2, gstream, table, za);
                normal4(A.key0, A.key1, (g0[t] >> 2) + 1, gstream, table, zb);
                const uint32_t sft = (uint32_t)(g0[t] & 3u);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t idx = sft + j;
                    float v = 0.0f;
#pragma unroll
                    for (int m = 0; m < 4; ++m) { if (idx == (uint32_t)m) v = za[m]; if (idx == (uint32_t)(m + 4)) v = zb[m]; }
                    z[j] = v;
                }
            }
            // written once, read by a later launch long after it has left the caches (4 GB at config 3): streaming store
            if (i4[t] < n4) __builtin_nontemporal_store(f32x4{ sq * z[0], sq * z[1], sq * z[2], sq * z[3] }, out + i4[t]);
        }
    }
}

__global__ void __launch_bounds__(FM_BLOCK) fm_fill_kernel(float4* __restrict__ p, float v, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * FM_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * FM_BLOCK)
        p[i] = make_float4(v, v, v, v);
}

// one lane per {Σ, Σ², min, max} block: 32 bytes from host-coherent memory (written earlier in this stream by the launches that took the
// moments) to out[i]
__global__ void __launch_bounds__(64) fm_gather_moments_kernel(const DevGatherArgs A, double* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= A.count) return;
    const volatile double* s = reinterpret_cast<const volatile double*>(A.src[i]);
    const double v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
    out[(size_t)i * 4 + 0] = v0; out[(size_t)i * 4 + 1] = v1; out[(size_t)i * 4 + 2] = v2; out[(size_t)i * 4 + 3] = v3;
}

// The moments of `world` path shards of `count` vectors, [shard][vector][Σ, Σ², min, max] (the receive buffer of an all-gather), combined
// per vector by the rule of fmhip_expectation_combine (abi.cpp: combine_moments): sums added in shard order — the same bits on every
// device, whatever algorithm the collective used —, java.lang.Math.min / max (NaN-propagating, -0.0 < +0.0).  One lane per vector.
__global__ void __launch_bounds__(64) fm_combine_moments_kernel(const double* __restrict__ gathered, const uint32_t world, const uint32_t count, double* __restrict__ out)
{
    const uint32_t k = blockIdx.x * 64u + threadIdx.x;
    if (k >= count) return;
    double sum = gathered[(size_t)k * 4 + 0], sumsq = gathered[(size_t)k * 4 + 1], mn = gathered[(size_t)k * 4 + 2], mx = gathered[(size_t)k * 4 + 3];
    for (uint32_t r = 1; r < world; ++r) {
        const double* g = gathered + ((size_t)r * count + k) * 4;
        sum += g[0]; sumsq += g[1];
        const double gmin = g[2], gmax = g[3];
        mn = (mn != mn || gmin != gmin) ? __builtin_nan("") : (gmin < mn || (gmin == mn && __builtin_signbit(gmin))) ? gmin : mn;
        mx = (mx != mx || gmax != gmax) ? __builtin_nan("") : (gmax > mx || (gmax == mx && !__builtin_signbit(gmax))) ? gmax : mx;
    }
    out[(size_t)k * 4 + 0] = sum; out[(size_t)k * 4 + 1] = sumsq; out[(size_t)k * 4 + 2] = mn; out[(size_t)k * 4 + 3] = mx;
}

// ---------------------------------------------------------------------------------------------
// Host-side launchers (the only functions the runtime calls)
// ---------------------------------------------------------------------------------------------

template <int NRED, int E, int NREG, int NIN_T, typename RegVec>
static hipError_t launch_program_v(const DevProgramArgs& a, const uint64_t* rows, double* partials, dim3 grid, hipStream_t st)
{
    if (a.use_inline) hipLaunchKernelGGL((fm_program_kernel<NRED, true, E, NREG, NIN_T, RegVec>),  grid, dim3(FM_BLOCK), 0, st, a, rows, partials);
    else              hipLaunchKernelGGL((fm_program_kernel<NRED, false, E, NREG, NIN_T, RegVec>), grid, dim3(FM_BLOCK), 0, st, a, rows, partials);
    return hipGetLastError();
}

template <int NRED>
static hipError_t launch_program_nred(const DevProgramArgs& a, const uint64_t* rows, double* partials, dim3 grid, hipStream_t st)
{
    if (a.variant == 1) {
        if (a.n_in <= 3) return launch_program_v<NRED, 8, 9, 3, f32x9>(a, rows, partials, grid, st);
        return launch_program_v<NRED, 8, 9, 8, f32x9>(a, rows, partials, grid, st);
    }
    return launch_program_v<NRED, 4, 16, FM_MAX_IN, f32x16>(a, rows, partials, grid, st);
}

hipError_t launch_program(const DevProgramArgs& a, const uint64_t* rows, double* partials,
                          uint32_t blocks_per_row, uint32_t batch, hipStream_t st)
{
    const dim3 grid(blocks_per_row, batch, 1);
    switch (a.n_red) {
    case 0:  return launch_program_nred<0>(a, rows, partials, grid, st);
    case 1:  return launch_program_nred<1>(a, rows, partials, grid, st);
    case 2:  return launch_program_nred<2>(a, rows, partials, grid, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_bm(const DevBmArgs& a, uint32_t n_streams, hipStream_t st)
{
    const int64_t n4 = (a.n_paths + 3) >> 2;
    int64_t bx = (n4 + FM_BM_TILE - 1) / FM_BM_TILE;                // n_paths <= 2^31: at most 2^20 tiles
    if (bx < 1) bx = 1;
    const uint32_t by = (n_streams + FM_BM_STREAMS - 1) / FM_BM_STREAMS;
    hipLaunchKernelGGL(fm_bm_kernel, dim3((uint32_t)bx, by, 1), dim3(FM_BLOCK), 0, st, a, a.sqrt_dt, n_streams);
    return hipGetLastError();
}

hipError_t launch_fill(float* p, float v, int64_t n_padded, hipStream_t st)
{
    const int64_t n4 = n_padded >> 2;
    int64_t bx = (n4 + FM_BLOCK - 1) / FM_BLOCK;
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(fm_fill_kernel, dim3((uint32_t)bx), dim3(FM_BLOCK), 0, st, reinterpret_cast<float4*>(p), v, n4);
    return hipGetLastError();
}

// Every kernel of this library made resident NOW (Engine::init): the HIP runtime loads device code lazily, at a kernel's first launch,
// and that load needs device memory — a caller whose pool has grown to the size of the device by then (handles released late, by a
// garbage collector) met "no kernel image is available" at the interpreter tier's first launch, thousands of launches into its run.
template <int NRED> static hipError_t preload_program_nred()
{
    hipFuncAttributes at;
    hipError_t e = hipSuccess;
    auto one = [&](const void* f) { if (e == hipSuccess) e = hipFuncGetAttributes(&at, f); };
    one(reinterpret_cast<const void*>(&fm_program_kernel<NRED, true, 8, 9, 3, f32x9>));   one(reinterpret_cast<const void*>(&fm_program_kernel<NRED, false, 8, 9, 3, f32x9>));
    one(reinterpret_cast<const void*>(&fm_program_kernel<NRED, true, 8, 9, 8, f32x9>));   one(reinterpret_cast<const void*>(&fm_program_kernel<NRED, false, 8, 9, 8, f32x9>));
    one(reinterpret_cast<const void*>(&fm_program_kernel<NRED, true, 4, 16, FM_MAX_IN, f32x16>)); one(reinterpret_cast<const void*>(&fm_program_kernel<NRED, false, 4, 16, FM_MAX_IN, f32x16>));
    return e;
}
hipError_t preload_kernels()
{
    hipFuncAttributes at;
    hipError_t e = preload_program_nred<0>();
    if (e == hipSuccess) e = preload_program_nred<1>();
    if (e == hipSuccess) e = preload_program_nred<2>();
    if (e == hipSuccess) e = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&fm_bm_kernel));
    if (e == hipSuccess) e = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&fm_fill_kernel));
    if (e == hipSuccess) e = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&fm_gather_moments_kernel));
    if (e == hipSuccess) e = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&fm_combine_moments_kernel));
    return e;
}

hipError_t launch_combine_moments(const double* gathered, uint32_t world, uint32_t count, double* out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(fm_combine_moments_kernel, dim3((count + 63u) / 64u), dim3(64), 0, st, gathered, world, count, out);
    return hipGetLastError();
}

hipError_t launch_gather_moments(const DevGatherArgs& a, double* out, hipStream_t st)
{
    if (a.count == 0) return hipSuccess;
    hipLaunchKernelGGL(fm_gather_moments_kernel, dim3((a.count + 63u) / 64u), dim3(64), 0, st, a, out);
    return hipGetLastError();
}

} // namespace fm
