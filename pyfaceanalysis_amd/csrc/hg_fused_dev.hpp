// Device-side definitions shared by the fused-plan translation units (hg_fused.hip: planner, executor,
// mid/top layer kernels; hg_fused_front.hip: first-layer kernels; hg_fused_igsfa.hip: iGSFA kernels).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hg_common.hpp"

namespace hg {
namespace fused {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxMT = 4;            // up to 64 outputs per affine in the fused plan

// Every stage is padded to a UNIFORM node structure (same K-block count, tile counts and
// expansion list for all its nodes; missing pieces are zero weights), so all weight / bias
// addresses are arithmetic on the node index and the only per-node table is the list of source
// blocks of GEMM 1.
constexpr int kMaxFuncs = 4;

struct DChunk {   // stage 0: a group of consecutive nodes whose input columns share one LDS tile
    int32_t node_begin, node_count, run_begin, run_count, n_cols, piece_begin, n_pieces, pad1;
};
struct DRun {
    int32_t start, len, lds_off, pad;
};

// k_stage01p: two LDS tiles (one barrier per tile group) or one (two barriers, half the LDS).  With the packed
// tile layout (128 + 4 words per row) both fit three workgroups per CU and measure the same (152-154 us).
constexpr bool kDoubleBuffer01 = true;

struct StageParams {
    const f32x4* afrag;   // [node][ A1: kb1 x MT1 | A2: MT1 x nf x MT2 ] blocks of 64 x f32x4
    const float* bias;    // [node][ (MT1 + MT2) x 16 ]
    const int2* kb1tab;   // [node][kb1] {source block, k-steps}           (stages > 0)
    const f32x4* in;      // input activation, fragment order               (stages > 0)
    f32x4* out;
    int32_t n_nodes, kb1, nf, has_exp;
    int32_t node_blocks, bias_floats, n_tiles, nb_in, nb_out, mto;
    int32_t nodes_per_group, nodes_per_wg, n_chunks, tile_groups, tile_parts;   // k_igsfa: nodes_per_wg = waves of a workgroup that take batch tiles
    uint32_t nk2p[kMaxMT];     // per z tile: 4 bits of k-steps per expansion function
    uint32_t funcp;            // 4 bits of ExpKind per expansion function
    float expo[kMaxFuncs];
    // stage 0
    const DChunk* chunks;
    const DRun* runs;
    const int2* piece_col;     // per chunk piece (4 columns) -> {first source column, LDS word offset}
    const int32_t* koff;       // [node*kb1 + kb][g][r] LDS word offsets
    const float* kmean;        // same shape: means subtracted by the loader
    const int32_t* kcol;       // k_stage01d: [node][g] first of the four contiguous source columns of lane group g
    const void* x;
    int64_t ldx, n_rows;
    int32_t lds_stride, nk_last, vec4, contig4;
    int32_t ig_has_lr, ig_folded;   // k_igsfa: residual GEMM present; folded form (first GEMM covers all output tiles, no others)
    unsigned long long* stamps;   // diagnostic build only (HIGSFA_STAMP): per-wave cycle stamps
    // Packed remainder tiles (stages whose output ends in a tile of <= 4 real rows, 4x4 form): the last tile of FOUR sibling
    // nodes shares one block — node n's row values in register pack_slot[n] % 4 of block pack_base + pack_slot[n] / 4 — instead of one block each;
    // the full tiles of node n start at block n * (mto - 1).  pack_base > 0 switches it on (producer side); a consumer sees
    // it only through its K-block table (source block, first k-step, k-steps).
    // Round 5: where the consuming stage runs the whole-visit-prefetch instantiation of k_stage (the packed block sits at a position
    // of the node's K-block list that is known at compile time), a packed block is stored SLOT-MAJOR — float [slot][lane], slot s =
    // 256 contiguous bytes (pack_soa) — so that a node's store of its remainder rows, one float per lane, covers two whole 128-byte
    // lines.  In the lane-major form (float [lane][slot]) the four siblings each write 4 of every 16 bytes of all eight lines of
    // the block and every line goes to HBM four times (profiles/r04_traffic.json: the front kernel wrote 204.5 MB for a 167.8 MB
    // buffer).  The consumer reads such a block one register at a time, 256 contiguous bytes per load, and only the two slots it
    // multiplies (load_kblock_soa): the other half of the block, its cousins' rows, is never fetched.
    int32_t pack_base;
    int32_t pack_soa;             // producer side: packed blocks are written slot-major
    int32_t pair_chunks;          // k_stage: block -> chunk map that puts chunks 2c, 2c + 1 on one XCD (lane-major packed output of two-node chunks)
    int32_t pack_in;              // consumer side: source blocks >= pack_in of the input are slot-major packed blocks (INT32_MAX: none)
    const int32_t* pack_slot;     // [node] -> 4 * (shared block) + register: siblings under one parent of the next layer share a block
    int32_t a4x4;                 // k_stage01p: layer-1 remainder fragments are already stored in 4x4 form
    // k_stage_prod (hg_fused_prod.hip): table-driven expansion (products, clip)
    const int2* etab;             // [neb][16] {kind << 16 | k << 8 | i, exponent bits}
    int32_t neb, has_clip;
    float clip_lo, clip_hi;
    // Dynamic tile-group queue of the persistent sweep kernels: one counter per node chunk (16 words apart).  A workgroup's
    // first tile group is its `part`; every further one is tile_parts + (atomicAdd(counter, 1) - work_base), taken two
    // iterations ahead so that the atomic's round trip is never waited for.  Counters are never reset: every workgroup
    // stops at its first failing grab, so a launch advances each counter by exactly tile_groups and the host adds that
    // to work_base (32-bit wrap-around is harmless).  Why: the SIMD arbiter favours the oldest wave, so with a static
    // split the first workgroup of a CU finishes 25 % before the third and the kernel ends with a third of its waves.
    uint32_t* work_ctr;
    uint32_t work_base;
    int32_t* err;       // host-visible error word: a bounded poll that ran out writes here (checked by the host at the next call)
    int32_t whatif;     // diagnostic build only (HIGSFA_WHATIF): bit 0 = k_stage reads every input block from the tile's first block (cache-hot),
                        // bit 1 = node_tail stores nothing — timing experiments, results are wrong
};

__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ float pow_abs(float v, float p) {
    // |v|^p = exp2(p * log2|v|); v = 0 -> log2 = -inf -> exp2 = 0 exactly
    return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(__builtin_fabsf(v)));
}

__device__ __forceinline__ f32x4 pow_abs4(f32x4 z, float p) {
    f32x4 e;
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = pow_abs(z[r], p);
    return e;
}

// Element-wise expansion function on an accumulator tile; `func` is wave-uniform.
// Two forms, chosen per kernel by measurement (same box, tools/ab_build.sh):
//  * apply_func: the compiler if-converts the three cases, i.e. it evaluates the power for every function of the
//    expansion, identity included (3.5x the v_log / v_exp in the front kernel's loop) — and the big sweep kernels are 2-3 %
//    FASTER that way than with the branches below (straight-line code between the MFMA runs schedules better; the
//    transcendental unit is otherwise idle);
//  * apply_func_uniform: identity costs nothing, one v_log + v_exp per value behind a real scalar branch (the empty asm
//    statement cannot be speculated); the latency-bound small kernels (k_stage_splitm, k_tail) gain 10 % from it.
__device__ __forceinline__ f32x4 apply_func(int func, float expo, f32x4 z) {
    f32x4 e;
    if (func == (int)E_IDENTITY) {
        e = z;
    } else if (func == (int)E_ABS_POW) {
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = pow_abs(z[r], expo);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_copysignf(pow_abs(z[r], expo), z[r]);
    }
    return e;
}

__device__ __forceinline__ f32x4 apply_func_uniform(int func, float expo, f32x4 z) {
    if (func == (int)E_IDENTITY) return z;
    asm volatile("" ::: "memory");
    f32x4 e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float p = pow_abs(z[r], expo);
        e[r] = func == (int)E_SIGNED_POW ? __builtin_copysignf(p, z[r]) : p;
    }
    return e;
}

#ifndef HG_POW_PREFETCH
#define HG_POW_PREFETCH 1      // (A/B switch: HIGSFA_CXXFLAGS=-DHG_POW_PREFETCH=0)
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// One K-block: acc[mt][t] += A[mt] (16 x 16, four k-steps) * B[t].  `wp` points at the block's first
// A fragment (+lane); fragments of consecutive m-tiles are 64 f32x4 apart.
// k-steps r0 .. nk-1 of the block are multiplied (r0 > 0: a packed remainder block, whose leading k-steps belong to other nodes).
template <int MT, int T, bool KOUT = false, typename WP>
__device__ __forceinline__ void gemm_block(WP wp, const f32x4 (&b)[T], f32x4 (&acc)[MT][T], int nk, int r0 = 0) {
#ifndef HG_GEMM4_KOUTER      // (A/B switch: tools/ab_build.sh "-DHG_GEMM4_KOUTER")
    if constexpr (MT >= 4 && T >= 2 && !KOUT) {
        // m-tile outer: one A fragment live (4 registers instead of 16).  In isolation this order is the slower one (a branch
        // and an exposed LDS read per pair of MFMAs: 88 % of the MFMA peak against 96 %, tools/ubench/mfma_loop.hip V5 / V4),
        // but inside the generic k_stage<4,4,2> — 116 VGPRs this way, 127 the other — it measures 3 % FASTER on layers 3-5.
        // (KOUT: the compile-time-expansion instantiations have registers to spare — 106 / 117 — and gain 0.3 % from the other order.)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 a = wp[mt * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r >= r0 && r < nk) {
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[mt][t] = MFMA16(a[r], b[t][r], acc[mt][t]);
                }
        }
        return;
    }
#endif
    {
        // k-step outer, m-tile inner: ONE branch per k-step and MT x T MFMAs behind it, all A fragments of the block read
        // from LDS up front (layer 2: 112 -> 105 us)
        f32x4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = wp[mt * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r >= r0 && r < nk) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[mt][t] = MFMA16(a[mt][r], b[t][r], acc[mt][t]);
            }
    }
}

// The same with the block's A fragments already in registers (k-step outer): for callers that load the NEXT block's fragments
// from LDS before they multiply the current one (k_stage's K-block loop, round 4), so that no MFMA waits for an LDS read.
template <int MT, int T>
__device__ __forceinline__ void gemm_block_regs(const f32x4 (&a)[MT], const f32x4 (&b)[T], f32x4 (&acc)[MT][T], int nk, int r0 = 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r >= r0 && r < nk) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[mt][t] = MFMA16(a[mt][r], b[t][r], acc[mt][t]);
        }
}

// Remainder tiles.  When an affine has 16 m + (1..4) outputs its last 16-row tile holds at most four real rows
// (features 16 m + i sit at rows 4 i = lane group g, register 0).  Such a tile runs on
// v_mfma_f32_4x4x1_16B_f32 instead — 16 independent 4x4 blocks per instruction, 8 cycles instead of 32:
// block b = lane / 4 takes the k index g = lane / 16 of the k-step and sub-images 4 (b % 4) .. + 3, so the B
// operand is the very same register as for the 16x16 form; the A fragment is stored in "4x4 form" (lane
// (g, i = lane % 4) holds row 4 i of lane group g of the ordinary fragment — permuted on the host for the
// LDS-resident weights, with ds_bpermute for the register-resident ones); the four k partial sums (lane
// groups) are added with two cross-lane steps and lane group g keeps row g.  Same products, 3/4 of the
// padding multiplications of those tiles gone.
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// The four k partial sums of a 4x4-form tile sit in the four lane groups (rows of 16 lanes), one register per remainder row; lane
// group g wants the total of register g.  A reduce-scatter on the vector ALU in three swaps and three adds (round 4; it was a
// full sum of every register in every row — eight swaps, eight adds and a select chain): v_permlane16_swap with d0 / d1
// exchanges the odd rows of one with the even rows of the other, so one add leaves (row 0 + row 1) of d0 in the even rows and of
// d1 in the odd rows; the same for d2 / d3; v_permlane32_swap of the two results exchanges the halves, and the last add leaves
// the total of d_g in row g.  Every total is (r0 + r1) + (r2 + r3) with the operands in the old order: the same bits.
__device__ __forceinline__ float rem4_total(f32x4 d) {
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[0]), __float_as_uint(d[1]), false, false);
    const float t = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[2]), __float_as_uint(d[3]), false, false);
    const float u = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(u), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The same totals through LDS (round 5): the wave writes its four partial-sum registers lane-major (one ds_write_b128) and every lane reads the
// register of ITS lane group from the four lane groups (four conflict-free ds_read_b32), then (v0 + v1) + (v2 + v3) — rem4_total's operands and
// association, hence its bits.  LDS instructions do not go through the port the MFMAs share with the vector ALU; the three permlane swaps
// do, at ~19 cycles each (tools/ubench/mfma_loop.hip V6 / V8: 65 against 41 cycles per reduction with every wave of a SIMD reducing at once).
// scratch: 256 floats of this wave's own.
__device__ __forceinline__ float rem4_total_lds(f32x4 d, float* scratch, int lane) {
    *(f32x4*)(scratch + lane * 4) = d;
    const int g = lane >> 4, j = lane & 15;
    const float v0 = scratch[(0 * 16 + j) * 4 + g], v1 = scratch[(1 * 16 + j) * 4 + g], v2 = scratch[(2 * 16 + j) * 4 + g], v3 = scratch[(3 * 16 + j) * 4 + g];
    return (v0 + v1) + (v2 + v3);
}

// acc (a remainder tile's accumulator: row g of the tile = register 0 of lane group g) += the 4x4-form partial sums d
__device__ __forceinline__ void add_rem4(f32x4& acc, f32x4 d) { acc[0] += rem4_total(d); }

// gemm_block with the last m-tile in 4x4 form: tiles 0 .. MT-2 accumulate in acc, the last one in d4.
template <int MT, int T, typename WP>
__device__ __forceinline__ void gemm_block_rem(WP wp, const f32x4 (&b)[T], f32x4 (&acc)[MT][T], f32x4 (&d4)[T], int nk, int r0 = 0) {
    f32x4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = wp[mt * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r >= r0 && r < nk) {
#pragma unroll
            for (int mt = 0; mt < MT - 1; ++mt)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[mt][t] = MFMA16(a[mt][r], b[t][r], acc[mt][t]);
#pragma unroll
            for (int t = 0; t < T; ++t) d4[t] = MFMA4(a[MT - 1][r], b[t][r], d4[t]);
        }
}

// Slot-major packed K-block (StageParams::pack_soa) for T batch tiles.  Table entry y = k-steps | first slot << 16: the block's A
// fragments are stored with this node's k-steps FIRST (the planner shifts them), so slot0 + r is loaded into register r and the
// multiplication runs k-steps 0 .. nk-1 like any other block.  TWO: at most two k-steps (plan-time condition of the static path).
template <int T, bool TWO>
__device__ __forceinline__ void load_kblock_soa(const StageParams& P, const uint32_t (&row)[T], int sb, int y, int lane, f32x4 (&b)[T]) {
    const int s0 = y >> 16, s1 = min(s0 + 1, 3);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float* f = (const float*)(P.in + (size_t)(row[t] + (uint32_t)sb) * 64) + lane;
        b[t] = f32x4{f[s0 * 64], f[s1 * 64], 0.f, 0.f};
    }
    if constexpr (!TWO) {
        if ((y & 255) > 2) {
            const int s2 = min(s0 + 2, 3), s3 = min(s0 + 3, 3);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float* f = (const float*)(P.in + (size_t)(row[t] + (uint32_t)sb) * 64) + lane;
                b[t][2] = f[s2 * 64];
                b[t][3] = f[s3 * 64];
            }
        }
    }
}

// One input K-block for T batch tiles (`row`: first block of each tile's row in the input activation; sb, y: the K-block table
// entry, wave-uniform), for kernels that learn at run time whether a block is slot-major (the small-batch and generic forms)
template <int T>
__device__ __forceinline__ void load_kblock(const StageParams& P, const uint32_t (&row)[T], int sb, int y, int lane, f32x4 (&b)[T]) {
    if (sb >= P.pack_in) {
        load_kblock_soa<T, false>(P, row, sb, y, lane, b);
        return;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) b[t] = P.in[(size_t)(row[t] + (uint32_t)sb) * 64 + lane];
}

// Where a node's remainder rows (register 0 of its last output tile) go: slot `slot` of the packed blocks (StageParams::pack_base)
__device__ __forceinline__ float* packed_slot_ptr(const StageParams& P, int tile, int slot, int lane) {
    float* blk = (float*)(P.out + ((size_t)tile * P.nb_out + P.pack_base + (slot >> 2)) * 64);
    return P.pack_soa ? blk + (slot & 3) * 64 + lane : blk + lane * 4 + (slot & 3);
}

// Second half of a node: expansion of the z accumulators in registers, second affine, store.
// wA2 / b2 point at this node's A2 fragments (+lane) and bias-2 fragment; address space (LDS or
// global) is resolved after inlining.
// FS: the expansion is (identity, |x|^p) and known at compile time (as in the front kernel): no function loop, no kind branches.
template <int MT1, int MT2, int T, bool REM = false, bool FS = false, typename WP, typename BP>
__device__ __forceinline__ void node_tail(const StageParams& P, WP wA2, BP b2, int node, f32x4 (&z)[MT1][T],
                                          const int (&tile)[T], int lane, float* rscr = nullptr) {      // rscr: this wave's 256 floats for rem4_total_lds (REM)
    const int g = lane >> 4;
    const int out_blk = node * P.mto;
#ifdef HIGSFA_DIAG
    const bool no_store = (P.whatif & 2) != 0;
#else
    constexpr bool no_store = false;
#endif
    if (!P.has_exp) {
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + out_blk + mt) * 64 + lane] = z[mt][t];
        return;
    }
    f32x4 y[MT2][T];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
        f32x4 bb = *(const f32x4*)(b2 + mt * 16 + g * 4);
#pragma unroll
        for (int t = 0; t < T; ++t) y[mt][t] = bb;
    }
    const float ex0 = P.expo[0], ex1 = P.expo[1], ex2 = P.expo[2], ex3 = P.expo[3];
    const uint32_t funcp = P.funcp;
    const int nf = P.nf;
    f32x4 d4[T];      // REM: 4x4-form accumulators of the last output tile
#pragma unroll
    for (int t = 0; t < T; ++t) d4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (Round 4 also measured the second affine with its A fragments alternating between two register sets, like the K-block loop
    // of k_stage: no gain at one tile per wave (layer 7: 17.3 against 17.4 us), and at two tiles per wave the second set spills —
    // 96 bytes of scratch at 128 VGPRs, layers 3-6 +17 %.  Not kept; the |x|^p block's fragments are requested before the power.)
    if constexpr (FS) {
#pragma unroll
        for (int mt1 = 0; mt1 < MT1; ++mt1) {
            const uint32_t nkp = P.nk2p[mt1];
            const int nk0 = nkp & 15, nk1 = (nkp >> 4) & 15;
            if constexpr (REM) {
                gemm_block_rem<MT2, T>(wA2 + (mt1 * 2) * MT2 * 64, z[mt1], y, d4, nk0);
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t) e[t] = pow_abs4(z[mt1][t], ex1);
                gemm_block_rem<MT2, T>(wA2 + (mt1 * 2 + 1) * MT2 * 64, e, y, d4, nk1);
            } else if constexpr (!HG_POW_PREFETCH) {
                gemm_block<MT2, T, true>(wA2 + (mt1 * 2) * MT2 * 64, z[mt1], y, nk0);
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t) e[t] = pow_abs4(z[mt1][t], ex1);
                gemm_block<MT2, T, true>(wA2 + (mt1 * 2 + 1) * MT2 * 64, e, y, nk1);
            } else {
                gemm_block<MT2, T, true>(wA2 + (mt1 * 2) * MT2 * 64, z[mt1], y, nk0);
                // the |x|^p block's A fragments are requested from LDS BEFORE the power is evaluated: the transcendentals cover the read
                f32x4 ap[MT2];
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) ap[mt] = (wA2 + (mt1 * 2 + 1) * MT2 * 64)[mt * 64];
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t) e[t] = pow_abs4(z[mt1][t], ex1);
                gemm_block_regs<MT2, T>(ap, e, y, nk1);
            }
        }
    } else
#pragma unroll
    for (int mt1 = 0; mt1 < MT1; ++mt1) {
        const uint32_t nkp = P.nk2p[mt1];
        for (int fi = 0; fi < nf; ++fi) {
            const int nk = (nkp >> (4 * fi)) & 15;
            if (nk == 0) continue;
            f32x4 e[T];
            const int fk = (funcp >> (4 * fi)) & 15;
            const float ex = fi == 0 ? ex0 : (fi == 1 ? ex1 : (fi == 2 ? ex2 : ex3));
#pragma unroll
            for (int t = 0; t < T; ++t) e[t] = apply_func(fk, ex, z[mt1][t]);
            if constexpr (REM) gemm_block_rem<MT2, T>(wA2 + (mt1 * nf + fi) * MT2 * 64, e, y, d4, nk);
            else gemm_block<MT2, T>(wA2 + (mt1 * nf + fi) * MT2 * 64, e, y, nk);
        }
    }
    if constexpr (REM) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (rscr) y[MT2 - 1][t][0] += rem4_total_lds(d4[t], rscr, lane);
            else add_rem4(y[MT2 - 1][t], d4[t]);
        }
        if (P.pack_base > 0) {      // full tiles as blocks; the remainder rows into this node's register of the shared block
#pragma unroll
            for (int mt = 0; mt < MT2 - 1; ++mt)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (tile[t] < P.n_tiles && !no_store) P.out[((size_t)tile[t] * P.nb_out + node * (MT2 - 1) + mt) * 64 + lane] = y[mt][t];
            const int slot = __builtin_amdgcn_readfirstlane(P.pack_slot[node]);
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles && !no_store)
                    *packed_slot_ptr(P, tile[t], slot, lane) = y[MT2 - 1][t][0];
            return;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (tile[t] < P.n_tiles && !no_store) P.out[((size_t)tile[t] * P.nb_out + out_blk + mt) * 64 + lane] = y[mt][t];
}


// k_tail (hg_fused_tail.hip): the LAST one to three layers of the hierarchy (1 / 2+1 / 4+2+1 nodes in the preset nets) as one
// launch in which a workgroup takes T batch tiles through all of them, activations in LDS, and writes the caller's
// row-major y (first y_cols columns) straight from the accumulators — no k_unpack pass, and only the output tiles that
// hold a requested column are computed in the last node's second affine.
struct TailStage {
    const f32x4* afrag;
    const float* bias;
    const int2* kb1tab;
    int32_t n_nodes, kb1, nf, has_exp, node_blocks, bias_floats, nb_out, mto, mt1, mt2;
    uint32_t nk2p[kMaxMT], funcp;
    float expo[kMaxFuncs];
};
constexpr int kMaxTail = 3;
struct TailParams {
    TailStage st[kMaxTail];
    const f32x4* in;          // input of the first fused layer, fragment order (global memory)
    void* y;                  // caller's row-major output
    const int32_t* col_of;    // [output block of the last layer][16 features] -> caller column, -1 = none
    int64_t ldy, n_rows;
    int32_t y_cols, y_f64;
    int32_t n_stages, n_tiles, nb_in;
    int32_t act_blocks, e_blocks;      // LDS: two activation buffers of act_blocks x T KiB, one expansion buffer of e_blocks x T KiB
    // k_subtree only: n_sub independent sub-trees of the fused layers, one per workgroup (st[k].n_nodes = nodes of ONE sub-tree in
    // layer k, st[k > 0].kb1tab = source blocks numbered within the sub-tree's LDS buffer); the last fused layer's tiles go to out_frag in fragment order (nb_out_frag blocks per batch tile)
    f32x4* out_frag;
    const int32_t* sub_nodes[kMaxTail];      // per fused layer: [sub-tree][position] -> node of the layer
    int32_t n_sub, nb_out_frag;
    unsigned long long* stamps;              // diagnostic build only (HIGSFA_STAMP): 16 wall-clock stamps per wave
};
size_t tail_lds_bytes(const TailParams& P, int T);
int tail_waves(const TailParams& P);
void launch_tail(const TailParams& P, int T, hipStream_t st);
void launch_subtree(const TailParams& P, hipStream_t st);


typedef void (*StageFn)(StageParams);
typedef void (*StageFn2)(StageParams, StageParams);

// kernels living in the other translation units
StageFn pick_stage0(int mt1, int mt2, int T, int x_dtype);            // hg_fused_front.hip
StageFn pick_stage0p(int x_dtype);
StageFn2 pick_stage01p(int x_dtype, bool stamp, bool rem4, bool fspec);
int stage01p_tiles(bool rem4, bool fspec);
StageFn2 pick_stage01d(int x_dtype, bool stamp, bool wgq);       // every wave on its own, no LDS tile (hg_fused_front.hip)
StageFn pick_igsfa(int ms, int mo, int T, int kb1);                   // hg_fused_igsfa.hip
StageFn pick_igfold(int mo, int T, bool fs = false);
StageFn pick_prod(int mt1, int mt2, int T);                           // hg_fused_prod.hip
void launch_igfold_split(const StageParams& P, int mo, int n_tiles, hipStream_t st);                                                          // hg_fused_igsfa.hip
void launch_im2frag(const void* x, int x_dtype, int64_t ldx, int64_t n_rows, int n_tiles, int nb, const int32_t* gcol, f32x4* out,
                    int vec4, hipStream_t st);

// 4 consecutive input elements -> 4 floats (16-byte / 4-byte / 32-byte loads)
template <typename XT> struct Vec4Load;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kBufferFlags = 0x00020000;      // raw buffer, 32-bit data format
template <> struct Vec4Load<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *(const f32x4*)p; }
    // same through a buffer resource: scalar base, 32-bit byte offset per lane, out-of-range reads return 0
    static __device__ __forceinline__ f32x4 buf(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    }
};
template <> struct Vec4Load<uint8_t> {
    static __device__ __forceinline__ f32x4 ld(const uint8_t* p) {
        uint32_t w = *(const uint32_t*)p;
        f32x4 v;
        v[0] = (float)(w & 0xff);
        v[1] = (float)((w >> 8) & 0xff);
        v[2] = (float)((w >> 16) & 0xff);
        v[3] = (float)(w >> 24);
        return v;
    }
    static __device__ __forceinline__ f32x4 buf(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
        return f32x4{(float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff), (float)(w >> 24)};
    }
};
template <> struct Vec4Load<double> {
    static __device__ __forceinline__ f32x4 ld(const double* p) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        f64x2 a = *(const f64x2*)p, b = *(const f64x2*)(p + 2);
        f32x4 v;
        v[0] = (float)a[0];
        v[1] = (float)a[1];
        v[2] = (float)b[0];
        v[3] = (float)b[1];
        return v;
    }
    static __device__ __forceinline__ f32x4 buf(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 a = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        const f64x2 b = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0));
        return f32x4{(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
    }
};


}  // namespace fused
}  // namespace hg
