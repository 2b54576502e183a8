// Fused plan, first-layer kernels: the caller's row-major sub-image matrix -> fragment-order activations
// (layer 0 alone, its persistent variant, and layers 0+1 in one kernel).  See hg_fused.hip for the data
// layout and the planner.
#include "hg_fused_dev.hpp"

namespace hg {
namespace fused {

// Stage 0: input = caller's row-major sub-image matrix.  The WG stages, for T batch tiles, the
// column runs its node chunk needs (full 16 B/lane coalesced row segments when alignment allows)
// into an LDS tile [sub-image][column]; each wave then takes every 4th node of the chunk and
// reads its receptive field out of LDS (one ds_read_b128 when the four k-steps of a lane are
// contiguous, e.g. 4-pixel-wide fields), subtracting the node's input mean on the way.
template <int MT1, int MT2, int T, typename XT>
__global__ void __launch_bounds__(512) k_stage0(StageParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    float* lds = (float*)smem;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6, g = lane >> 4, j = lane & 15;
    const int ci = blockIdx.x % P.n_chunks, grp = blockIdx.x / P.n_chunks;
    const DChunk ck = P.chunks[ci];
    int tile[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tile[t] = grp * T + t;
    const XT* x = (const XT*)P.x;
    const int stride = P.lds_stride;
    if (P.vec4) {
        const int pps = ck.n_pieces;             // 16-byte pieces per sub-image
        const int total = T * 16 * pps;
        constexpr int NB = 8;                    // loads in flight per thread
        for (int base = 0; base < total; base += nthr * NB) {
            f32x4 v[NB];
            int dsto[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int idx = base + k * nthr + tid;
                dsto[k] = -1;
                if (idx < total) {
                    const int sj = idx / pps, pc = idx - sj * pps;
                    const int2 pcol = P.piece_col[ck.piece_begin + pc];   // {source column, LDS word offset}
                    const int tl = tile[0] + sj / 16;
                    const int64_t row = (int64_t)tl * 16 + (sj & 15);
                    dsto[k] = sj * stride + pcol.y;
                    if (tl < P.n_tiles && row < P.n_rows)
                        v[k] = Vec4Load<XT>::ld(x + row * P.ldx + pcol.x);
                    else
                        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (dsto[k] >= 0) *(f32x4*)(lds + dsto[k]) = v[k];
        }
        if (tid < T * 16) lds[tid * stride + stride - 1] = 0.f;
    } else {
        for (int t = 0; t < T; ++t)
            for (int jj = wave; jj < 16; jj += nw) {
                const int64_t row = (int64_t)tile[t] * 16 + jj;
                float* dst = lds + (t * 16 + jj) * stride;
                const bool ok = tile[t] < P.n_tiles && row < P.n_rows;
                const XT* src = x + (ok ? row : 0) * P.ldx;
                for (int ri = 0; ri < ck.run_count; ++ri) {
                    const DRun rn = P.runs[ck.run_begin + ri];
                    for (int e = lane; e < rn.len; e += 64) dst[rn.lds_off + e] = ok ? (float)src[rn.start + e] : 0.f;
                }
                if (lane == 0) dst[stride - 1] = 0.f;  // the "zero column" padded k positions point at
            }
    }
    __syncthreads();
    for (int ni = ck.node_begin + wave; ni < ck.node_begin + ck.node_count; ni += nw) {
        const f32x4* wA1 = P.afrag + (size_t)ni * P.node_blocks * 64 + lane;
        const f32x4* wA2 = wA1 + P.kb1 * MT1 * 64;
        const float* b1 = P.bias + (size_t)ni * P.bias_floats;
        f32x4 z[MT1][T];
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
            f32x4 bb = *(const f32x4*)(b1 + mt * 16 + g * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) z[mt][t] = bb;
        }
        for (int kbi = 0; kbi < P.kb1; ++kbi) {
            const size_t ent = ((size_t)ni * P.kb1 + kbi) * 16 + g * 4;
            const i32x4 off = *(const i32x4*)(P.koff + ent);
            const f32x4 mu = *(const f32x4*)(P.kmean + ent);
            f32x4 bf[T];
            if (P.contig4) {
#pragma unroll
                for (int t = 0; t < T; ++t) bf[t] = *(const f32x4*)(lds + (t * 16 + j) * stride + off[0]) - mu;
            } else {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float* base = lds + (t * 16 + j) * stride;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bf[t][r] = base[off[r]] - mu[r];
                }
            }
            gemm_block<MT1, T>(wA1 + kbi * MT1 * 64, bf, z, kbi == P.kb1 - 1 ? P.nk_last : 4);
        }
        node_tail<MT1, MT2, T>(P, wA2, b1 + MT1 * 16, ni, z, tile, lane);
    }
}

// Stage 0, persistent + software-pipelined variant for small first-layer nodes (one K-block, one
// tile in and out, <= 2 expansion functions, 16-byte contiguous receptive-field rows): the
// workgroup owns one node chunk, keeps the weights of its nodes in REGISTERS (2 node slots per
// wave), and sweeps tile groups part, part + tile_parts, ...  While tile group i is multiplied out
// of the LDS tile, the row segments of tile group i+1 are already in flight from HBM into registers
// (8 x 16 B per thread); they are written to LDS after the barrier that ends the compute phase.
// No load is issued inside the compute phase, so the in-order vmcnt queue never forces the
// prefetch to land early.
template <int T, typename XT>
__global__ void __launch_bounds__(512) k_stage0p(StageParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    float* lds = (float*)smem;
    constexpr int NB = 8, NPW = 2;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6, g = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = blockIdx.x % P.n_chunks, part = blockIdx.x / P.n_chunks;
    const DChunk ck = P.chunks[ci];
    const XT* x = (const XT*)P.x;
    const int stride = P.lds_stride;
    const int n_groups = (P.n_tiles + T - 1) / T;
    // --- per-thread staging bookkeeping (the same pieces for every tile group)
    const int pps = ck.n_pieces, total = T * 16 * pps;
    int p_col[NB], p_dst[NB];   // source column; (sub-image << 24 | LDS word offset) or -1
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int idx = k * nthr + tid;
        p_col[k] = 0;
        p_dst[k] = -1;
        if (idx < total) {
            const int sj = idx / pps, pc = idx - sj * pps;
            const int2 pcol = P.piece_col[ck.piece_begin + pc];
            p_col[k] = pcol.x;
            p_dst[k] = (sj << 24) | (sj * stride + pcol.y);
        }
    }
    // --- weights of this wave's node slots, resident in registers for the whole sweep
    int w_off[NPW];
    f32x4 w_mu[NPW], w_a1[NPW], w_a2[NPW][2], w_b1[NPW], w_b2[NPW];
    bool w_ok[NPW];
#pragma unroll
    for (int sl = 0; sl < NPW; ++sl) {
        const int nl = wave + sl * nw;
        w_ok[sl] = nl < ck.node_count;
        const int ni = ck.node_begin + (w_ok[sl] ? nl : 0);
        const size_t ent = (size_t)ni * 16 + g * 4;       // kb1 == 1
        w_off[sl] = P.koff[ent];
        w_mu[sl] = *(const f32x4*)(P.kmean + ent);
        const f32x4* wp = P.afrag + (size_t)ni * P.node_blocks * 64 + lane;
        w_a1[sl] = wp[0];
        w_a2[sl][0] = wp[64];
        w_a2[sl][1] = P.nf > 1 ? wp[128] : wp[64];
        const float* bp = P.bias + (size_t)ni * P.bias_floats + g * 4;
        w_b1[sl] = *(const f32x4*)bp;
        w_b2[sl] = *(const f32x4*)(bp + 16);
    }
    const int nk1 = P.nk_last;
    const int nk2a = P.nk2p[0] & 15, nk2b = (P.nk2p[0] >> 4) & 15;
    const int fk0 = P.funcp & 15, fk1 = (P.funcp >> 4) & 15;
    const float ex0 = P.expo[0], ex1 = P.expo[1];

    auto fetch = [&](int grp, f32x4 (&v)[NB]) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t row = (int64_t)grp * (T * 16) + (p_dst[k] >> 24);
            if (p_dst[k] >= 0 && row < P.n_rows)
                v[k] = Vec4Load<XT>::ld(x + row * P.ldx + p_col[k]);
            else
                v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 v[NB];
    if (part < n_groups) fetch(part, v);
    for (int grp = part; grp < n_groups; grp += P.tile_parts) {
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (p_dst[k] >= 0) *(f32x4*)(lds + (p_dst[k] & 0xffffff)) = v[k];
        if (tid < T * 16) lds[tid * stride + stride - 1] = 0.f;
        __syncthreads();
        if (grp + P.tile_parts < n_groups) fetch(grp + P.tile_parts, v);
        int tile[T];
#pragma unroll
        for (int t = 0; t < T; ++t) tile[t] = grp * T + t;
#pragma unroll
        for (int sl = 0; sl < NPW; ++sl) {
            if (!w_ok[sl]) continue;
            const int ni = ck.node_begin + wave + sl * nw;
            f32x4 z[1][T], y[1][T], bf[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bf[t] = *(const f32x4*)(lds + (t * 16 + j) * stride + w_off[sl]) - w_mu[sl];
                z[0][t] = w_b1[sl];
                y[0][t] = w_b2[sl];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nk1) {
#pragma unroll
                    for (int t = 0; t < T; ++t) z[0][t] = MFMA16(w_a1[sl][r], bf[t][r], z[0][t]);
                }
            {
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t) e[t] = apply_func(fk0, ex0, z[0][t]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nk2a) {
#pragma unroll
                        for (int t = 0; t < T; ++t) y[0][t] = MFMA16(w_a2[sl][0][r], e[t][r], y[0][t]);
                    }
            }
            if (P.nf > 1) {
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t) e[t] = apply_func(fk1, ex1, z[0][t]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nk2b) {
#pragma unroll
                        for (int t = 0; t < T; ++t) y[0][t] = MFMA16(w_a2[sl][1][r], e[t][r], y[0][t]);
                    }
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + ni) * 64 + lane] = y[0][t];
        }
        __syncthreads();
    }
}

// Stages 0 AND 1 in one persistent kernel (the U11L front end): same structure as k_stage0p, but a
// wave's two node slots are ADJACENT layer-0 nodes 2w, 2w+1 — the two children of layer-1 node w of
// the chunk — so their output accumulators are, in registers, the two K-blocks of that layer-1
// node's first affine.  Layer-1 weights (4 + 8 fragment blocks) are register resident too.  The
// layer-0 activation (64 KiB per sub-image written and read back) never exists in memory.
// Requirements checked on the host: layer 0 as for k_stage0p; layer-1 node n reads exactly the
// blocks of layer-0 nodes 2n and 2n+1; both layer-1 affines have <= 32 outputs; same <= 2 functions.
// REM4: both layer-1 affines have 17..20 outputs, i.e. their second 16-row tile holds only four real rows:
// those tiles run in the 4x4 MFMA form (hg_fused_dev.hpp, "Remainder tiles"; 22 % of this kernel's MFMA time).
// FSPEC = 1: both layers expand with exactly (identity, |x|^p) — the expansion of every preset network — known at compile
// time: no function-kind branches, the second-tile k-step count of a REM4 layer 1 is the constant 1, and the loop body is
// one basic block in which the scheduler can place a wave's expansion arithmetic in the shadow of its own MFMAs.
template <typename XT, bool STAMP = false, bool REM4 = false, int TT = 2, int FSPEC = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(TT == 2 ? 3 : TT == 1 ? 4 : 2, TT == 2 ? 3 : TT == 1 ? 4 : 2))) k_stage01p(StageParams P, StageParams Q) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    float* lds = (float*)smem;
    unsigned long long rt_entry = 0;
    if (STAMP) rt_entry = __builtin_amdgcn_s_memrealtime();
    constexpr int T = TT, NB = 2 * TT, NPW = 2;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = blockIdx.x % P.n_chunks, part = blockIdx.x / P.n_chunks;
    const DChunk ck = P.chunks[ci];
    const XT* x = (const XT*)P.x;
    const int stride = P.lds_stride;
    const int n_groups = (P.n_tiles + T - 1) / T;
    const int pps = ck.n_pieces, total = T * 16 * pps;
    const XT* p_src[NB];   // address of the piece in tile group 0; a group advances every piece by T*16 rows
    int p_dst[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int idx = k * nthr + tid;
        p_src[k] = x;
        p_dst[k] = -1;
        if (idx < total) {
            const int sj = idx / pps, pc = idx - sj * pps;
            const int2 pcol = P.piece_col[ck.piece_begin + pc];
            p_src[k] = x + (int64_t)sj * P.ldx + pcol.x;
            p_dst[k] = (sj << 24) | (sj * stride + pcol.y);
        }
    }
    const int64_t grp_step = (int64_t)(T * 16) * P.ldx;
    // layer-0 weights of the two slots
    // Means and biases (10 vectors of 16 floats per wave; a lane needs the four of its group g) live in
    // LDS behind the two tiles and are re-read every tile group: 40 VGPRs less, which is what lets a
    // third wave per SIMD fit (168 VGPRs) — the weights proper stay in registers.
    float* cst = lds + (kDoubleBuffer01 ? 2 : 1) * (T * 16 * stride) + wave * 160;
    enum { C_MU = 0, C_B1 = 2, C_B2 = 4, C_QB1 = 6, C_QB2 = 8 };
    int w_off[NPW];
    f32x4 w_a1[NPW], w_a2[NPW][2];
    const bool w_ok = 2 * wave + 1 < ck.node_count;    // both children present (chunks hold whole pairs)
#pragma unroll
    for (int sl = 0; sl < NPW; ++sl) {
        const int ni = ck.node_begin + (w_ok ? 2 * wave + sl : 0);
        const size_t ent = (size_t)ni * 16 + g * 4;
        w_off[sl] = P.koff[ent];
        if (j == 0) *(f32x4*)(cst + (C_MU + sl) * 16 + g * 4) = *(const f32x4*)(P.kmean + ent);
        const f32x4* wp = P.afrag + (size_t)ni * P.node_blocks * 64 + lane;
        w_a1[sl] = wp[0];
        w_a2[sl][0] = wp[64];
        w_a2[sl][1] = P.nf > 1 ? wp[128] : wp[64];
        const float* bp = P.bias + (size_t)ni * P.bias_floats + g * 4;
        if (j == 0) {
            *(f32x4*)(cst + (C_B1 + sl) * 16 + g * 4) = *(const f32x4*)bp;
            *(f32x4*)(cst + (C_B2 + sl) * 16 + g * 4) = *(const f32x4*)(bp + 16);
        }
    }
    // layer-1 node of this wave: A1 [kb 0..1][mt 0..1], A2 [mt1 0..1][fi 0..1][mt2 0..1], biases
    const int n1 = (ck.node_begin >> 1) + (w_ok ? wave : 0);
    f32x4 q_a1[2][2], q_a2[2][2][2];
    {
        const f32x4* wq = Q.afrag + (size_t)n1 * Q.node_blocks * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) q_a1[kb][mt] = wq[(kb * 2 + mt) * 64];
        const f32x4* wq2 = wq + Q.kb1 * 2 * 64;
#pragma unroll
        for (int m1 = 0; m1 < 2; ++m1)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) q_a2[m1][fi][mt] = wq2[((m1 * Q.nf + (fi < Q.nf ? fi : 0)) * 2 + mt) * 64];
        const float* bq = Q.bias + (size_t)n1 * Q.bias_floats + g * 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            if (j == 0) {
                *(f32x4*)(cst + (C_QB1 + mt) * 16 + g * 4) = *(const f32x4*)(bq + mt * 16);
                *(f32x4*)(cst + (C_QB2 + mt) * 16 + g * 4) = *(const f32x4*)(bq + 32 + mt * 16);
            }
    }
    if (REM4 && !Q.a4x4) {
        // 4x4 form of the second-tile fragments: lane (g, i = lane % 4) takes row 4 i of lane group g
        // (Q.a4x4: the planner stored them in that form already, as for the k_stage REM instantiations)
        const int src = ((lane & 48) | ((lane & 3) << 2)) << 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                q_a1[kb][1][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(q_a1[kb][1][r])));
#pragma unroll
            for (int m1 = 0; m1 < 2; ++m1)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
                    q_a2[m1][fi][1][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(q_a2[m1][fi][1][r])));
        }
    }
    const int fk0 = P.funcp & 15, fk1 = (P.funcp >> 4) & 15;
    const float ex0 = P.expo[0], ex1 = P.expo[1];
    const int qfk0 = Q.funcp & 15, qfk1 = (Q.funcp >> 4) & 15;
    const float qex0 = Q.expo[0], qex1 = Q.expo[1];

    auto fetch = [&](int grp, f32x4 (&v)[NB]) {
        const int64_t goff = (int64_t)grp * grp_step;            // wave-uniform
        const int64_t rows_left = P.n_rows - (int64_t)grp * (T * 16);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (p_dst[k] >= 0 && (p_dst[k] >> 24) < rows_left)
                v[k] = Vec4Load<XT>::ld(p_src[k] + goff);
            else
                v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 v[NB];
    // tile-group queue (StageParams::work_ctr): this workgroup's first group is `part`, the rest are grabbed by thread 0
    // two iterations ahead and passed through an LDS slot that alternates with the tile buffers
    int* qslot = (int*)(lds + (kDoubleBuffer01 ? 2 : 1) * (T * 16 * stride) + (nthr >> 6) * 160);
    uint32_t* qctr = P.work_ctr + (size_t)ci * 16;
    const uint32_t q_dyn = (uint32_t)(n_groups - P.tile_parts);
    // increment-with-wrap rather than fetch_add: the compiler's atomic optimizer rewrites the latter into a wave-wide reduction +
    // readfirstlane, which waits for the result on the spot.  The raw return value is kept as it is until it is needed (one
    // iteration later), so that no wait is placed next to the atomic.
    auto grab_raw = [&]() -> uint32_t { return __builtin_amdgcn_atomic_inc32(qctr, 0xffffffffu, __ATOMIC_RELAXED, "agent"); };
    auto grab_group = [&](uint32_t raw) -> int {
        const uint32_t k = raw - P.work_base;
        return k < q_dyn ? (int)(P.tile_parts + k) : n_groups;
    };
    uint32_t q_raw = 0;                        // thread 0: the grab in flight (group of the iteration after next)
    bool q_more = tid == 0;                    // thread 0: no grab has failed yet
    if (q_more) q_raw = grab_raw();
    if (part < n_groups) fetch(part, v);
    // two LDS tiles, used alternately: one barrier per tile group is enough (a wave that writes tile
    // i+1 has passed barrier i, i.e. every wave has finished reading tile i-1, which shares its buffer)
    const int buf_words = T * 16 * stride;
    float* lds0 = lds;
    int flip = 0;
    unsigned long long t_w = 0, t_f = 0, t_l0 = 0, t_l1 = 0, t_all0 = 0, rt0 = 0, ts = 0;
    int n_it = 0;
    if (STAMP) {
        t_all0 = stamp_now();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    // Deferred stores: gfx950 returns loads and stores out of order with respect to each other, so waiting for the prefetched
    // sub-image rows (vmcnt(0)) would also wait for whatever stores were issued since — a full write round trip per tile group
    // when the results are stored at the end of an iteration.  They are therefore kept in registers and stored one iteration
    // later, right after the barrier, together with the next prefetch: every wait then sees only stores that are a whole
    // compute phase old.
    constexpr bool DEFER = REM4 && FSPEC == 1;      // the generic instantiation has no registers to spare: it stores at once
    f32x4 st_y0[T];
    float st_y1[T];
    int st_grp = -1;
    const int pk_slot = (REM4 && Q.pack_base > 0) ? __builtin_amdgcn_readfirstlane(Q.pack_slot[n1]) : 0;
    auto flush = [&]() {
        if (st_grp < 0) return;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tl = st_grp * T + t;
            if (tl < P.n_tiles) {
                Q.out[((size_t)tl * Q.nb_out + n1) * 64 + lane] = st_y0[t];
                *packed_slot_ptr(Q, tl, pk_slot, lane) = st_y1[t];
            }
        }
        st_grp = -1;
    };
    for (int grp = part; grp < n_groups;) {
        if (STAMP) ts = stamp_now();
        const int slot = flip;
        if (kDoubleBuffer01) {
            lds = lds0 + flip * buf_words;
        } else {
            __syncthreads();   // single tile: every wave is done with the previous tile group
        }
        flip ^= 1;
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (p_dst[k] >= 0) *(f32x4*)(lds + (p_dst[k] & 0xffffff)) = v[k];
        if (tid < T * 16) lds[tid * stride + stride - 1] = 0.f;
        if (tid == 0) {
            const int q_next = q_more ? grab_group(q_raw) : n_groups;
            q_more = q_next < n_groups;
            qslot[slot] = q_next;
        }
        __syncthreads();
        const int grp_next = __builtin_amdgcn_readfirstlane(qslot[slot]);
        if (STAMP) { unsigned long long t = stamp_now(); t_w += t - ts; ts = t; }
        if (grp_next < n_groups) fetch(grp_next, v);
        if (DEFER) flush();      // after the loads: the compiler waits for everything outstanding before it reuses their registers
        if (q_more) q_raw = grab_raw();      // likewise the next grab: issued here, looked at one iteration later
        if (STAMP) { unsigned long long t = stamp_now(); t_f += t - ts; ts = t; }
        if (w_ok) {
            int tile[T];
#pragma unroll
            for (int t = 0; t < T; ++t) tile[t] = grp * T + t;
            // ---- layer 0: two children
            f32x4 y0[NPW][T];
#pragma unroll
            for (int sl = 0; sl < NPW; ++sl) {
                f32x4 z[T], bf[T];
                const f32x4 mu = *(const f32x4*)(cst + (C_MU + sl) * 16 + g * 4);
                const f32x4 cb1 = *(const f32x4*)(cst + (C_B1 + sl) * 16 + g * 4), cb2 = *(const f32x4*)(cst + (C_B2 + sl) * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    bf[t] = *(const f32x4*)(lds + (t * 16 + j) * stride + w_off[sl]) - mu;
                    z[t] = cb1;
                    y0[sl][t] = cb2;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < T; ++t) z[t] = MFMA16(w_a1[sl][r], bf[t][r], z[t]);
                {
                    f32x4 e[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) e[t] = FSPEC == 1 ? z[t] : apply_func(fk0, ex0, z[t]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < T; ++t) y0[sl][t] = MFMA16(w_a2[sl][0][r], e[t][r], y0[sl][t]);
                }
                {
                    f32x4 e[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) e[t] = FSPEC == 1 ? pow_abs4(z[t], ex1) : apply_func(fk1, ex1, z[t]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < T; ++t) y0[sl][t] = MFMA16(w_a2[sl][1][r], e[t][r], y0[sl][t]);
                }
            }
            if (STAMP) { unsigned long long t = stamp_now(); t_l0 += t - ts; ts = t; }
            // ---- layer 1: K-blocks of the first affine are the children's accumulators
            f32x4 z1[2][T], y1[2][T];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    z1[mt][t] = *(const f32x4*)(cst + (C_QB1 + mt) * 16 + g * 4);
                    y1[mt][t] = *(const f32x4*)(cst + (C_QB2 + mt) * 16 + g * 4);
                }
            f32x4 d4[T];      // REM4: 4x4-form accumulators of the second tile (rows x k partial sums)
#pragma unroll
            for (int t = 0; t < T; ++t) d4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int t = 0; t < T; ++t) z1[0][t] = MFMA16(q_a1[kb][0][r], y0[kb][t][r], z1[0][t]);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if constexpr (REM4) d4[t] = MFMA4(q_a1[kb][1][r], y0[kb][t][r], d4[t]);
                        else z1[1][t] = MFMA16(q_a1[kb][1][r], y0[kb][t][r], z1[1][t]);
                    }
                }
            if constexpr (REM4) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    add_rem4(z1[1][t], d4[t]);
                    d4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) {        // z tile 0: full, branch-free
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t)
                    e[t] = FSPEC == 1 ? (fi == 0 ? z1[0][t] : pow_abs4(z1[0][t], qex1)) : apply_func(fi == 0 ? qfk0 : qfk1, fi == 0 ? qex0 : qex1, z1[0][t]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int t = 0; t < T; ++t) y1[0][t] = MFMA16(q_a2[0][fi][0][r], e[t][r], y1[0][t]);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if constexpr (REM4) d4[t] = MFMA4(q_a2[0][fi][1][r], e[t][r], d4[t]);
                        else y1[1][t] = MFMA16(q_a2[0][fi][1][r], e[t][r], y1[1][t]);
                    }
                }
            }
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) {        // z tile 1: partial (runtime k-step count)
                const int nk = (FSPEC == 1 && REM4) ? 1 : (int)((Q.nk2p[1] >> (4 * fi)) & 15);
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t)
                    e[t] = FSPEC == 1 ? (fi == 0 ? z1[1][t] : pow_abs4(z1[1][t], qex1)) : apply_func(fi == 0 ? qfk0 : qfk1, fi == 0 ? qex0 : qex1, z1[1][t]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nk) {
#pragma unroll
                        for (int t = 0; t < T; ++t) y1[0][t] = MFMA16(q_a2[1][fi][0][r], e[t][r], y1[0][t]);
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            if constexpr (REM4) d4[t] = MFMA4(q_a2[1][fi][1][r], e[t][r], d4[t]);
                            else y1[1][t] = MFMA16(q_a2[1][fi][1][r], e[t][r], y1[1][t]);
                        }
                    }
            }
            if constexpr (REM4) {
#pragma unroll
                for (int t = 0; t < T; ++t) add_rem4(y1[1][t], d4[t]);
            }
            if (REM4 && Q.pack_base > 0) {     // packed remainder tiles (hg_fused_dev.hpp, StageParams::pack_base): stored one iteration later
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    st_y0[t] = y1[0][t];
                    st_y1[t] = y1[1][t][0];
                }
                st_grp = grp;
                if (!DEFER) flush();
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        if (tile[t] < P.n_tiles) Q.out[((size_t)tile[t] * Q.nb_out + (size_t)n1 * Q.mto + mt) * 64 + lane] = y1[mt][t];
            }
            if (STAMP) { unsigned long long t = stamp_now(); t_l1 += t - ts; ts = t; ++n_it; }
        }
        grp = grp_next;
    }
    if (DEFER) flush();
    if (STAMP && lane == 0 && P.stamps) {
        unsigned long long* o = P.stamps + ((size_t)blockIdx.x * 8 + wave) * 12;
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        o[0] = t_w; o[1] = t_f; o[2] = t_l0; o[3] = t_l1;
        o[4] = stamp_now() - t_all0;
        o[5] = rt1 - rt0;
        o[6] = (unsigned long long)n_it;
        o[7] = rt_entry; o[8] = rt0; o[9] = rt1;
    }
}

// Stages 0 AND 1, every wave on its own ("direct" form of k_stage01p for the case its compile-time instantiation covers:
// contiguous 4-column pieces, both layers expanding with (identity, |x|^p), layer-1 remainder tiles in 4x4 form).
// A wave owns layer-1 node w of its chunk and the two layer-0 children; lane (g, j) loads the four contiguous input
// columns of lane group g of sub-image j straight from the caller's matrix into the MFMA B operand — 16 bytes per lane,
// the eight loads that share a 128-byte line (two children x four waves of the workgroup) are issued within one
// iteration of each other and meet in L1 / L2 — so there is no LDS tile, no cooperative index arithmetic and NO BARRIER:
// in k_stage01p every tile group ends with the four waves of a workgroup (four SIMDs, each with its own mix of
// co-resident waves) waiting for the slowest.  One batch tile per pass; tiles are handed out per wave from the queue of
// its layer-1 node (StageParams::work_ctr, counter index = layer-1 node); loads, deferred stores and the next grab are
// issued together right after the previous iteration's data has been taken over, and are looked at one iteration later.
// 100 VGPRs at four waves per SIMD (five waves: 96 VGPRs with spills, 170 us against 141).
// WGQ: ONE tile queue per chunk instead of one per layer-1 node.  Wave 0 grabs and publishes the tile of pass i + 1 in an LDS
// ring while it works on pass i; the other waves of the workgroup read it there, so that the four waves walk the same tiles
// at nearly the same time and the 128-byte input lines and packed output blocks they share are fetched and written once
// (the per-wave queues decorrelate them: 389 MB read / 235 MB written against 279 / 201).  No barrier in the loop: a
// follower waits only when it is more than one pass ahead of wave 0, wave 0 only when a follower is eight passes behind.
template <typename XT, bool STAMP = false, bool WGQ = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_stage01d(StageParams P, StageParams Q) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    unsigned long long rt_entry = 0;
    if (STAMP) rt_entry = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = blockIdx.x % P.n_chunks, part = blockIdx.x / P.n_chunks;
    const DChunk ck = P.chunks[ci];
    // LDS: 4 x 160 floats of per-wave constants | ring of 8 {pass, tile} | progress word per wave | (HG_REM4_LDS) 256 floats of scratch per wave
    unsigned long long* const ring = (unsigned long long*)((float*)smem + 4 * 160);      // entry = pass << 32 | tile
    int* const prog = (int*)(ring + 8);
#ifndef HG_REM4_LDS
#define HG_REM4_LDS 1      // (A/B switch) lane-group sums of the 4x4-form tiles through LDS instead of permlane swaps
#endif
    float* const rscr = (float*)smem + 4 * 160 + 32 + wave * 256;
    if (WGQ) {
        if (tid < 8) ring[tid] = ~0ull;
        if (tid < 4) prog[tid] = 0;
        __syncthreads();                               // the only barrier: before any wave leaves
    }
    if (2 * wave + 1 >= ck.node_count) return;         // chunks hold whole pairs; no barrier below
    const int n_active = ck.node_count >> 1;           // waves of this workgroup that work
    const XT* x = (const XT*)P.x;
    float* cst = (float*)smem + wave * 160;            // this wave's means and biases (10 vectors of 16 floats): read back every tile
    enum { C_MU = 0, C_B1 = 2, C_B2 = 4, C_QB1 = 6, C_QB2 = 8 };
    uint32_t src[2];          // byte offset of this lane's four columns inside a batch tile's rows
    f32x4 w_a1[2], w_a2[2][2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int ni = ck.node_begin + 2 * wave + sl;
        const size_t ent = (size_t)ni * 16 + g * 4;
        src[sl] = (uint32_t)(j * (int)P.ldx + P.kcol[ni * 4 + g]) * (uint32_t)sizeof(XT);
        if (j == 0) *(f32x4*)(cst + (C_MU + sl) * 16 + g * 4) = *(const f32x4*)(P.kmean + ent);
        const f32x4* wp = P.afrag + (size_t)ni * P.node_blocks * 64 + lane;
        w_a1[sl] = wp[0];
        w_a2[sl][0] = wp[64];
        w_a2[sl][1] = wp[128];
        const float* bp = P.bias + (size_t)ni * P.bias_floats + g * 4;
        if (j == 0) {
            *(f32x4*)(cst + (C_B1 + sl) * 16 + g * 4) = *(const f32x4*)bp;
            *(f32x4*)(cst + (C_B2 + sl) * 16 + g * 4) = *(const f32x4*)(bp + 16);
        }
    }
    const int n1 = (ck.node_begin >> 1) + wave;
    // layer-1 weights: first affine [kb][mt]; second affine, first z tile [fi][mt]; of the second z tile (four real rows = ONE
    // k-step) only that k-step's value per lane is kept
    f32x4 q_a1[2][2], q_a2[2][2];
    float q_a2t[2][2];
    {
        const f32x4* wq = Q.afrag + (size_t)n1 * Q.node_blocks * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) q_a1[kb][mt] = wq[(kb * 2 + mt) * 64];
        const f32x4* wq2 = wq + Q.kb1 * 2 * 64;
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                q_a2[fi][mt] = wq2[(fi * 2 + mt) * 64];
                q_a2t[fi][mt] = wq2[((2 + fi) * 2 + mt) * 64][0];
            }
        const float* bq = Q.bias + (size_t)n1 * Q.bias_floats + g * 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            if (j == 0) {
                *(f32x4*)(cst + (C_QB1 + mt) * 16 + g * 4) = *(const f32x4*)(bq + mt * 16);
                *(f32x4*)(cst + (C_QB2 + mt) * 16 + g * 4) = *(const f32x4*)(bq + 32 + mt * 16);
            }
    }
    if (!Q.a4x4) {       // 4x4 form of the second-tile fragments (see k_stage01p)
        const int bsrc = ((lane & 48) | ((lane & 3) << 2)) << 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                q_a1[kb][1][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bsrc, __float_as_int(q_a1[kb][1][r])));
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
                q_a2[fi][1][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bsrc, __float_as_int(q_a2[fi][1][r])));
        }
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) q_a2t[fi][1] = __int_as_float(__builtin_amdgcn_ds_bpermute(bsrc, __float_as_int(q_a2t[fi][1])));
    }
    const float ex1 = P.expo[1], qex1 = Q.expo[1];
    const int n_tiles = P.n_tiles;
    const int64_t tile_step = (int64_t)16 * P.ldx;

    // Loads and stores go through buffer resources rebuilt per tile (scalar base + one 32-bit offset per lane, no 64-bit
    // address arithmetic in vector registers); rows past the end of the batch are out of range and read as zero.
    const uint32_t row_bytes = (uint32_t)P.ldx * (uint32_t)sizeof(XT);
    auto fetch = [&](int tile, f32x4 (&v)[2]) {
#ifdef HIGSFA_DIAG
        if (P.whatif & 1) tile = part;          // timing experiment: every pass reads the wave's first tile again (cache-hot)
#endif
        const int64_t rows = P.n_rows - (int64_t)tile * 16;
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc((void*)(x + tile * tile_step), 0, (int)((rows < 16 ? (uint32_t)rows : 16u) * row_bytes), kBufferFlags);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) v[sl] = Vec4Load<XT>::buf(r, src[sl]);
    };
    // tile queue of this layer-1 node: first tile = part, then tile_parts + (counter - base); lane 0 grabs
    uint32_t* qctr = P.work_ctr + (size_t)(WGQ ? ci : n1) * 16;
    const bool grabber = !WGQ || wave == 0;            // wave-uniform
    const uint32_t q_dyn = (uint32_t)(n_tiles - P.tile_parts);
    auto grab_raw = [&]() -> uint32_t { return __builtin_amdgcn_atomic_inc32(qctr, 0xffffffffu, __ATOMIC_RELAXED, "agent"); };
    uint32_t q_raw = 0;
    bool q_more = true;          // wave-uniform: no grab has failed yet
    if (grabber && lane == 0) q_raw = grab_raw();
    f32x4 v[2];
    int pass = 0;
    auto decode = [&](uint32_t raw) -> int {
        const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw) - P.work_base;
        return k < q_dyn ? (int)(P.tile_parts + k) : n_tiles;
    };
    int tile = part;
    if (tile < n_tiles) fetch(tile, v);
    // deferred stores (see k_stage01p)
    // (the host only picks this kernel for a layer 1 that writes packed remainder tiles: Q.pack_base > 0)
    f32x4 st_y0;
    float st_y1 = 0.f;
    int st_tile = -1;
    const int pk_slot = __builtin_amdgcn_readfirstlane(Q.pack_slot[n1]);
    // lane-major packed block: float [lane][slot]; slot-major (StageParams::pack_soa): float [slot][lane]
    const uint32_t pk_off = (uint32_t)(Q.pack_base + (pk_slot >> 2)) * 1024u + (uint32_t)(pk_slot & 3) * (Q.pack_soa ? 256u : 4u);
    const uint32_t pk_lane = (uint32_t)lane * (Q.pack_soa ? 4u : 16u);
    const uint32_t out_tile_bytes = (uint32_t)Q.nb_out * 1024u;
    auto flush = [&]() {
        if (st_tile < 0) return;
#ifdef HIGSFA_DIAG
        if (P.whatif & 2) { st_tile = -1; return; }      // timing experiment: no stores
#endif
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(Q.out + (size_t)st_tile * Q.nb_out * 64), 0, (int)out_tile_bytes, kBufferFlags);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, st_y0), r, (uint32_t)lane * 16u, (uint32_t)n1 * 1024u, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(st_y1), r, pk_lane, pk_off, 0);
        st_tile = -1;
    };
    unsigned long long t_top = 0, t_l0 = 0, t_l1 = 0, t_all0 = 0, rt0 = 0, ts = 0;
    int n_it = 0;
    if (STAMP) {
        t_all0 = stamp_now();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    while (tile < n_tiles) {
        // keeps the compiler from hoisting the ten constant vectors out of the loop (40 VGPRs: spills at four waves per SIMD);
        // they are meant to be re-read from LDS every tile
        asm volatile("" ::: "memory");
        if (STAMP) ts = stamp_now();
        int next = n_tiles;
        if (grabber && q_more) {
            next = decode(q_raw);
            q_more = next < n_tiles;
        }
        if (WGQ) {
            constexpr int kSpinLimit = 1 << 22;        // bounded polls: a bug must not hang the GPU
            if (wave == 0) {
                // publish the tile of pass + 1 (or the end mark); the entry goes into the slot of entry pass - 7, which every
                // follower must have read.  (Publishing two passes ahead measured SLOWER: 159 us against 147.)
                if (pass >= 7) {
                    for (int w = 1; w < n_active; ++w) {
                        int spins = 0;
                        while (__builtin_amdgcn_readfirstlane(*(volatile int*)(prog + w)) < pass - 7 && ++spins < kSpinLimit) __builtin_amdgcn_s_sleep(2);
                        if (spins >= kSpinLimit && lane == 0) *P.err = 2;
                    }
                }
                if (lane == 0) *(volatile unsigned long long*)(ring + ((pass + 1) & 7)) = ((unsigned long long)(uint32_t)(pass + 1) << 32) | (uint32_t)next;
            } else {
                // follower: the tile of THIS pass was fixed when it read the entry one pass ago; now it needs entry pass + 1
                int spins = 0, e_pass, e_tile;
                do {
                    const unsigned long long e = *(volatile unsigned long long*)(ring + ((pass + 1) & 7));
                    e_pass = __builtin_amdgcn_readfirstlane((int)(e >> 32));
                    e_tile = __builtin_amdgcn_readfirstlane((int)(uint32_t)e);
                    if (e_pass == pass + 1) break;
                    __builtin_amdgcn_s_sleep(2);
                } while (++spins < kSpinLimit);
                next = e_pass == pass + 1 ? e_tile : n_tiles;
                if (e_pass != pass + 1 && lane == 0) *P.err = 3;
                if (lane == 0) *(volatile int*)(prog + wave) = pass + 1;      // entry pass + 1 has been read
            }
            ++pass;
        }
        // ---- layer 0, first affine of both children; the sub-image registers are free after it, so the next tile's loads
        // (and with them the deferred stores and the next grab) go out here
        f32x4 z0[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const f32x4 bf = v[sl] - *(const f32x4*)(cst + (C_MU + sl) * 16 + g * 4);
            z0[sl] = *(const f32x4*)(cst + (C_B1 + sl) * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) z0[sl] = MFMA16(w_a1[sl][r], bf[r], z0[sl]);
        }
        if (next < n_tiles) fetch(next, v);
        flush();
        if (grabber && q_more && lane == 0) q_raw = grab_raw();
        if (STAMP) { unsigned long long t = stamp_now(); t_top += t - ts; ts = t; }
        f32x4 y0[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            y0[sl] = *(const f32x4*)(cst + (C_B2 + sl) * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) y0[sl] = MFMA16(w_a2[sl][0][r], z0[sl][r], y0[sl]);
            const f32x4 e = pow_abs4(z0[sl], ex1);
#pragma unroll
            for (int r = 0; r < 4; ++r) y0[sl] = MFMA16(w_a2[sl][1][r], e[r], y0[sl]);
        }
        if (STAMP) { unsigned long long t = stamp_now(); t_l0 += t - ts; ts = t; }
        // ---- layer 1
        f32x4 z1[2], y1[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            z1[mt] = *(const f32x4*)(cst + (C_QB1 + mt) * 16 + g * 4);
            y1[mt] = *(const f32x4*)(cst + (C_QB2 + mt) * 16 + g * 4);
        }
        f32x4 d4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z1[0] = MFMA16(q_a1[kb][0][r], y0[kb][r], z1[0]);
                d4 = MFMA4(q_a1[kb][1][r], y0[kb][r], d4);
            }
        if constexpr (HG_REM4_LDS) z1[1][0] += rem4_total_lds(d4, rscr, lane);
        else add_rem4(z1[1], d4);
        d4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            const f32x4 e = fi == 0 ? z1[0] : pow_abs4(z1[0], qex1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y1[0] = MFMA16(q_a2[fi][0][r], e[r], y1[0]);
                d4 = MFMA4(q_a2[fi][1][r], e[r], d4);
            }
        }
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {        // second z tile: four real rows = one k-step
            const float e0 = fi == 0 ? z1[1][0] : pow_abs(z1[1][0], qex1);
            y1[0] = MFMA16(q_a2t[fi][0], e0, y1[0]);
            d4 = MFMA4(q_a2t[fi][1], e0, d4);
        }
        if constexpr (HG_REM4_LDS) y1[1][0] += rem4_total_lds(d4, rscr, lane);
        else add_rem4(y1[1], d4);
        st_y0 = y1[0];
        st_y1 = y1[1][0];
        st_tile = tile;
        if (STAMP) { unsigned long long t = stamp_now(); t_l1 += t - ts; ts = t; ++n_it; }
        tile = next;
    }
    flush();
    if (STAMP && lane == 0 && P.stamps) {
        unsigned long long* o = P.stamps + ((size_t)blockIdx.x * 8 + wave) * 12;
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        o[0] = t_top; o[1] = 0; o[2] = t_l0; o[3] = t_l1;
        o[4] = stamp_now() - t_all0;
        o[5] = rt1 - rt0;
        o[6] = (unsigned long long)n_it;
        o[7] = rt_entry; o[8] = rt0; o[9] = rt1;
    }
}

StageFn2 pick_stage01d(int x_dtype, bool stamp, bool wgq) {
#ifdef HIGSFA_DIAG
    if (stamp && x_dtype == HG_F32) return wgq ? (StageFn2)k_stage01d<float, true, true> : (StageFn2)k_stage01d<float, true, false>;
#endif
    if (!wgq) return x_dtype == HG_U8 ? (StageFn2)k_stage01d<uint8_t, false, false> : x_dtype == HG_F32 ? (StageFn2)k_stage01d<float, false, false> : (StageFn2)k_stage01d<double, false, false>;
    return x_dtype == HG_U8 ? (StageFn2)k_stage01d<uint8_t> : x_dtype == HG_F32 ? (StageFn2)k_stage01d<float> : (StageFn2)k_stage01d<double>;
}

template <int MT1, int MT2, typename XT>
static StageFn pick_stage0_t(int T) {
    if (T == 4) return k_stage0<MT1, MT2, 4, XT>;
    return k_stage0<MT1, MT2, 1, XT>;
}
template <int MT1, typename XT>
static StageFn pick_stage0_m2(int mt2, int T) {
    switch (mt2) {
        case 1: return pick_stage0_t<MT1, 1, XT>(T);
        case 2: return pick_stage0_t<MT1, 2, XT>(T);
        case 3: return pick_stage0_t<MT1, 3, XT>(T);
        default: return pick_stage0_t<MT1, 4, XT>(T);
    }
}
template <typename XT>
static StageFn pick_stage0_x(int mt1, int mt2, int T) {
    switch (mt1) {
        case 1: return pick_stage0_m2<1, XT>(mt2, T);
        case 2: return pick_stage0_m2<2, XT>(mt2, T);
        case 3: return pick_stage0_m2<3, XT>(mt2, T);
        default: return pick_stage0_m2<4, XT>(mt2, T);
    }
}
StageFn pick_stage0(int mt1, int mt2, int T, int x_dtype) {
    switch (x_dtype) {
        case HG_U8: return pick_stage0_x<uint8_t>(mt1, mt2, T);
        case HG_F32: return pick_stage0_x<float>(mt1, mt2, T);
        default: return pick_stage0_x<double>(mt1, mt2, T);
    }
}
StageFn pick_stage0p(int x_dtype) {
    return x_dtype == HG_U8 ? (StageFn)k_stage0p<4, uint8_t> : x_dtype == HG_F32 ? (StageFn)k_stage0p<4, float> : (StageFn)k_stage0p<4, double>;
}
// Tiles per pass of the instantiation pick_stage01p returns (the host sizes LDS and the group count with it): the
// compile-time-expansion form takes ONE tile per pass at four waves per SIMD (124 VGPRs; 146 us against 153 with two
// tiles at three waves), the generic form two tiles at three waves (168 VGPRs).
int stage01p_tiles(bool rem4, bool fspec) { return fspec ? 1 : 2; }

StageFn2 pick_stage01p(int x_dtype, bool stamp, bool rem4, bool fspec) {
    if (rem4) {
#ifdef HIGSFA_DIAG
        if (stamp && x_dtype == HG_F32) return fspec ? (StageFn2)k_stage01p<float, true, true, 1, 1> : (StageFn2)k_stage01p<float, true, true>;
#endif
        if (fspec)
            return x_dtype == HG_U8 ? (StageFn2)k_stage01p<uint8_t, false, true, 1, 1>
                                    : x_dtype == HG_F32 ? (StageFn2)k_stage01p<float, false, true, 1, 1> : (StageFn2)k_stage01p<double, false, true, 1, 1>;
        return x_dtype == HG_U8 ? (StageFn2)k_stage01p<uint8_t, false, true>
                                : x_dtype == HG_F32 ? (StageFn2)k_stage01p<float, false, true> : (StageFn2)k_stage01p<double, false, true>;
    }
#ifdef HIGSFA_DIAG
    if (stamp && x_dtype == HG_F32) return (StageFn2)k_stage01p<float, true>;
#endif
    if (fspec)      // layer 1 without remainder tiles (e.g. folded iGSFA nodes): still one tile per pass, expansion at compile time
        return x_dtype == HG_U8 ? (StageFn2)k_stage01p<uint8_t, false, false, 1, 1>
                                : x_dtype == HG_F32 ? (StageFn2)k_stage01p<float, false, false, 1, 1> : (StageFn2)k_stage01p<double, false, false, 1, 1>;
    return x_dtype == HG_U8 ? (StageFn2)k_stage01p<uint8_t> : x_dtype == HG_F32 ? (StageFn2)k_stage01p<float> : (StageFn2)k_stage01p<double>;
}

}  // namespace fused
}  // namespace hg
