// k_tail: the top of the hierarchy — the last one to three ordinary layers (4, 2, 1 nodes in the 11-layer nets) — as ONE
// launch, ending in the caller's row-major y.
//
// Why: at the top there is next to no arithmetic left (layers 8-10 of U11L-128 hold 1.8 % of the FLOPs) and a launch is all
// latency: start, one round of weight / input loads, two short dependent MFMA chains with an LDS exchange between them,
// store, end — 7-14 us per layer whatever the batch, plus 5 us for the k_unpack pass that turned fragment order into the
// caller's rows (profiles/r02_summary.md: 37 us for 0.83 GFLOP).  Here a workgroup of (nodes of the widest fused layer) x
// (m-tiles) waves takes T batch tiles through all fused layers: wave w is (node, m-tile) of the layer it is working on, like
// k_stage_splitm (hg_fused.hip), every weight block is read by exactly one wave straight from L2, the activations between
// the layers stay in LDS in fragment order, and the last layer's accumulators go straight to y[:, :y_cols] — of the last
// node's second affine only the output tiles that hold a requested column are computed (the caller reads sl[:, 0:k],
// FaceDetectUpdated.py:709-719).
//
// Same products in the same order as k_stage / k_stage_splitm (bias first, K-blocks and k-steps ascending), so a network gives
// the same bits whichever of the kernels runs its top layers.
#include <hip/hip_runtime.h>

#include <map>
#include <utility>

#include "hg_fused_dev.hpp"

namespace hg {
namespace fused {

namespace {

constexpr int KB = 8;      // K-blocks loaded per batch (all of a 120-input node's)

enum { OUT_LDS = 0, OUT_Y = 1, OUT_FRAG = 2 };

// Diagnostic build (HIGSFA_DIAG, tools/build_diag_lib.sh; HIGSFA_STAMP=<stage> picks the launch): wall-clock stamps (100 MHz) of
// every wave at the kernel's entry and, per layer, when its loads have arrived, when its expanded tiles are stored, behind the
// first barrier, when its outputs are stored, behind the second barrier.  The product build compiles none of it.
#ifdef HIGSFA_DIAG
#define TAIL_STAMP(i)                                                                                            \
    do {                                                                                                         \
        if (P.stamps && lane == 0) P.stamps[((size_t)blockIdx.x * 16 + w) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define TAIL_LOADS_DONE() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define TAIL_STAMP(i)
#define TAIL_LOADS_DONE()
#endif

// One layer for this workgroup's T tiles.  FIRST: inputs come from global memory (P.in), else from the LDS buffer `src`.
// OUT_Y: outputs go to the caller's y; OUT_FRAG: to P.out_frag in fragment order; OUT_LDS: to the LDS buffer `dst` (block-major:
// block b of tile t at (b * T + t) * 64).  k_subtree: the workgroup's nodes are nmap[0 .. S.n_nodes - 1] of the layer.
template <int T, bool FIRST, int OUT>
__device__ __forceinline__ void tail_layer(const TailParams& P, const TailStage& S, const f32x4* src, f32x4* dst, f32x4* ebuf,
                                           const int (&tile)[T], const uint32_t (&trow)[T], int w, int lane,
                                           const int32_t* nmap = nullptr, int sidx = 0) {
    constexpr bool LAST = OUT == OUT_Y;
    const int g = lane >> 4;
    const int wpn = S.has_exp ? (S.mt1 > S.mt2 ? S.mt1 : S.mt2) : S.mt1;      // waves per node
    const int node = w / wpn, mw = w - node * wpn;
    const bool on = node < S.n_nodes;
    const int gnode = nmap ? nmap[on ? node : 0] : on ? node : 0;      // (wave-uniform: a scalar load)
    const f32x4* wnode = S.afrag + (size_t)gnode * S.node_blocks * 64 + lane;
    const float* bnode = S.bias + (size_t)gnode * S.bias_floats + g * 4;
    const int2* kt = S.kb1tab + (size_t)gnode * S.kb1;
    const int nf = S.nf, mt1n = S.mt1, mt2n = S.mt2;
    const int out_blk = node * S.mto + mw;
    // which caller column each of this lane's four values of the output tile goes to (LAST only)
    int col[4] = {-1, -1, -1, -1};
    bool need_out = true;
    if constexpr (LAST) {
        const bool emits = on && mw < (S.has_exp ? mt2n : mt1n);
        bool mine = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (emits) col[r] = P.col_of[out_blk * 16 + 4 * r + g];
            if (col[r] >= P.y_cols) col[r] = -1;
            mine |= col[r] >= 0;
        }
        need_out = __builtin_amdgcn_ballot_w64(mine) != 0;       // a tile without a requested column is not computed
    }
    auto emit = [&](const f32x4 (&v)[T]) {
        if constexpr (LAST) {
            const int j = lane & 15;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t row = (int64_t)tile[t] * 16 + j;
                if (tile[t] < P.n_tiles && row < P.n_rows) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col[r] >= 0) {
                            if (P.y_f64) ((double*)P.y)[row * P.ldy + col[r]] = (double)v[t][r];
                            else ((float*)P.y)[row * P.ldy + col[r]] = v[t][r];
                        }
                }
            }
        } else if constexpr (OUT == OUT_FRAG) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out_frag[((size_t)tile[t] * P.nb_out_frag + gnode * S.mto + mw) * 64 + lane] = v[t];
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) dst[((size_t)out_blk * T + t) * 64 + lane] = v[t];
        }
    };
    const bool g1 = on && mw < mt1n && (S.has_exp || need_out);
    const bool g2 = on && S.has_exp && mw < mt2n && need_out;
    // GEMM-2 weights of this wave's output tile do not depend on z: fetch them first
    // (measured late in round 5: requesting them BEHIND the first GEMM's blocks, so that the first MFMA does not wait for their 8 KiB
    // per wave as well, costs 2 us per short call — profiles/r05_tail_warm.txt)
    f32x4 a2[KB];
    const int k2n = mt1n * nf;
    if (g2) {
#pragma unroll
        for (int k = 0; k < KB; ++k)
            if (k < k2n) a2[k] = wnode[((size_t)S.kb1 * mt1n + (size_t)k * mt2n + mw) * 64];
    }
    if (g1) {
        f32x4 z[T];
        const f32x4 bb = *(const f32x4*)(bnode + mw * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) z[t] = bb;
        for (int k0 = 0; k0 < S.kb1; k0 += KB) {
            // the whole batch of K-block entries first (scalar loads that do not wait for one another; the table has 8 spare
            // entries behind the last node), then every weight / input block of the batch, then the MFMAs
            int2 ent[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) ent[k] = kt[k0 + k];
            f32x4 a1[KB], bf[KB][T];
            int nks[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const bool real = k0 + k < S.kb1;
                nks[k] = real ? ent[k].y : 0;
                const int sb = real ? ent[k].x : ent[0].x, wk = real ? k0 + k : k0;      // (blocks beyond the node's: a harmless re-read, no MFMA)
                a1[k] = wnode[((size_t)wk * mt1n + mw) * 64];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if constexpr (FIRST) bf[k][t] = P.in[(size_t)(trow[t] + sb) * 64 + lane];
                    else bf[k][t] = src[((size_t)sb * T + t) * 64 + lane];
                }
            }
#ifdef HIGSFA_DIAG
            if (k0 == 0) {
                TAIL_LOADS_DONE();
                TAIL_STAMP(1 + 5 * sidx);
            }
#endif
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int nk = nks[k] & 255, r0 = nks[k] >> 8;      // (r0 > 0: a packed remainder block of the layer below)
                if (nks[k] == 4) {      // a whole block: four k-steps, no branch between them
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < T; ++t) z[t] = MFMA16(a1[k][r], bf[k][t][r], z[t]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r >= r0 && r < nk) {
#pragma unroll
                            for (int t = 0; t < T; ++t) z[t] = MFMA16(a1[k][r], bf[k][t][r], z[t]);
                        }
                }
            }
        }
        if (!S.has_exp) {
            emit(z);
        } else {
            for (int fi = 0; fi < nf; ++fi) {
                const int fk = (S.funcp >> (4 * fi)) & 15;
                const float ex = S.expo[fi];
#pragma unroll
                for (int t = 0; t < T; ++t) ebuf[((((size_t)node * nf + fi) * mt1n + mw) * T + t) * 64 + lane] = apply_func_uniform(fk, ex, z[t]);
            }
        }
    }
    TAIL_STAMP(2 + 5 * sidx);
    __syncthreads();                 // expanded tiles of every node are in LDS (a linear layer passes straight through)
    TAIL_STAMP(3 + 5 * sidx);
    if (g2) {
        f32x4 y[T];
        const f32x4 bb = *(const f32x4*)(bnode + (mt1n + mw) * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) y[t] = bb;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (k >= k2n) continue;
            const int mt1 = k / nf, fi = k - mt1 * nf;
            const int nk = (S.nk2p[mt1] >> (4 * fi)) & 15;
            f32x4 e[T];
#pragma unroll
            for (int t = 0; t < T; ++t) e[t] = ebuf[((((size_t)node * nf + fi) * mt1n + mt1) * T + t) * 64 + lane];
            if (nk == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < T; ++t) y[t] = MFMA16(a2[k][r], e[t][r], y[t]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nk) {
#pragma unroll
                        for (int t = 0; t < T; ++t) y[t] = MFMA16(a2[k][r], e[t][r], y[t]);
                    }
            }
        }
        emit(y);
    }
    TAIL_STAMP(4 + 5 * sidx);
    if constexpr (OUT == OUT_LDS) __syncthreads();      // this layer's output tiles are in LDS; ebuf is free again
    TAIL_STAMP(5 + 5 * sidx);
}

template <int T, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_tail(TailParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TAIL_STAMP(0);
    int tile[T];
    uint32_t trow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        tile[t] = blockIdx.x * T + t;
        trow[t] = (uint32_t)(tile[t] < P.n_tiles ? tile[t] : tile[0]) * (uint32_t)P.nb_in;
    }
    f32x4* act0 = smem;
    f32x4* act1 = smem + (size_t)P.act_blocks * T * 64;
    f32x4* ebuf = act1 + (size_t)P.act_blocks * T * 64;
    if (P.n_stages == 1) {
        tail_layer<T, true, OUT_Y>(P, P.st[0], nullptr, nullptr, ebuf, tile, trow, w, lane, nullptr, 0);
    } else if (P.n_stages == 2) {
        tail_layer<T, true, OUT_LDS>(P, P.st[0], nullptr, act0, ebuf, tile, trow, w, lane, nullptr, 0);
        tail_layer<T, false, OUT_Y>(P, P.st[1], act0, nullptr, ebuf, tile, trow, w, lane, nullptr, 1);
    } else {
        tail_layer<T, true, OUT_LDS>(P, P.st[0], nullptr, act0, ebuf, tile, trow, w, lane, nullptr, 0);
        tail_layer<T, false, OUT_LDS>(P, P.st[1], act0, act1, ebuf, tile, trow, w, lane, nullptr, 1);
        tail_layer<T, false, OUT_Y>(P, P.st[2], act1, nullptr, ebuf, tile, trow, w, lane, nullptr, 2);
    }
}

// k_subtree: two or three layers BELOW the top as one launch for short batches.  Where the receptive fields tile the layer below
// without overlap (every node of a layer is read by exactly one node of the next: all through the upper half of the preset
// hierarchies), a run of layers falls into n_sub independent sub-trees (the planner lists each one's nodes: plan_subtree in
// hg_fused.hip); a workgroup takes one sub-tree for one batch tile the way k_tail takes the whole top, and writes the sub-tree's
// root tiles back in fragment order.  A short batch (one 1080p frame's later cascade stages: 18 .. 348 windows) leaves every
// per-layer launch at its floor of 9-12 us; a run replaces three of them.  Consecutive workgroups are the sub-trees of one tile.
// What it costs: a workgroup pulls its sub-tree's weights (up to 784 KiB for 4 + 2 + 1 nodes) through ONE compute unit's L1, so a
// run is used only while all its workgroups are resident at once (FusedExec::sub_run_pays; profiles/r05_tail_stamps.txt).
template <int T, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_subtree(TailParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TAIL_STAMP(0);
    const int sub = blockIdx.x % P.n_sub, tb = blockIdx.x / P.n_sub;
    int tile[T];
    uint32_t trow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        tile[t] = tb * T + t;
        trow[t] = (uint32_t)(tile[t] < P.n_tiles ? tile[t] : tile[0]) * (uint32_t)P.nb_in;
    }
    f32x4* act0 = smem;
    f32x4* act1 = smem + (size_t)P.act_blocks * T * 64;
    f32x4* ebuf = act1 + (size_t)P.act_blocks * T * 64;
    const int32_t* m0 = P.sub_nodes[0] + sub * P.st[0].n_nodes;
    const int32_t* m1 = P.sub_nodes[1] + sub * P.st[1].n_nodes;
    if (P.n_stages == 2) {
        tail_layer<T, true, OUT_LDS>(P, P.st[0], nullptr, act0, ebuf, tile, trow, w, lane, m0, 0);
        tail_layer<T, false, OUT_FRAG>(P, P.st[1], act0, nullptr, ebuf, tile, trow, w, lane, m1, 1);
    } else {
        const int32_t* m2 = P.sub_nodes[2] + sub * P.st[2].n_nodes;
        tail_layer<T, true, OUT_LDS>(P, P.st[0], nullptr, act0, ebuf, tile, trow, w, lane, m0, 0);
        tail_layer<T, false, OUT_LDS>(P, P.st[1], act0, act1, ebuf, tile, trow, w, lane, m1, 1);
        tail_layer<T, false, OUT_FRAG>(P, P.st[2], act1, nullptr, ebuf, tile, trow, w, lane, m2, 2);
    }
}

template <int T, bool SUB = false>
void launch_t(const TailParams& P, int waves, unsigned grid, size_t lds, hipStream_t st) {
    auto go = [&](auto fn) {
        if (lds > 64 * 1024) {      // raise the dynamic-LDS limit once per (device, function)
            static thread_local std::map<std::pair<int, const void*>, size_t> raised;
            int dev = 0;
            HG_HIP(hipGetDevice(&dev));
            size_t& have = raised[{dev, (const void*)fn}];
            if (have < lds) {
                HG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                have = lds;
            }
        }
        hipLaunchKernelGGL(fn, grid, waves * 64, lds, st, P);
    };
    if constexpr (SUB) {
        if (waves <= 8) go(k_subtree<T, 8>);
        else go(k_subtree<T, 16>);
    } else {
        if (waves <= 4) go(k_tail<T, 4>);
        else if (waves <= 8) go(k_tail<T, 8>);
        else go(k_tail<T, 16>);
    }
}

}  // namespace

int tail_waves(const TailParams& P) {
    int wv = 1;
    for (int s = 0; s < P.n_stages; ++s) {
        const TailStage& S = P.st[s];
        wv = std::max(wv, S.n_nodes * (S.has_exp ? std::max(S.mt1, S.mt2) : S.mt1));
    }
    return wv <= 4 ? 4 : wv <= 8 ? 8 : 16;
}

size_t tail_lds_bytes(const TailParams& P, int T) { return ((size_t)2 * P.act_blocks + P.e_blocks) * T * 1024; }

void launch_tail(const TailParams& P, int T, hipStream_t st) {
    const int waves = tail_waves(P);
    const unsigned grid = (unsigned)((P.n_tiles + T - 1) / T);
    const size_t lds = tail_lds_bytes(P, T);
    if (T == 2) launch_t<2>(P, waves, grid, lds, st);
    else launch_t<1>(P, waves, grid, lds, st);
    HG_HIP(hipGetLastError());
}

void launch_subtree(const TailParams& P, hipStream_t st) {
    if (P.n_stages < 2 || P.n_sub < 1) fail(HG_ERR_STATE, "k_subtree: two or three layers, at least one sub-tree");
    const int waves = tail_waves(P);
    launch_t<1, true>(P, waves, (unsigned)P.n_sub * (unsigned)P.n_tiles, tail_lds_bytes(P, 1), st);
    HG_HIP(hipGetLastError());
}

}  // namespace fused
}  // namespace hg
