// Fused executor: one kernel per network layer ("stage" = Switchboard gather + per-node
// [affine -> element-wise expansion -> affine]), fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// Data layout ("fragment order").  A batch tile is 16 sub-images.  Every node output is cut
// into feature tiles of 16; one (batch tile, feature tile) pair is a 1 KiB block
//       block[lane 0..63][reg 0..3]   lane = 16*g + j   (j = sub-image in tile, g = 0..3)
// holding feature q = 4*reg + g of the tile for sub-image j.  This is exactly the C/D register
// image of v_mfma_f32_16x16x4_f32 (col = lane&15, row = 4*(lane>>4) + reg) under the row
// assignment row(q) = 4*(q&3) + (q>>2), so
//   * a producer stores each accumulator as one coalesced 16 B/lane (1 KiB/wave) write,
//   * a consumer loads a block with one 16 B/lane read and has FOUR k-steps of the MFMA B
//     operand in registers (k-step r: lane group g supplies feature 4r+g) — no shuffles, no LDS,
//   * inside a node the first affine's accumulators ARE the B operand of the second affine;
//     the expansion (|x|^0.8, ...) is applied to them in registers.
// The K order of every contraction is therefore permuted; the planner permutes the weight
// matrices to match (A fragments, [k-block][m-tile][lane][4]) and folds Switchboards into the
// K-block lists, so a gather costs nothing at run time.  Stage 0 reads the caller's row-major
// sub-image matrix: coalesced row segments go through an LDS tile and the 4x4 (or any) receptive
// field is picked out of LDS by per-lane offsets.
//
// Reference semantics restated: SURVEY.md §8a rows a3-a7 (Switchboard, Layer, PCANode,
// GeneralExpansionNode, SFANode) behind the call FaceDetectUpdated.py:699.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <sstream>

#include "hg_common.hpp"

namespace hg {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxMT = 4;            // up to 64 outputs per affine in the fused plan
constexpr int kStage0MaxCols = 1022;  // columns of one sub-image staged in LDS per chunk

// ---- host-side normal form ---------------------------------------------------------------------
struct Aff {  // y = (x - a) W + b
    int in = 0, out = 0;
    std::vector<double> a, W, b;
};

Aff aff_of(const TNode& n) { return Aff{(int)n.in_dim, (int)n.out_dim, n.a, n.W, n.b}; }

Aff fold(const Aff& f, const Aff& s) {  // s(f(x)) = (x - f.a)(f.W s.W) + (f.b - s.a) s.W + s.b
    Aff r;
    r.in = f.in;
    r.out = s.out;
    r.a = f.a;
    r.W.assign((size_t)f.in * s.out, 0.0);
    for (int i = 0; i < f.in; ++i)
        for (int k = 0; k < f.out; ++k) {
            double w = f.W[(size_t)i * f.out + k];
            if (w == 0.0) continue;
            for (int j = 0; j < s.out; ++j) r.W[(size_t)i * s.out + j] += w * s.W[(size_t)k * s.out + j];
        }
    r.b = s.b;
    for (int k = 0; k < f.out; ++k) {
        double d = f.b[k] - s.a[k];
        for (int j = 0; j < s.out; ++j) r.b[j] += d * s.W[(size_t)k * s.out + j];
    }
    return r;
}

struct FNode {
    int in_off = 0, in_dim = 0, out_dim = 0;
    Aff A1, A2;
    bool has_exp = false;
    std::vector<ExpFunc> funcs;
};

struct FStage {
    std::vector<int32_t> conn;  // stage input column -> column of the previous frame (or of x)
    std::vector<FNode> nodes;
    int out_w = 0;
};

typedef std::vector<const TNode*> LeafSeq;

bool flatten_leafs(const TNode& n, LeafSeq& s, std::string& why) {
    switch (n.kind) {
        case K_AFFINE:
        case K_EXPANSION: s.push_back(&n); return true;
        case K_IDENTITY: return true;
        case K_FLOWNODE:
        case K_FLOW:
            for (auto& c : n.ch)
                if (!flatten_leafs(*c, s, why)) return false;
            return true;
        default: why = std::string("node kind ") + kind_name(n.kind) + " inside a layer is not covered by the fused plan"; return false;
    }
}

struct ChainT {
    int in_dim, out_dim;
    LeafSeq seq;
};

bool to_chains(const TNode& n, std::vector<ChainT>& out, std::string& why) {
    auto one = [&](const TNode& c) {
        ChainT ch{(int)c.in_dim, (int)c.out_dim, {}};
        if (!flatten_leafs(c, ch.seq, why)) return false;
        out.push_back(std::move(ch));
        return true;
    };
    if (n.kind == K_LAYER) {
        for (auto& c : n.ch)
            if (!one(*c)) return false;
        return true;
    }
    if (n.kind == K_CLONELAYER) {
        for (uint32_t i = 0; i < n.aux; ++i)
            if (!one(*n.ch[0])) return false;
        return true;
    }
    return one(n);
}

bool canon(const ChainT& c, int in_off, FNode& fn, std::string& why) {
    fn.in_off = in_off;
    fn.in_dim = c.in_dim;
    fn.out_dim = c.out_dim;
    int phase = 0;  // 0: before A1, 1: in A1, 2: after E, 3: in A2
    for (const TNode* l : c.seq) {
        if (l->kind == K_AFFINE) {
            if (phase == 0) { fn.A1 = aff_of(*l); phase = 1; }
            else if (phase == 1) fn.A1 = fold(fn.A1, aff_of(*l));
            else if (phase == 2) { fn.A2 = aff_of(*l); phase = 3; }
            else fn.A2 = fold(fn.A2, aff_of(*l));
        } else {  // expansion
            if (phase != 1) { why = "node chain is not [affine][expansion][affine]"; return false; }
            for (const ExpFunc& f : l->funcs)
                if (f.kind > E_SIGNED_POW) { why = "expansion with cross-column products (QT / pair products)"; return false; }
            fn.funcs = l->funcs;
            fn.has_exp = true;
            phase = 2;
        }
    }
    if (phase == 0) { why = "layer node without an affine part"; return false; }
    if (phase == 2) { why = "node chain ends in an expansion"; return false; }
    if (fn.A1.out > 16 * kMaxMT || (fn.has_exp && fn.A2.out > 16 * kMaxMT)) {
        why = "affine with more than 64 outputs";
        return false;
    }
    return true;
}

bool build_stages(const TNode& root, std::vector<FStage>& stages, std::string& why) {
    std::vector<int32_t> pending;  // composition of switchboards since the last layer group
    bool have_pending = false;
    std::vector<ChainT> group;
    int frame_w = root.in_dim;

    auto close_group = [&]() -> bool {
        if (group.empty()) return true;
        FStage st;
        int in_w = 0;
        for (auto& c : group) in_w += c.in_dim;
        if (have_pending) {
            if ((int)pending.size() != in_w) { why = "internal: connection count"; return false; }
            st.conn = pending;
        } else {
            if (in_w != frame_w) { why = "internal: frame width"; return false; }
            st.conn.resize(in_w);
            for (int i = 0; i < in_w; ++i) st.conn[i] = i;
        }
        int off = 0;
        for (auto& c : group) {
            FNode fn;
            if (!canon(c, off, fn, why)) return false;
            off += c.in_dim;
            st.out_w += fn.out_dim;
            st.nodes.push_back(std::move(fn));
        }
        frame_w = st.out_w;
        stages.push_back(std::move(st));
        group.clear();
        pending.clear();
        have_pending = false;
        return true;
    };

    for (auto& cp : root.ch) {
        const TNode& c = *cp;
        if (c.kind == K_SWITCHBOARD) {
            if (!close_group()) return false;
            if (have_pending) {
                std::vector<int32_t> comp(c.conn.size());
                for (size_t i = 0; i < c.conn.size(); ++i) comp[i] = pending[c.conn[i]];
                pending.swap(comp);
            } else {
                pending = c.conn;
                have_pending = true;
            }
            continue;
        }
        std::vector<ChainT> chains;
        if (!to_chains(c, chains, why)) return false;
        bool merged = false;
        if (!group.empty() && group.size() == chains.size()) {
            merged = true;
            for (size_t k = 0; k < chains.size(); ++k)
                if (group[k].out_dim != chains[k].in_dim) { merged = false; break; }
            // merging [A][E][A] + another [A ...] is fine (folds); anything after A2 with an expansion is not
            if (merged)
                for (size_t k = 0; k < chains.size() && merged; ++k) {
                    int n_exp = 0;
                    for (auto* l : group[k].seq) n_exp += l->kind == K_EXPANSION;
                    for (auto* l : chains[k].seq) n_exp += l->kind == K_EXPANSION;
                    if (n_exp > 1) merged = false;
                }
            if (merged)
                for (size_t k = 0; k < chains.size(); ++k) {
                    group[k].out_dim = chains[k].out_dim;
                    for (auto* l : chains[k].seq) group[k].seq.push_back(l);
                }
        }
        if (!merged) {
            if (!close_group()) return false;
            group = std::move(chains);
        }
    }
    if (!close_group()) return false;
    if (have_pending) { why = "flow ends in a switchboard"; return false; }
    if (stages.empty()) { why = "no layer in the flow"; return false; }
    return true;
}

// ---- device descriptors ----------------------------------------------------------------------
struct DNode {
    int32_t kb_begin, kb_count;  // GEMM-1 K-blocks
    int32_t a1_blk, b1_off;      // A1 fragments [kb][mt1] (1 KiB blocks); bias fragment [MT1][16] floats
    int32_t k2_begin, nf;        // GEMM-2 K-blocks [mt1][nf]
    int32_t a2_blk, b2_off;      // A2 fragments [mt1][nf][mt2]
    int32_t out_blk, pad0, pad1, pad2;
};
struct DKB1 {
    int32_t src;  // stage>0: block index inside the input batch-tile row; stage 0: entry of the offset table
    int32_t nk;   // k-steps used (1..4)
};
struct DKB2 {
    int32_t nk, func;  // func: ExpKind
    float expo;
    int32_t pad;
};
struct DChunk {
    int32_t node_begin, node_count, run_begin, run_count;
};
struct DRun {
    int32_t start, len, lds_off, pad;
};

struct StageParams {
    const DNode* nodes;
    const DKB1* kb1;
    const DKB2* kb2;
    const f32x4* afrag;   // all A fragments of the stage, 64 x f32x4 per block
    const float* bias;    // bias fragments
    const f32x4* in;      // input activation (stage > 0)
    f32x4* out;           // output activation
    int32_t n_nodes, nodes_per_wg, n_tiles, nb_in, nb_out, has_exp;
    // stage 0 only
    const DChunk* chunks;
    const DRun* runs;
    const i32x4* koff;    // [entry][g] -> 4 LDS word offsets (r = 0..3)
    const f32x4* kmean;   // [entry][g] -> 4 pre-subtracted means
    const void* x;
    int64_t ldx, n_rows;
    int32_t lds_stride, n_chunks;
};

__device__ __forceinline__ float pow_abs(float v, float p) {
    // |v|^p = exp2(p * log2|v|); v = 0 -> log2 = -inf -> exp2 = 0 exactly
    return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(__builtin_fabsf(v)));
}

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

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// GEMM-2 half of a node: expansion of the z accumulators in registers, second affine, store.
template <int MT1, int MT2, int T>
__device__ __forceinline__ void node_tail(const StageParams& P, const DNode& nd, f32x4 (&z)[MT1][T], const int (&tile)[T],
                                          int lane) {
    const int g = lane >> 4;
    if (!P.has_exp) {
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + nd.out_blk + mt) * 64 + lane] = z[mt][t];
        return;
    }
    f32x4 y[MT2][T];
    const float* b2 = P.bias + nd.b2_off;
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
        f32x4 bb = *(const f32x4*)(b2 + mt * 16 + g * 4);
#pragma unroll
        for (int t = 0; t < T; ++t) y[mt][t] = bb;
    }
#pragma unroll
    for (int mt1 = 0; mt1 < MT1; ++mt1) {
        for (int fi = 0; fi < nd.nf; ++fi) {
            const DKB2 kb = P.kb2[nd.k2_begin + mt1 * nd.nf + fi];
            if (kb.nk == 0) continue;
            f32x4 e[T];
#pragma unroll
            for (int t = 0; t < T; ++t) e[t] = apply_func(kb.func, kb.expo, z[mt1][t]);
            const f32x4* ap = P.afrag + ((size_t)nd.a2_blk + (size_t)(mt1 * nd.nf + fi) * MT2) * 64 + lane;
            f32x4 a[MT2];
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) a[mt] = ap[mt * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < kb.nk) {
#pragma unroll
                    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                        for (int t = 0; t < T; ++t) y[mt][t] = MFMA16(a[mt][r], e[t][r], y[mt][t]);
                }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + nd.out_blk + mt) * 64 + lane] = y[mt][t];
}

// Stage > 0: input in fragment order.  WG = 4 waves; wave w owns T batch tiles; all waves walk the
// same nodes (weights hit L1/L2), grid = node chunks x batch-tile groups.
template <int MT1, int MT2, int T>
__global__ void __launch_bounds__(256) k_stage(StageParams P, int n_groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int chunk = blockIdx.x / n_groups, grp = blockIdx.x % n_groups;
    int tile[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tile[t] = (grp * 4 + wave) * T + t;
    if (tile[0] >= P.n_tiles) return;
    const int n0 = chunk * P.nodes_per_wg;
    const int n1 = min(n0 + P.nodes_per_wg, P.n_nodes);
    for (int ni = n0; ni < n1; ++ni) {
        const DNode nd = P.nodes[ni];
        f32x4 z[MT1][T];
        const float* b1 = P.bias + nd.b1_off;
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
            f32x4 bb = *(const f32x4*)(b1 + mt * 16 + g * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) z[mt][t] = bb;
        }
        for (int kbi = 0; kbi < nd.kb_count; ++kbi) {
            const DKB1 kb = P.kb1[nd.kb_begin + kbi];
            f32x4 bf[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int tl = tile[t] < P.n_tiles ? tile[t] : tile[0];
                bf[t] = P.in[((size_t)tl * P.nb_in + kb.src) * 64 + lane];
            }
            const f32x4* ap = P.afrag + ((size_t)nd.a1_blk + (size_t)kbi * MT1) * 64 + lane;
            f32x4 a[MT1];
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) a[mt] = ap[mt * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < kb.nk) {
#pragma unroll
                    for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
                        for (int t = 0; t < T; ++t) z[mt][t] = MFMA16(a[mt][r], bf[t][r], z[mt][t]);
                }
        }
        node_tail<MT1, MT2, T>(P, nd, z, tile, lane);
    }
}

// Stage 0: input = caller's row-major sub-image matrix.  The WG stages, for T batch tiles, the
// column runs its node chunk needs (coalesced along the row) into LDS; each wave then takes every
// 4th node of the chunk and reads its receptive field out of LDS via per-lane offsets.
template <int MT1, int MT2, int T, typename XT>
__global__ void __launch_bounds__(256) k_stage0(StageParams P, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const int ci = blockIdx.x / n_groups, grp = blockIdx.x % n_groups;
    const DChunk ck = P.chunks[ci];
    int tile[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tile[t] = grp * T + t;
    const XT* x = (const XT*)P.x;
    const int stride = P.lds_stride;
    // --- stage the input runs: wave w copies sub-images w, w+4, ... of every tile
    for (int t = 0; t < T; ++t) {
        for (int jj = wave; jj < 16; jj += 4) {
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
    for (int ni = ck.node_begin + wave; ni < ck.node_begin + ck.node_count; ni += 4) {
        const DNode nd = P.nodes[ni];
        f32x4 z[MT1][T];
        const float* b1 = P.bias + nd.b1_off;
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
            f32x4 bb = *(const f32x4*)(b1 + mt * 16 + g * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) z[mt][t] = bb;
        }
        for (int kbi = 0; kbi < nd.kb_count; ++kbi) {
            const DKB1 kb = P.kb1[nd.kb_begin + kbi];
            const i32x4 off = P.koff[(size_t)kb.src * 4 + g];
            const f32x4 mu = P.kmean[(size_t)kb.src * 4 + g];
            f32x4 bf[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float* base = lds + (t * 16 + j) * stride;
#pragma unroll
                for (int r = 0; r < 4; ++r) bf[t][r] = base[off[r]] - mu[r];
            }
            const f32x4* ap = P.afrag + ((size_t)nd.a1_blk + (size_t)kbi * MT1) * 64 + lane;
            f32x4 a[MT1];
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) a[mt] = ap[mt * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < kb.nk) {
#pragma unroll
                    for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
                        for (int t = 0; t < T; ++t) z[mt][t] = MFMA16(a[mt][r], bf[t][r], z[mt][t]);
                }
        }
        node_tail<MT1, MT2, T>(P, nd, z, tile, lane);
    }
}

// Fragment order -> caller's row-major y (first y_cols columns).
template <typename YT>
__global__ void k_unpack(const float* __restrict__ act, int nb, const int32_t* __restrict__ col_base, YT* __restrict__ y,
                         int64_t ldy, int64_t n, int cols) {
    int64_t total = n * cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t row = idx / cols;
        int c = (int)(idx - row * cols);
        int64_t tile = row >> 4;
        int jj = (int)(row & 15);
        y[row * ldy + c] = (YT)act[tile * nb * 256 + col_base[c] + jj * 4];
    }
}

// ---- launch tables -----------------------------------------------------------------------------
typedef void (*StageFn)(StageParams, int);

template <int MT1, int MT2>
StageFn pick_stage_t(int T) {
    if (T == 2) return k_stage<MT1, MT2, 2>;
    return k_stage<MT1, MT2, 1>;
}
template <int MT1>
StageFn pick_stage_m2(int mt2, int T) {
    switch (mt2) {
        case 1: return pick_stage_t<MT1, 1>(T);
        case 2: return pick_stage_t<MT1, 2>(T);
        case 3: return pick_stage_t<MT1, 3>(T);
        default: return pick_stage_t<MT1, 4>(T);
    }
}
StageFn pick_stage(int mt1, int mt2, int T) {
    switch (mt1) {
        case 1: return pick_stage_m2<1>(mt2, T);
        case 2: return pick_stage_m2<2>(mt2, T);
        case 3: return pick_stage_m2<3>(mt2, T);
        default: return pick_stage_m2<4>(mt2, T);
    }
}

template <int MT1, int MT2, typename XT>
StageFn pick_stage0_t(int T) {
    if (T == 2) return k_stage0<MT1, MT2, 2, XT>;
    return k_stage0<MT1, MT2, 1, XT>;
}
template <int MT1, typename XT>
StageFn pick_stage0_m2(int mt2, int T) {
    switch (mt2) {
        case 1: return pick_stage0_t<MT1, 1, XT>(T);
        case 2: return pick_stage0_t<MT1, 2, XT>(T);
        case 3: return pick_stage0_t<MT1, 3, XT>(T);
        default: return pick_stage0_t<MT1, 4, XT>(T);
    }
}
template <typename XT>
StageFn pick_stage0_x(int mt1, int mt2, int T) {
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

// ---- the executor --------------------------------------------------------------------------------
inline int q_of_row(int i) { return 4 * (i & 3) + (i >> 2); }  // tile row -> tile-local feature (involution)

struct HostStage {
    int mt1 = 1, mt2 = 1, nb_out = 0, nb_in = 0, n_nodes = 0;
    bool has_exp = false;
    std::vector<DNode> nodes;
    std::vector<DKB1> kb1;
    std::vector<DKB2> kb2;
    std::vector<float> afrag;  // blocks of 256 floats
    std::vector<float> bias;
    // stage 0
    std::vector<DChunk> chunks;
    std::vector<DRun> runs;
    std::vector<int32_t> koff;   // [entry][g][r]
    std::vector<float> kmean;
    int lds_stride = 0;
    int64_t mfma_per_tile = 0;   // MFMA instructions per batch tile (padded work)
    std::string name;
    // device
    DevBuf d_nodes, d_kb1, d_kb2, d_afrag, d_bias, d_chunks, d_runs, d_koff, d_kmean;
};

class FusedExecutor : public Executor {
public:
    FusedExecutor(const TNode& root, std::vector<FStage>&& fs) : in_dim_(root.in_dim), out_dim_(root.out_dim) {
        // feature -> (block, q) maps of the previous stage's output frame
        std::vector<int32_t> prev_blk, prev_q;  // per previous-frame column
        int prev_nb = 0;
        for (size_t si = 0; si < fs.size(); ++si) {
            FStage& st = fs[si];
            stages_.emplace_back();
            HostStage& hs = stages_.back();
            hs.n_nodes = (int)st.nodes.size();
            hs.has_exp = st.nodes[0].has_exp;
            for (auto& nd : st.nodes) {
                if (nd.has_exp != hs.has_exp) fail(HG_ERR_FORMAT, "fused: mixed node forms in one layer");
                hs.mt1 = std::max(hs.mt1, (nd.A1.out + 15) / 16);
                if (nd.has_exp) hs.mt2 = std::max(hs.mt2, (nd.A2.out + 15) / 16);
            }
            if (!hs.has_exp) hs.mt2 = 1;
            hs.nb_in = prev_nb;
            const int mto = hs.has_exp ? hs.mt2 : hs.mt1;
            if (si == 0) plan_stage0_inputs(st, hs);
            std::vector<int32_t> cur_blk, cur_q;
            int out_blk = 0;
            for (size_t ni = 0; ni < st.nodes.size(); ++ni) {
                FNode& nd = st.nodes[ni];
                DNode dn{};
                // ---- GEMM 1 -------------------------------------------------------------------
                const int p = nd.A1.out;
                // K-blocks and, for every (kblock, r, g), the list of consumer input positions
                std::vector<std::vector<int>> kpos;  // [kb*16 + 4r+g] -> positions c
                std::vector<DKB1> kbs;
                std::vector<double> bias1 = nd.A1.b;
                if (si == 0) {
                    const int nkb = (nd.in_dim + 15) / 16;
                    for (int kb = 0; kb < nkb; ++kb) {
                        int valid = std::min(16, nd.in_dim - kb * 16);
                        kbs.push_back(DKB1{s0_entry_base_[ni] + kb, (valid + 3) / 4});
                        for (int q = 0; q < 16; ++q) {
                            kpos.emplace_back();
                            if (q < valid) kpos.back().push_back(kb * 16 + q);
                        }
                    }
                    // means are subtracted by the loader in fp32; the fp64 remainder goes into the bias
                    for (int c = 0; c < nd.in_dim; ++c) {
                        double rem = nd.A1.a[c] - (double)(float)nd.A1.a[c];
                        for (int o = 0; o < p; ++o) bias1[o] -= rem * nd.A1.W[(size_t)c * p + o];
                    }
                } else {
                    std::map<int, int> blk_index;  // source block -> local kb index
                    for (int c = 0; c < nd.in_dim; ++c) {
                        int pc = st.conn[nd.in_off + c];
                        int blk = prev_blk[pc], q = prev_q[pc];
                        auto it = blk_index.find(blk);
                        int kb;
                        if (it == blk_index.end()) {
                            kb = (int)kbs.size();
                            blk_index[blk] = kb;
                            kbs.push_back(DKB1{blk, 0});
                            for (int qq = 0; qq < 16; ++qq) kpos.emplace_back();
                        } else {
                            kb = it->second;
                        }
                        kpos[kb * 16 + q].push_back(c);
                        kbs[kb].nk = std::max(kbs[kb].nk, q / 4 + 1);
                    }
                    for (int c = 0; c < nd.in_dim; ++c)  // (x - a) W + b = x W + (b - a W)
                        for (int o = 0; o < p; ++o) bias1[o] -= nd.A1.a[c] * nd.A1.W[(size_t)c * p + o];
                }
                dn.kb_begin = (int)hs.kb1.size();
                dn.kb_count = (int)kbs.size();
                dn.a1_blk = (int)(hs.afrag.size() / 256);
                for (size_t kb = 0; kb < kbs.size(); ++kb) {
                    hs.kb1.push_back(kbs[kb]);
                    hs.mfma_per_tile += (int64_t)kbs[kb].nk * hs.mt1;
                    for (int mt = 0; mt < hs.mt1; ++mt) {
                        size_t base = hs.afrag.size();
                        hs.afrag.resize(base + 256, 0.f);
                        for (int lane = 0; lane < 64; ++lane) {
                            int i = lane & 15, gg = lane >> 4;
                            int fo = 16 * mt + q_of_row(i);
                            if (fo >= p) continue;
                            for (int r = 0; r < 4; ++r) {
                                double w = 0;
                                for (int c : kpos[kb * 16 + 4 * r + gg]) w += nd.A1.W[(size_t)c * p + fo];
                                hs.afrag[base + lane * 4 + r] = (float)w;
                            }
                        }
                    }
                }
                dn.b1_off = (int)hs.bias.size();
                for (int mt = 0; mt < hs.mt1; ++mt)
                    for (int gg = 0; gg < 4; ++gg)
                        for (int r = 0; r < 4; ++r) {
                            int fo = 16 * mt + 4 * r + gg;
                            hs.bias.push_back(fo < p ? (float)bias1[fo] : 0.f);
                        }
                // ---- GEMM 2 -------------------------------------------------------------------
                int n_out = p;
                if (hs.has_exp) {
                    const int s = nd.A2.out;
                    n_out = s;
                    dn.nf = (int)nd.funcs.size();
                    dn.k2_begin = (int)hs.kb2.size();
                    dn.a2_blk = (int)(hs.afrag.size() / 256);
                    std::vector<int> foff(nd.funcs.size());
                    int eo = 0;
                    for (size_t fi = 0; fi < nd.funcs.size(); ++fi) {
                        foff[fi] = eo;
                        eo += nd.funcs[fi].out_dim(p);
                    }
                    if (eo != nd.A2.in) fail(HG_ERR_DIM, "fused: expansion width %d != second affine input_dim %d", eo, nd.A2.in);
                    std::vector<double> bias2 = nd.A2.b;
                    for (int c = 0; c < nd.A2.in; ++c)
                        for (int o = 0; o < s; ++o) bias2[o] -= nd.A2.a[c] * nd.A2.W[(size_t)c * s + o];
                    for (int mt1 = 0; mt1 < hs.mt1; ++mt1)
                        for (size_t fi = 0; fi < nd.funcs.size(); ++fi) {
                            const ExpFunc& f = nd.funcs[fi];
                            int used = f.used(p);
                            int valid = std::max(0, std::min(16, used - 16 * mt1));
                            DKB2 kb{(valid + 3) / 4, (int32_t)f.kind, (float)f.expo, 0};
                            hs.kb2.push_back(kb);
                            hs.mfma_per_tile += (int64_t)kb.nk * hs.mt2;
                            for (int mt2 = 0; mt2 < hs.mt2; ++mt2) {
                                size_t base = hs.afrag.size();
                                hs.afrag.resize(base + 256, 0.f);
                                for (int lane = 0; lane < 64; ++lane) {
                                    int i = lane & 15, gg = lane >> 4;
                                    int fo = 16 * mt2 + q_of_row(i);
                                    if (fo >= s) continue;
                                    for (int r = 0; r < 4; ++r) {
                                        int fz = 16 * mt1 + 4 * r + gg;
                                        if (fz >= used) continue;
                                        hs.afrag[base + lane * 4 + r] = (float)nd.A2.W[(size_t)(foff[fi] + fz) * s + fo];
                                    }
                                }
                            }
                        }
                    dn.b2_off = (int)hs.bias.size();
                    for (int mt = 0; mt < hs.mt2; ++mt)
                        for (int gg = 0; gg < 4; ++gg)
                            for (int r = 0; r < 4; ++r) {
                                int fo = 16 * mt + 4 * r + gg;
                                hs.bias.push_back(fo < s ? (float)bias2[fo] : 0.f);
                            }
                }
                dn.out_blk = out_blk;
                for (int f = 0; f < n_out; ++f) {
                    cur_blk.push_back(out_blk + f / 16);
                    cur_q.push_back(f % 16);
                }
                out_blk += mto;
                hs.nodes.push_back(dn);
            }
            hs.nb_out = out_blk;
            prev_blk.swap(cur_blk);
            prev_q.swap(cur_q);
            prev_nb = hs.nb_out;
            max_nb_ = std::max(max_nb_, hs.nb_out);
            padded_flops_ += hs.mfma_per_tile * 2048 / 16;
            std::ostringstream os;
            os << "fused stage " << si << ": " << hs.n_nodes << " nodes, MT " << hs.mt1 << "x" << hs.mt2 << ", "
               << hs.mfma_per_tile << " MFMA/tile, " << hs.afrag.size() * 4 / 1024 << " KiB weights, out " << hs.nb_out << " blocks/tile";
            hs.name = os.str();
        }
        // final frame: column -> offset inside a batch-tile row
        col_base_.resize(out_dim_);
        for (int c = 0; c < out_dim_; ++c) {
            int q = prev_q[c];
            col_base_[c] = prev_blk[c] * 256 + (q & 3) * 64 + (q >> 2);
        }
    }

    int plan_kind() const override { return HG_PLAN_FUSED; }
    int n_stages() const override { return (int)stages_.size() + 1; }
    std::string stage_name(int i) const override {
        return i < (int)stages_.size() ? stages_[i].name : std::string("fused unpack (fragment order -> row-major y)");
    }
    std::string describe() const override {
        std::ostringstream os;
        os << "plan: FUSED (fragment-order activations, v_mfma_f32_16x16x4_f32)\n";
        for (int i = 0; i < n_stages(); ++i) os << "  [" << i << "] " << stage_name(i) << "\n";
        return os.str();
    }
    int64_t weight_bytes() const override {
        int64_t t = 0;
        for (auto& s : stages_) t += (int64_t)(s.afrag.size() + s.bias.size()) * 4;
        return t;
    }
    int64_t padded_flops_per_row() const override { return padded_flops_; }
    int64_t workspace_bytes() const override { return (int64_t)(bufA_.bytes + bufB_.bytes); }

    void to_device() override {
        for (auto& s : stages_) {
            s.d_nodes.upload(s.nodes.data(), s.nodes.size() * sizeof(DNode));
            s.d_kb1.upload(s.kb1.data(), s.kb1.size() * sizeof(DKB1));
            if (!s.kb2.empty()) s.d_kb2.upload(s.kb2.data(), s.kb2.size() * sizeof(DKB2));
            s.d_afrag.upload(s.afrag.data(), s.afrag.size() * 4);
            s.d_bias.upload(s.bias.data(), s.bias.size() * 4);
            if (!s.chunks.empty()) {
                s.d_chunks.upload(s.chunks.data(), s.chunks.size() * sizeof(DChunk));
                s.d_runs.upload(s.runs.data(), s.runs.size() * sizeof(DRun));
                s.d_koff.upload(s.koff.data(), s.koff.size() * 4);
                s.d_kmean.upload(s.kmean.data(), s.kmean.size() * 4);
            }
        }
        d_col_base_.upload(col_base_.data(), col_base_.size() * 4);
        int dev = 0;
        HG_HIP(hipGetDevice(&dev));
        for (int mt1 = 1; mt1 <= kMaxMT; ++mt1) (void)mt1;
    }

    void reserve(int64_t rows) override {
        int64_t tiles = (rows + 15) / 16;
        size_t need = (size_t)tiles * max_nb_ * 1024;
        bufA_.alloc(need);
        bufB_.alloc(need);
        cap_rows_ = std::max(cap_rows_, tiles * 16);
    }

    void run(const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols, int64_t ldy,
             hipStream_t st, hipEvent_t* ev) override {
        if (n > cap_rows_) reserve(n);
        const int n_tiles = (int)((n + 15) / 16);
        int e = 0;
        if (ev) HG_HIP(hipEventRecord(ev[e++], st));
        f32x4* cur = (f32x4*)bufA_.p;
        f32x4* nxt = (f32x4*)bufB_.p;
        for (size_t si = 0; si < stages_.size(); ++si) {
            HostStage& s = stages_[si];
            StageParams P{};
            P.nodes = (const DNode*)s.d_nodes.p;
            P.kb1 = (const DKB1*)s.d_kb1.p;
            P.kb2 = (const DKB2*)s.d_kb2.p;
            P.afrag = (const f32x4*)s.d_afrag.p;
            P.bias = (const float*)s.d_bias.p;
            P.in = cur;
            P.out = nxt;
            P.n_nodes = s.n_nodes;
            P.n_tiles = n_tiles;
            P.nb_in = s.nb_in;
            P.nb_out = s.nb_out;
            P.has_exp = s.has_exp ? 1 : 0;
            if (si == 0) {
                P.chunks = (const DChunk*)s.d_chunks.p;
                P.runs = (const DRun*)s.d_runs.p;
                P.koff = (const i32x4*)s.d_koff.p;
                P.kmean = (const f32x4*)s.d_kmean.p;
                P.x = x;
                P.ldx = ldx;
                P.n_rows = n;
                P.lds_stride = s.lds_stride;
                P.n_chunks = (int)s.chunks.size();
                const int T = n_tiles >= 2 ? 2 : 1;
                const int groups = (n_tiles + T - 1) / T;
                const int64_t blocks = (int64_t)groups * P.n_chunks;
                if (blocks > 0x7fffffffll) fail(HG_ERR_ARG, "batch too large");
                size_t lds_bytes = (size_t)T * 16 * s.lds_stride * 4;
                StageFn fn = pick_stage0(s.mt1, s.mt2, T, x_dtype);
                hipLaunchKernelGGL(fn, (unsigned)blocks, 256, lds_bytes, st, P, groups);
            } else {
                // enough workgroups to fill 256 CUs several times over when the batch allows
                int T = ((int64_t)n_tiles * s.n_nodes >= 8 * 1024) ? 2 : 1;
                int groups = (n_tiles + 4 * T - 1) / (4 * T);
                int per_wg = (int)std::max<int64_t>(1, (int64_t)s.n_nodes * groups / 2048);
                P.nodes_per_wg = per_wg;
                int chunks = (s.n_nodes + per_wg - 1) / per_wg;
                const int64_t blocks = (int64_t)groups * chunks;
                if (blocks > 0x7fffffffll) fail(HG_ERR_ARG, "batch too large");
                StageFn fn = pick_stage(s.mt1, s.mt2, T);
                hipLaunchKernelGGL(fn, (unsigned)blocks, 256, 0, st, P, groups);
            }
            std::swap(cur, nxt);
            if (ev) HG_HIP(hipEventRecord(ev[e++], st));
        }
        const HostStage& last = stages_.back();
        unsigned grid = (unsigned)std::min<int64_t>((n * y_cols + 255) / 256, 4096);
        if (y_dtype == HG_F32)
            hipLaunchKernelGGL(k_unpack<float>, grid, 256, 0, st, (const float*)cur, last.nb_out, (const int32_t*)d_col_base_.p,
                               (float*)y, ldy, n, (int)y_cols);
        else if (y_dtype == HG_F64)
            hipLaunchKernelGGL(k_unpack<double>, grid, 256, 0, st, (const float*)cur, last.nb_out, (const int32_t*)d_col_base_.p,
                               (double*)y, ldy, n, (int)y_cols);
        else
            fail(HG_ERR_ARG, "output dtype must be f32 or f64");
        if (ev) HG_HIP(hipEventRecord(ev[e++], st));
        HG_HIP(hipGetLastError());
    }

    void release() override {
        bufA_.free();
        bufB_.free();
        d_col_base_.free();
        for (auto& s : stages_) {
            s.d_nodes.free(); s.d_kb1.free(); s.d_kb2.free(); s.d_afrag.free(); s.d_bias.free();
            s.d_chunks.free(); s.d_runs.free(); s.d_koff.free(); s.d_kmean.free();
        }
        cap_rows_ = 0;
    }

private:
    // Stage 0: group consecutive nodes into chunks whose distinct input columns fit the LDS tile,
    // turn each chunk's column set into contiguous runs, and record for every node input position
    // its word offset inside the staged row.
    void plan_stage0_inputs(const FStage& st, HostStage& hs) {
        const int n = (int)st.nodes.size();
        s0_entry_base_.assign(n, 0);
        int max_cols = 0;
        int ni = 0;
        int entry = 0;
        while (ni < n) {
            std::vector<int32_t> cols;
            int n1 = ni;
            while (n1 < n) {
                std::vector<int32_t> c2 = cols;
                const FNode& nd = st.nodes[n1];
                for (int c = 0; c < nd.in_dim; ++c) c2.push_back(st.conn[nd.in_off + c]);
                std::sort(c2.begin(), c2.end());
                c2.erase(std::unique(c2.begin(), c2.end()), c2.end());
                if ((int)c2.size() > kStage0MaxCols) break;
                cols.swap(c2);
                ++n1;
                if (n1 - ni >= 64) break;
            }
            if (n1 == ni) fail(HG_ERR_FORMAT, "fused: first-layer node with more than %d inputs", kStage0MaxCols);
            DChunk ck{ni, n1 - ni, (int)hs.runs.size(), 0};
            std::map<int32_t, int32_t> lds_of;
            int off = 0;
            for (size_t i = 0; i < cols.size();) {
                size_t k = i + 1;
                while (k < cols.size() && cols[k] == cols[k - 1] + 1) ++k;
                hs.runs.push_back(DRun{cols[i], (int)(k - i), off, 0});
                for (size_t m = i; m < k; ++m) lds_of[cols[m]] = off + (int)(m - i);
                off += (int)(k - i);
                i = k;
                ++ck.run_count;
            }
            max_cols = std::max(max_cols, off);
            for (int k = ni; k < n1; ++k) {
                const FNode& nd = st.nodes[k];
                s0_entry_base_[k] = entry;
                const int nkb = (nd.in_dim + 15) / 16;
                for (int kb = 0; kb < nkb; ++kb, ++entry)
                    for (int g = 0; g < 4; ++g)
                        for (int r = 0; r < 4; ++r) {
                            int c = kb * 16 + 4 * r + g;
                            if (c < nd.in_dim) {
                                hs.koff.push_back(lds_of[st.conn[nd.in_off + c]]);
                                hs.kmean.push_back((float)nd.A1.a[c]);
                            } else {
                                hs.koff.push_back(-1);  // patched to the zero column below
                                hs.kmean.push_back(0.f);
                            }
                        }
            }
            hs.chunks.push_back(ck);
            ni = n1;
        }
        // row stride: >= max_cols + 1 (zero column), == 2 (mod 32) so the 16 sub-images x 2 lane
        // groups of one ds_read_b32 half-wave hit 32 distinct banks
        int stride = max_cols + 1;
        while (stride % 32 != 2) ++stride;
        hs.lds_stride = stride;
        for (auto& o : hs.koff)
            if (o < 0) o = stride - 1;
    }

    int in_dim_, out_dim_;
    std::vector<HostStage> stages_;
    std::vector<int32_t> s0_entry_base_;
    std::vector<int32_t> col_base_;
    DevBuf d_col_base_, bufA_, bufB_;
    int max_nb_ = 0;
    int64_t padded_flops_ = 0, cap_rows_ = 0;
};

}  // namespace

std::unique_ptr<Executor> make_fused_executor(const TNode& root, std::string* why_not) {
    std::vector<FStage> stages;
    std::string why;
    if (!build_stages(root, stages, why)) {
        if (why_not) *why_not = why;
        return nullptr;
    }
    for (size_t i = 0; i < stages.size(); ++i) {
        bool he = stages[i].nodes[0].has_exp;
        for (auto& n : stages[i].nodes)
            if (n.has_exp != he) {
                if (why_not) *why_not = "layer mixes node forms";
                return nullptr;
            }
    }
    if (why_not) why_not->clear();
    return std::make_unique<FusedExecutor>(root, std::move(stages));
}

}  // namespace hg
