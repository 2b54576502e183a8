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
#include <cstdlib>
#include <map>
#include <sstream>
#include <tuple>

#include "hg_fused_dev.hpp"

namespace hg {

namespace {

using namespace fused;

// Diagnostic switches (DESIGN.md "Diagnostic environment variables"), read ONCE when a plan is built — never on the
// execute path.  None changes results beyond rounding.
struct FusedOptions {
    bool no_rem4 = false, ig_nofold = false, ig_resident = false, debug = false, no_prefetch_all = false;
    int stamp_stage = -1, ig_w = 0, ig_t = 0;
    // batches of up to this many 16-row tiles run the top layers as one persistent launch (0: never).  Measured on U11L-128
    // (tools/small_batch2.py): 5-10 % of a call up to N = 128, a loss from N = 340 (one workgroup per node and slice cannot
    // match the per-layer kernels' throughput), so the default stops at 8 tiles.
    bool no_pack = false;         // HIGSFA_NO_PACK: remainder tiles as whole blocks
    uint32_t wq_start = 0;        // HIGSFA_WQ_START: initial value of the tile-queue counters (tests: wrap-around)
    bool no_wgq = false;          // HIGSFA_NO_WGQ: k_stage01d with one tile queue per layer-1 node instead of one per chunk (2-3 % faster, +29 % HBM bytes)
    bool no_direct = false;       // HIGSFA_NO_DIRECT: front kernel always stages the input rows through LDS (k_stage01p)
    int tail_max = 3;             // HIGSFA_TAIL: most layers k_tail fuses at the top of the hierarchy (0: off — per-layer launches + k_unpack)
    bool no_fspec = false;        // HIGSFA_NO_FSPEC: front kernel without the compile-time (identity, abs-power) expansion
    int subtree_max_tiles = 64;   // HIGSFA_SUBTREE: batches of up to this many 16-row tiles may run layers below the top as sub-trees (k_subtree); 0: never
    int subtree_max_wgs = 256;    // HIGSFA_SUBTREE_WGS: ... while sub-trees x tiles stays within this many workgroups
    int splitm_max_nodes = 4;     // experiments: HIGSFA_SPLITM_MAX
    int splitm_max_wgs = 512;     // HIGSFA_SPLITM_WGS: largest k_stage_splitm grid for layers of more than splitm_max_nodes nodes
    int shape_variant = 0;        // experiments: HIGSFA_SHAPES
    int stage_w = 0, stage_t = 0; // experiments: HIGSFA_STAGE_SHAPE=waves,tiles for every k_stage launch
    int stage_parts = 0;          // experiments: HIGSFA_STAGE_PARTS=tile parts of every k_stage launch
    int stage_only = -1;          // experiments: HIGSFA_STAGE_ONLY=<stage>: the two knobs above for that stage only
    static FusedOptions from_env() {
        FusedOptions o;
        o.no_rem4 = getenv("HIGSFA_NO_REM4") != nullptr;
        o.ig_nofold = getenv("HIGSFA_IG_NOFOLD") != nullptr;
        o.ig_resident = getenv("HIGSFA_IG_RESIDENT") != nullptr;
        o.debug = getenv("HIGSFA_DEBUG") != nullptr;
        o.no_prefetch_all = getenv("HIGSFA_NO_PREFETCH_ALL") != nullptr;
        if (const char* e = getenv("HIGSFA_STAMP")) o.stamp_stage = atoi(e);
        if (const char* e = getenv("HIGSFA_IG_SHAPE")) sscanf(e, "%d,%d", &o.ig_w, &o.ig_t);

        o.no_pack = getenv("HIGSFA_NO_PACK") != nullptr;
        o.no_fspec = getenv("HIGSFA_NO_FSPEC") != nullptr;
        o.no_direct = getenv("HIGSFA_NO_DIRECT") != nullptr;
        o.no_wgq = getenv("HIGSFA_NO_WGQ") != nullptr;
        if (const char* e = getenv("HIGSFA_WQ_START")) o.wq_start = (uint32_t)strtoul(e, nullptr, 0);
        if (const char* e = getenv("HIGSFA_SPLITM_MAX")) o.splitm_max_nodes = atoi(e);
        if (const char* e = getenv("HIGSFA_SPLITM_WGS")) o.splitm_max_wgs = atoi(e);
        if (const char* e = getenv("HIGSFA_SHAPES")) o.shape_variant = atoi(e);
        if (const char* e = getenv("HIGSFA_STAGE_SHAPE")) sscanf(e, "%d,%d", &o.stage_w, &o.stage_t);
        if (const char* e = getenv("HIGSFA_STAGE_PARTS")) o.stage_parts = atoi(e);
        if (const char* e = getenv("HIGSFA_STAGE_ONLY")) o.stage_only = atoi(e);
        if (const char* e = getenv("HIGSFA_SUBTREE")) o.subtree_max_tiles = std::max(0, atoi(e));
        if (const char* e = getenv("HIGSFA_SUBTREE_WGS")) o.subtree_max_wgs = std::max(0, atoi(e));
        if (const char* e = getenv("HIGSFA_TAIL")) o.tail_max = std::max(0, std::min(atoi(e), kMaxTail));
        return o;
    }
};

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
    bool has_prod = false;              // cross-column products in the expansion -> k_stage_prod
    bool has_clip = false;              // CutoffNode between expansion and second affine
    double clip_lo = 0, clip_hi = 0;
    // iGSFA node: x0 = x - mean; s = sfa(expand(x0)) (scale folded in);
    // r = x0 - lr(s); q = pca(r); y = [s, q]
    bool is_ig = false, ig_has_lr = false;
    int ig_k = 0;
    std::vector<double> ig_mean;
    Aff ig_sfa, ig_lr, ig_pca;
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
        case K_EXPANSION:
        case K_HEAD:
        case K_CUTOFF:
        case K_IGSFA: s.push_back(&n); return true;
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

// An iGSFA node is ONE affine map of its expanded input when the expansion contains the identity over all
// columns (e P = x0 for a selection matrix P):
//     s = e Ws + cs,           cs = bs - as Ws                       (scale already folded into Ws, bs)
//     l = e (Ws Wl) + cl,      cl = (cs - al) Wl + bl                (reconstruction, when present)
//     q = e (P - Ws Wl) Wp + (-cl - ap) Wp + bp
// so y = [s, q] = e [Ws | (P - Ws Wl) Wp] + const, folded here in float64.  Nodes of up to 64 inputs then run
// as ordinary nodes (first affine = x - mean, second = the folded map) on the kernels tuned for them —
// including the fused first-two-layers kernel, which also makes the k_im2frag pass unnecessary; the MFMA
// count is about the same (one GEMM over 2 d_in instead of three smaller ones).  Wider nodes stay on k_igsfa
// (their identity first affine would need more than kMaxMT tiles) but use the same folded map there: one GEMM
// from the expanded input fragments to all output tiles instead of the G1 -> G2 -> G3 chain.
// HIGSFA_IG_NOFOLD=1 keeps every iGSFA node on the three-GEMM form (tests).
bool igsfa_affine(const FNode& fn, Aff& A2, bool nofold) {
    const int d = fn.in_dim, k = fn.ig_k, q = fn.ig_pca.out;
    if (nofold) return false;
    int E = 0, id_off = -1;
    for (const ExpFunc& f : fn.funcs) {
        if (f.kind == E_IDENTITY && f.used(d) == d && id_off < 0) id_off = E;
        E += f.out_dim(d);
    }
    if (id_off < 0 || fn.ig_sfa.in != E || fn.ig_sfa.out != k) return false;
    const Aff &S = fn.ig_sfa, &Lr = fn.ig_lr, &Pc = fn.ig_pca;
    std::vector<double> cs(k), Wsl((size_t)E * d, 0.0), cl(d, 0.0);
    for (int j = 0; j < k; ++j) {
        double v = S.b[j];
        for (int e = 0; e < E; ++e) v -= S.a[e] * S.W[(size_t)e * k + j];
        cs[j] = v;
    }
    if (fn.ig_has_lr) {
        for (int e = 0; e < E; ++e)
            for (int j = 0; j < k; ++j) {
                const double w = S.W[(size_t)e * k + j];
                if (w == 0.0) continue;
                for (int c = 0; c < d; ++c) Wsl[(size_t)e * d + c] += w * Lr.W[(size_t)j * d + c];
            }
        for (int c = 0; c < d; ++c) {
            double v = Lr.b[c];
            for (int j = 0; j < k; ++j) v += (cs[j] - Lr.a[j]) * Lr.W[(size_t)j * d + c];
            cl[c] = v;
        }
    }
    A2 = Aff();
    A2.in = E;
    A2.out = k + q;
    A2.a.assign(E, 0.0);
    A2.W.assign((size_t)E * (k + q), 0.0);
    A2.b.assign(k + q, 0.0);
    for (int e = 0; e < E; ++e) {
        for (int j = 0; j < k; ++j) A2.W[(size_t)e * (k + q) + j] = S.W[(size_t)e * k + j];
        for (int c = 0; c < d; ++c) {
            const double m = ((e == id_off + c) ? 1.0 : 0.0) - Wsl[(size_t)e * d + c];
            if (m == 0.0) continue;
            for (int j = 0; j < q; ++j) A2.W[(size_t)e * (k + q) + k + j] += m * Pc.W[(size_t)c * q + j];
        }
    }
    for (int j = 0; j < k; ++j) A2.b[j] = cs[j];
    for (int j = 0; j < q; ++j) {
        double v = Pc.b[j];
        for (int c = 0; c < d; ++c) v += (-cl[c] - Pc.a[c]) * Pc.W[(size_t)c * q + j];
        A2.b[k + j] = v;
    }
    return true;
}

void fold_igsfa(FNode& fn, bool nofold) {
    const int d = fn.in_dim;
    Aff A2;
    if (d > 16 * kMaxMT || !igsfa_affine(fn, A2, nofold)) return;
    Aff A1;
    A1.in = A1.out = d;
    A1.a = fn.ig_mean;
    A1.a.resize(d, 0.0);
    A1.W.assign((size_t)d * d, 0.0);
    for (int c = 0; c < d; ++c) A1.W[(size_t)c * d + c] = 1.0;
    A1.b.assign(d, 0.0);
    fn.A1 = std::move(A1);
    fn.A2 = std::move(A2);
    fn.is_ig = false;
}

bool canon(const ChainT& c, int in_off, FNode& fn, std::string& why, const FusedOptions& opt) {
    fn.in_off = in_off;
    fn.in_dim = c.in_dim;
    fn.out_dim = c.out_dim;
    for (const TNode* l : c.seq)
        if (l->kind == K_IGSFA) {
            if (c.seq.size() != 1) { why = "iGSFA node combined with other nodes in one chain"; return false; }
            if (l->sfa->out_dim != l->aux) { why = "iGSFA node whose sfa_node has more outputs than it preserves"; return false; }
            if (l->in_dim > 128 || l->out_dim > 16 * kMaxMT) { why = "iGSFA node with more than 128 inputs or 64 outputs"; return false; }
            fn.is_ig = true;
            fn.ig_k = (int)l->aux;
            fn.ig_mean = l->x_mean;
            fn.ig_sfa = aff_of(*l->sfa);
            for (int r = 0; r < fn.ig_sfa.in; ++r)
                for (int cc = 0; cc < fn.ig_sfa.out; ++cc) fn.ig_sfa.W[(size_t)r * fn.ig_sfa.out + cc] *= l->magn[cc];
            {   // ((e - a) W + b) * magn = (e - a)(W magn) + b magn
                for (int cc = 0; cc < fn.ig_sfa.out; ++cc) fn.ig_sfa.b[cc] *= l->magn[cc];
            }
            fn.ig_has_lr = (bool)l->lr;
            if (l->lr) fn.ig_lr = aff_of(*l->lr);
            fn.ig_pca = aff_of(*l->pca);
            if (l->exp_node) {
                for (const ExpFunc& f : l->exp_node->funcs)
                    if (f.kind > E_SIGNED_POW) { why = "expansion with cross-column products (QT / pair products)"; return false; }
                fn.funcs = l->exp_node->funcs;
            } else {
                fn.funcs = {ExpFunc{E_IDENTITY, 0, 0, 1.0}};
            }
            fn.has_exp = true;
            fold_igsfa(fn, opt.ig_nofold);
            return true;
        }
    int phase = 0;  // 0: before A1, 1: in A1, 2: after E, 3: in A2
    auto head = [](Aff& a, int keep) {        // HeadNode after an affine: keep its first `keep` outputs
        Aff r;
        r.in = a.in;
        r.out = keep;
        r.a = a.a;
        r.W.resize((size_t)a.in * keep);
        for (int i = 0; i < a.in; ++i)
            for (int o = 0; o < keep; ++o) r.W[(size_t)i * keep + o] = a.W[(size_t)i * a.out + o];
        r.b.assign(a.b.begin(), a.b.begin() + keep);
        a = std::move(r);
    };
    for (const TNode* l : c.seq) {
        if (l->kind == K_AFFINE) {
            if (phase == 0) { fn.A1 = aff_of(*l); phase = 1; }
            else if (phase == 1) fn.A1 = fold(fn.A1, aff_of(*l));
            else if (phase == 2) { fn.A2 = aff_of(*l); phase = 3; }
            else fn.A2 = fold(fn.A2, aff_of(*l));
        } else if (l->kind == K_HEAD) {
            if (phase == 1) head(fn.A1, (int)l->out_dim);
            else if (phase == 3) head(fn.A2, (int)l->out_dim);
            else { why = "HeadNode that does not follow an affine node"; return false; }
        } else if (l->kind == K_CUTOFF) {
            if (phase != 2 || fn.has_clip) { why = "CutoffNode anywhere but between the expansion and the second affine"; return false; }
            fn.has_clip = true;
            fn.clip_lo = l->lo;
            fn.clip_hi = l->hi;
        } else {  // expansion
            if (phase != 1) { why = "node chain is not [affine][expansion][affine]"; return false; }
            for (const ExpFunc& f : l->funcs)
                if (f.kind > E_SIGNED_POW) fn.has_prod = true;
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

bool build_stages(const TNode& root, std::vector<FStage>& stages, std::string& why, const FusedOptions& opt) {
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
            if (!canon(c, off, fn, why, opt)) return false;
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
        if (c.kind == K_IDENTITY) continue;      // mdp IdentityNode between layers: nothing to execute
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
                    int n_exp = 0, n_ig = 0;
                    for (auto* l : group[k].seq) { n_exp += l->kind == K_EXPANSION; n_ig += l->kind == K_IGSFA; }
                    for (auto* l : chains[k].seq) { n_exp += l->kind == K_EXPANSION; n_ig += l->kind == K_IGSFA; }
                    if (n_exp > 1 || n_ig > 0) merged = false;
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

// Stages > 0.  A workgroup of NW waves owns NW*T batch tiles and walks a range of nodes; the
// weights of `nodes_per_group` nodes at a time are copied once into LDS and shared by all waves
// (A fragments by ds_read_b128); activation fragments come straight from HBM/L2 with one 16 B/lane
// coalesced load per K-block and tile, prefetched one K-block ahead.
// REM: the last tile of both affines holds <= 4 real rows and its A fragments are stored in 4x4 form
// (hg_fused_dev.hpp, "Remainder tiles").
// KBF > 0 (== P.kb1, checked at launch): all K-blocks of a node visit are fetched at once, one whole visit ahead — issued
// before the second half of the previous node, whose MFMAs (no global loads of their own) cover the L2 / HBM round trip.
// For small nodes (layer 2: four blocks of <= 4 k-steps) the one-block-ahead stream leaves each wave waiting ~1.5k cycles
// per block (in-kernel stamps: GEMM 1 took 6x its MFMA time); costs KBF * T * 4 registers.
#ifndef HG_A_PINGPONG
#define HG_A_PINGPONG 1      // (A/B switch: HIGSFA_CXXFLAGS=-DHG_A_PINGPONG=0)
#endif
#ifdef HIGSFA_DIAG
#define HG_HOT(blk) ((P.whatif & 1) ? ((blk) & 1) : (blk))      // timing experiment: all input blocks from the first two of the tile row
#else
#define HG_HOT(blk) (blk)
#endif
// PK >= 0 (only with KBF): K-block PK of every node is a slot-major packed block of the stage below (StageParams::pack_soa): its
// two slots are loaded with two 4-byte loads per lane at a position of the code that is fixed at compile time — a run-time test
// per K-block in the streaming loops below cost layers 3-7, which never see such a block, 5 % (profiles/r05_packed_stores.txt).
template <int MT1, int MT2, int T, bool STAMP = false, bool REM = false, int KBF = 0, bool FS = false, int PK = -1>
__global__ void __launch_bounds__(512, 4) k_stage(StageParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, nw = nthr >> 6, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps tile/row bookkeeping in SGPRs
    // XCD-aware decode: blocks b, b+8, ... share an XCD (and its L2); give each XCD whole node
    // chunks so a chunk's weights are fetched into one L2 only.
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    // part-major within an XCD: the first workgroups dispatched — the oldest on their CUs, which the SIMD arbiter favours — are
    // part 0 of EVERY chunk, so every node's tiles are served by fast and slow workgroups alike (chunk-major: +0.4 % per step)
    const int cpx = (P.n_chunks + 7) >> 3;
    // pair_chunks: chunks 2c and 2c + 1 hold the four siblings of one lane-major packed output block; on ONE XCD, in the same part,
    // their partial-line stores meet in that XCD's L2 before the line is written back (otherwise in two L2s: never)
    const int jq = kq % cpx;
    const int chunk = P.pair_chunks ? (((xcd + 8 * (jq >> 1)) << 1) | (jq & 1)) : xcd + 8 * jq, part = kq / cpx;
    if (chunk >= P.n_chunks) return;
    const int n_begin = chunk * P.nodes_per_wg;
    const int n_end = min(n_begin + P.nodes_per_wg, P.n_nodes);
    const int npg = P.nodes_per_group;
    float* sb = (float*)(smem + (size_t)npg * P.node_blocks * 64);
    int2* stab = (int2*)(sb + npg * P.bias_floats);
    // (Round 5: the lane-group sums of the 4x4-form tiles through LDS — rem4_total_lds, which takes 1.9 us off the front kernel — make THIS kernel
    // slower, layer 2 91.8 -> 105.5 us: its LDS queue is full of A-fragment reads, and the reduction's reads wait behind them and its lgkmcnt(0)
    // for all of them.  The permlane form stays here.)

    unsigned long long t_copy = 0, t_g1 = 0, t_tail = 0, t_all0 = 0;
    int n_it = 0;
    unsigned long long rt0 = 0;
    if (STAMP) {
        t_all0 = stamp_now();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int g0 = n_begin; g0 < n_end; g0 += npg) {
        const int gn = min(npg, n_end - g0);
        unsigned long long tc0 = 0;
        if (STAMP) tc0 = stamp_now();
        __syncthreads();
        {   // cooperative copy of the group's weights, 8 x 16 B in flight per thread (an L2 round trip is
            // ~2k cycles here: the bytes in flight per CU set the copy rate)
            const f32x4* src = P.afrag + (size_t)g0 * P.node_blocks * 64;
#ifdef HIGSFA_DIAG
            // timing experiment (HIGSFA_WHATIF bit 2; wrong results): no weight copy at all — what hiding it could gain at most
            const int nvec = (P.whatif & 4) ? 0 : gn * P.node_blocks * 64;
#else
            const int nvec = gn * P.node_blocks * 64;
#endif
            int i = tid;
            for (; i + 7 * nthr < nvec; i += 8 * nthr) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[i + u * nthr];
#pragma unroll
                for (int u = 0; u < 8; ++u) smem[i + u * nthr] = v[u];
            }
            for (; i < nvec; i += nthr) smem[i] = src[i];
            const float* bsrc = P.bias + (size_t)g0 * P.bias_floats;
            for (int k = tid; k < gn * P.bias_floats; k += nthr) sb[k] = bsrc[k];
            const int2* tsrc = P.kb1tab + (size_t)g0 * P.kb1;
            for (int k = tid; k < gn * P.kb1; k += nthr) stab[k] = tsrc[k];
        }
        __syncthreads();
        // The workgroup keeps this node group's weights in LDS and sweeps its share of the batch:
        // tile groups part, part + tile_parts, ...  (no barrier inside: waves run free).  The
        // activation fragments form ONE prefetch stream across K-blocks, nodes and tile groups:
        // while block i is multiplied, block i+1 — possibly the first block of the next node or of
        // the next tile group — is already in flight, so no visit starts with an exposed load.
        if (STAMP) t_copy += stamp_now() - tc0;
        int tile[T];
        uint32_t trow[T], trow_nx[T];   // first block of the tile's row in the input activation
#pragma unroll
        for (int t = 0; t < T; ++t) tile[t] = (part * nw + wave) * T + t;
        if (tile[0] >= P.n_tiles) continue;
#pragma unroll
        for (int t = 0; t < T; ++t) trow[t] = (uint32_t)(tile[t] < P.n_tiles ? tile[t] : tile[0]) * (uint32_t)P.nb_in;
        if constexpr (KBF > 0) {
            f32x4 bq[KBF][T];
#pragma unroll
            for (int kbi = 0; kbi < KBF; ++kbi) {
                const int2 e0 = stab[kbi];
                const int sb0 = HG_HOT(__builtin_amdgcn_readfirstlane(e0.x));
                if (kbi == PK) load_kblock_soa<T, true>(P, trow, sb0, __builtin_amdgcn_readfirstlane(e0.y), lane, bq[kbi]);
                else {
#pragma unroll
                    for (int t = 0; t < T; ++t) bq[kbi][t] = P.in[(size_t)(trow[t] + sb0) * 64 + lane];
                }
            }
            for (int grp = part; grp < P.tile_groups; grp += P.tile_parts) {
                const int tn0 = ((grp + P.tile_parts) * nw + wave) * T;
                const bool has_next = grp + P.tile_parts < P.tile_groups && tn0 < P.n_tiles;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int tn = tn0 + t;
                    trow_nx[t] = has_next ? (uint32_t)(tn < P.n_tiles ? tn : tn0) * (uint32_t)P.nb_in : trow[t];
                }
                for (int ln = 0; ln < gn; ++ln) {
                    const f32x4* wA1 = smem + (size_t)ln * P.node_blocks * 64 + lane;
                    const f32x4* wA2 = wA1 + KBF * MT1 * 64;
                    const float* b1 = sb + ln * P.bias_floats;
                    const int2* kt = stab + ln * KBF;
                    f32x4 z[MT1][T];
#pragma unroll
                    for (int mt = 0; mt < MT1; ++mt) {
                        f32x4 bb = *(const f32x4*)(b1 + mt * 16 + g * 4);
#pragma unroll
                        for (int t = 0; t < T; ++t) z[mt][t] = bb;
                    }
                    f32x4 d4[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) d4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    unsigned long long ts0 = 0, ts1 = 0;
                    if (STAMP) ts0 = stamp_now();
#pragma unroll
                    for (int kbi = 0; kbi < KBF; ++kbi) {
                        const int nkr = __builtin_amdgcn_readfirstlane(kt[kbi].y), nk = nkr & 255, r0 = (nkr >> 8) & 255;
                        if constexpr (REM) gemm_block_rem<MT1, T>(wA1 + kbi * MT1 * 64, bq[kbi], z, d4, nk, r0);
                        else gemm_block<MT1, T, FS>(wA1 + kbi * MT1 * 64, bq[kbi], z, nk, r0);
                    }
                    if constexpr (REM) {
#pragma unroll
                        for (int t = 0; t < T; ++t) add_rem4(z[MT1 - 1][t], d4[t]);
                    }
                    {   // next visit's blocks: next node of this group on the same tiles, or the group's first node on the next tiles
                        const bool in_group = ln + 1 < gn;
                        const int2* ktn = in_group ? kt + KBF : stab;
                        uint32_t rw[T];
#pragma unroll
                        for (int t = 0; t < T; ++t) rw[t] = in_group ? trow[t] : trow_nx[t];
#pragma unroll
                        for (int kbi = 0; kbi < KBF; ++kbi) {
                            const int2 en = ktn[kbi];
                            const int sbn = HG_HOT(__builtin_amdgcn_readfirstlane(en.x));
                            if (kbi == PK) load_kblock_soa<T, true>(P, rw, sbn, __builtin_amdgcn_readfirstlane(en.y), lane, bq[kbi]);
                            else {
#pragma unroll
                                for (int t = 0; t < T; ++t) bq[kbi][t] = P.in[(size_t)(rw[t] + sbn) * 64 + lane];
                            }
                        }
                    }
                    if (STAMP) ts1 = stamp_now();
                    node_tail<MT1, MT2, T, REM, FS>(P, wA2, b1 + MT1 * 16, g0 + ln, z, tile, lane);
                    if (STAMP) {
                        unsigned long long ts2 = stamp_now();
                        t_g1 += ts1 - ts0;
                        t_tail += ts2 - ts1;
                        ++n_it;
                    }
                }
                if (!has_next) break;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    tile[t] = tn0 + t;
                    trow[t] = trow_nx[t];
                }
            }
            continue;
        }
        f32x4 bf[T], bfn[T];
        int nk;
        {
            const int2 kb = stab[0];
            const int sb0 = HG_HOT(__builtin_amdgcn_readfirstlane(kb.x));
            nk = __builtin_amdgcn_readfirstlane(kb.y);
            if constexpr (REM) load_kblock<T>(P, trow, sb0, nk, lane, bf);
            else {
#pragma unroll
                for (int t = 0; t < T; ++t) bf[t] = P.in[(size_t)(trow[t] + sb0) * 64 + lane];
            }
        }
        for (int grp = part; grp < P.tile_groups; grp += P.tile_parts) {
            // rows of the tile group after this one (or this one again when it is the last)
            const int tn0 = ((grp + P.tile_parts) * nw + wave) * T;
            const bool has_next = grp + P.tile_parts < P.tile_groups && tn0 < P.n_tiles;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int tn = tn0 + t;
                trow_nx[t] = has_next ? (uint32_t)(tn < P.n_tiles ? tn : tn0) * (uint32_t)P.nb_in : trow[t];
            }
            for (int ln = 0; ln < gn; ++ln) {
                const f32x4* wA1 = smem + (size_t)ln * P.node_blocks * 64 + lane;
                const f32x4* wA2 = wA1 + P.kb1 * MT1 * 64;
                const float* b1 = sb + ln * P.bias_floats;
                const int2* kt = stab + ln * P.kb1;
                f32x4 z[MT1][T];
#pragma unroll
                for (int mt = 0; mt < MT1; ++mt) {
                    f32x4 bb = *(const f32x4*)(b1 + mt * 16 + g * 4);
#pragma unroll
                    for (int t = 0; t < T; ++t) z[mt][t] = bb;
                }
                f32x4 d4[T];      // REM: 4x4-form accumulators of the last z tile
#pragma unroll
                for (int t = 0; t < T; ++t) d4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                unsigned long long ts0 = 0, ts1 = 0;
                if (STAMP) ts0 = stamp_now();
                if constexpr (FS && !REM && HG_A_PINGPONG) {
                    // A fragments of block kbi + 1 are read from LDS BEFORE block kbi is multiplied, into the other of two register
                    // sets (the loop body twice, so that no set is ever copied): the MFMAs of a block wait for no LDS read
                    // (-0.6 % per step, profiles/r04_ab_lds_prefetch.txt).  Reading the K-block table entry two blocks ahead as
                    // well, so that the input prefetch does not wait for its LDS read either, measured 0.4 % SLOWER: not kept.
                    f32x4 a0[MT1], a1[MT1];
#pragma unroll
                    for (int mt = 0; mt < MT1; ++mt) a0[mt] = wA1[mt * 64];
                    auto kstep = [&](int kbi, const f32x4 (&ac)[MT1], f32x4 (&an)[MT1]) {
                        const bool in_node = kbi + 1 < P.kb1;
                        const bool in_group = in_node || ln + 1 < gn;
                        const int2 kbn = in_node ? kt[kbi + 1] : (ln + 1 < gn ? kt[P.kb1] : stab[0]);
                        const int sbn = HG_HOT(__builtin_amdgcn_readfirstlane(kbn.x));
                        const int nkn = __builtin_amdgcn_readfirstlane(kbn.y);
#pragma unroll
                        for (int t = 0; t < T; ++t) bfn[t] = P.in[(size_t)((in_group ? trow[t] : trow_nx[t]) + sbn) * 64 + lane];
                        if (in_node) {
#pragma unroll
                            for (int mt = 0; mt < MT1; ++mt) an[mt] = wA1[((kbi + 1) * MT1 + mt) * 64];
                        }
                        gemm_block_regs<MT1, T>(ac, bf, z, nk & 255, (nk >> 8) & 255);
#pragma unroll
                        for (int t = 0; t < T; ++t) bf[t] = bfn[t];
                        nk = nkn;
                    };
                    for (int kbi = 0; kbi < P.kb1; kbi += 2) {
                        kstep(kbi, a0, a1);
                        if (kbi + 1 < P.kb1) kstep(kbi + 1, a1, a0);
                    }
                } else
                for (int kbi = 0; kbi < P.kb1; ++kbi) {
                    const bool in_node = kbi + 1 < P.kb1;
                    const bool in_group = in_node || ln + 1 < gn;
                    const int2 kbn = in_node ? kt[kbi + 1] : (ln + 1 < gn ? kt[P.kb1] : stab[0]);
                    const int sbn = HG_HOT(__builtin_amdgcn_readfirstlane(kbn.x));
                    const int nkn = __builtin_amdgcn_readfirstlane(kbn.y);
                    if constexpr (REM) {      // (only remainder-tile stages can have slot-major packed input: plan_slot_major)
                        uint32_t rw[T];
#pragma unroll
                        for (int t = 0; t < T; ++t) rw[t] = in_group ? trow[t] : trow_nx[t];
                        load_kblock<T>(P, rw, sbn, nkn, lane, bfn);
                    } else {
#pragma unroll
                        for (int t = 0; t < T; ++t) bfn[t] = P.in[(size_t)((in_group ? trow[t] : trow_nx[t]) + sbn) * 64 + lane];
                    }
                    if constexpr (REM) gemm_block_rem<MT1, T>(wA1 + kbi * MT1 * 64, bf, z, d4, nk & 255, (nk >> 8) & 255);
                    else gemm_block<MT1, T, FS>(wA1 + kbi * MT1 * 64, bf, z, nk & 255, (nk >> 8) & 255);
#pragma unroll
                    for (int t = 0; t < T; ++t) bf[t] = bfn[t];
                    nk = nkn;
                }
                if constexpr (REM) {
#pragma unroll
                    for (int t = 0; t < T; ++t) add_rem4(z[MT1 - 1][t], d4[t]);
                }
                if (STAMP) ts1 = stamp_now();
                node_tail<MT1, MT2, T, REM, FS>(P, wA2, b1 + MT1 * 16, g0 + ln, z, tile, lane);
                if (STAMP) {
                    unsigned long long ts2 = stamp_now();
                    t_g1 += ts1 - ts0;
                    t_tail += ts2 - ts1;
                    ++n_it;
                }
            }
            if (!has_next) break;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                tile[t] = tn0 + t;
                trow[t] = trow_nx[t];
            }
        }
    }
    if (STAMP && lane == 0 && P.stamps) {
        unsigned long long* o = P.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        o[5] = rt1 - rt0;
        o[6] = rt0;
        o[7] = rt1;
        o[0] = t_copy;
        o[1] = t_g1;
        o[2] = t_tail;
        o[3] = stamp_now() - t_all0;
        o[4] = (unsigned long long)n_it;
    }
}

// Stages > 0 with FEW nodes (the top of the hierarchy: 16, 8, 4, 2, 1 nodes).  There are not enough
// (node, tile) pairs to fill 1024 SIMDs with whole nodes, and copying 64 KiB of weights into LDS
// per workgroup is all latency.  Here a workgroup of max(MT1, MT2) waves shares ONE node and T
// tiles; wave w computes m-tile w of GEMM 1 and later m-tile w of GEMM 2, so every weight block is
// read by exactly one wave (straight from L2, all loads of a phase issued up front), and the
// expanded z tiles are exchanged through 2*MT1*T KiB of LDS.
template <int T, bool REM = false>
__global__ void __launch_bounds__(256) k_stage_splitm(StageParams P, int mt1n, int mt2n) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    constexpr int KB = 8;   // K-blocks loaded per batch
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int node = blockIdx.x % P.n_nodes, grp = blockIdx.x / P.n_nodes;
    int tile[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tile[t] = grp * T + t;
    uint32_t trow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) trow[t] = (uint32_t)(tile[t] < P.n_tiles ? tile[t] : tile[0]) * (uint32_t)P.nb_in;
    const f32x4* wnode = P.afrag + (size_t)node * P.node_blocks * 64 + lane;
    const float* bnode = P.bias + (size_t)node * P.bias_floats + g * 4;
    const int2* kt = P.kb1tab + (size_t)node * P.kb1;
    const int nf = P.nf;
    // GEMM-2 weights of this wave's output tile do not depend on z: fetch them first
    f32x4 a2[KB];
    const int k2n = mt1n * nf;
    if (P.has_exp && w < mt2n) {
#pragma unroll
        for (int k = 0; k < KB; ++k)
            if (k < k2n) a2[k] = wnode[((size_t)P.kb1 * mt1n + (size_t)k * mt2n + w) * 64];
    }
    // Remainder tiles (P.a4x4: the stage's last tile of both affines is stored in 4x4 form, hg_fused_dev.hpp): the wave that owns
    // that tile multiplies with v_mfma_f32_4x4x1 into d4 and folds the four partial sums afterwards — the same products in the
    // same order as the k_stage REM instantiations
    const bool rem1 = REM && w == mt1n - 1, rem2 = REM && w == mt2n - 1;      // (REM instantiation <=> P.a4x4)
    f32x4 z[T], d4[T];
#pragma unroll
    for (int t = 0; t < T; ++t) d4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (w < mt1n) {
        const f32x4 bb = *(const f32x4*)(bnode + w * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) z[t] = bb;
        for (int k0 = 0; k0 < P.kb1; k0 += KB) {
            // the whole batch of K-block entries first (scalar loads that do not wait for one another; the table has 8 spare
            // entries behind the last node), then every weight / input block of the batch, then the MFMAs
            int2 ent[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) ent[k] = kt[k0 + k];
            f32x4 a1[KB], bf[KB][T];
            int nks[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const bool real = k0 + k < P.kb1;
                nks[k] = real ? ent[k].y : 0;
                const int sb = real ? ent[k].x : ent[0].x, wk = real ? k0 + k : k0;      // (blocks beyond the node's: a harmless re-read, no MFMA)
                a1[k] = wnode[((size_t)wk * mt1n + w) * 64];
                load_kblock<T>(P, trow, sb, real ? ent[k].y : ent[0].y, lane, bf[k]);
            }
            if (REM && rem1) {
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const int nk = nks[k] & 255, r0 = (nks[k] >> 8) & 255;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r >= r0 && r < nk) {
#pragma unroll
                            for (int t = 0; t < T; ++t) d4[t] = MFMA4(a1[k][r], bf[k][t][r], d4[t]);
                        }
                }
            } else {
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const int nk = nks[k] & 255, r0 = (nks[k] >> 8) & 255;      // (r0 > 0: a lane-major packed remainder block)
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
        }
        if (rem1) {
#pragma unroll
            for (int t = 0; t < T; ++t) add_rem4(z[t], d4[t]);
        }
        if (!P.has_exp) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + node * P.mto + w) * 64 + lane] = z[t];
            return;   // linear stage: no barrier follows on any path
        }
        for (int fi = 0; fi < nf; ++fi) {
            const int fk = (P.funcp >> (4 * fi)) & 15;
            const float ex = P.expo[fi];
#pragma unroll
            for (int t = 0; t < T; ++t) smem[((fi * mt1n + w) * T + t) * 64 + lane] = apply_func_uniform(fk, ex, z[t]);
        }
    } else if (!P.has_exp) {
        return;
    }
    __syncthreads();
    if (w >= mt2n) return;
#pragma unroll
    for (int t = 0; t < T; ++t) d4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 y[T];
    {
        const f32x4 bb = *(const f32x4*)(bnode + (mt1n + w) * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) y[t] = bb;
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        if (k >= k2n) continue;
        const int mt1 = k / nf, fi = k - mt1 * nf;
        const int nk = (P.nk2p[mt1] >> (4 * fi)) & 15;
        f32x4 e[T];
#pragma unroll
        for (int t = 0; t < T; ++t) e[t] = smem[((fi * mt1n + mt1) * T + t) * 64 + lane];
        if (REM && rem2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nk) {
#pragma unroll
                    for (int t = 0; t < T; ++t) d4[t] = MFMA4(a2[k][r], e[t][r], d4[t]);
                }
        } else if (nk == 4) {
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
    if (rem2) {
#pragma unroll
        for (int t = 0; t < T; ++t) add_rem4(y[t], d4[t]);
    }
    if (REM && P.pack_base > 0) {      // packed remainder tiles (StageParams::pack_base): full tiles as blocks, the remainder rows into the shared block
        if (rem2) {
            const int slot = __builtin_amdgcn_readfirstlane(P.pack_slot[node]);
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) *packed_slot_ptr(P, tile[t], slot, lane) = y[t][0];
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + node * (mt2n - 1) + w) * 64 + lane] = y[t];
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
        if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + node * P.mto + w) * 64 + lane] = y[t];
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
StageFn pick_stage(int mt1, int mt2, int T, bool rem = false, int kbf = 0, bool fs = false, int pk = -1) {
    if (pk >= 0) {      // slot-major packed input at K-block 1 of 3 (plan_slot_major admits exactly these shapes)
        if (pk == 1 && kbf == 3 && T == 2 && rem && mt1 == 3 && mt2 == 3) return fs ? (StageFn)k_stage<3, 3, 2, false, true, 3, true, 1> : (StageFn)k_stage<3, 3, 2, false, true, 3, false, 1>;
        if (pk == 1 && kbf == 3 && T == 2 && rem && mt1 == 2 && mt2 == 2) return (StageFn)k_stage<2, 2, 2, false, true, 3, false, 1>;
        return nullptr;
    }
    if (fs) {       // expansion (identity, |x|^p) at compile time: the shapes of the preset networks' middle layers
        if (kbf == 3 && T == 2 && rem && mt1 == 3 && mt2 == 3) return (StageFn)k_stage<3, 3, 2, false, true, 3, true>;
        if (kbf == 4 && T == 2 && !rem && mt1 == 3 && mt2 == 3) return (StageFn)k_stage<3, 3, 2, false, false, 4, true>;
        if (kbf == 0 && !rem && mt1 == 4 && mt2 == 4) return T == 2 ? (StageFn)k_stage<4, 4, 2, false, false, 0, true> : (StageFn)k_stage<4, 4, 1, false, false, 0, true>;
    }
    if (kbf == 4 && T == 2) {     // whole-visit prefetch (small nodes of four K-blocks; plan time checks kb1 == 4)
        if (rem && mt1 == 3 && mt2 == 3) return (StageFn)k_stage<3, 3, 2, false, true, 4>;
        if (rem && mt1 == 2 && mt2 == 2) return (StageFn)k_stage<2, 2, 2, false, true, 4>;
        if (!rem && mt1 == 3 && mt2 == 3) return (StageFn)k_stage<3, 3, 2, false, false, 4>;
        if (!rem && mt1 == 2 && mt2 == 2) return (StageFn)k_stage<2, 2, 2, false, false, 4>;
    }
    if (kbf == 3 && T == 2) {     // the same with the children's remainder tiles packed into one shared block
        if (rem && mt1 == 3 && mt2 == 3) return (StageFn)k_stage<3, 3, 2, false, true, 3>;
        if (rem && mt1 == 2 && mt2 == 2) return (StageFn)k_stage<2, 2, 2, false, true, 3>;
    }
    if (rem) {      // instantiated where the synthetic and test networks need it (plan time checks the same list)
        if (mt1 == 3 && mt2 == 3) return T == 2 ? (StageFn)k_stage<3, 3, 2, false, true> : (StageFn)k_stage<3, 3, 1, false, true>;
        if (mt1 == 2 && mt2 == 2) return T == 2 ? (StageFn)k_stage<2, 2, 2, false, true> : (StageFn)k_stage<2, 2, 1, false, true>;
    }
    switch (mt1) {
        case 1: return pick_stage_m2<1>(mt2, T);
        case 2: return pick_stage_m2<2>(mt2, T);
        case 3: return pick_stage_m2<3>(mt2, T);
        default: return pick_stage_m2<4>(mt2, T);
    }
}


// ---- the executor --------------------------------------------------------------------------------
inline int q_of_row(int i) { return 4 * (i & 3) + (i >> 2); }  // tile row -> tile-local feature (involution)

constexpr int kStage0ChunkCols = 128;   // columns of one sub-image staged per chunk (T = 4 tiles -> ~66 KiB LDS)
constexpr int kWeightLdsKiB = 64;       // target size of a node group's weights in LDS

// layers [begin, begin + len) as n independent sub-trees in one launch (k_subtree, short batches; plan_subtree)
// Per layer of the run: [sub-tree][position] -> node of the layer, and (layers above the run's first) the K-block table with source blocks
// renumbered to the sub-tree's own activation buffer in LDS (position of the source node in the layer below x mto + tile).
// set: runs are planned in two alternative sets (plan_subtree); a call uses the runs of ONE set.
struct SubRun {
    int begin = 0, len = 0, n = 0, act_blocks = 0, e_blocks = 0, set = 0;
    std::vector<int32_t> nodes[kMaxTail], tab[kMaxTail];
    DevBuf d_nodes[kMaxTail], d_tab[kMaxTail];
};

struct HostStage {
    int mt1 = 1, mt2 = 1, mto = 1, nb_out = 0, nb_in = 0, n_nodes = 0, kb1 = 0, nf = 0;
    int node_blocks = 0, bias_floats = 0, nk_last = 4;
    int p_max = 0, s_max = 0;   // widest first / second affine of the layer (real outputs)
    bool rem4 = false;          // last tiles of both affines in 4x4 form (k_stage REM instantiations)
    bool pack_out = false;      // output: the remainder tiles of four sibling nodes share one block (StageParams::pack_base)
    bool pack_soa = false;      // ... stored slot-major (StageParams::pack_soa; plan_slot_major)
    int pack_in = 0x7fffffff;   // input: source blocks from this one on are slot-major packed blocks of the stage below (StageParams::pack_in)
    int pk_kbi = -1;            // ... and sit at this position of every node's K-block list
    std::vector<int32_t> pack_slot;
    DevBuf d_pack_slot;
    bool has_exp = false, contig4 = false, vec_ok = false;
    std::vector<ExpFunc> funcs;
    uint8_t nk2[kMaxMT][kMaxFuncs] = {};
    std::vector<float> afrag, bias;
    std::vector<int32_t> kb1tab;  // int2 pairs
    // stage 0
    std::vector<DChunk> chunks;
    std::vector<DRun> runs;
    std::vector<int32_t> piece_col, koff;
    std::vector<float> kmean;
    std::vector<int32_t> kcol;     // k_stage01d: [node][g] first source column of the lane group's four (empty: not applicable)
    bool direct_ok = false;
    int lds_stride = 0, max_chunk_nodes = 0, max_chunk_pieces = 0;
    int kind = 0;            // 0: affine-expansion-affine layer, 1: row-major -> fragment gather, 2: iGSFA layer, 3: table-driven expansion
    int neb = 0;             // kind 3: K-blocks of the expanded input
    bool has_clip = false;
    float clip_lo = 0, clip_hi = 0;
    std::vector<int32_t> etab;
    DevBuf d_etab;
    bool from_x = false;     // reads the caller's row-major matrix
    bool ig_has_lr = false, ig_folded = false;
    int ig_nks[kMaxMT] = {};  // k-steps of each slow-feature tile
    std::vector<int32_t> gcol;
    DevBuf d_gcol;
    int64_t mfma_per_tile = 0;
    int64_t mfma16_tile = 0, mfma4_tile = 0;      // instructions issued per batch tile, all nodes of the layer
    int64_t ks1_tile = 0, ks2_tile = 0;      // k-steps of the first / second affine summed over the layer's nodes (per batch tile): issue accounting
    std::string name;
    DevBuf d_afrag, d_bias, d_kb1tab, d_chunks, d_runs, d_piece, d_koff, d_kmean, d_kcol;
};

class FusedExecutor : public Executor {
public:
    FusedExecutor(const TNode& root, std::vector<FStage>&& fs, const FusedOptions& opt) : opt_(opt), out_dim_(root.out_dim) {
        std::vector<int32_t> prev_blk, prev_q;  // per column of the previous stage's output frame
        int prev_nb = 0;
        bool prev_packed = false;                // previous stage stores packed remainder tiles (its consumer decodes r0)
        for (size_t si = 0; si < fs.size(); ++si) {
            FStage& st = fs[si];
            if (st.nodes[0].is_ig) {
                if (stages_.empty()) add_gather0(st, prev_blk, prev_q, prev_nb);
                build_ig_stage(st, prev_blk, prev_q, prev_nb);
                prev_packed = false;
                continue;
            }
            {
                bool table_driven = st.nodes[0].has_clip;
                for (auto& nd : st.nodes) table_driven = table_driven || nd.has_prod;
                if (table_driven) {
                    if (stages_.empty()) add_gather0(st, prev_blk, prev_q, prev_nb);
                    build_prod_stage(st, prev_blk, prev_q, prev_nb, (int)si);
                    prev_packed = false;
                    continue;
                }
            }
            if (si > 0 && stages_.empty()) fail(HG_ERR_FORMAT, "internal: stage order");
            stages_.emplace_back();
            HostStage& hs = stages_.back();
            hs.from_x = si == 0;
            const int n = (int)st.nodes.size();
            hs.n_nodes = n;
            hs.has_exp = st.nodes[0].has_exp;
            hs.funcs = st.nodes[0].funcs;
            hs.nf = (int)hs.funcs.size();
            int p_max = 0;
            for (auto& nd : st.nodes) {
                hs.mt1 = std::max(hs.mt1, (nd.A1.out + 15) / 16);
                if (nd.has_exp) hs.mt2 = std::max(hs.mt2, (nd.A2.out + 15) / 16);
                p_max = std::max(p_max, nd.A1.out);
                if (nd.has_exp) hs.s_max = std::max(hs.s_max, nd.A2.out);
            }
            hs.p_max = p_max;
            if (!hs.has_exp) hs.mt2 = 1;
            hs.mto = hs.has_exp ? hs.mt2 : hs.mt1;
            hs.nb_in = prev_nb;
            for (int mt1 = 0; mt1 < hs.mt1; ++mt1)
                for (int fi = 0; fi < hs.nf; ++fi) {
                    int valid = std::max(0, std::min(16, hs.funcs[fi].used(p_max) - 16 * mt1));
                    hs.nk2[mt1][fi] = (uint8_t)((valid + 3) / 4);
                }

            // ---- per node: K-blocks of GEMM 1 and, for every (kblock, q), the consumer input positions
            struct NodeK {
                std::vector<int> src, nk;
                std::vector<std::vector<int>> kpos;  // [kb*16 + q] -> positions c
            };
            std::vector<NodeK> nks(n);
            if (si == 0) plan_stage0_inputs(st, hs);
            for (int ni = 0; ni < n; ++ni) {
                FNode& nd = st.nodes[ni];
                NodeK& K = nks[ni];
                if (si == 0) {
                    const int nkb = (nd.in_dim + 15) / 16;
                    for (int kb = 0; kb < nkb; ++kb) {
                        // K slot (k-step r, lane group g) <- input position s0_pos(r, g) of this block
                        int valid = std::min(16, nd.in_dim - kb * 16), nk = 0;
                        K.src.push_back(0);
                        for (int q = 0; q < 16; ++q) {
                            K.kpos.emplace_back();
                            const int pos = s0_pos(q >> 2, q & 3);
                            if (pos < valid) {
                                K.kpos.back().push_back(kb * 16 + pos);
                                nk = std::max(nk, (q >> 2) + 1);
                            }
                        }
                        K.nk.push_back(nk);
                    }
                } else {
                    std::map<int, int> blk_index;
                    for (int c = 0; c < nd.in_dim; ++c) {
                        int pc = st.conn[nd.in_off + c];
                        int blk = prev_blk[pc], q = prev_q[pc];
                        auto it = blk_index.find(blk);
                        int kb;
                        if (it == blk_index.end()) {
                            kb = (int)K.src.size();
                            blk_index[blk] = kb;
                            K.src.push_back(blk);
                            K.nk.push_back(0);
                            for (int qq = 0; qq < 16; ++qq) K.kpos.emplace_back();
                        } else {
                            kb = it->second;
                        }
                        K.kpos[kb * 16 + q].push_back(c);
                        K.nk[kb] = std::max(K.nk[kb], q / 4 + 1);
                    }
                }
                hs.kb1 = std::max(hs.kb1, (int)K.src.size());
            }
            hs.node_blocks = hs.kb1 * hs.mt1 + (hs.has_exp ? hs.mt1 * hs.nf * hs.mt2 : 0);
            hs.bias_floats = (hs.mt1 + (hs.has_exp ? hs.mt2 : 0)) * 16;
            if (si > 0 && (size_t)hs.node_blocks * 1024 + (size_t)hs.bias_floats * 4 + (size_t)hs.kb1 * 8 > 150 * 1024)
                fail(HG_ERR_FORMAT, "fused: one node needs %d KiB of weight fragments, more than a workgroup's LDS", hs.node_blocks);
            hs.afrag.assign((size_t)n * hs.node_blocks * 256, 0.f);
            hs.bias.assign((size_t)n * hs.bias_floats, 0.f);
            if (si > 0) hs.kb1tab.assign((size_t)n * hs.kb1 * 2, 0);

            std::vector<int32_t> cur_blk, cur_q, node_out;
            for (int ni = 0; ni < n; ++ni) {
                FNode& nd = st.nodes[ni];
                NodeK& K = nks[ni];
                const int p = nd.A1.out;
                float* wnode = hs.afrag.data() + (size_t)ni * hs.node_blocks * 256;
                float* bnode = hs.bias.data() + (size_t)ni * hs.bias_floats;
                // bias 1: (x - a) W + b = x W + (b - a W); stage 0 subtracts fl32(a) in the loader and
                // keeps only the fp64 remainder here
                std::vector<double> bias1 = nd.A1.b;
                for (int c = 0; c < nd.in_dim; ++c) {
                    double av = si == 0 ? nd.A1.a[c] - (double)(float)nd.A1.a[c] : nd.A1.a[c];
                    if (av == 0.0) continue;
                    for (int o = 0; o < p; ++o) bias1[o] -= av * nd.A1.W[(size_t)c * p + o];
                }
                for (size_t kb = 0; kb < K.src.size(); ++kb) {
                    int r0 = 0;      // leading k-steps of a packed block that belong to other nodes' rows: skipped (k_stage only)
                    if (si > 0 && prev_packed) {
                        r0 = 4;
                        for (int q = 0; q < 16; ++q)
                            if (!K.kpos[kb * 16 + q].empty()) r0 = std::min(r0, q / 4);
                        if (r0 >= K.nk[kb]) r0 = 0;
                    }
                    if (si > 0) {
                        hs.kb1tab[((size_t)ni * hs.kb1 + kb) * 2] = K.src[kb];
                        hs.kb1tab[((size_t)ni * hs.kb1 + kb) * 2 + 1] = K.nk[kb] | (r0 << 8);
                    }
                    hs.mfma_per_tile += (int64_t)(K.nk[kb] - r0) * hs.mt1;
                    hs.ks1_tile += K.nk[kb] - r0;
                    for (int mt = 0; mt < hs.mt1; ++mt) {
                        float* blk = wnode + ((size_t)kb * hs.mt1 + mt) * 256;
                        for (int lane = 0; lane < 64; ++lane) {
                            int i = lane & 15, gg = lane >> 4;
                            int fo = 16 * mt + q_of_row(i);
                            if (fo >= p) continue;
                            for (int r = 0; r < 4; ++r) {
                                double w = 0;
                                for (int c : K.kpos[kb * 16 + 4 * r + gg]) w += nd.A1.W[(size_t)c * p + fo];
                                blk[lane * 4 + r] = (float)w;
                            }
                        }
                    }
                }
                if (si > 0)  // padded K-blocks: any valid source block, zero k-steps
                    for (int kb = (int)K.src.size(); kb < hs.kb1; ++kb) hs.kb1tab[((size_t)ni * hs.kb1 + kb) * 2] = K.src[0];
                if (si == 0 && (int)K.src.size() == hs.kb1) hs.nk_last = std::max(ni == 0 ? 0 : hs.nk_last, K.nk.back());
                for (int mt = 0; mt < hs.mt1; ++mt)
                    for (int gg = 0; gg < 4; ++gg)
                        for (int r = 0; r < 4; ++r) {
                            int fo = 16 * mt + 4 * r + gg;
                            bnode[mt * 16 + gg * 4 + r] = fo < p ? (float)bias1[fo] : 0.f;
                        }
                int n_out = p;
                if (hs.has_exp) {
                    const int s = nd.A2.out;
                    n_out = s;
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
                    float* w2 = wnode + (size_t)hs.kb1 * hs.mt1 * 256;
                    for (int mt1 = 0; mt1 < hs.mt1; ++mt1)
                        for (int fi = 0; fi < hs.nf; ++fi) {
                            const int used = nd.funcs[fi].used(p);
                            hs.mfma_per_tile += (int64_t)hs.nk2[mt1][fi] * hs.mt2;
                            hs.ks2_tile += hs.nk2[mt1][fi];
                            for (int mt2 = 0; mt2 < hs.mt2; ++mt2) {
                                float* blk = w2 + ((size_t)(mt1 * hs.nf + fi) * hs.mt2 + mt2) * 256;
                                for (int lane = 0; lane < 64; ++lane) {
                                    int i = lane & 15, gg = lane >> 4;
                                    int fo = 16 * mt2 + q_of_row(i);
                                    if (fo >= s) continue;
                                    for (int r = 0; r < 4; ++r) {
                                        int fz = 16 * mt1 + 4 * r + gg;
                                        if (fz >= used) continue;
                                        blk[lane * 4 + r] = (float)nd.A2.W[(size_t)(foff[fi] + fz) * s + fo];
                                    }
                                }
                            }
                        }
                    for (int mt = 0; mt < hs.mt2; ++mt)
                        for (int gg = 0; gg < 4; ++gg)
                            for (int r = 0; r < 4; ++r) {
                                int fo = 16 * mt + 4 * r + gg;
                                bnode[hs.mt1 * 16 + mt * 16 + gg * 4 + r] = fo < s ? (float)bias2[fo] : 0.f;
                            }
                }
                node_out.push_back(n_out);
            }
            // Remainder tiles (hg_fused_dev.hpp): when the last tile of BOTH affines holds 1..4 real rows, store its
            // A fragments in 4x4 form for the k_stage REM instantiations.  Only stages that always run on
            // k_stage: not the first two (front kernels read the ordinary form) and more than 4 nodes
            // (k_stage_splitm takes the small ones).
            {
                const int r1 = hs.p_max - 16 * (hs.mt1 - 1), r2 = hs.s_max - 16 * (hs.mt2 - 1);
                hs.rem4 = si >= 1 && hs.has_exp && n > 4 && hs.mt1 == hs.mt2 && (hs.mt1 == 2 || hs.mt1 == 3) && r1 >= 1 && r1 <= 4 &&
                          r2 >= 1 && r2 <= 4 && !opt_.no_rem4;
                if (hs.rem4) {
                    auto to4x4 = [](float* blk) {
                        float old[256];
                        std::copy(blk, blk + 256, old);
                        for (int l = 0; l < 64; ++l) {
                            const int src = (l & 48) | ((l & 3) << 2);
                            for (int r = 0; r < 4; ++r) blk[l * 4 + r] = old[src * 4 + r];
                        }
                    };
                    for (int ni = 0; ni < n; ++ni) {
                        float* wnode = hs.afrag.data() + (size_t)ni * hs.node_blocks * 256;
                        for (int kb = 0; kb < hs.kb1; ++kb) to4x4(wnode + ((size_t)kb * hs.mt1 + hs.mt1 - 1) * 256);
                        float* w2 = wnode + (size_t)hs.kb1 * hs.mt1 * 256;
                        for (int b2 = 0; b2 < hs.mt1 * hs.nf; ++b2) to4x4(w2 + ((size_t)b2 * hs.mt2 + hs.mt2 - 1) * 256);
                    }
                    // the 4x4 tiles cost a quarter of the MFMA time of a 16x16 tile
                }
            }
            // Packed remainder tiles: only where this stage always runs on a kernel that writes them (k_stage REM, or the fused
            // front kernel for stage 1) and the next stage always runs on one that decodes them (k_stage: an ordinary layer of
            // more than 16 nodes — not the split-m, chain, product or iGSFA kernels).
            hs.pack_out = false;
            if (hs.rem4 && n % 4 == 0 && !opt_.no_pack && si + 1 < fs.size()) {
                const FStage& nx = fs[si + 1];
                bool ok = nx.nodes.size() > 16;
                for (auto& nd : nx.nodes) ok = ok && !nd.is_ig && !nd.has_prod && !nd.has_clip && nd.has_exp;
                hs.pack_out = ok;
            }
            if (hs.pack_out) {
                // which four nodes share a block: order the nodes by the node of the next layer that reads their remainder rows
                // (first reader), so that the children of one parent — and of its neighbour — sit in one block: the parent then
                // reads them as ONE K-block.  Any grouping is correct; this one saves loads.
                const FStage& nx = fs[si + 1];
                std::vector<int> col0(n, 0), reader(n, 1 << 30);
                for (int ni = 1; ni < n; ++ni) col0[ni] = col0[ni - 1] + node_out[ni - 1];
                std::vector<int> owner;           // output column -> node
                for (int ni = 0; ni < n; ++ni) owner.insert(owner.end(), node_out[ni], ni);
                for (size_t pj = 0; pj < nx.nodes.size(); ++pj)
                    for (int c = 0; c < nx.nodes[pj].in_dim; ++c) {
                        const int col = nx.conn[nx.nodes[pj].in_off + c];
                        const int ni = owner[col];
                        if (col - col0[ni] >= 16 * (hs.mto - 1)) reader[ni] = std::min(reader[ni], (int)pj);
                    }
                std::vector<int> order(n);
                for (int ni = 0; ni < n; ++ni) order[ni] = ni;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return reader[a] < reader[b]; });
                hs.pack_slot.assign(n, 0);
                for (int rk = 0; rk < n; ++rk) hs.pack_slot[order[rk]] = rk;
            }
            for (int ni = 0; ni < n; ++ni)
                for (int f = 0; f < node_out[ni]; ++f) {
                    if (hs.pack_out && f >= 16 * (hs.mto - 1)) {
                        cur_blk.push_back(n * (hs.mto - 1) + hs.pack_slot[ni] / 4);
                        cur_q.push_back(4 * (hs.pack_slot[ni] % 4) + (f - 16 * (hs.mto - 1)));
                    } else {
                        cur_blk.push_back(ni * (hs.pack_out ? hs.mto - 1 : hs.mto) + f / 16);
                        cur_q.push_back(f % 16);
                    }
                }
            hs.nb_out = hs.pack_out ? n * (hs.mto - 1) + n / 4 : n * hs.mto;
            prev_packed = hs.pack_out;
            prev_blk.swap(cur_blk);
            prev_q.swap(cur_q);
            prev_nb = hs.nb_out;
            max_nb_ = std::max(max_nb_, hs.nb_out);
            {
                // What the kernels ISSUE per 16-row tile: an affine's tile of <= 4 real rows runs on v_mfma_f32_4x4x1 (512 FLOP, 8 cycles)
                // where the stage has remainder tiles — decided above for k_stage, at run time with the same rule for layer 1 inside the
                // front kernel — every other tile on v_mfma_f32_16x16x4 (2048 FLOP, 32 cycles).  Round 4 counted every tile as 16 x 16.
                const bool front_rem = si == 1 && hs.has_exp && hs.mt1 == 2 && hs.mt2 == 2 && hs.p_max <= 20 && hs.s_max <= 20 && hs.nk2[1][0] <= 1 &&
                                       hs.nk2[1][1] <= 1 && !opt_.no_rem4;
                const int rem = (hs.rem4 || front_rem) ? 1 : 0;
                hs.mfma16_tile = hs.ks1_tile * (hs.mt1 - rem) + (hs.has_exp ? hs.ks2_tile * (hs.mt2 - rem) : 0);
                hs.mfma4_tile = rem ? hs.ks1_tile + hs.ks2_tile : 0;
            }
            padded_flops_ += (hs.mfma16_tile * 2048 + hs.mfma4_tile * 512) / 16;
            std::ostringstream os;
            os << "fused stage " << si << (hs.rem4 ? (hs.pack_out ? " (4x4 remainder tiles, packed four to a block)" : " (4x4 remainder tiles)") : "") << ": " << hs.n_nodes << " nodes, K-blocks " << hs.kb1 << ", tiles " << hs.mt1 << "x" << hs.mt2
               << ", " << hs.mfma_per_tile << " MFMA/tile (issued: " << hs.mfma16_tile << " x 16x16x4 + " << hs.mfma4_tile << " x 4x4x1), " << hs.afrag.size() * 4 / 1024
               << " KiB weights, out " << hs.nb_out << " blocks/tile";
            hs.name = os.str();
        }
        plan_slot_major();
        fuse01_ = can_fuse01();
        if (fuse01_) {
            stages_[0].name += "  [+ stage 1 fused in the same persistent kernel when the input allows 16-byte loads]";
        }
        col_base_.resize(out_dim_);
        col_of_.assign((size_t)std::max(prev_nb, 1) * 16, -1);      // inverse map for k_tail: (output block, feature of the tile) -> caller column
        for (int c = 0; c < out_dim_; ++c) {
            int q = prev_q[c];
            col_base_[c] = prev_blk[c] * 256 + (q & 3) * 64 + (q >> 2);
            col_of_[(size_t)prev_blk[c] * 16 + q] = c;
        }
        for (auto& hs : stages_)      // k_tail reads 8 K-block entries at once from a node's first: 8 spare ones behind the last node's
            if (!hs.kb1tab.empty()) hs.kb1tab.resize(hs.kb1tab.size() + 16, 0);
        plan_tail();
        plan_subtree();
    }

    int plan_kind() const override { return HG_PLAN_FUSED; }
    int n_stages() const override { return (int)stages_.size() + 1; }
    std::string stage_name(int i) const override {
        if (i < (int)stages_.size()) return stages_[i].name;
        return tail_begin_ >= 0 ? std::string("row-major y written by the top-of-hierarchy launch (no unpack pass)")
                                : std::string("fused unpack (fragment order -> row-major y)");
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
            s.d_afrag.upload(s.afrag.data(), s.afrag.size() * 4);
            s.d_bias.upload(s.bias.data(), s.bias.size() * 4);
            if (!s.kb1tab.empty()) s.d_kb1tab.upload(s.kb1tab.data(), s.kb1tab.size() * 4);
            if (!s.gcol.empty()) s.d_gcol.upload(s.gcol.data(), s.gcol.size() * 4);
            if (!s.etab.empty()) s.d_etab.upload(s.etab.data(), s.etab.size() * 4);
            if (!s.pack_slot.empty()) s.d_pack_slot.upload(s.pack_slot.data(), s.pack_slot.size() * 4);
            if (!s.chunks.empty()) {
                s.d_chunks.upload(s.chunks.data(), s.chunks.size() * sizeof(DChunk));
                s.d_runs.upload(s.runs.data(), s.runs.size() * sizeof(DRun));
                s.piece_col.resize(std::max<size_t>(s.piece_col.size(), 2));
                s.d_piece.upload(s.piece_col.data(), s.piece_col.size() * 4);
                s.d_koff.upload(s.koff.data(), s.koff.size() * 4);
                s.d_kmean.upload(s.kmean.data(), s.kmean.size() * 4);
                if (!s.kcol.empty()) s.d_kcol.upload(s.kcol.data(), s.kcol.size() * 4);
            }
        }
        for (SubRun& r : sub_runs_)
            for (int k = 0; k < r.len; ++k) {
                r.d_nodes[k].upload(r.nodes[k].data(), r.nodes[k].size() * 4);
                if (k > 0) r.d_tab[k].upload(r.tab[k].data(), r.tab[k].size() * 4);
            }
        d_col_base_.upload(col_base_.data(), col_base_.size() * 4);
        d_col_of_.upload(col_of_.data(), col_of_.size() * 4);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cus_ = prop.multiProcessorCount;
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
        check_device_error();      // of an earlier call: the word is host memory the kernels write through
        if (n > cap_rows_) reserve(n);
        // (Cutting a batch into row ranges on separate streams, plain or staggered, was measured and dropped: 0.69-0.84 ms
        // against 0.65 at N = 4096 — DESIGN.md §6.1.)
        run_range(x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy, st, ev, (f32x4*)bufA_.p, (f32x4*)bufB_.p);
    }

    void run_range(const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols, int64_t ldy,
                   hipStream_t st, hipEvent_t* ev, f32x4* bufA, f32x4* bufB) {
        const int n_tiles = (int)((n + 15) / 16);
        const int sub_set = pick_sub_set(n_tiles);
        int e = 0;
        if (ev) HG_HIP(hipEventRecord(ev[e++], st));
        f32x4* cur = bufA;
        f32x4* nxt = bufB;
        for (size_t si = 0; si < stages_.size(); ++si) {
            HostStage& s = stages_[si];
            auto base_params = [&](HostStage& hs, const f32x4* in, f32x4* out) {
                StageParams R{};
                R.afrag = (const f32x4*)hs.d_afrag.p;
                R.bias = (const float*)hs.d_bias.p;
                R.kb1tab = (const int2*)hs.d_kb1tab.p;
                R.in = in;
                R.out = out;
                R.n_nodes = hs.n_nodes;
                R.kb1 = hs.kb1;
                R.nf = hs.nf;
                R.has_exp = hs.has_exp ? 1 : 0;
                R.node_blocks = hs.node_blocks;
                R.bias_floats = hs.bias_floats;
                R.n_tiles = n_tiles;
                R.nb_in = hs.nb_in;
                R.nb_out = hs.nb_out;
                R.mto = hs.mto;
                R.pack_base = hs.pack_out ? hs.n_nodes * (hs.mto - 1) : 0;
                R.pack_soa = hs.pack_soa ? 1 : 0;
                R.pack_in = hs.pack_in;
                R.pack_slot = (const int32_t*)hs.d_pack_slot.p;
                R.a4x4 = hs.rem4 ? 1 : 0;
                for (int fi = 0; fi < hs.nf; ++fi) {
                    R.funcp |= (uint32_t)hs.funcs[fi].kind << (4 * fi);
                    R.expo[fi] = (float)hs.funcs[fi].expo;
                    for (int mt1 = 0; mt1 < hs.mt1; ++mt1) R.nk2p[mt1] |= (uint32_t)hs.nk2[mt1][fi] << (4 * fi);
                }
                return R;
            };
            if (tail_begin_ >= 0 && (int)si == tail_start(n_tiles)) {
                // the top of the hierarchy as one launch that ends in the caller's rows (hg_fused_tail.hip)
                TailParams TP = tail_params((int)si, cur, n_tiles, y, y_dtype, y_cols, ldy, n);
                // two tiles per workgroup only for narrow tops on long batches, and only while their doubled LDS image fits
                const int T = (n_tiles >= 512 && tail_waves(TP) <= 8 && tail_lds_bytes(TP, 2) <= (size_t)160 * 1024) ? 2 : 1;
#ifdef HIGSFA_DIAG
                const bool stamped = opt_.stamp_stage == (int)si;
                if (stamped) tail_stamps_begin(TP, (size_t)((n_tiles + T - 1) / T), st);
#endif
                launch_tail(TP, T, st);
#ifdef HIGSFA_DIAG
                if (stamped) tail_stamps_report("top-of-hierarchy launch", (int)si, TP.n_stages, (size_t)((n_tiles + T - 1) / T), st);
#endif
                if (ev)
                    for (size_t k = si; k <= stages_.size(); ++k) HG_HIP(hipEventRecord(ev[e++], st));   // the first event carries the launch's time
                HG_HIP(hipGetLastError());
                return;
            }
            const SubRun* sr = nullptr;
            for (const SubRun& r : sub_runs_)
                if (r.begin == (int)si && r.set == sub_set && sub_run_pays(r, n_tiles)) sr = &r;
            if (sr) {
                // a short batch: these layers as independent sub-trees in one launch (hg_fused_tail.hip)
                const int sub_len_ = sr->len;
                TailParams TP = subtree_params(*sr, cur, nxt, n_tiles);
#ifdef HIGSFA_DIAG
                const bool stamped = opt_.stamp_stage == (int)si;
                if (stamped) tail_stamps_begin(TP, (size_t)sr->n * n_tiles, st);
#endif
                launch_subtree(TP, st);
#ifdef HIGSFA_DIAG
                if (stamped) tail_stamps_report("sub-tree launch", (int)si, TP.n_stages, (size_t)sr->n * n_tiles, st);
#endif
                std::swap(cur, nxt);
                if (ev)
                    for (int k = 0; k < sub_len_; ++k) HG_HIP(hipEventRecord(ev[e++], st));   // the first event carries the launch's time
                si += sub_len_ - 1;
                continue;
            }
            StageParams P = base_params(s, cur, nxt);
            if (s.kind == 1) {        // row-major input -> fragment order
                const size_t esz0 = dtype_size(x_dtype), al0 = x_dtype == HG_U8 ? 4 : 16;
                const int v4 = (s.vec_ok && ldx % 4 == 0 && ((uintptr_t)x % al0) == 0 && (ldx * esz0) % al0 == 0) ? 1 : 0;
                launch_im2frag(x, x_dtype, ldx, n, n_tiles, s.nb_out, (const int32_t*)s.d_gcol.p, nxt, v4, st);
                std::swap(cur, nxt);
                if (ev) HG_HIP(hipEventRecord(ev[e++], st));
                continue;
            }
            if (s.kind == 2 && s.ig_folded && s.n_nodes <= 8 && s.nf <= 2 && s.kb1 <= 8 && (int64_t)s.n_nodes * n_tiles <= 16384 &&
                !opt_.ig_resident) {
                // top of an iGSFA hierarchy: one node and two tiles per small workgroup, output tiles split over waves
                P.ig_folded = 1;
                launch_igfold_split(P, s.mt2, n_tiles, st);
                std::swap(cur, nxt);
                if (ev) HG_HIP(hipEventRecord(ev[e++], st));
                continue;
            }
            if (s.kind == 2) {        // iGSFA layer
                // Waves that take tiles (nwt) x tiles per wave (T): the largest shape that still gives every CU
                // a workgroup (the top layers have 16 .. 1 nodes: with 8 x 2 a single node would run on 16 CUs).
                // Measured on the 11-layer net (us per layer, 4096 rows): 8x2 beats 4x2 / 16x1 / 8x1 wherever
                // it fills the chip; T = 1 loses 30-50 % (half the MFMAs per A fragment read from LDS) even where it
                // is the only way to give every CU a workgroup; waves
                // per workgroup must be a multiple of 4 (6x2 places 2,2,1,1 waves on the SIMDs and a second
                // workgroup no longer fits).  Nodes of 5-6 input blocks fit 3 waves per SIMD: 12x2 there.
                const size_t ig_lds = (size_t)s.node_blocks * 1024 + (size_t)s.bias_floats * 4 + (size_t)s.kb1 * 8;
                int nwt = 1, T = 1;
                {
                    static const int cand[][2] = {{12, 2}, {8, 2}, {4, 2}};
                    const int forced_w = opt_.ig_w, forced_t = opt_.ig_t;   // experiments (HIGSFA_IG_SHAPE)
                    for (auto& c : cand) {
                        if (forced_w && (c[0] != forced_w || c[1] != forced_t)) continue;
                        if (c[0] == 12 && (s.ig_folded || !(s.kb1 == 5 || s.kb1 == 6))) continue;   // k_igfold: <= 8 waves
                        const int64_t tg = (n_tiles + c[0] * c[1] - 1) / (c[0] * c[1]);
                        if (forced_w || (c[0] * c[1] <= std::max(n_tiles, 1) && tg * s.n_nodes >= 256)) {
                            nwt = c[0];
                            T = c[1];
                            break;
                        }
                    }
                    if (nwt == 1 && T == 1 && !forced_w) {
                        // too few (node, tile) pairs to fill the chip: still two tiles per wave — 4 x 2 with 32
                        // workgroups beats 1 x 1 with 256 by a third on the single top node (every workgroup
                        // copies the node's 64-80 KiB of weights, and T = 1 halves the MFMAs per LDS read)
                        T = n_tiles >= 2 ? 2 : 1;
                        nwt = n_tiles >= 8 ? 4 : n_tiles >= 4 ? 2 : 1;
                    }
                }
                // folded layers stream their input blocks (k_igfold); HIGSFA_IG_RESIDENT=1 keeps them on k_igsfa
                const bool igfold = s.ig_folded && !opt_.ig_resident;
                const bool ig_fs = igfold && s.nf == 2 && s.funcs[0].kind == E_IDENTITY && s.funcs[1].kind == E_ABS_POW && !opt_.no_fspec;
                StageFn fn = igfold ? pick_igfold(s.mt2, T, ig_fs) : pick_igsfa(s.mt1, s.mt2, T, s.kb1);
                const int ig_occ = resident_blocks(fn, std::max(nwt, 4) * 64, ig_lds);
                const int nw = std::max(nwt, 4);   // never fewer than 4 waves to copy a node's weights
                P.nodes_per_wg = nwt;
                P.ig_has_lr = s.ig_has_lr ? 1 : 0;
                P.ig_folded = s.ig_folded ? 1 : 0;
                P.nk2p[0] = 0;
                for (int ms = 0; ms < s.mt1; ++ms) P.nk2p[0] |= (uint32_t)s.ig_nks[ms] << (4 * ms);
                P.tile_groups = (n_tiles + nwt * T - 1) / (nwt * T);
                const int npg = 1;   // one node per workgroup measured fastest (236/180 us vs 246-288/220-269 us with 32-96 KiB groups on the 11-layer net)
                P.nodes_per_group = npg;
                P.n_chunks = (s.n_nodes + npg - 1) / npg;
                P.tile_parts = std::max(1, std::min(P.tile_groups, 256 * ig_occ / std::max(1, P.n_chunks)));
                const size_t lds_bytes = ig_lds;
                set_lds_limit(fn, lds_bytes);
                if (opt_.debug) fprintf(stderr, "[igsfa stage %d] nodes %d kb1 %d shape %dx%d occ %d lds %zu tile_groups %d parts %d\n", (int)si, s.n_nodes, s.kb1, nwt, T, ig_occ, lds_bytes, P.tile_groups, P.tile_parts);
                hipLaunchKernelGGL(fn, (unsigned)(P.n_chunks * P.tile_parts), nw * 64, lds_bytes, st, P);
                std::swap(cur, nxt);
                if (ev) HG_HIP(hipEventRecord(ev[e++], st));
                continue;
            }
            if (s.from_x) {
                P.chunks = (const DChunk*)s.d_chunks.p;
                P.runs = (const DRun*)s.d_runs.p;
                P.piece_col = (const int2*)s.d_piece.p;
                P.koff = (const int32_t*)s.d_koff.p;
                P.kmean = (const float*)s.d_kmean.p;
                P.kcol = (const int32_t*)s.d_kcol.p;
                P.x = x;
                P.ldx = ldx;
                P.n_rows = n;
                P.lds_stride = s.lds_stride;
                P.nk_last = s.nk_last;
                P.contig4 = s.contig4 ? 1 : 0;
                P.n_chunks = (int)s.chunks.size();
                const size_t esz = dtype_size(x_dtype);
                const size_t valign = x_dtype == HG_U8 ? 4 : 16;
                P.vec4 = (s.vec_ok && ldx % 4 == 0 && ((uintptr_t)x % valign) == 0 && (ldx * esz) % valign == 0) ? 1 : 0;
                const int T = (n_tiles >= 4 && s.mt1 * s.mt2 <= 4) ? 4 : 1;
                const int groups = (n_tiles + T - 1) / T;
                size_t lds_bytes = (size_t)T * 16 * s.lds_stride * 4;
                bool rem4 = false, fspec = false;
                if (fuse01_) {
                    // second tiles of both layer-1 affines hold <= 4 real rows: 4x4x1 MFMA form (HIGSFA_NO_REM4: off)
                    rem4 = stages_[1].rem4 || (stages_[1].p_max <= 20 && stages_[1].s_max <= 20 && stages_[1].nk2[1][0] <= 1 &&
                                               stages_[1].nk2[1][1] <= 1 && !opt_.no_rem4);
                    auto id_pow = [](const HostStage& hs) { return hs.nf == 2 && hs.funcs[0].kind == E_IDENTITY && hs.funcs[1].kind == E_ABS_POW; };
                    fspec = id_pow(s) && id_pow(stages_[1]) && !opt_.no_fspec;
                }
                const int FT = stage01p_tiles(rem4, fspec);      // batch tiles per pass of the fused front kernel
                if (fuse01_ && P.vec4 && n_tiles >= FT) {
                    // layers 0 and 1 in one persistent kernel; layer 1 writes where its own launch would
                    StageParams Q = base_params(stages_[1], nullptr, cur);
                    // every wave on its own (k_stage01d) where the input layout allows it, else the LDS-staged kernel
                    const bool direct = fspec && rem4 && s.direct_ok && stages_[1].pack_out && (int64_t)16 * ldx * (int64_t)esz + 2048 < 0x7fffffffll && s.max_chunk_nodes <= 8 && !opt_.no_direct;   // (k_stage01d: 32-bit byte offsets inside a tile's rows; at most four waves)
                    const bool wgq = !opt_.no_wgq;       // k_stage01d: one tile queue per chunk, shared by the waves of a workgroup
                    StageFn2 fn = direct ? pick_stage01d(x_dtype, false, wgq) : pick_stage01p(x_dtype, false, rem4, fspec);
                    const int thr01 = 64 * std::max(1, (s.max_chunk_nodes + 1) / 2);   // one wave per pair of layer-0 nodes
                    // LDS: the tile buffers of FT batch tiles (k_stage01p only) + 10 vectors of 16 floats (means, biases) per wave
                    const size_t lds2 = direct ? (size_t)4 * 160 * 4 + 32 * 4 + (size_t)4 * 256 * 4      // constants | ring + progress words | reduction scratch per wave
                                               : (size_t)(kDoubleBuffer01 ? 2 : 1) * FT * 16 * s.lds_stride * 4 + (size_t)(thr01 / 64) * 160 * 4 + 16;   // + tile-group queue slots
                    const int groups2 = (n_tiles + FT - 1) / FT;
                    int occ = 1;
                    set_lds_limit((StageFn)fn, lds2);
                    {
                        auto key = std::make_tuple((const void*)fn, thr01, lds2);
                        auto it = occ_.find(key);
                        if (it == occ_.end()) {
                            int nb = 0;
                            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)fn, thr01, lds2) != hipSuccess || nb < 1) nb = 1;
                            occ_[key] = nb;
                            occ = nb;
                        } else {
                            occ = it->second;
                        }
                    }
                    P.tile_parts = std::max(1, std::min(groups2, 256 * occ / std::max(1, P.n_chunks)));
                    if (opt_.debug) fprintf(stderr, "[front] occ %d threads %d lds %zu chunks %d tile_parts %d\n", occ, thr01, lds2, P.n_chunks, P.tile_parts);
#ifdef HIGSFA_DIAG
                    if (opt_.stamp_stage == 0 && x_dtype == HG_F32) {
                        fn = direct ? pick_stage01d(HG_F32, true, wgq) : pick_stage01p(HG_F32, true, rem4, fspec);
                        stamp_blocks_ = P.n_chunks * P.tile_parts;
                        stamp_buf_.alloc((size_t)stamp_blocks_ * 8 * 12 * 8);
                        HG_HIP(hipMemsetAsync(stamp_buf_.p, 0, stamp_buf_.bytes, st));
                        P.stamps = (unsigned long long*)stamp_buf_.p;
                    }
#endif
                    P.err = device_error_word();
                    {   // k_stage01p: one queue of tile groups per chunk; k_stage01d: one queue of tiles per layer-1 node (or per chunk)
                        WorkQueue& wq = direct ? (wgq ? wq_direct_wg_ : wq_direct_) : wq_front_;
                        P.work_ctr = work_counters(wq, direct && !wgq ? stages_[1].n_nodes : P.n_chunks, st);
                        P.work_base = wq.base;
                        wq.base += (uint32_t)groups2;      // what this launch adds to every counter (StageParams::work_ctr)
                    }
#ifdef HIGSFA_DIAG
                    if (const char* e = getenv("HIGSFA_WHATIF")) P.whatif = atoi(e);
#endif
                    hipLaunchKernelGGL(fn, (unsigned)(P.n_chunks * P.tile_parts), thr01, lds2, st, P, Q);
                    if (P.stamps) {
                        HG_HIP(hipStreamSynchronize(st));
                        std::vector<unsigned long long> h((size_t)stamp_blocks_ * 8 * 12);
                        HG_HIP(hipMemcpy(h.data(), stamp_buf_.p, h.size() * 8, hipMemcpyDeviceToHost));
                        double a[7] = {0, 0, 0, 0, 0, 0, 0}, nwv = 0;
                        unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0, 0, 0};     // wall clock (100 MHz): entry, loop start, end
                        for (size_t i = 0; i < h.size(); i += 12)
                            if (h[i + 4]) {
                                for (int k = 0; k < 7; ++k) a[k] += h[i + k];
                                nwv += 1;
                                for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], h[i + 7 + k]); hi[k] = std::max(hi[k], h[i + 7 + k]); }
                            }
                        {
                            double xs[8][4] = {}, ps[16][3] = {};
                            for (size_t i = 0, w = 0; i < h.size(); i += 12, ++w)
                                if (h[i + 4]) {
                                    const int blk = (int)(w / 8), xc = blk % 8, pt = std::min(15, blk / P.n_chunks);
                                    const double life = (h[i + 9] - h[i + 8]) * 0.01, clk = (double)h[i + 4] / (double)h[i + 5] * 100.0;
                                    xs[xc][0] += 1; xs[xc][1] += life; xs[xc][2] = std::max(xs[xc][2], life); xs[xc][3] += clk;
                                    ps[pt][0] += 1; ps[pt][1] += life; ps[pt][2] = std::max(ps[pt][2], life);
                                }
                            for (int k = 0; k < 8; ++k)
                                if (xs[k][0] > 0) fprintf(stderr, "[stamp stage 0+1] xcd %d: waves %.0f loop us mean %.1f max %.1f clock %.0f MHz\n", k, xs[k][0], xs[k][1] / xs[k][0], xs[k][2], xs[k][3] / xs[k][0]);
                            for (int k = 0; k < 16; ++k)
                                if (ps[k][0] > 0) fprintf(stderr, "[stamp stage 0+1] part %d: waves %.0f loop us mean %.1f max %.1f\n", k, ps[k][0], ps[k][1] / ps[k][0], ps[k][2]);
                        }
                        fprintf(stderr, "[stamp stage 0+1] wall clock, us after the first wave's entry: entries until %.1f, loop starts %.1f..%.1f, ends %.1f..%.1f\n",
                                (hi[0] - lo[0]) * 0.01, (lo[1] - lo[0]) * 0.01, (hi[1] - lo[0]) * 0.01, (lo[2] - lo[0]) * 0.01, (hi[2] - lo[0]) * 0.01);
                        fprintf(stderr, "[stamp stage 0+1] waves %.0f clock %.0f MHz lifetime %.1f us, iterations %.1f; cycles per iteration: "
                                        "lds-write+barrier %.0f, fetch issue %.0f, layer0 %.0f, layer1+store %.0f (sum %.0f)\n",
                                nwv, a[4] / a[5] * 100.0, a[5] / nwv / 100.0, a[6] / nwv, a[0] / a[6], a[1] / a[6], a[2] / a[6], a[3] / a[6],
                                (a[0] + a[1] + a[2] + a[3]) / a[6]);
                    }
                    if (ev) {
                        HG_HIP(hipEventRecord(ev[e++], st));   // stage 0 (carries the fused time)
                        HG_HIP(hipEventRecord(ev[e++], st));   // stage 1 (fused: no launch of its own)
                    }
                    si = 1;   // cur / nxt unchanged: stage 2 reads `cur`
                    continue;
                }
                const bool persistent = T == 4 && P.vec4 && s.contig4 && s.has_exp && s.mt1 == 1 && s.mt2 == 1 && s.kb1 == 1 &&
                                        s.nf <= 2 && s.max_chunk_nodes <= 16 && 64 * s.max_chunk_pieces <= 8 * 512;
                if (persistent) {
                    StageFn fn = pick_stage0p(x_dtype);
                    const int occ = resident_blocks(fn, 512, lds_bytes);
                    int parts = std::max(1, std::min(groups, 256 * occ / std::max(1, P.n_chunks)));
                    P.tile_parts = parts;
                    hipLaunchKernelGGL(fn, (unsigned)(P.n_chunks * parts), 512, lds_bytes, st, P);
                } else {
                    const int64_t blocks = (int64_t)groups * P.n_chunks;
                    if (blocks > 0x7fffffffll) fail(HG_ERR_ARG, "batch too large");
                    StageFn fn = pick_stage0(s.mt1, s.mt2, T, x_dtype);
                    set_lds_limit(fn, lds_bytes);
                    hipLaunchKernelGGL(fn, (unsigned)blocks, T == 4 ? 512 : 256, lds_bytes, st, P);
                }
            } else {
                // k_stage_splitm (one node and T tiles per small workgroup, weights straight from L2, no LDS copy): the top of the
                // hierarchy at any batch size, and EVERY ordinary layer while its grid is small enough to be resident at once —
                // a k_stage workgroup first copies 27-52 KiB of weights into LDS (5 us), which small batches never earn back
                // (N = 16: 92 -> 70 us per call, N = 340: 137 -> 129, N = 1024: the same; grids of more than ~500 workgroups: k_stage wins)
                const int T_sm = n_tiles >= 2 * 256 / std::max(1, s.n_nodes) ? 2 : 1;
                const int64_t wgs_sm = (int64_t)((n_tiles + T_sm - 1) / T_sm) * s.n_nodes;
                if (s.kind == 0 && (!s.rem4 || s.mt1 == s.mt2) && (s.rem4 || !s.pack_out) && s.mt1 * s.nf <= 8 &&
                    (s.n_nodes <= opt_.splitm_max_nodes ? (int64_t)s.n_nodes * n_tiles <= 8192 : wgs_sm <= opt_.splitm_max_wgs)) {
                    const int T = T_sm;
                    const int groups = (n_tiles + T - 1) / T;
                    const int nwv = std::max(s.mt1, s.has_exp ? s.mt2 : 1);
                    size_t lds_bytes = (size_t)std::max(1, s.nf) * s.mt1 * T * 1024;
                    if (s.rem4) {
                        if (T == 2) hipLaunchKernelGGL((k_stage_splitm<2, true>), (unsigned)(groups * s.n_nodes), nwv * 64, lds_bytes, st, P, s.mt1, s.mt2);
                        else hipLaunchKernelGGL((k_stage_splitm<1, true>), (unsigned)(groups * s.n_nodes), nwv * 64, lds_bytes, st, P, s.mt1, s.mt2);
                    } else if (T == 2)
                        hipLaunchKernelGGL(k_stage_splitm<2>, (unsigned)(groups * s.n_nodes), nwv * 64, lds_bytes, st, P, s.mt1, s.mt2);
                    else
                        hipLaunchKernelGGL(k_stage_splitm<1>, (unsigned)(groups * s.n_nodes), nwv * 64, lds_bytes, st, P, s.mt1, s.mt2);
                    std::swap(cur, nxt);
                    if (ev) HG_HIP(hipEventRecord(ev[e++], st));
                    continue;
                }
                if (s.kind == 3) {
                    launch_prod(s, P, n_tiles, st);
                    std::swap(cur, nxt);
                    if (ev) HG_HIP(hipEventRecord(ev[e++], st));
                    continue;
                }
                // node groups sized so a group's weights are ~64 KiB of LDS (always >= 1 node)
                const int npg = std::max(1, std::min(s.n_nodes, kWeightLdsKiB / std::max(1, s.node_blocks)));
                const int n_groups = (s.n_nodes + npg - 1) / npg;
                // waves x tiles per workgroup: the largest shape that still yields >= 512 workgroups
                // waves x tiles per workgroup: the largest shape that still gives every CU a
                // workgroup; never fewer than 4 waves to copy a node's weights unless the batch is tiny
                static const int shapes0[][2] = {{8, 2}, {4, 2}, {4, 1}, {0, 0}};
                static const int shapes1[][2] = {{8, 2}, {8, 1}, {4, 1}, {0, 0}};
                static const int shapes2[][2] = {{8, 2}, {8, 1}, {4, 2}, {4, 1}};
                const int (*shapes)[2] = opt_.shape_variant == 1 ? shapes1 : opt_.shape_variant == 3 ? shapes0 : shapes2;      // default: 8 x 1 before 4 x 2 (layer 7: 20.6 -> 19.0 us)
                int nw = 4, T = 1;
                for (int si2 = 0; si2 < 4; ++si2) {
                    const int* sh = shapes[si2];
                    if (!sh[0]) break;
                    int64_t tg = (n_tiles + sh[0] * sh[1] - 1) / (sh[0] * sh[1]);
                    if (sh[0] * sh[1] <= n_tiles && tg * n_groups >= 256) {
                        nw = sh[0];
                        T = sh[1];
                        break;
                    }
                }
                if (opt_.stage_w > 0 && (opt_.stage_only < 0 || opt_.stage_only == (int)si)) {      // experiments (HIGSFA_STAGE_SHAPE=waves,tiles; HIGSFA_STAGE_ONLY=<stage>)
                    nw = opt_.stage_w;
                    T = opt_.stage_t;
                }
                while (nw * T > std::max(n_tiles, 1) && nw > 1) nw >>= 1;
                const int tile_groups = (n_tiles + nw * T - 1) / (nw * T);
                // persistent sweep: each workgroup copies its node group's weights once and walks
                // tile groups part, part + tile_parts, ...; aim at ~3 workgroups per CU in total
                // cost(P) = rounds of resident workgroups x (weight copy + tile iterations per workgroup)
                const int64_t g8 = (int64_t)(n_groups + 7) / 8 * 8;
                size_t lds_probe = (size_t)npg * s.node_blocks * 1024 + (size_t)npg * s.bias_floats * 4 + (size_t)npg * s.kb1 * 8;
                const int kbf = (T == 2 && s.mt1 == s.mt2 && (s.mt1 == 2 || s.mt1 == 3) && !opt_.no_prefetch_all)
                                    ? (s.kb1 == 4 ? 4 : (s.kb1 == 3 && s.rem4 ? 3 : 0)) : 0;
                const bool fs = s.has_exp && s.nf == 2 && s.funcs[0].kind == E_IDENTITY && s.funcs[1].kind == E_ABS_POW && !opt_.no_fspec;
                // slot-major packed input (plan_slot_major): the whole-visit-prefetch instantiation reads it at a fixed code position; any
                // other shape of this call (one tile per wave: small batches) takes the generic remainder-tile loop, which tests per block
                StageFn fn = s.pk_kbi >= 0 ? (kbf == 3 ? pick_stage(s.mt1, s.mt2, T, s.rem4, kbf, fs, s.pk_kbi) : nullptr) : pick_stage(s.mt1, s.mt2, T, s.rem4, kbf, fs);
                if (!fn) fn = pick_stage(s.mt1, s.mt2, T, s.rem4, 0, false);
                const double capacity = 256.0 * resident_blocks(fn, nw * 64, lds_probe);
                int tile_parts = 1;
                double best = 1e300;
                for (int pp = 1; pp <= tile_groups; ++pp) {
                    double rounds = std::ceil(g8 * pp / capacity);
                    double cost = rounds * (0.35 + (double)((tile_groups + pp - 1) / pp));
                    if (cost < best - 1e-9) {
                        best = cost;
                        tile_parts = pp;
                    }
                }
                if (opt_.stage_parts > 0 && (opt_.stage_only < 0 || opt_.stage_only == (int)si)) tile_parts = std::min(opt_.stage_parts, tile_groups);      // experiments (HIGSFA_STAGE_PARTS)
                P.nodes_per_group = npg;
                P.nodes_per_wg = npg;
                P.n_chunks = n_groups;
                P.pair_chunks = (s.pack_out && !s.pack_soa && npg == 2 && n_groups % 16 == 0 && !getenv("HIGSFA_NO_PAIR")) ? 1 : 0;
                P.tile_groups = tile_groups;
                P.tile_parts = tile_parts;
                const int64_t blocks = (int64_t)((P.n_chunks + 7) / 8) * 8 * tile_parts;
                if (blocks > 0x7fffffffll) fail(HG_ERR_ARG, "batch too large");
                size_t lds_bytes = (size_t)npg * s.node_blocks * 1024 + (size_t)npg * s.bias_floats * 4 + (size_t)npg * s.kb1 * 8;
#ifdef HIGSFA_DIAG
                const bool stamp_kbf3 = s.mt1 == 3 && s.mt2 == 3 && T == 2 && s.rem4 && kbf == 3;
                if (opt_.stamp_stage == (int)si && (stamp_kbf3 || (s.mt1 == s.mt2 && (s.mt1 == 4 || s.mt1 == 3) && T == 2 && !s.rem4 && kbf == 0))) {
                    // diagnostic instantiation with s_memtime stamps (never used in timed runs)
                    fn = stamp_kbf3 ? (StageFn)k_stage<3, 3, 2, true, true, 3> : s.mt1 == 4 ? (StageFn)k_stage<4, 4, 2, true> : (StageFn)k_stage<3, 3, 2, true>;
                    stamp_buf_.alloc((size_t)blocks * 8 * 8 * 8);
                    HG_HIP(hipMemsetAsync(stamp_buf_.p, 0, stamp_buf_.bytes, st));
                    P.stamps = (unsigned long long*)stamp_buf_.p;
                    stamp_blocks_ = (int)blocks;
                }
#endif
                set_lds_limit(fn, lds_bytes);
#ifdef HIGSFA_DIAG
                if (const char* e = getenv("HIGSFA_WHATIF")) P.whatif = atoi(e);
#endif
                hipLaunchKernelGGL(fn, (unsigned)blocks, nw * 64, lds_bytes, st, P);
                if (P.stamps) {
                    HG_HIP(hipStreamSynchronize(st));
                    std::vector<unsigned long long> h((size_t)stamp_blocks_ * 8 * 8);
                    HG_HIP(hipMemcpy(h.data(), stamp_buf_.p, h.size() * 8, hipMemcpyDeviceToHost));
                    double c = 0, g1 = 0, tl = 0, all = 0, it = 0, nwv = 0, rt = 0;
                    unsigned long long lo0 = ~0ull, hi0 = 0, lo1 = ~0ull, hi1 = 0;
                    std::vector<unsigned long long> ends;
                    for (size_t i = 0; i < h.size(); i += 8)
                        if (h[i + 3]) {
                            c += h[i]; g1 += h[i + 1]; tl += h[i + 2]; all += h[i + 3]; it += h[i + 4]; rt += h[i + 5]; nwv += 1;
                            lo0 = std::min(lo0, h[i + 6]); hi0 = std::max(hi0, h[i + 6]); lo1 = std::min(lo1, h[i + 7]); hi1 = std::max(hi1, h[i + 7]);
                            ends.push_back(h[i + 7]);
                        }
                    std::sort(ends.begin(), ends.end());
                    if (!ends.empty())
                        fprintf(stderr, "[stamp stage %d] wall clock, us after the first wave's start: starts until %.1f, ends %.1f (10%%) %.1f (50%%) %.1f (90%%) %.1f (last); grid %lld blocks x %d waves\n",
                                (int)si, (hi0 - lo0) * 0.01, (ends[ends.size() / 10] - lo0) * 0.01, (ends[ends.size() / 2] - lo0) * 0.01,
                                (ends[ends.size() * 9 / 10] - lo0) * 0.01, (hi1 - lo0) * 0.01, (long long)blocks, nw);
                    fprintf(stderr, "[stamp stage %d] in-kernel clock %.0f MHz, wave lifetime %.1f us\n", (int)si, all / rt * 100.0, rt / nwv / 100.0);
                    fprintf(stderr, "[stamp stage %d] waves %.0f  avg cycles/wave: total %.0f  copy+barrier %.0f  gemm1 %.0f  tail %.0f  node-iterations %.1f  (per iteration: gemm1 %.0f tail %.0f)\n",
                            (int)si, nwv, all / nwv, c / nwv, g1 / nwv, tl / nwv, it / nwv, g1 / it, tl / it);
                }
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
        wq_front_.ctr.free();
        wq_direct_.ctr.free();
        wq_direct_wg_.ctr.free();
        if (err_host_) (void)hipHostFree(err_host_);
        err_host_ = nullptr;
        err_dev_ = nullptr;
        d_col_base_.free();
        d_col_of_.free();
        for (SubRun& r : sub_runs_)
            for (int k = 0; k < kMaxTail; ++k) { r.d_nodes[k].free(); r.d_tab[k].free(); }
        for (auto& s : stages_) {
            s.d_afrag.free(); s.d_bias.free(); s.d_kb1tab.free(); s.d_chunks.free();
            s.d_runs.free(); s.d_piece.free(); s.d_koff.free(); s.d_kmean.free(); s.d_kcol.free(); s.d_gcol.free(); s.d_etab.free(); s.d_pack_slot.free();
        }
        cap_rows_ = 0;
    }

private:
    // resident workgroups per CU for (kernel, block size, LDS) — cached occupancy query
    int resident_blocks(StageFn fn, int threads, size_t lds) {
        set_lds_limit(fn, lds);
        auto key = std::make_tuple((const void*)fn, threads, lds);
        auto it = occ_.find(key);
        if (it != occ_.end()) return it->second;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)fn, threads, lds) != hipSuccess || nb < 1) nb = 1;
        occ_[key] = nb;
        return nb;
    }

    // Counters of the dynamic tile queues (16 words apart); zeroed once, then only ever advanced.  One set per kernel that
    // uses them (every launch advances all of ITS counters by the same amount).
    struct WorkQueue {
        DevBuf ctr;
        uint32_t base = 0;
    };
    uint32_t* work_counters(WorkQueue& q, int n, hipStream_t st) {
        const size_t need = (size_t)n * 64;
        if (q.ctr.bytes < need) {
            q.ctr.alloc(std::max<size_t>(need, 64 * 1024));
            // counters and base start at the same value and only their difference matters; HIGSFA_WQ_START puts them just below
            // 2^32 so that a test can cross the wrap-around that a long-running process reaches after ~10^7 calls
            HG_HIP(hipMemsetD32Async((hipDeviceptr_t)q.ctr.p, (int)opt_.wq_start, q.ctr.bytes / 4, st));
            q.base = opt_.wq_start;
        }
        return (uint32_t*)q.ctr.p;
    }

    // Error word shared with the kernels that poll (k_chain, k_stage01d with the workgroup queue): pinned, device-mapped
    // host memory, so the host can look at it without synchronising.  A poll only runs out on a bug; the run that hit it
    // produced wrong features, and every later call on this flow fails loudly.
    int32_t* device_error_word() {
        if (!err_host_) {
            HG_HIP(hipHostMalloc((void**)&err_host_, 64, hipHostMallocMapped));
            *err_host_ = 0;
            HG_HIP(hipHostGetDevicePointer((void**)&err_dev_, err_host_, 0));
        }
        return err_dev_;
    }
    void check_errors() override { check_device_error(); }
    void check_device_error() {
        if (err_host_ && *(volatile int32_t*)err_host_ != 0)
            fail(HG_ERR_DEVICE, "fused: a bounded poll in a persistent kernel ran out (code %d); the features of that call are invalid", (int)*err_host_);
    }

    void set_lds_limit(StageFn fn, size_t bytes) {
        if (bytes <= 64 * 1024) return;
        if (bytes > 160 * 1024) fail(HG_ERR_FORMAT, "fused: a node needs %zu bytes of LDS (> 160 KiB)", bytes);
        auto it = lds_set_.find((const void*)fn);
        if (it != lds_set_.end() && it->second >= bytes) return;
        HG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        lds_set_[(const void*)fn] = 160 * 1024;
    }

    // Stage 0: group consecutive nodes into chunks whose distinct input columns fit the LDS tile,
    // turn each chunk's column set into contiguous runs, and record for every node input position
    // its word offset inside the staged row.
    void plan_stage0_inputs(const FStage& st, HostStage& hs) {
        const int n = (int)st.nodes.size();
        int kb1 = 0, max_in = 0;
        for (auto& nd : st.nodes) {
            kb1 = std::max(kb1, (nd.in_dim + 15) / 16);
            max_in = std::max(max_in, nd.in_dim);
        }
        // K-slot assignment inside a block of 16 input positions: slot (r, g) <- position 4r+g, or the
        // transposed 4g+r when that makes the four k-steps of a lane contiguous in the input row
        // (e.g. 4-pixel-wide receptive fields: one ds_read_b128 per fragment instead of four b32)
        s0_transpose_ = true;
        for (auto& nd : st.nodes) {
            for (int c = 0; c < nd.in_dim && s0_transpose_; c += 4)
                for (int r = 1; r < 4 && c + r < nd.in_dim; ++r)
                    if (st.conn[nd.in_off + c + r] != st.conn[nd.in_off + c] + r) s0_transpose_ = false;
            if (nd.in_dim % 4) s0_transpose_ = false;
        }
        const int col_budget = std::max(kStage0ChunkCols, (max_in + 3) / 4 * 4);
        if (col_budget > 2048) fail(HG_ERR_FORMAT, "fused: first-layer node with %d inputs", max_in);
        hs.koff.assign((size_t)n * kb1 * 16, -1);
        hs.kmean.assign((size_t)n * kb1 * 16, 0.f);
        int max_cols = 0, ni = 0;
        bool vec_ok = true, contig = true;
        while (ni < n) {
            std::vector<int32_t> cols;
            int n1 = ni;
            while (n1 < n) {
                std::vector<int32_t> c2 = cols;
                const FNode& nd = st.nodes[n1];
                for (int c = 0; c < nd.in_dim; ++c) c2.push_back(st.conn[nd.in_off + c]);
                std::sort(c2.begin(), c2.end());
                c2.erase(std::unique(c2.begin(), c2.end()), c2.end());
                if ((int)c2.size() > col_budget && n1 > ni) break;
                cols.swap(c2);
                ++n1;
            }
            DChunk ck{ni, n1 - ni, (int)hs.runs.size(), 0, 0, (int)hs.piece_col.size() / 2, 0, 0};
            std::map<int32_t, int32_t> lds_of;
            // Four short runs (4-pixel-high fields: run = pixel row = lane group g of the transposed K slots) pack
            // into 128 words: a ds_read_b128 serves lanes in four 16-lane groups that each mix HALF of lane group
            // g = 0 with half of g = 1 (or g = 2 with g = 3; MI355X_MICROARCH.md §LDS), so only those pairs have to
            // agree modulo the 64-word bank row: offsets 0, 64, 32, 96.  Half the LDS of one-run-per-bank-row.
            int n_runs = 0, max_len = 0;
            for (size_t i = 0; i < cols.size();) {
                size_t k = i + 1;
                while (k < cols.size() && cols[k] == cols[k - 1] + 1) ++k;
                ++n_runs;
                max_len = std::max(max_len, (int)(k - i));
                i = k;
            }
            const bool packed4 = s0_transpose_ && n_runs == 4 && max_len <= 32;
            static const int packed_off[4] = {0, 64, 32, 96};
            int off = 0, run_i = 0, hi = 0;
            for (size_t i = 0; i < cols.size(); ++run_i) {
                size_t k = i + 1;
                while (k < cols.size() && cols[k] == cols[k - 1] + 1) ++k;
                const int len = (int)(k - i);
                // otherwise every run starts on a 64-word (256 B = one LDS bank row) boundary: the four lane
                // groups of a ds_read_b128 then differ only by multiples of the bank row and the 16
                // sub-images of a group (stride == 4 mod 64) take 16 distinct 16-byte slots
                off = packed4 ? packed_off[run_i] : (off + 63) / 64 * 64;
                hs.runs.push_back(DRun{cols[i], len, off, 0});
                if (cols[i] % 4 || len % 4) vec_ok = false;
                for (size_t m = i; m < k; ++m) lds_of[cols[m]] = off + (int)(m - i);
                for (int pc = 0; pc + 3 < len; pc += 4) {
                    hs.piece_col.push_back(cols[i] + pc);
                    hs.piece_col.push_back(off + pc);
                    ++ck.n_pieces;
                }
                off += len;
                hi = std::max(hi, off);
                i = k;
                ++ck.run_count;
            }
            off = hi;
            ck.n_cols = off;
            max_cols = std::max(max_cols, off);
            for (int k = ni; k < n1; ++k) {
                const FNode& nd = st.nodes[k];
                for (int kb = 0; kb < kb1; ++kb)
                    for (int g = 0; g < 4; ++g)
                        for (int r = 0; r < 4; ++r) {
                            int c = kb * 16 + s0_pos(r, g);
                            size_t e = (((size_t)k * kb1 + kb) * 4 + g) * 4 + r;
                            if (c < nd.in_dim) {
                                hs.koff[e] = lds_of[st.conn[nd.in_off + c]];
                                hs.kmean[e] = (float)nd.A1.a[c];
                            }
                        }
            }
            hs.chunks.push_back(ck);
            hs.max_chunk_nodes = std::max(hs.max_chunk_nodes, ck.node_count);
            hs.max_chunk_pieces = std::max(hs.max_chunk_pieces, ck.n_pieces);
            ni = n1;
        }
        // Row stride of the LDS tile: >= max_cols + 1 (last word = the zero column padded k positions
        // read), == 4 (mod 64) words so that the 16 sub-images of a ds_read_b128 lane group land on 16
        // distinct 16-byte slots of the 256-byte bank row (rows stay 16-byte aligned).
        int stride = max_cols + 1;
        while (stride % 64 != 4) ++stride;
        hs.lds_stride = stride;
        for (size_t e = 0; e < hs.koff.size(); e += 4) {
            bool any_pad = false;
            for (int r = 0; r < 4; ++r)
                if (hs.koff[e + r] < 0) {
                    hs.koff[e + r] = stride - 1;
                    any_pad = true;
                }
            if (any_pad || hs.koff[e] % 4) contig = false;
            for (int r = 1; r < 4; ++r)
                if (hs.koff[e + r] != hs.koff[e] + r) contig = false;
        }
        hs.contig4 = contig;
        hs.vec_ok = vec_ok;
        // k_stage01d: every lane group of every node reads four contiguous, 16-byte aligned source columns
        bool direct = contig && vec_ok && s0_transpose_ && kb1 == 1;
        hs.kcol.assign((size_t)n * 4, 0);
        for (int k = 0; k < n && direct; ++k) {
            const FNode& nd = st.nodes[k];
            if (nd.in_dim != 16) { direct = false; break; }
            for (int g = 0; g < 4; ++g) {
                const int c0 = s0_pos(0, g), col = st.conn[nd.in_off + c0];
                for (int r = 0; r < 4; ++r)
                    if (s0_pos(r, g) != c0 + r || st.conn[nd.in_off + c0 + r] != col + r) direct = false;
                if (col % 4) direct = false;
                hs.kcol[(size_t)k * 4 + g] = col;
            }
        }
        hs.direct_ok = direct;
        if (!direct) hs.kcol.clear();
    }

    // First layer of iGSFA nodes: a gather pseudo-stage turns the row-major input into fragment-order
    // blocks (one block per node K-block), after which the iGSFA layer reads blocks like any other.
    void add_gather0(FStage& st, std::vector<int32_t>& prev_blk, std::vector<int32_t>& prev_q, int& prev_nb) {
        stages_.emplace_back();
        HostStage& hs = stages_.back();
        hs.kind = 1;
        hs.from_x = true;
        const int n = (int)st.nodes.size();
        int KB = 0;
        bool tr = true;
        for (auto& nd : st.nodes) {
            KB = std::max(KB, (nd.in_dim + 15) / 16);
            for (int c = 0; c < nd.in_dim && tr; c += 4)
                for (int r = 1; r < 4 && c + r < nd.in_dim; ++r)
                    if (st.conn[nd.in_off + c + r] != st.conn[nd.in_off + c] + r) tr = false;
            if (nd.in_dim % 4) tr = false;
        }
        hs.n_nodes = n;
        hs.kb1 = KB;
        hs.nb_out = n * KB;
        hs.gcol.assign((size_t)n * KB * 16, -1);
        const int in_w = (int)st.conn.size();
        prev_blk.assign(in_w, 0);
        prev_q.assign(in_w, 0);
        for (int ni = 0; ni < n; ++ni) {
            const FNode& nd = st.nodes[ni];
            for (int c = 0; c < nd.in_dim; ++c) {
                const int kb = c / 16, pp = c % 16;
                const int q = tr ? 4 * (pp % 4) + pp / 4 : pp;     // slot q = 4r + g
                hs.gcol[((size_t)ni * KB + kb) * 16 + (q & 3) * 4 + (q >> 2)] = st.conn[nd.in_off + c];
                prev_blk[nd.in_off + c] = ni * KB + kb;
                prev_q[nd.in_off + c] = q;
            }
        }
        hs.vec_ok = tr;
        for (size_t e4 = 0; e4 < hs.gcol.size() && hs.vec_ok; e4 += 4)
            if (hs.gcol[e4] >= 0 && hs.gcol[e4 + 3] == hs.gcol[e4] + 3 && hs.gcol[e4] % 4) hs.vec_ok = false;
        for (int i = 0; i < in_w; ++i) st.conn[i] = i;
        prev_nb = hs.nb_out;
        max_nb_ = std::max(max_nb_, hs.nb_out);
        std::ostringstream os;
        os << "fused gather: row-major input -> fragment order, " << n << " nodes x " << KB << " K-blocks";
        hs.name = os.str();
    }

    void build_ig_stage(FStage& st, std::vector<int32_t>& prev_blk, std::vector<int32_t>& prev_q, int& prev_nb) {
        stages_.emplace_back();
        HostStage& hs = stages_.back();
        hs.kind = 2;
        hs.has_exp = true;
        const int n = (int)st.nodes.size();
        hs.n_nodes = n;
        hs.funcs = st.nodes[0].funcs;
        hs.nf = (int)hs.funcs.size();
        hs.ig_has_lr = st.nodes[0].ig_has_lr;
        hs.nb_in = prev_nb;
        int k_max = 0, out_max = 0;
        struct NodeK {
            std::vector<int> src, nk, pos;   // per K-block: source block, k-steps; per slot: input position or -1
        };
        std::vector<NodeK> nks(n);
        for (int ni = 0; ni < n; ++ni) {
            FNode& nd = st.nodes[ni];
            if (nd.ig_has_lr != hs.ig_has_lr) fail(HG_ERR_FORMAT, "fused: iGSFA nodes of one layer differ in reconstruct_with_sfa");
            k_max = std::max(k_max, nd.ig_k);
            out_max = std::max(out_max, nd.out_dim);
            NodeK& K = nks[ni];
            std::map<int, int> blk_index;
            for (int c = 0; c < nd.in_dim; ++c) {
                const int pc = st.conn[nd.in_off + c];
                const int blk = prev_blk[pc], q = prev_q[pc];
                auto it = blk_index.find(blk);
                int kb;
                if (it == blk_index.end()) {
                    kb = (int)K.src.size();
                    blk_index[blk] = kb;
                    K.src.push_back(blk);
                    K.nk.push_back(0);
                    for (int qq = 0; qq < 16; ++qq) K.pos.push_back(-1);
                } else {
                    kb = it->second;
                }
                if (K.pos[kb * 16 + q] >= 0) fail(HG_ERR_FORMAT, "fused: iGSFA node reads one input column twice");
                K.pos[kb * 16 + q] = c;
                K.nk[kb] = std::max(K.nk[kb], q / 4 + 1);
            }
            hs.kb1 = std::max(hs.kb1, (int)K.src.size());
        }
        if (hs.kb1 > 8) fail(HG_ERR_FORMAT, "fused: iGSFA node input spans more than 8 source blocks");
        // folded form (see igsfa_affine): one GEMM from the expanded input to all output tiles
        std::vector<Aff> folded_a2(n);
        bool folded = true;
        for (int ni = 0; ni < n && folded; ++ni) folded = igsfa_affine(st.nodes[ni], folded_a2[ni], opt_.ig_nofold);
        hs.ig_folded = folded;
        if (folded) hs.ig_has_lr = false;
        const int KB = hs.kb1, MO = (out_max + 15) / 16, MS = folded ? MO : (k_max + 15) / 16, nf = hs.nf;
        hs.mt1 = MS;
        hs.mt2 = MO;
        hs.mto = MO;
        for (int ms = 0; ms < MS; ++ms) hs.ig_nks[ms] = (std::min(16, k_max - 16 * ms) + 3) / 4;
        hs.node_blocks = folded ? nf * KB * MO : nf * KB * MS + KB * MS + KB * MO;
        hs.bias_floats = MO * 16 + 2 * KB * 16;
        if ((size_t)hs.node_blocks * 1024 + (size_t)hs.bias_floats * 4 + (size_t)KB * 8 > 150 * 1024)
            fail(HG_ERR_FORMAT, "fused: one iGSFA node needs %d KiB of weight fragments, more than a workgroup's LDS", hs.node_blocks);
        hs.afrag.assign((size_t)n * hs.node_blocks * 256, 0.f);
        hs.bias.assign((size_t)n * hs.bias_floats, 0.f);
        hs.kb1tab.assign((size_t)n * KB * 2, 0);
        std::vector<int32_t> cur_blk, cur_q;
        for (int ni = 0; ni < n; ++ni) {
            FNode& nd = st.nodes[ni];
            NodeK& K = nks[ni];
            const int d = nd.in_dim, k = nd.ig_k, Q = nd.ig_pca.out;
            if (nd.ig_sfa.out != k || nd.ig_pca.in != d || k + Q != nd.out_dim) fail(HG_ERR_DIM, "fused: iGSFA node dimensions");
            std::vector<int> foff(nf), used(nf);
            int eo = 0;
            for (int fi = 0; fi < nf; ++fi) {
                foff[fi] = eo;
                used[fi] = nd.funcs[fi].used(d);
                eo += nd.funcs[fi].out_dim(d);
            }
            if (eo != nd.ig_sfa.in) fail(HG_ERR_DIM, "fused: iGSFA expansion width != sfa input_dim");
            float* wn = hs.afrag.data() + (size_t)ni * hs.node_blocks * 256;
            float* w1 = wn;
            float* w2 = w1 + (size_t)nf * KB * MS * 256;
            float* w3 = w2 + (size_t)KB * MS * 256;
            float* bn = hs.bias.data() + (size_t)ni * hs.bias_floats;
            for (int kb = 0; kb < KB; ++kb) {
                const bool real = kb < (int)K.src.size();
                hs.kb1tab[((size_t)ni * KB + kb) * 2] = real ? K.src[kb] : K.src[0];
                hs.kb1tab[((size_t)ni * KB + kb) * 2 + 1] = real ? K.nk[kb] : 0;
                if (!real) continue;
                hs.mfma_per_tile += (int64_t)K.nk[kb] * (folded ? nf * MO : nf * MS + MO);
                for (int ms = 0; ms < MS && hs.ig_has_lr; ++ms) hs.mfma_per_tile += hs.ig_nks[ms];
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 15, gg = lane >> 4;
                    for (int r = 0; r < 4; ++r) {
                        const int cs = K.pos[kb * 16 + 4 * r + gg];           // input position of this k-slot
                        if (folded) {     // rows = all output features, k-slots = expanded input positions
                            const Aff& F = folded_a2[ni];
                            for (int fi = 0; fi < nf; ++fi)
                                for (int mo = 0; mo < MO; ++mo) {
                                    const int f = 16 * mo + q_of_row(i);
                                    if (f < k + Q && cs >= 0 && cs < used[fi])
                                        w1[(((size_t)fi * KB + kb) * MO + mo) * 256 + lane * 4 + r] =
                                            (float)F.W[(size_t)(foff[fi] + cs) * (k + Q) + f];
                                }
                            continue;
                        }
                        // W1: rows = slow features, k-slots = expanded input positions
                        for (int fi = 0; fi < nf; ++fi)
                            for (int ms = 0; ms < MS; ++ms) {
                                const int fs = 16 * ms + q_of_row(i);
                                if (fs < k && cs >= 0 && cs < used[fi])
                                    w1[(((size_t)fi * KB + kb) * MS + ms) * 256 + lane * 4 + r] =
                                        (float)nd.ig_sfa.W[(size_t)(foff[fi] + cs) * k + fs];
                            }
                        // W3: rows = output features k.., k-slots = residual positions
                        for (int mo = 0; mo < MO; ++mo) {
                            const int f = 16 * mo + q_of_row(i);
                            if (f >= k && f < k + Q && cs >= 0)
                                w3[((size_t)kb * MO + mo) * 256 + lane * 4 + r] = (float)nd.ig_pca.W[(size_t)cs * Q + (f - k)];
                        }
                        // W2: rows = residual positions of this block, k-slots = slow features of tile ms
                        const int crow = K.pos[kb * 16 + q_of_row(i)];
                        if (hs.ig_has_lr && crow >= 0)
                            for (int ms = 0; ms < MS; ++ms) {
                                const int fs = 16 * ms + 4 * r + gg;
                                if (fs < k) w2[((size_t)kb * MS + ms) * 256 + lane * 4 + r] = (float)(-nd.ig_lr.W[(size_t)fs * d + crow]);
                            }
                    }
                }
                for (int gg = 0; gg < 4; ++gg)
                    for (int r = 0; r < 4; ++r) {
                        const int c = K.pos[kb * 16 + 4 * r + gg];
                        if (c < 0) continue;
                        double brv = 0;
                        if (hs.ig_has_lr) {
                            brv = -nd.ig_lr.b[c];
                            for (int fs = 0; fs < k; ++fs) brv += nd.ig_lr.a[fs] * nd.ig_lr.W[(size_t)fs * d + c];
                        }
                        bn[MO * 16 + kb * 16 + gg * 4 + r] = (float)brv;
                        bn[MO * 16 + KB * 16 + kb * 16 + gg * 4 + r] = (float)nd.ig_mean[c];
                    }
            }
            for (int mo = 0; mo < MO; ++mo)
                for (int gg = 0; gg < 4; ++gg)
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * mo + 4 * r + gg;
                        double v = 0;
                        if (folded) {
                            if (f < k + Q) v = folded_a2[ni].b[f];
                        } else if (f < k) {
                            v = nd.ig_sfa.b[f];
                            for (int e = 0; e < nd.ig_sfa.in; ++e) v -= nd.ig_sfa.a[e] * nd.ig_sfa.W[(size_t)e * k + f];
                        } else if (f < k + Q) {
                            v = nd.ig_pca.b[f - k];
                            for (int c = 0; c < d; ++c) v -= nd.ig_pca.a[c] * nd.ig_pca.W[(size_t)c * Q + (f - k)];
                        }
                        bn[mo * 16 + gg * 4 + r] = (float)v;
                    }
            for (int f = 0; f < nd.out_dim; ++f) {
                cur_blk.push_back(ni * MO + f / 16);
                cur_q.push_back(f % 16);
            }
        }
        hs.nb_out = n * MO;
        prev_blk.swap(cur_blk);
        prev_q.swap(cur_q);
        prev_nb = hs.nb_out;
        max_nb_ = std::max(max_nb_, hs.nb_out);
        padded_flops_ += hs.mfma_per_tile * 2048 / 16;
        std::ostringstream os;
        os << "fused iGSFA stage" << (folded ? " (folded to one GEMM)" : "") << ": " << n << " nodes, K-blocks " << KB << ", slow tiles " << MS << ", out tiles " << MO << ", "
           << hs.mfma_per_tile << " MFMA/tile, " << hs.afrag.size() * 4 / 1024 << " KiB weights";
        hs.name = os.str();
    }

    // Layer whose expansion holds cross-column products or is followed by a CutoffNode (k_stage_prod).  GEMM 1 as in any
    // stage > 0; the expanded input is described column by column for the WIDEST node of the layer, 16 columns per K-block
    // of GEMM 2; a narrower node's weights are scattered into that column order (columns it lacks get zero rows).
    void build_prod_stage(FStage& st, std::vector<int32_t>& prev_blk, std::vector<int32_t>& prev_q, int& prev_nb, int si) {
        stages_.emplace_back();
        HostStage& hs = stages_.back();
        hs.kind = 3;
        hs.has_exp = true;
        const int n = (int)st.nodes.size();
        hs.n_nodes = n;
        const std::vector<ExpFunc> all = st.nodes[0].funcs;
        const int nf_all = (int)all.size();
        std::vector<int> elem_of(nf_all, -1);          // function -> index among the element-wise ones
        for (int fi = 0; fi < nf_all; ++fi)
            if (all[fi].kind <= E_SIGNED_POW) {
                elem_of[fi] = (int)hs.funcs.size();
                hs.funcs.push_back(all[fi]);
            }
        hs.nf = (int)hs.funcs.size();
        if (hs.nf > kMaxFuncs) fail(HG_ERR_FORMAT, "fused: more than 4 element-wise expansion functions");
        hs.has_clip = st.nodes[0].has_clip;
        hs.clip_lo = (float)st.nodes[0].clip_lo;
        hs.clip_hi = (float)st.nodes[0].clip_hi;
        hs.nb_in = prev_nb;
        for (auto& nd : st.nodes) {
            if (!nd.has_exp) fail(HG_ERR_FORMAT, "fused: linear node in a layer with product expansions");
            hs.p_max = std::max(hs.p_max, nd.A1.out);
            hs.s_max = std::max(hs.s_max, nd.A2.out);
        }
        hs.mt1 = (hs.p_max + 15) / 16;
        hs.mt2 = (hs.s_max + 15) / 16;
        hs.mto = hs.mt2;
        for (int mt1 = 0; mt1 < hs.mt1; ++mt1)
            for (int fi = 0; fi < hs.nf; ++fi) {
                const int valid = std::max(0, std::min(16, hs.funcs[fi].used(hs.p_max) - 16 * mt1));
                hs.nk2[mt1][fi] = (uint8_t)((valid + 3) / 4);
            }
        // product columns of a node of width p: (function, i, k) in the order GeneralExpansionNode stacks them
        struct Col { int fi, i, k; };
        auto products = [&](int p) {
            std::vector<Col> cols;
            for (int fi = 0; fi < nf_all; ++fi) {
                const ExpFunc& f = all[fi];
                const int u = f.used(p);
                if (f.kind == E_QUADRATIC) { for (int i = 0; i < u; ++i) for (int k = i; k < u; ++k) cols.push_back({fi, i, k}); }
                else if (f.kind == E_PAIR_ADJ) for (int i = 0; i + (int)f.k < u; ++i) cols.push_back({fi, i, i + (int)f.k});
                else if (f.kind == E_PAIR_BAND)
                    for (int off = 0; off < (int)f.k; ++off) for (int i = 0; i + off < u; ++i) cols.push_back({fi, i, i + off});
            }
            return cols;
        };
        const std::vector<Col> cmax = products(hs.p_max);
        const int E = (int)cmax.size();
        hs.neb = (E + 15) / 16;
        hs.nk_last = hs.neb ? (std::min(16, E - 16 * (hs.neb - 1)) + 3) / 4 : 0;
        hs.etab.assign((size_t)std::max(hs.neb, 1) * 32, 0);
        for (int c = 0; c < E; ++c) {
            hs.etab[2 * c] = (int32_t)(0x80000000u | (uint32_t)(cmax[c].i * 64));     // byte offset of feature i in [feature][16 sub-images]
            hs.etab[2 * c + 1] = cmax[c].k * 64;
        }
        // K-blocks of GEMM 1
        struct NodeK {
            std::vector<int> src, nk;
            std::vector<std::vector<int>> kpos;
        };
        std::vector<NodeK> nks(n);
        for (int ni = 0; ni < n; ++ni) {
            FNode& nd = st.nodes[ni];
            NodeK& K = nks[ni];
            std::map<int, int> blk_index;
            for (int c = 0; c < nd.in_dim; ++c) {
                const int pc = st.conn[nd.in_off + c], blk = prev_blk[pc], q = prev_q[pc];
                auto it = blk_index.find(blk);
                int kb;
                if (it == blk_index.end()) {
                    kb = (int)K.src.size();
                    blk_index[blk] = kb;
                    K.src.push_back(blk);
                    K.nk.push_back(0);
                    for (int qq = 0; qq < 16; ++qq) K.kpos.emplace_back();
                } else {
                    kb = it->second;
                }
                K.kpos[kb * 16 + q].push_back(c);
                K.nk[kb] = std::max(K.nk[kb], q / 4 + 1);
            }
            hs.kb1 = std::max(hs.kb1, (int)K.src.size());
        }
        hs.node_blocks = hs.kb1 * hs.mt1 + (hs.mt1 * hs.nf + hs.neb) * hs.mt2;
        hs.bias_floats = (hs.mt1 + hs.mt2) * 16;
        if ((size_t)hs.node_blocks * 1024 + (size_t)hs.bias_floats * 4 + (size_t)hs.kb1 * 8 + (size_t)hs.neb * 128 + (size_t)4 * hs.mt1 * 1024 > 150 * 1024)
            fail(HG_ERR_FORMAT, "fused: one node needs %d KiB of weight fragments (%d product columns), more than a workgroup's LDS", hs.node_blocks, E);
        hs.afrag.assign((size_t)n * hs.node_blocks * 256, 0.f);
        hs.bias.assign((size_t)n * hs.bias_floats, 0.f);
        hs.kb1tab.assign((size_t)n * hs.kb1 * 2, 0);
        std::vector<int32_t> cur_blk, cur_q;
        for (int ni = 0; ni < n; ++ni) {
            FNode& nd = st.nodes[ni];
            NodeK& K = nks[ni];
            const int p = nd.A1.out, sdim = nd.A2.out;
            float* wnode = hs.afrag.data() + (size_t)ni * hs.node_blocks * 256;
            float* bnode = hs.bias.data() + (size_t)ni * hs.bias_floats;
            std::vector<double> bias1 = nd.A1.b;
            for (int c = 0; c < nd.in_dim; ++c)
                for (int o = 0; o < p; ++o) bias1[o] -= nd.A1.a[c] * nd.A1.W[(size_t)c * p + o];
            for (int kb = 0; kb < hs.kb1; ++kb) {
                const bool real = kb < (int)K.src.size();
                hs.kb1tab[((size_t)ni * hs.kb1 + kb) * 2] = real ? K.src[kb] : K.src[0];
                hs.kb1tab[((size_t)ni * hs.kb1 + kb) * 2 + 1] = real ? K.nk[kb] : 0;
                if (!real) continue;
                hs.mfma_per_tile += (int64_t)K.nk[kb] * hs.mt1;
                for (int mt = 0; mt < hs.mt1; ++mt) {
                    float* blk = wnode + ((size_t)kb * hs.mt1 + mt) * 256;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 15, gg = lane >> 4, fo = 16 * mt + q_of_row(i);
                        if (fo >= p) continue;
                        for (int r = 0; r < 4; ++r) {
                            double w = 0;
                            for (int c : K.kpos[kb * 16 + 4 * r + gg]) w += nd.A1.W[(size_t)c * p + fo];
                            blk[lane * 4 + r] = (float)w;
                        }
                    }
                }
            }
            for (int mt = 0; mt < hs.mt1; ++mt)
                for (int gg = 0; gg < 4; ++gg)
                    for (int r = 0; r < 4; ++r) {
                        const int fo = 16 * mt + 4 * r + gg;
                        bnode[mt * 16 + gg * 4 + r] = fo < p ? (float)bias1[fo] : 0.f;
                    }
            // rows of this node's W2: its own expanded columns in GeneralExpansionNode order
            std::vector<int> foff(nf_all);
            int eo = 0;
            for (int fi = 0; fi < nf_all; ++fi) {
                foff[fi] = eo;
                eo += all[fi].out_dim(p);
            }
            if (eo != nd.A2.in) fail(HG_ERR_DIM, "fused: expansion width %d != second affine input_dim %d", eo, nd.A2.in);
            std::map<std::tuple<int, int, int>, int> row_of;      // product (function, i, k) -> row
            {
                const std::vector<Col> cn = products(p);
                std::vector<int> cnt(nf_all, 0);
                for (auto& c : cn) row_of[std::make_tuple(c.fi, c.i, c.k)] = foff[c.fi] + cnt[c.fi]++;
            }
            std::vector<double> bias2 = nd.A2.b;
            for (int c = 0; c < nd.A2.in; ++c)
                for (int o = 0; o < sdim; ++o) bias2[o] -= nd.A2.a[c] * nd.A2.W[(size_t)c * sdim + o];
            float* w2 = wnode + (size_t)hs.kb1 * hs.mt1 * 256;
            for (int mt1 = 0; mt1 < hs.mt1; ++mt1)
                for (int fa = 0; fa < nf_all; ++fa) {
                    const int fi = elem_of[fa];
                    if (fi < 0) continue;
                    const int used = all[fa].used(p);
                    hs.mfma_per_tile += (int64_t)hs.nk2[mt1][fi] * hs.mt2;
                    for (int mt2 = 0; mt2 < hs.mt2; ++mt2) {
                        float* blk = w2 + ((size_t)(mt1 * hs.nf + fi) * hs.mt2 + mt2) * 256;
                        for (int lane = 0; lane < 64; ++lane) {
                            const int i = lane & 15, gg = lane >> 4, fo = 16 * mt2 + q_of_row(i);
                            if (fo >= sdim) continue;
                            for (int r = 0; r < 4; ++r) {
                                const int fz = 16 * mt1 + 4 * r + gg;
                                if (fz >= used) continue;
                                blk[lane * 4 + r] = (float)nd.A2.W[(size_t)(foff[fa] + fz) * sdim + fo];
                            }
                        }
                    }
                }
            float* wp = w2 + (size_t)hs.mt1 * hs.nf * hs.mt2 * 256;
            for (int eb = 0; eb < hs.neb; ++eb) {
                hs.mfma_per_tile += (int64_t)(eb + 1 < hs.neb ? 4 : hs.nk_last) * hs.mt2;
                for (int mt2 = 0; mt2 < hs.mt2; ++mt2) {
                    float* blk = wp + ((size_t)eb * hs.mt2 + mt2) * 256;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 15, gg = lane >> 4, fo = 16 * mt2 + q_of_row(i);
                        if (fo >= sdim) continue;
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * eb + 4 * r + gg;
                            if (c >= E) continue;
                            auto it = row_of.find(std::make_tuple(cmax[c].fi, cmax[c].i, cmax[c].k));
                            if (it == row_of.end()) continue;        // a column only wider nodes have
                            blk[lane * 4 + r] = (float)nd.A2.W[(size_t)it->second * sdim + fo];
                        }
                    }
                }
            }
            for (int mt = 0; mt < hs.mt2; ++mt)
                for (int gg = 0; gg < 4; ++gg)
                    for (int r = 0; r < 4; ++r) {
                        const int fo = 16 * mt + 4 * r + gg;
                        bnode[hs.mt1 * 16 + mt * 16 + gg * 4 + r] = fo < sdim ? (float)bias2[fo] : 0.f;
                    }
            for (int f = 0; f < sdim; ++f) {
                cur_blk.push_back(ni * hs.mto + f / 16);
                cur_q.push_back(f % 16);
            }
        }
        hs.nb_out = n * hs.mto;
        prev_blk.swap(cur_blk);
        prev_q.swap(cur_q);
        prev_nb = hs.nb_out;
        max_nb_ = std::max(max_nb_, hs.nb_out);
        padded_flops_ += hs.mfma_per_tile * 2048 / 16;
        std::ostringstream os;
        os << "fused stage " << si << " (table-driven expansion: products" << (hs.has_clip ? ", clip" : "") << "): " << n << " nodes, K-blocks " << hs.kb1
           << ", tiles " << hs.mt1 << "x" << hs.mt2 << ", " << hs.nf << " element-wise functions, " << E << " product columns in " << hs.neb
           << " K-blocks, " << hs.mfma_per_tile << " MFMA/tile, " << hs.afrag.size() * 4 / 1024 << " KiB weights";
        hs.name = os.str();
    }

    void launch_prod(HostStage& s, StageParams& P, int n_tiles, hipStream_t st) {
        // per-wave z image: T * mt1 KiB; the node group takes what is left of ~52 KiB (three workgroups per CU: the product
        // columns are LDS reads and multiplies between short MFMA runs, which only other waves can cover), more if one node needs it
        static const int shapes[][2] = {{8, 2}, {4, 2}, {4, 1}};
        int nw = 4, T = 1;
        for (auto& sh : shapes) {
            const int64_t tg = (n_tiles + sh[0] * sh[1] - 1) / (sh[0] * sh[1]);
            if (sh[0] * sh[1] <= n_tiles && tg * s.n_nodes >= 256) {
                nw = sh[0];
                T = sh[1];
                break;
            }
        }
        while (nw * T > std::max(n_tiles, 1) && nw > 1) nw >>= 1;
        const size_t zs_bytes = (size_t)nw * T * s.mt1 * 1024, et_bytes = (size_t)s.neb * 128;
        const size_t per_node = (size_t)s.node_blocks * 1024 + (size_t)s.bias_floats * 4 + (size_t)s.kb1 * 8;
        const size_t fixed = zs_bytes + et_bytes;
        const int npg = (int)std::max<size_t>(1, std::min<size_t>(s.n_nodes, fixed + per_node <= 52 * 1024 ? (52 * 1024 - fixed) / per_node : 1));
        const int n_groups = (s.n_nodes + npg - 1) / npg;
        const int tile_groups = (n_tiles + nw * T - 1) / (nw * T);
        const size_t lds_bytes = (size_t)npg * per_node + et_bytes + zs_bytes;
        StageFn fn = pick_prod(s.mt1, s.mt2, T);
        const double capacity = 256.0 * resident_blocks(fn, nw * 64, lds_bytes);
        const int64_t g8 = (int64_t)(n_groups + 7) / 8 * 8;
        int tile_parts = 1;
        double best = 1e300;
        for (int pp = 1; pp <= tile_groups; ++pp) {
            const double rounds = std::ceil(g8 * pp / capacity);
            const double cost = rounds * (0.35 + (double)((tile_groups + pp - 1) / pp));
            if (cost < best - 1e-9) {
                best = cost;
                tile_parts = pp;
            }
        }
        P.nodes_per_group = npg;
        P.nodes_per_wg = npg;
        P.n_chunks = n_groups;
        P.tile_groups = tile_groups;
        P.tile_parts = tile_parts;
        P.etab = (const int2*)s.d_etab.p;
        P.neb = s.neb;
        P.nk_last = s.nk_last;
        P.has_clip = s.has_clip ? 1 : 0;
        P.clip_lo = s.clip_lo;
        P.clip_hi = s.clip_hi;
        const int64_t blocks = (int64_t)((n_groups + 7) / 8) * 8 * tile_parts;
        if (blocks > 0x7fffffffll) fail(HG_ERR_ARG, "batch too large");
        set_lds_limit(fn, lds_bytes);
        hipLaunchKernelGGL(fn, (unsigned)blocks, nw * 64, lds_bytes, st, P);
    }

    // The suffix of the stage list that k_tail runs as one launch: ordinary layers (no remainder / packed tiles, K-blocks of the
    // second affine within one load batch) whose widest one needs at most 16 waves — the 4-2-1 nodes at the top of the preset
    // networks — and whose LDS tiles fit; the last layer always qualifies on its own when it is ordinary (then the launch is
    // k_stage_splitm's work plus the row-major store).
    // Slot-major packed blocks (StageParams::pack_soa) between a producer that packs and a consumer whose large-batch kernel is the
    // whole-visit-prefetch instantiation with remainder tiles (k_stage<.., REM, KBF = 3, .., PK>): every node of the consumer must
    // read exactly one packed block, at the same position of its K-block list, with at most two k-steps.  The consumer's A
    // fragments of that block are shifted so that its k-steps come first, and the table entry carries the first slot instead of
    // a first k-step (load_kblock_soa).  Same products in the same order as the lane-major form.  HIGSFA_NO_SOA=1: off.
    void plan_slot_major() {
        if (getenv("HIGSFA_NO_SOA")) return;
        for (size_t si = 0; si + 1 < stages_.size(); ++si) {
            HostStage& pr = stages_[si];
            HostStage& co = stages_[si + 1];
            if (!pr.pack_out || co.kind != 0 || !co.rem4 || co.kb1 != 3 || co.mt1 != co.mt2 || (co.mt1 != 2 && co.mt1 != 3)) continue;
            const int base = pr.n_nodes * (pr.mto - 1);
            int pk = -1;
            bool ok = true;
            for (int ni = 0; ni < co.n_nodes && ok; ++ni) {
                int seen = 0;
                for (int kb = 0; kb < co.kb1; ++kb) {
                    const int src = co.kb1tab[((size_t)ni * co.kb1 + kb) * 2], y = co.kb1tab[((size_t)ni * co.kb1 + kb) * 2 + 1];
                    if (src < base) continue;
                    const int nk = y & 255, r0 = y >> 8;
                    ++seen;
                    if (nk == 0 || nk - r0 > 2 || (pk >= 0 && pk != kb)) ok = false;
                    pk = kb;
                }
                if (seen != 1) ok = false;
            }
            if (!ok || pk != 1) continue;      // (instantiated for position 1 of 3: two children with one remainder block between their full tiles)
            for (int ni = 0; ni < co.n_nodes; ++ni) {
                int32_t& y = co.kb1tab[((size_t)ni * co.kb1 + pk) * 2 + 1];
                const int nk = y & 255, r0 = y >> 8;
                float* wnode = co.afrag.data() + (size_t)ni * co.node_blocks * 256;
                for (int mt = 0; mt < co.mt1; ++mt) {
                    float* blk = wnode + ((size_t)pk * co.mt1 + mt) * 256;
                    for (int lane = 0; lane < 64; ++lane) {
                        float v[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int r = r0; r < nk; ++r) v[r - r0] = blk[lane * 4 + r];
                        for (int r = 0; r < 4; ++r) blk[lane * 4 + r] = v[r];
                    }
                }
                y = (nk - r0) | (r0 << 16);
            }
            pr.pack_soa = true;
            co.pack_in = base;
            co.pk_kbi = pk;
            pr.name += "  [packed blocks slot-major]";
        }
    }

    void plan_tail() {
        tail_begin_ = -1;
        if (opt_.tail_max <= 0) return;
        const int ns = (int)stages_.size();
        int b = ns;
        int act_blocks = 0, e_blocks = 0;
        while (b > (fuse01_ ? 2 : 1) && ns - b < opt_.tail_max) {
            const HostStage& s = stages_[b - 1];
            if (s.kind != 0 || s.from_x || s.rem4 || s.pack_out || s.nf > kMaxFuncs) break;
            if (s.has_exp && s.mt1 * s.nf > 8) break;
            const int waves = s.n_nodes * (s.has_exp ? std::max(s.mt1, s.mt2) : s.mt1);
            if (waves > 16) break;
            const int act = b - 1 < ns - 1 ? std::max(act_blocks, s.nb_out) : act_blocks;      // the last layer's output goes to y
            const int eb = std::max(e_blocks, s.has_exp ? s.n_nodes * s.nf * s.mt1 : 0);
            if (((size_t)2 * act + eb) * 1024 > 150 * 1024) break;
            act_blocks = act;
            e_blocks = eb;
            --b;
        }
        if (b == ns) return;
        tail_begin_ = b;
        tail_act_blocks_ = act_blocks;
        tail_e_blocks_ = e_blocks;
        for (int i = b; i < ns; ++i)
            stages_[i].name += i == b ? (ns - b > 1 ? "  [this and the layers above: ONE launch, activations in LDS, writes the caller's rows]"
                                                    : "  [writes the caller's rows: no unpack pass]")
                                      : "  [in the top-of-hierarchy launch]";
    }

    // Layers below the top that fall into independent sub-trees (k_subtree, hg_fused_tail.hip): runs of two or three ordinary
    // layers under the top-of-hierarchy launch of a short batch, if the nodes each root (node of a run's last layer) draws on,
    // layer by layer, are as many for every root and shared with no other root.
    void plan_subtree() {
        sub_runs_.clear();
        if (opt_.subtree_max_tiles <= 0) return;
        const int ns = (int)stages_.size();
        // (the layer under a short batch's k_tail launch runs alone: tail_start)
        const int end = tail_begin_ < 0 ? ns : (ns - tail_begin_ >= 3 ? tail_begin_ + 1 : tail_begin_);
        // Two alternative sets of runs, each taken from the top down, a run ending where the one above begins.  Set 0 starts right under
        // the top launch (U11L-128: layers 6-8 as 4 sub-trees of 4 + 2 + 1 nodes, then layers 3-5 as 32 sub-trees; layers 0-2 pack their
        // remainder tiles and stay per-layer launches), set 1 one layer lower (layers 5-7 as 8 sub-trees).  A call takes the set whose
        // usable runs (sub_run_pays) cover more layers and, if equal, have more sub-trees (pick_sub_set): set 1 for 130 .. 512 rows,
        // where it is 2 us faster per call than set 0's four sub-trees (profiles/r05_subtree_call_times.txt).
        plan_subtree_set(end, 0);
        if (!sub_runs_.empty()) plan_subtree_set(end - 1, 1);
    }

    void plan_subtree_set(int end, int set) {
        while (end - 1 >= 2) {
            const int last = end - 1;
            const int k = stages_[last].n_nodes;
            int b = last + 1, act_blocks = 0, e_blocks = 0;
            std::vector<std::vector<int32_t>> members(kMaxTail), tabs(kMaxTail);      // by distance from `last`
            while (k >= 4 && b > (fuse01_ ? 2 : 1) && last + 1 - b < kMaxTail) {
                const HostStage& s = stages_[b - 1];
                if (s.kind != 0 || s.from_x || s.rem4 || s.pack_out || s.nf > kMaxFuncs || s.n_nodes % k) break;
                if (s.has_exp && s.mt1 * s.nf > 8) break;
                const int per = s.n_nodes / k;
                if (per * (s.has_exp ? std::max(s.mt1, s.mt2) : s.mt1) > 16) break;
                std::vector<int32_t> mem((size_t)k * per, -1), tab;
                if (b - 1 == last) {
                    for (int j = 0; j < k; ++j) mem[j] = j;
                } else {
                    // nodes of this layer under each root, in the order the layer above first reads them
                    const HostStage& up = stages_[b];
                    const std::vector<int32_t>& mup = members[last - b];
                    const int per_up = up.n_nodes / k;
                    std::vector<int32_t> owner(s.n_nodes, -1), pos(s.n_nodes, -1);
                    bool ok = s.nb_out == s.n_nodes * s.mto;
                    tab.assign(up.kb1tab.size(), 0);
                    for (int j = 0; j < k && ok; ++j) {
                        int have = 0;
                        for (int q = 0; q < per_up && ok; ++q) {
                            const int ni = mup[(size_t)j * per_up + q];
                            for (int kb = 0; kb < up.kb1 && ok; ++kb) {
                                const size_t at = ((size_t)ni * up.kb1 + kb) * 2;
                                const int src = up.kb1tab[at];
                                if (src < 0 || src >= s.nb_out) { ok = false; break; }
                                const int sn = src / s.mto;
                                if (owner[sn] < 0) {
                                    if (have == per) { ok = false; break; }
                                    owner[sn] = j;
                                    pos[sn] = have;
                                    mem[(size_t)j * per + have++] = sn;
                                } else if (owner[sn] != j) {
                                    ok = false;
                                    break;
                                }
                                tab[at] = pos[sn] * s.mto + src % s.mto;
                                tab[at + 1] = up.kb1tab[at + 1];
                            }
                        }
                        if (have != per) ok = false;
                    }
                    if (!ok) break;
                }
                const int act = b - 1 < last ? std::max(act_blocks, per * s.mto) : act_blocks;
                const int eb = std::max(e_blocks, s.has_exp ? per * s.nf * s.mt1 : 0);
                if (((size_t)2 * act + eb) * 1024 > 150 * 1024) break;
                act_blocks = act;
                e_blocks = eb;
                members[last - (b - 1)] = std::move(mem);
                if (b - 1 < last) tabs[last - b] = std::move(tab);      // the table of the layer above this one
                --b;
            }
            if (last + 1 - b < 2) {      // no run ends here: this layer stays a launch of its own
                end = last;
                continue;
            }
            SubRun r;
            r.begin = b;
            r.len = last + 1 - b;
            r.n = k;
            r.act_blocks = act_blocks;
            r.e_blocks = e_blocks;
            r.set = set;
            for (int i = b; i <= last; ++i) {
                r.nodes[i - b] = members[last - i];
                if (i > b) r.tab[i - b] = tabs[last - i];
            }
            sub_runs_.push_back(std::move(r));
            if (set != 0) {      // (the stage names describe set 0)
                stages_[b].name += "  [or, where more layers or more sub-trees can run that way: " + std::to_string(last + 1 - b) + " layers from here as " + std::to_string(k) + " sub-trees]";
                end = b;
                continue;
            }
            for (int i = b; i <= last; ++i) {
                stages_[i].name += i == b ? "  [batches of up to " + std::to_string(std::min(opt_.subtree_max_tiles, opt_.subtree_max_wgs / k) * 16) + " rows: this and the next " +
                                                std::to_string(last - b) + " layer(s) as " + std::to_string(k) + " sub-trees in ONE launch]"
                                          : "  [in the sub-tree launch for short batches]";
            }
            end = b;
        }
    }

#ifdef HIGSFA_DIAG
    // k_tail / k_subtree with wall-clock stamps (hg_fused_tail.hip: TAIL_STAMP): where a wave's time goes, layer by layer
    void tail_stamps_begin(TailParams& TP, size_t blocks, hipStream_t st) {
        stamp_buf_.alloc(blocks * 16 * 16 * 8);
        HG_HIP(hipMemsetAsync(stamp_buf_.p, 0, stamp_buf_.bytes, st));
        TP.stamps = (unsigned long long*)stamp_buf_.p;
    }
    void tail_stamps_report(const char* what, int si, int n_layers, size_t blocks, hipStream_t st) {
        HG_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(blocks * 16 * 16);
        HG_HIP(hipMemcpy(h.data(), stamp_buf_.p, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i < h.size(); i += 16)
            if (h[i]) t0 = std::min(t0, h[i]);
        static const char* names[5] = {"loads in", "expanded", "barrier 1", "outputs stored", "barrier 2"};
        double sum[16] = {}, mx[16] = {}, cnt[16] = {};
        for (size_t i = 0; i < h.size(); i += 16)
            for (int k = 0; k < 16; ++k)
                if (h[i + k]) {
                    const double us = (double)(h[i + k] - t0) * 0.01;
                    sum[k] += us; cnt[k] += 1; mx[k] = std::max(mx[k], us);
                }
        fprintf(stderr, "[stamp stage %d] %s, %zu workgroups: us after the first wave's entry, mean (max) over the waves that took part\n", si, what, blocks);
        fprintf(stderr, "[stamp stage %d]   entry %.2f (%.2f)\n", si, cnt[0] ? sum[0] / cnt[0] : 0.0, mx[0]);
        for (int l = 0; l < n_layers; ++l) {
            fprintf(stderr, "[stamp stage %d]   layer %d:", si, l);
            for (int k = 0; k < 5; ++k) {
                const int q = 1 + 5 * l + k;
                fprintf(stderr, "  %s %.2f (%.2f; %.0f waves)", names[k], cnt[q] ? sum[q] / cnt[q] : 0.0, mx[q], cnt[q]);
            }
            fprintf(stderr, "\n");
        }
    }
#endif

    static void fill_tail_stage(TailStage& S, HostStage& hs) {
        S.afrag = (const f32x4*)hs.d_afrag.p;
        S.bias = (const float*)hs.d_bias.p;
        S.kb1tab = (const int2*)hs.d_kb1tab.p;
        S.n_nodes = hs.n_nodes;
        S.kb1 = hs.kb1;
        S.nf = hs.nf;
        S.has_exp = hs.has_exp ? 1 : 0;
        S.node_blocks = hs.node_blocks;
        S.bias_floats = hs.bias_floats;
        S.nb_out = hs.nb_out;
        S.mto = hs.mto;
        S.mt1 = hs.mt1;
        S.mt2 = hs.mt2;
        for (int fi = 0; fi < hs.nf; ++fi) {
            S.funcp |= (uint32_t)hs.funcs[fi].kind << (4 * fi);
            S.expo[fi] = (float)hs.funcs[fi].expo;
            for (int mt1 = 0; mt1 < hs.mt1; ++mt1) S.nk2p[mt1] |= (uint32_t)hs.nk2[mt1][fi] << (4 * fi);
        }
    }

    TailParams subtree_params(const SubRun& sr, const f32x4* in, f32x4* out, int n_tiles) {
        const int sub_begin_ = sr.begin, sub_len_ = sr.len, sub_n_ = sr.n, sub_act_blocks_ = sr.act_blocks, sub_e_blocks_ = sr.e_blocks;
        TailParams TP{};
        TP.n_stages = sub_len_;
        for (int k = 0; k < sub_len_; ++k) {
            HostStage& hs = stages_[sub_begin_ + k];
            fill_tail_stage(TP.st[k], hs);
            TP.st[k].n_nodes /= sub_n_;
            TP.sub_nodes[k] = (const int32_t*)sr.d_nodes[k].p;
            if (k > 0) TP.st[k].kb1tab = (const int2*)sr.d_tab[k].p;
        }
        TP.in = in;
        TP.out_frag = out;
        TP.n_sub = sub_n_;
        TP.nb_out_frag = stages_[sub_begin_ + sub_len_ - 1].nb_out;
        TP.n_tiles = n_tiles;
        TP.nb_in = stages_[sub_begin_].nb_in;
        TP.act_blocks = sub_act_blocks_;
        TP.e_blocks = sub_e_blocks_;
        return TP;
    }

    // Three fused layers pay off from ~1400 rows on (call times against N, profiles/r03_call_times.txt: 16 waves per workgroup walk
    // the three layers' latencies one after the other — 23 us however small the batch, against 6 us for a k_stage_splitm launch of
    // the 4-node layer plus 13 us for the two layers above it); below that the launch starts one layer later.  Same bits either way.
    // A sub-tree launch pays while all its workgroups (sub-trees x batch tiles, each pulling its sub-tree's weights through ONE compute
    // unit's L1) are resident at once, one per CU: measured on U11L-128 (profiles/r05_subtree_call_times.txt) 8 sub-trees gain up to 512
    // rows = 256 workgroups and lose from 728; 32 sub-trees gain up to 44 rows and lose from 130.
    bool sub_run_pays(const SubRun& r, int n_tiles) const {
        return n_tiles <= opt_.subtree_max_tiles && (int64_t)r.n * n_tiles <= opt_.subtree_max_wgs;
    }

    // the set of runs a call of n_tiles uses: more layers inside usable runs first, then more sub-trees in the smallest of them
    int pick_sub_set(int n_tiles) const {
        int best = 0, best_cov = -1, best_k = 0;
        for (int set = 0; set < 2; ++set) {
            int cov = 0, mink = 0x7fffffff;
            for (const SubRun& r : sub_runs_)
                if (r.set == set && sub_run_pays(r, n_tiles)) {
                    cov += r.len;
                    mink = std::min(mink, r.n);
                }
            if (cov > best_cov || (cov == best_cov && cov > 0 && mink > best_k)) {
                best = set;
                best_cov = cov;
                best_k = mink;
            }
        }
        return best;
    }

    int tail_start(int n_tiles) const {
        const int ns = (int)stages_.size();
        bool short_batch = n_tiles < 96;
        const int sub_set = pick_sub_set(n_tiles);
        for (const SubRun& r : sub_runs_)      // (a sub-tree run that takes the top launch's first layer: only with HIGSFA_SUBTREE_WGS raised)
            if (r.set == sub_set && r.begin <= tail_begin_ && tail_begin_ < r.begin + r.len && sub_run_pays(r, n_tiles)) short_batch = true;
        return (ns - tail_begin_ >= 3 && short_batch) ? tail_begin_ + 1 : tail_begin_;
    }

    TailParams tail_params(int begin, const f32x4* in, int n_tiles, void* y, int y_dtype, int64_t y_cols, int64_t ldy, int64_t n) {
        TailParams TP{};
        const int ns = (int)stages_.size();
        TP.n_stages = ns - begin;
        for (int k = 0; k < TP.n_stages; ++k) {
            fill_tail_stage(TP.st[k], stages_[begin + k]);
        }
        TP.in = in;
        TP.y = y;
        TP.col_of = (const int32_t*)d_col_of_.p;
        TP.ldy = ldy;
        TP.n_rows = n;
        TP.y_cols = (int32_t)y_cols;
        TP.y_f64 = y_dtype == HG_F64 ? 1 : 0;
        if (y_dtype != HG_F32 && y_dtype != HG_F64) fail(HG_ERR_ARG, "output dtype must be f32 or f64");
        TP.n_tiles = n_tiles;
        TP.nb_in = stages_[begin].nb_in;
        TP.act_blocks = tail_act_blocks_;
        TP.e_blocks = tail_e_blocks_;
        return TP;
    }

    // Layers 0 and 1 can share one kernel when a wave's two layer-0 node slots are exactly the two
    // children of one layer-1 node (see k_stage01p).
    bool can_fuse01() const {
        if (stages_.size() < 2) return false;
        const HostStage& a = stages_[0];
        const HostStage& b = stages_[1];
        if (a.kind != 0 || b.kind != 0 || !a.from_x) return false;
        if (!(a.has_exp && a.mt1 == 1 && a.mt2 == 1 && a.kb1 == 1 && a.nf == 2 && a.contig4 && a.vec_ok)) return false;
        if (a.nk_last != 4 || a.nk2[0][0] != 4 || a.nk2[0][1] != 4) return false;   // the kernel runs all four k-steps unconditionally
        if (a.max_chunk_nodes > 16 || a.max_chunk_pieces > 64) return false;
        for (auto& c : a.chunks)
            if ((c.node_begin & 1) || (c.node_count & 1)) return false;
        if (!(b.has_exp && b.mt1 == 2 && b.mt2 == 2 && b.kb1 == 2 && b.nf == 2 && b.nk2[0][0] == 4 && b.nk2[0][1] == 4)) return false;
        if (b.n_nodes * 2 != a.n_nodes) return false;
        for (int n = 0; n < b.n_nodes; ++n)
            if (b.kb1tab[(size_t)n * 4] != 2 * n || b.kb1tab[(size_t)n * 4 + 2] != 2 * n + 1) return false;
        return true;
    }

    int s0_pos(int r, int g) const { return s0_transpose_ ? 4 * g + r : 4 * r + g; }

    FusedOptions opt_;
    int out_dim_;
    bool s0_transpose_ = false, fuse01_ = false;
    std::vector<HostStage> stages_;
    std::vector<int32_t> col_base_, col_of_;
    DevBuf d_col_base_, d_col_of_, bufA_, bufB_, stamp_buf_;
    int tail_begin_ = -1;         // first stage of the top-of-hierarchy launch (k_tail); -1: none
    int tail_act_blocks_ = 0, tail_e_blocks_ = 0;
    std::vector<SubRun> sub_runs_;      // k_subtree runs (short batches)
    WorkQueue wq_front_, wq_direct_, wq_direct_wg_;
    int32_t* err_host_ = nullptr;
    int32_t* err_dev_ = nullptr;
    int stamp_blocks_ = 0;
    std::map<const void*, size_t> lds_set_;
    std::map<std::tuple<const void*, int, size_t>, int> occ_;
    int max_nb_ = 0;
    int64_t padded_flops_ = 0, cap_rows_ = 0;
    int n_cus_ = 256;
};

}  // namespace

std::unique_ptr<Executor> make_fused_executor(const TNode& root, std::string* why_not) {
    std::vector<FStage> stages;
    std::string why;
    const FusedOptions opt = FusedOptions::from_env();
    if (!build_stages(root, stages, why, opt)) {
        if (why_not) *why_not = why;
        return nullptr;
    }
    for (auto& st : stages) {
        const FNode& f0 = st.nodes[0];
        bool table_driven = f0.has_clip;
        for (auto& n : st.nodes) table_driven = table_driven || n.has_prod;
        if (f0.funcs.size() > (size_t)kMaxFuncs && !table_driven) {
            if (why_not) *why_not = "more than 4 expansion functions";
            return nullptr;
        }
        for (auto& n : st.nodes) {
            bool same = n.has_exp == f0.has_exp && n.is_ig == f0.is_ig && n.funcs.size() == f0.funcs.size() && n.has_clip == f0.has_clip &&
                        (!n.has_clip || (n.clip_lo == f0.clip_lo && n.clip_hi == f0.clip_hi));
            for (size_t i = 0; same && i < n.funcs.size(); ++i)
                same = n.funcs[i].kind == f0.funcs[i].kind && n.funcs[i].expo == f0.funcs[i].expo && n.funcs[i].sel == f0.funcs[i].sel &&
                       n.funcs[i].k == f0.funcs[i].k;
            if (!same) {
                if (why_not) *why_not = "nodes of one layer use different expansions";
                return nullptr;
            }
        }
    }
    if (why_not) why_not->clear();
    try {
        return std::make_unique<FusedExecutor>(root, std::move(stages), opt);
    } catch (const Error& e) {      // a structure the fused kernels do not cover: generic plan instead
        if (why_not) *why_not = e.what();
        return nullptr;
    }
}

}  // namespace hg
