// Blob ("HGSFAFL1") -> flow tree.  Format: pyfaceanalysis_amd/blob.py.  Every read is bounds
// checked and every index / width a later stage derives from the blob (switchboard connections,
// expansion selections and offsets) is validated here: a malformed blob yields HG_ERR_FORMAT /
// HG_ERR_DIM, never an out-of-range access on the host or on the device.
#include <cmath>

#include "hg_common.hpp"

namespace hg {

namespace {

struct Reader {
    const uint8_t* base;
    size_t size, pos = 0;
    int depth = 0;

    const uint8_t* take(size_t n) {
        if (n > size || pos > size - n) fail(HG_ERR_FORMAT, "blob truncated at byte %zu (+%zu)", pos, n);
        const uint8_t* p = base + pos;
        pos += n;
        pos += (8 - (pos & 7)) & 7;
        return p;
    }
    uint32_t u32(const uint8_t* p) {
        uint32_t v;
        memcpy(&v, p, 4);
        return v;
    }
    void f64(std::vector<double>& out, size_t count) {
        if (count > (size - pos) / 8 + 1) fail(HG_ERR_FORMAT, "blob: array of %zu doubles exceeds blob", count);
        const uint8_t* p = take(count * 8);
        out.resize(count);
        if (count) memcpy(out.data(), p, count * 8);
    }
};

constexpr uint32_t kMaxDim = 1u << 24;

// The execution plans know ONE form of an iGSFA node: s = sfa(e) * magn (per column), r = x0 - lr(s).
// The other variants of the blob record are rewritten into it here, in float64 (n = normalised slow
// features, M = the scaling as a matrix, diag(magn) or the stored one):
//   scaling matrix:   s = n M = (e - a) (W M) + b M            -> sfa.W <- W M, sfa.b <- b M, magn <- 1
//   lr on unscaled n: lr(n) = (s M^-1 - a_l) W_l + b_l = (s - a_l M)(M^-1 W_l) + b_l
//                                                              -> lr.a <- a_l M, lr.W <- M^-1 W_l
// (M must be invertible for the second; a scaling that is not cannot be expressed and is rejected).
void normalise_igsfa(TNode& n, bool lr_unscaled, bool scale_matrix) {
    const size_t S = n.sfa->out_dim, E = n.sfa->in_dim, d = n.in_dim;
    std::vector<double> M;   // S x S row-major, only when needed
    if (scale_matrix || lr_unscaled) {
        if (scale_matrix) M = n.magn;
        else {
            M.assign(S * S, 0.0);
            for (size_t j = 0; j < S; ++j) M[j * S + j] = n.magn[j];
        }
        for (double v : M)
            if (!std::isfinite(v)) fail(HG_ERR_FORMAT, "igsfa: non-finite scaling");
    }
    if (lr_unscaled) {
        TNode& L = *n.lr;
        // a' = a_l M
        std::vector<double> a2(S, 0.0);
        for (size_t i = 0; i < S; ++i)
            for (size_t j = 0; j < S; ++j) a2[j] += L.a[i] * M[i * S + j];
        // W' = M^-1 W_l: solve M X = W_l by Gaussian elimination with partial pivoting
        std::vector<double> A = M, X = L.W;   // X: S x d
        for (size_t c = 0; c < S; ++c) {
            size_t piv = c;
            for (size_t rr = c + 1; rr < S; ++rr)
                if (std::fabs(A[rr * S + c]) > std::fabs(A[piv * S + c])) piv = rr;
            if (!(std::fabs(A[piv * S + c]) > 1e-300))
                fail(HG_ERR_FORMAT, "igsfa: the slow-feature scaling is singular, so a linear reconstruction from the unscaled "
                                    "features cannot be expressed");
            if (piv != c) {
                for (size_t j = 0; j < S; ++j) std::swap(A[piv * S + j], A[c * S + j]);
                for (size_t j = 0; j < d; ++j) std::swap(X[piv * d + j], X[c * d + j]);
            }
            const double inv = 1.0 / A[c * S + c];
            for (size_t rr = 0; rr < S; ++rr) {
                if (rr == c) continue;
                const double f = A[rr * S + c] * inv;
                if (f == 0.0) continue;
                for (size_t j = c; j < S; ++j) A[rr * S + j] -= f * A[c * S + j];
                for (size_t j = 0; j < d; ++j) X[rr * d + j] -= f * X[c * d + j];
            }
        }
        for (size_t c = 0; c < S; ++c) {
            const double inv = 1.0 / A[c * S + c];
            for (size_t j = 0; j < d; ++j) X[c * d + j] *= inv;
        }
        L.a.swap(a2);
        L.W.swap(X);
    }
    if (scale_matrix) {
        TNode& F = *n.sfa;
        std::vector<double> W2(E * S, 0.0), b2(S, 0.0);
        for (size_t e = 0; e < E; ++e)
            for (size_t i = 0; i < S; ++i) {
                const double w = F.W[e * S + i];
                if (w == 0.0) continue;
                for (size_t j = 0; j < S; ++j) W2[e * S + j] += w * M[i * S + j];
            }
        for (size_t i = 0; i < S; ++i)
            for (size_t j = 0; j < S; ++j) b2[j] += F.b[i] * M[i * S + j];
        F.W.swap(W2);
        F.b.swap(b2);
        n.magn.assign(S, 1.0);
    }
}

std::unique_ptr<TNode> read_node(Reader& r) {
    if (++r.depth > 64) fail(HG_ERR_FORMAT, "blob: nesting deeper than 64");
    const uint8_t* h = r.take(16);
    auto n = std::make_unique<TNode>();
    n->kind = r.u32(h);
    n->in_dim = r.u32(h + 4);
    n->out_dim = r.u32(h + 8);
    n->aux = r.u32(h + 12);
    if (n->in_dim == 0 || n->out_dim == 0 || n->in_dim > kMaxDim || n->out_dim > kMaxDim)
        fail(HG_ERR_FORMAT, "blob: node kind %u has invalid dims %u -> %u", n->kind, n->in_dim, n->out_dim);
    switch (n->kind) {
        case K_FLOW:
        case K_FLOWNODE: {
            if (n->aux == 0) fail(HG_ERR_FORMAT, "blob: empty flow");
            uint32_t cur = n->in_dim;
            for (uint32_t i = 0; i < n->aux; ++i) {
                auto c = read_node(r);
                if (c->in_dim != cur)
                    fail(HG_ERR_DIM, "flow node %u (%s) expects input_dim %u but receives %u", i,
                         kind_name(c->kind), c->in_dim, cur);
                cur = c->out_dim;
                n->ch.push_back(std::move(c));
            }
            if (cur != n->out_dim) fail(HG_ERR_DIM, "flow output_dim %u != last node's %u", n->out_dim, cur);
            break;
        }
        case K_LAYER: {
            if (n->aux == 0) fail(HG_ERR_FORMAT, "blob: empty layer");
            uint64_t si = 0, so = 0;
            for (uint32_t i = 0; i < n->aux; ++i) {
                auto c = read_node(r);
                si += c->in_dim;
                so += c->out_dim;
                n->ch.push_back(std::move(c));
            }
            if (si != n->in_dim || so != n->out_dim) fail(HG_ERR_DIM, "layer dims do not equal the sums over its nodes");
            break;
        }
        case K_CLONELAYER: {
            if (n->aux == 0) fail(HG_ERR_FORMAT, "blob: clone layer with 0 copies");
            auto c = read_node(r);
            if ((uint64_t)c->in_dim * n->aux != n->in_dim || (uint64_t)c->out_dim * n->aux != n->out_dim)
                fail(HG_ERR_DIM, "clone layer dims mismatch");
            n->ch.push_back(std::move(c));
            break;
        }
        case K_SWITCHBOARD: {
            if (n->aux != n->out_dim) fail(HG_ERR_FORMAT, "switchboard: connection count != output_dim");
            const uint8_t* p = r.take((size_t)n->aux * 4);
            n->conn.resize(n->aux);
            memcpy(n->conn.data(), p, (size_t)n->aux * 4);
            for (int32_t c : n->conn)
                if (c < 0 || (uint32_t)c >= n->in_dim) fail(HG_ERR_FORMAT, "switchboard: connection %d out of range", c);
            break;
        }
        case K_AFFINE: {
            if (n->aux > 5) fail(HG_ERR_FORMAT, "affine: unknown subtype %u", n->aux);
            r.f64(n->a, n->in_dim);
            r.f64(n->W, (size_t)n->in_dim * n->out_dim);
            r.f64(n->b, n->out_dim);
            break;
        }
        case K_EXPANSION: {
            if (n->aux == 0 || n->aux > 64) fail(HG_ERR_FORMAT, "expansion: bad function count %u", n->aux);
            uint64_t total = 0;
            for (uint32_t i = 0; i < n->aux; ++i) {
                const uint8_t* p = r.take(24);
                ExpFunc f;
                f.kind = r.u32(p);
                f.sel = r.u32(p + 4);
                f.k = r.u32(p + 8);
                memcpy(&f.expo, p + 16, 8);
                if (f.kind > E_PAIR_BAND) fail(HG_ERR_FORMAT, "expansion: unknown function kind %u", f.kind);
                // sel = number of leading columns the function reads (0 = all; more than there are = all, as numpy slicing
                // clamps); k = pair distance
                const uint64_t u = (f.sel && f.sel < n->in_dim) ? f.sel : n->in_dim;
                if ((f.kind == E_ABS_POW || f.kind == E_SIGNED_POW) && !(std::isfinite(f.expo) && f.expo > 0.0))
                    fail(HG_ERR_FORMAT, "expansion: function %u has exponent %g (must be finite and > 0)", i, f.expo);
                if (f.kind == E_PAIR_ADJ && (f.k == 0 || f.k >= u))
                    fail(HG_ERR_FORMAT, "expansion: pair distance %u outside 1..%llu", f.k, (unsigned long long)(u - 1));
                // widths in 64 bits: u (u + 1) / 2 overflows int from u = 46341
                if (f.kind == E_PAIR_BAND && (f.k == 0 || f.k > u))
                    fail(HG_ERR_FORMAT, "expansion: band of %u offsets outside 1..%llu", f.k, (unsigned long long)u);
                const uint64_t w = f.kind <= E_SIGNED_POW ? u : f.kind == E_QUADRATIC ? u * (u + 1) / 2
                                   : f.kind == E_PAIR_BAND ? (uint64_t)f.k * u - (uint64_t)f.k * (f.k - 1) / 2 : u - f.k;
                if (w == 0 || w > kMaxDim) fail(HG_ERR_FORMAT, "expansion: function %u is %llu columns wide", i, (unsigned long long)w);
                if (w != (uint64_t)f.out_dim((int)n->in_dim)) fail(HG_ERR_FORMAT, "internal: expansion width");
                total += w;
                n->funcs.push_back(f);
            }
            if (total != n->out_dim) fail(HG_ERR_DIM, "expansion: output_dim %u != sum of function widths %llu", n->out_dim,
                                          (unsigned long long)total);
            break;
        }
        case K_IGSFA: {
            const uint8_t* p = r.take(8);
            const uint32_t has_exp = r.u32(p), flags = r.u32(p + 4);
            // flags: bit0 lr_node present, bit1 lr_node reads the UNSCALED slow features, bit2 the scaling is a matrix
            if (flags & ~7u) fail(HG_ERR_FORMAT, "igsfa: unknown flags 0x%x", flags);
            const bool has_lr = flags & 1u, lr_unscaled = flags & 2u, scale_matrix = flags & 4u;
            r.f64(n->x_mean, n->in_dim);
            uint32_t e_dim = n->in_dim;
            if (has_exp) {
                n->exp_node = read_node(r);
                if (n->exp_node->kind != K_EXPANSION || n->exp_node->in_dim != n->in_dim)
                    fail(HG_ERR_FORMAT, "igsfa: exp_node must be an EXPANSION of the node input");
                e_dim = n->exp_node->out_dim;
            }
            n->sfa = read_node(r);
            if (n->sfa->kind != K_AFFINE || n->sfa->in_dim != e_dim) fail(HG_ERR_DIM, "igsfa: sfa_node dims mismatch");
            const size_t S = n->sfa->out_dim;
            if (scale_matrix && S > 4096) fail(HG_ERR_FORMAT, "igsfa: scaling matrix of %zu x %zu", S, S);
            r.f64(n->magn, scale_matrix ? S * S : S);
            if (has_lr) {
                n->lr = read_node(r);
                if (n->lr->kind != K_AFFINE || n->lr->in_dim != n->sfa->out_dim || n->lr->out_dim != n->in_dim)
                    fail(HG_ERR_DIM, "igsfa: lr_node dims mismatch");
            }
            n->pca = read_node(r);
            if (n->pca->kind != K_AFFINE || n->pca->in_dim != n->in_dim) fail(HG_ERR_DIM, "igsfa: pca_node dims mismatch");
            if (n->aux > n->sfa->out_dim || n->aux + n->pca->out_dim != n->out_dim)
                fail(HG_ERR_DIM, "igsfa: output_dim != num_sfa_features_preserved + pca output_dim");
            normalise_igsfa(*n, has_lr && lr_unscaled, scale_matrix);
            break;
        }
        case K_IDENTITY:
            if (n->in_dim != n->out_dim) fail(HG_ERR_DIM, "identity: dims differ");
            break;
        case K_HEAD:
            if (n->out_dim > n->in_dim) fail(HG_ERR_DIM, "head: output_dim > input_dim");
            break;
        case K_CUTOFF: {
            if (n->in_dim != n->out_dim) fail(HG_ERR_DIM, "cutoff: dims differ");
            const uint8_t* p = r.take(16);
            memcpy(&n->lo, p, 8);
            memcpy(&n->hi, p + 8, 8);
            break;
        }
        default:
            fail(HG_ERR_FORMAT, "blob: unknown node kind %u", n->kind);
    }
    --r.depth;
    return n;
}

}  // namespace

const char* kind_name(uint32_t kind) {
    static const char* names[] = {"?", "Flow", "Switchboard", "Layer", "CloneLayer", "Affine", "GeneralExpansion",
                                  "iGSFA", "Identity", "Head", "Cutoff", "FlowNode"};
    return kind <= K_FLOWNODE ? names[kind] : "?";
}

std::unique_ptr<TNode> parse_blob(const void* blob, size_t nbytes) {
    if (!blob) fail(HG_ERR_ARG, "null blob");
    if (nbytes < 24 + 16) fail(HG_ERR_FORMAT, "blob too small (%zu bytes)", nbytes);
    const uint8_t* p = (const uint8_t*)blob;
    if (memcmp(p, "HGSFAFL1", 8) != 0) fail(HG_ERR_FORMAT, "bad magic (not a HGSFAFL1 flow blob)");
    uint32_t version;
    uint64_t total;
    memcpy(&version, p + 8, 4);
    memcpy(&total, p + 16, 8);
    if (version != 1) fail(HG_ERR_FORMAT, "unsupported blob version %u", version);
    if (total != nbytes) fail(HG_ERR_FORMAT, "blob size field %llu != buffer size %zu", (unsigned long long)total, nbytes);
    Reader r{p, nbytes, 24};
    auto root = read_node(r);
    if (root->kind != K_FLOW) fail(HG_ERR_FORMAT, "root record is not a FLOW");
    return root;
}

int64_t tree_flops(const TNode& n) {
    int64_t t = 0;
    switch (n.kind) {
        case K_AFFINE: return 2ll * n.in_dim * n.out_dim;
        case K_CLONELAYER: return (int64_t)n.aux * tree_flops(*n.ch[0]);
        case K_IGSFA:
            t = tree_flops(*n.sfa) + tree_flops(*n.pca);
            if (n.lr) t += tree_flops(*n.lr);
            return t;
        default:
            for (auto& c : n.ch) t += tree_flops(*c);
            return t;
    }
}

}  // namespace hg
