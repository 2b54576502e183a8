// Fused plan, iGSFA layers: gather pre-pass and the three-GEMM node kernel.
#include "hg_fused_dev.hpp"

namespace hg {
namespace fused {

// Row-major input -> fragment order (used in front of a first layer of iGSFA nodes, whose kernel
// reads fragment-order blocks like every later layer).  One wave per (batch tile, block): lane (g, j)
// gathers the four columns of its four k-steps for sub-image j.
template <typename XT>
__global__ void __launch_bounds__(256) k_im2frag(const XT* __restrict__ x, int64_t ldx, int64_t n_rows, int n_tiles, int nb,
                                                 const int32_t* __restrict__ gcol, f32x4* __restrict__ out, int vec4) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= (int64_t)n_tiles * nb) return;
    const int tile = (int)(wid / nb), blk = (int)(wid - (int64_t)tile * nb);
    const int64_t row = (int64_t)tile * 16 + j;
    const i32x4 c = *(const i32x4*)(gcol + (size_t)blk * 16 + g * 4);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < n_rows) {
        const XT* xr = x + row * ldx;
        if (vec4 && c[0] >= 0 && c[3] == c[0] + 3) {     // four contiguous, 16-byte aligned columns
            v = Vec4Load<XT>::ld(xr + c[0]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c[r] >= 0) v[r] = (float)xr[c[r]];
        }
    }
    out[(size_t)wid * 64 + lane] = v;
}

// A layer of iGSFA nodes.  Same workgroup structure as k_stage (node
// weights once into LDS, persistent sweep over tile groups), three chained GEMMs per node, all
// operands in registers:
//   x0[kb] = input fragments - mean                                   (K-blocks of the node input)
//   y[ms] += W1[fi][kb][ms] * f_fi(x0[kb])          s = scaled slow features (rows of y tiles < MS)
//   x0[kb] = x0[kb] + bias_r[kb] + W2[kb][ms] * y[ms]                 r = x0 - lr(s): the input fragment
//                                                                     IS the C operand (same layout)
//   y[mo] += W3[kb][mo] * x0[kb]                    q = pca(r) lands in the remaining rows of y
// Output tiles hold [s, q] in the caller's column order.
template <int MS, int MO, int T, int KBM>   // KBM: K-block capacity of a node input (2, 4, 6 or 8; <= 128 inputs)
__global__ void __launch_bounds__((KBM == 6 ? 768 : 512), (KBM <= 4 ? 4 : KBM <= 6 ? 3 : 2)) k_igsfa(StageParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup owns a group of `nodes_per_group` consecutive nodes (their weights fit LDS together)
    const int npg = P.nodes_per_group;
    const int grp_id = blockIdx.x % P.n_chunks, part = blockIdx.x / P.n_chunks;
    const int g0 = grp_id * npg, gn = min(npg, P.n_nodes - g0);
    float* sb = (float*)(smem + (size_t)npg * P.node_blocks * 64);
    int2* stab = (int2*)(sb + npg * P.bias_floats);
    {
        const f32x4* src = P.afrag + (size_t)g0 * P.node_blocks * 64;
        const int nvec = gn * P.node_blocks * 64;
        int i = tid;
        for (; i + 7 * nthr < nvec; i += 8 * nthr) {     // 8 x 16 B in flight per thread
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
    const int kb1 = P.kb1, nf = P.nf;
    // all nw waves copy; only the first nwt take batch tiles (layers of a few nodes: more, smaller
    // workgroups fill the chip, and a lone wave would copy its node's weights at a crawl)
    const int nwt = P.nodes_per_wg > 0 ? min(P.nodes_per_wg, nw) : nw;
    for (int grp = part; grp < P.tile_groups; grp += P.tile_parts) {
        int tile[T];
#pragma unroll
        for (int t = 0; t < T; ++t) tile[t] = (grp * nwt + wave) * T + t;
        if (wave >= nwt || tile[0] >= P.n_tiles) break;
      for (int ln = 0; ln < gn; ++ln) {
        const int node = g0 + ln;
        const f32x4* w1 = smem + (size_t)ln * P.node_blocks * 64 + lane;   // [fi][kb][ms]
        const f32x4* w2 = w1 + (size_t)nf * kb1 * MS * 64;              // [kb][ms]
        const f32x4* w3 = w2 + (size_t)kb1 * MS * 64;                   // [kb][mo]
        const float* by = sb + ln * P.bias_floats;                      // [MO][16]
        const float* br = by + MO * 16;                                 // [kb][16]
        const float* mu = br + kb1 * 16;                                // [kb][16]
        const int2* ktab = stab + ln * kb1;
        f32x4 x0[KBM][T];
        int nk1[KBM];
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb) {
            nk1[kb] = 0;
            if (kb < kb1) {
                const int2 e = ktab[kb];
                const int sbk = __builtin_amdgcn_readfirstlane(e.x);
                nk1[kb] = __builtin_amdgcn_readfirstlane(e.y);
                const f32x4 m = *(const f32x4*)(mu + kb * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int tl = tile[t] < P.n_tiles ? tile[t] : tile[0];
                    x0[kb][t] = P.in[((size_t)tl * P.nb_in + sbk) * 64 + lane] - m;
                }
            }
        }
        f32x4 y[MO][T];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
            const f32x4 bb = *(const f32x4*)(by + mo * 16 + g * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) y[mo][t] = bb;
        }
        // G1: slow features (folded form: all output features) from the expanded input.  The A fragments are
        // consumed in LDS order, one (function, block, tile) step after the other: the next one is always in
        // flight while the current one is multiplied (a lone wave per SIMD has nothing else to hide the read).
        {
            const f32x4* wq = w1;
            f32x4 a_nx = *wq;
            for (int fi = 0; fi < nf; ++fi) {
                const int fk = (P.funcp >> (4 * fi)) & 15;
                const float ex = P.expo[fi];
#pragma unroll
                for (int kb = 0; kb < KBM; ++kb) {
                    if (kb >= kb1) continue;
                    f32x4 e[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) e[t] = apply_func(fk, ex, x0[kb][t]);
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {      // only the tiles that hold slow features
                        const f32x4 a = a_nx;
                        wq += 64;
                        a_nx = *wq;                        // past the last step: the next block of this node's LDS image, unused
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r < nk1[kb]) {
#pragma unroll
                                for (int t = 0; t < T; ++t) y[ms][t] = MFMA16(a[r], e[t][r], y[ms][t]);
                            }
                    }
                }
            }
        }
        // G2: residual r = x0 - lr(s), accumulated into the input fragments
        if (P.ig_has_lr) {
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb) {
                if (kb >= kb1) continue;
                const f32x4 bb = *(const f32x4*)(br + kb * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < T; ++t) x0[kb][t] += bb;
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    const f32x4 a = w2[((size_t)kb * MS + ms) * 64];
                    const int nks = (P.nk2p[0] >> (4 * ms)) & 15;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < nks) {
#pragma unroll
                            for (int t = 0; t < T; ++t) x0[kb][t] = MFMA16(a[r], y[ms][t][r], x0[kb][t]);
                        }
                }
            }
        }
        // G3: q = pca(r) into the remaining rows of the output tiles (folded form: G1 already produced them)
        if (!P.ig_folded) {
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb) {
                if (kb >= kb1) continue;
                gemm_block<MO, T>(w3 + ((size_t)kb * MO) * 64, x0[kb], y, nk1[kb]);
            }
        }
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + (size_t)node * MO + mo) * 64 + lane] = y[mo][t];
      }
    }
}

// Folded iGSFA layer (hg_fused.hip, igsfa_affine): y = f(x - mean) W + c is a single GEMM from the expanded
// input fragments to the MO output tiles, and nothing needs the input fragments afterwards — so they are
// streamed: one K-block of the node input in registers at a time (the next one, possibly the first block of
// the next batch-tile group, already in flight), expanded under every function and multiplied into the
// output tiles at once.  ~100 VGPRs instead of 160-185: four waves per SIMD like k_stage.  Same workgroup
// structure and LDS image ([fi][kb][mo] fragments, biases, means, block table) as k_igsfa.
// FS: the expansion is (identity, |x|^p), known at compile time (as in k_stage / the front kernel): no function loop, no kind branches
template <int MO, int T, bool FS = false>
__global__ void __launch_bounds__(512, 4) k_igfold(StageParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int node = blockIdx.x % P.n_chunks, part = blockIdx.x / P.n_chunks;   // one node per workgroup
    float* sb = (float*)(smem + (size_t)P.node_blocks * 64);
    int2* stab = (int2*)(sb + P.bias_floats);
    {
        const f32x4* src = P.afrag + (size_t)node * P.node_blocks * 64;
        const int nvec = P.node_blocks * 64;
        int i = tid;
        for (; i + 7 * nthr < nvec; i += 8 * nthr) {     // 8 x 16 B in flight per thread
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i + u * nthr];
#pragma unroll
            for (int u = 0; u < 8; ++u) smem[i + u * nthr] = v[u];
        }
        for (; i < nvec; i += nthr) smem[i] = src[i];
        const float* bsrc = P.bias + (size_t)node * P.bias_floats;
        for (int k = tid; k < P.bias_floats; k += nthr) sb[k] = bsrc[k];
        const int2* tsrc = P.kb1tab + (size_t)node * P.kb1;
        for (int k = tid; k < P.kb1; k += nthr) stab[k] = tsrc[k];
    }
    __syncthreads();
    const int kb1 = P.kb1, nf = P.nf;
    const int nwt = P.nodes_per_wg > 0 ? min(P.nodes_per_wg, nw) : nw;
    if (wave >= nwt) return;
    const f32x4* w1 = smem + lane;                        // [fi][kb][mo]
    const float* by = sb;                                 // [MO][16]
    const float* mu = sb + MO * 16 + kb1 * 16;            // [kb][16]  (after the unused residual biases)
    int tile[T];
    uint32_t trow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        tile[t] = (part * nwt + wave) * T + t;
        trow[t] = (uint32_t)(tile[t] < P.n_tiles ? tile[t] : tile[0]) * (uint32_t)P.nb_in;
    }
    if (tile[0] >= P.n_tiles) return;
    f32x4 bf[T], bfn[T];
    int nk = __builtin_amdgcn_readfirstlane(stab[0].y);
    {
        const int sb0 = __builtin_amdgcn_readfirstlane(stab[0].x);
#pragma unroll
        for (int t = 0; t < T; ++t) bf[t] = P.in[(size_t)(trow[t] + sb0) * 64 + lane];
    }
    for (int grp = part; grp < P.tile_groups; grp += P.tile_parts) {
        const int tn0 = ((grp + P.tile_parts) * nwt + wave) * T;
        const bool has_next = grp + P.tile_parts < P.tile_groups && tn0 < P.n_tiles;
        uint32_t trow_nx[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tn = tn0 + t;
            trow_nx[t] = has_next ? (uint32_t)(tn < P.n_tiles ? tn : tn0) * (uint32_t)P.nb_in : trow[t];
        }
        f32x4 y[MO][T];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
            const f32x4 bb = *(const f32x4*)(by + mo * 16 + g * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) y[mo][t] = bb;
        }
        for (int kb = 0; kb < kb1; ++kb) {
            // next block of this tile group, or the first block of the next one
            const bool in_node = kb + 1 < kb1;
            const int2 kbn = stab[in_node ? kb + 1 : 0];
            const int sbn = __builtin_amdgcn_readfirstlane(kbn.x), nkn = __builtin_amdgcn_readfirstlane(kbn.y);
#pragma unroll
            for (int t = 0; t < T; ++t) bfn[t] = P.in[(size_t)((in_node ? trow[t] : trow_nx[t]) + sbn) * 64 + lane];
            const f32x4 m = *(const f32x4*)(mu + kb * 16 + g * 4);
            f32x4 x0[T];
#pragma unroll
            for (int t = 0; t < T; ++t) x0[t] = bf[t] - m;
            if constexpr (FS) {
                gemm_block<MO, T, true>(w1 + ((size_t)kb * MO) * 64, x0, y, nk);
                f32x4 e[T];
#pragma unroll
                for (int t = 0; t < T; ++t) e[t] = pow_abs4(x0[t], P.expo[1]);
                gemm_block<MO, T, true>(w1 + ((size_t)(kb1 + kb) * MO) * 64, e, y, nk);
            } else
            for (int fi = 0; fi < nf; ++fi) {
                f32x4 e[T];
                const int fk = (P.funcp >> (4 * fi)) & 15;
                const float ex = P.expo[fi];
#pragma unroll
                for (int t = 0; t < T; ++t) e[t] = apply_func(fk, ex, x0[t]);
                gemm_block<MO, T>(w1 + ((size_t)(fi * kb1 + kb) * MO) * 64, e, y, nk);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) bf[t] = bfn[t];
            nk = nkn;
        }
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + (size_t)node * MO + mo) * 64 + lane] = y[mo][t];
        if (!has_next) break;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            tile[t] = tn0 + t;
            trow[t] = trow_nx[t];
        }
    }
}

template <int MS, int MO>
static StageFn pick_igsfa_t(int T, int kb1) {
    // the input fragments of a node stay in registers through all three GEMMs: the block capacity sets the
    // register count and with it the waves per SIMD (4 / 4 / 3 / 2)
    if (kb1 <= 2) return T == 2 ? (StageFn)k_igsfa<MS, MO, 2, 2> : (StageFn)k_igsfa<MS, MO, 1, 2>;
    if (kb1 <= 4) return T == 2 ? (StageFn)k_igsfa<MS, MO, 2, 4> : (StageFn)k_igsfa<MS, MO, 1, 4>;
    if (kb1 <= 6) return T == 2 ? (StageFn)k_igsfa<MS, MO, 2, 6> : (StageFn)k_igsfa<MS, MO, 1, 6>;
    return T == 2 ? (StageFn)k_igsfa<MS, MO, 2, 8> : (StageFn)k_igsfa<MS, MO, 1, 8>;
}
// Folded iGSFA layers with FEW nodes (the top of the hierarchy): not enough (node, tile) pairs to fill the
// chip with whole nodes, and copying a node's 64 KiB of weights into LDS per workgroup is all latency.  Like
// k_stage_splitm: a workgroup shares ONE node and T batch tiles; wave w < kb1 loads input block w, subtracts
// the mean and writes the expanded fragments to LDS; after one barrier wave w < MO computes output tile w
// from all of them, its A fragments straight from L2 (each read by exactly one wave of the workgroup, all
// requested before the barrier).  nf <= 2, kb1 <= 8.
template <int T>
__global__ void __launch_bounds__(512) k_igfold_split(StageParams P, int mo_n) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];      // [fi][kb][t][64]
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int node = blockIdx.x % P.n_nodes, grp = blockIdx.x / P.n_nodes;
    const int kb1 = P.kb1, nf = P.nf;
    int tile[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tile[t] = grp * T + t;
    const f32x4* wnode = P.afrag + (size_t)node * P.node_blocks * 64 + lane;      // [fi][kb][mo]
    const float* bnode = P.bias + (size_t)node * P.bias_floats;                    // [MO][16] | [kb][16] unused | [kb][16] means
    const int2* kt = P.kb1tab + (size_t)node * kb1;
    // output-tile weights: 2 x 8 fragments at most, requested before anything waits
    f32x4 a[2][8];
    if (w < mo_n) {
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int kb = 0; kb < 8; ++kb)
                if (fi < nf && kb < kb1) a[fi][kb] = wnode[((size_t)(fi * kb1 + kb) * mo_n + w) * 64];
    }
    if (w < kb1) {
        const int2 e = kt[w];
        const f32x4 m = *(const f32x4*)(bnode + mo_n * 16 + kb1 * 16 + w * 16 + g * 4);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tl = tile[t] < P.n_tiles ? tile[t] : tile[0];
            const f32x4 x0 = P.in[((size_t)tl * P.nb_in + e.x) * 64 + lane] - m;
            for (int fi = 0; fi < nf; ++fi)
                smem[((fi * kb1 + w) * T + t) * 64 + lane] = apply_func((P.funcp >> (4 * fi)) & 15, P.expo[fi], x0);
        }
    }
    __syncthreads();
    if (w >= mo_n) return;
    f32x4 y[T];
    {
        const f32x4 bb = *(const f32x4*)(bnode + w * 16 + g * 4);
#pragma unroll
        for (int t = 0; t < T; ++t) y[t] = bb;
    }
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (fi >= nf || kb >= kb1) continue;
            const int nk = kt[kb].y;
            f32x4 e[T];
#pragma unroll
            for (int t = 0; t < T; ++t) e[t] = smem[((fi * kb1 + kb) * T + t) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nk) {
#pragma unroll
                    for (int t = 0; t < T; ++t) y[t] = MFMA16(a[fi][kb][r], e[t][r], y[t]);
                }
        }
#pragma unroll
    for (int t = 0; t < T; ++t)
        if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + (size_t)node * mo_n + w) * 64 + lane] = y[t];
}

void launch_igfold_split(const StageParams& P, int mo, int n_tiles, hipStream_t st) {
    const int T = n_tiles >= 2 ? 2 : 1;
    const int groups = (n_tiles + T - 1) / T;
    const int nwv = P.kb1 > mo ? P.kb1 : mo;
    const size_t lds = (size_t)P.nf * P.kb1 * T * 1024;
    if (T == 2)
        hipLaunchKernelGGL(k_igfold_split<2>, (unsigned)(groups * P.n_nodes), nwv * 64, lds, st, P, mo);
    else
        hipLaunchKernelGGL(k_igfold_split<1>, (unsigned)(groups * P.n_nodes), nwv * 64, lds, st, P, mo);
}

StageFn pick_igfold(int mo, int T, bool fs) {
    if (fs && mo == 4) return T == 2 ? (StageFn)k_igfold<4, 2, true> : (StageFn)k_igfold<4, 1, true>;
    if (fs && mo == 3) return T == 2 ? (StageFn)k_igfold<3, 2, true> : (StageFn)k_igfold<3, 1, true>;
    if (fs && mo == 2) return T == 2 ? (StageFn)k_igfold<2, 2, true> : (StageFn)k_igfold<2, 1, true>;
    switch (mo) {
        case 1: return T == 2 ? (StageFn)k_igfold<1, 2> : (StageFn)k_igfold<1, 1>;
        case 2: return T == 2 ? (StageFn)k_igfold<2, 2> : (StageFn)k_igfold<2, 1>;
        case 3: return T == 2 ? (StageFn)k_igfold<3, 2> : (StageFn)k_igfold<3, 1>;
        default: return T == 2 ? (StageFn)k_igfold<4, 2> : (StageFn)k_igfold<4, 1>;
    }
}

StageFn pick_igsfa(int ms, int mo, int T, int kb1) {   // ms <= mo (the slow features are a prefix of the output)
    switch (ms * 10 + mo) {
        case 11: return pick_igsfa_t<1, 1>(T, kb1);
        case 12: return pick_igsfa_t<1, 2>(T, kb1);
        case 13: return pick_igsfa_t<1, 3>(T, kb1);
        case 14: return pick_igsfa_t<1, 4>(T, kb1);
        case 22: return pick_igsfa_t<2, 2>(T, kb1);
        case 23: return pick_igsfa_t<2, 3>(T, kb1);
        case 24: return pick_igsfa_t<2, 4>(T, kb1);
        case 33: return pick_igsfa_t<3, 3>(T, kb1);
        case 34: return pick_igsfa_t<3, 4>(T, kb1);
        default: return pick_igsfa_t<4, 4>(T, kb1);
    }
}

void launch_im2frag(const void* x, int x_dtype, int64_t ldx, int64_t n_rows, int n_tiles, int nb, const int32_t* gcol, f32x4* out,
                    int vec4, hipStream_t st) {
    const int64_t waves = (int64_t)n_tiles * nb;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (x_dtype == HG_U8)
        hipLaunchKernelGGL(k_im2frag<uint8_t>, grid, 256, 0, st, (const uint8_t*)x, ldx, n_rows, n_tiles, nb, gcol, out, vec4);
    else if (x_dtype == HG_F32)
        hipLaunchKernelGGL(k_im2frag<float>, grid, 256, 0, st, (const float*)x, ldx, n_rows, n_tiles, nb, gcol, out, vec4);
    else
        hipLaunchKernelGGL(k_im2frag<double>, grid, 256, 0, st, (const double*)x, ldx, n_rows, n_tiles, nb, gcol, out, vec4);
}

}  // namespace fused
}  // namespace hg
