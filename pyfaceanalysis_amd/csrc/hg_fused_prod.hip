// Fused layer kernel for nodes whose expansion contains cross-column products (cuicuilco.nonlinear_expansion QT,
// pair_prodsadj*_ex, sel_exp(k, QT) — module aliases FaceDetectUpdated.py:57,62; every face / eye network of the
// reference is a "Non-Linear ... 11 Layer Network", Pipelines/Pipeline_experimental.txt:7) or a CutoffNode between the
// expansion and the second affine.  Same plan as k_stage (hg_fused.hip): a workgroup keeps a node group's weights in LDS
// and sweeps its share of the batch tiles; GEMM 1 is identical.  The second half differs: a product z_i * z_k needs two
// FEATURES of one sub-image, which sit in different lane groups / registers of the accumulator tile.  So GEMM 2 runs in two
// parts: the element-wise functions of the expansion exactly as in k_stage (the z accumulators are the B operand, one
// K-block per (z tile, function)), then the product columns, 16 per K-block (k-step r / lane group g <- product column
// 16 b + 4 r + g), read as two words of a per-wave LDS image zs[feature][sub-image] of the z tiles through a table of byte
// offsets.  The table is stage-uniform (built for the widest node; narrower nodes have zero weights on the columns they
// lack).  A CutoffNode (numpy.clip) applies to every expanded value.
#include "hg_fused_dev.hpp"

namespace hg {
namespace fused {

namespace {

__device__ __forceinline__ float clip_keep_nan(float v, float lo, float hi) {      // numpy.clip: a NaN stays a NaN
    const float c = fminf(fmaxf(v, lo), hi);
    return v != v ? v : c;
}

template <int MT1, int MT2, int T>
__global__ void __launch_bounds__(512, 2) k_stage_prod(StageParams P) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, nw = nthr >> 6, g = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int chunk = xcd + 8 * (kq / P.tile_parts), part = kq % P.tile_parts;
    if (chunk >= P.n_chunks) return;
    const int n_begin = chunk * P.nodes_per_wg;
    const int n_end = min(n_begin + P.nodes_per_wg, P.n_nodes);
    const int npg = P.nodes_per_group;
    float* sb = (float*)(smem + (size_t)npg * P.node_blocks * 64);
    int2* stab = (int2*)(sb + npg * P.bias_floats);
    int2* etab = stab + npg * P.kb1;                                   // [neb][16] {byte offset of z_i | valid << 31, byte offset of z_k}
    const char* zs = (const char*)(etab + P.neb * 16) + (size_t)wave * T * MT1 * 1024 + j * 4;   // per wave: [t][feature][sub-image]
    for (int k = tid; k < P.neb * 16; k += nthr) etab[k] = P.etab[k];
    const int nfe = P.nf;                     // element-wise functions
    const float ex0 = P.expo[0], ex1 = P.expo[1], ex2 = P.expo[2], ex3 = P.expo[3];
    for (int g0 = n_begin; g0 < n_end; g0 += npg) {
        const int gn = min(npg, n_end - g0);
        __syncthreads();
        {
            const f32x4* src = P.afrag + (size_t)g0 * P.node_blocks * 64;
            const int nvec = gn * P.node_blocks * 64;
            int i = tid;
            for (; i + 3 * nthr < nvec; i += 4 * nthr) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = src[i + u * nthr];
#pragma unroll
                for (int u = 0; u < 4; ++u) smem[i + u * nthr] = v[u];
            }
            for (; i < nvec; i += nthr) smem[i] = src[i];
            const float* bsrc = P.bias + (size_t)g0 * P.bias_floats;
            for (int k = tid; k < gn * P.bias_floats; k += nthr) sb[k] = bsrc[k];
            const int2* tsrc = P.kb1tab + (size_t)g0 * P.kb1;
            for (int k = tid; k < gn * P.kb1; k += nthr) stab[k] = tsrc[k];
        }
        __syncthreads();
        for (int grp = part; grp < P.tile_groups; grp += P.tile_parts) {
            int tile[T];
            uint32_t trow[T];
#pragma unroll
            for (int t = 0; t < T; ++t) tile[t] = (grp * nw + wave) * T + t;
            if (tile[0] >= P.n_tiles) break;
#pragma unroll
            for (int t = 0; t < T; ++t) trow[t] = (uint32_t)(tile[t] < P.n_tiles ? tile[t] : tile[0]) * (uint32_t)P.nb_in;
            for (int ln = 0; ln < gn; ++ln) {
                const f32x4* wA1 = smem + (size_t)ln * P.node_blocks * 64 + lane;
                const f32x4* wA2 = wA1 + P.kb1 * MT1 * 64;
                const float* b1 = sb + ln * P.bias_floats;
                const int2* kt = stab + ln * P.kb1;
                f32x4 z[MT1][T];
#pragma unroll
                for (int mt = 0; mt < MT1; ++mt) {
                    const f32x4 bb = *(const f32x4*)(b1 + mt * 16 + g * 4);
#pragma unroll
                    for (int t = 0; t < T; ++t) z[mt][t] = bb;
                }
                f32x4 bf[T];
                {
                    const int sb0 = __builtin_amdgcn_readfirstlane(kt[0].x);
#pragma unroll
                    for (int t = 0; t < T; ++t) bf[t] = P.in[(size_t)(trow[t] + sb0) * 64 + lane];
                }
                for (int kbi = 0; kbi < P.kb1; ++kbi) {
                    const int nk = __builtin_amdgcn_readfirstlane(kt[kbi].y);
                    f32x4 bfn[T];
                    const int sbn = __builtin_amdgcn_readfirstlane(kt[kbi + 1 < P.kb1 ? kbi + 1 : kbi].x);
#pragma unroll
                    for (int t = 0; t < T; ++t) bfn[t] = P.in[(size_t)(trow[t] + sbn) * 64 + lane];
                    gemm_block<MT1, T>(wA1 + kbi * MT1 * 64, bf, z, nk);
#pragma unroll
                    for (int t = 0; t < T; ++t) bf[t] = bfn[t];
                }
                // z -> LDS image [feature][sub-image] (the wave's own region; a wave's LDS operations complete in order)
                if (P.neb > 0) {
                    float* zw = (float*)zs;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) zw[(t * MT1 * 16 + mt * 16 + 4 * r + g) * 16] = z[mt][t][r];
                }
                f32x4 y[MT2][T];
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) {
                    const f32x4 bb = *(const f32x4*)(b1 + MT1 * 16 + mt * 16 + g * 4);
#pragma unroll
                    for (int t = 0; t < T; ++t) y[mt][t] = bb;
                }
                // part 1: element-wise functions, from the accumulators (as node_tail in hg_fused_dev.hpp)
#pragma unroll
                for (int mt1 = 0; mt1 < MT1; ++mt1) {
                    const uint32_t nkp = P.nk2p[mt1];
                    for (int fi = 0; fi < nfe; ++fi) {
                        const int nk = (nkp >> (4 * fi)) & 15;
                        if (nk == 0) continue;
                        const int fk = (P.funcp >> (4 * fi)) & 15;
                        const float ex = fi == 0 ? ex0 : (fi == 1 ? ex1 : (fi == 2 ? ex2 : ex3));
                        f32x4 e[T];
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            e[t] = apply_func(fk, ex, z[mt1][t]);
                            if (P.has_clip) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) e[t][r] = clip_keep_nan(e[t][r], P.clip_lo, P.clip_hi);
                            }
                        }
                        gemm_block<MT2, T>(wA2 + (mt1 * nfe + fi) * MT2 * 64, e, y, nk);
                    }
                }
                // part 2: product columns, from the LDS image
                const f32x4* wP = wA2 + MT1 * nfe * MT2 * 64;
                for (int eb = 0; eb < P.neb; ++eb) {
                    const int nk = eb + 1 < P.neb ? 4 : P.nk_last;
                    f32x4 e[T];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int2 en = etab[eb * 16 + r * 4 + g];
                        const bool valid = en.x < 0;
                        const int oa = en.x & 0x7fffffff, ob = en.y;
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            const float a = *(const float*)(zs + t * MT1 * 1024 + oa), b = *(const float*)(zs + t * MT1 * 1024 + ob);
                            float v = valid ? a * b : 0.f;
                            if (P.has_clip) v = clip_keep_nan(v, P.clip_lo, P.clip_hi);
                            e[t][r] = v;
                        }
                    }
                    gemm_block<MT2, T>(wP + eb * MT2 * 64, e, y, nk);
                }
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        if (tile[t] < P.n_tiles) P.out[((size_t)tile[t] * P.nb_out + (g0 + ln) * P.mto + mt) * 64 + lane] = y[mt][t];
            }
        }
    }
}

template <int MT1, int MT2>
StageFn pick_t(int T) {
    return T == 2 ? (StageFn)k_stage_prod<MT1, MT2, 2> : (StageFn)k_stage_prod<MT1, MT2, 1>;
}
template <int MT1>
StageFn pick_m2(int mt2, int T) {
    switch (mt2) {
        case 1: return pick_t<MT1, 1>(T);
        case 2: return pick_t<MT1, 2>(T);
        case 3: return pick_t<MT1, 3>(T);
        default: return pick_t<MT1, 4>(T);
    }
}

}  // namespace

StageFn pick_prod(int mt1, int mt2, int T) {
    switch (mt1) {
        case 1: return pick_m2<1>(mt2, T);
        case 2: return pick_m2<2>(mt2, T);
        case 3: return pick_m2<3>(mt2, T);
        default: return pick_m2<4>(mt2, T);
    }
}

}  // namespace fused
}  // namespace hg
