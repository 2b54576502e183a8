// The top of the hierarchy in ONE launch.  The last layers of a flow have few nodes (U11L: 16, 8, 4, 2, 1): one launch per
// layer cannot fill the chip with whole nodes and pays a launch boundary, a weight fetch and a drain for 1-15 us of work
// (layers 6-10: 94 us at N = 4096 for 28 us of MFMA time; ~60 us at any small N, SURVEY.md §6: a real frame calls with N <= 728).
//
// Here every (layer, node) pair owns `slices` persistent workgroups that keep the node's weights in LDS for the whole
// launch and walk the batch in tile groups; a child layer hands a finished tile group to its parents through memory with a
// flag per (layer, group, node), so all layers of the chain run concurrently as a pipeline and the workgroups of the
// upper layers load their weights while the lower ones still compute.
//
//   work split     workgroup = 2 teams x NW waves (NW = max(MT1, MT2)); a team takes one batch tile per pass; wave m of a
//                  team computes m-tile m of GEMM 1, the expanded tiles are exchanged through LDS, wave m computes m-tile m
//                  of GEMM 2 (the split of k_stage_splitm: a tile's latency is 1 / NW of a whole-node wave's).  Each
//                  accumulator sees its products in the same order as in k_stage / k_stage_splitm: bit-identical results.
//   hand-off       the write-through form of cdna_hip_programming.md Guideline 16 (R1) / MI355X_MICROARCH.md "Valid forms",
//                  first table row — no release or acquire fence (the fence form measured 10.6 us per layer here, this
//                  one see profiles/): EVERY activation store of the kernel is a 16-byte `sc1` buffer store and EVERY
//                  activation load a 16-byte `sc1` buffer load to registers; producer: every wave `s_waitcnt vmcnt(0)`,
//                  workgroup barrier, ONE lane stores the launch's generation number into flags[layer][group][node]
//                  (relaxed agent atomic = `sc1` store); consumer: one wave polls the <= 16 flags of the group (relaxed agent
//                  loads, `s_sleep` between polls, BOUNDED: a time-out sets an error word and goes on, so the grid always
//                  drains), then the workgroup barrier, then the loads.  One workgroup per CU, hipMalloc'ed buffers.
//                  Generation numbers grow from launch to launch: no counter is ever reset, stale flags never match.
//   residency      grid = sum(nodes) x slices <= the number of CUs (one 80 KiB workgroup per CU); workgroups are numbered
//                  layer by layer, so under the observed in-order dispatch a resident consumer's producers are resident or
//                  done; correctness never depends on it (bounded spins).
//   buffers        every layer of the chain writes its own region (no ping-pong: a parent must not overwrite what a slower
//                  sibling of its child still reads).
#include "hg_fused_dev.hpp"

namespace hg {
namespace fused {

namespace {

constexpr int kSpinLimit = 1 << 22;      // polls of ~0.3 us: a stuck hand-off gives up after about a second

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16-byte write-through (sc1) accesses: aux = 16 (cdna_hip_programming.md Guideline 16, R1)
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}

template <int MT1, int MT2>
__global__ void __launch_bounds__(128 * (MT1 > MT2 ? MT1 : MT2), 1) k_chain(ChainParams C) {
    constexpr int NW = MT1 > MT2 ? MT1 : MT2;      // waves per team
    constexpr int KBM = 8;                         // K-blocks a node may have
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave / NW, m = wave % NW;
    int c = 0;
    while (c + 1 < C.n_stages && (int)blockIdx.x >= C.st[c + 1].wg_begin) ++c;
    const ChainStage& S = C.st[c];
    const int idx = blockIdx.x - S.wg_begin, node = idx / C.slices, slice = idx % C.slices;
    // LDS: weights | bias | K-block table | exchange tiles [team][fi][mt1]
    float* sb = (float*)(smem + (size_t)S.node_blocks * 64);
    int2* stab = (int2*)(sb + S.bias_floats);
    f32x4* zs = (f32x4*)(stab + KBM) + (size_t)team * S.nf * MT1 * 64;
    {
        const f32x4* src = S.afrag + (size_t)node * S.node_blocks * 64;
        const int nvec = S.node_blocks * 64, nthr = blockDim.x;
        int i = tid;
        for (; i + 7 * nthr < nvec; i += 8 * nthr) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i + u * nthr];
#pragma unroll
            for (int u = 0; u < 8; ++u) smem[i + u * nthr] = v[u];
        }
        for (; i < nvec; i += nthr) smem[i] = src[i];
        const float* bsrc = S.bias + (size_t)node * S.bias_floats;
        for (int k = tid; k < S.bias_floats; k += nthr) sb[k] = bsrc[k];
        const int2* tsrc = S.kb1tab + (size_t)node * S.kb1;
        for (int k = tid; k < KBM; k += nthr) stab[k] = k < S.kb1 ? tsrc[k] : int2{tsrc[0].x, 0};
    }
    __syncthreads();
    const f32x4* wA1 = smem + lane;
    const f32x4* wA2 = wA1 + S.kb1 * MT1 * 64;
    const int nf = S.nf;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)S.in, 0, C.n_tiles * S.nb_in * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)S.out, 0, C.n_tiles * S.nb_out * 1024, 0x00020000);
    const int passes = C.tiles_per_group / 2;
    for (int grp = slice; grp < C.n_groups; grp += C.slices) {
        if (c > 0) {     // wait for every node of the layer below to have published this group
            if (wave == 0) {
                const ChainStage& Pv = C.st[c - 1];
                const uint32_t* f = C.flags + ((size_t)(c - 1) * C.n_groups + grp) * 16;
                int spins = 0;
                for (;;) {
                    const uint32_t v = lane < Pv.n_nodes ? __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : C.gen;
                    if (__ballot(v != C.gen) == 0ull) break;
                    if (++spins > kSpinLimit) {
                        if (lane == 0) *C.err = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps the loads below the poll
            }
            __syncthreads();
        }
        for (int p = 0; p < passes; ++p) {
            const int tile = grp * C.tiles_per_group + p * 2 + team;
            const bool live = tile < C.n_tiles;                       // wave-uniform; barriers below are reached either way
            const uint32_t trow = (uint32_t)(live ? tile : 0) * (uint32_t)S.nb_in;
            f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live && m < MT1) {
                f32x4 bq[KBM];
#pragma unroll
                for (int kb = 0; kb < KBM; ++kb)
                    if (kb < S.kb1) bq[kb] = ld_sc1(rin, (trow + (uint32_t)__builtin_amdgcn_readfirstlane(stab[kb].x)) * 1024u + lane * 16u);
                z = *(const f32x4*)(sb + m * 16 + g * 4);
#pragma unroll
                for (int kb = 0; kb < KBM; ++kb) {
                    if (kb >= S.kb1) break;
                    const int nk = __builtin_amdgcn_readfirstlane(stab[kb].y);
                    const f32x4 a = wA1[(kb * MT1 + m) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < nk) z = MFMA16(a[r], bq[kb][r], z);
                }
                if (!S.has_exp) {
                    st_sc1(rout, ((uint32_t)tile * S.nb_out + node * S.mto + m) * 1024u + lane * 16u, z);
                } else {
                    for (int fi = 0; fi < nf; ++fi) zs[(fi * MT1 + m) * 64 + lane] = apply_func_uniform((S.funcp >> (4 * fi)) & 15, S.expo[fi], z);
                }
            }
            if (S.has_exp) {
                __syncthreads();
                if (live && m < MT2) {
                    f32x4 y = *(const f32x4*)(sb + (MT1 + m) * 16 + g * 4);
#pragma unroll
                    for (int mt1 = 0; mt1 < MT1; ++mt1) {
                        const uint32_t nkp = S.nk2p[mt1];
                        for (int fi = 0; fi < nf; ++fi) {
                            const int nk = (nkp >> (4 * fi)) & 15;
                            if (nk == 0) continue;
                            const f32x4 e = zs[(fi * MT1 + mt1) * 64 + lane];
                            const f32x4 a = wA2[((mt1 * nf + fi) * MT2 + m) * 64];
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (r < nk) y = MFMA16(a[r], e[r], y);
                        }
                    }
                    st_sc1(rout, ((uint32_t)tile * S.nb_out + node * S.mto + m) * 1024u + lane * 16u, y);
                }
                __syncthreads();      // the exchange tiles are rewritten by the next pass
            }
        }
        if (c + 1 < C.n_stages) {    // publish the group
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(C.flags + ((size_t)c * C.n_groups + grp) * 16 + node, C.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

typedef void (*ChainFn)(ChainParams);

ChainFn pick(int mt1, int mt2) {
    if (mt1 == 4 && mt2 == 4) return k_chain<4, 4>;
    if (mt1 == 3 && mt2 == 3) return k_chain<3, 3>;
    if (mt1 == 2 && mt2 == 2) return k_chain<2, 2>;
    if (mt1 == 1 && mt2 == 1) return k_chain<1, 1>;
    if (mt1 == 4 && mt2 == 2) return k_chain<4, 2>;
    if (mt1 == 2 && mt2 == 1) return k_chain<2, 1>;
    return nullptr;
}

}  // namespace

bool chain_supported(int mt1, int mt2) { return pick(mt1, mt2) != nullptr; }

size_t chain_lds_bytes(int node_blocks, int bias_floats, int mt1, int nf) {
    return (size_t)node_blocks * 1024 + (size_t)bias_floats * 4 + 8 * 8 + (size_t)2 * nf * mt1 * 1024;
}

void launch_chain(const ChainParams& C, int mt1, int mt2, int grid, size_t lds, hipStream_t st) {
    ChainFn fn = pick(mt1, mt2);
    if (!fn) fail(HG_ERR_STATE, "internal: chain kernel shape");
    static thread_local const void* raised = nullptr;
    if (lds > 64 * 1024 && raised != (const void*)fn) {
        HG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = (const void*)fn;
    }
    const int nw = std::max(mt1, mt2);
    hipLaunchKernelGGL(fn, (unsigned)grid, 128 * nw, lds, st, C);
}

}  // namespace fused
}  // namespace hg
