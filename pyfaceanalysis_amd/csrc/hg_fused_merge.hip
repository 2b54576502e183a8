// k_stage_mg: several consecutive middle layers of the hierarchy in ONE launch (round 5; MergeParams in hg_fused_dev.hpp).
//
// Why.  rocprofv3 shows a step's launches back to back, yet two independent batches in flight run 8-10 % faster than one after the
// other: what they fill is INSIDE the launches — every k_stage workgroup starts by copying its node's 52-64 KiB of weights into LDS
// (5 us with nothing on the matrix pipe), and the SIMD arbiter's oldest-first rule lets the workgroup that reached a CU first run
// ahead, so a launch's last quarter runs at reduced occupancy.  A timing experiment that let consecutive k_stage
// launches run side by side on two streams without waiting for one another (HIGSFA_WHATIF_OVERLAP: wrong results, times only) took
// 13-18 us off the 505 us step with two cross-stream hand-offs still in it.  Here the layers share one grid: layer s's workgroups
// sit behind layer s - 1's in block order, are dispatched as those drain, copy their weights while the others still multiply, and
// wait — per 16-tile group, on a counter in device memory — only for the activation blocks they are about to read.
//
// The arithmetic of a layer is k_stage's compile-time-expansion path (k_stage<MT1, MT2, 2, .., FS>: (identity, |x|^p) expansion, two
// tiles per wave, eight waves, A fragments of the first affine read one K-block ahead into alternating register sets): the same
// products in the same order, hence the same bits as the per-layer launches (tests: HIGSFA_NO_MERGE=1 against the default).
//
// Dependencies.  Block order guarantees progress: every block of layer s - 1 has a lower index than any block of layer s, each XCD
// dispatches its share of the grid in index order, and a resident workgroup waits only for blocks of lower index — so the unfinished
// block of lowest index is always resident or next in line on an XCD whose resident blocks have all finished.  Visibility across the
// eight L2s: producers store write-through (sc1) and wait for their stores before one lane adds 1 to the group's counter
// (device-scope atomic); consumers poll the counter (device-scope atomic load) and read activations with sc1 loads; no buffer is
// written twice inside a launch.  (The recipe of round 2's persistent chain, which ran the top of the hierarchy this way.)
#include <hip/hip_runtime.h>

#include "hg_fused_dev.hpp"

namespace hg {
namespace fused {

namespace {

constexpr int kSpinLimit = 1 << 16;      // x s_sleep(127) (~4 us): far beyond a launch's duration; a bug must not hang the GPU
constexpr int kCtrPad = 64;              // counters 256 bytes apart: one per cache line and memory channel, not sixteen in one line

template <int MT1, int MT2>
__global__ void __launch_bounds__(512, 4) k_stage_mg(MergeParams M) {
    constexpr int T = 2;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    int s = 0;
    while (s + 1 < M.n_stages && (int)blockIdx.x >= M.blk_end[s]) ++s;
    const StageParams& P = M.st[s];
    const int blk = (int)blockIdx.x - (s ? M.blk_end[s - 1] : 0);      // (layers start at multiples of 8: blk & 7 is the XCD)
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blk & 7, kq = blk >> 3;
    const int cpx = (P.n_chunks + 7) >> 3;
    const int chunk = xcd + 8 * (kq % cpx), part = kq / cpx;
    if (chunk >= P.n_chunks) return;      // padding block of the XCD-aware grid: owns nothing, signals nothing
    const bool consumer = s > 0, producer = s + 1 < M.n_stages;
    uint32_t* const done_out = M.done + (size_t)s * M.done_stride;
    const uint32_t* const done_in = M.done + (size_t)(consumer ? s - 1 : 0) * M.done_stride;
    const uint32_t want_in = consumer ? M.want[s - 1] : 0u;
    const int g0 = chunk * P.nodes_per_wg;
    const int gn = min(P.nodes_per_group, P.n_nodes - g0);
    float* sb = (float*)(smem + (size_t)P.nodes_per_group * P.node_blocks * 64);
    int2* stab = (int2*)(sb + P.nodes_per_group * P.bias_floats);
    {   // the node group's weights -> LDS (as k_stage); independent of the layer below: runs while that layer still computes
        const f32x4* src = P.afrag + (size_t)g0 * P.node_blocks * 64;
        const int nvec = gn * P.node_blocks * 64;
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

    auto signal_group = [&](int grp) {      // this wave's tiles of group grp are in memory (the caller has waited for its stores)
        if (producer && lane == 0) __hip_atomic_fetch_add(done_out + (size_t)grp * kCtrPad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wait_group = [&](int grp) {        // every producer wave of group grp has signalled
        if (!consumer) return;
        int spins = 0;
        while ((int32_t)(__hip_atomic_load(done_in + (size_t)grp * kCtrPad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want_in) < 0) {
            if (++spins > kSpinLimit) {
                if (lane == 0) *M.err = 4;
                break;
            }
            // a waiting wave asks again every ~4 us: thousands of waves of the next layer may be resident and waiting while this layer's
            // last groups are in work, and their polls travel to memory (device scope) through the same fabric as the activations
            __builtin_amdgcn_s_sleep(127);
        }
        // the poll is a relaxed atomic: nothing stops the COMPILER from moving the activation loads below in front of the loop (the
        // hardware issues in order and the loop's exit needs the counter's value) — this does
        asm volatile("" ::: "memory");
#ifdef HG_MG_ACQUIRE_FENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // diagnostic: the documented agent-scope acquire (buffer_inv sc1)
#endif
    };
    int tile[T];
    uint32_t trow[T], trow_nx[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tile[t] = (part * nw + wave) * T + t;
    if (tile[0] >= P.n_tiles) {      // a wave without tiles still belongs to the producers its consumers count
        for (int grp = part; grp < P.tile_groups; grp += P.tile_parts) signal_group(grp);
        return;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) trow[t] = (uint32_t)(tile[t] < P.n_tiles ? tile[t] : tile[0]) * (uint32_t)P.nb_in;
    // input / output through buffer resources (sc1 accesses; byte offsets fit 32 bits: checked by the host)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, 0, (int)((uint32_t)P.n_tiles * (uint32_t)P.nb_in * 1024u), kBufferFlags);
    const Sc1Store store{__builtin_amdgcn_make_buffer_rsrc((void*)P.out, 0, (int)((uint32_t)P.n_tiles * (uint32_t)P.nb_out * 1024u), kBufferFlags)};
    const uint32_t loff = (uint32_t)lane * 16u;
#ifndef HG_MG_LOAD_AUX
#define HG_MG_LOAD_AUX 16
#endif
    auto ldb = [&](uint32_t blk_index) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, loff, blk_index * 1024u, HG_MG_LOAD_AUX)); };

    wait_group(part);
    f32x4 bf[T], bfn[T];
    int nk;
    {
        const int2 kb = stab[0];
        const int sb0 = __builtin_amdgcn_readfirstlane(kb.x);
        nk = __builtin_amdgcn_readfirstlane(kb.y);
#pragma unroll
        for (int t = 0; t < T; ++t) bf[t] = ldb(trow[t] + (uint32_t)sb0);
    }
    int grp = part;
    for (;; grp += P.tile_parts) {
        // rows of the tile group after this one (or this one again when it is the last); its first block is prefetched during this
        // group's last node, so the group must be complete below BEFORE this one is worked on
        const int tn0 = ((grp + P.tile_parts) * nw + wave) * T;
        const bool has_next = grp + P.tile_parts < P.tile_groups && tn0 < P.n_tiles;
        if (has_next) wait_group(grp + P.tile_parts);
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
                const f32x4 bb = *(const f32x4*)(b1 + mt * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < T; ++t) z[mt][t] = bb;
            }
            f32x4 a0[MT1], a1[MT1];
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) a0[mt] = wA1[mt * 64];
            auto kstep = [&](int kbi, const f32x4 (&ac)[MT1], f32x4 (&an)[MT1]) {
                const bool in_node = kbi + 1 < P.kb1;
                const bool in_group = in_node || ln + 1 < gn;
                const int2 kbn = in_node ? kt[kbi + 1] : (ln + 1 < gn ? kt[P.kb1] : stab[0]);
                const int sbn = __builtin_amdgcn_readfirstlane(kbn.x);
                const int nkn = __builtin_amdgcn_readfirstlane(kbn.y);
#pragma unroll
                for (int t = 0; t < T; ++t) bfn[t] = ldb((in_group ? trow[t] : trow_nx[t]) + (uint32_t)sbn);
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
            node_tail<MT1, MT2, T, false, true>(P, wA2, b1 + MT1 * 16, g0 + ln, z, tile, lane, store);
        }
        if (producer) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this group's stores are written through (and the prefetched block has arrived)
#ifdef HG_MG_RELEASE_FENCE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // diagnostic: the documented agent-scope release (buffer_wbl2 sc1 + wait)
#endif
            signal_group(grp);
        }
        if (!has_next) break;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            tile[t] = tn0 + t;
            trow[t] = trow_nx[t];
        }
    }
    for (grp += P.tile_parts; grp < P.tile_groups; grp += P.tile_parts) signal_group(grp);      // groups of this workgroup in which this wave has no tile
}

}  // namespace

const void* stage_merged_fn(int mt1, int mt2) {
    if (mt1 == 4 && mt2 == 4) return (const void*)k_stage_mg<4, 4>;
    return nullptr;
}

void launch_stage_merged(const MergeParams& M, int mt1, int mt2, unsigned blocks, size_t lds_bytes, hipStream_t st) {
    if (mt1 == 4 && mt2 == 4) hipLaunchKernelGGL((k_stage_mg<4, 4>), blocks, 512, lds_bytes, st, M);
    else fail(HG_ERR_STATE, "internal: no merged instantiation for %d x %d tiles", mt1, mt2);
}

}  // namespace fused
}  // namespace hg
