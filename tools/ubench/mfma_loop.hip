// Microbenchmark: what does the inner loop shape of k_stage cost on gfx950?
// build: hipcc -O3 --offload-arch=gfx950 mfma_loop.hip -o mfma_loop ; run: ./mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
// lane-group reduce-scatter of a 4x4-form accumulator (hg_fused_dev.hpp rem4_total)
__device__ __forceinline__ float rem4_total(f32x4 d) {
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[0]), __float_as_uint(d[1]), false, false);
    const float t = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[2]), __float_as_uint(d[3]), false, false);
    const float u = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(u), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The same reduce-scatter through LDS (round 5): the wave writes its four partial-sum registers lane-major (one ds_write_b128) and every lane reads
// the register of ITS lane group from the four lane groups (four conflict-free ds_read_b32), then (v0 + v1) + (v2 + v3) — the operands and the
// association of rem4_total, hence the same bits — with three vector adds instead of three permlane swaps + three adds on the port the MFMAs need.
__device__ __forceinline__ float rem4_total_lds(f32x4 d, float* scratch, int lane) {
    *(f32x4*)(scratch + lane * 4) = d;
    const int g = lane >> 4, j = lane & 15;
    const float v0 = scratch[(0 * 16 + j) * 4 + g], v1 = scratch[(1 * 16 + j) * 4 + g], v2 = scratch[(2 * 16 + j) * 4 + g], v3 = scratch[(3 * 16 + j) * 4 + g];
    return (v0 + v1) + (v2 + v3);
}

// MT = 4 m-tiles, T = 2 batch tiles -> 8 accumulators, 32 MFMAs per "K-block" iteration
template <int V>
__global__ void __launch_bounds__(512) k(const f32x4* __restrict__ w, const int2* __restrict__ tab, float* out, int iters, int nk_in) {
    extern __shared__ f32x4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += blockDim.x) smem[i] = w[i];     // 64 KiB of "weights"
    int2* st = (int2*)(smem + 4096);
    if (tid < 64) st[tid] = tab[tid];
    __syncthreads();
    f32x4 acc[4][2];
    for (int m = 0; m < 4; ++m) for (int t = 0; t < 2; ++t) acc[m][t] = f32x4{0, 0, 0, 0};
    f32x4 bf[2] = {w[lane], w[64 + lane]};
    f32x4 a[4], an[4];
    f32x4 d4[3][2];
    for (int q = 0; q < 3; ++q) for (int t = 0; t < 2; ++t) d4[q][t] = f32x4{0, 0, 0, 0};
    const f32x4* wl = smem + lane;
    if (V == 2) for (int m = 0; m < 4; ++m) a[m] = wl[m * 64];
    int nk = nk_in;
    for (int it = 0; it < iters; ++it) {
        const int kb = it & 15;
        if (V == 0) {
            for (int m = 0; m < 4; ++m) a[m] = bf[m & 1];
        } else if (V == 1 || V == 3 || V == 4) {
            if (V == 4) {
                int2 e = st[kb];
                nk = __builtin_amdgcn_readfirstlane(e.y);
            }
            for (int m = 0; m < 4; ++m) a[m] = wl[(kb * 4 + m) * 64];
        } else if (V == 2) {
            const int kn = (it + 1) & 15;
            for (int m = 0; m < 4; ++m) an[m] = wl[(kn * 4 + m) * 64];
        }
        if (V == 6 || V == 7 || V == 8) {
            // Round 5: a 60-row affine as 3 full m-tiles on the 16x16x4 form + its 12-row tail as three 4-row
            // groups on v_mfma_f32_4x4x1 (8 cycles each instead of a fourth 32-cycle tile), the tail's A values read from a compact
            // LDS image (16 distinct 16-byte words per group: lanes with the same (g, i) read the same word) and the three
            // reduce-scatters every `visit` K-blocks (V6; V7 leaves them out to price them).  Same "useful" rows as V4.
            int2 e = st[kb];
            nk = __builtin_amdgcn_readfirstlane(e.y);
            for (int m = 0; m < 3; ++m) a[m] = wl[(kb * 4 + m) * 64];
            f32x4 a4[3];
            const f32x4* w4 = smem + (kb * 4 + 3) * 64 + ((lane >> 4) * 4 + (lane & 3));
            for (int q = 0; q < 3; ++q) a4[q] = w4[q * 16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r >= nk) continue;
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = MFMA16(a[m][r], bf[t][r], acc[m][t]);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int t = 0; t < 2; ++t) d4[q][t] = MFMA4(a4[q][r], bf[t][r], d4[q][t]);
            }
            if ((V == 6 || V == 8) && (it % 12) == 11) {
                float* scratch = (float*)(smem + 4096 + 64) + (tid >> 6) * 256;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        acc[3][t][q] += V == 8 ? rem4_total_lds(d4[q][t], scratch, lane) : rem4_total(d4[q][t]);
                        d4[q][t] = f32x4{0, 0, 0, 0};
                    }
                }
            }
            continue;
        }
        if (V == 5) {      // the order gemm_block used until round 2: m-tile outer (one A fragment live), k-step inner
            int2 e = st[kb];
            nk = __builtin_amdgcn_readfirstlane(e.y);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 am = wl[(kb * 4 + m) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r >= nk) continue;
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = MFMA16(am[r], bf[t][r], acc[m][t]);
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if ((V == 3 || V == 4) && r >= nk) continue;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = MFMA16(a[m][r], bf[t][r], acc[m][t]);
        }
        if (V == 2) for (int m = 0; m < 4; ++m) a[m] = an[m];
    }
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int m = 0; m < 4; ++m) for (int t = 0; t < 2; ++t) s += acc[m][t];
    for (int q = 0; q < 3; ++q) for (int t = 0; t < 2; ++t) s += d4[q][t];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
}

template <int V>
void run(const char* name, const f32x4* w, const int2* tab, float* out, int threads, int blocks_per_cu) {
    const int iters = 4096;
    const int blocks = 256 * blocks_per_cu;
    size_t lds = 4096 * 16 + 64 * 16 + 8 * 1024;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, blocks, threads, lds, 0, w, tab, out, iters, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, blocks, threads, lds, 0, w, tab, out, iters, 4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * 32.0 * 2048.0;
    printf("%-34s %4d thr x %d/CU: %7.3f ms  %6.1f TFLOP/s (%.0f%% of 157.3)\n", name, threads, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main(int argc, char** argv) {
    printf("operands: %s\n", argc > 1 ? "random in [-0.01, 0.01)" : "constant 0.001");
    f32x4* w; int2* tab; float* out;
    hipMalloc(&w, 4096 * 16 + 4096); hipMalloc(&tab, 64 * 8); hipMalloc(&out, 256 * 4 * 1024 * 4);
    std::vector<float> hw(4096 * 4 + 1024, 0.001f);
    if (argc > 1) { unsigned st = 12345u; for (auto& v : hw) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 65536.0f * 0.02f - 0.01f; } }
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    std::vector<int2> ht(64, int2{3, 4}); hipMemcpy(tab, ht.data(), 64 * 8, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 3; ++cfg) {
        int thr = cfg == 0 ? 256 : 512, per = cfg == 2 ? 2 : 1;   // 1, 2, 4 waves per SIMD
        run<0>("V0 registers only", w, tab, out, thr, per);
        run<1>("V1 + A frags from LDS (just in time)", w, tab, out, thr, per);
        run<2>("V2 + A frags from LDS (prefetched)", w, tab, out, thr, per);
        run<3>("V3 V1 + nk branches", w, tab, out, thr, per);
        run<4>("V4 V3 + LDS table/readfirstlane", w, tab, out, thr, per);
        run<5>("V5 V4 with m-tile outer, k-step inner", w, tab, out, thr, per);
        run<6>("V6 V4, 12-row tail as 3 x 4x4x1 + reduce", w, tab, out, thr, per);
        run<7>("V7 V6 without the reduce-scatters", w, tab, out, thr, per);
        run<8>("V8 V6, reduce-scatters through LDS", w, tab, out, thr, per);
    }
    return 0;
}
