// Microbenchmark: can a SIMD of gfx950 run fp32 MFMAs of one wave and VALU work of ANOTHER wave at the same time?
// Workgroups of 512 threads (2 waves per SIMD at one workgroup per CU, 4 at two).  Waves 0-3 run 32 MFMAs per iteration,
// waves 4-7 256 v_fma_f32 (full rate) or 64 v_exp_f32 (quarter rate) + 64 v_mul_f32.  If the two
// overlapped, the time of the mixed run would be max(MFMA alone, VALU alone); if they share the issue port / data path it is
// the sum.   build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// mode bit 0: even waves do MFMAs; bit 1: odd waves do FMAs; bit 2: odd waves do v_exp
__global__ void __launch_bounds__(512) k(float* out, int iters, int mode, float seed) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (((wave >> 2) & 1) == 0) {      // waves w and w + 4 share a SIMD: one of them multiplies, the other does vector work
        if (mode & 1) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float a = seed, b = seed * 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = MFMA16(a, b, acc[i]);          // 32 MFMAs = 1024 pipe cycles
            }
            for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        if (mode & 2) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = seed + i;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);   // 256 FMAs = 1024 issue cycles
            }
            for (int i = 0; i < 8; ++i) r += v[i];
        }
        if (mode & 4) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = seed * 0.01f * (i + 1);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.25f;   // 64 v_exp (+ 64 v_mul) = 1024 + 256 cycles
            }
            for (int i = 0; i < 8; ++i) r += v[i];
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static float run(float* out, int per_cu, int mode) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, 256 * per_cu, 512, 0, 0, out, iters, mode, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, 256 * per_cu, 512, 0, 0, out, iters, mode, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 2 * 512 * 4);
    for (int per = 1; per <= 2; ++per) {
        const float m = run(out, per, 1), f = run(out, per, 2), t = run(out, per, 4), mf = run(out, per, 3), mt = run(out, per, 5);
        printf("%d waves per SIMD (half MFMA, half VALU):  MFMA alone %.3f ms | FMA alone %.3f | v_exp alone %.3f | MFMA + FMA %.3f (sum %.3f, max %.3f) | MFMA + v_exp %.3f (sum %.3f, max %.3f)\n",
               2 * per, m, f, t, mf, m + f, m > f ? m : f, mt, m + t, m > t ? m : t);
    }
    return 0;
}
