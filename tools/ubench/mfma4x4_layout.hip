// Pins down the operand / result lane layout of v_mfma_f32_4x4x1_16B_f32 on gfx950 (one wave).
// For every source lane p: A = [lane == p], B = 1 -> which (lane, reg) of D light up; same for B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        f32x4 c = {0, 0, 0, 0};
        f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(l == p ? 1.f : 0.f, 1.f, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[((0 * 64 + p) * 64 + l) * 4 + r] = d[r];
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, l == p ? 1.f : 0.f, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[((1 * 64 + p) * 64 + l) * 4 + r] = d[r];
    }
}
int main() {
    float* d;
    hipMalloc(&d, 2 * 64 * 64 * 4 * 4);
    hipLaunchKernelGGL(probe, 1, 64, 0, 0, d);
    std::vector<float> h(2 * 64 * 64 * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w)
        for (int p = 0; p < 64; p += (p < 8 ? 1 : 7)) {
            printf("%s lane %2d ->", w ? "B" : "A", p);
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r)
                    if (h[((w * 64 + p) * 64 + l) * 4 + r] != 0.f) printf(" (l%d,r%d)", l, r);
            printf("\n");
        }
    return 0;
}
