// Host-side packing for the ndarray-in / ndarray-out call (hg_flow_execute, FaceDetectUpdated.py:699).
// Plain C++ (compiled by g++, not as HIP): AVX2 bodies chosen at run time, scalar fallback.
//
// Rows of the caller's matrix that hold only integers 0..255 — what images_asarray returns for mode "L" images
// (face_analysis.py:786) — travel as uint8: 1/8 (float64) or 1/4 (float32) of the PCIe bytes.  The device widens
// uint8 -> float exactly and rounds float64 -> float exactly for these values, so the result is bit-identical to
// sending the wide type.  The functions return false as soon as a block holds anything else (fractions, negatives,
// values above 255, NaN / Inf): the caller then sends the row block in its own type.
#include <immintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstring>

namespace hg {

namespace {

template <typename T>
inline bool narrow_scalar(const T* src, uint8_t* dst, int64_t n) {
    int bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const T v = src[i];
        const T c = v >= (T)0 ? (v <= (T)255 ? v : (T)256) : (T)256;     // NaN and out-of-range values -> 256 (never equal to v)
        const int iv = (int)c;
        bad |= (iv >> 8) | ((T)iv != v);
        dst[i] = (uint8_t)iv;
    }
    return bad == 0;
}

constexpr int64_t kBlock = 2048;      // elements checked between two early-out tests

// cvtt of NaN / out-of-range gives 0x80000000, which converts back to a value != v: one equality test covers everything
// except integers outside 0..255, which the OR of all converted values catches.
__attribute__((target("avx2"))) bool narrow_f64_avx2(const double* src, uint8_t* dst, int64_t n) {
    int64_t i = 0;
    while (i + 16 <= n) {
        const int64_t stop = (n - i < kBlock ? n : i + kBlock) - 15;
        __m256d eq = _mm256_castsi256_pd(_mm256_set1_epi32(-1));
        __m128i ors = _mm_setzero_si128();
        for (; i < stop; i += 16) {
            const __m256d v0 = _mm256_loadu_pd(src + i), v1 = _mm256_loadu_pd(src + i + 4), v2 = _mm256_loadu_pd(src + i + 8),
                          v3 = _mm256_loadu_pd(src + i + 12);
            const __m128i i0 = _mm256_cvttpd_epi32(v0), i1 = _mm256_cvttpd_epi32(v1), i2 = _mm256_cvttpd_epi32(v2), i3 = _mm256_cvttpd_epi32(v3);
            eq = _mm256_and_pd(eq, _mm256_and_pd(_mm256_and_pd(_mm256_cmp_pd(_mm256_cvtepi32_pd(i0), v0, _CMP_EQ_OQ),
                                                               _mm256_cmp_pd(_mm256_cvtepi32_pd(i1), v1, _CMP_EQ_OQ)),
                                                 _mm256_and_pd(_mm256_cmp_pd(_mm256_cvtepi32_pd(i2), v2, _CMP_EQ_OQ),
                                                               _mm256_cmp_pd(_mm256_cvtepi32_pd(i3), v3, _CMP_EQ_OQ))));
            ors = _mm_or_si128(ors, _mm_or_si128(_mm_or_si128(i0, i1), _mm_or_si128(i2, i3)));
            _mm_storeu_si128((__m128i*)(dst + i), _mm_packus_epi16(_mm_packs_epi32(i0, i1), _mm_packs_epi32(i2, i3)));
        }
        if (_mm256_movemask_pd(eq) != 0xF || !_mm_testz_si128(ors, _mm_set1_epi32(~255))) return false;
    }
    return narrow_scalar(src + i, dst + i, n - i);
}

__attribute__((target("avx2"))) bool narrow_f32_avx2(const float* src, uint8_t* dst, int64_t n) {
    int64_t i = 0;
    while (i + 16 <= n) {
        const int64_t stop = (n - i < kBlock ? n : i + kBlock) - 15;
        __m256 eq = _mm256_castsi256_ps(_mm256_set1_epi32(-1));
        __m256i ors = _mm256_setzero_si256();
        for (; i < stop; i += 16) {
            const __m256 v0 = _mm256_loadu_ps(src + i), v1 = _mm256_loadu_ps(src + i + 8);
            const __m256i i0 = _mm256_cvttps_epi32(v0), i1 = _mm256_cvttps_epi32(v1);
            eq = _mm256_and_ps(eq, _mm256_and_ps(_mm256_cmp_ps(_mm256_cvtepi32_ps(i0), v0, _CMP_EQ_OQ),
                                                 _mm256_cmp_ps(_mm256_cvtepi32_ps(i1), v1, _CMP_EQ_OQ)));
            ors = _mm256_or_si256(ors, _mm256_or_si256(i0, i1));
            const __m128i lo = _mm_packs_epi32(_mm256_castsi256_si128(i0), _mm256_extracti128_si256(i0, 1));
            const __m128i hi = _mm_packs_epi32(_mm256_castsi256_si128(i1), _mm256_extracti128_si256(i1, 1));
            _mm_storeu_si128((__m128i*)(dst + i), _mm_packus_epi16(lo, hi));
        }
        if (_mm256_movemask_ps(eq) != 0xFF || !_mm256_testz_si256(ors, _mm256_set1_epi32(~255))) return false;
    }
    return narrow_scalar(src + i, dst + i, n - i);
}

const bool kHaveAvx2 = __builtin_cpu_supports("avx2");

}  // namespace

// memcpy whose destination is write-combined device memory (the host path's direct mode): non-temporal 32-byte stores for the
// aligned middle, plain copies for the edges.
__attribute__((target("avx2"))) static void stream_copy_avx2(uint8_t* d, const uint8_t* s, size_t n) {
    const size_t head = std::min<size_t>(n, (32 - ((uintptr_t)d & 31)) & 31);
    memcpy(d, s, head);
    d += head, s += head, n -= head;
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i)), b = _mm256_loadu_si256((const __m256i*)(s + i + 32)),
                      c = _mm256_loadu_si256((const __m256i*)(s + i + 64)), e = _mm256_loadu_si256((const __m256i*)(s + i + 96));
        _mm256_stream_si256((__m256i*)(d + i), a);
        _mm256_stream_si256((__m256i*)(d + i + 32), b);
        _mm256_stream_si256((__m256i*)(d + i + 64), c);
        _mm256_stream_si256((__m256i*)(d + i + 96), e);
    }
    for (; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i*)(d + i), _mm256_loadu_si256((const __m256i*)(s + i)));
    memcpy(d + i, s + i, n - i);
}

void stream_copy(void* dst, const void* src, size_t bytes) {
    if (kHaveAvx2) stream_copy_avx2((uint8_t*)dst, (const uint8_t*)src, bytes);
    else memcpy(dst, src, bytes);
}

void store_fence() { _mm_sfence(); }

bool narrow_row_f64(const double* src, uint8_t* dst, int64_t n) {
    return kHaveAvx2 ? narrow_f64_avx2(src, dst, n) : narrow_scalar(src, dst, n);
}

bool narrow_row_f32(const float* src, uint8_t* dst, int64_t n) {
    return kHaveAvx2 ? narrow_f32_avx2(src, dst, n) : narrow_scalar(src, dst, n);
}

}  // namespace hg
