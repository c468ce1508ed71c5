/*
 * oracle/exact_intrinsics.h -- TEST INFRASTRUCTURE, not product code.
 *
 * Force-included (-include) when building the "exact" variant of the compiled reference
 * (oracle/_ref/libnudge_ref_exact.so).  The reference wraps four hardware-approximate x86
 * instructions (reference nudge.cpp:296-302 `_mm_rsqrt_ps/_mm_rcp_ps`, nudge.cpp:604-610
 * `_mm256_rsqrt_ps/_mm256_rcp_ps`).  Their result bits are implementation-defined (they differ
 * between Intel and AMD hosts and from the GPU's v_rsq_f32/v_rcp_f32), so no other machine can
 * reproduce them.  This header re-routes those four intrinsics -- by macro renaming AFTER
 * <immintrin.h> has been included, the reference source is compiled where it lies, unmodified --
 * to correctly rounded IEEE sequences:
 *      rsqrt(x) := 1.0f / sqrtf(x)      (sqrt then divide, both correctly rounded)
 *      recip(x) := 1.0f / x
 * which is exactly what the HIP kernels compute (nudge_amd/csrc/nh_math.h nh_rsqrt/nh_recip).
 */
#ifndef NH_ORACLE_EXACT_INTRINSICS_H
#define NH_ORACLE_EXACT_INTRINSICS_H
#include <immintrin.h>

static inline __m128 nh_exact_rsqrt_ps(__m128 x) { return _mm_div_ps(_mm_set1_ps(1.0f), _mm_sqrt_ps(x)); }
static inline __m128 nh_exact_rcp_ps(__m128 x) { return _mm_div_ps(_mm_set1_ps(1.0f), x); }
#ifdef __AVX__
static inline __m256 nh_exact_rsqrt256_ps(__m256 x) { return _mm256_div_ps(_mm256_set1_ps(1.0f), _mm256_sqrt_ps(x)); }
static inline __m256 nh_exact_rcp256_ps(__m256 x) { return _mm256_div_ps(_mm256_set1_ps(1.0f), x); }
#endif

#define _mm_rsqrt_ps nh_exact_rsqrt_ps
#define _mm_rcp_ps nh_exact_rcp_ps
#define _mm256_rsqrt_ps nh_exact_rsqrt256_ps
#define _mm256_rcp_ps nh_exact_rcp256_ps
#endif
