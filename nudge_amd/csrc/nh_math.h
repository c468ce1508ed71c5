// nh_math.h -- scalar fp32 building blocks of the HIP kernels.
//
// Every function is `NH_HD` (host+device) so that the SAME per-item arithmetic the kernels run can be
// compiled by g++ into tests/hostsim (a CPU-side unit-test harness that diffs it against the compiled
// reference) -- the product only ever calls these from device code.
//
// Numeric contract: the translation units including this header are built with
// -ffp-contract=off; fused multiply-add happens ONLY through nh_madd/nh_msub, exactly where the
// reference writes simd_float::madd/msub (reference nudge.cpp:270-284).  Division and sqrt are
// correctly rounded (HIP default), so nh_rsqrt/nh_recip match the "exact" oracle build
// (oracle/exact_intrinsics.h), not the implementation-defined x86 rsqrtps/rcpps bits.
#ifndef NH_MATH_H
#define NH_MATH_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define NH_HD __host__ __device__ __forceinline__
#else
#define NH_HD inline __attribute__((always_inline))
#endif

struct nh_f3 { float x, y, z; };
struct nh_quat { float x, y, z, s; };   // (v, s) like the reference's Rotation (nudge.cpp:874-877)

NH_HD uint32_t nh_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
NH_HD float nh_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// First operand is returned on NaN and on equality (reference nudge.cpp:286-294: _mm_min_ps(y, x)).
NH_HD float nh_min(float x, float y) { return (y < x) ? y : x; }
NH_HD float nh_max(float x, float y) { return (y > x) ? y : x; }
NH_HD float nh_abs(float x) { return nh_asfloat(nh_asuint(x) & 0x7fffffffu); }          // nudge.cpp:308-310
NH_HD float nh_neg(float x) { return nh_asfloat(nh_asuint(x) ^ 0x80000000u); }          // nudge.cpp:63-65
NH_HD float nh_xorf(float x, uint32_t bits) { return nh_asfloat(nh_asuint(x) ^ bits); }
NH_HD uint32_t nh_signbit(float x) { return nh_asuint(x) >> 31; }
NH_HD float nh_copysign_or(float mag, float sgn) { return nh_asfloat((nh_asuint(sgn) & 0x80000000u) | nh_asuint(mag)); }

NH_HD float nh_madd(float x, float y, float z) { return __builtin_fmaf(x, y, z); }      // nudge.cpp:270-276
NH_HD float nh_msub(float x, float y, float z) { return __builtin_fmaf(x, y, -z); }     // nudge.cpp:278-284
NH_HD float nh_rsqrt(float x) { return 1.0f / sqrtf(x); }                               // exact stand-in for nudge.cpp:296-298
NH_HD float nh_recip(float x) { return 1.0f / x; }                                      // exact stand-in for nudge.cpp:300-302

#if defined(__HIP_DEVICE_COMPILE__)
// Short device sequence for 1/sqrt(x): hardware estimates + one fused correction each.  Whenever the `_ok` predicate holds for
// the short result it IS the correctly rounded value (= nh_rsqrt, bit for bit) -- verified over all 2^32 inputs by
// tests/fastmath/exhaustive.hip -- so the guarded form takes the exact form only when some lane's result falls outside (zero,
// negative, non-finite arguments): same bits as nh_rsqrt, a third of the instructions.  Used where reciprocal roots are dense and
// off the critical dependency chain (narrowphase edge axes); the solver sweeps keep the exact forms (measured: no gain there).
__device__ __forceinline__ float nh_recip_fast(float x) {
	float r = __builtin_amdgcn_rcpf(x);
	float e = __builtin_fmaf(-x, r, 1.0f);
	return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ bool nh_recip_fast_ok(float r) { return __builtin_amdgcn_classf(r, 0x108 /* -normal | +normal */); }
__device__ __forceinline__ float nh_rsqrt_fast(float x) {
	// arguments below 2^-64 are scaled by 2^64 so that the correction term stays a normal number; both scalings are exact
	const bool small = x < 5.421010862427522e-20f;
	const float xs = small ? x * 18446744073709551616.0f : x;
	float y = __builtin_amdgcn_rsqf(xs);
	float s = xs * y;                                     // sqrt(xs) estimate, then RN(sqrt(xs))
	float e = __builtin_fmaf(-s, s, xs);
	s = __builtin_fmaf(e, 0.5f * y, s);
	s = small ? s * 2.3283064365386963e-10f : s;
	return nh_recip_fast(s);
}
__device__ __forceinline__ bool nh_rsqrt_fast_ok(float r) { return __builtin_amdgcn_classf(r, 0x100 /* +normal */); }
__device__ __forceinline__ float nh_rsqrt_guarded(float x) {
	float r = nh_rsqrt_fast(x);
	if (__builtin_expect(__builtin_amdgcn_ballot_w64(!nh_rsqrt_fast_ok(r)) != 0, 0)) r = nh_rsqrt(x);
	return r;
}
// Forms for the solver sweeps, where the argument is very often EXACTLY zero (a box at rest has no tangential velocity and no friction
// impulse: 1/0 = inf is the common case there, and the reference clamps it with min()).  The hardware estimates return the right
// infinity for +-0, so a select keeps it; arguments below 2^-64 (reciprocal root) are not rescaled but sent to the exact form.  Same
// proof obligation as above -- short != exact implies the predicate fails -- over all 2^32 inputs in tests/fastmath/exhaustive.hip.
__device__ __forceinline__ float nh_recip_fast_z(float x) {
	float r = __builtin_amdgcn_rcpf(x);
	float e = __builtin_fmaf(-x, r, 1.0f);
	float y = __builtin_fmaf(e, r, r);
	return x == 0.0f ? r : y;                              // 1 / +-0 = +-inf, which the estimate returns
}
// bad: NaN, denormal or zero result (arguments at either end of the range; 1 / +-inf = +-0 takes the exact form too)
__device__ __forceinline__ bool nh_recip_fast_z_bad(float r) { return __builtin_amdgcn_classf(r, 0x0F3); }
__device__ __forceinline__ float nh_rsqrt_fast_z(float x) {
	float y = __builtin_amdgcn_rsqf(x);
	float s = x * y;
	float e = __builtin_fmaf(-s, s, x);
	s = __builtin_fmaf(e, 0.5f * y, s);                    // RN(sqrt(x)) for x >= 2^-64
	float r = __builtin_amdgcn_rcpf(s);
	float e2 = __builtin_fmaf(-s, r, 1.0f);
	float q = __builtin_fmaf(e2, r, r);
	return x == 0.0f ? y : q;
}
// bad: anything but a positive normal number or +inf
__device__ __forceinline__ bool nh_rsqrt_fast_z_bad(float r) { return !(r >= 1.17549435e-38f); }
__device__ __forceinline__ bool nh_rsqrt_fast_z_small(float x) { return x < 5.421010862427522e-20f; }      // (and not zero: see the guard)
__device__ __forceinline__ float nh_recip_guarded(float x) {
	float r = nh_recip_fast_z(x);
	if (__builtin_expect(__builtin_amdgcn_ballot_w64(nh_recip_fast_z_bad(r)) != 0, 0)) r = nh_recip(x);
	return r;
}
__device__ __forceinline__ float nh_rsqrt_guarded_z(float x) {
	float r = nh_rsqrt_fast_z(x);
	const uint64_t zero = __builtin_amdgcn_ballot_w64(x == 0.0f);
	const uint64_t bad = __builtin_amdgcn_ballot_w64(nh_rsqrt_fast_z_bad(r)) | (__builtin_amdgcn_ballot_w64(nh_rsqrt_fast_z_small(x)) & ~zero);
	if (__builtin_expect(bad != 0, 0)) r = nh_rsqrt(x);
	return r;
}
#else
NH_HD float nh_rsqrt_guarded(float x) { return nh_rsqrt(x); }
NH_HD float nh_recip_guarded(float x) { return nh_recip(x); }
NH_HD float nh_rsqrt_guarded_z(float x) { return nh_rsqrt(x); }
#endif

// x86 cvttps2dq semantics (nudge.cpp:336-338): truncate; out of range or NaN -> 0x80000000.
NH_HD int32_t nh_toint(float x) {
	if (!(x > -2147483904.0f && x < 2147483648.0f)) return (int32_t)0x80000000u;
	return (int32_t)x;
}

NH_HD nh_f3 nh_make3(float x, float y, float z) { nh_f3 r = { x, y, z }; return r; }
NH_HD nh_f3 operator+(nh_f3 a, nh_f3 b) { return nh_make3(a.x + b.x, a.y + b.y, a.z + b.z); }
NH_HD nh_f3 operator-(nh_f3 a, nh_f3 b) { return nh_make3(a.x - b.x, a.y - b.y, a.z - b.z); }
NH_HD nh_f3 operator*(float a, nh_f3 b) { return nh_make3(a * b.x, a * b.y, a * b.z); }
NH_HD nh_f3 operator*(nh_f3 a, float b) { return nh_make3(a.x * b, a.y * b, a.z * b); }
NH_HD float nh_dot(nh_f3 a, nh_f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // nudge.cpp:1104-1106
NH_HD nh_f3 nh_cross(nh_f3 a, nh_f3 b) {                                                // nudge.cpp:1112-1115
	return nh_make3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
NH_HD nh_f3 nh_qv(nh_quat q) { return nh_make3(q.x, q.y, q.z); }

// Rotation * float3 (nudge.cpp:1117-1120)
NH_HD nh_f3 nh_rotate(nh_quat q, nh_f3 r) {
	nh_f3 t = 2.0f * nh_cross(nh_qv(q), r);
	return r + q.s * t + nh_cross(nh_qv(q), t);
}

// Rotation * Rotation (nudge.cpp:1122-1126)
NH_HD nh_quat nh_qmul(nh_quat l, nh_quat r) {
	nh_f3 v = nh_qv(r) * l.s + nh_qv(l) * r.s + nh_cross(nh_qv(l), nh_qv(r));
	nh_quat q = { v.x, v.y, v.z, l.s * r.s - nh_dot(nh_qv(l), nh_qv(r)) };
	return q;
}

// matrix(q) (nudge.cpp:1142-1163): three column vectors c0,c1,c2.
struct nh_m33 { nh_f3 c0, c1, c2; };
NH_HD nh_m33 nh_matrix(nh_quat q) {
	float kx = q.x + q.x, ky = q.y + q.y, kz = q.z + q.z;
	float xx = kx * q.x, yy = ky * q.y, zz = kz * q.z;
	float xy = kx * q.y, xz = kx * q.z, yz = ky * q.z;
	float sx = kx * q.s, sy = ky * q.s, sz = kz * q.s;
	nh_m33 m = {
		{ 1.0f - yy - zz, xy + sz, xz - sy },
		{ xy - sz, 1.0f - xx - zz, yz + sx },
		{ xz + sy, yz - sx, 1.0f - xx - yy },
	};
	return m;
}

// ---- Morton code of a collider's AABB min corner (nudge.cpp:2606-2645, 3143-3149) -------------------
NH_HD void nh_dilate3(uint32_t x, unsigned offset, uint32_t& lo32, uint32_t& hi32) {
	uint32_t lo24 = x & 0xffu, hi24 = (x >> 8) & 0xffu;
	lo24 = (lo24 | (lo24 << 8)) & 0x0f00f00fu;  hi24 = (hi24 | (hi24 << 8)) & 0x0f00f00fu;
	lo24 = (lo24 | (lo24 << 4)) & 0xc30c30c3u;  hi24 = (hi24 | (hi24 << 4)) & 0xc30c30c3u;
	lo24 = (lo24 | (lo24 << 2)) & 0x49249249u;  hi24 = (hi24 | (hi24 << 2)) & 0x49249249u;
	lo32 = (lo24 << offset) | (hi24 << (24 + offset));
	hi32 = hi24 >> (8 - offset);
}

// 48-bit Morton key (x at bit 2, y at 1, z at 0) of quantised coordinates.
NH_HD uint64_t nh_morton48(int32_t qx, int32_t qy, int32_t qz) {
	uint32_t lx, hx, ly, hy, lz, hz;
	nh_dilate3((uint32_t)qx, 2, lx, hx);
	nh_dilate3((uint32_t)qy, 1, ly, hy);
	nh_dilate3((uint32_t)qz, 0, lz, hz);
	uint32_t lo = lx | ly | lz;
	uint32_t hi = (hx | hy | hz) & 0xffffu;
	return ((uint64_t)hi << 32) | lo;
}

// scale = min over axes of 65535 * recip(max - min)  (nudge.cpp:3096-3099; min() keeps the first operand on NaN)
NH_HD float nh_morton_scale(nh_f3 smin, nh_f3 smax) {
	float s0 = 65535.0f * nh_recip(smax.x - smin.x);
	float s1 = 65535.0f * nh_recip(smax.y - smin.y);
	float s2 = 65535.0f * nh_recip(smax.z - smin.z);
	float t0 = nh_min(s0, s2);    // lanes of min(shuffle<0,1,2,2>, shuffle<2,2,0,1>)
	float t1 = nh_min(s1, s2);
	return nh_min(t0, t1);        // lane 0 of min(t, shuffle<1,0,3,2>(t))
}

NH_HD uint64_t nh_morton_of(nh_f3 p, float scale, nh_f3 smin_scaled) {
	int32_t qx = nh_toint(nh_msub(p.x, scale, smin_scaled.x));
	int32_t qy = nh_toint(nh_msub(p.y, scale, smin_scaled.y));
	int32_t qz = nh_toint(nh_msub(p.z, scale, smin_scaled.z));
	return nh_morton48(qx, qy, qz);
}

#endif
