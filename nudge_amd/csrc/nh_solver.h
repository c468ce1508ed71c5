// nh_solver.h -- per-contact solver arithmetic: world inverse inertia, constraint row build, warm start,
// one projected-Gauss-Seidel update, cached-impulse export; plus the per-body integrator.
//
// Scalar restatement-for-GPU of the reference's 8-wide AVX2 loops with the SAME operation order and the
// same fused-multiply-add placement (reference nudge.cpp:4182-4199, 4386-4633, 4646-4854, 4864-4883,
// 4890-4925).  See nh_math.h for the numeric contract.
//
// DERIVED WORK NOTICE.  The per-contact / per-pair arithmetic in this file follows rasmusbarr/nudge's nudge.cpp statement by statement
// (nh_build_row / nh_warm_start / nh_apply_one / nh_export_impulse / nh_advance_body: nudge.cpp:4182-4199, 4393-4854, 4864-4883, 4890-4925;
// the one-body forms nh_*1 are those with the static body's terms removed): operation order and fused-multiply-add placement are the numeric contract of bit-exact parity
// with that library, so those functions are a scalar restatement of its SIMD loops, not an independent design.  nudge is
//     Copyright (c) 2017 Rasmus Barringer
// and distributed under the MIT license: "Permission is hereby granted, free of charge, to any person obtaining a copy of this software
// and associated documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
// rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit persons to whom
// the Software is furnished to do so, subject to the following conditions: The above copyright notice and this permission notice shall
// be included in all copies or substantial portions of the Software.  THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND,
// EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND
// NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN
// ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE
// SOFTWARE."
#ifndef NH_SOLVER_H
#define NH_SOLVER_H

#include "nh_math.h"

// division and reciprocal root of the solver rows: the guarded short sequences of nh_math.h (bit-identical to nh_recip / nh_rsqrt by the
// exhaustive check, a third of the instructions -- the fused one-body solver and the one-workgroup solvers are bound by instruction issue);
// -DNH_SOLVER_EXACT_DIV selects the plain forms
#if defined(NH_SOLVER_EXACT_DIV)
NH_HD float nh_recip_s(float x) { return nh_recip(x); }
NH_HD float nh_rsqrt_s(float x) { return nh_rsqrt(x); }
#else
NH_HD float nh_recip_s(float x) { return nh_recip_guarded(x); }
NH_HD float nh_rsqrt_s(float x) { return nh_rsqrt_guarded_z(x); }
#endif

// World-space inverse inertia R diag R^T, 6 unique terms (nudge.cpp:4182-4197).
struct nh_inertia { float xx, yy, zz, xy, xz, yz; };

NH_HD nh_inertia nh_world_inertia(nh_quat q, float ix, float iy, float iz) {
	nh_m33 m = nh_matrix(q);
	nh_inertia t;
	t.xx = ix * m.c0.x * m.c0.x + iy * m.c1.x * m.c1.x + iz * m.c2.x * m.c2.x;
	t.yy = ix * m.c0.y * m.c0.y + iy * m.c1.y * m.c1.y + iz * m.c2.y * m.c2.y;
	t.zz = ix * m.c0.z * m.c0.z + iy * m.c1.z * m.c1.z + iz * m.c2.z * m.c2.z;
	t.xy = ix * m.c0.x * m.c0.y + iy * m.c1.x * m.c1.y + iz * m.c2.x * m.c2.y;
	t.xz = ix * m.c0.x * m.c0.z + iy * m.c1.x * m.c1.z + iz * m.c2.x * m.c2.z;
	t.yz = ix * m.c0.y * m.c0.z + iy * m.c1.y * m.c1.z + iz * m.c2.y * m.c2.z;
	return t;
}

// One constraint row: the 39 floats of ContactConstraintV (nudge.cpp:903-958) for ONE contact, + 1 pad
// (160 B = ten 16-byte loads).  a-side angular terms are stored negated like the reference (4539-4549).
struct nh_row {
	float pa_z, pa_x, pa_y;
	float pb_z, pb_x, pb_y;
	float n_x, u_x, v_x;
	float n_y, u_y, v_y;
	float n_z, u_z, v_z;
	float bias, friction, normal_velocity_to_normal_impulse;
	float friction_coefficient_x, friction_coefficient_y, friction_coefficient_z;
	float na_x, na_y, na_z;
	float nb_x, nb_y, nb_z;
	float ua_x, ua_y, ua_z;
	float va_x, va_y, va_z;
	float ub_x, ub_y, ub_z;
	float vb_x, vb_y, vb_z;
	float pad;
};

struct nh_state { float normal, friction_x, friction_y; };   // ContactConstraintStateV, one lane (nudge.cpp:960-964)

// Body velocity record carried through the solver: layout of BodyMomentum with unused0 = mass_inverse
// (nudge.cpp:4198) -- exactly what the reference gathers/scatters.
struct nh_vel { float vx, vy, vz, mass_inverse, wx, wy, wz, w_unused; };

struct nh_contact_in { float px, py, pz, penetration, nx, ny, nz, friction; };

#define NH_ALLOWED_PENETRATION 1e-3f   // nudge.cpp:49
#define NH_BIAS_FACTOR 2.0f            // nudge.cpp:50

// Row build (nudge.cpp:4393-4561).  apos/bpos = body positions, ia/ib = world inverse inertia.
NH_HD void nh_build_row(const nh_contact_in& c, nh_f3 apos, nh_f3 bpos, const nh_inertia& A, const nh_inertia& B,
                        float a_mass_inverse, float b_mass_inverse, nh_row& r) {
	float nx = c.nx, ny = c.ny, nz = c.nz;
	float pa_x = c.px - apos.x, pa_y = c.py - apos.y, pa_z = c.pz - apos.z;
	float pb_x = c.px - bpos.x, pb_y = c.py - bpos.y, pb_z = c.pz - bpos.z;

	float na_xt = pa_y * nz - pa_z * ny, na_yt = pa_z * nx - pa_x * nz, na_zt = pa_x * ny - pa_y * nx;
	float na_x = A.xx * na_xt + A.xy * na_yt + A.xz * na_zt;
	float na_y = A.xy * na_xt + A.yy * na_yt + A.yz * na_zt;
	float na_z = A.xz * na_xt + A.yz * na_yt + A.zz * na_zt;

	float nb_xt = pb_y * nz - pb_z * ny, nb_yt = pb_z * nx - pb_x * nz, nb_zt = pb_x * ny - pb_y * nx;
	float nb_x = B.xx * nb_xt + B.xy * nb_yt + B.xz * nb_zt;
	float nb_y = B.xy * nb_xt + B.yy * nb_yt + B.yz * nb_zt;
	float nb_z = B.xz * nb_xt + B.yz * nb_yt + B.zz * nb_zt;

	// cross(na, pa), cross(nb, pb)
	na_xt = na_y * pa_z - na_z * pa_y; na_yt = na_z * pa_x - na_x * pa_z; na_zt = na_x * pa_y - na_y * pa_x;
	nb_xt = nb_y * pb_z - nb_z * pb_y; nb_yt = nb_z * pb_x - nb_x * pb_z; nb_zt = nb_x * pb_y - nb_y * pb_x;

	float rx = na_xt + nb_xt, ry = na_yt + nb_yt, rz = na_zt + nb_zt;
	float r_dot_n = rx * nx + ry * ny + rz * nz;

	float mass_inverse = a_mass_inverse + b_mass_inverse;
	float nvtni = mass_inverse + r_dot_n;
	// AVX2 build: cmp_neq is ordered (false on NaN), nudge.cpp:636-638, 4439-4440
	bool nonzero = (nvtni < 0.0f) || (nvtni > 0.0f);
	nvtni = nonzero ? nh_neg(nh_recip_s(nvtni)) : 0.0f;

	float bias = (-NH_BIAS_FACTOR) * nh_max(c.penetration - NH_ALLOWED_PENETRATION, 0.0f) * nvtni;

	// smoothly varying tangent basis (nudge.cpp:4445-4456)
	float s = nh_abs(nx);
	float u_x = nz * s;
	float u_y = u_x - nz;
	float u_z = nh_madd(nx - ny, s, ny);
	u_x = nh_neg(u_x);
	{
		float f = nh_rsqrt_s(u_x * u_x + u_y * u_y + u_z * u_z);
		u_x *= f; u_y *= f; u_z *= f;
	}
	float v_x = u_y * nz - u_z * ny, v_y = u_z * nx - u_x * nz, v_z = u_x * ny - u_y * nx;

	float ua_x = pa_y * u_z - pa_z * u_y, ua_y = pa_z * u_x - pa_x * u_z, ua_z = pa_x * u_y - pa_y * u_x;
	float va_x = pa_y * v_z - pa_z * v_y, va_y = pa_z * v_x - pa_x * v_z, va_z = pa_x * v_y - pa_y * v_x;
	float ub_x = pb_y * u_z - pb_z * u_y, ub_y = pb_z * u_x - pb_x * u_z, ub_z = pb_x * u_y - pb_y * u_x;
	float vb_x = pb_y * v_z - pb_z * v_y, vb_y = pb_z * v_x - pb_x * v_z, vb_z = pb_x * v_y - pb_y * v_x;

	float a_duu = A.xx * ua_x * ua_x + A.yy * ua_y * ua_y + A.zz * ua_z * ua_z;
	float a_dvv = A.xx * va_x * va_x + A.yy * va_y * va_y + A.zz * va_z * va_z;
	float a_duv = A.xx * ua_x * va_x + A.yy * ua_y * va_y + A.zz * ua_z * va_z;
	float a_suu = A.xy * ua_x * ua_y + A.xz * ua_x * ua_z + A.yz * ua_y * ua_z;
	float a_svv = A.xy * va_x * va_y + A.xz * va_x * va_z + A.yz * va_y * va_z;
	float a_suv = A.xy * (ua_x * va_y + ua_y * va_x) + A.xz * (ua_x * va_z + ua_z * va_x) + A.yz * (ua_y * va_z + ua_z * va_y);

	float b_duu = B.xx * ub_x * ub_x + B.yy * ub_y * ub_y + B.zz * ub_z * ub_z;
	float b_dvv = B.xx * vb_x * vb_x + B.yy * vb_y * vb_y + B.zz * vb_z * vb_z;
	float b_duv = B.xx * ub_x * vb_x + B.yy * ub_y * vb_y + B.zz * ub_z * vb_z;
	float b_suu = B.xy * ub_x * ub_y + B.xz * ub_x * ub_z + B.yz * ub_y * ub_z;
	float b_svv = B.xy * vb_x * vb_y + B.xz * vb_x * vb_z + B.yz * vb_y * vb_z;
	float b_suv = B.xy * (ub_x * vb_y + ub_y * vb_x) + B.xz * (ub_x * vb_z + ub_z * vb_x) + B.yz * (ub_y * vb_z + ub_z * vb_y);

	float friction_x = mass_inverse + a_duu + a_suu + a_suu + b_duu + b_suu + b_suu;
	float friction_y = mass_inverse + a_dvv + a_svv + a_svv + b_dvv + b_svv + b_svv;
	float friction_z = a_duv + a_duv + a_suv + a_suv + b_duv + b_duv + b_suv + b_suv;

	float ua_xt = A.xx * ua_x + A.xy * ua_y + A.xz * ua_z;
	float ua_yt = A.xy * ua_x + A.yy * ua_y + A.yz * ua_z;
	float ua_zt = A.xz * ua_x + A.yz * ua_y + A.zz * ua_z;
	float va_xt = A.xx * va_x + A.xy * va_y + A.xz * va_z;
	float va_yt = A.xy * va_x + A.yy * va_y + A.yz * va_z;
	float va_zt = A.xz * va_x + A.yz * va_y + A.zz * va_z;
	float ub_xt = B.xx * ub_x + B.xy * ub_y + B.xz * ub_z;
	float ub_yt = B.xy * ub_x + B.yy * ub_y + B.yz * ub_z;
	float ub_zt = B.xz * ub_x + B.yz * ub_y + B.zz * ub_z;
	float vb_xt = B.xx * vb_x + B.xy * vb_y + B.xz * vb_z;
	float vb_yt = B.xy * vb_x + B.yy * vb_y + B.yz * vb_z;
	float vb_zt = B.xz * vb_x + B.yz * vb_y + B.zz * vb_z;

	r.pa_x = pa_x; r.pa_y = pa_y; r.pa_z = pa_z;
	r.pb_x = pb_x; r.pb_y = pb_y; r.pb_z = pb_z;
	r.n_x = nx; r.n_y = ny; r.n_z = nz;
	r.u_x = u_x; r.u_y = u_y; r.u_z = u_z;
	r.v_x = v_x; r.v_y = v_y; r.v_z = v_z;
	r.bias = bias; r.friction = c.friction; r.normal_velocity_to_normal_impulse = nvtni;
	r.friction_coefficient_x = friction_x; r.friction_coefficient_y = friction_y; r.friction_coefficient_z = friction_z;
	r.ua_x = nh_neg(ua_xt); r.ua_y = nh_neg(ua_yt); r.ua_z = nh_neg(ua_zt);
	r.va_x = nh_neg(va_xt); r.va_y = nh_neg(va_yt); r.va_z = nh_neg(va_zt);
	r.na_x = nh_neg(na_x); r.na_y = nh_neg(na_y); r.na_z = nh_neg(na_z);
	r.ub_x = ub_xt; r.ub_y = ub_yt; r.ub_z = ub_zt;
	r.vb_x = vb_xt; r.vb_y = vb_yt; r.vb_z = vb_zt;
	r.nb_x = nb_x; r.nb_y = nb_y; r.nb_z = nb_z;
	r.pad = 0.0f;
}

// Warm start (nudge.cpp:4563-4632): project the cached world-space impulse on the new basis, clamp to the
// friction cone, apply to both bodies.
NH_HD void nh_warm_start(const nh_row& r, float cix, float ciy, float ciz, nh_vel& a, nh_vel& b, nh_state& st) {
	float normal_impulse = nh_max(r.n_x * cix + r.n_y * ciy + r.n_z * ciz, 0.0f);
	float max_friction_impulse = normal_impulse * r.friction;
	float fx = r.u_x * cix + r.u_y * ciy + r.u_z * ciz;
	float fy = r.v_x * cix + r.v_y * ciy + r.v_z * ciz;
	float scale = fx * fx + fy * fy;
	scale = nh_rsqrt_s(scale);
	scale = scale * max_friction_impulse;
	scale = nh_min(1.0f, scale);
	fx = fx * scale;
	fy = fy * scale;

	float lx = fx * r.u_x + fy * r.v_x + r.n_x * normal_impulse;
	float ly = fx * r.u_y + fy * r.v_y + r.n_y * normal_impulse;
	float lz = fx * r.u_z + fy * r.v_z + r.n_z * normal_impulse;

	float aax = fx * r.ua_x + fy * r.va_x + normal_impulse * r.na_x;
	float aay = fx * r.ua_y + fy * r.va_y + normal_impulse * r.na_y;
	float aaz = fx * r.ua_z + fy * r.va_z + normal_impulse * r.na_z;
	float bax = fx * r.ub_x + fy * r.vb_x + normal_impulse * r.nb_x;
	float bay = fx * r.ub_y + fy * r.vb_y + normal_impulse * r.nb_y;
	float baz = fx * r.ub_z + fy * r.vb_z + normal_impulse * r.nb_z;

	a.vx -= lx * a.mass_inverse; a.vy -= ly * a.mass_inverse; a.vz -= lz * a.mass_inverse;
	a.wx += aax; a.wy += aay; a.wz += aaz;
	b.vx += lx * b.mass_inverse; b.vy += ly * b.mass_inverse; b.vz += lz * b.mass_inverse;
	b.wx += bax; b.wy += bay; b.wz += baz;

	st.normal = normal_impulse; st.friction_x = fx; st.friction_y = fy;
}

// One PGS update of one contact (nudge.cpp:4649-4853), explicit FMAs exactly where the reference has madd.
NH_HD void nh_apply_one(const nh_row& r, nh_state& st, nh_vel& a, nh_vel& b) {
	float v_xa = nh_madd(a.wy, r.pa_z, a.vx);
	float v_ya = nh_madd(a.wz, r.pa_x, a.vy);
	float v_za = nh_madd(a.wx, r.pa_y, a.vz);

	float v_xb = nh_madd(b.wy, r.pb_z, b.vx);
	float v_yb = nh_madd(b.wz, r.pb_x, b.vy);
	float v_zb = nh_madd(b.wx, r.pb_y, b.vz);

	v_xa = nh_madd(b.wz, r.pb_y, v_xa);
	v_ya = nh_madd(b.wx, r.pb_z, v_ya);
	v_za = nh_madd(b.wy, r.pb_x, v_za);

	v_xb = nh_madd(a.wz, r.pa_y, v_xb);
	v_yb = nh_madd(a.wx, r.pa_z, v_yb);
	v_zb = nh_madd(a.wy, r.pa_x, v_zb);

	float v_x = v_xb - v_xa, v_y = v_yb - v_ya, v_z = v_zb - v_za;

	float t_z = r.n_x * v_x;
	float t_x = v_x * r.u_x;
	float t_y = v_x * r.v_x;

	float normal_bias = r.bias;
	float old_normal_impulse = st.normal;
	float normal_factor = r.normal_velocity_to_normal_impulse;

	t_z = nh_madd(r.n_y, v_y, t_z);
	t_x = nh_madd(v_y, r.u_y, t_x);
	t_y = nh_madd(v_y, r.v_y, t_y);

	normal_bias = normal_bias + old_normal_impulse;

	t_z = nh_madd(r.n_z, v_z, t_z);
	t_x = nh_madd(v_z, r.u_z, t_x);
	t_y = nh_madd(v_z, r.v_z, t_y);

	float normal_impulse = nh_madd(normal_factor, t_z, normal_bias);

	float t_xx = t_x * t_x, t_yy = t_y * t_y, t_xy = t_x * t_y;
	float tl2 = t_xx + t_yy;

	normal_impulse = nh_max(normal_impulse, 0.0f);

	t_x *= tl2;
	t_y *= tl2;

	st.normal = normal_impulse;

	float max_friction_impulse = normal_impulse * r.friction;
	normal_impulse = normal_impulse - old_normal_impulse;

	float friction_factor = t_xx * r.friction_coefficient_x;
	float linear_impulse_x = r.n_x * normal_impulse;
	friction_factor = nh_madd(t_yy, r.friction_coefficient_y, friction_factor);
	float linear_impulse_y = r.n_y * normal_impulse;
	friction_factor = nh_madd(t_xy, r.friction_coefficient_z, friction_factor);
	float linear_impulse_z = r.n_z * normal_impulse;

	friction_factor = nh_recip_s(friction_factor);

	a.wx = nh_madd(r.na_x, normal_impulse, a.wx);
	a.wy = nh_madd(r.na_y, normal_impulse, a.wy);
	a.wz = nh_madd(r.na_z, normal_impulse, a.wz);

	float old_friction_impulse_x = st.friction_x, old_friction_impulse_y = st.friction_y;

	friction_factor = nh_min(1e+6f, friction_factor);

	float friction_impulse_x = t_x * friction_factor;
	float friction_impulse_y = t_y * friction_factor;

	friction_impulse_x = old_friction_impulse_x - friction_impulse_x;
	friction_impulse_y = old_friction_impulse_y - friction_impulse_y;

	float friction_clamp_scale = friction_impulse_x * friction_impulse_x + friction_impulse_y * friction_impulse_y;
	friction_clamp_scale = nh_rsqrt_s(friction_clamp_scale);

	b.wx = nh_madd(r.nb_x, normal_impulse, b.wx);
	b.wy = nh_madd(r.nb_y, normal_impulse, b.wy);
	b.wz = nh_madd(r.nb_z, normal_impulse, b.wz);

	friction_clamp_scale = friction_clamp_scale * max_friction_impulse;
	friction_clamp_scale = nh_min(1.0f, friction_clamp_scale);

	friction_impulse_x = friction_impulse_x * friction_clamp_scale;
	friction_impulse_y = friction_impulse_y * friction_clamp_scale;

	st.friction_x = friction_impulse_x;
	st.friction_y = friction_impulse_y;

	friction_impulse_x -= old_friction_impulse_x;
	friction_impulse_y -= old_friction_impulse_y;

	linear_impulse_x = nh_madd(r.u_x, friction_impulse_x, linear_impulse_x);
	linear_impulse_y = nh_madd(r.u_y, friction_impulse_x, linear_impulse_y);
	linear_impulse_z = nh_madd(r.u_z, friction_impulse_x, linear_impulse_z);

	linear_impulse_x = nh_madd(r.v_x, friction_impulse_y, linear_impulse_x);
	linear_impulse_y = nh_madd(r.v_y, friction_impulse_y, linear_impulse_y);
	linear_impulse_z = nh_madd(r.v_z, friction_impulse_y, linear_impulse_z);

	float a_mass_inverse_neg = nh_neg(a.mass_inverse);

	a.vx = nh_madd(linear_impulse_x, a_mass_inverse_neg, a.vx);
	a.vy = nh_madd(linear_impulse_y, a_mass_inverse_neg, a.vy);
	a.vz = nh_madd(linear_impulse_z, a_mass_inverse_neg, a.vz);

	a.wx = nh_madd(r.ua_x, friction_impulse_x, a.wx);
	a.wy = nh_madd(r.ua_y, friction_impulse_x, a.wy);
	a.wz = nh_madd(r.ua_z, friction_impulse_x, a.wz);

	a.wx = nh_madd(r.va_x, friction_impulse_y, a.wx);
	a.wy = nh_madd(r.va_y, friction_impulse_y, a.wy);
	a.wz = nh_madd(r.va_z, friction_impulse_y, a.wz);

	a.w_unused = 0.0f;

	b.vx = nh_madd(linear_impulse_x, b.mass_inverse, b.vx);
	b.vy = nh_madd(linear_impulse_y, b.mass_inverse, b.vy);
	b.vz = nh_madd(linear_impulse_z, b.mass_inverse, b.vz);

	b.wx = nh_madd(r.ub_x, friction_impulse_x, b.wx);
	b.wy = nh_madd(r.ub_y, friction_impulse_x, b.wy);
	b.wz = nh_madd(r.ub_z, friction_impulse_x, b.wz);

	b.wx = nh_madd(r.vb_x, friction_impulse_y, b.wx);
	b.wy = nh_madd(r.vb_y, friction_impulse_y, b.wy);
	b.wz = nh_madd(r.vb_z, friction_impulse_y, b.wz);

	b.w_unused = 0.0f;
}

// ---- one-body forms ---------------------------------------------------------------------------------------------------
// A contact between a dynamic body D and an INERT static world body S (velocity, inverse mass and inverse inertia all +0:
// what body 0 is in every scene of the reference).  In the general functions above every S-side quantity is then a product
// with zero: S's velocity stays +0 for good (x*0 + 0 = +0 in round-to-nearest for finite x), S's half of the row
// (nb/ub/vb or na/ua/va) is +-0, and S's lever arm only ever appears as fma(0, p, y).  The functions below are the general
// ones with those operations removed -- same operations, same order, same fused multiply-adds on everything that can be
// non-zero -- so D's velocity, the impulses and the exported cache entries come out bit-identical for finite inputs
// (tests/test_cpu_oracle.py::test_one_body_forms_equal_general_forms checks that on the golden scenes).  The only
// numbers that can differ are signs of exact zeros that are provably never observed: friction_coefficient_z (only ever
// added to a non-negative accumulator) and r (added to a positive inverse mass).
// A row shrinks from 39 to 27 floats: four of them fit in registers with room for a second wave per SIMD.

struct nh_row1 {
	float p_z, p_x, p_y;                   // lever arm of D
	float n_x, u_x, v_x;
	float n_y, u_y, v_y;
	float n_z, u_z, v_z;
	float bias, friction, normal_velocity_to_normal_impulse;
	float friction_coefficient_x, friction_coefficient_y, friction_coefficient_z;
	float nd_x, nd_y, nd_z;                // angular response of D to the normal / u / v impulse (negated when D is "a", like nh_row)
	float ud_x, ud_y, ud_z;
	float vd_x, vd_y, vd_z;
};

// nh_build_row with S's terms removed.  `d_is_a`: D plays the role of body a (S = b), else D = b.
NH_HD void nh_build_row1(const nh_contact_in& c, nh_f3 dpos, const nh_inertia& D, float d_mass_inverse, bool d_is_a, nh_row1& r) {
	float nx = c.nx, ny = c.ny, nz = c.nz;
	float p_x = c.px - dpos.x, p_y = c.py - dpos.y, p_z = c.pz - dpos.z;

	float n_xt = p_y * nz - p_z * ny, n_yt = p_z * nx - p_x * nz, n_zt = p_x * ny - p_y * nx;
	float nd_x = D.xx * n_xt + D.xy * n_yt + D.xz * n_zt;
	float nd_y = D.xy * n_xt + D.yy * n_yt + D.yz * n_zt;
	float nd_z = D.xz * n_xt + D.yz * n_yt + D.zz * n_zt;

	// cross(nd, p)
	float rx = nd_y * p_z - nd_z * p_y, ry = nd_z * p_x - nd_x * p_z, rz = nd_x * p_y - nd_y * p_x;
	float r_dot_n = rx * nx + ry * ny + rz * nz;

	float mass_inverse = d_mass_inverse;
	float nvtni = mass_inverse + r_dot_n;
	bool nonzero = (nvtni < 0.0f) || (nvtni > 0.0f);
	nvtni = nonzero ? nh_neg(nh_recip_s(nvtni)) : 0.0f;

	float bias = (-NH_BIAS_FACTOR) * nh_max(c.penetration - NH_ALLOWED_PENETRATION, 0.0f) * nvtni;

	float s = nh_abs(nx);
	float u_x = nz * s;
	float u_y = u_x - nz;
	float u_z = nh_madd(nx - ny, s, ny);
	u_x = nh_neg(u_x);
	{
		float f = nh_rsqrt_s(u_x * u_x + u_y * u_y + u_z * u_z);
		u_x *= f; u_y *= f; u_z *= f;
	}
	float v_x = u_y * nz - u_z * ny, v_y = u_z * nx - u_x * nz, v_z = u_x * ny - u_y * nx;

	float ud_x = p_y * u_z - p_z * u_y, ud_y = p_z * u_x - p_x * u_z, ud_z = p_x * u_y - p_y * u_x;
	float vd_x = p_y * v_z - p_z * v_y, vd_y = p_z * v_x - p_x * v_z, vd_z = p_x * v_y - p_y * v_x;

	float duu = D.xx * ud_x * ud_x + D.yy * ud_y * ud_y + D.zz * ud_z * ud_z;
	float dvv = D.xx * vd_x * vd_x + D.yy * vd_y * vd_y + D.zz * vd_z * vd_z;
	float duv = D.xx * ud_x * vd_x + D.yy * ud_y * vd_y + D.zz * ud_z * vd_z;
	float suu = D.xy * ud_x * ud_y + D.xz * ud_x * ud_z + D.yz * ud_y * ud_z;
	float svv = D.xy * vd_x * vd_y + D.xz * vd_x * vd_z + D.yz * vd_y * vd_z;
	float suv = D.xy * (ud_x * vd_y + ud_y * vd_x) + D.xz * (ud_x * vd_z + ud_z * vd_x) + D.yz * (ud_y * vd_z + ud_z * vd_y);

	float friction_x = mass_inverse + duu + suu + suu;
	float friction_y = mass_inverse + dvv + svv + svv;
	float friction_z = duv + duv + suv + suv;

	float ud_xt = D.xx * ud_x + D.xy * ud_y + D.xz * ud_z;
	float ud_yt = D.xy * ud_x + D.yy * ud_y + D.yz * ud_z;
	float ud_zt = D.xz * ud_x + D.yz * ud_y + D.zz * ud_z;
	float vd_xt = D.xx * vd_x + D.xy * vd_y + D.xz * vd_z;
	float vd_yt = D.xy * vd_x + D.yy * vd_y + D.yz * vd_z;
	float vd_zt = D.xz * vd_x + D.yz * vd_y + D.zz * vd_z;

	r.p_x = p_x; r.p_y = p_y; r.p_z = p_z;
	r.n_x = nx; r.n_y = ny; r.n_z = nz;
	r.u_x = u_x; r.u_y = u_y; r.u_z = u_z;
	r.v_x = v_x; r.v_y = v_y; r.v_z = v_z;
	r.bias = bias; r.friction = c.friction; r.normal_velocity_to_normal_impulse = nvtni;
	r.friction_coefficient_x = friction_x; r.friction_coefficient_y = friction_y; r.friction_coefficient_z = friction_z;
	if (d_is_a) {
		r.nd_x = nh_neg(nd_x); r.nd_y = nh_neg(nd_y); r.nd_z = nh_neg(nd_z);
		r.ud_x = nh_neg(ud_xt); r.ud_y = nh_neg(ud_yt); r.ud_z = nh_neg(ud_zt);
		r.vd_x = nh_neg(vd_xt); r.vd_y = nh_neg(vd_yt); r.vd_z = nh_neg(vd_zt);
	} else {
		r.nd_x = nd_x; r.nd_y = nd_y; r.nd_z = nd_z;
		r.ud_x = ud_xt; r.ud_y = ud_yt; r.ud_z = ud_zt;
		r.vd_x = vd_xt; r.vd_y = vd_yt; r.vd_z = vd_zt;
	}
}

// nh_warm_start, D's half
NH_HD void nh_warm_start1(const nh_row1& r, float cix, float ciy, float ciz, nh_vel& d, bool d_is_a, nh_state& st) {
	float normal_impulse = nh_max(r.n_x * cix + r.n_y * ciy + r.n_z * ciz, 0.0f);
	float max_friction_impulse = normal_impulse * r.friction;
	float fx = r.u_x * cix + r.u_y * ciy + r.u_z * ciz;
	float fy = r.v_x * cix + r.v_y * ciy + r.v_z * ciz;
	float scale = fx * fx + fy * fy;
	scale = nh_rsqrt_s(scale);
	scale = scale * max_friction_impulse;
	scale = nh_min(1.0f, scale);
	fx = fx * scale;
	fy = fy * scale;

	float lx = fx * r.u_x + fy * r.v_x + r.n_x * normal_impulse;
	float ly = fx * r.u_y + fy * r.v_y + r.n_y * normal_impulse;
	float lz = fx * r.u_z + fy * r.v_z + r.n_z * normal_impulse;

	float ax = fx * r.ud_x + fy * r.vd_x + normal_impulse * r.nd_x;
	float ay = fx * r.ud_y + fy * r.vd_y + normal_impulse * r.nd_y;
	float az = fx * r.ud_z + fy * r.vd_z + normal_impulse * r.nd_z;

	if (d_is_a) { d.vx -= lx * d.mass_inverse; d.vy -= ly * d.mass_inverse; d.vz -= lz * d.mass_inverse; }
	else        { d.vx += lx * d.mass_inverse; d.vy += ly * d.mass_inverse; d.vz += lz * d.mass_inverse; }
	d.wx += ax; d.wy += ay; d.wz += az;

	st.normal = normal_impulse; st.friction_x = fx; st.friction_y = fy;
}

// nh_apply_one, D's half
NH_HD void nh_apply_one1(const nh_row1& r, nh_state& st, nh_vel& d, bool d_is_a) {
	// velocity of the contact point on D, split the way the general form splits it between its two accumulators
	float l_x = nh_madd(d.wy, r.p_z, d.vx);
	float l_y = nh_madd(d.wz, r.p_x, d.vy);
	float l_z = nh_madd(d.wx, r.p_y, d.vz);
	float c_x = nh_madd(d.wz, r.p_y, 0.0f);
	float c_y = nh_madd(d.wx, r.p_z, 0.0f);
	float c_z = nh_madd(d.wy, r.p_x, 0.0f);
	// general: v = v_b - v_a with (v_a, v_b) = (l, c) when D is a, (c, l) when D is b
	float v_x = d_is_a ? (c_x - l_x) : (l_x - c_x);
	float v_y = d_is_a ? (c_y - l_y) : (l_y - c_y);
	float v_z = d_is_a ? (c_z - l_z) : (l_z - c_z);

	float t_z = r.n_x * v_x;
	float t_x = v_x * r.u_x;
	float t_y = v_x * r.v_x;

	float normal_bias = r.bias;
	float old_normal_impulse = st.normal;
	float normal_factor = r.normal_velocity_to_normal_impulse;

	t_z = nh_madd(r.n_y, v_y, t_z);
	t_x = nh_madd(v_y, r.u_y, t_x);
	t_y = nh_madd(v_y, r.v_y, t_y);

	normal_bias = normal_bias + old_normal_impulse;

	t_z = nh_madd(r.n_z, v_z, t_z);
	t_x = nh_madd(v_z, r.u_z, t_x);
	t_y = nh_madd(v_z, r.v_z, t_y);

	float normal_impulse = nh_madd(normal_factor, t_z, normal_bias);

	float t_xx = t_x * t_x, t_yy = t_y * t_y, t_xy = t_x * t_y;
	float tl2 = t_xx + t_yy;

	normal_impulse = nh_max(normal_impulse, 0.0f);

	t_x *= tl2;
	t_y *= tl2;

	st.normal = normal_impulse;

	float max_friction_impulse = normal_impulse * r.friction;
	normal_impulse = normal_impulse - old_normal_impulse;

	float friction_factor = t_xx * r.friction_coefficient_x;
	float linear_impulse_x = r.n_x * normal_impulse;
	friction_factor = nh_madd(t_yy, r.friction_coefficient_y, friction_factor);
	float linear_impulse_y = r.n_y * normal_impulse;
	friction_factor = nh_madd(t_xy, r.friction_coefficient_z, friction_factor);
	float linear_impulse_z = r.n_z * normal_impulse;

	friction_factor = nh_recip_s(friction_factor);

	d.wx = nh_madd(r.nd_x, normal_impulse, d.wx);
	d.wy = nh_madd(r.nd_y, normal_impulse, d.wy);
	d.wz = nh_madd(r.nd_z, normal_impulse, d.wz);

	float old_friction_impulse_x = st.friction_x, old_friction_impulse_y = st.friction_y;

	friction_factor = nh_min(1e+6f, friction_factor);

	float friction_impulse_x = t_x * friction_factor;
	float friction_impulse_y = t_y * friction_factor;

	friction_impulse_x = old_friction_impulse_x - friction_impulse_x;
	friction_impulse_y = old_friction_impulse_y - friction_impulse_y;

	float friction_clamp_scale = friction_impulse_x * friction_impulse_x + friction_impulse_y * friction_impulse_y;
	friction_clamp_scale = nh_rsqrt_s(friction_clamp_scale);

	friction_clamp_scale = friction_clamp_scale * max_friction_impulse;
	friction_clamp_scale = nh_min(1.0f, friction_clamp_scale);

	friction_impulse_x = friction_impulse_x * friction_clamp_scale;
	friction_impulse_y = friction_impulse_y * friction_clamp_scale;

	st.friction_x = friction_impulse_x;
	st.friction_y = friction_impulse_y;

	friction_impulse_x -= old_friction_impulse_x;
	friction_impulse_y -= old_friction_impulse_y;

	linear_impulse_x = nh_madd(r.u_x, friction_impulse_x, linear_impulse_x);
	linear_impulse_y = nh_madd(r.u_y, friction_impulse_x, linear_impulse_y);
	linear_impulse_z = nh_madd(r.u_z, friction_impulse_x, linear_impulse_z);

	linear_impulse_x = nh_madd(r.v_x, friction_impulse_y, linear_impulse_x);
	linear_impulse_y = nh_madd(r.v_y, friction_impulse_y, linear_impulse_y);
	linear_impulse_z = nh_madd(r.v_z, friction_impulse_y, linear_impulse_z);

	float signed_mass_inverse = d_is_a ? nh_neg(d.mass_inverse) : d.mass_inverse;

	d.vx = nh_madd(linear_impulse_x, signed_mass_inverse, d.vx);
	d.vy = nh_madd(linear_impulse_y, signed_mass_inverse, d.vy);
	d.vz = nh_madd(linear_impulse_z, signed_mass_inverse, d.vz);

	d.wx = nh_madd(r.ud_x, friction_impulse_x, d.wx);
	d.wy = nh_madd(r.ud_y, friction_impulse_x, d.wy);
	d.wz = nh_madd(r.ud_z, friction_impulse_x, d.wz);

	d.wx = nh_madd(r.vd_x, friction_impulse_y, d.wx);
	d.wy = nh_madd(r.vd_y, friction_impulse_y, d.wy);
	d.wz = nh_madd(r.vd_z, friction_impulse_y, d.wz);

	d.w_unused = 0.0f;
}

NH_HD void nh_export_impulse1(const nh_row1& r, const nh_state& st, float& ix, float& iy, float& iz) {
	ix = st.normal * r.n_x + st.friction_x * r.u_x + st.friction_y * r.v_x;
	iy = st.normal * r.n_y + st.friction_x * r.u_y + st.friction_y * r.v_y;
	iz = st.normal * r.n_z + st.friction_x * r.u_z + st.friction_y * r.v_z;
}

// ---- radial forms ------------------------------------------------------------------------------------------------------
// A contact whose two lever arms are PARALLEL to its normal (p_a = alpha * n, p_b = beta * n) between bodies of ISOTROPIC inverse
// inertia (k * identity) -- every sphere-sphere contact and every contact of a sphere with the inert static world: the ball pit of
// config 4.  With those two facts the 39-float row collapses algebraically (nh_build_row, term by term):
//     na = nb = 0                          (p x n = 0)               => normal_velocity_to_normal_impulse = -1 / (1/m_a + 1/m_b)
//     ua = k_a alpha v, va = -k_a alpha u, ub = -k_b beta v, vb = k_b beta u     (n x u = -v, n x v = u for v = u x n)
//     friction_coefficient_x = _y = 1/m_a + 1/m_b + k_a alpha^2 + k_b beta^2,   friction_coefficient_z = 0
// so a row is 12 floats (48 B) instead of 40 and the tangent frame is recomputed from n exactly as nh_build_row computes it.  The
// level / colour scheduled path streams rows from HBM once per sweep, and 16 sweeps over 12 M contacts is what config 4 costs.
// These are the SAME formulas evaluated with exact zeros dropped and the isotropic inertia applied as a scalar; they round
// differently from the general forms in the last bits (k * (p x u) instead of a 3x3 product whose off-diagonal terms are ~1e-8 k), so
// they are used only in the default (colour-ordered) sweep -- already a different Gauss-Seidel order than the reference's --
// never under NH_FLAG_EXACT_ORDER.  tests/test_cpu_oracle.py::test_radial_forms_agree_with_general_forms bounds the difference.
struct nh_rowr {
	float n_x, n_y, n_z, bias;
	float alpha, beta, friction, normal_velocity_to_normal_impulse;
	float friction_coefficient, c_a, c_b, pad;        // c_a = k_a * alpha, c_b = k_b * beta: what nh_rowr_derive gives (the blocked solver, nh_blocks.h,
	                                                  // streams only the first eight floats and recomputes these three from the two bodies)
};

// the part of a radial row that follows from alpha, beta and the two bodies' inverse mass / isotropic inverse inertia k
NH_HD void nh_rowr_derive(float alpha, float beta, float a_mass_inverse, float b_mass_inverse, float k_a, float k_b, float& friction_coefficient, float& c_a, float& c_b) {
	float mass_inverse = a_mass_inverse + b_mass_inverse;
	friction_coefficient = mass_inverse + k_a * alpha * alpha + k_b * beta * beta;
	c_a = k_a * alpha; c_b = k_b * beta;
}

// the smoothly varying tangent basis of nh_build_row (nudge.cpp:4445-4456)
NH_HD void nh_tangents(float nx, float ny, float nz, nh_f3& u, nh_f3& v) {
	float s = nh_abs(nx);
	float u_x = nz * s;
	float u_y = u_x - nz;
	float u_z = nh_madd(nx - ny, s, ny);
	u_x = nh_neg(u_x);
	float f = nh_rsqrt_s(u_x * u_x + u_y * u_y + u_z * u_z);
	u_x *= f; u_y *= f; u_z *= f;
	u = nh_make3(u_x, u_y, u_z);
	v = nh_make3(u_y * nz - u_z * ny, u_z * nx - u_x * nz, u_x * ny - u_y * nx);
}

// Is side `p` (lever arm of a body with inverse inertia diag(ix, iy, iz)) radial?  `inert`: the body is the inert static world,
// whose lever arm never matters.  The perpendicular part of a sphere's lever arm is rounding noise of (contact point - centre):
// anything below 1e-3 of the arm counts as parallel (and is dropped).
NH_HD bool nh_side_is_radial(float px, float py, float pz, float nx, float ny, float nz, float ix, float iy, float iz, bool inert) {
	if (inert) return true;
	if (!(nh_asuint(ix) == nh_asuint(iy) && nh_asuint(iy) == nh_asuint(iz))) return false;
	float cx = py * nz - pz * ny, cy = pz * nx - px * nz, cz = px * ny - py * nx;
	return cx * cx + cy * cy + cz * cz <= 1e-6f * (px * px + py * py + pz * pz);
}

NH_HD void nh_build_rowr(const nh_contact_in& c, nh_f3 apos, nh_f3 bpos, float k_a, float k_b, float a_mass_inverse, float b_mass_inverse,
                         bool a_inert, bool b_inert, nh_rowr& r) {
	float alpha = a_inert ? 0.0f : (c.px - apos.x) * c.nx + (c.py - apos.y) * c.ny + (c.pz - apos.z) * c.nz;
	float beta = b_inert ? 0.0f : (c.px - bpos.x) * c.nx + (c.py - bpos.y) * c.ny + (c.pz - bpos.z) * c.nz;
	float mass_inverse = a_mass_inverse + b_mass_inverse;
	bool nonzero = (mass_inverse < 0.0f) || (mass_inverse > 0.0f);
	float nvtni = nonzero ? nh_neg(nh_recip_s(mass_inverse)) : 0.0f;
	r.n_x = c.nx; r.n_y = c.ny; r.n_z = c.nz;
	r.bias = (-NH_BIAS_FACTOR) * nh_max(c.penetration - NH_ALLOWED_PENETRATION, 0.0f) * nvtni;
	r.alpha = alpha; r.beta = beta;
	r.normal_velocity_to_normal_impulse = nvtni;
	nh_rowr_derive(alpha, beta, a_mass_inverse, b_mass_inverse, k_a, k_b, r.friction_coefficient, r.c_a, r.c_b);
	r.friction = c.friction; r.pad = 0.0f;
}

NH_HD void nh_warm_start_r(const nh_rowr& r, float cix, float ciy, float ciz, nh_vel& a, nh_vel& b, nh_state& st) {
	nh_f3 u, v;
	nh_tangents(r.n_x, r.n_y, r.n_z, u, v);
	float normal_impulse = nh_max(r.n_x * cix + r.n_y * ciy + r.n_z * ciz, 0.0f);
	float max_friction_impulse = normal_impulse * r.friction;
	float fx = u.x * cix + u.y * ciy + u.z * ciz;
	float fy = v.x * cix + v.y * ciy + v.z * ciz;
	float scale = fx * fx + fy * fy;
	scale = nh_rsqrt_s(scale);
	scale = scale * max_friction_impulse;
	scale = nh_min(1.0f, scale);
	fx = fx * scale;
	fy = fy * scale;
	float lx = fx * u.x + fy * v.x + r.n_x * normal_impulse;
	float ly = fx * u.y + fy * v.y + r.n_y * normal_impulse;
	float lz = fx * u.z + fy * v.z + r.n_z * normal_impulse;
	float wx = fx * v.x - fy * u.x, wy = fx * v.y - fy * u.y, wz = fx * v.z - fy * u.z;
	a.vx -= lx * a.mass_inverse; a.vy -= ly * a.mass_inverse; a.vz -= lz * a.mass_inverse;
	a.wx += r.c_a * wx; a.wy += r.c_a * wy; a.wz += r.c_a * wz;
	b.vx += lx * b.mass_inverse; b.vy += ly * b.mass_inverse; b.vz += lz * b.mass_inverse;
	b.wx -= r.c_b * wx; b.wy -= r.c_b * wy; b.wz -= r.c_b * wz;
	st.normal = normal_impulse; st.friction_x = fx; st.friction_y = fy;
}

// nh_apply_one for a radial row: p_a = alpha n, p_b = beta n  =>  w x p = s (w x n)
NH_HD void nh_apply_radial(const nh_rowr& r, nh_state& st, nh_vel& a, nh_vel& b) {
	nh_f3 u, v;
	nh_tangents(r.n_x, r.n_y, r.n_z, u, v);
	// relative velocity of the contact point: (v_b + w_b x p_b) - (v_a + w_a x p_a)
	float dwx = r.beta * b.wx - r.alpha * a.wx, dwy = r.beta * b.wy - r.alpha * a.wy, dwz = r.beta * b.wz - r.alpha * a.wz;
	float v_x = (b.vx - a.vx) + (dwy * r.n_z - dwz * r.n_y);
	float v_y = (b.vy - a.vy) + (dwz * r.n_x - dwx * r.n_z);
	float v_z = (b.vz - a.vz) + (dwx * r.n_y - dwy * r.n_x);

	float t_z = nh_madd(r.n_z, v_z, nh_madd(r.n_y, v_y, r.n_x * v_x));
	float t_x = nh_madd(v_z, u.z, nh_madd(v_y, u.y, v_x * u.x));
	float t_y = nh_madd(v_z, v.z, nh_madd(v_y, v.y, v_x * v.x));

	float old_normal_impulse = st.normal;
	float normal_impulse = nh_madd(r.normal_velocity_to_normal_impulse, t_z, r.bias + old_normal_impulse);
	float t_xx = t_x * t_x, t_yy = t_y * t_y;
	float tl2 = t_xx + t_yy;
	normal_impulse = nh_max(normal_impulse, 0.0f);
	t_x *= tl2;
	t_y *= tl2;
	st.normal = normal_impulse;
	float max_friction_impulse = normal_impulse * r.friction;
	normal_impulse = normal_impulse - old_normal_impulse;

	float friction_factor = nh_madd(t_yy, r.friction_coefficient, t_xx * r.friction_coefficient);
	friction_factor = nh_recip_s(friction_factor);
	friction_factor = nh_min(1e+6f, friction_factor);

	float old_fx = st.friction_x, old_fy = st.friction_y;
	float fx = old_fx - t_x * friction_factor;
	float fy = old_fy - t_y * friction_factor;
	float clamp = nh_rsqrt_s(fx * fx + fy * fy);
	clamp = clamp * max_friction_impulse;
	clamp = nh_min(1.0f, clamp);
	fx = fx * clamp;
	fy = fy * clamp;
	st.friction_x = fx;
	st.friction_y = fy;
	fx -= old_fx;
	fy -= old_fy;

	float lx = nh_madd(v.x, fy, nh_madd(u.x, fx, r.n_x * normal_impulse));
	float ly = nh_madd(v.y, fy, nh_madd(u.y, fx, r.n_y * normal_impulse));
	float lz = nh_madd(v.z, fy, nh_madd(u.z, fx, r.n_z * normal_impulse));
	float wx = fx * v.x - fy * u.x, wy = fx * v.y - fy * u.y, wz = fx * v.z - fy * u.z;

	float a_mass_inverse_neg = nh_neg(a.mass_inverse);
	a.vx = nh_madd(lx, a_mass_inverse_neg, a.vx);
	a.vy = nh_madd(ly, a_mass_inverse_neg, a.vy);
	a.vz = nh_madd(lz, a_mass_inverse_neg, a.vz);
	a.wx = nh_madd(r.c_a, wx, a.wx);
	a.wy = nh_madd(r.c_a, wy, a.wy);
	a.wz = nh_madd(r.c_a, wz, a.wz);
	a.w_unused = 0.0f;
	b.vx = nh_madd(lx, b.mass_inverse, b.vx);
	b.vy = nh_madd(ly, b.mass_inverse, b.vy);
	b.vz = nh_madd(lz, b.mass_inverse, b.vz);
	float c_b_neg = nh_neg(r.c_b);
	b.wx = nh_madd(c_b_neg, wx, b.wx);
	b.wy = nh_madd(c_b_neg, wy, b.wy);
	b.wz = nh_madd(c_b_neg, wz, b.wz);
	b.w_unused = 0.0f;
}

NH_HD void nh_export_impulse_r(const nh_rowr& r, const nh_state& st, float& ix, float& iy, float& iz) {
	nh_f3 u, v;
	nh_tangents(r.n_x, r.n_y, r.n_z, u, v);
	ix = st.normal * r.n_x + st.friction_x * u.x + st.friction_y * v.x;
	iy = st.normal * r.n_y + st.friction_x * u.y + st.friction_y * v.y;
	iz = st.normal * r.n_z + st.friction_x * u.z + st.friction_y * v.z;
}

// true iff body 0 qualifies as S above (all bits zero)
NH_HD bool nh_is_inert(const float* velocity, const float* angular_velocity, const float* inertia_inverse, float mass_inverse) {
	uint32_t acc = nh_asuint(mass_inverse);
	for (int k = 0; k < 3; ++k) acc |= nh_asuint(velocity[k]) | nh_asuint(angular_velocity[k]) | nh_asuint(inertia_inverse[k]);
	return acc == 0u;
}

// update_cached_impulses, one contact (nudge.cpp:4872-4882)
NH_HD void nh_export_impulse(const nh_row& r, const nh_state& st, float& ix, float& iy, float& iz) {
	ix = st.normal * r.n_x + st.friction_x * r.u_x + st.friction_y * r.v_x;
	iy = st.normal * r.n_y + st.friction_x * r.u_y + st.friction_y * r.v_y;
	iz = st.normal * r.n_z + st.friction_x * r.u_z + st.friction_y * r.v_z;
}

// advance, one body (nudge.cpp:4893-4924).  Returns the new idle counter.
NH_HD uint8_t nh_advance_body(float* pos, float* rot, const float* vel, const float* ang, uint8_t idle, float time_step) {
	float half_time_step = 0.5f * time_step;
	nh_f3 velocity = nh_make3(vel[0], vel[1], vel[2]);
	nh_f3 angular_velocity = nh_make3(ang[0], ang[1], ang[2]);
	if (nh_dot(velocity, velocity) < 1e-2f && nh_dot(angular_velocity, angular_velocity) < 1e-1f) {
		if (idle < 0xff) ++idle;
	} else {
		idle = 0;
	}
	nh_quat dr = { angular_velocity.x, angular_velocity.y, angular_velocity.z, 0.0f };
	nh_quat q = { rot[0], rot[1], rot[2], rot[3] };
	dr = nh_qmul(dr, q);
	dr.x *= half_time_step; dr.y *= half_time_step; dr.z *= half_time_step; dr.s *= half_time_step;

	pos[0] += velocity.x * time_step;
	pos[1] += velocity.y * time_step;
	pos[2] += velocity.z * time_step;

	float rx = rot[0] + dr.x, ry = rot[1] + dr.y, rz = rot[2] + dr.z, rs = rot[3] + dr.s;
	float f = nh_rsqrt_s(rs * rs + rx * rx + ry * ry + rz * rz);   // nudge.cpp:1128-1133
	rot[0] = rx * f; rot[1] = ry * f; rot[2] = rz * f; rot[3] = rs * f;
	return idle;
}

// gravity + damping of one body: the caller-side loop of the sample app (example/main.cpp:290-305)
NH_HD void nh_gravity_damping(float* vel, float* ang, float gx_dt, float gy_dt, float gz_dt, float damping) {
	vel[0] -= gx_dt; vel[1] -= gy_dt; vel[2] -= gz_dt;
	vel[0] *= damping; vel[1] *= damping; vel[2] *= damping;
	ang[0] *= damping; ang[1] *= damping; ang[2] *= damping;
}

#endif
