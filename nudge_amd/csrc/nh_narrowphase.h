// nh_narrowphase.h -- per-pair narrowphase: box-box SAT + face clipping + edge-edge, box-sphere,
// sphere-sphere.  One GPU lane handles one collider pair from the first SAT axis to the emitted contacts
// (the reference streams 4 pairs per SSE iteration through three compaction phases instead,
// reference nudge.cpp:1177-2487); the arithmetic per pair keeps the reference's operation order so that
// contact counts and feature tags are identical and floats match the exact-math oracle.
//
// Conventions shared with the reference:
//   * a pair arrives as (a, b) with a = the collider LATER in Morton order (nudge.cpp:1202-1203, 3495)
//   * emitted contacts are canonicalised so that "a" is the collider with the LARGER tag
//     (nudge.cpp:2074-2087 for faces, 2131-2132 for edges); normal points from a to b
//   * feature word = a-features in one 16-bit half, b-features in the other
//
// DERIVED WORK NOTICE.  The per-contact / per-pair arithmetic in this file follows rasmusbarr/nudge's nudge.cpp statement by statement
// (nh_box_box_eval: nudge.cpp:1177-2487 incl. its feature-tag tables; nh_sphere_sphere / nh_box_sphere: nudge.cpp:2489-2604): operation order and fused-multiply-add placement are the numeric contract of bit-exact parity
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
#ifndef NH_NARROWPHASE_H
#define NH_NARROWPHASE_H

#include "nh_math.h"

struct nh_xform {            // world transform of a collider (layout of nudge.h:34-38)
	float px, py, pz;
	uint32_t body;
	float qx, qy, qz, qs;
};

struct nh_contact_out {      // nudge.h:61-66 + feature word
	float px, py, pz, penetration;
	float nx, ny, nz, friction;
	uint32_t feature;
};

struct nh_pair_result {
	int count;               // number of contacts written to out[]
	int swapped;             // 1: canonical "a" is the pair's second collider (input b)
};

// Result of the box-box test before contacts are materialised.  Every array is indexed with compile-time
// constants only (fully unrolled loops), so on the GPU it lives in registers, not in scratch memory.
struct nh_bb_result {
	int kind;                // 0: separated, 1: one edge-edge contact (in `edge`), 2: face clipping (mask over 16 candidates)
	int swapped;
	unsigned mask;
	unsigned tag_swap;
	nh_contact_out edge;
	float lx[16], ly[16], lz[16], penetration[16];   // candidate points in a's local frame
	uint32_t tags[16];
	float w0[3], w1[3], w2[3], wn[3], apos[3];
};

NH_HD int nh_bb_count(const nh_bb_result& r) {
	if (r.kind == 1) return 1;
	if (r.kind == 2) return __builtin_popcount(r.mask);
	return 0;
}

// materialise candidate `index` (0..15) of a face result (nudge.cpp:2095-2108)
NH_HD void nh_bb_contact(const nh_bb_result& r, int index, float lx, float ly, float lz, float pen, uint32_t tag, nh_contact_out& o) {
	(void)index;
	o.px = r.w0[0] * lx + r.w1[0] * ly + r.w2[0] * lz + r.apos[0];
	o.py = r.w0[1] * lx + r.w1[1] * ly + r.w2[1] * lz + r.apos[1];
	o.pz = r.w0[2] * lx + r.w1[2] * ly + r.w2[2] * lz + r.apos[2];
	o.penetration = pen;
	o.nx = r.wn[0]; o.ny = r.wn[1]; o.nz = r.wn[2];
	o.friction = 0.5f;
	o.feature = r.tag_swap ? ((tag >> 16) | (tag << 16)) : tag;
}

// ---------------------------------------------------------------------------------------------------
// Phase-1/2 shared: relative rotation matrix "a to b" as three vectors vx,vy,vz (nudge.cpp:1228-1268).
struct nh_rel { float m[9]; };   // m[r*3+c], r = vector (x,y,z), c = component: layout of a_to_b[] (nudge.cpp:1507-1519)

NH_HD nh_rel nh_relative(nh_quat a, nh_quat b) {
	// t = cross(b.v, a.v)
	float tx = b.y * a.z - b.z * a.y;
	float ty = b.z * a.x - b.x * a.z;
	float tz = b.x * a.y - b.y * a.x;
	float rx = a.x * b.s - b.x * a.s - tx;
	float ry = a.y * b.s - b.y * a.s - ty;
	float rz = a.z * b.s - b.z * a.s - tz;
	float rs = (a.x * b.x + a.y * b.y + a.z * b.z + a.s * b.s);
	float kx = rx + rx, ky = ry + ry, kz = rz + rz;
	float xx = kx * rx, yy = ky * ry, zz = kz * rz;
	float xy = kx * ry, xz = kx * rz, yz = ky * rz;
	float sx = kx * rs, sy = ky * rs, sz = kz * rs;
	nh_rel r;
	r.m[0] = 1.0f - yy - zz; r.m[1] = xy + sz;        r.m[2] = xz - sy;
	r.m[3] = xy - sz;        r.m[4] = 1.0f - xx - zz; r.m[5] = yz + sx;
	r.m[6] = xz + sy;        r.m[7] = yz - sx;        r.m[8] = 1.0f - xx - yy;
	return r;
}

// b_offset: position of b's centre in a's frame (nudge.cpp:1340-1349 / 1554-1564); delta = a.pos - b.pos
NH_HD nh_f3 nh_b_offset(nh_quat a, nh_f3 delta) {
	nh_f3 av = nh_qv(a);
	nh_f3 t = nh_cross(delta, av);
	t.x += t.x; t.y += t.y; t.z += t.z;
	nh_f3 u = nh_cross(av, t);
	return nh_make3(u.x - delta.x - a.s * t.x, u.y - delta.y - a.s * t.y, u.z - delta.z - a.s * t.z);
}

// a_offset: position of a's centre in b's frame (nudge.cpp:1324-1334)
NH_HD nh_f3 nh_a_offset(nh_quat b, nh_f3 delta) {
	nh_f3 bv = nh_qv(b);
	nh_f3 t = nh_cross(bv, delta);
	t.x += t.x; t.y += t.y; t.z += t.z;
	nh_f3 u = nh_cross(bv, t);
	return nh_make3(u.x + delta.x - b.s * t.x, u.y + delta.y - b.s * t.y, u.z + delta.z - b.s * t.z);
}

// ---------------------------------------------------------------------------------------------------
// Edge-edge contact (phase 3, nudge.cpp:2143-2484).  a = collider with the larger tag.
NH_HD void nh_box_box_edge(const nh_xform& A, const nh_xform& B, const float* sa, const float* sb,
                           uint32_t edge, float penetration, nh_contact_out* out) {
	float ab[9], bb[9];   // basis[r*3+c]: r = local axis, c = world component (nudge.cpp:2190-2251)
	{
		float kx = A.qx + A.qx, ky = A.qy + A.qy, kz = A.qz + A.qz;
		float xx = kx * A.qx, yy = ky * A.qy, zz = kz * A.qz, xy = kx * A.qy, xz = kx * A.qz, yz = ky * A.qz;
		float sx = kx * A.qs, sy = ky * A.qs, sz = kz * A.qs;
		ab[0] = 1.0f - yy - zz; ab[1] = xy + sz; ab[2] = xz - sy;
		ab[3] = xy - sz; ab[4] = 1.0f - xx - zz; ab[5] = yz + sx;
		ab[6] = xz + sy; ab[7] = yz - sx; ab[8] = 1.0f - xx - yy;
	}
	{
		float kx = B.qx + B.qx, ky = B.qy + B.qy, kz = B.qz + B.qz;
		float xx = kx * B.qx, yy = ky * B.qy, zz = kz * B.qz, xy = kx * B.qy, xz = kx * B.qz, yz = ky * B.qz;
		float sx = kx * B.qs, sy = ky * B.qs, sz = kz * B.qs;
		bb[0] = 1.0f - yy - zz; bb[1] = xy + sz; bb[2] = xz - sy;
		bb[3] = xy - sz; bb[4] = 1.0f - xx - zz; bb[5] = yz + sx;
		bb[6] = xz + sy; bb[7] = yz - sx; bb[8] = 1.0f - xx - yy;
	}
	// select edge directions: bit0 -> y, bit1 -> z (nudge.cpp:2258-2278)
	unsigned ea = edge & 0xffffu, eb = edge >> 16;
	unsigned ra = (ea & 2u) ? 2u : ((ea & 1u) ? 1u : 0u);
	unsigned rb = (eb & 2u) ? 2u : ((eb & 1u) ? 1u : 0u);
	float ux = ab[ra * 3 + 0], uy = ab[ra * 3 + 1], uz = ab[ra * 3 + 2];
	float vx = bb[rb * 3 + 0], vy = bb[rb * 3 + 1], vz = bb[rb * 3 + 2];

	float nx = uy * vz - uz * vy;
	float ny = uz * vx - ux * vz;
	float nz = ux * vy - uy * vx;

	float dx = B.px - A.px, dy = B.py - A.py, dz = B.pz - A.pz;
	uint32_t flip = nh_asuint(nx * dx + ny * dy + nz * dz) & 0x80000000u;
	nx = nh_xorf(nx, flip); ny = nh_xorf(ny, flip); nz = nh_xorf(nz, flip);

	float asx = ab[0] * nx + ab[1] * ny + ab[2] * nz;
	float asy = ab[3] * nx + ab[4] * ny + ab[5] * nz;
	float asz = ab[6] * nx + ab[7] * ny + ab[8] * nz;
	float bsx = bb[0] * nx + bb[1] * ny + bb[2] * nz;
	float bsy = bb[3] * nx + bb[4] * ny + bb[5] * nz;
	float bsz = bb[6] * nx + bb[7] * ny + bb[8] * nz;
	uint32_t sax = nh_asuint(asx) & 0x80000000u, say = nh_asuint(asy) & 0x80000000u, saz = nh_asuint(asz) & 0x80000000u;
	uint32_t sbx = nh_asuint(bsx) & 0x80000000u, sby = nh_asuint(bsy) & 0x80000000u, sbz = nh_asuint(bsz) & 0x80000000u;

	// feature tag (nudge.cpp:2378-2390)
	uint32_t edge_x = (sax >> 31) | ((sbx ^ 0x80000000u) >> 15);
	uint32_t edge_y = (say >> 30) | ((sby ^ 0x80000000u) >> 14);
	uint32_t edge_z = (saz >> 29) | ((sbz ^ 0x80000000u) >> 13);
	uint32_t ewl = ((ea + 1u) + (ea >> 1)) & 0xffffu, ewh = ((eb + 1u) + (eb >> 1)) & 0xffffu;   // 1 << edge per half
	uint32_t edge_w = ewl | (ewh << 16);
	uint32_t tag_hi = edge_x | edge_y | edge_z | edge_w;
	uint32_t tag_lo = ~edge_w & tag_hi;
	uint32_t tag = tag_lo | (tag_hi << 8);

	float sax_f = nh_xorf(sa[0], sax), say_f = nh_xorf(sa[1], say), saz_f = nh_xorf(sa[2], saz);
	float sbx_f = nh_xorf(sb[0], sbx), sby_f = nh_xorf(sb[1], sby), sbz_f = nh_xorf(sb[2], sbz);

	float a0 = ab[0] * sax_f, a1 = ab[1] * sax_f, a2 = ab[2] * sax_f;
	float a3 = ab[3] * say_f, a4 = ab[4] * say_f, a5 = ab[5] * say_f;
	float a6 = ab[6] * saz_f, a7 = ab[7] * saz_f, a8 = ab[8] * saz_f;
	float b0 = bb[0] * sbx_f, b1 = bb[1] * sbx_f, b2 = bb[2] * sbx_f;
	float b3 = bb[3] * sby_f, b4 = bb[4] * sby_f, b5 = bb[5] * sby_f;
	float b6 = bb[6] * sbz_f, b7 = bb[7] * sbz_f, b8 = bb[8] * sbz_f;

	float cax = a0 + a3 + a6 + A.px, cay = a1 + a4 + a7 + A.py, caz = a2 + a5 + a8 + A.pz;
	float cbx = b0 + b3 + b6 - B.px, cby = b1 + b4 + b7 - B.py, cbz = b2 + b5 + b8 - B.pz;   // negated, as in the reference

	float ox = cax + cbx, oy = cay + cby, oz = caz + cbz;
	float ia = ux * ux + uy * uy + uz * uz;
	float ib = ux * vx + uy * vy + uz * vz;
	float ic = vx * vx + vy * vy + vz * vz;
	float id = ox * ux + oy * uy + oz * uz;
	float ie = ox * vx + oy * vy + oz * vz;
	float ir = 0.5f / (ia * ic - ib * ib);
	float s_a = (ib * ie - ic * id) * ir;
	float s_b = (ia * ie - ib * id) * ir;

	out->px = (cax - cbx) * 0.5f + ux * s_a + vx * s_b;
	out->py = (cay - cby) * 0.5f + uy * s_a + vy * s_b;
	out->pz = (caz - cbz) * 0.5f + uz * s_a + vz * s_b;

	float f = nh_rsqrt_guarded(nx * nx + ny * ny + nz * nz);
	out->nx = nx * f; out->ny = ny * f; out->nz = nz * f;
	out->penetration = penetration;
	out->friction = 0.5f;
	out->feature = tag;
}

// ---------------------------------------------------------------------------------------------------
// Full box-box test of one pair: SAT over 15 axes, then either one edge-edge contact or up to 16 clipped candidates
// (8 is the geometric maximum) described by res.mask.  A/B, sa/sb, tag_a/tag_b describe the pair in ARRIVAL order
// (A = later in Morton order).
NH_HD void nh_box_box_eval(nh_xform A, nh_xform B, const float* size_a, const float* size_b,
                           uint32_t tag_a, uint32_t tag_b, nh_bb_result& res) {
	res.kind = 0; res.swapped = 0; res.mask = 0; res.tag_swap = 0;
	float sa[3] = { size_a[0], size_a[1], size_a[2] };
	float sb[3] = { size_b[0], size_b[1], size_b[2] };
	int swapped = 0;          // A/B currently exchanged relative to arrival order
	float face_penetration;
	unsigned a_face;

	// ---- phase 1: six face axes (nudge.cpp:1214-1390) ----
	{
		nh_quat qa = { A.qx, A.qy, A.qz, A.qs }, qb = { B.qx, B.qy, B.qz, B.qs };
		nh_rel r = nh_relative(qa, qb);
		float vxx = nh_abs(r.m[0]), vxy = nh_abs(r.m[1]), vxz = nh_abs(r.m[2]);
		float vyx = nh_abs(r.m[3]), vyy = nh_abs(r.m[4]), vyz = nh_abs(r.m[5]);
		float vzx = nh_abs(r.m[6]), vzy = nh_abs(r.m[7]), vzz = nh_abs(r.m[8]);

		float pax = sb[0] + vxx * sa[0] + vyx * sa[1] + vzx * sa[2];
		float pay = sb[1] + vxy * sa[0] + vyy * sa[1] + vzy * sa[2];
		float paz = sb[2] + vxz * sa[0] + vyz * sa[1] + vzz * sa[2];
		float pbx = sa[0] + vxx * sb[0] + vxy * sb[1] + vxz * sb[2];
		float pby = sa[1] + vyx * sb[0] + vyy * sb[1] + vyz * sb[2];
		float pbz = sa[2] + vzx * sb[0] + vzy * sb[1] + vzz * sb[2];

		nh_f3 delta = nh_make3(A.px - B.px, A.py - B.py, A.pz - B.pz);
		nh_f3 ao = nh_a_offset(qb, delta);
		pax -= nh_abs(ao.x); pay -= nh_abs(ao.y); paz -= nh_abs(ao.z);
		nh_f3 bo = nh_b_offset(qa, delta);
		pbx -= nh_abs(bo.x); pby -= nh_abs(bo.y); pbz -= nh_abs(bo.z);

		float payz = nh_min(pay, paz), pbyz = nh_min(pby, pbz);
		float pa = nh_min(pax, payz), pb = nh_min(pbx, pbyz);
		float p = nh_min(pa, pb);

		unsigned aface = (payz == pa ? 1u : 0u) + (paz == pa ? 1u : 0u);
		unsigned bface = (pbyz == pb ? 1u : 0u) + (pbz == pb ? 1u : 0u);
		bool swap = (pa == p);
		if (!(p > 0.0f)) return;
		face_penetration = p;
		a_face = swap ? aface : bface;
		if (swap) {
			nh_xform t = A; A = B; B = t;
			float t0 = sa[0], t1 = sa[1], t2 = sa[2];
			sa[0] = sb[0]; sa[1] = sb[1]; sa[2] = sb[2];
			sb[0] = t0; sb[1] = t1; sb[2] = t2;
			uint32_t tt = tag_a; tag_a = tag_b; tag_b = tt;
			swapped = 1;
		}
	}

	// ---- phase 2: nine edge axes (nudge.cpp:1451-1675) ----
	nh_quat qa = { A.qx, A.qy, A.qz, A.qs }, qb = { B.qx, B.qy, B.qz, B.qs };
	nh_rel rel = nh_relative(qa, qb);
	const float* a_to_b = rel.m;
	nh_f3 delta = nh_make3(A.px - B.px, A.py - B.py, A.pz - B.pz);
	nh_f3 b_offset = nh_b_offset(qa, delta);

	float penetration = face_penetration;
	unsigned a_edge = 0, b_edge = 0;
	{
		float epa[9], epb[9];
		for (unsigned i = 0; i < 3; ++i) {
			float acx = a_to_b[0 * 3 + i], acy = a_to_b[1 * 3 + i], acz = a_to_b[2 * 3 + i];
			float bcx = a_to_b[i * 3 + 0], bcy = a_to_b[i * 3 + 1], bcz = a_to_b[i * 3 + 2];
			float ac2x = acx * acx, ac2y = acy * acy, ac2z = acz * acz;
			float bc2x = bcx * bcx, bc2y = bcy * bcy, bc2z = bcz * bcz;
			float aacx = nh_abs(acx), aacy = nh_abs(acy), aacz = nh_abs(acz);
			float abcx = nh_abs(bcx), abcy = nh_abs(bcy), abcz = nh_abs(bcz);
			float r_a0 = ac2y + ac2z, r_a1 = ac2z + ac2x, r_a2 = ac2x + ac2y;
			float r_b0 = bc2y + bc2z, r_b1 = bc2z + bc2x, r_b2 = bc2x + bc2y;
			const float nan_threshold = 1e-3f;
			// rsqrt OR all-ones when r <= threshold -> NaN (nudge.cpp:1613-1619)
			const float qnan = nh_asfloat(0xffffffffu);
			r_a0 = (r_a0 <= nan_threshold) ? qnan : nh_rsqrt_guarded((r_a0 <= nan_threshold) ? 1.0f : r_a0);
			r_a1 = (r_a1 <= nan_threshold) ? qnan : nh_rsqrt_guarded((r_a1 <= nan_threshold) ? 1.0f : r_a1);
			r_a2 = (r_a2 <= nan_threshold) ? qnan : nh_rsqrt_guarded((r_a2 <= nan_threshold) ? 1.0f : r_a2);
			r_b0 = (r_b0 <= nan_threshold) ? qnan : nh_rsqrt_guarded((r_b0 <= nan_threshold) ? 1.0f : r_b0);
			r_b1 = (r_b1 <= nan_threshold) ? qnan : nh_rsqrt_guarded((r_b1 <= nan_threshold) ? 1.0f : r_b1);
			r_b2 = (r_b2 <= nan_threshold) ? qnan : nh_rsqrt_guarded((r_b2 <= nan_threshold) ? 1.0f : r_b2);
			float pa0 = aacy * sa[2] + aacz * sa[1];
			float pa1 = aacz * sa[0] + aacx * sa[2];
			float pa2 = aacx * sa[1] + aacy * sa[0];
			float pb0 = abcy * sb[2] + abcz * sb[1];
			float pb1 = abcz * sb[0] + abcx * sb[2];
			float pb2 = abcx * sb[1] + abcy * sb[0];
			float o0 = nh_abs(acy * b_offset.z - acz * b_offset.y);
			float o1 = nh_abs(acz * b_offset.x - acx * b_offset.z);
			float o2 = nh_abs(acx * b_offset.y - acy * b_offset.x);
			epa[i * 3 + 0] = (pa0 - o0) * r_a0;
			epa[i * 3 + 1] = (pa1 - o1) * r_a1;
			epa[i * 3 + 2] = (pa2 - o2) * r_a2;
			epb[i * 3 + 0] = pb0 * r_b0;
			epb[i * 3 + 1] = pb1 * r_b1;
			epb[i * 3 + 2] = pb2 * r_b2;
		}
		for (unsigned i = 0; i < 3; ++i) {
			for (unsigned j = 0; j < 3; ++j) {
				float p = epa[i * 3 + j] + epb[j * 3 + i];
				bool m = penetration > p;
				penetration = nh_min(penetration, p);
				if (m) { a_edge = j; b_edge = i; }
			}
		}
	}

	bool is_edge = face_penetration > penetration + 1e-3f;
	bool overlapping = penetration > 0.0f;
	if (!overlapping) return;

	if (is_edge) {
		// canonicalise: a = larger tag (nudge.cpp:2127-2132), then phase 3
		uint32_t feature;
		if (tag_a > tag_b) {
			feature = a_edge | (b_edge << 16);
			nh_box_box_edge(A, B, sa, sb, feature, penetration, &res.edge);
		} else {
			feature = b_edge | (a_edge << 16);
			nh_box_box_edge(B, A, sb, sa, feature, penetration, &res.edge);
			swapped ^= 1;
		}
		res.kind = 1;
		res.swapped = swapped;
		return;
	}

	// ---- face-face clipping (nudge.cpp:1678-2112) ----
	float dirs[3] = { nh_abs(a_to_b[a_face * 3 + 0]), nh_abs(a_to_b[a_face * 3 + 1]), nh_abs(a_to_b[a_face * 3 + 2]) };
	float c0[3] = { a_to_b[0], a_to_b[3], a_to_b[6] };
	float c1[3] = { a_to_b[1], a_to_b[4], a_to_b[7] };
	float c2[3] = { a_to_b[2], a_to_b[5], a_to_b[8] };
	float bo[3] = { b_offset.x, b_offset.y, b_offset.z };

	// most aligned face of b (nudge.cpp:1719-1745)
	bool m1 = dirs[1] >= nh_max(dirs[2], dirs[0]);
	bool m2 = dirs[2] >= nh_max(dirs[1], dirs[0]);
	for (int k = 0; k < 3; ++k) { c0[k] *= sb[0]; c1[k] *= sb[1]; c2[k] *= sb[2]; }
	unsigned b_face = 0;
	if (m2) {
		for (int k = 0; k < 3; ++k) { float t = c0[k]; c0[k] = c2[k]; c2[k] = c1[k]; c1[k] = t; }
		b_face = 2;
	} else if (m1) {
		for (int k = 0; k < 3; ++k) { float t = c0[k]; c0[k] = c1[k]; c1[k] = c2[k]; c2[k] = t; }
		b_face = 1;
	}
	float c[3] = { c0[0], c0[1], c0[2] };
	const float* dxv = c1;
	const float* dyv = c2;

	unsigned b_positive_face_bit = ((nh_asuint(bo[a_face]) ^ nh_asuint(c[a_face])) >> 31);
	unsigned b_offset_neg = nh_signbit(bo[a_face]);
	if (!b_positive_face_bit) { c[0] = nh_neg(c[0]); c[1] = nh_neg(c[1]); c[2] = nh_neg(c[2]); }
	c[0] += bo[0]; c[1] += bo[1]; c[2] += bo[2];

	// quads[axis] = { a.size, c, dx, dy } ; transformed axes (nudge.cpp:1764-1778)
	unsigned AX = (a_face + 1) % 3, AY = (a_face + 2) % 3, AZ = a_face;
	float sx = sa[AX], sy = sa[AY], sz_ = sa[AZ];
	float cx = c[AX], cy = c[AY], cz = c[AZ];
	float dxx = dxv[AX], dxy_ = dxv[AY], dxz = dxv[AZ];
	float dyx = dyv[AX], dyy = dyv[AY], dyz = dyv[AZ];

	float support_x[16], support_y[16], support_z[16];
	float* penetrations = res.penetration;
	uint32_t* support_tags = res.tags;
	unsigned mask;
	{
		// sign patterns over the 4 lanes
		const uint32_t N = 0x80000000u, P = 0u;
		const uint32_t s_npnp[4] = { N, P, N, P }, s_pnpn[4] = { P, N, P, N }, s_nnpp[4] = { N, N, P, P };
		bool mask0[4], mask1[4];
		float k0 = cx * dyy - cy * dyx;
		float k1 = cx * dxy_ - cy * dxx;
		float k2 = dxx * dyy - dxy_ * dyx;
		float ox = k0, oy = k1, delta_max = nh_abs(k2);
		float sdxy0 = dxx * sy, sdxy1 = dxy_ * sx, sdxy2 = dyx * sy, sdxy3 = dyy * sx;
		for (int l = 0; l < 4; ++l) {
			float corner0x = nh_xorf(sx, s_pnpn[l]);
			float corner0y = nh_xorf(sy, s_nnpp[l]);
			float corner1x = cx + nh_xorf(dxx, s_npnp[l]) + nh_xorf(dyx, s_nnpp[l]);
			float corner1y = cy + nh_xorf(dxy_, s_npnp[l]) + nh_xorf(dyy, s_nnpp[l]);
			float delta_x = ox + nh_xorf(sdxy2, s_nnpp[l]) + nh_xorf(sdxy3, s_npnp[l]);
			float delta_y = oy + nh_xorf(sdxy0, s_nnpp[l]) + nh_xorf(sdxy1, s_npnp[l]);
			bool inside_x = nh_abs(corner1x) <= sx;
			bool inside_y = nh_abs(corner1y) <= sy;
			mask0[l] = nh_max(nh_abs(delta_x), nh_abs(delta_y)) <= delta_max;
			mask1[l] = inside_x && inside_y;
			support_x[l] = corner0x; support_y[l] = corner0y;
			support_x[4 + l] = corner1x; support_y[4 + l] = corner1y;
		}
		// Don't allow edge intersections if both vertices are inside (nudge.cpp:1834-1836).
		bool em[8] = {
			mask0[3] && mask0[1], mask0[2] && mask0[0], mask0[0] && mask0[1], mask0[2] && mask0[3],
			mask1[1] && mask1[0], mask1[3] && mask1[2], mask1[2] && mask1[0], mask1[3] && mask1[1],
		};

		// edges of quad b against the bounds of quad a (nudge.cpp:1844-1898)
		const float dxyv[4] = { dxx, dxy_, dyx, dyy };
		float rdxy[4] = { 1.0f / dxyv[0], 1.0f / dxyv[1], 1.0f / dxyv[2], 1.0f / dxyv[3] };
		const int i0022[4] = { 0, 0, 2, 2 }, i1133[4] = { 1, 1, 3, 3 }, i2200[4] = { 2, 2, 0, 0 }, i3311[4] = { 3, 3, 1, 1 };
		unsigned edge_axis_near = 0, edge_axis_far = 0;
		bool mask_a[4], mask_b[4];
		for (int l = 0; l < 4; ++l) {
			float offset_x = dxyv[i0022[l]], offset_y = dxyv[i1133[l]];
			float pivot_x = cx + nh_xorf(dxyv[i2200[l]], s_npnp[l]);
			float pivot_y = cy + nh_xorf(dxyv[i3311[l]], s_npnp[l]);
			float pos_x = nh_copysign_or(sx, offset_x);
			float pos_y = nh_copysign_or(sy, offset_y);
			float rx = rdxy[i0022[l]], ry = rdxy[i1133[l]];
			float near_x = (pos_x + pivot_x) * rx, far_x = (pos_x - pivot_x) * rx;
			float near_y = (pos_y + pivot_y) * ry, far_y = (pos_y - pivot_y) * ry;
			float a = nh_min(1.0f, near_x), b = nh_min(1.0f, far_x);
			if (a > near_y) edge_axis_near |= 1u << l;
			if (b > far_y) edge_axis_far |= 1u << l;
			a = nh_min(a, near_y); b = nh_min(b, far_y);
			support_x[8 + l] = pivot_x - offset_x * a;
			support_y[8 + l] = pivot_y - offset_y * a;
			support_x[12 + l] = pivot_x + offset_x * b;
			support_y[12 + l] = pivot_y + offset_y * b;
			bool m = (a + b) > 0.0f;
			mask_a[l] = !(a == 1.0f) && m;     // _mm_cmpneq_ps is unordered: true on NaN (nudge.cpp:1886)
			mask_b[l] = !(b == 1.0f) && m;
		}
		mask = 0;
		for (int l = 0; l < 4; ++l) {
			if (mask0[l]) mask |= 1u << l;
			if (mask1[l]) mask |= 1u << (4 + l);
			if (!em[l] && mask_a[l]) mask |= 1u << (8 + l);
			if (!em[4 + l] && mask_b[l]) mask |= 1u << (12 + l);
		}

		// vertex / edge labels (nudge.cpp:1902-1970)
		unsigned a_sign_face_bit = b_offset_neg ? (1u << a_face) : 0u;
		unsigned b_sign_face_bit = b_positive_face_bit ? 0u : (1u << b_face);
		unsigned a_vertices = 0x12003624u >> (3 - a_face);
		unsigned b_vertices = 0x00122436u >> (3 - b_face);
		unsigned a_face_bits = 0xffff0000u | a_sign_face_bit;
		unsigned b_face_bits = 0x0000ffffu | (b_sign_face_bit << 16);
		support_tags[0] = ((a_vertices >> 0) & 0x7) | a_face_bits;
		support_tags[1] = ((a_vertices >> 8) & 0x7) | a_face_bits;
		support_tags[2] = ((a_vertices >> 16) & 0x7) | a_face_bits;
		support_tags[3] = ((a_vertices >> 24) & 0x7) | a_face_bits;
		support_tags[4] = ((b_vertices << 16) & 0x70000) | b_face_bits;
		support_tags[5] = ((b_vertices << 8) & 0x70000) | b_face_bits;
		support_tags[6] = ((b_vertices >> 0) & 0x70000) | b_face_bits;
		support_tags[7] = ((b_vertices >> 8) & 0x70000) | b_face_bits;

		unsigned edge_axis_winding = nh_signbit(dxyv[0]) | (nh_signbit(dxyv[1]) << 1) | (nh_signbit(dxyv[2]) << 2) | (nh_signbit(dxyv[3]) << 3);
		unsigned a_near_edge[4], a_far_edge[4];
		for (int l = 0; l < 4; ++l) {
			unsigned yn = (edge_axis_near >> l) & 1u;
			a_near_edge[l] = yn * 2 + ((edge_axis_winding >> ((l < 2 ? 0 : 2) + yn)) & 1u);
		}
		edge_axis_winding ^= 0xf;
		for (int l = 0; l < 4; ++l) {
			unsigned yf = (edge_axis_far >> l) & 1u;
			a_far_edge[l] = yf * 2 + ((edge_axis_winding >> ((l < 2 ? 0 : 2) + yf)) & 1u);
		}
		uint64_t a_edge_map = 0x1200362424003612ull >> (3 - a_face);
		uint64_t b_edge_map = 0x2400361212003624ull >> (3 - b_face);
		unsigned face_bits = a_sign_face_bit | (a_sign_face_bit << 8) | (b_sign_face_bit << 16) | (b_sign_face_bit << 24);
		for (int l = 0; l < 4; ++l) {
			unsigned b_edge_l = ((unsigned)((b_edge_map >> (l << 4)) & 0x0707) << 16) | face_bits;
			support_tags[8 + l] = (unsigned)((a_edge_map >> (a_near_edge[l] << 4)) & 0x0707) | b_edge_l;
			support_tags[12 + l] = (unsigned)((a_edge_map >> (a_far_edge[l] << 4)) & 0x0707) | b_edge_l;
		}
	}

	// z-plane through face b (nudge.cpp:1973-2019)
	{
		float zn0 = dxy_ * dyz - dxz * dyy;
		float zn1 = dxz * dyx - dxx * dyz;
		float zn2 = dxx * dyy - dxy_ * dyx;
		float dotc = cx * zn0 + cy * zn1 + cz * zn2;
		float inv = 1.0f / zn2;
		float plane0 = nh_neg(zn0) * inv, plane1 = nh_neg(zn1) * inv, plane2 = dotc * inv;
		uint32_t z_sign = b_offset_neg ? 0x80000000u : 0u;
		float half_signed = nh_xorf(0.5f, z_sign);
		unsigned penetration_mask = 0;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			float x = support_x[i], y = support_y[i];
			float z = x * plane0 + y * plane1 + plane2;
			float pen = sz_ - nh_xorf(z, z_sign);
			z += pen * half_signed;
			if (pen > 0.0f) penetration_mask |= 1u << i;
			penetrations[i] = pen;
			support_z[i] = z;
		}
		mask &= penetration_mask;
	}

	// inverse transform + rotation of a to world (nudge.cpp:2021-2056)
	float w0[3], w1[3], w2[3];
	{
		float qx = A.qx, qy = A.qy, qz = A.qz, qs = A.qs;
		float kx = qx + qx, ky = qy + qy, kz = qz + qz, ks = nh_neg(qs + qs);
		w0[0] = nh_neg((ky * qy + kz * qz) - 1.0f);
		w0[1] = (kx * qy + kz * qs) - 0.0f;
		w0[2] = (kx * qz + ks * qy) - 0.0f;
		w1[0] = (kx * qy + ks * qz) - 0.0f;
		w1[1] = nh_neg((kz * qz + kx * qx) - 1.0f);
		w1[2] = (ky * qz + kx * qs) - 0.0f;
		w2[0] = (kx * qz + ky * qs) - 0.0f;
		w2[1] = (ky * qz + ks * qx) - 0.0f;
		w2[2] = nh_neg((kx * qx + ky * qy) - 1.0f);
	}
	const float* wn_src = a_face == 0 ? w0 : (a_face == 1 ? w1 : w2);
	float wn[3] = { wn_src[0], wn_src[1], wn_src[2] };
	if (b_offset_neg) { wn[0] = nh_neg(wn[0]); wn[1] = nh_neg(wn[1]); wn[2] = nh_neg(wn[2]); }

	unsigned tag_swap = 0;
	if (tag_b > tag_a) {
		tag_swap = 16;
		swapped ^= 1;
		wn[0] = nh_neg(wn[0]); wn[1] = nh_neg(wn[1]); wn[2] = nh_neg(wn[2]);
	}

	// local axes of the support blocks: X' = (a_face+1)%3, Y' = (a_face+2)%3, Z' = a_face (nudge.cpp:2021-2026);
	// selected with compile-time indices so the candidate arrays stay in registers
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		float bx = support_x[i], by = support_y[i], bz = support_z[i];
		res.lx[i] = a_face == 0 ? bz : (a_face == 1 ? by : bx);
		res.ly[i] = a_face == 0 ? bx : (a_face == 1 ? bz : by);
		res.lz[i] = a_face == 0 ? by : (a_face == 1 ? bx : bz);
	}
	for (int k = 0; k < 3; ++k) { res.w0[k] = w0[k]; res.w1[k] = w1[k]; res.w2[k] = w2[k]; res.wn[k] = wn[k]; }
	res.apos[0] = A.px; res.apos[1] = A.py; res.apos[2] = A.pz;
	res.kind = 2;
	res.mask = mask;
	res.tag_swap = tag_swap;
	res.swapped = swapped;
}

// Convenience wrapper (host-side unit tests): materialises the contacts into out[], in candidate order.
NH_HD nh_pair_result nh_box_box(nh_xform A, nh_xform B, const float* size_a, const float* size_b,
                                uint32_t tag_a, uint32_t tag_b, nh_contact_out* out) {
	nh_bb_result r;
	nh_box_box_eval(A, B, size_a, size_b, tag_a, tag_b, r);
	nh_pair_result res = { 0, r.swapped };
	if (r.kind == 1) { out[0] = r.edge; res.count = 1; }
	else if (r.kind == 2) {
		for (int index = 0; index < 16; ++index) {
			if (!((r.mask >> index) & 1u)) continue;
			nh_bb_contact(r, index, r.lx[index], r.ly[index], r.lz[index], r.penetration[index], r.tags[index], out[res.count++]);
		}
	}
	return res;
}

// ---------------------------------------------------------------------------------------------------
// sphere-sphere (nudge.cpp:2489-2521): a = pair's FIRST collider in Morton order; normal a -> b.
NH_HD int nh_sphere_sphere(float ra, float rb, const nh_xform& A, const nh_xform& B, nh_contact_out* out) {
	float r = ra + rb;
	nh_f3 dp = nh_make3(B.px - A.px, B.py - A.py, B.pz - A.pz);
	float l2 = nh_dot(dp, dp);
	if (l2 > r * r) return 0;
	nh_f3 n;
	float l = sqrtf(l2);
	if (l2 > 1e-4f) n = dp * (1.0f / l);
	else n = nh_make3(1.0f, 0.0f, 0.0f);
	nh_f3 p = nh_make3(A.px, A.py, A.pz) + n * (l - rb);
	out->px = p.x; out->py = p.y; out->pz = p.z;
	out->penetration = r - l;
	out->nx = n.x; out->ny = n.y; out->nz = n.z;
	out->friction = 0.5f;
	out->feature = 0;
	return 1;
}

// box-sphere (nudge.cpp:2523-2604): a = box, b = sphere; normal box -> sphere.
NH_HD int nh_box_sphere(const float* size, float radius, const nh_xform& A, const nh_xform& B, nh_contact_out* out) {
	nh_quat a_to_world = { A.qx, A.qy, A.qz, A.qs };
	nh_quat world_to_a = { -A.qx, -A.qy, -A.qz, A.qs };
	nh_f3 offset_b = nh_rotate(world_to_a, nh_make3(B.px - A.px, B.py - A.py, B.pz - A.pz));
	float dx = fabsf(offset_b.x), dy = fabsf(offset_b.y), dz = fabsf(offset_b.z);
	float w = size[0] + radius, h = size[1] + radius, d = size[2] + radius;
	if (dx >= w || dy >= h || dz >= d) return 0;
	nh_f3 n;
	float penetration;
	float r = radius;
	unsigned outside_x = dx > size[0], outside_y = dy > size[1], outside_z = dz > size[2];
	if (outside_x + outside_y + outside_z >= 2) {
		nh_f3 corner = nh_make3(
			outside_x ? (offset_b.x > 0.0f ? size[0] : -size[0]) : offset_b.x,
			outside_y ? (offset_b.y > 0.0f ? size[1] : -size[1]) : offset_b.y,
			outside_z ? (offset_b.z > 0.0f ? size[2] : -size[2]) : offset_b.z);
		nh_f3 dp = offset_b - corner;
		float l2 = nh_dot(dp, dp);
		if (l2 > r * r) return 0;
		float l = sqrtf(l2);
		float m = 1.0f / l;
		n = dp * m;
		penetration = r - l;
	} else if (w - dx < h - dy && w - dx < d - dz) {
		n = nh_make3(offset_b.x > 0.0f ? 1.0f : -1.0f, 0.0f, 0.0f);
		penetration = w - dx;
	} else if (h - dy < d - dz) {
		n = nh_make3(0.0f, offset_b.y > 0.0f ? 1.0f : -1.0f, 0.0f);
		penetration = h - dy;
	} else {
		n = nh_make3(0.0f, 0.0f, offset_b.z > 0.0f ? 1.0f : -1.0f);
		penetration = d - dz;
	}
	nh_f3 p = offset_b - n * r;
	p = nh_rotate(a_to_world, p) + nh_make3(A.px, A.py, A.pz);
	n = nh_rotate(a_to_world, n);
	out->px = p.x; out->py = p.y; out->pz = p.z;
	out->penetration = penetration;
	out->nx = n.x; out->ny = n.y; out->nz = n.z;
	out->friction = 0.5f;
	out->feature = 0;
	return 1;
}

#endif
