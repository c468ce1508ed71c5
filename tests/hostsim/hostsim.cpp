// tests/hostsim/hostsim.cpp -- CPU-side UNIT-TEST harness (never part of the product, never a fallback).
//
// Compiles the per-item device functions of nudge_amd/csrc/nh_*.h with g++ so that the arithmetic the HIP
// kernels execute can be diffed against the compiled reference (oracle/_ref) on this GPU-less dev box.
// The parallel plumbing of the kernels (sort / scan / compaction / scheduling) is NOT modelled here: this
// harness uses O(n^2) loops and takes the solver order from the oracle.
#include "../../nudge_amd/csrc/nh_math.h"
#include "../../nudge_amd/csrc/nh_narrowphase.h"
#include "../../nudge_amd/csrc/nh_solver.h"
#include <vector>
#include <algorithm>

struct Xf { float p[3]; uint32_t body; float q[4]; };

extern "C" {

// World transforms + AABBs + morton keys for all colliders (boxes first, then spheres).
// Returns number of contacts; outputs sorted by (key, feature).
int hs_collide(const Xf* body_xf, uint32_t nbox, const Xf* box_xf, const float* box_size /*4 per*/, const uint32_t* box_tags,
               uint32_t nsph, const Xf* sph_xf, const float* sph_radius, const uint32_t* sph_tags,
               uint32_t cap, float* out_contacts /*8 per*/, uint32_t* out_bodies /*2 per*/, uint64_t* out_keys, uint32_t* out_features) {
	uint32_t n = nbox + nsph;
	std::vector<nh_xform> xf(n);
	std::vector<float> mn(3 * n), mx(3 * n);
	std::vector<uint32_t> tags(n);
	for (uint32_t i = 0; i < n; ++i) {
		const Xf& l = i < nbox ? box_xf[i] : sph_xf[i - nbox];
		const Xf& b = body_xf[l.body];
		nh_quat bq = { b.q[0], b.q[1], b.q[2], b.q[3] }, lq = { l.q[0], l.q[1], l.q[2], l.q[3] };
		nh_f3 p = nh_rotate(bq, nh_make3(l.p[0], l.p[1], l.p[2])) + nh_make3(b.p[0], b.p[1], b.p[2]);
		nh_quat q = nh_qmul(bq, lq);
		xf[i].px = p.x; xf[i].py = p.y; xf[i].pz = p.z; xf[i].body = l.body;
		xf[i].qx = q.x; xf[i].qy = q.y; xf[i].qz = q.z; xf[i].qs = q.s;
		float sx, sy, sz;
		if (i < nbox) {
			nh_m33 m = nh_matrix(q);
			const float* s = box_size + 4 * i;
			nh_f3 c0 = m.c0 * s[0], c1 = m.c1 * s[1], c2 = m.c2 * s[2];
			sx = fabsf(c0.x) + fabsf(c1.x) + fabsf(c2.x);
			sy = fabsf(c0.y) + fabsf(c1.y) + fabsf(c2.y);
			sz = fabsf(c0.z) + fabsf(c1.z) + fabsf(c2.z);
			tags[i] = box_tags[i];
		} else {
			sx = sy = sz = sph_radius[i - nbox];
			tags[i] = sph_tags[i - nbox];
		}
		mn[3*i+0] = p.x - sx; mn[3*i+1] = p.y - sy; mn[3*i+2] = p.z - sz;
		mx[3*i+0] = p.x + sx; mx[3*i+1] = p.y + sy; mx[3*i+2] = p.z + sz;
	}
	// morton keys
	nh_f3 smin = nh_make3(mn[0], mn[1], mn[2]), smax = smin;
	for (uint32_t i = 1; i < n; ++i) {
		smin.x = nh_min(smin.x, mn[3*i]); smin.y = nh_min(smin.y, mn[3*i+1]); smin.z = nh_min(smin.z, mn[3*i+2]);
		smax.x = nh_max(smax.x, mn[3*i]); smax.y = nh_max(smax.y, mn[3*i+1]); smax.z = nh_max(smax.z, mn[3*i+2]);
	}
	float scale = nh_morton_scale(smin, smax);
	nh_f3 smin_s = nh_make3(smin.x * scale, smin.y * scale, smin.z * scale);
	std::vector<uint64_t> mk(n);
	for (uint32_t i = 0; i < n; ++i) mk[i] = nh_morton_of(nh_make3(mn[3*i], mn[3*i+1], mn[3*i+2]), scale, smin_s);

	struct Rec { uint64_t key; uint32_t feature; uint32_t seq; float c[8]; uint32_t ba, bb; };
	std::vector<Rec> recs;
	uint32_t seq = 0;
	for (uint32_t i = 0; i < n; ++i) for (uint32_t j = i + 1; j < n; ++j) {
		bool ov = mx[3*j] > mn[3*i] && mx[3*i] > mn[3*j] && mx[3*j+1] > mn[3*i+1] && mx[3*i+1] > mn[3*j+1] && mx[3*j+2] > mn[3*i+2] && mx[3*i+2] > mn[3*j+2];
		if (!ov) continue;
		if (xf[i].body == xf[j].body) continue;
		// first = earlier in morton order (ties: lower index)
		uint32_t first = (mk[i] < mk[j] || (mk[i] == mk[j] && i < j)) ? i : j;
		uint32_t second = first == i ? j : i;
		nh_contact_out out[16];
		int cnt = 0; uint64_t key = 0; uint32_t ba = 0, bb = 0;
		bool fs = first >= nbox, ss = second >= nbox;
		if (!fs && !ss) {
			// narrowphase a = later in morton order
			uint32_t a = second, b = first;
			nh_pair_result r = nh_box_box(xf[a], xf[b], box_size + 4*a, box_size + 4*b, tags[a], tags[b], out);
			cnt = r.count;
			uint32_t ca = r.swapped ? b : a, cb = r.swapped ? a : b;
			key = (uint64_t)tags[ca] | ((uint64_t)tags[cb] << 32);
			ba = xf[ca].body; bb = xf[cb].body;
		} else if (fs && ss) {
			uint32_t a = first, b = second;   // sphere-sphere: a = earlier (pair >> 16)
			cnt = nh_sphere_sphere(sph_radius[a - nbox], sph_radius[b - nbox], xf[a], xf[b], out);
			key = (uint64_t)tags[a] | ((uint64_t)tags[b] << 32);
			ba = xf[a].body; bb = xf[b].body;
		} else {
			uint32_t a = fs ? second : first, b = fs ? first : second;   // a = box, b = sphere
			cnt = nh_box_sphere(box_size + 4*a, sph_radius[b - nbox], xf[a], xf[b], out);
			key = (uint64_t)tags[a] | ((uint64_t)tags[b] << 32);
			ba = xf[a].body; bb = xf[b].body;
		}
		for (int k = 0; k < cnt; ++k) {
			Rec rc; rc.key = key; rc.feature = out[k].feature; rc.seq = seq++;
			rc.c[0] = out[k].px; rc.c[1] = out[k].py; rc.c[2] = out[k].pz; rc.c[3] = out[k].penetration;
			rc.c[4] = out[k].nx; rc.c[5] = out[k].ny; rc.c[6] = out[k].nz; rc.c[7] = out[k].friction;
			rc.ba = ba; rc.bb = bb;
			recs.push_back(rc);
		}
	}
	// reference tag order: (b_tag, a_tag, feature)
	std::stable_sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) {
		uint64_t kx = (x.key >> 32) << 32 | (x.key & 0xffffffffu), ky = (y.key >> 32) << 32 | (y.key & 0xffffffffu);
		if (kx != ky) return kx < ky;
		return x.feature < y.feature;
	});
	uint32_t m = (uint32_t)std::min<size_t>(recs.size(), cap);
	for (uint32_t i = 0; i < m; ++i) {
		memcpy(out_contacts + 8*i, recs[i].c, 32);
		out_bodies[2*i] = recs[i].ba; out_bodies[2*i+1] = recs[i].bb;
		out_keys[i] = recs[i].key; out_features[i] = recs[i].feature;
	}
	return (int)recs.size();
}

// setup + warm start + `iterations` x apply + export, with contacts visited in the given order
// (order[i] = contact index; duplicates -- the reference's padding lanes -- are skipped).
void hs_solve(uint32_t nbodies, const Xf* body_xf, const float* props /*4 per: inertia_inverse xyz, mass_inverse*/, float* momentum /*8 per*/,
              uint32_t ncontacts, const float* contacts /*8 per*/, const uint32_t* bodies /*2 per*/, const float* cached /*4 per*/,
              uint32_t norder, const uint32_t* order, uint32_t iterations, float* out_impulses /*4 per*/) {
	std::vector<nh_row> rows(ncontacts);
	std::vector<nh_state> st(ncontacts);
	std::vector<uint32_t> seq;
	std::vector<uint8_t> seen(ncontacts, 0);
	for (uint32_t i = 0; i < norder; ++i) { uint32_t c = order[i]; if (!seen[c]) { seen[c] = 1; seq.push_back(c); } }
	for (uint32_t i = 0; i < nbodies; ++i) momentum[8*i+3] = props[4*i+3];
	auto load = [&](uint32_t b) { nh_vel v; memcpy(&v, momentum + 8*b, 32); return v; };
	auto store = [&](uint32_t b, const nh_vel& v) { memcpy(momentum + 8*b, &v, 32); };
	for (uint32_t c : seq) {
		uint32_t a = bodies[2*c], b = bodies[2*c+1];
		nh_quat qa = { body_xf[a].q[0], body_xf[a].q[1], body_xf[a].q[2], body_xf[a].q[3] };
		nh_quat qb = { body_xf[b].q[0], body_xf[b].q[1], body_xf[b].q[2], body_xf[b].q[3] };
		nh_inertia IA = nh_world_inertia(qa, props[4*a], props[4*a+1], props[4*a+2]);
		nh_inertia IB = nh_world_inertia(qb, props[4*b], props[4*b+1], props[4*b+2]);
		nh_contact_in ci; memcpy(&ci, contacts + 8*c, 32);
		nh_build_row(ci, nh_make3(body_xf[a].p[0], body_xf[a].p[1], body_xf[a].p[2]), nh_make3(body_xf[b].p[0], body_xf[b].p[1], body_xf[b].p[2]),
		             IA, IB, props[4*a+3], props[4*b+3], rows[c]);
		nh_vel va = load(a), vb = load(b);
		nh_warm_start(rows[c], cached[4*c], cached[4*c+1], cached[4*c+2], va, vb, st[c]);
		store(a, va); store(b, vb);
	}
	for (uint32_t it = 0; it < iterations; ++it)
		for (uint32_t c : seq) {
			uint32_t a = bodies[2*c], b = bodies[2*c+1];
			nh_vel va = load(a), vb = load(b);
			nh_apply_one(rows[c], st[c], va, vb);
			store(a, va); store(b, vb);
		}
	for (uint32_t c = 0; c < ncontacts; ++c) {
		if (!seen[c]) continue;
		nh_export_impulse(rows[c], st[c], out_impulses[4*c], out_impulses[4*c+1], out_impulses[4*c+2]);
	}
}

// Same solve, but contacts against body 0 go through the ONE-BODY forms (nh_row1 / nh_apply_one1 ...) the fused GPU kernel uses
// when body 0 is inert.  Returns the number of contacts that took that path (0 if body 0 is not inert).
int hs_solve_one_body(uint32_t nbodies, const Xf* body_xf, const float* props, float* momentum,
                      uint32_t ncontacts, const float* contacts, const uint32_t* bodies, const float* cached,
                      uint32_t norder, const uint32_t* order, uint32_t iterations, float* out_impulses) {
	if (!nh_is_inert(momentum, momentum + 4, props, props[3])) return 0;
	std::vector<nh_row> rows(ncontacts);
	std::vector<nh_row1> rows1(ncontacts);
	std::vector<nh_state> st(ncontacts);
	std::vector<uint32_t> seq;
	std::vector<uint8_t> seen(ncontacts, 0);
	for (uint32_t i = 0; i < norder; ++i) { uint32_t c = order[i]; if (!seen[c]) { seen[c] = 1; seq.push_back(c); } }
	for (uint32_t i = 0; i < nbodies; ++i) momentum[8*i+3] = props[4*i+3];
	auto load = [&](uint32_t b) { nh_vel v; memcpy(&v, momentum + 8*b, 32); return v; };
	auto store = [&](uint32_t b, const nh_vel& v) { memcpy(momentum + 8*b, &v, 32); };
	auto one = [&](uint32_t c) { return (bodies[2*c] == 0) != (bodies[2*c+1] == 0); };
	int taken = 0;
	for (uint32_t c : seq) {
		uint32_t a = bodies[2*c], b = bodies[2*c+1];
		nh_contact_in ci; memcpy(&ci, contacts + 8*c, 32);
		if (one(c)) {
			uint32_t d = a ? a : b;
			nh_quat qd = { body_xf[d].q[0], body_xf[d].q[1], body_xf[d].q[2], body_xf[d].q[3] };
			nh_inertia ID = nh_world_inertia(qd, props[4*d], props[4*d+1], props[4*d+2]);
			nh_build_row1(ci, nh_make3(body_xf[d].p[0], body_xf[d].p[1], body_xf[d].p[2]), ID, props[4*d+3], a != 0, rows1[c]);
			nh_vel vd = load(d);
			nh_warm_start1(rows1[c], cached[4*c], cached[4*c+1], cached[4*c+2], vd, a != 0, st[c]);
			store(d, vd);
			++taken;
			continue;
		}
		nh_quat qa = { body_xf[a].q[0], body_xf[a].q[1], body_xf[a].q[2], body_xf[a].q[3] };
		nh_quat qb = { body_xf[b].q[0], body_xf[b].q[1], body_xf[b].q[2], body_xf[b].q[3] };
		nh_inertia IA = nh_world_inertia(qa, props[4*a], props[4*a+1], props[4*a+2]);
		nh_inertia IB = nh_world_inertia(qb, props[4*b], props[4*b+1], props[4*b+2]);
		nh_build_row(ci, nh_make3(body_xf[a].p[0], body_xf[a].p[1], body_xf[a].p[2]), nh_make3(body_xf[b].p[0], body_xf[b].p[1], body_xf[b].p[2]),
		             IA, IB, props[4*a+3], props[4*b+3], rows[c]);
		nh_vel va = load(a), vb = load(b);
		nh_warm_start(rows[c], cached[4*c], cached[4*c+1], cached[4*c+2], va, vb, st[c]);
		store(a, va); store(b, vb);
	}
	for (uint32_t it = 0; it < iterations; ++it)
		for (uint32_t c : seq) {
			uint32_t a = bodies[2*c], b = bodies[2*c+1];
			if (one(c)) {
				uint32_t d = a ? a : b;
				nh_vel vd = load(d);
				nh_apply_one1(rows1[c], st[c], vd, a != 0);
				store(d, vd);
				continue;
			}
			nh_vel va = load(a), vb = load(b);
			nh_apply_one(rows[c], st[c], va, vb);
			store(a, va); store(b, vb);
		}
	for (uint32_t c = 0; c < ncontacts; ++c) {
		if (!seen[c]) continue;
		if (one(c)) nh_export_impulse1(rows1[c], st[c], out_impulses[4*c], out_impulses[4*c+1], out_impulses[4*c+2]);
		else nh_export_impulse(rows[c], st[c], out_impulses[4*c], out_impulses[4*c+1], out_impulses[4*c+2]);
	}
	// the general path zeroes unused1 of every body it touches, body 0 included (nudge.cpp:4823, 4849)
	return taken;
}

// Same solve with RADIAL rows (nh_rowr, nh_solver.h) wherever a contact qualifies (both lever arms parallel to the normal, isotropic
// inertia, or the inert static world): what the colour-scheduled GPU sweeps of the default solver order run.  Returns how many did.
int hs_solve_radial(uint32_t nbodies, const Xf* body_xf, const float* props, float* momentum,
                    uint32_t ncontacts, const float* contacts, const uint32_t* bodies, const float* cached,
                    uint32_t norder, const uint32_t* order, uint32_t iterations, float* out_impulses) {
	const bool inert0 = nh_is_inert(momentum, momentum + 4, props, props[3]);
	std::vector<nh_row> rows(ncontacts);
	std::vector<nh_rowr> rowsr(ncontacts);
	std::vector<uint8_t> radial(ncontacts, 0);
	std::vector<nh_state> st(ncontacts);
	std::vector<uint32_t> seq;
	std::vector<uint8_t> seen(ncontacts, 0);
	for (uint32_t i = 0; i < norder; ++i) { uint32_t c = order[i]; if (!seen[c]) { seen[c] = 1; seq.push_back(c); } }
	for (uint32_t i = 0; i < nbodies; ++i) momentum[8*i+3] = props[4*i+3];
	auto load = [&](uint32_t b) { nh_vel v; memcpy(&v, momentum + 8*b, 32); return v; };
	auto store = [&](uint32_t b, const nh_vel& v) { memcpy(momentum + 8*b, &v, 32); };
	int taken = 0;
	for (uint32_t c : seq) {
		uint32_t a = bodies[2*c], b = bodies[2*c+1];
		nh_contact_in ci; memcpy(&ci, contacts + 8*c, 32);
		nh_f3 pa = nh_make3(body_xf[a].p[0], body_xf[a].p[1], body_xf[a].p[2]), pb = nh_make3(body_xf[b].p[0], body_xf[b].p[1], body_xf[b].p[2]);
		const bool ia = inert0 && a == 0, ib = inert0 && b == 0;
		radial[c] = nh_side_is_radial(ci.px - pa.x, ci.py - pa.y, ci.pz - pa.z, ci.nx, ci.ny, ci.nz, props[4*a], props[4*a+1], props[4*a+2], ia) &&
		            nh_side_is_radial(ci.px - pb.x, ci.py - pb.y, ci.pz - pb.z, ci.nx, ci.ny, ci.nz, props[4*b], props[4*b+1], props[4*b+2], ib);
		nh_vel va = load(a), vb = load(b);
		if (radial[c]) {
			nh_build_rowr(ci, pa, pb, props[4*a], props[4*b], props[4*a+3], props[4*b+3], ia, ib, rowsr[c]);
			nh_warm_start_r(rowsr[c], cached[4*c], cached[4*c+1], cached[4*c+2], va, vb, st[c]);
			++taken;
		} else {
			nh_quat qa = { body_xf[a].q[0], body_xf[a].q[1], body_xf[a].q[2], body_xf[a].q[3] };
			nh_quat qb = { body_xf[b].q[0], body_xf[b].q[1], body_xf[b].q[2], body_xf[b].q[3] };
			nh_build_row(ci, pa, pb, nh_world_inertia(qa, props[4*a], props[4*a+1], props[4*a+2]), nh_world_inertia(qb, props[4*b], props[4*b+1], props[4*b+2]),
			             props[4*a+3], props[4*b+3], rows[c]);
			nh_warm_start(rows[c], cached[4*c], cached[4*c+1], cached[4*c+2], va, vb, st[c]);
		}
		store(a, va); store(b, vb);
	}
	for (uint32_t it = 0; it < iterations; ++it)
		for (uint32_t c : seq) {
			uint32_t a = bodies[2*c], b = bodies[2*c+1];
			nh_vel va = load(a), vb = load(b);
			if (radial[c]) nh_apply_radial(rowsr[c], st[c], va, vb); else nh_apply_one(rows[c], st[c], va, vb);
			store(a, va); store(b, vb);
		}
	for (uint32_t c = 0; c < ncontacts; ++c) {
		if (!seen[c]) continue;
		if (radial[c]) nh_export_impulse_r(rowsr[c], st[c], out_impulses[4*c], out_impulses[4*c+1], out_impulses[4*c+2]);
		else nh_export_impulse(rows[c], st[c], out_impulses[4*c], out_impulses[4*c+1], out_impulses[4*c+2]);
	}
	return taken;
}

void hs_advance(uint32_t nactive, const uint32_t* active, Xf* body_xf, const float* momentum, uint8_t* idle, float dt) {
	for (uint32_t n = 0; n < nactive; ++n) {
		uint32_t i = active[n];
		idle[i] = nh_advance_body(body_xf[i].p, body_xf[i].q, momentum + 8*i, momentum + 8*i + 4, idle[i], dt);
	}
}

void hs_gravity(uint32_t nactive, const uint32_t* active, float* momentum, float dt, float g, float rate) {
	float damping = 1.0f - dt * rate;
	float dv = g * dt;
	for (uint32_t n = 0; n < nactive; ++n) {
		uint32_t i = active[n];
		nh_gravity_damping(momentum + 8*i, momentum + 8*i + 4, 0.0f * dt, dv, 0.0f * dt, damping);
	}
}

}
