// nh_fused.hip -- the ONE-KERNEL still step (round 6; include/nudge_hip.h note 9, nh_internal.h: nh_FusedState).
//
// A plain still step inside an nh_step call (nobody moving, nobody able to sleep, the sub-step before it did its k_xform<true>: "xform ahead") used to be two
// launches: k_narrowphase<*, true, false> over the kept pairs -- up to four 36-byte contacts per pair written to the pair's raw slots -- and the fused one-body
// solver, which read them back.  In the worlds this form runs in every dynamic body owns exactly one kept pair (its contact-list record), so the lane that solves a
// body can evaluate that pair itself (nudge.cpp:1177-2604, nh_narrowphase.h), keep the <= 4 contacts in LDS, and go on to rows, warm start, sweeps, export, gravity,
// advance and the next sub-step's transforms (nudge.cpp:4350-4925, nh_solver.h) without the contacts ever reaching HBM: 144 B per pair not written, 144 B not read,
// one launch boundary less.
//
// Two things the launch boundary between narrowphase and solver used to provide have to be provided otherwise:
//
// 1. THE DENSE INDEX.  The order in which a body's contacts are solved depends on their indices i in the dense, tag-ordered contact list of the step (nudge.cpp:4206-4339:
//    batch(i) = (i >> 7) * 16 + (i & 15)), i.e. on how many contacts the records BEFORE the body's record in tag order have in this step.  Lanes therefore run in
//    TAG ORDER -- wave w takes positions 64 w .. 64 w + 63 of the kept tag order (the table k_fused_map writes once per layout) -- and the number of contacts gained or
//    lost before a wave travels by a decoupled look-back (every wave publishes the change of its 64 records, then the running total, in one 64-bit word tagged with the
//    step's sequence number: no clearing between steps; a wave only ever waits for waves with smaller indices, which the dispatcher has started before it).
//
// 2. ALL CHECKS BEFORE ANY WRITE.  A still step that meets something it does not own (a key that changed, a fifth contact, a contact between two dynamic bodies ...)
//    raises st->still_failed_seq and must leave the world as it was, to be run again in full.  In one launch a wave may find that out after others have advanced
//    their bodies, so the kernel is OUT OF PLACE: body state (transform, momentum, idle counter), the slot cache and the colliders' world transforms / AABBs are read
//    from one set of buffers and written to the other (nh_FusedState: `cur`); the host swaps the roles when it launches and swaps them back for every launch whose
//    verdict turns out to be "failed".  The caller's body arrays are buffer 0, so fused steps are launched in pairs inside an nh_step call and the sub-step a call ends
//    with is never fused (it also leaves the contacts in their slots, which the views of note 9 are exported from).
//
// Everything arithmetic is the device functions the two kernels use (nh_box_box_eval, nh_build_row1, nh_apply_one1, ...), in their order: the results are the bits
// of the two-kernel form (tests/test_gpu_still.py::test_fused_*), of a library that never speculates, and of the reference.
#include "nh_internal.h"
#include "nh_narrowphase.h"
#include <type_traits>
#include "nh_solver.h"

#define NH_FZ_OWN 0x80000000u          // fz.x: the record is its dynamic body's contact-list record (the lane solves that body)
#define NH_FZ_IS_A 0x80000000u         // fz.y: ... and the body plays "a" in its contacts (NH_BODY_REC_IS_A)
#define NH_FZ_DEAD 0x80000000u         // fz.z: the kept pair is dead (generation stamps)
#define NH_FZ_DYN_IS_B 0x80000000u     // fz.w: the dynamic body's collider is the pair's second one
#define NH_FZ_AGG 1u

__device__ __forceinline__ uint64_t fz_pack(uint32_t seq, uint32_t flag, int value) { return ((uint64_t)seq << 32) | ((uint64_t)flag << 30) | (uint64_t)((uint32_t)(value + (1 << 29)) & 0x3FFFFFFFu); }
__device__ __forceinline__ int fz_value(uint64_t s) { return (int)((uint32_t)s & 0x3FFFFFFFu) - (1 << 29); }
__device__ __forceinline__ uint32_t fz_flag(uint64_t s, uint32_t seq) { return (uint32_t)(s >> 32) == seq ? (((uint32_t)s >> 30) & 3u) : 0u; }

__device__ __forceinline__ uint32_t fz_slot_key_of(uint32_t i) { return (i >> 7) * 16u + (i & 15u); }          // (nh_solve.hip: slot_key_of, closed form)

// ---- the position table: once per layout (a full step, or a still step in movers form, voids it) ----------------------------------------------------------------
// fz[p] for tag-order position p: the record (= kept pair) there, its dynamic body and whether the record is that body's own, the pair's two colliders.
__global__ __launch_bounds__(256) void k_fused_map(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ body_rec,
                                                   const uint2* __restrict__ kept, uint32_t kept_cap, uint32_t pair_cap, const uint8_t* __restrict__ gen,
                                                   const float4* __restrict__ aabb_min, uint4* __restrict__ fz) {
	const uint32_t nrec = st->records;
	const uint32_t n_bb = min(st->pairs, pair_cap);
	const uint32_t index_mask = gen ? NH_GEN_INDEX : 0xFFFFFFFFu;
	const bool stamps = gen && st->moved_count != 0u;
	for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < nrec; p += gridDim.x * blockDim.x) {
		const uint32_t r = sorted_idx[p];
		const uint2 kp = r < n_bb ? kept[r] : kept[kept_cap - 1u - (r - n_bb)];
		const uint32_t ca = kp.x & index_mask, cb = kp.y & index_mask;
		bool dead = false;
		if (stamps) dead = !((uint32_t)(gen[ca] & 0x7Fu) == (kp.x >> NH_GEN_SHIFT) && (uint32_t)(gen[cb] & 0x7Fu) == (kp.y >> NH_GEN_SHIFT));
		const uint32_t ba = __float_as_uint(aabb_min[ca].w), bb = __float_as_uint(aabb_min[cb].w);
		uint32_t dyn = 0u, own = 0u, is_a = 0u;
		if ((ba == 0u) != (bb == 0u)) {
			dyn = ba ? ba : bb;
			const uint32_t br = body_rec[dyn];
			if (br != NH_BODY_REC_NONE && (br & ~NH_BODY_REC_IS_A) == r) { own = NH_FZ_OWN; is_a = (br & NH_BODY_REC_IS_A) ? NH_FZ_IS_A : 0u; }
		}
		fz[p] = make_uint4(r | own, dyn | is_a, ca | (dead ? NH_FZ_DEAD : 0u), cb | ((dyn && bb == dyn) ? NH_FZ_DYN_IS_B : 0u));
	}
}

// the static world's colliders (and every tag) into the other set of collider buffers: a fused step reads one set and writes the other, and only the colliders of
// dynamic bodies are written by its lanes.  Once per k_xform<true>, i.e. once per nh_step call that fuses.
__global__ __launch_bounds__(256) void k_fused_copy_static(uint32_t n, const float4* __restrict__ xf_in, const float4* __restrict__ min_in, const float4* __restrict__ max_in,
                                                           float4* __restrict__ xf_out, float4* __restrict__ min_out, float4* __restrict__ max_out) {
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const float4 mn = min_in[c];
		if (__float_as_uint(mn.w) != 0u) continue;
		min_out[c] = mn; max_out[c] = max_in[c];
		xf_out[2u * (size_t)c] = xf_in[2u * (size_t)c]; xf_out[2u * (size_t)c + 1u] = xf_in[2u * (size_t)c + 1u];
	}
}

// body arrays of the shadow set -> the caller's (a run of fused steps that could not be ended in pairs: rare)
__global__ __launch_bounds__(256) void k_fused_copy_bodies(uint32_t n, const float4* __restrict__ xf_in, const float4* __restrict__ mom_in, const uint8_t* __restrict__ idle_in,
                                                           float4* __restrict__ xf_out, float4* __restrict__ mom_out, uint8_t* __restrict__ idle_out) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		xf_out[2u * (size_t)i] = xf_in[2u * (size_t)i]; xf_out[2u * (size_t)i + 1u] = xf_in[2u * (size_t)i + 1u];
		mom_out[2u * (size_t)i] = mom_in[2u * (size_t)i]; mom_out[2u * (size_t)i + 1u] = mom_in[2u * (size_t)i + 1u];
		idle_out[i] = idle_in[i];
	}
}

struct nh_FusedView {
	const uint4* fz; uint32_t nrec, pos_waves; unsigned long long* status;
	// per 64 position waves (segments) and per 4096 (super-segments, behind them in the same array): waves heard from | sum of their changes << 32 -- this launch's, and the
	// next launch's words, which this launch clears
	unsigned long long* seg; unsigned long long* sup; unsigned long long* seg_next; uint32_t seg_words;
	// the pair's evaluation (k_narrowphase<*, true, false>): colliders of the set the LAST step wrote
	const nh_xform* xf; const float4* aabb_min; const float4* aabb_max; const uint32_t* ctag;
	const nh_BoxCollider* box_data; const nh_SphereCollider* sph_data; uint32_t nbox, pair_cap;
	const uint64_t* rec_key; uint32_t* cnt_sorted; uint32_t* start_sorted;
	// the slot cache, in and out
	const float4* sc_imp_in; const uint32_t* sc_feat_in; const uint32_t* sc_count_in;
	float4* sc_imp_out; uint32_t* sc_feat_out; uint32_t* sc_count_out;
	// body state, in and out
	const nh_Transform* xf_in; const nh_BodyMomentum* mom_in; const uint8_t* idle_in;
	nh_Transform* xf_out; nh_BodyMomentum* mom_out; uint8_t* idle_out;
	const nh_BodyProperties* props; uint32_t nbodies;
	const uint32_t* body_rec; const uint32_t* body_col;
	uint32_t parity, seq, guard_seq, iterations;
	uint32_t debug_fail;          // test hook (option "fused_fail_in"): this launch raises the failure flag at the very END of one of its middle waves -- when others have long written
	float gx_dt, gy_dt, gz_dt, damping, time_step;
};

// One wave per workgroup, two waves per SIMD (the box-box SAT and the four-contact solver each need ~200 VGPRs; their phases follow each other).
// Workgroups [0, pos_waves): tag-order positions; [pos_waves, ...): dynamic bodies WITHOUT a record (they only fall), by body index.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_still_fused(nh_DevState* __restrict__ st, nh_FusedView fv, nh_AheadView av) {
	constexpr int M = 4;
	const uint32_t lane = nh_lane();
	const uint32_t parity = fv.parity, seq = fv.seq;
	// (the segment sums of the step after this one: nobody looks at them during this launch -- cleared whether or not this step goes through, the host counts on it)
	if (blockIdx.x == 0) for (uint32_t k = lane; k < fv.seg_words; k += 64u) fv.seg_next[k] = 0ull;
	// ---- round 1: who am I (asked for before the verdict below is waited for: one memory round trip for both) ----
	const bool pos_wave = blockIdx.x < fv.pos_waves;
	uint4 z = make_uint4(0u, 0u, 0u, 0u);
	uint32_t free_rec = 0u, free_col = 0xFFFFFFFFu;
	{
		const uint32_t p0 = blockIdx.x * 64u + lane;
		const uint32_t x0 = min((pos_wave ? 0u : blockIdx.x - fv.pos_waves) * 64u + lane, fv.nbodies - 1u);
		if (pos_wave) z = fv.fz[p0 < fv.nrec ? p0 : 0u];
		else { free_rec = fv.body_rec[x0]; free_col = fv.body_col[x0]; }
	}
	// ---- what every workgroup finds out for itself (the same words, the same verdict) ----
	{
		const nh_BodyMomentum m0 = fv.mom_in[0];
		const nh_BodyProperties p0 = fv.props[0];
		const bool inert = nh_is_inert(m0.velocity, m0.angular_velocity, p0.inertia_inverse, p0.mass_inverse);
		// (what k_narrowphase<.., true, false>(ahead) checks at its start: the last step's lanes met a leaver or a sleeper, a world the lanes cannot stand in for k_xform
		// in, a layout that is not the last full step's, a kept list that is not the layout's, another record count than the table was built for)
		const bool bad = !inert || st->ahead_failed_seq >= seq || st->ahead_multi != 0u || !st->lay_valid || st->fat_count != st->pairs || st->fat_count_sph != st->pairs_sph || st->records != fv.nrec;
		if (bad && blockIdx.x == 0 && lane == 0) { atomicMax(&st->still_failed_seq, seq); atomicAdd(&st->still_diff[3], 1u); }
		if (bad || st->still_failed_seq >= fv.guard_seq) {
			// (a wave that leaves because ANOTHER wave of this launch has raised the flag still says so in its look-back word: nobody waits for it)
			if (blockIdx.x < fv.pos_waves && lane == 0) {
				__hip_atomic_store(&fv.status[blockIdx.x], fz_pack(seq, NH_FZ_AGG, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const unsigned long long old = atomicAdd(&fv.seg[blockIdx.x >> 6], 1ull);
				if ((uint32_t)old == 63u) atomicAdd(&fv.sup[blockIdx.x >> 12], ((old >> 32) << 32) + 64ull);
			}
			return;
		}
	}
	if (blockIdx.x == 0) {
		// what k_xform<true> does for the step after this one (nobody touches the other parity's words during this step)
		if (lane == 0) {
			st->max_idle[parity ^ 1u] = 0u; st->delta_count[parity ^ 1u] = 0u; st->delta_overflow[parity ^ 1u] = 0u; st->still_esc[parity ^ 1u] = 0u;
			st->still_asleep[parity ^ 1u] = 0u; st->still_sleeping[parity ^ 1u] = 0u; st->still_culled[parity ^ 1u] = 0u;
			for (int k = 0; k < 3; ++k) { st->still_smin[parity ^ 1u][k] = 0xffffffffu; st->still_smax[parity ^ 1u][k] = 0u; }
			st->still_fat0 = st->fat_count + st->fat_count_sph;
			st->active = fv.nbodies - 1u; st->sleeping = 0u; st->culled = 0u;          // (plain form: nobody asleep -- the solver's counters)
		}
		for (uint32_t k = lane; k < NH_AHEAD_PARTS * 8u; k += 64u) (&st->ahead_part[parity ^ 1u][0][0])[k] = (k & 7u) < 3u ? 0xffffffffu : 0u;
	}
	__shared__ float4 s_imp[64 * M];                  // cached impulses in (wave-cooperative, slot order); solved impulses out
	__shared__ float4 s_c0[64 * M];                   // this step's contacts: position + penetration, by (lane, slot) -- later the lane's collider for the next step's transforms
	__shared__ uint32_t s_f[64 * M];                  // ... and their feature words
	__shared__ float4 s_body[5][64];                  // momentum (2), properties, transform (2): fetched straight to LDS while the pair is evaluated
	__shared__ uint32_t s_park[9][64];                // key on file (2), count and start on file, cached count, cached feature words (4): parked across the contact computation
	__shared__ uint32_t s_first[64], s_dst[64 * M];
	const uint32_t n_bb = min(st->pairs, fv.pair_cap);
	uint32_t r = 0, body = 0, ca = 0, cb = 0, base = 0, p = 0;
	bool live = false, own = false, is_a_body = false, dead = false, dyn_is_b = false, free_body = false;
	uint32_t col = 0xFFFFFFFFu;
	if (pos_wave) {
		p = blockIdx.x * 64u + lane;
		live = p < fv.nrec;
		if (live) {
			r = z.x & ~NH_FZ_OWN; own = (z.x & NH_FZ_OWN) != 0u;
			body = z.y & ~NH_FZ_IS_A; is_a_body = (z.y & NH_FZ_IS_A) != 0u;
			ca = z.z & ~NH_FZ_DEAD; dead = (z.z & NH_FZ_DEAD) != 0u;
			cb = z.w & ~NH_FZ_DYN_IS_B; dyn_is_b = (z.w & NH_FZ_DYN_IS_B) != 0u;
			base = r < n_bb ? 4u * r : 4u * n_bb + (r - n_bb);
			if (own) col = dyn_is_b ? cb : ca;
		}
	} else {
		const uint32_t x = (blockIdx.x - fv.pos_waves) * 64u + lane;
		free_body = x < fv.nbodies && (x == 0u || free_rec == NH_BODY_REC_NONE);
		body = free_body ? x : 0u;
		if (free_body && x != 0u) col = free_col;
		if (__builtin_amdgcn_ballot_w64(free_body) == 0ull) return;
	}
	const bool solves = own || (free_body && body != 0u);          // this lane owns a dynamic body: gravity, contacts if any, advance, the next step's transforms
	const bool sph_rec = live && r >= n_bb;
	s_first[lane] = own ? base : 0xFFFFFFFFu;
	__builtin_amdgcn_wave_barrier();
	// ---- round 2: everything else at once.  What the solver phase needs goes straight to LDS (no register holds it through the SAT) ----
	{
		const uint32_t xb = solves ? body : 0u;
		const float4* mp = reinterpret_cast<const float4*>(fv.mom_in + xb);
		const float4* tp = reinterpret_cast<const float4*>(fv.xf_in + xb);
		nh_glds16(mp, &s_body[0][lane]); nh_glds16(mp + 1, &s_body[1][lane]);
		nh_glds16(reinterpret_cast<const float4*>(fv.props + xb), &s_body[2][lane]);
		nh_glds16(tp, &s_body[3][lane]); nh_glds16(tp + 1, &s_body[4][lane]);
		if (pos_wave) {
#pragma unroll
			for (int k = 0; k < M; ++k) {
				const uint32_t e = 64u * k + lane, j = e / M;
				const uint32_t bj = s_first[j];
				// (a record of a pair with a sphere owns ONE slot: its entries 1 .. 3 fetch slot 0 again and are never looked at)
				const uint32_t c = bj != 0xFFFFFFFFu ? (bj < 4u * n_bb ? bj + (e - j * M) : bj) : 0u;
				nh_glds16(fv.sc_imp_in + c, &s_imp[e]);
			}
		}
	}
	uint32_t idle0 = 0;
	if (solves) idle0 = fv.idle_in[body];
	int count = 0;
	uint64_t key = 0;
	uint32_t body_a = 0, body_b = 0;
	bool invalid = false, changed = false;
	float nx = 0.0f, ny = 0.0f, nz = 0.0f, fric = 0.5f;          // the contacts of one pair share normal and friction (face clipping: res.wn; an edge or a sphere pair has one contact)
	// the Morton frame of this step (nudge.cpp:3096-3100) from what the last step's lanes gathered in NH_AHEAD_PARTS places + the static world's share
	float mscale = 0.0f;
	nh_f3 mmin = nh_make3(0.0f, 0.0f, 0.0f);
	uint32_t start = 0, cnt = 0, oc = 0;
	uint32_t of[M] = { 0u, 0u, 0u, 0u };
	if (pos_wave) {
		uint32_t fr_min[3], fr_max[3], fr_top;
		{
			const uint32_t* part = &st->ahead_part[parity][lane & (NH_AHEAD_PARTS - 1u)][0];
			fr_top = part[6];
			for (int k = 0; k < 3; ++k) { fr_min[k] = min(part[k], st->still_static_min[k]); fr_max[k] = max(part[3 + k], st->still_static_max[k]); }
		}
		// the pair: both boxes, both world transforms, shapes and tags, the record's key / count / start on file, the slot cache's count and feature words
		const float4 amin = fv.aabb_min[ca], amax = fv.aabb_max[ca], bmin = fv.aabb_min[cb], bmax = fv.aabb_max[cb];
		const nh_xform XA = fv.xf[ca], XB = fv.xf[cb];
		const uint32_t TA = fv.ctag[ca], TB = fv.ctag[cb];
		const nh_BoxCollider SA = fv.box_data[ca < fv.nbox ? ca : 0u], SB = fv.box_data[cb < fv.nbox ? cb : 0u];
		float RA = 0.0f, RB = 0.0f;
		if (__builtin_amdgcn_ballot_w64(sph_rec) != 0ull) { RA = fv.sph_data[ca >= fv.nbox ? ca - fv.nbox : 0u].radius; RB = fv.sph_data[cb >= fv.nbox ? cb - fv.nbox : 0u].radius; }
		{
			const uint64_t pk = fv.rec_key[r];
			s_park[0][lane] = (uint32_t)pk; s_park[1][lane] = (uint32_t)(pk >> 32);
			s_park[2][lane] = fv.cnt_sorted[p < fv.nrec ? p : 0u]; s_park[3][lane] = fv.start_sorted[p < fv.nrec ? p : 0u];
			s_park[4][lane] = fv.sc_count_in[r];
			if (sph_rec) { s_park[5][lane] = fv.sc_feat_in[base]; s_park[6][lane] = 0u; s_park[7][lane] = 0u; s_park[8][lane] = 0u; }
			else { const uint4 of4 = *reinterpret_cast<const uint4*>(fv.sc_feat_in + base); s_park[5][lane] = of4.x; s_park[6][lane] = of4.y; s_park[7][lane] = of4.z; s_park[8][lane] = of4.w; }
		}
		for (int k = 0; k < 3; ++k)
			for (int d = 32; d >= 1; d >>= 1) { fr_min[k] = min(fr_min[k], (uint32_t)__shfl_xor((int)fr_min[k], d)); fr_max[k] = max(fr_max[k], (uint32_t)__shfl_xor((int)fr_max[k], d)); }
		for (int d = 32; d >= 1; d >>= 1) fr_top = max(fr_top, (uint32_t)__shfl_xor((int)fr_top, d));
		if (blockIdx.x == 0 && lane == 0) {
			// (filed where a step with a k_xform<true> of its own would have it: the host's mirror, the sleep prediction)
			for (int k = 0; k < 3; ++k) { st->still_smin[parity][k] = fr_min[k]; st->still_smax[parity][k] = fr_max[k]; }
			st->max_idle[parity] = fr_top;
		}
		{
			const nh_f3 smin = nh_make3(nh_float_unflip(fr_min[0]), nh_float_unflip(fr_min[1]), nh_float_unflip(fr_min[2]));
			const nh_f3 smax = nh_make3(nh_float_unflip(fr_max[0]), nh_float_unflip(fr_max[1]), nh_float_unflip(fr_max[2]));
			mscale = nh_morton_scale(smin, smax);
			mmin = nh_make3(smin.x * mscale, smin.y * mscale, smin.z * mscale);
		}
		// ---- the pair (k_narrowphase<*, true, false>, one record per lane): SAT / clipping first -- the contact COUNT is what the other waves wait for ----
		nh_bb_result bb;
		bb.kind = 0; bb.mask = 0;
		nh_contact_out single;
		single.px = single.py = single.pz = single.penetration = single.nx = single.ny = single.nz = 0.0f; single.friction = 0.5f; single.feature = 0u;
		if (live) {
			bool hit = nh_aabb_overlap(amin, amax, bmin, bmax) && (__float_as_uint(bmin.w) != __float_as_uint(amin.w)) && !dead;
			const uint64_t ka = nh_morton_of(nh_make3(amin.x, amin.y, amin.z), mscale, mmin), kb = nh_morton_of(nh_make3(bmin.x, bmin.y, bmin.z), mscale, mmin);
			const bool a_first = (ka < kb) || (ka == kb && ca < cb);
			invalid = !hit;
			// narrowphase "a" = later in Morton order (emit_pair)
			const bool a_is_cb = hit && a_first;
			const nh_xform A = a_is_cb ? XB : XA, B = a_is_cb ? XA : XB;
			const uint32_t ta = a_is_cb ? TB : TA, tb = a_is_cb ? TA : TB;
			const uint32_t a = invalid ? 0u : (a_is_cb ? cb : ca), b = invalid ? 0u : (a_is_cb ? ca : cb);
			const bool a_sph = a >= fv.nbox, b_sph = b >= fv.nbox;
			if (invalid) {
				key = ~0ull;
			} else if (!sph_rec) {
				const nh_BoxCollider sa = a_is_cb ? SB : SA, sb = a_is_cb ? SA : SB;
				nh_box_box_eval(A, B, sa.size, sb.size, ta, tb, bb);
				count = nh_bb_count(bb);
				key = (uint64_t)(bb.swapped ? tb : ta) | ((uint64_t)(bb.swapped ? ta : tb) << 32);
				body_a = bb.swapped ? B.body : A.body;
				body_b = bb.swapped ? A.body : B.body;
			} else {
				const float ra = a_is_cb ? RB : RA, rb = a_is_cb ? RA : RB;
				if (a_sph && b_sph) {
					// sphere-sphere: first = earlier in Morton order = pair.y (nudge.cpp:3775-3776)
					count = nh_sphere_sphere(rb, ra, B, A, &single);
					key = (uint64_t)tb | ((uint64_t)ta << 32);
					body_a = B.body; body_b = A.body;
				} else {
					// box-sphere: box is always "a" (nudge.cpp:3746-3751, 3759-3768)
					const nh_xform BX = a_sph ? B : A, SP = a_sph ? A : B;
					const nh_BoxCollider sz = a_sph ? (a_is_cb ? SA : SB) : (a_is_cb ? SB : SA);
					count = nh_box_sphere(sz.size, a_sph ? ra : rb, BX, SP, &single);
					key = (uint64_t)(a_sph ? tb : ta) | ((uint64_t)(a_sph ? ta : tb) << 32);
					body_a = BX.body; body_b = SP.body;
				}
			}
			// what a still step relies on, per record (k_narrowphase: still_record)
			const uint64_t was_key = (uint64_t)s_park[0][lane] | ((uint64_t)s_park[1][lane] << 32);
			const bool same_key = was_key == key || invalid;
			bool ok = same_key && count <= 4;
			if (ok && count) ok = (body_a == 0u) != (body_b == 0u) && own;
			if (!ok) { atomicAdd(&st->still_diff[!same_key ? 0 : (count > 4 ? 1 : 2)], 1u); changed = true; count = 0; bb.kind = 0; }
		}
		if (__builtin_amdgcn_ballot_w64(changed) != 0ull && lane == 0) atomicMax(&st->still_failed_seq, seq);
		// ---- the dense index.  Contacts gained or lost BEFORE this wave's records travel in three kinds of word, each written once per wave and none waited for in a
		// chain: the wave's own change (status[w], tagged with the step's number: no clearing), the sum over its SEGMENT of 64 waves and over its SUPER-SEGMENT of 4096
		// (low half: waves heard from, high half: the sum; cleared for the next launch by this launch's first wave).  A wave asks for the words of the waves before it in
		// its segment, the segments before its own in its super-segment and the super-segments before its own ALL AT ONCE -- one round of independent polls whatever its
		// index (an agent-scope load that has to see another XCD's store takes ~2.7 us here: the textbook look-back, a walk of dependent rounds, cost 120 us per launch)
		// -- right after the SAT, and looks at the answers when the contacts have been written out to LDS.
		const uint32_t w = blockIdx.x;
		const uint32_t cnt_old = s_park[2][lane];
		const int d = live ? count - (int)cnt_old : 0;
		int incl = d;
		for (int s = 1; s < 64; s <<= 1) { const int y = __shfl_up(incl, s); if ((int)lane >= s) incl += y; }
		const int agg = __shfl(incl, 63);
		// (4096 waves adding to ONE super-segment word queue up behind each other -- measured: the kernel took 1.0 ms -- so a super-segment hears from its 64 segments,
		// not from its waves: the wave that completes a segment passes the segment's sum on)
		unsigned long long seg_old = 0ull;
		if (lane == 0) {
			__hip_atomic_store(&fv.status[w], fz_pack(seq, NH_FZ_AGG, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			seg_old = atomicAdd(&fv.seg[w >> 6], ((unsigned long long)(uint32_t)agg << 32) + 1ull);
		}
		const bool want_a = lane < (w & 63u), want_b = lane < ((w >> 6) & 63u), want_c = lane < (w >> 12);
		const unsigned long long* const pa_at = &fv.status[(w & ~63u) + (want_a ? lane : 0u)];
		const unsigned long long* const pb_at = &fv.seg[((w >> 12) << 6) + (want_b ? lane : 0u)];
		const unsigned long long* const pc_at = &fv.sup[want_c ? lane : 0u];
		unsigned long long pa = __hip_atomic_load(pa_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		unsigned long long pb = __hip_atomic_load(pb_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		unsigned long long pc = __hip_atomic_load(pc_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		// ---- the contacts, out to LDS by (lane, slot): position + penetration and feature word; one normal and one friction per pair ----
		if (live && count) {
			if (bb.kind == 2) {
				int k = 0;
				nx = bb.wn[0]; ny = bb.wn[1]; nz = bb.wn[2];
#pragma unroll
				for (int index = 0; index < 16; ++index) {
					if (((bb.mask >> index) & 1u) && k < count && k < M) {
						nh_contact_out o;
						nh_bb_contact(bb, index, bb.lx[index], bb.ly[index], bb.lz[index], bb.penetration[index], bb.tags[index], o);
						s_c0[M * lane + k] = make_float4(o.px, o.py, o.pz, o.penetration);
						s_f[M * lane + k] = o.feature;
						++k;
					}
				}
			} else if (bb.kind == 1) {
				s_c0[M * lane] = make_float4(bb.edge.px, bb.edge.py, bb.edge.pz, bb.edge.penetration);
				s_f[M * lane] = bb.edge.feature;
				nx = bb.edge.nx; ny = bb.edge.ny; nz = bb.edge.nz; fric = bb.edge.friction;
			} else {
				s_c0[M * lane] = make_float4(single.px, single.py, single.pz, single.penetration);
				s_f[M * lane] = single.feature;
				nx = single.nx; ny = single.ny; nz = single.nz; fric = single.friction;
			}
		}
		if (lane == 0 && (uint32_t)seg_old == 63u) atomicAdd(&fv.sup[w >> 12], ((unsigned long long)((uint32_t)(seg_old >> 32) + (uint32_t)agg) << 32) + 64ull);
		// ---- ... and the answers: everything before this wave has said its change? ----
		int before = 0;
		{
			uint32_t spins = 0;
			bool gave_up = false;
			while (true) {
				const bool miss_a = want_a && fz_flag(pa, seq) == 0u, miss_b = want_b && (uint32_t)pb < 64u, miss_c = want_c && (uint32_t)pc < 4096u;
				if (__builtin_amdgcn_ballot_w64(miss_a || miss_b || miss_c) == 0ull) break;
				// (bounded: a word that never comes fails the step instead of hanging the GPU; and once the step has failed anyway nobody needs to wait)
				if (++spins > (1u << 18) || ((spins & 31u) == 0u && __hip_atomic_load(&st->still_failed_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seq)) { gave_up = true; break; }
				__builtin_amdgcn_s_sleep(2);
				if (miss_a) pa = __hip_atomic_load(pa_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (miss_b) pb = __hip_atomic_load(pb_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (miss_c) pc = __hip_atomic_load(pc_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			int part = 0;
			if (!gave_up) part = (want_a ? fz_value(pa) : 0) + (want_b ? (int)(uint32_t)(pb >> 32) : 0) + (want_c ? (int)(uint32_t)(pc >> 32) : 0);
			for (int s = 32; s >= 1; s >>= 1) part += __shfl_xor(part, s);
			before = part;
			if (gave_up && lane == 0) { atomicMax(&st->still_failed_seq, seq); atomicAdd(&st->still_diff[3], 1u); }
		}
		const int shift = before + incl - d;
		start = s_park[3][lane] + (uint32_t)shift;
		if (live) {
			if (shift) fv.start_sorted[p] = start;
			if (d) fv.cnt_sorted[p] = (uint32_t)count;
			if (!own) fv.sc_count_out[r] = s_park[4][lane];          // (a record nobody solves keeps what it had, in the other set too)
		}
		if (w + 1u == fv.pos_waves && lane == 0) {
			// the contact count of the step = the total behind the last record: kept like the starts
			const uint32_t total = fv.start_sorted[fv.nrec] + (uint32_t)(before + agg);
			if (before + agg) fv.start_sorted[fv.nrec] = total;
			st->contacts = total;
		}
		cnt = own ? min((uint32_t)count, 4u) : 0u;
		oc = min(s_park[4][lane], 4u);
#pragma unroll
		for (int k = 0; k < M; ++k) of[k] = s_park[5 + k][lane];
	}
	if (__builtin_amdgcn_ballot_w64(solves || free_body) == 0ull) return;          // (free_body: the lane that carries body 0 over)
	// ---- the body (k_solve_one_body<4, true, true, 1, true, true>): order of solving, warm start, rows, sweeps, export, gravity, advance ----
	uint32_t nf[M], sl[M], jm[M];
#pragma unroll
	for (int k = 0; k < M; ++k) { nf[k] = (uint32_t)k < cnt ? s_f[M * lane + k] : 0u; sl[k] = (uint32_t)k; jm[k] = 0xFFFFFFFFu; }
	{
		// dense index of slot k = first + rank of its feature word (ties by slot: k_gather_contacts); the order of solving = by slot key of that index, then by index
		// (k_adj_simple's rule); the cache entry a contact starts from = last step's slot with the same feature word
		uint32_t ci[M], kk[M];
#pragma unroll
		for (int k = 0; k < M; ++k) {
			uint32_t rk = 0;
#pragma unroll
			for (int j = 0; j < M; ++j) rk += ((uint32_t)j < cnt && (uint32_t)k < cnt && (nf[j] < nf[k] || (nf[j] == nf[k] && j < k))) ? 1u : 0u;
			const bool on = (uint32_t)k < cnt;
			ci[k] = on ? start + rk : 0xFFFFFFFFu;
			kk[k] = on ? fz_slot_key_of(ci[k]) : 0xFFFFFFFFu;
		}
#pragma unroll
		for (int q = 1; q < M; ++q) {
#pragma unroll
			for (int j = q; j > 0; --j) {
				const bool sw2 = (kk[j] < kk[j - 1]) || (kk[j] == kk[j - 1] && ci[j] < ci[j - 1]);
				const uint32_t tc = sw2 ? ci[j - 1] : ci[j], tk = sw2 ? kk[j - 1] : kk[j], ts = sw2 ? sl[j - 1] : sl[j];
				ci[j - 1] = sw2 ? ci[j] : ci[j - 1]; kk[j - 1] = sw2 ? kk[j] : kk[j - 1]; sl[j - 1] = sw2 ? sl[j] : sl[j - 1];
				ci[j] = tc; kk[j] = tk; sl[j] = ts;
			}
		}
#pragma unroll
		for (int q = 0; q < M; ++q) {
			const bool on = (uint32_t)q < cnt;
			uint32_t f = 0, wj = 0xFFFFFFFFu;
#pragma unroll
			for (int k = 0; k < M; ++k) if (sl[q] == (uint32_t)k) f = nf[k];
#pragma unroll
			for (int j = M - 1; j >= 0; --j) if ((uint32_t)j < oc && of[j] == f) wj = (uint32_t)j;
			jm[q] = on ? wj : 0xFFFFFFFFu;
			s_dst[lane * M + q] = on ? base + sl[q] : 0xFFFFFFFFu;          // (where the results go: read by the wave after the barrier before the result stores)
		}
		// this step's feature words and count become the slot cache's (the impulses follow at the end)
		if (own) {
#pragma unroll
			for (int k = 0; k < M; ++k) if ((uint32_t)k < cnt) fv.sc_feat_out[base + k] = nf[k];
			fv.sc_count_out[r] = cnt;
		}
	}
	// body state out of LDS (the LDS-direct loads of round 2: everything this wave has asked for has arrived by now -- the pair's registers came behind them)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
	float4 m0 = s_body[0][lane], m1 = s_body[1][lane];
	const float4 pr = s_body[2][lane], t0 = s_body[3][lane], t1 = s_body[4][lane];
	if (solves) {
		float vel[3] = { m0.x, m0.y, m0.z }, ang[3] = { m1.x, m1.y, m1.z };
		nh_gravity_damping(vel, ang, fv.gx_dt, fv.gy_dt, fv.gz_dt, fv.damping);
		m0.x = vel[0]; m0.y = vel[1]; m0.z = vel[2]; m1.x = ang[0]; m1.y = ang[1]; m1.z = ang[2];
	}
	nh_vel v = { m0.x, m0.y, m0.z, pr.w, m1.x, m1.y, m1.z, m1.w };
	const nh_quat qx = { t1.x, t1.y, t1.z, t1.w };
	const nh_inertia IX = nh_world_inertia(qx, pr.x, pr.y, pr.z);
	const nh_f3 posx = nh_make3(t0.x, t0.y, t0.z);
	nh_row1 rw[M];
	nh_state s[M];
#pragma unroll
	for (int q = 0; q < M; ++q) {
		if ((uint32_t)q < cnt) {
			const float4 c0 = s_c0[M * lane + sl[q]];
			nh_contact_in ci = { c0.x, c0.y, c0.z, c0.w, nx, ny, nz, fric };
			nh_build_row1(ci, posx, IX, v.mass_inverse, is_a_body, rw[q]);
			const float4 wv = s_imp[lane * M + (jm[q] != 0xFFFFFFFFu ? jm[q] : 0u)];
			const bool hit = jm[q] != 0xFFFFFFFFu;
			nh_warm_start1(rw[q], hit ? wv.x : 0.0f, hit ? wv.y : 0.0f, hit ? wv.z : 0.0f, v, is_a_body, s[q]);
		}
	}
	// the lane's collider for the NEXT step's transforms: 80 bytes straight to LDS while the sweeps run (the contact slots: nobody looks at them any more)
	float4* const s_ah = s_c0;
	float4* const s_ah4 = reinterpret_cast<float4*>(s_f);
	float ah_rad = 0.0f;
	{
		__builtin_amdgcn_wave_barrier();          // (every lane has read its contacts)
		const uint32_t cc = col != 0xFFFFFFFFu ? col : 0u;
		const bool is_box = cc < av.nbox;
		const float4* lp = reinterpret_cast<const float4*>(is_box ? av.box_xf + cc : av.sph_xf + (cc - av.nbox));
		const float4* sp4 = is_box ? reinterpret_cast<const float4*>(av.box_data + cc) : lp;          // (a sphere's radius -- four bytes -- comes by register)
		if (!is_box) ah_rad = av.sph_data[cc - av.nbox].radius;
		nh_glds16(lp, &s_ah[lane]); nh_glds16(lp + 1, &s_ah[64u + lane]);
		nh_glds16(av.fat_box + 2u * (size_t)cc, &s_ah[128u + lane]); nh_glds16(av.fat_box + 2u * (size_t)cc + 1u, &s_ah[192u + lane]);
		nh_glds16(sp4, &s_ah4[lane]);
	}
	if (__builtin_amdgcn_ballot_w64(solves && is_a_body) == 0ull) {
		for (uint32_t it = 0; it < fv.iterations; ++it) {
#pragma unroll
			for (int q = 0; q < M; ++q)
				if ((uint32_t)q < cnt) nh_apply_one1(rw[q], s[q], v, false);
		}
	} else {
		for (uint32_t it = 0; it < fv.iterations; ++it) {
#pragma unroll
			for (int q = 0; q < M; ++q)
				if ((uint32_t)q < cnt) nh_apply_one1(rw[q], s[q], v, is_a_body);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the five loads issued before the sweeps: nothing else of this wave is in flight)
	const float4 ah_l0 = s_ah[lane], ah_l1 = s_ah[64u + lane], ah_fmn = s_ah[128u + lane], ah_fmx = s_ah[192u + lane], ah_sz = s_ah4[lane];
	// results: the exported impulse (update_cached_impulses, nudge.cpp:4872-4882) per contact, written by the wave slot by slot
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for (int q = 0; q < M; ++q) {
		if ((uint32_t)q < cnt) {
			float ox, oy, oz;
			nh_export_impulse1(rw[q], s[q], ox, oy, oz);
			s_imp[lane * M + q] = make_float4(ox, oy, oz, 0.0f);
		}
	}
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for (int k = 0; k < M; ++k) {
		const uint32_t e = 64u * k + lane;
		const uint32_t c = pos_wave ? s_dst[e] : 0xFFFFFFFFu;
		if (c != 0xFFFFFFFFu) fv.sc_imp_out[c] = s_imp[e];
	}
	uint32_t nmin[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, nmax[3] = { 0u, 0u, 0u }, ntop = 0u;
	bool nfail = false;
	if (solves) {
		reinterpret_cast<float4*>(fv.mom_out + body)[0] = make_float4(v.vx, v.vy, v.vz, v.mass_inverse);
		reinterpret_cast<float4*>(fv.mom_out + body)[1] = make_float4(v.wx, v.wy, v.wz, v.w_unused);
		float pos[3] = { t0.x, t0.y, t0.z }, rot[4] = { t1.x, t1.y, t1.z, t1.w };
		const float vel[3] = { v.vx, v.vy, v.vz }, ang[3] = { v.wx, v.wy, v.wz };
		const uint8_t idle_out = nh_advance_body(pos, rot, vel, ang, (uint8_t)idle0, fv.time_step);
		fv.idle_out[body] = idle_out;
		reinterpret_cast<float4*>(fv.xf_out + body)[0] = make_float4(pos[0], pos[1], pos[2], t0.w);
		reinterpret_cast<float4*>(fv.xf_out + body)[1] = make_float4(rot[0], rot[1], rot[2], rot[3]);
		// the next step's k_xform<true> for the body's collider (nh_collide.hip: k_xform; nh_solve.hip: AHEAD)
		ntop = idle_out; nfail = idle_out == 0xffu;
		const uint32_t c = col;
		if (c != 0xFFFFFFFFu) {
			const bool is_box = c < av.nbox;
			float4 sz = ah_sz;
			if (!is_box) sz.x = ah_rad;
			const nh_quat bq = { rot[0], rot[1], rot[2], rot[3] };
			const nh_quat lq = { ah_l1.x, ah_l1.y, ah_l1.z, ah_l1.w };
			const nh_f3 pw = nh_rotate(bq, nh_make3(ah_l0.x, ah_l0.y, ah_l0.z)) + nh_make3(pos[0], pos[1], pos[2]);
			const nh_quat q = nh_qmul(bq, lq);
			float sx, sy, sz2;
			if (is_box) {
				const nh_m33 mm = nh_matrix(q);
				const nh_f3 c0 = mm.c0 * sz.x, c1 = mm.c1 * sz.y, c2 = mm.c2 * sz.z;
				sx = fabsf(c0.x) + fabsf(c1.x) + fabsf(c2.x);
				sy = fabsf(c0.y) + fabsf(c1.y) + fabsf(c2.y);
				sz2 = fabsf(c0.z) + fabsf(c1.z) + fabsf(c2.z);
			} else {
				sx = sy = sz2 = sz.x;
			}
			const float mnx = pw.x - sx, mny = pw.y - sy, mnz = pw.z - sz2, mxx = pw.x + sx, mxy = pw.y + sy, mxz = pw.z + sz2;
			float4* const wx = reinterpret_cast<float4*>(av.xf) + 2u * (size_t)c;
			wx[0] = make_float4(pw.x, pw.y, pw.z, ah_l0.w);          // (l0.w: the collider's body -- this one)
			wx[1] = make_float4(q.x, q.y, q.z, q.s);
			av.aabb_min[c] = make_float4(mnx, mny, mnz, ah_l0.w);
			av.aabb_max[c] = make_float4(mxx, mxy, mxz, __uint_as_float(0u));          // (plain form: nobody asleep)
			nfail |= !(mnx >= ah_fmn.x && mny >= ah_fmn.y && mnz >= ah_fmn.z && mxx <= ah_fmx.x && mxy <= ah_fmx.y && mxz <= ah_fmx.z);       // (NaN: out)
			nmin[0] = nmax[0] = nh_float_flip(mnx); nmin[1] = nmax[1] = nh_float_flip(mny); nmin[2] = nmax[2] = nh_float_flip(mnz);
		}
	} else if (free_body) {
		// body 0: carried over to the other set as it is
		reinterpret_cast<float4*>(fv.mom_out)[0] = m0; reinterpret_cast<float4*>(fv.mom_out)[1] = m1;
		reinterpret_cast<float4*>(fv.xf_out)[0] = t0; reinterpret_cast<float4*>(fv.xf_out)[1] = t1;
		fv.idle_out[0] = fv.idle_in[0];
	}
	{
		// the wave's share of the NEXT step's scene bounds, largest idle counter, and whether that step would have failed in its k_xform<true>
		const uint32_t np = parity ^ 1u;
#pragma unroll
		for (int k = 0; k < 3; ++k)
			for (int d = 32; d >= 1; d >>= 1) { nmin[k] = min(nmin[k], (uint32_t)__shfl_xor((int)nmin[k], d)); nmax[k] = max(nmax[k], (uint32_t)__shfl_xor((int)nmax[k], d)); }
		for (int d = 32; d >= 1; d >>= 1) ntop = max(ntop, (uint32_t)__shfl_xor((int)ntop, d));
		const unsigned long long bad = __ballot(nfail);
		uint32_t* const part = &st->ahead_part[np][blockIdx.x & (NH_AHEAD_PARTS - 1u)][0];
		const uint32_t mine_min = lane == 0u ? nmin[0] : (lane == 1u ? nmin[1] : nmin[2]);
		const uint32_t mine_max = lane == 3u ? nmax[0] : (lane == 4u ? nmax[1] : (lane == 5u ? nmax[2] : ntop));
		if (lane < 3u) atomicMin(&part[lane], mine_min);
		else if (lane < 7u) atomicMax(&part[lane], mine_max);
		else if (lane == 7u && bad != 0ull) atomicMax(&st->ahead_failed_seq, seq + 1u);
	}
	if (fv.debug_fail && blockIdx.x == fv.pos_waves / 2u && lane == 0) { atomicMax(&st->still_failed_seq, seq); atomicAdd(&st->still_diff[3], 1u); }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------------------------------------
static void fz_swap_sets(nh_context* ctx) {
	nh_FusedState& f = ctx->fused;
	std::swap(ctx->sc_imp, f.sc_imp_alt); std::swap(ctx->sc_feat, f.sc_feat_alt); std::swap(ctx->sc_count, f.sc_count_alt);
	std::swap(ctx->own_xf, f.own_xf_alt); std::swap(ctx->own_aabb_min, f.own_aabb_min_alt); std::swap(ctx->own_aabb_max, f.own_aabb_max_alt);
	f.cur ^= 1u;
}

void nh_fused_free(nh_context* ctx) {
	nh_FusedState& f = ctx->fused;
	void* bufs[] = { f.fz, f.status, f.seg, f.sc_imp_alt, f.sc_feat_alt, f.sc_count_alt, f.own_xf_alt, f.own_aabb_min_alt, f.own_aabb_max_alt, f.shadow_xf, f.shadow_mom, f.shadow_idle };
	// (the pointers may have changed places with the context's: both sets are freed, whichever name they go by -- nh_destroy frees ctx->sc_* / own_* itself)
	for (void* b : bufs) if (b) (void)hipFree(b);
	memset(&f, 0, sizeof(f));
}

// buffers of the second set, sized like the first (allocated when the first fused step of a world is about to be launched)
static int fz_ensure(nh_context* ctx, uint32_t B, uint32_t C) {
	nh_FusedState& f = ctx->fused;
	const uint32_t pair_cap = ctx->lay_capacity;
	if (f.pair_capacity != pair_cap) {
		void** bufs[] = { (void**)&f.fz, (void**)&f.status, (void**)&f.seg, (void**)&f.sc_imp_alt, (void**)&f.sc_feat_alt, (void**)&f.sc_count_alt };
		for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
		f.pair_capacity = 0; f.map_valid = false;
		const size_t cache_slots = (size_t)5 * pair_cap + 64u;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.fz, sizeof(uint4) * ((size_t)pair_cap + 64u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.status, sizeof(unsigned long long) * ((size_t)pair_cap / 64u + 64u)));
		NH_HIP_CHECK(ctx, hipMemsetAsync(f.status, 0, sizeof(unsigned long long) * ((size_t)pair_cap / 64u + 64u), ctx->stream));
		f.seg_words = pair_cap / 4096u + 64u + 64u;          // segments (one per 64 position waves), then up to 64 super-segments
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.seg, sizeof(unsigned long long) * 2u * (size_t)f.seg_words));
		NH_HIP_CHECK(ctx, hipMemsetAsync(f.seg, 0, sizeof(unsigned long long) * 2u * (size_t)f.seg_words, ctx->stream));
		f.seg_clean[0] = f.seg_clean[1] = true;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.sc_imp_alt, sizeof(float4) * cache_slots));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.sc_feat_alt, sizeof(uint32_t) * cache_slots));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.sc_count_alt, sizeof(uint32_t) * (size_t)pair_cap + 64u));
		f.pair_capacity = pair_cap;
	}
	if (f.collider_capacity != ctx->own_capacity || f.collider_capacity < C) {
		void** bufs[] = { (void**)&f.own_xf_alt, (void**)&f.own_aabb_min_alt, (void**)&f.own_aabb_max_alt };
		for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
		f.collider_capacity = 0;
		const size_t cc = ctx->own_capacity;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.own_xf_alt, sizeof(nh_xform) * cc + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.own_aabb_min_alt, sizeof(float4) * cc + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.own_aabb_max_alt, sizeof(float4) * cc + 64u));
		f.collider_capacity = ctx->own_capacity;
	}
	if (f.body_capacity < B) {
		void** bufs[] = { (void**)&f.shadow_xf, (void**)&f.shadow_mom, (void**)&f.shadow_idle };
		for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
		f.body_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.shadow_xf, sizeof(nh_Transform) * (size_t)B + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.shadow_mom, sizeof(nh_BodyMomentum) * (size_t)B + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&f.shadow_idle, (size_t)B + 64u));
		f.body_capacity = B;
	}
	return NH_OK;
}

// May the still step nh_collide is about to launch -- plain form, started at the narrowphase (xform ahead), another sub-step of the call behind it -- be a fused one?
// Decided on the host from what it knows; everything else is checked by the kernel.
bool nh_fused_wanted(nh_context* ctx, uint32_t B, uint32_t C, uint32_t steps_left) {
	nh_FusedState& f = ctx->fused;
	const nh_StillStep& ss = ctx->still;
	if (!f.enabled || ss.no_ahead || ss.ahead_world_bad || !ss.ahead_map_ok || !ctx->body_col || ctx->body_col_capacity < B || !ctx->own_xf) return false;
	if (ctx->env_solver_waves == 4 || ctx->env_solver_waves == 2) return false;
	if (ctx->timing && ctx->timing_filter.empty()) return false;          // (per-kernel timing of every launch: the two-kernel form is what the tables name)
	if (ctx->stream_state.every) return false;                            // (frames are copied from the caller's arrays behind every nh_advance)
	if (!ctx->h_state->records || ctx->h_state->records > ctx->lay_capacity || ctx->h_state->records > (1u << 24)) return false;          // (the look-back's three levels: 64^3 waves)
	// in pairs: bodies are read from one set of arrays and written to the other, and the call must end in the caller's
	if (f.cur == 0u) return steps_left >= 2u;
	return steps_left >= 1u;
	(void)C;
}

// the state is in the second set and the next step cannot be a fused one: bodies back into the caller's arrays, library-owned buffers change names
int nh_fused_come_home(nh_context* ctx) {
	nh_FusedState& f = ctx->fused;
	if (f.cur == 0u) return NH_OK;
	const nh_BodyData& b = ctx->still.lay_bodies;
	NH_LAUNCH(ctx, "fused_copy_bodies", k_fused_copy_bodies, nh_grid_for(b.count, 256, 2048), 256, b.count, reinterpret_cast<const float4*>(f.shadow_xf), reinterpret_cast<const float4*>(f.shadow_mom),
	          f.shadow_idle, reinterpret_cast<float4*>(b.transforms), reinterpret_cast<float4*>(b.momentum), b.idle_counters);
	// (the library-owned sets simply keep the names they have: ctx->sc_* / own_* ARE the current ones)
	f.cur = 0u; f.homecomings++;
	return NH_OK;
}

// a fused launch whose verdict was "failed" (or that was launched behind one): it wrote nothing that counts -- the sets change back
void nh_fused_void_last(nh_context* ctx, uint32_t launches) {
	nh_FusedState& f = ctx->fused;
	for (uint32_t k = 0; k < launches && f.unconfirmed; ++k) { fz_swap_sets(ctx); f.unconfirmed--; f.steps--; }
}

int nh_fused_launch(nh_context* ctx, const nh_BodyData* bodies, uint32_t iterations, uint32_t guard_seq) {
	nh_FusedState& f = ctx->fused;
	nh_StillStep& ss = ctx->still;
	const nh_ColliderData& cd = ss.colliders;
	const uint32_t B = bodies->count, C = cd.boxes.count + cd.spheres.count;
	{ int rc = fz_ensure(ctx, B, C); if (rc) return rc; }
	nh_DevState* st = ctx->d_state;
	const uint32_t nrec = ctx->h_state->records;
	if (!f.map_valid) {
		const uint8_t* gen = (C < (1u << NH_GEN_SHIFT) && !ctx->env_no_incremental) ? ctx->fat_gen : (const uint8_t*)nullptr;
		NH_LAUNCH(ctx, "fused_map", k_fused_map, nh_grid_for(nrec, 256, 2048), 256, st, ctx->sort_sorted_idx, ctx->body_rec, ctx->fat_pairs, ctx->fat_pair_capacity, ctx->lay_capacity, gen,
		          ctx->own_aabb_min, f.fz);
		f.map_valid = true; f.map_records = nrec;
	}
	if (!f.statics_copied) {
		NH_LAUNCH(ctx, "fused_copy_static", k_fused_copy_static, nh_grid_for(C, 256, 2048), 256, C, reinterpret_cast<const float4*>(ctx->own_xf), ctx->own_aabb_min, ctx->own_aabb_max,
		          reinterpret_cast<float4*>(f.own_xf_alt), f.own_aabb_min_alt, f.own_aabb_max_alt);
		f.statics_copied = true;
	}
	nh_FusedView fv;
	fv.fz = f.fz; fv.nrec = nrec; fv.pos_waves = (nrec + 63u) / 64u; fv.status = f.status;
	{
		// the segment words of this launch must be zero: the fused launch before it cleared them if it was the step before this one -- otherwise a memset does
		const uint32_t par = ctx->step_parity & 1u;
		const uint32_t nseg = (fv.pos_waves + 63u) / 64u;
		if (nseg + 64u > f.seg_words || fv.pos_waves > (1u << 18)) return NH_ERR_INVALID;          // (64 super-segments of 4096 waves: 16.7 M records; nh_fused_wanted never asks for more)
		if (!f.seg_clean[par]) NH_HIP_CHECK(ctx, hipMemsetAsync(f.seg + (size_t)par * f.seg_words, 0, sizeof(unsigned long long) * f.seg_words, ctx->stream));
		fv.seg = f.seg + (size_t)par * f.seg_words; fv.sup = fv.seg + (f.seg_words - 64u); fv.seg_next = f.seg + (size_t)(par ^ 1u) * f.seg_words; fv.seg_words = f.seg_words;
		f.seg_clean[par] = false; f.seg_clean[par ^ 1u] = true;
	}
	fv.xf = ctx->own_xf; fv.aabb_min = ctx->own_aabb_min; fv.aabb_max = ctx->own_aabb_max; fv.ctag = ctx->own_ctag;
	fv.box_data = cd.boxes.data; fv.sph_data = cd.spheres.data; fv.nbox = cd.boxes.count; fv.pair_cap = ctx->lay_capacity;
	fv.rec_key = ctx->sort_keys_by_position; fv.cnt_sorted = ctx->cnt_sorted; fv.start_sorted = ctx->start_sorted;
	fv.sc_imp_in = ctx->sc_imp; fv.sc_feat_in = ctx->sc_feat; fv.sc_count_in = ctx->sc_count;
	fv.sc_imp_out = f.sc_imp_alt; fv.sc_feat_out = f.sc_feat_alt; fv.sc_count_out = f.sc_count_alt;
	const bool home = f.cur == 0u;
	fv.xf_in = home ? bodies->transforms : f.shadow_xf; fv.mom_in = home ? bodies->momentum : f.shadow_mom; fv.idle_in = home ? bodies->idle_counters : f.shadow_idle;
	fv.xf_out = home ? f.shadow_xf : bodies->transforms; fv.mom_out = home ? f.shadow_mom : bodies->momentum; fv.idle_out = home ? f.shadow_idle : bodies->idle_counters;
	fv.props = bodies->properties; fv.nbodies = B;
	fv.body_rec = ctx->body_rec; fv.body_col = ctx->body_col;
	fv.parity = ctx->step_parity; fv.seq = ctx->collide_seq; fv.guard_seq = guard_seq; fv.iterations = iterations;
	fv.debug_fail = 0u;
	if (f.debug_fail_in) { if (--f.debug_fail_in == 0u) fv.debug_fail = 1u; }
	fv.gx_dt = ctx->grav.gx_dt; fv.gy_dt = ctx->grav.gy_dt; fv.gz_dt = ctx->grav.gz_dt; fv.damping = ctx->grav.damping; fv.time_step = ctx->grav.time_step;
	nh_AheadView av = { ctx->body_col, cd.boxes.transforms, cd.boxes.data, cd.spheres.transforms, cd.spheres.data, cd.boxes.count, f.own_xf_alt, f.own_aabb_min_alt, f.own_aabb_max_alt, ctx->fat_box };
	const uint32_t grid = fv.pos_waves + (B + 63u) / 64u;
	NH_LAUNCH(ctx, "still_fused", k_still_fused, grid, 64, st, fv, av);
	fz_swap_sets(ctx);
	f.unconfirmed++; f.steps++;
	return NH_OK;
}
