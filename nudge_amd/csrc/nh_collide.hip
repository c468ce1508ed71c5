// nh_collide.hip -- nh_collide: world transforms + AABBs, exact-pair-set broadphase, narrowphase, islands /
// sleeping, tag-ordered contact output.  Replaces reference nudge.cpp:3000-4009.
//
// What is kept from the reference is the RESULT: the set of strictly overlapping AABB pairs
// (nudge.cpp:3306-3308), which collider of a pair plays "a" in the narrowphase (the one later in Morton order
// of the AABB min corners, nudge.cpp:3086-3172, 3495, 1202-1203), the per-pair contact arithmetic
// (nh_narrowphase.h), island membership / sleeping rules at both levels (coarse: nudge.cpp:3500-3703, over the broadphase pairs, before
// the narrowphase; fine: 3788-4006, over the contacts) and the tag order of contacts.
// How it is computed is GPU-first: the reference's O(C^2/1024) grouped all-pairs sweep is replaced by a
// uniform grid over "small" colliders (cell = power of two >= their largest extent) plus a brute-force pass for
// the few large ones (ground slabs); union-find runs lock-free with atomicCAS hooking.
#include "nh_internal.h"
#include "nh_narrowphase.h"
#include <type_traits>
#include "nh_solver.h"          // (nh_is_inert: k_pair_begin checks body 0 for the step it opens)
#include <stdlib.h>

// ---- 1. world transforms, AABBs, bounds, extent histogram ---------------------------------------------
// aabb_min[c] = (min.xyz, bits(body)), aabb_max[c] = (max.xyz, bits(max extent))
// 1024-thread workgroups, one per CU: the scene bounds / histogram end in a handful of global atomics per WORKGROUP on the same
// few addresses (~12 ns each, serialised), so the fewer workgroups the better.
// inflation of the kept pair list's boxes: 1/32 of the last grid's cell (nothing is known before the first grid)
#define NH_ESC_MAX 4096u             // colliders re-inserted per step at most (more leave their boxes: rebuild)
// leaver x (moved + large) box tests a step may spend before regrouping everybody is the cheaper way.  One lane per combination, 64 K lanes: sixteen tests per collider of the
// world cost what a fifth of a rebuild costs (4.4 M tests for 2048 boxes in the air over a million at rest: ~20 us against ~250 us -- and a rebuild voids the layout the still
// steps live on; rounds 3-5 allowed half a test per collider, which sent exactly that world into a rebuild on 84 of 96 steps)
// (... and when a thirty-second of the world has been re-inserted since the last rebuild -- stale grid entries, dead pairs in the kept list, a layout that is no longer "in place" --
// regrouping everybody IS the cheaper way: the old half-a-test rule did that by accident for small worlds in which everybody had moved once)
#define NH_REINSERT_COMBOS(ncolliders) ((uint64_t)(ncolliders) * 16u)
#define NH_MOVED_MAX 16384u          // colliders re-inserted since the last rebuild at most (their grid entries are stale: the next leaver is tested against each of them)
// (NH_GEN_SHIFT / NH_GEN_INDEX -- kept pairs carry the generation stamps of their two colliders above the 24-bit collider index: nh_internal.h)
__device__ __forceinline__ float nh_kept_margin(float cell_inv) { return cell_inv > 0.0f ? 0.03125f / cell_inv : 0.05f; }

// STILL (nh_collide, "9. still steps"): the step was launched as one in which nothing changes structurally.  The kernel then only CHECKS -- a collider
// outside its inflated box, a body asleep: the step fails (st->still_fail) and is replayed in full -- and gathers nothing a rebuild would need; it also does
// what k_collide_begin does for a full step as far as a still step needs it (largest idle counter; next step's flags and bounds cleared).
template<bool STILL>
__global__ __launch_bounds__(1024) void k_xform(nh_DevState* __restrict__ st,
                                               const nh_Transform* __restrict__ body_xf,
                                               const nh_Transform* __restrict__ box_xf, const nh_BoxCollider* __restrict__ box_data, const uint32_t* __restrict__ box_tags, uint32_t nbox,
                                               const nh_Transform* __restrict__ sph_xf, const nh_SphereCollider* __restrict__ sph_data, const uint32_t* __restrict__ sph_tags, uint32_t nsph,
                                               nh_xform* __restrict__ xf, float4* __restrict__ aabb_min, float4* __restrict__ aabb_max, uint32_t* __restrict__ ctag,
                                               uint4* __restrict__ zero16, uint32_t zero16_count, float4* fat_box, uint32_t direct,
                                               uint8_t* __restrict__ gen, uint32_t* __restrict__ esc_mark, uint32_t* __restrict__ esc_list, uint32_t* __restrict__ moved_list, uint32_t seq,
                                               const uint8_t* __restrict__ idle = nullptr, uint32_t nbodies = 0u, uint32_t parity = 0u, uint8_t* __restrict__ awake_out = nullptr,
                                               // STILL, sleepers form, a sub-step of an nh_step call that is not its first: a collider whose body is asleep now and carried the
                                               // mark when the sub-step before this one wrote these arrays has not moved in between (nobody advances a sleeping body, and the
                                               // caller cannot touch anything inside a call): its world transform, AABB, tag and mark stand -- only its share of the scene bounds
                                               // is taken (from the AABB on file).  What an asleep collider costs is 37 bytes instead of 217
                                               uint32_t skip_asleep = 0u) {
	// `awake_out` (STILL): SLEEPERS form (nh_internal.h) -- a body asleep does not fail the step; it is counted, noted (awake_out[i]: the active list is a view of it) and
	// its colliders carry the mark in aabb_max.w, where the still narrowphase finds it without another memory round trip
	__shared__ uint32_t hist[STILL ? 1 : 1024];
	__shared__ uint32_t red[12];
	__shared__ uint32_t s_asleep;
	if (!STILL) hist[threadIdx.x] = 0;
	if (threadIdx.x < 3) { red[threadIdx.x] = 0xffffffffu; red[3 + threadIdx.x] = 0u; red[6 + threadIdx.x] = 0xffffffffu; red[9 + threadIdx.x] = 0u; }
	__syncthreads();
	uint32_t n = nbox + nsph;
	// (the cell counters of the counting sort further down are cleared here: one launch less than a memset of their own)
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < zero16_count; k += gridDim.x * blockDim.x) zero16[k] = make_uint4(0u, 0u, 0u, 0u);
	uint32_t lmin[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, lmax[3] = { 0, 0, 0 };
	// bounds of the "small" colliders under LAST step's size split (nh_grid_setup keeps them if the split has not moved)
	const uint32_t guess_exp = st->small_exp;
	uint32_t gmn[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, gmx[3] = { 0, 0, 0 };
	// kept pair list (below, "6. kept pairs"): is every AABB still inside the inflated box the kept pairs were found with?
	// (`direct`: this step searches the grid with the exact boxes and keeps nothing, nh_collide)
	const bool fat_valid = st->fat_valid != 0u && !direct;
	const float margin = direct ? 0.0f : nh_kept_margin(st->cell_inv);           // (k_grid_setup records the same value for k_cell_keys before it moves the grid)
	bool escaped = STILL && !fat_valid;
	if (STILL) {
		// the per-body part of k_collide_begin: nobody may be asleep (the host has ruled it out; checked), the largest idle counter for its next prediction
		uint32_t top = 0, nas = 0;
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1u; i < nbodies; i += gridDim.x * blockDim.x) {
			const uint32_t cnt = idle[i]; top = max(top, cnt);
			if (awake_out) { awake_out[i] = cnt == 0xffu ? (uint8_t)0 : (uint8_t)1; nas += cnt == 0xffu ? 1u : 0u; }
			else escaped |= cnt == 0xffu;
		}
		for (int d = 32; d >= 1; d >>= 1) { top = max(top, (uint32_t)__shfl_xor((int)top, d)); nas += (uint32_t)__shfl_xor((int)nas, d); }
		if (threadIdx.x < 64u) { red[0] = 0u; s_asleep = 0u; }          // (red[0] reused below after the barrier; here: the workgroup's largest counter)
		__syncthreads();
		if (nh_lane() == 0 && top) atomicMax(&red[0], top);
		if (nh_lane() == 0 && nas) atomicAdd(&s_asleep, nas);
		__syncthreads();
		if (threadIdx.x == 0) {
			if (red[0]) atomicMax(&st->max_idle[parity], red[0]);
			if (s_asleep) atomicAdd(&st->still_asleep[parity], s_asleep);
			if (blockIdx.x == 0) {
				// what the NEXT step (the other parity) accumulates into: nobody touches it during this one
				st->max_idle[parity ^ 1u] = 0u; st->delta_count[parity ^ 1u] = 0u; st->delta_overflow[parity ^ 1u] = 0u; st->still_esc[parity ^ 1u] = 0u;
				st->still_asleep[parity ^ 1u] = 0u; st->still_sleeping[parity ^ 1u] = 0u; st->still_culled[parity ^ 1u] = 0u;
				for (int k = 0; k < 3; ++k) { st->still_smin[parity ^ 1u][k] = 0xffffffffu; st->still_smax[parity ^ 1u][k] = 0u; }
				st->still_fat0 = st->fat_count + st->fat_count_sph;          // (nobody appends to the kept list while this kernel runs: k_reinsert's verdict needs a value that stands still)
			}
		}
		// (sleepers form: where THIS step's narrowphase counts its sleeping pairs)
		if (awake_out && blockIdx.x == 0 && threadIdx.x < NH_SLEEP_PARTS * 2u) (&st->sleep_part[parity][0][0])[threadIdx.x] = 0u;
		// (xform ahead, nh_internal.h: what this step's solver may gather for the next step)
		if (blockIdx.x == 0 && threadIdx.x < NH_AHEAD_PARTS * 8u) (&st->ahead_part[parity ^ 1u][0][0])[threadIdx.x] = (threadIdx.x & 7u) < 3u ? 0xffffffffu : 0u;
		__syncthreads();
		if (threadIdx.x < 3) { red[threadIdx.x] = 0xffffffffu; red[3 + threadIdx.x] = 0u; }
		__syncthreads();
	}
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		if (STILL && skip_asleep) {
			const float4 omin = aabb_min[c];
			const uint32_t omark = reinterpret_cast<const uint32_t*>(aabb_max)[4u * (size_t)c + 3u], ob = __float_as_uint(omin.w);
			if (omark == 1u && ob != 0u && ob < nbodies && idle[ob] == 0xffu) {
				uint32_t f;
				f = nh_float_flip(omin.x); lmin[0] = min(lmin[0], f); lmax[0] = max(lmax[0], f);
				f = nh_float_flip(omin.y); lmin[1] = min(lmin[1], f); lmax[1] = max(lmax[1], f);
				f = nh_float_flip(omin.z); lmin[2] = min(lmin[2], f); lmax[2] = max(lmax[2], f);
				continue;
			}
		}
		bool is_box = c < nbox;
		nh_Transform l = is_box ? box_xf[c] : sph_xf[c - nbox];
		nh_Transform b = body_xf[l.body];
		// Transform * Transform (nudge.cpp:1165-1175)
		nh_quat bq = { b.rotation[0], b.rotation[1], b.rotation[2], b.rotation[3] };
		nh_quat lq = { l.rotation[0], l.rotation[1], l.rotation[2], l.rotation[3] };
		nh_f3 p = nh_rotate(bq, nh_make3(l.position[0], l.position[1], l.position[2])) + nh_make3(b.position[0], b.position[1], b.position[2]);
		nh_quat q = nh_qmul(bq, lq);
		nh_xform w;
		w.px = p.x; w.py = p.y; w.pz = p.z; w.body = l.body;
		w.qx = q.x; w.qy = q.y; w.qz = q.z; w.qs = q.s;
		float sx, sy, sz;
		if (is_box) {
			// |R| * size (nudge.cpp:3027-3037)
			nh_m33 m = nh_matrix(q);
			nh_BoxCollider bc = box_data[c];
			nh_f3 c0 = m.c0 * bc.size[0], c1 = m.c1 * bc.size[1], c2 = m.c2 * bc.size[2];
			sx = fabsf(c0.x) + fabsf(c1.x) + fabsf(c2.x);
			sy = fabsf(c0.y) + fabsf(c1.y) + fabsf(c2.y);
			sz = fabsf(c0.z) + fabsf(c1.z) + fabsf(c2.z);
			ctag[c] = box_tags[c];
		} else {
			sx = sy = sz = sph_data[c - nbox].radius;
			ctag[c] = sph_tags[c - nbox];
		}
		float mnx = p.x - sx, mny = p.y - sy, mnz = p.z - sz;
		float mxx = p.x + sx, mxy = p.y + sy, mxz = p.z + sz;
		float ext = fmaxf(fmaxf(mxx - mnx, mxy - mny), mxz - mnz);
		xf[c] = w;
		aabb_min[c] = make_float4(mnx, mny, mnz, __uint_as_float(l.body));
		// (STILL: nobody reads the extent; the fourth word says whether the collider's body is asleep -- sleepers form)
		if (STILL) aabb_max[c] = make_float4(mxx, mxy, mxz, __uint_as_float((awake_out && l.body && idle[l.body] == 0xffu) ? 1u : 0u));
		else aabb_max[c] = make_float4(mxx, mxy, mxz, ext);
		// the inflated box a rebuild would sort and search (k_cell_keys writes it then): the grid's statistics (size class, bounds of the small
		// ones) are taken from it; the Morton frame (lmin / lmax) is the exact AABBs' (nudge.cpp:3086-3100)
		const float fx0 = mnx - margin, fy0 = mny - margin, fz0 = mnz - margin, fx1 = mxx + margin, fy1 = mxy + margin, fz1 = mxz + margin;
		const float fext = fmaxf(fmaxf(fx1 - fx0, fy1 - fy0), fz1 - fz0);
		bool out = false;
		if (fat_valid) {
			const float4 fmn = fat_box[2u * c], fmx = fat_box[2u * c + 1u];
			out = !(mnx >= fmn.x && mny >= fmn.y && mnz >= fmn.z && mxx <= fmx.x && mxy <= fmx.y && mxz <= fmx.z);       // (NaN: out)
		}
		if (gen) {
			// Re-insertion ("7." below): a SMALL collider that has left its box takes the candidate box as its new one right here, gets the next generation
			// stamp (its kept pairs die) and goes on this step's list of leavers; k_reinsert finds its new pairs.  Large colliders, worn-out stamps, full
			// lists and boxes that are no longer small for the grid end in the rebuild instead.
			// (STILL with stamps = a still step in MOVERS form, nh_internal.h "LOCAL speculation": the same, counted by step parity; whoever cannot be re-inserted fails the step)
			const uint32_t g = out ? gen[c] : 0u;
			bool again = out && !(g & 0x80u) && (g & 0x7Fu) < 127u && ((__float_as_uint(fext) >> 21) & 1023u) <= guess_exp && fext == fext;
			const uint32_t eslot = nh_wave_reserve1(STILL ? &st->still_esc[parity] : &st->esc_count, again);
			const uint32_t mslot = nh_wave_reserve1(&st->moved_count, again && (g & 0x7Fu) == 0u);
			if (again && (eslot >= NH_ESC_MAX || ((g & 0x7Fu) == 0u && mslot >= NH_MOVED_MAX))) again = false;
			if (again) {
				esc_list[eslot] = c;
				if ((g & 0x7Fu) == 0u) moved_list[mslot] = c;
				gen[c] = (uint8_t)(g + 1u); esc_mark[c] = seq;
				fat_box[2u * c] = make_float4(fx0, fy0, fz0, __uint_as_float(l.body)); fat_box[2u * c + 1u] = make_float4(fx1, fy1, fz1, fext);
				out = false;
			}
		}
		escaped |= out;
		uint32_t f;
		f = nh_float_flip(mnx); lmin[0] = min(lmin[0], f); lmax[0] = max(lmax[0], f);
		f = nh_float_flip(mny); lmin[1] = min(lmin[1], f); lmax[1] = max(lmax[1], f);
		f = nh_float_flip(mnz); lmin[2] = min(lmin[2], f); lmax[2] = max(lmax[2], f);
		// exponent histogram: one LDS atomic per distinct exponent in the wave (usually 2-3), not one per lane
		if (STILL) continue;
		uint32_t ebin = (__float_as_uint(fext) >> 21) & 1023u;          // size class: exponent + two mantissa bits (quarter octaves)
		if (ebin <= guess_exp && fext == fext) {
			f = nh_float_flip(fx0); gmn[0] = min(gmn[0], f); gmx[0] = max(gmx[0], f);
			f = nh_float_flip(fy0); gmn[1] = min(gmn[1], f); gmx[1] = max(gmx[1], f);
			f = nh_float_flip(fz0); gmn[2] = min(gmn[2], f); gmx[2] = max(gmx[2], f);
		}
		unsigned long long todo = __ballot(1);
		while (todo) {
			uint32_t e0 = (uint32_t)__shfl((int)ebin, (int)(__ffsll((long long)todo) - 1));
			unsigned long long same = __ballot(ebin == e0);
			if (nh_lane() == (unsigned)(__ffsll((long long)same) - 1)) atomicAdd(&hist[e0], (uint32_t)__popcll(same));
			todo &= ~same;
		}
	}
	for (int k = 0; k < 3; ++k) {
		for (int d = 32; d >= 1; d >>= 1) {
			lmin[k] = min(lmin[k], (uint32_t)__shfl_xor((int)lmin[k], d)); lmax[k] = max(lmax[k], (uint32_t)__shfl_xor((int)lmax[k], d));
			if (!STILL) { gmn[k] = min(gmn[k], (uint32_t)__shfl_xor((int)gmn[k], d)); gmx[k] = max(gmx[k], (uint32_t)__shfl_xor((int)gmx[k], d)); }
		}
		if ((threadIdx.x & 63) == 0) { atomicMin(&red[k], lmin[k]); atomicMax(&red[3 + k], lmax[k]); if (!STILL) { atomicMin(&red[6 + k], gmn[k]); atomicMax(&red[9 + k], gmx[k]); } }
	}
	if (STILL) {
		{ const unsigned long long esc = __ballot(escaped); if (esc != 0ull && nh_lane() == 0) { atomicMax(&st->still_failed_seq, seq); atomicAdd(&st->still_diff[3], (uint32_t)__popcll(esc)); } }
		__syncthreads();
		if (threadIdx.x < 3) { atomicMin(&st->still_smin[parity][threadIdx.x], red[threadIdx.x]); atomicMax(&st->still_smax[parity][threadIdx.x], red[3 + threadIdx.x]); }
		return;
	}
	if (__ballot(escaped) != 0ull && nh_lane() == 0) st->fat_escaped = 1u;          // (plain store of a constant)
	__syncthreads();
	if (hist[threadIdx.x]) atomicAdd(&st->exp_hist[threadIdx.x], hist[threadIdx.x]);
	if (threadIdx.x < 3) {
		atomicMin(&st->smin[threadIdx.x], red[threadIdx.x]); atomicMax(&st->smax[threadIdx.x], red[3 + threadIdx.x]);
		if (red[6 + threadIdx.x] <= red[9 + threadIdx.x]) { atomicMin(&st->gmin[threadIdx.x], red[6 + threadIdx.x]); atomicMax(&st->gmax[threadIdx.x], red[9 + threadIdx.x]); }
	}
}

// ---- 2. grid parameters: ONE single-wave launch between the transform pass and the cell keys (Morton frame, small / large split, grid
// origin, cell size and dimensions).  (Folding it into the last workgroup of k_xform to finish was measured: the agent-scope fence that
// needs -- an L2 write-back per workgroup on this multi-die part -- cost 100 us, twenty times the launch it saved.)
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return *p; }

__global__ __launch_bounds__(64) void k_grid_setup(nh_DevState* st, uint32_t ncolliders, uint32_t max_cells, uint32_t direct, uint32_t kept_cap) {
	// small/large split: allow at most max(64, C/1024) colliders above the grid's cell size.  e = largest size class kept small =
	// the highest class whose suffix count (colliders in classes >= e) exceeds the budget.  One wave: lane l owns classes 16l..16l+15.
	const uint32_t budget = max(ncolliders / 1024u, 64u);
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t h[16], mine = 0;
#pragma unroll
	for (int k = 0; k < 16; ++k) { h[k] = ld_agent(&st->exp_hist[16u * lane + k]); mine += h[k]; }
	uint32_t suffix = mine;                                    // inclusive suffix sum over lanes >= l
	for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_down(suffix, d); if (lane + d < 64u) suffix += y; }
	const unsigned long long over = __ballot(suffix > budget);
	int e = 0;
	if (over) {
		const uint32_t top = 63u - (uint32_t)__clzll(over);    // highest lane whose suffix exceeds the budget: the class is in its range
		int mine_e = 0;
		if (lane == top) {
			uint32_t above = suffix - mine;
			mine_e = (int)(16u * lane + 15u);
			while (mine_e > (int)(16u * lane) && above + h[mine_e - (int)(16u * lane)] <= budget) { above += h[mine_e - (int)(16u * lane)]; --mine_e; }
		}
		e = __shfl(mine_e, (int)top);
	}
	if (lane != 0) return;
	nh_f3 smin = nh_make3(nh_float_unflip(ld_agent(&st->smin[0])), nh_float_unflip(ld_agent(&st->smin[1])), nh_float_unflip(ld_agent(&st->smin[2])));
	nh_f3 smax = nh_make3(nh_float_unflip(ld_agent(&st->smax[0])), nh_float_unflip(ld_agent(&st->smax[1])), nh_float_unflip(ld_agent(&st->smax[2])));
	// Morton frame (nudge.cpp:3096-3100)
	float scale = nh_morton_scale(smin, smax);
	st->morton_scale = scale;
	st->morton_min[0] = smin.x * scale; st->morton_min[1] = smin.y * scale; st->morton_min[2] = smin.z * scale;
	// kept pair list: while it is valid and nobody has left its inflated box the grid below is not needed this step -- every kernel of the chain
	// (k_cell_keys .. k_large_pairs) leaves at once and k_kept_filter re-tests the kept pairs; otherwise the candidate boxes k_xform just
	// wrote become the current ones and the chain rebuilds the list from them
	// (re-insertions append to the kept list and leave dead pairs behind: when this step's -- 64 pairs per leaver allowed for -- would fill it beyond seven
	// eighths it is rebuilt instead, which compacts it; and when more than a sixteenth of the world has left, regrouping everybody is the cheaper way)
	const uint32_t leavers = st->esc_count;
	// (... or when the leavers' tests against everybody who has moved since the rebuild -- quadratic -- outgrow the regrouping they stand in for)
	const bool crowded = leavers != 0u && ((uint64_t)st->fat_count + st->fat_count_sph + 64ull * leavers > (uint64_t)kept_cap / 8u * 7u || (uint64_t)leavers * 16u > ncolliders ||
	                                       (uint64_t)leavers * (st->moved_count + st->large) > NH_REINSERT_COMBOS(ncolliders) || (uint64_t)st->moved_count * 32u > ncolliders);
	const bool rebuild = direct || st->fat_valid == 0u || st->fat_escaped != 0u || crowded;
	st->fat_rebuild = rebuild ? 1u : 0u;
	st->fat_reinsert = !rebuild && st->esc_count != 0u ? 1u : 0u;
	if (!rebuild) { st->fat_inserts += st->esc_count; return; }
	st->moved_count = 0u;
	st->fat_margin = direct ? 0.0f : nh_kept_margin(st->cell_inv);           // (of the grid about to be replaced: what k_xform used)
	st->fat_valid = direct ? 0u : 1u; st->fat_count = 0u; st->fat_count_sph = 0u; st->fat_rebuilds += 1u;
	st->large = 0u;             // (counted again by k_cell_keys of this rebuild; steps that re-use the kept list keep the count of the rebuild they live on)
	// e = largest size class (exponent << 2 | two mantissa bits) kept "small"; never let the cell underflow to denormals
	if (e < (64 << 2)) e = 64 << 2;
	if (e > (190 << 2)) e = 190 << 2;
	// k_xform gathered the bounds of the small colliders' min corners under last step's split.  If the split has moved they are the bounds of
	// the wrong subset: this step the grid spans the min corners of ALL colliders instead (a superset: correct, at worst a coarser grid for one
	// step; the next step guesses right again)
	uint32_t gmn[3], gmx[3];
	const bool moved = st->small_exp != (uint32_t)e;
	for (int k = 0; k < 3; ++k) { gmn[k] = ld_agent(moved ? &st->smin[k] : &st->gmin[k]); gmx[k] = ld_agent(moved ? &st->smax[k] : &st->gmax[k]); }
	st->small_exp = (uint32_t)e;
	const float cell = __uint_as_float((uint32_t)(e + 1) << 21);     // upper edge of the class: > every small extent, at most 25 % above the largest
	float inv = 1.0f / cell;
	// grid origin and dimensions; if the table is too small the CELL grows (never clamp one axis: that piles colliders up)
	if (gmn[0] > gmx[0]) {      // no small collider at all
		st->origin[0] = st->origin[1] = st->origin[2] = 0.0f;
		st->dims[0] = st->dims[1] = st->dims[2] = 1;
		st->cell_inv = inv;
		return;
	}
	const nh_f3 gmin = nh_make3(nh_float_unflip(gmn[0]), nh_float_unflip(gmn[1]), nh_float_unflip(gmn[2]));
	const nh_f3 gmax = nh_make3(nh_float_unflip(gmx[0]), nh_float_unflip(gmx[1]), nh_float_unflip(gmx[2]));
	st->origin[0] = gmin.x; st->origin[1] = gmin.y; st->origin[2] = gmin.z;
	// (with fewer colliders than the budget every one of them is "large" and the cell is tiny: the loop then walks down to a single cell)
	st->dims[0] = st->dims[1] = st->dims[2] = 1;
	for (int it = 0; it < 256; ++it) {
		float dx = (gmax.x - gmin.x) * inv, dy = (gmax.y - gmin.y) * inv, dz = (gmax.z - gmin.z) * inv;
		uint64_t nx = (uint64_t)fminf(fmaxf(dx, 0.0f), 1e7f) + 2, ny = (uint64_t)fminf(fmaxf(dy, 0.0f), 1e7f) + 2, nz = (uint64_t)fminf(fmaxf(dz, 0.0f), 1e7f) + 2;
		if (nx * ny * nz < max_cells) { st->dims[0] = (uint32_t)nx; st->dims[1] = (uint32_t)ny; st->dims[2] = (uint32_t)nz; break; }
		inv *= 0.5f;
	}
	st->cell_inv = inv;
}

__device__ __forceinline__ int cell_coord(float x, float origin, float inv) {
	float f = (x - origin) * inv;
	f = fminf(fmaxf(f, -1.0f), 16777216.0f);
	return (int)floorf(f);
}

// ---- 3. cell keys, morton keys, large list; 4. counting sort by cell -----------------------------------------------
// The colliders are grouped by grid cell with a COUNTING sort: k_cell_keys takes a rank inside the cell from an atomic counter,
// the counters are scanned into cell starts, k_cell_scatter writes the AABBs in cell order.  (The order INSIDE a cell is
// whatever the atomics gave; nothing downstream depends on it: a pair is emitted once whichever member comes first, its roles
// are decided by Morton keys, and contacts come out in tag order.)  Large colliders follow the small ones, in list order.
__global__ __launch_bounds__(256) void k_cell_keys(nh_DevState* __restrict__ st, uint32_t n,
                                                   const float4* __restrict__ aabb_min, const float4* __restrict__ aabb_max, float4* __restrict__ box,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ rank, uint32_t* __restrict__ large_list,
                                                   uint32_t* __restrict__ cstart, uint8_t* __restrict__ gen) {
	if (!st->fat_rebuild) return;
	// (the inflated boxes of this rebuild: box[2c] = min, body; box[2c + 1] = max, extent -- written here, read by the rest of the chain and,
	// in the steps that re-use the kept pairs, by k_xform)
	const float margin = st->fat_margin;
	const float inv = st->cell_inv;
	const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
	const int nx = (int)st->dims[0], ny = (int)st->dims[1], nz = (int)st->dims[2];
	const uint32_t small_exp = st->small_exp;
	uint32_t rounded = (n + 63u) & ~63u;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < rounded; c += gridDim.x * blockDim.x) {
		bool valid = c < n;
		bool large = false;
		uint32_t key = NH_LARGE_KEY;
		if (valid) {
			float4 mn = aabb_min[c], mx = aabb_max[c];
			mn.x -= margin; mn.y -= margin; mn.z -= margin; mx.x += margin; mx.y += margin; mx.z += margin;
			mx.w = fmaxf(fmaxf(mx.x - mn.x, mx.y - mn.y), mx.z - mn.z);
			box[2u * c] = mn; box[2u * c + 1u] = mx;
			int cx0 = cell_coord(mn.x, ox, inv), cy0 = cell_coord(mn.y, oy, inv), cz0 = cell_coord(mn.z, oz, inv);
			int cx1 = cell_coord(mx.x, ox, inv), cy1 = cell_coord(mx.y, oy, inv), cz1 = cell_coord(mx.z, oz, inv);
			uint32_t e = (__float_as_uint(mx.w) >> 21) & 1023u;
			// "small" = spans at most 2 cells per axis in THIS arithmetic (makes the neighbour search exact)
			large = e > small_exp || (cx1 - cx0) > 1 || (cy1 - cy0) > 1 || (cz1 - cz0) > 1 || !(mx.w == mx.w);
			if (!large) {
				int cx = min(max(cx0, 0), nx - 1), cy = min(max(cy0, 0), ny - 1), cz = min(max(cz0, 0), nz - 1);
				key = (uint32_t)((cz * ny + cy) * nx + cx);
				rank[c] = atomicAdd(&cstart[key], 1u);
			}
			keys[c] = key;
		}
		uint32_t slot = nh_wave_reserve1(&st->large, valid && large);
		if (valid && large) { large_list[slot] = c; rank[c] = slot; }
		if (valid && gen) gen[c] = large ? 0x80u : 0u;            // generation 0: the grid entry written below is this collider's box (bit 7: not in the grid)
	}
}

// exclusive scan of the per-cell counts (n = table size, a multiple of CS_TILE) into cell starts, in place; cstart[n] = total
#define CS_PER_THREAD 16
#define CS_TILE (256 * CS_PER_THREAD)
__global__ __launch_bounds__(256) void k_cscan_sums(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ block_sums) {
	if (!st->fat_rebuild) return;
	__shared__ uint32_t wsum[4];
	const uint4* p = reinterpret_cast<const uint4*>(cnt + (size_t)blockIdx.x * CS_TILE + threadIdx.x * CS_PER_THREAD);
	uint32_t s = 0;
#pragma unroll
	for (int q = 0; q < CS_PER_THREAD / 4; ++q) { uint4 v = p[q]; s += v.x + v.y + v.z + v.w; }
	for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void k_cscan_final(const nh_DevState* __restrict__ st, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ block_sums, uint32_t n, uint32_t* __restrict__ out) {
	if (!st->fat_rebuild) return;
	__shared__ uint32_t wsum[4];
	__shared__ uint32_t red[4];
	// carry = sum of the block sums before this block
	uint32_t ca = 0;
	for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256) ca += block_sums[i];
	for (int d = 32; d >= 1; d >>= 1) ca += __shfl_xor(ca, d);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ca;
	uint4* p = reinterpret_cast<uint4*>(cnt + (size_t)blockIdx.x * CS_TILE + threadIdx.x * CS_PER_THREAD);
	uint4* po = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * CS_TILE + threadIdx.x * CS_PER_THREAD);       // (the starts live in a table of their own: the grid outlasts the step)
	uint4 v[CS_PER_THREAD / 4];
	uint32_t mine = 0;
#pragma unroll
	// (the counts are cleared as they are consumed: the table is the library's and starts the next rebuild clean -- clearing 4 cells per collider at the start of
	// EVERY step, rebuild or not, was 33 MB of a landed step's 85 MB of k_xform writes at a million boxes)
	for (int q = 0; q < CS_PER_THREAD / 4; ++q) { v[q] = p[q]; p[q] = make_uint4(0u, 0u, 0u, 0u); mine += v[q].x + v[q].y + v[q].z + v[q].w; }
	uint32_t x = mine;
	for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
	if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
	__syncthreads();
	uint32_t run = red[0] + red[1] + red[2] + red[3] + x - mine;
	for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
	for (int q = 0; q < CS_PER_THREAD / 4; ++q) {
		uint4 o;
		o.x = run; run += v[q].x; o.y = run; run += v[q].y; o.z = run; run += v[q].z; o.w = run; run += v[q].w;
		po[q] = o;
	}
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = run;
}

// cell-sorted copy of the AABBs: ONE 32-byte record per collider, sbox[2 pos] = (min.xyz, body), sbox[2 pos + 1] = (max.xyz, collider index),
// so that a candidate of the pair search costs one cache line, not two.  (Morton keys are not carried along: the pair search computes them
// from the min corners of the few pairs that actually overlap.)
__global__ __launch_bounds__(256) void k_cell_scatter(const nh_DevState* __restrict__ st, uint32_t n, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ rank,
                                                      const uint32_t* __restrict__ cstart, const float4* __restrict__ box,
                                                      uint32_t* __restrict__ skeys, float4* __restrict__ sbox) {
	if (!st->fat_rebuild) return;
	const uint32_t nsmall = n - st->large;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const uint32_t key = keys[c];
		const uint32_t pos = key != NH_LARGE_KEY ? cstart[key] + rank[c] : nsmall + rank[c];
		float4 mn = box[2u * c], mx = box[2u * c + 1u];
		mx.w = __uint_as_float(c);                 // sorted copy carries (body, collider index) in the w lanes
		skeys[pos] = key;
		sbox[2u * pos] = mn; sbox[2u * pos + 1u] = mx;
	}
}

__device__ __forceinline__ bool aabb_overlap(const float4& amin, const float4& amax, const float4& bmin, const float4& bmax) { return nh_aabb_overlap(amin, amax, bmin, bmax); }

// Pair emission.  A single global counter takes ~12 ns per atomic (same-address atomics serialise in L2), so
// one atomic per wave per hit-iteration would cost more than the search itself: every wave stages its pairs in
// its own LDS buffer and reserves global space once per ~450 pairs.
// The pair list is split BY SHAPE as it is emitted (the reference partitions its pairs by type before the narrowphase for the same
// reason, nudge.cpp:3705-3751): box-box pairs fill `pairs` from the front (counter st->pairs), pairs with a sphere from the back (counter
// st->pairs_sph, slot pair_cap-1-k), so that each narrowphase launch runs ONE kind of arithmetic in all its lanes.
#ifndef FP_WBUF
#define FP_WBUF 512
#endif
struct pair_stage { uint2* buf; uint32_t count; uint32_t cap; uint32_t nbox; uint32_t mask = 0xFFFFFFFFu; };      // (mask: where a staged pair carries generation stamps above its collider indices)

__device__ __forceinline__ void write_staged(nh_DevState* st, uint2* pairs, uint32_t pair_cap, const pair_stage& ps, uint32_t base_bb, uint32_t base_sph) {
	// lanes take the staged pairs 64 at a time; a pair's slot = its type's base + its rank among the staged pairs of that type
	uint32_t run_bb = 0, run_sph = 0;
	for (uint32_t k0 = 0; k0 < ps.count; k0 += 64) {
		const uint32_t k = k0 + nh_lane();
		const bool on = k < ps.count;
		const uint2 pr = on ? ps.buf[k] : make_uint2(0u, 0u);
		const bool sph = on && ((pr.x & ps.mask) >= ps.nbox || (pr.y & ps.mask) >= ps.nbox);
		const unsigned long long bs = __ballot(sph), bb = __ballot(on && !sph);
		const unsigned long long below = (1ull << nh_lane()) - 1ull;
		if (on) {
			const uint32_t slot = sph ? base_sph + run_sph + (uint32_t)__popcll(bs & below) : base_bb + run_bb + (uint32_t)__popcll(bb & below);
			if (slot < pair_cap) pairs[sph ? pair_cap - 1u - slot : slot] = pr; else st->error = NH_ERR_PAIR_CAPACITY;
		}
		run_sph += (uint32_t)__popcll(bs); run_bb += (uint32_t)__popcll(bb);
	}
}

__device__ __forceinline__ void count_staged(const pair_stage& ps, uint32_t& n_bb, uint32_t& n_sph) {
	uint32_t sph = 0;
	for (uint32_t k = nh_lane(); k < ps.count; k += 64) { const uint2 pr = ps.buf[k]; sph += ((pr.x & ps.mask) >= ps.nbox || (pr.y & ps.mask) >= ps.nbox) ? 1u : 0u; }
	for (int d = 32; d >= 1; d >>= 1) sph += (uint32_t)__shfl_xor((int)sph, d);
	n_sph = sph; n_bb = ps.count - sph;
}

// (`cnt_bb`, `cnt_sph`: the two counters of the list being filled -- st->pairs / st->pairs_sph for this step's pairs, st->fat_count / st->fat_count_sph
// for the kept list)
__device__ __forceinline__ void flush_pairs(nh_DevState* st, uint2* pairs, uint32_t pair_cap, pair_stage& ps, uint32_t* cnt_bb, uint32_t* cnt_sph) {
	if (!ps.count) return;
	__builtin_amdgcn_wave_barrier();
	uint32_t n_bb, n_sph;
	count_staged(ps, n_bb, n_sph);
	uint32_t base = 0;
	if (nh_lane() == 0 && n_bb) base = atomicAdd(cnt_bb, n_bb);
	if (nh_lane() == 1 && n_sph) base = atomicAdd(cnt_sph, n_sph);
	const uint32_t base_bb = __shfl(base, 0), base_sph = __shfl(base, 1);
	write_staged(st, pairs, pair_cap, ps, base_bb, base_sph);
	__builtin_amdgcn_wave_barrier();
	ps.count = 0;
}

// final flush once per workgroup, not once per wave (every wave ends with a partly filled stage)
__device__ __forceinline__ void flush_pairs_block(nh_DevState* st, uint2* pairs, uint32_t pair_cap, pair_stage& ps, uint32_t* cnt_bb, uint32_t* cnt_sph) {
	__shared__ uint32_t wave_bb[8], wave_sph[8];            // (workgroups of up to 8 waves)
	__shared__ uint32_t block_bb, block_sph;
	const uint32_t nwaves = blockDim.x >> 6;
	__builtin_amdgcn_wave_barrier();
	uint32_t n_bb, n_sph;
	count_staged(ps, n_bb, n_sph);
	if (nh_lane() == 0) { wave_bb[threadIdx.x >> 6] = n_bb; wave_sph[threadIdx.x >> 6] = n_sph; }
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < nwaves; ++w) t += wave_bb[w]; block_bb = t ? atomicAdd(cnt_bb, t) : 0u; }
	if (threadIdx.x == 64) { uint32_t t = 0; for (uint32_t w = 0; w < nwaves; ++w) t += wave_sph[w]; block_sph = t ? atomicAdd(cnt_sph, t) : 0u; }
	__syncthreads();
	uint32_t base_bb = block_bb, base_sph = block_sph;
	for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { base_bb += wave_bb[w]; base_sph += wave_sph[w]; }
	write_staged(st, pairs, pair_cap, ps, base_bb, base_sph);
}

__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b);

// `islands`: non-null while some body is asleep -- the coarse islands (nudge.cpp:3575-3650: sets over the broadphase pairs, body 0 ignored) are
// united right where a pair is found
// (an edge between two AWAKE bodies cannot change which sets are active -- a set is active iff it holds an awake body, and a path from a sleeping
// body to an awake one reaches its first awake body over edges that all have a sleeping end -- so only edges with a sleeping end are united:
// while a few bodies of a million-sphere pile doze off, almost every edge is skipped)
__device__ __forceinline__ void emit_pair(nh_DevState* st, uint2* pairs, uint32_t pair_cap, uint64_t ka, uint64_t kb, uint32_t ca, uint32_t cb, bool hit, pair_stage& ps,
                                          uint32_t* islands, const uint8_t* __restrict__ idle, uint32_t body_a, uint32_t body_b) {
	unsigned long long bal = __ballot(hit);
	if (!bal) return;
	if (hit && islands && body_a && body_b && (idle[body_a] == 0xff || idle[body_b] == 0xff)) uf_union(islands, body_a, body_b);
	if (hit) {
		bool a_first = (ka < kb) || (ka == kb && ca < cb);
		// narrowphase "a" = later in Morton order (nudge.cpp:3495, 1202-1203)
		ps.buf[ps.count + (uint32_t)__popcll(bal & ((1ull << nh_lane()) - 1ull))] = a_first ? make_uint2(cb, ca) : make_uint2(ca, cb);
	}
	ps.count += (uint32_t)__popcll(bal);
	if (ps.count > ps.cap - 64) flush_pairs(st, pairs, pair_cap, ps, &st->pairs, &st->pairs_sph);
}

// ---- kept pair list: what the grid search emits -----------------------------------------------------------------------------------
// Pairs of overlapping INFLATED boxes, unordered and unfiltered (same-body pairs included: who belongs to which body is looked at when a pair is
// used, k_kept_filter), staged per wave and split by shape like the pairs above (box-box from the front of the kept buffer, pairs with a sphere from its
// back: a kept pair then has a POSITION that lasts until the next rebuild, see k_kept_filter).
__device__ __forceinline__ void emit_kept(nh_DevState* st, uint2* __restrict__ kept, uint32_t kept_cap, uint32_t ca, uint32_t cb, bool hit, pair_stage& ps) {
	const unsigned long long bal = __ballot(hit);
	if (!bal) return;
	if (hit) ps.buf[ps.count + (uint32_t)__popcll(bal & ((1ull << nh_lane()) - 1ull))] = make_uint2(ca, cb);
	ps.count += (uint32_t)__popcll(bal);
	if (ps.count > ps.cap - 64) flush_pairs(st, kept, kept_cap, ps, &st->fat_count, &st->fat_count_sph);
}

// Two sphere colliders whose boxes overlap but which cannot touch.  In a pit of spheres 40 % of the overlapping boxes belong to spheres that do not
// touch: they would become records that hold no contact, sorted, scanned and gathered with the rest.  The search knows both boxes, and a sphere's box
// IS the sphere (centre = middle, radius = half the edge), up to the rounding of centre -/+ radius: with a margin of ~8 ulp of the coordinates and
// 1e-5 of the radii "surely apart" never drops a pair the narrowphase would give a contact (nudge.cpp:2495: l2 > r * r means none).  Only in the
// DIRECT search (the kept list has to stay a superset across steps), and only between two bodies that are AWAKE: an edge with a sleeping end takes part in
// the coarse islands and may have to be reported as a sleeping pair although its boxes merely overlap (nudge.cpp:3575-3703).
__device__ __forceinline__ bool spheres_surely_apart(const float4& amin, const float4& amax, const float4& bmin, const float4& bmax) {
	const float ax = 0.5f * (amin.x + amax.x), ay = 0.5f * (amin.y + amax.y), az = 0.5f * (amin.z + amax.z);
	const float bx = 0.5f * (bmin.x + bmax.x), by = 0.5f * (bmin.y + bmax.y), bz = 0.5f * (bmin.z + bmax.z);
	const float r = 0.5f * (amax.x - amin.x) + 0.5f * (bmax.x - bmin.x);
	const float dx = bx - ax, dy = by - ay, dz = bz - az;
	const float e = 1e-6f * (fabsf(ax) + fabsf(ay) + fabsf(az) + fabsf(bx) + fabsf(by) + fabsf(bz)) + 1e-5f * r;
	const float s = r + e;
	return dx * dx + dy * dy + dz * dz > s * s;
}

// What the grid search does with two overlapping boxes.  KEPT mode: the boxes are inflated, the pair goes to the kept list as it is.  DIRECT mode (a
// world in which somebody leaves its box every step: keeping pairs would only add the filter pass): the boxes are the exact AABBs, and the pair --
// unless both colliders sit on one body -- goes straight to this step's pairs, roles by Morton order, coarse islands united (emit_pair).
template<bool DIRECT>
__device__ __forceinline__ void emit_found(nh_DevState* st, uint2* __restrict__ out, uint32_t out_cap, const float4& amin, const float4& bmin, uint32_t ca, uint32_t cb, bool hit,
                                           pair_stage& ps, float mscale, const nh_f3& mmin, uint32_t* islands, const uint8_t* __restrict__ idle) {
	if (DIRECT) {
		const uint32_t abody = __float_as_uint(amin.w), bbody = __float_as_uint(bmin.w);
		hit = hit && abody != bbody;
		uint64_t ka = 0, kb = 0;
		if (hit) { ka = nh_morton_of(nh_make3(amin.x, amin.y, amin.z), mscale, mmin); kb = nh_morton_of(nh_make3(bmin.x, bmin.y, bmin.z), mscale, mmin); }
		emit_pair(st, out, out_cap, ka, kb, ca, cb, hit, ps, islands, idle, abody, bbody);
	} else {
		emit_kept(st, out, out_cap, ca, cb, hit, ps);
	}
}

// ---- 5. pair search: grid neighbours + large list ---------------------------------------------------------
// One lane per small collider A, in cell order.  A pair is emitted by its member with the LOWER sorted index, so A only looks FORWARD:
// a collider B that overlaps A has its (min corner) cell in [c_min(A) - 1, c_max(A)] per axis, and one that comes after A in cell order
// lies in A's own row of cells at or after A's cell, in the following rows of A's cell layer, or in the next layer -- at most five rows,
// each ONE contiguous range of the cell-sorted list (cells that are neighbours along x are neighbours in the table).  The ten table
// reads go out together; the candidates are then taken FP_BATCH at a time with all their loads (two 16-byte AABB halves and the
// Morton key) in flight at once: three or four memory round trips per lane instead of one per candidate.
#ifndef FP_BATCH
#define FP_BATCH 2          // (A/B on one box, 1 M boxes: 2 -> 42 us, 4 -> 45 us)
#endif
// SUB: lanes per collider.  A SMALL world (a pile of a thousand bodies) neither fills the machine with one lane per collider nor has short candidate
// lists (its cells are as large as its largest box, its boxes lie on top of each other): SUB consecutive lanes then share a collider and take every
// SUB-th candidate of its ranges -- 16 x the lanes, 1/16 of the dependent rounds (the sample application's pile: 84 -> ~10 us).  Each lane emits its
// own hits; nothing else changes.
template<int SUB, bool DIRECT>
__global__ __launch_bounds__(256) void k_find_pairs(nh_DevState* __restrict__ st, uint32_t n, uint32_t nbox,
                                                    const uint32_t* __restrict__ keys, const float4* __restrict__ sbox,
                                                    const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ large_list,
                                                    const float4* __restrict__ box,
                                                    uint2* __restrict__ kept, uint32_t kept_cap,
                                                    uint32_t* __restrict__ coarse_parent, uint32_t parity, const uint8_t* __restrict__ idle) {
	// (KEPT mode: runs on the inflated boxes and fills the kept list; roles, same-body pairs and sleeping sets are k_kept_filter's business.
	// DIRECT mode: `box` / `sbox` hold the exact AABBs and `kept` is this step's pair list)
	if (!st->fat_rebuild) return;
	uint32_t* const islands = DIRECT && st->any_idle[parity] ? coarse_parent : nullptr;
	const float mscale = st->morton_scale;
	const nh_f3 mmin = nh_make3(st->morton_min[0], st->morton_min[1], st->morton_min[2]);
	const float inv = st->cell_inv;
	const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
	const int nx = (int)st->dims[0], ny = (int)st->dims[1], nz = (int)st->dims[2];
	const uint32_t nlarge = st->large;
	__shared__ uint2 wave_buf[4][FP_WBUF];
	pair_stage ps = { wave_buf[threadIdx.x >> 6], 0u, FP_WBUF, nbox };
	const uint32_t rounded = ((n + 63u) & ~63u) * SUB;
	for (uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x; gi < rounded; gi += gridDim.x * blockDim.x) {
		const uint32_t i = gi / SUB, sub = gi % SUB;
		bool valid = i < n;
		float4 amin = make_float4(0, 0, 0, 0), amax = amin;
		uint32_t key = NH_LARGE_KEY, ca = 0;
		if (valid) { amin = sbox[2u * i]; amax = sbox[2u * i + 1u]; key = keys[i]; ca = __float_as_uint(amax.w); }
		const bool small = valid && key != NH_LARGE_KEY;
		// the forward rows: begin / end of their candidate ranges (empty rows: begin = end = 0)
		uint32_t rb[5] = { 0, 0, 0, 0, 0 }, re[5] = { 0, 0, 0, 0, 0 };
		if (small) {
			const int cx0 = cell_coord(amin.x, ox, inv), cy0 = cell_coord(amin.y, oy, inv), cz0 = cell_coord(amin.z, oz, inv);
			const int cx1 = cell_coord(amax.x, ox, inv), cy1 = cell_coord(amax.y, oy, inv), cz1 = cell_coord(amax.z, oz, inv);
			const int x0 = min(max(cx0 - 1, 0), nx - 1), x1 = min(max(cx1, 0), nx - 1);
			const int y0 = min(max(cy0 - 1, 0), ny - 1), y1 = min(max(cy1, 0), ny - 1);
			const int z1 = min(max(cz1, 0), nz - 1);
			// A's own cell (the clamped one its key was made of)
			const int ay = min(max(cy0, 0), ny - 1), az = min(max(cz0, 0), nz - 1);
			// row 0: A's row, from the entry after A to the end of cell x1
			{ const uint32_t row = (uint32_t)((az * ny + ay) * nx); rb[0] = i + 1u; re[0] = cstart[row + (uint32_t)x1 + 1u]; }
			// row 1: the next row of A's layer
			if (y1 > ay) { const uint32_t row = (uint32_t)((az * ny + ay + 1) * nx); rb[1] = cstart[row + (uint32_t)x0]; re[1] = cstart[row + (uint32_t)x1 + 1u]; }
			// rows 2..4: the next layer, rows y0..y1 (at most three: y0 >= c_min - 1, y1 <= c_min + 1)
			if (z1 > az) {
#pragma unroll
				for (int r = 0; r < 3; ++r) {
					const int cy = y0 + r;
					if (cy <= y1) { const uint32_t row = (uint32_t)(((az + 1) * ny + cy) * nx); rb[2 + r] = cstart[row + (uint32_t)x0]; re[2 + r] = cstart[row + (uint32_t)x1 + 1u]; }
				}
			}
			if (re[0] < rb[0]) re[0] = rb[0];
		}
		// walk the ranges, FP_BATCH candidates per round (wave-uniform loop, lanes without work idle)
		uint32_t r = 0, j = rb[0] + sub, jend = re[0];
		for (;;) {
			uint32_t cand[FP_BATCH];
			bool have[FP_BATCH];
#pragma unroll
			for (int k = 0; k < FP_BATCH; ++k) {
				while (j >= jend && r < 4u) {
					++r;
					j = (r == 1u ? rb[1] : (r == 2u ? rb[2] : (r == 3u ? rb[3] : rb[4]))) + sub;
					jend = r == 1u ? re[1] : (r == 2u ? re[2] : (r == 3u ? re[3] : re[4]));
				}
				have[k] = j < jend;
				cand[k] = have[k] ? j : 0u;
				if (have[k]) j += SUB;
			}
			if (!__any(have[0])) break;
			float4 bmin[FP_BATCH], bmax[FP_BATCH];
#pragma unroll
			for (int k = 0; k < FP_BATCH; ++k) if (have[k]) { bmin[k] = sbox[2u * cand[k]]; bmax[k] = sbox[2u * cand[k] + 1u]; }
#pragma unroll
			for (int k = 0; k < FP_BATCH; ++k) {
				if (!__any(have[k])) break;
				bool hit = false;
				uint32_t cb = 0;
				if (have[k]) {
					cb = __float_as_uint(bmax[k].w);
					hit = aabb_overlap(amin, amax, bmin[k], bmax[k]);
					if (DIRECT && hit && ca >= nbox && cb >= nbox && spheres_surely_apart(amin, amax, bmin[k], bmax[k]))
						hit = islands && (idle[__float_as_uint(amin.w)] == 0xffu || idle[__float_as_uint(bmin[k].w)] == 0xffu);      // (two awake spheres that cannot touch: no pair; `islands` is null while nobody sleeps at all)
				}
				emit_found<DIRECT>(st, kept, kept_cap, amin, bmin[k], ca, cb, hit, ps, mscale, mmin, islands, idle);
			}
		}
		// large x larger-indexed large (small x large is found from the large side: k_large_pairs)
		const bool a_large = valid && key == NH_LARGE_KEY && sub == 0u;
		if (__ballot(a_large)) {
			for (uint32_t l = 0; l < nlarge; ++l) {
				uint32_t cb = large_list[l];
				float4 bmin = box[2u * cb], bmax = box[2u * cb + 1u];
				bool hit = a_large && cb > ca && aabb_overlap(amin, amax, bmin, bmax);
				emit_found<DIRECT>(st, kept, kept_cap, amin, bmin, ca, cb, hit, ps, mscale, mmin, islands, idle);
			}
		}
	}
	flush_pairs_block(st, kept, kept_cap, ps, DIRECT ? &st->pairs : &st->fat_count, DIRECT ? &st->pairs_sph : &st->fat_count_sph);
}

// small x large pairs, found from the LARGE side: a large collider (ground slab, wall) visits the grid cells its AABB
// covers instead of every small collider testing every large one (124 slabs x 1 M boxes = 1.2e8 tests otherwise).
// One reservation of pair slots (an atomic on ONE counter: ~12 ns each, serialised) per workgroup: few, large workgroup shares --
// about LP_BLOCKS workgroups are active in all, so a large collider is cut into LP_BLOCKS / (number of large colliders) slices:
// 8 for the 124 ground slabs of the drop scenes, 200 for the five walls of the ball pit (whose 25,000 cell rows each would
// otherwise be walked one after the other by a handful of waves)
#ifndef LP_THREADS
#define LP_THREADS 512              // threads per workgroup; LP_GRID workgroups are launched (A/B: 512 x 512 -> 21 us, 2048 x 256 -> 25 us at 1 M boxes:
#define LP_GRID 512u                // every workgroup ends with one reservation on the same counter)
#define LP_BLOCKS 512u
#endif
#define LP_WBUF 1024
template<bool DIRECT>
__global__ __launch_bounds__(LP_THREADS) void k_large_pairs(nh_DevState* __restrict__ st, uint32_t nbox, const float4* __restrict__ sbox,
                                                     const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ large_list,
                                                     const float4* __restrict__ box,
                                                     uint2* __restrict__ kept, uint32_t kept_cap,
                                                     uint32_t* __restrict__ coarse_parent, uint32_t parity, const uint8_t* __restrict__ idle) {
	if (!st->fat_rebuild) return;
	uint32_t* const islands = DIRECT && st->any_idle[parity] ? coarse_parent : nullptr;
	const float mscale = st->morton_scale;
	const nh_f3 mmin = nh_make3(st->morton_min[0], st->morton_min[1], st->morton_min[2]);
	const float inv = st->cell_inv;
	const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
	const int nx = (int)st->dims[0], ny = (int)st->dims[1], nz = (int)st->dims[2];
	const uint32_t nlarge = st->large;
	__shared__ uint2 wave_buf[LP_THREADS / 64][LP_WBUF];
	pair_stage ps = { wave_buf[threadIdx.x >> 6], 0u, LP_WBUF, nbox };
	uint32_t slices = nlarge ? LP_BLOCKS / nlarge : 1u;
	slices = min(max(slices, 1u), gridDim.x);
	const uint32_t slice = blockIdx.x % slices;
	for (uint32_t l = blockIdx.x / slices; l < nlarge; l += gridDim.x / slices) {
		const uint32_t ca = large_list[l];
		const float4 amin = box[2u * ca], amax = box[2u * ca + 1u];
		// a small collider B overlapping A has its min-corner cell in [c_min(A) - 1, c_max(A)] per axis
		int x0 = min(max(cell_coord(amin.x, ox, inv) - 1, 0), nx - 1), x1 = min(max(cell_coord(amax.x, ox, inv), 0), nx - 1);
		int y0 = min(max(cell_coord(amin.y, oy, inv) - 1, 0), ny - 1), y1 = min(max(cell_coord(amax.y, oy, inv), 0), ny - 1);
		int z0 = min(max(cell_coord(amin.z, oz, inv) - 1, 0), nz - 1), z1 = min(max(cell_coord(amax.z, oz, inv), 0), nz - 1);
		// an x-row of cells is one contiguous range of the cell-sorted order: a wave takes rows, its lanes take consecutive candidates
		// (coalesced, no lane idling on an empty cell)
		const uint32_t wy = (uint32_t)(y1 - y0 + 1), wz = (uint32_t)(z1 - z0 + 1);
		const uint32_t nrows = wy * wz;
		const uint32_t lane = threadIdx.x & 63u;
		for (uint32_t r = slice * (LP_THREADS / 64u) + (threadIdx.x >> 6); r < nrows; r += slices * (LP_THREADS / 64u)) {       // wave-uniform
			const uint32_t cy = r % wy, cz = r / wy;
			const uint32_t row = (uint32_t)(((z0 + (int)cz) * ny + (y0 + (int)cy)) * nx);
			const uint32_t j0 = cstart[row + (uint32_t)x0], jend = cstart[row + (uint32_t)x1 + 1u];
			for (uint32_t base = j0; base < jend; base += 64u) {
				const uint32_t cand = base + lane;
				const bool have = cand < jend;
				bool hit = false;
				uint32_t cb = 0;
				float4 bmin = make_float4(0, 0, 0, 0);
				if (have) {
					bmin = sbox[2u * cand];
					const float4 bmax = sbox[2u * cand + 1u];
					cb = __float_as_uint(bmax.w);
					hit = aabb_overlap(amin, amax, bmin, bmax);
				}
				emit_found<DIRECT>(st, kept, kept_cap, amin, bmin, ca, cb, hit, ps, mscale, mmin, islands, idle);
			}
		}
	}
	flush_pairs_block(st, kept, kept_cap, ps, DIRECT ? &st->pairs : &st->fat_count, DIRECT ? &st->pairs_sph : &st->fat_count_sph);
}

// ---- 7. re-insertion: the kept list survives a few colliders leaving their boxes ------------------------------------------------------------
// A collider that has left its inflated box used to cost everybody the rebuild (regroup 1 M colliders, search, 117 us) -- one rolling ball in a world at
// rest, every step.  Now k_xform gives a SMALL leaver a new box and the next GENERATION stamp on the spot, and this kernel finds the pairs of the new box:
//   * a kept pair carries the stamps of its two colliders (7 bits each, above the 24-bit indices); k_kept_filter skips pairs whose stamps are not the
//     current ones -- every old pair of the leaver dies without anybody searching for it; dead pairs stay where they are until the next rebuild;
//   * everybody who has NOT moved since the rebuild (stamp 0) still lies in the box the kept grid sorted it by: the leaver visits the cells around its
//     new box (min-corner cell in [c_min - 1, c_max] per axis, as k_large_pairs does) and the large list, skipping stale entries (stamp != 0);
//   * everybody who HAS moved since (the `moved` list, at most NH_MOVED_MAX) is tested directly with its current box; two leavers of one step meet once
//     (the lower index is taken by the higher).
// The result is again a superset of the overlapping pairs with every pair in it once, so k_kept_filter gives exactly the pairs of a full search
// (NH_NO_INCREMENTAL=1 and NH_NO_KEPT_PAIRS=1 must give the same bits: test_kept_pair_list_*).  One wave per leaver, lanes across the candidates.
// Rebuilds remain for: large colliders on the move, more than NH_ESC_MAX leavers in a step (a falling scene), a full moved list, a worn-out stamp,
// a crowded kept buffer, changed collider counts.
// A leaver's OLD pair with the collider its body's contacts come from (still mode only; nh_internal.h "LOCAL speculation").  Re-boxing kills every kept pair of the
// leaver (stamps) and the search appends the pairs of the new box -- a body that hops and comes down on the ground it left would so lose the RECORD of that pair (its
// place in the tag order, its slots, the body's body_rec) and get an appended pair without one: the landing would need a full step.  But the layout knows that pair:
// body_rec[body] is its record, a record's position is its kept pair's (in-place layout).  When the new box still overlaps that partner's box, the old pair is
// re-stamped where it stands instead of a new one being appended: same pair set, and the record lives on -- empty while the body is in the air, with contacts again
// when it lands.
struct nh_OldPair { uint32_t partner, kpos; uint2 kp; bool leaver_is_x; };
__device__ __forceinline__ nh_OldPair nh_old_pair_of(const nh_DevState* __restrict__ st, uint32_t ca, const float4* __restrict__ box, const uint32_t* __restrict__ body_rec,
                                                     const uint2* __restrict__ kept, uint32_t kept_cap, uint32_t pair_cap) {
	nh_OldPair o = { 0xFFFFFFFFu, 0u, make_uint2(0u, 0u), false };
	if (!body_rec || !st->lay_valid || !st->lay_in_place) return o;
	const uint32_t body = __float_as_uint(box[2u * ca].w);
	if (!body) return o;
	const uint32_t br = body_rec[body];
	if (br == NH_BODY_REC_NONE) return o;
	const uint32_t r = br & ~NH_BODY_REC_IS_A, n_bb = min(st->pairs, pair_cap), n_sph = min(st->pairs_sph, pair_cap - n_bb);
	if (r >= n_bb + n_sph) return o;
	o.kpos = r < n_bb ? r : kept_cap - 1u - (r - n_bb);
	o.kp = kept[o.kpos];
	const uint32_t x = o.kp.x & NH_GEN_INDEX, y = o.kp.y & NH_GEN_INDEX;
	if (x == ca) { o.partner = y; o.leaver_is_x = true; } else if (y == ca) { o.partner = x; o.leaver_is_x = false; }
	return o;
}

__global__ __launch_bounds__(256) void k_reinsert(nh_DevState* __restrict__ st, uint32_t nbox, const float4* __restrict__ sbox, const uint32_t* __restrict__ cstart,
                                                  const uint32_t* __restrict__ large_list, const float4* __restrict__ box, const uint8_t* __restrict__ gen,
                                                  const uint32_t* __restrict__ esc_mark, const uint32_t* __restrict__ esc_list, const uint32_t* __restrict__ moved_list, uint32_t seq,
                                                  uint2* __restrict__ kept, uint32_t kept_cap, uint32_t still = 0u, uint32_t parity = 0u, uint32_t ncolliders = 0u,
                                                  const uint32_t* __restrict__ body_rec = nullptr, uint32_t pair_cap = 0u) {
	// `still`: launched by a still step in MOVERS form (nh_internal.h, "LOCAL speculation") behind k_xform<true>, which has re-boxed this step's leavers: there is no
	// k_grid_setup in a still step, so k_grid_setup's verdict is taken here, by every workgroup from words nobody writes meanwhile.  The boxes HAVE changed: when
	// the leavers cannot be re-inserted (too many for the kept buffer or for the quadratic tests) the kept list is void (fat_valid = 0: the replay rebuilds it)
	uint32_t leavers = still ? st->still_esc[parity] : st->esc_count;
	if (still) {
		if (!leavers) return;
		const bool crowded = (uint64_t)st->still_fat0 + 64ull * leavers > (uint64_t)kept_cap / 8u * 7u || (uint64_t)leavers * 16u > ncolliders ||
		                     (uint64_t)leavers * (st->moved_count + st->large) > NH_REINSERT_COMBOS(ncolliders) || (uint64_t)st->moved_count * 32u > ncolliders;
		if (crowded) {
			if (blockIdx.x == 0 && threadIdx.x == 0) { st->fat_valid = 0u; atomicMax(&st->still_failed_seq, seq); atomicAdd(&st->still_diff[3], 1u); }
			return;
		}
		if (blockIdx.x == 0 && threadIdx.x == 0) st->fat_inserts += min(leavers, NH_ESC_MAX);
	} else if (!st->fat_reinsert) return;
	const uint32_t ne = min(leavers, NH_ESC_MAX), nm = min(st->moved_count, NH_MOVED_MAX), nlarge = st->large;
	const float inv = st->cell_inv;
	const float ox = st->origin[0], oy = st->origin[1], oz = st->origin[2];
	const int nx = (int)st->dims[0], ny = (int)st->dims[1], nz = (int)st->dims[2];
	__shared__ uint2 wave_buf[4][FP_WBUF];
	pair_stage ps = { wave_buf[threadIdx.x >> 6], 0u, FP_WBUF, nbox, NH_GEN_INDEX };
	const uint32_t lane = nh_lane();
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	// (a) the kept grid: everybody whose box has not changed since the rebuild.  One wave per leaver; the rows of cells around its box are ranges of the
	// cell-sorted list: the lanes fetch all the range ends at once, then share out the candidates of all rows together -- three dependent memory round trips
	// per leaver whatever the number of rows (one row after the other it was thirty, and ONE leaver cost as much as the rebuild it was meant to save)
	for (uint32_t e = wave; e < ne; e += nwaves) {                // (wave-uniform)
		const uint32_t ca = esc_list[e];
		const float4 amin = box[2u * ca], amax = box[2u * ca + 1u];
		const uint32_t pa = ca | (uint32_t)(gen[ca] & 0x7Fu) << NH_GEN_SHIFT;
		const nh_OldPair old = still ? nh_old_pair_of(st, ca, box, body_rec, kept, kept_cap, pair_cap) : nh_OldPair{ 0xFFFFFFFFu, 0u, make_uint2(0u, 0u), false };      // (wave-uniform)
		bool revived = false;
		const int x0 = min(max(cell_coord(amin.x, ox, inv) - 1, 0), nx - 1), x1 = min(max(cell_coord(amax.x, ox, inv), 0), nx - 1);
		const int y0 = min(max(cell_coord(amin.y, oy, inv) - 1, 0), ny - 1), y1 = min(max(cell_coord(amax.y, oy, inv), 0), ny - 1);
		const int z0 = min(max(cell_coord(amin.z, oz, inv) - 1, 0), nz - 1), z1 = min(max(cell_coord(amax.z, oz, inv), 0), nz - 1);
		const uint32_t wy = (uint32_t)(y1 - y0 + 1), nrows = min(wy * (uint32_t)(z1 - z0 + 1), 64u);        // (a small collider: at most 3 x 3 rows, 4 x 4 in the worst rounding)
		if (wy * (uint32_t)(z1 - z0 + 1) > 64u && lane == 0u) st->error = NH_ERR_INVALID;                     // (k_xform lets only small colliders through: never -- but never silently)
		uint32_t j0 = 0, len = 0;
		if (lane < nrows) {
			const uint32_t row = (uint32_t)(((z0 + (int)(lane / wy)) * ny + (y0 + (int)(lane % wy))) * nx);
			j0 = cstart[row + (uint32_t)x0];
			len = cstart[row + (uint32_t)x1 + 1u] - j0;
		}
		uint32_t incl = len;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += y; }
		const uint32_t total = (uint32_t)__shfl((int)incl, 63);
		for (uint32_t base = 0; base < total; base += 64u) {
			const uint32_t t = base + lane;
			uint32_t cand = 0;
			for (uint32_t r = 0; r < nrows; ++r) {
				const uint32_t end = (uint32_t)__shfl((int)incl, (int)r), l = (uint32_t)__shfl((int)len, (int)r), first = (uint32_t)__shfl((int)j0, (int)r);
				if (t < end && t >= end - l) cand = first + (t - (end - l));
			}
			bool hit = false;
			uint32_t cb = 0;
			if (t < total) {
				const float4 bmin = sbox[2u * cand], bmax = sbox[2u * cand + 1u];
				cb = __float_as_uint(bmax.w);
				hit = aabb_overlap(amin, amax, bmin, bmax) && (gen[cb] & 0x7Fu) == 0u;              // (a stale entry -- the leaver's own among them -- is on the moved list)
				if (hit && cb == old.partner) { hit = false; revived = true; }                      // (the pair the body's record belongs to: re-stamped below, not appended)
			}
			emit_kept(st, kept, kept_cap, pa, cb, hit, ps);
		}
		if (__ballot(revived) != 0ull && lane == 0u) kept[old.kpos] = old.leaver_is_x ? make_uint2(pa, old.kp.y) : make_uint2(old.kp.x, pa);
	}
	// (b) leaver x (large colliders: never re-inserted, their boxes are the rebuild's) and leaver x (everybody who has moved since the rebuild, with the box
	// it has now): one lane per combination, all of them spread over the launch
	const uint32_t others = nlarge + nm;
	const uint64_t combos = (uint64_t)ne * others, rounded = (combos + 63u) & ~63ull;
	for (uint64_t cbase = (uint64_t)wave * 64u; cbase < rounded; cbase += (uint64_t)nwaves * 64u) {
		const uint64_t idx = cbase + lane;
		bool hit = false;
		uint32_t pa = 0, pb = 0;
		if (idx < combos) {
			const uint32_t e = (uint32_t)(idx / others), q = (uint32_t)(idx % others);
			const uint32_t ca = esc_list[e];
			const uint32_t cb = q < nlarge ? large_list[q] : moved_list[q - nlarge];
			// two leavers of one step meet once: the higher index takes the lower (both stand on the moved list)
			const bool skip = cb == ca || (q >= nlarge && esc_mark[cb] == seq && cb > ca);
			if (!skip) hit = aabb_overlap(box[2u * ca], box[2u * ca + 1u], box[2u * cb], box[2u * cb + 1u]);
			pa = ca | (uint32_t)(gen[ca] & 0x7Fu) << NH_GEN_SHIFT;
			pb = cb | (uint32_t)(gen[cb] & 0x7Fu) << NH_GEN_SHIFT;
			if (hit && still) {
				// (the ground a hopping body comes back to is a LARGE collider: this is where its old pair is found again -- see nh_old_pair_of; one lane per combination,
				// so this lane is the only one that can hold it)
				const nh_OldPair old = nh_old_pair_of(st, ca, box, body_rec, kept, kept_cap, pair_cap);
				if (cb == old.partner) {
					hit = false;
					kept[old.kpos] = old.leaver_is_x ? make_uint2(pa, (old.kp.y & NH_GEN_INDEX) | (pb & ~NH_GEN_INDEX)) : make_uint2((old.kp.x & NH_GEN_INDEX) | (pb & ~NH_GEN_INDEX), pa);
				}
			}
		}
		emit_kept(st, kept, kept_cap, pa, pb, hit, ps);
	}
	flush_pairs_block(st, kept, kept_cap, ps, &st->fat_count, &st->fat_count_sph);
}

// ---- 6. kept pairs -> this step's pairs ------------------------------------------------------------------------------------------------
// Temporal coherence.  The grid search above runs on AABBs inflated by 1/32 of a cell and its result is KEPT.  As long as every collider's
// AABB lies inside the inflated box it had then (checked by k_xform every step; a moved, resized or re-parented collider simply leaves its
// box), two AABBs that overlap now had overlapping inflated boxes then: the kept list is a superset of this step's pairs, and this one pass
// -- the EXACT boxes of this step, the strict test of nudge.cpp:3306-3308, bodies and roles as they are now -- produces exactly the pairs the
// full search would (k_grid_setup decides on the device; the kernels of the chain are launched either way and leave at once when they are not
// needed).  A scene at rest re-uses its list for hundreds of steps: at 1 M boxes the regrouping and the search cost 117 us, this pass 15.
// NH_NO_KEPT_PAIRS=1 rebuilds every step; the results must be the same bit for bit (test_kept_pair_list_*).
// Two ways of writing the result.  COMPACT: hits are staged and appended (box-box pairs from the front of `pairs`, pairs with a sphere from the back),
// in whatever order the workgroups reserve their slots.  IN PLACE: a kept pair's position in the kept buffer is also its position in `pairs`, a miss
// leaves an invalid entry (~0, ~0) there -- chosen when the last pass found at least 15 of 16 kept pairs overlapping and nothing was rebuilt since.
// A record (one per pair, k_narrowphase) then sits where it sat last step, and when its key is the same too -- in a landed scene it is, for every pair,
// step after step -- the tag sort of the records has nothing to do (nh_collide: "sort re-use").
#define NH_PAIR_INVALID 0xFFFFFFFFu
__global__ __launch_bounds__(256) void k_kept_filter(nh_DevState* __restrict__ st, const uint2* __restrict__ kept, uint32_t kept_cap, uint32_t nbox,
                                                     const float4* __restrict__ aabb_min, const float4* __restrict__ aabb_max,
                                                     uint2* __restrict__ pairs, uint32_t pair_cap,
                                                     uint32_t* __restrict__ coarse_parent, uint32_t parity, const uint8_t* __restrict__ idle, uint32_t allow_in_place,
                                                     const uint8_t* __restrict__ gen) {
	uint32_t* const islands = st->any_idle[parity] ? coarse_parent : nullptr;
	// generation stamps (k_reinsert): looked at only once somebody has been re-inserted since the rebuild -- until then every stamp is 0
	const uint32_t index_mask = gen ? NH_GEN_INDEX : 0xFFFFFFFFu;
	const bool stamps = gen && st->moved_count != 0u;
	// Morton frame (nudge.cpp:3096-3100): which collider of a pair plays "a" is decided by the Morton order of the AABB min corners
	const float mscale = st->morton_scale;
	const nh_f3 mmin = nh_make3(st->morton_min[0], st->morton_min[1], st->morton_min[2]);
	__shared__ uint2 wave_buf[4][FP_WBUF];
	__shared__ uint32_t s_hits;
	pair_stage ps = { wave_buf[threadIdx.x >> 6], 0u, FP_WBUF, nbox };
	const uint32_t n_bb = min(st->fat_count, kept_cap), n_sph = min(st->fat_count_sph, kept_cap - n_bb);
	const uint32_t n = n_bb + n_sph;
	if (blockIdx.x == 0 && threadIdx.x == 0 && (uint64_t)st->fat_count + st->fat_count_sph > kept_cap) st->error = NH_ERR_PAIR_CAPACITY;
	// (every workgroup derives the same decision from values nobody writes during this kernel)
	const bool in_place = allow_in_place && !st->fat_rebuild && n <= pair_cap && (uint64_t)st->fat_hits_prev * 16u >= (uint64_t)n * 15u;
	if (threadIdx.x == 0) s_hits = 0;
	if (blockIdx.x == 0 && threadIdx.x == 0 && in_place) { st->pairs = n_bb; st->pairs_sph = n_sph; st->lay_in_place = 1u; }
	__syncthreads();
	// One reservation on the pair counters per WORKGROUP and flush (a same-address atomic costs ~12 ns, serialised: with one per wave the counters, not the
	// memory system, set the pace): the waves stage their pairs separately but flush together, so the trip count is uniform over the workgroup
	const uint32_t rounded = (n + blockDim.x - 1u) / blockDim.x * blockDim.x;
	uint32_t hits = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += gridDim.x * blockDim.x) {
		bool hit = false;
		uint32_t ca = 0, cb = 0, abody = 0, bbody = 0;
		uint64_t ka = 0, kb = 0;
		if (i < n) {
			const uint2 pr = i < n_bb ? kept[i] : kept[kept_cap - 1u - (i - n_bb)];
			ca = pr.x & index_mask; cb = pr.y & index_mask;
			const float4 amin = aabb_min[ca], amax = aabb_max[ca], bmin = aabb_min[cb], bmax = aabb_max[cb];
			abody = __float_as_uint(amin.w); bbody = __float_as_uint(bmin.w);
			hit = aabb_overlap(amin, amax, bmin, bmax) && (bbody != abody);
			if (stamps && hit) hit = (uint32_t)(gen[ca] & 0x7Fu) == (pr.x >> NH_GEN_SHIFT) && (uint32_t)(gen[cb] & 0x7Fu) == (pr.y >> NH_GEN_SHIFT);      // (a pair of a box somebody has left since)
			if (hit) { ka = nh_morton_of(nh_make3(amin.x, amin.y, amin.z), mscale, mmin); kb = nh_morton_of(nh_make3(bmin.x, bmin.y, bmin.z), mscale, mmin); }
		}
		hits += hit ? 1u : 0u;
		if (in_place) {
			if (hit && islands && abody && bbody && (idle[abody] == 0xff || idle[bbody] == 0xff)) uf_union(islands, abody, bbody);
			if (i < n) {
				const bool a_first = (ka < kb) || (ka == kb && ca < cb);
				const uint2 out = hit ? (a_first ? make_uint2(cb, ca) : make_uint2(ca, cb)) : make_uint2(NH_PAIR_INVALID, NH_PAIR_INVALID);       // narrowphase "a" = later in Morton order (emit_pair)
				pairs[i < n_bb ? i : pair_cap - 1u - (i - n_bb)] = out;
			}
		} else {
			emit_pair(st, pairs, pair_cap, ka, kb, ca, cb, hit, ps, islands, idle, abody, bbody);
			if (__syncthreads_or(ps.count > ps.cap - 192u)) { flush_pairs_block(st, pairs, pair_cap, ps, &st->pairs, &st->pairs_sph); ps.count = 0; __syncthreads(); }
		}
	}
	if (!in_place) flush_pairs_block(st, pairs, pair_cap, ps, &st->pairs, &st->pairs_sph);
	// how many kept pairs overlap: next step's choice between the two ways
	for (int d = 32; d >= 1; d >>= 1) hits += (uint32_t)__shfl_xor((int)hits, d);
	if (nh_lane() == 0 && hits) atomicAdd(&s_hits, hits);
	__syncthreads();
	if (threadIdx.x == 0 && s_hits) atomicAdd(&st->fat_hits, s_hits);
}

// ---- 6. narrowphase -----------------------------------------------------------------------------------------
// raw contacts are appended in arbitrary order; one record per pair that produced contacts.
// record i belongs to pair i: its first 4 contacts sit at raw[4*i .. 4*i+3], further ones in the overflow area
// (struct nh_Record, NH_REC_SLEEPING: nh_internal.h -- the records are library-owned and outlive the step)

// SPH = false: the box-box pairs (front of `pairs`, records 0 .. n_bb-1); SPH = true: the pairs with a sphere (back of `pairs`, records
// n_bb .. n_bb+n_sph-1).  Record r owns the raw contact slots 4r .. 4r+3 either way (a sphere pair uses the first one).
// STILL (nh_collide, "9. still steps"): the pairs are taken straight from the KEPT list (what k_kept_filter does in place: exact AABB test, bodies,
// stamps, roles by Morton order -- a kept pair's position is its record's) and the contacts go to the record's own raw slots like in a full step: a pair that
// gains or loses a contact moves nobody else's data.  What a still step RELIES on is checked -- the record's key is what it was (tag order unchanged), at most
// four contacts, a pair with contacts joins a dynamic body to the static world and is the one pair that body's solver lane knows (body_rec) -- and raises
// st->still_fail otherwise: the step is replayed in full.  The record's count goes to its place in the tag order (cnt_sorted): one scan later every record
// knows where its contacts sit in the dense list.
// LOCAL (STILL only; nh_internal.h, "LOCAL speculation"): the form that knows appended pairs (movers) and sleeping bodies (sleepers).  A world in which nobody has left
// its box lately and nobody can be asleep -- the bench's steady state -- runs the LOCAL = false instantiation: the loop of round 4 (rocprofv3: 91 us at 1 M boxes; the
// one form with everything in it, measured first: 100.6)
template<bool SPH, bool STILL, bool LOCAL = false>
__global__ __launch_bounds__(256) void k_narrowphase(nh_DevState* __restrict__ st, const uint2* __restrict__ pairs,
                                                     const nh_xform* __restrict__ xf, const uint32_t* __restrict__ ctag,
                                                     const nh_BoxCollider* __restrict__ box_data, const nh_SphereCollider* __restrict__ sph_data, uint32_t nbox,
                                                     nh_Contact* __restrict__ raw_data, uint32_t* __restrict__ raw_feature, uint32_t pair_cap, uint32_t overflow_cap,
                                                     uint64_t* __restrict__ rec_key, nh_Record* __restrict__ rec, uint32_t* __restrict__ rec_idx,
                                                     const uint32_t* __restrict__ coarse_root, const uint32_t* __restrict__ coarse_active, uint32_t parity,
                                                     // contact ownership of a partitioned world (nh_set_first_ghost_body): bodies >= first_ghost are ghosts; 0 = off
                                                     uint32_t first_ghost,
                                                     // STILL only: the kept list (`pairs`) and its capacity, this step's exact AABBs, the stamps, tag-order positions / counts, the bodies' records
                                                     uint32_t kept_cap = 0u, const float4* __restrict__ aabb_min = nullptr, const float4* __restrict__ aabb_max = nullptr,
                                                     const uint8_t* __restrict__ gen = nullptr, const uint32_t* __restrict__ lay_rank = nullptr, uint32_t* __restrict__ cnt_sorted = nullptr,
                                                     const uint32_t* __restrict__ body_rec = nullptr, uint32_t seq = 0u, int2* __restrict__ delta = nullptr, uint32_t delta_scan = 0u,
                                                     // STILL, sleepers form (nh_internal.h): the slot cache's per-record counts (entries kept aside for a sleeping pair); nullptr = nobody can be asleep
                                                     uint32_t* __restrict__ sc_count = nullptr,
                                                     // STILL, plain form: no k_xform<true> has run this step -- the last step's solver did its work (nh_internal.h: xform ahead)
                                                     uint32_t ahead = 0u,
                                                     // STILL, sleepers form: where a dropped slot-cache count is noted until the step's verdict is in (nh_internal.h: sc_undo)
                                                     uint64_t* __restrict__ sc_undo = nullptr,
                                                     // STILL, sleepers form: a record that WAS a sleeping pair when the step before wrote it and whose two ends are the static world or
                                                     // asleep now is what it was -- counted, nothing else fetched (30 bytes per pair instead of 230)
                                                     uint32_t quick_sleepers = 0u) {
	if (STILL && !LOCAL && ahead) {
		// what k_xform<true> does for the step after this one (nobody touches the other parity's words during this step: the solver that gathers into them comes later)
		if (blockIdx.x == 0 && threadIdx.x == 0) {
			st->max_idle[parity ^ 1u] = 0u; st->delta_count[parity ^ 1u] = 0u; st->delta_overflow[parity ^ 1u] = 0u; st->still_esc[parity ^ 1u] = 0u;
			st->still_asleep[parity ^ 1u] = 0u; st->still_sleeping[parity ^ 1u] = 0u; st->still_culled[parity ^ 1u] = 0u;
			for (int k = 0; k < 3; ++k) { st->still_smin[parity ^ 1u][k] = 0xffffffffu; st->still_smax[parity ^ 1u][k] = 0u; }
			st->still_fat0 = st->fat_count + st->fat_count_sph;
		}
		if (blockIdx.x == 0) for (uint32_t k = threadIdx.x; k < NH_AHEAD_PARTS * 8u; k += blockDim.x) (&st->ahead_part[parity ^ 1u][0][0])[k] = (k & 7u) < 3u ? 0xffffffffu : 0u;
		// ... and what it would have found: a collider outside its inflated box, a body asleep (the solver lane that advanced the body saw it), a world the lanes cannot
		// stand in for k_xform in -- this step has not happened (every workgroup reads the same words and comes to the same verdict)
		if (st->ahead_failed_seq >= seq || st->ahead_multi != 0u) {
			if (blockIdx.x == 0 && threadIdx.x == 0) { atomicMax(&st->still_failed_seq, seq); atomicAdd(&st->still_diff[3], 1u); }
			return;
		}
	}
	// coarse islands (nudge.cpp:3500-3703): a pair whose bodies belong to a set of sleeping bodies (sets over the BROADPHASE pairs and the
	// user's connections) never reaches the narrowphase: it becomes a sleeping pair, larger tag in the low half (3697)
	const bool some_asleep = !STILL && st->any_idle[parity] != 0;
	constexpr int NS = SPH ? 1 : 4;
	__shared__ float4 s_c0[NS][256], s_c1[NS][256];
	__shared__ uint32_t s_f[NS][256];
	__shared__ uint8_t s_n[NS][64];
	__shared__ uint32_t s_pre[STILL ? 3 : 1][STILL ? 256 : 1];          // STILL: the record's old key and tag-order position, fetched with everything else
	const uint32_t n_bb = min(st->pairs, pair_cap), n_sph = min(st->pairs_sph, pair_cap - n_bb);
	if (!STILL && blockIdx.x == 0 && threadIdx.x == 0) {
		st->records = n_bb + n_sph;            // (both launches write the same number: either may be the only one)
		if ((uint64_t)st->pairs + st->pairs_sph > pair_cap) st->error = NH_ERR_PAIR_CAPACITY;
	}
	// STILL: Morton frame of this step (nudge.cpp:3096-3100; k_grid_setup's arithmetic) from the bounds k_xform<true> gathered; stamps as in k_kept_filter
	float mscale = 0.0f;
	nh_f3 mmin = nh_make3(0.0f, 0.0f, 0.0f);
	uint32_t fr_min[3] = { 0u, 0u, 0u }, fr_max[3] = { 0u, 0u, 0u }, fr_top = 0u;
	uint32_t i_first = 0u;
	uint2 kp_first = make_uint2(0u, 0u);
	auto still_frame = [&]() {
		if (!LOCAL && ahead) {
			for (int k = 0; k < 3; ++k)
				for (int d = 32; d >= 1; d >>= 1) { fr_min[k] = min(fr_min[k], (uint32_t)__shfl_xor((int)fr_min[k], d)); fr_max[k] = max(fr_max[k], (uint32_t)__shfl_xor((int)fr_max[k], d)); }
			for (int d = 32; d >= 1; d >>= 1) fr_top = max(fr_top, (uint32_t)__shfl_xor((int)fr_top, d));
			// (the first wave files the result where a step with a k_xform<true> of its own would have it: the host's mirror, the sleep prediction)
			if (blockIdx.x == 0 && threadIdx.x == 0) {
				for (int k = 0; k < 3; ++k) { st->still_smin[parity][k] = fr_min[k]; st->still_smax[parity][k] = fr_max[k]; }
				st->max_idle[parity] = fr_top;
			}
		}
		const nh_f3 smin = nh_make3(nh_float_unflip(fr_min[0]), nh_float_unflip(fr_min[1]), nh_float_unflip(fr_min[2]));
		const nh_f3 smax = nh_make3(nh_float_unflip(fr_max[0]), nh_float_unflip(fr_max[1]), nh_float_unflip(fr_max[2]));
		mscale = nh_morton_scale(smin, smax);
		mmin = nh_make3(smin.x * mscale, smin.y * mscale, smin.z * mscale);
	};
	const uint32_t index_mask = (STILL && gen) ? NH_GEN_INDEX : 0xFFFFFFFFu;
	const bool stamps = STILL && gen && st->moved_count != 0u;
	if (STILL) {
		for (int k = 0; k < 3; ++k) { fr_min[k] = st->still_smin[parity][k]; fr_max[k] = st->still_smax[parity][k]; }
		if (!LOCAL && ahead) {
			// the bounds (and the largest idle counter) the last step's solver lanes gathered in NH_AHEAD_PARTS places, the static world's share beside them: asked for
			// here, put together by every wave for itself (still_frame) once its lanes' first kept pairs are on their way as well
			const uint32_t* part = &st->ahead_part[parity][nh_lane() & (NH_AHEAD_PARTS - 1u)][0];
			fr_top = part[6];
			for (int k = 0; k < 3; ++k) { fr_min[k] = min(part[k], st->still_static_min[k]); fr_max[k] = max(part[3 + k], st->still_static_max[k]); }
		}
		// (every lane's FIRST kept pair is asked for before the frame is put together: the wave waits once for both)
		i_first = blockIdx.x * blockDim.x + threadIdx.x;
#ifndef NH_NO_PAIR_PREFETCH
		{ const uint32_t ic = min(i_first, kept_cap - 1u); kp_first = pairs[SPH ? kept_cap - 1u - ic : ic]; }
#else
		i_first = 0xFFFFFFFFu;
#endif
		still_frame();
		// (the layout must be the last full step's, and the kept list at least as long as the pair list it was written from: what lies behind are pairs APPENDED by
		// still steps in movers form since -- below)
		// (LOCAL = false: the host launches this form only when no still step in movers form has appended a pair since the layout was written -- checked all the same)
		if (blockIdx.x == 0 && threadIdx.x == 0 && (!st->lay_valid || (LOCAL ? (st->fat_count < st->pairs || st->fat_count_sph < st->pairs_sph) : (st->fat_count != st->pairs || st->fat_count_sph != st->pairs_sph))))
			atomicMax(&st->still_failed_seq, seq);
	}
	const uint32_t n = SPH ? n_sph : n_bb, first_rec = SPH ? n_bb : 0u;
	// STILL, LOCAL speculation (nh_internal.h): kept pairs behind the layout's -- appended by k_reinsert for colliders that left their boxes in a still step -- have no
	// record, no raw slots and no place in the tag order.  They are evaluated like the others (exact boxes, stamps, roles, contact arithmetic); as long as none of them
	// yields a CONTACT they are what a full step would file as records without contacts, which nothing downstream looks at; the first contact fails the step, and the
	// full step that follows gives every kept pair its record again.
	uint32_t n_app = 0;
	if (STILL && LOCAL) {
		const uint32_t k_bb = min(st->fat_count, kept_cap), k_sph = min(st->fat_count_sph, kept_cap - k_bb);
		const uint32_t kc = SPH ? k_sph : k_bb;
		n_app = kc > n ? kc - n : 0u;
	}
	// sort re-use (nh_collide): `rec_key` is the context's buffer of keys by record position, still holding last step's -- is every key what it was?
	bool changed = false;
	// STILL: what the step relies on, per record (see above); the count goes to the record's place in the tag order
	uint32_t n_sleeping = 0, n_culled = 0;          // sleepers form: this lane's sleeping pairs and the cache entries kept aside for them
	auto still_record = [&](uint32_t ri, uint64_t key, int count, uint32_t body_a, uint32_t body_b, uint64_t was_key, uint32_t pos, bool vanished, bool sleeping = false) {
		if (LOCAL && sleeping) {
			// A pair of a sleeping body with the static world: a sleeping pair (nudge.cpp:3669-3703) -- no contact, the pair's word (`key`: larger tag low, 3697) in the
			// list of sleeping pairs, and its cached impulses kept aside (4064-4101) when the cache's tag -- the contact key on file -- IS that word; otherwise they are
			// gone, as in the reference.  From now on the record carries the pair's word (the tag order on file no longer matches the keys: a full step sorts again).
			vanished = true; count = 0;
			// (the drop happens before the step's verdict is known: the old count is noted under this step's number and comes back if the step did not happen --
			// nh_still_undo_drops.  The key on file needs no such care: a replay compares every key with this buffer and sorts again where they differ, and sort_valid = 0 asks for that anyway)
			if (was_key != key) { const uint32_t oc = sc_count[ri]; if (oc) { sc_undo[ri] = ((uint64_t)seq << 32) | oc; sc_count[ri] = 0u; } rec_key[ri] = key; st->sort_valid = 0u; }
			else n_culled += min(sc_count[ri], 4u);
			n_sleeping += 1u;
		}
		// (`vanished`: the kept pair is dead -- a collider of it has been re-boxed since, stamps -- or its exact boxes do not overlap this step: a record without
		// contacts wherever the full step's sort would put it, so its old place serves; the key it had stays on file, and if the pair overlaps again with that key
		// it is the same record in the same place)
		const bool same_key = was_key == key || vanished;
		bool ok = same_key && count <= 4;
		if (ok && count) {
			const uint32_t dyn = body_a ? body_a : body_b;
			ok = (body_a == 0u) != (body_b == 0u) && (body_rec[dyn] & ~NH_BODY_REC_IS_A) == ri;
		}
		if (!ok) { atomicAdd(&st->still_diff[!same_key ? 0 : (count > 4 ? 1 : 2)], 1u); changed = true; }          // (statistics; rare)
		rec[ri].count = (LOCAL && sleeping) ? NH_REC_SLEEPING : (uint32_t)count;
		// the record's count in its place of the tag order; a CHANGE is listed (position, difference): the solver lanes shift their first contact by the changes before
		// them -- a handful per step in a world at rest -- instead of the whole order being scanned again.  Too many for the list: the step fails (delta_overflow)
		const uint32_t old = cnt_sorted[pos];
		if (old != (uint32_t)count) {
			cnt_sorted[pos] = (uint32_t)count;
			if (!st->delta_overflow[parity]) {
				const uint32_t k = atomicAdd(&st->delta_count[parity], 1u);
				if (k < NH_DELTA_MAX) delta[parity * NH_DELTA_MAX + k] = make_int2((int)pos, count - (int)old);
				else { st->delta_overflow[parity] = 1u; if (!delta_scan) atomicMax(&st->still_failed_seq, seq); }          // (more changes than the list holds: the scan runs -- large worlds, where the host has launched it -- or this is not a still step)
			}
		}
	};
	uint32_t rounded = (n + n_app + 63u) & ~63u;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += gridDim.x * blockDim.x) {
		const bool live = i < n, app = STILL && LOCAL && !live && i < n + n_app;
		nh_bb_result bb;
		bb.kind = 0; bb.mask = 0;
		nh_contact_out single;          // sphere paths produce at most one contact
		int count = 0;
		uint64_t key = 0;
		uint32_t body_a = 0, body_b = 0;
		bool asleep = false;
		const uint32_t ri = first_rec + i;        // record index
		bool invalid = false, sleeper_pair = false;
		// STILL: everything a pair needs goes out in ONE round trip behind the pair itself -- both boxes, both world transforms, shapes and tags, the record's old key and
		// tag-order position -- and which of the two is "a" is a register select afterwards (fetching the transforms only once the roles were known, and the key and the
		// position only when the contacts were done, made four dependent round trips of it)
		nh_xform XA = nh_xform(), XB = nh_xform();
		uint32_t TA = 0, TB = 0, CB = 0;
		nh_BoxCollider SA = nh_BoxCollider(), SB = nh_BoxCollider();
		float RA = 0.0f, RB = 0.0f;
		auto XF = [&](uint32_t c) -> nh_xform { if (STILL) return c == CB ? XB : XA; return xf[c]; };
		auto TAG = [&](uint32_t c) -> uint32_t { if (STILL) return c == CB ? TB : TA; return ctag[c]; };
		auto BOX = [&](uint32_t c) -> nh_BoxCollider { if (STILL) return c == CB ? SB : SA; return box_data[c]; };
		auto RAD = [&](uint32_t c) -> float { if (STILL) return c == CB ? RB : RA; return sph_data[c - nbox].radius; };
		bool quick = false;
		if (STILL && LOCAL && quick_sleepers && sc_count && live) {
			uint2 kp = kp_first;
			if (i != i_first) kp = pairs[SPH ? kept_cap - 1u - i : i];
			const uint32_t ca = kp.x & index_mask, cb = kp.y & index_mask;
			const uint32_t* const mx = reinterpret_cast<const uint32_t*>(aabb_max); const uint32_t* const mn = reinterpret_cast<const uint32_t*>(aabb_min);
			const uint32_t ma = mx[4u * (size_t)ca + 3u], mb = mx[4u * (size_t)cb + 3u], ba = mn[4u * (size_t)ca + 3u], bb2 = mn[4u * (size_t)cb + 3u];
			const uint32_t rc = reinterpret_cast<const uint32_t*>(rec)[4u * (size_t)ri + 3u];
			const uint32_t oc = sc_count[ri];
			// (both ends the static world or marked asleep -- one of them marked: the other end of a sleeping record is the static world)
			quick = rc == NH_REC_SLEEPING && (ma | mb) == 1u && (ba == 0u || ma == 1u) && (bb2 == 0u || mb == 1u);
			if (quick) { n_sleeping += 1u; n_culled += min(oc, 4u); }
		}
		if ((live && !quick) || app) {
			uint2 pr;
			if (STILL) {
				uint2 kp = kp_first;
				if (i != i_first) kp = pairs[SPH ? kept_cap - 1u - i : i];
				const uint32_t ca = kp.x & index_mask, cb = kp.y & index_mask;
				const float4 amin = aabb_min[ca], amax = aabb_max[ca], bmin = aabb_min[cb], bmax = aabb_max[cb];
				XA = xf[ca]; XB = xf[cb]; TA = ctag[ca]; TB = ctag[cb]; CB = cb;
				SA = box_data[ca < nbox ? ca : 0u]; SB = box_data[cb < nbox ? cb : 0u];
				if (SPH) { RA = sph_data[ca >= nbox ? ca - nbox : 0u].radius; RB = sph_data[cb >= nbox ? cb - nbox : 0u].radius; }
				// (parked in LDS across the contact computation: three more live registers there are three more spilled ones)
				if (live) { const uint64_t pk = rec_key[ri]; s_pre[0][threadIdx.x] = (uint32_t)pk; s_pre[1][threadIdx.x] = (uint32_t)(pk >> 32); s_pre[2][threadIdx.x] = lay_rank[ri]; }
				bool hit = aabb_overlap(amin, amax, bmin, bmax) && (__float_as_uint(bmin.w) != __float_as_uint(amin.w));
				if (stamps && hit) hit = (uint32_t)(gen[ca] & 0x7Fu) == (kp.x >> NH_GEN_SHIFT) && (uint32_t)(gen[cb] & 0x7Fu) == (kp.y >> NH_GEN_SHIFT);
				// sleepers form: k_xform<true> marked the colliders of sleeping bodies (aabb_max.w).  With the static world: a sleeping pair.  Two dynamic bodies with a
				// sleeping end, or a pair nobody has a record for: the body's set holds more than itself -- not this step's business
				if (LOCAL && sc_count && hit && (__float_as_uint(amax.w) | __float_as_uint(bmax.w)) != 0u) {
					if (app || (__float_as_uint(amin.w) != 0u && __float_as_uint(bmin.w) != 0u)) { changed = true; atomicAdd(&st->still_diff[3], 1u); }
					else sleeper_pair = true;
				}
				const uint64_t ka = nh_morton_of(nh_make3(amin.x, amin.y, amin.z), mscale, mmin), kb = nh_morton_of(nh_make3(bmin.x, bmin.y, bmin.z), mscale, mmin);
				const bool a_first = (ka < kb) || (ka == kb && ca < cb);
				pr = hit ? (a_first ? make_uint2(cb, ca) : make_uint2(ca, cb)) : make_uint2(NH_PAIR_INVALID, NH_PAIR_INVALID);       // narrowphase "a" = later in Morton order (emit_pair)
			} else {
				pr = pairs[SPH ? pair_cap - 1u - i : i];
			}
			invalid = pr.x == NH_PAIR_INVALID;       // (k_kept_filter, in-place form: a kept pair that does not overlap this step -- an empty record, key ~0)
			if (invalid) { pr = make_uint2(0u, 0u); key = ~0ull; }
			uint32_t a = pr.x, b = pr.y;            // a = later in Morton order
			bool a_sph = a >= nbox, b_sph = b >= nbox;
			bool coarse_asleep = invalid || sleeper_pair;
			if (STILL && LOCAL && sleeper_pair) { const uint32_t ta = TAG(a), tb = TAG(b); key = ta > tb ? ((uint64_t)ta | ((uint64_t)tb << 32)) : ((uint64_t)tb | ((uint64_t)ta << 32)); }
			if (some_asleep && !invalid) {
				const uint32_t ba = xf[a].body, bb2 = xf[b].body;
				const uint32_t set = coarse_root[ba ? ba : bb2];            // sets[a] | sets[b] with sets[0] = 0 (nudge.cpp:3687)
				coarse_asleep = !coarse_active[set];
				if (coarse_asleep) {
					const uint32_t ta = ctag[a], tb = ctag[b];
					key = ta > tb ? ((uint64_t)ta | ((uint64_t)tb << 32)) : ((uint64_t)tb | ((uint64_t)ta << 32));
					body_a = ba; body_b = bb2;
					asleep = true;
				}
			}
			if (coarse_asleep) {
			} else if (!SPH) {
				nh_xform A = XF(a), B = XF(b);
				nh_BoxCollider sa = BOX(a), sb = BOX(b);
				uint32_t ta = TAG(a), tb = TAG(b);
				nh_box_box_eval(A, B, sa.size, sb.size, ta, tb, bb);
				count = nh_bb_count(bb);
				key = (uint64_t)(bb.swapped ? tb : ta) | ((uint64_t)(bb.swapped ? ta : tb) << 32);
				body_a = bb.swapped ? B.body : A.body;
				body_b = bb.swapped ? A.body : B.body;
			} else if (a_sph && b_sph) {
				// sphere-sphere: first = earlier in Morton order = pair.y (nudge.cpp:3775-3776)
				nh_xform A = XF(b), B = XF(a);
				count = nh_sphere_sphere(RAD(b), RAD(a), A, B, &single);
				key = (uint64_t)TAG(b) | ((uint64_t)TAG(a) << 32);
				body_a = A.body; body_b = B.body;
			} else {
				// box-sphere: box is always "a" (nudge.cpp:3746-3751, 3759-3768)
				uint32_t bx = a_sph ? b : a, sp = a_sph ? a : b;
				nh_xform A = XF(bx), B = XF(sp);
				nh_BoxCollider sz = BOX(bx);
				count = nh_box_sphere(sz.size, RAD(sp), A, B, &single);
				key = (uint64_t)TAG(bx) | ((uint64_t)TAG(sp) << 32);
				body_a = A.body; body_b = B.body;
			}
			// one owner per contact in a partitioned world ("contact owned by the rank owning the body with the larger collider tag", the a > b
			// canonicalisation of nudge.cpp:2074-2087): a pair with the static world belongs to its dynamic body, a pair of two dynamic bodies to the body of
			// the collider with the larger tag; a pair whose owner is a ghost here is solved by the rank that owns that body and makes no contact on this one
			if (!STILL && first_ghost && count) {
				const uint32_t ba = xf[a].body, bb2 = xf[b].body;
				const uint32_t owner = ba == 0u ? bb2 : (bb2 == 0u ? ba : (ctag[a] > ctag[b] ? ba : bb2));
				if (owner >= first_ghost) { count = 0; bb.kind = 0; }
			}
			// (an appended pair -- no record -- may overlap, but a contact of it belongs to a layout only a full step can make)
			if (LOCAL && app) { if (count) { changed = true; atomicAdd(&st->still_diff[1], 1u); } count = 0; bb.kind = 0; bb.mask = 0; }
		}
		if (SPH) {
			// one contact at most, ONE raw slot per sphere record (they follow the four-slot blocks of the box-box records: slot 4 n_bb + i, so that
			// consecutive lanes write -- and k_gather_contacts reads -- consecutive 32 bytes instead of every fourth slot)
			if (live) {
				if (count) {
					const size_t slot = (size_t)4 * n_bb + i;
					float4* dp = reinterpret_cast<float4*>(raw_data + slot);
					dp[0] = make_float4(single.px, single.py, single.pz, single.penetration);
					dp[1] = make_float4(single.nx, single.ny, single.nz, single.friction);
					raw_feature[slot] = single.feature;
				}
				if (STILL) { if (!quick) still_record(ri, key, count, body_a, body_b, (uint64_t)s_pre[0][threadIdx.x] | ((uint64_t)s_pre[1][threadIdx.x] << 32), s_pre[2][threadIdx.x], invalid, sleeper_pair); }
				else {
				{ const uint64_t was = rec_key[ri]; changed |= was != key; rec_key[ri] = key; }
				nh_Record r = { body_a, body_b, 0u, asleep ? NH_REC_SLEEPING : (uint32_t)count };
				if (asleep) st->any_sleeping = 1;
				rec[ri] = r;
				rec_idx[ri] = ri;
				}
			}
			continue;
		}
		// more than 4 contacts per pair is the exception (askew face contacts): only those touch the global counter
		uint32_t extra = (!STILL && count > 4) ? (uint32_t)(count - 4) : 0u;
		uint32_t ovf = 0;
		if (!STILL && __ballot(extra != 0)) ovf = nh_wave_reserve(&st->raw_contacts, extra);
		// The first four contacts of a pair go to its four raw slots: the wave's 64 pairs own ONE contiguous run of 256 slots, so the
		// lanes post their contacts to LDS and the wave writes the run with consecutive lanes on consecutive slots (a lane writing
		// its own 128 bytes would put 64 different lines behind every store instruction).  Contacts 5+ (rare) go out directly.
		float4* w_c0 = s_c0[threadIdx.x >> 6];
		float4* w_c1 = s_c1[threadIdx.x >> 6];
		uint32_t* w_f = s_f[threadIdx.x >> 6];
		uint8_t* w_n = s_n[threadIdx.x >> 6];
		const uint32_t lane = nh_lane();
		if (live) {
			if (extra && ovf + extra > overflow_cap) { st->error = NH_ERR_CONTACT_CAPACITY; count = 4; }
			auto put = [&](int k, const nh_contact_out& o) {
				if (STILL && k >= 4) return;
				if (k < 4) {
					w_c0[4u * lane + k] = make_float4(o.px, o.py, o.pz, o.penetration);
					w_c1[4u * lane + k] = make_float4(o.nx, o.ny, o.nz, o.friction);
					w_f[4u * lane + k] = o.feature;
				} else {
					size_t dst = (size_t)4 * pair_cap + ovf + (k - 4);
					float4* dp = reinterpret_cast<float4*>(raw_data + dst);
					dp[0] = make_float4(o.px, o.py, o.pz, o.penetration);
					dp[1] = make_float4(o.nx, o.ny, o.nz, o.friction);
					raw_feature[dst] = o.feature;
				}
			};
			if (bb.kind == 2) {
				int k = 0;
#pragma unroll
				for (int index = 0; index < 16; ++index) {
					if (((bb.mask >> index) & 1u) && k < count) {
						nh_contact_out o;
						nh_bb_contact(bb, index, bb.lx[index], bb.ly[index], bb.lz[index], bb.penetration[index], bb.tags[index], o);
						put(k, o);
						++k;
					}
				}
			} else if (bb.kind == 1) {
				put(0, bb.edge);
			}
			if (STILL) { if (!quick) still_record(ri, key, count, body_a, body_b, (uint64_t)s_pre[0][threadIdx.x] | ((uint64_t)s_pre[1][threadIdx.x] << 32), s_pre[2][threadIdx.x], invalid, sleeper_pair); }
			else {
				{ const uint64_t was = rec_key[ri]; changed |= was != key; rec_key[ri] = key; }
				nh_Record r = { body_a, body_b, ovf, asleep ? NH_REC_SLEEPING : (uint32_t)count };
				if (asleep) st->any_sleeping = 1;
				rec[ri] = r;
				rec_idx[ri] = ri;
			}
		}
		w_n[lane] = (uint8_t)(live ? (count < 4 ? count : 4) : 0);
		__builtin_amdgcn_wave_barrier();
		{
			const size_t run = (size_t)4 * (ri - lane);            // first raw slot of the wave's pairs
#pragma unroll
			for (int r4 = 0; r4 < 4; ++r4) {
				const uint32_t e = 64u * r4 + lane;
				if ((e & 3u) < (uint32_t)w_n[e >> 2]) {
					float4* dp = reinterpret_cast<float4*>(raw_data + run + e);
					dp[0] = w_c0[e]; dp[1] = w_c1[e];
					raw_feature[run + e] = w_f[e];
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	if (STILL) {
		if (__ballot(changed) != 0ull && nh_lane() == 0) atomicMax(&st->still_failed_seq, seq);
		if (LOCAL && sc_count) {
			// sleepers form: the step's sleeping pairs and kept-aside cache entries, one pair of atomics per workgroup that saw any
			__shared__ uint32_t s_sl[2];
			if (threadIdx.x < 2u) s_sl[threadIdx.x] = 0u;
			__syncthreads();
			for (int d = 32; d >= 1; d >>= 1) { n_sleeping += (uint32_t)__shfl_xor((int)n_sleeping, d); n_culled += (uint32_t)__shfl_xor((int)n_culled, d); }
			if (nh_lane() == 0 && n_sleeping) { atomicAdd(&s_sl[0], n_sleeping); atomicAdd(&s_sl[1], n_culled); }
			__syncthreads();
			if (threadIdx.x == 0 && s_sl[0]) { uint32_t* const part = &st->sleep_part[parity][blockIdx.x & (NH_SLEEP_PARTS - 1u)][0]; atomicAdd(&part[0], s_sl[0]); if (s_sl[1]) atomicAdd(&part[1], s_sl[1]); }
		}
		return;
	}
	if (__ballot(changed) != 0ull && nh_lane() == 0) st->keys_changed = 1u;          // (plain store of a constant)
}

// XFORM AHEAD (nh_internal.h: nh_StillStep).  Once per nh_step call, behind the k_xform<true> of its first plain still step: which collider a body's solver lane
// transforms for the next step (body_col, cleared to NH_NONE by the host), whether every dynamic body has at most one (st->ahead_multi), and the scene bounds of the
// colliders no lane owns -- the static world's (body 0), which nothing moves inside the call.
__global__ __launch_bounds__(256) void k_ahead_map(nh_DevState* __restrict__ st, const float4* __restrict__ aabb_min, uint32_t n, uint32_t* __restrict__ body_col) {
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		st->ahead_multi = 0u;
		for (int k = 0; k < 3; ++k) { st->still_static_min[k] = 0xffffffffu; st->still_static_max[k] = 0u; }
	}
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const uint32_t b = __float_as_uint(aabb_min[c].w);
		if (b) body_col[b] = c;          // (several colliders of one body: any of them wins, k_ahead_check sees the others)
	}
}
// (`idle`: sleepers ahead -- a collider of a body asleep is nobody's lane's either: its share is taken here, with the static world's, while the sleeping set stands still)
__global__ __launch_bounds__(256) void k_ahead_check(nh_DevState* __restrict__ st, const float4* __restrict__ aabb_min, uint32_t n, const uint32_t* __restrict__ body_col, const uint8_t* __restrict__ idle = nullptr) {
	__shared__ uint32_t red[6];
	if (threadIdx.x < 3) { red[threadIdx.x] = 0xffffffffu; red[3 + threadIdx.x] = 0u; }
	__syncthreads();
	uint32_t lmin[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, lmax[3] = { 0, 0, 0 };
	bool multi = false;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const float4 mn = aabb_min[c];
		const uint32_t b = __float_as_uint(mn.w);
		if (b) { multi |= body_col[b] != c; if (!(idle && idle[b] == 0xffu)) continue; }
		uint32_t f;
		f = nh_float_flip(mn.x); lmin[0] = min(lmin[0], f); lmax[0] = max(lmax[0], f);
		f = nh_float_flip(mn.y); lmin[1] = min(lmin[1], f); lmax[1] = max(lmax[1], f);
		f = nh_float_flip(mn.z); lmin[2] = min(lmin[2], f); lmax[2] = max(lmax[2], f);
	}
	if (__ballot(multi) != 0ull && nh_lane() == 0) st->ahead_multi = 1u;          // (plain store of a constant)
	for (int k = 0; k < 3; ++k) {
		for (int d = 32; d >= 1; d >>= 1) { lmin[k] = min(lmin[k], (uint32_t)__shfl_xor((int)lmin[k], d)); lmax[k] = max(lmax[k], (uint32_t)__shfl_xor((int)lmax[k], d)); }
		if ((threadIdx.x & 63) == 0) { atomicMin(&red[k], lmin[k]); atomicMax(&red[3 + k], lmax[k]); }
	}
	__syncthreads();
	if (threadIdx.x < 3 && red[threadIdx.x] <= red[3 + threadIdx.x]) { atomicMin(&st->still_static_min[threadIdx.x], red[threadIdx.x]); atomicMax(&st->still_static_max[threadIdx.x], red[3 + threadIdx.x]); }
}

// PAIR AHEAD (nh_internal.h: nh_StillStep).  Once per layout: which kept pairs are NO body's contact-list record -- two neighbours whose inflated boxes overlap, a body
// hovering above ground it does not touch yet.  k_pair_mark stamps every record that is some body's own; k_pair_list lists the others (st->pair_unowned, cleared by
// k_ahead_map).  The solver lanes evaluate their bodies' own pairs for the next step; k_pair_begin evaluates the listed ones.
// (`nbodies`: in a partitioned world stepped through nh_partition_step only the OWNED bodies -- a ghost's lane does not evaluate its pair, the ghost arrives anew every step)
__global__ __launch_bounds__(256) void k_pair_mark(const uint32_t* __restrict__ body_rec, uint32_t nbodies, uint32_t* __restrict__ mark, uint32_t stamp) {
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x + 1u; x < nbodies; x += gridDim.x * blockDim.x) {
		const uint32_t br = body_rec[x];
		if (br != NH_BODY_REC_NONE) mark[br & ~NH_BODY_REC_IS_A] = stamp;
	}
}
// (`rec`: a world with sleepers -- the record of a sleeping pair is nobody's either, a sleeping body has no contact-list record, but nobody has to evaluate it: it stays what it is
// while its body sleeps, and a body that wakes does so in a full step, after which the list is made again)
__global__ __launch_bounds__(256) void k_pair_list(nh_DevState* __restrict__ st, const uint32_t* __restrict__ mark, uint32_t stamp, uint32_t* __restrict__ list, uint32_t cap, const nh_Record* __restrict__ rec = nullptr) {
	const uint32_t n = st->records;
	for (uint32_t r0 = blockIdx.x * blockDim.x; r0 < n; r0 += gridDim.x * blockDim.x) {
		const uint32_t r = r0 + threadIdx.x;
		const bool mine = r < n && mark[r] != stamp && !(rec && rec[r].count == NH_REC_SLEEPING);
		const uint32_t slot = nh_wave_reserve1(&st->pair_unowned, mine);
		if (mine && slot < cap) list[slot] = r;
	}
}

// A still step whose pairs the LAST step's solver lanes have evaluated starts at the solver; this small launch does first what the narrowphase used to do besides: the
// checks that concern the step as a whole, the scene frame of the step filed where a k_xform<true> would have put it (the host's mirror, the sleep prediction), the other
// parity's words cleared for what this step's lanes gather (first workgroup) -- one check of its own: the lanes decided which collider of a pair plays "a" with the LAST
// frame's cells as their yardstick (k_solve_one_body<.., PAIR>), so this step's cells must not be much larger -- and the kept pairs that are NO body's own (k_pair_list),
// evaluated like k_narrowphase<*, true, false> evaluates them: a record nobody's lane solves must stay without contacts and keep its key, or this is not a still step.
__global__ __launch_bounds__(256) void k_pair_begin(nh_DevState* __restrict__ st, uint32_t parity, uint32_t seq, const uint32_t* __restrict__ list, uint32_t list_cap,
                                                    const uint2* __restrict__ kept, uint32_t kept_cap, uint32_t pair_cap, const uint8_t* __restrict__ gen,
                                                    const nh_xform* __restrict__ xf, const float4* __restrict__ aabb_min, const float4* __restrict__ aabb_max, const uint32_t* __restrict__ ctag,
                                                    const nh_BoxCollider* __restrict__ box_data, const nh_SphereCollider* __restrict__ sph_data, uint32_t nbox,
                                                    const uint64_t* __restrict__ rec_key, nh_Record* __restrict__ rec,
                                                    const uint32_t* __restrict__ lay_rank, uint32_t* __restrict__ cnt_sorted, const uint32_t* __restrict__ body_rec, int2* __restrict__ delta,
                                                    nh_Contact* __restrict__ raw_data, uint32_t* __restrict__ raw_feature,
                                                    const nh_BodyMomentum* __restrict__ momentum, const nh_BodyProperties* __restrict__ props, uint32_t delta_scan,
                                                    // sleepers ahead (nh_internal.h): the number of bodies of a world with sleepers in it -- the step's sleeper counts are the last step's
                                                    uint32_t sleepers_bodies = 0u) {
	const uint32_t lane = nh_lane();
	uint32_t fr_min[3], fr_max[3], fr_top;
	{
		const uint32_t* part = &st->ahead_part[parity][lane & (NH_AHEAD_PARTS - 1u)][0];
		fr_top = part[6];
		for (int k = 0; k < 3; ++k) { fr_min[k] = min(part[k], st->still_static_min[k]); fr_max[k] = max(part[3 + k], st->still_static_max[k]); }
	}
	for (int k = 0; k < 3; ++k)
		for (int d = 32; d >= 1; d >>= 1) { fr_min[k] = min(fr_min[k], (uint32_t)__shfl_xor((int)fr_min[k], d)); fr_max[k] = max(fr_max[k], (uint32_t)__shfl_xor((int)fr_max[k], d)); }
	for (int d = 32; d >= 1; d >>= 1) fr_top = max(fr_top, (uint32_t)__shfl_xor((int)fr_top, d));
	const nh_f3 smin = nh_make3(nh_float_unflip(fr_min[0]), nh_float_unflip(fr_min[1]), nh_float_unflip(fr_min[2]));
	const nh_f3 smax = nh_make3(nh_float_unflip(fr_max[0]), nh_float_unflip(fr_max[1]), nh_float_unflip(fr_max[2]));
	const float mscale = nh_morton_scale(smin, smax);
	const nh_f3 mmin = nh_make3(smin.x * mscale, smin.y * mscale, smin.z * mscale);
	const uint32_t n_list = st->pair_unowned;
	if (blockIdx.x == 0 && threadIdx.x < 64u) {
		const uint32_t op = parity ^ 1u;
		const float scale_was = nh_morton_scale(nh_make3(nh_float_unflip(st->still_smin[op][0]), nh_float_unflip(st->still_smin[op][1]), nh_float_unflip(st->still_smin[op][2])),
		                                        nh_make3(nh_float_unflip(st->still_smax[op][0]), nh_float_unflip(st->still_smax[op][1]), nh_float_unflip(st->still_smax[op][2])));
		// (body 0 inert: the solver checks it too -- here so that the step's verdict is complete when this launch ends, nh_StillStep::early_verdict)
		const nh_BodyMomentum m0 = momentum[0];
		const nh_BodyProperties p0 = props[0];
		const bool inert = nh_is_inert(m0.velocity, m0.angular_velocity, p0.inertia_inverse, p0.mass_inverse);
		const bool bad = st->ahead_failed_seq >= seq || st->ahead_multi != 0u || !st->lay_valid || st->fat_count != st->pairs || st->fat_count_sph != st->pairs_sph ||
		                 n_list > list_cap || !(mscale >= 0.75f * scale_was) || !inert;
		__builtin_amdgcn_wave_barrier();
		if (lane == 0) {
			if (bad) { atomicMax(&st->still_failed_seq, seq); atomicAdd(&st->still_diff[3], 1u); }
			if (!(mscale >= 0.75f * scale_was)) atomicAdd(&st->pair_diag[2], 1u);
			if (n_list > list_cap) atomicAdd(&st->pair_diag[3], 1u);
			for (int k = 0; k < 3; ++k) { st->still_smin[parity][k] = fr_min[k]; st->still_smax[parity][k] = fr_max[k]; }
			uint32_t top = fr_top;
			if (sleepers_bodies) {
				// (nobody fell asleep or woke since the last step -- that step's lanes would have failed this one: k_solve_one_body<.., AHEAD> -- so what a k_xform<true> and a
				// narrowphase in sleepers form would count now is what they counted then; the solver's first thread turns the parity words into the step's counters)
				const uint32_t asleep = (sleepers_bodies - 1u) - min(st->active, sleepers_bodies - 1u);
				st->still_asleep[parity] = asleep; st->still_sleeping[parity] = st->sleeping; st->still_culled[parity] = st->culled;
				if (asleep) top = 0xffu;
			}
			st->max_idle[parity] = top;
			// what k_xform<true> does for the step after this one
			st->max_idle[op] = 0u; st->delta_count[op] = 0u; st->delta_overflow[op] = 0u; st->still_esc[op] = 0u;
			st->still_asleep[op] = 0u; st->still_sleeping[op] = 0u; st->still_culled[op] = 0u;
			for (int k = 0; k < 3; ++k) { st->still_smin[op][k] = 0xffffffffu; st->still_smax[op][k] = 0u; }
			st->still_fat0 = st->fat_count + st->fat_count_sph;
		}
		for (uint32_t k = lane; k < NH_AHEAD_PARTS * 8u; k += 64u) (&st->ahead_part[op][0][0])[k] = (k & 7u) < 3u ? 0xffffffffu : 0u;
		if (sleepers_bodies) for (uint32_t k = lane; k < NH_SLEEP_PARTS * 2u; k += 64u) (&st->sleep_part[parity][0][0])[k] = 0u;          // (no narrowphase adds to them in such a step)
	}
	// ---- the kept pairs that are nobody's: exact boxes, stamps, roles by Morton order, contact arithmetic -- and the verdict: no contact, the key on file ----
	const uint32_t n_bb = min(st->pairs, pair_cap);
	const uint32_t index_mask = gen ? NH_GEN_INDEX : 0xFFFFFFFFu;
	const bool stamps = gen && st->moved_count != 0u;
	bool changed = false;
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < min(n_list, list_cap); j += gridDim.x * blockDim.x) {
		const uint32_t r = list[j];
		if (sleepers_bodies && rec[r].count == NH_REC_SLEEPING) continue;          // (a listed pair whose body has fallen asleep since: a settled sleeping pair, nobody's business)
		const uint2 kp = r < n_bb ? kept[r] : kept[kept_cap - 1u - (r - n_bb)];
		const uint32_t ca = kp.x & index_mask, cb = kp.y & index_mask;
		const float4 amin = aabb_min[ca], amax = aabb_max[ca], bmin = aabb_min[cb], bmax = aabb_max[cb];
		bool hit = aabb_overlap(amin, amax, bmin, bmax) && (__float_as_uint(bmin.w) != __float_as_uint(amin.w));
		if (stamps && hit) hit = (uint32_t)(gen[ca] & 0x7Fu) == (kp.x >> NH_GEN_SHIFT) && (uint32_t)(gen[cb] & 0x7Fu) == (kp.y >> NH_GEN_SHIFT);
		// (sleepers ahead: the boxes of a pair with a SLEEPING end overlap -- an awake body has come within reach of a sleeper, which wakes it (its set holds more than itself,
		// nudge.cpp:3575-3650), or a sleeper's pair that was no sleeping pair has become one: either way the three kernels' business, k_narrowphase<*, true, true> -- not this step's)
		if (sleepers_bodies && hit && (__float_as_uint(amax.w) | __float_as_uint(bmax.w)) != 0u) { changed = true; atomicAdd(&st->still_diff[3], 1u); hit = false; }
		int count = 0;
		uint64_t key = ~0ull;
		uint32_t body_a = 0, body_b = 0;
		nh_bb_result bb;
		bb.kind = 0; bb.mask = 0;
		nh_contact_out single;
		single.px = single.py = single.pz = single.penetration = single.nx = single.ny = single.nz = 0.0f; single.friction = 0.5f; single.feature = 0u;
		if (hit) {
			const uint64_t ka = nh_morton_of(nh_make3(amin.x, amin.y, amin.z), mscale, mmin), kb = nh_morton_of(nh_make3(bmin.x, bmin.y, bmin.z), mscale, mmin);
			const bool a_first = (ka < kb) || (ka == kb && ca < cb);
			const uint32_t a = a_first ? cb : ca, b = a_first ? ca : cb;          // narrowphase "a" = later in Morton order (emit_pair)
			const nh_xform A = xf[a], B = xf[b];
			const uint32_t ta = ctag[a], tb = ctag[b];
			const bool a_sph = a >= nbox, b_sph = b >= nbox;
			if (!a_sph && !b_sph) {
				const nh_BoxCollider sa = box_data[a], sb = box_data[b];
				nh_box_box_eval(A, B, sa.size, sb.size, ta, tb, bb);
				count = nh_bb_count(bb);
				key = (uint64_t)(bb.swapped ? tb : ta) | ((uint64_t)(bb.swapped ? ta : tb) << 32);
				body_a = bb.swapped ? B.body : A.body; body_b = bb.swapped ? A.body : B.body;
			} else if (a_sph && b_sph) {
				count = nh_sphere_sphere(sph_data[b - nbox].radius, sph_data[a - nbox].radius, B, A, &single);
				key = (uint64_t)tb | ((uint64_t)ta << 32);
				body_a = B.body; body_b = A.body;
			} else {
				const uint32_t bx = a_sph ? b : a, sp = a_sph ? a : b;
				const nh_BoxCollider sz = box_data[bx];
				const nh_xform BX = a_sph ? B : A, SP = a_sph ? A : B;
				count = nh_box_sphere(sz.size, sph_data[sp - nbox].radius, BX, SP, &single);
				key = (uint64_t)(a_sph ? tb : ta) | ((uint64_t)(a_sph ? ta : tb) << 32);
				body_a = BX.body; body_b = SP.body;
			}
		}
		// still_record (k_narrowphase): the key on file -- or the pair gone --, at most four contacts, and a pair WITH contacts joins a dynamic body to the static world and
		// is that body's own record (a ghost's, in a partitioned world: its lane solves it, this launch evaluates it)
		const uint64_t was_key = rec_key[r];
		const uint32_t pos = lay_rank[r];
		bool ok = (was_key == key || !hit) && count <= 4;
		if (ok && count) {
			const uint32_t dyn = body_a ? body_a : body_b;
			ok = (body_a == 0u) != (body_b == 0u) && (body_rec[dyn] & ~NH_BODY_REC_IS_A) == r;
		}
		if (!ok) { changed = true; atomicAdd(&st->still_diff[(was_key != key && hit) ? 0 : (count > 4 ? 1 : 2)], 1u); count = 0; }
		else if (count) {
			const size_t base = r < n_bb ? (size_t)4 * r : (size_t)4 * n_bb + (r - n_bb);
			if (bb.kind == 2) {
				int k = 0;
#pragma unroll
				for (int index = 0; index < 16; ++index) {
					if (((bb.mask >> index) & 1u) && k < count) {
						nh_contact_out o;
						nh_bb_contact(bb, index, bb.lx[index], bb.ly[index], bb.lz[index], bb.penetration[index], bb.tags[index], o);
						float4* dp = reinterpret_cast<float4*>(raw_data + base + k);
						dp[0] = make_float4(o.px, o.py, o.pz, o.penetration); dp[1] = make_float4(o.nx, o.ny, o.nz, o.friction);
						raw_feature[base + k] = o.feature;
						++k;
					}
				}
			} else {
				const nh_contact_out o = bb.kind == 1 ? bb.edge : single;
				float4* dp = reinterpret_cast<float4*>(raw_data + base);
				dp[0] = make_float4(o.px, o.py, o.pz, o.penetration); dp[1] = make_float4(o.nx, o.ny, o.nz, o.friction);
				raw_feature[base] = o.feature;
			}
		}
		rec[r].count = (uint32_t)count;
		const uint32_t old = cnt_sorted[pos];
		if (old != (uint32_t)count) {
			cnt_sorted[pos] = (uint32_t)count;
			if (!st->delta_overflow[parity]) {
				const uint32_t k = atomicAdd(&st->delta_count[parity], 1u);
				if (k < NH_DELTA_MAX) delta[parity * NH_DELTA_MAX + k] = make_int2((int)pos, count - (int)old);
				else { st->delta_overflow[parity] = 1u; if (!delta_scan) changed = true; }
			}
		}
	}
	if (__ballot(changed) != 0ull && lane == 0) atomicMax(&st->still_failed_seq, seq);
}

// Asleep steps (nh_internal.h: nh_AsleepState): is every collider's world AABB (k_xform's arithmetic, bit for bit) and tag what it was when the world was found asleep,
// and every body still asleep?  One lane per collider / body; any difference raises st->asleep_failed.
__global__ __launch_bounds__(256) void k_asleep_check(nh_DevState* __restrict__ st, const nh_Transform* __restrict__ body_xf, const uint8_t* __restrict__ idle, uint32_t nbodies,
                                                      const nh_Transform* __restrict__ box_xf, const nh_BoxCollider* __restrict__ box_data, const uint32_t* __restrict__ box_tags, uint32_t nbox,
                                                      const nh_Transform* __restrict__ sph_xf, const nh_SphereCollider* __restrict__ sph_data, const uint32_t* __restrict__ sph_tags, uint32_t nsph,
                                                      const float4* __restrict__ was_min, const float4* __restrict__ was_max, const uint32_t* __restrict__ was_tag) {
	bool differs = false;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1u; i < nbodies; i += gridDim.x * blockDim.x) differs |= idle[i] != 0xffu;
	const uint32_t n = nbox + nsph;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const bool is_box = c < nbox;
		const nh_Transform l = is_box ? box_xf[c] : sph_xf[c - nbox];
		if (l.body >= nbodies) { differs = true; continue; }
		const nh_Transform b = body_xf[l.body];
		const nh_quat bq = { b.rotation[0], b.rotation[1], b.rotation[2], b.rotation[3] };
		const nh_quat lq = { l.rotation[0], l.rotation[1], l.rotation[2], l.rotation[3] };
		const nh_f3 p = nh_rotate(bq, nh_make3(l.position[0], l.position[1], l.position[2])) + nh_make3(b.position[0], b.position[1], b.position[2]);
		const nh_quat q = nh_qmul(bq, lq);
		float sx, sy, sz;
		uint32_t tag;
		if (is_box) {
			const nh_m33 m = nh_matrix(q);
			const nh_BoxCollider bc = box_data[c];
			const nh_f3 c0 = m.c0 * bc.size[0], c1 = m.c1 * bc.size[1], c2 = m.c2 * bc.size[2];
			sx = fabsf(c0.x) + fabsf(c1.x) + fabsf(c2.x);
			sy = fabsf(c0.y) + fabsf(c1.y) + fabsf(c2.y);
			sz = fabsf(c0.z) + fabsf(c1.z) + fabsf(c2.z);
			tag = box_tags[c];
		} else {
			sx = sy = sz = sph_data[c - nbox].radius;
			tag = sph_tags[c - nbox];
		}
		const float4 mn = was_min[c], mx = was_max[c];
		const float mnx = p.x - sx, mny = p.y - sy, mnz = p.z - sz, mxx = p.x + sx, mxy = p.y + sy, mxz = p.z + sz;
		differs |= __float_as_uint(mnx) != __float_as_uint(mn.x) || __float_as_uint(mny) != __float_as_uint(mn.y) || __float_as_uint(mnz) != __float_as_uint(mn.z) ||
		           __float_as_uint(mxx) != __float_as_uint(mx.x) || __float_as_uint(mxy) != __float_as_uint(mx.y) || __float_as_uint(mxz) != __float_as_uint(mx.z) ||
		           __float_as_uint(mn.w) != l.body || tag != was_tag[c];
	}
	if (__ballot(differs) != 0ull && nh_lane() == 0) st->asleep_failed = 1u;          // (plain store of a constant)
}

// The one check of an nh_step call on a world that went to sleep (nh_solve.hip: nh_step): 0 = nothing has changed, 1 = something has, < 0 = error code.
int nh_asleep_verify(nh_context* ctx, const nh_BodyData* bodies, const nh_ColliderData* colliders) {
	nh_AsleepState& as = ctx->asleep;
	const uint32_t C = colliders->boxes.count + colliders->spheres.count;
	if (C != as.C || bodies->count != as.B || !as.aabb_min || !bodies->idle_counters) return 1;
	NH_HIP_CHECK(ctx, hipMemsetAsync(&ctx->d_state->asleep_failed, 0, sizeof(uint32_t), ctx->stream));
	NH_LAUNCH(ctx, "asleep_check", k_asleep_check, nh_grid_for(C > bodies->count ? C : bodies->count, 256, 4096), 256, ctx->d_state, bodies->transforms, bodies->idle_counters, bodies->count,
	          colliders->boxes.transforms, colliders->boxes.data, colliders->boxes.tags, colliders->boxes.count,
	          colliders->spheres.transforms, colliders->spheres.data, colliders->spheres.tags, colliders->spheres.count, as.aabb_min, as.aabb_max, as.tags);
	uint32_t failed = 1u;
	NH_HIP_CHECK(ctx, hipMemcpyAsync(&failed, &ctx->d_state->asleep_failed, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
	return failed ? 1 : 0;
}

// ... and the copies it compares with: this step's world AABBs and tags (the caller's arena still holds them), taken when the second all-asleep step in a row ends
// (round-4 advisor finding: the copy used to be taken in finish_setup, inside nh_apply_impulses, out of nh_collide's scratch -- memory of the by-value arena that
// nh_read_cached_impulses and nh_setup_contact_constraints had long handed out again; valid only because nothing happened to write there in a world asleep.  Now the
// copy is enqueued by nh_collide itself, right behind k_xform, in the steps that can become the second of the streak; finish_setup only adopts it)
int nh_asleep_remember(nh_context* ctx) {
	nh_AsleepState& as = ctx->asleep;
	if (!as.snap_C || as.snap_seq != ctx->collide_seq || as.snap_C != as.last_C) return 1;
	as.C = as.snap_C;
	return 0;
}

static int asleep_snapshot(nh_context* ctx) {
	nh_AsleepState& as = ctx->asleep;
	const uint32_t C = as.last_C;
	as.snap_C = 0;
	if (!C || !as.last_aabb_min) return 1;
	if (as.capacity < C) {
		void** bufs[] = { (void**)&as.aabb_min, (void**)&as.aabb_max, (void**)&as.tags };
		for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
		as.capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&as.aabb_min, sizeof(float4) * (size_t)C));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&as.aabb_max, sizeof(float4) * (size_t)C));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&as.tags, sizeof(uint32_t) * (size_t)C));
		as.capacity = C;
	}
	NH_HIP_CHECK(ctx, hipMemcpyAsync(as.aabb_min, as.last_aabb_min, sizeof(float4) * (size_t)C, hipMemcpyDeviceToDevice, ctx->stream));
	NH_HIP_CHECK(ctx, hipMemcpyAsync(as.aabb_max, as.last_aabb_max, sizeof(float4) * (size_t)C, hipMemcpyDeviceToDevice, ctx->stream));
	NH_HIP_CHECK(ctx, hipMemcpyAsync(as.tags, as.last_ctag, sizeof(uint32_t) * (size_t)C, hipMemcpyDeviceToDevice, ctx->stream));
	as.snap_C = C; as.snap_seq = ctx->collide_seq;
	return 0;
}

// ---- 7. islands (lock-free union-find), active bodies, sleeping -------------------------------------------
// find with path halving: every visited node is re-pointed at its grandparent (an ancestor stays an ancestor whatever other
// lanes do, and parent[x] <= x is preserved), so the long chains of one big island (ball pit) collapse as they are walked
__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t x) {
	uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	while (p != x) {
		uint32_t gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		x = p;
		p = gp;
	}
	return x;
}

__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
	for (;;) {
		a = uf_find(parent, a);
		b = uf_find(parent, b);
		if (a == b) return;
		if (a < b) { uint32_t t = a; a = b; b = t; }       // hook the larger root under the smaller: root = min index
		uint32_t old = atomicCAS(&parent[a], a, b);
		if (old == a) return;
	}
}

__global__ __launch_bounds__(256) void k_uf_union_records(const nh_DevState* __restrict__ st, const nh_Record* __restrict__ rec, uint32_t* __restrict__ parent, uint32_t parity,
                                                          const uint8_t* __restrict__ idle) {
	if (!st->any_idle[parity]) return;          // nobody sleeps: every set is active, no need to know the sets
	uint32_t n = st->records;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		nh_Record r = rec[i];
		// body 0 is the static world (nudge.cpp:3874-3876); pairs dropped at the coarse level have no contacts; edges between two awake bodies: see emit_pair
		if ((r.count & ~NH_REC_SLEEPING) && r.body_a && r.body_b && (idle[r.body_a] == 0xff || idle[r.body_b] == 0xff)) uf_union(parent, r.body_a, r.body_b);
	}
}

__global__ __launch_bounds__(256) void k_uf_union_connections(const nh_DevState* __restrict__ st, const nh_BodyPair* __restrict__ con, uint32_t ncon, uint32_t* __restrict__ parent, uint32_t parity,
                                                              const uint8_t* __restrict__ idle) {
	if (!st->any_idle[parity]) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ncon; i += gridDim.x * blockDim.x) {
		nh_BodyPair p = con[i];
		if (p.a && p.b && (idle[p.a] == 0xff || idle[p.b] == 0xff)) uf_union(parent, p.a, p.b);                          // nudge.cpp:3799-3807
	}
}

// flatten + "a set is active if any member's idle counter != 0xff" (nudge.cpp:3960-3963)
__global__ __launch_bounds__(256) void k_uf_flatten(const nh_DevState* __restrict__ st, uint32_t parity, uint32_t nbodies, uint32_t* __restrict__ parent, const uint8_t* __restrict__ idle, uint32_t* __restrict__ set_active, uint32_t* __restrict__ root_of) {
	if (!st->any_idle[parity]) return;          // nobody sleeps: every body is active, the sets are not needed (k_active_write)
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		uint32_t r = uf_find(parent, i);
		root_of[i] = r;
		if (i >= 1 && idle[i] != 0xff) set_active[r] = 1;
	}
}

__global__ __launch_bounds__(256) void k_active_flags(const nh_DevState* __restrict__ st, uint32_t parity, uint32_t nbodies, const uint32_t* __restrict__ root_of, const uint32_t* __restrict__ set_active, uint32_t* __restrict__ flags) {
	if (!st->any_idle[parity]) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x)
		flags[i] = (i >= 1 && set_active[root_of[i]]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_active_write(nh_DevState* __restrict__ st, uint32_t nbodies, const uint32_t* __restrict__ root_of, const uint32_t* __restrict__ set_active,
                                                      const uint32_t* __restrict__ scan, uint32_t* __restrict__ indices, uint32_t capacity, uint32_t parity, uint32_t no_islands,
                                                      const uint32_t* __restrict__ block_top, uint32_t n_block_top) {
	if (blockIdx.x == 0 && threadIdx.x < 64u) {
		// largest idle counter seen by k_collide_begin's workgroups -> the counter block the host mirrors
		uint32_t top = 0;
		for (uint32_t k = threadIdx.x; k < n_block_top; k += 64u) top = max(top, block_top[k]);
		for (int d = 32; d >= 1; d >>= 1) top = max(top, (uint32_t)__shfl_xor((int)top, d));
		if (threadIdx.x == 0) st->max_idle[parity] = top;
	}
	// `no_islands`: the host ruled out sleepers and launched no island kernel; if the device disagrees the idle counters were changed behind
	// the library's back (nh_bodies_changed): reported, and the step falls back to "everybody active" instead of reading unset arrays
	if (st->stale_hint && blockIdx.x == 0 && threadIdx.x == 0) { st->error = NH_ERR_STALE_HINT; st->stale_hint = 0; }
	if (no_islands || !st->any_idle[parity]) {
		// nobody sleeps: the active list is every body but the static world, ascending (nudge.cpp:3966-3971 with all sets active)
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1u; i < nbodies; i += gridDim.x * blockDim.x) {
			if (i - 1u < capacity) indices[i - 1u] = i; else st->error = NH_ERR_ACTIVE_CAPACITY;
		}
		if (blockIdx.x == 0 && threadIdx.x == 0) st->active = nbodies ? nbodies - 1u : 0u;
		return;
	}
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		if (i >= 1 && set_active[root_of[i]]) {
			uint32_t pos = scan[i];
			if (pos < capacity) indices[pos] = i; else st->error = NH_ERR_ACTIVE_CAPACITY;
		}
	}
}

// contacts of inactive sets leave the contact list and become sleeping pairs (nudge.cpp:3976-4003)
__global__ __launch_bounds__(256) void k_filter_records(nh_DevState* __restrict__ st, nh_Record* __restrict__ rec,
                                                        const uint32_t* __restrict__ root_of, const uint32_t* __restrict__ set_active, uint32_t parity) {
	if (!st->any_idle[parity]) return;          // nobody sleeps: no record can belong to a sleeping set
	const uint32_t n = st->records;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const nh_Record r = rec[i];
		if (r.count && !(r.count & NH_REC_SLEEPING)) {
			const uint32_t set = root_of[r.body_a ? r.body_a : r.body_b];      // sets[a] | sets[b] with sets[0] = 0 (nudge.cpp:3988)
			// no contacts any more; the record keeps its key: sleeping pairs come out in tag order with the sorted records (k_gather_contacts)
			if (!set_active[set]) { rec[i].count = NH_REC_SLEEPING; st->any_sleeping = 1; }
		}
	}
}

// ---- 8. tag-ordered output ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sorted_counts(nh_DevState* __restrict__ st, const uint32_t* __restrict__ rec_idx, const nh_Record* __restrict__ rec,
                                                       uint32_t* __restrict__ counts, uint32_t* __restrict__ sleep_flags, uint32_t order_kept, uint32_t* __restrict__ lay_rank) {
	uint32_t n = st->records;
	// sort re-use: `rec_idx` is the context's kept tag order (order_kept), sorted this step or still holding from an earlier one -- either way it now
	// belongs to the keys k_narrowphase left in the context's buffer and to this record count
	if (blockIdx.x == 0 && threadIdx.x == 0) { st->sort_valid = order_kept; st->records_kept = n; }
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t ri = rec_idx[i];
		nh_Record r = rec[ri];
		lay_rank[ri] = i;                      // (where a still step puts the record's count: nh_internal.h, contact storage by slot)
		counts[i] = r.count & ~NH_REC_SLEEPING;
		sleep_flags[i] = (r.count & NH_REC_SLEEPING) ? 1u : 0u;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) { sleep_flags[n] = 0; counts[n] = 0; }
}

// Tag-ordered output.  A wave takes 64 consecutive records (= collider pairs with contacts, in tag order); their contacts
// occupy ONE contiguous range of the output.  Each lane ranks its record's contacts by feature (reference: low 32 tag bits,
// nudge.cpp:4031-4036) and posts "output slot -> raw slot, owner lane" to LDS; then the wave copies the range with one lane
// per output contact: coalesced 32-B stores, no search.
__global__ __launch_bounds__(256) void k_gather_contacts(nh_DevState* __restrict__ st, const uint32_t* __restrict__ rec_idx, const nh_Record* __restrict__ rec, const uint64_t* __restrict__ rec_key_sorted,
                                                         const uint32_t* __restrict__ starts, const nh_Contact* __restrict__ raw_data, const uint32_t* __restrict__ raw_feature,
                                                         nh_Contact* __restrict__ data, nh_BodyPair* __restrict__ bodies, uint64_t* __restrict__ tags, uint32_t* __restrict__ features, uint32_t capacity, uint32_t pair_cap,
                                                         const uint32_t* __restrict__ sleep_scan, uint64_t* __restrict__ sleeping_pairs, uint32_t* __restrict__ deg, uint32_t nbodies,
                                                         uint32_t* __restrict__ dense_slot, uint32_t order_kept, uint32_t view_only) {
	// view_only: the dense list is laid out again for the caller from what still steps keep by slot (nh_still_sync_outputs): no per-body bookkeeping, no sleeping pairs
	__shared__ uint32_t s_src[4][1024];
	unsigned long long* __restrict__ pair_counter = reinterpret_cast<unsigned long long*>(deg + 2u * NH_DEG_STRIDE(nbodies));
	uint32_t* __restrict__ first_contact = deg + 4u * NH_DEG_STRIDE(nbodies);
	__shared__ uint32_t s_feat[4][1024];
	__shared__ uint8_t s_own[4][1024];
	const uint32_t nrec = st->records;
	const uint32_t n_bb = min(st->pairs, pair_cap);
	if (st->contacts > capacity && blockIdx.x == 0 && threadIdx.x == 0) st->error = NH_ERR_CONTACT_CAPACITY;
	// The layout this launch leaves (per record: first contact | contact count << 28, below) is what a still step compares itself with and writes through:
	// usable when a record's position is its kept pair's (k_kept_filter wrote in place), nothing sleeps, and everything fits
	if (blockIdx.x == 0 && threadIdx.x == 0 && !view_only)
		// (records of sleeping pairs are part of a usable layout since round 5 -- whether a step with sleepers may follow is the host's call: finish_setup)
		st->lay_valid = (st->lay_in_place && order_kept && !st->error && st->contacts <= capacity && st->contacts < (1u << 28)) ? 1u : 0u;
	const uint32_t wave = threadIdx.x >> 6, lane = nh_lane();
	uint32_t* src = s_src[wave];
	uint32_t* sft = s_feat[wave];
	uint8_t* own = s_own[wave];
	for (uint32_t lo0 = (blockIdx.x * 4u + wave) * 64u; lo0 < nrec; lo0 += gridDim.x * 256u) {
		const uint32_t lo = lo0 + lane;
		uint32_t cnt = 0, start = 0, body_a = 0, body_b = 0, base4 = 0, baseo = 0;
		bool sph_rec = false;
		uint64_t key = 0;
		if (lo < nrec) {
			const uint32_t ri = rec_idx[lo];
			const nh_Record r = rec[ri];
			start = starts[lo];
			cnt = (r.count & NH_REC_SLEEPING) ? 0u : (r.count > 16u ? 16u : r.count);
			key = rec_key_sorted[lo];
			// sleeping pairs in ascending tag order (the reference sorts them at nudge.cpp:4008): ordered compaction of the sorted records
			if ((r.count & NH_REC_SLEEPING) && sleeping_pairs && !view_only) {
				const uint32_t p = sleep_scan[lo];
				if (p < capacity) sleeping_pairs[p] = key; else st->error = NH_ERR_CONTACT_CAPACITY;
			}
			body_a = r.body_a; body_b = r.body_b;
			sph_rec = ri >= n_bb;                   // (records of pairs with a sphere follow the box-box ones and own ONE raw slot each: k_narrowphase<true>)
			base4 = sph_rec ? 4u * n_bb + (ri - n_bb) : 4u * ri; baseo = 4u * pair_cap + r.overflow;
			// contact degree of the two bodies (body 0 is the static world): the solver's adjacency starts from these
			// and what lets setup skip the general CSR build for a body that sits in ONE pair with the static world: its contacts are start .. start+cnt-1
			// ONE atomic per body and pair: degree and pair info share a 64-bit counter (two 32-bit atomics per body used to make this kernel
			// atomic-bound in a pit of 11 M contacts); no value is asked back -- a returning atomic is a memory round trip the wave waits for
			if (cnt && start + cnt <= capacity && !view_only) {
				// (where a body of the one-pair class finds its record and the record's place in the tag order: k_adj_simple -> body_rec / body_pos)
				if (body_a) first_contact[body_a + NH_DEG_STRIDE(nbodies)] = lo;
				if (body_b) first_contact[body_b + NH_DEG_STRIDE(nbodies)] = lo;
#if defined(NH_GATHER_32)      // A/B only: the two 32-bit atomics per body of rounds 1 and 2, on the halves of the same counter
				if (body_a) { uint32_t* h = reinterpret_cast<uint32_t*>(&pair_counter[body_a]); atomicAdd(h, cnt); atomicAdd(h + 1, body_b ? 0x10001u : 1u); first_contact[body_a] = start | 0x80000000u; }
				if (body_b) { uint32_t* h = reinterpret_cast<uint32_t*>(&pair_counter[body_b]); atomicAdd(h, cnt); atomicAdd(h + 1, body_a ? 0x10001u : 1u); first_contact[body_b] = start; }
#else
				if (body_a) { atomicAdd(&pair_counter[body_a], (unsigned long long)cnt | ((unsigned long long)(body_b ? 0x10001u : 1u) << 32)); first_contact[body_a] = start | 0x80000000u; }      // top bit: plays "a"
				if (body_b) { atomicAdd(&pair_counter[body_b], (unsigned long long)cnt | ((unsigned long long)(body_a ? 0x10001u : 1u) << 32)); first_contact[body_b] = start; }
#endif
			}
		}
		// (first output slot of the wave's records: the smallest start among those that HAVE contacts -- a record without any may carry a start that still steps
		// have not kept up to date, nh_internal.h "contact storage by slot")
		uint32_t base = cnt ? start : 0xFFFFFFFFu;
		for (int d = 32; d >= 1; d >>= 1) base = min(base, (uint32_t)__shfl_xor((int)base, d));
		{
			const uint32_t rel = start - base;
			if (cnt && sph_rec) {
				src[rel] = base4; sft[rel] = raw_feature[base4]; own[rel] = (uint8_t)lane;          // (one contact, one slot)
			} else if (cnt && cnt <= 4) {
				const uint4 fv = *reinterpret_cast<const uint4*>(raw_feature + base4);
				const uint32_t f[4] = { fv.x, fv.y, fv.z, fv.w };
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					if ((uint32_t)j < cnt) {
						uint32_t rank = 0;
#pragma unroll
						for (int i = 0; i < 4; ++i) rank += ((uint32_t)i < cnt && (f[i] < f[j] || (f[i] == f[j] && i < j))) ? 1u : 0u;
						src[rel + rank] = base4 + j; sft[rel + rank] = f[j]; own[rel + rank] = (uint8_t)lane;
					}
				}
			} else if (cnt) {
				for (uint32_t j = 0; j < cnt; ++j) {
					const uint32_t sj = j < 4 ? base4 + j : baseo + (j - 4);
					const uint32_t fj = raw_feature[sj];
					uint32_t rank = 0;
					for (uint32_t i = 0; i < cnt; ++i) {
						const uint32_t fi = raw_feature[i < 4 ? base4 + i : baseo + (i - 4)];
						rank += (fi < fj || (fi == fj && i < j)) ? 1u : 0u;
					}
					src[rel + rank] = sj; sft[rel + rank] = fj; own[rel + rank] = (uint8_t)lane;
				}
			}
		}
		uint32_t total = (lo < nrec && cnt) ? (start - base + cnt) : 0u;
		for (int d = 32; d >= 1; d >>= 1) total = max(total, (uint32_t)__shfl_xor((int)total, d));
		__builtin_amdgcn_wave_barrier();
		for (uint32_t k0 = 0; k0 < total; k0 += 64) {           // wave-uniform: every lane takes part in the shuffles
			const uint32_t k = k0 + lane;
			const bool on = k < total;
			const uint32_t o = on ? own[k] : 0u;
			const uint32_t ba = (uint32_t)__shfl((int)body_a, (int)o), bb = (uint32_t)__shfl((int)body_b, (int)o);
			const uint32_t klo = (uint32_t)__shfl((int)(uint32_t)key, (int)o), khi = (uint32_t)__shfl((int)(uint32_t)(key >> 32), (int)o);
			const uint32_t d = base + k;
			if (on && d < capacity) {
				const float4* sp = reinterpret_cast<const float4*>(raw_data + src[k]);
				float4 c0 = sp[0], c1 = sp[1];
				float4* dp = reinterpret_cast<float4*>(data + d);
				dp[0] = c0; dp[1] = c1;
				nh_BodyPair bp = { ba, bb };
				bodies[d] = bp;
				tags[d] = (uint64_t)klo | ((uint64_t)khi << 32);
				features[d] = sft[k];
				dense_slot[d] = src[k];
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// counters, bounds, and the union-find / island arrays (parent[i] = i, nothing active yet)
// Also notes whether ANY body is asleep (idle counter saturated): if none is, every island is active whatever its shape and the
// union-find over the contact records -- expensive when millions of bodies form one island -- is skipped.  The flag is double
// buffered by step parity: this step's launch sets any_idle[parity] with plain stores and clears the other one for the next step.
__global__ __launch_bounds__(256) void k_collide_begin(nh_DevState* st, uint32_t ncolliders, uint32_t nbodies, uint32_t* __restrict__ parent, uint32_t* __restrict__ set_active, uint32_t* __restrict__ deg,
                                                       const uint8_t* __restrict__ idle, uint32_t parity, uint32_t* __restrict__ coarse_parent, uint32_t* __restrict__ coarse_active, uint32_t no_islands,
                                                       uint32_t* __restrict__ block_top, uint32_t drop_kept_pairs, uint32_t drop_sort_order) {
	__shared__ uint32_t s_top[4];
	// `no_islands`: the host has ruled out that anybody is asleep this step (nh_collide): no island kernel will run, their arrays stay untouched
	bool asleep = false;
	uint32_t top = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		if (!no_islands) { parent[i] = i; set_active[i] = 0; coarse_parent[i] = i; coarse_active[i] = 0; }
		const uint32_t c = i >= 1 ? idle[i] : 0u;
		top = max(top, c);
		asleep |= c == 0xffu;
	}
	// (asleep although the host ruled it out: a broken promise, see nh_bodies_changed.  The step then runs as if nobody slept -- every kernel keyed
	// on any_idle stays off the unset island arrays -- and k_active_write reports NH_ERR_STALE_HINT)
	if (asleep) { if (no_islands) st->stale_hint = 1; else st->any_idle[parity] = 1; }
	// largest idle counter of the world, for the host's sleep prediction: per workgroup here (plain stores: thousands of atomics on one
	// address would cost ~12 ns each, serialised), finished by k_active_write
	for (int d = 32; d >= 1; d >>= 1) top = max(top, (uint32_t)__shfl_xor((int)top, d));
	if (nh_lane() == 0) s_top[threadIdx.x >> 6] = top;
	__syncthreads();
	if (threadIdx.x == 0) block_top[blockIdx.x] = max(max(s_top[0], s_top[1]), max(s_top[2], s_top[3]));
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		st->any_idle[parity ^ 1u] = 0;
		// (what a still step of the other parity -- the next step, if the host launches it as one -- accumulates into, and the verdict on this step's layout)
		st->max_idle[parity ^ 1u] = 0u; st->lay_valid = 0u; st->lay_in_place = 0u; st->delta_count[parity ^ 1u] = 0u; st->delta_overflow[parity ^ 1u] = 0u; st->still_esc[parity ^ 1u] = 0u;
		st->still_asleep[parity ^ 1u] = 0u; st->still_sleeping[parity ^ 1u] = 0u; st->still_culled[parity ^ 1u] = 0u;
		for (int k = 0; k < 3; ++k) { st->still_smin[parity ^ 1u][k] = 0xffffffffu; st->still_smax[parity ^ 1u][k] = 0u; }
	}
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < NH_DEG_WORDS(nbodies); i += gridDim.x * blockDim.x) deg[i] = 0;      // degrees, pair counters, first contacts (NH_DEG_STRIDE)
	if (blockIdx.x != 0) return;
	uint32_t t = threadIdx.x;
	if (t == 0) {
		st->colliders = ncolliders; st->pairs = 0; st->pairs_sph = 0; st->raw_contacts = 0; st->records = 0; st->contacts = 0; st->sleeping = 0;
		st->active = 0; st->culled = 0; st->general_contacts = 0; st->levels = 0; st->error = 0;
		st->has_other = 0; st->has_unstable = 0; st->unleveled = 0; st->has_static8 = 0; st->has_staticN = 0; st->any_sleeping = 0; st->has_pending = 0;
		st->sort_buckets = st->sort_buckets_next;
		st->fat_escaped = 0; st->fat_rebuild = 0; st->esc_count = 0; st->fat_reinsert = 0;
		st->fat_hits_prev = st->fat_hits; st->fat_hits = 0;
		st->keys_changed = 0;
		if (drop_sort_order) st->sort_valid = 0;
		if (drop_kept_pairs) st->fat_valid = 0;       // (collider counts or buffers changed, or NH_NO_KEPT_PAIRS: the kept pair list describes another world)
		for (int k = 0; k < 3; ++k) { st->smin[k] = 0xffffffffu; st->smax[k] = 0u; st->gmin[k] = 0xffffffffu; st->gmax[k] = 0u; }
	}
	for (uint32_t k = t; k < 1024u; k += blockDim.x) st->exp_hist[k] = 0;
}

// =================================================================================================================
// ---- 9. still steps (nh_internal.h: nh_StillStep) --------------------------------------------------------------------------------------------------------
static bool same_bodies(const nh_BodyData& a, const nh_BodyData& b) {
	return a.transforms == b.transforms && a.properties == b.properties && a.momentum == b.momentum && a.idle_counters == b.idle_counters && a.count == b.count;
}
static bool same_colliders(const nh_ColliderData& a, const nh_ColliderData& b) {
	return a.boxes.tags == b.boxes.tags && a.boxes.data == b.boxes.data && a.boxes.transforms == b.boxes.transforms && a.boxes.count == b.boxes.count &&
	       a.spheres.tags == b.spheres.tags && a.spheres.data == b.spheres.data && a.spheres.transforms == b.spheres.transforms && a.spheres.count == b.spheres.count;
}
static bool same_contacts(const nh_ContactData& a, const nh_ContactData& b) {
	return a.data == b.data && a.bodies == b.bodies && a.tags == b.tags && a.features == b.features && a.capacity == b.capacity && a.sleeping_pairs == b.sleeping_pairs;
}

static int collide_impl(nh_context* ctx, nh_ActiveBodies* active_bodies, nh_ContactData* contacts, const nh_BodyData* bodies, const nh_ColliderData* colliders,
                        const nh_BodyConnections* body_connections, nh_Arena temporary, bool want_still);

extern "C" int nh_collide(nh_context* ctx, nh_ActiveBodies* active_bodies, nh_ContactData* contacts,
                          const nh_BodyData* bodies, const nh_ColliderData* colliders,
                          const nh_BodyConnections* body_connections, nh_Arena temporary) {
	if (!ctx || !active_bodies || !contacts || !bodies || !colliders) return NH_ERR_INVALID;
	NH_HIP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	nh_StillStep& ss = ctx->still;
	ss.active = false; ss.resolved = false; ss.setup_d = nullptr;
	// A still step?  The last step must have left everything one relies on (ok_next: decided at its round trip), the contact cache must be that step's contact
	// list, the caller must be in the mode that promises the sample's call order, and the world must be the one the kept layout describes.
	const bool held = ss.hold != 0u;          // (back-off after failed still steps: nh_internal.h)
	if (held) ss.hold--;
	const bool want_still = !held && !ss.disabled && !ctx->first_ghost && ss.ok_next && ss.cache_ok && ss.slots_current && !ss.appended && (ctx->flags & NH_FLAG_FUSED_STEP) && !(ctx->flags & (NH_FLAG_SYNC_COUNTS | NH_FLAG_EXACT_ORDER)) &&
	                        same_bodies(*bodies, ss.lay_bodies) && same_colliders(*colliders, ss.lay_colliders) && same_contacts(*contacts, ss.lay_contacts) &&
	                        active_bodies->indices == ss.lay_active && active_bodies->capacity == ss.lay_active_capacity;
	// (kept for the replay of a still step that fails, and as what the next layout belongs to)
	ss.active_out = active_bodies; ss.contacts_out = contacts; ss.bodies = *bodies; ss.colliders = *colliders;
	ss.has_connections = body_connections != nullptr;
	if (body_connections) ss.connections = *body_connections;
	ss.temporary = temporary;
	return collide_impl(ctx, active_bodies, contacts, bodies, colliders, body_connections, temporary, want_still);
}

// A still step that has not been confirmed becomes a full one: nh_collide again, in full, with the arguments it was given -- and the adjacency kernel the still
// setup left out, if setup has run.  Called by every entry point outside the sample's call order (nh_flush_pending) and by the solver's round trip when a check
// failed (first_apply).  Nothing the still kernels wrote matters: contact data is rewritten, momentum / transforms / cache were not touched.
int nh_still_collide_again(nh_context* ctx) {
	nh_StillStep& ss = ctx->still;
	const bool after_collide = ctx->after_collide, may_overlap = ctx->gravity_may_overlap;
	ss.active = false; ss.resolved = false; ss.ok_next = false;
	ss.failed++; ss.note_failure();
	ss.replaying = true;
	const int rc = collide_impl(ctx, ss.active_out, ss.contacts_out, &ss.bodies, &ss.colliders, ss.has_connections ? &ss.connections : nullptr, ss.temporary, false);
	ss.replaying = false;
	ctx->after_collide = after_collide; ctx->gravity_may_overlap = may_overlap;
	return rc;
}

// The dense, tag-ordered contact list of the API laid out again from what still steps keep by slot (records, raw slots, counts and starts in tag order): the
// same kernel a full step uses, without its per-body bookkeeping.  For whoever looks at contacts->data / bodies / tags / features after still steps.
int nh_still_view_contacts(nh_context* ctx) {
	const nh_ContactData& c = ctx->still.lay_contacts;
	const uint32_t pair_cap = ctx->lay_capacity;
	NH_LAUNCH(ctx, "gather_contacts", k_gather_contacts, nh_grid_for(pair_cap, 256, 8192), 256, ctx->d_state, ctx->sort_sorted_idx, ctx->rec, ctx->sort_sorted_keys, ctx->start_sorted,
	          ctx->raw_data, ctx->raw_feature, c.data, c.bodies, c.tags, c.features, c.capacity, pair_cap, (const uint32_t*)nullptr, (uint64_t*)nullptr, ctx->deg, ctx->still.bodies.count,
	          ctx->dense_slot, 1u, 1u);
	return NH_OK;
}

static int collide_impl(nh_context* ctx, nh_ActiveBodies* active_bodies, nh_ContactData* contacts, const nh_BodyData* bodies, const nh_ColliderData* colliders,
                        const nh_BodyConnections* body_connections, nh_Arena temporary, bool want_still) {
	ctx->adv.done = false;
	ctx->step_parity ^= 1u;
	ctx->collide_seq++;
	ctx->collide_mark = ctx->advance_count;
	// can anybody be asleep?  (counters rise by at most one per nh_advance since the nh_collide that measured them)
	const bool no_islands = ctx->idle_bound >= 0 && (uint64_t)ctx->idle_bound + (ctx->advance_count - ctx->idle_bound_mark) < 0xffu;
	ctx->islands_skipped = no_islands;
	ctx->idle_unknown = false;                   // this call looks at the counters as they are now (k_collide_begin): its maximum may be adopted again
	nh_DevState* st = ctx->d_state;
	const uint32_t nbox = colliders->boxes.count, nsph = colliders->spheres.count;
	const uint32_t C = nbox + nsph;
	const uint32_t B = bodies->count;
	const uint32_t cap = contacts->capacity;
	ctx->body_count = B;

	int err = NH_OK;
	nh_Arena* A = &temporary;
	nh_xform* xf = nh_arena_array<nh_xform>(A, C, &err);
	float4* aabb_min = nh_arena_array<float4>(A, C, &err);
	float4* aabb_max = nh_arena_array<float4>(A, C, &err);
	uint32_t* ctag = nh_arena_array<uint32_t>(A, C, &err);
	ctx->asleep.last_aabb_min = aabb_min; ctx->asleep.last_aabb_max = aabb_max; ctx->asleep.last_ctag = ctag; ctx->asleep.last_C = C;          // (nh_asleep_remember)
	uint32_t* keys_a = nh_arena_array<uint32_t>(A, C, &err);
	uint32_t* keys_b = nh_arena_array<uint32_t>(A, C, &err);
	uint32_t* vals_a = nh_arena_array<uint32_t>(A, C, &err);
	// broadphase pairs (each owns 4 raw contact slots, further contacts overflow): nh_set_pair_capacity, else half the contact capacity
	const uint32_t pair_cap = ctx->pair_capacity ? ctx->pair_capacity : cap / 2 + 1024;
	const bool onesweep = !ctx->env_sort_classic;                    // one kernel per radix pass (nh_util.hip); NH_SORT_CLASSIC=1: three
	size_t hist_words = 256 * NH_SORT_GRID + 512;
	if (onesweep && nh_sort_scratch_words(C > pair_cap ? C : pair_cap) > hist_words) hist_words = nh_sort_scratch_words(C > pair_cap ? C : pair_cap);
	uint32_t* hist = nh_arena_array<uint32_t>(A, hist_words, &err);
	uint32_t* scan_tmp = nh_arena_array<uint32_t>(A, 2 * NH_SORT_GRID + 64, &err);
	// grid table: a power of two >= 4 cells per collider, 2^16 .. 2^24 entries (flat scenes populate one layer of a grid that is three cells high:
	// with fewer entries the cell has to double -- four times the candidates per collider; measured at 2 M mixed bodies: pair search 153 -> 80 us)
	uint64_t want = 1; while (want < 4ull * C) want <<= 1;
	if (want < (1u << 16)) want = 1u << 16;
	if (want > (1u << 24)) want = 1u << 24;
	const uint32_t max_cells = (uint32_t)want;
	int cell_bits = 0; while ((1ull << cell_bits) < want) ++cell_bits;
	uint32_t* cscan_sums = nh_arena_array<uint32_t>(A, max_cells / CS_TILE + 8, &err);
	const uint32_t overflow_cap = cap;
	uint2* pairs = nh_arena_array<uint2>(A, pair_cap, &err);
	uint64_t* rec_key_b = nh_arena_array<uint64_t>(A, pair_cap, &err);
	uint32_t* rec_idx_a = nh_arena_array<uint32_t>(A, pair_cap, &err);
	uint32_t* rec_idx_b = nh_arena_array<uint32_t>(A, pair_cap, &err);
	uint2* sort_place = nh_arena_array<uint2>(A, pair_cap, &err);
	uint32_t* sleep_flags = nh_arena_array<uint32_t>(A, pair_cap + 1, &err);
	uint32_t* parent = nh_arena_array<uint32_t>(A, B, &err);
	uint32_t* set_active = nh_arena_array<uint32_t>(A, B, &err);
	uint32_t* root_of = nh_arena_array<uint32_t>(A, B, &err);
	uint32_t* flags = nh_arena_array<uint32_t>(A, B + 1, &err);
	uint32_t* coarse_parent = nh_arena_array<uint32_t>(A, B, &err);
	uint32_t* coarse_active = nh_arena_array<uint32_t>(A, B, &err);
	uint32_t* coarse_root = nh_arena_array<uint32_t>(A, B, &err);
	const uint32_t begin_grid = nh_grid_for(B, 256, 2048);
	uint32_t* block_top = nh_arena_array<uint32_t>(A, begin_grid, &err);
	if (err) return err;

	hipStream_t s = ctx->stream;
	bool drop_sort_order = ctx->env_no_sort_reuse;
	if (ctx->deg_capacity < NH_DEG_WORDS(B)) {
		// library-owned (layout: NH_DEG_STRIDE in nh_internal.h): per body the contact degree, the pair info (collider pairs it is in | pairs with a
		// dynamic partner << 16) and the first contact of its last pair -- written while the contacts are laid out, consumed by setup
		if (ctx->deg) NH_HIP_CHECK(ctx, hipFree(ctx->deg));
		ctx->deg = nullptr; ctx->deg_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->deg, sizeof(uint32_t) * NH_DEG_WORDS(B)));
		ctx->deg_capacity = (uint32_t)NH_DEG_WORDS(B);
	}

	if (ctx->sort_capacity != pair_cap) {
		// library-owned: splitters of the seeded tag sort (they persist from one nh_collide to the next) + its bucket counters
		if (ctx->sort_splitters) NH_HIP_CHECK(ctx, hipFree(ctx->sort_splitters));
		if (ctx->sort_counts) NH_HIP_CHECK(ctx, hipFree(ctx->sort_counts));
		if (ctx->sort_starts) NH_HIP_CHECK(ctx, hipFree(ctx->sort_starts));
		if (ctx->sort_keys_by_position) NH_HIP_CHECK(ctx, hipFree(ctx->sort_keys_by_position));
		if (ctx->sort_sorted_keys) NH_HIP_CHECK(ctx, hipFree(ctx->sort_sorted_keys));
		if (ctx->sort_sorted_idx) NH_HIP_CHECK(ctx, hipFree(ctx->sort_sorted_idx));
		ctx->sort_keys_by_position = nullptr; ctx->sort_sorted_keys = nullptr; ctx->sort_sorted_idx = nullptr;
		ctx->sort_splitters = nullptr; ctx->sort_counts = nullptr; ctx->sort_starts = nullptr; ctx->sort_capacity = 0; ctx->sort_seeded = false;
		const size_t nbk = (size_t)nh_bucket_sort_max_buckets(ctx, pair_cap) + 1u;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sort_splitters, sizeof(uint64_t) * nbk));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sort_counts, sizeof(uint32_t) * nbk));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sort_starts, sizeof(uint32_t) * nbk));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->sort_counts, 0, sizeof(uint32_t) * nbk, s));
		// sort re-use: the record keys by record position (k_narrowphase compares and rewrites them every step) and the tag order of the last sort
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sort_keys_by_position, sizeof(uint64_t) * (size_t)pair_cap));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sort_sorted_keys, sizeof(uint64_t) * (size_t)pair_cap));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sort_sorted_idx, sizeof(uint32_t) * (size_t)pair_cap));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->sort_keys_by_position, 0, sizeof(uint64_t) * (size_t)pair_cap, s));
		drop_sort_order = true;
		ctx->sort_capacity = pair_cap;
	}
	// Kept pair list or direct search?  Keeping pairs pays while the list outlives a few steps.  In a world where somebody leaves its box every step
	// (millions of bodies of which a few always roll) every step rebuilds and the filter pass only adds to it: after eight rebuilds in a row (the count
	// rides back with every step's round trip) the next 64 steps search the grid with the exact boxes and write the pairs directly, then the list gets
	// another chance.
	{
		const uint32_t seen = ctx->h_state->fat_rebuilds;
		if (ctx->bp_direct_left > 0) { if (--ctx->bp_direct_left == 0) ctx->bp_rebuild_streak = 0; }
		else {
			ctx->bp_rebuild_streak = seen != ctx->bp_seen_rebuilds ? ctx->bp_rebuild_streak + 1u : 0u;
			if (ctx->bp_rebuild_streak >= 8u) ctx->bp_direct_left = 64;
		}
		ctx->bp_seen_rebuilds = seen;
	}
	const bool bp_direct = ctx->env_no_fat || ctx->bp_direct_left > 0;
	// library-owned: the kept pair list of the broadphase and the two buffers of inflated boxes (they persist from one nh_collide to the next)
	bool drop_kept = ctx->fat_nbox != nbox || ctx->fat_nsph != nsph;
	// (the kept list holds pairs of INFLATED boxes: more than overlap exactly -- twice the pair capacity, at least 4096)
	const uint32_t want_kept = pair_cap > 0x7fffffffu / 2u ? 0xfffffffeu : (2u * pair_cap > 4096u ? 2u * pair_cap : 4096u);
	if (ctx->fat_pair_capacity != want_kept) {
		if (ctx->fat_pairs) NH_HIP_CHECK(ctx, hipFree(ctx->fat_pairs));
		ctx->fat_pairs = nullptr; ctx->fat_pair_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fat_pairs, sizeof(uint2) * (size_t)want_kept));
		ctx->fat_pair_capacity = want_kept;
		drop_kept = true;
	}
	if (ctx->fat_collider_capacity < C) {
		if (ctx->fat_box) NH_HIP_CHECK(ctx, hipFree(ctx->fat_box));
		ctx->fat_box = nullptr; ctx->fat_collider_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fat_box, sizeof(float4) * 2u * (size_t)C));
		ctx->fat_collider_capacity = C;
		drop_kept = true;
	}
	ctx->fat_nbox = nbox; ctx->fat_nsph = nsph;
	// library-owned as well: the grid of the last rebuild (cell-sorted boxes, cell starts, large list) and the book-keeping of re-insertions -- a
	// collider that leaves its box is looked up in THAT grid steps later, so none of it may live in the caller's temporary arena
	if (ctx->grid_cells < max_cells || ctx->grid_collider_capacity < ctx->fat_collider_capacity) {
		if (ctx->grid_sbox) NH_HIP_CHECK(ctx, hipFree(ctx->grid_sbox));
		if (ctx->grid_cstart) NH_HIP_CHECK(ctx, hipFree(ctx->grid_cstart));
		if (ctx->grid_counts) NH_HIP_CHECK(ctx, hipFree(ctx->grid_counts));
		if (ctx->grid_large) NH_HIP_CHECK(ctx, hipFree(ctx->grid_large));
		if (ctx->fat_gen) NH_HIP_CHECK(ctx, hipFree(ctx->fat_gen));
		if (ctx->fat_esc_mark) NH_HIP_CHECK(ctx, hipFree(ctx->fat_esc_mark));
		if (ctx->fat_esc_list) NH_HIP_CHECK(ctx, hipFree(ctx->fat_esc_list));
		if (ctx->fat_moved_list) NH_HIP_CHECK(ctx, hipFree(ctx->fat_moved_list));
		ctx->grid_sbox = nullptr; ctx->grid_cstart = nullptr; ctx->grid_counts = nullptr; ctx->grid_large = nullptr; ctx->fat_gen = nullptr; ctx->fat_esc_mark = nullptr;
		ctx->fat_esc_list = nullptr; ctx->fat_moved_list = nullptr; ctx->grid_cells = 0; ctx->grid_collider_capacity = 0;
		const size_t cc = ctx->fat_collider_capacity;              // (>= C: set with the boxes above)
		const uint32_t cells = max_cells > ctx->grid_cells ? max_cells : ctx->grid_cells;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grid_sbox, sizeof(float4) * 2u * cc + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grid_cstart, sizeof(uint32_t) * ((size_t)cells + 8u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grid_counts, sizeof(uint32_t) * ((size_t)cells + 8u)));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->grid_counts, 0, sizeof(uint32_t) * ((size_t)cells + 8u), s));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->grid_large, sizeof(uint32_t) * cc + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fat_gen, cc + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fat_esc_mark, sizeof(uint32_t) * cc + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fat_esc_list, sizeof(uint32_t) * NH_ESC_MAX));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fat_moved_list, sizeof(uint32_t) * NH_MOVED_MAX));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->fat_esc_mark, 0, sizeof(uint32_t) * cc + 64u, s));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->fat_gen, 0, cc + 64u, s));
		ctx->grid_cells = cells; ctx->grid_collider_capacity = (uint32_t)cc;
		drop_kept = true;
	}
	// re-insertion of colliders that leave their boxes (k_reinsert): kept-list steps only, indices that leave room for the stamps
	const bool incremental = !bp_direct && !ctx->env_no_incremental && C < (1u << NH_GEN_SHIFT);
	// library-owned: contact storage by slot, the records, their place in the tag order, the slot cache; per body class / contact-list record / record (nh_internal.h)
	bool lay_fresh = false;
	if (ctx->lay_capacity != pair_cap || ctx->lay_contact_capacity != cap) {
		void** bufs[] = { (void**)&ctx->raw_data, (void**)&ctx->raw_feature, (void**)&ctx->rec, (void**)&ctx->lay_rank, (void**)&ctx->cnt_sorted, (void**)&ctx->start_sorted, (void**)&ctx->dense_slot,
		                  (void**)&ctx->sc_imp, (void**)&ctx->sc_feat, (void**)&ctx->sc_count, (void**)&ctx->sc_undo, (void**)&ctx->pair_mark, (void**)&ctx->pair_list, (void**)&ctx->exp_cnt, (void**)&ctx->exp_start, (void**)&ctx->exp_scan_tmp, (void**)&ctx->still_delta };
		for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
		ctx->lay_capacity = 0; ctx->lay_contact_capacity = 0;
		const size_t slots = (size_t)4 * pair_cap + cap, cache_slots = (size_t)5 * pair_cap + 64u;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->raw_data, sizeof(nh_Contact) * slots));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->raw_feature, sizeof(uint32_t) * slots + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->rec, sizeof(nh_Record) * (size_t)pair_cap + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->lay_rank, sizeof(uint32_t) * (size_t)pair_cap + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->cnt_sorted, sizeof(uint32_t) * ((size_t)pair_cap + 2u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->start_sorted, sizeof(uint32_t) * ((size_t)pair_cap + 2u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->dense_slot, sizeof(uint32_t) * (size_t)cap + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sc_imp, sizeof(float4) * cache_slots));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sc_feat, sizeof(uint32_t) * cache_slots));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sc_count, sizeof(uint32_t) * (size_t)pair_cap + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->sc_undo, sizeof(uint64_t) * (size_t)pair_cap + 64u));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->sc_undo, 0, sizeof(uint64_t) * (size_t)pair_cap + 64u, s));
		ctx->still.undo_dirty = false;
		ctx->pair_list_capacity = pair_cap / 8u + 1024u;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->pair_mark, sizeof(uint32_t) * (size_t)pair_cap + 64u));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->pair_mark, 0, sizeof(uint32_t) * (size_t)pair_cap + 64u, s));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->pair_list, sizeof(uint32_t) * (size_t)ctx->pair_list_capacity));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->exp_cnt, sizeof(uint32_t) * ((size_t)pair_cap + 2u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->exp_start, sizeof(uint32_t) * ((size_t)pair_cap + 2u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->exp_scan_tmp, sizeof(uint32_t) * (2u * NH_SORT_GRID + 64u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->still_delta, sizeof(int2) * 2u * NH_DELTA_MAX));
		ctx->raw_slots = (uint32_t)slots;
		ctx->lay_capacity = pair_cap; ctx->lay_contact_capacity = cap; lay_fresh = true;
		ctx->fused.map_valid = false;
	}
	if (ctx->lay_body_capacity < B) {
		void** bufs[] = { (void**)&ctx->lay_class, (void**)&ctx->lay_simple, (void**)&ctx->body_rec, (void**)&ctx->body_pos };
		for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
		ctx->lay_body_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->lay_class, (size_t)B + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->lay_simple, sizeof(uint2) * (size_t)B + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->body_rec, sizeof(uint32_t) * (size_t)B + 64u));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->body_pos, sizeof(uint32_t) * (size_t)B + 64u));
		ctx->lay_body_capacity = B; lay_fresh = true;
	}
	nh_Contact* raw_data = ctx->raw_data; uint32_t* raw_feature = ctx->raw_feature; nh_Record* rec = ctx->rec; uint32_t* rec_counts = ctx->cnt_sorted;
	if (lay_fresh) { ctx->still.slots_current = false; ctx->still.ok_next = false; }
	{
		nh_StillStep& ss = ctx->still;
		if (!same_colliders(*colliders, ss.lay_colliders)) ss.ahead_world_bad = false;          // (other colliders: whether bodies carry several of them is found out again)
		ss.lay_bodies = *bodies; ss.lay_colliders = *colliders; ss.lay_contacts = *contacts; ss.lay_active = active_bodies->indices; ss.lay_active_capacity = active_bodies->capacity;
		ss.appended = false;
		// ---- the still step: three launches instead of the thirty below ----
		// SLEEPERS form (nh_internal.h): somebody may be asleep (the host cannot rule it out) -- bodies asleep in sets of their own are the step's business, without
		// user connections (they join sets: nudge.cpp:3511-3575) and with the per-body notes the views are made of
		const bool may_sleep = !no_islands;
		const bool sleepers_ok = !ss.no_local && !(body_connections && body_connections->count);
		if (may_sleep && sleepers_ok && ctx->still_awake_capacity < B) {
			if (ctx->still_awake) NH_HIP_CHECK(ctx, hipFree(ctx->still_awake));
			ctx->still_awake = nullptr; ctx->still_awake_capacity = 0;
			NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->still_awake, (size_t)B + 64u));
			ctx->still_awake_capacity = B;
		}
		const bool still = want_still && (no_islands || sleepers_ok) && !bp_direct && !drop_kept && !drop_sort_order && !lay_fresh && !ctx->env_sort_classic && ctx->sort_seeded && !ctx->env_sort_radix &&
		                   !ctx->env_no_sort_reuse && C != 0u;
		if (still) {
			const bool sleepers = may_sleep;
			ss.sleepers = sleepers;
			nh_DevState* st = ctx->d_state;
			// MOVERS form (nh_internal.h, "LOCAL speculation"): while somebody has been leaving its inflated box lately, the step re-boxes its leavers and re-inserts them
			// into the kept list itself (one more launch, which leaves at once in a step nobody moves in); otherwise a leaver fails the step, and the full step that
			// re-inserts it switches this on
			const bool movers = incremental && !ss.no_local && ss.movers_left != 0u;
			// (the colliders' world transforms, boxes and tags of a still step live in buffers of the library's own: nh_internal.h, xform ahead)
			if (ctx->own_capacity < C) {
				void** bufs[] = { (void**)&ctx->own_xf, (void**)&ctx->own_aabb_min, (void**)&ctx->own_aabb_max, (void**)&ctx->own_ctag };
				for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
				ctx->own_capacity = 0;
				NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->own_xf, sizeof(nh_xform) * (size_t)C + 64u));
				NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->own_aabb_min, sizeof(float4) * (size_t)C + 64u));
				NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->own_aabb_max, sizeof(float4) * (size_t)C + 64u));
				NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->own_ctag, sizeof(uint32_t) * (size_t)C + 64u));
				ctx->own_capacity = C; ss.ahead_ready = false;
			}
			xf = ctx->own_xf; aabb_min = ctx->own_aabb_min; aabb_max = ctx->own_aabb_max; ctag = ctx->own_ctag;
			// XFORM AHEAD: the solver of the step before this one -- same nh_step call, plain form -- has written all of that for the dynamic bodies' colliders; this step
			// starts at the narrowphase
			// SLEEPERS AHEAD (nh_internal.h): the sleeping set has stood still for a few confirmed steps -- the awake bodies' lanes work ahead as in the plain form
			if (ss.sleep_backoff) ss.sleep_backoff--;
			// (the step in which the host can no longer rule sleepers out is the one before the first of them falls asleep: what stood still until now was a world awake)
			if (sleepers && !ss.was_sleepers) ss.sleep_stable = 0u;
			ss.was_sleepers = sleepers;
			const bool sleep_stable = sleepers && !ss.no_sleeper_ahead && !ss.no_pair && ss.sleep_stable >= 4u && ss.sleep_backoff == 0u && !ctx->step_hook && ss.pipelined && !ctx->fused.enabled && C <= (2u << 20);
			const bool plain = !movers && (!sleepers || sleep_stable) && !ss.appended_pairs && (nbox >= 2 || nsph != 0u);
			// (sleepers: the whole pair step or the three kernels -- a step that starts at the narrowphase would need its sleepers form there)
			const bool ahead_step = plain && ss.ahead_ready && !ss.no_ahead && (!sleepers || ss.pair_ready);
			ss.ahead_ready = false; ss.ahead_plain = plain;
			// THE ONE-KERNEL STILL STEP (nh_fused.hip): a plain step that starts at the narrowphase, inside nh_step with late verdicts, and is not the call's last -- nothing is
			// launched here; the solver call launches k_still_fused, which evaluates every kept pair in the lane that solves its body
			const bool fused = ahead_step && !sleepers && ss.pipelined && ss.more_steps && nh_fused_wanted(ctx, B, C, ss.steps_left);
			ss.fused = fused;
			if (movers) ctx->fused.map_valid = false;          // (re-stamped and appended pairs: the position table describes another kept list)
			if (!fused && ctx->fused.cur != 0u) {
				// the bodies live in the shadow arrays (an odd run of fused steps) and this step works in place: the run's last verdict first -- a failed step's output
				// must not be copied over the state it has to be replayed from -- then the bodies go home
				if (ss.verdict.pending && nh_still_verdict_now(ctx)) return NH_INTERNAL_STILL_FAILED;
				{ int rc = nh_fused_come_home(ctx); if (rc) return rc; }
			}
			// PAIR AHEAD (nh_internal.h): the last step's solver lanes evaluated this step's pairs too -- this step starts at the solver, behind one wave of bookkeeping
			const bool pair_step = ahead_step && ss.pair_ready && !fused && !ss.no_pair;
			ss.pair_ready = false; ss.pair_step = pair_step; ss.early_verdict = false;
			if (pair_step) ss.pair_steps++;
			if (pair_step && sleepers && ss.sleep_run < 0xffffu) ss.sleep_run++;
			if (!ahead_step) ctx->fused.statics_copied = false;          // (k_xform<true> rewrites the current set of collider buffers, the static world's entries included)
			if (ahead_step) ss.ahead_steps++;
			else
			NH_LAUNCH(ctx, "xform_still", (k_xform<true>), nh_grid_for(C > B ? C : B, 1024, 256), 1024, st, bodies->transforms,
			          colliders->boxes.transforms, colliders->boxes.data, colliders->boxes.tags, nbox,
			          colliders->spheres.transforms, colliders->spheres.data, colliders->spheres.tags, nsph,
			          xf, aabb_min, aabb_max, ctag, (uint4*)nullptr, 0u, ctx->fat_box, 0u,
			          movers ? ctx->fat_gen : (uint8_t*)nullptr, ctx->fat_esc_mark, ctx->fat_esc_list, ctx->fat_moved_list, ctx->collide_seq, bodies->idle_counters, B, ctx->step_parity,
			          sleepers ? ctx->still_awake : (uint8_t*)nullptr, (sleepers && ss.substep > 0u && !ctx->step_hook && !ss.no_sleeper_skip) ? 1u : 0u);          // (the narrowphase's `sleeper_skip` below)
			// (sleepers ahead: the static world's share of the scene bounds holds the sleepers' -- a step with a k_xform<true> and a narrowphase of its own may be the one in
			// which somebody falls asleep without failing anything, so every such step of a world with sleepers takes the map again before lanes work ahead on it)
			if (sleepers && !ahead_step) ss.ahead_map_ok = false;
			if (!ahead_step && plain && ss.more_steps && !ss.no_ahead && !ss.ahead_world_bad && !ss.ahead_map_ok && bodies->idle_counters) {
				// (once per nh_step call: which collider each body's lane stands in for, and the static world's share of the scene bounds)
				if (ctx->body_col_capacity < B) {
					if (ctx->body_col) NH_HIP_CHECK(ctx, hipFree(ctx->body_col));
					ctx->body_col = nullptr; ctx->body_col_capacity = 0;
					NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->body_col, sizeof(uint32_t) * (size_t)B + 64u));
					ctx->body_col_capacity = B;
				}
				NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->body_col, 0xff, sizeof(uint32_t) * (size_t)B, ctx->stream));
				NH_LAUNCH(ctx, "ahead_map", k_ahead_map, nh_grid_for(C, 256, 2048), 256, st, aabb_min, C, ctx->body_col);
				NH_LAUNCH(ctx, "ahead_check", k_ahead_check, nh_grid_for(C, 256, 1024), 256, st, aabb_min, C, ctx->body_col, sleepers ? bodies->idle_counters : (const uint8_t*)nullptr);
				if (ss.pair_owned_seq == 0u && !ss.no_pair && ctx->pair_mark) {
					// (pair ahead: the kept pairs that are no body's own, listed once per layout)
					NH_HIP_CHECK(ctx, hipMemsetAsync(&st->pair_unowned, 0, sizeof(uint32_t), ctx->stream));
					NH_LAUNCH(ctx, "pair_mark", k_pair_mark, nh_grid_for(B, 256, 2048), 256, ctx->body_rec, (ctx->halo_ghost_first && ctx->halo_ghost_first < B) ? ctx->halo_ghost_first : B, ctx->pair_mark, ctx->collide_seq);
					NH_LAUNCH(ctx, "pair_list", k_pair_list, nh_grid_for(pair_cap, 256, 2048), 256, st, ctx->pair_mark, ctx->collide_seq, ctx->pair_list, ctx->pair_list_capacity, sleepers ? ctx->rec : (const nh_Record*)nullptr);
					ss.pair_owned_seq = ctx->collide_seq;
				}
				ss.ahead_map_ok = true;
			}
			if (movers)
				NH_LAUNCH(ctx, "reinsert_still", k_reinsert, 256, 256, st, nbox, ctx->grid_sbox, ctx->grid_cstart, ctx->grid_large, ctx->fat_box, ctx->fat_gen, ctx->fat_esc_mark, ctx->fat_esc_list,
				          ctx->fat_moved_list, ctx->collide_seq, ctx->fat_pairs, ctx->fat_pair_capacity, 1u, ctx->step_parity, C, ctx->body_rec, pair_cap);
			const uint8_t* gen = incremental ? ctx->fat_gen : (const uint8_t*)nullptr;
			// (a world dozing off changes thousands of contact counts per step: the scan, not the list -- but not in a pair step of a world whose sleeping set stands still,
			// sleepers ahead: its lanes listed their changes like a plain world's, and the two launches that leave at once are 4 us of an 80 us step)
			const bool delta_scan = C > (2u << 20) || (sleepers && !pair_step);
			uint32_t* const sc_count = sleepers ? ctx->sc_count : (uint32_t*)nullptr;
			if (sleepers) ss.undo_dirty = true;
			// (the LOCAL form of the narrowphase -- appended pairs, sleeping bodies -- only where it can matter: this step or an earlier one since the layout ran in movers
			// form, or somebody may be asleep)
			if (movers) ss.appended_pairs = true;
			const bool local = sleepers || ss.appended_pairs;
			// (sleepers form inside an nh_step call, not its first sub-step: what was asleep when the sub-step before this one looked and is asleep now has not moved)
			const uint32_t sleeper_skip = (sleepers && ss.substep > 0u && !ctx->step_hook && !ss.no_sleeper_skip) ? 1u : 0u;
#define NH_NARROW_STILL(SPHERES, NAME, GRIDMAX) do { if (local) NH_LAUNCH(ctx, NAME, (k_narrowphase<SPHERES, true, true>), nh_grid_for(pair_cap, 256, GRIDMAX), 256, st, ctx->fat_pairs, xf, ctag, \
				          colliders->boxes.data, colliders->spheres.data, nbox, raw_data, raw_feature, pair_cap, 0u, ctx->sort_keys_by_position, rec, (uint32_t*)nullptr, \
				          (const uint32_t*)nullptr, (const uint32_t*)nullptr, ctx->step_parity, 0u, ctx->fat_pair_capacity, aabb_min, aabb_max, gen, ctx->lay_rank, ctx->cnt_sorted, ctx->body_rec, ctx->collide_seq, ctx->still_delta, delta_scan ? 1u : 0u, sc_count, 0u, ctx->sc_undo, sleeper_skip); \
				else NH_LAUNCH(ctx, NAME, (k_narrowphase<SPHERES, true, false>), nh_grid_for(pair_cap, 256, GRIDMAX), 256, st, ctx->fat_pairs, xf, ctag, \
				          colliders->boxes.data, colliders->spheres.data, nbox, raw_data, raw_feature, pair_cap, 0u, ctx->sort_keys_by_position, rec, (uint32_t*)nullptr, \
				          (const uint32_t*)nullptr, (const uint32_t*)nullptr, ctx->step_parity, 0u, ctx->fat_pair_capacity, aabb_min, aabb_max, gen, ctx->lay_rank, ctx->cnt_sorted, ctx->body_rec, ctx->collide_seq, ctx->still_delta, delta_scan ? 1u : 0u, (uint32_t*)nullptr, ahead_step ? 1u : 0u); } while (0)
			if (pair_step) {
			NH_LAUNCH(ctx, "pair_begin", k_pair_begin, 32, 256, st, ctx->step_parity, ctx->collide_seq, ctx->pair_list, ctx->pair_list_capacity, ctx->fat_pairs, ctx->fat_pair_capacity, pair_cap, gen,
			                         xf, aabb_min, aabb_max, ctag, colliders->boxes.data, colliders->spheres.data, nbox, ctx->sort_keys_by_position, rec,
			                         ctx->lay_rank, ctx->cnt_sorted, ctx->body_rec, ctx->still_delta, raw_data, raw_feature, bodies->momentum, bodies->properties, delta_scan ? 1u : 0u, sleepers ? B : 0u);
			ss.early_verdict = false;
			if (ctx->step_hook && !ss.pipelined && ss.h_ring[0] && !ctx->timing) {
				NH_HIP_CHECK(ctx, hipMemcpyAsync(ss.h_ring[0], st, NH_COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
				NH_HIP_CHECK(ctx, hipEventRecord(ss.ev_ring[0], ctx->stream));
				ss.early_verdict = true;
			}
			}
			else if (!fused) {
			if (nbox >= 2) NH_NARROW_STILL(false, "narrowphase_still", 4096);
			if (nsph) NH_NARROW_STILL(true, "narrowphase_sph_still", 8192);
			}
			// every record's first contact in the dense (tag-ordered) list = the scan of the counts in tag order: kept from step to step, shifted by the listed changes
			// (the solver keeps the total).  More pairs changing their count than the list holds: a world of up to two million colliders fails the step (a handful change
			// per million and step at rest); a larger one runs the scan instead -- two launches that leave at once otherwise, nothing next to its step
			if (delta_scan && !fused) nh_scan_u32(ctx, ctx->cnt_sorted, ctx->start_sorted, &st->records, 1, scan_tmp, nullptr, &st->delta_overflow[ctx->step_parity]);
			ss.active = true; ss.resolved = false; ss.launched++;
			ctx->gravity_may_overlap = false;
			ctx->after_collide = true;
			return NH_OK;
		}
	}
	// a full step.  (nh_step: a still step whose verdict has not been looked at yet must have happened before anything is built on it)
	if (ctx->still.verdict.pending && nh_still_verdict_now(ctx)) return NH_INTERNAL_STILL_FAILED;
	ctx->still.fused = false; ctx->fused.map_valid = false; ctx->fused.statics_copied = false;
	ctx->still.pair_ready = false; ctx->still.pair_step = false; ctx->still.early_verdict = false; ctx->still.pair_world_bad = false; ctx->still.pair_world_ok = false; ctx->still.pair_owned_seq = 0u;          // (another layout: whether every kept pair is some body's own is found out again)
	{ int rc = nh_fused_come_home(ctx); if (rc) return rc; }          // (every fused step before this one is confirmed: the bodies it left in the shadow arrays go home)
	// The solver reads the caller's cache arrays and this nh_collide lays the dense contact list out -- whatever still steps kept by slot goes home first
	{ int rc = nh_still_export_cache(ctx); if (rc) return rc; }
	ctx->still.ahead_ready = false; ctx->still.ahead_plain = false;
	if (!no_islands) ctx->still.ahead_map_ok = false;          // (sleepers ahead: the static world's share of the scene bounds holds the sleepers' -- and who sleeps may change in a full step)
	ctx->still.contacts_stale = false; ctx->still.slots_current = false; ctx->still.views_sleepers = false; ctx->still.appended_pairs = false;          // (a full step writes the contact list, the sleeping pairs and the active list itself)
	NH_LAUNCH(ctx, "collide_begin", k_collide_begin, begin_grid, 256, st, C, B, parent, set_active, ctx->deg, bodies->idle_counters, ctx->step_parity, coarse_parent, coarse_active, no_islands ? 1u : 0u, block_top,
	          drop_kept ? 1u : 0u, drop_sort_order ? 1u : 0u);
	const bool seeded_sort = !ctx->env_sort_classic && ctx->sort_seeded && !ctx->env_sort_radix;      // (NH_SORT_RADIX=1: radix passes every step)
	uint64_t* rec_key_a = ctx->sort_keys_by_position;
	if (C) {
		NH_LAUNCH(ctx, "xform_aabb", (k_xform<false>), nh_grid_for(C, 1024, 256), 1024, st, bodies->transforms,
		          colliders->boxes.transforms, colliders->boxes.data, colliders->boxes.tags, nbox,
		          colliders->spheres.transforms, colliders->spheres.data, colliders->spheres.tags, nsph,
		          xf, aabb_min, aabb_max, ctag, (uint4*)nullptr, 0u, ctx->fat_box, bp_direct ? 1u : 0u,
		          incremental ? ctx->fat_gen : (uint8_t*)nullptr, ctx->fat_esc_mark, ctx->fat_esc_list, ctx->fat_moved_list, ctx->collide_seq);
		// (asleep steps, nh_internal.h: this step may turn out to be the second fixed point in a row -- its boxes and tags are kept while they are still where k_xform put them)
		if (ctx->asleep.streak >= 1u && !ctx->asleep.disabled) (void)asleep_snapshot(ctx);
		// Morton frame; and the decision whether the grid chain below runs this step or the kept pair list is re-used (its kernels are launched
		// either way: the decision is made on the device, and a kernel that leaves at once costs two microseconds)
		NH_LAUNCH(ctx, "grid_setup", k_grid_setup, 1, 64, st, C, max_cells, bp_direct ? 1u : 0u, ctx->fat_pair_capacity);
		float4* fbox = ctx->fat_box;
		float4* sbox = ctx->grid_sbox; uint32_t* large_list = ctx->grid_large; uint32_t* cell_start = ctx->grid_cstart; uint32_t* cstart = ctx->grid_counts;       // (cstart: the cell COUNTS of a rebuild, all zero between rebuilds)
		uint2* kept = ctx->fat_pairs;
		const uint32_t kept_cap = ctx->fat_pair_capacity;
		const bool measure_skip = ctx->env_measure_skip && ctx->collide_seq > 66u;       // NH_MEASURE_SKIP_CHAINS=1: MEASUREMENT ONLY (wrong as soon as anything moves): what the launches that leave at once cost
		if (!measure_skip) {
		NH_LAUNCH(ctx, "cell_keys", k_cell_keys, nh_grid_for(C, 256, 2048), 256, st, C, aabb_min, aabb_max, fbox, keys_a, vals_a, large_list, cstart, incremental ? ctx->fat_gen : (uint8_t*)nullptr);
		NH_LAUNCH(ctx, "cell_scan", k_cscan_sums, max_cells / CS_TILE, 256, st, cstart, cscan_sums);
		NH_LAUNCH(ctx, "cell_scan", k_cscan_final, max_cells / CS_TILE, 256, st, cstart, cscan_sums, max_cells, cell_start);
		NH_LAUNCH(ctx, "cell_scatter", k_cell_scatter, nh_grid_for(C, 256, 2048), 256, st, C, keys_a, vals_a, cell_start, fbox, keys_b, sbox);
		}
		// (KEPT: the chain fills the kept list, k_kept_filter makes this step's pairs of it; DIRECT: the chain writes this step's pairs itself)
		const bool many_lanes = ctx->env_fp_sub ? ctx->env_fp_sub > 1 : C <= 16384u;
#define NH_FIND_PAIRS(SUBL, DIR, OUT, OUTCAP) NH_LAUNCH(ctx, "find_pairs", (k_find_pairs<SUBL, DIR>), nh_grid_for((uint64_t)C * SUBL, 256, 4096), 256, st, C, nbox, keys_b, sbox, cell_start, large_list, fbox, OUT, OUTCAP, \
		          coarse_parent, ctx->step_parity, bodies->idle_counters)
		if (bp_direct) {
			if (many_lanes) NH_FIND_PAIRS(16, true, pairs, pair_cap); else NH_FIND_PAIRS(1, true, pairs, pair_cap);
			NH_LAUNCH(ctx, "large_pairs", (k_large_pairs<true>), LP_GRID, LP_THREADS, st, nbox, sbox, cell_start, large_list, fbox, pairs, pair_cap, coarse_parent, ctx->step_parity, bodies->idle_counters);
		} else {
			if (!measure_skip) {
			if (many_lanes) NH_FIND_PAIRS(16, false, kept, kept_cap); else NH_FIND_PAIRS(1, false, kept, kept_cap);
			NH_LAUNCH(ctx, "large_pairs", (k_large_pairs<false>), LP_GRID, LP_THREADS, st, nbox, sbox, cell_start, large_list, fbox, kept, kept_cap, coarse_parent, ctx->step_parity, bodies->idle_counters);
			}
			// colliders that left their boxes this step (a few: otherwise the chain above has run): new boxes, new pairs appended to the kept list
			if (incremental)
				NH_LAUNCH(ctx, "reinsert", k_reinsert, 256, 256, st, nbox, sbox, cell_start, large_list, fbox, ctx->fat_gen, ctx->fat_esc_mark, ctx->fat_esc_list, ctx->fat_moved_list,
				          ctx->collide_seq, kept, kept_cap);
			// this step's pairs: the kept ones whose exact AABBs overlap now (every step)
			NH_LAUNCH(ctx, "kept_filter", k_kept_filter, nh_grid_for(kept_cap, 256, 1024), 256, st, kept, kept_cap, nbox, aabb_min, aabb_max, pairs, pair_cap, coarse_parent, ctx->step_parity, bodies->idle_counters,
			          (seeded_sort && !ctx->env_no_sort_reuse) ? 1u : 0u, incremental ? ctx->fat_gen : (const uint8_t*)nullptr);
		}
		// coarse islands over the broadphase pairs (united where the pairs were emitted) + connections: only when some body is asleep (a device
		// flag: the kernels leave at once otherwise)
		if (!no_islands && body_connections && body_connections->count)
			NH_LAUNCH(ctx, "uf_union_connections", k_uf_union_connections, nh_grid_for(body_connections->count, 256, 512), 256, st, body_connections->data, body_connections->count, coarse_parent, ctx->step_parity, bodies->idle_counters);
		if (!no_islands) NH_LAUNCH(ctx, "coarse_flatten", k_uf_flatten, nh_grid_for(B, 256, 512), 256, st, ctx->step_parity, B, coarse_parent, bodies->idle_counters, coarse_active, coarse_root);
		// box-box pairs and pairs with a sphere are separate lists (emit_pair): one launch each, each running one kind of arithmetic.
		// (the host does not know the counts: a launch over a list that turns out empty costs a few microseconds, so the launch for a
		// shape the world does not contain is skipped outright)
		if (nbox >= 2)
			NH_LAUNCH(ctx, "narrowphase", (k_narrowphase<false, false>), nh_grid_for(pair_cap, 256, 4096), 256, st, pairs, xf, ctag,
			          colliders->boxes.data, colliders->spheres.data, nbox, raw_data, raw_feature, pair_cap, overflow_cap, rec_key_a, rec, rec_idx_a,
			          coarse_root, coarse_active, ctx->step_parity, ctx->first_ghost);
		if (nsph)
			NH_LAUNCH(ctx, "narrowphase_sph", (k_narrowphase<true, false>), nh_grid_for(pair_cap, 256, 8192), 256, st, pairs, xf, ctag,
			          colliders->boxes.data, colliders->spheres.data, nbox, raw_data, raw_feature, pair_cap, overflow_cap, rec_key_a, rec, rec_idx_a,
			          coarse_root, coarse_active, ctx->step_parity, ctx->first_ghost);
	}
	// islands over contact records (+ user connections): a chain of small kernels that the tag sort of the records does not depend
	// on -- it runs on the side stream while the caller's stream sorts (fork here, join before k_sorted_counts)
	// (only while the side chain is small: the radix passes spin-wait on each other and must not share the machine with a long
	// kernel -- with millions of records in one island the union-find alone takes a millisecond)
	const bool fork = !no_islands && ctx->side != nullptr && (seeded_sort || ctx->h_state->records <= 1500000u);
	hipStream_t main_stream = ctx->stream;
	if (fork) {
		NH_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, main_stream));
		NH_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
		ctx->stream = ctx->side;
	}
	if (!no_islands) {
		if (body_connections && body_connections->count)
			NH_LAUNCH(ctx, "uf_union_connections", k_uf_union_connections, nh_grid_for(body_connections->count, 256, 2048), 256, st, body_connections->data, body_connections->count, parent, ctx->step_parity, bodies->idle_counters);
		NH_LAUNCH(ctx, "uf_union_records", k_uf_union_records, nh_grid_for(pair_cap, 256, 2048), 256, st, rec, parent, ctx->step_parity, bodies->idle_counters);
		NH_LAUNCH(ctx, "uf_flatten", k_uf_flatten, nh_grid_for(B, 256, 2048), 256, st, ctx->step_parity, B, parent, bodies->idle_counters, set_active, root_of);
		NH_LAUNCH(ctx, "active_flags", k_active_flags, nh_grid_for(B, 256, 2048), 256, st, ctx->step_parity, B, root_of, set_active, flags);
		// number of bodies is known on the host: reuse the device-count scan with a constant count of 0 + extra
		nh_scan_u32(ctx, flags, flags, &st->pad0 /* always 0 */, B, scan_tmp, &st->active, &st->any_idle[ctx->step_parity]);    // (skipped on the device when nobody sleeps)
	}
	// (nobody asleep -- known on the device, or already on the host: the active list is 1 .. B-1, written without looking at any set)
	NH_LAUNCH(ctx, "active_write", k_active_write, nh_grid_for(B, 256, 2048), 256, st, B, root_of, set_active, flags, active_bodies->indices, active_bodies->capacity, ctx->step_parity, no_islands ? 1u : 0u, block_top, begin_grid);
	// (what a gravity call on the side stream waits for; NH_FLAG_FUSED_STEP never takes that road, and an event on the stream is a bubble of a few microseconds)
	if (ctx->side && !(ctx->flags & NH_FLAG_FUSED_STEP)) NH_HIP_CHECK(ctx, hipEventRecord(ctx->ev_active, ctx->stream));         // (ctx->stream is the side stream inside a fork)
	if (!no_islands) NH_LAUNCH(ctx, "filter_records", k_filter_records, nh_grid_for(pair_cap, 256, 2048), 256, st, rec, root_of, set_active, ctx->step_parity);
	if (fork) {
		ctx->stream = main_stream;
		NH_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join, ctx->side));
	}
	// tag order: sort records by (b_tag, a_tag), then lay contacts out pair by pair
	int key_bits = (int)ctx->tag_bits;
	if (key_bits < 1) key_bits = 1;
	if (key_bits > 32) key_bits = 32;
	// keys are a | b<<32: sort the low field, then the high field (stable) -- dropped records carry ~0 and end up last
	// Sort re-use.  The record keys live in a context buffer indexed by record position; k_narrowphase has compared every key with what that position
	// held last step.  When nothing differs (and the record count is the same) the order the last sort left in the context's buffers still holds: the four
	// kernels below leave at once.  That happens whenever k_kept_filter wrote the pairs in place and no pair changed its key -- a landed scene, step after step.
	const uint64_t* sorted_keys = rec_key_a;
	const uint32_t* sorted_idx = rec_idx_a;
	if (seeded_sort) {
		if (!(ctx->env_measure_skip && ctx->collide_seq > 66u))
		nh_bucket_sort_u64_u32(ctx, rec_key_a, rec_key_b, rec_idx_a, rec_idx_b, pair_cap, sort_place, key_bits, ctx->sort_sorted_keys, ctx->sort_sorted_idx);
		sorted_keys = ctx->sort_sorted_keys; sorted_idx = ctx->sort_sorted_idx;
	} else if (onesweep) {
		if (nh_onesweep_u64_u32_two_fields(ctx, rec_key_a, rec_key_b, rec_idx_a, rec_idx_b, &st->records, pair_cap, ctx->h_state->records, hist, key_bits)) {
			uint64_t* t = rec_key_a; rec_key_a = rec_key_b; rec_key_b = t; uint32_t* u = rec_idx_a; rec_idx_a = rec_idx_b; rec_idx_b = u;
		}
	} else {
		if (nh_sort_u64_u32(ctx, rec_key_a, rec_key_b, rec_idx_a, rec_idx_b, &st->records, hist, 0, ((key_bits + 7) / 8) * 8)) {
			uint64_t* t = rec_key_a; rec_key_a = rec_key_b; rec_key_b = t; uint32_t* u = rec_idx_a; rec_idx_a = rec_idx_b; rec_idx_b = u;
		}
		if (nh_sort_u64_u32(ctx, rec_key_a, rec_key_b, rec_idx_a, rec_idx_b, &st->records, hist, 32, 32 + ((key_bits + 7) / 8) * 8)) {
			uint64_t* t = rec_key_a; rec_key_a = rec_key_b; rec_key_b = t; uint32_t* u = rec_idx_a; rec_idx_a = rec_idx_b; rec_idx_b = u;
		}
	}
	// (dropped records carry count 0, so where they land in the order is irrelevant)
	if (!seeded_sort) { sorted_keys = rec_key_a; sorted_idx = rec_idx_a; }         // (the radix passes leave the result in whichever buffer they ended in)
	if (onesweep && !seeded_sort) { nh_bucket_sort_seed(ctx, rec_key_a, pair_cap); ctx->sort_seeded = true; }
	if (fork) NH_HIP_CHECK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
	NH_LAUNCH(ctx, "sorted_counts", k_sorted_counts, nh_grid_for(pair_cap, 256, 2048), 256, st, sorted_idx, rec, rec_counts, sleep_flags, seeded_sort ? 1u : 0u, ctx->lay_rank);
	// contact starts and sleeping-pair slots: two scans over the sorted records in one pair of launches
	nh_scan2_u32(ctx, rec_counts, ctx->start_sorted, &st->contacts, sleep_flags, sleep_flags, &st->sleeping, &st->records, 1, scan_tmp);
	NH_LAUNCH(ctx, "gather_contacts", k_gather_contacts, nh_grid_for(pair_cap, 256, 8192), 256, st, sorted_idx, rec, sorted_keys, ctx->start_sorted, raw_data, raw_feature,
	          contacts->data, contacts->bodies, contacts->tags, contacts->features, cap, pair_cap, sleep_flags, contacts->sleeping_pairs, ctx->deg, B,
	          ctx->dense_slot, seeded_sort ? 1u : 0u, 0u);

	if (ctx->flags & NH_FLAG_SYNC_COUNTS) {
		nh_Counts c;
		int rc = nh_read_counts(ctx, &c);
		if (rc) return rc;
		contacts->count = c.contacts;
		contacts->sleeping_count = c.sleeping_pairs;
		active_bodies->count = c.active_bodies;
		if (c.error) return (int)c.error;
	}
	ctx->gravity_may_overlap = ctx->side != nullptr;       // until any other entry point enqueues work
	ctx->after_collide = true;
	return NH_OK;
}

// ---- contacts appended by the caller after nh_collide -----------------------------------------------------------------------------------------------
// The reference lets its caller append contacts between collide() and read_cached_impulses() ("Custom contacts can be added here",
// example/main.cpp:287): they simply take part in the tag sort of read_cached_impulses (nudge.cpp:4027-4044) and everything downstream walks the
// sorted order.  Here the list nh_collide returns is already IN tag order and carries per-body bookkeeping counted while it was laid out (degrees,
// pair counts, first contact), so appended contacts are merged into that order and the bookkeeping is counted again -- a slow path, paid only by
// the steps that use it: one pass over all contacts.
__device__ __forceinline__ bool app_less(uint64_t ta, uint32_t fa, uint64_t tb, uint32_t fb) { return ta < tb || (ta == tb && fa < fb); }

// rank of every appended contact among the appended ones (ties: position), and the sorted keys
// (every k_app_* kernel leaves at once when the merged list would not fit the caller's arrays: k_app_count reports NH_ERR_CONTACT_CAPACITY and nothing is written out of bounds)
__global__ __launch_bounds__(256) void k_app_rank(const nh_DevState* __restrict__ st, uint32_t extra, const uint64_t* __restrict__ tags, const uint32_t* __restrict__ features,
                                                  uint32_t* __restrict__ rank, uint64_t* __restrict__ skey, uint32_t* __restrict__ sfeat, uint32_t capacity) {
	const uint32_t K = st->contacts;
	if ((uint64_t)K + extra > capacity) return;
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < extra; j += gridDim.x * blockDim.x) {
		const uint64_t t = tags[K + j]; const uint32_t f = features[K + j];
		uint32_t r = 0;
		for (uint32_t i = 0; i < extra; ++i) {
			const uint64_t ti = tags[K + i]; const uint32_t fi = features[K + i];
			r += (app_less(ti, fi, t, f) || (ti == t && fi == f && i < j)) ? 1u : 0u;
		}
		rank[j] = r; skey[r] = t; sfeat[r] = f;
	}
}

// new position of every contact: an old one moves up by the appended ones that sort before it, an appended one lands behind the old ones with a key <= its own
__global__ __launch_bounds__(256) void k_app_positions(const nh_DevState* __restrict__ st, uint32_t extra, const uint64_t* __restrict__ tags, const uint32_t* __restrict__ features,
                                                       const uint32_t* __restrict__ rank, const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sfeat, uint32_t* __restrict__ pos, uint32_t capacity) {
	const uint32_t K = st->contacts;
	if ((uint64_t)K + extra > capacity) return;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < K + extra; i += gridDim.x * blockDim.x) {
		const uint64_t t = tags[i]; const uint32_t f = features[i];
		if (i < K) {
			uint32_t lo = 0, hi = extra;                         // appended contacts with a key < (t, f)
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (app_less(skey[mid], sfeat[mid], t, f)) lo = mid + 1u; else hi = mid; }
			pos[i] = i + lo;
		} else {
			uint32_t lo = 0, hi = K;                             // old contacts with a key <= (t, f)
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (!app_less(t, f, tags[mid], features[mid])) lo = mid + 1u; else hi = mid; }
			pos[i] = rank[i - K] + lo;
		}
	}
}

__global__ __launch_bounds__(256) void k_app_copy(const nh_DevState* __restrict__ st, uint32_t extra, const nh_Contact* __restrict__ data, const nh_BodyPair* __restrict__ bodies,
                                                  const uint64_t* __restrict__ tags, const uint32_t* __restrict__ features, float4* __restrict__ t_data, nh_BodyPair* __restrict__ t_bodies,
                                                  uint64_t* __restrict__ t_tags, uint32_t* __restrict__ t_features, uint32_t capacity) {
	if ((uint64_t)st->contacts + extra > capacity) return;
	const uint32_t n = st->contacts + extra;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		t_data[2 * (size_t)i] = reinterpret_cast<const float4*>(data + i)[0]; t_data[2 * (size_t)i + 1] = reinterpret_cast<const float4*>(data + i)[1];
		t_bodies[i] = bodies[i]; t_tags[i] = tags[i]; t_features[i] = features[i];
	}
}

__global__ __launch_bounds__(256) void k_app_scatter(const nh_DevState* __restrict__ st, uint32_t extra, const uint32_t* __restrict__ pos, const float4* __restrict__ t_data,
                                                     const nh_BodyPair* __restrict__ t_bodies, const uint64_t* __restrict__ t_tags, const uint32_t* __restrict__ t_features,
                                                     nh_Contact* __restrict__ data, nh_BodyPair* __restrict__ bodies, uint64_t* __restrict__ tags, uint32_t* __restrict__ features, uint32_t capacity) {
	if ((uint64_t)st->contacts + extra > capacity) return;
	const uint32_t n = st->contacts + extra;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t p = pos[i];
		reinterpret_cast<float4*>(data + p)[0] = t_data[2 * (size_t)i]; reinterpret_cast<float4*>(data + p)[1] = t_data[2 * (size_t)i + 1];
		bodies[p] = t_bodies[i]; tags[p] = t_tags[i]; features[p] = t_features[i];
	}
}

// the per-body bookkeeping of k_gather_contacts, from the merged list: a run of equal tags AND equal bodies is one collider pair (appended contacts may share a tag --
// the reference only sorts by it -- while naming different bodies: each such stretch is a pair of its own for its two bodies)
__global__ __launch_bounds__(256) void k_app_recount(nh_DevState* __restrict__ st, uint32_t extra, const nh_BodyPair* __restrict__ bodies, const uint64_t* __restrict__ tags,
                                                     uint32_t* __restrict__ deg, uint32_t nbodies, uint32_t capacity) {
	if ((uint64_t)st->contacts + extra > capacity) return;
	unsigned long long* __restrict__ pair_counter = reinterpret_cast<unsigned long long*>(deg + 2u * NH_DEG_STRIDE(nbodies));
	uint32_t* __restrict__ first_contact = deg + 4u * NH_DEG_STRIDE(nbodies);
	const uint32_t n = st->contacts + extra;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const nh_BodyPair p = bodies[i];
		if (i > 0 && tags[i - 1] == tags[i] && bodies[i - 1].a == p.a && bodies[i - 1].b == p.b) continue;          // not the head of its run
		const uint64_t t = tags[i];
		uint32_t len = 1;
		while (i + len < n && tags[i + len] == t && bodies[i + len].a == p.a && bodies[i + len].b == p.b) ++len;
		if (p.a) { atomicAdd(&pair_counter[p.a], (unsigned long long)len | ((unsigned long long)(p.b ? 0x10001u : 1u) << 32)); first_contact[p.a] = i | 0x80000000u; }
		if (p.b) { atomicAdd(&pair_counter[p.b], (unsigned long long)len | ((unsigned long long)(p.a ? 0x10001u : 1u) << 32)); first_contact[p.b] = i; }
	}
}

__global__ void k_app_count(nh_DevState* st, uint32_t extra, uint32_t capacity) {
	if (st->contacts + extra > capacity) st->error = NH_ERR_CONTACT_CAPACITY; else st->contacts += extra;
}

extern "C" int nh_append_contacts(nh_context* ctx, nh_ContactData* contacts, const nh_BodyData* bodies, uint32_t extra, uint32_t* positions, nh_Arena temporary) {
	if (!ctx || !contacts || !bodies) return NH_ERR_INVALID;
	if (!extra) return NH_OK;
	if (extra > 65536u || extra > contacts->capacity) return NH_ERR_INVALID;      // (the appended contacts are ranked against each other by comparison: a slow path for a handful)
	if (ctx->setup_seq == ctx->collide_seq || !ctx->deg) return NH_ERR_STALE_SETUP;      // after nh_collide, before nh_read_cached_impulses / nh_setup_contact_constraints
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	{ int rc = nh_still_sync_outputs(ctx); if (rc) return rc; }
	ctx->still.appended = true; ctx->still.ok_next = false;
	nh_DevState* st = ctx->d_state;
	const uint32_t cap = contacts->capacity, B = bodies->count;
	int err = NH_OK;
	uint32_t* rank = nh_arena_array<uint32_t>(&temporary, extra, &err);
	uint64_t* skey = nh_arena_array<uint64_t>(&temporary, extra, &err);
	uint32_t* sfeat = nh_arena_array<uint32_t>(&temporary, extra, &err);
	uint32_t* pos = positions ? positions : nh_arena_array<uint32_t>(&temporary, cap, &err);
	float4* t_data = nh_arena_array<float4>(&temporary, 2 * (size_t)cap, &err);
	nh_BodyPair* t_bodies = nh_arena_array<nh_BodyPair>(&temporary, cap, &err);
	uint64_t* t_tags = nh_arena_array<uint64_t>(&temporary, cap, &err);
	uint32_t* t_features = nh_arena_array<uint32_t>(&temporary, cap, &err);
	if (err) return err;
	if ((ctx->flags & NH_FLAG_SYNC_COUNTS) && (uint64_t)contacts->count + extra > cap) return NH_ERR_CONTACT_CAPACITY;
	NH_LAUNCH(ctx, "append_rank", k_app_rank, nh_grid_for(extra, 256, 1024), 256, st, extra, contacts->tags, contacts->features, rank, skey, sfeat, cap);
	NH_LAUNCH(ctx, "append_positions", k_app_positions, nh_grid_for(cap, 256, 2048), 256, st, extra, contacts->tags, contacts->features, rank, skey, sfeat, pos, cap);
	NH_LAUNCH(ctx, "append_copy", k_app_copy, nh_grid_for(cap, 256, 2048), 256, st, extra, contacts->data, contacts->bodies, contacts->tags, contacts->features, t_data, t_bodies, t_tags, t_features, cap);
	NH_LAUNCH(ctx, "append_scatter", k_app_scatter, nh_grid_for(cap, 256, 2048), 256, st, extra, pos, t_data, t_bodies, t_tags, t_features, contacts->data, contacts->bodies, contacts->tags, contacts->features, cap);
	// degrees, pair info, first contact: counted again
	NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->deg, 0, sizeof(uint32_t) * NH_DEG_WORDS(B), ctx->stream));
	NH_LAUNCH(ctx, "append_recount", k_app_recount, nh_grid_for(cap, 256, 2048), 256, st, extra, contacts->bodies, contacts->tags, ctx->deg, B, cap);
	NH_LAUNCH(ctx, "append_count", k_app_count, 1, 1, st, extra, cap);
	if (ctx->flags & NH_FLAG_SYNC_COUNTS) {
		nh_Counts c;
		int rc = nh_read_counts(ctx, &c);
		if (rc) return rc;
		if (c.error) return (int)c.error;
		contacts->count = c.contacts;
	}
	return NH_OK;
}
