// nh_solve.hip -- contact cache (read/write), constraint setup, the sequential-impulse solver, cached-impulse
// export, gravity/damping and the integrator.  Replaces reference nudge.cpp:4011-4926 and the caller-side
// loop example/main.cpp:290-305.
//
// Solver order.  The reference runs projected Gauss-Seidel over 8-wide batches produced by a sequential
// greedy scheduler (nudge.cpp:4206-4339); the only thing about that schedule that influences the result is,
// for every dynamic body, the ORDER in which its contacts are visited (contacts that share no dynamic body
// commute exactly; body 0 is static and never conflicts, nudge.cpp:4238-4240).  So every contact gets a
// "slot key" = the index of the reference batch it would be in -- either replayed exactly on the device
// (NH_FLAG_EXACT_ORDER) or by the closed form batch(i) = (i/128)*16 + i%16, which is what the scheduler
// produces whenever it meets no lane conflict (single-layer drop scenes) -- and two engines execute that
// order:
//   * one-body fast path: a body whose contacts are all against body 0 is an independent problem; one lane
//     owns it, keeps its velocity in registers and walks its contacts in slot order for all iterations;
//   * level schedule: remaining contacts get level = 1 + max(level of the previous contact on either body);
//     levels run one after the other, contacts inside a level are independent.  Executing levels in order
//     is exactly the sequential Gauss-Seidel sweep in slot order.
#include "nh_internal.h"
int nh_asleep_remember(nh_context* ctx);            // nh_collide.hip (asleep steps: nh_internal.h)
#include <type_traits>
#include <atomic>
#include <string.h>
#include "nh_solver.h"
#include "nh_narrowphase.h"          // (PAIR AHEAD: the still solver's lanes evaluate their bodies' own collider pairs for the next sub-step)

#define NH_NONE 0xFFFFFFFFu
// body classes: the one-body fast path by contact count (<= 4, <= 8, more) or the level-scheduled path
#define NH_CLS_NONE 0u
#define NH_CLS_STATIC4 1u
#define NH_CLS_STATIC8 2u
#define NH_CLS_STATICN 3u
#define NH_CLS_GENERAL 4u
#define NH_CLS_PENDING 0xFFu         // k_adj_simple could not settle it: k_adj_fill / k_adj_sort do
// one-body bodies classified by k_adj_sort (after the counts round trip): the kernel speculatively launched before it has not seen them
#define NH_CLS_STATIC4_LATE 5u
#define NH_CLS_STATIC8_LATE 6u
#define NH_UNSET 0xFFFFFFFFu

__device__ __forceinline__ bool tag_less(uint64_t ta, uint32_t fa, uint64_t tb, uint32_t fb) { return ta < tb || (ta == tb && fa < fb); }

// ---- gravity + damping over the active list (example/main.cpp:290-305) -------------------------------------
__global__ __launch_bounds__(256) void k_gravity(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ active, nh_BodyMomentum* __restrict__ momentum,
                                                 float gx_dt, float gy_dt, float gz_dt, float damping) {
	uint32_t n = st->active;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t i = active[k];
		nh_BodyMomentum m = momentum[i];
		nh_gravity_damping(m.velocity, m.angular_velocity, gx_dt, gy_dt, gz_dt, damping);
		momentum[i] = m;
	}
}

// ---- advance (nudge.cpp:4886-4926) ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_advance(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ active, nh_Transform* __restrict__ xf,
                                                 const nh_BodyMomentum* __restrict__ momentum, uint8_t* __restrict__ idle, float time_step) {
	uint32_t n = st->active;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t i = active[k];
		nh_Transform t = xf[i];
		nh_BodyMomentum m = momentum[i];
		idle[i] = nh_advance_body(t.position, t.rotation, m.velocity, m.angular_velocity, idle[i], time_step);
		xf[i] = t;
	}
}

// NH_FLAG_FUSED_STEP: the bodies the fused one-body solver owns (class STATIC4 while the static world is inert) get gravity / damping and their
// advance inside that kernel; these two take everybody else on the active list
__global__ __launch_bounds__(256) void k_gravity_rest(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ active, const uint8_t* __restrict__ body_class,
                                                      nh_BodyMomentum* __restrict__ momentum, float gx_dt, float gy_dt, float gz_dt, float damping) {
	const uint32_t n = st->active;
	const bool inert = st->static_inert != 0u;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t i = active[k];
		if (inert && body_class[i] == NH_CLS_STATIC4) continue;
		nh_BodyMomentum m = momentum[i];
		nh_gravity_damping(m.velocity, m.angular_velocity, gx_dt, gy_dt, gz_dt, damping);
		momentum[i] = m;
	}
}

__global__ __launch_bounds__(256) void k_advance_rest(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ active, const uint8_t* __restrict__ body_class,
                                                      nh_Transform* __restrict__ xf, const nh_BodyMomentum* __restrict__ momentum, uint8_t* __restrict__ idle, float time_step) {
	const uint32_t n = st->active;
	const bool inert = st->static_inert != 0u;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t i = active[k];
		if (inert && body_class[i] == NH_CLS_STATIC4) continue;
		nh_Transform t = xf[i];
		nh_BodyMomentum m = momentum[i];
		idle[i] = nh_advance_body(t.position, t.rotation, m.velocity, m.angular_velocity, idle[i], time_step);
		xf[i] = t;
	}
}

// ---- read_cached_impulses (nudge.cpp:4021-4108) ------------------------------------------------------------------
// contacts are already in tag order, so the reference's merge-join becomes one binary search per contact.
__global__ __launch_bounds__(256) void k_cache_lookup(const nh_DevState* __restrict__ st, const uint64_t* __restrict__ tags, const uint32_t* __restrict__ features,
                                                      const uint64_t* __restrict__ ctags, const uint32_t* __restrict__ cfeatures, const nh_CachedContactImpulse* __restrict__ cdata,
                                                      nh_CachedContactImpulse* __restrict__ out, const nh_BodyPair* __restrict__ bodies, const uint8_t* __restrict__ body_class) {
	uint32_t n = st->contacts, m = st->cache;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		if (body_class) {
			// only contacts of the >8-contact one-body class and of the level-scheduled path read their warm start from `out`; the
			// fused one-body solver looks its own up and may already have put its SOLVED impulses there
			const nh_BodyPair p = bodies[i];
			const uint32_t ca = p.a ? body_class[p.a] : 0u, cb = p.b ? body_class[p.b] : 0u;
			if (ca != NH_CLS_GENERAL && cb != NH_CLS_GENERAL && ca != NH_CLS_STATICN && cb != NH_CLS_STATICN) continue;
		}
		uint64_t t = tags[i]; uint32_t f = features[i];
		uint32_t lo = 0, hi = m;
		// steady state: the cache is last step's contact list, so the entry usually sits at the same index
		if (i < m && ctags[i] == t && cfeatures[i] == f) { lo = i; hi = i; }
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (tag_less(ctags[mid], cfeatures[mid], t, f)) lo = mid + 1; else hi = mid; }
		nh_CachedContactImpulse r = { { 0.0f, 0.0f, 0.0f }, 0.0f };
		if (lo < m && ctags[lo] == t && cfeatures[lo] == f) r = cdata[lo];
		out[i] = r;
	}
}

// ... over a LIST of contacts: the general ones of a world in which they are few (k_contact_class: `general_list`, st->general_contacts of them) -- two bodies touching among a
// million that rest on the ground cost the lookup 43 us of reading every contact's bodies and classes to find them
__global__ __launch_bounds__(256) void k_cache_lookup_listed(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ list, const uint64_t* __restrict__ tags, const uint32_t* __restrict__ features,
                                                             const uint64_t* __restrict__ ctags, const uint32_t* __restrict__ cfeatures, const nh_CachedContactImpulse* __restrict__ cdata,
                                                             nh_CachedContactImpulse* __restrict__ out) {
	const uint32_t n = st->contacts, m = st->cache, g = st->general_contacts;
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < g; j += gridDim.x * blockDim.x) {
		const uint32_t i = list[j];
		if (i >= n) continue;
		const uint64_t t = tags[i]; const uint32_t f = features[i];
		uint32_t lo = 0, hi = m;
		if (i < m && ctags[i] == t && cfeatures[i] == f) { lo = i; hi = i; }
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (tag_less(ctags[mid], cfeatures[mid], t, f)) lo = mid + 1; else hi = mid; }
		nh_CachedContactImpulse r = { { 0.0f, 0.0f, 0.0f }, 0.0f };
		if (lo < m && ctags[lo] == t && cfeatures[lo] == f) r = cdata[lo];
		out[i] = r;
	}
}

// cached impulses of sleeping pairs are kept aside (nudge.cpp:4064-4101)
__global__ __launch_bounds__(256) void k_cull_flags(const nh_DevState* __restrict__ st, const uint64_t* __restrict__ ctags, const uint64_t* __restrict__ sleeping, uint32_t* __restrict__ flags) {
	uint32_t m = st->cache, ns = st->sleeping;
	if (blockIdx.x == 0 && threadIdx.x == 0) flags[m] = 0;       // sentinel so that scan[m] = number of culled entries
	if (ns == 0) return;                                         // nothing sleeps: the scan and the write below are skipped on the device
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
		uint64_t t = ctags[j];
		uint32_t lo = 0, hi = ns;
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (sleeping[mid] < t) lo = mid + 1; else hi = mid; }
		flags[j] = (lo < ns && sleeping[lo] == t) ? 1u : 0u;
	}
}

__global__ __launch_bounds__(256) void k_cull_write(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ flags_in, const uint32_t* __restrict__ scan,
                                                    const uint64_t* __restrict__ ctags, const uint32_t* __restrict__ cfeatures, const nh_CachedContactImpulse* __restrict__ cdata,
                                                    uint64_t* __restrict__ otags, uint32_t* __restrict__ ofeatures, nh_CachedContactImpulse* __restrict__ odata) {
	uint32_t m = st->cache;
	if (st->sleeping == 0) return;
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
		// flags were scanned in place into `scan`; an entry is culled iff scan[j+1] != scan[j]
		uint32_t p = scan[j], q = scan[j + 1];
		if (q != p) { otags[p] = ctags[j]; ofeatures[p] = cfeatures[j]; odata[p] = cdata[j]; }
		(void)flags_in;
	}
}

// ---- write_cached_impulses (nudge.cpp:4110-4158): merge of two sorted runs by rank -------------------------------
__global__ __launch_bounds__(256) void k_write_cache(nh_DevState* __restrict__ st, const uint64_t* __restrict__ tags, const uint32_t* __restrict__ features, const nh_CachedContactImpulse* __restrict__ imp,
                                                     const uint64_t* __restrict__ ktags, const uint32_t* __restrict__ kfeatures, const nh_CachedContactImpulse* __restrict__ kdata,
                                                     uint64_t* __restrict__ otags, uint32_t* __restrict__ ofeatures, nh_CachedContactImpulse* __restrict__ odata, uint32_t capacity) {
	uint32_t n = st->contacts, m = st->culled;
	uint32_t total = n + m;
	if (total > capacity) { if (blockIdx.x == 0 && threadIdx.x == 0) { st->error = NH_ERR_CACHE_CAPACITY; st->cache = 0; } return; }
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
		if (i < n) {
			// contact i goes after every culled entry with key <= its key ("if (a < b) contact else culled")
			uint64_t t = tags[i]; uint32_t f = features[i];
			uint32_t lo = 0, hi = m;
			while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (!tag_less(t, f, ktags[mid], kfeatures[mid])) lo = mid + 1; else hi = mid; }
			uint32_t pos = i + lo;
			otags[pos] = t; ofeatures[pos] = f; odata[pos] = imp[i];
		} else {
			uint32_t j = i - n;
			uint64_t t = ktags[j]; uint32_t f = kfeatures[j];
			uint32_t lo = 0, hi = n;
			while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (tag_less(tags[mid], features[mid], t, f)) lo = mid + 1; else hi = mid; }
			uint32_t pos = j + lo;
			otags[pos] = t; ofeatures[pos] = f; odata[pos] = kdata[j];
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) st->cache = total;
}

// ---- setup: per-body adjacency in solver order -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zero_u32(uint32_t* __restrict__ p, uint32_t n) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

// Contacts of one collider pair are contiguous in tag order: the first lane of each run does the pair's bookkeeping
// (one atomic per pair and body instead of one per contact).
// (a run = equal tags AND equal bodies: contacts appended by the caller may share a tag between different body pairs, nh_append_contacts)
__device__ __forceinline__ uint32_t run_length(const uint64_t* __restrict__ tags, const nh_BodyPair* __restrict__ bodies, uint32_t i, uint32_t n) {
	const nh_BodyPair p = bodies[i];
	if (i > 0 && tags[i - 1] == tags[i] && bodies[i - 1].a == p.a && bodies[i - 1].b == p.b) return 0;           // not the head of its run
	uint64_t t = tags[i];
	uint32_t len = 1;
	while (i + len < n && tags[i + len] == t && bodies[i + len].a == p.a && bodies[i + len].b == p.b) ++len;
	return len;
}

// general CSR fill, only for the bodies k_adj_simple left pending (several pairs, or a dynamic partner)
__global__ __launch_bounds__(256) void k_adj_fill(const nh_DevState* __restrict__ st, const nh_BodyPair* __restrict__ bodies, const uint64_t* __restrict__ tags, const uint32_t* __restrict__ off,
                                                  uint32_t* __restrict__ cursor, uint32_t* __restrict__ adj, const uint8_t* __restrict__ body_class, uint32_t which) {
	// `which`: 3 = the lists of pending AND general bodies; 1 = pending bodies only (the general ones are left to the per-block colouring of the blocked
	// solver, which needs no lists); 2 = general bodies only (the second pass when that turned out not to apply)
	if (!st->has_pending) return;
	uint32_t n = st->contacts;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		uint32_t len = run_length(tags, bodies, i, n);
		if (!len) continue;
		nh_BodyPair p = bodies[i];
		const uint32_t ca = p.a ? body_class[p.a] : NH_CLS_NONE, cb = p.b ? body_class[p.b] : NH_CLS_NONE;      // (general bodies settled by k_adj_simple need their lists too)
		if ((ca == NH_CLS_PENDING && (which & 1u)) || (ca == NH_CLS_GENERAL && (which & 2u))) { uint32_t base = off[p.a] + atomicAdd(&cursor[p.a], len); for (uint32_t k = 0; k < len; ++k) adj[base + k] = i + k; }
		if ((cb == NH_CLS_PENDING && (which & 1u)) || (cb == NH_CLS_GENERAL && (which & 2u))) { uint32_t base = off[p.b] + atomicAdd(&cursor[p.b], len); for (uint32_t k = 0; k < len; ++k) adj[base + k] = i + k; }
	}
}

__device__ __forceinline__ uint32_t slot_key_of(uint32_t i, const uint32_t* __restrict__ slot_key) {
	// closed form of the reference's round-robin scheduler when it meets no lane conflict:
	// contact i -> bucket i%16, lane (i/16)%8; batches fill (and are emitted) bucket by bucket every 128 contacts.
	return slot_key ? slot_key[i] : ((i >> 7) * 16u + (i & 15u));
}

// One lane per body.  A body that sits in exactly ONE collider pair, with the static world, and has <= 8 contacts -- the box on the
// ground -- needs no CSR build: its contacts are first .. first+d-1 (a pair's contacts are adjacent in tag order), recorded here in slot
// order as ONE word pair the solver reads next to the body state: simple[x] = (first contact, count | 3-bit offsets in slot order << 4 |
// body-is-"a" << 28).  Degrees come straight from nh_collide's counters: no scan, no adjacency array on this path (k_adj_from_simple
// writes the CSR form on demand).  Everything else is left PENDING for the general k_adj_fill / k_adj_sort, which do not even start
// when nothing is pending.
#define NH_FIRST_IS_A 0x80000000u        // in first_contact[]: the body plays "a" in its last pair (k_gather_contacts)
__global__ __launch_bounds__(256) void k_adj_simple(nh_DevState* __restrict__ st, uint32_t nbodies, uint32_t* __restrict__ deg, const unsigned long long* __restrict__ pair_counter,
                                                    const uint32_t* __restrict__ first_contact, const uint32_t* __restrict__ slot_key,
                                                    uint8_t* __restrict__ body_class, const nh_BodyProperties* __restrict__ props, nh_BodyMomentum* __restrict__ momentum,
                                                    uint2* __restrict__ simple, uint32_t* __restrict__ body_rec, uint32_t* __restrict__ body_pos, const uint32_t* __restrict__ sorted_idx) {
	// body_rec / body_pos (nh_internal.h, contact storage by slot): for a body of the one-pair class the record its contacts come from (| role) and that record's
	// place in the tag order -- what a still step's solver lane starts from; NH_BODY_REC_NONE for a body without contacts
	bool other = false, unstable = false;
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < nbodies; x += gridDim.x * blockDim.x) {
		if (x == 0) {
			body_class[0] = 0;
			momentum[0].unused0 = props[0].mass_inverse;             // the reference stashes mass_inverse in unused0 of EVERY body (nudge.cpp:4198)
			const nh_BodyMomentum m0 = momentum[0];
			const nh_BodyProperties p0 = props[0];
			st->static_inert = nh_is_inert(m0.velocity, m0.angular_velocity, p0.inertia_inverse, p0.mass_inverse) ? 1u : 0u;
			deg[0] = 0u;
			continue;
		}
		const unsigned long long pc = pair_counter[x];
		const uint32_t d = (uint32_t)pc;
		deg[x] = d;                                 // (the degrees as an array of their own: what the CSR scan reads, ensure_csr)
		uint32_t cls = NH_CLS_NONE;
		if (d) {
			const uint32_t inf = (uint32_t)(pc >> 32);
			if ((inf & 0xFFFFu) == 1u && (inf >> 16) == 0u && d <= 8u) {
				const uint32_t fc = first_contact[x];
				const uint32_t f = fc & ~NH_FIRST_IS_A;
				// contacts f .. f+d-1 in slot order (insertion sort in registers: a 4-entry network for the usual box on the ground, 8 otherwise)
				auto emit = [&](auto tag) {
					constexpr int W = decltype(tag)::value;
					uint32_t c[W], k[W];
#pragma unroll
					for (int q = 0; q < W; ++q) { c[q] = (uint32_t)q < d ? f + q : 0xFFFFFFFFu; k[q] = (uint32_t)q < d ? slot_key_of(c[q], slot_key) : 0xFFFFFFFFu; }
#pragma unroll
					for (int q = 1; q < W; ++q) {
#pragma unroll
						for (int j = q; j > 0; --j) {
							bool sw = (k[j] < k[j - 1]) || (k[j] == k[j - 1] && c[j] < c[j - 1]);
							uint32_t tc = sw ? c[j - 1] : c[j], tk = sw ? k[j - 1] : k[j];
							c[j - 1] = sw ? c[j] : c[j - 1]; k[j - 1] = sw ? k[j] : k[j - 1];
							c[j] = tc; k[j] = tk;
						}
					}
					uint32_t perm = 0;
#pragma unroll
					for (int q = 0; q < W; ++q) if ((uint32_t)q < d) perm |= (c[q] - f) << (3 * q);
					simple[x] = make_uint2(f, d | (perm << 4) | ((fc & NH_FIRST_IS_A) ? (1u << 28) : 0u));
					if (body_rec) {
						const uint32_t pos = first_contact[x + NH_DEG_STRIDE(nbodies)];
						body_pos[x] = pos;
						body_rec[x] = (sorted_idx ? sorted_idx[pos] : NH_BODY_REC_NONE) | ((fc & NH_FIRST_IS_A) ? NH_BODY_REC_IS_A : 0u);
					}
				};
				if (d <= 4u) emit(std::integral_constant<int, 4>()); else emit(std::integral_constant<int, 8>());
				cls = d <= 4u ? NH_CLS_STATIC4 : NH_CLS_STATIC8;
				if (cls == NH_CLS_STATIC8) st->has_static8 = 1;
			} else if (!slot_key && (inf >> 16) != 0u) {
				// default (colour) order and a dynamic partner: a general body, settled here -- its adjacency list is filled like a pending body's
				// (k_adj_fill) but needs neither the per-body sort by slot key nor the predecessor links of the exact order (k_adj_sort walks and
				// sorts every list it classifies: 0.5 ms in a pit of 4 M spheres that are ALL of this kind)
				cls = NH_CLS_GENERAL;
				st->has_pending = 1;
				momentum[x].unused0 = props[x].mass_inverse;         // (nudge.cpp:4198: the level-scheduled sweeps read it from there)
			} else {
				cls = NH_CLS_PENDING;
				st->has_pending = 1;
			}
		} else {
			momentum[x].unused0 = props[x].mass_inverse;
			if (body_rec) { body_rec[x] = NH_BODY_REC_NONE; body_pos[x] = 0u; }
		}
		body_class[x] = (uint8_t)cls;
		other |= cls != NH_CLS_STATIC4;
		unstable |= cls != NH_CLS_STATIC4 && cls != NH_CLS_NONE;
	}
	if (__builtin_amdgcn_ballot_w64(unstable) != 0 && nh_lane() == 0) st->has_unstable = 1u;
	// (one plain store per wave that has such a body: no counting)
	if (__builtin_amdgcn_ballot_w64(other) != 0 && nh_lane() == 0) st->has_other = 1u;
}

// CSR form of the lists k_adj_simple recorded (only when a kernel that walks off[] / adj[] is about to run: several apply calls per
// setup, a static world that is not inert, bodies of the 5..8-contact class)
__global__ __launch_bounds__(256) void k_adj_from_simple(uint32_t nbodies, const uint8_t* __restrict__ body_class, const uint32_t* __restrict__ off,
                                                         const uint2* __restrict__ simple, uint32_t* __restrict__ adj) {
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < nbodies; x += gridDim.x * blockDim.x) {
		const uint32_t cls = body_class[x];
		if (cls != NH_CLS_STATIC4 && cls != NH_CLS_STATIC8) continue;
		const uint2 sp = simple[x];
		const uint32_t b = off[x], d = sp.y & 15u;
		for (uint32_t q = 0; q < d; ++q) adj[b + q] = sp.x + ((sp.y >> (4 + 3 * q)) & 7u);
	}
}

// one lane per body: sort its contact list by slot key, classify; bodies off the fast path also record per-contact predecessors
__global__ __launch_bounds__(256) void k_adj_sort(nh_DevState* __restrict__ st, uint32_t nbodies, const nh_BodyPair* __restrict__ bodies, const uint32_t* __restrict__ off,
                                                  uint32_t* __restrict__ adj, const uint32_t* __restrict__ slot_key, uint8_t* __restrict__ body_class,
                                                  uint32_t* __restrict__ pred_a, uint32_t* __restrict__ pred_b,
                                                  const nh_BodyProperties* __restrict__ props, nh_BodyMomentum* __restrict__ momentum) {
	if (!st->has_pending) return;
	uint32_t rounded = (nbodies + 63u) & ~63u;
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < rounded; x += gridDim.x * blockDim.x) {
		uint32_t cls = 0, d = 0;
		if (x >= 1 && x < nbodies && body_class[x] == NH_CLS_PENDING) {
			uint32_t b = off[x], e = off[x + 1];
			d = e - b;
			if (d) {
				bool static_only = true;
				if (d <= 8) {
					// short list: sort in registers (fully unrolled insertion sort), write back only what moved
					uint32_t c[8], k[8];
#pragma unroll
					for (int q = 0; q < 8; ++q) {
						c[q] = (uint32_t)q < d ? adj[b + q] : 0xFFFFFFFFu;
						k[q] = (uint32_t)q < d ? slot_key_of(c[q], slot_key) : 0xFFFFFFFFu;
					}
#pragma unroll
					for (int q = 1; q < 8; ++q) {
#pragma unroll
						for (int j = q; j > 0; --j) {
							bool sw = (k[j] < k[j - 1]) || (k[j] == k[j - 1] && c[j] < c[j - 1]);
							uint32_t tc = sw ? c[j - 1] : c[j], tk = sw ? k[j - 1] : k[j];
							c[j - 1] = sw ? c[j] : c[j - 1]; k[j - 1] = sw ? k[j] : k[j - 1];
							c[j] = tc; k[j] = tk;
						}
					}
#pragma unroll
					for (int q = 0; q < 8; ++q) {
						if ((uint32_t)q < d) {
							adj[b + q] = c[q];
							nh_BodyPair p = bodies[c[q]];
							if ((p.a == x ? p.b : p.a) != 0) static_only = false;
						}
					}
				} else {
					for (uint32_t q = b + 1; q < e; ++q) {
						uint32_t c = adj[q];
						uint32_t kc = slot_key_of(c, slot_key);
						uint32_t j = q;
						while (j > b) {
							uint32_t dd = adj[j - 1];
							uint32_t kd = slot_key_of(dd, slot_key);
							if (kd < kc || (kd == kc && dd < c)) break;
							adj[j] = dd; --j;
						}
						adj[j] = c;
					}
					for (uint32_t q = b; q < e; ++q) {
						nh_BodyPair p = bodies[adj[q]];
						if ((p.a == x ? p.b : p.a) != 0) static_only = false;
					}
				}
				if (!static_only) {
					uint32_t prev = NH_NONE;
					for (uint32_t q = b; q < e; ++q) {
						uint32_t c = adj[q];
						if (bodies[c].a == x) pred_a[c] = prev; else pred_b[c] = prev;
						prev = c;
					}
				}
				cls = static_only ? (d <= 4u ? NH_CLS_STATIC4_LATE : (d <= 8u ? NH_CLS_STATIC8_LATE : NH_CLS_STATICN)) : NH_CLS_GENERAL;
			}
			body_class[x] = (uint8_t)cls;
			if (cls == NH_CLS_STATIC8_LATE) st->has_static8 = 1;
			if (cls == NH_CLS_STATICN) st->has_staticN = 1;
			// (mass_inverse -> momentum.unused0, nudge.cpp:4198: the level-scheduled sweeps read it from there)
			if (cls == NH_CLS_NONE || cls == NH_CLS_GENERAL) momentum[x].unused0 = props[x].mass_inverse;
		}
	}
}

// general contacts: flag, count, level = UNSET; `general_list` = their indices, ascending.
// A workgroup owns a contiguous chunk of the contacts: it counts its general contacts, reserves its share of the list with ONE atomic on the
// global counter (one per wave cost 2 ms for the 12 M contacts of the ball pit: same-address atomics serialise), and fills it in order.
__global__ __launch_bounds__(256) void k_contact_class(nh_DevState* __restrict__ st, const nh_BodyPair* __restrict__ bodies, const uint8_t* __restrict__ body_class,
                                                       uint32_t* __restrict__ level, uint32_t* __restrict__ general_list, uint32_t* __restrict__ pred_a, uint32_t* __restrict__ pred_b) {
	if (!st->has_pending) return;             // every body with contacts is on the one-body path: no general contacts
	__shared__ uint32_t s_wave[4], s_base;
	const uint32_t n = st->contacts;
	const uint32_t chunk = (((n + gridDim.x - 1u) / gridDim.x) + 255u) & ~255u;
	const uint32_t begin = min(blockIdx.x * chunk, n), end = min(begin + chunk, n);
	const uint32_t lane = nh_lane(), wave = threadIdx.x >> 6;
	// pass 1: classify (level[] remembers the verdict), count
	uint32_t mine = 0;
	for (uint32_t i = begin + threadIdx.x; i < end; i += 256) {
		const nh_BodyPair p = bodies[i];
		const bool general = (p.a && body_class[p.a] == NH_CLS_GENERAL) || (p.b && body_class[p.b] == NH_CLS_GENERAL);
		level[i] = general ? NH_UNSET : 0u;
		if (general) { if (!p.a) pred_a[i] = NH_NONE; if (!p.b) pred_b[i] = NH_NONE; ++mine; }
	}
	for (int d = 32; d >= 1; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d);
	if (lane == 0) s_wave[wave] = mine;
	__syncthreads();
	if (threadIdx.x == 0) { const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]; s_base = total ? atomicAdd(&st->general_contacts, total) : 0u; }
	__syncthreads();
	// pass 2: ordered fill (block-uniform trip count; a wave's share of a 256-contact slice follows the shares of the waves before it)
	uint32_t cursor = s_base;
	for (uint32_t i0 = begin; i0 < end; i0 += 256) {
		const uint32_t i = i0 + threadIdx.x;
		const bool general = i < end && level[i] == NH_UNSET;
		const unsigned long long bal = __ballot(general);
		__syncthreads();
		if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
		__syncthreads();
		uint32_t before = 0, total = 0;
		for (uint32_t w = 0; w < 4; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
		if (general) general_list[cursor + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = i;
		cursor += total;
	}
}

// ---- exact replay of the reference's greedy batch scheduler (nudge.cpp:4206-4339), one wave -------------------
#define GR_BUCKETS 16
#define GR_OPEN 64          // open (vacant) batches kept per bucket in LDS
// Exact-order mode, common case first.  When the scheduler below never meets a lane conflict, contact i ends up in batch
// (i / 128) * 16 + i % 16 (its bucket's batch completes with every eighth contact of the bucket, batches are emitted in that order,
// leftovers bucket by bucket).  The first conflict the scheduler can meet is two contacts of ONE such batch sharing a dynamic body, so if no
// batch of the closed form holds one, the closed form IS the schedule: checked here with one lane per batch; only otherwise the
// sequential replay runs (it is a single wave walking all contacts: milliseconds for tens of thousands).
__global__ __launch_bounds__(256) void k_order_check(nh_DevState* __restrict__ st, const nh_BodyPair* __restrict__ bodies, uint32_t* __restrict__ slot_key, uint32_t seq) {
	const uint32_t n = st->contacts;
	const uint32_t nbatch = ((n + 127u) / 128u) * 16u;
	for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nbatch; t += gridDim.x * blockDim.x) {
		const uint32_t first = (t >> 4) * 128u + (t & 15u);
		uint32_t ca[8], cb[8];
		bool conflict = false;
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const uint32_t i = first + 16u * k;
			ca[k] = NH_NONE; cb[k] = NH_NONE;
			if (i < n) {
				const nh_BodyPair bp = bodies[i];
				ca[k] = bp.a ? bp.a : bp.b; cb[k] = bp.b ? bp.b : bp.a;      // dependencies on body 0 do not count (nudge.cpp:4238-4240)
				slot_key[i] = t;
#pragma unroll
				for (int j = 0; j < k; ++j) conflict |= ca[j] == ca[k] || cb[j] == ca[k] || ca[j] == cb[k] || cb[j] == cb[k];
			}
		}
		if (conflict) st->order_conflict = seq;
	}
}

// Open batches beyond the GR_OPEN kept in LDS spill to global memory (`spill_ab` / `spill_idx`: GR_BUCKETS x spill_cap x 8 entries from the
// arena): a dynamic hub body with thousands of contacts (a tray of boxes) makes every contact of a bucket conflict, so no batch completes
// and the open list grows with the contact count -- the reference sizes these arrays by contacts.count (nudge.cpp:4222-4223).
struct gr_store {
	uint2 (*lds_ab)[GR_OPEN + 1][8];
	uint32_t (*lds_idx)[GR_OPEN + 1][8];
	uint2* spill_ab; uint32_t* spill_idx; uint32_t spill_cap;
	__device__ __forceinline__ size_t at(uint32_t b, uint32_t j, uint32_t l) const { return ((size_t)b * spill_cap + (j - (GR_OPEN + 1u))) * 8u + l; }
	__device__ __forceinline__ uint2 ab(uint32_t b, uint32_t j, uint32_t l) const {
		if (j <= GR_OPEN) return lds_ab[b][j][l];
		const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(spill_ab + at(b, j, l)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
	}
	__device__ __forceinline__ void set_ab(uint32_t b, uint32_t j, uint32_t l, uint2 v) const {
		if (j <= GR_OPEN) lds_ab[b][j][l] = v;
		else __hip_atomic_store(reinterpret_cast<unsigned long long*>(spill_ab + at(b, j, l)), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__device__ __forceinline__ uint32_t idx(uint32_t b, uint32_t j, uint32_t l) const {
		return j <= GR_OPEN ? lds_idx[b][j][l] : __hip_atomic_load(spill_idx + at(b, j, l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__device__ __forceinline__ void set_idx(uint32_t b, uint32_t j, uint32_t l, uint32_t v) const {
		if (j <= GR_OPEN) lds_idx[b][j][l] = v; else __hip_atomic_store(spill_idx + at(b, j, l), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
};

__global__ __launch_bounds__(64) void k_greedy_replay(nh_DevState* __restrict__ st, const nh_BodyPair* __restrict__ bodies, uint32_t* __restrict__ slot_key, uint32_t seq,
                                                      uint2* __restrict__ spill_ab, uint32_t* __restrict__ spill_idx, uint32_t spill_cap) {
	if (st->order_conflict != seq) return;          // k_order_check: the closed form it wrote is the schedule
	__shared__ uint2 pair_ab[GR_BUCKETS][GR_OPEN + 1][8];
	__shared__ uint32_t slot_idx[GR_BUCKETS][GR_OPEN + 1][8];
	__shared__ uint32_t vacancy[GR_BUCKETS];
	__shared__ uint32_t emitted;
	const gr_store S = { pair_ab, slot_idx, spill_ab, spill_idx, spill_cap };
	const uint32_t max_open = GR_OPEN + spill_cap;      // largest legal index of the all-invalid padding entry
	const uint32_t lane = threadIdx.x, sub = lane >> 3, ln = lane & 7;
	const uint32_t n = st->contacts;
	if (lane < GR_BUCKETS) { vacancy[lane] = 0; }
	if (lane == 0) emitted = 0;
	for (uint32_t k = lane; k < GR_BUCKETS * 8; k += 64) pair_ab[k >> 3][0][k & 7] = make_uint2(NH_NONE, NH_NONE);
	__syncthreads();
	for (uint32_t i = 0; i < n; ++i) {
		nh_BodyPair bp = bodies[i];
		uint32_t bucket = i % GR_BUCKETS;
		uint32_t ca = bp.a ? bp.a : bp.b, cb = bp.b ? bp.b : bp.a;     // ignore dependencies on body 0
		uint32_t vac = vacancy[bucket];
		// first open batch (or the all-invalid padding entry at index vac) without a conflicting lane
		uint32_t j = NH_NONE;
		for (uint32_t base = 0; j == NH_NONE; base += 8) {
			uint32_t jb = base + sub;
			bool in_range = jb <= vac;
			uint2 ab = in_range ? S.ab(bucket, jb, ln) : make_uint2(NH_NONE, NH_NONE);
			bool conflict = in_range && (ab.x == ca || ab.y == ca || ab.x == cb || ab.y == cb);
			unsigned long long bal = __ballot(conflict);
			unsigned long long rng = __ballot(in_range);
			for (uint32_t s = 0; s < 8; ++s) {
				bool ok = ((rng >> (s * 8)) & 1ull) && (((bal >> (s * 8)) & 0xffull) == 0ull);
				if (ok) { j = base + s; break; }
			}
		}
		// first free lane of that batch
		uint2 mine = S.ab(bucket, j, ln);
		unsigned long long freeb = __ballot(sub == 0 && mine.x == NH_NONE && mine.y == NH_NONE);
		uint32_t free_lane = (uint32_t)__ffsll((long long)(freeb & 0xffull)) - 1u;
		__syncthreads();
		if (lane == 0) {
			S.set_idx(bucket, j, free_lane, i);
			S.set_ab(bucket, j, free_lane, make_uint2(ca, cb));
		}
		__syncthreads();
		bool changed = false;
		if (j == vac) {
			vac = vac + 1;
			changed = true;
			if (vac > max_open) { if (lane == 0) st->error = NH_ERR_SCHEDULER_CAPACITY; return; }
		} else if (free_lane == 7) {
			// batch complete: emit it, move the last open batch into its place
			uint32_t e = emitted;
			if (lane < 8) slot_key[S.idx(bucket, j, lane)] = e;
			vac = vac - 1;
			__syncthreads();
			if (lane < 8) {
				S.set_ab(bucket, j, lane, S.ab(bucket, vac, lane));
				S.set_idx(bucket, j, lane, S.idx(bucket, vac, lane));
			}
			if (lane == 0) emitted = e + 1;
			changed = true;
		}
		__syncthreads();
		if (changed) {
			if (lane == 0) vacancy[bucket] = vac;
			if (lane < 8) S.set_ab(bucket, vac, lane, make_uint2(NH_NONE, NH_NONE));
		}
		__syncthreads();
	}
	// leftovers, bucket by bucket (nudge.cpp:4316-4337)
	uint32_t e = emitted;
	for (uint32_t b = 0; b < GR_BUCKETS; ++b) {
		uint32_t vac = vacancy[b];
		for (uint32_t j = 0; j < vac; ++j) {
			if (lane < 8) {
				uint2 ab = S.ab(b, j, lane);
				if (!(ab.x == NH_NONE && ab.y == NH_NONE)) slot_key[S.idx(b, j, lane)] = e;
			}
			++e;
		}
	}
}

// ---- body state access -------------------------------------------------------------------------------------------------
__device__ __forceinline__ nh_vel load_vel(const nh_BodyMomentum* __restrict__ momentum, const nh_BodyProperties* __restrict__ props, uint32_t i) {
	nh_BodyMomentum m = momentum[i];
	nh_vel v;
	v.vx = m.velocity[0]; v.vy = m.velocity[1]; v.vz = m.velocity[2];
	v.mass_inverse = props[i].mass_inverse;          // the reference stashes it in unused0 (nudge.cpp:4198)
	v.wx = m.angular_velocity[0]; v.wy = m.angular_velocity[1]; v.wz = m.angular_velocity[2];
	v.w_unused = m.unused1;
	return v;
}

__device__ __forceinline__ void store_vel(nh_BodyMomentum* __restrict__ momentum, uint32_t i, const nh_vel& v) {
	nh_BodyMomentum m;
	m.velocity[0] = v.vx; m.velocity[1] = v.vy; m.velocity[2] = v.vz; m.unused0 = v.mass_inverse;
	m.angular_velocity[0] = v.wx; m.angular_velocity[1] = v.wy; m.angular_velocity[2] = v.wz; m.unused1 = v.w_unused;
	momentum[i] = m;
}

__device__ __forceinline__ nh_inertia body_inertia(const nh_Transform* __restrict__ xf, const nh_BodyProperties* __restrict__ props, uint32_t i) {
	nh_Transform t = xf[i];
	nh_BodyProperties p = props[i];
	nh_quat q = { t.rotation[0], t.rotation[1], t.rotation[2], t.rotation[3] };
	return nh_world_inertia(q, p.inertia_inverse[0], p.inertia_inverse[1], p.inertia_inverse[2]);
}

__device__ __forceinline__ void store_row(float* __restrict__ rows, uint32_t c, const nh_row& r) {
	float4* dst = reinterpret_cast<float4*>(rows + (size_t)c * 40);
	const float4* src = reinterpret_cast<const float4*>(&r);
#pragma unroll
	for (int k = 0; k < 10; ++k) dst[k] = src[k];
}

__device__ __forceinline__ void load_row(const float* __restrict__ rows, uint32_t c, nh_row& r) {
	const float4* src = reinterpret_cast<const float4*>(rows + (size_t)c * 40);
	float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
	for (int k = 0; k < 10; ++k) dst[k] = src[k];
}

// ---- one-body fast path: setup (rows + warm start) ----------------------------------------------------------------------
template<bool STORE>
__global__ __launch_bounds__(256) void k_setup_static(uint32_t nbodies, const uint8_t* __restrict__ body_class, const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj,
                                                      const nh_Contact* __restrict__ contacts, const nh_BodyPair* __restrict__ bodies, nh_CachedContactImpulse* __restrict__ cached,
                                                      const nh_Transform* __restrict__ xf, const nh_BodyProperties* __restrict__ props, nh_BodyMomentum* __restrict__ momentum,
                                                      float* __restrict__ rows, float4* __restrict__ states, const uint32_t* __restrict__ d_any) {
	if (d_any && !*d_any) return;
	const nh_inertia I0 = body_inertia(xf, props, 0);
	const nh_Transform t0 = xf[0];
	const nh_f3 pos0 = nh_make3(t0.position[0], t0.position[1], t0.position[2]);
	const nh_vel vel0 = load_vel(momentum, props, 0);
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < nbodies; x += gridDim.x * blockDim.x) {
		uint32_t cls = body_class[x];
		if (STORE ? (cls != NH_CLS_STATICN) : (cls != NH_CLS_STATIC4 && cls != NH_CLS_STATIC8)) continue;
		nh_inertia IX = body_inertia(xf, props, x);
		nh_Transform tx = xf[x];
		nh_f3 posx = nh_make3(tx.position[0], tx.position[1], tx.position[2]);
		nh_vel v = load_vel(momentum, props, x);
		nh_vel z = vel0;
		uint32_t b = off[x], e = off[x + 1];
		// results of up to 8 contacts are kept in registers and stored back to back after the loop: stores that are
		// spread out in time reach HBM as partial lines (measured 2x write amplification)
		float4 sbuf[8];
		nh_CachedContactImpulse obuf[8];
#pragma unroll 1
		for (uint32_t q = b; q < e; ++q) {
			uint32_t c = adj[q];
			nh_Contact ct = contacts[c];
			nh_contact_in ci = { ct.position[0], ct.position[1], ct.position[2], ct.penetration, ct.normal[0], ct.normal[1], ct.normal[2], ct.friction };
			nh_CachedContactImpulse w = cached[c];
			bool x_is_a = bodies[c].a == x;
			nh_row r;
			nh_state s;
			if (x_is_a) {
				nh_build_row(ci, posx, pos0, IX, I0, v.mass_inverse, z.mass_inverse, r);
				nh_warm_start(r, w.impulse[0], w.impulse[1], w.impulse[2], v, z, s);
			} else {
				nh_build_row(ci, pos0, posx, I0, IX, z.mass_inverse, v.mass_inverse, r);
				nh_warm_start(r, w.impulse[0], w.impulse[1], w.impulse[2], z, v, s);
			}
			// keep update_cached_impulses valid even if apply_impulses is never called
			nh_CachedContactImpulse o;
			nh_export_impulse(r, s, o.impulse[0], o.impulse[1], o.impulse[2]);
			o.unused = 0.0f;
			float4 sv = make_float4(s.normal, s.friction_x, s.friction_y, 0.0f);
			if (STORE) {
				// (rows are not stored: bodies with more than 8 static contacts rebuild them in every sweep, see k_apply_static<0>)
				states[c] = sv;
				cached[c] = o;
			} else {
				uint32_t k = q - b;
#pragma unroll
				for (int j = 0; j < 8; ++j) if (k == (uint32_t)j) { sbuf[j] = sv; obuf[j] = o; }
			}
		}
		if (!STORE) {
#pragma unroll
			for (int j = 0; j < 8; ++j) if (b + j < e) states[adj[b + j]] = sbuf[j];
#pragma unroll
			for (int j = 0; j < 8; ++j) if (b + j < e) cached[adj[b + j]] = obuf[j];
		}
		store_vel(momentum, x, v);
	}
}

// ---- one-body fast path: `iterations` PGS sweeps ---------------------------------------------------------------------------
// MAXC > 0: the body's <= MAXC constraint rows stay in registers across all iterations (160 B each): every row is
// read from HBM once per call, not once per iteration.  MAXC == 0: any contact count, rows re-read per iteration.
// FUSED: the call also does what nh_setup_contact_constraints deferred for these bodies -- fetch the cached impulse of every
// contact (read_cached_impulses, nudge.cpp:4057-4083: same-index probe, then binary search in the tag-sorted cache), build the
// rows and warm start (nudge.cpp:4563-4632) -- before the sweeps, so a contact's row exists only in registers, once per step.
struct nh_CacheView { const uint64_t* tags; const uint32_t* features; const uint64_t* ctags; const uint32_t* cfeatures; const nh_CachedContactImpulse* cdata; const nh_DevState* st; };

template<int MAXC, bool FUSED>
__global__ __launch_bounds__(256) void k_apply_static(uint32_t nbodies, const uint8_t* __restrict__ body_class, const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj,
                                                      const nh_BodyPair* __restrict__ bodies, const nh_BodyProperties* __restrict__ props, nh_BodyMomentum* __restrict__ momentum,
                                                      const float* __restrict__ rows, float4* __restrict__ states, uint32_t iterations,
                                                      const nh_Contact* __restrict__ contacts, const nh_Transform* __restrict__ xf, nh_CachedContactImpulse* __restrict__ impulses,
                                                      nh_CacheView cv, uint32_t cls_a, uint32_t cls_b) {
	const nh_vel vel0 = load_vel(momentum, props, 0);
	const nh_inertia I0 = body_inertia(xf, props, 0);
	const nh_Transform t0 = xf[0];
	const nh_f3 pos0 = nh_make3(t0.position[0], t0.position[1], t0.position[2]);
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < nbodies; x += gridDim.x * blockDim.x) {
		{ const uint32_t my_class = body_class[x]; if (my_class != cls_a && my_class != cls_b) continue; }
		nh_vel v = load_vel(momentum, props, x);
		nh_vel z = vel0;
		uint32_t b = off[x], e = off[x + 1];
		if (MAXC > 0) {
			constexpr int M = MAXC > 0 ? MAXC : 1;
			nh_row r[M];
			nh_state s[M];
			bool is_a[M];
			uint32_t cid[M];
			uint32_t cnt = e - b;
			// rows are a pure function of (contact, body transforms, properties): rebuilt here, never stored
			const nh_inertia IX = body_inertia(xf, props, x);
			const nh_Transform tx = xf[x];
			const nh_f3 posx = nh_make3(tx.position[0], tx.position[1], tx.position[2]);
			// phase A: everything the body needs from memory, as independent loads (one round trip at this occupancy)
			nh_Contact ctc[M];
			float wx[M], wy[M], wz[M];
			bool miss[M];
			const uint32_t m = FUSED ? cv.st->cache : 0u;
#pragma unroll
			for (int q = 0; q < M; ++q) cid[q] = (uint32_t)q < cnt ? adj[b + q] : 0u;
#pragma unroll
			for (int q = 0; q < M; ++q) {
				miss[q] = false; wx[q] = wy[q] = wz[q] = 0.0f;
				if ((uint32_t)q < cnt) {
					const uint32_t c = cid[q];
					is_a[q] = bodies[c].a == x;
					ctc[q] = contacts[c];
					if (FUSED) {
						// steady state: the cache is last step's contact list, so the entry usually sits at the same index
						const uint64_t t = cv.tags[c]; const uint32_t f = cv.features[c];
						const bool in = c < m;
						const uint64_t ct = in ? cv.ctags[c] : 0ull; const uint32_t cf = in ? cv.cfeatures[c] : 0u;
						nh_CachedContactImpulse wc = { { 0.0f, 0.0f, 0.0f }, 0.0f };
						if (in) wc = cv.cdata[c];
						const bool hit = in && ct == t && cf == f;
						if (hit) { wx[q] = wc.impulse[0]; wy[q] = wc.impulse[1]; wz[q] = wc.impulse[2]; }
						miss[q] = !hit;
					} else {
						float4 sv = states[c];
						s[q].normal = sv.x; s[q].friction_x = sv.y; s[q].friction_y = sv.z;
					}
				}
			}
			if (FUSED) {
#pragma unroll
				for (int q = 0; q < M; ++q) {
					if (miss[q]) {      // new or shifted contact: binary search in the tag-sorted cache (nudge.cpp:4057-4083 is a merge-join)
						const uint32_t c = cid[q];
						const uint64_t t = cv.tags[c]; const uint32_t f = cv.features[c];
						uint32_t lo = 0, hi = m;
						while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (tag_less(cv.ctags[mid], cv.cfeatures[mid], t, f)) lo = mid + 1; else hi = mid; }
						if (lo < m && cv.ctags[lo] == t && cv.cfeatures[lo] == f) { nh_CachedContactImpulse w = cv.cdata[lo]; wx[q] = w.impulse[0]; wy[q] = w.impulse[1]; wz[q] = w.impulse[2]; }
					}
				}
			}
			// phase B: rows, and (fused) the warm start in slot order
#pragma unroll
			for (int q = 0; q < M; ++q) {
				if ((uint32_t)q < cnt) {
					const nh_Contact& ct = ctc[q];
					nh_contact_in ci = { ct.position[0], ct.position[1], ct.position[2], ct.penetration, ct.normal[0], ct.normal[1], ct.normal[2], ct.friction };
					if (is_a[q]) nh_build_row(ci, posx, pos0, IX, I0, v.mass_inverse, z.mass_inverse, r[q]);
					else         nh_build_row(ci, pos0, posx, I0, IX, z.mass_inverse, v.mass_inverse, r[q]);
					if (FUSED) {
						if (is_a[q]) nh_warm_start(r[q], wx[q], wy[q], wz[q], v, z, s[q]);
						else         nh_warm_start(r[q], wx[q], wy[q], wz[q], z, v, s[q]);
					}
				}
			}
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q) {
					if ((uint32_t)q < cnt) {
						if (is_a[q]) nh_apply_one(r[q], s[q], v, z); else nh_apply_one(r[q], s[q], z, v);
					}
				}
			}
#pragma unroll
			for (int q = 0; q < M; ++q)
				if ((uint32_t)q < cnt) {
					states[cid[q]] = make_float4(s[q].normal, s[q].friction_x, s[q].friction_y, 0.0f);
					nh_CachedContactImpulse o;        // update_cached_impulses (nudge.cpp:4872-4882) for this contact
					nh_export_impulse(r[q], s[q], o.impulse[0], o.impulse[1], o.impulse[2]);
					o.unused = 0.0f;
					impulses[cid[q]] = o;
				}
		} else {
			// any contact count (rare: a body resting on more than 8 static contacts): rows are rebuilt from the contact in every sweep
			// instead of being stored -- a pure function of (contact, body transforms, properties), so the same bits every time
			const nh_inertia IXn = body_inertia(xf, props, x);
			const nh_Transform txn = xf[x];
			const nh_f3 posxn = nh_make3(txn.position[0], txn.position[1], txn.position[2]);
			for (uint32_t it = 0; it < iterations; ++it) {
				for (uint32_t q = b; q < e; ++q) {
					uint32_t c = adj[q];
					const bool x_is_a = bodies[c].a == x;
					nh_Contact ct = contacts[c];
					nh_contact_in ci = { ct.position[0], ct.position[1], ct.position[2], ct.penetration, ct.normal[0], ct.normal[1], ct.normal[2], ct.friction };
					nh_row r;
					if (x_is_a) nh_build_row(ci, posxn, pos0, IXn, I0, v.mass_inverse, z.mass_inverse, r);
					else        nh_build_row(ci, pos0, posxn, I0, IXn, z.mass_inverse, v.mass_inverse, r);
					float4 sv = states[c];
					nh_state s = { sv.x, sv.y, sv.z };
					if (x_is_a) nh_apply_one(r, s, v, z); else nh_apply_one(r, s, z, v);
					states[c] = make_float4(s.normal, s.friction_x, s.friction_y, 0.0f);
					if (it + 1 == iterations) {
						nh_CachedContactImpulse o;
						nh_export_impulse(r, s, o.impulse[0], o.impulse[1], o.impulse[2]);
						o.unused = 0.0f;
						impulses[c] = o;
					}
				}
			}
		}
		store_vel(momentum, x, v);
	}
}

// ---- one-body fast path, inert static world (the usual case): one-body forms of nh_solver.h ---------------------------------
// Same contract as k_apply_static<MAXC, FUSED>, for MAXC in {4, 8}; rows are the 27-float nh_row1.
//
// Warm-start lookup (FUSED).  The cache is last step's contact list (plus culled entries), sorted like this step's, so a
// contact sits where it sat last step unless contacts before it appeared or vanished -- which shifts EVERYTHING after them.
// `hint[x]` therefore remembers the index body x's first contact had last step: guess = hint[x] + (c - first contact of x).
// A wrong guess is repaired by looking at the 8 entries around it (sorted: if they bracket the key without containing it the
// contact is proven new), and only then by binary search (the reference does a linear merge-join, nudge.cpp:4057-4083).
#define NH_HINT_NONE 0xFFFFFFFFu

__device__ __forceinline__ bool cache_probe(const nh_CacheView& cv, uint32_t m, uint32_t guess, uint64_t t, uint32_t f, float& wx, float& wy, float& wz) {
	// the three loads go out together (the impulse is fetched before the tags are known to match: one memory round trip, not two)
	if (guess < m) {
		const uint64_t ct = cv.ctags[guess];
		const uint32_t cf = cv.cfeatures[guess];
		const float4 w = *reinterpret_cast<const float4*>(cv.cdata + guess);
		if (ct == t && cf == f) { wx = w.x; wy = w.y; wz = w.z; return true; }
	}
	return false;
}

// (arguments and result by value: a reference to the view or to the impulse would pin them in scratch memory across the call, and the scratch
// traffic shares the vector memory counter with the global loads of the kernel)
__device__ __noinline__ float4 cache_search(const uint64_t* __restrict__ ctags, const uint32_t* __restrict__ cfeatures, const nh_CachedContactImpulse* __restrict__ cdata,
                                            uint32_t m, uint32_t guess, uint64_t t, uint32_t f) {
	const float4 none = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	uint32_t lo = 0, hi = m;
	if (guess != NH_HINT_NONE && m) {
		// window [g-4, g+4): entries are sorted by (tag, feature)
		const uint32_t g = guess < m ? guess : m - 1;
		const uint32_t w0 = g >= 4 ? g - 4 : 0, w1 = g + 4 < m ? g + 4 : m;
		bool below = false, above = false;
		for (uint32_t k = w0; k < w1; ++k) {
			const uint64_t ct = ctags[k]; const uint32_t cf = cfeatures[k];
			if (ct == t && cf == f) return *reinterpret_cast<const float4*>(cdata + k);
			if (tag_less(ct, cf, t, f)) below = true; else above = true;
		}
		if ((below || w0 == 0) && (above || w1 == m)) return none;       // bracketed and absent: a new contact
		if (!below) hi = w0; else lo = w1;
	}
	while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (tag_less(ctags[mid], cfeatures[mid], t, f)) lo = mid + 1; else hi = mid; }
	if (lo < m && ctags[lo] == t && cfeatures[lo] == f) return *reinterpret_cast<const float4*>(cdata + lo);
	return none;
}

// Memory access.  One lane owns one body, but a body's per-contact records (contact, pair, tags, cache entry: 8 arrays) are
// fetched by the WAVE: slot e = 4 * body + q of the wave's 64 bodies is loaded by lane e % 64 in round e / 64, so that
// consecutive lanes read consecutive contacts (a body's contacts are adjacent in tag order, and so are its neighbours')
// instead of 64 lanes striding 128 bytes apart; the records reach their owner through LDS.  Results go back the same way.
// Four 27-float rows fit in 256 registers: two waves per SIMD (the compiler would otherwise spend 260 and run one).
struct nh_ob_slot { float4 c0, c1, w; };      // contact (32 B), cached impulse xyz + flags in w.w (bit 0 miss, bit 1 body is "a")

// STILL (nh_internal.h: nh_StillStep, contact storage by slot).  The lane of body x starts from the body's RECORD (body_rec: the one collider pair, with the
// static world, its contacts come from; none: the body only falls) instead of a contact-list record of the dense list: this step's contact count and first dense
// index from the record's place in the tag order (cnt_sorted / start_sorted: the still narrowphase and one scan wrote them), its contacts and feature words from the
// record's own raw slots, last step's feature words and solved impulses from the slot cache.  From those it derives what the full path reads from arrays: the
// dense index of a contact (first + rank of its feature word, nudge.cpp:4027-4044), the order its contacts are solved in (slot keys of those indices,
// nudge.cpp:4206-4339: k_adj_simple's rule), and each contact's warm start (the cached impulse with the same feature word, nudge.cpp:4057-4083 -- a pair's
// entries can only be the pair's own).  The solved impulses and this step's feature words go back to the slot cache.  EVERY dynamic body is the kernel's: one
// without contacts gets gravity and its advance here.  If an earlier kernel of the step raised st->still_fail, or body 0 is not inert, the kernel leaves first.
#ifndef NH_SOLVER_WPE
#define NH_SOLVER_WPE 2          // wavefronts per SIMD the four-contact solver is compiled for (A/B: -DNH_SOLVER_WPE=3)
#endif
struct nh_StillView { const uint32_t* body_rec; const uint32_t* body_pos; const uint32_t* cnt_sorted; uint32_t* start_sorted; const uint32_t* raw_feature; uint32_t* sc_feat; uint32_t* sc_count; uint32_t pair_cap;
                     const int2* delta; uint32_t parity; };

// AHEAD (STILL, plain form; nh_internal.h: xform ahead): the lane that has advanced its body also does the NEXT step's k_xform<true> for the body's one collider -- world
// transform, AABB, the test against the inflated box, scene bounds, largest idle counter -- with k_xform's own arithmetic (nh_collide.hip: k_xform, k_asleep_check).
// PAIR (AHEAD only; nh_internal.h: pair ahead): ... and the next step's NARROWPHASE for the body's one kept pair, from what the lane holds in registers and the static
// partner's transform, box and shape: contacts into the record's raw slots, count into its place in the tag order, a change of the count onto the next step's list.
template<int MAXC, bool FUSED, bool CONTIG = false, int NW = 4, bool STILL = false, bool AHEAD = false, bool PAIR = false, bool PART = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(MAXC == 4 ? NH_SOLVER_WPE : 1, MAXC == 4 ? NH_SOLVER_WPE : 1))) void k_solve_one_body(uint32_t nbodies, const uint8_t* __restrict__ body_class, const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj,
                                                        const nh_BodyPair* __restrict__ bodies, const nh_BodyProperties* __restrict__ props, nh_BodyMomentum* __restrict__ momentum,
                                                        float4* __restrict__ states, uint32_t iterations,
                                                        const nh_Contact* __restrict__ contacts, nh_Transform* xf, nh_CachedContactImpulse* __restrict__ impulses,
                                                        nh_CacheView cv, uint32_t* __restrict__ hint, uint32_t cls_a, uint32_t cls_b, uint32_t require_inert,
                                                        const uint2* __restrict__ simple = nullptr, nh_FusedStep fs = nh_FusedStep(), nh_StillView sv = nh_StillView(),
                                                        nh_AheadView av = nh_AheadView()) {
	// CONTIG: every body of the class comes from k_adj_simple (one collider pair with the static world): its contact list is the record
	// simple[x] -- one load next to the body state instead of the dependent chain off[x] -> adj[] -> contact
	constexpr int M = MAXC;
	// launched before the host knew whether body 0 is inert: if it is not, the general kernels take these bodies after the round trip
	// `require_inert`: bit 0 -- launched before the host knew whether body 0 is inert (leave if it is not); bit 1 -- NH_FLAG_SINGLE_APPLY: no later
	// nh_apply_impulses will continue from the solver states, so they are not stored
	const bool keep_states = !(require_inert & 2u);
	if (nbodies == 0u) return;
	if (STILL && CONTIG && NW == 1 && !PART && (fs.bits & 8u) && blockIdx.x != 0u) {
		// (sleepers form: a wave all of whose bodies are asleep has nothing to do in any instantiation -- it finds out with ONE load, before the words and the list every other
		// wave starts with, and leaves; the first workgroup stays: its first thread files the step's counters)
		const uint32_t x0 = blockIdx.x * 64u + threadIdx.x;
		const uint32_t i0 = (x0 < nbodies) ? (uint32_t)fs.idle[x0] : 0xffu;
		if (__builtin_amdgcn_ballot_w64(i0 != 0xffu) == 0ull) return;
	}
	// (the two words of device state -- is body 0 inert, how long is the cache -- are fetched beside the first round of loads, not before it)
	uint32_t inert_word = (require_inert & 1u) ? cv.st->static_inert : 1u;
	if (STILL) {
		// (every workgroup reads the same words and comes to the same verdict)
		const nh_BodyMomentum m0 = momentum[0];
		const nh_BodyProperties p0 = props[0];
		const bool inert = nh_is_inert(m0.velocity, m0.angular_velocity, p0.inertia_inverse, p0.mass_inverse);
		if (!inert && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&const_cast<nh_DevState*>(cv.st)->still_failed_seq, fs.seq);
		if (!inert || cv.st->still_failed_seq >= fs.guard_seq) {
			// (the verdict the host is waiting for: see below)
			if (fs.host_counters && blockIdx.x == 0 && threadIdx.x == 0 && !(PART && fs.part == 1u)) {
				const volatile uint32_t* src = reinterpret_cast<const volatile uint32_t*>(cv.st);
#pragma unroll
				for (uint32_t k = 0; k < NH_COUNTER_WORDS; ++k) fs.host_counters[k] = src[k];
				__threadfence_system();
				__hip_atomic_store(fs.host_counters + NH_COUNTER_WORDS, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);          // (early counters: the word the host may poll for)
			}
			return;
		}
		inert_word = 1u;
	}
	__shared__ nh_ob_slot s_slot[NW][64 * M];
	__shared__ uint32_t s_cid[NW][64 * M];
	__shared__ uint32_t s_hint[NW][64], s_first[NW][64];
	__shared__ int2 s_delta[STILL ? NH_DELTA_MAX : 1];                   // STILL: this step's changes of contact counts (tag-order position, difference)
	// AHEAD: the body's collider (fetched with the first round of loads).  What the lane needs of it -- local transform, size, inflated box: 80 bytes -- is asked for when
	// the rows have been built and delivered straight to LDS while the sweeps run: no register holds it, and the wave's contact slots (s_slot: nobody looks at them
	// between the row build and the results) are where it lands
	__shared__ uint32_t s_col[AHEAD ? NW : 1][AHEAD ? 64 : 1];
	__shared__ uint32_t s_pk[PAIR ? NW : 1][PAIR ? 4 : 1][PAIR ? 64 : 1];          // PAIR: the body's kept pair (2 words) and the record's key on file (2), fetched with the first round of loads
	uint32_t n_delta = 0;
	if (STILL) {
		// The first contact of every record in the dense list is kept from step to step (start_sorted); the records whose count changed this step are listed, and a lane
		// shifts its record's start by the changes BEFORE it in the tag order.  (More changes than the list holds: the scan has run instead, nothing to shift.)
		n_delta = cv.st->delta_overflow[sv.parity] ? 0u : min(cv.st->delta_count[sv.parity], NH_DELTA_MAX);
		for (uint32_t k = threadIdx.x; k < n_delta; k += blockDim.x) s_delta[k] = sv.delta[sv.parity * NH_DELTA_MAX + k];
		if (blockIdx.x == 0 && threadIdx.x == 0 && !(PART && fs.part == 1u)) {          // (halo split: the interior launch does what a step does once)
			// the contact count of the step = the total behind the last record: kept like the starts
			const uint32_t nrec = cv.st->records;
			int shift = 0;
			for (uint32_t k = 0; k < n_delta; ++k) shift += sv.delta[sv.parity * NH_DELTA_MAX + k].y;
			const uint32_t total = sv.start_sorted[nrec] + (uint32_t)shift;
			if (shift) sv.start_sorted[nrec] = total;
			nh_DevState* const stw = const_cast<nh_DevState*>(cv.st);
			stw->contacts = total;
			// (sleepers form, nh_internal.h: what k_xform<true> and the still narrowphase counted becomes the step's counters; all zero when nobody sleeps)
			uint32_t n_sleeping = stw->still_sleeping[sv.parity], n_culled = stw->still_culled[sv.parity];
			if (fs.bits & 4u) {          // (sleepers form: the narrowphase's share, nh_internal.h: sleep_part)
				for (uint32_t k = 0; k < NH_SLEEP_PARTS; ++k) { n_sleeping += stw->sleep_part[sv.parity][k][0]; n_culled += stw->sleep_part[sv.parity][k][1]; }
			}
			stw->active = (nbodies - 1u) - min(stw->still_asleep[sv.parity], nbodies - 1u); stw->sleeping = n_sleeping; stw->culled = n_culled;
			// nh_step with late verdicts: everything a still step can fail on has been decided before this launch (k_xform<true>, the narrowphase or k_pair_begin) or at its
			// top (body 0) -- what the lanes find wanting fails the NEXT step -- so the step's counters are final here, and this thread leaves them where the host will look
			// (pinned memory, visible when the launch has ended: the host waits for the event behind it).  The copy the runtime made instead cost 6 us of stream time per step.
			if (fs.host_counters) {
				const volatile uint32_t* src = reinterpret_cast<const volatile uint32_t*>(cv.st);
#pragma unroll
				for (uint32_t k = 0; k < NH_COUNTER_WORDS; ++k) fs.host_counters[k] = src[k];
				// (early counters: ... and this step's number behind them, released at system scope -- the host that waits for the LAST verdict of an nh_step call polls for
				// it and returns while this launch runs: the caller's next call queues up behind it instead of behind a wake-up)
				__threadfence_system();
				__hip_atomic_store(fs.host_counters + NH_COUNTER_WORDS, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
		__syncthreads();
	}
	if (!STILL && fs.host_counters && blockIdx.x == 0 && threadIdx.x == 0) {
		// EARLY COUNTERS (nh_internal.h): nothing in a full step's one-body solver writes the step's counters -- they are final as this launch starts.  Its first thread leaves them
		// in the host's pinned block, then the nh_collide number behind them (a release at system scope: whoever sees the number sees the words); the host polls that word
		const volatile uint32_t* src = reinterpret_cast<const volatile uint32_t*>(cv.st);
#pragma unroll
		for (uint32_t k = 0; k < NH_COUNTER_WORDS; ++k) fs.host_counters[k] = src[k];
		__threadfence_system();
		__hip_atomic_store(fs.host_counters + NH_COUNTER_WORDS, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	const uint32_t wave = threadIdx.x >> 6, lane = nh_lane();
	nh_ob_slot* slot = s_slot[wave];
	uint32_t* scid = s_cid[wave];
	const uint32_t m = (FUSED && !STILL) ? cv.st->cache : 0u;
	nh_CachedContactImpulse* const out_impulses = STILL ? const_cast<nh_CachedContactImpulse*>(cv.cdata) : impulses;
	const uint32_t rounded = (nbodies + (64u * NW - 1u)) / (64u * NW) * (64u * NW);
	// block-uniform trip count; the CONTIG launch covers every body with one pass (no loop: nothing is carried in registers across iterations)
	uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	if (PART && fs.part == 1u) {
		// (the listed bodies, in whatever order the list was made: a lane is a body wherever it sits; lanes behind the list's end stand on the static world, nobody's)
		const uint32_t cnt_listed = *fs.xcount;
		x = x < cnt_listed ? fs.xlist[x] : 0u;
	}
	else if (x >= rounded) return;
	do {
		// Two memory round trips per body, not one per dependent step: (1) class, contact-list record and cache hint -- unconditional loads at a
		// clamped index, a few bytes wasted on bodies of other classes; (2) everything per contact for all slots at once -- contact, tag, feature,
		// the cache entry at the guessed position and its impulse, again at clamped indices and without branches, so that the ~28 loads of a lane
		// are in flight together -- with the body state issued behind them (it is needed only in phase B).  Loads under divergent branches are
		// issued and awaited branch by branch: that form spent 61 % of the wave cycles parked at s_waitcnt (SQ_WAIT_ANY; now 33 %: profiles/r02_sq_wave_states.json).
		const uint32_t xc = min(x, nbodies - 1u);
		const uint32_t my_class = STILL ? 0u : body_class[xc];
		uint2 sp = make_uint2(0u, 0u);
		uint32_t h = NH_HINT_NONE;
		if (CONTIG && !STILL) sp = simple[xc];
		if (FUSED && !STILL) h = hint[xc];
		uint32_t br = NH_BODY_REC_NONE, bpos = 0u, idle0 = 0u;
		if (STILL) { br = sv.body_rec[xc]; bpos = sv.body_pos[xc]; if (CONTIG && (fs.bits & 4u)) idle0 = fs.idle[xc]; }          // (bit 2 of fs.bits: sleepers form -- somebody may be asleep)
		uint32_t col = 0xFFFFFFFFu;
		if (AHEAD) col = av.body_col[xc];
		if (!inert_word) return;
		// (STILL: a body asleep -- only a step in sleepers form gets this far with one -- is nobody's: no gravity, no contacts, no advance, nothing stored; nudge.cpp:3669-3703, 4896-4898)
		bool mine = STILL ? (x >= 1u && x < nbodies && idle0 != 0xffu) : (x < nbodies && (my_class == cls_a || my_class == cls_b));
		if (PART && fs.part == 2u && fs.xskip[xc] != 0u) mine = false;          // (solved by the boundary launch)
		const uint32_t wave_x0 = x - lane;
		uint32_t b = 0, cnt = 0;
		// (a wave none of whose 64 bodies is of the kernel's classes -- a ball pit, where four million bodies are the blocked solver's and a handful sit alone on the
		// floor -- goes straight to its next 64: the second round of loads is the expensive one)
		if (!STILL && __builtin_amdgcn_ballot_w64(mine) == 0ull) { if (CONTIG) break; x += gridDim.x * blockDim.x; continue; }
		// (STILL, sleepers form: a wave all of whose bodies are asleep has nothing to fetch, solve, advance or store)
		// (sleepers ahead: ... nor anything to do ahead -- what its lanes would gather for the next step is the neutral element of everything gathered)
		if (STILL && CONTIG && (fs.bits & 8u) && __builtin_amdgcn_ballot_w64(mine) == 0ull) break;
		if (!mine) { sp = make_uint2(0u, 0u); h = NH_HINT_NONE; br = NH_BODY_REC_NONE; }
		uint32_t cid[M];
		uint32_t first = 0xFFFFFFFFu;
		uint32_t still_base = 0, still_rec = 0;
		bool still_has = false;
		// STILL: two memory round trips, like the full form -- (1) the body's record and tag-order position (above); (2) EVERYTHING else at once: the record's count,
		// start and cached count, the feature words of its four slots (this step's and the cached ones), the four contacts and the four cached impulses IN SLOT ORDER
		// (the wave-cooperative loads below), the body state.  Which slot is solved when, and which cached impulse it starts from, is worked out afterwards in
		// registers -- the version that first read counts, then features, then sorted, then fetched the contacts in solve order waited for four dependent round trips
		uint32_t st_start = 0, st_cnt = 0, st_oc = 0;
		uint32_t nf[M], of[M];
		uint32_t sl[M], jm[M];          // solve order -> slot of the contact, slot of the cached impulse it starts from (0xFFFFFFFF: none)
#pragma unroll
		for (int k = 0; k < M; ++k) { nf[k] = 0u; of[k] = 0u; sl[k] = (uint32_t)k; jm[k] = 0xFFFFFFFFu; }
		if (STILL) {
			still_has = br != NH_BODY_REC_NONE;
			still_rec = still_has ? (br & ~NH_BODY_REC_IS_A) : 0u;
			const uint32_t n_bb = min(cv.st->pairs, sv.pair_cap);
			still_base = still_rec < n_bb ? 4u * still_rec : 4u * n_bb + (still_rec - n_bb);       // (a record of a pair with a sphere owns ONE slot, behind the box-box records' four each)
			const uint32_t bp = still_has ? bpos : 0u;
			st_cnt = sv.cnt_sorted[bp]; st_start = sv.start_sorted[bp]; st_oc = sv.sc_count[still_rec];
			if (PAIR) {
				const uint2 kp = still_rec < n_bb ? av.kept[still_rec] : av.kept[av.kept_cap - 1u - (still_rec - n_bb)];
				const uint64_t pk = av.rec_key[still_rec];
				s_pk[wave][0][lane] = kp.x; s_pk[wave][1][lane] = kp.y; s_pk[wave][2][lane] = (uint32_t)pk; s_pk[wave][3][lane] = (uint32_t)(pk >> 32);
			}
#pragma unroll
			for (int k = 0; k < M; ++k) { nf[k] = sv.raw_feature[still_base + k]; of[k] = sv.sc_feat[still_base + k]; }
			s_first[wave][lane] = still_has ? still_base : 0xFFFFFFFFu;          // (where the wave-cooperative loads find this body's four slots)
		} else {
		if (CONTIG) cnt = sp.y & 15u;
		else if (mine) { b = off[x]; cnt = off[x + 1] - b; }
#pragma unroll
		for (int q = 0; q < M; ++q) {
			if (CONTIG) cid[q] = (uint32_t)q < cnt ? sp.x + ((sp.y >> (4 + 3 * q)) & 7u) : 0xFFFFFFFFu;
			else cid[q] = (uint32_t)q < cnt ? adj[b + q] : 0xFFFFFFFFu;
			first = min(first, cid[q]);
		}
		}
		if (!STILL) {
#pragma unroll
			for (int q = 0; q < M; ++q) scid[lane * M + q] = cid[q];
			s_hint[wave][lane] = h; s_first[wave][lane] = first;
		}
		// (nh_partition_step: a ghost's lane solves and advances it -- an owned body may rest on it -- but its transforms and pair for the next step come with the next halo)
		const bool ghost_lane = AHEAD && fs.ghost_first != 0u && x >= fs.ghost_first;
		if (AHEAD) { if (!mine || ghost_lane) col = 0xFFFFFFFFu; s_col[wave][lane] = col; }
		__builtin_amdgcn_wave_barrier();
		// phase A: the wave fetches the per-contact records slot by slot; all loads first, then the comparisons
		uint32_t sc[M], sflags[M];
		float4 sc0[M], sc1[M], sw[M];
		uint64_t stag[M], sctag[M];
		uint32_t sfeat[M], scfeat[M], sguess[M];
#pragma unroll
		for (int k = 0; k < M; ++k) {
			const uint32_t e = 64u * k + lane, j = e / M;
			if (STILL) { const uint32_t bj = s_first[wave][j]; sc[k] = bj != 0xFFFFFFFFu ? bj + (e - j * M) : 0xFFFFFFFFu; }          // (slot e mod M of body j's record)
			else sc[k] = scid[e];
			const uint32_t c = sc[k] != 0xFFFFFFFFu ? sc[k] : 0u;
			sc0[k] = reinterpret_cast<const float4*>(contacts + c)[0];
			sc1[k] = reinterpret_cast<const float4*>(contacts + c)[1];
			// (CONTIG: the body's role in its one pair rides in simple[x], bit 28 -- no load of the body pair)
			sflags[k] = (!CONTIG && bodies[c].a == wave_x0 + j) ? 2u : 0u;
			if (STILL) {
				sw[k] = *reinterpret_cast<const float4*>(cv.cdata + c);          // (the impulse cached in the SAME slot; matched to a contact by feature word below)
			} else if (FUSED) {
				const uint32_t hj = s_hint[wave][j];
				sguess[k] = hj != NH_HINT_NONE ? hj + (sc[k] - s_first[wave][j]) : sc[k];
				const uint32_t g = sguess[k] < m ? sguess[k] : 0u;
				stag[k] = cv.tags[c]; sfeat[k] = cv.features[c];
				sctag[k] = cv.ctags[g]; scfeat[k] = cv.cfeatures[g];
				sw[k] = *reinterpret_cast<const float4*>(cv.cdata + g);
			} else {
				sw[k] = states[c];
			}
		}
		// body state: 16-byte loads (the records are 16 / 32 bytes and aligned, include/nudge_hip.h)
		float4 m0 = make_float4(0, 0, 0, 0), m1 = m0, pr = m0, t0 = m0, t1 = make_float4(0, 0, 0, 1);
		uint32_t idle_early = 0;
		if (mine) {
			m0 = reinterpret_cast<const float4*>(momentum + x)[0]; m1 = reinterpret_cast<const float4*>(momentum + x)[1];
			pr = *reinterpret_cast<const float4*>(props + x);
			t0 = reinterpret_cast<const float4*>(xf + x)[0]; t1 = reinterpret_cast<const float4*>(xf + x)[1];
			if (CONTIG && (fs.bits & 2u)) idle_early = (STILL && (fs.bits & 4u)) ? idle0 : fs.idle[x];
			if (CONTIG && (fs.bits & 1u)) {          // NH_FLAG_FUSED_STEP: gravity / damping (k_gravity) on the way in
				float vel[3] = { m0.x, m0.y, m0.z }, ang[3] = { m1.x, m1.y, m1.z };
				nh_gravity_damping(vel, ang, fs.gx_dt, fs.gy_dt, fs.gz_dt, fs.damping);
				m0.x = vel[0]; m0.y = vel[1]; m0.z = vel[2]; m1.x = ang[0]; m1.y = ang[1]; m1.z = ang[2];
			}
		}
#pragma unroll
		for (int k = 0; k < M; ++k) {
			const uint32_t e = 64u * k + lane;
			nh_ob_slot o;
			o.c0 = make_float4(0, 0, 0, 0); o.c1 = o.c0; o.w = o.c0;
			if (sc[k] != 0xFFFFFFFFu) {
				uint32_t flags = sflags[k];
				o.c0 = sc0[k]; o.c1 = sc1[k];
				if (STILL) {
					o.w = make_float4(sw[k].x, sw[k].y, sw[k].z, __uint_as_float(flags));
				} else if (FUSED) {
					const bool hit = sguess[k] < m && sctag[k] == stag[k] && scfeat[k] == sfeat[k];
					if (!hit) flags |= 1u;
					o.w = hit ? make_float4(sw[k].x, sw[k].y, sw[k].z, __uint_as_float(flags)) : make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(flags));
				} else {
					o.w = make_float4(sw[k].x, sw[k].y, sw[k].z, __uint_as_float(flags));
				}
			}
			slot[e] = o;
		}
		if (STILL) {
			uint32_t oc = 0;
			if (still_has) {
				cnt = min(st_cnt, 4u); oc = min(st_oc, 4u);
				uint32_t start = st_start;
				int shift = 0;
				for (uint32_t k = 0; k < n_delta; ++k) shift += (uint32_t)s_delta[k].x < bpos ? s_delta[k].y : 0;
				if (shift) { start += (uint32_t)shift; sv.start_sorted[bpos] = start; }
				st_start = start;
			}
			// dense index of slot k = first + rank of its feature word (ties by slot: k_gather_contacts); the order of solving = by slot key of that index, then by index
			// (k_adj_simple's rule); the cache entry a contact starts from = last step's slot with the same feature word
			uint32_t ci[M], key[M];
#pragma unroll
			for (int k = 0; k < M; ++k) {
				uint32_t r = 0;
#pragma unroll
				for (int j = 0; j < M; ++j) r += ((uint32_t)j < cnt && (uint32_t)k < cnt && (nf[j] < nf[k] || (nf[j] == nf[k] && j < k))) ? 1u : 0u;
				const bool on = (uint32_t)k < cnt;
				ci[k] = on ? st_start + r : 0xFFFFFFFFu;
				key[k] = on ? slot_key_of(ci[k], nullptr) : 0xFFFFFFFFu;
				sl[k] = (uint32_t)k;
			}
#pragma unroll
			for (int q = 1; q < M; ++q) {
#pragma unroll
				for (int j = q; j > 0; --j) {
					const bool sw2 = (key[j] < key[j - 1]) || (key[j] == key[j - 1] && ci[j] < ci[j - 1]);
					const uint32_t tc = sw2 ? ci[j - 1] : ci[j], tk = sw2 ? key[j - 1] : key[j], ts = sw2 ? sl[j - 1] : sl[j];
					ci[j - 1] = sw2 ? ci[j] : ci[j - 1]; key[j - 1] = sw2 ? key[j] : key[j - 1]; sl[j - 1] = sw2 ? sl[j] : sl[j - 1];
					ci[j] = tc; key[j] = tk; sl[j] = ts;
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
				cid[q] = on ? still_base + sl[q] : 0xFFFFFFFFu;
				scid[lane * M + q] = cid[q];          // (where the results go: read by the wave after the barrier before the result stores)
			}
			// this step's feature words and count become the slot cache's (the impulses follow at the end)
			if (still_has) {
#pragma unroll
				for (int k = 0; k < M; ++k) if ((uint32_t)k < cnt) sv.sc_feat[still_base + k] = nf[k];
				sv.sc_count[still_rec] = cnt;
			}
		}
		__builtin_amdgcn_wave_barrier();
		nh_vel v = { m0.x, m0.y, m0.z, pr.w, m1.x, m1.y, m1.z, m1.w };
		const nh_quat qx = { t1.x, t1.y, t1.z, t1.w };
		const nh_inertia IX = nh_world_inertia(qx, pr.x, pr.y, pr.z);
		const nh_f3 posx = nh_make3(t0.x, t0.y, t0.z);
		nh_row1 r[M];
		nh_state s[M];
		bool is_a[M];
		// phase B: rows, and (fused) the warm start in slot order
#pragma unroll
		for (int q = 0; q < M; ++q) {
			is_a[q] = false;
			if ((uint32_t)q < cnt) {
				const nh_ob_slot o = slot[lane * M + (STILL ? sl[q] : (uint32_t)q)];
				const uint32_t flags = STILL ? 0u : __float_as_uint(o.w.w);
				is_a[q] = STILL ? (br & NH_BODY_REC_IS_A) != 0 : (CONTIG ? ((sp.y >> 28) & 1u) != 0 : (flags & 2u) != 0);
				nh_contact_in ci = { o.c0.x, o.c0.y, o.c0.z, o.c0.w, o.c1.x, o.c1.y, o.c1.z, o.c1.w };
				nh_build_row1(ci, posx, IX, v.mass_inverse, is_a[q], r[q]);
				if (FUSED) {
					float wx = o.w.x, wy = o.w.y, wz = o.w.z;
					if (STILL) {
						const float4 wv = slot[lane * M + (jm[q] != 0xFFFFFFFFu ? jm[q] : 0u)].w;
						const bool hit = jm[q] != 0xFFFFFFFFu;
						wx = hit ? wv.x : 0.0f; wy = hit ? wv.y : 0.0f; wz = hit ? wv.z : 0.0f;
					}
					if (!STILL && (flags & 1u)) {
						const float4 found = cache_search(cv.ctags, cv.cfeatures, cv.cdata, m, h != NH_HINT_NONE ? h + (cid[q] - first) : cid[q], cv.tags[cid[q]], cv.features[cid[q]]);
						wx = found.x; wy = found.y; wz = found.z;
					}
					nh_warm_start1(r[q], wx, wy, wz, v, is_a[q], s[q]);
				} else {
					s[q].normal = o.w.x; s[q].friction_x = o.w.y; s[q].friction_y = o.w.z;
				}
			}
		}
		if (FUSED && !STILL && mine) hint[x] = first;          // where this body's contacts start in what becomes the next step's cache
		// PAIR: the static partner of the body's one kept pair -- its world transform, box, shape and tag (this call's k_xform<true> wrote them; nothing moves it) -- asked
		// for before the sweeps (nobody waits for it), used when the body's own collider has its new transform
		nh_xform pp_x = nh_xform();
		float4 pp_min = make_float4(0, 0, 0, 0), pp_max = pp_min, pp_shape = pp_min;
		uint32_t pp_tag = 0, pp_own_tag = 0, pp_partner = 0xFFFFFFFFu, pp_bits = 0;          // pp_bits: 1 = evaluate, 2 = the pair is dead (stamps), 4 = the record is not a pair of the body's collider
		if (PAIR && CONTIG && (fs.bits & 2u) && mine && still_has && s_col[wave][lane] != 0xFFFFFFFFu) {
			const uint32_t ah_col = s_col[wave][lane];
			const uint32_t kx = s_pk[wave][0][lane], ky = s_pk[wave][1][lane];
			const uint32_t index_mask = av.gen ? NH_GEN_INDEX : 0xFFFFFFFFu;
			const uint32_t ca = kx & index_mask, cb = ky & index_mask;
			pp_partner = ca == ah_col ? cb : ca;
			pp_bits = 1u | ((ca != ah_col && cb != ah_col) ? 4u : 0u);
			if (av.gen && cv.st->moved_count != 0u && !((uint32_t)(av.gen[ca] & 0x7Fu) == (kx >> NH_GEN_SHIFT) && (uint32_t)(av.gen[cb] & 0x7Fu) == (ky >> NH_GEN_SHIFT))) pp_bits |= 2u;
			pp_x = av.xf[pp_partner]; pp_min = av.aabb_min[pp_partner]; pp_max = av.aabb_max[pp_partner];
			if (pp_partner < av.nbox) pp_shape = *reinterpret_cast<const float4*>(av.box_data + pp_partner);
			else pp_shape.x = av.sph_data[pp_partner - av.nbox].radius;
			pp_tag = av.ctag[pp_partner]; pp_own_tag = av.ctag[ah_col];
		}
		// (CONTIG: a body's role is one bit for all its contacts, and in most scenes the static world is always "a": when no lane of the wave
		// has D = a the role selects drop out of the sweep)
		float4* const s_ah = reinterpret_cast<float4*>(slot);          // AHEAD: piece k of lane l at s_ah[64 k + l]
		float ah_rad = 0.0f;
		if (AHEAD) {
			__builtin_amdgcn_wave_barrier();          // (every lane has read its slots)
			const uint32_t c0 = s_col[wave][lane];
			const uint32_t cc = c0 != 0xFFFFFFFFu ? c0 : 0u;          // (unconditional loads at a clamped index, like everything else here)
			const bool is_box = cc < av.nbox;
			const float4* lp = reinterpret_cast<const float4*>(is_box ? av.box_xf + cc : av.sph_xf + (cc - av.nbox));
			const float4* sp4 = is_box ? reinterpret_cast<const float4*>(av.box_data + cc) : lp;          // (a sphere's radius -- four bytes -- comes by register)
			if (!is_box) ah_rad = av.sph_data[cc - av.nbox].radius;
			nh_glds16(lp, &s_ah[lane]); nh_glds16(lp + 1, &s_ah[64u + lane]);
			nh_glds16(av.fat_box + 2u * (size_t)cc, &s_ah[128u + lane]); nh_glds16(av.fat_box + 2u * (size_t)cc + 1u, &s_ah[192u + lane]);
			nh_glds16(sp4, &s_ah[256u + lane]);
		}
		if (CONTIG && __builtin_amdgcn_ballot_w64(mine && (STILL ? (still_has && (br & NH_BODY_REC_IS_A) != 0) : ((sp.y >> 28) & 1u) != 0)) == 0) {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) nh_apply_one1(r[q], s[q], v, false);
			}
		} else {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) nh_apply_one1(r[q], s[q], v, is_a[q]);
			}
		}
		// NH_FLAG_FUSED_STEP: the advance (k_advance) on the way out, from the transform held since phase A (7 registers; a reload would be a third exposed round trip)
		float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
		uint32_t idle_in = 0;
		const bool advance_here = CONTIG && (fs.bits & 2u) && mine;
		// (AHEAD: the collider's 80 bytes out of LDS before the results take their place there and before any store is issued -- a wait for the loads that brought them
		// must not turn into a wait for this wave's stores)
		float4 ah_l0 = make_float4(0, 0, 0, 0), ah_l1 = ah_l0, ah_fmn = ah_l0, ah_fmx = ah_l0, ah_sz = ah_l0;
		uint32_t ah_col = 0xFFFFFFFFu;
		if (AHEAD) {
			ah_col = s_col[wave][lane];
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the five loads issued before the sweeps: nothing else of this wave is in flight)
			ah_l0 = s_ah[lane]; ah_l1 = s_ah[64u + lane]; ah_fmn = s_ah[128u + lane]; ah_fmx = s_ah[192u + lane]; ah_sz = s_ah[256u + lane];
		}
		if (advance_here) { a0 = t0; a1 = t1; idle_in = idle_early; }
		// results: state + exported impulse (update_cached_impulses, nudge.cpp:4872-4882) per contact, written by the wave slot by slot
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int q = 0; q < M; ++q) {
			if ((uint32_t)q < cnt) {
				float ox, oy, oz;
				nh_export_impulse1(r[q], s[q], ox, oy, oz);
				slot[lane * M + q].c0 = make_float4(s[q].normal, s[q].friction_x, s[q].friction_y, 0.0f);
				slot[lane * M + q].c1 = make_float4(ox, oy, oz, 0.0f);
			}
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int k = 0; k < M; ++k) {
			const uint32_t e = 64u * k + lane;
			const uint32_t c = scid[e];
			if (c != 0xFFFFFFFFu) {
				if (keep_states) states[c] = slot[e].c0;
				*reinterpret_cast<float4*>(out_impulses + c) = slot[e].c1;
			}
		}
		if (mine) {
			reinterpret_cast<float4*>(momentum + x)[0] = make_float4(v.vx, v.vy, v.vz, v.mass_inverse);
			reinterpret_cast<float4*>(momentum + x)[1] = make_float4(v.wx, v.wy, v.wz, v.w_unused);
		}
		// AHEAD: what this lane's collider contributes to the next step's scene bounds (min corner, order-preserving encoding), the body's new idle counter, and whether
		// the next step would have failed in its k_xform<true> (collider outside its inflated box; body asleep)
		uint32_t nmin[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, nmax[3] = { 0u, 0u, 0u }, ntop = 0u;
		bool nfail = false;
		int pp_count = 0;
		float pair_scale = 0.0f;
		if (PAIR) {
			__builtin_amdgcn_wave_barrier();          // (every lane of the wave has read where the results went)
#pragma unroll
			for (int k = 0; k < M; ++k) scid[lane * M + k] = 0xFFFFFFFFu;          // (where the next step's contacts go: nowhere, unless the lane says so below)
			// the Morton scale of THIS step's scene frame (nudge.cpp:3096-3099): the yardstick of the role rule below
			const nh_DevState* const sp = cv.st;
			const nh_f3 smin = nh_make3(nh_float_unflip(sp->still_smin[sv.parity][0]), nh_float_unflip(sp->still_smin[sv.parity][1]), nh_float_unflip(sp->still_smin[sv.parity][2]));
			const nh_f3 smax = nh_make3(nh_float_unflip(sp->still_smax[sv.parity][0]), nh_float_unflip(sp->still_smax[sv.parity][1]), nh_float_unflip(sp->still_smax[sv.parity][2]));
			pair_scale = nh_morton_scale(smin, smax);
			// more kept pairs that are nobody's than their list holds (k_pair_list): somebody would go unevaluated -- the next step is not this form's
			if (sp->pair_unowned > av.unowned_cap) nfail = true;
		}
		if (advance_here) {
			float pos[3] = { a0.x, a0.y, a0.z }, rot[4] = { a1.x, a1.y, a1.z, a1.w };
			const float vel[3] = { v.vx, v.vy, v.vz }, ang[3] = { v.wx, v.wy, v.wz };
			const uint8_t idle_out = nh_advance_body(pos, rot, vel, ang, (uint8_t)idle_in, fs.time_step);
			fs.idle[x] = idle_out;
			reinterpret_cast<float4*>(xf + x)[0] = make_float4(pos[0], pos[1], pos[2], a0.w);
			reinterpret_cast<float4*>(xf + x)[1] = make_float4(rot[0], rot[1], rot[2], rot[3]);
			if (AHEAD && !ghost_lane) {
				ntop = idle_out; nfail = idle_out == 0xffu;
				const uint32_t c = ah_col;
				if (c != 0xFFFFFFFFu) {
					const bool is_box = c < av.nbox;
					const float4 l0 = ah_l0, l1 = ah_l1, fmn = ah_fmn, fmx = ah_fmx;
					float4 sz = ah_sz;
					if (!is_box) sz.x = ah_rad;
					// Transform * Transform (nudge.cpp:1165-1175), |R| * size (3027-3037): k_xform's lines
					const nh_quat bq = { rot[0], rot[1], rot[2], rot[3] };
					const nh_quat lq = { l1.x, l1.y, l1.z, l1.w };
					const nh_f3 p = nh_rotate(bq, nh_make3(l0.x, l0.y, l0.z)) + nh_make3(pos[0], pos[1], pos[2]);
					const nh_quat q = nh_qmul(bq, lq);
					float sx, sy, sz2;
					if (is_box) {
						const nh_m33 m = nh_matrix(q);
						const nh_f3 c0 = m.c0 * sz.x, c1 = m.c1 * sz.y, c2 = m.c2 * sz.z;
						sx = fabsf(c0.x) + fabsf(c1.x) + fabsf(c2.x);
						sy = fabsf(c0.y) + fabsf(c1.y) + fabsf(c2.y);
						sz2 = fabsf(c0.z) + fabsf(c1.z) + fabsf(c2.z);
					} else {
						sx = sy = sz2 = sz.x;
					}
					const float mnx = p.x - sx, mny = p.y - sy, mnz = p.z - sz2, mxx = p.x + sx, mxy = p.y + sy, mxz = p.z + sz2;
					float4* const wx = reinterpret_cast<float4*>(av.xf) + 2u * (size_t)c;
					wx[0] = make_float4(p.x, p.y, p.z, l0.w);          // (l0.w: the collider's body -- this one)
					wx[1] = make_float4(q.x, q.y, q.z, q.s);
					av.aabb_min[c] = make_float4(mnx, mny, mnz, l0.w);
					av.aabb_max[c] = make_float4(mxx, mxy, mxz, __uint_as_float(0u));          // (plain form: nobody asleep)
					nfail |= !(mnx >= fmn.x && mny >= fmn.y && mnz >= fmn.z && mxx <= fmx.x && mxy <= fmx.y && mxz <= fmx.z);       // (NaN: out)
					nmin[0] = nmax[0] = nh_float_flip(mnx); nmin[1] = nmax[1] = nh_float_flip(mny); nmin[2] = nmax[2] = nh_float_flip(mnz);
					if (PAIR && (pp_bits & 1u)) {
						// ---- the NEXT step's narrowphase for this body's pair (k_narrowphase<*, true, false>: exact boxes, stamps, roles, contact arithmetic, still_record) ----
						const float4 omin = make_float4(mnx, mny, mnz, l0.w), omax = make_float4(mxx, mxy, mxz, 0.0f);
						nh_xform OX; OX.px = p.x; OX.py = p.y; OX.pz = p.z; OX.body = __float_as_uint(l0.w); OX.qx = q.x; OX.qy = q.y; OX.qz = q.z; OX.qs = q.s;
						const bool hit = nh_aabb_overlap(omin, omax, pp_min, pp_max) && __float_as_uint(pp_min.w) != OX.body && !(pp_bits & 2u);
						// roles: "a" = the collider LATER in the Morton order of the AABB min corners in the next step's scene frame -- decided without that frame when one corner is
						// <= the other in every coordinate and two cells of THIS step's frame less in one (k_pair_begin checks the next frame's cells against this one's)
						const float ddx = mnx - pp_min.x, ddy = mny - pp_min.y, ddz = mnz - pp_min.z;
						const float far = 2.0f;
						const bool partner_first = ddx >= 0.0f && ddy >= 0.0f && ddz >= 0.0f && fmaxf(fmaxf(ddx, ddy), ddz) * pair_scale >= far;
						const bool own_first = ddx <= 0.0f && ddy <= 0.0f && ddz <= 0.0f && fmaxf(fmaxf(-ddx, -ddy), -ddz) * pair_scale >= far;
						bool pfail = (pp_bits & 4u) != 0u || (hit && !partner_first && !own_first) || __float_as_uint(pp_min.w) != 0u;          // (a partner that is not the static world: not this lane's to decide)
						if (pfail) atomicAdd(&const_cast<nh_DevState*>(cv.st)->pair_diag[(hit && !partner_first && !own_first) ? 0 : 1], 1u);
						uint64_t key = ~0ull;
						uint32_t body_a = 0, body_b = 0;
						nh_bb_result bb;
						bb.kind = 0; bb.mask = 0;
						nh_contact_out single;
						single.px = single.py = single.pz = single.penetration = single.nx = single.ny = single.nz = 0.0f; single.friction = 0.5f; single.feature = 0u;
						if (hit && !pfail) {
							const bool a_is_own = partner_first;          // (the partner comes first in Morton order: the body's collider is the later one)
							const nh_xform A = a_is_own ? OX : pp_x, B = a_is_own ? pp_x : OX;
							const uint32_t ta = a_is_own ? pp_own_tag : pp_tag, tb = a_is_own ? pp_tag : pp_own_tag;
							const bool own_sph = !is_box, par_sph = pp_partner >= av.nbox;
							const bool a_sph = a_is_own ? own_sph : par_sph, b_sph = a_is_own ? par_sph : own_sph;
							const float own_size[3] = { sz.x, sz.y, sz.z }, par_size[3] = { pp_shape.x, pp_shape.y, pp_shape.z };
							const float ra = a_is_own ? sz.x : pp_shape.x, rb = a_is_own ? pp_shape.x : sz.x;          // (radii, where the collider is a sphere)
							if (!a_sph && !b_sph) {
								nh_box_box_eval(A, B, a_is_own ? own_size : par_size, a_is_own ? par_size : own_size, ta, tb, bb);
								pp_count = nh_bb_count(bb);
								key = (uint64_t)(bb.swapped ? tb : ta) | ((uint64_t)(bb.swapped ? ta : tb) << 32);
								body_a = bb.swapped ? B.body : A.body; body_b = bb.swapped ? A.body : B.body;
							} else if (a_sph && b_sph) {
								// sphere-sphere: first = earlier in Morton order = pair.y (nudge.cpp:3775-3776)
								pp_count = nh_sphere_sphere(rb, ra, B, A, &single);
								key = (uint64_t)tb | ((uint64_t)ta << 32);
								body_a = B.body; body_b = A.body;
							} else {
								// box-sphere: box is always "a" (nudge.cpp:3746-3751, 3759-3768)
								const nh_xform BX = a_sph ? B : A, SP = a_sph ? A : B;
								const bool box_is_own = a_sph ? !a_is_own : a_is_own;
								pp_count = nh_box_sphere(box_is_own ? own_size : par_size, a_sph ? ra : rb, BX, SP, &single);
								key = (uint64_t)(a_sph ? tb : ta) | ((uint64_t)(a_sph ? ta : tb) << 32);
								body_a = BX.body; body_b = SP.body;
							}
						}
						// what the next still step relies on, per record (still_record)
						{
							const uint64_t was_key = (uint64_t)s_pk[wave][2][lane] | ((uint64_t)s_pk[wave][3][lane] << 32);
							const bool vanished = !hit;
							const bool same_key = was_key == key || vanished;
							bool ok = same_key && pp_count <= 4;
							if (ok && pp_count) ok = (body_a == 0u) != (body_b == 0u);
							if (!ok) { pfail = true; pp_count = 0; bb.kind = 0; atomicAdd(&const_cast<nh_DevState*>(cv.st)->pair_diag[1], 1u); }
						}
						nfail |= pfail;
						// the contacts: staged in the wave's slots (free by now), written out by the wave below
						if (pp_count) {
							nh_ob_slot* const stage = slot + lane * M;
							if (bb.kind == 2) {
								int k = 0;
#pragma unroll
								for (int index = 0; index < 16; ++index) {
									if (((bb.mask >> index) & 1u) && k < pp_count && k < M) {
										nh_contact_out o;
										nh_bb_contact(bb, index, bb.lx[index], bb.ly[index], bb.lz[index], bb.penetration[index], bb.tags[index], o);
										stage[k].c0 = make_float4(o.px, o.py, o.pz, o.penetration); stage[k].c1 = make_float4(o.nx, o.ny, o.nz, o.friction); stage[k].w.x = __uint_as_float(o.feature);
										scid[lane * M + k] = still_base + (uint32_t)k;
										++k;
									}
								}
							} else {
								const nh_contact_out o = bb.kind == 1 ? bb.edge : single;
								stage[0].c0 = make_float4(o.px, o.py, o.pz, o.penetration); stage[0].c1 = make_float4(o.nx, o.ny, o.nz, o.friction); stage[0].w.x = __uint_as_float(o.feature);
								scid[lane * M] = still_base;
							}
						}
						// the record's count, in the record and in its place of the tag order; a CHANGE goes onto the next step's list (k_narrowphase: still_record)
						av.rec[still_rec].count = (uint32_t)pp_count;
						if (st_cnt != (uint32_t)pp_count) {
							nh_DevState* const stw = const_cast<nh_DevState*>(cv.st);
							const uint32_t np2 = sv.parity ^ 1u;
							av.cnt_sorted[bpos] = (uint32_t)pp_count;
							if (!stw->delta_overflow[np2]) {
								const uint32_t k = atomicAdd(&stw->delta_count[np2], 1u);
								if (k < NH_DELTA_MAX) av.delta[np2 * NH_DELTA_MAX + k] = make_int2((int)bpos, pp_count - (int)st_cnt);
								else { stw->delta_overflow[np2] = 1u; if (!av.delta_scan) nfail = true; }          // (more changes than the list holds: the next step scans the counts again -- large worlds, where the host launches the scan -- or is not this form's)
							}
						}
					}
				}
			}
		}
		if (PAIR) {
			// the next step's contacts out to the records' raw slots, by the wave (consecutive lanes on consecutive slots wherever the records are consecutive)
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int k = 0; k < M; ++k) {
				const uint32_t e = 64u * k + lane;
				const uint32_t c = scid[e];
				if (c != 0xFFFFFFFFu) {
					float4* dp = reinterpret_cast<float4*>(av.raw_data + c);
					dp[0] = slot[e].c0; dp[1] = slot[e].c1;
					av.raw_feature[c] = __float_as_uint(slot[e].w.x);
				}
			}
		}
		if (AHEAD) {
			// the wave's share goes to the NEXT step's words (the other parity; cleared by this step's k_xform<true> or, in a step that had none, by its narrowphase)
			nh_DevState* const stw = const_cast<nh_DevState*>(cv.st);
			const uint32_t np = sv.parity ^ 1u;
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				for (int d = 32; d >= 1; d >>= 1) { nmin[k] = min(nmin[k], (uint32_t)__shfl_xor((int)nmin[k], d)); nmax[k] = max(nmax[k], (uint32_t)__shfl_xor((int)nmax[k], d)); }
			}
			for (int d = 32; d >= 1; d >>= 1) ntop = max(ntop, (uint32_t)__shfl_xor((int)ntop, d));
			const unsigned long long bad = __ballot(nfail);
			// (NH_AHEAD_PARTS places, results not awaited: a wave neither queues behind sixteen thousand others on one address nor waits for an answer; the next
			// step's narrowphase puts the places together, with the static world's share)
			uint32_t* const part = &stw->ahead_part[np][(blockIdx.x * NW + wave) & (NH_AHEAD_PARTS - 1u)][0];
			const uint32_t mine_min = lane == 0u ? nmin[0] : (lane == 1u ? nmin[1] : nmin[2]);
			const uint32_t mine_max = lane == 3u ? nmax[0] : (lane == 4u ? nmax[1] : (lane == 5u ? nmax[2] : ntop));
			if (lane < 3u) atomicMin(&part[lane], mine_min);
			else if (lane < 7u) atomicMax(&part[lane], mine_max);
			else if (lane == 7u && bad != 0ull) atomicMax(&stw->ahead_failed_seq, fs.seq + 1u);
		}
		__builtin_amdgcn_wave_barrier();
		if (CONTIG) break;
		x += gridDim.x * blockDim.x;
	} while (x < rounded);
}

// ---- level-scheduled path ------------------------------------------------------------------------------------------------------
// one atomicAdd / atomicMax per workgroup (same-address atomics cost ~12 ns each on this chip)
__device__ __forceinline__ void block_report(nh_DevState* st, uint32_t left, uint32_t lmax, uint32_t round) {
	// progress counters alternate between rounds; each round clears the counter of the next one (nobody touches it meanwhile)
	uint32_t* mine = (round & 1u) ? &st->unleveled_odd : &st->unleveled;
	uint32_t* next = (round & 1u) ? &st->unleveled : &st->unleveled_odd;
	if (blockIdx.x == 0 && threadIdx.x == 0) *next = 0;
	__shared__ uint32_t s_left[4], s_max[4];
	for (int d = 32; d >= 1; d >>= 1) { left += (uint32_t)__shfl_xor((int)left, d); lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d)); }
	if ((threadIdx.x & 63) == 0) { s_left[threadIdx.x >> 6] = left; s_max[threadIdx.x >> 6] = lmax; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t l = s_left[0] + s_left[1] + s_left[2] + s_left[3];
		uint32_t m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
		if (l) atomicAdd(mine, l);
		if (m) atomicMax(&st->levels, m);
	}
}

__global__ __launch_bounds__(256) void k_level_relax(nh_DevState* __restrict__ st, const uint32_t* __restrict__ general_list, const uint32_t* __restrict__ pred_a, const uint32_t* __restrict__ pred_b, uint32_t* __restrict__ level, uint32_t round) {
	uint32_t n = st->general_contacts;
	uint32_t left = 0, lmax = 0;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t c = general_list[k];
		if (__hip_atomic_load(&level[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != NH_UNSET) continue;
		uint32_t pa = pred_a[c], pb = pred_b[c];
		uint32_t la = pa == NH_NONE ? 0u : __hip_atomic_load(&level[pa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		uint32_t lb = pb == NH_NONE ? 0u : __hip_atomic_load(&level[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (la == NH_UNSET || lb == NH_UNSET) { ++left; continue; }
		uint32_t l = (la > lb ? la : lb) + 1u;
		__hip_atomic_store(&level[c], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		lmax = max(lmax, l);
	}
	block_report(st, left, lmax, round);
}


// Default (throughput) order for contacts off the one-body fast path: a deterministic greedy COLOURING of the contact graph
// (two contacts conflict iff they share a dynamic body).  Jones-Plassmann rounds with fixed hashed priorities: a contact whose
// priority beats every still-uncoloured neighbour takes the smallest colour none of its neighbours holds.  Two such contacts
// are never adjacent, so a round is race free in place.  colour+1 is stored where the exact mode stores the dependency level,
// and the same executor runs colour classes in order: a valid Gauss-Seidel sweep (a different one than the reference's; on
// chaotic scenes that is the statistical-parity regime), with ~max-degree classes instead of one level per
// link of the longest dependency chain.
__global__ void k_level_reset_progress(nh_DevState* st) { st->unleveled = 0; st->unleveled_odd = 0; }

__device__ __forceinline__ uint32_t colour_priority(uint32_t c) { uint32_t h = c * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return h; }

__global__ __launch_bounds__(256) void k_colour_round(nh_DevState* __restrict__ st, const uint32_t* __restrict__ general_list, const nh_BodyPair* __restrict__ bodies,
                                                      const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj, uint32_t* __restrict__ level, uint32_t round, uint32_t n_list) {
	// n_list != 0: `general_list` is the compacted list of the contacts that were still uncoloured after an earlier batch of rounds
	uint32_t n = n_list ? n_list : st->general_contacts;
	uint32_t left = 0, lmax = 0;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t c = general_list[k];
		if (__hip_atomic_load(&level[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != NH_UNSET) continue;
		const uint32_t pc = colour_priority(c);
		nh_BodyPair p = bodies[c];
		bool is_max = true;
		unsigned long long used = 0ull;        // colours 1..64 held by neighbours
		bool overflow = false;
		for (int side = 0; side < 2 && is_max; ++side) {
			uint32_t x = side ? p.b : p.a;
			if (!x) continue;
			uint32_t b = off[x], e = off[x + 1];
			for (uint32_t q = b; q < e; ++q) {
				uint32_t m = adj[q];
				if (m == c) continue;
				uint32_t lm = __hip_atomic_load(&level[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (lm == NH_UNSET) {
					uint32_t pm = colour_priority(m);
					if (pm > pc || (pm == pc && m > c)) { is_max = false; break; }
				} else if (lm >= 1 && lm <= 64) {
					used |= 1ull << (lm - 1);
				} else if (lm > 64) {
					overflow = true;
				}
			}
		}
		if (!is_max) { ++left; continue; }
		uint32_t colour;
		if (~used) colour = (uint32_t)__ffsll((long long)~used);          // smallest free colour in 1..64
		else { colour = 65; overflow = true; }
		if (overflow && colour > 64) {
			// more than 64 colours around one contact: fall back to "one past the largest neighbour colour"
			uint32_t mx = 64;
			for (int side = 0; side < 2; ++side) {
				uint32_t x = side ? p.b : p.a;
				if (!x) continue;
				for (uint32_t q = off[x]; q < off[x + 1]; ++q) {
					uint32_t lm = __hip_atomic_load(&level[adj[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (lm != NH_UNSET && lm > mx) mx = lm;
				}
			}
			colour = mx + 1;
		}
		__hip_atomic_store(&level[c], colour, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		lmax = max(lmax, colour);
	}
	block_report(st, left, lmax, round);
}

// counting sort of the general contacts by level / colour.  Levels are few and hot: histogram and cursors are kept per
// workgroup in LDS, global atomics happen once per (workgroup, level).
// Colouring, speculative form (the default; NH_COLOUR_JP=1 selects the Jones-Plassmann rounds above).  A round is two kernels:
//   try     every uncoloured contact picks the smallest colour that none of its FINISHED neighbours holds -> tent[c]   (reads only `level`,
//           which nobody writes meanwhile)
//   settle  two neighbours that picked the same colour in this round are in conflict: the one with the lower hashed priority stays
//           uncoloured, everybody else takes its pick.  A neighbour showing my pick must be from this round (an older one would have
//           been in my `used` set), and its colour reads the same from `level` or `tent`, so the outcome does not depend on timing.
// Most contacts settle in the first round (a pick conflicts only with same-round neighbours), so a handful of rounds replace the ~30 rounds
// in which Jones-Plassmann colours one local maximum per neighbourhood.
// round 0 of a step: last step's colour of every contact that existed then (it comes back with the warm-start impulse: spare word of the
// cache record) is its pick; new contacts make no pick yet (0) and choose in the next round against the settled ones
__global__ __launch_bounds__(256) void k_colour_seed(nh_DevState* __restrict__ st, const uint32_t* __restrict__ list, const nh_CachedContactImpulse* __restrict__ warm, uint32_t* __restrict__ tent) {
	const uint32_t n = st->general_contacts;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t c = list[k];
		const uint32_t old = __float_as_uint(warm[c].unused);
		tent[c] = old >= 1u && old <= NH_MAX_LEVELS ? old : 0u;
	}
}

// The same, and the seeds are FINAL at once (the whole of round 0).  Two contacts that both carry a colour from the cache cannot be in conflict: they
// share a body, so they shared it in the step that wrote the cache -- contacts of one body are written by one step: a sleeping set's entries are kept
// aside and come back together, and a set has no contact with a body outside it -- and that step's colouring was proper.  Only the contacts without a
// colour (new this step) go through the try / settle rounds, against the finished ones.  NH_COLOUR_CHECK_SEEDS=1 runs the seeds through a settle round
// instead (k_colour_seed + k_colour_settle over all contacts, 1.4 ms at 11 M contacts); the outcome must be the same.
__global__ __launch_bounds__(256) void k_colour_seed_final(nh_DevState* __restrict__ st, const uint32_t* __restrict__ list, const nh_CachedContactImpulse* __restrict__ warm,
                                                           uint32_t* __restrict__ level, uint32_t* __restrict__ tent) {
	const uint32_t n = st->general_contacts;
	uint32_t left = 0, lmax = 0;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t c = list[k];
		const uint32_t old = __float_as_uint(warm[c].unused);
		if (old >= 1u && old <= NH_MAX_LEVELS) { level[c] = old; tent[c] = old; lmax = max(lmax, old); }      // (tent: the seed as k_colour_validate reads it)
		else { tent[c] = 0u; ++left; }
	}
	block_report(st, left, lmax, 0u);
}

// The argument above holds for a cache this library wrote for this world.  The cache is caller-owned memory: one written by the reference library,
// one with an uninitialised `unused`, a collider re-pointed to another body between two steps, or two contacts that matched the same cache entry
// would hand two contacts of one body the same colour -- and the sweeps would then update that body's momentum concurrently.  So the seeds get ONE
// cheap look: one lane per general body walks the body's contact list (2 G reads against the ~11 G of a settle round) for a colour held twice; of two
// seeded contacts of a body with the same seed the one with the lower hashed priority becomes uncoloured again (decided on the immutable seeds in
// `tent`, so the outcome does not depend on timing) and is coloured by the try / settle rounds like a new contact.
__global__ __launch_bounds__(256) void k_colour_validate(nh_DevState* __restrict__ st, uint32_t nbodies, const uint8_t* __restrict__ body_class, const uint32_t* __restrict__ off,
                                                         const uint32_t* __restrict__ adj, uint32_t* level, const uint32_t* __restrict__ tent) {
	uint32_t demoted = 0;
	for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < nbodies; x += gridDim.x * blockDim.x) {
		if (x == 0u || body_class[x] != NH_CLS_GENERAL) continue;
		const uint32_t b = off[x], e = off[x + 1];
		unsigned long long seen = 0ull;
		bool clash = false;
		for (uint32_t q = b; q < e; ++q) {
			const uint32_t sc = tent[adj[q]];
			if (!sc) continue;
			if (sc > 64u) { clash = true; continue; }             // (more than 64 colours around one body: take the slow look)
			const unsigned long long bit = 1ull << (sc - 1u);
			clash |= (seen & bit) != 0ull;
			seen |= bit;
		}
		if (!clash) continue;
		for (uint32_t q = b; q < e; ++q) {
			const uint32_t c = adj[q], sc = tent[c];
			if (!sc) continue;
			const uint32_t pc = colour_priority(c);
			for (uint32_t r = b; r < e; ++r) {
				const uint32_t m = adj[r];
				if (m == c || tent[m] != sc) continue;
				const uint32_t pm = colour_priority(m);
				if (pm > pc || (pm == pc && m > c)) { if (atomicExch(&level[c], NH_UNSET) != NH_UNSET) ++demoted; break; }
			}
		}
	}
	for (int d = 32; d >= 1; d >>= 1) demoted += (uint32_t)__shfl_xor((int)demoted, d);
	if (nh_lane() == 0u && demoted) atomicAdd(&st->unleveled, demoted);      // round 0's progress counter (block_report)
}

// the smallest colour none of c's FINISHED neighbours holds
__device__ __forceinline__ uint32_t colour_pick(uint32_t c, const nh_BodyPair* __restrict__ bodies, const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj, const uint32_t* level) {
	const nh_BodyPair p = bodies[c];
	unsigned long long used = 0ull;        // colours 1..64 held by finished neighbours
	uint32_t mx = 64;
	for (int side = 0; side < 2; ++side) {
		const uint32_t x = side ? p.b : p.a;
		if (!x) continue;
		for (uint32_t q = off[x]; q < off[x + 1]; ++q) {
			const uint32_t m = adj[q];
			if (m == c) continue;
			const uint32_t lm = level[m];
			if (lm == NH_UNSET) continue;
			if (lm >= 1 && lm <= 64) used |= 1ull << (lm - 1);
			else if (lm > mx) mx = lm;
		}
	}
	return ~used ? (uint32_t)__ffsll((long long)~used) : mx + 1u;      // more than 64 colours around one contact: one past the largest
}

// does c keep its pick `mine`?  Not if a neighbour holds or picked the same colour and has the higher hashed priority.
__device__ __forceinline__ bool colour_loses(uint32_t c, uint32_t mine, const nh_BodyPair* __restrict__ bodies, const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj,
                                             uint32_t* level, const uint32_t* tent) {
	const uint32_t pc = colour_priority(c);
	const nh_BodyPair p = bodies[c];
	for (int side = 0; side < 2; ++side) {
		const uint32_t x = side ? p.b : p.a;
		if (!x) continue;
		for (uint32_t q = off[x]; q < off[x + 1]; ++q) {
			const uint32_t m = adj[q];
			if (m == c) continue;
			const uint32_t lm = __hip_atomic_load(&level[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t cm = lm != NH_UNSET ? lm : tent[m];       // (tent[m] of a contact finished long ago is stale, but then lm is set)
			if (cm != mine) continue;
			const uint32_t pm = colour_priority(m);
			if (pm > pc || (pm == pc && m > c)) return true;
		}
	}
	return false;
}

__global__ __launch_bounds__(256) void k_colour_try(nh_DevState* __restrict__ st, const uint32_t* __restrict__ list, uint32_t n_list, const nh_BodyPair* __restrict__ bodies,
                                                    const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj, const uint32_t* __restrict__ level, uint32_t* __restrict__ tent) {
	const uint32_t n = n_list ? n_list : st->general_contacts;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t c = list[k];
		if (level[c] != NH_UNSET) continue;
		tent[c] = colour_pick(c, bodies, off, adj, level);
	}
}

__global__ __launch_bounds__(256) void k_colour_settle(nh_DevState* __restrict__ st, const uint32_t* __restrict__ list, uint32_t n_list, const nh_BodyPair* __restrict__ bodies,
                                                       const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj, uint32_t* __restrict__ level, const uint32_t* __restrict__ tent, uint32_t round) {
	const uint32_t n = n_list ? n_list : st->general_contacts;
	uint32_t left = 0, lmax = 0;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t c = list[k];
		if (__hip_atomic_load(&level[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != NH_UNSET) continue;       // finished in an earlier round (only I write my own entry)
		const uint32_t mine = tent[c];
		if (mine == 0u) { ++left; continue; }                 // (a new contact in round 0: no pick yet)
		if (colour_loses(c, mine, bodies, off, adj, level, tent)) { ++left; continue; }
		__hip_atomic_store(&level[c], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		lmax = max(lmax, mine);
	}
	block_report(st, left, lmax, round);
}

// ---- small general sets: the whole colouring and the level order in ONE workgroup -------------------------------------------------------------
// Seeds, try / settle rounds, level histogram, offsets and the scatter into level order are seven kinds of launches and, between the rounds, one host
// round trip each (the host asks whether anybody is still uncoloured).  For a few thousand contacts the round trips ARE the cost (the sample
// application's pile: 0.37 of its 1.0 ms step).  One 1024-thread workgroup does all of it with __syncthreads() between the phases; same rules as the
// grid kernels (colour_pick / colour_loses), so the colouring is the same proper, deterministic one.  Writes st->levels, the level offsets and the
// "level has a full row" flags' zero state where k_level_offsets / k_zero_u32 would.
#define NH_COLOUR_SMALL_MAX 8192u
#define NH_COLOUR_STAGE_WORDS 28672u
__global__ __launch_bounds__(1024) void k_colour_small(nh_DevState* __restrict__ st, const uint32_t* __restrict__ list, const nh_CachedContactImpulse* __restrict__ warm,
                                                       const nh_BodyPair* __restrict__ bodies, const uint32_t* __restrict__ off, const uint32_t* __restrict__ adj,
                                                       uint32_t* level_g, uint32_t* tent_g, uint32_t* __restrict__ hist /* 2 x (NH_MAX_LEVELS + 2) */, uint32_t* __restrict__ order,
                                                       uint32_t nbodies) {
	__shared__ uint32_t s_hist[NH_MAX_LEVELS + 2];
	__shared__ uint32_t s_left, s_max, s_wsum[16];
	// STAGED: the walks below -- for every contact the lists of its two bodies, for every entry the neighbour's colour -- are chains of dependent loads, a dozen entries
	// long; out of L2 every link is most of a microsecond and the kernel 126 us of the sample pile's 780.  When the world is small enough the colours, the picks, the list
	// offsets and the lists themselves are copied to LDS first (112 KB) and the same code walks them there
	__shared__ uint32_t s_stage[NH_COLOUR_STAGE_WORDS];
	const uint32_t n = min(st->general_contacts, NH_COLOUR_SMALL_MAX), T = 1024u, tid = threadIdx.x;
	const uint32_t nc = st->contacts, nb1 = nbodies + 1u, na = nbodies ? off[nbodies] : 0u;
	const bool staged = nbodies != 0u && (unsigned long long)2u * nc + nb1 + na <= (unsigned long long)NH_COLOUR_STAGE_WORDS;          // (the same for every thread)
	uint32_t* level = level_g; uint32_t* tent = tent_g;
	if (staged) {
		level = s_stage; tent = s_stage + nc;
		uint32_t* const s_off = s_stage + 2u * nc; uint32_t* const s_adj = s_off + nb1;
		for (uint32_t i = tid; i < nc; i += T) level[i] = level_g[i];          // (tent: written for every listed contact below before anybody reads it)
		for (uint32_t i = tid; i < nb1; i += T) s_off[i] = off[i];
		for (uint32_t i = tid; i < na; i += T) s_adj[i] = adj[i];
		off = s_off; adj = s_adj;
	}
	if (tid == 0) { s_left = 0; s_max = 0; }
	for (uint32_t l = tid; l < NH_MAX_LEVELS + 2; l += T) { s_hist[l] = 0; hist[NH_MAX_LEVELS + 2 + l] = 0; }
	__syncthreads();
	// round 0: colours that came back with the cache are final (k_colour_seed_final)
	uint32_t lmax = 0, left = 0;
	for (uint32_t k = tid; k < n; k += T) {
		const uint32_t c = list[k];
		const uint32_t old = __float_as_uint(warm[c].unused);
		if (old >= 1u && old <= NH_MAX_LEVELS) { level[c] = old; tent[c] = old; lmax = max(lmax, old); }
		else { tent[c] = 0u; ++left; }
	}
	__syncthreads();
	// the seeds come from caller-owned memory: of two seeded neighbours with the same seed the lower priority one is coloured again (k_colour_validate)
	for (uint32_t k = tid; k < n; k += T) {
		const uint32_t c = list[k], sc = tent[c];
		if (!sc) continue;
		const uint32_t pc = colour_priority(c);
		const nh_BodyPair p = bodies[c];
		bool lose = false;
		for (int side = 0; side < 2 && !lose; ++side) {
			const uint32_t x = side ? p.b : p.a;
			if (!x) continue;
			for (uint32_t q = off[x]; q < off[x + 1]; ++q) {
				const uint32_t m = adj[q];
				if (m == c || tent[m] != sc) continue;
				const uint32_t pm = colour_priority(m);
				if (pm > pc || (pm == pc && m > c)) { lose = true; break; }
			}
		}
		if (lose) { level[c] = NH_UNSET; ++left; }
	}
	if (left) atomicAdd(&s_left, left);
	__syncthreads();
	for (uint32_t round = 1; s_left != 0u && round <= NH_MAX_LEVELS + 8u; ++round) {
		__syncthreads();                                     // (everybody has read s_left)
		if (tid == 0) s_left = 0;
		for (uint32_t k = tid; k < n; k += T) {
			const uint32_t c = list[k];
			if (level[c] == NH_UNSET) tent[c] = colour_pick(c, bodies, off, adj, level);
		}
		__syncthreads();
		left = 0;
		for (uint32_t k = tid; k < n; k += T) {
			const uint32_t c = list[k];
			if (__hip_atomic_load(&level[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != NH_UNSET) continue;
			const uint32_t mine = tent[c];
			if (colour_loses(c, mine, bodies, off, adj, level, tent)) { ++left; continue; }
			__hip_atomic_store(&level[c], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			lmax = max(lmax, mine);
		}
		if (left) atomicAdd(&s_left, left);
		__syncthreads();
	}
	// levels, histogram, exclusive offsets (k_level_hist + k_level_offsets), scatter into level order (k_level_scatter)
	if (lmax) atomicMax(&s_max, lmax);
	for (uint32_t k = tid; k < n; k += T) { const uint32_t l = level[list[k]]; if (l <= NH_MAX_LEVELS) atomicAdd(&s_hist[l], 1u); else st->error = NH_ERR_INVALID; }
	__syncthreads();
	if (tid == 0) { st->levels = s_max; if (s_left) st->error = NH_ERR_INVALID; }
	const uint32_t per = (NH_MAX_LEVELS + 2 + T - 1) / T;
	const uint32_t base = tid * per;
	uint32_t sum = 0;
	for (uint32_t i = 0; i < per; ++i) if (base + i < NH_MAX_LEVELS + 2) sum += s_hist[base + i];
	uint32_t x = sum;
	for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)(tid & 63) >= d) x += y; }
	if ((tid & 63) == 63) s_wsum[tid >> 6] = x;
	__syncthreads();
	uint32_t run = x - sum;
	for (uint32_t w = 0; w < (tid >> 6); ++w) run += s_wsum[w];
	for (uint32_t i = 0; i < per; ++i) if (base + i < NH_MAX_LEVELS + 2) { const uint32_t v = s_hist[base + i]; hist[base + i] = run; s_hist[base + i] = run; run += v; }
	__syncthreads();
	for (uint32_t k = tid; k < n; k += T) { const uint32_t c = list[k]; const uint32_t l = level[c]; if (l <= NH_MAX_LEVELS) order[atomicAdd(&s_hist[l], 1u)] = c; if (staged) level_g[c] = l; }
}

// the contacts of `list` that are still uncoloured, in any order (a round's outcome does not depend on who looks at which contact).
// A workgroup owns a contiguous chunk: it counts, reserves its output range with ONE atomic on the global cursor (same-address atomics are
// ~12 ns each, serialised), and fills the range through a cursor in LDS.
__global__ __launch_bounds__(256) void k_colour_compact(nh_DevState* __restrict__ st, const uint32_t* __restrict__ list, uint32_t n_list, const uint32_t* __restrict__ level, uint32_t* __restrict__ out) {
	__shared__ uint32_t s_count, s_base, s_cursor;
	const uint32_t n = n_list ? n_list : st->general_contacts;
	const uint32_t chunk = (n + gridDim.x - 1u) / gridDim.x;
	const uint32_t begin = min(blockIdx.x * chunk, n), end = min(begin + chunk, n);
	if (threadIdx.x == 0) { s_count = 0; s_cursor = 0; }
	__syncthreads();
	uint32_t mine = 0;
	for (uint32_t k = begin + threadIdx.x; k < end; k += 256) mine += level[list[k]] == NH_UNSET ? 1u : 0u;
	for (int d = 32; d >= 1; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d);
	if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_count, mine);
	__syncthreads();
	if (threadIdx.x == 0) s_base = s_count ? atomicAdd(&st->colour_cursor, s_count) : 0u;
	__syncthreads();
	const uint32_t base = s_base;
	for (uint32_t k0 = begin; k0 < end; k0 += 256) {              // block-uniform trip count
		const uint32_t k = k0 + threadIdx.x;
		uint32_t c = 0;
		bool keep = false;
		if (k < end) { c = list[k]; keep = level[c] == NH_UNSET; }
		const uint32_t slot = nh_wave_reserve1(&s_cursor, keep);
		if (keep) out[base + slot] = c;
	}
}

__global__ __launch_bounds__(256) void k_level_hist(nh_DevState* __restrict__ st, const uint32_t* __restrict__ general_list, const uint32_t* __restrict__ level, uint32_t* __restrict__ hist) {
	__shared__ uint32_t h[NH_MAX_LEVELS + 2];
	for (uint32_t l = threadIdx.x; l < NH_MAX_LEVELS + 2; l += blockDim.x) h[l] = 0;
	__syncthreads();
	uint32_t n = st->general_contacts;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t l = level[general_list[k]];
		if (l <= NH_MAX_LEVELS) atomicAdd(&h[l], 1u); else st->error = NH_ERR_INVALID;
	}
	__syncthreads();
	for (uint32_t l = threadIdx.x; l < NH_MAX_LEVELS + 2; l += blockDim.x) if (h[l]) atomicAdd(&hist[l], h[l]);
}

__global__ __launch_bounds__(1024) void k_level_offsets(uint32_t* __restrict__ hist /* NH_MAX_LEVELS+2: in counts, out exclusive offsets */, uint32_t* __restrict__ cursor) {
	__shared__ uint32_t wsum[16];
	const uint32_t per = (NH_MAX_LEVELS + 2 + 1023) / 1024;
	uint32_t base = threadIdx.x * per, s = 0;
	for (uint32_t i = 0; i < per; ++i) if (base + i < NH_MAX_LEVELS + 2) s += hist[base + i];
	uint32_t x = s;
	for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
	if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
	__syncthreads();
	uint32_t woff = 0;
	for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
	uint32_t run = woff + x - s;
	for (uint32_t i = 0; i < per; ++i) if (base + i < NH_MAX_LEVELS + 2) { uint32_t v = hist[base + i]; hist[base + i] = run; cursor[base + i] = run; run += v; }
}

__global__ __launch_bounds__(256) void k_level_scatter(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ general_list, const uint32_t* __restrict__ level, uint32_t* __restrict__ cursor, uint32_t* __restrict__ order) {
	__shared__ uint32_t h[NH_MAX_LEVELS + 2];
	for (uint32_t l = threadIdx.x; l < NH_MAX_LEVELS + 2; l += blockDim.x) h[l] = 0;
	__syncthreads();
	uint32_t n = st->general_contacts;
	// each workgroup owns a contiguous chunk so that both passes see the same contacts
	uint32_t chunk = (n + gridDim.x - 1) / gridDim.x;
	uint32_t begin = min(blockIdx.x * chunk, n), end = min(begin + chunk, n);
	for (uint32_t k = begin + threadIdx.x; k < end; k += blockDim.x) {
		uint32_t l = level[general_list[k]];
		if (l <= NH_MAX_LEVELS) atomicAdd(&h[l], 1u);
	}
	__syncthreads();
	for (uint32_t l = threadIdx.x; l < NH_MAX_LEVELS + 2; l += blockDim.x) { uint32_t c = h[l]; h[l] = c ? atomicAdd(&cursor[l], c) : 0u; }
	__syncthreads();
	for (uint32_t k = begin + threadIdx.x; k < end; k += blockDim.x) {
		uint32_t c = general_list[k];
		uint32_t l = level[c];
		if (l <= NH_MAX_LEVELS) order[atomicAdd(&h[l], 1u)] = c;
	}
}

// ---- general contacts: everything per contact lives in LEVEL ORDER (position k in level_order) ------------------------------------
// rows as structure-of-float4 (chunk q of contact k at rows4[q * stride + k]: a wave's ten row loads are ten contiguous
// kilobytes), the body pair, the solver state.  The sweeps then stream rows / pairs / states and only gather the two momentum
// records (32 B each; mass_inverse rides in momentum.unused0 like in the reference, nudge.cpp:4198 -- k_adj_sort put it there).
__device__ __forceinline__ void store_row_soa(float4* __restrict__ rows4, size_t stride, uint32_t k, const nh_row& r) {
	const float4* src = reinterpret_cast<const float4*>(&r);
#pragma unroll
	for (int q = 0; q < 10; ++q) rows4[(size_t)q * stride + k] = src[q];
}

__device__ __forceinline__ void load_row_soa(const float4* __restrict__ rows4, size_t stride, uint32_t k, nh_row& r) {
	float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
	for (int q = 0; q < 10; ++q) dst[q] = rows4[(size_t)q * stride + k];
}

__device__ __forceinline__ nh_vel load_vel_m(const nh_BodyMomentum* __restrict__ momentum, uint32_t i) {
	const float4 m0 = reinterpret_cast<const float4*>(momentum + i)[0], m1 = reinterpret_cast<const float4*>(momentum + i)[1];
	nh_vel v = { m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w };
	return v;
}

__device__ __forceinline__ void store_vel_m(nh_BodyMomentum* __restrict__ momentum, uint32_t i, const nh_vel& v) {
	reinterpret_cast<float4*>(momentum + i)[0] = make_float4(v.vx, v.vy, v.vz, v.mass_inverse);
	reinterpret_cast<float4*>(momentum + i)[1] = make_float4(v.wx, v.wy, v.wz, v.w_unused);
}

// Rows of the general contacts, in level order.  A contact that qualifies (nh_side_is_radial on both sides: sphere contacts) and is not
// solved in the reference's exact order gets the 12-float radial row (nh_rowr: chunks 0..2 of its column in `rows4`), flagged in bit 31 of
// gpair.x; everything else the full 40-float row.
#define NH_RADIAL_BIT 0x80000000u
__device__ __forceinline__ void store_rowr_soa(float4* __restrict__ rows4, size_t stride, uint32_t k, const nh_rowr& r) {
	const float4* src = reinterpret_cast<const float4*>(&r);
#pragma unroll
	for (int q = 0; q < 3; ++q) rows4[(size_t)q * stride + k] = src[q];
}

__device__ __forceinline__ void load_rowr_soa(const float4* __restrict__ rows4, size_t stride, uint32_t k, nh_rowr& r) {
	float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
	for (int q = 0; q < 3; ++q) dst[q] = rows4[(size_t)q * stride + k];
}

__global__ __launch_bounds__(256) void k_rows_general(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ order,
                                                      const nh_Contact* __restrict__ contacts, const nh_BodyPair* __restrict__ bodies,
                                                      const nh_Transform* __restrict__ xf, const nh_BodyProperties* __restrict__ props,
                                                      float4* __restrict__ rows4, size_t stride, uint2* __restrict__ gpair, uint32_t allow_radial, uint32_t static_inert,
                                                      const uint32_t* __restrict__ level, uint32_t* __restrict__ level_has_full,
                                                      const nh_CachedContactImpulse* __restrict__ warm_in, float4* __restrict__ gstates, uint32_t* __restrict__ lpair, uint32_t packed_end) {
	// warm_in (blocked solver, nh_blocks.h): the cached impulse of every contact is copied to where its solver state will be, so that the warm start
	// streams the same arrays as a sweep; lpair: the contact's local index pair learns whether its row is radial
	uint32_t n = st->general_contacts;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		uint32_t c = order[k];
		nh_BodyPair p = bodies[c];
		nh_Contact ct = contacts[c];
		nh_contact_in ci = { ct.position[0], ct.position[1], ct.position[2], ct.penetration, ct.normal[0], ct.normal[1], ct.normal[2], ct.friction };
		const float4 ta0 = reinterpret_cast<const float4*>(xf + p.a)[0], tb0 = reinterpret_cast<const float4*>(xf + p.b)[0];
		const float4 pa = *reinterpret_cast<const float4*>(props + p.a), pb = *reinterpret_cast<const float4*>(props + p.b);
		const nh_f3 apos = nh_make3(ta0.x, ta0.y, ta0.z), bpos = nh_make3(tb0.x, tb0.y, tb0.z);
		const bool ia = static_inert && p.a == 0u, ib = static_inert && p.b == 0u;
		const bool radial = allow_radial &&
			nh_side_is_radial(ci.px - apos.x, ci.py - apos.y, ci.pz - apos.z, ci.nx, ci.ny, ci.nz, pa.x, pa.y, pa.z, ia) &&
			nh_side_is_radial(ci.px - bpos.x, ci.py - bpos.y, ci.pz - bpos.z, ci.nx, ci.ny, ci.nz, pb.x, pb.y, pb.z, ib);
		if (radial) {
			nh_rowr r;
			nh_build_rowr(ci, apos, bpos, pa.x, pb.x, pa.w, pb.w, ia, ib, r);
			store_rowr_soa(rows4, stride, k, r);
		} else {
			nh_inertia IA = body_inertia(xf, props, p.a), IB = body_inertia(xf, props, p.b);
			nh_row r;
			nh_build_row(ci, apos, bpos, IA, IB, pa.w, pb.w, r);
			store_row_soa(rows4, stride, k, r);
			level_has_full[level ? level[c] : 1u] = 1u;             // this level / colour class needs the kernel that knows full rows (plain store of a constant; per-block colours: one flag for all)
		}
		gpair[k] = make_uint2(p.a | (radial ? NH_RADIAL_BIT : 0u), p.b);
		if (warm_in) {
			// (contacts owned by a block, k < packed_end, keep their three state floats packed: 12 bytes each in the same memory)
			const float4 w = *reinterpret_cast<const float4*>(warm_in + c);
			if (k < packed_end) { float* ps = reinterpret_cast<float*>(gstates) + 3 * (size_t)k; ps[0] = w.x; ps[1] = w.y; ps[2] = w.z; }
			else gstates[k] = w;
		}
		if (lpair && radial) lpair[k] |= NH_BLK_RADIAL;
	}
}

// RADIAL_ONLY: every contact of the class has a radial row (the host knows from k_rows_general's per-class flags): the kernel without the
// 40-float path needs a third of the registers and runs at full occupancy -- what a gather / scatter bound sweep wants.
template<bool WARM, bool RADIAL_ONLY = false>
__global__ __launch_bounds__(256) void k_level_exec(const uint32_t* __restrict__ order, uint32_t begin, uint32_t end,
                                                    const uint2* __restrict__ gpair, const nh_CachedContactImpulse* __restrict__ cached,
                                                    nh_BodyMomentum* __restrict__ momentum,
                                                    const float4* __restrict__ rows4, size_t stride, float4* __restrict__ gstates, uint32_t keep_w = 0u) {
	// keep_w (blocked solver: residual contacts on its block-ordered momentum copy): the spare word of a momentum record is not ours to clear
	for (uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x; k < end; k += gridDim.x * blockDim.x) {
		uint2 p = gpair[k];
		const bool radial = RADIAL_ONLY || (p.x & NH_RADIAL_BIT) != 0u;
		p.x &= ~NH_RADIAL_BIT;
		nh_vel a = load_vel_m(momentum, p.x), b = load_vel_m(momentum, p.y);
		const float aw = a.w_unused, bw = b.w_unused;
		nh_state s;
		float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (WARM) w = cached ? *reinterpret_cast<const float4*>(cached + order[k]) : gstates[k];      // (blocked solver: the cached impulse waits in the state's place)
		else { const float4 sv = gstates[k]; s.normal = sv.x; s.friction_x = sv.y; s.friction_y = sv.z; }
		if (radial) {
			nh_rowr r;
			load_rowr_soa(rows4, stride, k, r);
			if (WARM) nh_warm_start_r(r, w.x, w.y, w.z, a, b, s); else nh_apply_radial(r, s, a, b);
		} else if (!RADIAL_ONLY) {
			nh_row r;
			load_row_soa(rows4, stride, k, r);
			if (WARM) nh_warm_start(r, w.x, w.y, w.z, a, b, s); else nh_apply_one(r, s, a, b);
		}
		gstates[k] = make_float4(s.normal, s.friction_x, s.friction_y, 0.0f);
		if (keep_w) { a.w_unused = aw; b.w_unused = bw; }
		if (p.x) store_vel_m(momentum, p.x, a);      // body 0 is the static world: never written
		if (p.y) store_vel_m(momentum, p.y, b);
	}
}

// ---- small general sets: ONE workgroup, constraint rows resident in LDS across the warm start and all sweeps ------------------------------------
// The level / colour schedule needs a synchronisation between consecutive levels.  Across workgroups that is a kernel boundary (a device-wide
// barrier costs more than one on this part); inside ONE workgroup it is __syncthreads().  So when the whole general set fits one compute unit's
// LDS -- 704 rows of 160 B (or 1792 radial rows of 48 B), their states and body pairs: 155 KB of the 160 KB -- one workgroup loads the rows ONCE and walks
// levels x sweeps with barriers only: the north star's "constraints staged in LDS across solver iterations".  Momentum stays in global memory:
// the waves of a workgroup share their compute unit's L1, stores write through it, and __syncthreads() orders them -- the ordinary way a
// workgroup communicates through global memory.  Arithmetic and order are exactly those of k_level_exec (levels in order, any order inside a level).
#define NH_RES_MAX_FULL 704u            // general sets with a full (40-float) row: 10 x 704 sixteen-byte words of row storage
#define NH_RES_MAX_RADIAL 1792u         // all rows radial (12 floats): 3 x 1792 words
#define NH_RES_ROW_WORDS 7168u
#define NH_RES_THREADS 1024
__global__ __launch_bounds__(NH_RES_THREADS) void k_solve_resident(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ level_off, uint32_t levels,
                                                                   const uint32_t* __restrict__ order, const uint2* __restrict__ gpair, const nh_CachedContactImpulse* __restrict__ cached,
                                                                   nh_BodyMomentum* momentum, const float4* __restrict__ rows4, size_t stride, float4* __restrict__ gstates,
                                                                   uint32_t iterations, uint32_t warm, uint32_t cap /* NH_RES_MAX_FULL or _RADIAL: the row stride in LDS */) {
	__shared__ float4 s_rows[NH_RES_ROW_WORDS];          // chunk q of contact k at s_rows[q * cap + k]
	__shared__ float4 s_state[NH_RES_MAX_RADIAL];
	__shared__ uint2 s_pair[NH_RES_MAX_RADIAL];
	const uint32_t G = min(st->general_contacts, cap);
	if (levels == 0u) levels = st->levels;               // (the one-workgroup colouring leaves the count on the device: no host round trip)
	for (uint32_t k = threadIdx.x; k < G; k += NH_RES_THREADS) {
		const uint2 p = gpair[k];
		s_pair[k] = p;
		const uint32_t chunks = (p.x & NH_RADIAL_BIT) ? 3u : 10u;
		for (uint32_t q = 0; q < chunks; ++q) s_rows[q * cap + k] = rows4[(size_t)q * stride + k];
		s_state[k] = warm ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : gstates[k];
	}
	__syncthreads();
	auto one = [&](uint32_t k, bool warm_start) {
		uint2 p = s_pair[k];
		const bool radial = (p.x & NH_RADIAL_BIT) != 0u;
		p.x &= ~NH_RADIAL_BIT;
		nh_vel a = load_vel_m(momentum, p.x), b = load_vel_m(momentum, p.y);
		nh_state s;
		float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (warm_start) w = *reinterpret_cast<const float4*>(cached + order[k]);
		else { const float4 sv = s_state[k]; s.normal = sv.x; s.friction_x = sv.y; s.friction_y = sv.z; }
		if (radial) {
			nh_rowr r;
			float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
			for (uint32_t q = 0; q < 3; ++q) dst[q] = s_rows[q * cap + k];
			if (warm_start) nh_warm_start_r(r, w.x, w.y, w.z, a, b, s); else nh_apply_radial(r, s, a, b);
		} else {
			nh_row r;
			float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
			for (uint32_t q = 0; q < 10; ++q) dst[q] = s_rows[q * cap + k];
			if (warm_start) nh_warm_start(r, w.x, w.y, w.z, a, b, s); else nh_apply_one(r, s, a, b);
		}
		s_state[k] = make_float4(s.normal, s.friction_x, s.friction_y, 0.0f);
		if (p.x) store_vel_m(momentum, p.x, a);      // body 0 is the static world: never written
		if (p.y) store_vel_m(momentum, p.y, b);
	};
	if (warm) {
		for (uint32_t l = 1; l <= levels; ++l) {
			const uint32_t b = level_off[l], e = min(level_off[l + 1], G);
			for (uint32_t k = b + threadIdx.x; k < e; k += NH_RES_THREADS) one(k, true);
			__syncthreads();
		}
	}
	for (uint32_t it = 0; it < iterations; ++it)
		for (uint32_t l = 1; l <= levels; ++l) {
			const uint32_t b = level_off[l], e = min(level_off[l + 1], G);
			for (uint32_t k = b + threadIdx.x; k < e; k += NH_RES_THREADS) one(k, false);
			__syncthreads();
		}
	for (uint32_t k = threadIdx.x; k < G; k += NH_RES_THREADS) gstates[k] = s_state[k];
}

// ---- update_cached_impulses (nudge.cpp:4857-4884) --------------------------------------------------------------------------------
// (contacts on the one-body fast path export their impulse at the end of every nh_apply_impulses call)
// ---- mid-size general sets in small worlds: ONE workgroup, every body's momentum resident in LDS ------------------------------------------------
// When the rows do not fit one compute unit's LDS (k_solve_resident) but the WORLD is small -- at most NH_RESB_MAX_BODIES bodies: the sample
// application's 1536-body pile -- the other half of the per-sweep traffic does: one workgroup keeps the momentum of every body in LDS (32 B each) for
// the warm start and all sweeps, streams rows, pairs and states from L2, and separates levels with __syncthreads() instead of kernel boundaries.
// What a level costs is then one LDS round trip, not a memory round trip, PROVIDED its rows are already there: every thread fetches its contact of
// the NEXT level (pair, row, state: they do not depend on the sweep in progress) before it processes the current one.  Same arithmetic and level order
// as k_level_exec; a thread always handles the same contacts, so a state it wrote is its own to read back.
#define NH_RESB_MAX_BODIES 3584u
#define NH_RESB_MAX_LEVELS 2046u
#define NH_RESB_THREADS 512
struct nh_resb_item { float4 row[10]; float4 sv; uint2 p; uint32_t k; bool valid; };
__global__ __launch_bounds__(NH_RESB_THREADS) void k_solve_resident_bodies(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ level_off, const uint32_t* __restrict__ level_full,
                                                                           uint32_t levels, const uint32_t* __restrict__ order, const uint2* __restrict__ gpair,
                                                                           const nh_CachedContactImpulse* __restrict__ cached, nh_BodyMomentum* momentum, uint32_t nbodies,
                                                                           const float4* __restrict__ rows4, size_t stride, float4* gstates, uint32_t iterations, uint32_t warm) {
	__shared__ float4 s_m0[NH_RESB_MAX_BODIES], s_m1[NH_RESB_MAX_BODIES];
	__shared__ uint32_t s_off[NH_RESB_MAX_LEVELS + 2];
	__shared__ uint8_t s_full[NH_RESB_MAX_LEVELS + 2];
	const uint32_t G = st->general_contacts, T = NH_RESB_THREADS, tid = threadIdx.x;
	if (levels == 0u) levels = st->levels;               // (the one-workgroup colouring leaves the count on the device: no host round trip)
	if (levels > NH_RESB_MAX_LEVELS) { if (tid == 0) const_cast<nh_DevState*>(st)->error = NH_ERR_INVALID; return; }
	for (uint32_t x = tid; x < nbodies; x += T) { s_m0[x] = reinterpret_cast<const float4*>(momentum + x)[0]; s_m1[x] = reinterpret_cast<const float4*>(momentum + x)[1]; }
	for (uint32_t l = tid; l <= levels + 1u; l += T) { s_off[l] = min(level_off[l], G); s_full[l] = (uint8_t)(level_full[l] != 0u); }
	__syncthreads();
	// the contact this thread handles first in level l: loads issued here, consumed one level later
	auto fetch = [&](uint32_t l, bool with_state, nh_resb_item& it) {
		const uint32_t k = s_off[l] + tid;
		it.valid = k < s_off[l + 1];
		it.k = k;
		if (it.valid) {
			it.p = gpair[k];
			const uint32_t chunks = s_full[l] ? 10u : 3u;            // (a level without a full row holds radial rows only: three chunks)
#pragma unroll
			for (uint32_t q = 0; q < 10; ++q) if (q < chunks) it.row[q] = rows4[(size_t)q * stride + k];
			if (with_state) it.sv = gstates[k];
		}
	};
	auto process = [&](const nh_resb_item& it, bool warm_start, bool state_fetched) {
		uint2 p = it.p;
		const bool radial = (p.x & NH_RADIAL_BIT) != 0u;
		p.x &= ~NH_RADIAL_BIT;
		const float4 a0 = s_m0[p.x], a1 = s_m1[p.x], b0 = s_m0[p.y], b1 = s_m1[p.y];
		nh_vel a = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w }, b = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
		nh_state s;
		float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (warm_start) w = *reinterpret_cast<const float4*>(cached + order[it.k]);
		else { const float4 sv = state_fetched ? it.sv : gstates[it.k]; s.normal = sv.x; s.friction_x = sv.y; s.friction_y = sv.z; }
		if (radial) {
			nh_rowr r;
			float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
			for (uint32_t q = 0; q < 3; ++q) dst[q] = it.row[q];
			if (warm_start) nh_warm_start_r(r, w.x, w.y, w.z, a, b, s); else nh_apply_radial(r, s, a, b);
		} else {
			nh_row r;
			float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
			for (uint32_t q = 0; q < 10; ++q) dst[q] = it.row[q];
			if (warm_start) nh_warm_start(r, w.x, w.y, w.z, a, b, s); else nh_apply_one(r, s, a, b);
		}
		gstates[it.k] = make_float4(s.normal, s.friction_x, s.friction_y, 0.0f);
		if (p.x) { s_m0[p.x] = make_float4(a.vx, a.vy, a.vz, a.mass_inverse); s_m1[p.x] = make_float4(a.wx, a.wy, a.wz, a.w_unused); }      // body 0 is the static world: never written
		if (p.y) { s_m0[p.y] = make_float4(b.vx, b.vy, b.vz, b.mass_inverse); s_m1[p.y] = make_float4(b.wx, b.wy, b.wz, b.w_unused); }
	};
	// the passes as one sequence of level steps: [warm start: levels 1..L] then `iterations` x [levels 1..L]
	const uint32_t passes = (warm ? 1u : 0u) + iterations;
	if (passes == 0u || levels == 0u) return;
	const bool prefetch_state = levels > 1u;            // (with one level the next step's state is the one being written right now)
	nh_resb_item cur, nxt;
	fetch(1u, prefetch_state && !warm, cur);
	for (uint32_t pass = 0; pass < passes; ++pass) {
		const bool warm_pass = warm && pass == 0u;
		for (uint32_t l = 1; l <= levels; ++l) {
			const bool last = l == levels;
			const bool more = !last || pass + 1u < passes;
			if (more) fetch(last ? 1u : l + 1u, prefetch_state && !(warm_pass && !last), nxt);        // (the next step is a sweep unless it is still the warm start)
			if (cur.valid) process(cur, warm_pass, prefetch_state && !warm_pass);
			// a level with more contacts than threads: the rest, fetched in place
			for (uint32_t k = s_off[l] + tid + T; k < s_off[l + 1]; k += T) {
				nh_resb_item extra;
				extra.k = k; extra.valid = true; extra.p = gpair[k];
#pragma unroll
				for (uint32_t q = 0; q < 10; ++q) if (q < (s_full[l] ? 10u : 3u)) extra.row[q] = rows4[(size_t)q * stride + k];
				process(extra, warm_pass, false);
			}
			__syncthreads();
			if (more) cur = nxt;
		}
	}
	for (uint32_t x = tid + (tid == 0 ? T : 0u); x < nbodies; x += T) {            // (x = 0, the static world, is never written back)
		reinterpret_cast<float4*>(momentum + x)[0] = s_m0[x]; reinterpret_cast<float4*>(momentum + x)[1] = s_m1[x];
	}
}

__global__ __launch_bounds__(256) void k_update_impulses(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ order, const float4* __restrict__ rows4, size_t stride,
                                                         const float4* __restrict__ gstates, nh_CachedContactImpulse* __restrict__ out, const uint32_t* __restrict__ colour,
                                                         const uint2* __restrict__ gpair, uint32_t packed_end = 0u) {
	// packed_end (blocked solver): the states of the contacts owned by a block, k < packed_end, are packed (three floats each, same memory)
	// colour != nullptr (default solver order): the contact's colour rides in the spare word of its cached impulse, so that the next
	// step's colouring can start from it (k_colour_seed)
	uint32_t n = st->general_contacts;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		nh_state s;
		if (k < packed_end) { const float* ps = reinterpret_cast<const float*>(gstates) + 3 * (size_t)k; s.normal = ps[0]; s.friction_x = ps[1]; s.friction_y = ps[2]; }
		else { const float4 sv = gstates[k]; s.normal = sv.x; s.friction_x = sv.y; s.friction_y = sv.z; }
		float ox, oy, oz;
		if (gpair[k].x & NH_RADIAL_BIT) {
			nh_rowr r;
			load_rowr_soa(rows4, stride, k, r);
			nh_export_impulse_r(r, s, ox, oy, oz);
		} else {
			// (only n, u, v of the full row are needed: chunks 1..3 hold them)
			nh_row r;
			load_row_soa(rows4, stride, k, r);
			nh_export_impulse(r, s, ox, oy, oz);
		}
		const uint32_t c = order[k];
		*reinterpret_cast<float4*>(out + c) = make_float4(ox, oy, oz, colour ? __uint_as_float(colour[c]) : 0.0f);
	}
}

#include "nh_blocks.h"

// =====================================================================================================================================
static nh_ContactImpulseData* new_impulse_data(nh_context* ctx) {
	if (ctx->impulse_ring.empty()) { ctx->impulse_ring.resize(64, nullptr); ctx->constraint_ring.resize(64, nullptr); }
	uint32_t k = ctx->ring_pos % 64;
	if (!ctx->impulse_ring[k]) ctx->impulse_ring[k] = new nh_ContactImpulseData();
	return ctx->impulse_ring[k];
}

static nh_ContactConstraintData* new_constraint_data(nh_context* ctx) {
	if (ctx->constraint_ring.empty()) { ctx->impulse_ring.resize(64, nullptr); ctx->constraint_ring.resize(64, nullptr); }
	uint32_t k = ctx->ring_pos % 64;
	if (!ctx->constraint_ring[k]) ctx->constraint_ring[k] = new nh_ContactConstraintData();
	return ctx->constraint_ring[k];
}

extern "C" int nh_apply_gravity_damping(nh_context* ctx, const nh_ActiveBodies* active_bodies, const nh_BodyData* bodies,
                                        float time_step, const float gravity[3], float damping_rate) {
	if (!ctx || !active_bodies || !bodies || !gravity) return NH_ERR_INVALID;
	const bool follows_collide = ctx->gravity_may_overlap, after_collide = ctx->after_collide;
	// (a still step stays speculative across the gravity call that directly follows its nh_collide on the same arrays -- the one the solver will fuse)
	const bool in_sequence = ctx->still.active && !ctx->still.resolved && after_collide && !ctx->pending && bodies->momentum == ctx->still.bodies.momentum &&
	                         active_bodies->indices == ctx->still.lay_active;
	{ int rc = nh_flush_pending(ctx, false, in_sequence); if (rc) return rc; }
	float damping = 1.0f - time_step * damping_rate;
	if ((ctx->flags & NH_FLAG_FUSED_STEP) && after_collide) {
		// applied by the fused solver when it loads a body (first_apply), by k_gravity_rest for the others; any other entry point settles it first
		ctx->grav.pending = true;
		ctx->grav.gx_dt = gravity[0] * time_step; ctx->grav.gy_dt = gravity[1] * time_step; ctx->grav.gz_dt = gravity[2] * time_step;
		ctx->grav.damping = damping; ctx->grav.time_step = time_step;
		ctx->grav.active = active_bodies->indices; ctx->grav.momentum = bodies->momentum; ctx->grav.body_count = bodies->count;
		return NH_OK;
	}
	// directly after nh_collide (the sample's loop, example/main.cpp:286-305): on the side stream, behind the kernel that wrote the active list and
	// beside the tag sort / contact gather still queued on the caller's stream (they never touch momentum); the caller's stream then waits
	const bool overlap = follows_collide && ctx->side != nullptr;
	hipStream_t main_stream = ctx->stream;
	if (overlap) { NH_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_active, 0)); ctx->stream = ctx->side; }
	NH_LAUNCH(ctx, "gravity_damping", k_gravity, nh_grid_for(bodies->count, 256, 2048), 256, ctx->d_state, active_bodies->indices, bodies->momentum,
	          gravity[0] * time_step, gravity[1] * time_step, gravity[2] * time_step, damping);
	if (overlap) {
		ctx->stream = main_stream;
		NH_HIP_CHECK(ctx, hipEventRecord(ctx->ev_gravity, ctx->side));
		NH_HIP_CHECK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_gravity, 0));
	}
	return NH_OK;
}

extern "C" int nh_advance(nh_context* ctx, const nh_ActiveBodies* active_bodies, const nh_BodyData* bodies, float time_step) {
	if (!ctx || !active_bodies || !bodies) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	ctx->advance_count++;
	if (ctx->adv.done) {
		// NH_FLAG_FUSED_STEP: the one-body solver advanced its own bodies with the time step gravity was given
		ctx->adv.done = false;
		if (nh_asuint(time_step) != nh_asuint(ctx->adv.time_step)) return NH_ERR_INVALID;
		if (ctx->adv.rest)
			NH_LAUNCH(ctx, "advance_rest", k_advance_rest, nh_grid_for(bodies->count, 256, 2048), 256, ctx->d_state, active_bodies->indices, ctx->adv.body_class, bodies->transforms,
			          bodies->momentum, bodies->idle_counters, time_step);
		return ctx->stream_state.every ? nh_stream_after_advance(ctx) : NH_OK;
	}
	NH_LAUNCH(ctx, "advance", k_advance, nh_grid_for(bodies->count, 256, 2048), 256, ctx->d_state, active_bodies->indices, bodies->transforms,
	          bodies->momentum, bodies->idle_counters, time_step);
	return ctx->stream_state.every ? nh_stream_after_advance(ctx) : NH_OK;
}

// cached impulses of sleeping pairs are kept aside (nudge.cpp:4064-4101); skipped entirely when nothing sleeps (culled = 0 since nh_collide)
static void run_cull(nh_context* ctx, nh_ContactImpulseData* d, uint32_t sleeping_on_host) {
	if (!d->cull_pending) return;
	d->cull_pending = false;
	if (sleeping_on_host == 0) return;
	nh_DevState* st = ctx->d_state;
	const uint32_t ccap = d->cache_capacity;
	NH_LAUNCH(ctx, "cull_flags", k_cull_flags, nh_grid_for(ccap, 256, 4096), 256, st, d->ctags, d->sleeping_pairs, d->cull_flags);
	nh_scan_u32(ctx, d->cull_flags, d->cull_flags, &st->cache, 1, d->cull_tmp, &st->culled, &st->sleeping);   // +1: the sentinel, so scan[j+1] exists for every j
	NH_LAUNCH(ctx, "cull_write", k_cull_write, nh_grid_for(ccap, 256, 4096), 256, st, d->cull_flags, d->cull_flags, d->ctags, d->cfeatures, d->cdata,
	          d->culled_tags, d->culled_features, d->culled_data);
}

extern "C" int nh_read_cached_impulses(nh_context* ctx, const nh_ContactCache* cache, const nh_ContactData* contacts, nh_Arena* memory, nh_ContactImpulseData** out) {
	if (!ctx || !cache || !contacts || !memory || !out) return NH_ERR_INVALID;
	{
		// still step: only on the cache and the contact list it was launched for
		const nh_StillStep& ss = ctx->still;
		const bool in_sequence = ss.active && !ss.resolved && !ctx->pending && cache->tags == ss.cache_tags && cache->features == ss.cache_features && cache->data == ss.cache_data &&
		                         cache->capacity == ss.cache_capacity && contacts->data == ss.lay_contacts.data && contacts->tags == ss.lay_contacts.tags;
		int rc = nh_flush_pending(ctx, true, in_sequence); if (rc) return rc;
	}
	const uint32_t kcap = contacts->capacity, ccap = cache->capacity;
	int err = NH_OK;
	ctx->ring_pos++;
	nh_ContactImpulseData* d = new_impulse_data(ctx);
	d->data = nh_arena_array<nh_CachedContactImpulse>(memory, kcap, &err);
	d->culled_tags = nh_arena_array<uint64_t>(memory, ccap, &err);
	d->culled_features = nh_arena_array<uint32_t>(memory, ccap, &err);
	d->culled_data = nh_arena_array<nh_CachedContactImpulse>(memory, ccap, &err);
	d->capacity = kcap; d->culled_capacity = ccap;
	// culling of cached impulses of sleeping pairs (nudge.cpp:4064-4101) only has work when something sleeps; the scratch for it is
	// reserved here, the kernels run once the host knows the sleeping-pair count (run_cull)
	d->cull_flags = nh_arena_array<uint32_t>(memory, ccap + 1, &err);
	d->cull_tmp = nh_arena_array<uint32_t>(memory, 2 * NH_SORT_GRID + 64, &err);
	if (err) return err;
	// the per-contact lookup is deferred: bodies on the one-body path fetch their impulses inside the fused solver kernel, and
	// materialize_lookup() fills d->data for everything else (only if such contacts exist, which setup learns)
	d->lookup_pending = true; d->ctx = ctx; d->consumed = false;
	d->tags = contacts->tags; d->features = contacts->features;
	d->ctags = cache->tags; d->cfeatures = cache->features; d->cdata = cache->data;
	d->cull_pending = contacts->sleeping_pairs != nullptr;
	d->sleeping_pairs = contacts->sleeping_pairs;
	d->cache_capacity = ccap;
	if (d->cull_pending && (ctx->flags & NH_FLAG_SYNC_COUNTS)) {
		// reference semantics: nh_collide has just synchronised, the count is on the host
		run_cull(ctx, d, ctx->h_state->sleeping);
	}
	*out = d;
	return NH_OK;
}

// `bodies` / `body_class` given: restrict to the contacts that read their warm start from d->data (see k_cache_lookup)
// (`general_list`, `general`: the contacts that read their warm start from d->data are the listed general ones and nothing else -- no body of the rare many-contact class -- and
// they are few: the lookup walks the list)
static void materialize_lookup(nh_context* ctx, nh_ContactImpulseData* d, const nh_BodyPair* bodies = nullptr, const uint8_t* body_class = nullptr, const uint32_t* general_list = nullptr, uint32_t general = 0u) {
	if (!d->lookup_pending) return;
	d->lookup_pending = false;
	if (general_list && general != 0u && (uint64_t)general * 16u < d->capacity) {
		NH_LAUNCH(ctx, "cache_lookup", k_cache_lookup_listed, nh_grid_for(general, 256, 4096), 256, ctx->d_state, general_list, d->tags, d->features, d->ctags, d->cfeatures, d->cdata, d->data);
		return;
	}
	NH_LAUNCH(ctx, "cache_lookup", k_cache_lookup, nh_grid_for(d->capacity, 256, 16384), 256, ctx->d_state, d->tags, d->features, d->ctags, d->cfeatures, d->cdata, d->data, bodies, body_class);
}

extern "C" const nh_CachedContactImpulse* nh_contact_impulses_device(const nh_ContactImpulseData* d) {
	if (!d) return nullptr;
	nh_ContactImpulseData* m = const_cast<nh_ContactImpulseData*>(d);
	if (m->ctx) {
		// before the solver has run this holds the warm-start impulses, afterwards the solved ones: make either visible
		if (m->ctx->pending) nh_flush_pending(m->ctx);
		else if (!m->consumed) materialize_lookup(m->ctx, m);
		nh_StillStep& ss = m->ctx->still;
		if (ss.active && ss.resolved && m->consumed && m->cdata == ss.cache_data) {
			// a still step: the solved impulses live in the slot cache; in tag order they are what the exported cache holds (nothing is culled in a still step)
			if (nh_still_sync_outputs(m->ctx, NH_VIEW_CACHE) == NH_OK)
				hipMemcpyAsync(m->data, ss.cache.data, sizeof(nh_CachedContactImpulse) * (size_t)(m->capacity < ss.cache.capacity ? m->capacity : ss.cache.capacity), hipMemcpyDeviceToDevice, m->ctx->stream);
		}
	}
	return d->data;
}

__global__ void k_cache_to_slots(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ dense_slot, const nh_CachedContactImpulse* __restrict__ imp,
                                 const uint32_t* __restrict__ features, const nh_Record* __restrict__ rec, float4* __restrict__ sc_imp, uint32_t* __restrict__ sc_feat, uint32_t* __restrict__ sc_count);

__global__ void k_culled_to_slots(const nh_DevState* __restrict__ st, const uint64_t* __restrict__ ctags, const uint32_t* __restrict__ cfeatures,
                                  const nh_CachedContactImpulse* __restrict__ cdata, const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ sorted_idx,
                                  const nh_Record* __restrict__ rec, uint32_t pair_cap, float4* __restrict__ sc_imp, uint32_t* __restrict__ sc_feat, uint32_t* __restrict__ sc_count);

extern "C" int nh_write_cached_impulses(nh_context* ctx, nh_ContactCache* cache, const nh_ContactData* contacts, nh_ContactImpulseData* imp) {
	if (!ctx || !cache || !contacts || !imp) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	{
		nh_StillStep& ss = ctx->still;
		// a still step that went through: the solver has written every contact's impulse into its cache entry, tags and features are last step's -- nothing to do
		if (ss.active && ss.resolved && cache->data == ss.cache_data && imp->ctx == ctx && imp->consumed) return NH_OK;
		// a full step: afterwards the cache IS this step's contact list when nothing was culled (the round trip has told) -- what the next still step relies on
		// (sleepers form: entries kept aside for sleeping pairs are part of such a cache -- they go to the slots of their records below)
		ss.cache_ok = imp->consumed && !imp->cull_pending && !(ctx->flags & NH_FLAG_SYNC_COUNTS) && ((ctx->h_state->sleeping == 0u && ctx->h_state->culled == 0u) || !ss.no_local) &&
		              contacts->data == ss.lay_contacts.data && contacts->tags == ss.lay_contacts.tags;
		ss.cache_tags = cache->tags; ss.cache_features = cache->features; ss.cache_data = cache->data; ss.cache_capacity = cache->capacity;
		ss.cache = *cache;
		ss.slots_current = false;
	}
	if (!imp->consumed) materialize_lookup(ctx, imp);      // no setup ran on this handle: the cache is rewritten from the looked-up impulses
	if (imp->cull_pending) {
		nh_Counts c;
		int rc = nh_read_counts(ctx, &c);
		if (rc) return rc;
		run_cull(ctx, imp, c.sleeping_pairs);
	}
	nh_DevState* st = ctx->d_state;
	NH_LAUNCH(ctx, "write_cache", k_write_cache, nh_grid_for((uint64_t)contacts->capacity + cache->capacity, 256, 4096), 256, st,
	          contacts->tags, contacts->features, imp->data, imp->culled_tags, imp->culled_features, imp->culled_data,
	          cache->tags, cache->features, cache->data, cache->capacity);
	if (ctx->still.ok_next && ctx->still.cache_ok && !ctx->still.disabled && (ctx->flags & NH_FLAG_FUSED_STEP) && ctx->sort_seeded) {
		// the next step may be a still one: the cache goes to the slots as well (solved impulse + feature word to the raw slot each contact came from)
		NH_LAUNCH(ctx, "cache_to_slots", k_cache_to_slots, nh_grid_for(contacts->capacity, 256, 4096), 256, st, ctx->dense_slot, imp->data, contacts->features, ctx->rec,
		          ctx->sc_imp, ctx->sc_feat, ctx->sc_count);
		// (the entries kept aside for sleeping pairs, nudge.cpp:4064-4101: to the slots of the pairs' records -- a step in sleepers form keeps them there)
		// (the host's mirror knows the step's sleeping pairs from the round trip; how many entries were kept aside is counted on the device after it: the kernel reads that)
		if (ctx->h_state->sleeping)
			NH_LAUNCH(ctx, "culled_to_slots", k_culled_to_slots, nh_grid_for(imp->culled_capacity, 256, 2048), 256, st, imp->culled_tags, imp->culled_features, imp->culled_data, ctx->sort_sorted_keys,
			          ctx->sort_sorted_idx, ctx->rec, ctx->lay_capacity, ctx->sc_imp, ctx->sc_feat, ctx->sc_count);
		ctx->still.slots_current = true;
	}
	if (ctx->flags & NH_FLAG_SYNC_COUNTS) {
		nh_Counts c;
		int rc = nh_read_counts(ctx, &c);
		if (rc) return rc;
		cache->count = c.cache;
		if (c.error) return (int)c.error;
	}
	return NH_OK;
}

extern "C" int nh_setup_contact_constraints(nh_context* ctx, const nh_ActiveBodies* active_bodies, const nh_ContactData* contacts,
                                            const nh_BodyData* bodies, nh_ContactImpulseData* imp, nh_Arena* memory, nh_ContactConstraintData** out) {
	if (!ctx || !contacts || !bodies || !imp || !memory || !out) return NH_ERR_INVALID;
	{
		const nh_StillStep& ss = ctx->still;
		const bool in_sequence = ss.active && !ss.resolved && !ctx->pending && contacts->data == ss.lay_contacts.data && contacts->bodies == ss.lay_contacts.bodies &&
		                         bodies->momentum == ss.bodies.momentum && bodies->transforms == ss.bodies.transforms && bodies->count == ss.bodies.count && imp->ctx == ctx && !imp->consumed;
		int rc = nh_flush_pending(ctx, true, in_sequence); if (rc) return rc;
	}
	(void)active_bodies;
	nh_DevState* st = ctx->d_state;
	const uint32_t kcap = contacts->capacity;
	const uint32_t B = bodies->count;
	int err = NH_OK;
	nh_ContactConstraintData* d = new_constraint_data(ctx);
	d->rows = nh_arena_array<float>(memory, (size_t)kcap * 40, &err);
	d->states = nh_arena_array<float>(memory, (size_t)kcap * 4, &err);
	if (!ctx->deg || ctx->deg_capacity < NH_DEG_WORDS(B)) return NH_ERR_INVALID;      // nh_collide of this step sized and filled it
	// one setup per collide: the fill cursors and the contact layout below belong to the last nh_collide (header note 8)
	if (ctx->setup_seq == ctx->collide_seq) return NH_ERR_STALE_SETUP;
	if ((ctx->flags & NH_FLAG_SYNC_COUNTS) && contacts->count != ctx->h_state->contacts) return NH_ERR_STALE_SETUP;
	ctx->setup_seq = ctx->collide_seq;
	d->body_off = nh_arena_array<uint32_t>(memory, (size_t)B + 2u, &err);           // CSR offsets (scan of the degrees nh_collide counted)
	d->adj = nh_arena_array<uint32_t>(memory, (size_t)kcap * 2, &err);
	if (!ctx->lay_class || ctx->lay_body_capacity < B) return NH_ERR_INVALID;        // (sized by this step's nh_collide)
	d->body_class = ctx->lay_class;          // library-owned: a still step reads what the last full step's k_adj_simple left here
	d->simple = ctx->lay_simple;
	d->level_order = nh_arena_array<uint32_t>(memory, kcap, &err);
	d->gpair = nh_arena_array<uint2>(memory, kcap, &err);
	d->gstates = nh_arena_array<float4>(memory, kcap, &err);
	uint32_t* cursor = ctx->deg + NH_DEG_STRIDE(B);                                   // fill cursors (zeroed by nh_collide)
	uint32_t* pred_a = nh_arena_array<uint32_t>(memory, kcap, &err);
	uint32_t* pred_b = nh_arena_array<uint32_t>(memory, kcap, &err);
	uint32_t* level = nh_arena_array<uint32_t>(memory, kcap, &err);
	uint32_t* general_list = nh_arena_array<uint32_t>(memory, kcap, &err);
	uint32_t* slot_key = (ctx->flags & NH_FLAG_EXACT_ORDER) ? nh_arena_array<uint32_t>(memory, kcap, &err) : nullptr;
	uint32_t* tent = (ctx->flags & NH_FLAG_EXACT_ORDER) ? nullptr : nh_arena_array<uint32_t>(memory, kcap, &err);      // colouring: this round's picks
	// exact order: spill space of the scheduler replay's open batches (k_greedy_replay), at most 8192 per bucket = 12.6 MB
	const uint32_t spill_cap = slot_key ? (kcap / GR_BUCKETS + 1u < 8192u ? kcap / GR_BUCKETS + 1u : 8192u) : 0u;
	uint2* spill_ab = slot_key ? nh_arena_array<uint2>(memory, (size_t)GR_BUCKETS * spill_cap * 8u, &err) : nullptr;
	uint32_t* spill_idx = slot_key ? nh_arena_array<uint32_t>(memory, (size_t)GR_BUCKETS * spill_cap * 8u, &err) : nullptr;
	uint32_t* level_hist = nh_arena_array<uint32_t>(memory, 2 * (NH_MAX_LEVELS + 2), &err);       // [0, L+2): histogram -> offsets; [L+2, 2L+4): class has a full row
	uint32_t* level_cursor = nh_arena_array<uint32_t>(memory, NH_MAX_LEVELS + 2, &err);
	uint32_t* tmp = nh_arena_array<uint32_t>(memory, 2 * NH_SORT_GRID + 64, &err);
	if (err) return err;
	d->contact_capacity = kcap; d->body_count = B; d->bodies = contacts->bodies;
	d->contact_data = contacts->data; d->impulses = imp->data; d->general_list = general_list;
	d->levels = 0; d->general_contacts = 0;
	d->blk.active = false; d->blk.warm_pending = false; d->blk.local = false;

	// (degrees were counted by nh_collide while it laid the contacts out; their scan into CSR offsets waits until somebody needs it: ensure_csr)
	d->csr_ready = false;
	if (slot_key) {
		const uint32_t seq = ++ctx->order_seq ? ctx->order_seq : ++ctx->order_seq;       // never 0: tells this call's verdict from an older one
		NH_LAUNCH(ctx, "order_check", k_order_check, nh_grid_for(kcap / 8u + 16u, 256, 1024), 256, st, contacts->bodies, slot_key, seq);
		NH_LAUNCH(ctx, "greedy_replay", k_greedy_replay, 1, 64, st, contacts->bodies, slot_key, seq, spill_ab, spill_idx, spill_cap);
	}
	if (ctx->still.active && !ctx->still.resolved) ctx->still.setup_d = d;      // still step: classes and records are last step's (nh_still_abandon launches the kernel if it comes to that)
	else
	NH_LAUNCH(ctx, "adjacency_simple", k_adj_simple, nh_grid_for(B, 256, 4096), 256, st, B, ctx->deg, reinterpret_cast<const unsigned long long*>(ctx->deg + 2u * NH_DEG_STRIDE(B)), ctx->deg + 4u * NH_DEG_STRIDE(B), slot_key,
	          d->body_class, bodies->properties, bodies->momentum, d->simple, ctx->body_rec, ctx->body_pos, ctx->sort_seeded ? ctx->sort_sorted_idx : (const uint32_t*)nullptr);
	if (ctx->hint_capacity < B) {
		// library-owned, persistent across steps: per body, where its contacts started in the previous step's list (warm-start hint)
		if (ctx->hint) NH_HIP_CHECK(ctx, hipFree(ctx->hint));
		ctx->hint = nullptr; ctx->hint_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->hint, sizeof(uint32_t) * (size_t)B));
		NH_HIP_CHECK(ctx, hipMemsetAsync(ctx->hint, 0xFF, sizeof(uint32_t) * (size_t)B, ctx->stream));
		ctx->hint_capacity = B;
	}
	// No host round trip here.  The one-body path (lookup + rows + warm start) runs fused with the first sweeps, and everything that
	// needs the device counters is finished behind it (finish_setup), by nh_apply_impulses or by the next call that observes
	// momentum / impulses / counters (nh_flush_pending).
	d->cont.contacts = *contacts;
	d->cont.cursor = cursor; d->cont.pred_a = pred_a; d->cont.pred_b = pred_b; d->cont.level = level; d->cont.slot_key = slot_key;
	d->cont.level_hist = level_hist; d->cont.level_cursor = level_cursor; d->cont.tmp = tmp; d->cont.tent = tent;
	d->has_static8 = d->has_staticN = d->has_late = false; d->static_inert = false; d->general_lists = true;
	d->finish_pending = true;
	d->setup_pending = true; d->imp = imp; d->bodies_at_setup = *bodies;
	imp->consumed = true;
	ctx->pending = d;
	*out = d;
	return NH_OK;
}

// The part of setup_contact_constraints that needs the device counters on the host: general adjacency for bodies k_adj_simple left
// pending, culling when something sleeps, the >8-contact one-body class, and the level-scheduled path (colouring / levels, rows,
// warm start).  Runs once per setup, with ctx->pending already cleared.
// CSR adjacency (body_off[] offsets, adj[] lists) -- needed by every kernel but the speculative one-body launch
static void ensure_csr(nh_context* ctx, nh_ContactConstraintData* d) {
	if (d->csr_ready) return;
	d->csr_ready = true;
	const uint32_t B = d->body_count;
	nh_scan_u32(ctx, ctx->deg, d->body_off, &ctx->d_state->pad0, B + 1, d->cont.tmp, nullptr);
	NH_LAUNCH(ctx, "adjacency_from_simple", k_adj_from_simple, nh_grid_for(B, 256, 4096), 256, B, d->body_class, d->body_off, d->simple, d->adj);
}

// ---- spatially blocked solver (nh_blocks.h): host side ---------------------------------------------------------------------------------------------
static int blk_reserve(nh_context* ctx, uint32_t B, uint32_t kcap) {
	nh_BlkBuffers& k = ctx->blk;
	if (!k.tables) {
		const size_t nb = NH_BLK_MAX_BLOCKS + 8u, nbins = (size_t)(NH_BLK_MAX_BLOCKS + 1u) * NH_BLK_MAX_COLOURS + 8u;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.blk_count, sizeof(uint32_t) * nb));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.own_base, sizeof(uint32_t) * nb));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.bins, sizeof(uint32_t) * nbins));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.bin_off, sizeof(uint32_t) * nbins));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.ghost_base, sizeof(uint32_t) * nb));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.ghost_count, sizeof(uint32_t) * nb));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.phase_list, sizeof(uint32_t) * 8u * NH_BLK_MAX_BLOCKS));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.desc, sizeof(uint32_t) * 8u * NH_BLK_MAX_BLOCKS * NH_BLK_DESC_WORDS));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.scan_tmp, sizeof(uint32_t) * (2u * NH_SORT_GRID + 64u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.chain_slot, sizeof(uint32_t) * nb));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.chain_nbr, sizeof(uint32_t) * nb * NH_BLK_NBR_WORDS));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.chain_done, sizeof(uint32_t) * nb));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.chain_ticket, sizeof(uint32_t) * NH_BLK_CHAIN_PASSES));
		k.tables = true;
	}
	if (k.body_capacity < B + 1u) {
		if (k.brank) hipFree(k.brank);
		if (k.bm) hipFree(k.bm);
		if (k.pos_body) hipFree(k.pos_body);
		if (k.touch) hipFree(k.touch);
		k.brank = nullptr; k.bm = nullptr; k.pos_body = nullptr; k.touch = nullptr; k.body_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.brank, sizeof(uint2) * (size_t)(B + 1u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.bm, sizeof(float4) * 2u * (size_t)(B + 1u)));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.pos_body, sizeof(uint32_t) * (size_t)(B + 1u)));
		if (k.check) NH_HIP_CHECK(ctx, hipMalloc((void**)&k.touch, sizeof(uint32_t) * 2u * (size_t)(B + 1u)));
		k.body_capacity = B + 1u;
	}
	if (k.contact_capacity < kcap) {
		if (k.lpair) hipFree(k.lpair);
		if (k.ghost_list) hipFree(k.ghost_list);
		k.lpair = nullptr; k.ghost_list = nullptr; k.contact_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.lpair, sizeof(uint32_t) * (size_t)kcap));
		NH_HIP_CHECK(ctx, hipMalloc((void**)&k.ghost_list, sizeof(uint32_t) * (size_t)kcap));
		k.contact_capacity = kcap;
	}
	return NH_OK;
}

// Lays the general contacts out by (owning block, colour) and builds the per-block tables.  Leaves d->blk.active false when the blocked form does not
// apply (the caller then takes the per-colour global launches): too few contacts, too many colours, a block that cannot be made to fit.
static int blk_setup(nh_context* ctx, nh_ContactConstraintData* d, uint32_t G, bool local) {
	nh_BlkBuffers& k = ctx->blk;
	nh_DevState* st = ctx->d_state;
	// `local`: tried BEFORE any world-wide colouring exists -- contacts are binned by owning block only and every block colours its own (k_blk_prepare_local);
	// gives up (not active) when some contact is residual or a block needs more colours than its descriptor holds: the caller colours world-wide and comes again
	d->blk.active = false; d->blk.local = false;
	const uint32_t B = d->body_count, kcap = d->contact_capacity, L = local ? 1u : d->levels;
	if (k.disabled || (ctx->flags & NH_FLAG_EXACT_ORDER) || G < k.min_contacts || L < 1u || L > NH_BLK_MAX_COLOURS || !d->static_inert) return NH_OK;
	{ int rc = blk_reserve(ctx, B, kcap); if (rc) return rc; }
	const nh_ContactData* contacts = &d->cont.contacts;
	const nh_BodyData* bodies = &d->bodies_at_setup;
	uint32_t* rank = d->cont.pred_a; uint32_t* bin = d->cont.pred_b;          // (exact-order scratch: free in the default order once the colouring is done)
	const size_t nbins = (size_t)(NH_BLK_MAX_BLOCKS + 1u) * L + 1u;
	nh_BlkGrid* h = &ctx->h_state->blk;
	bool ok = false;
	for (int attempt = 0; attempt < 6 && !ok; ++attempt) {
		NH_HIP_CHECK(ctx, hipMemsetAsync(k.blk_count, 0, sizeof(uint32_t) * (NH_BLK_MAX_BLOCKS + 8u), ctx->stream));
		NH_HIP_CHECK(ctx, hipMemsetAsync(k.bins, 0, sizeof(uint32_t) * (nbins + 7u), ctx->stream));
		NH_LAUNCH(ctx, "blk_reset", k_blk_reset, 1, 1, st);
		NH_LAUNCH(ctx, "blk_bounds", k_blk_bounds, nh_grid_for(B, 256, 1024), 256, st, B, d->body_class, bodies->transforms);
		NH_LAUNCH(ctx, "blk_grid", k_blk_grid, 1, 1, st, k.scale, (float)k.target);
		NH_LAUNCH(ctx, "blk_body", k_blk_body, nh_grid_for(B, 256, 4096), 256, st, B, d->body_class, bodies->transforms, k.blk_count, k.brank);
		NH_LAUNCH(ctx, "blk_contact", k_blk_contact, nh_grid_for(G, 256, 8192), 256, st, d->general_list, contacts->bodies, local ? (const uint32_t*)nullptr : (const uint32_t*)d->cont.level, k.brank, L, k.bins, rank, bin);
		nh_scan_u32(ctx, k.blk_count, k.own_base, &st->pad0, NH_BLK_MAX_BLOCKS + 1u, k.scan_tmp, nullptr);
		nh_scan_u32(ctx, k.bins, k.bin_off, &st->pad0, (uint32_t)nbins, k.scan_tmp, nullptr);
		NH_LAUNCH(ctx, "blk_scatter", k_blk_scatter, nh_grid_for(G, 256, 8192), 256, st, d->general_list, rank, bin, k.bin_off, d->level_order);
		if (local)         // (the colour-sorted order goes to `rank`'s memory, free since the scatter: it becomes the setup's level order below)
			NH_LAUNCH(ctx, "blk_prepare", k_blk_prepare_local, NH_BLK_MAX_BLOCKS, NH_BLK_PREP_THREADS, st, k.bin_off, k.blk_count, k.own_base, d->level_order, contacts->bodies, k.brank,
			          (const nh_CachedContactImpulse*)d->imp->data, d->cont.tent /* (not `level`: a failed attempt must leave that as k_contact_class wrote it) */, rank, k.lpair, k.ghost_base, k.ghost_count, k.ghost_list, k.phase_list, k.desc);
		else
		NH_LAUNCH(ctx, "blk_prepare", k_blk_prepare, NH_BLK_MAX_BLOCKS, 256, st, L, k.bin_off, k.blk_count, k.own_base, d->level_order, contacts->bodies, k.brank, k.lpair,
		          k.ghost_base, k.ghost_count, k.ghost_list, k.phase_list, k.desc);
		NH_HIP_CHECK(ctx, hipMemcpyAsync(h, &st->blk, sizeof(nh_BlkGrid), hipMemcpyDeviceToHost, ctx->stream));
		NH_HIP_CHECK(ctx, hipMemcpyAsync(d->blk.res_off, k.bin_off + (size_t)NH_BLK_MAX_BLOCKS * L, sizeof(uint32_t) * (L + 1u), hipMemcpyDeviceToHost, ctx->stream));
		NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
		if (!h->nblk) return NH_OK;
		if (local && (h->far_contacts || h->too_many_colours)) {
			if (k.prof) fprintf(stderr, "blk_profile setup: per-block colouring gave up (residual %u, too many colours %u, most rounds %u, most anew %u, overflow %u)\n", h->far_contacts, h->too_many_colours, h->max_rounds, h->max_listed, h->overflow);
			return NH_OK;
		}         // (a contact no block can own / a block of more than 32 colours: the world-wide colouring it is)
		if (h->overflow) { k.scale *= 0.8f; continue; }
		ok = true;
		// the edge follows the largest load: room for the pile to shift, no retries
		const float load = (float)h->max_load / (float)NH_BLK_CAP;
		if (load > 0.90f) k.scale *= 0.97f; else if (load < 0.60f && k.scale < 4.0f) k.scale *= 1.03f;
	}
	if (!ok) return NH_OK;
	for (int p = 0; p < 8; ++p) d->blk.phase_count[p] = h->phase_count[p];
	if (k.prof && (k.prof_calls % 64u) == 63u) fprintf(stderr, "blk_profile setup: %s colouring, most colours in a block %u (most rounds %u, most contacts coloured anew %u), largest load %u of %u, near contacts %u, residual %u\n", local ? "per-block" : "world-wide", local ? h->max_colours : L, h->max_rounds, h->max_listed, h->max_load, NH_BLK_CAP, h->near_contacts, h->far_contacts);
	d->blk.L = local ? NH_BLK_LOCAL_L : L;
	d->blk.local = local;
	if (local) {
		// every contact is owned by a block; the order k_blk_prepare_local wrote is the setup's level order from here on
		for (uint32_t l = 1; l <= NH_BLK_LOCAL_L; ++l) d->blk.res_off[l] = d->blk.res_off[0];
		uint32_t* t = d->level_order; d->level_order = rank; d->cont.pred_a = t;
	}
	// rows (and the cached impulses, in the solver states' place) in block order; which colours hold a full row
	uint32_t* level_hist = d->cont.level_hist;
	NH_LAUNCH(ctx, "zero_u32", k_zero_u32, 8, 256, level_hist, 2 * (NH_MAX_LEVELS + 2));
	if (local && !ctx->env_blk_rows_global) {
		// one workgroup per block, the block's bodies staged in LDS (k_blk_rows): every general contact is owned by a block here
		nh_BlkPhases ph;
		ph.start[0] = 0;
		for (int p = 0; p < 8; ++p) ph.start[p + 1] = ph.start[p] + d->blk.phase_count[p];
		NH_LAUNCH(ctx, "blk_slots", k_blk_slots, nh_grid_for(B, 256, 4096), 256, B, k.brank, k.own_base, k.pos_body);
		if (ph.start[8])
			NH_LAUNCH(ctx, "rows_general", k_blk_rows, ph.start[8], NH_BLK_THREADS, st, ph, k.desc, k.ghost_list, k.pos_body, d->level_order, contacts->data, bodies->transforms, bodies->properties,
			          (float4*)d->rows, (size_t)kcap, d->gpair, level_hist + (NH_MAX_LEVELS + 2), (const nh_CachedContactImpulse*)d->imp->data, reinterpret_cast<float*>(d->gstates), k.lpair);
	} else
	NH_LAUNCH(ctx, "rows_general", k_rows_general, nh_grid_for(G, 256, 4096), 256, st, d->level_order, contacts->data, contacts->bodies, bodies->transforms, bodies->properties,
	          (float4*)d->rows, (size_t)kcap, d->gpair, 1u, 1u, local ? (const uint32_t*)nullptr : (const uint32_t*)d->cont.level, level_hist + (NH_MAX_LEVELS + 2), (const nh_CachedContactImpulse*)d->imp->data, d->gstates, k.lpair, d->blk.res_off[0]);
	if (d->blk.res_off[L] > d->blk.res_off[0])
		NH_LAUNCH(ctx, "blk_residual_pairs", k_blk_residual_pairs, nh_grid_for(d->blk.res_off[L] - d->blk.res_off[0], 256, 4096), 256, d->blk.res_off[0], d->blk.res_off[L], k.brank, k.own_base, d->gpair);
	NH_HIP_CHECK(ctx, hipMemcpyAsync(d->level_full, level_hist + (NH_MAX_LEVELS + 2), sizeof(uint32_t) * (L + 2u), hipMemcpyDeviceToHost, ctx->stream));
	NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
	d->blk.radial_only = true;
	for (uint32_t l = 1; l <= L; ++l) d->blk.radial_only &= d->level_full[l] == 0u;
	if (local) for (uint32_t l = 2; l <= NH_BLK_LOCAL_L + 1u; ++l) d->level_full[l] = d->level_full[1];       // (one flag for all colours)
	if (k.check) {
		// NH_BLK_CHECK=1 (tests): the structure the sweeps rely on, launch by launch
		NH_LAUNCH(ctx, "blk_gather", k_blk_gather, nh_grid_for(B, 256, 4096), 256, B, k.brank, k.own_base, (const float4*)bodies->momentum, bodies->properties, k.bm, k.pos_body);
		for (int p = 0; p < 8; ++p) {
			if (!d->blk.phase_count[p]) continue;
			NH_HIP_CHECK(ctx, hipMemsetAsync(k.touch, 0, sizeof(uint32_t) * 2u * (size_t)(B + 1u), ctx->stream));
			NH_LAUNCH(ctx, "blk_check", k_blk_check, d->blk.phase_count[p], 256, st, k.phase_list + (size_t)p * NH_BLK_MAX_BLOCKS, d->blk.L, k.bin_off, k.blk_count, k.own_base, k.ghost_base,
			          k.ghost_count, k.ghost_list, k.lpair, d->level_order, contacts->bodies, local ? (const uint32_t*)nullptr : (const uint32_t*)d->cont.level, k.pos_body, k.touch, k.touch + (B + 1u),
			          local ? (const uint32_t*)(k.desc + (size_t)p * NH_BLK_MAX_BLOCKS * NH_BLK_DESC_WORDS) : (const uint32_t*)nullptr);
		}
		NH_HIP_CHECK(ctx, hipMemcpyAsync(h, &st->blk, sizeof(nh_BlkGrid), hipMemcpyDeviceToHost, ctx->stream));
		NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
		if (h->check_errors) { fprintf(stderr, "nudge_hip: blocked solver structure check failed (%u errors)\n", h->check_errors); return NH_ERR_INVALID; }
	}
	// chained sweeps (option "blk_chain"): which task is a block's, which tasks of earlier colours it waits for; tags and tickets start over
	d->blk.chain = false; d->blk.chain_pass = 0;
	if (k.chain && local && d->blk.res_off[d->blk.L] == d->blk.res_off[0]) {
		nh_BlkPhases ph;
		ph.start[0] = 0;
		for (int p = 0; p < 8; ++p) ph.start[p + 1] = ph.start[p] + d->blk.phase_count[p];
		if (ph.start[8]) {
			NH_HIP_CHECK(ctx, hipMemsetAsync(k.chain_slot, 0xff, sizeof(uint32_t) * NH_BLK_MAX_BLOCKS, ctx->stream));
			NH_HIP_CHECK(ctx, hipMemsetAsync(k.chain_ticket, 0, sizeof(uint32_t) * NH_BLK_CHAIN_PASSES, ctx->stream));
			NH_LAUNCH(ctx, "blk_chain_slots", k_blk_chain_slots, (ph.start[8] + 255u) / 256u, 256, ph, k.phase_list, k.chain_slot);
			NH_LAUNCH(ctx, "blk_chain_nbr", k_blk_chain_nbr, (ph.start[8] + 255u) / 256u, 256, st, ph, k.phase_list, k.chain_slot, k.chain_nbr, k.chain_done);
			d->blk.chain = true;
		}
	}
	d->blk.active = true;
	d->blk.warm_pending = true;
	return NH_OK;
}

// warm start (once per setup) and `iterations` sweeps of the blocked form: momentum into block order, per pass eight block-colour launches + the
// residual colours, momentum back
static void blk_run(nh_context* ctx, nh_ContactConstraintData* d, const nh_BodyData* bodies, bool warm, uint32_t iterations) {
	nh_BlkBuffers& k = ctx->blk;
	const uint32_t B = d->body_count, L = d->blk.L;
	const uint32_t passes = (warm ? 1u : 0u) + iterations;
	if (!passes) return;
	const size_t stride = (size_t)d->contact_capacity;
	NH_LAUNCH(ctx, "blk_gather", k_blk_gather, nh_grid_for(B, 256, 4096), 256, B, k.brank, k.own_base, (const float4*)bodies->momentum, bodies->properties, k.bm, (uint32_t*)nullptr);
	for (uint32_t pass = 0; pass < passes; ++pass) {
		const bool w = warm && pass == 0u;
		if (d->blk.chain && d->blk.chain_pass < NH_BLK_CHAIN_PASSES) {
			// one launch for the eight block colours of this pass
			nh_BlkChain ch;
			ch.ph.start[0] = 0;
			for (int p = 0; p < 8; ++p) ch.ph.start[p + 1] = ch.ph.start[p] + d->blk.phase_count[p];
			ch.nbr = k.chain_nbr; ch.done = k.chain_done; ch.ticket = k.chain_ticket + d->blk.chain_pass; ch.tag = ++d->blk.chain_pass; ch.error = &ctx->d_state->error;
			unsigned long long* pr = (k.prof && pass + 1u == passes) ? k.prof : (unsigned long long*)nullptr;
#define NH_BLK_CHAINED(NAME, W, R) NH_LAUNCH(ctx, NAME, (k_blk_sweep<W, R, true>), ch.ph.start[8], NH_BLK_THREADS, k.desc, L, k.ghost_list, k.lpair, \
			          (const float4*)d->rows, stride, reinterpret_cast<float*>(d->gstates), k.bm, pr, ch)
			if (w) { if (d->blk.radial_only) NH_BLK_CHAINED("blk_warm", true, true); else NH_BLK_CHAINED("blk_warm", true, false); }
			else { if (d->blk.radial_only) NH_BLK_CHAINED("blk_sweep", false, true); else NH_BLK_CHAINED("blk_sweep", false, false); }
#undef NH_BLK_CHAINED
			continue;
		}
		for (int p = 0; p < 8; ++p) {
			const uint32_t n = d->blk.phase_count[p];
			if (!n) continue;
			const uint32_t* list = k.desc + (size_t)p * NH_BLK_MAX_BLOCKS * NH_BLK_DESC_WORDS;
#define NH_BLK_SWEEP(NAME, W, R) NH_LAUNCH(ctx, NAME, (k_blk_sweep<W, R>), n, NH_BLK_THREADS, list, L, k.ghost_list, k.lpair, \
			          (const float4*)d->rows, stride, reinterpret_cast<float*>(d->gstates), k.bm, (k.prof && pass + 1u == passes) ? k.prof + 4u * (size_t)p * NH_BLK_MAX_BLOCKS : (unsigned long long*)nullptr)
			if (w) { if (d->blk.radial_only) NH_BLK_SWEEP("blk_warm", true, true); else NH_BLK_SWEEP("blk_warm", true, false); }
			else { if (d->blk.radial_only) NH_BLK_SWEEP("blk_sweep", false, true); else NH_BLK_SWEEP("blk_sweep", false, false); }
#undef NH_BLK_SWEEP
		}
		for (uint32_t l = 0; l < L; ++l) {
			const uint32_t b = d->blk.res_off[l], e = d->blk.res_off[l + 1];
			if (e <= b) continue;
			nh_BodyMomentum* bm = reinterpret_cast<nh_BodyMomentum*>(k.bm);
			const bool full = d->level_full[l + 1u] != 0u;
#define NH_BLK_RES(NAME, W, R) NH_LAUNCH(ctx, NAME, (k_level_exec<W, R>), nh_grid_for(e - b, 256, 8192), 256, d->level_order, b, e, d->gpair, (const nh_CachedContactImpulse*)nullptr, bm, \
			          (const float4*)d->rows, stride, d->gstates, 1u)
			if (w) { if (full) NH_BLK_RES("warm_level", true, false); else NH_BLK_RES("warm_level", true, true); }
			else { if (full) NH_BLK_RES("apply_level", false, false); else NH_BLK_RES("apply_level", false, true); }
#undef NH_BLK_RES
		}
	}
	if (k.prof && (++k.prof_calls % 64u) == 0u) {
		// NH_BLK_PROFILE=1: the last sweep's eight launches, per workgroup: prologue / colours / epilogue in microseconds (100 MHz clock)
		hipStreamSynchronize(ctx->stream);
		std::vector<unsigned long long> h(4u * NH_BLK_MAX_BLOCKS);
		unsigned long long prev_end = 0;
		for (int p = 0; p < 8; ++p) {
			const uint32_t n = d->blk.phase_count[p];
			if (!n) continue;
			hipMemcpy(h.data(), k.prof + 4u * (size_t)p * NH_BLK_MAX_BLOCKS, sizeof(unsigned long long) * 4u * n, hipMemcpyDeviceToHost);
			double s0 = 0, s1 = 0, s2 = 0, m0 = 0, m1 = 0, m2 = 0;
			unsigned long long first = ~0ull, lastt = 0;
			for (uint32_t i = 0; i < n; ++i) {
				const double a = 0.01 * (double)(h[4 * i + 1] - h[4 * i]), b = 0.01 * (double)(h[4 * i + 2] - h[4 * i + 1]), c = 0.01 * (double)(h[4 * i + 3] - h[4 * i + 2]);
				s0 += a; s1 += b; s2 += c; if (a > m0) m0 = a; if (b > m1) m1 = b; if (c > m2) m2 = c;
				if (h[4 * i] < first) first = h[4 * i];
				if (h[4 * i + 3] > lastt) lastt = h[4 * i + 3];
			}
			fprintf(stderr, "blk_profile phase %d: %u workgroups, first start to last end %.1f us (%.1f us after the previous launch's last end); mean (max) us: prologue %.2f (%.2f), colours %.2f (%.2f), epilogue %.2f (%.2f)\n",
			        p, n, 0.01 * (double)(lastt - first), prev_end ? 0.01 * (double)((long long)first - (long long)prev_end) : 0.0, s0 / n, m0, s1 / n, m1, s2 / n, m2);
			prev_end = lastt;
		}
	}
	NH_LAUNCH(ctx, "blk_scatter_back", k_blk_scatter_back, nh_grid_for(B, 256, 4096), 256, B, k.brank, k.own_base, (const float4*)k.bm, (float4*)bodies->momentum);
}

// LOCAL speculation (nh_internal.h): every round trip -- a full step's, a still step's verdict -- tells whether somebody left its inflated box in that step; the
// movers form of the still step stays on for 16 steps after the last one who did
static void still_note_movers(nh_context* ctx, const nh_DevState* h, uint32_t seq = 0u) {          // `seq`: the nh_collide the counters belong to (0: a full step's round trip)
	nh_StillStep& ss = ctx->still;
	// (a still step in sleepers form that found NOBODY awake: the next step is a full one -- two of those in a row start the asleep steps, which cost nothing)
	if (ss.sleepers && h->active == 0u) ss.ok_next = false;
	// (sleepers ahead: how long has the sleeping set stood still?  Counted over confirmed still steps -- `seq` != 0 -- by the active count they report)
	if (seq != 0u) { if (h->active == ss.sleep_last_active) { if (ss.sleep_stable < 0xffffu) ss.sleep_stable++; } else { ss.sleep_stable = 0u; ss.sleep_last_active = h->active; } }
#ifdef NH_DEBUG_PAIR
	fprintf(stderr, "[pair] verdict: map_ok %d unowned %u records %u pairs %u+%u world_bad %d ok %d ready %d\n", (int)ss.ahead_map_ok, h->pair_unowned, h->records, h->pairs, h->pairs_sph, (int)ss.pair_world_bad, (int)ss.pair_world_ok, (int)ss.pair_ready);
#endif
	// (... counted by k_pair_owned in the nh_collide numbered pair_owned_seq: counters of an earlier step, or of a full step -- which voids the count -- say nothing)
	if (ss.pair_owned_seq != 0u && seq >= ss.pair_owned_seq) { if (h->pair_unowned > ctx->pair_list_capacity) ss.pair_world_bad = true; else ss.pair_world_ok = true; }         // (pair ahead: some kept pair is nobody's -- k_pair_owned; the step that relied on it has failed itself)
	if (h->ahead_multi) ss.ahead_world_bad = true;          // (xform ahead: some body carries several colliders -- k_ahead_check; the step that relied on the map has failed itself)
	if (h->fat_inserts != ss.seen_inserts) ss.movers_left = 16u;
	else if (ss.movers_left) ss.movers_left--;
	ss.seen_inserts = h->fat_inserts; ss.seen_rebuilds = h->fat_rebuilds;
}

// EARLY COUNTERS (nh_internal.h): the counters the launch numbered `seq` left in the pinned block as it started.  The host polls the word behind them; should the stream run dry
// without it (a launch that never happened) the ordinary copy answers instead -- nothing can hang here that could not hang in hipStreamSynchronize
static int read_counts_early(nh_context* ctx, nh_Counts* out, uint32_t seq) {
	volatile uint32_t* const flag = ctx->h_early + NH_COUNTER_WORDS;
	bool seen = false;
	for (uint32_t spins = 0; ; ++spins) {
		if (*flag == seq) { seen = true; break; }
		if ((spins & 255u) == 255u) {
			const hipError_t q = hipStreamQuery(ctx->stream);
			if (q == hipSuccess) { seen = *flag == seq; break; }          // (everything enqueued has run: the word is there, or the launch was not)
			if (q != hipErrorNotReady) { ctx->last_hip_error = (int)q; return NH_ERR_HIP; }
		}
		__builtin_ia32_pause();
	}
	if (!seen) { ctx->early_fallbacks++; return nh_read_counts(ctx, out); }
	ctx->gravity_may_overlap = false; ctx->after_collide = false;          // (as nh_read_counts leaves them)
	std::atomic_thread_fence(std::memory_order_acquire);
	memcpy(ctx->h_state, ctx->h_early, NH_COUNTER_WORDS * sizeof(uint32_t));
	ctx->early_reads++;
	nh_counts_from_mirror(ctx, out);
	return NH_OK;
}

static int finish_setup(nh_context* ctx, nh_ContactConstraintData* d) {
	nh_DevState* st = ctx->d_state;
	const nh_ContactData* contacts = &d->cont.contacts;
	const nh_BodyData* bodies = &d->bodies_at_setup;
	nh_ContactImpulseData* imp = d->imp;
	const uint32_t kcap = d->contact_capacity, B = d->body_count;
	uint32_t* pred_a = d->cont.pred_a; uint32_t* pred_b = d->cont.pred_b; uint32_t* level = d->cont.level;
	uint32_t* slot_key = d->cont.slot_key; uint32_t* level_hist = d->cont.level_hist; uint32_t* level_cursor = d->cont.level_cursor;
	uint32_t* general_list = d->general_list;
	// one round trip: body classes, whether anything sleeps, whether the general adjacency build has work
	// (early counters, nh_internal.h: the speculative one-body launch before this call has them on their way to the host as it STARTS -- the solver runs on while the host reads)
	nh_Counts c;
	int rc = d->early_seq ? read_counts_early(ctx, &c, d->early_seq) : nh_read_counts(ctx, &c);
	d->early_seq = 0u;
	if (rc) return rc;
	if (c.error) return (int)c.error;
	d->has_late = ctx->h_state->has_pending != 0;
	{
		// may the NEXT step be launched as a still step?  (nh_internal.h: nh_StillStep)  This step laid its contacts out in a layout a still step can write through, every
		// dynamic body sits in one pair with an inert static world (the fused solver owns them all), nothing sleeps, nothing was culled
		const nh_DevState* h = ctx->h_state;
		// (SLEEPERS form, nh_internal.h: sleeping pairs and kept-aside cache entries no longer rule the next step out -- as long as somebody is awake; a world in which
		// nobody is belongs to the asleep steps below, which need full steps to get going)
		const bool sleepers_ok = !ctx->still.no_local;
		const bool nobody_sleeps = !h->any_sleeping && h->sleeping == 0u && h->culled == 0u;
		ctx->still.ok_next = h->lay_valid && !h->has_unstable && h->static_inert && !h->has_pending && !h->has_static8 && !h->has_staticN && (nobody_sleeps || (sleepers_ok && h->active != 0u)) &&
		                     h->general_contacts == 0u && !h->error;
		// ... and nobody moved: a collider that left its inflated box in THIS step (re-inserted, or the kept list rebuilt) will most likely leave it again in the next
		// one -- a body in the air does on every step -- and a still step that fails costs two steps' worth of launches
		// (LOCAL speculation, nh_internal.h: a still step in movers form re-inserts its own leavers -- then only a REBUILD of the kept list, which also voids the
		// layout, rules the next step out; a re-insertion switches the movers form on for the steps that follow)
		const bool local = !ctx->still.no_local && !ctx->env_no_incremental && !ctx->env_no_fat;
		if ((h->fat_inserts != ctx->still.seen_inserts && !local) || h->fat_rebuilds != ctx->still.seen_rebuilds) ctx->still.ok_next = false;
		still_note_movers(ctx, h);
		// asleep steps (nh_internal.h: nh_AsleepState): was this step the fixed point of a world in which every body is asleep -- nobody active, no contact, every cache
		// entry kept aside -- and the second one in a row with the same counts?  Then nh_step may take the steps that follow as done (after its own check)
		nh_AsleepState& as = ctx->asleep;
		const uint32_t pairs_now = h->pairs + h->pairs_sph;
		// (the cache: entries of sleeping pairs are kept aside and written back, nudge.cpp:4064-4101, by kernels that run after this round trip -- what shows here is that
		//  the count the LAST step left is the count it found, i.e. nothing but kept-aside entries was in it)
		const bool fixed_point = h->active == 0u && h->contacts == 0u && h->general_contacts == 0u && !h->error && h->sleeping != 0u && !ctx->first_ghost;
		if (fixed_point && as.streak && as.prev_pairs == pairs_now && as.prev_sleeping == h->sleeping && as.prev_cache == h->cache) as.streak++;
		else as.streak = fixed_point ? 1u : 0u;
		as.prev_pairs = pairs_now; as.prev_sleeping = h->sleeping; as.prev_cache = h->cache;
		if (as.streak == 2u && !as.disabled) { as.B = d->body_count; if (nh_asleep_remember(ctx)) as.streak = 0; }
	}
	// NH_FLAG_FUSED_STEP: gravity / damping of the bodies the fused solver does not own, before anything else reads their momentum -- not launched
	// at all when every body is the fused solver's (a landed drop scene)
	ctx->adv.rest = ctx->h_state->has_other != 0 || !ctx->h_state->static_inert;
	if (ctx->grav.rest_pending) {
		ctx->grav.rest_pending = false;
		if (ctx->adv.rest)
			NH_LAUNCH(ctx, "gravity_rest", k_gravity_rest, nh_grid_for(ctx->grav.body_count, 256, 2048), 256, st, ctx->grav.active, d->body_class, ctx->grav.momentum,
			          ctx->grav.gx_dt, ctx->grav.gy_dt, ctx->grav.gz_dt, ctx->grav.damping);
	}
	if (d->has_late || ctx->h_state->has_static8 || !ctx->h_state->static_inert) ensure_csr(ctx, d);
	if (d->has_late) {
		// bodies in several pairs or with dynamic partners (piles, pits): general CSR fill + sort + classification, then the counts again
		// (a world that had enough general contacts for the blocked solver last step will most likely colour them block by block again: the general bodies'
		// lists -- which only the world-wide colouring reads -- are then left out, and filled in a second pass if it comes to that)
		const bool local_candidate = !(ctx->flags & NH_FLAG_EXACT_ORDER) && !ctx->blk.disabled && !ctx->env_blk_global_colours && ctx->last_general_contacts >= ctx->blk.min_contacts;
		d->general_lists = !local_candidate;
		NH_LAUNCH(ctx, "adjacency_fill", k_adj_fill, nh_grid_for(kcap, 256, 4096), 256, st, contacts->bodies, contacts->tags, d->body_off, d->cont.cursor, d->adj, d->body_class, local_candidate ? 1u : 3u);
		NH_LAUNCH(ctx, "adjacency_sort", k_adj_sort, nh_grid_for(B, 256, 4096), 256, st, B, contacts->bodies, d->body_off, d->adj, slot_key, d->body_class, pred_a, pred_b, bodies->properties, bodies->momentum);
		NH_LAUNCH(ctx, "contact_class", k_contact_class, nh_grid_for(kcap, 256, 2048), 256, st, contacts->bodies, d->body_class, level, general_list, pred_a, pred_b);
		rc = nh_read_counts(ctx, &c);
		if (rc) return rc;
		if (c.error) return (int)c.error;
	}
	run_cull(ctx, imp, c.sleeping_pairs);
	d->general_contacts = c.general_contacts;
	ctx->last_general_contacts = c.general_contacts;
	d->has_static8 = ctx->h_state->has_static8 != 0;
	d->has_staticN = ctx->h_state->has_staticN != 0;
	d->static_inert = ctx->h_state->static_inert != 0;
	if (d->has_staticN) {
		// bodies with more than 8 static contacts are rare: the warm start reads the materialised lookup
		materialize_lookup(ctx, imp, contacts->bodies, d->body_class);
		NH_LAUNCH(ctx, "setup_staticN", (k_setup_static<true>), nh_grid_for(B, 256, 8192), 256, B, d->body_class, d->body_off, d->adj,
		          contacts->data, contacts->bodies, imp->data, bodies->transforms, bodies->properties, bodies->momentum, d->rows, (float4*)d->states, (const uint32_t*)nullptr);
	}
	if (c.general_contacts) {
		const uint32_t G = c.general_contacts;
		materialize_lookup(ctx, imp, contacts->bodies, d->body_class, d->has_staticN ? (const uint32_t*)nullptr : general_list, G);
		// a large set in default order with no contact that no block can own: the blocks colour their own contacts (nh_blocks.h, k_blk_prepare_local) --
		// no world-wide colouring, no adjacency lists of the general bodies
		if (!(ctx->flags & NH_FLAG_EXACT_ORDER) && !ctx->env_blk_global_colours) {
			rc = blk_setup(ctx, d, G, true);
			if (rc) return rc;
			if (d->blk.active) return NH_OK;
		}
		if (!d->general_lists && d->has_late) {
			// (the per-block colouring did not apply after all: the lists of the general bodies the adjacency build left out)
			NH_LAUNCH(ctx, "adjacency_fill", k_adj_fill, nh_grid_for(kcap, 256, 4096), 256, st, contacts->bodies, contacts->tags, d->body_off, d->cont.cursor, d->adj, d->body_class, 2u);
			d->general_lists = true;
		}
		// A few thousand contacts in default order: colouring and level order by ONE workgroup, no host round trip until the solver's (k_colour_small)
		if (!(ctx->flags & NH_FLAG_EXACT_ORDER) && !ctx->env_colour_jp && !ctx->env_no_resident && !ctx->env_colour_check_seeds && G <= NH_COLOUR_SMALL_MAX &&
		    (G <= NH_RES_MAX_FULL || B <= NH_RESB_MAX_BODIES)) {
			NH_LAUNCH(ctx, "colour_small", k_colour_small, 1, 1024, st, general_list, imp->data, contacts->bodies, d->body_off, d->adj, level, d->cont.tent, level_hist, d->level_order, B);
			NH_LAUNCH(ctx, "rows_general", k_rows_general, nh_grid_for(G, 256, 4096), 256, st, d->level_order, contacts->data, contacts->bodies, bodies->transforms, bodies->properties,
			          (float4*)d->rows, (size_t)kcap, d->gpair, 1u, d->static_inert ? 1u : 0u, level, level_hist + (NH_MAX_LEVELS + 2),
			          (const nh_CachedContactImpulse*)nullptr, (float4*)nullptr, (uint32_t*)nullptr, 0u);
			d->levels = 0;                                  // on the device (st->levels)
			d->resident = G <= NH_RES_MAX_FULL; d->resident_cap = NH_RES_MAX_FULL; d->resident_bodies = !d->resident;
			if (d->resident)
				NH_LAUNCH(ctx, "solve_resident", k_solve_resident, 1, NH_RES_THREADS, st, level_hist, 0u, d->level_order, d->gpair, imp->data,
				          bodies->momentum, (const float4*)d->rows, (size_t)kcap, d->gstates, 0u, 1u, d->resident_cap);
			else
				NH_LAUNCH(ctx, "solve_resident_bodies", k_solve_resident_bodies, 1, NH_RESB_THREADS, st, level_hist, level_hist + (NH_MAX_LEVELS + 2), 0u, d->level_order, d->gpair,
				          imp->data, bodies->momentum, B, (const float4*)d->rows, (size_t)kcap, d->gstates, 0u, 1u);
			return NH_OK;
		}
		// relaxation: every round finalises at least the next level
		uint32_t rounds = 0;
		NH_LAUNCH(ctx, "level_reset", k_level_reset_progress, 1, 1, st);
		// colouring: most contacts are settled by the first rounds; the later ones walk a compacted list of the rest (pred_a / pred_b, which
		// only the exact mode uses, serve as its two buffers)
		const bool colour_jp = ctx->env_colour_jp;
		const uint32_t* colour_list = general_list;
		uint32_t colour_n = 0;                         // 0: the whole general list
		uint32_t* spare[2] = { pred_a, pred_b };
		int spare_at = 0;
		for (;;) {
			// (short batches first while colouring: the list shrinks fastest in the first rounds, and a batch ends with its compaction)
			const int batch = (ctx->flags & NH_FLAG_EXACT_ORDER) ? 8 : (colour_jp ? (rounds < 4 ? 2 : (rounds < 8 ? 4 : 8)) : 1);
			for (int r = 0; r < batch; ++r, ++rounds) {
				if (ctx->flags & NH_FLAG_EXACT_ORDER)
					NH_LAUNCH(ctx, "level_relax", k_level_relax, nh_grid_for(G, 256, 1024), 256, st, general_list, pred_a, pred_b, level, rounds);
				else if (colour_jp)
					NH_LAUNCH(ctx, "colour_round", k_colour_round, nh_grid_for(colour_n ? colour_n : G, 256, 1024), 256, st, colour_list, contacts->bodies, d->body_off, d->adj, level, rounds, colour_n);
				else {
					if (rounds == 0 && !ctx->env_colour_check_seeds) {
						NH_LAUNCH(ctx, "colour_seed", k_colour_seed_final, nh_grid_for(G, 256, 2048), 256, st, general_list, imp->data, level, d->cont.tent);
						NH_LAUNCH(ctx, "colour_validate", k_colour_validate, nh_grid_for(B, 256, 4096), 256, st, B, d->body_class, d->body_off, d->adj, level, d->cont.tent);
						continue;
					}
					if (rounds == 0)
						NH_LAUNCH(ctx, "colour_seed", k_colour_seed, nh_grid_for(G, 256, 2048), 256, st, general_list, imp->data, d->cont.tent);
					else
					NH_LAUNCH(ctx, "colour_try", k_colour_try, nh_grid_for(colour_n ? colour_n : G, 256, 2048), 256, st, colour_list, colour_n, contacts->bodies, d->body_off, d->adj, level, d->cont.tent);
					NH_LAUNCH(ctx, "colour_settle", k_colour_settle, nh_grid_for(colour_n ? colour_n : G, 256, 2048), 256, st, colour_list, colour_n, contacts->bodies, d->body_off, d->adj, level, d->cont.tent, rounds);
				}
			}
			rc = nh_read_counts(ctx, &c);
			if (rc) return rc;
			uint32_t left = ((rounds - 1) & 1u) ? ctx->h_state->unleveled_odd : ctx->h_state->unleveled;
			if (left == 0) break;
			if (rounds > NH_MAX_LEVELS + 8) return NH_ERR_INVALID;
			if (!(ctx->flags & NH_FLAG_EXACT_ORDER) && left <= (colour_n ? colour_n : G) / 2u) {
				NH_HIP_CHECK(ctx, hipMemsetAsync(&st->colour_cursor, 0, sizeof(uint32_t), ctx->stream));
				NH_LAUNCH(ctx, "colour_compact", k_colour_compact, nh_grid_for(colour_n ? colour_n : G, 256, 1024), 256, st, colour_list, colour_n, level, spare[spare_at]);
				colour_list = spare[spare_at]; spare_at ^= 1; colour_n = left;
			}
		}
		if (c.levels > NH_MAX_LEVELS) return NH_ERR_INVALID;
		d->levels = c.levels;
		// a large set in default order: contacts owned by spatial blocks, swept in LDS block by block (nh_blocks.h)
		rc = blk_setup(ctx, d, G, false);
		if (rc) return rc;
		if (d->blk.active) return NH_OK;
		NH_LAUNCH(ctx, "zero_u32", k_zero_u32, 8, 256, level_hist, 2 * (NH_MAX_LEVELS + 2));       // histogram + the per-class "has a full row" flags behind it
		NH_LAUNCH(ctx, "level_hist", k_level_hist, nh_grid_for(G, 256, 512), 256, st, general_list, level, level_hist);
		NH_LAUNCH(ctx, "level_offsets", k_level_offsets, 1, 1024, level_hist, level_cursor);
		NH_LAUNCH(ctx, "level_scatter", k_level_scatter, nh_grid_for(G, 256, 512), 256, st, general_list, level, level_cursor, d->level_order);
		NH_LAUNCH(ctx, "rows_general", k_rows_general, nh_grid_for(G, 256, 4096), 256, st, d->level_order, contacts->data, contacts->bodies, bodies->transforms, bodies->properties,
		          (float4*)d->rows, (size_t)kcap, d->gpair, (ctx->flags & NH_FLAG_EXACT_ORDER) ? 0u : 1u, d->static_inert ? 1u : 0u,
		          level, level_hist + (NH_MAX_LEVELS + 2), (const nh_CachedContactImpulse*)nullptr, (float4*)nullptr, (uint32_t*)nullptr, 0u);
		NH_HIP_CHECK(ctx, hipMemcpyAsync(d->level_off, level_hist, sizeof(uint32_t) * (d->levels + 2), hipMemcpyDeviceToHost, ctx->stream));
		NH_HIP_CHECK(ctx, hipMemcpyAsync(d->level_full, level_hist + (NH_MAX_LEVELS + 2), sizeof(uint32_t) * (d->levels + 2), hipMemcpyDeviceToHost, ctx->stream));
		NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
		// A small general set (a pile of a few hundred boxes, a pit of a thousand balls) is solved by ONE workgroup with its rows resident in LDS for
		// the warm start and all sweeps (k_solve_resident): one launch per nh_apply_impulses call instead of one per level and sweep
		bool all_radial = true;
		for (uint32_t l = 1; l <= d->levels; ++l) all_radial &= d->level_full[l] == 0u;
		d->resident_cap = all_radial ? NH_RES_MAX_RADIAL : NH_RES_MAX_FULL;
		d->resident = G <= d->resident_cap && !ctx->env_no_resident;
		d->resident_bodies = false;
		if (d->resident) {
			NH_LAUNCH(ctx, "solve_resident", k_solve_resident, 1, NH_RES_THREADS, st, level_hist, d->levels, d->level_order, d->gpair, imp->data,
			          bodies->momentum, (const float4*)d->rows, (size_t)kcap, d->gstates, 0u, 1u, d->resident_cap);
			return NH_OK;
		}
		// more rows than one compute unit's LDS holds, but a small world: every body's momentum resident instead (k_solve_resident_bodies)
		d->resident_bodies = B <= NH_RESB_MAX_BODIES && d->levels <= NH_RESB_MAX_LEVELS && !ctx->env_no_resident;
		if (d->resident_bodies) {
			NH_LAUNCH(ctx, "solve_resident_bodies", k_solve_resident_bodies, 1, NH_RESB_THREADS, st, level_hist, level_hist + (NH_MAX_LEVELS + 2), d->levels, d->level_order, d->gpair,
			          imp->data, bodies->momentum, B, (const float4*)d->rows, (size_t)kcap, d->gstates, 0u, 1u);
			return NH_OK;
		}
		// warm start in level order (levels are numbered from 1)
		for (uint32_t l = 1; l <= d->levels; ++l) {
			uint32_t b = d->level_off[l], e = d->level_off[l + 1];
			if (e > b && !d->level_full[l])
				NH_LAUNCH(ctx, "warm_level", (k_level_exec<true, true>), nh_grid_for(e - b, 256, 8192), 256, d->level_order, b, e, d->gpair, imp->data,
				          bodies->momentum, (const float4*)d->rows, (size_t)kcap, d->gstates);
			else if (e > b)
				NH_LAUNCH(ctx, "warm_level", (k_level_exec<true>), nh_grid_for(e - b, 256, 4096), 256, d->level_order, b, e, d->gpair, imp->data,
				          bodies->momentum, (const float4*)d->rows, (size_t)kcap, d->gstates);
		}
	}
	return NH_OK;
}

// Solver launches for the one-body classes.  `early4_done`: the STATIC4 bodies classified by k_adj_simple have already been taken by
// the speculative launch (only possible when body 0 is inert: that launch checks it on the device and leaves otherwise).
static void launch_apply_static(nh_context* ctx, nh_ContactConstraintData* d, const nh_BodyData* bodies, uint32_t iterations, bool fused, bool early4_done, bool drop_states = false) {
	const uint32_t B = d->body_count;
	if (!(fused && early4_done && !d->has_late && !d->has_static8 && !d->has_staticN)) ensure_csr(ctx, d);      // (the common step launches nothing below)
	nh_CacheView cv = { nullptr, nullptr, nullptr, nullptr, nullptr, ctx->d_state };
	if (fused) { cv.tags = d->imp->tags; cv.features = d->imp->features; cv.ctags = d->imp->ctags; cv.cfeatures = d->imp->cfeatures; cv.cdata = d->imp->cdata; }
	auto go = [&](const char* name, auto kernel, uint32_t cls_a, uint32_t cls_b) {
		NH_LAUNCH(ctx, name, kernel, nh_grid_for(B, 256, 8192), 256, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
		          d->rows, (float4*)d->states, iterations, d->contact_data, bodies->transforms, d->impulses, cv, cls_a, cls_b);
	};
	auto go1 = [&](const char* name, auto kernel, uint32_t cls_a, uint32_t cls_b) {
		NH_LAUNCH(ctx, name, kernel, nh_grid_for(B, 256, 8192), 256, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
		          (float4*)d->states, iterations, d->contact_data, bodies->transforms, d->impulses, cv, ctx->hint, cls_a, cls_b,
		          (fused && drop_states) ? 2u : 0u, (const uint2*)nullptr, nh_FusedStep(), nh_StillView(), nh_AheadView());
	};
	const bool late = d->has_late;
	if (d->static_inert) {
		// body 0 is inert (all-zero velocity, inverse mass, inverse inertia): one-body forms, 27-float rows
		const bool need4 = !early4_done || late;
		const uint32_t a4 = early4_done ? NH_CLS_STATIC4_LATE : NH_CLS_STATIC4;
		if (fused) {
			if (need4) go1("solve_one_body", k_solve_one_body<4, true>, a4, NH_CLS_STATIC4_LATE);
			if (d->has_static8) go1("solve_one_body8", k_solve_one_body<8, true>, NH_CLS_STATIC8, NH_CLS_STATIC8_LATE);
		} else {
			if (need4) go1("solve_one_body", k_solve_one_body<4, false>, a4, NH_CLS_STATIC4_LATE);
			if (d->has_static8) go1("solve_one_body8", k_solve_one_body<8, false>, NH_CLS_STATIC8, NH_CLS_STATIC8_LATE);
		}
	} else if (fused) {
		go("apply_static", k_apply_static<4, true>, NH_CLS_STATIC4, NH_CLS_STATIC4_LATE);
		if (d->has_static8) go("apply_static8", k_apply_static<8, true>, NH_CLS_STATIC8, NH_CLS_STATIC8_LATE);
	} else {
		go("apply_static", k_apply_static<4, false>, NH_CLS_STATIC4, NH_CLS_STATIC4_LATE);
		if (d->has_static8) go("apply_static8", k_apply_static<8, false>, NH_CLS_STATIC8, NH_CLS_STATIC8_LATE);
	}
	if (d->has_staticN && iterations) go("apply_staticN", k_apply_static<0, false>, NH_CLS_STATICN, NH_CLS_STATICN);
}

// First solver call after a setup (or its flush with zero sweeps): the common one-body kernel goes out BEFORE the counters round trip --
// the round trip then overlaps with it instead of idling the GPU -- and the rest follows once the host knows what else there is.
static int settle_gravity(nh_context* ctx) {
	if (!ctx->grav.pending) return NH_OK;
	ctx->grav.pending = false;
	NH_LAUNCH(ctx, "gravity_damping", k_gravity, nh_grid_for(ctx->grav.body_count, 256, 2048), 256, ctx->d_state, ctx->grav.active, ctx->grav.momentum,
	          ctx->grav.gx_dt, ctx->grav.gy_dt, ctx->grav.gz_dt, ctx->grav.damping);
	return NH_OK;
}

// `applied`: called by nh_apply_impulses itself (not by the flush of a setup nobody applied): with NH_FLAG_FUSED_STEP the caller has promised
// that this is the step's only solver call, so the bodies of the fused kernel are advanced there too
static inline bool cd_count_over(const nh_ColliderData& cd) { return (uint64_t)cd.boxes.count + cd.spheres.count >= (1ull << 31); }
static int first_apply(nh_context* ctx, nh_ContactConstraintData* d, const nh_BodyData* bodies, uint32_t iterations, bool applied = false) {
	// the flush of a setup nobody applied yet keeps the solver states whatever the flags say: an nh_apply_impulses may still follow (nh_read_counts
	// or nh_synchronize between setup and apply must not break the step)
	const bool from_apply = applied && (ctx->flags & NH_FLAG_FUSED_STEP), drop_states = applied && (ctx->flags & NH_FLAG_SINGLE_APPLY);
	d->states_kept = !drop_states;
	bool early4_done = false;
	if (ctx->still.active && !ctx->still.resolved) {
		// ---- still step: ONE launch, then the round trip that confirms it ----
		nh_StillStep& ss = ctx->still;
		const bool usable = applied && from_apply && drop_states && iterations != 0u && d->finish_pending && ss.setup_d == d && ctx->grav.pending && ctx->grav.momentum == bodies->momentum &&
		                    bodies->idle_counters && bodies->momentum == ss.bodies.momentum && bodies->transforms == ss.bodies.transforms && d->imp->cdata == ss.cache_data && ss.slots_current;
		bool confirmed = false;
		if (usable) {
			const uint32_t B = d->body_count;
			nh_CacheView cv = { nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const nh_CachedContactImpulse*>(ctx->sc_imp), ctx->d_state };
			nh_StillView sv = { ctx->body_rec, ctx->body_pos, ctx->cnt_sorted, ctx->start_sorted, ctx->raw_feature, ctx->sc_feat, ctx->sc_count, ctx->lay_capacity, ctx->still_delta, ctx->step_parity };
			nh_FusedStep fs = nh_FusedStep();
			fs.gx_dt = ctx->grav.gx_dt; fs.gy_dt = ctx->grav.gy_dt; fs.gz_dt = ctx->grav.gz_dt; fs.damping = ctx->grav.damping; fs.time_step = ctx->grav.time_step;
			fs.bits = 3u | (ss.sleepers ? 4u : 0u) | ((ss.sleepers && !ss.no_sleeper_skip) ? 8u : 0u); fs.idle = bodies->idle_counters; fs.seq = ctx->collide_seq;          // (bit 3: waves of sleeping bodies leave at once)
			fs.ghost_first = ctx->halo_ghost_first;
			// (late verdicts: the solver's first thread writes the step's counters into the pinned ring slot itself -- not the fused look-back kernel, which keeps the copy)
			const bool self_report = ss.pipelined && ss.h_ring[0] && !ss.fused;
			fs.host_counters = self_report ? reinterpret_cast<uint32_t*>(ss.h_ring[ctx->collide_seq & 1u]) : nullptr;
			fs.guard_seq = ss.verdict.pending ? ss.verdict.seq : ctx->collide_seq;          // (a solver behind an unconfirmed still step leaves when THAT one failed)
#define NH_SOLVE_STILL(NWAVES) NH_LAUNCH(ctx, "solve_still", (k_solve_one_body<4, true, true, NWAVES, true>), (B + 64u * NWAVES - 1u) / (64u * NWAVES), 64 * NWAVES, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum, \
		          (float4*)d->states, iterations, ctx->raw_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, 2u, d->simple, fs, sv, nh_AheadView())
			// XFORM AHEAD (nh_internal.h): another sub-step of this nh_step call follows, the step is in the plain form and every dynamic body has its one collider on the
			// map -- the lanes do the next step's k_xform<true> on their way out
			ctx->halo_split.launched = false;          // (this step's solver has not gone out in two parts yet)
			const bool ahead = ss.ahead_plain && ss.more_steps && ss.ahead_map_ok && !ss.no_ahead && !ss.ahead_world_bad && ctx->env_solver_waves != 4 && ctx->env_solver_waves != 2 &&
			                   ctx->own_xf && ctx->body_col && ctx->body_col_capacity >= B;
			// PAIR AHEAD (nh_internal.h): ... and the next sub-step's narrowphase for the body's own pair -- that sub-step then starts at the solver
			const bool pair = ahead && !ss.fused && !ss.no_pair && !ss.pair_world_bad && ss.pair_owned_seq != 0u && ctx->pair_list && ctx->fat_pairs && ctx->own_ctag && !(cd_count_over(ss.colliders));
			if (ss.fused) {
				// ONE launch for the whole step (nh_fused.hip): pair evaluation, solver, advance and the next sub-step's transforms
				int rc = nh_fused_launch(ctx, bodies, iterations, fs.guard_seq);
				if (rc) return rc;
			} else if (ahead) {
				const nh_ColliderData& cd = ss.colliders;
				const uint32_t C = cd.boxes.count + cd.spheres.count;
				const uint8_t* gen = (C < (1u << NH_GEN_SHIFT) && !ctx->env_no_incremental) ? ctx->fat_gen : (const uint8_t*)nullptr;
				nh_AheadView av = { ctx->body_col, cd.boxes.transforms, cd.boxes.data, cd.spheres.transforms, cd.spheres.data, cd.boxes.count, ctx->own_xf, ctx->own_aabb_min, ctx->own_aabb_max, ctx->fat_box,
				                    ctx->fat_pairs, ctx->fat_pair_capacity, ctx->lay_capacity, ctx->sort_keys_by_position, ctx->own_ctag, gen,
				                    ctx->rec, ctx->raw_data, ctx->raw_feature, ctx->cnt_sorted, ctx->still_delta, ctx->pair_list_capacity, C > (2u << 20) ? 1u : 0u };
				// HALO SPLIT (nh_partition_step): the bodies the neighbours wait for on a second stream, behind the step's prologue like the interior -- the exchange of the NEXT
				// sub-step is enqueued behind them once the verdict is in (below) and travels while the interior is being solved
				const bool split = pair && ctx->halo_split.armed && ss.early_verdict && ss.steps_left > 0u && !ctx->timing && ctx->halo_split.bound != 0u;
				if (split) {
					nh_FusedStep fi = fs, fb = fs;
					fi.part = 2u; fi.xskip = ctx->halo_split.mask;
					fb.part = 1u; fb.xlist = ctx->halo_split.list; fb.xcount = ctx->halo_split.count; fb.host_counters = nullptr;
					NH_LAUNCH(ctx, "solve_still", (k_solve_one_body<4, true, true, 1, true, true, true, true>), (B + 63u) / 64u, 64, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
					          (float4*)d->states, iterations, ctx->raw_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, 2u, d->simple, fi, sv, av);
					NH_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->halo_split.stream, ss.ev_ring[0], 0));          // (the counters' copy behind k_pair_begin: the prologue has run)
					hipLaunchKernelGGL((k_solve_one_body<4, true, true, 1, true, true, true, true>), dim3((ctx->halo_split.bound + 63u) / 64u), dim3(64), 0, ctx->halo_split.stream,
					                   B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
					                   (float4*)d->states, iterations, ctx->raw_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, 2u, d->simple, fb, sv, av);
					ctx->halo_split.launched = true;
				} else if (pair)
				NH_LAUNCH(ctx, "solve_still", (k_solve_one_body<4, true, true, 1, true, true, true>), (B + 63u) / 64u, 64, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
				          (float4*)d->states, iterations, ctx->raw_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, 2u, d->simple, fs, sv, av);
				else
				NH_LAUNCH(ctx, "solve_still", (k_solve_one_body<4, true, true, 1, true, true>), (B + 63u) / 64u, 64, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
				          (float4*)d->states, iterations, ctx->raw_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, 2u, d->simple, fs, sv, av);
			} else
			if (ctx->env_solver_waves == 4) NH_SOLVE_STILL(4); else if (ctx->env_solver_waves == 2) NH_SOLVE_STILL(2); else NH_SOLVE_STILL(1);
			ss.ahead_ready = ahead || ss.fused;
			ss.pair_ready = pair && !ss.fused;
			if (ss.pipelined && ss.h_ring[0]) {
				// nh_step: first the verdict of the still step BEFORE this one (its counters landed long ago) ...
				if (ss.verdict.pending) {
					NH_HIP_CHECK(ctx, hipEventSynchronize(ss.ev_ring[ss.verdict.slot]));
					const nh_DevState* h = ss.h_ring[ss.verdict.slot];
					if (h->still_failed_seq >= ss.verdict.seq || h->error) return NH_INTERNAL_STILL_FAILED;       // (nh_step cleans up and runs both steps again)
					ss.confirmed_seq = ss.verdict.seq;
					if (ss.verdict_fused && ctx->fused.unconfirmed) ctx->fused.unconfirmed--;
					memcpy(ctx->h_state, h, NH_COUNTER_WORDS * sizeof(uint32_t));
					if (!ctx->idle_unknown) { ctx->idle_bound = (int)h->max_idle[ss.verdict.parity]; ctx->idle_bound_mark = ss.verdict.collide_mark; }
					ss.verdict.pending = false;
					still_note_movers(ctx, h, ss.verdict.seq);
				}
				// ... then this step's counters on their way, to be looked at by the next step
				const int slot = (int)(ctx->collide_seq & 1u);
				if (!self_report) NH_HIP_CHECK(ctx, hipMemcpyAsync(ss.h_ring[slot], ctx->d_state, NH_COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
				NH_HIP_CHECK(ctx, hipEventRecord(ss.ev_ring[slot], ctx->stream));
				ss.verdict.pending = true; ss.verdict.seq = ctx->collide_seq; ss.verdict.parity = ctx->step_parity; ss.verdict.slot = slot; ss.verdict.collide_mark = ctx->collide_mark;
				ss.verdict.self_report = self_report && !ctx->halo_split.launched;
				ss.verdict_fused = ss.fused;
				confirmed = true;          // (until the next step says otherwise)
			} else if (ss.early_verdict) {
				// nh_partition_step, a step that started at the solver: everything such a step can fail on is checked by its PROLOGUE (k_pair_begin: the step-wide words,
				// body 0, the pairs that are nobody's or a ghost's -- the lanes' checks belong to the NEXT step), whose counters nh_collide sent on their way before this
				// solver was launched.  The host waits for THAT copy -- the solver runs on -- so the next sub-step's halo and launches queue up behind it: no bubble
				ss.early_verdict = false;
				NH_HIP_CHECK(ctx, hipEventSynchronize(ss.ev_ring[0]));
				const nh_DevState* h = ss.h_ring[0];
				confirmed = h->still_failed_seq < ctx->collide_seq && h->error == 0u;
				if (confirmed) {
					ss.confirmed_seq = ctx->collide_seq;
					memcpy(ctx->h_state, h, NH_COUNTER_WORDS * sizeof(uint32_t));
					if (!ctx->idle_unknown) { ctx->idle_bound = (int)h->max_idle[ctx->step_parity]; ctx->idle_bound_mark = ctx->collide_mark; }
					still_note_movers(ctx, h, ctx->collide_seq);
					// (halo split: the step happened -- what the neighbours get of it may leave now, behind the boundary launch on its stream)
					if (ctx->halo_split.launched && ctx->halo_split.after_verdict) { int rc = ctx->halo_split.after_verdict(ctx, ctx->halo_split.user); if (rc) return rc; }
				} else {
					NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));          // (the solver has left at once; the replay starts from a quiet stream like every other)
					if (ctx->halo_split.launched) { NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->halo_split.stream)); ctx->halo_split.launched = false; }
				}
			} else {
				// the step's one round trip: did every check hold?
				NH_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_state, ctx->d_state, NH_COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
				NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
				if (ctx->timing) nh_timer_collect(ctx);
				confirmed = ctx->h_state->still_failed_seq < ctx->collide_seq && ctx->h_state->error == 0u;
				if (confirmed) ss.confirmed_seq = ctx->collide_seq;
				if (confirmed) { nh_Counts c; nh_counts_from_mirror(ctx, &c); still_note_movers(ctx, ctx->h_state, ctx->collide_seq); }              // (the sleep prediction: largest idle counter this step saw)
			}
		}
		if (confirmed) {
			ss.resolved = true; ss.setup_d = nullptr; ss.note_confirmed();
			ss.cache_stale = true; ss.contacts_stale = true;        // (the caller's cache and dense contact arrays are last written by an earlier step: nh_still_sync_outputs)
			if (ss.sleepers) { ss.views_sleepers = true; ss.sleep_pairs_current = false; ss.active_current = false; }            // (... and so are its active list and sleeping pairs)
			d->finish_pending = false; d->has_late = false; d->has_static8 = d->has_staticN = false; d->static_inert = true; d->general_contacts = 0; d->levels = 0;
			d->imp->cull_pending = false; d->imp->lookup_pending = false;
			ctx->last_general_contacts = 0;
			ctx->grav.pending = false; ctx->grav.rest_pending = false;
			ctx->adv.done = true; ctx->adv.time_step = ctx->grav.time_step; ctx->adv.body_class = d->body_class; ctx->adv.rest = false;
			return NH_OK;
		}
		// not a still step after all (a check failed on the device, or the caller left the sample's order): the step again, in full
		{ int rc = nh_still_abandon(ctx); if (rc) return rc; }
	}
	if (!d->finish_pending) { int rc = settle_gravity(ctx); if (rc) return rc; }
	if (d->finish_pending) {
		d->finish_pending = false;
		const uint32_t B = d->body_count;
		nh_CacheView cv = { d->imp->tags, d->imp->features, d->imp->ctags, d->imp->cfeatures, d->imp->cdata, ctx->d_state };
		nh_FusedStep fs = nh_FusedStep();
		const bool gravity_here = ctx->grav.pending && ctx->grav.momentum == bodies->momentum;
		if (ctx->grav.pending && !gravity_here) { int rc = settle_gravity(ctx); if (rc) return rc; }
		if (gravity_here) {
			ctx->grav.pending = false;
			fs.gx_dt = ctx->grav.gx_dt; fs.gy_dt = ctx->grav.gy_dt; fs.gz_dt = ctx->grav.gz_dt; fs.damping = ctx->grav.damping; fs.time_step = ctx->grav.time_step;
			fs.bits = 1u;
			if (from_apply && bodies->idle_counters) { fs.bits |= 2u; fs.idle = bodies->idle_counters; ctx->adv.done = true; ctx->adv.time_step = fs.time_step; ctx->adv.body_class = d->body_class; }
		}
#define NH_SOLVE_CONTIG(NWAVES) NH_LAUNCH(ctx, "solve_one_body", (k_solve_one_body<4, true, true, NWAVES>), (B + 64u * NWAVES - 1u) / (64u * NWAVES), 64 * NWAVES, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum, \
		          (float4*)d->states, iterations, d->contact_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, \
		          1u | (drop_states ? 2u : 0u), d->simple, fs, nh_StillView(), nh_AheadView())
		// (early counters: not with EVERY launch timed -- the events of a step are collected at its round trip, when they must have happened; timing restricted to one kernel, what
		// bench.py keeps on, collects at nh_kernel_times -- and not under the legacy observer contract, whose round trips do more than read)
		d->early_seq = 0u;
		if (ctx->h_early && !ctx->no_early_counts && B != 0u && !(ctx->timing && ctx->timing_filter.empty()) && !ctx->sync_exports_views && !(ctx->flags & NH_FLAG_SYNC_COUNTS)) {
			fs.host_counters = ctx->h_early; fs.seq = ctx->collide_seq; d->early_seq = ctx->collide_seq;
		}
		if (ctx->env_solver_waves == 4) NH_SOLVE_CONTIG(4); else if (ctx->env_solver_waves == 2) NH_SOLVE_CONTIG(2); else NH_SOLVE_CONTIG(1);
		ctx->grav.rest_pending = gravity_here;        // everybody else's gravity: finish_setup, once the round trip has told whether there is anybody else
		int rc = finish_setup(ctx, d);
		if (rc) return rc;
		early4_done = d->static_inert;          // (if body 0 is not inert the speculative launch left at once and the general kernels take everything)
	}
	launch_apply_static(ctx, d, bodies, iterations, true, early4_done, drop_states);
	return NH_OK;
}

int nh_still_collide_again(nh_context* ctx);       // nh_collide.hip
int nh_asleep_verify(nh_context* ctx, const nh_BodyData* bodies, const nh_ColliderData* colliders);      // nh_collide.hip
int nh_still_view_contacts(nh_context* ctx);       // nh_collide.hip: the dense contact list laid out again from the slots (k_gather_contacts, view only)

// ---- the contact cache between its two homes: the caller's arrays (tag order) and the slots (nh_internal.h, contact storage by slot) ------------------------
// full step -> slots: the solved impulse and feature word of dense contact c go to the raw slot c came from (k_gather_contacts recorded it), every record's count
__global__ __launch_bounds__(256) void k_cache_to_slots(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ dense_slot, const nh_CachedContactImpulse* __restrict__ imp,
                                                        const uint32_t* __restrict__ features, const nh_Record* __restrict__ rec, float4* __restrict__ sc_imp, uint32_t* __restrict__ sc_feat,
                                                        uint32_t* __restrict__ sc_count) {
	const uint32_t n = st->contacts, nrec = st->records;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const uint32_t slot = dense_slot[c];
		sc_imp[slot] = *reinterpret_cast<const float4*>(imp + c);
		sc_feat[slot] = features[c];
	}
	for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += gridDim.x * blockDim.x) { const uint32_t k = rec[r].count; sc_count[r] = (k & NH_REC_SLEEPING) ? 0u : k; }
}

// entries kept aside for sleeping pairs (the culled arrays of this step's nh_ContactImpulseData: tag order, st->culled of them) -> the slots of the pairs' records.  A
// record is found by its key -- a sleeping record's key is the pair's word, which is what the cache tag equals for every entry that was kept (k_cull_flags) -- in the
// tag order of the layout; an entry's place among its record's slots is its place in the run of equal tags (the entries are ranked by feature word, like the slots' export)
__global__ __launch_bounds__(256) void k_culled_to_slots(const nh_DevState* __restrict__ st, const uint64_t* __restrict__ ctags, const uint32_t* __restrict__ cfeatures,
                                                         const nh_CachedContactImpulse* __restrict__ cdata, const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ sorted_idx,
                                                         const nh_Record* __restrict__ rec, uint32_t pair_cap, float4* __restrict__ sc_imp, uint32_t* __restrict__ sc_feat, uint32_t* __restrict__ sc_count) {
	const uint32_t n = st->culled, nrec = st->records, n_bb = min(st->pairs, pair_cap);
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const uint64_t t = ctags[j];
		uint32_t k = 0;                                   // place in the run of equal tags
		while (k < 4u && j > k && ctags[j - 1u - k] == t) ++k;
		uint32_t lo = 0, hi = nrec;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted_keys[mid] < t) lo = mid + 1u; else hi = mid; }
		if (lo >= nrec || sorted_keys[lo] != t) continue;
		const uint32_t ri = sorted_idx[lo];
		if (!(rec[ri].count & NH_REC_SLEEPING)) continue;
		const bool sph = ri >= n_bb;
		if (k >= (sph ? 1u : 4u)) continue;
		const uint32_t base = sph ? 4u * n_bb + (ri - n_bb) : 4u * ri;
		sc_imp[base + k] = *reinterpret_cast<const float4*>(cdata + j);
		sc_feat[base + k] = cfeatures[j];
		if (j + 1u >= n || ctags[j + 1u] != t) sc_count[ri] = k + 1u;          // (the last entry of the run knows how many there are)
	}
}

// slots -> the caller's cache arrays, in tag order: counts in tag order (scanned by the host's launch in between), then the entries ranked by feature word
__global__ __launch_bounds__(256) void k_slot_counts_sorted(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ sc_count, uint32_t* __restrict__ out) {
	const uint32_t nrec = st->records;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= nrec; i += gridDim.x * blockDim.x) out[i] = i < nrec ? min(sc_count[sorted_idx[i]], 4u) : 0u;
}

__global__ __launch_bounds__(256) void k_slots_to_cache(nh_DevState* __restrict__ st, const uint32_t* __restrict__ sorted_idx, const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ cnt,
                                                        const uint32_t* __restrict__ start, const float4* __restrict__ sc_imp, const uint32_t* __restrict__ sc_feat, uint32_t pair_cap,
                                                        uint64_t* __restrict__ otags, uint32_t* __restrict__ ofeatures, nh_CachedContactImpulse* __restrict__ odata, uint32_t capacity) {
	const uint32_t nrec = st->records, n_bb = min(st->pairs, pair_cap);
	if (blockIdx.x == 0 && threadIdx.x == 0) { const uint32_t total = start[nrec]; if (total > capacity) { st->error = NH_ERR_CACHE_CAPACITY; st->cache = 0u; } else st->cache = total; }
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += gridDim.x * blockDim.x) {
		// two dependent round trips: the record's place and count, then its four slots' feature words and impulses at once (a record of a pair with a sphere owns one
		// slot: the other three loads fall into the neighbours' slots and are not used)
		const uint32_t k = min(cnt[i], 4u), ri = sorted_idx[i], first = start[i];
		const uint64_t key = sorted_keys[i];
		const uint32_t base = ri < n_bb ? 4u * ri : 4u * n_bb + (ri - n_bb);
		uint32_t f[4]; float4 w[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) { f[j] = sc_feat[base + j]; w[j] = sc_imp[base + j]; }
		if (!k || first + k > capacity) continue;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if ((uint32_t)j < k) {
				uint32_t r = 0;
#pragma unroll
				for (int q = 0; q < 4; ++q) r += ((uint32_t)q < k && (f[q] < f[j] || (f[q] == f[j] && q < j))) ? 1u : 0u;
				otags[first + r] = key; ofeatures[first + r] = f[j];
				*reinterpret_cast<float4*>(odata + first + r) = w[j];
			}
		}
	}
}

// Sleepers form: the slot-cache counts a still narrowphase dropped in a step that did not happen (nh_internal.h: sc_undo) come back.  Called wherever a still step is
// given up -- failed on the device (still_forget_failed), or left by the caller between its nh_collide and its solver (nh_still_abandon) -- BEFORE the slot cache goes
// home to the caller's arrays, which is what the full replay warm-starts from.  Notes of confirmed steps (numbers <= confirmed_seq) are simply cleared.
__global__ __launch_bounds__(256) void k_sleep_undo(uint64_t* __restrict__ sc_undo, uint32_t* __restrict__ sc_count, uint32_t n, uint32_t confirmed_seq) {
	for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
		const uint64_t u = sc_undo[r];
		if (u == 0ull) continue;
		if ((uint32_t)(u >> 32) > confirmed_seq) sc_count[r] = (uint32_t)u;
		sc_undo[r] = 0ull;
	}
}

int nh_still_undo_drops(nh_context* ctx) {
	nh_StillStep& ss = ctx->still;
	if (!ss.undo_dirty || !ctx->sc_undo || !ctx->lay_capacity) return NH_OK;
	ss.undo_dirty = false;
	NH_LAUNCH(ctx, "sleep_undo", k_sleep_undo, nh_grid_for(ctx->lay_capacity, 256, 2048), 256, ctx->sc_undo, ctx->sc_count, ctx->lay_capacity, ss.confirmed_seq);
	return NH_OK;
}

int nh_still_export_cache(nh_context* ctx) {
	nh_StillStep& ss = ctx->still;
	if (!ss.cache_stale) return NH_OK;
	ss.cache_stale = false;
	nh_DevState* st = ctx->d_state;
	const uint32_t P = ctx->lay_capacity;
	// (scratch: the tag-order starts of the dense VIEW are this step's -- a failed still step may already have overwritten them -- so the cache gets its own scan;
	// dense_slot is free between two full steps)
	uint32_t* tmp_cnt = ctx->exp_cnt; uint32_t* tmp_start = ctx->exp_start;
	NH_LAUNCH(ctx, "slot_counts", k_slot_counts_sorted, nh_grid_for(P, 256, 2048), 256, st, ctx->sort_sorted_idx, ctx->sc_count, tmp_cnt);
	nh_scan_u32(ctx, tmp_cnt, tmp_start, &st->records, 1, ctx->exp_scan_tmp, nullptr);
	NH_LAUNCH(ctx, "slots_to_cache", k_slots_to_cache, nh_grid_for(P, 256, 4096), 256, st, ctx->sort_sorted_idx, ctx->sort_sorted_keys, tmp_cnt, tmp_start, ctx->sc_imp, ctx->sc_feat, P,
	          ss.cache.tags, ss.cache.features, ss.cache.data, ss.cache.capacity);
	return NH_OK;
}

// ---- views of a world with sleepers (nh_internal.h, "SLEEPERS form"): the list of sleeping pairs and the active list, as a full step would have written them -------
__global__ __launch_bounds__(256) void k_view_sleep_keys(const nh_DevState* __restrict__ st, const nh_Record* __restrict__ rec, const uint64_t* __restrict__ rec_key, uint64_t* __restrict__ out) {
	const uint32_t n = st->records;
	for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) out[r] = (rec[r].count & NH_REC_SLEEPING) ? rec_key[r] : ~0ull;      // (a sleeping record carries the pair's word)
}
__global__ __launch_bounds__(256) void k_view_sleep_copy(nh_DevState* __restrict__ st, const uint64_t* __restrict__ sorted, uint64_t* __restrict__ out, uint32_t capacity) {
	const uint32_t n = st->sleeping;
	if (n > capacity) { if (blockIdx.x == 0 && threadIdx.x == 0) st->error = NH_ERR_CONTACT_CAPACITY; return; }
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = sorted[i];
}
__global__ __launch_bounds__(256) void k_view_awake_flags(const uint8_t* __restrict__ awake, uint32_t nbodies, uint32_t* __restrict__ flags) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= nbodies; i += gridDim.x * blockDim.x) flags[i] = (i >= 1u && i < nbodies && awake[i]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_view_active_write(nh_DevState* __restrict__ st, const uint8_t* __restrict__ awake, const uint32_t* __restrict__ scan, uint32_t nbodies,
                                                           uint32_t* __restrict__ indices, uint32_t capacity) {
	for (uint32_t i = 1u + blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		if (!awake[i]) continue;
		const uint32_t at = scan[i];
		if (at < capacity) indices[at] = i; else st->error = NH_ERR_ACTIVE_CAPACITY;
	}
}

static int still_view_sleepers(nh_context* ctx, uint32_t what) {
	nh_StillStep& ss = ctx->still;
	nh_DevState* st = ctx->d_state;
	if (what & NH_VIEW_CONTACTS) {
		// sleeping pairs in ascending order (nudge.cpp:4008): the words of the sleeping records, sorted -- every other record sorts behind them
		const uint32_t P = ctx->lay_capacity;
		if (ss.lay_contacts.sleeping_pairs) {
			if (ctx->exp_sleep_capacity < P) {
				void** bufs[] = { (void**)&ctx->exp_sleep_a, (void**)&ctx->exp_sleep_b, (void**)&ctx->exp_sleep_hist };
				for (void** b : bufs) { if (*b) NH_HIP_CHECK(ctx, hipFree(*b)); *b = nullptr; }
				ctx->exp_sleep_capacity = 0;
				NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->exp_sleep_a, sizeof(uint64_t) * (size_t)P + 64u));
				NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->exp_sleep_b, sizeof(uint64_t) * (size_t)P + 64u));
				NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->exp_sleep_hist, sizeof(uint32_t) * (256u * NH_SORT_GRID + 512u)));
				ctx->exp_sleep_capacity = P;
			}
			NH_LAUNCH(ctx, "view_sleep_keys", k_view_sleep_keys, nh_grid_for(P, 256, 2048), 256, st, ctx->rec, ctx->sort_keys_by_position, ctx->exp_sleep_a);
			uint64_t* a = ctx->exp_sleep_a; uint64_t* b = ctx->exp_sleep_b;
			int bits = (int)ctx->tag_bits; if (bits < 1) bits = 1; if (bits > 32) bits = 32;
			const int top = ((bits + 7) / 8) * 8;
			if (nh_sort_u64(ctx, a, b, &st->records, ctx->exp_sleep_hist, 0, top)) { uint64_t* t = a; a = b; b = t; }
			if (nh_sort_u64(ctx, a, b, &st->records, ctx->exp_sleep_hist, 32, 32 + top)) { uint64_t* t = a; a = b; b = t; }
			NH_LAUNCH(ctx, "view_sleep_copy", k_view_sleep_copy, nh_grid_for(P, 256, 1024), 256, st, a, ss.lay_contacts.sleeping_pairs, ss.lay_contacts.capacity);
		}
	}
	if ((what & NH_VIEW_ACTIVE) && ss.lay_active && ctx->still_awake) {
		const uint32_t B = ss.lay_bodies.count;
		if (ctx->exp_flags_capacity < B + 2u) {
			if (ctx->exp_flags) NH_HIP_CHECK(ctx, hipFree(ctx->exp_flags));
			ctx->exp_flags = nullptr; ctx->exp_flags_capacity = 0;
			NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->exp_flags, sizeof(uint32_t) * ((size_t)B + 66u)));
			ctx->exp_flags_capacity = B + 2u;
		}
		NH_LAUNCH(ctx, "view_awake_flags", k_view_awake_flags, nh_grid_for(B, 256, 2048), 256, ctx->still_awake, B, ctx->exp_flags);
		nh_scan_u32(ctx, ctx->exp_flags, ctx->exp_flags, &st->pad0 /* always 0 */, B, ctx->exp_scan_tmp, nullptr);
		NH_LAUNCH(ctx, "view_active_write", k_view_active_write, nh_grid_for(B, 256, 2048), 256, st, ctx->still_awake, ctx->exp_flags, B, const_cast<uint32_t*>(ss.lay_active), ss.lay_active_capacity);
	}
	return NH_OK;
}

int nh_still_sync_outputs(nh_context* ctx, uint32_t what) {
	nh_StillStep& ss = ctx->still;
	if (ss.active && !ss.resolved) return NH_OK;          // (an unconfirmed still step is abandoned by the caller first: nh_flush_pending)
	int rc = NH_OK;
	if (what & NH_VIEW_CACHE) rc = nh_still_export_cache(ctx);
	if (rc) return rc;
	if ((what & NH_VIEW_CONTACTS) && ss.contacts_stale) { ss.contacts_stale = false; rc = nh_still_view_contacts(ctx); if (!rc && ss.views_sleepers) rc = still_view_sleepers(ctx, NH_VIEW_CONTACTS); ss.sleep_pairs_current = true; }
	else if ((what & NH_VIEW_CONTACTS) && ss.views_sleepers && !ss.sleep_pairs_current) { rc = still_view_sleepers(ctx, NH_VIEW_CONTACTS); ss.sleep_pairs_current = true; }
	if (rc) return rc;
	if ((what & NH_VIEW_ACTIVE) && ss.views_sleepers && !ss.active_current) { rc = still_view_sleepers(ctx, NH_VIEW_ACTIVE); ss.active_current = true; }
	return rc;
}

extern "C" int nh_export_views(nh_context* ctx, uint32_t what) {
	if (!ctx || (what & ~(uint32_t)NH_VIEW_ALL)) return NH_ERR_INVALID;          // (NH_VIEW_ALL = contacts | cache | active)
	NH_HIP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx, true); if (rc) return rc; }
	return nh_still_sync_outputs(ctx, what);
}

// A still step that has not been confirmed, met by anything but the next call of the sample's order (or failed on the device): the step is launched again in full.
int nh_still_abandon(nh_context* ctx) {
	nh_StillStep& ss = ctx->still;
	if (!ss.active || ss.resolved || ss.replaying) return NH_OK;
	nh_ContactConstraintData* d = ss.setup_d;
	ss.setup_d = nullptr;
	int rc = nh_still_undo_drops(ctx);                    // (what this step's narrowphase dropped from the slot cache in sleepers form comes back first)
	if (rc) return rc;
	rc = nh_still_export_cache(ctx);                      // (the slot cache holds the last confirmed step's impulses: the full solver reads the caller's arrays)
	if (rc) return rc;
	ss.contacts_stale = false;                         // (the replay lays the dense list out itself)
	rc = nh_still_collide_again(ctx);
	if (rc) return rc;
	if (d) {
		// the adjacency kernel the still setup left out (default order only: still steps are not launched in exact-order mode)
		const uint32_t B = d->body_count;
		NH_LAUNCH(ctx, "adjacency_simple", k_adj_simple, nh_grid_for(B, 256, 4096), 256, ctx->d_state, B, ctx->deg, reinterpret_cast<const unsigned long long*>(ctx->deg + 2u * NH_DEG_STRIDE(B)),
		          ctx->deg + 4u * NH_DEG_STRIDE(B), (const uint32_t*)nullptr, d->body_class, d->bodies_at_setup.properties, d->bodies_at_setup.momentum, d->simple,
		          ctx->body_rec, ctx->body_pos, ctx->sort_seeded ? ctx->sort_sorted_idx : (const uint32_t*)nullptr);
	}
	return NH_OK;
}

int nh_flush_pending(nh_context* ctx, bool keep_gravity, bool in_sequence) {
	if (!in_sequence) { int rc = nh_still_abandon(ctx); if (rc) return rc; }
	ctx->gravity_may_overlap = false;            // some entry point other than nh_collide is enqueueing work: gravity no longer "directly follows" it
	ctx->after_collide = false;
	nh_ContactConstraintData* d = ctx->pending;
	if (d) {
		ctx->pending = nullptr;
		if (d->setup_pending) {
			d->setup_pending = false;
			int rc = first_apply(ctx, d, &d->bodies_at_setup, 0);       // zero sweeps: lookup + rows + warm start only
			if (rc) return rc;
			if (d->general_contacts && d->blk.active && d->blk.warm_pending) { d->blk.warm_pending = false; blk_run(ctx, d, &d->bodies_at_setup, true, 0u); }
		}
	}
	// NH_FLAG_FUSED_STEP: gravity waits for the solver only across nh_read_cached_impulses / nh_setup_contact_constraints / nh_apply_impulses
	return keep_gravity ? NH_OK : settle_gravity(ctx);
}

extern "C" int nh_apply_impulses(nh_context* ctx, nh_ContactConstraintData* d, const nh_BodyData* bodies, uint32_t iterations) {
	if (!ctx || !d || !bodies) return NH_ERR_INVALID;
	if (ctx->pending && ctx->pending != d) { int rc = nh_flush_pending(ctx); if (rc) return rc; }
	if (ctx->still.active && !ctx->still.resolved && !(ctx->pending == d && d->setup_pending)) { int rc = nh_still_abandon(ctx); if (rc) return rc; }
	ctx->gravity_may_overlap = false; ctx->after_collide = false;
	const bool first = d->setup_pending;
	if (!first && !d->states_kept) return NH_ERR_INVALID;      // NH_FLAG_SINGLE_APPLY: the states of the first call were not kept
	d->setup_pending = false;
	ctx->pending = nullptr;
	if (first) { int rc = first_apply(ctx, d, bodies, iterations, true); if (rc) return rc; }
	else { int rc = settle_gravity(ctx); if (rc) return rc; launch_apply_static(ctx, d, bodies, iterations, false, false); }
	if (d->general_contacts && d->blk.active) {
		const bool warm = d->blk.warm_pending;
		d->blk.warm_pending = false;
		blk_run(ctx, d, bodies, warm, iterations);
	} else if (d->general_contacts && d->resident) {
		// small general set: one workgroup, rows in LDS for all sweeps (k_solve_resident)
		if (iterations)
			NH_LAUNCH(ctx, "solve_resident", k_solve_resident, 1, NH_RES_THREADS, ctx->d_state, d->cont.level_hist, d->levels, d->level_order, d->gpair, (const nh_CachedContactImpulse*)nullptr,
			          bodies->momentum, (const float4*)d->rows, (size_t)d->contact_capacity, d->gstates, iterations, 0u, d->resident_cap);
	} else if (d->general_contacts && d->resident_bodies) {
		if (iterations)
			NH_LAUNCH(ctx, "solve_resident_bodies", k_solve_resident_bodies, 1, NH_RESB_THREADS, ctx->d_state, d->cont.level_hist, d->cont.level_hist + (NH_MAX_LEVELS + 2), d->levels,
			          d->level_order, d->gpair, (const nh_CachedContactImpulse*)nullptr, bodies->momentum, d->body_count, (const float4*)d->rows, (size_t)d->contact_capacity, d->gstates,
			          iterations, 0u);
	} else if (d->general_contacts) {
		for (uint32_t it = 0; it < iterations; ++it)
			for (uint32_t l = 1; l <= d->levels; ++l) {
				uint32_t b = d->level_off[l], e = d->level_off[l + 1];
				if (e > b && !d->level_full[l])
					NH_LAUNCH(ctx, "apply_level", (k_level_exec<false, true>), nh_grid_for(e - b, 256, 8192), 256, d->level_order, b, e, d->gpair, (const nh_CachedContactImpulse*)nullptr,
					          bodies->momentum, (const float4*)d->rows, (size_t)d->contact_capacity, d->gstates);
				else if (e > b)
					NH_LAUNCH(ctx, "apply_level", (k_level_exec<false>), nh_grid_for(e - b, 256, 4096), 256, d->level_order, b, e, d->gpair, (const nh_CachedContactImpulse*)nullptr,
					          bodies->momentum, (const float4*)d->rows, (size_t)d->contact_capacity, d->gstates);
			}
	}
	return NH_OK;
}

extern "C" int nh_update_cached_impulses(nh_context* ctx, nh_ContactConstraintData* d, nh_ContactImpulseData* imp) {
	if (!ctx || !d || !imp) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	if (d->general_contacts)
		NH_LAUNCH(ctx, "update_impulses", k_update_impulses, nh_grid_for(d->general_contacts, 256, 4096), 256, ctx->d_state, d->level_order, (const float4*)d->rows, (size_t)d->contact_capacity,
		          (const float4*)d->gstates, imp->data, (ctx->flags & NH_FLAG_EXACT_ORDER) ? (const uint32_t*)nullptr : (const uint32_t*)((d->blk.active && d->blk.local) ? d->cont.tent : d->cont.level), d->gpair,
		          d->blk.active ? d->blk.res_off[0] : 0u);
	return NH_OK;
}

// ---- halo records (multi-GPU partitions, include/nudge_hip.h) ---------------------------------------------------------------------
struct nh_HaloRecord { float position[3]; float rotation[4]; float momentum[8]; uint32_t idle; };     // 64 B
static_assert(sizeof(nh_HaloRecord) == NH_HALO_RECORD_BYTES, "halo record layout");

__global__ __launch_bounds__(256) void k_halo_pack(const nh_Transform* __restrict__ xf, const nh_BodyMomentum* __restrict__ momentum, const uint8_t* __restrict__ idle,
                                                   const uint32_t* __restrict__ indices, uint32_t count, nh_HaloRecord* __restrict__ out) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
		const uint32_t i = indices[k];
		const float4 t0 = reinterpret_cast<const float4*>(xf + i)[0], t1 = reinterpret_cast<const float4*>(xf + i)[1];
		const float4 m0 = reinterpret_cast<const float4*>(momentum + i)[0], m1 = reinterpret_cast<const float4*>(momentum + i)[1];
		float4* o = reinterpret_cast<float4*>(out + k);
		o[0] = make_float4(t0.x, t0.y, t0.z, t1.x);
		o[1] = make_float4(t1.y, t1.z, t1.w, m0.x);
		o[2] = make_float4(m0.y, m0.z, m0.w, m1.x);
		o[3] = make_float4(m1.y, m1.z, m1.w, __uint_as_float((uint32_t)idle[i]));
	}
}

__global__ __launch_bounds__(256) void k_halo_unpack(nh_Transform* __restrict__ xf, nh_BodyMomentum* __restrict__ momentum, uint8_t* __restrict__ idle,
                                                     uint32_t first_slot, uint32_t count, const nh_HaloRecord* __restrict__ in) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
		const uint32_t i = first_slot + k;
		const float4* r = reinterpret_cast<const float4*>(in + k);
		const float4 a = r[0], b = r[1], c = r[2], d = r[3];
		nh_Transform* t = xf + i;
		t->position[0] = a.x; t->position[1] = a.y; t->position[2] = a.z;             // t->body keeps what the owner of the slot put there
		reinterpret_cast<float4*>(t)[1] = make_float4(a.w, b.x, b.y, b.z);
		reinterpret_cast<float4*>(momentum + i)[0] = make_float4(b.w, c.x, c.y, c.z);
		reinterpret_cast<float4*>(momentum + i)[1] = make_float4(c.w, d.x, d.y, d.z);
		idle[i] = (uint8_t)__float_as_uint(d.w);
	}
}

extern "C" int nh_halo_pack(nh_context* ctx, const nh_BodyData* bodies, const uint32_t* indices, uint32_t count, void* out) {
	if (!ctx || !bodies || (count && (!indices || !out))) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	if (count) NH_LAUNCH(ctx, "halo_pack", k_halo_pack, nh_grid_for(count, 256, 1024), 256, bodies->transforms, bodies->momentum, bodies->idle_counters, indices, count, (nh_HaloRecord*)out);
	return NH_OK;
}

// nh_halo_update: the same for records of the SAME bodies that already occupy those slots, one step further on their owner (the per-step halo of a
// partitioned world).  Their idle counters rise by at most one per step there as here, so what the last measuring nh_collide saw of them still bounds
// them: the host's sleep prediction survives (a sleeper the prediction had ruled out is still caught on the device: NH_ERR_STALE_HINT).
static int halo_unpack(nh_context* ctx, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in, bool same_bodies);
extern "C" int nh_halo_update(nh_context* ctx, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in) { return halo_unpack(ctx, bodies, first_slot, count, in, true); }
extern "C" int nh_halo_unpack(nh_context* ctx, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in) { return halo_unpack(ctx, bodies, first_slot, count, in, false); }

static int halo_unpack(nh_context* ctx, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in, bool same_bodies) {
	if (!ctx || !bodies || (count && !in) || (uint64_t)first_slot + count > bodies->count) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	if (count && !same_bodies) { ctx->idle_bound = -1; ctx->idle_unknown = true; }             // idle counters arrive from another world: nothing is known about them until the next nh_collide has looked
	if (count) NH_LAUNCH(ctx, "halo_unpack", k_halo_unpack, nh_grid_for(count, 256, 1024), 256, bodies->transforms, bodies->momentum, bodies->idle_counters, first_slot, count, (const nh_HaloRecord*)in);
	return NH_OK;
}

// (the two kernels on a stream of the caller's choice, nothing completed, nothing noted: nh_partition_step's exchange beside the interior's solver -- halo split)
void nh_halo_pack_on(hipStream_t stream, const nh_BodyData* bodies, const uint32_t* indices, uint32_t count, void* out) {
	if (count) hipLaunchKernelGGL(k_halo_pack, dim3(nh_grid_for(count, 256, 1024)), dim3(256), 0, stream, bodies->transforms, bodies->momentum, bodies->idle_counters, indices, count, (nh_HaloRecord*)out);
}
void nh_halo_update_on(hipStream_t stream, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in) {
	if (count) hipLaunchKernelGGL(k_halo_unpack, dim3(nh_grid_for(count, 256, 1024)), dim3(256), 0, stream, bodies->transforms, bodies->momentum, bodies->idle_counters, first_slot, count, (const nh_HaloRecord*)in);
}

// ---- nh_step: the sample's sub-step loop (example/main.cpp:274-328) as one entry point ----------------------------------------------------------------------
// The eight calls, `steps` times.  Besides sparing the caller eight crossings of the ABI per step, the library -- driving the call order itself -- may look at a still
// step's verdict one step late (nh_internal.h: nh_StillStep::pipelined), so that neither the host nor the GPU ever waits for the other inside the loop; a failed
// still step and the one launched behind it (both did nothing) are simply run again.  On return every step is confirmed.
int nh_still_verdict_now(nh_context* ctx) {
	// the pending verdict, waited for: 0 confirmed, 1 failed
	nh_StillStep& ss = ctx->still;
	if (!ss.verdict.pending) return 0;
	const nh_DevState* h = ss.h_ring[ss.verdict.slot];
	bool seen = false;
	if (ss.verdict.self_report && !ctx->no_early_counts && !(ctx->timing && ctx->timing_filter.empty())) {
		// early counters (nh_internal.h): the step's solver left counters and number in the ring slot as it STARTED -- everything a still step can fail on was decided by
		// then -- so the call's last verdict does not wait for the launch to end (the event is the fallback: a launch that never ran writes no number)
		volatile const uint32_t* const flag = reinterpret_cast<volatile const uint32_t*>(h) + NH_COUNTER_WORDS;
		for (uint32_t spins = 0; ; ++spins) {
			if (*flag == ss.verdict.seq) { seen = true; break; }
			if ((spins & 255u) == 255u && hipEventQuery(ss.ev_ring[ss.verdict.slot]) != hipErrorNotReady) break;
			__builtin_ia32_pause();
		}
		if (seen) { std::atomic_thread_fence(std::memory_order_acquire); ctx->early_reads++; }
	}
	if (!seen && hipEventSynchronize(ss.ev_ring[ss.verdict.slot]) != hipSuccess) return 1;
	ss.verdict.pending = false;
	if (h->still_failed_seq >= ss.verdict.seq || h->error) return 1;
	ss.confirmed_seq = ss.verdict.seq;
	if (ss.verdict_fused && ctx->fused.unconfirmed) ctx->fused.unconfirmed--;
	memcpy(ctx->h_state, h, NH_COUNTER_WORDS * sizeof(uint32_t));
	if (!ctx->idle_unknown) { ctx->idle_bound = (int)h->max_idle[ss.verdict.parity]; ctx->idle_bound_mark = ss.verdict.collide_mark; }
	still_note_movers(ctx, h, ss.verdict.seq);
	return 0;
}

// after a failed still step: nothing of it (or of the step launched behind it) has happened; the next nh_collide is a full one
static int still_forget_failed(nh_context* ctx, bool advanced, uint32_t voided) {
	nh_StillStep& ss = ctx->still;
	ss.verdict.pending = false;
	ss.active = false; ss.resolved = false; ss.setup_d = nullptr; ss.ok_next = false; ss.ahead_ready = false;
	ss.note_failure();
	if (ss.sleepers) {
		// (sleepers ahead: somebody fell asleep, most likely.  One sleeper now and then costs the form eight steps; failures in quick succession -- a world dozing off in a
		// trickle -- double that up to 64)
		if (ss.sleep_backoff_len < 8u || ss.sleep_run >= 32u) ss.sleep_backoff_len = 8u; else if (ss.sleep_backoff_len < 64u) ss.sleep_backoff_len *= 2u;
		ss.sleep_backoff = ss.sleep_backoff_len; ss.sleep_stable = 0u; ss.sleep_run = 0u;
	}
	ss.failed += voided;                               // (still steps launched that did not happen: the failed one, and the one launched behind it if it got that far)
	ctx->pending = nullptr;
	ctx->grav.pending = false; ctx->grav.rest_pending = false; ctx->adv.done = false;
	ctx->after_collide = false; ctx->gravity_may_overlap = false;
	if (advanced && ctx->advance_count) ctx->advance_count--;          // (the failed step's nh_advance was counted: the sleep prediction counts real ones)
	// fused launches among the voided steps (nh_fused.hip) wrote into the OTHER set of buffers: the sets change back, and if the state they were given lives in the
	// shadow arrays it goes home -- the replay is a full step, in place
	nh_fused_void_last(ctx, ctx->fused.unconfirmed);
	ss.fused = false; ss.verdict_fused = false;
	{ int rc = nh_fused_come_home(ctx); if (rc) return rc; }
	if (advanced) nh_stream_void_advance(ctx);          // (... and so does the state stream; a frame taken at that nh_advance shows the state BEFORE the step: it is withdrawn)
	{ int rc = nh_still_undo_drops(ctx); if (rc) return rc; }          // (sleepers form: slot-cache counts the voided steps' narrowphases dropped)
	return nh_still_export_cache(ctx);                                 // (the slot cache holds the last step that DID happen: the full solver reads the caller's arrays)
}

static bool same_bodies_arrays(const nh_BodyData& a, const nh_BodyData& b) {
	return a.transforms == b.transforms && a.properties == b.properties && a.momentum == b.momentum && a.idle_counters == b.idle_counters && a.count == b.count;
}
static bool same_collider_arrays(const nh_ColliderData& a, const nh_ColliderData& b) {
	return a.boxes.tags == b.boxes.tags && a.boxes.data == b.boxes.data && a.boxes.transforms == b.boxes.transforms && a.boxes.count == b.boxes.count &&
	       a.spheres.tags == b.spheres.tags && a.spheres.data == b.spheres.data && a.spheres.transforms == b.spheres.transforms && a.spheres.count == b.spheres.count;
}

extern "C" int nh_step(nh_context* ctx, const nh_StepArgs* a, uint32_t steps) {
	if (!ctx || !a || !a->active_bodies || !a->contacts || !a->bodies || !a->colliders || !a->contact_cache) return NH_ERR_INVALID;
	NH_HIP_CHECK(ctx, hipSetDevice(ctx->device));
	nh_StillStep& ss = ctx->still;
	if (!ss.h_ring[0] && !ss.ring_failed) {
		for (int k = 0; k < 2; ++k) {
			if (hipHostMalloc((void**)&ss.h_ring[k], sizeof(nh_DevState), 0) != hipSuccess || hipEventCreateWithFlags(&ss.ev_ring[k], hipEventDisableTiming) != hipSuccess) { ss.ring_failed = true; break; }
			memset(ss.h_ring[k], 0, sizeof(nh_DevState));
		}
		if (ss.ring_failed) {
			// whatever was created goes back, both slots end empty, and the loop runs with the verdict inside every step from now on (no retry per call)
			for (int k = 0; k < 2; ++k) {
				if (ss.h_ring[k]) { (void)hipHostFree(ss.h_ring[k]); ss.h_ring[k] = nullptr; }
				if (ss.ev_ring[k]) { (void)hipEventDestroy(ss.ev_ring[k]); ss.ev_ring[k] = nullptr; }
			}
		}
	}
	// (per-kernel timing of EVERY launch collects its events at the step's round trip; timing restricted to one kernel -- nh_set_timing_filter: two events per step, what
	// bench.py keeps on during its timed region -- leaves the loop as it is: the events are collected when the call's last verdict has been waited for)
	ss.pipelined = ss.h_ring[0] != nullptr && (ctx->flags & NH_FLAG_FUSED_STEP) && !(ctx->flags & (NH_FLAG_SYNC_COUNTS | NH_FLAG_EXACT_ORDER)) && !ss.disabled &&
	               !(ctx->timing && ctx->timing_filter.empty()) &&
	               !ctx->step_hook;          // (nh_partition_step: what a step sends to the neighbours must be a step that HAPPENED -- its verdict is looked at inside the step)
	ss.verdict.pending = false;
	ss.ahead_map_ok = false; ss.ahead_ready = false;          // (xform ahead, nh_internal.h: nothing carries over from another call -- the caller may have changed anything in between)
	int result = NH_OK;
	uint32_t i = 0;
	while (i < steps || ss.verdict.pending) {
		if (i >= steps) {
			// the last step's verdict, waited for; a failure sends the loop back one step
			if (nh_still_verdict_now(ctx) == 0) break;
			{ int rc = still_forget_failed(ctx, true, 1u); if (rc) { result = rc; break; } }
			i -= 1;
			continue;
		}
		// A world asleep (nh_internal.h: nh_AsleepState): two full steps in a row were its fixed point.  ONE check per call that nothing the caller owns has changed since
		// -- and the remaining steps of this call are done: nothing inside the library wakes a world in which nobody is awake.
		// (never inside nh_partition_step: the neighbours expect this rank's halo before every sub-step, whether anything moves here or not)
		if (ctx->asleep.streak >= 2u && !ctx->asleep.disabled && !ctx->step_hook && !ss.verdict.pending && !ctx->pending && (ctx->flags & NH_FLAG_FUSED_STEP) && !(ctx->flags & NH_FLAG_SYNC_COUNTS) && !ctx->timing &&
		    same_bodies_arrays(*a->bodies, ss.lay_bodies) && same_collider_arrays(*a->colliders, ss.lay_colliders) && a->contacts->data == ss.lay_contacts.data && a->contacts->tags == ss.lay_contacts.tags &&
		    a->contacts->sleeping_pairs == ss.lay_contacts.sleeping_pairs && a->active_bodies->indices == ss.lay_active && a->contact_cache->data == ss.cache_data && a->contact_cache->tags == ss.cache_tags && a->contact_cache->features == ss.cache_features) {
			const int v = nh_asleep_verify(ctx, a->bodies, a->colliders);
			if (v < 0) { result = -v; break; }
			if (v == 0) { ss.sleep_backoff = ss.sleep_backoff > steps - i ? ss.sleep_backoff - (steps - i) : 0u; ctx->asleep.steps += steps - i; if (ctx->stream_state.every) ctx->stream_state.advances += steps - i; i = steps; continue; }          // (no frames: nothing moves)
			ctx->asleep.streak = 0;
		}
		int rc;
		if (ctx->step_hook && (rc = ctx->step_hook(ctx, ctx->step_hook_user, i))) { result = rc; break; }
		ss.more_steps = i + 1u < steps; ss.steps_left = steps - 1u - i; ss.substep = i;
		nh_Arena temporary = a->arena;
		nh_ContactImpulseData* imp = nullptr;
		nh_ContactConstraintData* con = nullptr;
		if ((rc = nh_collide(ctx, a->active_bodies, a->contacts, a->bodies, a->colliders, a->body_connections, temporary)) ||
		    (rc = nh_apply_gravity_damping(ctx, a->active_bodies, a->bodies, a->time_step, a->gravity, a->damping_rate)) ||
		    (rc = nh_read_cached_impulses(ctx, a->contact_cache, a->contacts, &temporary, &imp)) ||
		    (rc = nh_setup_contact_constraints(ctx, a->active_bodies, a->contacts, a->bodies, imp, &temporary, &con)) ||
		    (rc = nh_apply_impulses(ctx, con, a->bodies, a->iterations))) {
			if (rc == NH_INTERNAL_STILL_FAILED && i > 0) {
				// the still step before this one failed: neither it nor this one has happened
				{ int rc2 = still_forget_failed(ctx, true, ctx->still.active ? 2u : 1u); if (rc2) { result = rc2; break; } }
				i -= 1;
				continue;
			}
			result = rc == NH_INTERNAL_STILL_FAILED ? NH_ERR_INVALID : rc;
			break;
		}
		if ((rc = nh_update_cached_impulses(ctx, con, imp)) || (rc = nh_write_cached_impulses(ctx, a->contact_cache, a->contacts, imp)) ||
		    (rc = nh_advance(ctx, a->active_bodies, a->bodies, a->time_step))) { result = rc; break; }
		++i;
	}
	ss.pipelined = false; ss.more_steps = false; ss.ahead_ready = false; ss.ahead_map_ok = false; ss.steps_left = 0u; ss.fused = false; ss.substep = 0u;
	if (result && ss.verdict.pending) { hipEventSynchronize(ss.ev_ring[ss.verdict.slot]); ss.verdict.pending = false; }
	// (fused steps are launched in pairs and every verdict of the call is in: the bodies are in the caller's arrays -- unless the call ends in an error)
	if (ctx->fused.cur != 0u) { ctx->fused.unconfirmed = 0u; int rc = nh_fused_come_home(ctx); if (rc && !result) result = rc; }
	ctx->fused.unconfirmed = 0u;
	return result;
}
