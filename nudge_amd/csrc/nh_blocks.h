// nh_blocks.h -- spatially blocked, LDS-resident coloured sweeps for LARGE general contact sets (included by nh_solve.hip).
//
// The problem (config 4: one connected pit of 4 M spheres, 11 M contacts, 13 colours, 16 sweeps).  A colour class is a set of contacts that share
// no dynamic body, so one launch may solve it in any order; the sweep is "colour 1, colour 2, ...".  Solved straight from HBM (k_level_exec) every
// colour launch gathers and scatters two 32-byte momentum records per contact from an array that does not survive in L2 from one launch to the
// next: thirteen passes over the whole momentum array per sweep, 224 launches per step, bound by memory transactions in flight.
//
// What this file does instead (the north star's "constraints staged in LDS across solver iterations", for sets no single workgroup can hold):
//   * the general bodies are binned into cubic BLOCKS of a uniform grid (edge chosen so that a block's bodies fit one workgroup's LDS:
//     NH_BLK_CAP momentum records, two workgroups per compute unit); a private copy of their momentum is kept in block order (`bm`), so a block's
//     own bodies are one contiguous, coalesced read;
//   * a contact belongs to ONE block: the block of its two bodies if they share one (or the other is the static world); else, if the two centres
//     are "near" (no coordinate differs by half a block edge), the block with the lower BLOCK COLOUR (parity of the three block coordinates, 8
//     colours), which borrows the other body as a GHOST for the duration of its launch; else the contact is RESIDUAL (a body larger than half a
//     block) and is solved by the per-colour global launches as before;
//   * a sweep is eight launches, one per block colour.  Blocks of one colour are never adjacent, and by the "near" rule two of them can never borrow
//     the same ghost (the ghost's centre would have to be within half an edge of two bodies that lie more than an edge apart), so all blocks of a
//     launch work on disjoint bodies.  Inside a launch one workgroup owns a block: it loads the block's bodies and ghosts into LDS ONCE, walks
//     the contact colours of its own list with __syncthreads() in between (rows, local index pairs and solver states stream in block order,
//     prefetched one colour ahead), and writes the momentum back ONCE.
// Momentum then crosses HBM about twice per sweep instead of thirteen times, and a sweep is 8 launches instead of one per colour.
// Order of a sweep: block colour by block colour, contact colour by contact colour inside a block, residual colours last -- every contact once,
// never two contacts of one body at a time: a valid Gauss-Seidel sweep, like the colour order it replaces a different one than the reference's
// (nudge.cpp:4646-4854 walks its greedy batches), i.e. the statistical-parity regime of SURVEY appendix C; radial rows as in k_level_exec.
//
// Everything is rebuilt every step from the contact list (positions, colours and pairs change every step in a pit that is still settling):
// bounds -> grid -> block and rank of every body -> owner and (block, colour) bin of every contact -> two scans -> scatter -> per block the ghost
// table and the local index pairs (k_blk_prepare, one workgroup per block, a hash table in LDS).  The host adapts the block edge from the largest
// load it reads back with the step's other counters, and retries with smaller blocks when a block overflows.
#ifndef NH_BLOCKS_H
#define NH_BLOCKS_H

#ifndef NH_BLK_CAP
#define NH_BLK_CAP 2400u                 // momentum records in one workgroup's LDS: slot 0 = the static world, own bodies, ghosts (76.8 KB: two workgroups per CU)
#define NH_BLK_THREADS 512
#endif
#define NH_BLK_RANK_BITS 12u
#define NH_BLK_RANK_MASK ((1u << NH_BLK_RANK_BITS) - 1u)
#define NH_BLK_HASH 4096u                // ghost table of k_blk_prepare (open addressing in LDS)
#define NH_BLK_NEAR 500                  // of 1024 per block edge: "near" = no coordinate of the two centres differs by more (12 units of slack for the quantisation)
#define NH_BLK_RESIDUAL NH_BLK_MAX_BLOCKS
#define NH_BLK_GROUPS 2                  // wavefront groups of a sweep workgroup that take the colour chunks in turn (below)
#define NH_BLK_CHUNK (NH_BLK_THREADS / NH_BLK_GROUPS)      // contacts of one colour solved in one step
#define NH_BLK_MAX_CHUNKS 96u            // steps of one block: its colours cut into chunks (a block with more is too big: smaller blocks)
#define NH_BLK_LOCAL_L 32u               // colours of the per-block colouring (k_blk_prepare_local); the descriptor holds L + 1 offsets
#define NH_BLK_PREP_THREADS 1024         // threads of k_blk_prepare_local (a block's ~2500 contacts: two or three per thread and pass; 512 threads: 0.74 ms, 256: 1.15 ms in the pit)
#define NH_BLK_MAX_CONTACTS 6144u        // contacts of one block the per-block colouring holds in LDS (a block with more is too big: smaller blocks)
#define NH_BLK_DESC_WORDS 40u            // per (block colour, workgroup) descriptor: block, own bodies, first slot, ghosts, ghost list base, the L + 1 bin offsets

// (nh_glds16 -- 16 bytes from a per-lane global address straight into LDS: nh_internal.h)

// ... the same with system scope (sc0 sc1): the line is fetched from memory whatever this XCD's L2 holds -- for data another workgroup of the SAME launch has
// written with system-scope stores (chained sweeps below; MI355X: the L2s of the eight XCDs are not coherent with each other)
__device__ __forceinline__ void nh_glds16_sys(const float4* src, float4* lds_dst) {
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 17);
}
typedef float nh_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nh_store16_sys(float4* dst, float4 v) {
	const nh_v4f w = { v.x, v.y, v.z, v.w };
	asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(w) : "memory");
}
#define NH_BLK_NBR_WORDS 28u             // chained sweeps: per task the number of tasks it waits for, then up to 26 of them (+ padding)
#define NH_BLK_CHAIN_PASSES 64u
struct nh_BlkPhases { uint32_t start[9]; };          // workgroups of the eight block colours, as ranges of one launch
struct nh_BlkChain { nh_BlkPhases ph; const uint32_t* nbr; uint32_t* done; uint32_t* ticket; uint32_t tag; uint32_t* error; };

// Places at per-key counters for the lanes of a wave (`valid` lanes; `key` = index of the counter): lanes with the same key find each other by ballots, and then ALL
// group leaders ask for their group's places at once -- one memory round trip per call, not one per distinct key in the wave (a returning atomic is a round trip the
// wave waits for; contacts and bodies arrive in spatial order, so a wave meets a handful of keys)
__device__ __forceinline__ uint32_t nh_wave_places(uint32_t* __restrict__ counters, bool valid, uint32_t key) {
	const uint32_t lane = nh_lane();
	uint32_t leader = lane, before = 0, group = 0;
	unsigned long long todo = __ballot(valid);
	while (todo) {
		const int first = __ffsll((long long)todo) - 1;
		const uint32_t k = (uint32_t)__shfl((int)key, first);
		const unsigned long long same = __ballot(valid && key == k) & todo;
		if ((same >> lane) & 1ull) { leader = (uint32_t)first; before = (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); group = (uint32_t)__popcll(same); }
		todo &= ~same;
	}
	uint32_t base = 0;
	if (valid && leader == lane) base = atomicAdd(&counters[key], group);
	base = (uint32_t)__shfl((int)base, (int)leader);
	return base + before;
}

__device__ __forceinline__ uint32_t blk_slot(const uint32_t* __restrict__ own_base, uint32_t br) { return 1u + own_base[br >> NH_BLK_RANK_BITS] + (br & NH_BLK_RANK_MASK); }

__global__ void k_blk_reset(nh_DevState* st) {
	nh_BlkGrid& g = st->blk;
	for (int k = 0; k < 3; ++k) { g.bmin[k] = 0xffffffffu; g.bmax[k] = 0u; }
	g.bodies = 0; g.nblk = 0; g.overflow = 0; g.max_load = 0; g.ghost_cursor = 0; g.check_errors = 0; g.near_contacts = 0; g.far_contacts = 0; g.max_colours = 0; g.too_many_colours = 0; g.max_rounds = 0; g.max_listed = 0;
	for (int k = 0; k < 8; ++k) g.phase_count[k] = 0;
}

// bounds of the general bodies' centres
__global__ __launch_bounds__(256) void k_blk_bounds(nh_DevState* __restrict__ st, uint32_t nbodies, const uint8_t* __restrict__ body_class, const nh_Transform* __restrict__ xf) {
	__shared__ uint32_t s_lo[3][4], s_hi[3][4], s_n[4];
	uint32_t lo[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, hi[3] = { 0u, 0u, 0u }, n = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		if (i == 0u || body_class[i] != NH_CLS_GENERAL) continue;
		const float4 t = reinterpret_cast<const float4*>(xf + i)[0];
		const uint32_t f[3] = { nh_float_flip(t.x), nh_float_flip(t.y), nh_float_flip(t.z) };
#pragma unroll
		for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], f[k]); hi[k] = max(hi[k], f[k]); }
		++n;
	}
	for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
		for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], d)); hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], d)); }
		n += (uint32_t)__shfl_xor((int)n, d);
	}
	if ((threadIdx.x & 63u) == 0u) {
		for (int k = 0; k < 3; ++k) { s_lo[k][threadIdx.x >> 6] = lo[k]; s_hi[k][threadIdx.x >> 6] = hi[k]; }
		s_n[threadIdx.x >> 6] = n;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const uint32_t total = s_n[0] + s_n[1] + s_n[2] + s_n[3];
		if (total) {
			// (a look first, past this XCD's L2: bounds stand after the first few workgroups, and a thousand workgroups x 7 atomics on one line are most of this kernel's time)
			for (int k = 0; k < 3; ++k) {
				const uint32_t lo_k = min(min(s_lo[k][0], s_lo[k][1]), min(s_lo[k][2], s_lo[k][3])), hi_k = max(max(s_hi[k][0], s_hi[k][1]), max(s_hi[k][2], s_hi[k][3]));
				if (lo_k < __hip_atomic_load(&st->blk.bmin[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&st->blk.bmin[k], lo_k);
				if (hi_k > __hip_atomic_load(&st->blk.bmax[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->blk.bmax[k], hi_k);
			}
			atomicAdd(&st->blk.bodies, total);
		}
	}
}

// the grid: the edge at which an average block (over the axes the world extends along) holds `target` bodies, times the host's correction;
// block coordinates are bit fields of the block id (at most NH_BLK_ID_BITS bits in all: the edge grows until they fit)
__global__ void k_blk_grid(nh_DevState* st, float scale, float target) {
	nh_BlkGrid& g = st->blk;
	g.nblk = 0;
	if (!g.bodies) return;
	float lo[3], e[3];
	for (int k = 0; k < 3; ++k) {
		lo[k] = nh_float_unflip(g.bmin[k]);
		e[k] = nh_float_unflip(g.bmax[k]) - lo[k];
		if (!(e[k] >= 0.0f) || !(e[k] < 1e30f) || !(lo[k] > -1e30f)) return;              // non-finite world: no grid (the caller falls back to the global launches)
	}
	float edge = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], 1e-3f)) * 1.001f;
	for (int it = 0; it < 600; ++it) {
		float cells = 1.0f;
		for (int k = 0; k < 3; ++k) cells *= fmaxf(e[k] / edge, 1.0f);
		if ((float)g.bodies / cells <= target) break;
		edge *= 0.97f;
	}
	edge *= scale;
	uint32_t bits[3], sum;
	for (;;) {
		sum = 0;
		for (int k = 0; k < 3; ++k) {
			const float cells = fminf(e[k] / edge, 1048575.0f);
			const uint32_t dims = (uint32_t)cells + 1u;
			uint32_t b = 0;
			while ((1u << b) < dims) ++b;
			bits[k] = b; sum += b;
		}
		if (sum <= NH_BLK_ID_BITS) break;
		edge *= 1.26f;
	}
	g.edge = edge; g.inv_edge = 1.0f / edge;
	for (int k = 0; k < 3; ++k) { g.origin[k] = lo[k]; g.bits[k] = bits[k]; }
	g.nblk = 1u << sum;
}

// block and rank of every general body; rank = arrival order at the block's counter (wave-aggregated: bodies come in index order, which is
// spatial order in a generated scene, so the lanes of a wave hit one or two counters).  brank.y = position inside the block, 3 x 10 bits.
__global__ __launch_bounds__(256) void k_blk_body(nh_DevState* __restrict__ st, uint32_t nbodies, const uint8_t* __restrict__ body_class, const nh_Transform* __restrict__ xf,
                                                  uint32_t* __restrict__ blk_count, uint2* __restrict__ brank) {
	if (!st->blk.nblk) return;
	const float o[3] = { st->blk.origin[0], st->blk.origin[1], st->blk.origin[2] };
	const float inv = st->blk.inv_edge * 1024.0f;
	const uint32_t bits[3] = { st->blk.bits[0], st->blk.bits[1], st->blk.bits[2] };
	const uint32_t rounded = (nbodies + 63u) & ~63u;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += gridDim.x * blockDim.x) {
		const bool general = i >= 1u && i < nbodies && body_class[i] == NH_CLS_GENERAL;
		uint32_t blk = 0, frac = 0;
		if (general) {
			const float4 t = reinterpret_cast<const float4*>(xf + i)[0];
			const float p[3] = { t.x, t.y, t.z };
			uint32_t shift = 0;
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				const int lim = (int)((1u << bits[k]) * 1024u) - 1;
				int q = (int)((p[k] - o[k]) * inv);
				q = q < 0 ? 0 : (q > lim ? lim : q);
				blk |= ((uint32_t)q >> 10) << shift;
				frac |= ((uint32_t)q & 1023u) << (10 * k);
				shift += bits[k];
			}
		}
		uint32_t rank = nh_wave_places(blk_count, general, blk);
		if (general) {
			if (rank >= NH_BLK_RANK_MASK) { st->blk.overflow = 1; rank = NH_BLK_RANK_MASK; }
			brank[i] = make_uint2(blk << NH_BLK_RANK_BITS | rank, frac);
		} else if (i < nbodies) brank[i] = make_uint2(NH_NONE, 0u);
	}
}

// owner block and (block, colour) bin of every general contact; rank in the bin = arrival order at its counter
__global__ __launch_bounds__(256) void k_blk_contact(nh_DevState* __restrict__ st, const uint32_t* __restrict__ general_list, const nh_BodyPair* __restrict__ bodies,
                                                     const uint32_t* __restrict__ level, const uint2* __restrict__ brank, uint32_t L, uint32_t* __restrict__ bins,
                                                     uint32_t* __restrict__ rank_out, uint32_t* __restrict__ bin_out) {
	if (!st->blk.nblk) return;
	const uint32_t n = st->general_contacts;
	const uint32_t b0 = st->blk.bits[0], b1 = st->blk.bits[1], b2 = st->blk.bits[2];
	const uint32_t m0 = (1u << b0) - 1u, m1 = (1u << b1) - 1u, m2 = (1u << b2) - 1u;
	uint32_t near = 0, far = 0;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t c = general_list[k];
		const nh_BodyPair p = bodies[c];
		uint32_t l = level ? level[c] : 1u;          // (no colours yet: the blocks colour their own contacts, k_blk_prepare_local; L = 1)
		if (l < 1u || l > L) { st->error = NH_ERR_INVALID; l = 1u; }
		const uint2 ra = p.a ? brank[p.a] : make_uint2(NH_NONE, 0u), rb = p.b ? brank[p.b] : make_uint2(NH_NONE, 0u);
		if ((p.a && ra.x == NH_NONE) || (p.b && rb.x == NH_NONE) || (!p.a && !p.b)) { st->error = NH_ERR_INVALID; rank_out[k] = 0; bin_out[k] = NH_NONE; continue; }
		uint32_t owner;
		const uint32_t A = ra.x >> NH_BLK_RANK_BITS, B = rb.x >> NH_BLK_RANK_BITS;
		if (!p.a) owner = B;
		else if (!p.b || A == B) owner = A;
		else {
			const int ca[3] = { (int)(A & m0), (int)((A >> b0) & m1), (int)((A >> (b0 + b1)) & m2) };
			const int cb[3] = { (int)(B & m0), (int)((B >> b0) & m1), (int)((B >> (b0 + b1)) & m2) };
			bool is_near = true;
			uint32_t col_a = 0, col_b = 0;
#pragma unroll
			for (int q = 0; q < 3; ++q) {
				const int qa = (ca[q] << 10) | (int)((ra.y >> (10 * q)) & 1023u), qb = (cb[q] << 10) | (int)((rb.y >> (10 * q)) & 1023u);
				const int dq = qa - qb;
				is_near = is_near && dq <= NH_BLK_NEAR && dq >= -NH_BLK_NEAR;
				col_a |= (uint32_t)(ca[q] & 1) << q; col_b |= (uint32_t)(cb[q] & 1) << q;
			}
			if (is_near) { owner = col_a < col_b ? A : B; ++near; }
			else { owner = NH_BLK_RESIDUAL; ++far; }
		}
		const uint32_t bin = owner * L + (l - 1u);
		// (contacts come in tag order, which is spatial order in a generated scene: the lanes of a wave hit a handful of bins -- one atomic per bin and wave;
		// with one bin per block, 2500 contacts each, a returning atomic per contact cost 1.9 ms in a pit of 11 M contacts)
		const uint32_t rnk = nh_wave_places(bins, true, bin);
		rank_out[k] = rnk;
		bin_out[k] = bin;
	}
	// (the `near` count was a statistic nobody read: thirty thousand waves adding to ONE word queue for ~12 ns each -- 0.39 of this kernel's 0.40 ms in a pit of 11 M
	//  contacts.  `far` -- contacts no block can own -- is rare, and the host wants it)
	(void)near;
	if (__ballot(far != 0u) != 0ull) {
		for (int d = 32; d >= 1; d >>= 1) far += (uint32_t)__shfl_xor((int)far, d);
		if (nh_lane() == 0u && far) atomicAdd(&st->blk.far_contacts, far);
	}
}

__global__ __launch_bounds__(256) void k_blk_scatter(const nh_DevState* __restrict__ st, const uint32_t* __restrict__ general_list, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ bin,
                                                     const uint32_t* __restrict__ bin_off, uint32_t* __restrict__ order) {
	if (!st->blk.nblk) return;
	const uint32_t n = st->general_contacts;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t b = bin[k];
		if (b != NH_NONE) order[bin_off[b] + rank[k]] = general_list[k];
	}
}

// One workgroup per block: the ghosts (bodies of neighbour blocks that this block's contacts touch), numbered through a hash table in LDS; the
// block-order slots of the ghosts; the local index pair of every contact of the block (0 = the static world, 1.. = own bodies by rank, then ghosts).
__global__ __launch_bounds__(256) void k_blk_prepare(nh_DevState* __restrict__ st, uint32_t L, const uint32_t* __restrict__ bin_off, const uint32_t* __restrict__ blk_count,
                                                     const uint32_t* __restrict__ own_base, const uint32_t* __restrict__ order, const nh_BodyPair* __restrict__ bodies,
                                                     const uint2* __restrict__ brank, uint32_t* __restrict__ lpair, uint32_t* __restrict__ ghost_base, uint32_t* __restrict__ ghost_count,
                                                     uint32_t* __restrict__ ghost_list, uint32_t* __restrict__ phase_list, uint32_t* __restrict__ desc) {
	__shared__ uint32_t s_key[NH_BLK_HASH], s_val[NH_BLK_HASH];
	__shared__ uint32_t s_n, s_base, s_fail, s_slot;
	const uint32_t b = blockIdx.x, tid = threadIdx.x;
	if (b >= st->blk.nblk) return;
	const uint32_t r0 = bin_off[b * L], r1 = bin_off[(b + 1u) * L];
	if (r0 == r1) { if (tid == 0) { ghost_base[b] = 0; ghost_count[b] = 0; } return; }
	const uint32_t n_own = blk_count[b];
	for (uint32_t t = tid; t < NH_BLK_HASH; t += 256u) s_key[t] = NH_NONE;
	if (tid == 0) { s_n = 0; s_fail = 0; }
	__syncthreads();
	for (uint32_t k = r0 + tid; k < r1; k += 256u) {
		const nh_BodyPair p = bodies[order[k]];
#pragma unroll
		for (int side = 0; side < 2; ++side) {
			const uint32_t x = side ? p.b : p.a;
			if (!x || (brank[x].x >> NH_BLK_RANK_BITS) == b) continue;
			uint32_t h = (x * 2654435761u) >> (32 - 12);
			bool done = false;
			for (uint32_t probe = 0; probe < NH_BLK_HASH; ++probe) {
				const uint32_t old = atomicCAS(&s_key[h], NH_NONE, x);
				if (old == NH_NONE || old == x) { done = true; break; }
				h = (h + 1u) & (NH_BLK_HASH - 1u);
			}
			if (!done) s_fail = 1;
		}
	}
	__syncthreads();
	for (uint32_t t = tid; t < NH_BLK_HASH; t += 256u) if (s_key[t] != NH_NONE) s_val[t] = atomicAdd(&s_n, 1u);
	__syncthreads();
	const uint32_t n_gh = s_n;
	uint32_t chunks = 0;                                      // (every thread counts: L is small)
	for (uint32_t l = 0; l < L; ++l) chunks += (bin_off[b * L + l + 1u] - bin_off[b * L + l] + NH_BLK_CHUNK - 1u) / NH_BLK_CHUNK;
	const bool fits = !s_fail && 1u + n_own + n_gh <= NH_BLK_CAP && chunks <= NH_BLK_MAX_CHUNKS;
	if (tid == 0) {
		atomicMax(&st->blk.max_load, 1u + n_own + n_gh);
		if (!fits) st->blk.overflow = 1;
		s_base = fits ? atomicAdd(&st->blk.ghost_cursor, n_gh) : 0u;
		ghost_base[b] = s_base; ghost_count[b] = fits ? n_gh : 0u;
		if (fits) {
			// block colour = parity of the three block coordinates (a coordinate whose field is zero bits wide is always 0)
			const uint32_t b0 = st->blk.bits[0], b1 = st->blk.bits[1], b2 = st->blk.bits[2];
			const uint32_t c0 = b & ((1u << b0) - 1u), c1 = (b >> b0) & ((1u << b1) - 1u), c2 = (b >> (b0 + b1)) & ((1u << b2) - 1u);
			const uint32_t col = (c0 & 1u) | (c1 & 1u) << 1 | (c2 & 1u) << 2;
			const uint32_t slot = col * NH_BLK_MAX_BLOCKS + atomicAdd(&st->blk.phase_count[col], 1u);
			phase_list[slot] = b;
			s_slot = slot;
			uint32_t* dsc = desc + (size_t)slot * NH_BLK_DESC_WORDS;
			dsc[0] = b; dsc[1] = n_own; dsc[2] = 1u + own_base[b]; dsc[3] = n_gh; dsc[4] = s_base;
		}
	}
	__syncthreads();
	if (!fits) return;                                   // (the host sees the overflow flag and retries the whole setup with smaller blocks)
	if (tid <= L) desc[(size_t)s_slot * NH_BLK_DESC_WORDS + 5u + tid] = bin_off[b * L + tid];
	const uint32_t base = s_base;
	for (uint32_t t = tid; t < NH_BLK_HASH; t += 256u) if (s_key[t] != NH_NONE) ghost_list[base + s_val[t]] = blk_slot(own_base, brank[s_key[t]].x);
	for (uint32_t k = r0 + tid; k < r1; k += 256u) {
		const nh_BodyPair p = bodies[order[k]];
		uint32_t loc[2];
#pragma unroll
		for (int side = 0; side < 2; ++side) {
			const uint32_t x = side ? p.b : p.a;
			uint32_t l = 0;
			if (x) {
				const uint32_t br = brank[x].x;
				if ((br >> NH_BLK_RANK_BITS) == b) l = 1u + (br & NH_BLK_RANK_MASK);
				else {
					uint32_t h = (x * 2654435761u) >> (32 - 12);
					while (s_key[h] != x) h = (h + 1u) & (NH_BLK_HASH - 1u);
					l = 1u + n_own + s_val[h];
				}
			}
			loc[side] = l;
		}
		lpair[k] = loc[0] | loc[1] << 16;
	}
}

// The same with the COLOURING done here, block by block (when no contact is residual -- the usual case: every contact then lives in exactly one block, and
// all a block needs from a colouring is that no two contacts of one colour in ITS list share a dynamic body).  The world-wide colouring this replaces
// needs per-body adjacency lists (23 M returning atomics in a pit of 4 M spheres), a seed and a validation pass and three to five relaxation rounds with a
// host round trip each: 2 ms of the pit's 16.  Here the block's contacts and a "colours in use" mask per local body sit in LDS:
//   * SEEDS: a contact that was there last step brings the colour it had (it rides in the spare word of its cached impulse, as for the world-wide
//     colouring).  Every seeded contact ORs its colour into its two bodies' masks; a body that is offered one colour twice (two contacts that were
//     coloured in different blocks last step, or garbage in a cache the library did not write) is noted in a conflict mask, and BOTH contacts of such a
//     clash give their seed up -- symmetric, so the outcome does not depend on who came first;
//   * the rest (new contacts, a few per cent) by Jones-Plassmann rounds over a compacted list: every uncoloured contact bids for its two bodies with a
//     hashed priority (atomicMin of a 64-bit value made of the contact's index in the caller's list: no arrival order in it), a contact that holds both
//     takes the lowest colour neither body uses.
// Then a counting sort by colour inside the block's segment (positions inside a colour are arrival order: contacts of one colour share no body, so
// nothing depends on it), the descriptor with the NH_BLK_LOCAL_L + 1 colour offsets, and the colours written back for the next step's seeds.
__global__ __launch_bounds__(NH_BLK_PREP_THREADS) void k_blk_prepare_local(nh_DevState* __restrict__ st, const uint32_t* __restrict__ bin_off, const uint32_t* __restrict__ blk_count,
                                                           const uint32_t* __restrict__ own_base, const uint32_t* __restrict__ order, const nh_BodyPair* __restrict__ bodies,
                                                           const uint2* __restrict__ brank, const nh_CachedContactImpulse* __restrict__ warm, uint32_t* __restrict__ colour_out,
                                                           uint32_t* __restrict__ order_out, uint32_t* __restrict__ lpair, uint32_t* __restrict__ ghost_base,
                                                           uint32_t* __restrict__ ghost_count, uint32_t* __restrict__ ghost_list, uint32_t* __restrict__ phase_list, uint32_t* __restrict__ desc) {
	constexpr uint32_t LL = NH_BLK_LOCAL_L, LIST = 2u * (2u * NH_BLK_HASH - NH_BLK_MAX_CONTACTS);       // (16-bit entries behind the index pairs)
	__shared__ uint32_t s_tab[2u * NH_BLK_HASH];                 // the ghost table (keys, values); afterwards the local index pairs of the block's contacts + the list of the uncoloured
	__shared__ unsigned long long s_mask[NH_BLK_CAP];            // colours in use, per local body
	__shared__ unsigned long long s_claim[NH_BLK_CAP];           // seed pass: colours offered twice; Jones-Plassmann rounds: the best bid
	__shared__ uint8_t s_col[NH_BLK_MAX_CONTACTS];
	__shared__ uint32_t s_cnt[LL + 1u], s_cur[LL];
	__shared__ uint32_t s_n, s_base, s_fail, s_slot, s_nlist;
	static_assert(2u * NH_BLK_HASH > NH_BLK_MAX_CONTACTS, "the index pairs reuse the ghost table");
	uint32_t* const s_key = s_tab; uint32_t* const s_val = s_tab + NH_BLK_HASH; uint32_t* const s_lp = s_tab;
	uint16_t* const s_list = reinterpret_cast<uint16_t*>(s_tab + NH_BLK_MAX_CONTACTS);
	const uint32_t b = blockIdx.x, tid = threadIdx.x;
	if (b >= st->blk.nblk) return;
	const uint32_t r0 = bin_off[b], r1 = bin_off[b + 1u];
	if (r0 == r1) { if (tid == 0) { ghost_base[b] = 0; ghost_count[b] = 0; } return; }
	const uint32_t n_own = blk_count[b], nc = r1 - r0;
	for (uint32_t t = tid; t < NH_BLK_HASH; t += NH_BLK_PREP_THREADS) s_key[t] = NH_NONE;
	if (tid == 0) { s_n = 0; s_fail = nc > NH_BLK_MAX_CONTACTS ? 1u : 0u; s_nlist = 0; }
	__syncthreads();
	for (uint32_t k = r0 + tid; k < r1; k += NH_BLK_PREP_THREADS) {
		const nh_BodyPair p = bodies[order[k]];
#pragma unroll
		for (int side = 0; side < 2; ++side) {
			const uint32_t x = side ? p.b : p.a;
			if (!x || (brank[x].x >> NH_BLK_RANK_BITS) == b) continue;
			uint32_t h = (x * 2654435761u) >> (32 - 12);
			bool done = false;
			for (uint32_t probe = 0; probe < NH_BLK_HASH; ++probe) {
				const uint32_t old = atomicCAS(&s_key[h], NH_NONE, x);
				if (old == NH_NONE || old == x) { done = true; break; }
				h = (h + 1u) & (NH_BLK_HASH - 1u);
			}
			if (!done) s_fail = 1;
		}
	}
	__syncthreads();
	for (uint32_t t = tid; t < NH_BLK_HASH; t += NH_BLK_PREP_THREADS) if (s_key[t] != NH_NONE) s_val[t] = atomicAdd(&s_n, 1u);
	__syncthreads();
	const uint32_t n_gh = s_n, nloc = 1u + n_own + n_gh;
	const bool fits = !s_fail && nloc <= NH_BLK_CAP;
	if (tid == 0) atomicMax(&st->blk.max_load, nloc);
	if (!fits) { if (tid == 0) { st->blk.overflow = 1; ghost_base[b] = 0; ghost_count[b] = 0; } return; }             // (uniform: the host retries with smaller blocks)
	if (tid == 0) { s_base = atomicAdd(&st->blk.ghost_cursor, n_gh); ghost_base[b] = s_base; ghost_count[b] = n_gh; }
	__syncthreads();
	const uint32_t base = s_base;
	for (uint32_t t = tid; t < NH_BLK_HASH; t += NH_BLK_PREP_THREADS) if (s_key[t] != NH_NONE) ghost_list[base + s_val[t]] = blk_slot(own_base, brank[s_key[t]].x);
	// local index pairs, parked in the output array (contact order of the segment) until the table's memory is free for them
	for (uint32_t k = r0 + tid; k < r1; k += NH_BLK_PREP_THREADS) {
		const nh_BodyPair p = bodies[order[k]];
		uint32_t loc[2];
#pragma unroll
		for (int side = 0; side < 2; ++side) {
			const uint32_t x = side ? p.b : p.a;
			uint32_t l = 0;
			if (x) {
				const uint32_t br = brank[x].x;
				if ((br >> NH_BLK_RANK_BITS) == b) l = 1u + (br & NH_BLK_RANK_MASK);
				else {
					uint32_t h = (x * 2654435761u) >> (32 - 12);
					while (s_key[h] != x) h = (h + 1u) & (NH_BLK_HASH - 1u);
					l = 1u + n_own + s_val[h];
				}
			}
			loc[side] = l;
		}
		lpair[k] = loc[0] | loc[1] << 16;
	}
	__syncthreads();
	for (uint32_t t = tid; t < nloc; t += NH_BLK_PREP_THREADS) { s_mask[t] = 0ull; s_claim[t] = 0ull; }
	__syncthreads();
	// seeds: the colour of last step, offered to both bodies
	for (uint32_t kl = tid; kl < nc; kl += NH_BLK_PREP_THREADS) {
		const uint32_t lp = lpair[r0 + kl], la = lp & 0xFFFFu, lb = lp >> 16;
		s_lp[kl] = lp;
		uint32_t seed = __float_as_uint(warm[order[r0 + kl]].unused);
		if (seed < 1u || seed > LL) seed = 0u;
		s_col[kl] = (uint8_t)seed;
		if (seed) {
			const unsigned long long bit = 1ull << (seed - 1u);
			if (la) { const unsigned long long old = atomicOr(&s_mask[la], bit); if (old & bit) atomicOr(&s_claim[la], bit); }
			if (lb) { const unsigned long long old = atomicOr(&s_mask[lb], bit); if (old & bit) atomicOr(&s_claim[lb], bit); }
		}
	}
	__syncthreads();
	// a colour offered twice to one body: both contacts give it up (the colour stays marked as used on that body: one colour wasted, no order in the outcome);
	// whoever is left without a colour goes on the list
	for (uint32_t kl = tid; kl < nc; kl += NH_BLK_PREP_THREADS) {
		uint32_t seed = s_col[kl];
		if (seed) {
			const uint32_t lp = s_lp[kl], la = lp & 0xFFFFu, lb = lp >> 16;
			const unsigned long long bit = 1ull << (seed - 1u);
			if ((la && (s_claim[la] & bit)) || (lb && (s_claim[lb] & bit))) { seed = 0u; s_col[kl] = 0; }
		}
		if (!seed) { const uint32_t at = atomicAdd(&s_nlist, 1u); if (at < LIST) s_list[at] = (uint16_t)kl; }
	}
	__syncthreads();
	const uint32_t nlist = s_nlist;
	const bool listed = nlist <= LIST;                // (else: a block without seeds -- the first step -- walks all its contacts in every round)
	const uint32_t nwalk = listed ? nlist : nc;
	// Jones-Plassmann rounds over the uncoloured
	for (uint32_t round = 0; round < 4096u && nwalk; ++round) {
		for (uint32_t i = tid; i < nwalk; i += NH_BLK_PREP_THREADS) {
			const uint32_t kl = listed ? (uint32_t)s_list[i] : i;
			if (s_col[kl]) continue;
			const uint32_t lp = s_lp[kl], la = lp & 0xFFFFu, lb = lp >> 16;
			if (la) s_claim[la] = ~0ull;
			if (lb) s_claim[lb] = ~0ull;
		}
		__syncthreads();
		for (uint32_t i = tid; i < nwalk; i += NH_BLK_PREP_THREADS) {
			const uint32_t kl = listed ? (uint32_t)s_list[i] : i;
			if (s_col[kl]) continue;
			const uint32_t lp = s_lp[kl], la = lp & 0xFFFFu, lb = lp >> 16, c = order[r0 + kl];
			const unsigned long long bid = (unsigned long long)(c * 2654435761u) << 32 | c;
			if (la) atomicMin(&s_claim[la], bid);
			if (lb) atomicMin(&s_claim[lb], bid);
		}
		__syncthreads();
		bool left = false;
		for (uint32_t i = tid; i < nwalk; i += NH_BLK_PREP_THREADS) {
			const uint32_t kl = listed ? (uint32_t)s_list[i] : i;
			if (s_col[kl]) continue;
			const uint32_t lp = s_lp[kl], la = lp & 0xFFFFu, lb = lp >> 16, c = order[r0 + kl];
			const unsigned long long bid = (unsigned long long)(c * 2654435761u) << 32 | c;
			if ((!la || s_claim[la] == bid) && (!lb || s_claim[lb] == bid)) {
				const unsigned long long used = (la ? s_mask[la] : 0ull) | (lb ? s_mask[lb] : 0ull);
				const uint32_t q = (uint32_t)__ffsll((long long)~used) - 1u;           // (the LOWEST free colour: a free colour picked by hash, to even the classes out, made 22 colours of 15 and four times the clashes among the seeds)
				if (q >= LL) { s_fail = 1; s_col[kl] = 1; }                  // (more colours than the descriptor holds: smaller blocks do not help, the caller falls back)
				else { s_col[kl] = (uint8_t)(q + 1u); if (la) s_mask[la] |= 1ull << q; if (lb) s_mask[lb] |= 1ull << q; }
			} else left = true;
		}
		if (!__syncthreads_or(left ? 1 : 0)) { if (tid == 0) { atomicMax(&st->blk.max_rounds, round + 1u); atomicMax(&st->blk.max_listed, nlist); } break; }
		if (round == 4095u && tid == 0) atomicMax(&st->blk.max_rounds, 4096u);
	}
	// counting sort by colour inside the segment; chunks of the sweep's pipeline
	if (tid <= LL) s_cnt[tid] = 0;
	for (uint32_t kl = tid; kl < nc; kl += NH_BLK_PREP_THREADS) if (!s_col[kl]) { s_fail = 1; s_col[kl] = 1; }          // (the round limit: never in practice)
	__syncthreads();
	for (uint32_t kl = tid; kl < nc; kl += NH_BLK_PREP_THREADS) atomicAdd(&s_cnt[s_col[kl] - 1u], 1u);
	__syncthreads();
	if (tid == 0) {
		uint32_t run = 0, chunks = 0, used = 0;
		for (uint32_t c = 0; c < LL; ++c) { const uint32_t n = s_cnt[c]; s_cnt[c] = run; s_cur[c] = 0; run += n; chunks += (n + NH_BLK_CHUNK - 1u) / NH_BLK_CHUNK; if (n) used = c + 1u; }
		s_cnt[LL] = run;
		atomicMax(&st->blk.max_colours, used);
		if (s_fail) st->blk.too_many_colours = 1;
		if (chunks > NH_BLK_MAX_CHUNKS) { st->blk.overflow = 1; s_fail = 1; }
		if (!s_fail) {
			const uint32_t b0 = st->blk.bits[0], b1 = st->blk.bits[1], b2 = st->blk.bits[2];
			const uint32_t c0 = b & ((1u << b0) - 1u), c1 = (b >> b0) & ((1u << b1) - 1u), c2 = (b >> (b0 + b1)) & ((1u << b2) - 1u);
			const uint32_t col = (c0 & 1u) | (c1 & 1u) << 1 | (c2 & 1u) << 2;
			const uint32_t slot = col * NH_BLK_MAX_BLOCKS + atomicAdd(&st->blk.phase_count[col], 1u);
			phase_list[slot] = b;
			s_slot = slot;
			uint32_t* dsc = desc + (size_t)slot * NH_BLK_DESC_WORDS;
			dsc[0] = b; dsc[1] = n_own; dsc[2] = 1u + own_base[b]; dsc[3] = n_gh; dsc[4] = base;
		}
	}
	__syncthreads();
	if (s_fail) return;
	if (tid <= LL) desc[(size_t)s_slot * NH_BLK_DESC_WORDS + 5u + tid] = r0 + s_cnt[tid];
	for (uint32_t kl = tid; kl < nc; kl += NH_BLK_PREP_THREADS) {
		const uint32_t q = s_col[kl] - 1u, c = order[r0 + kl];
		const uint32_t pos = r0 + s_cnt[q] + atomicAdd(&s_cur[q], 1u);
		order_out[pos] = c;
		lpair[pos] = s_lp[kl];
		colour_out[c] = q + 1u;                        // (next step's seed: k_update_impulses puts it into the cached impulse's spare word)
	}
}

// block-order slot -> body (the inverse of blk_slot): what a kernel that works block by block needs to find its bodies' records
__global__ __launch_bounds__(256) void k_blk_slots(uint32_t nbodies, const uint2* __restrict__ brank, const uint32_t* __restrict__ own_base, uint32_t* __restrict__ pos_body) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		if (i == 0u) { pos_body[0] = 0u; continue; }
		const uint32_t br = brank[i].x;
		if (br != NH_NONE) pos_body[blk_slot(own_base, br)] = i;
	}
}

// Rows of the contacts the blocks own, BLOCK BY BLOCK (per-block colouring only: every general contact is then owned).  k_rows_general walks the block-ordered
// contact list with one thread per contact and gathers, per contact, the two bodies' positions and inertias from the caller's arrays: in a pit of 11 M
// contacts it fetched 2.5 GB (PMC, raw) for 0.9 GB of distinct data.  Here one workgroup per block stages the block's bodies ONCE in LDS (position, inverse
// mass | inverse inertia, body index: the same 32 bytes per body the sweep keeps of the momentum) and its contacts read them from there; what is left in
// global memory per contact is its 32-byte record, its cached impulse and its place.  Same functions, same operands: the rows are the very bits
// k_rows_general writes (NH_BLK_ROWS_GLOBAL=1 keeps that kernel: the A/B of the tests).
__global__ __launch_bounds__(NH_BLK_THREADS) void k_blk_rows(const nh_DevState* __restrict__ st, nh_BlkPhases ph, const uint32_t* __restrict__ desc_all, const uint32_t* __restrict__ ghost_list,
                                                             const uint32_t* __restrict__ pos_body, const uint32_t* __restrict__ order, const nh_Contact* __restrict__ contacts,
                                                             const nh_Transform* __restrict__ xf, const nh_BodyProperties* __restrict__ props, float4* __restrict__ rows4, size_t stride,
                                                             uint2* __restrict__ gpair, uint32_t* __restrict__ level_has_full, const nh_CachedContactImpulse* __restrict__ warm_in,
                                                             float* __restrict__ pstates, uint32_t* __restrict__ lpair) {
	__shared__ float4 s_p[NH_BLK_CAP], s_i[NH_BLK_CAP];          // (position, inverse mass), (inverse inertia, body index)
	__shared__ uint32_t s_desc[NH_BLK_DESC_WORDS];
	const uint32_t tid = threadIdx.x;
	uint32_t p = 0;
	while (p < 7u && blockIdx.x >= ph.start[p + 1u]) ++p;
	const uint32_t* dsc = desc_all + ((size_t)p * NH_BLK_MAX_BLOCKS + (blockIdx.x - ph.start[p])) * NH_BLK_DESC_WORDS;
	if (tid < NH_BLK_DESC_WORDS) s_desc[tid] = dsc[tid];
	__syncthreads();
	const uint32_t n_own = s_desc[1], base = s_desc[2], n_gh = s_desc[3], gb = s_desc[4], nloc = 1u + n_own + n_gh;
	const uint32_t k0 = s_desc[5], k1 = s_desc[5u + NH_BLK_LOCAL_L];
	const bool static_inert = st->static_inert != 0u;
	for (uint32_t j = tid; j < nloc; j += NH_BLK_THREADS) {
		uint32_t body = 0;
		if (j) body = pos_body[j <= n_own ? base + (j - 1u) : ghost_list[gb + (j - 1u - n_own)]];
		const float4 t0 = reinterpret_cast<const float4*>(xf + body)[0];
		const float4 pr = *reinterpret_cast<const float4*>(props + body);
		s_p[j] = make_float4(t0.x, t0.y, t0.z, pr.w);
		s_i[j] = make_float4(pr.x, pr.y, pr.z, __uint_as_float(body));
	}
	__syncthreads();
	for (uint32_t k = k0 + tid; k < k1; k += NH_BLK_THREADS) {
		const uint32_t c = order[k];
		const nh_Contact ct = contacts[c];
		const uint32_t lp = lpair[k], la = lp & 0xFFFu, lb = (lp >> 16) & 0xFFFu;
		const float4 a_p = s_p[la], a_i = s_i[la], b_p = s_p[lb], b_i = s_i[lb];
		const uint32_t body_a = __float_as_uint(a_i.w), body_b = __float_as_uint(b_i.w);
		nh_contact_in ci = { ct.position[0], ct.position[1], ct.position[2], ct.penetration, ct.normal[0], ct.normal[1], ct.normal[2], ct.friction };
		const nh_f3 apos = nh_make3(a_p.x, a_p.y, a_p.z), bpos = nh_make3(b_p.x, b_p.y, b_p.z);
		const bool ia = static_inert && body_a == 0u, ib = static_inert && body_b == 0u;
		const bool radial = nh_side_is_radial(ci.px - apos.x, ci.py - apos.y, ci.pz - apos.z, ci.nx, ci.ny, ci.nz, a_i.x, a_i.y, a_i.z, ia) &&
		                    nh_side_is_radial(ci.px - bpos.x, ci.py - bpos.y, ci.pz - bpos.z, ci.nx, ci.ny, ci.nz, b_i.x, b_i.y, b_i.z, ib);
		if (radial) {
			nh_rowr r;
			nh_build_rowr(ci, apos, bpos, a_i.x, b_i.x, a_p.w, b_p.w, ia, ib, r);
			store_rowr_soa(rows4, stride, k, r);
		} else {
			nh_inertia IA = body_inertia(xf, props, body_a), IB = body_inertia(xf, props, body_b);          // (boxes: the rotation too -- from the caller's arrays)
			nh_row r;
			nh_build_row(ci, apos, bpos, IA, IB, a_p.w, b_p.w, r);
			store_row_soa(rows4, stride, k, r);
			level_has_full[1] = 1u;
		}
		gpair[k] = make_uint2(body_a | (radial ? NH_RADIAL_BIT : 0u), body_b);
		const float4 w = *reinterpret_cast<const float4*>(warm_in + c);
		float* ps = pstates + 3 * (size_t)k;
		ps[0] = w.x; ps[1] = w.y; ps[2] = w.z;
		if (radial) lpair[k] = lp | NH_BLK_RADIAL;
	}
}

// residual contacts are solved by the global per-colour launches on the block-ordered momentum copy: their pairs become block-order slots
__global__ __launch_bounds__(256) void k_blk_residual_pairs(uint32_t begin, uint32_t end, const uint2* __restrict__ brank, const uint32_t* __restrict__ own_base, uint2* __restrict__ gpair) {
	for (uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x; k < end; k += gridDim.x * blockDim.x) {
		const uint2 p = gpair[k];
		const uint32_t a = p.x & ~NH_RADIAL_BIT;
		gpair[k] = make_uint2((a ? blk_slot(own_base, brank[a].x) : 0u) | (p.x & NH_RADIAL_BIT), p.y ? blk_slot(own_base, brank[p.y].x) : 0u);
	}
}

// momentum of the general bodies, caller's array <-> block order.  The copy's records are (velocity, inverse mass | angular velocity, k): the inverse
// mass rides in the first spare word like in the reference (nudge.cpp:4198; k_adj_sort put it there), the second spare word carries the body's
// inverse inertia about x -- the scalar k of a body with isotropic inertia, which is all a radial row ever asks of it (nh_rowr_derive).
__global__ __launch_bounds__(256) void k_blk_gather(uint32_t nbodies, const uint2* __restrict__ brank, const uint32_t* __restrict__ own_base, const float4* __restrict__ momentum4,
                                                    const nh_BodyProperties* __restrict__ props, float4* __restrict__ bm4, uint32_t* __restrict__ pos_body) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		uint32_t slot = 0;
		if (i) { const uint32_t br = brank[i].x; if (br == NH_NONE) continue; slot = blk_slot(own_base, br); }
		float4 m1 = momentum4[2 * (size_t)i + 1];
		m1.w = props[i].inertia_inverse[0];
		bm4[2 * (size_t)slot] = momentum4[2 * (size_t)i]; bm4[2 * (size_t)slot + 1] = m1;
		if (pos_body) pos_body[slot] = i;
	}
}

__global__ __launch_bounds__(256) void k_blk_scatter_back(uint32_t nbodies, const uint2* __restrict__ brank, const uint32_t* __restrict__ own_base, const float4* __restrict__ bm4,
                                                          float4* __restrict__ momentum4) {
	for (uint32_t i = 1u + blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		const uint32_t br = brank[i].x;
		if (br == NH_NONE) continue;
		const uint32_t slot = blk_slot(own_base, br);
		float4 m1 = bm4[2 * (size_t)slot + 1];
		m1.w = 0.0f;                                   // (what k_level_exec leaves in the spare word of a body it has solved)
		momentum4[2 * (size_t)i] = bm4[2 * (size_t)slot]; momentum4[2 * (size_t)i + 1] = m1;
	}
}

// ---- the sweep: one workgroup per block of the launch's block colour ---------------------------------------------------------------------------------
// WARM: the warm start (nudge.cpp:4563-4632) instead of a sweep; the cached impulse of a contact sits where its solver state will (k_rows_general put it
// there), so both kinds of launch stream the same arrays.  RADIAL_ONLY: no contact of the set has a full row (the host knows from k_rows_general's
// flags).
//
// What a contact costs per sweep: the first eight floats of its radial row (n, bias, alpha, beta, friction, effective mass; the other three follow from the
// two bodies, nh_rowr_derive), its local index pair, three state floats in and out -- 60 bytes, streamed in block order.  A colour step by itself is a
// dependent chain (LDS read, ~150 instructions, LDS write, barrier) with a few hundred contacts in it, far too little to keep HBM busy; what keeps it busy
// is how far AHEAD the stream is fetched.  The workgroup's wavefronts are split into NG groups that take the colours in turn (group g: colours g, g + NG,
// ...; a block has a few hundred contacts per colour, so a group's threads cover one), every thread keeps three contacts in flight in three register sets
// that take turns (no copies: a copy would have to wait for the loads it moves), and so the loads of the next 2 NG colours -- about half of a block's
// contacts -- are under way while a colour is being solved.  All loads go out unconditionally at clamped positions: a load under a divergent branch is
// issued and awaited inside that branch.
// chained sweeps: task t = the t-th (block colour, block) in launch order.  Which task is a block's, and which tasks of EARLIER colours touch bodies it touches
// (any of its 26 neighbours in the grid that has work: a block borrows ghosts only from neighbours, and only neighbours borrow from it)
__global__ __launch_bounds__(256) void k_blk_chain_slots(nh_BlkPhases ph, const uint32_t* __restrict__ phase_list, uint32_t* __restrict__ slot_of_block) {
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= ph.start[8]) return;
	int p = 0;
	while (p < 7 && t >= ph.start[p + 1]) ++p;
	slot_of_block[phase_list[(size_t)p * NH_BLK_MAX_BLOCKS + (t - ph.start[p])]] = t;
}
__global__ __launch_bounds__(256) void k_blk_chain_nbr(const nh_DevState* __restrict__ st, nh_BlkPhases ph, const uint32_t* __restrict__ phase_list, const uint32_t* __restrict__ slot_of_block,
                                                       uint32_t* __restrict__ nbr, uint32_t* __restrict__ done) {
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= ph.start[8]) return;
	int p = 0;
	while (p < 7 && t >= ph.start[p + 1]) ++p;
	const uint32_t b = phase_list[(size_t)p * NH_BLK_MAX_BLOCKS + (t - ph.start[p])];
	const uint32_t b0 = st->blk.bits[0], b1 = st->blk.bits[1], b2 = st->blk.bits[2];
	const int d0 = 1 << b0, d1 = 1 << b1, d2 = 1 << b2;
	const int c0 = (int)(b & (uint32_t)(d0 - 1)), c1 = (int)((b >> b0) & (uint32_t)(d1 - 1)), c2 = (int)((b >> (b0 + b1)) & (uint32_t)(d2 - 1));
	uint32_t n = 0;
	for (int z = -1; z <= 1; ++z) for (int y = -1; y <= 1; ++y) for (int x = -1; x <= 1; ++x) {
		if (!x && !y && !z) continue;
		const int q0 = c0 + x, q1 = c1 + y, q2 = c2 + z;
		if (q0 < 0 || q0 >= d0 || q1 < 0 || q1 >= d1 || q2 < 0 || q2 >= d2) continue;
		const uint32_t s = slot_of_block[(uint32_t)q0 | (uint32_t)q1 << b0 | (uint32_t)q2 << (b0 + b1)];
		if (s != NH_NONE && s < ph.start[p]) nbr[(size_t)t * NH_BLK_NBR_WORDS + 1u + n++] = s;          // (an earlier colour: its task of this pass comes first)
	}
	nbr[(size_t)t * NH_BLK_NBR_WORDS] = n;
	done[t] = 0u;
}

struct nh_blk_item { float4 r0, r1; float sx, sy, sz; uint32_t lp; };      // (position and validity follow from the colour: not kept)
template<bool WARM, bool RADIAL_ONLY, bool CHAIN = false>
__global__ __launch_bounds__(NH_BLK_THREADS) __attribute__((amdgpu_waves_per_eu(RADIAL_ONLY ? 4 : 2, RADIAL_ONLY ? 4 : 2))) void k_blk_sweep(const uint32_t* __restrict__ desc, uint32_t L,
                                                              const uint32_t* __restrict__ ghost_list, const uint32_t* __restrict__ lpair, const float4* __restrict__ rows4, size_t stride,
                                                              float* __restrict__ pstates, float4* bm4, unsigned long long* __restrict__ prof, nh_BlkChain chain = nh_BlkChain()) {
	// prof (NH_BLK_PROFILE=1): per workgroup four readings of the 100 MHz wall clock -- start, bodies in, colours done, momentum out
	constexpr uint32_t NG = NH_BLK_GROUPS, T = NH_BLK_THREADS, TG = NH_BLK_CHUNK;
	const unsigned long long t_start = prof ? wall_clock64() : 0ull;
	__shared__ float4 s_m0[NH_BLK_CAP], s_m1[NH_BLK_CAP];
	__shared__ uint32_t s_desc[NH_BLK_DESC_WORDS];
	__shared__ uint32_t s_first[64];                       // first chunk of every colour
	__shared__ uint32_t s_cs[NH_BLK_MAX_CHUNKS + 1];       // chunk starts (positions in the block-ordered contact arrays); chunk m = [s_cs[m], s_cs[m + 1])
	const uint32_t tid = threadIdx.x, g = tid / TG, t = tid % TG;
	// CHAIN: all eight block colours in one launch.  The workgroup's task is the next ticket (tickets go out in the order workgroups START, so every task this one
	// may have to wait for below is running or done: no deadlock whatever the residency); `desc` = the descriptors of all colours
	uint32_t task = blockIdx.x;
	size_t pidx = blockIdx.x;          // (where the profile readings of this workgroup go)
	if (CHAIN) {
		__shared__ uint32_t s_task;
		if (tid == 0) s_task = atomicAdd(chain.ticket, 1u);
		__syncthreads();
		task = s_task;
		int p = 0;
		while (p < 7 && task >= chain.ph.start[p + 1]) ++p;
		desc += (size_t)p * NH_BLK_MAX_BLOCKS * NH_BLK_DESC_WORDS + (size_t)(task - chain.ph.start[p]) * NH_BLK_DESC_WORDS;
		pidx = (size_t)p * NH_BLK_MAX_BLOCKS + (task - chain.ph.start[p]);
	} else desc += (size_t)blockIdx.x * NH_BLK_DESC_WORDS;
	if (prof && threadIdx.x == 0) prof[4 * pidx] = t_start;
	// one load brings everything the workgroup needs to know about its block (k_blk_prepare wrote it where the task finds it)
	// (CHAIN: and, in the same round trip, the list of tasks it has to wait for)
	__shared__ uint32_t s_nbr[NH_BLK_NBR_WORDS];
	if (tid < NH_BLK_DESC_WORDS) s_desc[tid] = desc[tid];
	if (CHAIN && tid >= 64u && tid < 64u + NH_BLK_NBR_WORDS) s_nbr[tid - 64u] = chain.nbr[(size_t)task * NH_BLK_NBR_WORDS + (tid - 64u)];
	__syncthreads();
	const uint32_t n_own = s_desc[1], base = s_desc[2], n_gh = s_desc[3], gb = s_desc[4];
	const uint32_t* s_off = s_desc + 5;
	// A colour is solved in chunks of at most TG contacts, one chunk per step (the greedy colouring fills its low colours first: a block's first colours
	// hold several hundred contacts, its last ones a handful).  Chunks of one colour need no barrier between them, but as steps of ONE pipeline they
	// are all prefetched alike.  Chunk table: counts per colour, a wave scan, the starts.
	if (tid < 64u) {
		const uint32_t n = tid < L ? (s_off[tid + 1u] - s_off[tid] + TG - 1u) / TG : 0u;
		uint32_t x = n;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d); if ((int)tid >= d) x += y; }
		s_first[tid] = x - n;                              // (s_first[L] = number of chunks)
	}
	__syncthreads();
	const uint32_t nch = min(s_first[L], NH_BLK_MAX_CHUNKS);
	if (tid < L) { const uint32_t f = s_first[tid], n = s_first[tid + 1u] - f; for (uint32_t q = 0; q < n && f + q < NH_BLK_MAX_CHUNKS; ++q) s_cs[f + q] = s_off[tid] + q * TG; }
	if (tid == L) s_cs[nch] = s_off[L];
	__syncthreads();
	const uint32_t last = s_off[L] > s_off[0] ? s_off[L] - 1u : s_off[0];

	auto load_item = [&](uint32_t k, nh_blk_item& it) {
		it.lp = lpair[k];
		it.r0 = rows4[k];
		it.r1 = rows4[stride + k];
		const float* ps = pstates + 3 * (size_t)k;
		it.sx = ps[0]; it.sy = ps[1]; it.sz = ps[2];
	};
	// this thread's contact in its group's j-th chunk
	auto fetch = [&](uint32_t j, nh_blk_item& it) {
		const uint32_t m = j * NG + g;
		const uint32_t mc = min(m, nch - 1u);
		const uint32_t k = s_cs[mc] + t;
		load_item(m < nch && k < s_cs[mc + 1u] ? k : last, it);
	};
	auto process = [&](const nh_blk_item& it, uint32_t k) {
		const uint32_t la = it.lp & 0xFFFu, lb = (it.lp >> 16) & 0xFFFu;
		const float4 a0 = s_m0[la], a1 = s_m1[la], b0 = s_m0[lb], b1 = s_m1[lb];
		nh_vel a = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w }, bb = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
		nh_state s;
		if (!WARM) { s.normal = it.sx; s.friction_x = it.sy; s.friction_y = it.sz; }
		if (RADIAL_ONLY || (it.lp & NH_BLK_RADIAL)) {
			nh_rowr r;
			r.n_x = it.r0.x; r.n_y = it.r0.y; r.n_z = it.r0.z; r.bias = it.r0.w;
			r.alpha = it.r1.x; r.beta = it.r1.y; r.friction = it.r1.z; r.normal_velocity_to_normal_impulse = it.r1.w; r.pad = 0.0f;
			nh_rowr_derive(r.alpha, r.beta, a0.w, b0.w, a1.w, b1.w, r.friction_coefficient, r.c_a, r.c_b);
			if (WARM) nh_warm_start_r(r, it.sx, it.sy, it.sz, a, bb, s); else nh_apply_radial(r, s, a, bb);
		} else if (!RADIAL_ONLY) {
			// a full row (boxes): chunks 2..9 fetched in place -- scenes of boxes are not what this kernel is tuned for
			nh_row r;
			float4* dst = reinterpret_cast<float4*>(&r);
			dst[0] = it.r0; dst[1] = it.r1;
#pragma unroll
			for (int q = 2; q < 10; ++q) dst[q] = rows4[(size_t)q * stride + k];
			if (WARM) nh_warm_start(r, it.sx, it.sy, it.sz, a, bb, s); else nh_apply_one(r, s, a, bb);
		}
		float* ps = pstates + 3 * (size_t)k;
		ps[0] = s.normal; ps[1] = s.friction_x; ps[2] = s.friction_y;
		// slot 0 is the static world: never written.  The spare words (inverse mass, k) go back as they came.
		if (la) { s_m0[la] = make_float4(a.vx, a.vy, a.vz, a0.w); s_m1[la] = make_float4(a.wx, a.wy, a.wz, a1.w); }
		if (lb) { s_m0[lb] = make_float4(bb.vx, bb.vy, bb.vz, b0.w); s_m1[lb] = make_float4(bb.wx, bb.wy, bb.wz, b1.w); }
	};

	// Prologue: the block's bodies come in by LDS-direct loads (global_load_lds_dwordx4: a wave moves 64 consecutive 16-byte pieces from per-lane
	// addresses to 1 KB of LDS, no staging registers, nothing to wait for until the barrier) -- the own bodies are contiguous in block order, a ghost's
	// slot comes from the ghost list first (the first two per thread are asked for before anything else).  Two dependent round trips in all.
	constexpr uint32_t UG = 2;
	uint32_t gs[UG];
#pragma unroll
	for (uint32_t u = 0; u < UG; ++u) gs[u] = ghost_list[gb + min(tid + u * T, n_gh ? n_gh - 1u : 0u)];
	nh_blk_item A, B, C;
	fetch(0u, A);
	fetch(1u, B);
	if (CHAIN) {
		// wait for the tasks of earlier colours that touch this block's bodies (k_blk_chain_nbr): they publish their momentum with system-scope stores and then
		// their pass tag; the spin is bounded -- a tag that never comes is an error, not a hang
		if (tid < 64u) {
			const uint32_t cnt = s_nbr[0];
			if (tid < cnt) {
				const uint32_t other = s_nbr[1u + tid];
				uint32_t spins = 0;
				while (__hip_atomic_load(&chain.done[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != chain.tag) {
					__builtin_amdgcn_s_sleep(8);
					if (++spins > 4000000u) { *chain.error = NH_ERR_INVALID; break; }
				}
			}
		}
		__syncthreads();
	}
	auto body_in = [&](size_t slot, uint32_t at) {
		if (CHAIN) { nh_glds16_sys(bm4 + 2 * slot, &s_m0[at]); nh_glds16_sys(bm4 + 2 * slot + 1, &s_m1[at]); }
		else { nh_glds16(bm4 + 2 * slot, &s_m0[at]); nh_glds16(bm4 + 2 * slot + 1, &s_m1[at]); }
	};
	if (tid == 0) { nh_glds16(bm4, &s_m0[0]); nh_glds16(bm4 + 1, &s_m1[0]); }
	for (uint32_t i = tid; i < n_own; i += T) body_in((size_t)(base + i), 1u + i);
#pragma unroll
	for (uint32_t u = 0; u < UG; ++u) {
		const uint32_t i = tid + u * T;
		if (i < n_gh) body_in((size_t)gs[u], 1u + n_own + i);
	}
	for (uint32_t i = tid + UG * T; i < n_gh; i += T) body_in((size_t)ghost_list[gb + i], 1u + n_own + i);
	__syncthreads();
	if (prof && threadIdx.x == 0) prof[4 * pidx + 1] = wall_clock64();
	uint32_t j = 0;
#define NH_BLK_STEP(CUR, FREE)                                                                                                                           \
	{                                                                                                                                                    \
		fetch(j + 2u, FREE);                                                                                                                             \
		_Pragma("unroll") for (uint32_t gg = 0; gg < NG; ++gg) {                                                                                         \
			const uint32_t m = j * NG + gg;                                                                                                              \
			if (m < nch) {                                   /* (uniform over the workgroup) */                                                          \
				if (g == gg) {                               /* (uniform over a wavefront)  */                                                           \
					const uint32_t k = s_cs[m] + t;                                                                                                      \
					if (k < s_cs[m + 1u]) process(CUR, k);                                                                                               \
				}                                                                                                                                        \
				__syncthreads();                                                                                                                         \
			}                                                                                                                                            \
		}                                                                                                                                                \
		if (++j * NG >= nch) break;                                                                                                                      \
	}
	for (;;) { NH_BLK_STEP(A, C) NH_BLK_STEP(B, A) NH_BLK_STEP(C, B) }
#undef NH_BLK_STEP
	if (prof && threadIdx.x == 0) prof[4 * pidx + 2] = wall_clock64();
	// momentum out; the first ghost slots are still in registers
	auto body_out = [&](size_t slot, uint32_t at) {
		if (CHAIN) { nh_store16_sys(bm4 + 2 * slot, s_m0[at]); nh_store16_sys(bm4 + 2 * slot + 1, s_m1[at]); }
		else { bm4[2 * slot] = s_m0[at]; bm4[2 * slot + 1] = s_m1[at]; }
	};
	for (uint32_t i = tid; i < n_own; i += T) body_out((size_t)(base + i), 1u + i);
#pragma unroll
	for (uint32_t u = 0; u < UG; ++u) { const uint32_t i = tid + u * T; if (i < n_gh) body_out((size_t)gs[u], 1u + n_own + i); }
	for (uint32_t i = tid + UG * T; i < n_gh; i += T) body_out((size_t)ghost_list[gb + i], 1u + n_own + i);
	if (CHAIN) {
		// published: every store of the workgroup has left, then the tag (system scope, like the stores)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (tid == 0) __hip_atomic_store(&chain.done[task], chain.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (prof && threadIdx.x == 0) prof[4 * pidx + 3] = wall_clock64();
}

// ---- NH_BLK_CHECK=1: structural check of one block colour's launch -------------------------------------------------------------------------------------
// every local index of every contact resolves to the body the contact names; no block-order slot is touched by two blocks of the launch; a
// block's list holds no two contacts of one colour that share a dynamic body
__global__ __launch_bounds__(256) void k_blk_check(nh_DevState* __restrict__ st, const uint32_t* __restrict__ phase_list, uint32_t L, const uint32_t* __restrict__ bin_off,
                                                   const uint32_t* __restrict__ blk_count, const uint32_t* __restrict__ own_base, const uint32_t* __restrict__ ghost_base,
                                                   const uint32_t* __restrict__ ghost_count, const uint32_t* __restrict__ ghost_list, const uint32_t* __restrict__ lpair,
                                                   const uint32_t* __restrict__ order, const nh_BodyPair* __restrict__ bodies, const uint32_t* __restrict__ level,
                                                   const uint32_t* __restrict__ pos_body, uint32_t* __restrict__ touch, uint32_t* __restrict__ colour_of_slot /* 2 x slots: scratch */,
                                                   const uint32_t* __restrict__ desc_phase) {
	// (desc_phase: the blocks coloured their own contacts, k_blk_prepare_local -- the colour offsets are the descriptor's, there is no world-wide colour)
	const uint32_t b = phase_list[blockIdx.x];
	const uint32_t* offs = desc_phase ? desc_phase + (size_t)blockIdx.x * NH_BLK_DESC_WORDS + 5u : bin_off + (size_t)b * L;
	const uint32_t n_own = blk_count[b], base = 1u + own_base[b], n_gh = ghost_count[b], gb = ghost_base[b];
	uint32_t errors = 0;
	if (1u + n_own + n_gh > NH_BLK_CAP) ++errors;
	for (uint32_t l = 0; l < L; ++l) {
		for (uint32_t k = offs[l] + threadIdx.x; k < offs[l + 1u]; k += blockDim.x) {
			const uint32_t c = order[k];
			const nh_BodyPair p = bodies[c];
			if (level && level[c] != l + 1u) ++errors;
			const uint32_t lp = lpair[k];
			const uint32_t loc[2] = { lp & 0xFFFu, (lp >> 16) & 0xFFFu };
			for (int side = 0; side < 2; ++side) {
				const uint32_t x = side ? p.b : p.a, lc = loc[side];
				uint32_t slot;
				if (lc == 0u) slot = 0u;
				else if (lc <= n_own) slot = base + lc - 1u;
				else if (lc - 1u - n_own < n_gh) slot = ghost_list[gb + lc - 1u - n_own];
				else { ++errors; continue; }
				if (pos_body[slot] != x) ++errors;
				if (slot) {
					const uint32_t old = atomicExch(&touch[slot], b + 1u);
					if (old != 0u && old != b + 1u) ++errors;                       // another block of this launch works on the same body
					const uint32_t oldc = atomicExch(&colour_of_slot[slot], (l + 1u) | (k << 6));
					if (oldc != 0u && (oldc & 63u) == l + 1u && (oldc >> 6) != (k & 0x3FFFFFFu)) ++errors;      // two contacts of one colour on one body
				}
			}
		}
		__syncthreads();
	}
	if (errors) atomicAdd(&st->blk.check_errors, errors);
}

#endif
