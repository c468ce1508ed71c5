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

#define NH_BLK_CAP 2400u                 // momentum records in one workgroup's LDS: slot 0 = the static world, own bodies, ghosts (76.8 KB: two workgroups per CU)
#define NH_BLK_THREADS 512
#define NH_BLK_RANK_BITS 12u
#define NH_BLK_RANK_MASK ((1u << NH_BLK_RANK_BITS) - 1u)
#define NH_BLK_HASH 4096u                // ghost table of k_blk_prepare (open addressing in LDS)
#define NH_BLK_NEAR 500                  // of 1024 per block edge: "near" = no coordinate of the two centres differs by more (12 units of slack for the quantisation)
#define NH_BLK_RESIDUAL NH_BLK_MAX_BLOCKS

__device__ __forceinline__ uint32_t blk_slot(const uint32_t* __restrict__ own_base, uint32_t br) { return 1u + own_base[br >> NH_BLK_RANK_BITS] + (br & NH_BLK_RANK_MASK); }

__global__ void k_blk_reset(nh_DevState* st) {
	nh_BlkGrid& g = st->blk;
	for (int k = 0; k < 3; ++k) { g.bmin[k] = 0xffffffffu; g.bmax[k] = 0u; }
	g.bodies = 0; g.nblk = 0; g.overflow = 0; g.max_load = 0; g.ghost_cursor = 0; g.check_errors = 0; g.near_contacts = 0; g.far_contacts = 0;
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
			for (int k = 0; k < 3; ++k) {
				atomicMin(&st->blk.bmin[k], min(min(s_lo[k][0], s_lo[k][1]), min(s_lo[k][2], s_lo[k][3])));
				atomicMax(&st->blk.bmax[k], max(max(s_hi[k][0], s_hi[k][1]), max(s_hi[k][2], s_hi[k][3])));
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
	const uint32_t lane = nh_lane();
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
		uint32_t rank = 0;
		unsigned long long todo = __ballot(general);
		while (todo) {
			const int leader = __ffsll((long long)todo) - 1;
			const uint32_t lb = (uint32_t)__shfl((int)blk, leader);
			const unsigned long long same = __ballot(general && blk == lb) & todo;
			uint32_t base = 0;
			if ((int)lane == leader) base = atomicAdd(&blk_count[lb], (uint32_t)__popcll(same));
			base = (uint32_t)__shfl((int)base, leader);
			if ((same >> lane) & 1ull) rank = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
			todo &= ~same;
		}
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
		uint32_t l = level[c];
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
		rank_out[k] = atomicAdd(&bins[bin], 1u);
		bin_out[k] = bin;
	}
	for (int d = 32; d >= 1; d >>= 1) { near += (uint32_t)__shfl_xor((int)near, d); far += (uint32_t)__shfl_xor((int)far, d); }
	if (nh_lane() == 0u) { if (near) atomicAdd(&st->blk.near_contacts, near); if (far) atomicAdd(&st->blk.far_contacts, far); }
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
                                                     uint32_t* __restrict__ ghost_list, uint32_t* __restrict__ phase_list) {
	__shared__ uint32_t s_key[NH_BLK_HASH], s_val[NH_BLK_HASH];
	__shared__ uint32_t s_n, s_base, s_fail;
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
	const bool fits = !s_fail && 1u + n_own + n_gh <= NH_BLK_CAP;
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
			phase_list[col * NH_BLK_MAX_BLOCKS + atomicAdd(&st->blk.phase_count[col], 1u)] = b;
		}
	}
	__syncthreads();
	if (!fits) return;                                   // (the host sees the overflow flag and retries the whole setup with smaller blocks)
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

// residual contacts are solved by the global per-colour launches on the block-ordered momentum copy: their pairs become block-order slots
__global__ __launch_bounds__(256) void k_blk_residual_pairs(uint32_t begin, uint32_t end, const uint2* __restrict__ brank, const uint32_t* __restrict__ own_base, uint2* __restrict__ gpair) {
	for (uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x; k < end; k += gridDim.x * blockDim.x) {
		const uint2 p = gpair[k];
		const uint32_t a = p.x & ~NH_RADIAL_BIT;
		gpair[k] = make_uint2((a ? blk_slot(own_base, brank[a].x) : 0u) | (p.x & NH_RADIAL_BIT), p.y ? blk_slot(own_base, brank[p.y].x) : 0u);
	}
}

// momentum of the general bodies, caller's array <-> block order
__global__ __launch_bounds__(256) void k_blk_gather(uint32_t nbodies, const uint2* __restrict__ brank, const uint32_t* __restrict__ own_base, const float4* __restrict__ momentum4,
                                                    float4* __restrict__ bm4, uint32_t* __restrict__ pos_body) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		uint32_t slot = 0;
		if (i) { const uint32_t br = brank[i].x; if (br == NH_NONE) continue; slot = blk_slot(own_base, br); }
		bm4[2 * (size_t)slot] = momentum4[2 * (size_t)i]; bm4[2 * (size_t)slot + 1] = momentum4[2 * (size_t)i + 1];
		if (pos_body) pos_body[slot] = i;
	}
}

__global__ __launch_bounds__(256) void k_blk_scatter_back(uint32_t nbodies, const uint2* __restrict__ brank, const uint32_t* __restrict__ own_base, const float4* __restrict__ bm4,
                                                          float4* __restrict__ momentum4) {
	for (uint32_t i = 1u + blockIdx.x * blockDim.x + threadIdx.x; i < nbodies; i += gridDim.x * blockDim.x) {
		const uint32_t br = brank[i].x;
		if (br == NH_NONE) continue;
		const uint32_t slot = blk_slot(own_base, br);
		momentum4[2 * (size_t)i] = bm4[2 * (size_t)slot]; momentum4[2 * (size_t)i + 1] = bm4[2 * (size_t)slot + 1];
	}
}

// ---- the sweep: one workgroup per block of the launch's block colour ---------------------------------------------------------------------------------
// WARM: the warm start (nudge.cpp:4563-4632) instead of a sweep; the cached impulse of a contact sits where its solver state will (k_rows_general put it
// there), so both kinds of launch stream the same arrays.  RADIAL_ONLY: no contact of the set has a full row (the host knows from k_rows_general's
// flags): the kernel without the 40-float path runs at twice the occupancy.
template<int NQ> struct nh_blk_item { float4 row[NQ]; float4 sv; uint32_t lp, k; bool valid; };

template<bool WARM, bool RADIAL_ONLY>
__global__ __launch_bounds__(NH_BLK_THREADS) void k_blk_sweep(const uint32_t* __restrict__ phase_list, uint32_t L, const uint32_t* __restrict__ bin_off, const uint32_t* __restrict__ blk_count,
                                                              const uint32_t* __restrict__ own_base, const uint32_t* __restrict__ ghost_base, const uint32_t* __restrict__ ghost_count,
                                                              const uint32_t* __restrict__ ghost_list, const uint32_t* __restrict__ lpair, const float4* __restrict__ rows4, size_t stride,
                                                              float4* __restrict__ gstates, float4* bm4) {
	constexpr int NQ = RADIAL_ONLY ? 3 : 10;
	__shared__ float4 s_m0[NH_BLK_CAP], s_m1[NH_BLK_CAP];
	__shared__ uint32_t s_off[NH_BLK_MAX_COLOURS + 1];
	const uint32_t tid = threadIdx.x, T = NH_BLK_THREADS;
	const uint32_t b = phase_list[blockIdx.x];
	const uint32_t n_own = blk_count[b], base = 1u + own_base[b], n_gh = ghost_count[b], gb = ghost_base[b];
	if (tid <= L) s_off[tid] = bin_off[b * L + tid];
	__syncthreads();

	auto load_item = [&](uint32_t k, nh_blk_item<NQ>& it) {
		it.k = k;
		it.lp = lpair[k];
#pragma unroll
		for (int q = 0; q < 3; ++q) it.row[q] = rows4[(size_t)q * stride + k];
		if (!RADIAL_ONLY && !(it.lp & NH_BLK_RADIAL)) {
#pragma unroll
			for (int q = 3; q < NQ; ++q) it.row[q] = rows4[(size_t)q * stride + k];
		}
		it.sv = gstates[k];
	};
	auto fetch = [&](uint32_t l, nh_blk_item<NQ>& it) {
		const uint32_t k = s_off[l] + tid;
		it.valid = k < s_off[l + 1];
		if (it.valid) load_item(k, it);
	};
	auto process = [&](const nh_blk_item<NQ>& it) {
		const uint32_t la = it.lp & 0xFFFu, lb = (it.lp >> 16) & 0xFFFu;
		const float4 a0 = s_m0[la], a1 = s_m1[la], b0 = s_m0[lb], b1 = s_m1[lb];
		nh_vel a = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w }, bb = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
		nh_state s;
		if (!WARM) { s.normal = it.sv.x; s.friction_x = it.sv.y; s.friction_y = it.sv.z; }
		if (RADIAL_ONLY || (it.lp & NH_BLK_RADIAL)) {
			nh_rowr r;
			float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
			for (int q = 0; q < 3; ++q) dst[q] = it.row[q];
			if (WARM) nh_warm_start_r(r, it.sv.x, it.sv.y, it.sv.z, a, bb, s); else nh_apply_radial(r, s, a, bb);
		} else if (!RADIAL_ONLY) {
			nh_row r;
			float4* dst = reinterpret_cast<float4*>(&r);
#pragma unroll
			for (int q = 0; q < NQ; ++q) dst[q] = it.row[q];
			if (WARM) nh_warm_start(r, it.sv.x, it.sv.y, it.sv.z, a, bb, s); else nh_apply_one(r, s, a, bb);
		}
		gstates[it.k] = make_float4(s.normal, s.friction_x, s.friction_y, 0.0f);
		if (la) { s_m0[la] = make_float4(a.vx, a.vy, a.vz, a.mass_inverse); s_m1[la] = make_float4(a.wx, a.wy, a.wz, a.w_unused); }      // slot 0 is the static world: never written
		if (lb) { s_m0[lb] = make_float4(bb.vx, bb.vy, bb.vz, bb.mass_inverse); s_m1[lb] = make_float4(bb.wx, bb.wy, bb.wz, bb.w_unused); }
	};

	// the first colour's contacts are on their way while the bodies come in
	nh_blk_item<NQ> cur, nxt;
	fetch(0u, cur);
	if (tid == 0) { s_m0[0] = bm4[0]; s_m1[0] = bm4[1]; }
	for (uint32_t i = tid; i < n_own; i += T) { s_m0[1u + i] = bm4[2 * (size_t)(base + i)]; s_m1[1u + i] = bm4[2 * (size_t)(base + i) + 1]; }
	for (uint32_t i = tid; i < n_gh; i += T) { const size_t slot = ghost_list[gb + i]; s_m0[1u + n_own + i] = bm4[2 * slot]; s_m1[1u + n_own + i] = bm4[2 * slot + 1]; }
	__syncthreads();
	for (uint32_t l = 0; l < L; ++l) {
		const bool more = l + 1u < L;
		if (more) fetch(l + 1u, nxt);
		if (cur.valid) process(cur);
		for (uint32_t k = s_off[l] + tid + T; k < s_off[l + 1]; k += T) {            // a colour with more contacts than threads: the rest, fetched in place
			nh_blk_item<NQ> extra;
			load_item(k, extra);
			process(extra);
		}
		__syncthreads();
		if (more) cur = nxt;
	}
	for (uint32_t i = tid; i < n_own; i += T) { bm4[2 * (size_t)(base + i)] = s_m0[1u + i]; bm4[2 * (size_t)(base + i) + 1] = s_m1[1u + i]; }
	for (uint32_t i = tid; i < n_gh; i += T) { const size_t slot = ghost_list[gb + i]; bm4[2 * slot] = s_m0[1u + n_own + i]; bm4[2 * slot + 1] = s_m1[1u + n_own + i]; }
}

// ---- NH_BLK_CHECK=1: structural check of one block colour's launch -------------------------------------------------------------------------------------
// every local index of every contact resolves to the body the contact names; no block-order slot is touched by two blocks of the launch; a
// block's list holds no two contacts of one colour that share a dynamic body
__global__ __launch_bounds__(256) void k_blk_check(nh_DevState* __restrict__ st, const uint32_t* __restrict__ phase_list, uint32_t L, const uint32_t* __restrict__ bin_off,
                                                   const uint32_t* __restrict__ blk_count, const uint32_t* __restrict__ own_base, const uint32_t* __restrict__ ghost_base,
                                                   const uint32_t* __restrict__ ghost_count, const uint32_t* __restrict__ ghost_list, const uint32_t* __restrict__ lpair,
                                                   const uint32_t* __restrict__ order, const nh_BodyPair* __restrict__ bodies, const uint32_t* __restrict__ level,
                                                   const uint32_t* __restrict__ pos_body, uint32_t* __restrict__ touch, uint32_t* __restrict__ colour_of_slot /* 2 x slots: scratch */) {
	const uint32_t b = phase_list[blockIdx.x];
	const uint32_t n_own = blk_count[b], base = 1u + own_base[b], n_gh = ghost_count[b], gb = ghost_base[b];
	uint32_t errors = 0;
	if (1u + n_own + n_gh > NH_BLK_CAP) ++errors;
	for (uint32_t l = 0; l < L; ++l) {
		for (uint32_t k = bin_off[b * L + l] + threadIdx.x; k < bin_off[b * L + l + 1u]; k += blockDim.x) {
			const uint32_t c = order[k];
			const nh_BodyPair p = bodies[c];
			if (level[c] != l + 1u) ++errors;
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
