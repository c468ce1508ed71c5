// nh_partition.hip -- the multi-GPU path behind the C ABI (include/nudge_hip.h, "multi-GPU"): one x-slab of a world per context; migration, ghost
// listing, ghost installation + collider rebuild, per-step halo records and cut re-balancing as kernels on the context's stream with device-side counts.
// The reference has no counterpart (single-threaded); what makes a partitioned world reproduce the single one is that a contact's identity
// is its pair of collider TAGS (nudge.h:86, 93; nudge.cpp:2074-2087), which travel with the body.  Transport is the caller's (RCCL, torch.distributed).
//
// Order of things in a rank's arrays: body 0 (static world) | owned bodies, slots 1 .. n_owned | ghosts from the left neighbour | ghosts from the right one;
// colliders: the static ones the rank keeps | one per dynamic body, in slot order (boxes and spheres each in their own array).
#include "nh_internal.h"
#include "nh_math.h"
#include <math.h>
#include <string.h>

struct nh_PartDev {
	uint32_t zero;               // always 0 (a device-side count of nothing: nh_scan*_u32 take their length as count + extra)
	uint32_t count[2];           // records packed for the left / right neighbour by the last pack
	uint32_t kinds[2];           // boxes / spheres among the dynamic bodies (collider rebuild)
	uint32_t vmax_bits;          // largest |velocity| of an owned body, as the bits of a non-negative float (atomicMax)
	float kth[2];                // the two order statistics a new cut lies between
	uint32_t error;
};

// what travels with a body: its state and up to NH_PARTITION_MAX_COLLIDERS colliders (boxes first, then spheres), each as shape | local transform | tag
struct nh_PartCollider { float shape[4]; nh_Transform cxf; uint32_t tag; uint32_t pad; };      // 56 B
struct nh_FullRecord { nh_Transform xf; nh_BodyProperties props; nh_BodyMomentum momentum; nh_PartCollider col[NH_PARTITION_MAX_COLLIDERS]; uint8_t nbox, nsph, idle, pad; };
static_assert(sizeof(nh_PartCollider) == 56 && sizeof(nh_FullRecord) == NH_PARTITION_FULL_BYTES, "full record layout");
#define NP_MAXC NH_PARTITION_MAX_COLLIDERS

struct nh_partition {
	nh_context* ctx;
	nh_PartitionConfig cfg;
	double lo, hi, cut0[2];
	uint32_t n_owned, n_bodies, n_boxes, n_spheres;
	uint32_t ghost_out[2], ghost_in[2];
	uint64_t migrated_out, migrated_in, refreshes, cut_moves, quiet_refreshes;
	// body-parallel description of the colliders a dynamic body carries (device, body_capacity entries; NP_MAXC collider slots per body: its boxes first, then its spheres)
	float4* shape; nh_Transform* cxf; uint32_t* tag; uint8_t* nbx; uint8_t* nsp; float* reach;
	// lists of the owned slots sent to the left / right neighbour every step (written by pack_ghosts)
	uint32_t* list[2];
	// scratch: two flag / scan arrays of body_capacity + 1 words, scan temporaries, sort buffers, a staging copy of the body records for the compaction
	uint32_t* flag[2]; uint32_t* scan_tmp; uint32_t* sort_a; uint32_t* sort_b; uint32_t* sort_va; uint32_t* sort_vb; uint32_t* sort_hist;
	nh_FullRecord* stage;
	nh_PartDev* d; nh_PartDev* h;
	bool leavers_flagged;
	// optional transport for the PER-STEP halo (nh_partition_set_transport): an RCCL communicator and the five RCCL entry points, handed in by the host as plain
	// pointers -- the library neither links nor loads RCCL itself, it calls what the host's process has already loaded
	struct { void* comm; int (*group_start)(void); int (*group_end)(void);
	         int (*send)(const void*, size_t, int, int, void*, hipStream_t); int (*recv)(void*, size_t, int, int, void*, hipStream_t); int peer[2]; } rccl;
	void* step_out[2]; void* step_in[2]; uint32_t step_cap[2][2];      // message buffers of the per-step halo, [side][out / in], grown at refreshes
	// the ghost margin of a cut covers how far bodies on BOTH its sides travel in an epoch: this rank's top speed before migration (nh_partition_top_speed) and what
	// the two neighbours reported (nh_partition_set_peer_speeds); consumed by the next nh_partition_pack_ghosts
	double speed_pre, peer_speed[2];
	// single ownership of contacts that cross a cut (nh_set_first_ghost_body): the ghosts' momentum as it was at the last nh_partition_mark_ghosts (2 x float4 per ghost)
	float4* ghost_ref; uint32_t ghost_ref_capacity;
	// ... and WHICH ghost set that mark belongs to: nh_partition_unpack_ghosts starts a new generation, the mark adopts it, nh_partition_pack_deltas insists on it
	// (a refresh that changes the ghost set without outgrowing the buffer used to pass the capacity test and pack deltas against other bodies' references)
	uint64_t ghost_generation, ghost_ref_generation;
	// nh_partition_step: the arrays of the call in progress (the hook needs them), whether its first sub-step is preceded by a per-step exchange, loop-back records (one-rank rehearsal)
	const nh_StepArgs* step_args; uint32_t step_exchange_first, step_loopback;
	void* loop_out; void* loop_in; uint32_t* loop_idx; uint32_t loop_cap;
	// HALO SPLIT (round 6): the bodies a neighbour waits for -- listed for it, its ghosts here, the rehearsal's loop-back records -- are solved by a
	// launch of their own on `split_stream`; once the step's verdict is in, the NEXT sub-step's exchange is enqueued behind them and travels beside the interior's solver
	// (nh_internal.h: nh_FusedStep::part).  `exchanged_ahead`: the hook of the next sub-step finds its halo in place and only joins the streams
	hipStream_t split_stream; hipEvent_t split_done; uint32_t* split_list; uint32_t* split_count; uint8_t* split_mask; uint32_t split_list_cap, split_mask_cap;
	bool exchanged_ahead, no_overlap;
	uint64_t overlapped_exchanges;
};

#define NP_CHECK(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (ctx)->last_hip_error = (int)e_; return NH_ERR_HIP; } } while (0)

// the body-parallel collider description as the kernels see it: body i's colliders sit at [NP_MAXC i, NP_MAXC i + nbx[i] + nsp[i]): boxes, then spheres
struct nh_PartCols { float4* shape; nh_Transform* cxf; uint32_t* tag; uint8_t* nbx; uint8_t* nsp; float* reach; };

// ---- kernels ------------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float np_reach(const nh_Transform& c, const float4& s, bool box) {
	const float off = sqrtf(c.position[0] * c.position[0] + c.position[1] * c.position[1] + c.position[2] * c.position[2]);
	const float ext = box ? sqrtf(s.x * s.x + s.y * s.y + s.z * s.z) : s.x;
	return (off + ext) * 1.0001f;
}
__device__ __forceinline__ float np_body_reach(const nh_PartCols& pc, uint32_t b) {
	const uint32_t nb = pc.nbx[b], ns = pc.nsp[b];
	float r = 0.0f;
	for (uint32_t c = 0; c < nb + ns && c < NP_MAXC; ++c) r = fmaxf(r, np_reach(pc.cxf[(size_t)NP_MAXC * b + c], pc.shape[(size_t)NP_MAXC * b + c], c < nb));
	return r;
}

// creation: the collider arrays (static ones first, then the dynamic bodies' in body-slot order, a body's colliders adjacent, Transform.body = slot) -> the
// body-parallel description.  A collider's place among its body's = the run of colliders of that body just before it (boxes: from 0; spheres: behind the boxes).
__global__ __launch_bounds__(256) void k_part_adopt(uint32_t n_static, uint32_t n, uint32_t which, const uint32_t* __restrict__ tags, const void* __restrict__ data, const nh_Transform* __restrict__ xf,
                                                    uint32_t body_capacity, nh_PartCols pc, nh_PartDev* dv) {
	for (uint32_t c = n_static + blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const nh_Transform t = xf[c];
		const uint32_t b = t.body;
		if (b == 0u || b >= body_capacity) { dv->error = NH_ERR_INVALID; continue; }
		uint32_t rank = 0;
		while (rank < NP_MAXC && c - rank > n_static && xf[c - rank - 1u].body == b) ++rank;
		const bool last = c + 1u >= n || xf[c + 1u].body != b;
		const uint32_t at = (which == 0u ? 0u : (uint32_t)pc.nbx[b]) + rank;          // (spheres are adopted by a second launch: the box count is final by then)
		if (at >= NP_MAXC) { dv->error = NH_ERR_INVALID; continue; }
		float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (which == 0u) { const nh_BoxCollider bc = reinterpret_cast<const nh_BoxCollider*>(data)[c]; s = make_float4(bc.size[0], bc.size[1], bc.size[2], bc.unused); }
		else s.x = reinterpret_cast<const nh_SphereCollider*>(data)[c].radius;
		pc.shape[(size_t)NP_MAXC * b + at] = s; pc.cxf[(size_t)NP_MAXC * b + at] = t; pc.tag[(size_t)NP_MAXC * b + at] = tags[c];
		if (last) { if (which == 0u) pc.nbx[b] = (uint8_t)(rank + 1u); else pc.nsp[b] = (uint8_t)(rank + 1u); }
	}
}
__global__ __launch_bounds__(256) void k_part_reaches(uint32_t n, nh_PartCols pc) {
	for (uint32_t b = 1u + blockIdx.x * blockDim.x + threadIdx.x; b < n; b += gridDim.x * blockDim.x) pc.reach[b] = np_body_reach(pc, b);
}

// refresh phase 1: who leaves (x < lo: to the left; x >= hi: to the right).  flags are scanned in place afterwards
__global__ __launch_bounds__(256) void k_part_flag_leavers(uint32_t n, const nh_Transform* __restrict__ bt, float lo, float hi, uint32_t has_left, uint32_t has_right,
                                                           uint32_t* __restrict__ fl, uint32_t* __restrict__ fr) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
		uint32_t l = 0u, r = 0u;
		if (i >= 1u && i < n) {
			const float x = bt[i].position[0];
			l = (has_left && x < lo) ? 1u : 0u;
			r = (has_right && x >= hi) ? 1u : 0u;
		}
		fl[i] = l; fr[i] = r;           // (entry n: the sentinel that makes scan[i + 1] exist for every i)
	}
}

__device__ __forceinline__ nh_FullRecord np_full(uint32_t i, const nh_Transform* bt, const nh_BodyProperties* bp, const nh_BodyMomentum* bm, const uint8_t* bi, const nh_PartCols& pc) {
	nh_FullRecord r;
	r.xf = bt[i]; r.props = bp[i]; r.momentum = bm[i];
	r.nbox = pc.nbx[i]; r.nsph = pc.nsp[i]; r.idle = bi[i]; r.pad = 0;
	for (uint32_t c = 0; c < NP_MAXC; ++c) {
		nh_PartCollider k;
		memset(&k, 0, sizeof(k));
		if (c < (uint32_t)r.nbox + r.nsph) {
			const float4 s = pc.shape[(size_t)NP_MAXC * i + c];
			k.shape[0] = s.x; k.shape[1] = s.y; k.shape[2] = s.z; k.shape[3] = s.w;
			k.cxf = pc.cxf[(size_t)NP_MAXC * i + c]; k.tag = pc.tag[(size_t)NP_MAXC * i + c];
		}
		r.col[c] = k;
	}
	return r;
}

// ordered compaction of the flagged bodies' full records (sl / sr: the exclusive scans of the flags, n + 1 entries); optionally the slot lists too
__global__ __launch_bounds__(256) void k_part_pack_full(uint32_t n, const uint32_t* __restrict__ sl, const uint32_t* __restrict__ sr, uint32_t capacity,
                                                        const nh_Transform* __restrict__ bt, const nh_BodyProperties* __restrict__ bp, const nh_BodyMomentum* __restrict__ bm, const uint8_t* __restrict__ bi,
                                                        nh_PartCols pc, nh_FullRecord* __restrict__ out_l, nh_FullRecord* __restrict__ out_r, uint32_t* __restrict__ list_l, uint32_t* __restrict__ list_r, nh_PartDev* dv) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t pl = sl[i], pr = sr[i];
		const bool l = sl[i + 1] != pl, r = sr[i + 1] != pr;
		if (!l && !r) continue;
		const nh_FullRecord rec = np_full(i, bt, bp, bm, bi, pc);
		if (l) { if (pl < capacity && out_l) out_l[pl] = rec; else dv->error = NH_ERR_INVALID; if (list_l) list_l[pl] = i; }
		if (r) { if (pr < capacity && out_r) out_r[pr] = rec; else dv->error = NH_ERR_INVALID; if (list_r) list_r[pr] = i; }
	}
}

// the bodies that stay, in slot order, into the staging buffer (keep position = i - leavers before i); then back
__global__ __launch_bounds__(256) void k_part_stage_kept(uint32_t n, const uint32_t* __restrict__ sl, const uint32_t* __restrict__ sr,
                                                         const nh_Transform* __restrict__ bt, const nh_BodyProperties* __restrict__ bp, const nh_BodyMomentum* __restrict__ bm, const uint8_t* __restrict__ bi,
                                                         nh_PartCols pc, nh_FullRecord* __restrict__ stage) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t pl = sl[i], pr = sr[i];
		if (sl[i + 1] != pl || sr[i + 1] != pr) continue;
		stage[i - pl - pr] = np_full(i, bt, bp, bm, bi, pc);
	}
}

// full records -> body slots [at, at + m): the world's arrays and the body-parallel description (reach follows from the colliders)
__global__ __launch_bounds__(256) void k_part_install(uint32_t at, uint32_t m, const nh_FullRecord* __restrict__ in,
                                                      nh_Transform* __restrict__ bt, nh_BodyProperties* __restrict__ bp, nh_BodyMomentum* __restrict__ bm, uint8_t* __restrict__ bi, nh_PartCols pc) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
		const nh_FullRecord r = in[k];
		const uint32_t i = at + k;
		bt[i] = r.xf; bp[i] = r.props; bm[i] = r.momentum; bi[i] = r.idle;
		pc.nbx[i] = r.nbox; pc.nsp[i] = r.nsph;
		for (uint32_t c = 0; c < NP_MAXC; ++c) {
			pc.shape[(size_t)NP_MAXC * i + c] = make_float4(r.col[c].shape[0], r.col[c].shape[1], r.col[c].shape[2], r.col[c].shape[3]);
			pc.cxf[(size_t)NP_MAXC * i + c] = r.col[c].cxf; pc.tag[(size_t)NP_MAXC * i + c] = r.col[c].tag;
		}
		pc.reach[i] = np_body_reach(pc, i);
	}
}

// refresh phase 2: top speed of the owned bodies (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void k_part_vmax(uint32_t n, const nh_BodyMomentum* __restrict__ bm, nh_PartDev* dv) {
	float top = 0.0f;
	for (uint32_t i = 1u + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const nh_BodyMomentum m = bm[i];
		const float v = sqrtf(m.velocity[0] * m.velocity[0] + m.velocity[1] * m.velocity[1] + m.velocity[2] * m.velocity[2]);
		top = v > top ? v : top;           // (NaN: not taken)
	}
	for (int d = 32; d >= 1; d >>= 1) top = fmaxf(top, __shfl_xor(top, d));
	__shared__ float s_top[4];
	if ((threadIdx.x & 63u) == 0u) s_top[threadIdx.x >> 6] = top;
	__syncthreads();
	if (threadIdx.x == 0) { top = fmaxf(fmaxf(s_top[0], s_top[1]), fmaxf(s_top[2], s_top[3])); if (top > 0.0f) atomicMax(&dv->vmax_bits, __float_as_uint(top)); }
}

// ... who can reach across a cut within the coming epoch: distance to the cut < reach + max reach of anybody + 2 x (how far anything travels in an epoch)
// (`2 x drift` bounds the approach of the TWO bodies of a pair across the cut, so the speed must be the top speed on both sides of THAT cut: this rank's -- now, and
// before this refresh's migration took its fastest body away -- and the neighbour's, which the host hands in (speed_left / speed_right; round-4 advisor finding: with
// this rank's own speed only, a fast body owned by the neighbour reached a slow body here that nobody had listed for it, and tunnelled through until the next refresh))
__global__ __launch_bounds__(256) void k_part_flag_ghosts(uint32_t n, const nh_Transform* __restrict__ bt, const float* __restrict__ reach, float lo, float hi, uint32_t has_left, uint32_t has_right,
                                                          double max_reach, double epoch_dt, double speed_floor, double gravity_epoch_dt, const nh_PartDev* __restrict__ dv,
                                                          uint32_t* __restrict__ fl, uint32_t* __restrict__ fr, double speed_pre, double speed_left, double speed_right) {
	double own = (double)__uint_as_float(dv->vmax_bits);
	if (speed_pre > own) own = speed_pre;
	if (speed_floor > own) own = speed_floor;
	const double vl = speed_left > own ? speed_left : own, vr = speed_right > own ? speed_right : own;
	const float extra_l = (float)(max_reach + 2.0 * epoch_dt * (vl + gravity_epoch_dt)), extra_r = (float)(max_reach + 2.0 * epoch_dt * (vr + gravity_epoch_dt));
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
		uint32_t l = 0u, r = 0u;
		if (i >= 1u && i < n) {
			const float x = bt[i].position[0];
			l = (has_left && x - (reach[i] + extra_l) < lo) ? 1u : 0u;
			r = (has_right && x + (reach[i] + extra_r) >= hi) ? 1u : 0u;
		}
		fl[i] = l; fr[i] = r;
	}
}

// collider rebuild: how many boxes / spheres every dynamic body carries (scanned by the host's launch in between), then the ordered write behind the static colliders
__global__ __launch_bounds__(256) void k_part_flag_kinds(uint32_t n, const uint8_t* __restrict__ nbx, const uint8_t* __restrict__ nsp, uint32_t* __restrict__ f0, uint32_t* __restrict__ f1) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
		const bool body = i >= 1u && i < n;
		f0[i] = body ? (uint32_t)nbx[i] : 0u; f1[i] = body ? (uint32_t)nsp[i] : 0u;
	}
}

__global__ __launch_bounds__(256) void k_part_write_colliders(uint32_t n, const uint32_t* __restrict__ s0, const uint32_t* __restrict__ s1, nh_PartCols pc,
                                                              uint32_t n_static_box, uint32_t box_capacity, uint32_t* __restrict__ xt, nh_BoxCollider* __restrict__ xd, nh_Transform* __restrict__ xx,
                                                              uint32_t n_static_sph, uint32_t sph_capacity, uint32_t* __restrict__ st, nh_SphereCollider* __restrict__ sd, nh_Transform* __restrict__ sx,
                                                              nh_PartDev* dv) {
	for (uint32_t i = 1u + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t nb = pc.nbx[i], ns = pc.nsp[i];
		for (uint32_t k = 0; k < nb + ns && k < NP_MAXC; ++k) {
			nh_Transform t = pc.cxf[(size_t)NP_MAXC * i + k];
			t.body = i;                                  // Transform.body of a collider = its body's slot (nudge.h:36)
			const float4 s = pc.shape[(size_t)NP_MAXC * i + k];
			if (k < nb) {
				const uint32_t c = n_static_box + s0[i] + k;
				if (c >= box_capacity) { dv->error = NH_ERR_INVALID; continue; }
				nh_BoxCollider b = { { s.x, s.y, s.z }, s.w };
				xt[c] = pc.tag[(size_t)NP_MAXC * i + k]; xd[c] = b; xx[c] = t;
			} else {
				const uint32_t c = n_static_sph + s1[i] + (k - nb);
				if (c >= sph_capacity) { dv->error = NH_ERR_INVALID; continue; }
				nh_SphereCollider sp = { s.x };
				st[c] = pc.tag[(size_t)NP_MAXC * i + k]; sd[c] = sp; sx[c] = t;
			}
		}
	}
}

// re-balancing: x of the owned bodies as sortable keys
__global__ __launch_bounds__(256) void k_part_x_keys(uint32_t n, const nh_Transform* __restrict__ bt, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i + 1u < n; i += gridDim.x * blockDim.x) { keys[i] = nh_float_flip(bt[i + 1u].position[0]); vals[i] = i; }
}
__global__ void k_part_kth(const uint32_t* __restrict__ sorted, uint32_t ka, uint32_t kb, nh_PartDev* dv) { dv->kth[0] = nh_float_unflip(sorted[ka]); dv->kth[1] = nh_float_unflip(sorted[kb]); }
__global__ void k_part_reset(nh_PartDev* dv) { dv->count[0] = dv->count[1] = 0u; dv->kinds[0] = dv->kinds[1] = 0u; dv->vmax_bits = 0u; dv->zero = 0u; }

// ---- host ----------------------------------------------------------------------------------------------------------------------------------------------
static nh_PartCols part_cols(const nh_partition* p) { nh_PartCols pc = { p->shape, p->cxf, p->tag, p->nbx, p->nsp, p->reach }; return pc; }
static int part_read_back(nh_partition* p) {
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipMemcpyAsync(p->h, p->d, sizeof(nh_PartDev), hipMemcpyDeviceToHost, ctx->stream));
	NP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
	return p->h->error ? (int)p->h->error : NH_OK;
}

extern "C" int nh_partition_create(nh_partition** out, nh_context* ctx, const nh_PartitionConfig* cfg, const nh_BodyData* bodies, const nh_ColliderData* colliders) {
	if (!out || !ctx || !cfg || !bodies || !colliders) return NH_ERR_INVALID;
	if (cfg->body_capacity < cfg->n_owned + 1u || bodies->count != cfg->n_owned + 1u || cfg->ranks == 0u || cfg->rank >= cfg->ranks || cfg->epoch == 0u) return NH_ERR_INVALID;
	if (colliders->boxes.count < cfg->n_static_box || colliders->spheres.count < cfg->n_static_sph) return NH_ERR_INVALID;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	nh_partition* p = new nh_partition();
	memset(p, 0, sizeof(*p));
	p->ctx = ctx; p->cfg = *cfg; p->lo = cfg->lo; p->hi = cfg->hi; p->cut0[0] = cfg->lo; p->cut0[1] = cfg->hi;
	p->n_owned = cfg->n_owned; p->n_bodies = cfg->n_owned + 1u; p->n_boxes = colliders->boxes.count; p->n_spheres = colliders->spheres.count;
	const size_t cap = cfg->body_capacity;
	const size_t sort_words = 256u * NH_SORT_GRID + 512u;
	bool ok = hipMalloc((void**)&p->shape, sizeof(float4) * cap * NP_MAXC) == hipSuccess && hipMalloc((void**)&p->cxf, sizeof(nh_Transform) * cap * NP_MAXC) == hipSuccess &&
	          hipMalloc((void**)&p->tag, sizeof(uint32_t) * cap * NP_MAXC) == hipSuccess && hipMalloc((void**)&p->nbx, cap + 64u) == hipSuccess && hipMalloc((void**)&p->nsp, cap + 64u) == hipSuccess &&
	          hipMalloc((void**)&p->reach, sizeof(float) * cap) == hipSuccess &&
	          hipMalloc((void**)&p->list[0], sizeof(uint32_t) * cap) == hipSuccess && hipMalloc((void**)&p->list[1], sizeof(uint32_t) * cap) == hipSuccess &&
	          hipMalloc((void**)&p->flag[0], sizeof(uint32_t) * (cap + 2u)) == hipSuccess && hipMalloc((void**)&p->flag[1], sizeof(uint32_t) * (cap + 2u)) == hipSuccess &&
	          hipMalloc((void**)&p->scan_tmp, sizeof(uint32_t) * (2u * NH_SORT_GRID + 64u)) == hipSuccess &&
	          hipMalloc((void**)&p->sort_a, sizeof(uint32_t) * cap) == hipSuccess && hipMalloc((void**)&p->sort_b, sizeof(uint32_t) * cap) == hipSuccess &&
	          hipMalloc((void**)&p->sort_va, sizeof(uint32_t) * cap) == hipSuccess && hipMalloc((void**)&p->sort_vb, sizeof(uint32_t) * cap) == hipSuccess &&
	          hipMalloc((void**)&p->sort_hist, sizeof(uint32_t) * sort_words) == hipSuccess &&
	          hipMalloc((void**)&p->stage, sizeof(nh_FullRecord) * cap) == hipSuccess && hipMalloc((void**)&p->d, sizeof(nh_PartDev)) == hipSuccess &&
	          hipHostMalloc((void**)&p->h, sizeof(nh_PartDev), 0) == hipSuccess;
	if (!ok) { nh_partition_destroy(p); return NH_ERR_HIP; }
	hipStream_t s = ctx->stream;
	hipMemsetAsync(p->d, 0, sizeof(nh_PartDev), s);
	hipMemsetAsync(p->nbx, 0, cap, s); hipMemsetAsync(p->nsp, 0, cap, s);
	hipMemsetAsync(p->reach, 0, sizeof(float) * cap, s);
	hipMemsetAsync(p->shape, 0, sizeof(float4) * cap * NP_MAXC, s);
	hipMemsetAsync(p->cxf, 0, sizeof(nh_Transform) * cap * NP_MAXC, s);
	hipMemsetAsync(p->tag, 0, sizeof(uint32_t) * cap * NP_MAXC, s);
	const uint32_t nbx = colliders->boxes.count, nsp = colliders->spheres.count;
	// (boxes first: a body's spheres go behind its boxes, so the box counts must be final when the spheres are adopted -- two launches on one stream)
	if (nbx > cfg->n_static_box)
		hipLaunchKernelGGL(k_part_adopt, dim3(nh_grid_for(nbx - cfg->n_static_box, 256, 2048)), dim3(256), 0, s, cfg->n_static_box, nbx, 0u, colliders->boxes.tags, (const void*)colliders->boxes.data,
		                   colliders->boxes.transforms, cfg->body_capacity, part_cols(p), p->d);
	if (nsp > cfg->n_static_sph)
		hipLaunchKernelGGL(k_part_adopt, dim3(nh_grid_for(nsp - cfg->n_static_sph, 256, 2048)), dim3(256), 0, s, cfg->n_static_sph, nsp, 1u, colliders->spheres.tags, (const void*)colliders->spheres.data,
		                   colliders->spheres.transforms, cfg->body_capacity, part_cols(p), p->d);
	hipLaunchKernelGGL(k_part_reaches, dim3(nh_grid_for(p->n_bodies, 256, 2048)), dim3(256), 0, s, p->n_bodies, part_cols(p));
	int rc = part_read_back(p);
	if (rc) { nh_partition_destroy(p); return rc; }
	*out = p;
	return NH_OK;
}

extern "C" void nh_partition_destroy(nh_partition* p) {
	if (!p) return;
	hipSetDevice(p->ctx->device);
	hipStreamSynchronize(p->ctx->stream);
	for (int side = 0; side < 2; ++side) { if (p->step_out[side]) hipFree(p->step_out[side]); if (p->step_in[side]) hipFree(p->step_in[side]); }
	void* bufs[] = { p->shape, p->cxf, p->tag, p->nbx, p->nsp, p->reach, p->list[0], p->list[1], p->flag[0], p->flag[1], p->scan_tmp, p->sort_a, p->sort_b, p->sort_va, p->sort_vb, p->sort_hist, p->stage, p->d, p->ghost_ref,
	                 p->loop_out, p->loop_in, p->loop_idx, p->split_list, p->split_count, p->split_mask };
	for (void* b : bufs) if (b) hipFree(b);
	if (p->split_stream) { hipStreamSynchronize(p->split_stream); hipStreamDestroy(p->split_stream); }
	if (p->split_done) hipEventDestroy(p->split_done);
	if (p->h) hipHostFree(p->h);
	delete p;
}

extern "C" int nh_partition_info(nh_partition* p, nh_PartitionInfo* out) {
	if (!p || !out) return NH_ERR_INVALID;
	out->n_owned = p->n_owned; out->n_bodies = p->n_bodies; out->n_boxes = p->n_boxes; out->n_spheres = p->n_spheres;
	out->ghost_out[0] = p->ghost_out[0]; out->ghost_out[1] = p->ghost_out[1]; out->ghost_in[0] = p->ghost_in[0]; out->ghost_in[1] = p->ghost_in[1];
	out->lo = p->lo; out->hi = p->hi;
	out->migrated_out = p->migrated_out; out->migrated_in = p->migrated_in; out->refreshes = p->refreshes; out->cut_moves = p->cut_moves; out->quiet_refreshes = p->quiet_refreshes;
	return NH_OK;
}

#define NP_LAUNCH(name, kernel, n_items, ...) NH_LAUNCH(ctx, name, kernel, nh_grid_for((uint64_t)(n_items) + 1u, 256, 4096), 256, __VA_ARGS__)

extern "C" int nh_partition_pack_migrants(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right, uint32_t capacity_records, uint32_t counts[2]) {
	if (!p || !bodies || !counts) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	const uint32_t n = p->n_owned + 1u;
	const uint32_t has_left = p->cfg.rank > 0u && out_left, has_right = p->cfg.rank + 1u < p->cfg.ranks && out_right;
	NH_LAUNCH(ctx, "part_reset", k_part_reset, 1, 1, p->d);
	NP_LAUNCH("part_flag_leavers", k_part_flag_leavers, n, n, bodies->transforms, (float)p->lo, (float)p->hi, has_left ? 1u : 0u, has_right ? 1u : 0u, p->flag[0], p->flag[1]);
	nh_scan2_u32(ctx, p->flag[0], p->flag[0], &p->d->count[0], p->flag[1], p->flag[1], &p->d->count[1], &p->d->zero, n + 1u, p->scan_tmp);
	NP_LAUNCH("part_pack_full", k_part_pack_full, n, n, p->flag[0], p->flag[1], capacity_records, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters,
	          part_cols(p), (nh_FullRecord*)out_left, (nh_FullRecord*)out_right, (uint32_t*)nullptr, (uint32_t*)nullptr, p->d);
	int rc = part_read_back(p);
	if (rc) return rc;
	counts[0] = p->h->count[0]; counts[1] = p->h->count[1];
	if (counts[0] > capacity_records || counts[1] > capacity_records) return NH_ERR_INVALID;
	p->migrated_out += counts[0] + counts[1];
	p->leavers_flagged = true;
	return NH_OK;
}

extern "C" int nh_partition_unpack_migrants(nh_partition* p, const nh_BodyData* bodies, const void* in_left, uint32_t n_left, const void* in_right, uint32_t n_right) {
	if (!p || !bodies || (n_left && !in_left) || (n_right && !in_right) || !p->leavers_flagged) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	p->leavers_flagged = false;
	const uint32_t n = p->n_owned + 1u, left = p->h->count[0] + p->h->count[1], keep = n - left;
	if ((uint64_t)keep + n_left + n_right > p->cfg.body_capacity) return NH_ERR_INVALID;
	if (left) {
		// the bodies that stay keep their order: through the staging buffer (a gather in place would read slots another lane has already overwritten)
		NP_LAUNCH("part_stage_kept", k_part_stage_kept, n, n, p->flag[0], p->flag[1], bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, part_cols(p), p->stage);
		NP_LAUNCH("part_install", k_part_install, keep, 0u, keep, p->stage, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, part_cols(p));
	}
	uint32_t at = keep;
	if (n_left) { NP_LAUNCH("part_install", k_part_install, n_left, at, n_left, (const nh_FullRecord*)in_left, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, part_cols(p)); at += n_left; }
	if (n_right) { NP_LAUNCH("part_install", k_part_install, n_right, at, n_right, (const nh_FullRecord*)in_right, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, part_cols(p)); at += n_right; }
	p->n_owned = at - 1u;
	p->n_bodies = at;                       // (ghosts are gone until nh_partition_unpack_ghosts)
	p->ghost_in[0] = p->ghost_in[1] = 0u; p->ghost_generation++;
	p->migrated_in += n_left + n_right;
	return NH_OK;
}

// Top speed of the bodies this rank owns NOW (call it before nh_partition_pack_migrants): one kernel and the call's host synchronisation.  The host sends the value to
// both neighbours (with the migrant counts, say) and hands theirs to nh_partition_set_peer_speeds before nh_partition_pack_ghosts.
extern "C" int nh_partition_top_speed(nh_partition* p, const nh_BodyData* bodies, double* out) {
	if (!p || !bodies || !out) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	const uint32_t n = p->n_owned + 1u;
	NH_LAUNCH(ctx, "part_reset", k_part_reset, 1, 1, p->d);
	NH_LAUNCH(ctx, "part_vmax", k_part_vmax, nh_grid_for(n, 256, 1024), 256, n, bodies->momentum, p->d);
	int rc = part_read_back(p);
	if (rc) return rc;
	float v; memcpy(&v, &p->h->vmax_bits, sizeof(v));
	p->speed_pre = (double)v;
	*out = p->speed_pre;
	return NH_OK;
}

extern "C" int nh_partition_set_peer_speeds(nh_partition* p, double left, double right) {
	if (!p) return NH_ERR_INVALID;
	p->peer_speed[0] = left == left && left > 0.0 ? left : 0.0;          // (NaN / negative: no neighbour, or none reported)
	p->peer_speed[1] = right == right && right > 0.0 ? right : 0.0;
	return NH_OK;
}

// A QUIET refresh (round 6).  A refresh re-decides who leaves and who is a ghost, ships full records, appends the ghosts again, rebuilds the collider arrays and tells the
// context that its bodies have changed -- the next sub-step is a full step, the one after it a still step that starts from the transforms.  In a world at rest nothing of
// that changes anything: nobody has crossed a cut, and the bodies within reach of a cut (at the speeds measured NOW: nh_partition_top_speed / nh_partition_set_peer_speeds
// first, like before nh_partition_pack_ghosts -- nothing is consumed here) are exactly the ones on this rank's lists, in the same order.  Then the neighbours hold the
// right ghosts already and a per-step exchange is all this boundary needs.  One kernel chain and one read-back; `quiet` = 1 says so for THIS rank -- the host skips the
// refresh only when every rank of the job says so (a rank that refreshes re-orders what it sends).
__global__ __launch_bounds__(256) void k_part_sum_flags(uint32_t n, const uint32_t* __restrict__ f0, const uint32_t* __restrict__ f1, uint32_t* out0, uint32_t* out1) {
	uint32_t a = 0, b = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { a += f0[i]; b += f1[i]; }
	for (int d = 32; d >= 1; d >>= 1) { a += (uint32_t)__shfl_xor((int)a, d); b += (uint32_t)__shfl_xor((int)b, d); }
	if ((threadIdx.x & 63u) == 0u) { if (a) atomicAdd(out0, a); if (b) atomicAdd(out1, b); }
}
__global__ __launch_bounds__(256) void k_part_lists_flagged(const uint32_t* __restrict__ list0, uint32_t n0, const uint32_t* __restrict__ list1, uint32_t n1, uint32_t n,
                                                            const uint32_t* __restrict__ f0, const uint32_t* __restrict__ f1, uint32_t* __restrict__ misses) {
	uint32_t m = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += gridDim.x * blockDim.x) {
		const uint32_t b = i < n0 ? list0[i] : list1[i - n0];
		m += (b >= 1u && b < n && (i < n0 ? f0[b] : f1[b]) != 0u) ? 0u : 1u;
	}
	for (int d = 32; d >= 1; d >>= 1) m += (uint32_t)__shfl_xor((int)m, d);
	if ((threadIdx.x & 63u) == 0u && m) atomicAdd(misses, m);
}
extern "C" int nh_partition_refresh_is_quiet(nh_partition* p, const nh_BodyData* bodies, int* quiet) {
	if (!p || !bodies || !quiet) return NH_ERR_INVALID;
	*quiet = 0;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	if (p->refreshes == 0u || p->leavers_flagged) return NH_OK;          // (nothing to compare with yet: the first refresh installs ghosts and colliders)
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	const uint32_t n = p->n_owned + 1u;
	const uint32_t has_left = p->cfg.rank > 0u, has_right = p->cfg.rank + 1u < p->cfg.ranks;
	const double epoch_dt = (double)p->cfg.epoch * p->cfg.time_step;
	NH_LAUNCH(ctx, "part_reset", k_part_reset, 1, 1, p->d);
	NH_LAUNCH(ctx, "part_vmax", k_part_vmax, nh_grid_for(n, 256, 1024), 256, n, bodies->momentum, p->d);
	// (1) nobody leaves: count[0] = bodies beyond either cut
	NP_LAUNCH("part_flag_leavers", k_part_flag_leavers, n, n, bodies->transforms, (float)p->lo, (float)p->hi, has_left, has_right, p->flag[0], p->flag[1]);
	NH_LAUNCH(ctx, "part_sum_flags", k_part_sum_flags, nh_grid_for(n, 256, 256), 256, n, p->flag[0], p->flag[1], &p->d->count[0], &p->d->count[0]);
	// (2) the bodies within reach of a cut are the listed ones: as many (kinds[0 / 1]), and every listed one among them (count[1] = listed bodies that are not)
	NP_LAUNCH("part_flag_ghosts", k_part_flag_ghosts, n, n, bodies->transforms, p->reach, (float)p->lo, (float)p->hi, has_left, has_right,
	          p->cfg.max_reach, epoch_dt, p->cfg.speed_floor, fabs(p->cfg.gravity) * epoch_dt, p->d, p->flag[0], p->flag[1], p->speed_pre, p->peer_speed[0], p->peer_speed[1]);
	NH_LAUNCH(ctx, "part_sum_flags", k_part_sum_flags, nh_grid_for(n, 256, 256), 256, n, p->flag[0], p->flag[1], &p->d->kinds[0], &p->d->kinds[1]);
	if (p->ghost_out[0] + p->ghost_out[1])
		NH_LAUNCH(ctx, "part_lists_flagged", k_part_lists_flagged, nh_grid_for(p->ghost_out[0] + p->ghost_out[1], 256, 256), 256, p->list[0], p->ghost_out[0], p->list[1], p->ghost_out[1], n,
		          p->flag[0], p->flag[1], &p->d->count[1]);
	int rc = part_read_back(p);
	if (rc) return rc;
	*quiet = (p->h->count[0] == 0u && p->h->count[1] == 0u && p->h->kinds[0] == p->ghost_out[0] && p->h->kinds[1] == p->ghost_out[1]) ? 1 : 0;
	if (*quiet) p->quiet_refreshes++;
	return NH_OK;
}

extern "C" int nh_partition_pack_ghosts(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right, uint32_t capacity_records, uint32_t counts[2]) {
	if (!p || !bodies || !counts) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	const uint32_t n = p->n_owned + 1u;
	const uint32_t has_left = p->cfg.rank > 0u && out_left, has_right = p->cfg.rank + 1u < p->cfg.ranks && out_right;
	const double epoch_dt = (double)p->cfg.epoch * p->cfg.time_step;
	NH_LAUNCH(ctx, "part_reset", k_part_reset, 1, 1, p->d);
	NH_LAUNCH(ctx, "part_vmax", k_part_vmax, nh_grid_for(n, 256, 1024), 256, n, bodies->momentum, p->d);
	NP_LAUNCH("part_flag_ghosts", k_part_flag_ghosts, n, n, bodies->transforms, p->reach, (float)p->lo, (float)p->hi, has_left ? 1u : 0u, has_right ? 1u : 0u,
	          p->cfg.max_reach, epoch_dt, p->cfg.speed_floor, fabs(p->cfg.gravity) * epoch_dt, p->d, p->flag[0], p->flag[1], p->speed_pre, p->peer_speed[0], p->peer_speed[1]);
	p->speed_pre = 0.0; p->peer_speed[0] = p->peer_speed[1] = 0.0;          // (consumed: the next refresh measures and exchanges again)
	nh_scan2_u32(ctx, p->flag[0], p->flag[0], &p->d->count[0], p->flag[1], p->flag[1], &p->d->count[1], &p->d->zero, n + 1u, p->scan_tmp);
	NP_LAUNCH("part_pack_full", k_part_pack_full, n, n, p->flag[0], p->flag[1], capacity_records, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters,
	          part_cols(p), (nh_FullRecord*)out_left, (nh_FullRecord*)out_right, p->list[0], p->list[1], p->d);
	int rc = part_read_back(p);
	if (rc) return rc;
	counts[0] = p->h->count[0]; counts[1] = p->h->count[1];
	if (counts[0] > capacity_records || counts[1] > capacity_records) return NH_ERR_INVALID;
	p->ghost_out[0] = counts[0]; p->ghost_out[1] = counts[1];
	return NH_OK;
}

extern "C" int nh_partition_unpack_ghosts(nh_partition* p, nh_BodyData* bodies, nh_ColliderData* colliders, const void* in_left, uint32_t n_left, const void* in_right, uint32_t n_right) {
	if (!p || !bodies || !colliders || (n_left && !in_left) || (n_right && !in_right)) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	uint32_t at = p->n_owned + 1u;
	if ((uint64_t)at + n_left + n_right > p->cfg.body_capacity) return NH_ERR_INVALID;
	if (n_left) { NP_LAUNCH("part_install", k_part_install, n_left, at, n_left, (const nh_FullRecord*)in_left, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, part_cols(p)); at += n_left; }
	if (n_right) { NP_LAUNCH("part_install", k_part_install, n_right, at, n_right, (const nh_FullRecord*)in_right, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, part_cols(p)); at += n_right; }
	p->ghost_in[0] = n_left; p->ghost_in[1] = n_right;
	p->ghost_generation++;          // (whatever nh_partition_mark_ghosts noted belongs to the ghosts that were here before)
	p->n_bodies = at;
	// colliders: the static ones stay, then one per dynamic body slot (owned, then ghosts) -- boxes and spheres each compacted in slot order
	NH_LAUNCH(ctx, "part_reset", k_part_reset, 1, 1, p->d);
	NP_LAUNCH("part_flag_kinds", k_part_flag_kinds, at, at, p->nbx, p->nsp, p->flag[0], p->flag[1]);
	nh_scan2_u32(ctx, p->flag[0], p->flag[0], &p->d->kinds[0], p->flag[1], p->flag[1], &p->d->kinds[1], &p->d->zero, at + 1u, p->scan_tmp);
	NP_LAUNCH("part_write_colliders", k_part_write_colliders, at, at, p->flag[0], p->flag[1], part_cols(p),
	          p->cfg.n_static_box, p->cfg.box_capacity, colliders->boxes.tags, colliders->boxes.data, colliders->boxes.transforms,
	          p->cfg.n_static_sph, p->cfg.sphere_capacity, colliders->spheres.tags, colliders->spheres.data, colliders->spheres.transforms, p->d);
	int rc = part_read_back(p);
	if (rc) return rc;
	p->n_boxes = p->cfg.n_static_box + p->h->kinds[0]; p->n_spheres = p->cfg.n_static_sph + p->h->kinds[1];
	if (p->n_boxes > p->cfg.box_capacity || p->n_spheres > p->cfg.sphere_capacity) return NH_ERR_INVALID;
	bodies->count = at; colliders->boxes.count = p->n_boxes; colliders->spheres.count = p->n_spheres;
	p->refreshes++;
	return nh_bodies_changed(ctx);           // body records (idle counters included) were rewritten
}

extern "C" int nh_partition_pack_step(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right) {
	if (!p || !bodies) return NH_ERR_INVALID;
	if (p->ghost_out[0] && out_left) { int rc = nh_halo_pack(p->ctx, bodies, p->list[0], p->ghost_out[0], out_left); if (rc) return rc; }
	if (p->ghost_out[1] && out_right) { int rc = nh_halo_pack(p->ctx, bodies, p->list[1], p->ghost_out[1], out_right); if (rc) return rc; }
	return NH_OK;
}

extern "C" int nh_partition_unpack_step(nh_partition* p, const nh_BodyData* bodies, const void* in_left, const void* in_right) {
	if (!p || !bodies) return NH_ERR_INVALID;
	uint32_t at = p->n_owned + 1u;
	if (p->ghost_in[0]) { if (!in_left) return NH_ERR_INVALID; int rc = nh_halo_update(p->ctx, bodies, at, p->ghost_in[0], in_left); if (rc) return rc; }
	at += p->ghost_in[0];
	if (p->ghost_in[1]) { if (!in_right) return NH_ERR_INVALID; int rc = nh_halo_update(p->ctx, bodies, at, p->ghost_in[1], in_right); if (rc) return rc; }
	return NH_OK;
}

__global__ void k_part_set_sort_count(nh_PartDev* dv, uint32_t n) { dv->kinds[0] = n; }

extern "C" int nh_partition_choose_cut(nh_partition* p, const nh_BodyData* bodies, int direction, uint32_t neighbour_owned, double* cut_out) {
	if (!p || !bodies || !cut_out || (direction != -1 && direction != 1)) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	const uint32_t n = p->n_owned;
	const double old = direction < 0 ? p->lo : p->hi;
	double cut = old;
	if (n > 1u && n > neighbour_owned) {
		// hand half the difference -- at most 5 % of my bodies -- to the neighbour: the cut goes between the k-th and the (k+1)-th body counted from that side
		uint32_t k = (n - neighbour_owned) / 2u;
		const uint32_t five = (uint32_t)(0.05 * (double)n);
		if (k > five) k = five;
		if (k > n - 1u) k = n - 1u;
		if (k < 1u) k = 1u;
		NH_LAUNCH(ctx, "part_reset", k_part_set_sort_count, 1, 1, p->d, n);
		NP_LAUNCH("part_x_keys", k_part_x_keys, n, n + 1u, bodies->transforms, p->sort_a, p->sort_va);
		const int in_b = nh_sort_u32_u32(ctx, p->sort_a, p->sort_b, p->sort_va, p->sort_vb, &p->d->kinds[0], p->sort_hist, 0, 32);
		const uint32_t* sorted = in_b ? p->sort_b : p->sort_a;
		// towards the left neighbour the k smallest x leave (between the k-th and the (k+1)-th smallest); towards the right one the k largest
		const uint32_t ia = direction < 0 ? k - 1u : n - k, ib = direction < 0 ? k : n - k - 1u;
		NH_LAUNCH(ctx, "part_kth", k_part_kth, 1, 1, sorted, ia, ib, p->d);
		int rc = part_read_back(p);
		if (rc) return rc;
		cut = 0.5 * ((double)p->h->kth[0] + (double)p->h->kth[1]);
		// never further from where the cut started than the static colliders this rank kept allow
		const double c0 = p->cut0[direction < 0 ? 0 : 1];
		if (cut < c0 - p->cfg.cut_slack) cut = c0 - p->cfg.cut_slack;
		if (cut > c0 + p->cfg.cut_slack) cut = c0 + p->cfg.cut_slack;
		if (!(cut == cut) || cut > 1e300 || cut < -1e300) cut = old;
	}
	if (cut != old) { if (direction < 0) p->lo = cut; else p->hi = cut; p->cut_moves++; }
	*cut_out = cut;
	return NH_OK;
}

extern "C" int nh_partition_set_cut(nh_partition* p, int direction, double cut) {
	if (!p || (direction != -1 && direction != 1) || !(cut == cut)) return NH_ERR_INVALID;
	const double old = direction < 0 ? p->lo : p->hi;
	if (cut != old) { if (direction < 0) p->lo = cut; else p->hi = cut; p->cut_moves++; }
	return NH_OK;
}

// ---- the per-step halo through RCCL, driven from here (one call per step instead of pack / host-side transport / unpack) -------------------------------------
// `comm`: an ncclComm_t of the ranks of this world, in rank order; the four functions: ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv of the RCCL library the
// host process has loaded (plain pointers: nothing of RCCL is linked into or loaded by libnudge_hip.so).  left_peer / right_peer: the communicator ranks of the two
// neighbours (-1: none).
extern "C" int nh_partition_set_transport(nh_partition* p, void* comm, void* group_start, void* group_end, void* send, void* recv, int left_peer, int right_peer) {
	if (!p) return NH_ERR_INVALID;
	if (comm && (!group_start || !group_end || !send || !recv)) return NH_ERR_INVALID;
	p->rccl.comm = comm;
	p->rccl.group_start = reinterpret_cast<int (*)(void)>(group_start); p->rccl.group_end = reinterpret_cast<int (*)(void)>(group_end);
	p->rccl.send = reinterpret_cast<int (*)(const void*, size_t, int, int, void*, hipStream_t)>(send);
	p->rccl.recv = reinterpret_cast<int (*)(void*, size_t, int, int, void*, hipStream_t)>(recv);
	p->rccl.peer[0] = left_peer; p->rccl.peer[1] = right_peer;
	return NH_OK;
}

static int part_step_buffer(nh_partition* p, int side, int io, uint32_t records) {
	if (p->step_cap[side][io] >= records && (io ? p->step_in[side] : p->step_out[side])) return NH_OK;
	void** buf = io ? &p->step_in[side] : &p->step_out[side];
	if (*buf) { hipStreamSynchronize(p->ctx->stream); hipFree(*buf); *buf = nullptr; }
	const uint32_t cap = records + records / 4u + 256u;
	if (hipMalloc(buf, (size_t)cap * NH_HALO_RECORD_BYTES) != hipSuccess) return NH_ERR_HIP;
	p->step_cap[side][io] = cap;
	return NH_OK;
}

// ---- single ownership of contacts that cross a cut: the per-iteration exchange (include/nudge_hip.h; round 5: behind the C ABI) -------------------------------------------
// With nh_set_first_ghost_body a contact is solved by ONE rank (the owner of the body with the larger collider tag: nudge.cpp:2074-2087, 2131-2132), so a sweep here
// moves ghosts -- bodies another rank owns.  After every sweep (and after the warm start) what it did to them goes home: delta = momentum now - momentum at the last
// mark, added to the owner's record; then the owners' momentum goes out to their ghosts again and is marked.  All 32-byte momentum records; lengths are the ghost
// lists of the last refresh, in the neighbour's list order.  Same arithmetic as nudge_amd/partition.py's torch form (delta_pack / delta_unpack / momentum_pack /
// momentum_unpack / delta_mark), which stays as the independent check.
__global__ __launch_bounds__(256) void k_part_mark(uint32_t n, uint32_t at, const nh_BodyMomentum* __restrict__ bm, float4* __restrict__ ref) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const float4* m = reinterpret_cast<const float4*>(bm + at + k);
		ref[2u * k] = m[0]; ref[2u * k + 1u] = m[1];
	}
}
__global__ __launch_bounds__(256) void k_part_pack_delta(uint32_t n, uint32_t at, uint32_t ref_at, const nh_BodyMomentum* __restrict__ bm, const float4* __restrict__ ref, float4* __restrict__ out) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const float4* m = reinterpret_cast<const float4*>(bm + at + k);
		const float4 a0 = m[0], a1 = m[1], r0 = ref[2u * (ref_at + k)], r1 = ref[2u * (ref_at + k) + 1u];
		// (the two spare words of a momentum record are the engine's -- the inverse mass rides there during a solve, nudge.cpp:4358 -- and never travel)
		out[2u * k] = make_float4(a0.x - r0.x, a0.y - r0.y, a0.z - r0.z, 0.0f);
		out[2u * k + 1u] = make_float4(a1.x - r1.x, a1.y - r1.y, a1.z - r1.z, 0.0f);
	}
}
__global__ __launch_bounds__(256) void k_part_add_delta(uint32_t n, const uint32_t* __restrict__ list, nh_BodyMomentum* __restrict__ bm, const float4* __restrict__ in) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		float4* m = reinterpret_cast<float4*>(bm + list[k]);
		const float4 d0 = in[2u * k], d1 = in[2u * k + 1u];
		float4 a0 = m[0], a1 = m[1];
		a0.x += d0.x; a0.y += d0.y; a0.z += d0.z; a0.w += d0.w; a1.x += d1.x; a1.y += d1.y; a1.z += d1.z; a1.w += d1.w;          // (all eight words, like index_add_: the spare ones get + 0)
		m[0] = a0; m[1] = a1;
	}
}
__global__ __launch_bounds__(256) void k_part_pack_mom(uint32_t n, const uint32_t* __restrict__ list, const nh_BodyMomentum* __restrict__ bm, float4* __restrict__ out) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const float4* m = reinterpret_cast<const float4*>(bm + list[k]);
		out[2u * k] = m[0]; out[2u * k + 1u] = m[1];
	}
}
__global__ __launch_bounds__(256) void k_part_unpack_mom(uint32_t n, uint32_t at, nh_BodyMomentum* __restrict__ bm, const float4* __restrict__ in) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		float4* m = reinterpret_cast<float4*>(bm + at + k);
		m[0] = in[2u * k]; m[1] = in[2u * k + 1u];
	}
}

static int part_iteration_enter(nh_partition* p, const nh_BodyData* bodies) {
	if (!p || !bodies || !bodies->momentum) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	// (a deferred setup -- rows, warm start: header note 7 -- has to be in the momentum the records are taken from)
	return nh_flush_pending(ctx);
}

extern "C" int nh_partition_mark_ghosts(nh_partition* p, const nh_BodyData* bodies) {
	{ int rc = part_iteration_enter(p, bodies); if (rc) return rc; }
	nh_context* ctx = p->ctx;
	const uint32_t n = p->ghost_in[0] + p->ghost_in[1];
	if (p->ghost_ref_capacity < n) {
		if (p->ghost_ref) { NP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); NP_CHECK(ctx, hipFree(p->ghost_ref)); }
		p->ghost_ref = nullptr; p->ghost_ref_capacity = 0;
		const uint32_t cap = n + n / 4u + 256u;
		NP_CHECK(ctx, hipMalloc((void**)&p->ghost_ref, sizeof(float4) * 2u * (size_t)cap));
		p->ghost_ref_capacity = cap;
	}
	if (n) NP_LAUNCH("part_mark", k_part_mark, n, n, p->n_owned + 1u, bodies->momentum, p->ghost_ref);
	p->ghost_ref_generation = p->ghost_generation;
	return NH_OK;
}

extern "C" int nh_partition_pack_deltas(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right) {
	{ int rc = part_iteration_enter(p, bodies); if (rc) return rc; }
	nh_context* ctx = p->ctx;
	// (no mark since the ghosts changed: the references on file are other bodies' momenta)
	if (p->ghost_ref_capacity < p->ghost_in[0] + p->ghost_in[1] || p->ghost_ref_generation != p->ghost_generation) return NH_ERR_INVALID;
	uint32_t at = p->n_owned + 1u, ref_at = 0u;
	void* out[2] = { out_left, out_right };
	for (int side = 0; side < 2; ++side) {
		const uint32_t n = p->ghost_in[side];
		if (n) { if (!out[side]) return NH_ERR_INVALID; NP_LAUNCH("part_pack_delta", k_part_pack_delta, n, n, at, ref_at, bodies->momentum, p->ghost_ref, (float4*)out[side]); }
		at += n; ref_at += n;
	}
	return NH_OK;
}

extern "C" int nh_partition_unpack_deltas(nh_partition* p, const nh_BodyData* bodies, const void* in_left, const void* in_right) {
	{ int rc = part_iteration_enter(p, bodies); if (rc) return rc; }
	nh_context* ctx = p->ctx;
	const void* in[2] = { in_left, in_right };
	for (int side = 0; side < 2; ++side) {          // (left neighbour's first, then the right one's: a body listed for both gets the two sums in this order)
		const uint32_t n = p->ghost_out[side];
		if (n) { if (!in[side]) return NH_ERR_INVALID; NP_LAUNCH("part_add_delta", k_part_add_delta, n, n, p->list[side], bodies->momentum, (const float4*)in[side]); }
	}
	return NH_OK;
}

extern "C" int nh_partition_pack_momentum(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right) {
	{ int rc = part_iteration_enter(p, bodies); if (rc) return rc; }
	nh_context* ctx = p->ctx;
	void* out[2] = { out_left, out_right };
	for (int side = 0; side < 2; ++side) {
		const uint32_t n = p->ghost_out[side];
		if (n) { if (!out[side]) return NH_ERR_INVALID; NP_LAUNCH("part_pack_mom", k_part_pack_mom, n, n, p->list[side], bodies->momentum, (float4*)out[side]); }
	}
	return NH_OK;
}

extern "C" int nh_partition_unpack_momentum(nh_partition* p, const nh_BodyData* bodies, const void* in_left, const void* in_right) {
	{ int rc = part_iteration_enter(p, bodies); if (rc) return rc; }
	nh_context* ctx = p->ctx;
	const void* in[2] = { in_left, in_right };
	uint32_t at = p->n_owned + 1u;
	for (int side = 0; side < 2; ++side) {
		const uint32_t n = p->ghost_in[side];
		if (n) { if (!in[side]) return NH_ERR_INVALID; NP_LAUNCH("part_unpack_mom", k_part_unpack_mom, n, n, at, bodies->momentum, (const float4*)in[side]); }
		at += n;
	}
	return NH_OK;
}

static int part_step_buffer(nh_partition* p, int side, int io, uint32_t records);
// The whole exchange of one solver iteration driven by the library over the transport of nh_partition_set_transport, in ONE call: deltas of the ghosts back to their
// owners, the owners' momentum out to the ghosts, the ghosts marked -- two RCCL groups on the context's stream, no host synchronisation.
extern "C" int nh_partition_exchange_iteration(nh_partition* p, const nh_BodyData* bodies) {
	if (!p || !bodies || !p->rccl.comm) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	for (int side = 0; side < 2; ++side) {
		const uint32_t most = p->ghost_out[side] > p->ghost_in[side] ? p->ghost_out[side] : p->ghost_in[side];
		if (most) { int rc = part_step_buffer(p, side, 0, most); if (rc) return rc; rc = part_step_buffer(p, side, 1, most); if (rc) return rc; }
	}
	for (int pass = 0; pass < 2; ++pass) {
		// pass 0: deltas -- this rank sends what it did to the neighbour's bodies (its ghosts from that side) and receives what the neighbour did to the bodies listed for it;
		// pass 1: momentum -- the listed bodies go out, the ghosts come in
		{ int rc = pass == 0 ? nh_partition_pack_deltas(p, bodies, p->step_out[0], p->step_out[1]) : nh_partition_pack_momentum(p, bodies, p->step_out[0], p->step_out[1]); if (rc) return rc; }
		int nccl_rc = p->rccl.group_start();
		if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
		for (int side = 0; side < 2 && !nccl_rc; ++side) {
			const int peer = p->rccl.peer[side];
			if (peer < 0) continue;
			const uint32_t n_out = pass == 0 ? p->ghost_in[side] : p->ghost_out[side], n_in = pass == 0 ? p->ghost_out[side] : p->ghost_in[side];
			if (n_out) nccl_rc = p->rccl.send(p->step_out[side], (size_t)n_out * 32u, 1 /* ncclUint8 */, peer, p->rccl.comm, ctx->stream);
			if (!nccl_rc && n_in) nccl_rc = p->rccl.recv(p->step_in[side], (size_t)n_in * 32u, 1, peer, p->rccl.comm, ctx->stream);
		}
		{ const int end_rc = p->rccl.group_end(); if (!nccl_rc) nccl_rc = end_rc; }
		if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
		{ int rc = pass == 0 ? nh_partition_unpack_deltas(p, bodies, p->step_in[0], p->step_in[1]) : nh_partition_unpack_momentum(p, bodies, p->step_in[0], p->step_in[1]); if (rc) return rc; }
	}
	return nh_partition_mark_ghosts(p, bodies);
}

// pack -> ncclGroupStart; ncclSend / ncclRecv with both neighbours; ncclGroupEnd -> unpack, all on the context's stream, lengths those of the last refresh
extern "C" int nh_partition_exchange_step(nh_partition* p, const nh_BodyData* bodies) {
	if (!p || !bodies || !p->rccl.comm) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	for (int side = 0; side < 2; ++side) {
		if (p->ghost_out[side]) { int rc = part_step_buffer(p, side, 0, p->ghost_out[side]); if (rc) return rc; }
		if (p->ghost_in[side]) { int rc = part_step_buffer(p, side, 1, p->ghost_in[side]); if (rc) return rc; }
	}
	{ int rc = nh_partition_pack_step(p, bodies, p->step_out[0], p->step_out[1]); if (rc) return rc; }
	// (an error inside the group still closes it: a group left open on this thread would swallow every later collective of the communicator; the first RCCL result
	// that was not ncclSuccess is kept in nh_last_hip_error, negated so that it cannot be taken for a hipError_t)
	int nccl_rc = p->rccl.group_start();
	if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
	for (int side = 0; side < 2 && !nccl_rc; ++side) {
		const int peer = p->rccl.peer[side];
		if (peer < 0) continue;
		if (p->ghost_out[side]) nccl_rc = p->rccl.send(p->step_out[side], (size_t)p->ghost_out[side] * NH_HALO_RECORD_BYTES, 1 /* ncclUint8 */, peer, p->rccl.comm, ctx->stream);
		if (!nccl_rc && p->ghost_in[side]) nccl_rc = p->rccl.recv(p->step_in[side], (size_t)p->ghost_in[side] * NH_HALO_RECORD_BYTES, 1, peer, p->rccl.comm, ctx->stream);
	}
	{ const int end_rc = p->rccl.group_end(); if (!nccl_rc) nccl_rc = end_rc; }
	if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
	return nh_partition_unpack_step(p, bodies, p->step_in[0], p->step_in[1]);
}

// ---- nh_partition_step: K sub-steps of a slab in ONE library call (round 6) ---------------------------------------------------------------------
// The sample's loop (example/main.cpp:274-328) for a partitioned world: before every sub-step the ghosts take their owners' state (nh_partition_exchange_step), then the
// eight calls.  Driven from the host one sub-step at a time the world never got past the first link of the chain of still steps: every step launched k_xform<true> and
// its narrowphase.  Inside one nh_step call -- the exchange enqueued by the library between two sub-steps, through the hook below -- a step's solver lanes do the next
// step's transforms and pairs for the OWNED bodies (xform ahead, pair ahead).  GHOSTS are different: what a rank computes for a ghost is not what its owner computes
// (the order in which a body's contacts are solved depends on the contact's index in the RANK's dense list, nudge.cpp:4206-4339), and the owner's state is what
// arrives.  So a ghost's lane leaves the next step alone (nh_FusedStep::ghost_first); when the halo has been unpacked k_xform_ghosts does k_xform<true>'s work for the
// ghosts' colliders -- world transform, box, the test against the inflated box, scene bounds, idle counters -- and the records whose body is a ghost are on the list
// k_pair_begin evaluates (k_pair_mark marks owned bodies' records only).  Verdicts are looked at inside every step: what goes out to a neighbour is a step that happened.
__global__ __launch_bounds__(256) void k_xform_ghosts(nh_DevState* __restrict__ st, uint32_t first, uint32_t nbodies, const nh_Transform* __restrict__ body_xf, const uint8_t* __restrict__ idle,
                                                      const uint8_t* __restrict__ nbx, const uint8_t* __restrict__ nsp, const uint32_t* __restrict__ s0, const uint32_t* __restrict__ s1,
                                                      uint32_t n_static_box, uint32_t n_static_sph, uint32_t nbox,
                                                      const nh_Transform* __restrict__ box_xf, const nh_BoxCollider* __restrict__ box_data, const nh_Transform* __restrict__ sph_xf, const nh_SphereCollider* __restrict__ sph_data,
                                                      float4* __restrict__ xf, float4* __restrict__ aabb_min, float4* __restrict__ aabb_max, const float4* __restrict__ fat_box,
                                                      uint32_t parity_next, uint32_t seq_next) {
	uint32_t lmin[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, lmax[3] = { 0u, 0u, 0u }, top = 0u;
	bool fail = false;
	for (uint32_t b = first + blockIdx.x * blockDim.x + threadIdx.x; b < nbodies; b += gridDim.x * blockDim.x) {
		const uint32_t cnt = idle[b];
		top = max(top, cnt); fail |= cnt == 0xffu;
		const nh_Transform bt = body_xf[b];
		const nh_quat bq = { bt.rotation[0], bt.rotation[1], bt.rotation[2], bt.rotation[3] };
		const uint32_t nb = nbx[b], ns = nsp[b];
		for (uint32_t k = 0; k < nb + ns; ++k) {
			const bool is_box = k < nb;
			const uint32_t c = is_box ? n_static_box + s0[b] + k : nbox + n_static_sph + s1[b] + (k - nb);
			const nh_Transform l = is_box ? box_xf[c] : sph_xf[c - nbox];
			// Transform * Transform (nudge.cpp:1165-1175), |R| * size (3027-3037): k_xform's lines
			const nh_quat lq = { l.rotation[0], l.rotation[1], l.rotation[2], l.rotation[3] };
			const nh_f3 p = nh_rotate(bq, nh_make3(l.position[0], l.position[1], l.position[2])) + nh_make3(bt.position[0], bt.position[1], bt.position[2]);
			const nh_quat q = nh_qmul(bq, lq);
			float sx, sy, sz;
			if (is_box) {
				const nh_m33 m = nh_matrix(q);
				const nh_BoxCollider bc = box_data[c];
				const nh_f3 c0 = m.c0 * bc.size[0], c1 = m.c1 * bc.size[1], c2 = m.c2 * bc.size[2];
				sx = fabsf(c0.x) + fabsf(c1.x) + fabsf(c2.x);
				sy = fabsf(c0.y) + fabsf(c1.y) + fabsf(c2.y);
				sz = fabsf(c0.z) + fabsf(c1.z) + fabsf(c2.z);
			} else {
				sx = sy = sz = sph_data[c - nbox].radius;
			}
			const float mnx = p.x - sx, mny = p.y - sy, mnz = p.z - sz, mxx = p.x + sx, mxy = p.y + sy, mxz = p.z + sz;
			xf[2u * (size_t)c] = make_float4(p.x, p.y, p.z, __uint_as_float(l.body));
			xf[2u * (size_t)c + 1u] = make_float4(q.x, q.y, q.z, q.s);
			aabb_min[c] = make_float4(mnx, mny, mnz, __uint_as_float(l.body));
			aabb_max[c] = make_float4(mxx, mxy, mxz, __uint_as_float(0u));
			const float4 fmn = fat_box[2u * (size_t)c], fmx = fat_box[2u * (size_t)c + 1u];
			fail |= !(mnx >= fmn.x && mny >= fmn.y && mnz >= fmn.z && mxx <= fmx.x && mxy <= fmx.y && mxz <= fmx.z);       // (NaN: out)
			uint32_t f;
			f = nh_float_flip(mnx); lmin[0] = min(lmin[0], f); lmax[0] = max(lmax[0], f);
			f = nh_float_flip(mny); lmin[1] = min(lmin[1], f); lmax[1] = max(lmax[1], f);
			f = nh_float_flip(mnz); lmin[2] = min(lmin[2], f); lmax[2] = max(lmax[2], f);
		}
	}
	for (int k = 0; k < 3; ++k)
		for (int d = 32; d >= 1; d >>= 1) { lmin[k] = min(lmin[k], (uint32_t)__shfl_xor((int)lmin[k], d)); lmax[k] = max(lmax[k], (uint32_t)__shfl_xor((int)lmax[k], d)); }
	for (int d = 32; d >= 1; d >>= 1) top = max(top, (uint32_t)__shfl_xor((int)top, d));
	const unsigned long long bad = __ballot(fail);
	if (nh_lane() == 0) {
		// (into the places the solver lanes gather in: the step's first launch puts them together -- nh_solve.hip, AHEAD)
		uint32_t* const part = &st->ahead_part[parity_next][(blockIdx.x * 4u + (threadIdx.x >> 6)) & (NH_AHEAD_PARTS - 1u)][0];
		for (int k = 0; k < 3; ++k) { if (lmin[k] != 0xffffffffu) atomicMin(&part[k], lmin[k]); if (lmax[k]) atomicMax(&part[3 + k], lmax[k]); }
		if (top) atomicMax(&part[6], top);
		if (bad) atomicMax(&st->ahead_failed_seq, seq_next);
	}
}

__global__ __launch_bounds__(256) void k_part_iota(uint32_t* idx, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) idx[i] = i + 1u; }

static int part_loop_buffers(nh_partition* p, uint32_t m) {
	nh_context* ctx = p->ctx;
	if (p->loop_cap >= m) return NH_OK;
	if (p->loop_out) { NP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); if (p->split_stream) NP_CHECK(ctx, hipStreamSynchronize(p->split_stream)); (void)hipFree(p->loop_out); (void)hipFree(p->loop_in); (void)hipFree(p->loop_idx); }
	p->loop_out = p->loop_in = nullptr; p->loop_idx = nullptr; p->loop_cap = 0;
	NP_CHECK(ctx, hipMalloc(&p->loop_out, (size_t)m * NH_HALO_RECORD_BYTES)); NP_CHECK(ctx, hipMalloc(&p->loop_in, (size_t)m * NH_HALO_RECORD_BYTES));
	NP_CHECK(ctx, hipMalloc((void**)&p->loop_idx, sizeof(uint32_t) * (size_t)m));
	hipLaunchKernelGGL(k_part_iota, dim3(nh_grid_for(m, 256, 256)), dim3(256), 0, ctx->stream, p->loop_idx, m);
	p->loop_cap = m;
	return NH_OK;
}

// ---- HALO SPLIT (round 6; nh_internal.h: nh_FusedStep::part; option "halo_overlap", OFF by default) ---------------------------------------------
// (Built, bit-exact through the recording transport and the c5 world, and in the one-rank rehearsal -- 4096 records per sub-step to this rank and back through RCCL -- no
// faster than the exchange between two sub-steps: 0.535-0.538 against 0.529-0.534 ms per 2 M bodies, profiles/r06_halo_split_ab.log.  On one GPU there is no link latency
// to hide, and 25 us of pack / RCCL / unpack are what two more launches and a second stream cost.  Kept for a node where the exchange crosses xGMI.)
// The exchange used to sit BETWEEN two solver launches: pack -> RCCL -> unpack -> the ghosts' transforms, ~30 us in which the GPU did nothing else.  A pair step's solver
// now goes out in two parts behind the step's prologue: the bodies a neighbour waits for (listed for it, its ghosts here, the rehearsal's loop-back records) on a second
// stream, everybody else on the context's.  When the verdict is in -- the step happened -- the NEXT sub-step's exchange is enqueued behind the boundary launch and travels
// beside the interior's solver; the next sub-step's hook only joins the two streams.  Nothing an interior lane touches is touched by the exchange (a one-body lane reads
// and writes its own body, collider and record), and a ghost's lane has finished before its owner's state lands on it.
__global__ __launch_bounds__(256) void k_part_mark_list(const uint32_t* __restrict__ list, uint32_t n, uint32_t nbodies, uint8_t* __restrict__ mask) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint32_t b = list[i]; if (b >= 1u && b < nbodies) mask[b] = 1; }
}
__global__ __launch_bounds__(256) void k_part_mark_range(uint32_t first, uint32_t n, uint32_t nbodies, uint8_t* __restrict__ mask) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) if (first + i < nbodies) mask[first + i] = 1;
}
__global__ __launch_bounds__(256) void k_part_mask_to_list(const uint8_t* __restrict__ mask, uint32_t nbodies, uint32_t* __restrict__ list, uint32_t cap, uint32_t* __restrict__ count) {
	for (uint32_t b = 1u + blockIdx.x * blockDim.x + threadIdx.x; b < nbodies; b += gridDim.x * blockDim.x)
		if (mask[b]) { const uint32_t k = atomicAdd(count, 1u); if (k < cap) list[k] = b; }
}
static int part_after_verdict(nh_context* ctx, void* user);
static int part_arm_split(nh_partition* p, uint32_t loopback) {
	nh_context* ctx = p->ctx;
	ctx->halo_split.armed = false; ctx->halo_split.launched = false; p->exchanged_ahead = false;
	if (!ctx->halo_overlap || !p->rccl.comm || !(ctx->flags & NH_FLAG_FUSED_STEP)) return NH_OK;
	const uint32_t m = loopback ? (loopback < p->n_owned ? loopback : p->n_owned) : 0u;
	const uint32_t n_ghost = p->ghost_in[0] + p->ghost_in[1];
	const uint32_t bound = m + p->ghost_out[0] + p->ghost_out[1] + n_ghost;
	if (!bound) return NH_OK;
	if (!p->split_stream) { NP_CHECK(ctx, hipStreamCreateWithFlags(&p->split_stream, hipStreamNonBlocking)); NP_CHECK(ctx, hipEventCreateWithFlags(&p->split_done, hipEventDisableTiming)); }
	if (p->split_mask_cap < p->n_bodies || p->split_list_cap < bound || !p->split_count) {
		NP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); NP_CHECK(ctx, hipStreamSynchronize(p->split_stream));
		if (p->split_mask) (void)hipFree(p->split_mask); if (p->split_list) (void)hipFree(p->split_list);
		p->split_mask = nullptr; p->split_list = nullptr; p->split_mask_cap = p->split_list_cap = 0;
		NP_CHECK(ctx, hipMalloc((void**)&p->split_mask, (size_t)p->cfg.body_capacity + 64u)); p->split_mask_cap = p->cfg.body_capacity;
		NP_CHECK(ctx, hipMalloc((void**)&p->split_list, sizeof(uint32_t) * ((size_t)bound + bound / 2u + 64u))); p->split_list_cap = bound + bound / 2u;
		if (!p->split_count) NP_CHECK(ctx, hipMalloc((void**)&p->split_count, 64));
	}
	if (m) { int rc = part_loop_buffers(p, m); if (rc) return rc; }
	for (int side = 0; side < 2; ++side) {
		if (p->ghost_out[side]) { int rc = part_step_buffer(p, side, 0, p->ghost_out[side]); if (rc) return rc; }
		if (p->ghost_in[side]) { int rc = part_step_buffer(p, side, 1, p->ghost_in[side]); if (rc) return rc; }
	}
	NP_CHECK(ctx, hipMemsetAsync(p->split_mask, 0, p->n_bodies, ctx->stream));
	NP_CHECK(ctx, hipMemsetAsync(p->split_count, 0, sizeof(uint32_t), ctx->stream));
	if (m) hipLaunchKernelGGL(k_part_mark_range, dim3(nh_grid_for(m, 256, 256)), dim3(256), 0, ctx->stream, 1u, m, p->n_bodies, p->split_mask);
	for (int side = 0; side < 2; ++side) if (p->ghost_out[side])
		hipLaunchKernelGGL(k_part_mark_list, dim3(nh_grid_for(p->ghost_out[side], 256, 256)), dim3(256), 0, ctx->stream, p->list[side], p->ghost_out[side], p->n_bodies, p->split_mask);
	if (n_ghost) hipLaunchKernelGGL(k_part_mark_range, dim3(nh_grid_for(n_ghost, 256, 256)), dim3(256), 0, ctx->stream, p->n_owned + 1u, n_ghost, p->n_bodies, p->split_mask);
	hipLaunchKernelGGL(k_part_mask_to_list, dim3(nh_grid_for(p->n_bodies, 256, 1024)), dim3(256), 0, ctx->stream, p->split_mask, p->n_bodies, p->split_list, p->split_list_cap, p->split_count);
	ctx->halo_split.stream = p->split_stream; ctx->halo_split.list = p->split_list; ctx->halo_split.count = p->split_count; ctx->halo_split.mask = p->split_mask; ctx->halo_split.bound = bound;
	ctx->halo_split.after_verdict = part_after_verdict; ctx->halo_split.user = p;
	ctx->halo_split.armed = true;
	return NH_OK;
}

static void part_xform_ghosts(nh_partition* p, hipStream_t stream) {
	nh_context* ctx = p->ctx;
	const nh_BodyData* bodies = p->step_args->bodies;
	const nh_ColliderData* cd = p->step_args->colliders;
	const uint32_t n_ghost = p->ghost_in[0] + p->ghost_in[1], first = p->n_owned + 1u;
	hipLaunchKernelGGL(k_xform_ghosts, dim3(nh_grid_for(n_ghost, 256, 256)), dim3(256), 0, stream, ctx->d_state, first, bodies->count, bodies->transforms, bodies->idle_counters,
	                   p->nbx, p->nsp, p->flag[0], p->flag[1], p->cfg.n_static_box, p->cfg.n_static_sph, cd->boxes.count,
	                   cd->boxes.transforms, cd->boxes.data, cd->spheres.transforms, cd->spheres.data,
	                   reinterpret_cast<float4*>(ctx->own_xf), ctx->own_aabb_min, ctx->own_aabb_max, ctx->fat_box, ctx->step_parity ^ 1u, ctx->collide_seq + 1u);
}

// the verdict of a pair step whose solver went out in two parts is in, and it is "happened": the next sub-step's exchange, behind the boundary launch on its stream
static int part_after_verdict(nh_context* ctx, void* user) {
	nh_partition* p = static_cast<nh_partition*>(user);
	const nh_BodyData* bodies = p->step_args->bodies;
	hipStream_t s = p->split_stream;
	const uint32_t m = p->step_loopback ? (p->step_loopback < p->n_owned ? p->step_loopback : p->n_owned) : 0u;
	const uint32_t n_ghost = p->ghost_in[0] + p->ghost_in[1];
	int nccl_rc = 0;
	if (p->ghost_out[0] | p->ghost_out[1] | n_ghost) {
		for (int side = 0; side < 2; ++side) nh_halo_pack_on(s, bodies, p->list[side], p->ghost_out[side], p->step_out[side]);
		nccl_rc = p->rccl.group_start();
		if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
		for (int side = 0; side < 2 && !nccl_rc; ++side) {
			const int peer = p->rccl.peer[side];
			if (peer < 0) continue;
			if (p->ghost_out[side]) nccl_rc = p->rccl.send(p->step_out[side], (size_t)p->ghost_out[side] * NH_HALO_RECORD_BYTES, 1, peer, p->rccl.comm, s);
			if (!nccl_rc && p->ghost_in[side]) nccl_rc = p->rccl.recv(p->step_in[side], (size_t)p->ghost_in[side] * NH_HALO_RECORD_BYTES, 1, peer, p->rccl.comm, s);
		}
		{ const int end_rc = p->rccl.group_end(); if (!nccl_rc) nccl_rc = end_rc; }
		if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
		uint32_t at = p->n_owned + 1u;
		nh_halo_update_on(s, bodies, at, p->ghost_in[0], p->step_in[0]); at += p->ghost_in[0];
		nh_halo_update_on(s, bodies, at, p->ghost_in[1], p->step_in[1]);
	}
	if (m) {
		nh_halo_pack_on(s, bodies, p->loop_idx, m, p->loop_out);
		nccl_rc = p->rccl.group_start();
		if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
		const int self_peer = p->rccl.peer[1] >= 0 ? p->rccl.peer[1] : (int)p->cfg.rank;
		nccl_rc = p->rccl.send(p->loop_out, (size_t)m * NH_HALO_RECORD_BYTES, 1, self_peer, p->rccl.comm, s);
		if (!nccl_rc) nccl_rc = p->rccl.recv(p->loop_in, (size_t)m * NH_HALO_RECORD_BYTES, 1, self_peer, p->rccl.comm, s);
		{ const int end_rc = p->rccl.group_end(); if (!nccl_rc) nccl_rc = end_rc; }
		if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
		nh_halo_update_on(s, bodies, 1u, m, p->loop_in);
	}
	if (ctx->still.ahead_ready && n_ghost && ctx->own_xf) part_xform_ghosts(p, s);
	NP_CHECK(ctx, hipEventRecord(p->split_done, s));
	p->exchanged_ahead = true; p->overlapped_exchanges++;
	return NH_OK;
}

static int part_step_hook(nh_context* ctx, void* user, uint32_t sub_step) {
	nh_partition* p = static_cast<nh_partition*>(user);
	const nh_BodyData* bodies = p->step_args->bodies;
	if (p->exchanged_ahead) {
		// (halo split: this sub-step's halo left while the last one's interior was being solved -- the context's stream waits for it to have landed, that is all)
		p->exchanged_ahead = false;
		NP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, p->split_done, 0));
		return NH_OK;
	}
	// what the step before this one left for it (xform ahead): known BEFORE the exchange's entry points run -- they complete deferred work and look like a caller from outside
	const bool ahead = ctx->still.ahead_ready, pair = ctx->still.pair_ready;
	if (sub_step > 0u || p->step_exchange_first) {
		if (p->ghost_out[0] | p->ghost_out[1] | p->ghost_in[0] | p->ghost_in[1]) { int rc = nh_partition_exchange_step(p, bodies); if (rc) return rc; }
		if (p->step_loopback && p->rccl.comm) {
			// one-rank rehearsal: the first `step_loopback` owned bodies' records through the transport to this same rank and back onto themselves (the same bits)
			const uint32_t m = p->step_loopback < p->n_owned ? p->step_loopback : p->n_owned;
			{ int rc = part_loop_buffers(p, m); if (rc) return rc; }
			{ int rc = nh_halo_pack(ctx, bodies, p->loop_idx, m, p->loop_out); if (rc) return rc; }
			int nccl_rc = p->rccl.group_start();
			if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
			const int self_peer = p->rccl.peer[1] >= 0 ? p->rccl.peer[1] : (int)p->cfg.rank;          // (the rehearsal's "right neighbour" is this rank itself: RcclDirect.attach(loopback))
			nccl_rc = p->rccl.send(p->loop_out, (size_t)m * NH_HALO_RECORD_BYTES, 1, self_peer, p->rccl.comm, ctx->stream);
			if (!nccl_rc) nccl_rc = p->rccl.recv(p->loop_in, (size_t)m * NH_HALO_RECORD_BYTES, 1, self_peer, p->rccl.comm, ctx->stream);
			{ const int end_rc = p->rccl.group_end(); if (!nccl_rc) nccl_rc = end_rc; }
			if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
			{ int rc = nh_halo_update(ctx, bodies, 1u, m, p->loop_in); if (rc) return rc; }
		}
	}
	// the entry points above are "a caller from outside" to the still-step machinery; the chain of steps goes on all the same: nothing but the ghosts (and, in the
	// rehearsal, bodies that got their own bits back) has changed
	ctx->still.ahead_ready = ahead; ctx->still.pair_ready = pair;
	const uint32_t n_ghost = p->ghost_in[0] + p->ghost_in[1];
	if (ahead && n_ghost && ctx->own_xf) part_xform_ghosts(p, ctx->stream);
	return NH_OK;
}

extern "C" int nh_partition_step(nh_partition* p, const nh_StepArgs* args, uint32_t steps, uint32_t exchange_first, uint32_t loopback_records) {
	if (!p || !args || !args->bodies || !args->colliders) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	const bool needs_transport = (p->ghost_out[0] | p->ghost_out[1] | p->ghost_in[0] | p->ghost_in[1] | loopback_records) != 0u;
	if (needs_transport && !p->rccl.comm) return NH_ERR_INVALID;          // (the library enqueues the exchange itself: nh_partition_set_transport first)
	if (args->bodies->count != p->n_bodies) return NH_ERR_INVALID;
	p->step_args = args; p->step_exchange_first = exchange_first; p->step_loopback = loopback_records;
	ctx->step_hook = part_step_hook; ctx->step_hook_user = p;
	ctx->halo_ghost_first = (p->ghost_in[0] + p->ghost_in[1]) ? p->n_owned + 1u : 0u;
	{ int rc = part_arm_split(p, loopback_records); if (rc) { ctx->step_hook = nullptr; ctx->step_hook_user = nullptr; ctx->halo_ghost_first = 0u; p->step_args = nullptr; return rc; } }
	const int rc = nh_step(ctx, args, steps);
	ctx->step_hook = nullptr; ctx->step_hook_user = nullptr; ctx->halo_ghost_first = 0u;
	if (ctx->halo_split.armed) {
		// (the call's last sub-step is never split; an exchange still in flight here belongs to a call that ended in an error)
		if (p->exchanged_ahead || ctx->halo_split.launched) { (void)hipStreamSynchronize(p->split_stream); p->exchanged_ahead = false; }
		ctx->halo_split.armed = false; ctx->halo_split.launched = false;
	}
	p->step_args = nullptr;
	return rc;
}

// transport check: `bytes` of a pattern to both peers and from both (a peer equal to this rank: a loop-back) through the functions handed in, one group; 0 = what
// came back is what each peer sent (the pattern carries the sender's rank).  Synchronises the stream; the host decides what a time-out is (it can poll an event instead: see bench.py).
__global__ void k_part_pattern(uint32_t* buf, uint32_t n, uint32_t rank) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = (rank << 24) ^ (i * 2654435761u); }
__global__ void k_part_pattern_check(const uint32_t* buf, uint32_t n, uint32_t rank, nh_PartDev* dv) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) if (buf[i] != ((rank << 24) ^ (i * 2654435761u))) dv->error = NH_ERR_INVALID; }
extern "C" int nh_partition_transport_check(nh_partition* p, uint32_t bytes, int enqueue_only) {
	if (!p || !p->rccl.comm || bytes < 4u) return NH_ERR_INVALID;
	nh_context* ctx = p->ctx;
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	const uint32_t records = (bytes + NH_HALO_RECORD_BYTES - 1u) / NH_HALO_RECORD_BYTES, words = bytes / 4u;
	for (int side = 0; side < 2; ++side) if (p->rccl.peer[side] >= 0) {
		{ int rc = part_step_buffer(p, side, 0, records); if (rc) return rc; }
		{ int rc = part_step_buffer(p, side, 1, records); if (rc) return rc; }
	}
	hipLaunchKernelGGL(k_part_reset, dim3(1), dim3(1), 0, ctx->stream, p->d);
	for (int side = 0; side < 2; ++side) if (p->rccl.peer[side] >= 0)
		hipLaunchKernelGGL(k_part_pattern, dim3(64), dim3(256), 0, ctx->stream, (uint32_t*)p->step_out[side], words, p->cfg.rank);
	int nccl_rc = p->rccl.group_start();
	if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
	for (int side = 0; side < 2 && !nccl_rc; ++side) if (p->rccl.peer[side] >= 0) {
		nccl_rc = p->rccl.send(p->step_out[side], (size_t)words * 4u, 1, p->rccl.peer[side], p->rccl.comm, ctx->stream);
		if (!nccl_rc) nccl_rc = p->rccl.recv(p->step_in[side], (size_t)words * 4u, 1, p->rccl.peer[side], p->rccl.comm, ctx->stream);
	}
	{ const int end_rc = p->rccl.group_end(); if (!nccl_rc) nccl_rc = end_rc; }          // (closed whatever happened inside: see nh_partition_exchange_step)
	if (nccl_rc) { ctx->last_hip_error = -nccl_rc; return NH_ERR_HIP; }
	for (int side = 0; side < 2; ++side) if (p->rccl.peer[side] >= 0)
		hipLaunchKernelGGL(k_part_pattern_check, dim3(64), dim3(256), 0, ctx->stream, (const uint32_t*)p->step_in[side], words, (uint32_t)p->rccl.peer[side], p->d);
	if (enqueue_only) return NH_OK;            // (the caller polls the stream and then calls nh_partition_transport_result)
	return part_read_back(p);
}
extern "C" int nh_partition_transport_result(nh_partition* p) { return p ? part_read_back(p) : NH_ERR_INVALID; }
