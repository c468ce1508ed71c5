// nh_partition.hip -- the multi-GPU path behind the C ABI (include/nudge_hip.h, "multi-GPU"): one x-slab of a world per context; migration, ghost
// listing, ghost installation + collider rebuild, per-step halo records and cut re-balancing as kernels on the context's stream with device-side counts.
// The reference has no counterpart (single-threaded); what makes a partitioned world reproduce the single one is that a contact's identity
// is its pair of collider TAGS (nudge.h:86, 93; nudge.cpp:2074-2087), which travel with the body.  Transport is the caller's (RCCL, torch.distributed).
//
// Order of things in a rank's arrays: body 0 (static world) | owned bodies, slots 1 .. n_owned | ghosts from the left neighbour | ghosts from the right one;
// colliders: the static ones the rank keeps | one per dynamic body, in slot order (boxes and spheres each in their own array).
#include "nh_internal.h"
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

struct nh_FullRecord { nh_Transform xf; nh_BodyProperties props; nh_BodyMomentum momentum; float shape[4]; nh_Transform cxf; uint32_t tag; uint8_t kind, idle, pad0, pad1; };
static_assert(sizeof(nh_FullRecord) == NH_PARTITION_FULL_BYTES, "full record layout");

struct nh_partition {
	nh_context* ctx;
	nh_PartitionConfig cfg;
	double lo, hi, cut0[2];
	uint32_t n_owned, n_bodies, n_boxes, n_spheres;
	uint32_t ghost_out[2], ghost_in[2];
	uint64_t migrated_out, migrated_in, refreshes, cut_moves;
	// body-parallel description of the one collider a dynamic body carries (device, body_capacity entries)
	float4* shape; nh_Transform* cxf; uint32_t* tag; uint8_t* kind; float* reach;
	// lists of the owned slots sent to the left / right neighbour every step (written by pack_ghosts)
	uint32_t* list[2];
	// scratch: two flag / scan arrays of body_capacity + 1 words, scan temporaries, sort buffers, a staging copy of the body records for the compaction
	uint32_t* flag[2]; uint32_t* scan_tmp; uint32_t* sort_a; uint32_t* sort_b; uint32_t* sort_va; uint32_t* sort_vb; uint32_t* sort_hist;
	nh_FullRecord* stage;
	nh_PartDev* d; nh_PartDev* h;
	bool leavers_flagged;
};

#define NP_CHECK(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (ctx)->last_hip_error = (int)e_; return NH_ERR_HIP; } } while (0)

// ---- kernels ------------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float np_reach(const nh_Transform& c, const float4& s, uint32_t kind) {
	const float off = sqrtf(c.position[0] * c.position[0] + c.position[1] * c.position[1] + c.position[2] * c.position[2]);
	const float ext = kind == 0u ? sqrtf(s.x * s.x + s.y * s.y + s.z * s.z) : s.x;
	return (off + ext) * 1.0001f;
}

// creation: the collider arrays (static ones first, then one per dynamic body with Transform.body = slot) -> the body-parallel description
__global__ __launch_bounds__(256) void k_part_adopt(uint32_t n_static, uint32_t n, uint32_t which, const uint32_t* __restrict__ tags, const void* __restrict__ data, const nh_Transform* __restrict__ xf,
                                                    uint32_t body_capacity, float4* __restrict__ shape, nh_Transform* __restrict__ cxf, uint32_t* __restrict__ tag, uint8_t* __restrict__ kind,
                                                    float* __restrict__ reach, nh_PartDev* dv) {
	for (uint32_t c = n_static + blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		const nh_Transform t = xf[c];
		const uint32_t b = t.body;
		if (b == 0u || b >= body_capacity) { dv->error = NH_ERR_INVALID; continue; }
		float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (which == 0u) { const nh_BoxCollider bc = reinterpret_cast<const nh_BoxCollider*>(data)[c]; s = make_float4(bc.size[0], bc.size[1], bc.size[2], bc.unused); }
		else s.x = reinterpret_cast<const nh_SphereCollider*>(data)[c].radius;
		shape[b] = s; cxf[b] = t; tag[b] = tags[c]; kind[b] = (uint8_t)which; reach[b] = np_reach(t, s, which);
	}
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

__device__ __forceinline__ nh_FullRecord np_full(uint32_t i, const nh_Transform* bt, const nh_BodyProperties* bp, const nh_BodyMomentum* bm, const uint8_t* bi,
                                                  const float4* shape, const nh_Transform* cxf, const uint32_t* tag, const uint8_t* kind) {
	nh_FullRecord r;
	r.xf = bt[i]; r.props = bp[i]; r.momentum = bm[i];
	const float4 s = shape[i];
	r.shape[0] = s.x; r.shape[1] = s.y; r.shape[2] = s.z; r.shape[3] = s.w;
	r.cxf = cxf[i]; r.tag = tag[i]; r.kind = kind[i]; r.idle = bi[i]; r.pad0 = 0; r.pad1 = 0;
	return r;
}

// ordered compaction of the flagged bodies' full records (sl / sr: the exclusive scans of the flags, n + 1 entries); optionally the slot lists too
__global__ __launch_bounds__(256) void k_part_pack_full(uint32_t n, const uint32_t* __restrict__ sl, const uint32_t* __restrict__ sr, uint32_t capacity,
                                                        const nh_Transform* __restrict__ bt, const nh_BodyProperties* __restrict__ bp, const nh_BodyMomentum* __restrict__ bm, const uint8_t* __restrict__ bi,
                                                        const float4* __restrict__ shape, const nh_Transform* __restrict__ cxf, const uint32_t* __restrict__ tag, const uint8_t* __restrict__ kind,
                                                        nh_FullRecord* __restrict__ out_l, nh_FullRecord* __restrict__ out_r, uint32_t* __restrict__ list_l, uint32_t* __restrict__ list_r, nh_PartDev* dv) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t pl = sl[i], pr = sr[i];
		const bool l = sl[i + 1] != pl, r = sr[i + 1] != pr;
		if (!l && !r) continue;
		const nh_FullRecord rec = np_full(i, bt, bp, bm, bi, shape, cxf, tag, kind);
		if (l) { if (pl < capacity && out_l) out_l[pl] = rec; else dv->error = NH_ERR_INVALID; if (list_l) list_l[pl] = i; }
		if (r) { if (pr < capacity && out_r) out_r[pr] = rec; else dv->error = NH_ERR_INVALID; if (list_r) list_r[pr] = i; }
	}
}

// the bodies that stay, in slot order, into the staging buffer (keep position = i - leavers before i); then back
__global__ __launch_bounds__(256) void k_part_stage_kept(uint32_t n, const uint32_t* __restrict__ sl, const uint32_t* __restrict__ sr,
                                                         const nh_Transform* __restrict__ bt, const nh_BodyProperties* __restrict__ bp, const nh_BodyMomentum* __restrict__ bm, const uint8_t* __restrict__ bi,
                                                         const float4* __restrict__ shape, const nh_Transform* __restrict__ cxf, const uint32_t* __restrict__ tag, const uint8_t* __restrict__ kind,
                                                         nh_FullRecord* __restrict__ stage) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t pl = sl[i], pr = sr[i];
		if (sl[i + 1] != pl || sr[i + 1] != pr) continue;
		stage[i - pl - pr] = np_full(i, bt, bp, bm, bi, shape, cxf, tag, kind);
	}
}

// full records -> body slots [at, at + m): the world's arrays and the body-parallel description (reach follows from the collider)
__global__ __launch_bounds__(256) void k_part_install(uint32_t at, uint32_t m, const nh_FullRecord* __restrict__ in,
                                                      nh_Transform* __restrict__ bt, nh_BodyProperties* __restrict__ bp, nh_BodyMomentum* __restrict__ bm, uint8_t* __restrict__ bi,
                                                      float4* __restrict__ shape, nh_Transform* __restrict__ cxf, uint32_t* __restrict__ tag, uint8_t* __restrict__ kind, float* __restrict__ reach) {
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
		const nh_FullRecord r = in[k];
		const uint32_t i = at + k;
		bt[i] = r.xf; bp[i] = r.props; bm[i] = r.momentum; bi[i] = r.idle;
		const float4 s = make_float4(r.shape[0], r.shape[1], r.shape[2], r.shape[3]);
		shape[i] = s; cxf[i] = r.cxf; tag[i] = r.tag; kind[i] = r.kind; reach[i] = np_reach(r.cxf, s, r.kind);
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
__global__ __launch_bounds__(256) void k_part_flag_ghosts(uint32_t n, const nh_Transform* __restrict__ bt, const float* __restrict__ reach, float lo, float hi, uint32_t has_left, uint32_t has_right,
                                                          double max_reach, double epoch_dt, double speed_floor, double gravity_epoch_dt, const nh_PartDev* __restrict__ dv,
                                                          uint32_t* __restrict__ fl, uint32_t* __restrict__ fr) {
	const double vmax = (double)__uint_as_float(dv->vmax_bits);
	const double drift = epoch_dt * ((vmax > speed_floor ? vmax : speed_floor) + gravity_epoch_dt);
	const float extra = (float)(max_reach + 2.0 * drift);
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
		uint32_t l = 0u, r = 0u;
		if (i >= 1u && i < n) {
			const float x = bt[i].position[0];
			const float margin = reach[i] + extra;
			l = (has_left && x - margin < lo) ? 1u : 0u;
			r = (has_right && x + margin >= hi) ? 1u : 0u;
		}
		fl[i] = l; fr[i] = r;
	}
}

// collider rebuild: flags by kind over the dynamic bodies, then the ordered write behind the static colliders
__global__ __launch_bounds__(256) void k_part_flag_kinds(uint32_t n, const uint8_t* __restrict__ kind, uint32_t* __restrict__ f0, uint32_t* __restrict__ f1) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
		const uint32_t k = (i >= 1u && i < n) ? kind[i] : 255u;
		f0[i] = k == 0u ? 1u : 0u; f1[i] = k == 1u ? 1u : 0u;
	}
}

__global__ __launch_bounds__(256) void k_part_write_colliders(uint32_t n, const uint32_t* __restrict__ s0, const uint32_t* __restrict__ s1, const uint8_t* __restrict__ kind,
                                                              const float4* __restrict__ shape, const nh_Transform* __restrict__ cxf, const uint32_t* __restrict__ tag,
                                                              uint32_t n_static_box, uint32_t box_capacity, uint32_t* __restrict__ xt, nh_BoxCollider* __restrict__ xd, nh_Transform* __restrict__ xx,
                                                              uint32_t n_static_sph, uint32_t sph_capacity, uint32_t* __restrict__ st, nh_SphereCollider* __restrict__ sd, nh_Transform* __restrict__ sx,
                                                              nh_PartDev* dv) {
	for (uint32_t i = 1u + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t k = kind[i];
		nh_Transform t = cxf[i];
		t.body = i;                                  // Transform.body of a collider = its body's slot (nudge.h:36)
		const float4 s = shape[i];
		if (k == 0u) {
			const uint32_t c = n_static_box + s0[i];
			if (c >= box_capacity) { dv->error = NH_ERR_INVALID; continue; }
			nh_BoxCollider b = { { s.x, s.y, s.z }, s.w };
			xt[c] = tag[i]; xd[c] = b; xx[c] = t;
		} else if (k == 1u) {
			const uint32_t c = n_static_sph + s1[i];
			if (c >= sph_capacity) { dv->error = NH_ERR_INVALID; continue; }
			nh_SphereCollider sp = { s.x };
			st[c] = tag[i]; sd[c] = sp; sx[c] = t;
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
	if ((uint64_t)(colliders->boxes.count - cfg->n_static_box) + (colliders->spheres.count - cfg->n_static_sph) != cfg->n_owned) return NH_ERR_INVALID;      // one collider per dynamic body
	NP_CHECK(ctx, hipSetDevice(ctx->device));
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	nh_partition* p = new nh_partition();
	memset(p, 0, sizeof(*p));
	p->ctx = ctx; p->cfg = *cfg; p->lo = cfg->lo; p->hi = cfg->hi; p->cut0[0] = cfg->lo; p->cut0[1] = cfg->hi;
	p->n_owned = cfg->n_owned; p->n_bodies = cfg->n_owned + 1u; p->n_boxes = colliders->boxes.count; p->n_spheres = colliders->spheres.count;
	const size_t cap = cfg->body_capacity;
	const size_t sort_words = 256u * NH_SORT_GRID + 512u;
	bool ok = hipMalloc((void**)&p->shape, sizeof(float4) * cap) == hipSuccess && hipMalloc((void**)&p->cxf, sizeof(nh_Transform) * cap) == hipSuccess &&
	          hipMalloc((void**)&p->tag, sizeof(uint32_t) * cap) == hipSuccess && hipMalloc((void**)&p->kind, cap + 64u) == hipSuccess && hipMalloc((void**)&p->reach, sizeof(float) * cap) == hipSuccess &&
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
	hipMemsetAsync(p->kind, 0xFF, cap, s);
	hipMemsetAsync(p->reach, 0, sizeof(float) * cap, s);
	hipMemsetAsync(p->shape, 0, sizeof(float4) * cap, s);
	hipMemsetAsync(p->cxf, 0, sizeof(nh_Transform) * cap, s);
	hipMemsetAsync(p->tag, 0, sizeof(uint32_t) * cap, s);
	const uint32_t nbx = colliders->boxes.count, nsp = colliders->spheres.count;
	if (nbx > cfg->n_static_box)
		hipLaunchKernelGGL(k_part_adopt, dim3(nh_grid_for(nbx - cfg->n_static_box, 256, 2048)), dim3(256), 0, s, cfg->n_static_box, nbx, 0u, colliders->boxes.tags, (const void*)colliders->boxes.data,
		                   colliders->boxes.transforms, cfg->body_capacity, p->shape, p->cxf, p->tag, p->kind, p->reach, p->d);
	if (nsp > cfg->n_static_sph)
		hipLaunchKernelGGL(k_part_adopt, dim3(nh_grid_for(nsp - cfg->n_static_sph, 256, 2048)), dim3(256), 0, s, cfg->n_static_sph, nsp, 1u, colliders->spheres.tags, (const void*)colliders->spheres.data,
		                   colliders->spheres.transforms, cfg->body_capacity, p->shape, p->cxf, p->tag, p->kind, p->reach, p->d);
	int rc = part_read_back(p);
	if (rc) { nh_partition_destroy(p); return rc; }
	*out = p;
	return NH_OK;
}

extern "C" void nh_partition_destroy(nh_partition* p) {
	if (!p) return;
	hipSetDevice(p->ctx->device);
	hipStreamSynchronize(p->ctx->stream);
	void* bufs[] = { p->shape, p->cxf, p->tag, p->kind, p->reach, p->list[0], p->list[1], p->flag[0], p->flag[1], p->scan_tmp, p->sort_a, p->sort_b, p->sort_va, p->sort_vb, p->sort_hist, p->stage, p->d };
	for (void* b : bufs) if (b) hipFree(b);
	if (p->h) hipHostFree(p->h);
	delete p;
}

extern "C" int nh_partition_info(nh_partition* p, nh_PartitionInfo* out) {
	if (!p || !out) return NH_ERR_INVALID;
	out->n_owned = p->n_owned; out->n_bodies = p->n_bodies; out->n_boxes = p->n_boxes; out->n_spheres = p->n_spheres;
	out->ghost_out[0] = p->ghost_out[0]; out->ghost_out[1] = p->ghost_out[1]; out->ghost_in[0] = p->ghost_in[0]; out->ghost_in[1] = p->ghost_in[1];
	out->lo = p->lo; out->hi = p->hi;
	out->migrated_out = p->migrated_out; out->migrated_in = p->migrated_in; out->refreshes = p->refreshes; out->cut_moves = p->cut_moves;
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
	          p->shape, p->cxf, p->tag, p->kind, (nh_FullRecord*)out_left, (nh_FullRecord*)out_right, (uint32_t*)nullptr, (uint32_t*)nullptr, p->d);
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
		NP_LAUNCH("part_stage_kept", k_part_stage_kept, n, n, p->flag[0], p->flag[1], bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, p->shape, p->cxf, p->tag, p->kind, p->stage);
		NP_LAUNCH("part_install", k_part_install, keep, 0u, keep, p->stage, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, p->shape, p->cxf, p->tag, p->kind, p->reach);
	}
	uint32_t at = keep;
	if (n_left) { NP_LAUNCH("part_install", k_part_install, n_left, at, n_left, (const nh_FullRecord*)in_left, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, p->shape, p->cxf, p->tag, p->kind, p->reach); at += n_left; }
	if (n_right) { NP_LAUNCH("part_install", k_part_install, n_right, at, n_right, (const nh_FullRecord*)in_right, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, p->shape, p->cxf, p->tag, p->kind, p->reach); at += n_right; }
	p->n_owned = at - 1u;
	p->n_bodies = at;                       // (ghosts are gone until nh_partition_unpack_ghosts)
	p->ghost_in[0] = p->ghost_in[1] = 0u;
	p->migrated_in += n_left + n_right;
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
	          p->cfg.max_reach, epoch_dt, p->cfg.speed_floor, fabs(p->cfg.gravity) * epoch_dt, p->d, p->flag[0], p->flag[1]);
	nh_scan2_u32(ctx, p->flag[0], p->flag[0], &p->d->count[0], p->flag[1], p->flag[1], &p->d->count[1], &p->d->zero, n + 1u, p->scan_tmp);
	NP_LAUNCH("part_pack_full", k_part_pack_full, n, n, p->flag[0], p->flag[1], capacity_records, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters,
	          p->shape, p->cxf, p->tag, p->kind, (nh_FullRecord*)out_left, (nh_FullRecord*)out_right, p->list[0], p->list[1], p->d);
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
	if (n_left) { NP_LAUNCH("part_install", k_part_install, n_left, at, n_left, (const nh_FullRecord*)in_left, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, p->shape, p->cxf, p->tag, p->kind, p->reach); at += n_left; }
	if (n_right) { NP_LAUNCH("part_install", k_part_install, n_right, at, n_right, (const nh_FullRecord*)in_right, bodies->transforms, bodies->properties, bodies->momentum, bodies->idle_counters, p->shape, p->cxf, p->tag, p->kind, p->reach); at += n_right; }
	p->ghost_in[0] = n_left; p->ghost_in[1] = n_right;
	p->n_bodies = at;
	// colliders: the static ones stay, then one per dynamic body slot (owned, then ghosts) -- boxes and spheres each compacted in slot order
	NH_LAUNCH(ctx, "part_reset", k_part_reset, 1, 1, p->d);
	NP_LAUNCH("part_flag_kinds", k_part_flag_kinds, at, at, p->kind, p->flag[0], p->flag[1]);
	nh_scan2_u32(ctx, p->flag[0], p->flag[0], &p->d->kinds[0], p->flag[1], p->flag[1], &p->d->kinds[1], &p->d->zero, at + 1u, p->scan_tmp);
	NP_LAUNCH("part_write_colliders", k_part_write_colliders, at, at, p->flag[0], p->flag[1], p->kind, p->shape, p->cxf, p->tag,
	          p->cfg.n_static_box, p->cfg.box_capacity, colliders->boxes.tags, colliders->boxes.data, colliders->boxes.transforms,
	          p->cfg.n_static_sph, p->cfg.sphere_capacity, colliders->spheres.tags, colliders->spheres.data, colliders->spheres.transforms, p->d);
	int rc = part_read_back(p);
	if (rc) return rc;
	if (p->h->kinds[0] + p->h->kinds[1] != at - 1u) return NH_ERR_INVALID;            // a dynamic body without a collider of its own
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
