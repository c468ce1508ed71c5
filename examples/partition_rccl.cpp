// examples/partition_rccl.cpp -- C++ host of the MULTI-GPU path: one x-slab of a world per rank, the slab logic behind the C ABI
// (include/nudge_hip.h "multi-GPU": nh_partition_* -- migration, ghost listing / installation, collider rebuild, per-step halo records, cut re-balancing, all
// kernels with device-side counts), and THIS program as the transport: RCCL point-to-point (ncclSend / ncclRecv over xGMI) between one process per GPU, or
// device-to-device copies between several ranks held by one process on one GPU (NUDGE_LOCAL_RANKS=R: what a one-GPU box can run of it, and what the tests
// compare with the Python host nudge_amd/partition.py bit for bit -- both are thin transports over the same entry points).
//
//   every `epoch` steps:  [re-balance: the two ranks at a cut exchange their body counts; the heavier one calls nh_partition_choose_cut and sends the value]
//                         nh_partition_pack_migrants -> exchange -> nh_partition_unpack_migrants
//                         nh_partition_pack_ghosts   -> exchange -> nh_partition_unpack_ghosts
//   every other step:     nh_partition_pack_step     -> exchange -> nh_partition_unpack_step
//   then the reference's sub-step on the local world (example/main.cpp:274-328): the eight calls of the C ABI.
// No collective on the data path; per refresh one host synchronisation per rank (the message lengths), none on the other steps.
//
// The world: a scene file written by tests/partition_util.py (`--scene FILE --cuts x1,x2,... STEPS`), or the drop-tile world of BASELINE.json configs[4] from a
// counter-based generator (`COLS_PER_RANK ROWS SIDE STEPS`: every rank generates the same world and keeps its slab).
//
// build:  make -C nudge_amd/csrc partition_rccl
// run:    one process per GPU:   for r in 0 .. 7: RANK=$r WORLD_SIZE=8 LOCAL_RANK=$r NUDGE_NCCL_ID_FILE=/tmp/nudge.id nudge_amd/partition_rccl 6 46 90 200 &
//         one GPU, three ranks:  NUDGE_LOCAL_RANKS=3 nudge_amd/partition_rccl 2 2 24 120
//         NUDGE_RCCL_LOOPBACK=1 (WORLD_SIZE=1): the per-step records of the first bodies also travel ncclSend(self) / ncclRecv(self) -- RCCL executed on one GPU.
#include "../include/nudge_hip.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <unistd.h>
#include <vector>

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); exit(3); } } while (0)
#define NH_OK_(expr) do { int rc_ = (expr); if (rc_) { fprintf(stderr, "%s: %s\n", #expr, nh_error_string(rc_)); exit(4); } } while (0)
#define NCCL_OK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #expr, ncclGetErrorString(r_)); exit(5); } } while (0)

// ---- the whole world on the host (what nudge_amd/scenes.py builds; layouts of nudge_hip.h) ----------------------------------------------------------------
struct Scene {
	std::vector<nh_Transform> body_xf; std::vector<nh_BodyProperties> body_props; std::vector<nh_BodyMomentum> body_mom; std::vector<uint8_t> idle;
	std::vector<uint32_t> box_tags; std::vector<nh_BoxCollider> box_data; std::vector<nh_Transform> box_xf;
	std::vector<uint32_t> sph_tags; std::vector<nh_SphereCollider> sph_data; std::vector<nh_Transform> sph_xf;
	double time_step = 1.0 / 120.0, gravity = 9.82, damping_rate = 0.25;
	uint32_t iterations = 8;
};

template<class T> static void read_vec(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "scene file truncated\n"); exit(2); } }

// file layout (tests/partition_util.py: write_scene_file): "NHSC" | u32 version = 1, nb, nbox, nsph, iterations | f64 time_step, gravity, damping_rate | the ten arrays
static Scene load_scene(const char* path) {
	FILE* f = fopen(path, "rb");
	if (!f) { perror(path); exit(2); }
	char magic[4]; uint32_t head[5]; double par[3];
	if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "NHSC", 4) || fread(head, 4, 5, f) != 5 || head[0] != 1u || fread(par, 8, 3, f) != 3) { fprintf(stderr, "%s: not a scene file\n", path); exit(2); }
	Scene s;
	const size_t nb = head[1], nbox = head[2], nsph = head[3];
	s.iterations = head[4]; s.time_step = par[0]; s.gravity = par[1]; s.damping_rate = par[2];
	read_vec(f, s.body_xf, nb); read_vec(f, s.body_props, nb); read_vec(f, s.body_mom, nb); read_vec(f, s.idle, nb);
	read_vec(f, s.box_tags, nbox); read_vec(f, s.box_data, nbox); read_vec(f, s.box_xf, nbox);
	read_vec(f, s.sph_tags, nsph); read_vec(f, s.sph_data, nsph); read_vec(f, s.sph_xf, nsph);
	fclose(f);
	return s;
}

// counter-based generator (splitmix64): the same body on every rank that generates it
static float u01(uint64_t seed, uint64_t counter) {
	uint64_t x = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	x ^= x >> 31;
	return (float)((double)(x >> 40) * (1.0 / 16777216.0));
}

// cols x rows tiles of side x side boxes on a 3.0 pitch, one ground slab per tile (the shape of BASELINE.json configs[1] / configs[4])
static Scene drop_tiles(int cols, int rows, int side) {
	Scene s;
	const uint64_t seed = 0x9E3779B97F4A7C15ull ^ 5ull;
	const float pitch = 3.0f, half = side * 1.5f + 2.0f, tile_pitch = 2.0f * half + 4.0f;
	const uint32_t n_tiles = (uint32_t)(cols * rows);
	nh_Transform ident = {}; ident.rotation[3] = 1.0f;
	s.body_xf.push_back(ident); s.body_props.push_back(nh_BodyProperties()); s.body_mom.push_back(nh_BodyMomentum()); s.idle.push_back(0);
	for (int tz = 0; tz < rows; ++tz) for (int tx = 0; tx < cols; ++tx) {
		nh_Transform st = ident; st.position[0] = tx * tile_pitch; st.position[1] = -20.0f; st.position[2] = tz * tile_pitch; st.body = 0;
		nh_BoxCollider sb = { { half, 10.0f, half }, 0.0f };
		s.box_xf.push_back(st); s.box_data.push_back(sb); s.box_tags.push_back((uint32_t)(tz * cols + tx));          // static colliders: tags 0 .. n_tiles-1
	}
	for (int tz = 0; tz < rows; ++tz) for (int tx = 0; tx < cols; ++tx) {
		const float ox = tx * tile_pitch, oz = tz * tile_pitch;
		const uint32_t tile = (uint32_t)(tz * cols + tx);
		for (int iz = 0; iz < side; ++iz) for (int ix = 0; ix < side; ++ix) {
			const uint64_t g = (uint64_t)tile * side * side + (uint64_t)iz * side + ix;
			const float sx = 0.5f + 0.5f * u01(seed, 8 * g), sy = 0.5f + 0.5f * u01(seed, 8 * g + 1), sz = 0.5f + 0.5f * u01(seed, 8 * g + 2);
			const float yaw = 0.5f * u01(seed, 8 * g + 3);
			nh_Transform t = ident;
			t.position[0] = ox + (ix - (side - 1) * 0.5f) * pitch + 0.2f * u01(seed, 8 * g + 4);
			t.position[2] = oz + (iz - (side - 1) * 0.5f) * pitch + 0.2f * u01(seed, 8 * g + 5);
			t.position[1] = -10.0f + 0.5f + 0.5f * u01(seed, 8 * g + 6) + sy;
			t.rotation[1] = sinf(0.5f * yaw); t.rotation[3] = cosf(0.5f * yaw);
			const float mass = 8.0f * sx * sy * sz, k = mass / 3.0f;            // add_box of the sample application (example/main.cpp:112-145)
			nh_BodyProperties p;
			p.mass_inverse = 1.0f / mass;
			p.inertia_inverse[0] = 1.0f / (k * sy * sy + k * sz * sz); p.inertia_inverse[1] = 1.0f / (k * sx * sx + k * sz * sz); p.inertia_inverse[2] = 1.0f / (k * sx * sx + k * sy * sy);
			const uint32_t body = (uint32_t)s.body_xf.size();
			s.body_xf.push_back(t); s.body_props.push_back(p); s.body_mom.push_back(nh_BodyMomentum()); s.idle.push_back(0);
			nh_Transform local = ident; local.body = body;
			nh_BoxCollider b = { { sx, sy, sz }, 0.0f };
			s.box_xf.push_back(local); s.box_data.push_back(b); s.box_tags.push_back(n_tiles + (uint32_t)g);
		}
	}
	return s;
}

template<class T>
static T* to_device(const std::vector<T>& v, size_t capacity) {
	const size_t n = capacity > v.size() ? capacity : v.size();
	T* d = nullptr;
	HIP_OK(hipMalloc((void**)&d, sizeof(T) * (n ? n : 1)));
	HIP_OK(hipMemset(d, 0, sizeof(T) * (n ? n : 1)));
	if (!v.empty()) HIP_OK(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
	return d;
}

// ---- one rank: its slab of the scene as a device-resident world + the nh_partition that keeps it ---------------------------------------------------------
struct Rank {
	int rank = 0, ranks = 1;
	nh_context* ctx = nullptr;
	nh_partition* part = nullptr;
	nh_BodyData bodies = {}; nh_ColliderData colliders = {}; nh_BodyConnections connections = { nullptr, 0 };
	nh_ContactData contacts = {}; nh_ContactCache cache = {}; nh_ActiveBodies active = {}; nh_Arena arena = {};
	uint32_t body_capacity = 0, msg_capacity = 0;
	void* out[2] = { nullptr, nullptr }; void* in[2] = { nullptr, nullptr };          // message buffers: [0] left neighbour, [1] right neighbour
	uint32_t out_count[2] = { 0, 0 }, in_count[2] = { 0, 0 };
	bool has(int side) const { return side == 0 ? rank > 0 : rank + 1 < ranks; }
};

// the slab [lo, hi) of `scene` for `rank`: what nudge_amd/partition.py split_scene() builds (every static collider kept; slots in scene order)
static void make_rank(Rank& R, const Scene& s, const std::vector<double>& cuts, int rank, int device, hipStream_t stream, uint32_t epoch, double speed_floor, double max_reach, uint32_t min_extra, double slack,
                      uint32_t context_flags = NH_FLAG_SINGLE_APPLY | NH_FLAG_FUSED_STEP) {
	R.rank = rank; R.ranks = (int)cuts.size() + 1;
	const double lo = rank == 0 ? -HUGE_VAL : cuts[rank - 1], hi = rank + 1 == R.ranks ? HUGE_VAL : cuts[rank];
	const size_t nb = s.body_xf.size();
	std::vector<uint32_t> slot_of(nb, 0u);
	std::vector<nh_Transform> xf(1, s.body_xf[0]); std::vector<nh_BodyProperties> props(1, s.body_props[0]); std::vector<nh_BodyMomentum> mom(1, s.body_mom[0]); std::vector<uint8_t> idle(1, s.idle[0]);
	for (size_t i = 1; i < nb; ++i) {
		const float x = s.body_xf[i].position[0];
		if (!(x >= lo && x < hi)) continue;
		slot_of[i] = (uint32_t)xf.size();
		nh_Transform t = s.body_xf[i]; t.body = (uint32_t)i;                          // the global id rides in the field the engine does not use for bodies (nudge.h:36)
		xf.push_back(t); props.push_back(s.body_props[i]); mom.push_back(s.body_mom[i]); idle.push_back(s.idle[i]);
	}
	const uint32_t n = (uint32_t)xf.size();
	// colliders: the static ones, then one per owned dynamic body in SLOT order (boxes and spheres each in their own array)
	std::vector<uint32_t> bt, stg; std::vector<nh_BoxCollider> bd; std::vector<nh_SphereCollider> sd; std::vector<nh_Transform> bx, sx;
	for (size_t c = 0; c < s.box_xf.size(); ++c) if (s.box_xf[c].body == 0) { bt.push_back(s.box_tags[c]); bd.push_back(s.box_data[c]); bx.push_back(s.box_xf[c]); }
	for (size_t c = 0; c < s.sph_xf.size(); ++c) if (s.sph_xf[c].body == 0) { stg.push_back(s.sph_tags[c]); sd.push_back(s.sph_data[c]); sx.push_back(s.sph_xf[c]); }
	const uint32_t n_static_box = (uint32_t)bt.size(), n_static_sph = (uint32_t)stg.size();
	std::vector<std::pair<uint32_t, size_t>> dyn_box, dyn_sph;          // (slot, collider)
	for (size_t c = 0; c < s.box_xf.size(); ++c) { const uint32_t b = s.box_xf[c].body; if (b && slot_of[b]) dyn_box.push_back({ slot_of[b], c }); }
	for (size_t c = 0; c < s.sph_xf.size(); ++c) { const uint32_t b = s.sph_xf[c].body; if (b && slot_of[b]) dyn_sph.push_back({ slot_of[b], c }); }
	std::sort(dyn_box.begin(), dyn_box.end()); std::sort(dyn_sph.begin(), dyn_sph.end());
	for (auto& e : dyn_box) { nh_Transform t = s.box_xf[e.second]; t.body = e.first; bt.push_back(s.box_tags[e.second]); bd.push_back(s.box_data[e.second]); bx.push_back(t); }
	for (auto& e : dyn_sph) { nh_Transform t = s.sph_xf[e.second]; t.body = e.first; stg.push_back(s.sph_tags[e.second]); sd.push_back(s.sph_data[e.second]); sx.push_back(t); }
	{
		// a dynamic body carries up to NH_PARTITION_MAX_COLLIDERS colliders (compound bodies travel with all of them)
		std::vector<uint8_t> per_body(n, 0);
		for (auto& e : dyn_box) per_body[e.first]++;
		for (auto& e : dyn_sph) per_body[e.first]++;
		for (uint32_t b = 1; b < n; ++b) if (per_body[b] > NH_PARTITION_MAX_COLLIDERS) { fprintf(stderr, "rank %d: a dynamic body carries at most %d colliders\n", rank, NH_PARTITION_MAX_COLLIDERS); exit(2); }
	}
	const uint32_t extra = std::max<uint32_t>(min_extra, (uint32_t)(slack * n));
	R.body_capacity = n + extra;
	const uint32_t box_cap = (uint32_t)bt.size() + NH_PARTITION_MAX_COLLIDERS * extra, sph_cap = (uint32_t)stg.size() + NH_PARTITION_MAX_COLLIDERS * extra;
	uint32_t max_tag = 1;
	for (uint32_t t : s.box_tags) max_tag = std::max(max_tag, t);
	for (uint32_t t : s.sph_tags) max_tag = std::max(max_tag, t);
	uint32_t tag_bits = 8; while (tag_bits < 32 && (1ull << tag_bits) <= max_tag) ++tag_bits;

	HIP_OK(hipSetDevice(device));
	NH_OK_(nh_create(&R.ctx, device, stream, context_flags));
	NH_OK_(nh_set_tag_bits(R.ctx, tag_bits));
	R.bodies = { to_device(xf, R.body_capacity), to_device(props, R.body_capacity), to_device(mom, R.body_capacity), to_device(idle, R.body_capacity), n };
	R.colliders.boxes.tags = to_device(bt, box_cap); R.colliders.boxes.data = to_device(bd, box_cap); R.colliders.boxes.transforms = to_device(bx, box_cap); R.colliders.boxes.count = (uint32_t)bt.size();
	R.colliders.spheres.tags = to_device(stg, sph_cap); R.colliders.spheres.data = to_device(sd, sph_cap); R.colliders.spheres.transforms = to_device(sx, sph_cap); R.colliders.spheres.count = (uint32_t)stg.size();
	const uint32_t K = 8u * R.body_capacity;
	R.contacts.data = to_device(std::vector<nh_Contact>(), K); R.contacts.bodies = to_device(std::vector<nh_BodyPair>(), K); R.contacts.tags = to_device(std::vector<uint64_t>(), K);
	R.contacts.features = to_device(std::vector<uint32_t>(), K); R.contacts.sleeping_pairs = to_device(std::vector<uint64_t>(), K); R.contacts.capacity = K;
	R.cache = { to_device(std::vector<uint64_t>(), K), to_device(std::vector<uint32_t>(), K), to_device(std::vector<nh_CachedContactImpulse>(), K), K, 0 };
	R.active = { to_device(std::vector<uint32_t>(), R.body_capacity), R.body_capacity, 0 };
	const size_t ncol = (size_t)box_cap + sph_cap;
	size_t cells = 1; while (cells < 4 * ncol) cells <<= 1;
	cells = std::min<size_t>(std::max<size_t>(cells, (size_t)1 << 16), (size_t)1 << 24);
	R.arena.size = ((size_t)64 << 20) + ncol * 200 + cells * 8 + (size_t)K * 320 + ((size_t)K / 2 + 1024) * 216 + (size_t)R.body_capacity * 64;
	HIP_OK(hipMalloc(&R.arena.data, R.arena.size));
	nh_PartitionConfig cfg = {};
	cfg.rank = (uint32_t)rank; cfg.ranks = (uint32_t)R.ranks; cfg.lo = lo; cfg.hi = hi; cfg.n_owned = n - 1u; cfg.n_static_box = n_static_box; cfg.n_static_sph = n_static_sph;
	cfg.body_capacity = R.body_capacity; cfg.box_capacity = box_cap; cfg.sphere_capacity = sph_cap; cfg.epoch = epoch;
	cfg.time_step = s.time_step; cfg.gravity = s.gravity; cfg.speed_floor = speed_floor; cfg.max_reach = max_reach; cfg.cut_slack = HUGE_VAL;
	NH_OK_(nh_partition_create(&R.part, R.ctx, &cfg, &R.bodies, &R.colliders));
	R.msg_capacity = std::max<uint32_t>(4096u, R.body_capacity / 4u);
	for (int side = 0; side < 2; ++side) {
		HIP_OK(hipMalloc(&R.out[side], (size_t)NH_PARTITION_FULL_BYTES * R.msg_capacity)); HIP_OK(hipMalloc(&R.in[side], (size_t)NH_PARTITION_FULL_BYTES * R.msg_capacity));
	}
}

// the reference's sub-step on a rank's local world (example/main.cpp:274-328)
static void local_step(Rank& R, const Scene& s) {
	const float dt = (float)s.time_step, gravity[3] = { 0.0f, (float)s.gravity, 0.0f };
	nh_Arena temporary = R.arena;
	NH_OK_(nh_collide(R.ctx, &R.active, &R.contacts, &R.bodies, &R.colliders, &R.connections, temporary));
	NH_OK_(nh_apply_gravity_damping(R.ctx, &R.active, &R.bodies, dt, gravity, (float)s.damping_rate));
	nh_ContactImpulseData* impulses = nullptr;
	NH_OK_(nh_read_cached_impulses(R.ctx, &R.cache, &R.contacts, &temporary, &impulses));
	nh_ContactConstraintData* constraints = nullptr;
	NH_OK_(nh_setup_contact_constraints(R.ctx, &R.active, &R.contacts, &R.bodies, impulses, &temporary, &constraints));
	NH_OK_(nh_apply_impulses(R.ctx, constraints, &R.bodies, s.iterations));
	NH_OK_(nh_update_cached_impulses(R.ctx, constraints, impulses));
	NH_OK_(nh_write_cached_impulses(R.ctx, &R.cache, &R.contacts, impulses));
	NH_OK_(nh_advance(R.ctx, &R.active, &R.bodies, dt));
}

// ---- transport: what rank r's out[side] holds goes to its neighbour's in[1 - side].  LOCAL: device-to-device copies; RCCL: ncclSend / ncclRecv ----------
struct Transport {
	bool local = false;                          // several ranks in this process (one GPU)
	ncclComm_t comm = nullptr; int rank = 0, world = 1;
	hipStream_t stream = nullptr;
	uint64_t* d_cnt = nullptr;                   // RCCL: 4 words on the device (out left / right, in left / right) for lengths and cut values
};

// `known`: the lengths are what the last refresh agreed on (per-step records); otherwise they travel first (one stream synchronisation)
static void exchange(Transport& T, std::vector<Rank>& ranks, size_t record_bytes, bool known) {
	if (T.local) {
		for (Rank& R : ranks) for (int side = 0; side < 2; ++side) if (R.has(side)) {
			Rank& N = ranks[R.rank + (side ? 1 : -1)];
			N.in_count[1 - side] = R.out_count[side];
			if (R.out_count[side]) HIP_OK(hipMemcpyAsync(N.in[1 - side], R.out[side], record_bytes * R.out_count[side], hipMemcpyDeviceToDevice, T.stream));
		}
		return;
	}
	Rank& R = ranks[0];
	if (!known) {
		uint64_t h[4] = { R.out_count[0], R.out_count[1], 0, 0 };
		HIP_OK(hipMemcpyAsync(T.d_cnt, h, 16, hipMemcpyHostToDevice, T.stream));
		NCCL_OK(ncclGroupStart());
		for (int side = 0; side < 2; ++side) if (R.has(side)) {
			const int peer = T.rank + (side ? 1 : -1);
			NCCL_OK(ncclSend(T.d_cnt + side, 1, ncclUint64, peer, T.comm, T.stream)); NCCL_OK(ncclRecv(T.d_cnt + 2 + side, 1, ncclUint64, peer, T.comm, T.stream));
		}
		NCCL_OK(ncclGroupEnd());
		HIP_OK(hipMemcpyAsync(h, T.d_cnt, 32, hipMemcpyDeviceToHost, T.stream));
		HIP_OK(hipStreamSynchronize(T.stream));
		for (int side = 0; side < 2; ++side) R.in_count[side] = R.has(side) ? (uint32_t)h[2 + side] : 0u;
	}
	NCCL_OK(ncclGroupStart());
	for (int side = 0; side < 2; ++side) if (R.has(side)) {
		const int peer = T.rank + (side ? 1 : -1);
		if (R.out_count[side]) NCCL_OK(ncclSend(R.out[side], record_bytes * R.out_count[side], ncclUint8, peer, T.comm, T.stream));
		if (R.in_count[side]) NCCL_OK(ncclRecv(R.in[side], record_bytes * R.in_count[side], ncclUint8, peer, T.comm, T.stream));
	}
	NCCL_OK(ncclGroupEnd());
}

// one number per cut and side (a body count, a new cut): out[side][k] of rank k goes to in[1 - side][neighbour].  LOCAL: plain assignment
typedef std::vector<double> Values[2];
static void exchange_values(Transport& T, std::vector<Rank>& ranks, Values& out, Values& in) {
	if (T.local) {
		for (Rank& R : ranks) for (int side = 0; side < 2; ++side) if (R.has(side)) in[1 - side][R.rank + (side ? 1 : -1)] = out[side][R.rank];
		return;
	}
	Rank& R = ranks[0];
	double h[4] = { out[0][0], out[1][0], 0.0, 0.0 };
	HIP_OK(hipMemcpyAsync(T.d_cnt, h, 16, hipMemcpyHostToDevice, T.stream));
	NCCL_OK(ncclGroupStart());
	for (int side = 0; side < 2; ++side) if (R.has(side)) {
		const int peer = T.rank + (side ? 1 : -1);
		NCCL_OK(ncclSend(T.d_cnt + side, 1, ncclUint64, peer, T.comm, T.stream)); NCCL_OK(ncclRecv(T.d_cnt + 2 + side, 1, ncclUint64, peer, T.comm, T.stream));
	}
	NCCL_OK(ncclGroupEnd());
	HIP_OK(hipMemcpyAsync(h, T.d_cnt, 32, hipMemcpyDeviceToHost, T.stream));
	HIP_OK(hipStreamSynchronize(T.stream));
	in[0][0] = h[2]; in[1][0] = h[3];
}

int main(int argc, char** argv) {
	const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
	const int world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
	const int local_rank = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
	const int local_ranks = getenv("NUDGE_LOCAL_RANKS") ? atoi(getenv("NUDGE_LOCAL_RANKS")) : 0;
	const char* scene_path = nullptr; const char* out_path = nullptr; const char* cuts_arg = nullptr;
	std::vector<const char*> pos;
	uint32_t epoch = 16, min_extra = 4096; double speed_floor = 1.0, slack = 0.15; bool rebalance = true, single_owner = false, library_steps = false;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene_path = argv[++i];
		else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
		else if (!strcmp(argv[i], "--cuts") && i + 1 < argc) cuts_arg = argv[++i];
		else if (!strcmp(argv[i], "--epoch") && i + 1 < argc) epoch = (uint32_t)atoi(argv[++i]);
		else if (!strcmp(argv[i], "--speed-floor") && i + 1 < argc) speed_floor = atof(argv[++i]);
		else if (!strcmp(argv[i], "--min-extra") && i + 1 < argc) min_extra = (uint32_t)atoi(argv[++i]);
		else if (!strcmp(argv[i], "--slack") && i + 1 < argc) slack = atof(argv[++i]);
		else if (!strcmp(argv[i], "--no-rebalance")) rebalance = false;
		else if (!strcmp(argv[i], "--single-owner")) single_owner = true;
		else if (!strcmp(argv[i], "--library-steps")) library_steps = true;          // the sub-steps between two refreshes in ONE call: nh_partition_step (RCCL transport handed to the library)
		else pos.push_back(argv[i]);
	}
	const int n_ranks = local_ranks > 0 ? local_ranks : world;
	const int cols_per_rank = pos.size() > 0 ? atoi(pos[0]) : 4, rows = pos.size() > 1 ? atoi(pos[1]) : 2, side = pos.size() > 2 ? atoi(pos[2]) : 90;
	const unsigned steps = scene_path ? (pos.size() > 0 ? (unsigned)atoi(pos[0]) : 100u) : (pos.size() > 3 ? (unsigned)atoi(pos[3]) : 100u);
	const bool loopback = world == 1 && local_ranks == 0 && getenv("NUDGE_RCCL_LOOPBACK") && atoi(getenv("NUDGE_RCCL_LOOPBACK")) != 0;

	const int device = local_ranks > 0 ? 0 : local_rank;
	HIP_OK(hipSetDevice(device));
	hipStream_t stream;
	HIP_OK(hipStreamCreate(&stream));

	// ---- RCCL bootstrap (one process per GPU): rank 0 publishes the unique id through a file (one node, shared file system) ----
	Transport T;
	T.stream = stream; T.rank = rank; T.world = world; T.local = local_ranks > 0;
	if (local_ranks == 0 && (world > 1 || loopback)) {
		ncclUniqueId id;
		if (world > 1) {
			const char* path = getenv("NUDGE_NCCL_ID_FILE") ? getenv("NUDGE_NCCL_ID_FILE") : "/tmp/nudge_partition_rccl.id";
			if (rank == 0) {
				NCCL_OK(ncclGetUniqueId(&id));
				std::string tmp = std::string(path) + ".tmp";
				FILE* f = fopen(tmp.c_str(), "wb");
				if (!f || fwrite(&id, sizeof(id), 1, f) != 1) { perror(path); return 2; }
				fclose(f);
				rename(tmp.c_str(), path);
			} else {
				FILE* f = nullptr;
				for (int tries = 0; tries < 600 && !(f = fopen(path, "rb")); ++tries) usleep(100000);
				if (!f || fread(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "rank %d: no unique id at %s\n", rank, path); return 2; }
				fclose(f);
			}
		} else NCCL_OK(ncclGetUniqueId(&id));
		NCCL_OK(ncclCommInitRank(&T.comm, world, id, rank));
		HIP_OK(hipMalloc((void**)&T.d_cnt, 64));
	}

	// ---- the world and the cuts ----
	Scene scene = scene_path ? load_scene(scene_path) : drop_tiles(cols_per_rank * n_ranks, rows, side);
	std::vector<double> cuts;
	if (cuts_arg) { for (const char* p = cuts_arg; *p;) { char* e = nullptr; cuts.push_back(strtod(p, &e)); p = *e == ',' ? e + 1 : e; if (e == p) break; } }
	else if (n_ranks > 1) {
		// equal-count x-slabs: the quantiles of the bodies' x (+0.41: never exactly on a lattice line), like bench.py
		std::vector<float> xs;
		for (size_t i = 1; i < scene.body_xf.size(); ++i) xs.push_back(scene.body_xf[i].position[0]);
		std::sort(xs.begin(), xs.end());
		for (int r = 1; r < n_ranks; ++r) cuts.push_back((double)xs[std::min(xs.size() - 1, (size_t)((double)r / n_ranks * (xs.size() - 1)))] + 0.41);
	}
	if ((int)cuts.size() != n_ranks - 1) { fprintf(stderr, "%d ranks need %d cuts\n", n_ranks, n_ranks - 1); return 2; }
	// largest reach of any body of the WHOLE world (partition.py body_reach: |collider offset| + extent, rounded to float, x 1.0001)
	float max_reach = 0.0f;
	for (size_t c = 0; c < scene.box_xf.size(); ++c) if (scene.box_xf[c].body) {
		const nh_Transform& t = scene.box_xf[c]; const nh_BoxCollider& b = scene.box_data[c];
		const double off = sqrt((double)t.position[0] * t.position[0] + (double)t.position[1] * t.position[1] + (double)t.position[2] * t.position[2]);
		const double ext = sqrt((double)b.size[0] * b.size[0] + (double)b.size[1] * b.size[1] + (double)b.size[2] * b.size[2]);
		max_reach = std::max(max_reach, (float)(off + ext) * 1.0001f);
	}
	for (size_t c = 0; c < scene.sph_xf.size(); ++c) if (scene.sph_xf[c].body) {
		const nh_Transform& t = scene.sph_xf[c];
		const double off = sqrt((double)t.position[0] * t.position[0] + (double)t.position[1] * t.position[1] + (double)t.position[2] * t.position[2]);
		max_reach = std::max(max_reach, (float)(off + (double)scene.sph_data[c].radius) * 1.0001f);
	}

	std::vector<Rank> ranks(local_ranks > 0 ? local_ranks : 1);
	// (--single-owner: one nh_apply_impulses call per solver iteration, in the reference's solver order -- neither promise of the fused flags holds)
	for (size_t k = 0; k < ranks.size(); ++k) make_rank(ranks[k], scene, cuts, local_ranks > 0 ? (int)k : rank, device, stream, epoch, speed_floor, (double)max_reach, min_extra, slack,
	                                                    single_owner ? (uint32_t)NH_FLAG_EXACT_ORDER : (uint32_t)(NH_FLAG_SINGLE_APPLY | NH_FLAG_FUSED_STEP));
	const size_t total_bodies = scene.body_xf.size() - 1;

	// NUDGE_RCCL_LOOPBACK=1: per-step records of the first owned bodies to myself through RCCL, written back onto the same bodies (a no-op for the world)
	uint32_t n_loop = 0; uint32_t* d_loop = nullptr; void* loop_out = nullptr; void* loop_in = nullptr;
	if (loopback) {
		nh_PartitionInfo info; NH_OK_(nh_partition_info(ranks[0].part, &info));
		n_loop = std::min<uint32_t>(info.n_owned, 4096u);
		std::vector<uint32_t> idx(n_loop); for (uint32_t i = 0; i < n_loop; ++i) idx[i] = 1u + i;
		d_loop = to_device(idx, 1); HIP_OK(hipMalloc(&loop_out, (size_t)NH_HALO_RECORD_BYTES * (n_loop + 1))); HIP_OK(hipMalloc(&loop_in, (size_t)NH_HALO_RECORD_BYTES * (n_loop + 1)));
	}

	// --library-steps: the communicator and the four RCCL entry points go to the library (it links no RCCL), neighbours by communicator rank; a one-rank rehearsal names itself
	const bool use_library_steps = library_steps && !T.local && !single_owner && T.comm != nullptr;
	if (library_steps && !use_library_steps) { fprintf(stderr, "--library-steps needs the RCCL transport (one rank per process) and not --single-owner\n"); return 2; }
	if (use_library_steps)
		for (Rank& R : ranks)
			NH_OK_(nh_partition_set_transport(R.part, T.comm, (void*)&ncclGroupStart, (void*)&ncclGroupEnd, (void*)&ncclSend, (void*)&ncclRecv,
			                                  R.rank > 0 ? R.rank - 1 : -1, R.rank + 1 < n_ranks ? R.rank + 1 : (loopback ? R.rank : -1)));
	const size_t slots = T.local ? (size_t)n_ranks : (size_t)1;
	auto at = [&](const Rank& R) { return T.local ? (size_t)R.rank : (size_t)0; };
	HIP_OK(hipStreamSynchronize(stream));
	auto t0 = std::chrono::steady_clock::now();
	for (unsigned n = 0; n < steps; ++n) {
		if (n % epoch == 0 && n_ranks > 1) {
			// ---- REFRESH ----
			if (rebalance && n > 0) {
				Values mine = { std::vector<double>(slots, 0.0), std::vector<double>(slots, 0.0) }, theirs = { std::vector<double>(slots, 0.0), std::vector<double>(slots, 0.0) };
				for (Rank& R : ranks) { nh_PartitionInfo i; NH_OK_(nh_partition_info(R.part, &i)); mine[0][at(R)] = mine[1][at(R)] = (double)i.n_owned; }
				exchange_values(T, ranks, mine, theirs);
				// both sides of a cut evaluate the same rule on the same two counts; the heavier one picks the new place and sends it over
				Values cut_out = { std::vector<double>(slots, NAN), std::vector<double>(slots, NAN) }, cut_in = { std::vector<double>(slots, NAN), std::vector<double>(slots, NAN) };
				for (Rank& R : ranks) for (int side = 0; side < 2; ++side) if (R.has(side)) {
					const double a = mine[side][at(R)], b = theirs[side][at(R)];
					if (fabs(a - b) <= 0.05 * std::max(1.0, a + b) || a <= b) continue;
					double cut = 0.0;
					NH_OK_(nh_partition_choose_cut(R.part, &R.bodies, side ? +1 : -1, (uint32_t)b, &cut));
					cut_out[side][at(R)] = cut;
				}
				exchange_values(T, ranks, cut_out, cut_in);
				for (Rank& R : ranks) for (int side = 0; side < 2; ++side) if (R.has(side)) {
					const double a = mine[side][at(R)], b = theirs[side][at(R)], cut = cut_in[side][at(R)];
					if (fabs(a - b) > 0.05 * std::max(1.0, a + b) && a < b && cut == cut) NH_OK_(nh_partition_set_cut(R.part, side ? +1 : -1, cut));
				}
			}
			{
				// top speeds across the cuts: a cut's ghost margin must cover the fastest body on EITHER of its sides (include/nudge_hip.h: nh_partition_top_speed)
				Values mine = { std::vector<double>(slots, 0.0), std::vector<double>(slots, 0.0) }, theirs = { std::vector<double>(slots, 0.0), std::vector<double>(slots, 0.0) };
				for (Rank& R : ranks) { double v = 0.0; NH_OK_(nh_partition_top_speed(R.part, &R.bodies, &v)); mine[0][at(R)] = mine[1][at(R)] = v; }
				exchange_values(T, ranks, mine, theirs);
				for (Rank& R : ranks) NH_OK_(nh_partition_set_peer_speeds(R.part, R.has(0) ? theirs[0][at(R)] : 0.0, R.has(1) ? theirs[1][at(R)] : 0.0));
			}
			for (Rank& R : ranks) NH_OK_(nh_partition_pack_migrants(R.part, &R.bodies, R.has(0) ? R.out[0] : nullptr, R.has(1) ? R.out[1] : nullptr, R.msg_capacity, R.out_count));
			exchange(T, ranks, NH_PARTITION_FULL_BYTES, false);
			for (Rank& R : ranks) NH_OK_(nh_partition_unpack_migrants(R.part, &R.bodies, R.in[0], R.has(0) ? R.in_count[0] : 0u, R.in[1], R.has(1) ? R.in_count[1] : 0u));
			for (Rank& R : ranks) NH_OK_(nh_partition_pack_ghosts(R.part, &R.bodies, R.has(0) ? R.out[0] : nullptr, R.has(1) ? R.out[1] : nullptr, R.msg_capacity, R.out_count));
			exchange(T, ranks, NH_PARTITION_FULL_BYTES, false);
			for (Rank& R : ranks) NH_OK_(nh_partition_unpack_ghosts(R.part, &R.bodies, &R.colliders, R.in[0], R.has(0) ? R.in_count[0] : 0u, R.in[1], R.has(1) ? R.in_count[1] : 0u));
		} else if (n_ranks > 1 && !use_library_steps) {
			// ---- the per-step halo: the owners' state of this step overwrites the ghosts' (lengths fixed since the refresh) ----
			for (Rank& R : ranks) {
				NH_OK_(nh_partition_pack_step(R.part, &R.bodies, R.out[0], R.out[1]));
				nh_PartitionInfo i; NH_OK_(nh_partition_info(R.part, &i));
				R.out_count[0] = i.ghost_out[0]; R.out_count[1] = i.ghost_out[1]; R.in_count[0] = i.ghost_in[0]; R.in_count[1] = i.ghost_in[1];
			}
			exchange(T, ranks, NH_HALO_RECORD_BYTES, true);
			for (Rank& R : ranks) NH_OK_(nh_partition_unpack_step(R.part, &R.bodies, R.in[0], R.in[1]));
		}
		if (use_library_steps) {
			// ---- the sub-steps up to the next refresh in ONE library call: nh_partition_step enqueues the per-step halo (pack, ncclSend / ncclRecv, unpack, the ghosts'
			// transforms) between two sub-steps itself, so the chain of still steps -- xform ahead, pair ahead -- runs through the partitioned world (include/nudge_hip.h)
			const unsigned k = n_ranks > 1 ? std::min<unsigned>(steps - n, epoch - n % epoch) : steps - n;
			const float dt = (float)scene.time_step;
			for (Rank& R : ranks) {
				nh_StepArgs a;
				memset(&a, 0, sizeof(a));
				a.active_bodies = &R.active; a.contacts = &R.contacts; a.bodies = &R.bodies; a.colliders = &R.colliders; a.body_connections = &R.connections; a.contact_cache = &R.cache;
				a.arena = R.arena; a.time_step = dt; a.gravity[0] = 0.0f; a.gravity[1] = (float)scene.gravity; a.gravity[2] = 0.0f; a.damping_rate = (float)scene.damping_rate; a.iterations = scene.iterations;
				NH_OK_(nh_partition_step(R.part, &a, k, (n_ranks == 1 || n % epoch != 0) ? 1u : 0u, loopback ? n_loop : 0u));
			}
			n += k - 1u;
			continue;
		}
		if (loopback && n_loop) {
			Rank& R = ranks[0];
			NH_OK_(nh_halo_pack(R.ctx, &R.bodies, d_loop, n_loop, loop_out));
			NCCL_OK(ncclGroupStart());
			NCCL_OK(ncclSend(loop_out, (size_t)n_loop * NH_HALO_RECORD_BYTES, ncclUint8, 0, T.comm, stream));
			NCCL_OK(ncclRecv(loop_in, (size_t)n_loop * NH_HALO_RECORD_BYTES, ncclUint8, 0, T.comm, stream));
			NCCL_OK(ncclGroupEnd());
			NH_OK_(nh_halo_update(R.ctx, &R.bodies, 1, n_loop, loop_in));
		}
		if (!single_owner) { for (Rank& R : ranks) local_step(R, scene); continue; }
		// ---- single ownership of contacts that cross a cut (include/nudge_hip.h: nh_set_first_ghost_body and the per-iteration exchange) ----
		// EXCHANGE: what a rank's sweeps did to its ghosts goes back to their owners, the owners' momentum goes out to the ghosts, the ghosts are marked
		auto exchange_iteration = [&]() {
			for (Rank& R : ranks) {
				nh_PartitionInfo i; NH_OK_(nh_partition_info(R.part, &i));
				NH_OK_(nh_partition_pack_deltas(R.part, &R.bodies, R.out[0], R.out[1]));
				R.out_count[0] = i.ghost_in[0]; R.out_count[1] = i.ghost_in[1]; R.in_count[0] = i.ghost_out[0]; R.in_count[1] = i.ghost_out[1];
			}
			exchange(T, ranks, sizeof(nh_BodyMomentum), true);
			for (Rank& R : ranks) NH_OK_(nh_partition_unpack_deltas(R.part, &R.bodies, R.in[0], R.in[1]));
			for (Rank& R : ranks) {
				nh_PartitionInfo i; NH_OK_(nh_partition_info(R.part, &i));
				NH_OK_(nh_partition_pack_momentum(R.part, &R.bodies, R.out[0], R.out[1]));
				R.out_count[0] = i.ghost_out[0]; R.out_count[1] = i.ghost_out[1]; R.in_count[0] = i.ghost_in[0]; R.in_count[1] = i.ghost_in[1];
			}
			exchange(T, ranks, sizeof(nh_BodyMomentum), true);
			for (Rank& R : ranks) { NH_OK_(nh_partition_unpack_momentum(R.part, &R.bodies, R.in[0], R.in[1])); NH_OK_(nh_partition_mark_ghosts(R.part, &R.bodies)); }
		};
		const float dt = (float)scene.time_step, gravity[3] = { 0.0f, (float)scene.gravity, 0.0f };
		std::vector<nh_ContactImpulseData*> impulses(ranks.size(), nullptr);
		std::vector<nh_ContactConstraintData*> constraints(ranks.size(), nullptr);
		for (size_t k = 0; k < ranks.size(); ++k) {
			Rank& R = ranks[k];
			nh_PartitionInfo i; NH_OK_(nh_partition_info(R.part, &i));
			NH_OK_(nh_set_first_ghost_body(R.ctx, i.n_owned + 1u));
			nh_Arena temporary = R.arena;
			NH_OK_(nh_collide(R.ctx, &R.active, &R.contacts, &R.bodies, &R.colliders, &R.connections, temporary));
			NH_OK_(nh_apply_gravity_damping(R.ctx, &R.active, &R.bodies, dt, gravity, (float)scene.damping_rate));
			NH_OK_(nh_read_cached_impulses(R.ctx, &R.cache, &R.contacts, &temporary, &impulses[k]));
			NH_OK_(nh_partition_mark_ghosts(R.part, &R.bodies));          // (the warm start inside setup already moves the ghosts of the contacts owned here)
			NH_OK_(nh_setup_contact_constraints(R.ctx, &R.active, &R.contacts, &R.bodies, impulses[k], &temporary, &constraints[k]));
		}
		exchange_iteration();
		for (uint32_t it = 0; it < scene.iterations; ++it)
			for (int phase = 0; phase < 2; ++phase) {          // even ranks sweep, results travel, odd ranks sweep: one Gauss-Seidel sweep of the world in rank-parity order
				for (size_t k = 0; k < ranks.size(); ++k) if (ranks[k].rank % 2 == phase) NH_OK_(nh_apply_impulses(ranks[k].ctx, constraints[k], &ranks[k].bodies, 1u));
				exchange_iteration();
			}
		for (size_t k = 0; k < ranks.size(); ++k) {
			Rank& R = ranks[k];
			NH_OK_(nh_update_cached_impulses(R.ctx, constraints[k], impulses[k]));
			NH_OK_(nh_write_cached_impulses(R.ctx, &R.cache, &R.contacts, impulses[k]));
			NH_OK_(nh_advance(R.ctx, &R.active, &R.bodies, dt));
		}
	}
	for (Rank& R : ranks) NH_OK_(nh_synchronize(R.ctx));
	const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

	// ---- results: per rank a line; optionally every owned body (transform with the global id, momentum, idle) into a file the tests read ----
	FILE* fo = out_path ? fopen(out_path, "wb") : nullptr;
	uint64_t owned_all = 0;
	for (Rank& R : ranks) {
		nh_Counts c; NH_OK_(nh_read_counts(R.ctx, &c));
		if (c.error) { fprintf(stderr, "rank %d: device error: %s\n", R.rank, nh_error_string((int)c.error)); return 5; }
		nh_PartitionInfo info; NH_OK_(nh_partition_info(R.part, &info));
		std::vector<nh_Transform> xf(info.n_owned + 1u); std::vector<nh_BodyMomentum> mom(info.n_owned + 1u); std::vector<uint8_t> idle(info.n_owned + 1u);
		HIP_OK(hipMemcpy(xf.data(), R.bodies.transforms, sizeof(nh_Transform) * xf.size(), hipMemcpyDeviceToHost));
		HIP_OK(hipMemcpy(mom.data(), R.bodies.momentum, sizeof(nh_BodyMomentum) * mom.size(), hipMemcpyDeviceToHost));
		HIP_OK(hipMemcpy(idle.data(), R.bodies.idle_counters, idle.size(), hipMemcpyDeviceToHost));
		double checksum = 0.0;
		for (uint32_t i = 1; i <= info.n_owned; ++i) checksum += (double)xf[i].position[0] + 2.0 * xf[i].position[1] + 3.0 * xf[i].position[2];
		if (fo) for (uint32_t i = 1; i <= info.n_owned; ++i) { fwrite(&xf[i], sizeof(nh_Transform), 1, fo); fwrite(&mom[i], sizeof(nh_BodyMomentum), 1, fo); fwrite(&idle[i], 1, 1, fo); }
		owned_all += info.n_owned;
		printf("partition_rccl rank %d/%d: %u owned + %u ghost bodies, %u steps, last step %u contacts (%.3f per owned+ghost body), %.1f steps/s, refreshes %llu, migrated out %llu in %llu, cut moves %llu, "
		       "still steps %u, checksum %.6f\n", R.rank, n_ranks, info.n_owned, info.ghost_in[0] + info.ghost_in[1], steps, c.contacts, (double)c.contacts / std::max(1u, info.n_bodies - 1u), steps / seconds,
		       (unsigned long long)info.refreshes, (unsigned long long)info.migrated_out, (unsigned long long)info.migrated_in, (unsigned long long)info.cut_moves, c.still_steps, checksum);
	}
	if (fo) fclose(fo);
	if (local_ranks > 0 && owned_all != total_bodies) { fprintf(stderr, "bodies lost or duplicated: %llu owned of %zu\n", (unsigned long long)owned_all, total_bodies); return 7; }
	if (loopback) printf("partition_rccl loopback: %u halo records per step through ncclSend/ncclRecv(self), %u steps%s\n", n_loop, steps, use_library_steps ? " (inside nh_partition_step)" : "");
	for (Rank& R : ranks) { nh_partition_destroy(R.part); nh_destroy(R.ctx); }
	if (T.comm) NCCL_OK(ncclCommDestroy(T.comm));
	return 0;
}
