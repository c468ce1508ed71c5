// examples/partition_rccl.cpp -- C++ host for the MULTI-GPU path: one process per GPU, one x-slab of a drop-tile world per process,
// ghost-body halo exchange every step with ncclSend / ncclRecv (RCCL over xGMI) on the records of nh_halo_pack / nh_halo_unpack.
//
// The shape of BASELINE.json configs[4] (boxes dropping onto ground slabs): bodies never leave their slab sideways, so
// this example keeps STATIC ghost lists and does no migration (nudge_amd/partition.py is the general host: refresh epochs, migration,
// equal-count cuts; it talks to the same two C-ABI entry points through torch.distributed).  What it shows is the host side in the
// reference's own language on the performance path:
//   * every rank generates, from the same counter-based generator, the tiles that touch its slab: the bodies whose centre lies in
//     [lo, hi) are OWNED, those within `margin` outside are GHOSTS (slots after the owned ones, ordinary bodies to the engine);
//   * per step:  nh_halo_pack(owned bodies near a cut) -> ncclGroupStart; ncclSend / ncclRecv with both neighbours; ncclGroupEnd
//                -> nh_halo_unpack(into the ghost slots) -> the eight calls of the reference's sub-step (example/main.cpp:274-328);
//     everything on ONE HIP stream, no host synchronisation inside the loop, no collective on the data path;
//   * both sides of a cut enumerate the bodies in the same (generation) order, so a message needs no index list.
// Contacts between an owned body and a ghost are solved on both ranks and the owner's state wins at the next exchange (exact when no
// contact crosses a cut, which is the case here: boxes only touch the ground).
//
// build:  make -C nudge_amd/csrc partition_rccl            (hipcc/g++ + -lrccl -lamdhip64 -lnudge_hip)
// run:    one process per GPU, e.g.
//         for r in 0 1 2 3 4 5 6 7; do RANK=$r WORLD_SIZE=8 LOCAL_RANK=$r NUDGE_NCCL_ID_FILE=/tmp/nudge.id \
//             nudge_amd/partition_rccl 16 2 90 200 & done; wait
//         arguments: tile columns per rank, tile rows, boxes per tile side, steps
//         WORLD_SIZE=1 runs the whole world on one GPU without any communication (what tests/test_gpu_callsite.py does).
#include "../include/nudge_hip.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

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

// counter-based generator (splitmix64): the same body on every rank that generates it
static float u01(uint64_t seed, uint64_t counter) {
	uint64_t x = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	x ^= x >> 31;
	return (float)((double)(x >> 40) * (1.0 / 16777216.0));
}

struct Body { nh_Transform xf; nh_BodyProperties props; nh_BoxCollider box; uint32_t tag; };

template<class T>
static T* to_device(const std::vector<T>& v, size_t capacity = 0) {
	size_t n = capacity > v.size() ? capacity : v.size();
	T* d = nullptr;
	HIP_OK(hipMalloc((void**)&d, sizeof(T) * (n ? n : 1)));
	HIP_OK(hipMemset(d, 0, sizeof(T) * (n ? n : 1)));
	if (!v.empty()) HIP_OK(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
	return d;
}

int main(int argc, char** argv) {
	const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
	const int world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
	const int local_rank = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
	const int cols_per_rank = argc > 1 ? atoi(argv[1]) : 4, rows = argc > 2 ? atoi(argv[2]) : 2, side = argc > 3 ? atoi(argv[3]) : 90;
	const unsigned steps = argc > 4 ? (unsigned)atoi(argv[4]) : 100;
	const int cols = cols_per_rank * world;
	const uint32_t iterations = 8;
	const float time_step = 1.0f / 120.0f, gravity[3] = { 0.0f, 9.82f, 0.0f }, damping_rate = 0.25f;
	const uint64_t seed = 0x9E3779B97F4A7C15ull ^ 5ull;

	HIP_OK(hipSetDevice(local_rank));
	hipStream_t stream;
	HIP_OK(hipStreamCreate(&stream));

	// ---- RCCL bootstrap: rank 0 publishes the unique id through a file (one node, shared file system) ----
	ncclComm_t comm = nullptr;
	if (world > 1) {
		const char* path = getenv("NUDGE_NCCL_ID_FILE") ? getenv("NUDGE_NCCL_ID_FILE") : "/tmp/nudge_partition_rccl.id";
		ncclUniqueId id;
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
		NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
	}
	// NUDGE_RCCL_LOOPBACK=1 with WORLD_SIZE=1: a one-rank communicator, and every step the halo records of the first owned bodies travel
	// nh_halo_pack -> ncclSend(self) / ncclRecv(self) in one group -> nh_halo_update onto the SAME bodies (a no-op for the simulation: the
	// checksum must equal the run without it).  What a one-GPU box can execute of the communication path: RCCL point-to-point on the
	// library's stream between the two halo kernels.
	const bool loopback = world == 1 && getenv("NUDGE_RCCL_LOOPBACK") && atoi(getenv("NUDGE_RCCL_LOOPBACK")) != 0;
	if (loopback) {
		ncclUniqueId id;
		NCCL_OK(ncclGetUniqueId(&id));
		NCCL_OK(ncclCommInitRank(&comm, 1, id, 0));
	}

	// ---- the world: cols x rows tiles of side x side boxes on a 3.0 pitch, one ground slab per tile; x-slabs cut THROUGH tiles ----
	const float pitch = 3.0f, half = side * 1.5f + 2.0f, tile_pitch = 2.0f * half + 4.0f;
	const float x_begin = -half, x_end = (cols - 1) * tile_pitch + half;
	const float width = (x_end - x_begin) / world;
	const float lo = rank == 0 ? -1e30f : x_begin + rank * width + 0.41f;           // (+0.41: never exactly on a lattice line)
	const float hi = rank == world - 1 ? 1e30f : x_begin + (rank + 1) * width + 0.41f;
	const float margin = 8.0f;                                                       // > 2 x the largest box diagonal + sideways drift of a drop

	std::vector<Body> owned, ghost_left, ghost_right;       // ghost_left: owned by rank-1, within `margin` of my lo; ...
	std::vector<uint32_t> send_left, send_right;           // owned slots (1-based body indices) the neighbours hold as ghosts, generation order
	std::vector<nh_Transform> slab_xf;
	std::vector<nh_BoxCollider> slab_box;
	std::vector<uint32_t> slab_tags;
	const uint32_t n_tiles = (uint32_t)(cols * rows);
	for (int tz = 0; tz < rows; ++tz) for (int tx = 0; tx < cols; ++tx) {
		const float ox = tx * tile_pitch, oz = tz * tile_pitch;
		if (ox + half < lo - margin || ox - half > hi + margin) continue;        // the tile does not touch this rank
		const uint32_t tile = (uint32_t)(tz * cols + tx);
		nh_Transform st = {};
		st.position[0] = ox; st.position[1] = -20.0f; st.position[2] = oz; st.body = 0; st.rotation[3] = 1.0f;
		slab_xf.push_back(st);
		nh_BoxCollider sb = { { half, 10.0f, half }, 0.0f };
		slab_box.push_back(sb);
		slab_tags.push_back(tile);                                                 // static colliders: tags 0 .. n_tiles-1
		for (int iz = 0; iz < side; ++iz) for (int ix = 0; ix < side; ++ix) {
			const uint64_t g = (uint64_t)tile * side * side + (uint64_t)iz * side + ix;   // global body number (0-based among the dynamic ones)
			Body b = {};
			const float sx = 0.5f + 0.5f * u01(seed, 8 * g), sy = 0.5f + 0.5f * u01(seed, 8 * g + 1), sz = 0.5f + 0.5f * u01(seed, 8 * g + 2);
			const float yaw = 0.5f * u01(seed, 8 * g + 3);
			b.xf.position[0] = ox + (ix - (side - 1) * 0.5f) * pitch + 0.2f * u01(seed, 8 * g + 4);
			b.xf.position[2] = oz + (iz - (side - 1) * 0.5f) * pitch + 0.2f * u01(seed, 8 * g + 5);
			b.xf.position[1] = -10.0f + 0.5f + 0.5f * u01(seed, 8 * g + 6) + sy;
			b.xf.rotation[1] = sinf(0.5f * yaw); b.xf.rotation[3] = cosf(0.5f * yaw);
			const float mass = 8.0f * sx * sy * sz, k = mass / 3.0f;            // add_box of the sample application (example/main.cpp:112-145)
			b.props.mass_inverse = 1.0f / mass;
			b.props.inertia_inverse[0] = 1.0f / (k * sy * sy + k * sz * sz);
			b.props.inertia_inverse[1] = 1.0f / (k * sx * sx + k * sz * sz);
			b.props.inertia_inverse[2] = 1.0f / (k * sx * sx + k * sy * sy);
			b.box.size[0] = sx; b.box.size[1] = sy; b.box.size[2] = sz;
			b.tag = n_tiles + (uint32_t)g;                                          // collider tag = global, identical on every rank
			const float x = b.xf.position[0];
			if (x >= lo && x < hi) {
				owned.push_back(b);
				if (x < lo + margin && rank > 0) send_left.push_back((uint32_t)owned.size());
				if (x >= hi - margin && rank < world - 1) send_right.push_back((uint32_t)owned.size());
			} else if (x >= lo - margin && x < lo) ghost_left.push_back(b);
			else if (x >= hi && x < hi + margin) ghost_right.push_back(b);
		}
	}
	const uint32_t n_loop = loopback ? (uint32_t)(owned.size() < 4096 ? owned.size() : 4096) : 0u;
	if (loopback) for (uint32_t i = 0; i < n_loop; ++i) send_right.push_back(1u + i);           // (consecutive slots: nh_halo_update writes a range)
	const uint32_t n_owned = (uint32_t)owned.size(), n_gl = (uint32_t)ghost_left.size(), n_gr = (uint32_t)ghost_right.size();
	const uint32_t nb = 1 + n_owned + n_gl + n_gr, n_static = (uint32_t)slab_xf.size(), nbox = n_static + nb - 1;

	// body 0 = the static world; then owned, then the left ghosts, then the right ghosts.  One box collider per dynamic body.
	std::vector<nh_Transform> h_xf(nb), h_box_xf(nbox);
	std::vector<nh_BodyProperties> h_props(nb);
	std::vector<nh_BoxCollider> h_box(nbox);
	std::vector<uint32_t> h_tags(nbox);
	h_xf[0].rotation[3] = 1.0f;
	for (uint32_t s = 0; s < n_static; ++s) { h_box_xf[s] = slab_xf[s]; h_box[s] = slab_box[s]; h_tags[s] = slab_tags[s]; }
	uint32_t at = 1;
	for (const std::vector<Body>* list : { &owned, &ghost_left, &ghost_right })
		for (const Body& b : *list) {
			h_xf[at] = b.xf; h_props[at] = b.props;
			nh_Transform local = {}; local.rotation[3] = 1.0f; local.body = at;
			h_box_xf[n_static + at - 1] = local; h_box[n_static + at - 1] = b.box; h_tags[n_static + at - 1] = b.tag;
			++at;
		}

	nh_context* ctx = nullptr;
	NH_OK_(nh_create(&ctx, local_rank, stream, NH_FLAG_SINGLE_APPLY | NH_FLAG_FUSED_STEP));
	uint32_t tag_bits = 1; while ((1ull << tag_bits) <= (uint64_t)n_tiles + (uint64_t)n_tiles * side * side) ++tag_bits;
	NH_OK_(nh_set_tag_bits(ctx, tag_bits));
	const uint32_t K = 6u * nb + 4096u;
	nh_BodyData bodies = { to_device(h_xf), to_device(h_props), to_device(std::vector<nh_BodyMomentum>(nb)), to_device(std::vector<uint8_t>(nb)), nb };
	nh_ColliderData colliders = {};
	colliders.boxes.tags = to_device(h_tags); colliders.boxes.data = to_device(h_box); colliders.boxes.transforms = to_device(h_box_xf); colliders.boxes.count = nbox;
	colliders.spheres.tags = to_device(std::vector<uint32_t>(), 1); colliders.spheres.data = to_device(std::vector<nh_SphereCollider>(), 1);
	colliders.spheres.transforms = to_device(std::vector<nh_Transform>(), 1); colliders.spheres.count = 0;
	nh_BodyConnections connections = { nullptr, 0 };
	nh_ContactData contacts = {};
	contacts.data = to_device(std::vector<nh_Contact>(), K); contacts.bodies = to_device(std::vector<nh_BodyPair>(), K);
	contacts.tags = to_device(std::vector<uint64_t>(), K); contacts.features = to_device(std::vector<uint32_t>(), K);
	contacts.sleeping_pairs = to_device(std::vector<uint64_t>(), K); contacts.capacity = K;
	nh_ContactCache cache = { to_device(std::vector<uint64_t>(), K), to_device(std::vector<uint32_t>(), K), to_device(std::vector<nh_CachedContactImpulse>(), K), K, 0 };
	nh_ActiveBodies active = { to_device(std::vector<uint32_t>(), nb), nb, 0 };
	nh_Arena arena = {};
	arena.size = ((size_t)64 << 20) + (size_t)nbox * 200 + ((size_t)1 << 24) * 8 + (size_t)K * 320 + ((size_t)K / 2 + 1024) * 216 + (size_t)nb * 64;
	HIP_OK(hipMalloc(&arena.data, arena.size));

	// halo buffers: what I send to / receive from each neighbour (NH_HALO_RECORD_BYTES per body)
	uint32_t* d_send_left = to_device(send_left, 1); uint32_t* d_send_right = to_device(send_right, 1);
	void *out_left = nullptr, *out_right = nullptr, *in_left = nullptr, *in_right = nullptr;
	HIP_OK(hipMalloc(&out_left, (size_t)NH_HALO_RECORD_BYTES * (send_left.size() + 1))); HIP_OK(hipMalloc(&out_right, (size_t)NH_HALO_RECORD_BYTES * (send_right.size() + 1)));
	HIP_OK(hipMalloc(&in_left, (size_t)NH_HALO_RECORD_BYTES * (n_gl + 1))); HIP_OK(hipMalloc(&in_right, (size_t)NH_HALO_RECORD_BYTES * (n_gr + n_loop + 1)));
	if (world > 1) {
		// both sides must agree on the list lengths: what I call my right ghosts is what my right neighbour calls its left sends
		uint64_t mine[2] = { (uint64_t)send_left.size(), (uint64_t)send_right.size() }, theirs[2] = { 0, 0 };
		uint64_t* d_m = to_device(std::vector<uint64_t>(mine, mine + 2)); uint64_t* d_t = to_device(std::vector<uint64_t>(2));
		NCCL_OK(ncclGroupStart());
		if (rank > 0) { NCCL_OK(ncclSend(d_m, 1, ncclUint64, rank - 1, comm, stream)); NCCL_OK(ncclRecv(d_t, 1, ncclUint64, rank - 1, comm, stream)); }
		if (rank < world - 1) { NCCL_OK(ncclSend(d_m + 1, 1, ncclUint64, rank + 1, comm, stream)); NCCL_OK(ncclRecv(d_t + 1, 1, ncclUint64, rank + 1, comm, stream)); }
		NCCL_OK(ncclGroupEnd());
		HIP_OK(hipStreamSynchronize(stream));
		HIP_OK(hipMemcpy(theirs, d_t, sizeof(theirs), hipMemcpyDeviceToHost));
		// the left neighbour sent me the length of ITS right list (= my left ghosts), the right neighbour that of its left list
		if ((rank > 0 && theirs[0] != n_gl) || (rank < world - 1 && theirs[1] != n_gr)) { fprintf(stderr, "rank %d: ghost lists disagree (%llu/%u, %llu/%u)\n", rank, (unsigned long long)theirs[0], n_gl, (unsigned long long)theirs[1], n_gr); return 6; }
	}

	uint64_t sum_contacts = 0;
	HIP_OK(hipStreamSynchronize(stream));
	auto t0 = std::chrono::steady_clock::now();
	for (unsigned n = 0; n < steps; ++n) {
		if (world > 1) {
			// ---- ghost-body halo exchange: the owners' state of this step overwrites the ghosts' ----
			NH_OK_(nh_halo_pack(ctx, &bodies, d_send_left, (uint32_t)send_left.size(), out_left));
			NH_OK_(nh_halo_pack(ctx, &bodies, d_send_right, (uint32_t)send_right.size(), out_right));
			NCCL_OK(ncclGroupStart());
			if (rank > 0) {
				if (!send_left.empty()) NCCL_OK(ncclSend(out_left, send_left.size() * NH_HALO_RECORD_BYTES, ncclUint8, rank - 1, comm, stream));
				if (n_gl) NCCL_OK(ncclRecv(in_left, (size_t)n_gl * NH_HALO_RECORD_BYTES, ncclUint8, rank - 1, comm, stream));
			}
			if (rank < world - 1) {
				if (!send_right.empty()) NCCL_OK(ncclSend(out_right, send_right.size() * NH_HALO_RECORD_BYTES, ncclUint8, rank + 1, comm, stream));
				if (n_gr) NCCL_OK(ncclRecv(in_right, (size_t)n_gr * NH_HALO_RECORD_BYTES, ncclUint8, rank + 1, comm, stream));
			}
			NCCL_OK(ncclGroupEnd());
			NH_OK_(nh_halo_update(ctx, &bodies, 1 + n_owned, n_gl, in_left));
			NH_OK_(nh_halo_update(ctx, &bodies, 1 + n_owned + n_gl, n_gr, in_right));
		}
		if (loopback && n_loop) {
			NH_OK_(nh_halo_pack(ctx, &bodies, d_send_right, n_loop, out_right));
			NCCL_OK(ncclGroupStart());
			NCCL_OK(ncclSend(out_right, (size_t)n_loop * NH_HALO_RECORD_BYTES, ncclUint8, 0, comm, stream));
			NCCL_OK(ncclRecv(in_right, (size_t)n_loop * NH_HALO_RECORD_BYTES, ncclUint8, 0, comm, stream));
			NCCL_OK(ncclGroupEnd());
			NH_OK_(nh_halo_update(ctx, &bodies, 1, n_loop, in_right));
		}
		// ---- the reference's sub-step on the local world (example/main.cpp:274-328) ----
		nh_Arena temporary = arena;
		NH_OK_(nh_collide(ctx, &active, &contacts, &bodies, &colliders, &connections, temporary));
		NH_OK_(nh_apply_gravity_damping(ctx, &active, &bodies, time_step, gravity, damping_rate));
		nh_ContactImpulseData* impulses = nullptr;
		NH_OK_(nh_read_cached_impulses(ctx, &cache, &contacts, &temporary, &impulses));
		nh_ContactConstraintData* constraints = nullptr;
		NH_OK_(nh_setup_contact_constraints(ctx, &active, &contacts, &bodies, impulses, &temporary, &constraints));
		NH_OK_(nh_apply_impulses(ctx, constraints, &bodies, iterations));
		NH_OK_(nh_update_cached_impulses(ctx, constraints, impulses));
		NH_OK_(nh_write_cached_impulses(ctx, &cache, &contacts, impulses));
		NH_OK_(nh_advance(ctx, &active, &bodies, time_step));
	}
	NH_OK_(nh_synchronize(ctx));
	const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	nh_Counts c;
	NH_OK_(nh_read_counts(ctx, &c));
	if (c.error) { fprintf(stderr, "rank %d: device error: %s\n", rank, nh_error_string((int)c.error)); return 5; }
	sum_contacts = c.contacts;

	// owned bodies must still be inside slab + margin (this example has no migration), and a position checksum for comparisons between runs
	HIP_OK(hipMemcpy(h_xf.data(), bodies.transforms, sizeof(nh_Transform) * nb, hipMemcpyDeviceToHost));
	double checksum = 0.0;
	for (uint32_t i = 1; i <= n_owned; ++i) {
		const float x = h_xf[i].position[0];
		if (x < lo - margin * 0.5f || x > hi + margin * 0.5f) { fprintf(stderr, "rank %d: body %u left its slab (x = %g): this example does not migrate bodies\n", rank, i, x); return 7; }
		checksum += (double)h_xf[i].position[0] + 2.0 * h_xf[i].position[1] + 3.0 * h_xf[i].position[2];
	}
	printf("partition_rccl rank %d/%d: %u owned + %u ghost bodies, %u steps, last step %llu contacts (%.3f per owned+ghost body), %.1f steps/s, checksum %.6f\n",
	       rank, world, n_owned, n_gl + n_gr, steps, (unsigned long long)sum_contacts, (double)sum_contacts / (nb - 1), steps / seconds, checksum);
	if (loopback) printf("partition_rccl loopback: %u halo records per step through ncclSend/ncclRecv(self), %u steps\n", n_loop, steps);
	nh_destroy(ctx);
	if (comm) NCCL_OK(ncclCommDestroy(comm));
	return 0;
}
