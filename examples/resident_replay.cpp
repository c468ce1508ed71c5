// examples/resident_replay.cpp -- a C++ host driving the DEVICE-RESIDENT C ABI (include/nudge_hip.h) directly.
//
// The sub-step loop of the reference's sample (example/main.cpp:274-328: collide, gravity/damping, read_cached_impulses,
// setup_contact_constraints, apply_impulses, update_cached_impulses, write_cached_impulses, advance) with every array in HBM:
// the scene is uploaded once, the loop only enqueues work on one HIP stream, state comes back when somebody wants to look.
// This is the performance path an adopter ports a call site to; nudge_amd/compat/ is the
// no-change path.  Needs nothing from the reference: plain C++ + the HIP runtime + libnudge_hip.so.
//
// usage: resident_replay <scene.bin> <out.bin> <steps> [exact]
//   scene.bin / out.bin: the formats of tests/callsite/replay.cpp (reference layouts, 16-bit tags), so the two programs can be
//   compared byte for byte; `exact` selects the reference's greedy solver order (NH_FLAG_EXACT_ORDER).
#include "../include/nudge_hip.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); exit(3); } } while (0)
#define NH_OK_(expr) do { int rc_ = (expr); if (rc_) { fprintf(stderr, "%s: %s\n", #expr, nh_error_string(rc_)); exit(4); } } while (0)

template<class T>
static std::vector<T> read_array(FILE* f, size_t n) {
	std::vector<T> v(n);
	if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short scene file\n"); exit(2); }
	return v;
}

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
	if (argc < 4) { fprintf(stderr, "usage: %s scene.bin out.bin steps [exact | step]\n", argv[0]); return 2; }
	FILE* f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	std::vector<uint32_t> head = read_array<uint32_t>(f, 5);
	std::vector<float> par = read_array<float>(f, 3);
	const uint32_t nb = head[0], nbox = head[1], nsph = head[2], iterations = head[3], K = head[4];
	const float time_step = par[0], damping_rate = par[2];
	const float gravity[3] = { 0.0f, par[1], 0.0f };
	const unsigned steps = (unsigned)atoi(argv[3]);
	const bool exact = argc > 4 && !strcmp(argv[4], "exact");
	// "step": the whole loop as ONE call of nh_step under NH_FLAG_FUSED_STEP (the sample's call order promised): still steps, late verdicts, asleep steps
	const bool one_call = argc > 4 && !strcmp(argv[4], "step");

	// host copies in the file's (reference) layouts; tags widen to 32 bits
	std::vector<nh_Transform> h_xf = read_array<nh_Transform>(f, nb);
	std::vector<nh_BodyProperties> h_props = read_array<nh_BodyProperties>(f, nb);
	std::vector<nh_BodyMomentum> h_mom = read_array<nh_BodyMomentum>(f, nb);
	std::vector<uint8_t> h_idle = read_array<uint8_t>(f, nb);
	std::vector<uint16_t> t16 = read_array<uint16_t>(f, nbox);
	std::vector<uint32_t> h_box_tags(t16.begin(), t16.end());
	std::vector<nh_BoxCollider> h_box = read_array<nh_BoxCollider>(f, nbox);
	std::vector<nh_Transform> h_box_xf = read_array<nh_Transform>(f, nbox);
	t16 = read_array<uint16_t>(f, nsph);
	std::vector<uint32_t> h_sph_tags(t16.begin(), t16.end());
	std::vector<nh_SphereCollider> h_sph = read_array<nh_SphereCollider>(f, nsph);
	std::vector<nh_Transform> h_sph_xf = read_array<nh_Transform>(f, nsph);
	fclose(f);

	hipStream_t stream;
	HIP_OK(hipStreamCreate(&stream));
	nh_context* ctx = nullptr;
	NH_OK_(nh_create(&ctx, 0, stream, exact ? NH_FLAG_EXACT_ORDER : (one_call ? NH_FLAG_FUSED_STEP : 0u)));
	NH_OK_(nh_set_tag_bits(ctx, 16));

	// caller-owned arrays, now in HBM (the library never allocates them)
	nh_BodyData bodies = { to_device(h_xf), to_device(h_props), to_device(h_mom), to_device(h_idle), nb };
	nh_ColliderData colliders = {};
	colliders.boxes.tags = to_device(h_box_tags); colliders.boxes.data = to_device(h_box); colliders.boxes.transforms = to_device(h_box_xf); colliders.boxes.count = nbox;
	colliders.spheres.tags = to_device(h_sph_tags); colliders.spheres.data = to_device(h_sph); colliders.spheres.transforms = to_device(h_sph_xf); colliders.spheres.count = nsph;
	nh_BodyConnections connections = { nullptr, 0 };
	nh_ContactData contacts = {};
	contacts.data = to_device(std::vector<nh_Contact>(), K); contacts.bodies = to_device(std::vector<nh_BodyPair>(), K);
	contacts.tags = to_device(std::vector<uint64_t>(), K); contacts.features = to_device(std::vector<uint32_t>(), K);
	contacts.sleeping_pairs = to_device(std::vector<uint64_t>(), K); contacts.capacity = K;
	nh_ContactCache cache = { to_device(std::vector<uint64_t>(), K), to_device(std::vector<uint32_t>(), K), to_device(std::vector<nh_CachedContactImpulse>(), K), K, 0 };
	nh_ActiveBodies active = { to_device(std::vector<uint32_t>(), nb), nb, 0 };
	nh_Arena arena = {};
	arena.size = ((size_t)64 << 20) + (size_t)(nbox + nsph) * 200 + ((size_t)1 << 22) * 8 + (size_t)K * 420 + (size_t)nb * 64;
	HIP_OK(hipMalloc(&arena.data, arena.size));

	// A viewer's feed (example/main.cpp:176-272 draws from bodies.transforms after simulate()): NUDGE_STREAM_EVERY=N puts the transforms of every N-th sub-step into a
	// pinned host ring while the world goes on (include/nudge_hip.h: nh_stream_state); a render thread would poll nh_stream_latest and draw ring[slot]
	const unsigned stream_every = getenv("NUDGE_STREAM_EVERY") ? (unsigned)atoi(getenv("NUDGE_STREAM_EVERY")) : 0u;
	void* ring = nullptr;
	if (stream_every) {
		HIP_OK(hipHostMalloc(&ring, sizeof(nh_Transform) * (size_t)nb * 4u, 0));
		NH_OK_(nh_stream_state(ctx, &bodies, nb, ring, 4u, stream_every));
	}

	uint64_t sum_contacts = 0;
	HIP_OK(hipStreamSynchronize(stream));
	auto t0 = std::chrono::steady_clock::now();
	if (one_call) {
		nh_StepArgs args = {};
		args.active_bodies = &active; args.contacts = &contacts; args.bodies = &bodies; args.colliders = &colliders; args.body_connections = &connections;
		args.contact_cache = &cache; args.arena = arena; args.time_step = time_step; args.damping_rate = damping_rate; args.iterations = iterations;
		args.gravity[0] = gravity[0]; args.gravity[1] = gravity[1]; args.gravity[2] = gravity[2];
		NH_OK_(nh_step(ctx, &args, steps));
	}
	for (unsigned n = 0; n < (one_call ? 0u : steps); ++n) {
		nh_Arena temporary = arena;                                   // same scratch every sub-step (example/main.cpp:282)
		NH_OK_(nh_collide(ctx, &active, &contacts, &bodies, &colliders, &connections, temporary));
		NH_OK_(nh_apply_gravity_damping(ctx, &active, &bodies, time_step, gravity, damping_rate));
		nh_ContactImpulseData* impulses = nullptr;
		NH_OK_(nh_read_cached_impulses(ctx, &cache, &contacts, &temporary, &impulses));
		nh_ContactConstraintData* constraints = nullptr;
		NH_OK_(nh_setup_contact_constraints(ctx, &active, &contacts, &bodies, impulses, &temporary, &constraints));
		NH_OK_(nh_apply_impulses(ctx, constraints, &bodies, iterations));   // all sweeps in one call: rows never leave registers
		NH_OK_(nh_update_cached_impulses(ctx, constraints, impulses));
		NH_OK_(nh_write_cached_impulses(ctx, &cache, &contacts, impulses));
		NH_OK_(nh_advance(ctx, &active, &bodies, time_step));
		nh_Counts c;                                                  // (only for the statistic below; costs a synchronisation)
		NH_OK_(nh_read_counts(ctx, &c));
		if (c.error) { fprintf(stderr, "device error: %s\n", nh_error_string((int)c.error)); return 5; }
		sum_contacts += c.contacts;
	}
	NH_OK_(nh_synchronize(ctx));
	double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	nh_Counts c;
	NH_OK_(nh_read_counts(ctx, &c));
	if (stream_every) {
		nh_StreamInfo si;
		NH_OK_(nh_stream_latest(ctx, &si));
		const nh_Transform* frame = static_cast<const nh_Transform*>(ring) + (size_t)si.slot * nb;
		printf("resident_replay: state stream: %llu frames (%llu dropped), newest landed: sub-step %llu in slot %u, body 1 at (%.3f, %.3f, %.3f)\n", (unsigned long long)si.frames,
		       (unsigned long long)si.dropped, (unsigned long long)si.step, si.slot, si.valid && nb > 1 ? frame[1].position[0] : 0.0f, si.valid && nb > 1 ? frame[1].position[1] : 0.0f,
		       si.valid && nb > 1 ? frame[1].position[2] : 0.0f);
		NH_OK_(nh_stream_state(ctx, nullptr, 0, nullptr, 0, 0));
	}

	HIP_OK(hipMemcpy(h_xf.data(), bodies.transforms, sizeof(nh_Transform) * nb, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(h_mom.data(), bodies.momentum, sizeof(nh_BodyMomentum) * nb, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(h_idle.data(), bodies.idle_counters, nb, hipMemcpyDeviceToHost));
	FILE* o = fopen(argv[2], "wb");
	if (!o) { perror(argv[2]); return 2; }
	fwrite(h_xf.data(), sizeof(nh_Transform), nb, o);
	fwrite(h_mom.data(), sizeof(nh_BodyMomentum), nb, o);
	fwrite(h_idle.data(), 1, nb, o);
	fwrite(&sum_contacts, 8, 1, o);
	fwrite(&c.contacts, 4, 1, o);
	fwrite(&c.active_bodies, 4, 1, o);
	fclose(o);
	printf("resident_replay: %u steps, %u bodies, last step %u contacts, %u active, sum of contacts %llu, %.1f steps/s\n", steps, nb, c.contacts, c.active_bodies,
	       (unsigned long long)sum_contacts, steps / seconds);
	nh_destroy(ctx);
	return 0;
}
