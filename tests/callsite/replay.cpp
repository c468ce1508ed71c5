// tests/callsite/replay.cpp -- a CALL SITE of the reference API, written once, linked twice.
//
// This is what an adopter's code looks like: it includes the reference's `nudge.h`, owns every array in host
// memory (reference layouts, 16-bit indices), and drives one sub-step the way the reference's sample does
// (call order and arena handling of example/main.cpp:274-328; the gravity/damping loop runs on the host between
// collide() and read_cached_impulses()).  tests/callsite/Makefile links this file
//   (a) against the reference's own nudge.cpp                    -> oracle/_ref/replay_ref   (the checker)
//   (b) against nudge_amd/compat/nudge_compat.cpp + libnudge_hip -> nudge_amd/compat/_build/replay_hip
// and tests/test_gpu_callsite.py requires the two binaries to produce bit-identical worlds.
//
// usage: replay <scene.bin> <out.bin> <steps> [mode]
// mode (what the call site does besides the plain loop; every variant is legal use of the reference API):
//   plain             (default)
//   zero_iterations   never calls apply_impulses: the warm start applied by setup_contact_constraints (nudge.cpp:4563-4632) is all the solver does
//   peek_after_setup  reads bodies.momentum right after setup_contact_constraints (folds it into the checksum): the warm start must be visible there
//   custom_impulses   user impulses between two apply_impulses calls, the way custom constraints interleave (example/main.cpp:314-317)
//   two_worlds        the plain loop on TWO copies of the scene, each with arrays of its own, stepped by two threads at once: the reference has no global
//                     mutable state and is re-entrant for disjoint worlds (SURVEY 8(b) threading); both worlds must end up alike (exit code 3 otherwise)
//   custom_contacts   contacts appended behind the list collide() returned ("Custom contacts can be added here", example/main.cpp:287): a soft
//                     "shelf" that the first dynamic bodies rest against, with tags no collider pair produces
// scene.bin: u32 header { bodies, boxes, spheres, iterations, max_contacts } + f32 { time_step, gravity, damping_rate }
//            then Transform[bodies] BodyProperties[bodies] BodyMomentum[bodies] u8 idle[bodies]
//            u16 box_tags[boxes] BoxCollider[boxes] Transform[boxes] u16 sphere_tags[spheres] SphereCollider[spheres] Transform[spheres]
// out.bin:   Transform[bodies] BodyMomentum[bodies] u8 idle[bodies] u64 sum_of_contact_counts u32 last_contact_count u32 last_active_count u64 checksum
#include <nudge.h>

#include <stdint.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

template<class T>
static T* aligned_array(size_t n) {
	void* p = nullptr;
	if (posix_memalign(&p, 64, sizeof(T) * (n ? n : 1) + 64)) abort();
	memset(p, 0, sizeof(T) * (n ? n : 1) + 64);
	return static_cast<T*>(p);
}

template<class T>
static void read_array(FILE* f, T* dst, size_t n) {
	if (n && fread(dst, sizeof(T), n, f) != n) { fprintf(stderr, "replay: short scene file\n"); exit(2); }
}

static int world_main(int argc, char** argv, bool write_output, std::vector<unsigned char>* final_state);

int main(int argc, char** argv) {
	if (argc < 4) { fprintf(stderr, "usage: %s scene.bin out.bin steps [mode]\n", argv[0]); return 2; }
	if (argc > 4 && !strcmp(argv[4], "two_worlds")) {
		std::vector<unsigned char> a, b;
		int rb = 0;
		std::thread other([&] { rb = world_main(argc, argv, false, &b); });
		const int ra = world_main(argc, argv, true, &a);
		other.join();
		if (ra || rb) return ra ? ra : rb;
		if (a != b) { fprintf(stderr, "replay: the two worlds differ\n"); return 3; }
		return 0;
	}
	return world_main(argc, argv, true, nullptr);
}

static int world_main(int argc, char** argv, bool write_output, std::vector<unsigned char>* final_state) {
	FILE* f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	uint32_t head[5];
	float par[3];
	read_array(f, head, 5);
	read_array(f, par, 3);
	const uint32_t nb = head[0], nbox = head[1], nsph = head[2], iterations = head[3], max_contacts = head[4];
	const float time_step = par[0], gravity = par[1], damping_rate = par[2];
	const unsigned steps = (unsigned)atoi(argv[3]);
	const char* mode = argc > 4 ? argv[4] : "plain";
	const bool zero_iterations = !strcmp(mode, "zero_iterations"), peek = !strcmp(mode, "peek_after_setup"), custom = !strcmp(mode, "custom_impulses");
	const bool custom_contacts = !strcmp(mode, "custom_contacts");
	if (!zero_iterations && !peek && !custom && !custom_contacts && strcmp(mode, "two_worlds") && strcmp(mode, "plain") && strcmp(mode, "exact")) { fprintf(stderr, "replay: unknown mode %s\n", mode); return 2; }
	uint64_t checksum = 1469598103934665603ull;      // FNV-1a over whatever the mode looks at

	nudge::BodyData bodies = {};
	bodies.transforms = aligned_array<nudge::Transform>(nb);
	bodies.properties = aligned_array<nudge::BodyProperties>(nb);
	bodies.momentum = aligned_array<nudge::BodyMomentum>(nb);
	bodies.idle_counters = aligned_array<uint8_t>(nb);
	bodies.count = nb;
	read_array(f, bodies.transforms, nb);
	read_array(f, bodies.properties, nb);
	read_array(f, bodies.momentum, nb);
	read_array(f, bodies.idle_counters, nb);

	nudge::ColliderData colliders = {};
	colliders.boxes.tags = aligned_array<uint16_t>(nbox);
	colliders.boxes.data = aligned_array<nudge::BoxCollider>(nbox);
	colliders.boxes.transforms = aligned_array<nudge::Transform>(nbox);
	colliders.boxes.count = nbox;
	read_array(f, colliders.boxes.tags, nbox);
	read_array(f, colliders.boxes.data, nbox);
	read_array(f, colliders.boxes.transforms, nbox);
	colliders.spheres.tags = aligned_array<uint16_t>(nsph);
	colliders.spheres.data = aligned_array<nudge::SphereCollider>(nsph);
	colliders.spheres.transforms = aligned_array<nudge::Transform>(nsph);
	colliders.spheres.count = nsph;
	read_array(f, colliders.spheres.tags, nsph);
	read_array(f, colliders.spheres.data, nsph);
	read_array(f, colliders.spheres.transforms, nsph);
	fclose(f);

	nudge::ContactData contacts = {};
	contacts.data = aligned_array<nudge::Contact>(max_contacts);
	contacts.bodies = aligned_array<nudge::BodyPair>(max_contacts);
	contacts.tags = aligned_array<uint64_t>(max_contacts);
	contacts.sleeping_pairs = aligned_array<uint32_t>(max_contacts);
	contacts.capacity = max_contacts;

	nudge::ContactCache cache = {};
	cache.tags = aligned_array<uint64_t>(max_contacts);
	cache.data = aligned_array<nudge::CachedContactImpulse>(max_contacts);
	cache.capacity = max_contacts;

	nudge::ActiveBodies active = {};
	active.indices = aligned_array<uint16_t>(nb);
	active.capacity = nb;

	nudge::Arena arena = {};
	arena.size = ((size_t)64 << 20) + (size_t)max_contacts * 1400;
	arena.data = aligned_array<uint8_t>(arena.size);

	uint64_t sum_contacts = 0;
	const auto t0 = std::chrono::steady_clock::now();
	for (unsigned n = 0; n < steps; ++n) {
		nudge::Arena temporary = arena;                       // same scratch every sub-step
		nudge::BodyConnections connections = {};
		nudge::collide(&active, &contacts, bodies, colliders, connections, temporary);
		sum_contacts += contacts.count;
		if (custom_contacts) {
			// "Custom contacts can be added here" (example/main.cpp:287): every one of the first dynamic bodies gets two contacts with the static world
			// -- a floor under it at y = 3 (only while it is below) and a gentle sideways push -- tagged with collider tags nobody uses
			for (unsigned b = 1; b < nb && b <= 12 && contacts.count + 2 <= contacts.capacity; ++b) {
				const float* p = bodies.transforms[b].position;
				for (unsigned k = 0; k < 2; ++k) {
					if (k == 0 && p[1] > 3.0f) continue;
					nudge::Contact c = {};
					c.position[0] = p[0]; c.position[1] = k == 0 ? p[1] - 0.25f : p[1]; c.position[2] = p[2];
					c.penetration = k == 0 ? (3.0f - p[1] < 0.2f ? 3.0f - p[1] : 0.2f) : 0.01f;
					c.normal[0] = k == 0 ? 0.0f : 1.0f; c.normal[1] = k == 0 ? 1.0f : 0.0f; c.normal[2] = 0.0f;       // from a (the static world) towards b
					c.friction = 0.25f;
					const unsigned i = contacts.count++;
					contacts.data[i] = c;
					contacts.bodies[i].a = 0; contacts.bodies[i].b = (uint16_t)b;
					contacts.tags[i] = ((uint64_t)(0x7000u + k) | ((uint64_t)(0x7100u + b) << 16)) << 32 | (uint64_t)(k + 1u);
				}
			}
		}

		// caller-side gravity and damping over the active list
		const float damping = 1.0f - time_step * damping_rate;
		for (unsigned i = 0; i < active.count; ++i) {
			nudge::BodyMomentum& m = bodies.momentum[active.indices[i]];
			m.velocity[1] -= gravity * time_step;
			for (int k = 0; k < 3; ++k) { m.velocity[k] *= damping; m.angular_velocity[k] *= damping; }
		}

		nudge::ContactImpulseData* impulses = nudge::read_cached_impulses(cache, contacts, &temporary);
		nudge::ContactConstraintData* constraints = nudge::setup_contact_constraints(active, contacts, bodies, impulses, &temporary);
		if (peek) {
			const unsigned char* raw = reinterpret_cast<const unsigned char*>(bodies.momentum);
			for (size_t k = 0; k < sizeof(nudge::BodyMomentum) * (size_t)nb; ++k) { checksum ^= raw[k]; checksum *= 1099511628211ull; }
		}
		for (unsigned i = 0; i < (zero_iterations ? 0u : iterations); ++i) {
			nudge::apply_impulses(constraints, bodies);
			if (custom) {
				// a stand-in for a user constraint: small fixed impulses on the first few dynamic bodies after every sweep
				for (unsigned b = 1; b < nb && b <= 8; ++b) { bodies.momentum[b].angular_velocity[1] += 0.01f; bodies.momentum[b].velocity[0] -= 0.002f; }
			}
		}
		nudge::update_cached_impulses(constraints, impulses);
		nudge::write_cached_impulses(&cache, contacts, impulses);
		nudge::advance(active, bodies, time_step);
	}
	const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

	if (final_state) {
		const unsigned char* t = reinterpret_cast<const unsigned char*>(bodies.transforms);
		const unsigned char* m = reinterpret_cast<const unsigned char*>(bodies.momentum);
		final_state->assign(t, t + sizeof(nudge::Transform) * (size_t)nb);
		final_state->insert(final_state->end(), m, m + sizeof(nudge::BodyMomentum) * (size_t)nb);
		final_state->insert(final_state->end(), bodies.idle_counters, bodies.idle_counters + nb);
	}
	if (!write_output) return 0;
	FILE* o = fopen(argv[2], "wb");
	if (!o) { perror(argv[2]); return 2; }
	fwrite(bodies.transforms, sizeof(nudge::Transform), nb, o);
	fwrite(bodies.momentum, sizeof(nudge::BodyMomentum), nb, o);
	fwrite(bodies.idle_counters, 1, nb, o);
	fwrite(&sum_contacts, 8, 1, o);
	fwrite(&contacts.count, 4, 1, o);
	fwrite(&active.count, 4, 1, o);
	fwrite(&checksum, 8, 1, o);
	fclose(o);
	printf("replay: %u steps, %u bodies, last step %u contacts, %u active, sum of contacts %llu, %.1f steps/s\n", steps, nb, contacts.count, active.count, (unsigned long long)sum_contacts, steps / seconds);
	return 0;
}
