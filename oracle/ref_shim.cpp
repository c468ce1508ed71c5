// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE (the checker), never linked into the product.
//
// A thin extern "C" harness around the UNMODIFIED reference engine, compiled from the sources
// where they lie (/root/reference/nudge.cpp + nudge.h) by oracle/Makefile into
// oracle/_ref/libnudge_ref_{exact,shipped}.so.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load those libraries.
//
// It owns one "world" the way the reference's sample app does (example/main.cpp:359-395: caller
// allocated SoA arrays + a bump arena) and replays the reference's own step loop
// (example/main.cpp:274-328) headless, stage by stage, so that tests can diff stage boundaries.
//
// Nothing here restates arithmetic: every number comes out of the reference's own functions
// (nudge.h:134-146).  The two opaque result structs are mirrored below ONLY to expose the solver's
// contact order (constraint_to_contact) to the scheduler-replay test.
#include "nudge.h"

#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <atomic>
#include <chrono>

namespace {

// Mirrors of the private result structs (reference nudge.cpp:4011-4019 and 4160-4168); layout only.
struct ImpulseDataMirror {
	uint32_t* sorted_contacts;
	nudge::CachedContactImpulse* culled_data;
	uint64_t* culled_tags;
	unsigned culled_count;
	nudge::CachedContactImpulse* data;
};

struct ConstraintDataMirror {
	unsigned contact_count;
	void* momentum_to_velocity;
	uint32_t* constraint_to_contact;
	void* constraints;
	void* constraint_states;
	unsigned constraint_batches;
};

template<class T>
T* alloc_array(size_t n) {
	void* p = _mm_malloc(sizeof(T) * (n ? n : 1) + 64, 64);
	memset(p, 0, sizeof(T) * (n ? n : 1) + 64);
	return static_cast<T*>(p);
}

}

extern "C" {

struct RefWorld {
	nudge::Arena arena;
	nudge::BodyData bodies;
	nudge::ColliderData colliders;
	nudge::ContactData contacts;
	nudge::ContactCache cache;
	nudge::ActiveBodies active;
	nudge::BodyConnections connections;

	uint32_t max_bodies, max_boxes, max_spheres, max_contacts;

	// per-step transient state (valid between the staged calls of one step)
	nudge::Arena temporary;
	nudge::ContactImpulseData* impulses;
	nudge::ContactConstraintData* constraints;

	// statistics
	uint64_t sum_contacts;
	uint64_t steps_done;
};

// Plain-pointer view for ctypes/numpy.
struct RefWorldView {
	void* body_transforms;      // nudge::Transform[max_bodies]      (32 B)
	void* body_properties;      // nudge::BodyProperties[max_bodies] (16 B)
	void* body_momentum;        // nudge::BodyMomentum[max_bodies]   (32 B)
	uint8_t* idle_counters;
	uint16_t* box_tags;
	void* box_data;             // nudge::BoxCollider[max_boxes]     (16 B)
	void* box_transforms;       // nudge::Transform[max_boxes]
	uint16_t* sphere_tags;
	void* sphere_data;          // nudge::SphereCollider[max_spheres] (4 B)
	void* sphere_transforms;
	void* contact_data;         // nudge::Contact[max_contacts]      (32 B)
	void* contact_bodies;       // nudge::BodyPair[max_contacts]     (2 x u16)
	uint64_t* contact_tags;
	uint32_t* sleeping_pairs;
	uint64_t* cache_tags;
	void* cache_data;           // nudge::CachedContactImpulse[max_contacts] (16 B)
	uint16_t* active_indices;
	void* connections;          // nudge::BodyPair[max_bodies]
	uint32_t body_count, box_count, sphere_count;
	uint32_t contact_count, sleeping_count, cache_count, active_count, connection_count;
	uint32_t max_bodies, max_boxes, max_spheres, max_contacts;
};

void ref_set_ftz(int on) {
	_MM_SET_FLUSH_ZERO_MODE(on ? _MM_FLUSH_ZERO_ON : _MM_FLUSH_ZERO_OFF);
	_MM_SET_DENORMALS_ZERO_MODE(on ? _MM_DENORMALS_ZERO_ON : _MM_DENORMALS_ZERO_OFF);
}

const char* ref_variant() {
#ifdef NH_ORACLE_EXACT_INTRINSICS_H
	return "exact";
#else
	return "shipped";
#endif
}

int ref_simd_width() {
#ifdef __AVX2__
	return 8;
#else
	return 4;
#endif
}

RefWorld* ref_world_create(uint32_t max_bodies, uint32_t max_boxes, uint32_t max_spheres, uint32_t max_contacts, uint64_t arena_bytes) {
	RefWorld* w = new RefWorld();
	memset(w, 0, sizeof(*w));
	w->max_bodies = max_bodies;
	w->max_boxes = max_boxes;
	w->max_spheres = max_spheres;
	w->max_contacts = max_contacts;

	w->arena.size = arena_bytes;
	w->arena.data = _mm_malloc(arena_bytes, 4096);

	w->bodies.transforms = alloc_array<nudge::Transform>(max_bodies);
	w->bodies.properties = alloc_array<nudge::BodyProperties>(max_bodies);
	w->bodies.momentum = alloc_array<nudge::BodyMomentum>(max_bodies);
	w->bodies.idle_counters = alloc_array<uint8_t>(max_bodies);

	w->colliders.boxes.tags = alloc_array<uint16_t>(max_boxes);
	w->colliders.boxes.data = alloc_array<nudge::BoxCollider>(max_boxes);
	w->colliders.boxes.transforms = alloc_array<nudge::Transform>(max_boxes);
	w->colliders.spheres.tags = alloc_array<uint16_t>(max_spheres);
	w->colliders.spheres.data = alloc_array<nudge::SphereCollider>(max_spheres);
	w->colliders.spheres.transforms = alloc_array<nudge::Transform>(max_spheres);

	w->contacts.capacity = max_contacts;
	w->contacts.data = alloc_array<nudge::Contact>(max_contacts);
	w->contacts.bodies = alloc_array<nudge::BodyPair>(max_contacts);
	w->contacts.tags = alloc_array<uint64_t>(max_contacts);
	w->contacts.sleeping_pairs = alloc_array<uint32_t>(max_contacts);

	w->cache.capacity = max_contacts;
	w->cache.data = alloc_array<nudge::CachedContactImpulse>(max_contacts);
	w->cache.tags = alloc_array<uint64_t>(max_contacts);

	w->active.capacity = max_bodies;
	w->active.indices = alloc_array<uint16_t>(max_bodies);

	w->connections.data = alloc_array<nudge::BodyPair>(max_bodies);
	return w;
}

void ref_world_destroy(RefWorld* w) {
	if (!w) return;
	_mm_free(w->arena.data);
	_mm_free(w->bodies.transforms); _mm_free(w->bodies.properties); _mm_free(w->bodies.momentum); _mm_free(w->bodies.idle_counters);
	_mm_free(w->colliders.boxes.tags); _mm_free(w->colliders.boxes.data); _mm_free(w->colliders.boxes.transforms);
	_mm_free(w->colliders.spheres.tags); _mm_free(w->colliders.spheres.data); _mm_free(w->colliders.spheres.transforms);
	_mm_free(w->contacts.data); _mm_free(w->contacts.bodies); _mm_free(w->contacts.tags); _mm_free(w->contacts.sleeping_pairs);
	_mm_free(w->cache.data); _mm_free(w->cache.tags);
	_mm_free(w->active.indices);
	_mm_free(w->connections.data);
	delete w;
}

void ref_world_set_counts(RefWorld* w, uint32_t bodies, uint32_t boxes, uint32_t spheres, uint32_t connections) {
	w->bodies.count = bodies;
	w->colliders.boxes.count = boxes;
	w->colliders.spheres.count = spheres;
	w->connections.count = connections;
}

void ref_world_set_cache_count(RefWorld* w, uint32_t n) { w->cache.count = n; }
// (the reference lets the caller edit the contact list between collide and read_cached_impulses, example/main.cpp:287: a caller that REMOVES contacts shortens it)
void ref_world_set_contact_count(RefWorld* w, uint32_t n) { w->contacts.count = n; }

void ref_world_view(RefWorld* w, RefWorldView* v) {
	v->body_transforms = w->bodies.transforms;
	v->body_properties = w->bodies.properties;
	v->body_momentum = w->bodies.momentum;
	v->idle_counters = w->bodies.idle_counters;
	v->box_tags = w->colliders.boxes.tags;
	v->box_data = w->colliders.boxes.data;
	v->box_transforms = w->colliders.boxes.transforms;
	v->sphere_tags = w->colliders.spheres.tags;
	v->sphere_data = w->colliders.spheres.data;
	v->sphere_transforms = w->colliders.spheres.transforms;
	v->contact_data = w->contacts.data;
	v->contact_bodies = w->contacts.bodies;
	v->contact_tags = w->contacts.tags;
	v->sleeping_pairs = w->contacts.sleeping_pairs;
	v->cache_tags = w->cache.tags;
	v->cache_data = w->cache.data;
	v->active_indices = w->active.indices;
	v->connections = w->connections.data;
	v->body_count = w->bodies.count;
	v->box_count = w->colliders.boxes.count;
	v->sphere_count = w->colliders.spheres.count;
	v->contact_count = w->contacts.count;
	v->sleeping_count = w->contacts.sleeping_count;
	v->cache_count = w->cache.count;
	v->active_count = w->active.count;
	v->connection_count = w->connections.count;
	v->max_bodies = w->max_bodies;
	v->max_boxes = w->max_boxes;
	v->max_spheres = w->max_spheres;
	v->max_contacts = w->max_contacts;
}

// ---- staged step: the body of example/main.cpp:280-326, one call per line of that loop ----

void ref_stage_collide(RefWorld* w) {
	w->temporary = w->arena;                                                                   // main.cpp:282
	nudge::collide(&w->active, &w->contacts, w->bodies, w->colliders, w->connections, w->temporary); // main.cpp:286
}

// main.cpp:290-305 -- the caller-side gravity + damping loop (part of the timed step).
void ref_stage_gravity(RefWorld* w, float time_step, float gravity, float damping_rate) {
	float damping = 1.0f - time_step * damping_rate;
	float dv = gravity * time_step;
	for (unsigned i = 0; i < w->active.count; ++i) {
		unsigned index = w->active.indices[i];
		nudge::BodyMomentum& m = w->bodies.momentum[index];
		m.velocity[1] -= dv;
		m.velocity[0] *= damping;
		m.velocity[1] *= damping;
		m.velocity[2] *= damping;
		m.angular_velocity[0] *= damping;
		m.angular_velocity[1] *= damping;
		m.angular_velocity[2] *= damping;
	}
}

void ref_stage_read_cache(RefWorld* w) {
	w->impulses = nudge::read_cached_impulses(w->cache, w->contacts, &w->temporary);           // main.cpp:308
}

void ref_stage_setup(RefWorld* w) {
	w->constraints = nudge::setup_contact_constraints(w->active, w->contacts, w->bodies, w->impulses, &w->temporary); // main.cpp:311
}

void ref_stage_apply(RefWorld* w, uint32_t iterations) {
	for (unsigned i = 0; i < iterations; ++i)
		nudge::apply_impulses(w->constraints, w->bodies);                                      // main.cpp:314-317
}

void ref_stage_update(RefWorld* w) {
	nudge::update_cached_impulses(w->constraints, w->impulses);                                // main.cpp:320
}

void ref_stage_write_cache(RefWorld* w) {
	nudge::write_cached_impulses(&w->cache, w->contacts, w->impulses);                         // main.cpp:323
}

void ref_stage_advance(RefWorld* w, float time_step) {
	nudge::advance(w->active, w->bodies, time_step);                                           // main.cpp:326
}

// Introspection of the opaque results (test-only).
uint32_t ref_constraint_batches(RefWorld* w) {
	return reinterpret_cast<ConstraintDataMirror*>(w->constraints)->constraint_batches;
}

const uint32_t* ref_constraint_to_contact(RefWorld* w) {
	return reinterpret_cast<ConstraintDataMirror*>(w->constraints)->constraint_to_contact;
}

const uint32_t* ref_sorted_contacts(RefWorld* w) {
	return reinterpret_cast<ImpulseDataMirror*>(w->impulses)->sorted_contacts;
}

const void* ref_contact_impulses(RefWorld* w) {
	return reinterpret_cast<ImpulseDataMirror*>(w->impulses)->data;
}

uint32_t ref_culled_count(RefWorld* w) {
	return reinterpret_cast<ImpulseDataMirror*>(w->impulses)->culled_count;
}

void ref_world_step(RefWorld* w, uint32_t steps, uint32_t iterations, float time_step, float gravity, float damping_rate) {
	for (uint32_t n = 0; n < steps; ++n) {
		ref_stage_collide(w);
		ref_stage_gravity(w, time_step, gravity, damping_rate);
		ref_stage_read_cache(w);
		ref_stage_setup(w);
		ref_stage_apply(w, iterations);
		ref_stage_update(w);
		ref_stage_write_cache(w);
		ref_stage_advance(w, time_step);
		w->sum_contacts += w->contacts.count;
		w->steps_done += 1;
	}
}

uint64_t ref_world_sum_contacts(RefWorld* w) { return w->sum_contacts; }

// Tiled driver: independent worlds of <= 8191 dynamic colliders each are stepped by
// the unmodified reference.  The reference has no global mutable state, so tiles may run on
// different host threads.  Returns wall seconds.
double ref_tiles_step(RefWorld** worlds, uint32_t world_count, uint32_t steps, uint32_t iterations,
                      float time_step, float gravity, float damping_rate, uint32_t threads, int ftz) {
	if (threads < 1) threads = 1;
	std::atomic<uint32_t> next(0);
	auto t0 = std::chrono::steady_clock::now();
	auto worker = [&]() {
		ref_set_ftz(ftz);
		for (;;) {
			uint32_t i = next.fetch_add(1);
			if (i >= world_count) break;
			ref_world_step(worlds[i], steps, iterations, time_step, gravity, damping_rate);
		}
	};
	if (threads == 1) {
		worker();
	} else {
		std::vector<std::thread> pool;
		for (uint32_t t = 0; t < threads; ++t) pool.emplace_back(worker);
		for (auto& t : pool) t.join();
	}
	auto t1 = std::chrono::steady_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

}
