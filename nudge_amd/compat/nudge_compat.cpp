// nudge_compat.cpp -- the reference's C++ API (nudge.h:134-146, `namespace nudge`) implemented on libnudge_hip.so.
//
// This is the drop-in for EXISTING call sites: compile this file against the adopter's own `nudge.h` INSTEAD of
// `nudge.cpp` and link `-lnudge_hip`; `example/main.cpp:274-328` and `tests/main.cpp` then run unchanged, with
// every array still caller-owned HOST memory in the reference's 16-bit layouts.  Each entry point
//     1. mirrors the host arrays it reads into HBM (widening 16-bit tags / body indices to the 32-bit records of
//        include/nudge_hip.h),
//     2. calls the C-ABI function that replaces the reference function of the same name,
//     3. copies back exactly what the reference function writes (nudge.h comments / SURVEY 8(b) "outputs written
//        in place"), narrowing again.
// The opaque `ContactImpulseData*` / `ContactConstraintData*` live in the caller's Arena like the reference's
// (nudge.cpp:4022, 4174); they are small handles to the device-side objects.
//
// Because user code may touch the host arrays between any two calls (the gravity loop, custom constraints:
// example/main.cpp:288-305, 316), nothing is assumed to stay valid on the device across calls except what the
// reference itself keeps in opaque state.  That makes this layer PCIe-bound -- it is the compatibility path.
// The performance path is the device-resident C ABI (include/nudge_hip.h).
//
// Deviations from the reference, all forced by the implementation underneath:
//   * `contacts` come back in tag order (the order the reference's solver consumes) instead of emission order;
//     between collide() and read_cached_impulses() the caller may edit contact fields in place (friction,
//     penetration ...) and append contacts of its own behind the list (example/main.cpp:287: bump contacts.count),
//     but must not reorder or remove contacts;
//   * capacity overflows and arena exhaustion abort with a message (reference: assert / silent overflow);
//   * scratch does not come from the caller's Arena (it is device memory owned by this layer); only the two
//     opaque handles are allocated from it;
//   * the solver visits contacts in the reference's exact greedy batch order (NH_FLAG_EXACT_ORDER), so results
//     match an exact-math build of the reference bit for bit; the x86 approximate rsqrt/rcp of the shipped build
//     are not reproduced (their bits are CPU-vendor specific).
//
// Thread model: like the reference (no global mutable state, re-entrant for disjoint worlds, one thread per world at a time): every WORLD -- identified
// by the caller's arrays: bodies.transforms and contacts.data, which the reference API hands to every entry point that carries no opaque handle -- has
// its own session here (device mirrors, nh_context, HIP stream, mutex); sessions are found through a small registry.  Two threads stepping two worlds
// run concurrently; two threads inside ONE world are the caller's bug in the reference too (SURVEY 8(b) threading) and are serialised.
#include <nudge.h>

#include "../../include/nudge_hip.h"

#include <hip/hip_runtime_api.h>

#include <map>
#include <memory>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

[[noreturn]] void die(const char* what, const char* why) {
	fprintf(stderr, "nudge (HIP backend): %s: %s\n", what, why);
	abort();      // the reference's error convention is assert() (nudge.cpp:1000, 3010, 4118)
}

void hip_check(hipError_t e, const char* what) {
	if (e != hipSuccess) die(what, hipGetErrorString(e));
}

// growable device buffer
struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	void reserve(size_t bytes) {
		if (bytes <= cap) return;
		if (p) hip_check(hipFree(p), "hipFree");
		size_t want = bytes + bytes / 2 + 256;
		hip_check(hipMalloc(&p, want), "hipMalloc");
		hip_check(hipMemset(p, 0, want), "hipMemset");
		cap = want;
	}
	template<class T> T* as() const { return static_cast<T*>(p); }
};

void upload(DevBuf& d, const void* src, size_t bytes) {
	d.reserve(bytes ? bytes : 16);
	if (bytes) hip_check(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
}

void download(void* dst, const void* src, size_t bytes) {
	if (bytes) hip_check(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
}

struct Session {
	std::mutex lock;
	nh_context* ctx = nullptr;
	// device mirrors of the caller's arrays
	DevBuf body_xf, body_props, body_mom, body_idle;
	DevBuf box_tags, box_data, box_xf, sph_tags, sph_data, sph_xf;
	DevBuf connections;
	DevBuf c_data, c_bodies, c_tags, c_features, c_sleeping;
	DevBuf k_tags, k_features, k_data;
	DevBuf active;
	DevBuf arena;
	DevBuf positions;
	// the step in flight
	nh_Arena temporary = { nullptr, 0 };
	nh_ContactData contacts = {};
	nh_ActiveBodies active_bodies = {};
	uint32_t body_count = 0;
	// host staging
	std::vector<uint32_t> u32;
	std::vector<uint64_t> u64;
	std::vector<uint32_t> feat;
	// contacts appended by the caller after collide() ("Custom contacts can be added here", example/main.cpp:287): the device list is merged into tag
	// order (nh_append_contacts) while the caller's arrays keep their order; perm[i] = where the caller's contact i sits on the device
	std::vector<uint32_t> perm;
	std::vector<nudge::Contact> staged_contacts;
	uint32_t host_contacts = 0;          // contacts.count as the caller sees it

	hipStream_t stream = nullptr;
	int device = 0;
	// every entry point, after taking the session's lock: the world's device is the calling thread's current one (a world first seen by another thread would
	// otherwise get its mirrors on whatever device THIS thread happens to have current)
	void bind() { init(); hip_check(hipSetDevice(device), "hipSetDevice"); }
	void init() {
		if (ctx) return;
		if (const char* s = getenv("NUDGE_HIP_DEVICE")) device = atoi(s);
		hip_check(hipSetDevice(device), "hipSetDevice");
		// a stream of the world's own (a blocking one: the synchronous copies of this layer, on the legacy default stream, stay ordered with its kernels)
		hip_check(hipStreamCreate(&stream), "hipStreamCreate");
		int rc = nh_create(&ctx, device, stream, NH_FLAG_SYNC_COUNTS | NH_FLAG_EXACT_ORDER);
		if (rc) die("nh_create", nh_error_string(rc));
		nh_set_tag_bits(ctx, 16);       // the reference's collider tags are 16-bit (nudge.h:86, 93)
	}
};

// worlds by the caller's arrays.  The registry lock is held only while looking up / registering; a session's own lock for the duration of a call.
struct Registry {
	std::mutex lock;
	std::map<const void*, Session*> by_key;
	std::vector<std::unique_ptr<Session>> all;
	// (constructed on first use, i.e. after the HIP runtime: destroyed before it at exit -- contexts and streams go while the runtime is still alive)
	~Registry() {
		for (auto& s : all) {
			if (s->ctx) { nh_destroy(s->ctx); s->ctx = nullptr; }
			if (s->stream) { hipStreamDestroy(s->stream); s->stream = nullptr; }
		}
	}
};
Registry& registry() { static Registry r; return r; }

// collide(): the world of these bodies / this contact list, created on first sight
Session& session_of_world(const void* body_transforms, const void* contact_data) {
	Registry& r = registry();
	std::lock_guard<std::mutex> guard(r.lock);
	Session* s = nullptr;
	auto a = r.by_key.find(body_transforms);
	if (a != r.by_key.end()) s = a->second;
	if (!s) { auto b = r.by_key.find(contact_data); if (b != r.by_key.end()) s = b->second; }
	if (!s) { r.all.emplace_back(new Session()); s = r.all.back().get(); }
	if (body_transforms) r.by_key[body_transforms] = s;
	if (contact_data) r.by_key[contact_data] = s;
	return *s;
}

// any later entry point without an opaque handle: the world collide() registered these arrays for
Session& session_by_key(const void* key, const char* who) {
	Registry& r = registry();
	std::lock_guard<std::mutex> guard(r.lock);
	auto a = r.by_key.find(key);
	if (a != r.by_key.end()) return *a->second;
	// the reference has no call order: advance() or read_cached_impulses() on arrays collide() has not seen yet simply start a world
	(void)who;
	r.all.emplace_back(new Session());
	Session* s = r.all.back().get();
	if (key) r.by_key[key] = s;
	return *s;
}

void check(int rc, const char* what) { if (rc) die(what, nh_error_string(rc)); }

// The caller is done with a world (its arrays are about to be freed or re-used): the session found through either array goes, with its device mirrors, context
// and stream.  Without this call sessions live until the process ends (the reference has nothing to release, so an unmodified call site never tells).
void release_world(const void* key) {
	Registry& r = registry();
	std::lock_guard<std::mutex> guard(r.lock);
	auto a = r.by_key.find(key);
	if (a == r.by_key.end()) return;
	Session* s = a->second;
	for (auto it = r.by_key.begin(); it != r.by_key.end();) { if (it->second == s) it = r.by_key.erase(it); else ++it; }
	for (auto it = r.all.begin(); it != r.all.end(); ++it) if (it->get() == s) {
		{
			std::lock_guard<std::mutex> g2(s->lock);
			if (s->ctx) { hipSetDevice(s->device); nh_destroy(s->ctx); s->ctx = nullptr; }
			if (s->stream) { hipStreamDestroy(s->stream); s->stream = nullptr; }
			DevBuf* bufs[] = { &s->body_xf, &s->body_props, &s->body_mom, &s->body_idle, &s->box_tags, &s->box_data, &s->box_xf, &s->sph_tags, &s->sph_data, &s->sph_xf, &s->connections,
			                   &s->c_data, &s->c_bodies, &s->c_tags, &s->c_features, &s->c_sleeping, &s->k_tags, &s->k_features, &s->k_data, &s->active, &s->arena, &s->positions };
			for (DevBuf* b : bufs) if (b->p) { hipFree(b->p); b->p = nullptr; b->cap = 0; }
		}
		r.all.erase(it);
		break;
	}
}

nh_BodyData device_bodies(Session& s, const nudge::BodyData& b, bool xf, bool props, bool mom, bool idle) {
	if (xf) upload(s.body_xf, b.transforms, sizeof(nudge::Transform) * b.count);
	if (props) upload(s.body_props, b.properties, sizeof(nudge::BodyProperties) * b.count);
	if (mom) upload(s.body_mom, b.momentum, sizeof(nudge::BodyMomentum) * b.count);
	if (idle) upload(s.body_idle, b.idle_counters, b.count);
	// the records not uploaded by this call keep whatever an earlier call mirrored (or zeros)
	s.body_xf.reserve(sizeof(nudge::Transform) * b.count);
	s.body_props.reserve(sizeof(nudge::BodyProperties) * b.count);
	s.body_mom.reserve(sizeof(nudge::BodyMomentum) * b.count);
	s.body_idle.reserve(b.count ? b.count : 16);
	nh_BodyData d = { s.body_xf.as<nh_Transform>(), s.body_props.as<nh_BodyProperties>(), s.body_mom.as<nh_BodyMomentum>(), s.body_idle.as<uint8_t>(), b.count };
	s.body_count = b.count;
	return d;
}

void upload_tags16(Session& s, DevBuf& d, const uint16_t* tags, uint32_t n) {
	s.u32.resize(n ? n : 1);
	for (uint32_t i = 0; i < n; ++i) s.u32[i] = tags[i];
	upload(d, s.u32.data(), sizeof(uint32_t) * n);
}

// reference u64 contact tag = feature | (a16 | b16 << 16) << 32 (nudge.h:76)  <->  wide (a | b << 32, feature)
inline uint64_t narrow_tag(uint64_t wide, uint32_t feature) {
	uint64_t a = wide & 0xffffu, b = (wide >> 32) & 0xffffu;
	return ((a | (b << 16)) << 32) | feature;
}
inline void widen_tag(uint64_t tag, uint64_t& wide, uint32_t& feature) {
	uint64_t hi = tag >> 32;
	wide = (hi & 0xffffu) | (((hi >> 16) & 0xffffu) << 32);
	feature = (uint32_t)tag;
}

// the caller may have edited contact fields in place (custom friction etc.): its array goes up again, through the permutation if contacts were appended
void upload_contact_data(Session& s, const nudge::Contact* data, uint32_t n) {
	if (s.perm.empty()) { upload(s.c_data, data, sizeof(nudge::Contact) * (size_t)n); return; }
	s.staged_contacts.resize(n ? n : 1);
	for (uint32_t i = 0; i < n; ++i) s.staged_contacts[s.perm[i]] = data[i];
	upload(s.c_data, s.staged_contacts.data(), sizeof(nudge::Contact) * (size_t)n);
}

// opaque results: handles in the caller's arena (reference nudge.cpp:4022, 4174 allocate theirs there too)
struct ImpulseHandle { uint64_t magic; nh_ContactImpulseData* dev; Session* world; };
struct ConstraintHandle { uint64_t magic; nh_ContactConstraintData* dev; Session* world; };
const uint64_t IMPULSE_MAGIC = 0x6e68496d70756c73ull, CONSTRAINT_MAGIC = 0x6e68436f6e737472ull;

void* arena_take(nudge::Arena* arena, uintptr_t size, uintptr_t alignment) {
	uintptr_t data = (uintptr_t)arena->data, end = data + arena->size;
	data = (data + alignment - 1) & ~(alignment - 1);
	if (data + size > end) die("Arena", "out of memory");       // reference: assert (nudge.cpp:1000-1039)
	arena->data = (void*)(data + size);
	arena->size = end - (data + size);
	return (void*)data;
}

}  // namespace

namespace nudge {

// nudge.h:134, nudge.cpp:3000-4009
void collide(ActiveBodies* active_bodies, ContactData* contacts, BodyData bodies, ColliderData colliders, BodyConnections body_connections, Arena /*temporary: scratch is device memory*/) {
	Session& s = session_of_world(bodies.transforms, contacts->data);
	std::lock_guard<std::mutex> guard(s.lock);
	s.bind();
	const uint32_t nbox = colliders.boxes.count, nsph = colliders.spheres.count, K = contacts->capacity;

	nh_BodyData db = device_bodies(s, bodies, true, false, false, true);
	check(nh_bodies_changed(s.ctx), "bodies_changed");       // idle counters come from the caller's host array every step
	upload_tags16(s, s.box_tags, colliders.boxes.tags, nbox);
	upload(s.box_data, colliders.boxes.data, sizeof(BoxCollider) * nbox);
	upload(s.box_xf, colliders.boxes.transforms, sizeof(Transform) * nbox);
	upload_tags16(s, s.sph_tags, colliders.spheres.tags, nsph);
	upload(s.sph_data, colliders.spheres.data, sizeof(SphereCollider) * nsph);
	upload(s.sph_xf, colliders.spheres.transforms, sizeof(Transform) * nsph);
	nh_ColliderData dc = {};
	dc.boxes.tags = s.box_tags.as<uint32_t>(); dc.boxes.data = s.box_data.as<nh_BoxCollider>(); dc.boxes.transforms = s.box_xf.as<nh_Transform>(); dc.boxes.count = nbox;
	dc.spheres.tags = s.sph_tags.as<uint32_t>(); dc.spheres.data = s.sph_data.as<nh_SphereCollider>(); dc.spheres.transforms = s.sph_xf.as<nh_Transform>(); dc.spheres.count = nsph;

	s.u32.resize(2 * (size_t)body_connections.count + 2);
	for (uint32_t i = 0; i < body_connections.count; ++i) { s.u32[2 * i] = body_connections.data[i].a; s.u32[2 * i + 1] = body_connections.data[i].b; }
	upload(s.connections, s.u32.data(), sizeof(uint32_t) * 2 * body_connections.count);
	nh_BodyConnections dcon = { s.connections.as<nh_BodyPair>(), body_connections.count };

	s.c_data.reserve(sizeof(nh_Contact) * (size_t)K); s.c_bodies.reserve(sizeof(nh_BodyPair) * (size_t)K);
	s.c_tags.reserve(8 * (size_t)K); s.c_features.reserve(4 * (size_t)K); s.c_sleeping.reserve(8 * (size_t)K);
	s.active.reserve(4 * (size_t)(active_bodies->capacity ? active_bodies->capacity : 4));
	s.contacts = nh_ContactData{ s.c_data.as<nh_Contact>(), s.c_bodies.as<nh_BodyPair>(), s.c_tags.as<uint64_t>(), s.c_features.as<uint32_t>(), K, 0,
	                             contacts->sleeping_pairs ? s.c_sleeping.as<uint64_t>() : nullptr, 0 };
	s.active_bodies = nh_ActiveBodies{ s.active.as<uint32_t>(), active_bodies->capacity, 0 };

	// device scratch for the whole step (collide by value; read_cached_impulses / setup advance the copy)
	const uint64_t ncol = (uint64_t)nbox + nsph;
	uint64_t cells = 1; while (cells < 4 * ncol) cells <<= 1;
	if (cells < (1u << 16)) cells = 1u << 16;
	if (cells > (1u << 24)) cells = 1u << 24;
	// broadphase pairs are not bounded by the contact capacity (a pit of balls holds more overlapping AABB pairs than contacts): like the
	// reference, whose pair list is only bounded by the arena (nudge.cpp:3473), size the pair buffer from the colliders as well
	const uint64_t pair_cap = (uint64_t)K / 2 + 1024 > 16 * ncol + 1024 ? (uint64_t)K / 2 + 1024 : 16 * ncol + 1024;
	check(nh_set_pair_capacity(s.ctx, (uint32_t)pair_cap), "set_pair_capacity");
	const size_t arena_bytes = ((size_t)64 << 20) + ncol * 200 + cells * 8 + (size_t)K * 320 + (size_t)pair_cap * 216 + (size_t)bodies.count * 64;
	s.arena.reserve(arena_bytes);
	s.temporary = nh_Arena{ s.arena.p, s.arena.cap };

	check(nh_collide(s.ctx, &s.active_bodies, &s.contacts, &db, &dc, &dcon, s.temporary), "collide");

	// outputs written in place by the reference: contacts (data, bodies, tags, count, sleeping pairs), active bodies
	const uint32_t n = s.contacts.count;
	contacts->count = n;
	s.host_contacts = n; s.perm.clear();
	download(contacts->data, s.c_data.p, sizeof(Contact) * (size_t)n);
	s.u32.resize(2 * (size_t)n + 2); s.u64.resize((size_t)n + 1); s.feat.resize((size_t)n + 1);
	download(s.u32.data(), s.c_bodies.p, 8 * (size_t)n);
	download(s.u64.data(), s.c_tags.p, 8 * (size_t)n);
	download(s.feat.data(), s.c_features.p, 4 * (size_t)n);
	for (uint32_t i = 0; i < n; ++i) {
		contacts->bodies[i].a = (uint16_t)s.u32[2 * i];
		contacts->bodies[i].b = (uint16_t)s.u32[2 * i + 1];
		contacts->tags[i] = narrow_tag(s.u64[i], s.feat[i]);
	}
	contacts->sleeping_count = 0;
	if (contacts->sleeping_pairs) {
		const uint32_t ns = s.contacts.sleeping_count;
		s.u64.resize((size_t)ns + 1);
		download(s.u64.data(), s.c_sleeping.p, 8 * (size_t)ns);
		for (uint32_t i = 0; i < ns; ++i) contacts->sleeping_pairs[i] = (uint32_t)((s.u64[i] & 0xffffu) | (((s.u64[i] >> 32) & 0xffffu) << 16));
		contacts->sleeping_count = ns;
	}
	const uint32_t na = s.active_bodies.count;
	s.u32.resize((size_t)na + 1);
	download(s.u32.data(), s.active.p, 4 * (size_t)na);
	for (uint32_t i = 0; i < na; ++i) active_bodies->indices[i] = (uint16_t)s.u32[i];
	active_bodies->count = na;
}

// nudge.h:136, nudge.cpp:4021-4108
ContactImpulseData* read_cached_impulses(ContactCache contact_cache, ContactData contacts, Arena* memory) {
	Session& s = session_by_key(contacts.data, "read_cached_impulses");
	std::lock_guard<std::mutex> guard(s.lock);
	s.bind();
	if (contacts.count < s.host_contacts) die("read_cached_impulses", "contacts were removed since collide()");
	if (contacts.count > contacts.capacity) die("read_cached_impulses", "contacts.count exceeds contacts.capacity");
	if (contacts.count > s.host_contacts) {
		// custom contacts appended behind the list collide() returned (example/main.cpp:287): up they go, widened, and into tag order on the device
		const uint32_t old = s.host_contacts, extra = contacts.count - old;
		if (!s.perm.empty()) die("read_cached_impulses", "contacts appended twice in one step");
		hip_check(hipMemcpy(s.c_data.as<nh_Contact>() + old, contacts.data + old, sizeof(Contact) * (size_t)extra, hipMemcpyHostToDevice), "hipMemcpy H2D");
		s.u32.resize(2 * (size_t)extra); s.u64.resize(extra); s.feat.resize(extra);
		for (uint32_t i = 0; i < extra; ++i) {
			s.u32[2 * i] = contacts.bodies[old + i].a; s.u32[2 * i + 1] = contacts.bodies[old + i].b;
			widen_tag(contacts.tags[old + i], s.u64[i], s.feat[i]);
		}
		hip_check(hipMemcpy(s.c_bodies.as<nh_BodyPair>() + old, s.u32.data(), 8 * (size_t)extra, hipMemcpyHostToDevice), "hipMemcpy H2D");
		hip_check(hipMemcpy(s.c_tags.as<uint64_t>() + old, s.u64.data(), 8 * (size_t)extra, hipMemcpyHostToDevice), "hipMemcpy H2D");
		hip_check(hipMemcpy(s.c_features.as<uint32_t>() + old, s.feat.data(), 4 * (size_t)extra, hipMemcpyHostToDevice), "hipMemcpy H2D");
		s.positions.reserve(4 * (size_t)contacts.capacity + 16);
		nh_BodyData db = { s.body_xf.as<nh_Transform>(), s.body_props.as<nh_BodyProperties>(), s.body_mom.as<nh_BodyMomentum>(), s.body_idle.as<uint8_t>(), s.body_count };
		check(nh_append_contacts(s.ctx, &s.contacts, &db, extra, s.positions.as<uint32_t>(), s.temporary), "append_contacts");
		s.perm.resize(contacts.count);
		download(s.perm.data(), s.positions.p, 4 * (size_t)contacts.count);
		s.host_contacts = contacts.count;
	}
	upload_contact_data(s, contacts.data, contacts.count);
	const uint32_t m = contact_cache.count, cap = contact_cache.capacity;
	s.u64.resize((size_t)m + 1); s.feat.resize((size_t)m + 1);
	for (uint32_t i = 0; i < m; ++i) widen_tag(contact_cache.tags[i], s.u64[i], s.feat[i]);
	s.k_tags.reserve(8 * (size_t)cap + 16); s.k_features.reserve(4 * (size_t)cap + 16); s.k_data.reserve(16 * (size_t)cap + 16);
	upload(s.k_tags, s.u64.data(), 8 * (size_t)m);
	upload(s.k_features, s.feat.data(), 4 * (size_t)m);
	upload(s.k_data, contact_cache.data, sizeof(CachedContactImpulse) * (size_t)m);
	check(nh_set_cache_count(s.ctx, m), "set_cache_count");
	nh_ContactCache dk = { s.k_tags.as<uint64_t>(), s.k_features.as<uint32_t>(), s.k_data.as<nh_CachedContactImpulse>(), cap, m };
	nh_ContactImpulseData* dev = nullptr;
	check(nh_read_cached_impulses(s.ctx, &dk, &s.contacts, &s.temporary, &dev), "read_cached_impulses");
	ImpulseHandle* h = static_cast<ImpulseHandle*>(arena_take(memory, sizeof(ImpulseHandle), 16));
	h->magic = IMPULSE_MAGIC; h->dev = dev; h->world = &s;
	return reinterpret_cast<ContactImpulseData*>(h);
}

// nudge.h:138, nudge.cpp:4110-4158
void write_cached_impulses(ContactCache* contact_cache, ContactData contacts, ContactImpulseData* contact_impulses) {
	ImpulseHandle* h = reinterpret_cast<ImpulseHandle*>(contact_impulses);
	if (!h || h->magic != IMPULSE_MAGIC) die("write_cached_impulses", "not a ContactImpulseData of this backend");
	Session& s = *h->world;
	std::lock_guard<std::mutex> guard(s.lock);
	s.bind();
	(void)contacts;
	const uint32_t cap = contact_cache->capacity;
	s.k_tags.reserve(8 * (size_t)cap + 16); s.k_features.reserve(4 * (size_t)cap + 16); s.k_data.reserve(16 * (size_t)cap + 16);
	nh_ContactCache dk = { s.k_tags.as<uint64_t>(), s.k_features.as<uint32_t>(), s.k_data.as<nh_CachedContactImpulse>(), cap, 0 };
	check(nh_write_cached_impulses(s.ctx, &dk, &s.contacts, h->dev), "write_cached_impulses");
	const uint32_t m = dk.count;
	s.u64.resize((size_t)m + 1); s.feat.resize((size_t)m + 1);
	download(s.u64.data(), s.k_tags.p, 8 * (size_t)m);
	download(s.feat.data(), s.k_features.p, 4 * (size_t)m);
	for (uint32_t i = 0; i < m; ++i) contact_cache->tags[i] = narrow_tag(s.u64[i], s.feat[i]);
	download(contact_cache->data, s.k_data.p, sizeof(CachedContactImpulse) * (size_t)m);
	contact_cache->count = m;
}

// nudge.h:140, nudge.cpp:4170-4638
ContactConstraintData* setup_contact_constraints(ActiveBodies active_bodies, ContactData contacts, BodyData bodies, ContactImpulseData* contact_impulses, Arena* memory) {
	ImpulseHandle* h = reinterpret_cast<ImpulseHandle*>(contact_impulses);
	if (!h || h->magic != IMPULSE_MAGIC) die("setup_contact_constraints", "not a ContactImpulseData of this backend");
	Session& s = *h->world;
	std::lock_guard<std::mutex> guard(s.lock);
	s.bind();
	if (contacts.count != s.host_contacts) die("setup_contact_constraints", "contacts were added or removed since read_cached_impulses()");
	(void)active_bodies;
	upload_contact_data(s, contacts.data, contacts.count);
	nh_BodyData db = device_bodies(s, bodies, true, true, true, false);      // the gravity loop has touched momentum
	nh_ContactConstraintData* dev = nullptr;
	check(nh_setup_contact_constraints(s.ctx, &s.active_bodies, &s.contacts, &db, h->dev, &s.temporary, &dev), "setup_contact_constraints");
	// the C ABI defers row build + warm start to the first apply (nudge_hip.h note 7); the reference has applied the warm start when setup
	// returns (nudge.cpp:4563-4632) and host code may read momentum right away: complete the pending work before the download
	check(nh_synchronize(s.ctx), "synchronize");
	download(bodies.momentum, s.body_mom.p, sizeof(BodyMomentum) * (size_t)bodies.count);     // warm start + unused0 (4198)
	ConstraintHandle* c = static_cast<ConstraintHandle*>(arena_take(memory, sizeof(ConstraintHandle), 16));
	c->magic = CONSTRAINT_MAGIC; c->dev = dev; c->world = &s;
	return reinterpret_cast<ContactConstraintData*>(c);
}

// nudge.h:142, nudge.cpp:4640-4855
void apply_impulses(ContactConstraintData* data, BodyData bodies) {
	ConstraintHandle* c = reinterpret_cast<ConstraintHandle*>(data);
	if (!c || c->magic != CONSTRAINT_MAGIC) die("apply_impulses", "not a ContactConstraintData of this backend");
	Session& s = *c->world;
	std::lock_guard<std::mutex> guard(s.lock);
	s.bind();
	nh_BodyData db = device_bodies(s, bodies, false, false, true, false);    // custom constraint impulses may have touched momentum
	check(nh_apply_impulses(s.ctx, c->dev, &db, 1), "apply_impulses");
	download(bodies.momentum, s.body_mom.p, sizeof(BodyMomentum) * (size_t)bodies.count);
}

// nudge.h:144, nudge.cpp:4857-4884
void update_cached_impulses(ContactConstraintData* data, ContactImpulseData* contact_impulses) {
	ConstraintHandle* c = reinterpret_cast<ConstraintHandle*>(data);
	ImpulseHandle* h = reinterpret_cast<ImpulseHandle*>(contact_impulses);
	if (!c || c->magic != CONSTRAINT_MAGIC || !h || h->magic != IMPULSE_MAGIC || c->world != h->world) die("update_cached_impulses", "foreign opaque data");
	Session& s = *c->world;
	std::lock_guard<std::mutex> guard(s.lock);
	s.bind();
	check(nh_update_cached_impulses(s.ctx, c->dev, h->dev), "update_cached_impulses");
}

// nudge.h:146, nudge.cpp:4886-4926
void advance(ActiveBodies active_bodies, BodyData bodies, float time_step) {
	Session& s = session_by_key(bodies.transforms, "advance");
	std::lock_guard<std::mutex> guard(s.lock);
	s.bind();
	nh_BodyData db = device_bodies(s, bodies, true, false, true, true);
	// the active list is the one collide() produced (still on the device); the host copy must not have been edited
	if (active_bodies.count != s.active_bodies.count) die("advance", "active_bodies were edited since collide()");
	nh_ActiveBodies da = s.active_bodies;
	check(nh_advance(s.ctx, &da, &db, time_step), "advance");
	download(bodies.transforms, s.body_xf.p, sizeof(Transform) * (size_t)bodies.count);
	download(bodies.idle_counters, s.body_idle.p, bodies.count);
}

}  // namespace nudge

// Optional, for adopters that create and destroy worlds: nudge_hip_release_world(bodies.transforms) (or contacts.data) frees everything the GPU backend holds for
// the world these arrays belong to.  Declared in INTEGRATION.md; the reference API itself has no such call.
extern "C" void nudge_hip_release_world(const void* body_transforms_or_contact_data) { release_world(body_transforms_or_contact_data); }
