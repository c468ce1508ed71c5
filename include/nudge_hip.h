/*
 * nudge_hip.h -- C ABI of the MI355X-native stepping engine (libnudge_hip.so).
 *
 * This is the drop-in boundary for the hot path of rasmusbarr/nudge:
 *     collide -> [gravity] -> read_cached_impulses -> setup_contact_constraints
 *             -> apply_impulses x N -> update_cached_impulses -> write_cached_impulses -> advance
 * Each entry point below replaces the reference function cited next to it (reference nudge.h:134-146,
 * implemented in reference nudge.cpp:3000-4926) and keeps its argument meaning.  Differences, all forced
 * by the device boundary or by scale, are:
 *
 *   1. Every array pointer inside the structs is DEVICE memory (hipMalloc'ed by the caller, or carved out
 *      of a torch tensor): the caller still owns every array; per-step scratch and the device side of the
 *      two opaque result objects come out of the caller's `nh_Arena` (device memory) with
 *      the reference's bump semantics (nudge.h:29-32, nudge.cpp:990-1055): `nh_collide` takes the arena
 *      BY VALUE (scratch is dead on return), the two `Arena*` functions ADVANCE it.
 *      What the library DOES allocate itself (hipMalloc, freed by nh_destroy; all of it state that must survive from one
 *      step to the next, which the by-value arena cannot hold): the counter block (~5 KB device + a pinned host mirror);
 *      per body 4 B of warm-start hint and 16 B of contact degree / adjacency bookkeeping (grown to the largest
 *      `bodies->count` seen); the splitters and bucket counters of the tag sort (16 B per 1024 broadphase pairs of
 *      capacity); the broadphase's kept pair list (16 B per pair of pair capacity: twice the capacity in 8-byte pairs) and one
 *      inflated box per collider (32 B) -- nh_collide re-tests the kept pairs instead of searching the grid while no collider
 *      has left its box, and re-inserts the few that have (new box, new pairs) from the grid of its last search, which it keeps as well:
 *      per collider 32 B of cell-sorted boxes + 9 B of book-keeping, 4 B per grid cell (4 cells per collider, 2^16 .. 2^24); the contact records' keys by position and the tag order of the last sort (20 B per pair of capacity) --
 *      the sort is skipped while no record changes its key; a side stream and two events; two small host-side rings of opaque-object descriptors.
 *   2. Indices are widened: body indices and collider tags are 32-bit (the reference packs them in 16 bits:
 *      nudge.h:68-71, 86, 93, 126 and asserts <= 8192 colliders at nudge.cpp:3010).  A contact's identity
 *      is therefore split in two words: `tags[i]`  = a_tag | (uint64_t)b_tag << 32   (reference: high 32
 *      bits of its u64 tag, a16 | b16 << 16) and `features[i]` = the reference's low 32 feature bits.
 *      Ordering everywhere is the reference's: by b_tag, then a_tag, then feature.
 *   3. Functions return an error code (NH_OK = 0) instead of assert()-aborting (nudge.cpp:1000, 3010, 4118).
 *   4. Counts are produced on the device.  With NH_FLAG_SYNC_COUNTS (default) the host-visible `count`
 *      fields are filled in before the call returns, as in the reference; without it the calls only enqueue
 *      work on the context's stream and `nh_read_counts` fetches the numbers when the host wants them.
 *   5. `contacts->data/bodies/tags/features` are returned in TAG ORDER (the order the reference's solver
 *      consumes them in, nudge.cpp:4027-4044), not in narrowphase emission order: same set, canonical order.
 *   6. The caller-side gravity/damping loop of the sample app (example/main.cpp:290-305) runs over
 *      host-invisible memory here, so it is offered as `nh_apply_gravity_damping`.
 *   7. Deferred execution.  `nh_read_cached_impulses` and `nh_setup_contact_constraints` may postpone part of their
 *      work (the per-contact cache lookup; row build + warm start of bodies that only touch the static world) and
 *      run it fused with the first `nh_apply_impulses`, so that a constraint row lives only in registers; the one host
 *      round trip of a step (device counters) is also taken there, behind the solver kernel it has just launched.  Every
 *      entry point of this library that reads or writes momentum, impulses or counters first completes pending
 *      work, and so does `nh_synchronize`: through the API the reference's call-by-call semantics are kept.  Only
 *      foreign kernels enqueued on the stream BETWEEN setup and the first apply would see momentum without the
 *      warm start; call `nh_synchronize` (or any entry point) first if you need that.
 *
 *   8. The solver consumes the contact list as nh_collide laid it out (tag order, per-body degrees, adjacency seeds).
 *      Contacts appended by the caller after nh_collide (the reference allows that: "Custom contacts can be added here",
 *      example/main.cpp:287) are announced with `nh_append_contacts`, which merges them into that order and counts the per-body
 *      bookkeeping again; fields of existing contacts (friction, penetration ...) may simply be edited in place.
 *      One nh_setup_contact_constraints per nh_collide (a second one returns NH_ERR_STALE_SETUP).
 *
 *   9. Still steps (NH_FLAG_FUSED_STEP without NH_FLAG_SYNC_COUNTS / NH_FLAG_EXACT_ORDER).  A world at rest on static geometry repeats itself: the same pairs at
 *      the same places of the kept pair list, the same tag order, nearly the same contacts.  When a step ended in that state the library launches the NEXT one
 *      speculatively as a "still" step: transforms + AABBs, the narrowphase straight from the kept pair list into the pairs' own contact slots (library-owned),
 *      the fused solver reading those slots and a slot-indexed copy of the warm-start cache -- three kernels that CHECK everything they rely on (nobody outside
 *      its inflated box or asleep, every collider pair with the key it had, at most four contacts per body, all of them against the inert static world) instead
 *      of the ~25 launches that would find it out again.  A pair that gains or loses a contact is listed and the places of the contacts behind it shift (the
 *      solver order of nudge.cpp:4206-4339 is a function of a contact's dense tag-order index, which stays exact).  A failed check is reported with the step's
 *      one host round trip before anything irreversible has been written, and the step is run again in full from nh_collide's arguments: results are
 *      bit-identical to a library that never speculates (option "no_still"), which the tests check.
 *      Since round 5 two of those checks no longer fail the step (LOCAL speculation; option "no_local_still" restores them): a small collider outside its inflated box is
 *      given a new box and re-inserted into the kept pair list by the step itself (its pair with the ground it left keeps its record: taking off, flying and coming down
 *      again are still steps), and a body asleep in a set of its own (no overlapping AABB of another dynamic body, no user connections in the world) is left alone -- its
 *      pairs are sleeping pairs, its cached impulses are kept aside by the reference's rule (nudge.cpp:3669-3703, 4064-4101).  What still ends in a full step: a first
 *      contact with a collider the body was not resting on, two dynamic bodies touching, sets of several bodies with a sleeper among them, a rebuild of the kept list.
 *      Inside ONE nh_step call -- where the caller cannot touch anything between two sub-steps -- a still step in which nobody moves and nobody can be asleep also does
 *      the next sub-step's first kernel ("xform ahead"; option "no_xform_ahead"): the solver lane that has just advanced a body writes its collider's world transform
 *      and AABB, tests it against the inflated box and gathers the scene bounds, so that the next sub-step starts at the narrowphase; what the lane finds wanting fails
 *      THAT sub-step, which is run again in full.  Only for worlds whose dynamic bodies carry one collider each (checked on the device once per call).
 *      nh_Counts.ahead_steps counts the sub-steps that started that way.  (Round 6: the lane also evaluates its body's own collider pair for the next sub-step -- "pair ahead",
 *      nh_Counts.pair_steps: such a sub-step starts at the solver -- and while the set of sleeping bodies stands still a world WITH sleepers is stepped the same way, its
 *      sleepers part of the static world for those sub-steps ("sleepers ahead"; options "no_pair_ahead", "no_sleeper_ahead").)
 *      The caller's dense contact list (contacts->data / bodies / tags / features, contacts->sleeping_pairs), active list and contact cache are VIEWS under this regime: they are
 *      brought up to date by nh_export_views (what: NH_VIEW_CONTACTS, NH_VIEW_CACHE, NH_VIEW_ACTIVE), by nh_append_contacts, and by any step that does not qualify -- not by
 *      every step, and not by the cheap observers: nh_read_counts returns the counters and nh_synchronize waits for the stream, neither exports anything (the
 *      sample reads active_bodies.count every sub-step, example/main.cpp:293: that must not cost two passes over the contacts).  Call nh_export_views and then
 *      nh_synchronize before reading the arrays from the device yourself.  (BREAKING for hosts written against rounds 1-4, where nh_synchronize / nh_read_counts
 *      exported the views: nh_set_option(ctx, "sync_exports_views", 1) restores that contract at its old price; INTEGRATION.md.)  Any entry point outside the sample's
 *      call order between nh_collide and nh_apply_impulses turns a still step into a full one first, so everything it observes is what a full step
 *      produces.  nh_Counts.still_steps / still_replays count them.
 *
 * Threading: one context = one HIP stream = one world at a time; no global mutable state.
 */
#ifndef NUDGE_HIP_H
#define NUDGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------------- */
enum {
	NH_OK = 0,
	NH_ERR_INVALID = 1,          /* bad argument */
	NH_ERR_NO_DEVICE = 2,        /* no HIP device / runtime error */
	NH_ERR_ARENA = 3,            /* arena exhausted            (reference: assert nudge.cpp:1000-1039) */
	NH_ERR_CONTACT_CAPACITY = 4, /* contacts->capacity exceeded (reference: unchecked overflow)         */
	NH_ERR_CACHE_CAPACITY = 5,   /* contact_cache->capacity     (reference: assert nudge.cpp:4118)      */
	NH_ERR_ACTIVE_CAPACITY = 6,  /* active_bodies->capacity     (reference: unchecked overflow)         */
	NH_ERR_PAIR_CAPACITY = 7,    /* broadphase pair buffer: nh_set_pair_capacity (default contacts->capacity / 2 + 1024;
	                                AABB-overlap pairs can outnumber contacts in dense scenes: size it from the colliders) */
	NH_ERR_HIP = 8,              /* a HIP call failed; see nh_last_hip_error                            */
	NH_ERR_SCHEDULER_CAPACITY = 9, /* NH_FLAG_EXACT_ORDER only: the replay of the reference's greedy batch scheduler ran out of open
	                                batches (more than ~130,000 contacts on ONE dynamic body; the reference sizes this by
	                                contacts.count, nudge.cpp:4222-4223)                                  */
	NH_ERR_STALE_HINT = 11,      /* idle_counters were changed behind the library's back (see nh_bodies_changed): a body was asleep in a step
	                                for which the host had ruled that out                                 */
	NH_ERR_STALE_SETUP = 10      /* nh_setup_contact_constraints called twice for one nh_collide, or contacts->count changed in
	                                between: the solver's inputs are laid out by nh_collide (note 8)         */
};

/* ---- context flags ----------------------------------------------------------------------------- */
enum {
	NH_FLAG_SYNC_COUNTS = 1u,    /* fill host-side count fields before returning (reference semantics)   */
	NH_FLAG_SINGLE_APPLY = 4u,   /* the caller promises ONE nh_apply_impulses(..., iterations) per nh_setup_contact_constraints (the whole
	                                iteration loop of example/main.cpp:313-318 in one call): the per-contact solver states that only a
	                                further nh_apply_impulses call would read are then not written (16 B per contact per step); a
	                                second call for the same setup returns NH_ERR_INVALID.  Entry points that complete a deferred setup
	                                before the first apply (nh_read_counts, nh_synchronize ...) keep the states: the apply after them works */
	NH_FLAG_FUSED_STEP = 8u,     /* implies NH_FLAG_SINGLE_APPLY.  The caller promises the call order of the sample's step (example/main.cpp:286-330):
	                                nh_collide, nh_apply_gravity_damping(dt), nh_read_cached_impulses, nh_setup_contact_constraints, one
	                                nh_apply_impulses, nh_update_cached_impulses, nh_write_cached_impulses, nh_advance(dt), and does not touch body
	                                state in between.  Gravity / damping and the advance of the bodies whose only contacts are with the inert
	                                static world then happen inside their solver kernel (same arithmetic, two passes over the body state less);
	                                every other body goes through the ordinary kernels.  Any other entry point in between settles the pending
	                                gravity first; nh_advance with a different time step than gravity was given returns NH_ERR_INVALID -- a broken
	                                promise, found after the solver has moved its bodies by gravity's step: do not continue that world       */
	NH_FLAG_EXACT_ORDER = 2u     /* solver visits contacts in the reference's greedy batch order, replayed
	                                exactly on the device (nudge.cpp:4206-4339); default is the closed-form
	                                round-robin order, identical whenever the greedy scheduler meets no
	                                lane conflict (drop scenes)                                            */
};

/* ---- POD records: layouts identical to reference nudge.h unless marked (wide) -------------------- */
typedef struct nh_Arena { void* data; uintptr_t size; } nh_Arena;                                   /* nudge.h:29-32 */
typedef struct nh_Transform { float position[3]; uint32_t body; float rotation[4]; } nh_Transform;   /* nudge.h:34-38 */
typedef struct nh_BodyProperties { float inertia_inverse[3]; float mass_inverse; } nh_BodyProperties;/* nudge.h:40-43 */
typedef struct nh_BodyMomentum { float velocity[3]; float unused0; float angular_velocity[3]; float unused1; } nh_BodyMomentum; /* nudge.h:45-50 */
typedef struct nh_SphereCollider { float radius; } nh_SphereCollider;                                /* nudge.h:52-54 */
typedef struct nh_BoxCollider { float size[3]; float unused; } nh_BoxCollider;                       /* nudge.h:56-59 */
typedef struct nh_Contact { float position[3]; float penetration; float normal[3]; float friction; } nh_Contact; /* nudge.h:61-66 */
typedef struct nh_BodyPair { uint32_t a; uint32_t b; } nh_BodyPair;                                  /* nudge.h:68-71 (wide) */
typedef struct nh_CachedContactImpulse { float impulse[3]; float unused; } nh_CachedContactImpulse;  /* nudge.h:113-116 */
/* (`unused`: zero, as in the reference, under NH_FLAG_EXACT_ORDER; in the default solver order the library keeps the contact's
   colour of the last sweep there -- an integer bit pattern, never read as a number -- to warm-start the next step's colouring) */

typedef struct nh_ContactData {                                                                      /* nudge.h:73-82 (wide) */
	nh_Contact* data;
	nh_BodyPair* bodies;
	uint64_t* tags;            /* a_tag | b_tag << 32 */
	uint32_t* features;        /* feature word (reference: low 32 bits of the tag) */
	uint32_t capacity;
	uint32_t count;
	uint64_t* sleeping_pairs;  /* tag | tag << 32, ascending: pairs of sleeping sets dropped before the narrowphase carry the larger tag low
	                              (nudge.cpp:3697), pairs of sleeping contact islands the order of their contact tags (nudge.cpp:3979) */
	uint32_t sleeping_count;
} nh_ContactData;

typedef struct nh_ColliderData {                                                                     /* nudge.h:84-98 (wide tags) */
	struct { uint32_t* tags; nh_BoxCollider* data; nh_Transform* transforms; uint32_t count; } boxes;
	struct { uint32_t* tags; nh_SphereCollider* data; nh_Transform* transforms; uint32_t count; } spheres;
} nh_ColliderData;

typedef struct nh_BodyData {                                                                         /* nudge.h:100-106 */
	nh_Transform* transforms;
	nh_BodyProperties* properties;
	nh_BodyMomentum* momentum;
	uint8_t* idle_counters;
	uint32_t count;
} nh_BodyData;

typedef struct nh_BodyConnections { nh_BodyPair* data; uint32_t count; } nh_BodyConnections;         /* nudge.h:108-111 */

typedef struct nh_ContactCache {                                                                     /* nudge.h:118-123 (wide) */
	uint64_t* tags;
	uint32_t* features;
	nh_CachedContactImpulse* data;
	uint32_t capacity;
	uint32_t count;
} nh_ContactCache;

typedef struct nh_ActiveBodies { uint32_t* indices; uint32_t capacity; uint32_t count; } nh_ActiveBodies; /* nudge.h:125-129 (wide) */

typedef struct nh_ContactImpulseData nh_ContactImpulseData;         /* nudge.h:131, opaque */
typedef struct nh_ContactConstraintData nh_ContactConstraintData;   /* nudge.h:132, opaque */
typedef struct nh_context nh_context;

/* Device-side counters of the most recent step, mirrored to the host by nh_read_counts.  (Fields are only ever added at the END; a caller built against an
   older header must be recompiled: nh_read_counts fills the whole struct.) */
typedef struct nh_Counts {
	uint32_t colliders;         /* C */
	uint32_t pairs;             /* P: broadphase pairs surviving the same-body / sleeping filters */
	uint32_t contacts;          /* K */
	uint32_t sleeping_pairs;
	uint32_t active_bodies;
	uint32_t cache;             /* entries in the contact cache after the last write */
	uint32_t culled;            /* cached impulses kept aside for sleeping pairs */
	uint32_t large_colliders;   /* colliders handled by the brute-force "large x all" broadphase pass */
	uint32_t general_contacts;  /* contacts solved by the level-scheduled path (not the one-body fast path) */
	uint32_t levels;            /* dependency levels of that path */
	uint32_t error;             /* NH_* raised on the device (capacity overflows) */
	uint32_t has_other_bodies;  /* non-zero: some body is outside the one-body fast path's common class (several pairs, dynamic partners, no contacts) */
	uint32_t unleveled;         /* internal: progress of the level relaxation */
	uint32_t raw_pairs;         /* broadphase pairs before any filter (diagnostic) */
	uint32_t broadphase_rebuilds; /* nh_collide calls since nh_create that regrouped the colliders and searched the grid; the others re-used the kept pair list */
	uint32_t sort_reuses;       /* nh_collide calls since nh_create that skipped the tag sort of the contact records: every record sat where it sat the step
	                               before, with the same key */
	uint32_t broadphase_inserts; /* colliders that left their inflated box and were re-inserted into the kept pair list (new box, new pairs) without a rebuild,
	                               since nh_create */
	uint32_t still_steps;       /* steps since nh_create that went through as STILL steps (note 9): launched as a world whose contact layout is last step's, and confirmed */
	uint32_t still_replays;     /* ... launched as one, found otherwise (by the device or because the caller left the sample's call order) and run again in full */
	uint32_t still_diff[4];     /* what still steps have found changed, summed since nh_create: collider pairs with another key / another contact count / other feature words;
	                               colliders outside their inflated box or bodies asleep (diagnostic) */
	uint32_t blk_blocks, blk_bodies, blk_ghosts;   /* the blocked solver's tables of the last step that built them (large general sets): blocks of the grid, general bodies binned,
	                               ghost copies borrowed by blocks (a sweep loads and stores blk_bodies + blk_ghosts momentum records); diagnostic / measurement */
	uint32_t asleep_steps;      /* steps since nh_create that nh_step found to be steps of a world in which every body is asleep and nothing has changed: done without a launch */
	uint32_t ahead_steps;       /* still steps (launched, since nh_create) that started at the narrowphase: the solver of the sub-step before them, inside the same nh_step call, had
	                               already written their colliders' world transforms and boxes (note 9, "xform ahead") */
	uint32_t fused_steps;       /* ... and that went through as ONE out-of-place launch with a look-back (option "fused_still", off by default: measured slower) */
	uint32_t pair_steps;        /* still steps that started at the SOLVER: the solver lanes of the sub-step before them had evaluated every body's own collider pair for them as well
	                               (note 9, "pair ahead") -- a step of ONE launch behind a one-workgroup prologue */
	uint32_t pair_diag[4];      /* why such steps were refused, summed since nh_create: which collider plays "a" could not be told without the next scene frame / the record was not the
	                               lane's to evaluate (another key, a fifth contact, a partner that is not the static world) / the next frame's cells came out too large / some kept pair
	                               belongs to no body (diagnostic) */
} nh_Counts;

/* ---- lifecycle ---------------------------------------------------------------------------------- */
/* `stream` is a hipStream_t (or NULL for the default stream) on `device`; the context never creates one. */
int nh_create(nh_context** out, int device, void* stream, uint32_t flags);
void nh_destroy(nh_context* ctx);
int nh_set_flags(nh_context* ctx, uint32_t flags);
int nh_synchronize(nh_context* ctx);                         /* completes deferred work and waits for the stream; exports no view (note 9) */
int nh_read_counts(nh_context* ctx, nh_Counts* out);        /* synchronises the stream; exports no view (note 9) */
/* The caller's views of what still steps keep by slot (note 9), brought up to date on the context's stream: NH_VIEW_CONTACTS = contacts->data / bodies / tags /
   features in tag order, NH_VIEW_CACHE = the nh_ContactCache arrays (tags, features, data; nh_Counts.cache).  A no-op (two host-side flag tests) when the arrays
   are current -- after a full step they always are.  Replaces the reference's implicit guarantee that both arrays are current when collide() /
   write_cached_impulses() return (nudge.cpp:4009, 4158) for callers that read them from the device themselves. */
enum { NH_VIEW_CONTACTS = 1u /* + contacts->sleeping_pairs */, NH_VIEW_CACHE = 2u, NH_VIEW_ACTIVE = 4u /* active_bodies->indices */, NH_VIEW_ALL = 7u };
int nh_export_views(nh_context* ctx, uint32_t what);
int nh_set_cache_count(nh_context* ctx, uint32_t count);    /* restore a checkpointed ContactCache */
/* Tell the library that the caller has written `idle_counters` itself (initial upload excepted: a fresh context assumes nothing).  The
   library predicts on the host when a body can first be asleep -- counters rise by at most one per nh_advance -- and launches none of the
   island / sleeping kernels before that; counters written from outside invalidate the prediction until the next nh_collide has looked at
   them.  A forgotten call is detected on the device and reported as NH_ERR_STALE_HINT, never silently wrong. */
int nh_bodies_changed(nh_context* ctx);
int nh_set_tag_bits(nh_context* ctx, uint32_t bits);        /* collider tags are < 2^bits (default 32): fewer sort passes */
/* Capacity of the broadphase pair buffer (AABB-overlap pairs after the same-body filter), carved from the arena at ~210 B per
   pair by nh_collide.  0 (default) = contacts->capacity / 2 + 1024.  The reference has no such limit (its pair list lives in the
   arena too, nudge.cpp:3473): dense scenes hold more overlapping pairs than contacts (a settled ball pit: 1.7 x), so size it from
   the collider count there.  Overflow is reported as NH_ERR_PAIR_CAPACITY. */
int nh_set_pair_capacity(nh_context* ctx, uint32_t pairs);
/* Diagnostic switches for A/B runs and tests -- none is needed in production, each selects an older, slower or more talkative path; the library never reads the
   environment.  Names (value 0 / 1 unless noted): "no_still", "no_kept_pairs", "no_incremental", "no_sort_reuse", "sort_classic", "sort_radix", "bucket_tile" (n),
   "bucket_target" (n), "colour_jp", "colour_check_seeds", "no_resident", "solver_waves" (1, 2, 4), "fp_sub" (0, 1, 2, 4, 16), "cooperative", "no_blocks", "blk_check",
   "blk_min" (n), "blk_target" (n), "blk_rows_global", "blk_global_colours", "blk_profile", "no_asleep", "no_blk_chain", "no_local_still", "no_xform_ahead",
   "no_pair_ahead", "no_sleeper_skip", "no_sleeper_ahead", "no_early_counts", "halo_overlap" (1 = on), "fused_still", "fused_fail_in" (n: test hook), "sync_exports_views".  Unknown name: NH_ERR_INVALID.  Call right after nh_create.
   (nudge_amd/engine.py maps environment variables NH_<NAME> onto these calls for its tests: a convenience of that host, not of the library.) */
int nh_set_option(nh_context* ctx, const char* name, int value);
const char* nh_error_string(int code);
int nh_last_hip_error(nh_context* ctx);

/* ---- the hot path: one entry point per reference function ------------------------------------------ */
/* collide (nudge.h:134, nudge.cpp:3000-4009) */
int nh_collide(nh_context* ctx, nh_ActiveBodies* active_bodies, nh_ContactData* contacts,
               const nh_BodyData* bodies, const nh_ColliderData* colliders,
               const nh_BodyConnections* body_connections, nh_Arena temporary);

/* Custom contacts (example/main.cpp:287: "Custom contacts can be added here").  The caller has written `extra` complete contacts -- data, bodies,
   tags (a_tag | b_tag << 32: any pair of tags that no collider pair produces), features -- behind the list nh_collide returned, at indices
   [count, count + extra) of the arrays of `contacts` (device memory; without NH_FLAG_SYNC_COUNTS `count` is what nh_read_counts reports).  Call this
   after nh_collide and before nh_read_cached_impulses: the contacts are merged into tag order (the order the reference's solver consumes,
   nudge.cpp:4027-4044, 4172) and the per-body bookkeeping is counted again; the contact count grows by `extra` (contacts->count too under
   NH_FLAG_SYNC_COUNTS).  `positions` (device memory, count + extra words, or NULL): where every contact of the list as the caller left it -- the
   old ones, then the appended ones -- now sits.  `temporary`: scratch (~60 bytes per contact of capacity).  At most 65,536 contacts per call.
   Like in the reference, contacts appended here take no part in the sleeping islands nh_collide has already formed. */
int nh_append_contacts(nh_context* ctx, nh_ContactData* contacts, const nh_BodyData* bodies, uint32_t extra, uint32_t* positions, nh_Arena temporary);

/* caller-side loop of the sample app (example/main.cpp:290-305): v -= g*dt; v,w *= 1 - dt*damping_rate */
int nh_apply_gravity_damping(nh_context* ctx, const nh_ActiveBodies* active_bodies, const nh_BodyData* bodies,
                             float time_step, const float gravity[3], float damping_rate);

/* read_cached_impulses (nudge.h:136, nudge.cpp:4021-4108) */
int nh_read_cached_impulses(nh_context* ctx, const nh_ContactCache* contact_cache, const nh_ContactData* contacts,
                            nh_Arena* memory, nh_ContactImpulseData** out);

/* write_cached_impulses (nudge.h:138, nudge.cpp:4110-4158) */
int nh_write_cached_impulses(nh_context* ctx, nh_ContactCache* contact_cache, const nh_ContactData* contacts,
                             nh_ContactImpulseData* contact_impulses);

/* setup_contact_constraints (nudge.h:140, nudge.cpp:4170-4638) */
int nh_setup_contact_constraints(nh_context* ctx, const nh_ActiveBodies* active_bodies, const nh_ContactData* contacts,
                                 const nh_BodyData* bodies, nh_ContactImpulseData* contact_impulses,
                                 nh_Arena* memory, nh_ContactConstraintData** out);

/* apply_impulses (nudge.h:142, nudge.cpp:4640-4855): `iterations` consecutive calls of the reference */
int nh_apply_impulses(nh_context* ctx, nh_ContactConstraintData* data, const nh_BodyData* bodies, uint32_t iterations);

/* update_cached_impulses (nudge.h:144, nudge.cpp:4857-4884) */
int nh_update_cached_impulses(nh_context* ctx, nh_ContactConstraintData* data, nh_ContactImpulseData* contact_impulses);

/* advance (nudge.h:146, nudge.cpp:4886-4926) */
int nh_advance(nh_context* ctx, const nh_ActiveBodies* active_bodies, const nh_BodyData* bodies, float time_step);

/* ---- multi-GPU helpers (nudge_amd/partition.py): the per-step halo record of a body ---------------------- */
/* record = transform.position (12 B) | transform.rotation (16 B) | momentum (32 B) | idle counter (1 B) + 3 B pad = 64 B.
   nh_halo_pack gathers the records of the bodies listed in `indices` (device memory) into `out` (device memory, count x 64 B);
   nh_halo_unpack writes `count` records into the consecutive body slots first_slot .. first_slot+count-1 (the `body` field of
   their transforms -- unused by the engine for bodies, nudge.h:36 -- is left alone).  Transport between the two is the host's
   business (RCCL send/recv). */
#define NH_HALO_RECORD_BYTES 64
int nh_halo_pack(nh_context* ctx, const nh_BodyData* bodies, const uint32_t* indices, uint32_t count, void* out);
int nh_halo_unpack(nh_context* ctx, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in);
/* nh_halo_update: nh_halo_unpack for records of the SAME bodies that already occupy those slots, as their owner stepped them since the last exchange (the
   per-step halo; nh_halo_unpack is for slots that change hands: refresh, migration).  The difference is the library's sleep prediction (see
   nh_bodies_changed): idle counters rise by at most one per step on the owner as they do here, so an update does not invalidate it. */
int nh_halo_update(nh_context* ctx, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in);

/* ---- the sample's sub-step loop as ONE entry point (example/main.cpp:274-328: simulate()) ---------------------------------------------------------------------
   nh_step(ctx, args, steps) = `steps` times  nh_collide, nh_apply_gravity_damping, nh_read_cached_impulses, nh_setup_contact_constraints, nh_apply_impulses(iterations),
   nh_update_cached_impulses, nh_write_cached_impulses, nh_advance  on the arrays of `args` (scratch and the two opaque objects come out of `arena`, reset every step like
   the sample does, example/main.cpp:282).  Same results as the eight calls made by the caller -- bit for bit -- with two savings: one crossing of the ABI instead of
   eight per step, and, because the library drives the call order itself, the host round trip of a still step (note 9) is taken one step LATE, so that neither the host nor
   the GPU waits for the other inside the loop.  Every step is confirmed when the call returns.  Custom contacts / user impulses between the calls need the eight calls.
   A world in which EVERY body is asleep is a fixed point of the step (no active body, no contact, every overlapping pair a sleeping pair, the cache kept aside and
   written back as it was: nudge.cpp:3669-3703, 4064-4101, 4896-4898).  Once two full steps in a row have shown that, nh_step checks once per call -- one kernel: every
   collider's world AABB and tag what they were, every body still asleep -- that nothing the caller owns has changed, and takes the steps of the call as done
   (nh_Counts.asleep_steps); anything else runs them in full.  Results are those of a library that never does this (option "no_asleep"), which the tests check. */
typedef struct nh_StepArgs {
	nh_ActiveBodies* active_bodies; nh_ContactData* contacts; const nh_BodyData* bodies; const nh_ColliderData* colliders; const nh_BodyConnections* body_connections;
	nh_ContactCache* contact_cache; nh_Arena arena;
	float time_step; float gravity[3]; float damping_rate; uint32_t iterations;
} nh_StepArgs;
int nh_step(nh_context* ctx, const nh_StepArgs* args, uint32_t steps);

/* ---- state streaming: the GL-free viewer hook (the sample redraws from bodies.transforms after simulate(), example/main.cpp:176-272, 330) -------------------------
   A viewer, a recorder or a network peer wants body transforms on the HOST every few steps without stopping the world.  nh_stream_state arms that: every `every`-th
   nh_advance (the end of a sub-step, whoever drives the eight calls) the transforms of the first `count` bodies are copied device-to-device into a library-owned staging
   buffer on the context's stream (tens of microseconds at a million bodies, ordered behind the step that produced them) and from there, on the library's side stream,
   into slot (frame mod slots) of the caller's ring -- the world's stream never waits for the host link.  A frame whose predecessor is still on its way to the host is
   dropped, not queued (nh_StreamInfo.dropped).  Ring layout: `slots` frames of count x sizeof(nh_Transform) bytes, back to back; give it pinned memory (hipHostMalloc)
   or the copies are staged by the runtime.  nh_stream_latest: the newest frame that has LANDED (never blocks): its slot, and the number of nh_advance calls (since
   nh_stream_state) it shows.  every = 0 disarms.  Reference counterpart: none needed -- its arrays are host memory. */
typedef struct nh_StreamInfo { uint32_t slot; uint32_t valid; uint64_t step; uint64_t frames, dropped; } nh_StreamInfo;
int nh_stream_state(nh_context* ctx, const nh_BodyData* bodies, uint32_t count, void* host_ring, uint32_t slots, uint32_t every);
int nh_stream_latest(nh_context* ctx, nh_StreamInfo* out);

/* ---- multi-GPU: one x-slab of a world per context ------------------------------------------------------------------------------------
   The world is cut into slabs along x, one rank (process, GPU, nh_context) per slab [lo, hi).  A rank's arrays hold
       body slot 0            the static world with the static colliders the rank keeps (front of the collider arrays),
       slots 1 .. n_owned     the dynamic bodies whose centre lies in the slab: authoritative here,
       the tail               GHOSTS: copies of neighbour-owned bodies that may touch something owned here (left neighbour's first),
   and one collider per dynamic body, in body-slot order behind the static ones.  An `nh_partition` keeps the per-body collider description
   (shape, local transform, tag, kind), the slab, and the lists of who is sent where; everything below is device work on the context's
   stream with device-side counts -- what the host gets back are the message lengths it needs to post the messages.  TRANSPORT IS THE CALLER'S:
   records are written to / read from caller-provided device buffers (ncclSend / ncclRecv, torch.distributed ...);
   nudge_amd/partition.py and examples/partition_rccl.cpp are two such transports over these entry points.
     every `epoch` steps (and at step 0), REFRESH:
        [optional: re-balance -- nh_partition_choose_cut / nh_partition_set_cut]
        nh_partition_pack_migrants  -> exchange -> nh_partition_unpack_migrants     (bodies whose centre left the slab change owner)
        nh_partition_pack_ghosts    -> exchange -> nh_partition_unpack_ghosts       (owned bodies that can reach across a cut within the epoch; collider arrays rebuilt)
     every other step:
        nh_partition_pack_step      -> exchange -> nh_partition_unpack_step         (64-byte records of the listed bodies, fixed lengths)
   Reference counterpart: none (the reference is single-threaded); the contact identity that makes this work is the collider TAG
   (nudge.h:86, 93; nudge.cpp:2074-2087), which travels with a body.  A dynamic body carries up to NH_PARTITION_MAX_COLLIDERS colliders (compound bodies); in the
   collider arrays the colliders of dynamic bodies stand in body-slot order, a body's own adjacent (boxes in the box array, spheres in the sphere array). */
#define NH_PARTITION_MAX_COLLIDERS 4     /* colliders a dynamic body may carry across ranks (compound bodies, nudge.cpp:3023-3025, 3058-3060): its boxes, then its spheres */
#define NH_PARTITION_FULL_BYTES 308      /* transform 32 | properties 16 | momentum 32 | 4 x (shape 16 | collider transform 32 | tag 4 | pad 4) | boxes 1, spheres 1, idle 1, pad 1 */
typedef struct nh_partition nh_partition;
typedef struct nh_PartitionConfig {
	uint32_t rank, ranks;
	double lo, hi;                                      /* the slab along x; -HUGE_VAL / HUGE_VAL at the two ends */
	uint32_t n_owned;                                   /* dynamic bodies in slots 1 .. n_owned at creation */
	uint32_t n_static_box, n_static_sph;                /* static colliders at the front of the collider arrays (never touched) */
	uint32_t body_capacity, box_capacity, sphere_capacity;   /* records the caller's arrays have room for */
	uint32_t epoch;                                     /* steps between two refreshes */
	double time_step, gravity, speed_floor;             /* how far a body can travel in an epoch: epoch dt (max(top speed, floor) + g epoch dt) */
	double max_reach;                                   /* largest distance from a body origin to the far side of its collider, over the WHOLE world */
	double cut_slack;                                   /* a cut never moves further than this from where it started (static colliders kept); HUGE_VAL = unlimited */
} nh_PartitionConfig;
typedef struct nh_PartitionInfo {
	uint32_t n_owned, n_bodies, n_boxes, n_spheres;     /* bodies = 1 + owned + ghosts; colliders incl. the static ones */
	uint32_t ghost_out[2], ghost_in[2];                 /* [0] left neighbour, [1] right neighbour: bodies listed for / received from it at the last refresh */
	double lo, hi;
	uint64_t migrated_out, migrated_in, refreshes, cut_moves;
	uint64_t quiet_refreshes;                           /* refreshes this rank found it did not need (nh_partition_refresh_is_quiet said 1) */
} nh_PartitionInfo;
/* `bodies` / `colliders`: the rank's arrays as they are at creation (1 + n_owned bodies; the static colliders, then the dynamic bodies' in slot order, Transform.body = slot). */
int nh_partition_create(nh_partition** out, nh_context* ctx, const nh_PartitionConfig* config, const nh_BodyData* bodies, const nh_ColliderData* colliders);
void nh_partition_destroy(nh_partition* p);
int nh_partition_info(nh_partition* p, nh_PartitionInfo* out);
/* Refresh, phase 1.  pack: full records of the owned bodies whose centre is left of lo -> out_left, right of (or on) hi -> out_right (device buffers of `capacity_records`
   records each; either may be NULL at an end of the row); counts[0..1] = records written (the call synchronises the stream once to tell).
   unpack: the leavers are dropped (owned bodies stay in slot order), the arrivals appended, left neighbour's first; ghosts are gone until phase 2. */
int nh_partition_pack_migrants(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right, uint32_t capacity_records, uint32_t counts[2]);
int nh_partition_unpack_migrants(nh_partition* p, const nh_BodyData* bodies, const void* in_left, uint32_t n_left, const void* in_right, uint32_t n_right);
/* Between the two phases -- or before phase 1, with its counts -- the ranks at a cut tell each other their TOP SPEED: a cut's ghost margin is reach + 2 x (how far anything
   on EITHER side travels in an epoch), so it must know the fastest body of the neighbour too (a fast body owned there reaches a slow one here that this rank would
   otherwise never list for it).  nh_partition_top_speed: the largest |velocity| among the bodies this rank owns now (call it BEFORE nh_partition_pack_migrants: a body that
   migrates in this refresh then counts on both sides); nh_partition_set_peer_speeds: what the left / right neighbour reported (NAN or <= 0: none).  Both are consumed by the
   next nh_partition_pack_ghosts; a host that never calls them gets margins from this rank's own top speed and `speed_floor` alone -- safe only when the floor bounds the world. */
int nh_partition_top_speed(nh_partition* p, const nh_BodyData* bodies, double* out);
/* A QUIET refresh (round 6): *quiet = 1 when a refresh would change nothing on this rank -- no owned body has crossed a cut, and the bodies within reach of a cut (at the
   speeds of nh_partition_top_speed / nh_partition_set_peer_speeds: call those first; nothing is consumed) are exactly the ones listed for the neighbours at the last
   refresh, in the same order.  When EVERY rank of the job says so the host may skip the refresh and exchange a per-step halo instead (nudge_amd/partition.py does, with an
   all-reduce of the answers): the still steps of a world at rest then run through the epoch boundary.  Never 1 before the first refresh.  One stream synchronisation. */
int nh_partition_refresh_is_quiet(nh_partition* p, const nh_BodyData* bodies, int* quiet);
int nh_partition_set_peer_speeds(nh_partition* p, double left, double right);
/* Refresh, phase 2.  pack: the owned bodies within reach of a cut for the coming epoch are listed (the lists are kept for the per-step records) and their full
   records written.  unpack: the received ghosts are installed behind the owned bodies, the collider arrays rebuilt (static ones, then one per dynamic body in
   slot order), `bodies->count` and the collider counts of the caller's structs set, and the context told that body records changed (nh_bodies_changed). */
int nh_partition_pack_ghosts(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right, uint32_t capacity_records, uint32_t counts[2]);
int nh_partition_unpack_ghosts(nh_partition* p, nh_BodyData* bodies, nh_ColliderData* colliders, const void* in_left, uint32_t n_left, const void* in_right, uint32_t n_right);
/* Every other step: NH_HALO_RECORD_BYTES per listed body (nh_halo_pack's record); lengths are those of the last refresh (nh_partition_info: ghost_out / ghost_in). */
int nh_partition_pack_step(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right);
int nh_partition_unpack_step(nh_partition* p, const nh_BodyData* bodies, const void* in_left, const void* in_right);
/* The per-step halo driven by the library: nh_partition_exchange_step = pack -> ncclGroupStart; ncclSend / ncclRecv with both neighbours; ncclGroupEnd -> unpack, all
   enqueued on the context's stream in ONE call (the host's share of a step shrinks to two calls: this one and nh_step).  The library does not link RCCL: the host hands
   it an ncclComm_t of the world's ranks and the addresses of ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv of the RCCL library its process has loaded
   (nh_partition_set_transport; left_peer / right_peer: communicator ranks of the neighbours, -1 = none).  nh_partition_transport_check sends a pattern to both peers and
   receives theirs in one group (a peer equal to the own rank: a loop-back on one GPU); with enqueue_only the host polls the stream itself -- a neighbour that never answers must not hang
   it -- and fetches the verdict with nh_partition_transport_result. */
/* (nh_partition_transport_check may re-size the message buffers: after an enqueue_only check call nh_partition_transport_result BEFORE the next exchange or check.
   An RCCL call that fails makes the exchange return NH_ERR_HIP with the group closed; nh_last_hip_error then holds the ncclResult_t, negated.) */
int nh_partition_set_transport(nh_partition* p, void* comm, void* group_start, void* group_end, void* send, void* recv, int left_peer, int right_peer);
int nh_partition_exchange_step(nh_partition* p, const nh_BodyData* bodies);
/* K sub-steps of a slab in ONE call (the loop example/main.cpp:274-328 runs for a partitioned world: halo exchange, then the eight calls, K times): nh_step with the per-step
   exchange enqueued by the library between two sub-steps over the transport of nh_partition_set_transport (needed as soon as there is a ghost on either side).  The
   chain of still steps (note 9: xform ahead, pair ahead) runs through such a call for the OWNED bodies; ghosts' colliders are transformed and their pairs evaluated when
   the halo has arrived.  `exchange_first`: the first sub-step is preceded by a per-step exchange too (0 right after a refresh, which has just installed the ghosts).
   `loopback_records` (rehearsal on one rank): that many owned bodies' records also travel through the transport to this same rank every sub-step.  K must not reach past
   the next refresh: migration and ghost lists are the host's (nh_partition_pack_migrants ...). */
int nh_partition_step(nh_partition* p, const nh_StepArgs* args, uint32_t steps, uint32_t exchange_first, uint32_t loopback_records);
int nh_partition_transport_check(nh_partition* p, uint32_t bytes, int enqueue_only);
int nh_partition_transport_result(nh_partition* p);
/* Re-balancing ("move cuts when |count_r - mean| > 5 %").  The two ranks at a cut tell each other how many bodies they own (transport: caller); when the
   counts differ by more than `tolerance` of their sum the HEAVIER side calls nh_partition_choose_cut(direction, neighbour's count) -- the x that hands half the
   difference, at most 5 % of its bodies, to the neighbour; it moves its own cut and returns it -- and sends the value over; the lighter side calls nh_partition_set_cut.
   The bodies in between migrate with the refresh that follows.  direction: -1 = the cut towards the left neighbour, +1 = towards the right. */
int nh_partition_choose_cut(nh_partition* p, const nh_BodyData* bodies, int direction, uint32_t neighbour_owned, double* cut);
int nh_partition_set_cut(nh_partition* p, int direction, double cut);
/* One owner per contact that crosses a cut (SURVEY 8(e) "determinism rule": the contact is owned by the rank owning the body with the larger collider tag, mirroring the
   a > b canonicalisation of nudge.cpp:2074-2087 / 2131-2132).  With first_ghost > 0 the bodies [first_ghost, count) of the context's world are ghosts and nh_collide makes
   contacts only for collider pairs this rank owns: a pair with the static world belongs to its dynamic body, a pair of two dynamic bodies to the body whose collider has the
   larger tag -- if that body is a ghost here, the pair yields no contact on this rank (the rank that owns the body solves it).  The host returns what the solver did to the
   ghosts (momentum after a sweep minus before) to their owners and refreshes the ghosts from the owners after every nh_apply_impulses(..., 1): nudge_amd/partition.py
   `Partition(per_iteration=True, single_owner=True)`.  0 switches the rule off.  Still steps (note 9) are not launched while it is on. */
int nh_set_first_ghost_body(nh_context* ctx, uint32_t first_ghost);
/* ... and the per-iteration exchange that goes with it, behind the C ABI (round 5; the torch form in nudge_amd/partition.py is kept as the independent check).  Records are
   the 32-byte nh_BodyMomentum; lengths are the ghost lists of the last refresh (nh_partition_info: a rank SENDS ghost_in[side] deltas and ghost_out[side] momentum records
   to the neighbour on that side and receives the other number), in the neighbour's list order.  One step of a rank (examples/partition_rccl.cpp --single-owner):
       nh_set_first_ghost_body(n_owned + 1); nh_collide; gravity; nh_read_cached_impulses; nh_partition_mark_ghosts; nh_setup_contact_constraints; EXCHANGE;
       iterations x { even ranks: nh_apply_impulses(.., 1); EXCHANGE; odd ranks: nh_apply_impulses(.., 1); EXCHANGE }; update; write; advance
       EXCHANGE = pack_deltas -> transport -> unpack_deltas (owners add what the neighbours' sweeps did to their bodies); pack_momentum -> transport -> unpack_momentum
                  (ghosts take their owners' momentum); mark_ghosts -- or nh_partition_exchange_iteration, which does all of it over the transport of nh_partition_set_transport.
   Every call first completes deferred work (note 7), so the warm start is in the momentum the records are taken from. */
int nh_partition_mark_ghosts(nh_partition* p, const nh_BodyData* bodies);
int nh_partition_pack_deltas(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right);
int nh_partition_unpack_deltas(nh_partition* p, const nh_BodyData* bodies, const void* in_left, const void* in_right);
int nh_partition_pack_momentum(nh_partition* p, const nh_BodyData* bodies, void* out_left, void* out_right);
int nh_partition_unpack_momentum(nh_partition* p, const nh_BodyData* bodies, const void* in_left, const void* in_right);
int nh_partition_exchange_iteration(nh_partition* p, const nh_BodyData* bodies);

/* ---- introspection for tests / measurement --------------------------------------------------------- */
/* Device pointer to the per-contact warm-start impulses of an nh_ContactImpulseData (K x 16 B, contact order). */
const nh_CachedContactImpulse* nh_contact_impulses_device(const nh_ContactImpulseData* d);
/* Per-kernel device time of the last step, measured with HIP events on the context's stream when timing is on. */
int nh_enable_timing(nh_context* ctx, int on);
/* Restrict timing to kernels whose name equals `name` (NULL or "" = all): two events per step instead of hundreds.  (With timing of ALL kernels on, nh_step looks at every
   still step's verdict inside the step -- the events are collected at that round trip; restricted to one kernel it keeps its late verdicts and collects at the end of the call.) */
int nh_set_timing_filter(nh_context* ctx, const char* name);
/* Writes up to `cap` (name, milliseconds, launches) triples accumulated since the last reset; returns the number. */
typedef struct nh_KernelTime { const char* name; double ms; uint32_t launches; uint32_t reserved; } nh_KernelTime;
int nh_kernel_times(nh_context* ctx, nh_KernelTime* out, int cap, int reset);

#ifdef __cplusplus
}
#endif

#endif /* NUDGE_HIP_H */
