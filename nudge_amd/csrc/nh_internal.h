// nh_internal.h -- shared by the .hip translation units of libnudge_hip.so (not installed).
#ifndef NH_INTERNAL_H
#define NH_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/nudge_hip.h"
#include "nh_math.h"

#define NH_WAVE 64
#define NH_BLOCK 256
#define NH_MAX_CELLS 0xFFFFFFu          // cell keys fit 24 bits (3 radix passes); key NH_LARGE_KEY marks a large collider
#define NH_LARGE_KEY 0xFFFFFFu
#define NH_SORT_GRID 512                // blocks per radix-sort pass (each owns a contiguous chunk)
#define NH_MAX_LEVELS 4096

// ---- spatially blocked solver (nh_blocks.h): the block grid of this step, on the device ------------------
#define NH_BLK_ID_BITS 15u
#define NH_BLK_MAX_BLOCKS (1u << NH_BLK_ID_BITS)
#define NH_BLK_MAX_COLOURS 32u
#define NH_BLK_RADIAL 0x8000u            // in a contact's local index pair: it has a radial row (set by k_rows_general)
struct nh_BlkGrid {
	uint32_t bmin[3], bmax[3];        // bounds of the general bodies' centres (order-preserving uint encoding, atomicMin / Max)
	uint32_t bodies;                  // general bodies
	float origin[3];
	float edge, inv_edge;             // block edge length
	uint32_t bits[3];                 // block id = cx | cy << bits[0] | cz << (bits[0] + bits[1])
	uint32_t nblk;                    // 1 << (bits[0] + bits[1] + bits[2]); 0 = no grid (no general body, non-finite bounds)
	uint32_t overflow;                // some block's bodies + ghosts do not fit one workgroup's LDS (or its rank field): smaller blocks needed
	uint32_t max_load;                // largest 1 + own + ghost count of a block
	uint32_t ghost_cursor;            // entries of the ghost list handed out
	uint32_t phase_count[8];          // blocks with work, by block colour (parity of the three block coordinates)
	uint32_t check_errors;            // NH_BLK_CHECK=1: structural errors found by k_blk_check
	uint32_t near_contacts, far_contacts;   // contacts between two blocks left to the residual launches (far; `near` -- solved inside a block -- is no longer counted)
	uint32_t max_colours, too_many_colours; // per-block colouring (k_blk_prepare_local): most colours a block used; some block needed more than its descriptor holds
	uint32_t max_rounds, max_listed;        // ... most Jones-Plassmann rounds a block took, most contacts a block had to colour anew (statistics)
};

// ---- device-resident step state (one per context) ---------------------------------------------------
struct nh_DevState {
	// counters (zeroed at the start of nh_collide, except `cache`)
	uint32_t colliders;
	uint32_t pairs;
	uint32_t raw_contacts;
	uint32_t records;
	uint32_t contacts;
	uint32_t sleeping;
	uint32_t active;
	uint32_t cache;
	uint32_t culled;
	uint32_t large;
	uint32_t general_contacts;
	uint32_t levels;
	uint32_t error;
	uint32_t has_other;           // some body is NOT of the class the fused one-body solver owns (k_adj_simple; NH_FLAG_FUSED_STEP: the rest kernels have work)
	uint32_t unleveled;           // general contacts still without a level (relaxation progress)
	uint32_t pad0;                // always 0 (used as a constant device-side count)
	uint32_t has_static8, has_staticN;    // any body in the rarer fast-path classes (plain stores, no counting)
	uint32_t any_sleeping;                // some contact record belongs to a sleeping island this step
	uint32_t unleveled_odd;               // progress counter of odd relaxation / colouring rounds
	uint32_t static_inert;                // body 0 has all-zero velocity, inverse mass and inverse inertia (nh_is_inert)
	uint32_t has_pending;                 // some body needs the general adjacency build (k_adj_simple could not settle it)
	uint32_t pairs_sph;                   // broadphase pairs with a sphere (they fill the pair list from the back; `pairs` counts the box-box ones)
	uint32_t max_idle[2];                 // by step parity: largest idle counter seen by this step's nh_collide (host-side sleep prediction, nh_collide)
	uint32_t fat_rebuilds;                // broadphase rebuilds since nh_create (the steps in between re-used the kept pair list: below)
	uint32_t sort_reuses;                 // steps since nh_create whose record sort was skipped because last step's tag order still held
	uint32_t fat_inserts;                 // colliders re-inserted into the kept pair list since nh_create (they left their inflated box; no rebuild)
	// "still" steps (nh_collide.hip, "9. still steps"): a step the host launched speculatively as a world whose contact layout is last step's
	uint32_t still_failed_seq;            // the latest still step (its nh_collide's sequence number) that met something that is NOT as it was -- a collider left its box, a body is
	                                      // asleep, a collider pair changed its key or has more than four contacts or a partner the solver lane does not know, body 0 not inert:
	                                      // the solver of that step, and of every still step launched behind it before the host has looked, leaves without touching anything;
	                                      // the host replays from that step in full.  Only ever raised (atomicMax): no clearing, no window in which a verdict can be lost
	uint32_t ahead_multi;                 // k_ahead_check ("xform ahead", nh_StillStep): some body carries more than one collider -- its solver lane cannot stand in for k_xform<true>
	uint32_t lay_valid;                   // the kept layout (per record: first contact | count, feature words; per body: class + contact list record) describes the contact
	                                      // list the last FULL step laid out, and that step met the conditions a still step relies on (k_gather_contacts)
	uint32_t has_unstable;                // some body with contacts is outside the class a still step's solver owns (one collider pair, with the static world, <= 4 contacts): k_adj_simple
	uint32_t still_diff[4];               // why the last still step that failed did: records whose key / contact count / feature words differed, colliders outside their boxes or bodies asleep (statistics)
	uint32_t pair_unowned;                // k_pair_list ("pair ahead", nh_StillStep): kept pairs that are NO body's contact-list record (two neighbours whose inflated boxes overlap, a body
	                                      // hovering above ground it does not touch yet): no solver lane evaluates them -- they are listed, and k_pair_begin does (more than the list holds: no pair ahead)
	uint32_t pair_diag[4];                // why pair-ahead steps were refused, summed since nh_create: roles undecidable / record not the lane's to evaluate (key, count, partner) /
	                                      // the next frame's cells too large / a kept pair that is nobody's (statistics)
	// ---- (the 41 words above are what nh_read_counts mirrors to the host: NH_COUNTER_WORDS) ----
	// Temporal coherence of the broadphase (nh_collide.hip, "kept pair list"): the grid search runs over AABBs inflated by `fat_margin` and its
	// pairs are KEPT; while every collider's AABB stays inside its inflated box the kept list is a superset of the overlapping pairs and one filter
	// pass (exact AABBs, exact test) replaces the whole regrouping + search.
	uint32_t fat_count;                   // pairs in the kept list
	uint32_t fat_valid;                   // the kept list and the inflated boxes describe the current collider arrays
	uint32_t fat_escaped;                 // this step: some AABB left its inflated box (k_xform)
	uint32_t fat_rebuild;                 // this step: the grid chain runs (k_grid_setup: !fat_valid || fat_escaped)
	float fat_margin;                     // inflation of the boxes a rebuild writes this step (k_grid_setup -> k_cell_keys)
	uint32_t fat_count_sph;               // kept pairs with a sphere (they fill the kept buffer from the back; fat_count counts the box-box ones)
	uint32_t fat_hits, fat_hits_prev;     // kept pairs that overlapped in this / the last k_kept_filter pass
	// incremental insertion (nh_collide.hip, "7. re-insertion"): a few colliders left their boxes -- they get new boxes, their old pairs die (generation
	// stamps) and their new pairs are appended; the grid of the last rebuild keeps serving everybody who has not moved since
	uint32_t esc_count;                   // this step: colliders that left their box and were given a new one (k_xform)
	uint32_t moved_count;                 // colliders re-inserted at least once since the last rebuild (their grid entry is stale: the `moved` list)
	uint32_t fat_reinsert;                // this step: k_reinsert has work (k_grid_setup: no rebuild, esc_count > 0)
	// sort re-use (nh_collide): this step's records sit where last step's sat and carry the same keys => last step's tag order still holds
	uint32_t keys_changed;                // some record's key differs from the key that position had last step (k_narrowphase)
	uint32_t records_kept, records_prev;  // records of the step the kept tag order belongs to / as saved at the start of this step
	uint32_t sort_valid;                  // the kept tag order (context buffers) is a sort of the keys k_narrowphase compared against
	// scene bounds of AABB min corners, order-preserving uint encoding (atomicMin/Max)
	uint32_t smin[3];
	uint32_t smax[3];
	// the same over "small" (grid) colliders only: the broadphase grid spans just these
	uint32_t gmin[3];
	uint32_t gmax[3];
	// broadphase grid parameters (written by nh_grid_setup: the last workgroup of k_xform)
	float morton_scale;
	float morton_min[3];          // scene_min * scale
	float cell_inv;               // 1 / cell size (cell size is a power of two)
	float origin[3];
	uint32_t dims[3];
	uint32_t small_exp;           // colliders whose size class (extent bits >> 21: exponent + 2 mantissa bits) is <= this go to the grid (kept across steps: next step's guess)
	uint32_t redo_small;          // (unused)
	uint32_t exp_hist[1024];      // histogram of size classes
	uint32_t any_idle[2];         // by step parity: some body's idle counter is saturated (asleep)
	uint32_t sort_buckets;        // buckets of this step's tag sort = splitters of the previous step + 1 (nh_util.hip, bucket sort)
	uint32_t sort_buckets_next;   // written by this step's sort for the next one
	uint32_t colour_cursor;       // append cursor of the colouring's list compaction (nh_solve.hip)
	uint32_t stale_hint;          // a body was asleep in a step for which the host had ruled that out (k_collide_begin -> k_active_write -> NH_ERR_STALE_HINT)
	uint32_t order_conflict;      // exact-order mode: sequence number of the setup call whose closed-form schedule has a lane conflict (nh_solve.hip)
	uint32_t lay_in_place;        // this (full) step's k_kept_filter wrote the pairs IN PLACE: a record's position is its kept pair's position
	uint32_t delta_count[2], delta_overflow[2];   // by step parity: collider pairs whose contact count changed in this still step (their tag-order position and the change are listed
	                                              // in nh_context::still_delta; more than NH_DELTA_MAX of them: the counts are scanned again instead)
	uint32_t still_smin[2][3], still_smax[2][3];   // by step parity: scene bounds of a still step (k_xform<true>; cleared for the next step by the step before)
	uint32_t asleep_failed;                // k_asleep_check (nh_step, asleep steps): something is not what it was when the world went to sleep
	uint32_t still_top[2];        // by step parity: largest idle counter a still step saw
	// still steps with movers (nh_collide.hip, "9. still steps", LOCAL speculation): colliders that left their inflated boxes in a still step are given new boxes and
	// re-inserted into the kept pair list by the step itself (k_xform<true> with stamps, k_reinsert in still mode) instead of failing it
	uint32_t still_esc[2];        // by step parity: colliders a still step re-boxed (the esc_count of a full step; cleared for the next step by the step before)
	uint32_t still_fat0;          // kept pairs (both kinds) when this still step began: what k_reinsert's "crowded" verdict is taken from (the list grows while it runs)
	// still steps with SLEEPERS (nh_collide.hip, "9. still steps", LOCAL speculation): by step parity, what a still step in sleepers form counted -- bodies asleep, their
	// collider pairs (the step's sleeping pairs), the cache entries kept aside for them; the solver turns them into the step's counters (active / sleeping / culled)
	uint32_t still_asleep[2], still_sleeping[2], still_culled[2];
	// (the narrowphase's share -- sleeping pairs, entries kept aside -- arrives in NH_SLEEP_PARTS places per parity: four thousand workgroups on ONE address queued for
	//  12 ns each, 90 us of a launch that had 10 us of work; cleared by the sleepers-form k_xform<true> of the same step, summed by the solver's first thread)
	uint32_t sleep_part[2][64][2];
	// XFORM AHEAD (nh_StillStep): a still step's solver has done the NEXT step's k_xform<true> for its body's collider -- what it found (a collider outside its inflated box,
	// a body whose idle counter is saturated) fails that next step, not its own: the sequence number of the step it fails; and the part of the scene bounds no solver lane
	// owns (the static world's colliders)
	uint32_t ahead_failed_seq;
	uint32_t still_static_min[3], still_static_max[3];
	// ... and what the lanes gathered for the next step, by step parity: NH_AHEAD_PARTS partial results (min corner bounds 0-2 / 3-5, largest idle counter 6) that the
	// waves of the solver spread their atomics over -- 16 K waves on ONE address would queue for 12 ns each; the next step's narrowphase reduces them
	uint32_t ahead_part[2][64][8];
	nh_BlkGrid blk;               // spatially blocked solver (nh_blocks.h)
};

struct nh_TimerSlot { const char* name; hipEvent_t a, b; };

// what the fused one-body solver does besides solving (NH_FLAG_FUSED_STEP): bits 1 = gravity / damping on load, 2 = advance on store
#define NH_AHEAD_PARTS 64u
#define NH_SLEEP_PARTS 64u
struct nh_xform;
// what a solver lane needs to do the next step's k_xform<true> for its body's collider (nh_StillStep: xform ahead)
struct nh_AheadView { const uint32_t* body_col; const nh_Transform* box_xf; const nh_BoxCollider* box_data; const nh_Transform* sph_xf; const nh_SphereCollider* sph_data; uint32_t nbox;
                      nh_xform* xf; float4* aabb_min; float4* aabb_max; const float4* fat_box;
                      // PAIR AHEAD (nh_StillStep): what the lane needs to evaluate its body's own kept pair for the next sub-step -- the kept list, the keys on file, the stamps and
                      // tags -- and where the result goes: the record's raw contact slots, its count in the tag order, the next step's list of count changes
                      const uint2* kept; uint32_t kept_cap, pair_cap; const uint64_t* rec_key; const uint32_t* ctag; const uint8_t* gen;
                      struct nh_Record* rec; nh_Contact* raw_data; uint32_t* raw_feature; uint32_t* cnt_sorted; int2* delta; uint32_t unowned_cap, delta_scan; };
struct nh_FusedStep { float gx_dt, gy_dt, gz_dt, damping, time_step; uint32_t bits; uint8_t* idle; uint32_t seq, guard_seq;
                      uint32_t ghost_first;
                      uint32_t* host_counters;
                      // HALO SPLIT (nh_partition_step, round 6): the solver of a pair step as TWO launches on two streams -- the bodies a neighbour waits for (listed for it, its
                      // ghosts here, the rehearsal's loop-back records: `xlist`, *xcount of them, part = 1) and everybody else (part = 2: `xskip` marks the listed ones) -- so that
                      // the halo of the next sub-step travels while the interior is still being solved
                      uint32_t part; const uint32_t* xlist; const uint32_t* xcount; const uint8_t* xskip; };      // (nh_step, late verdicts: where the launch's first thread leaves the step's counters for the host -- pinned memory; nullptr: the host copies them itself)      // (nh_partition_step: bodies from here on are ghosts -- their lanes solve and advance them but leave the next step's transforms and pairs alone: the ghost arrives anew)
//      // (still steps: this step's sequence number; the
                                                                                                                                    // oldest still step whose verdict the host has not seen)

// library-owned buffers of the spatially blocked solver (nh_blocks.h), grown on demand
struct nh_BlkBuffers {
	uint2* brank;                     // per body: block << 12 | rank in the block, position inside the block (3 x 10 bits); NH_NONE = not a general body
	float4* bm;                       // momentum of the general bodies in block order (slot 0: the static world), 2 x float4 per body
	uint32_t* pos_body;               // NH_BLK_CHECK=1 only: body index by block-order slot
	uint32_t* touch;                  //   and who touched a slot in the phase being checked
	uint32_t body_capacity;
	uint32_t* lpair;                  // per contact position: local indices la | lb << 16 (| NH_BLK_RADIAL) into the owning block's LDS
	uint32_t* ghost_list;             // per block: block-order slots of the neighbour bodies it borrows
	uint32_t contact_capacity;
	uint32_t* blk_count; uint32_t* own_base;              // NH_BLK_MAX_BLOCKS + 1 each (+ padding)
	uint32_t* bins; uint32_t* bin_off;                    // (NH_BLK_MAX_BLOCKS + 1) x NH_BLK_MAX_COLOURS + 1: contacts by (owning block, colour); the last "block" = residual
	uint32_t* ghost_base; uint32_t* ghost_count;          // per block
	uint32_t* phase_list;                                 // 8 x NH_BLK_MAX_BLOCKS: blocks with work, by block colour
	uint32_t* desc;                                       // ... and what a sweep workgroup needs to know about each of them (NH_BLK_DESC_WORDS words)
	uint32_t* scan_tmp;
	bool tables;
	float scale;                      // block edge relative to the edge at which an average block holds `target` bodies: adapted from the largest load seen
	uint32_t target, min_contacts;    // NH_BLK_TARGET, NH_BLK_MIN
	bool disabled, check;             // NH_NO_BLOCKS=1, NH_BLK_CHECK=1
	// chained sweeps (option "blk_chain", nh_blocks.h: k_blk_sweep<.., CHAIN>): the eight block colours of a pass as ONE launch; a workgroup draws its task from a
	// ticket counter in (colour, block) order and waits only for the neighbour blocks of earlier colours -- no launch-wide tail between two colours
	bool chain; uint32_t* chain_slot; uint32_t* chain_nbr; uint32_t* chain_done; uint32_t* chain_ticket;      // block -> task; per task: count + tasks it waits for; per task: last pass done; per pass: ticket counter
	unsigned long long* prof; uint32_t prof_calls;       // NH_BLK_PROFILE=1: wall-clock readings of the sweep workgroups (8 launches x NH_BLK_MAX_BLOCKS x 4), printed now and then
};

// ---- still steps (nh_collide.hip, "9. still steps") --------------------------------------------------------------------------------------------------
// A world at rest on static geometry repeats itself: the same pairs at the same places of the kept pair list, the same contacts in the same order, every body
// with its one contact-list record, the contact cache = last step's contact list.  Everything a step does to FIND that out again -- grid chain, pair filter,
// tag sort, scans, contact gather, adjacency, cache merge: two dozen launches, most of them leaving at once -- is then skipped by the HOST, which launches
// three kernels (k_xform<true>, k_narrowphase<*, true>, the solver reading and writing the cache in place); they CHECK what they rely on (nobody left its
// inflated box or is asleep; every record has the key, contact count and feature words it had; body 0 inert) and raise st->still_fail otherwise -- after which
// each of them leaves without having changed anything that matters, and the host, which learns of it with the step's one round trip, replays the step in full.
// THE ONE-KERNEL STILL STEP (round 6; nh_fused.hip).  A plain still step that starts at the narrowphase (xform ahead) and is not the last sub-step of its nh_step call
// is ONE launch: the lane that solves a body evaluates the body's one kept pair itself and keeps the contacts on chip.  The kernel reads body state, slot cache and
// collider transforms from one set of buffers and writes the other (a check that fails in one wave must find the world as it was, although other waves have long
// advanced their bodies): `cur` says where the BODY state lives -- 0: the caller's arrays, 1: the shadow arrays below -- while the library-owned buffers simply
// change names with their twins (ctx->sc_* / own_* always name the current ones).  Fused steps come in pairs, so a call ends in the caller's arrays.
struct nh_FusedState {
	bool enabled;                   // option "fused_still" (OFF by default: measured slower than the two launches it replaces -- nh_fused.hip)
	uint32_t debug_fail_in;         // option "fused_fail_in" (tests): the n-th fused launch from now fails itself after its waves have written
	uint32_t cur;                   // which set holds the body state
	uint32_t unconfirmed;           // fused launches whose verdict the host has not seen (each changed the sets over: a failure changes them back)
	uint64_t steps, homecomings;    // statistics: fused steps that happened; copies of the shadow arrays back into the caller's outside the pairing
	unsigned long long* seg; uint32_t seg_words; bool seg_clean[2];          // segment sums of the look-back, by step parity (nh_fused.hip)
	uint4* fz; unsigned long long* status; bool map_valid; uint32_t map_records;      // per tag-order position: record, body, colliders (k_fused_map); the look-back words
	bool statics_copied;            // the static world's collider transforms / boxes are in both sets (since the last k_xform<true>)
	float4* sc_imp_alt; uint32_t* sc_feat_alt; uint32_t* sc_count_alt; uint32_t pair_capacity;
	struct nh_xform* own_xf_alt; float4* own_aabb_min_alt; float4* own_aabb_max_alt; uint32_t collider_capacity;
	nh_Transform* shadow_xf; nh_BodyMomentum* shadow_mom; uint8_t* shadow_idle; uint32_t body_capacity;
};

struct nh_StillStep {
	bool disabled;                  // nh option / NH_NO_STILL=1: never launch a step speculatively (A/B, tests)
	bool ok_next;                   // the last step left a layout, body classes and a cache that a still step may rely on (decided at its round trip / end)
	bool cache_ok;                  // the caller's contact cache IS the contact list of the last step (nh_write_cached_impulses wrote it with nothing culled)
	const void *cache_tags, *cache_features, *cache_data; uint32_t cache_capacity;
	bool active, resolved;          // this step was launched as a still step / ... and its round trip has confirmed it
	bool replaying;                 // inside the replay of a failed one
	bool appended;                  // nh_append_contacts changed the list since the layout was written
	bool slots_current;             // the slot cache (sc_*) holds the impulses of the last step (imported at the end of a full step that allows a still one, then kept by the solver)
	bool cache_stale, contacts_stale;   // still steps have run since the caller's cache arrays / dense contact arrays were last written
	nh_ContactCache cache;          // the caller's cache arrays (as given to the last nh_write_cached_impulses)
	// nh_step drives the sample's loop itself, so it can look at a still step's verdict ONE STEP LATE: the host never waits for the GPU inside the loop (the copy of the
	// counters it looks at has long landed), and the GPU never waits for the host.  A still step that failed did nothing, nor did the one launched behind it (their
	// solvers compare st->still_failed_seq with `guard`): nh_step runs both again, the first in full.
	bool pipelined;                 // inside nh_step
	struct { bool pending; uint32_t seq, parity; int slot; uint64_t collide_mark; bool self_report; } verdict;      // the still step whose counters the host has not looked at yet (self_report: the solver's first thread left them in the ring slot, its number behind them -- the host may poll for that word instead of waiting for the launch to END: early counters)
	nh_DevState* h_ring[2]; hipEvent_t ev_ring[2];      // pinned landing places of the counter copies, and when they have landed
	bool ring_failed;                                    // ... could not be created: nh_step looks at every verdict inside its step
	nh_ContactConstraintData* setup_d;      // its nh_setup_contact_constraints has run (without the adjacency kernel the replay then owes)
	// the arguments of its nh_collide, for the replay; and the arrays the kept layout belongs to (another world in the same context: not still)
	nh_ActiveBodies* active_out; nh_ContactData* contacts_out; nh_BodyData bodies; nh_ColliderData colliders; nh_BodyConnections connections; bool has_connections; nh_Arena temporary;
	nh_BodyData lay_bodies; nh_ColliderData lay_colliders; nh_ContactData lay_contacts; const uint32_t* lay_active; uint32_t lay_active_capacity;
	uint64_t launched, failed;      // statistics
	// PAIR AHEAD (round 6).  With xform ahead a plain still step was two launches: the narrowphase over the kept pairs and the solver.  In the world this form runs in nearly
	// every kept pair is some dynamic body's ONE pair with the static world (the few that are nobody's are listed once per layout: k_pair_list) -- so the solver lane that has just advanced a body and
	// written its collider's new world transform and box (xform ahead) goes on and evaluates that pair for the NEXT sub-step too (nh_box_box_eval & co. on what it holds in
	// registers + the static partner's transform, box and shape): the contacts go to the record's raw slots, the count to its place in the tag order, a change of the count
	// to the next step's list -- exactly what that step's narrowphase would have written.  The next sub-step is then ONE launch of the solver behind a small
	// prologue (k_pair_begin: the words the narrowphase's first wave used to clear, file and check, and the listed pairs that are nobody's, evaluated the old way).  No wave waits for another; nothing is written out of place: what
	// the lane finds wanting (another key, a fifth contact, a partner that is not where the rule below needs it) fails the NEXT step before it starts, like a leaver.
	// Which collider of a pair plays "a" is decided by the Morton order of the AABB min corners in the NEXT step's scene frame (nudge.cpp:3086-3100), which no lane knows;
	// but when one corner is <= the other in all three coordinates and clearly less in one (two quantisation cells of THIS step's frame), it comes first in the Morton
	// order of ANY frame whose cells are not much larger -- k_pair_begin checks that the next frame's are not (scale >= 3/4 of this one's).  A box on a slab always is.
	bool early_verdict;             // nh_partition_step, pair step: the counters behind k_pair_begin are on their way to h_ring[0] -- the step's verdict, ahead of its solver
	bool pair_ready;                // the solver of the LAST step launched evaluated the pairs of this one: the next nh_collide of this call launches k_pair_begin instead of the narrowphase
	bool pair_step;                 // the still step being launched started at the solver (statistics below)
	bool no_pair;                   // option "no_pair_ahead"
	uint32_t pair_owned_seq;        // the nh_collide (collide_seq) in which the kept pairs that are nobody's were last listed; 0: the list is void (a full step has laid the records out again)
	bool pair_world_bad;            // a round trip showed more such pairs than the list holds: not offered until the next full step
	bool pair_world_ok;             // ... a round trip showed the list complete (for the layout it was made from): only then are the lanes asked to evaluate their pairs
	uint64_t pair_steps;            // statistics: still steps that were ONE solver launch (+ its one-workgroup prologue)
	bool fused;                     // the still step being launched is a fused one (nh_FusedState): nh_collide launched nothing, the solver call launches k_still_fused
	bool verdict_fused;             // ... and so was the step whose verdict is pending
	uint32_t steps_left;            // nh_step: sub-steps of this call behind the one being launched
	uint32_t confirmed_seq;         // collide_seq of the newest still step whose verdict was "confirmed" (every drop noted in ctx->sc_undo under a later number is undone on failure)
	bool undo_dirty;                // a narrowphase in sleepers form has run since sc_undo was last swept
	// back-off: a world in which something keeps moving (a body in the air leaves its inflated box every step) fails every still step it is offered; after the k-th
	// failure in a row the next 2^k - 1 steps (at most 63) are not offered as still steps.  Sixteen confirmed still steps in a row forget the failures.
	uint32_t hold, fail_level, good_run;
	uint32_t seen_inserts, seen_rebuilds;          // re-insertion / rebuild counters at the last round trip (finish_setup, a still step's verdict: did anybody leave its box in that step?)
	// LOCAL speculation (round 5).  A still step used to fail as a whole when ONE collider left its inflated box: a world of a million boxes at rest paid the full
	// step for a single box in the air (-42 %).  While somebody has been leaving boxes lately (`movers_left` steps to go: renewed by every round trip that shows a
	// re-insertion) still steps are launched in MOVERS form: k_xform<true> re-boxes the leavers like a full step's does, k_reinsert appends their new pairs to the
	// kept list, and the still narrowphase takes a kept pair that has died (stamps) or no longer overlaps as a record without contacts, and the appended pairs as
	// pairs without a record that must not produce a contact -- a landing is still a full step.  Option "no_local_still": the round-4 behaviour.
	bool no_local; uint32_t movers_left;
	// XFORM AHEAD (round 5, verdict 2c).  k_xform<true> was 40 of a still step's 300 us at a million boxes, and cannot be folded into the narrowphase that follows it:
	// which collider of a pair plays "a" depends on the scene bounds of THIS step (nudge.cpp:3086-3100), a reduction over every collider.  But every dynamic body of a
	// still step has a solver lane that has just advanced it (NH_FLAG_FUSED_STEP) -- and between two sub-steps of ONE nh_step call the caller cannot touch anything.  So
	// the lane does the next step's k_xform for the body's one collider while the transform is still in registers: world transform, AABB, the test against the inflated
	// box, the bounds, the largest idle counter.  The next step then starts at the narrowphase (two launches instead of three); what the lane found wanting fails THAT
	// step (st->ahead_failed_seq), which is run again in full like any failed still step.  Only in the plain form (nobody moving, nobody able to sleep), only for worlds
	// whose dynamic bodies carry one collider each (checked once per call: k_ahead_map / k_ahead_check), only while another sub-step of the same call follows.
	// Option "no_xform_ahead": every still step launches k_xform<true> (A/B, tests).
	bool no_ahead;                  // option
	bool more_steps;                // nh_step: another sub-step of this call follows the one being launched
	bool ahead_map_ok;              // the body -> collider map was built in this nh_step call (the caller cannot have changed the colliders since)
	bool ahead_world_bad;           // ... and showed a body with several colliders (st->ahead_multi, seen at a round trip): not offered again until the colliders change
	bool ahead_plain;               // the still step being launched is in the plain form and has a narrowphase launch
	bool ahead_ready;               // the solver of the LAST step launched did this step's transforms: the next nh_collide of this call may skip k_xform<true>
	uint64_t ahead_steps;           // statistics: still steps that started at the narrowphase
	bool appended_pairs;            // a still step in movers form has run since the last full step: the kept list may be longer than the layout (the narrowphase's LOCAL form is needed)
	// SLEEPERS form (round 5).  Still steps needed everybody awake, asleep steps everybody asleep; a world in which SOME bodies sleep paid full steps (~40 launches).  The
	// reference's rule is per set (nudge.cpp:3575-3703): a body whose AABB overlaps no other dynamic body's forms a set of its own, asleep exactly when its idle counter
	// is saturated; its pairs with the static world become sleeping pairs (larger tag low, 3697), it is not active (no gravity, no advance), and its cached impulses are
	// kept aside and written back (4064-4101) when the cache tag equals the sleeping pair's word.  All of that is local to the body: in sleepers form k_xform<true> marks
	// the colliders of sleeping bodies (and counts them), the still narrowphase files their pairs as sleeping records without contacts -- an overlapping pair of two
	// DYNAMIC bodies with a sleeping end means a set of more than one body: the step fails, a full step sorts it out -- and the solver leaves sleeping bodies alone.
	// Active list and sleeping pairs are VIEWS like the contact list (nh_export_views).
	bool sleepers;                  // this still step was launched in sleepers form
	uint32_t substep;               // nh_step: index of the sub-step being launched within its call (0: the first, or a step driven through the stage calls)
	// SLEEPERS AHEAD (round 6): while the set of sleeping bodies stands still -- the host has seen the same active count in the last few confirmed steps -- a step in sleepers form
	// is launched like a plain one: xform ahead and pair ahead by the lanes of the AWAKE bodies, one solver launch per sub-step behind k_pair_begin.  Sleeping bodies are part of
	// the static world for such a step: their colliders' share of the scene bounds is taken once with the static world's (k_ahead_check), their records stay the settled sleeping
	// pairs they are, the step's sleeper counts are the last step's (k_pair_begin).  A body that falls asleep fails the NEXT step like in the plain form (the lane that advanced it
	// sees the saturated counter): that step is replayed in full, and the form is left alone for 8 steps -- 16, 32, 64 when failures follow each other within 32 pair steps.  Option "no_sleeper_ahead": off.
	bool no_sleeper_ahead;
	bool was_sleepers;              // the last still step was launched in sleepers form
	uint32_t sleep_backoff_len, sleep_run;          // how long the form is left alone after a failure (8 steps, doubled up to 64 by failures in quick succession), pair steps in sleepers form since the last one
	uint32_t sleep_last_active, sleep_stable, sleep_backoff;      // the active count of the latest confirmed sleepers-form step, how many in a row reported it, steps until the form may be tried again
	bool no_sleeper_skip;           // option "no_sleeper_skip": the sleepers form does everything for every collider / pair / wave, asleep or not (A/B, tests)
	bool views_sleepers;            // sleepers-form still steps have run since the caller's active list / sleeping pairs were last written by a full step
	bool sleep_pairs_current, active_current;       // ... and those two views have been exported since the last such step
	void note_failure() { hold = (1u << fail_level) - 1u; if (fail_level < 6u) ++fail_level; good_run = 0; }
	void note_confirmed() { if (++good_run >= 16u) fail_level = 0; }
};

// ASLEEP STEPS (nh_step only).  A world in which every body is asleep is a fixed point of the step: no active body, no contact, every overlapping pair a sleeping pair,
// the cache kept aside and written back as it was (nudge.cpp:3669-3703, 4064-4101, 4896-4898) -- the step reads everything and changes nothing.  When two full steps in
// a row have shown exactly that (their round trips: nobody active, no contact, the same pair / sleeping-pair / cache counts as the step before) the library
// keeps a copy of the colliders' world AABBs and tags; nh_step then verifies ONCE per call, with one kernel, that all of them are what they were and that every body is
// still asleep (the only things a caller could have changed between two calls without telling: nh_bodies_changed and friends end the regime themselves) -- and, if so,
// the steps of that call are done: nothing inside the library can wake a world nobody is awake in.  Anything else runs the steps in full.
struct nh_AsleepState {
	bool disabled;                   // option "no_asleep"
	uint32_t streak;                 // full steps in a row that were all-asleep fixed points
	uint32_t prev_pairs, prev_sleeping, prev_cache;
	float4 *aabb_min, *aabb_max; uint32_t* tags; uint32_t capacity;          // library-owned copies (world AABBs and tags of the step that made the streak 2)
	uint32_t C, B;                   // colliders / bodies of that step
	uint32_t snap_C, snap_seq;       // what the copies hold right now: colliders and the nh_collide call (collide_seq) they were taken in (0: nothing)
	const float4 *last_aabb_min, *last_aabb_max; const uint32_t* last_ctag; uint32_t last_C;     // this step's arrays in the caller's arena (nh_collide)
	uint64_t steps;                  // statistics
};

// state streaming (include/nudge_hip.h: nh_stream_state)
#define NH_STREAM_MAX_SLOTS 16
struct nh_StateStream {
	uint32_t every, slots, count; const nh_Transform* src; char* ring;
	nh_Transform* stage; uint32_t stage_capacity;                 // device staging copy (the world goes on while the frame travels)
	hipEvent_t staged, landed[NH_STREAM_MAX_SLOTS]; bool in_flight[NH_STREAM_MAX_SLOTS];
	uint64_t slot_step[NH_STREAM_MAX_SLOTS], slot_frame[NH_STREAM_MAX_SLOTS];
	uint64_t advances, frames, dropped; int last_slot;
};

struct nh_context {
	int device;
	int cu_count;                  // compute units of the device (persistent launches size themselves by it)
	hipStream_t stream;
	uint32_t flags;
	uint32_t tag_bits;
	nh_DevState* d_state;
	nh_DevState* h_state;          // pinned mirror
	// EARLY COUNTERS (round 6): a full step's counters are final when its one-body solver STARTS (nothing in that kernel writes them), so the launch's first thread leaves
	// them in this pinned block -- NH_COUNTER_WORDS words, then the nh_collide number as the word that says "complete" -- and the host, polling that word, has them
	// while the solver runs: what it launches next queues up behind the solver instead of behind a copy and a wake-up on an idle GPU (26 us of a 550 us step).
	uint32_t* h_early;             // pinned, NH_COUNTER_WORDS + 16 words; nullptr: could not be had, the host copies
	bool no_early_counts;          // option "no_early_counts" (A/B, tests): the copy behind the solver, as before
	uint64_t early_reads, early_fallbacks;      // round trips answered by the solver's first thread / by the copy after all (the stream ran dry without the word)
	int last_hip_error;
	// timing
	bool timing;
	bool timing_current;           // the launch being issued is timed (filter matched)
	std::string timing_filter;
	std::vector<nh_TimerSlot> timer_slots;
	std::vector<hipEvent_t> event_pool;
	std::map<std::string, std::pair<double, uint32_t>> kernel_ms;
	std::vector<std::string> name_store;
	// opaque result objects handed to the caller (ring)
	std::vector<nh_ContactImpulseData*> impulse_ring;
	std::vector<nh_ContactConstraintData*> constraint_ring;
	uint32_t ring_pos;
	// last collide: number of bodies (for sizing), contact capacity
	uint32_t body_count;
	// deferred one-body setup (see nh_ContactConstraintData::setup_pending)
	nh_ContactConstraintData* pending;
	// warm-start hints (per body: index of its first contact in the previous step's contact list); library-owned device memory
	uint32_t* hint;
	uint32_t hint_capacity;
	// per-body contact degree (filled by nh_collide's contact gather, turned into CSR offsets by setup) + fill cursors: 2 x (B + 2) words
	uint32_t* deg;
	uint32_t deg_capacity;
	// sort re-use (library-owned, pair capacity entries each): record keys by record position, and the tag order the last sort left
	uint64_t* sort_keys_by_position; uint64_t* sort_sorted_keys; uint32_t* sort_sorted_idx;
	bool env_no_sort_reuse;               // NH_NO_SORT_REUSE=1 (A/B, tests)
	// kept pair list of the broadphase (library-owned, persists from one nh_collide to the next): pairs of inflated boxes, and the boxes
	uint2* fat_pairs; uint32_t fat_pair_capacity;
	float4* fat_box; uint32_t fat_collider_capacity;
	// the grid of the last rebuild, kept for re-insertions: cell-sorted inflated boxes, their cell keys, cell starts, the large colliders; per collider
	// a generation byte (bit 7: large) and the nh_collide call it last left its box in; this step's leavers, everybody who has left since the rebuild
	float4* grid_sbox; uint32_t* grid_skeys; uint32_t* grid_cstart; uint32_t* grid_counts; uint32_t* grid_large; uint32_t grid_cells, grid_collider_capacity;
	uint8_t* fat_gen; uint32_t* fat_esc_mark; uint32_t* fat_esc_list; uint32_t* fat_moved_list;
	bool env_no_incremental;
	bool sync_exports_views;              // option "sync_exports_views": nh_synchronize / nh_read_counts also bring the caller's contact list, cache and active list up to date (the contract before round 5)
	bool env_blk_rows_global;             // NH_BLK_ROWS_GLOBAL=1: the rows of block-owned contacts by k_rows_general (one thread per contact) instead of k_blk_rows (A/B, tests)
	bool env_blk_global_colours;          // NH_BLK_GLOBAL_COLOURS=1: the blocked solver takes the world-wide colouring (A/B, tests)
	uint32_t last_general_contacts;       // general contacts of the last setup: does this step's adjacency build leave the general bodies to the per-block colouring?
	uint32_t fat_nbox, fat_nsph;          // collider counts of the last nh_collide (a change invalidates the list)
	uint32_t bp_seen_rebuilds, bp_rebuild_streak; int bp_direct_left;      // direct search instead of the kept list while every step rebuilds it (nh_collide)
	bool env_measure_skip;                // NH_MEASURE_SKIP_CHAINS=1: measurement only (nh_collide)
	bool env_no_fat;                      // NH_NO_KEPT_PAIRS=1: rebuild every step (A/B, tests)
	uint32_t step_parity;
	uint32_t order_seq;
	uint32_t pair_capacity;               // nh_set_pair_capacity; 0 = contacts->capacity / 2 + 1024
	uint32_t collide_seq, setup_seq;      // one nh_setup_contact_constraints per nh_collide (NH_ERR_STALE_SETUP)
	// Sleep prediction.  A body is asleep when its idle counter is saturated (0xff) and nh_advance raises a counter by at most one, so from the
	// largest counter an earlier nh_collide saw (mirrored with the step's one round trip) the host knows how soon anybody CAN be asleep:
	// until then the island / sleeping kernels are not launched at all.  idle_bound < 0: unknown (start, nh_bodies_changed, halo unpack).
	int idle_bound;                       // largest idle counter at the nh_collide that `idle_bound_mark` belongs to
	bool idle_unknown;                    // idle counters were rewritten from outside (nh_bodies_changed, nh_halo_unpack) since the last nh_collide looked at them:
	                                      // what that nh_collide measured says nothing about them (nh_read_counts must not adopt it); cleared by nh_collide
	uint64_t advance_count, collide_mark, idle_bound_mark;     // nh_advance calls so far; their number at the last nh_collide / at the measuring one
	bool islands_skipped;                 // the last nh_collide launched no island kernels (its prediction is checked on the device: NH_ERR_STALE_HINT)
	// behaviour switches read ONCE from the environment by nh_create (test / A-B knobs, see nh_context.hip)
	bool env_sort_classic, env_sort_radix, env_colour_jp, env_no_resident, env_colour_check_seeds;
	int env_solver_waves, env_fp_sub;
	bool os_plain_launch;                 // cooperative launches are not available here: the radix passes use plain launches bounded by os_resident
	int os_resident;                      // co-resident workgroups of a one-kernel radix pass on THIS device (nh_util.hip), 0 = not yet asked
	// tag sort seeded by the previous step's order (nh_util.hip): splitters persist across steps; counts / starts are scratch that must
	// outlive nothing but is zeroed by its own kernels.  `sort_seeded`: the splitters describe the previous nh_collide of this capacity
	uint64_t* sort_splitters;
	uint32_t* sort_counts;
	uint32_t* sort_starts;
	uint32_t sort_capacity;
	uint32_t sort_tile, sort_target;      // test knobs (NH_BUCKET_TILE, NH_BUCKET_TARGET), 0 = defaults
	bool sort_seeded;
	// fork/join inside one entry point: independent kernel chains run on `side` while the caller's stream does the long one
	hipStream_t side;
	hipEvent_t ev_fork, ev_join;
	// gravity / damping beside the tag sort: nh_collide records `ev_active` right after the active list is written; an nh_apply_gravity_damping
	// that FOLLOWS nh_collide directly (gravity_may_overlap) runs on the side stream behind that event -- concurrently with the sort / gather
	// chain still queued on the caller's stream, which never touches momentum -- and the caller's stream waits for `ev_gravity`
	hipEvent_t ev_active, ev_gravity;
	bool gravity_may_overlap;
	// NH_FLAG_FUSED_STEP: gravity / damping recorded by an nh_apply_gravity_damping that directly follows nh_collide (`after_collide`) and applied by
	// the fused one-body solver when it loads a body's momentum (everybody else: k_gravity_rest); the advance that solver performed for its bodies
	bool after_collide;
	struct { bool pending, rest_pending; float gx_dt, gy_dt, gz_dt, damping, time_step; const uint32_t* active; nh_BodyMomentum* momentum; uint32_t body_count; } grav;
	struct { bool done, rest; float time_step; const uint8_t* body_class; } adv;      // `rest`: some active body is not the fused solver's (the step's round trip told)
	nh_BlkBuffers blk;
	// Contact storage by SLOT (library-owned; written by every step, full or still).  A collider pair of the (kept) pair list is a RECORD; record r owns the raw
	// contact slots 4 r .. 4 r + 3 (a record of a pair with a sphere: the one slot 4 n_bb + (r - n_bb)), whatever the other records do -- so a pair that gains or
	// loses a contact moves nobody else's data.  The dense, tag-ordered contact list of the API is a VIEW of this: record r sits at position lay_rank[r] of the tag
	// order (kept from the last sort), its contacts start at start_sorted[lay_rank[r]] (the scan of the counts in that order) ranked by feature word.  A full step
	// lays the view out for the caller right away (k_gather_contacts); still steps leave it to the next call that looks (nh_still_sync_outputs).
	nh_Contact* raw_data; uint32_t* raw_feature; uint32_t raw_slots;        // 4 pair_cap + contact capacity slots (the tail: contacts 5+ of a pair, full steps only)
	struct nh_Record* rec;                                                   // per record: the two bodies, overflow position, contact count
	uint32_t* lay_rank;                                                      // per record: its position in the kept tag order
	uint32_t* cnt_sorted; uint32_t* start_sorted;                            // per tag-order position: contact count, first contact (pair_cap + 1 entries)
	uint32_t* dense_slot;                                                    // per contact of the last FULL step's list: the raw slot it came from
	// the contact cache by slot (still steps): the solved impulse and the feature word of the contact each slot held LAST step, and how many each record had
	float4* sc_imp; uint32_t* sc_feat; uint32_t* sc_count;
	// sleepers form: a still step's narrowphase DROPS the slot cache's entries of a pair it files as sleeping under another word than the contact key on file (the
	// reference's quirk, nudge.cpp:4064-4101) before the step's verdict is known.  What it dropped is noted here -- (step sequence number << 32) | old count, by record --
	// and put back when that step turns out not to have happened (nh_still_undo_drops): a replay must find the cache of the last step that DID happen
	uint64_t* sc_undo;
	// still steps keep their colliders' world transforms, AABBs and tags in buffers of the library's own (a full step's live in the caller's arena): with xform ahead
	// (nh_StillStep) they are written by one step's solver and read by the next step's narrowphase; and the body -> collider map those solver lanes go by
	struct nh_xform* own_xf; float4* own_aabb_min; float4* own_aabb_max; uint32_t* own_ctag; uint32_t own_capacity;
	uint32_t* body_col; uint32_t body_col_capacity;
	uint32_t* pair_mark; uint32_t* pair_list; uint32_t pair_list_capacity;      // pair ahead: per record the stamp of the listing that found it some body's own; the records that are nobody's
	uint8_t* still_awake; uint32_t still_awake_capacity;                     // per body: awake at the last sleepers-form still step's nh_collide (the active list is a view of it)
	uint64_t* exp_sleep_a; uint64_t* exp_sleep_b; uint32_t* exp_sleep_hist; uint32_t* exp_flags; uint32_t exp_sleep_capacity, exp_flags_capacity;      // scratch of the view exports (on demand)
	int2* still_delta;                                                       // 2 x NH_DELTA_MAX: (tag-order position, change of the contact count) of this still step, by step parity
	uint32_t* exp_cnt; uint32_t* exp_start; uint32_t* exp_scan_tmp;          // scratch of the export slots -> caller's cache (pair_cap + 2 words each; scan temporaries)
	uint32_t lay_capacity, lay_contact_capacity;                             // pair capacity / contact capacity the arrays above were sized for
	// per body (k_adj_simple): class, contact-list record of the full path; the record a body's contacts come from (| NH_BODY_REC_IS_A) and its tag-order position
	uint8_t* lay_class; uint2* lay_simple; uint32_t* body_rec; uint32_t* body_pos; uint32_t lay_body_capacity;
	nh_StillStep still;
	nh_FusedState fused;
	// nh_partition_step (nh_partition.hip): K sub-steps of a slab in one nh_step call -- before every sub-step's nh_collide the library calls the hook, which enqueues the
	// per-step halo exchange (and, for a step that starts further down the chain, the ghosts' transforms); halo_ghost_first: the first ghost body while such a call runs
	int (*step_hook)(nh_context* ctx, void* user, uint32_t sub_step); void* step_hook_user; uint32_t halo_ghost_first;
	// HALO SPLIT (nh_partition_step; nh_FusedStep::part): armed by the partition for the duration of a call -- the boundary bodies' list / count / mask, the second stream, and
	// what to enqueue on it once a pair step's verdict is in (the next sub-step's exchange: it then travels beside the interior's solver).  `launched`: this step's solver went
	// out in two parts
	struct { bool armed, launched; hipStream_t stream; const uint32_t* list; const uint32_t* count; const uint8_t* mask; uint32_t bound;
	         int (*after_verdict)(nh_context* ctx, void* user); void* user; } halo_split;
	bool halo_overlap;             // option "halo_overlap" (OFF by default: bit-exact, and no faster in the one-rank rehearsal -- profiles/r06_halo_split_ab.log): nh_partition_step sends the
	                               // next sub-step's halo beside the interior's solver instead of between two sub-steps
	nh_AsleepState asleep;
	uint32_t first_ghost;          // nh_set_first_ghost_body: bodies >= first_ghost are ghosts of a partitioned world (0 = no ownership rule)
	nh_StateStream stream_state;
};
int nh_stream_after_advance(nh_context* ctx);          // nh_advance -> state streaming

// Runs work that an earlier call deferred; every entry point that reads or writes momentum / impulses calls it first.
// `in_sequence`: the caller is the next call of the sample's step (gravity, read, setup, apply): a still step stays speculative across it; every other
// entry point first turns a still step that has not been confirmed yet into a full one (nh_still_abandon)
int nh_flush_pending(nh_context* ctx, bool keep_gravity = false, bool in_sequence = false);
int nh_still_abandon(nh_context* ctx);
// The caller's view of what still steps keep by slot -- the dense contact list and the contact cache in tag order -- brought up to date (every entry point outside
// the sample's call order that hands the arrays to the caller: nh_export_views, nh_append_contacts, nh_contact_impulses_device -- NOT nh_read_counts / nh_synchronize)
int nh_still_sync_outputs(nh_context* ctx, uint32_t what = 7u /* NH_VIEW_ALL */);
int nh_still_export_cache(nh_context* ctx);
int nh_still_undo_drops(nh_context* ctx);
// nh_fused.hip
bool nh_fused_wanted(nh_context* ctx, uint32_t B, uint32_t C, uint32_t steps_left);
int nh_fused_launch(nh_context* ctx, const nh_BodyData* bodies, uint32_t iterations, uint32_t guard_seq);
int nh_fused_come_home(nh_context* ctx);
void nh_fused_void_last(nh_context* ctx, uint32_t launches);
void nh_fused_free(nh_context* ctx);
void nh_stream_void_advance(nh_context* ctx);
int nh_still_verdict_now(nh_context* ctx);            // nh_step: the pending verdict of the last still step, waited for: 0 = it happened, 1 = it did not
#define NH_INTERNAL_STILL_FAILED (-1000)      // (never leaves the library) first_apply -> nh_step: the still step before this one failed, run both again
#define NH_DELTA_MAX 256u
#define NH_GEN_SHIFT 24u             // kept pairs carry the generation stamps of their two colliders above the 24-bit collider index (nh_collide.hip, "7. re-insertion")
#define NH_GEN_INDEX 0xFFFFFFu
#define NH_BODY_REC_NONE 0xFFFFFFFFu
#define NH_BODY_REC_IS_A 0x80000000u
struct nh_Record { uint32_t body_a, body_b, overflow, count; };          // one per collider pair that reached the narrowphase (nh_collide.hip)
#define NH_REC_SLEEPING 0x80000000u
void nh_counts_from_mirror(nh_context* ctx, nh_Counts* out);
#define NH_COUNTER_WORDS 41u          // leading words of nh_DevState that the host mirrors (nh_read_counts)

struct nh_ContactImpulseData {
	nh_CachedContactImpulse* data;        // K, contact (tag) order; holds the warm-start impulses once `lookup_pending` is false
	// the warm-start lookup is deferred: bodies on the one-body path fetch their cached impulses inside the fused solver kernel
	bool lookup_pending;
	bool consumed;                        // a setup_contact_constraints took this handle: every contact's impulse gets (re)written by the solver
	nh_context* ctx;
	const uint64_t* tags; const uint32_t* features;                       // this step's contacts
	const uint64_t* ctags; const uint32_t* cfeatures; const nh_CachedContactImpulse* cdata;   // the cache as read
	uint64_t* culled_tags;
	uint32_t* culled_features;
	nh_CachedContactImpulse* culled_data;
	uint32_t capacity;                    // contact capacity the arrays were sized for
	// culling of sleeping pairs' cached impulses, run once the host knows whether anything sleeps
	bool cull_pending;
	const uint64_t* sleeping_pairs;
	uint32_t* cull_flags; uint32_t* cull_tmp;
	uint32_t cache_capacity;
	uint32_t culled_capacity;
};

struct nh_ContactConstraintData {
	float* rows;                          // K x 40 floats (nh_row)
	float* states;                        // K x 4 floats (nh_state + pad)
	uint32_t* body_off;                   // B+1 CSR offsets into adj
	uint32_t* adj;                        // contact indices per dynamic body, solver order
	uint8_t* body_class;
	uint2* simple;                        // per body of k_adj_simple's classes: first contact, count | slot order << 4 (nh_solve.hip)                  // 0 none, 1 static-only, 2 general
	uint32_t* level_order;                // general contacts sorted by level
	uint2* gpair;                         // per general contact, LEVEL order: body pair
	float4* gstates;                      //   solver state (rows of general contacts are structure-of-float4 in `rows`, level order, stride = contact_capacity)
	uint32_t level_off[NH_MAX_LEVELS + 2];// host copy of level offsets
	uint32_t level_full[NH_MAX_LEVELS + 2];// host copy: the level / colour class holds a full (40-float) row, not only radial ones
	uint32_t levels;
	uint32_t early_seq;                   // early counters (nh_context::h_early): the one-body launch of this setup leaves the step's counters for the host under this number; 0: it does not
	uint32_t general_contacts;
	uint32_t contact_capacity;
	uint32_t body_count;
	bool has_static8, has_staticN, static_inert;
	uint32_t resident_cap;                // its row stride in LDS (all rows radial: 1792, else 704)
	bool resident_bodies;                 // small world, larger general set: every body's momentum in one workgroup's LDS (k_solve_resident_bodies)
	bool resident;                        // the general set is small enough for the one-workgroup LDS-resident solver (k_solve_resident)
	bool csr_ready;                       // body_off[] / adj[] hold the CSR adjacency (built on demand: ensure_csr in nh_solve.hip)
	const nh_BodyPair* bodies;
	// inputs kept for the one-body fast path, which rebuilds its rows in registers instead of storing them
	const nh_Contact* contact_data;
	nh_CachedContactImpulse* impulses;
	uint32_t* general_list;
	// one-body path: row build + warm start are deferred into the first nh_apply_impulses (or the next call that observes
	// momentum / impulses); `setup_pending` says they have not run yet
	bool setup_pending;
	bool general_lists;        // the adjacency lists of the general bodies have been filled (not when the per-block colouring of the blocked solver was expected to apply)
	bool states_kept;          // the first solver call stored its states: another nh_apply_impulses can continue (false after an applied NH_FLAG_SINGLE_APPLY call)
	nh_ContactImpulseData* imp;
	nh_BodyData bodies_at_setup;
	// nh_setup_contact_constraints returns WITHOUT a host round trip; what needs the device counters (general adjacency, culling,
	// the level-scheduled path) is finished by the first nh_apply_impulses / flush, behind the speculatively launched one-body solver
	bool finish_pending;
	bool has_late;                        // bodies classified after the round trip (classes *_LATE)
	struct {
		nh_ContactData contacts;
		uint32_t *cursor, *pred_a, *pred_b, *level, *slot_key, *level_hist, *level_cursor, *tmp, *tent;
	} cont;
	// spatially blocked solver (nh_blocks.h): the general contacts are laid out by (owning block, colour) and swept block by block in LDS
	struct {
		bool active, warm_pending, radial_only;
		bool local;                           // the blocks coloured their own contacts (k_blk_prepare_local): no world-wide colour, no general adjacency lists
		uint32_t L;                                   // colours
		uint32_t phase_count[8];
		bool chain; uint32_t chain_pass;              // chained sweeps usable for this setup (tables built, no residual contacts); passes launched so far
		uint32_t res_off[NH_BLK_MAX_COLOURS + 2];     // residual contacts (no block can own them) by colour: positions in the level-order arrays
	} blk;
};

// ---- arena (host-side bump allocation over device memory, reference nudge.cpp:990-1055) ---------------
static inline void* nh_arena_alloc(nh_Arena* arena, uintptr_t size, uintptr_t alignment, int* err) {
	uintptr_t data = (uintptr_t)arena->data;
	uintptr_t end = data + arena->size;
	uintptr_t mask = alignment - 1;
	data = (data + mask) & ~mask;
	if (data + size > end) { *err = NH_ERR_ARENA; return nullptr; }
	arena->data = (void*)(data + size);
	arena->size = end - (data + size);
	return (void*)data;
}

template<class T>
static inline T* nh_arena_array(nh_Arena* arena, uintptr_t count, int* err, uintptr_t alignment = 256) {
	return static_cast<T*>(nh_arena_alloc(arena, sizeof(T) * (count ? count : 1), alignment, err));
}

// ---- launch with optional per-kernel timing -----------------------------------------------------------
void nh_timer_begin(nh_context* ctx, const char* name);
void nh_timer_end(nh_context* ctx);
void nh_timer_collect(nh_context* ctx);

#define NH_LAUNCH(ctx, name, kernel, grid, block, ...)                                           \
	do {                                                                                         \
		if ((ctx)->timing) nh_timer_begin((ctx), name);                                          \
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->stream, __VA_ARGS__);      \
		if ((ctx)->timing) nh_timer_end((ctx));                                                  \
	} while (0)

// launch on the side stream (untimed; only between a fork and its join inside one entry point)
#define NH_LAUNCH_SIDE(ctx, kernel, grid, block, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->side, __VA_ARGS__)

#define NH_HIP_CHECK(ctx, expr)                                                                  \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) { (ctx)->last_hip_error = (int)e_; return NH_ERR_HIP; }            \
	} while (0)

// layout of ctx->deg (32-bit words, S = NH_DEG_STRIDE(bodies)): [0, S) contact degree per body (k_adj_simple writes it out of the counters below: the CSR
// scan reads it); [S, 2S) fill cursors of the CSR build (zero until then); [2S, 4S) one 64-bit counter per body, bumped ONCE per collider pair while the
// contacts are laid out: low word = contacts so far (the degree), high word = collider pairs | pairs with a dynamic partner
// << 16; [4S, 5S) first contact of the body's last pair (| NH_FIRST_IS_A)
#define NH_DEG_STRIDE(B) ((((size_t)(B)) + 3u) & ~(size_t)1u)
#define NH_DEG_WORDS(B) (6u * NH_DEG_STRIDE(B))       // ([5S, 6S): tag-order position of the record the first contact belongs to)

static inline uint32_t nh_grid_for(uint64_t n, uint32_t block = NH_BLOCK, uint32_t max_blocks = 4096) {
	uint64_t g = (n + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (uint32_t)g;
}

// ---- utilities implemented in nh_util.hip (all sizes come from DEVICE counters) ------------------------
// Stable LSD radix sorts, 8-bit digits, bits [begin_bit, end_bit).  Returns 0 if the result is in the *_a buffers, 1 if in *_b.
int nh_sort_u32_u32(nh_context* ctx, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                     const uint32_t* d_count, uint32_t* hist /* 256*NH_SORT_GRID+512 */, int begin_bit, int end_bit);
int nh_sort_u64_u32(nh_context* ctx, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                     const uint32_t* d_count, uint32_t* hist, int begin_bit, int end_bit);
int nh_sort_u64(nh_context* ctx, uint64_t* keys_a, uint64_t* keys_b,
                 const uint32_t* d_count, uint32_t* hist, int begin_bit, int end_bit);
// One-kernel-per-pass variant (see nh_util.hip).  `capacity` is a host-side upper bound of *d_count (sizes the launch);
// `expected` a guess of it (0 = none; sizes the grid, any value is correct); `scratch` needs nh_sort_scratch_words(capacity) words.
size_t nh_sort_scratch_words(uint32_t capacity);
int nh_onesweep_u32_u32(nh_context* ctx, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                         const uint32_t* d_count, uint32_t capacity, uint32_t expected, uint32_t* scratch, int begin_bit, int end_bit);
int nh_onesweep_u64_u32_two_fields(nh_context* ctx, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                                    const uint32_t* d_count, uint32_t capacity, uint32_t expected, uint32_t* scratch, int field_bits);
// Tag sort of st->records (key, value) pairs seeded by the previous step's result (see nh_util.hip).  Sorted data ends in keys_a / vals_a.
uint32_t nh_bucket_sort_max_buckets(const nh_context* ctx, uint32_t capacity);
void nh_bucket_sort_u64_u32(nh_context* ctx, const uint64_t* keys_a, uint64_t* keys_b, const uint32_t* vals_a, uint32_t* vals_b, uint32_t capacity, uint2* place, int field_bits,
                            uint64_t* keys_out, uint32_t* vals_out);
void nh_bucket_sort_seed(nh_context* ctx, const uint64_t* sorted_keys, uint32_t capacity);
// Exclusive scan of `in[0..n)` into `out` (may alias); n read from *d_count (+ `extra` elements);
// the grand total is also stored to *d_total if non-null.  `tmp` needs 2 * NH_SORT_GRID words.
// If `d_enable` is given and *d_enable == 0 the scan is skipped on the device (the total is reported as 0).
void nh_halo_pack_on(hipStream_t stream, const nh_BodyData* bodies, const uint32_t* indices, uint32_t count, void* out);          // nh_solve.hip
void nh_halo_update_on(hipStream_t stream, const nh_BodyData* bodies, uint32_t first_slot, uint32_t count, const void* in);
void nh_scan_u32(nh_context* ctx, const uint32_t* in, uint32_t* out, const uint32_t* d_count, uint32_t extra,
                 uint32_t* tmp, uint32_t* d_total, const uint32_t* d_enable = nullptr);
// the same for two arrays of equal length in one pair of launches
void nh_scan2_u32(nh_context* ctx, const uint32_t* in_a, uint32_t* out_a, uint32_t* d_total_a, const uint32_t* in_b, uint32_t* out_b, uint32_t* d_total_b,
                  const uint32_t* d_count, uint32_t extra, uint32_t* tmp);

// dst[0 .. *d_count * words_per_item) = src[...]
void nh_copy_back_u32(nh_context* ctx, const uint32_t* src, uint32_t* dst, const uint32_t* d_count, uint32_t words_per_item);

// ---- device helpers ------------------------------------------------------------------------------------
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned nh_lane() { return __lane_id(); }

// 16 bytes from a per-lane global address straight into LDS (lane-linear destination: consecutive lanes must name consecutive 16-byte pieces)
__device__ __forceinline__ void nh_glds16(const float4* src, float4* lds_dst) {
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ bool nh_aabb_overlap(const float4& amin, const float4& amax, const float4& bmin, const float4& bmax) {
	// strict overlap, nudge.cpp:3306-3308
	return (bmax.x > amin.x) && (amax.x > bmin.x) && (bmax.y > amin.y) && (amax.y > bmin.y) && (bmax.z > amin.z) && (amax.z > bmin.z);
}
__device__ __forceinline__ uint32_t nh_float_flip(float f) {
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float nh_float_unflip(uint32_t u) {
	u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
	return __uint_as_float(u);
}

// Wave-aggregated reservation: every active lane asks for `n` slots (may be 0); returns the lane's base.
__device__ __forceinline__ uint32_t nh_wave_reserve(uint32_t* counter, uint32_t n) {
	// inclusive prefix sum over the wave (all 64 lanes must call)
	uint32_t x = n;
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t y = __shfl_up(x, d);
		if ((int)nh_lane() >= d) x += y;
	}
	uint32_t total = __shfl(x, 63);
	uint32_t base = 0;
	if (nh_lane() == 63 && total) base = atomicAdd(counter, total);
	base = __shfl(base, 63);
	return base + x - n;
}

// Same for a 0/1 request: one ballot + popcount instead of a shuffle scan.
__device__ __forceinline__ uint32_t nh_wave_reserve1(uint32_t* counter, bool flag) {
	unsigned long long bal = __ballot(flag);
	if (!bal) return 0;
	unsigned leader = (unsigned)__ffsll((long long)bal) - 1u;
	uint32_t base = 0;
	if (nh_lane() == leader) base = atomicAdd(counter, (uint32_t)__popcll(bal));
	base = __shfl(base, (int)leader);
	return base + (uint32_t)__popcll(bal & ((1ull << nh_lane()) - 1ull));
}
#endif

#endif
