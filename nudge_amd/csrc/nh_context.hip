// nh_context.hip -- context lifecycle, device counters -> host, per-kernel timing, error strings.
#include "nh_internal.h"
#include <stdlib.h>
#include <string.h>

extern "C" int nh_create(nh_context** out, int device, void* stream, uint32_t flags) {
	if (!out) return NH_ERR_INVALID;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return NH_ERR_NO_DEVICE;
	if (hipSetDevice(device) != hipSuccess) return NH_ERR_NO_DEVICE;
	nh_context* ctx = new nh_context();
	ctx->device = device;
	ctx->stream = (hipStream_t)stream;
	ctx->asleep.streak = 0;
	ctx->flags = (flags & NH_FLAG_FUSED_STEP) ? (flags | NH_FLAG_SINGLE_APPLY) : flags;
	ctx->after_collide = false; ctx->grav.pending = false; ctx->grav.rest_pending = false; ctx->adv.done = false; ctx->adv.rest = true;
	ctx->tag_bits = 32;
	ctx->last_hip_error = 0;
	ctx->timing = false;
	ctx->timing_current = false;
	ctx->ring_pos = 0;
	ctx->body_count = 0;
	ctx->pending = nullptr;
	ctx->hint = nullptr;
	ctx->hint_capacity = 0;
	ctx->deg = nullptr;
	ctx->deg_capacity = 0;
	ctx->order_seq = 0;
	ctx->sort_splitters = nullptr; ctx->sort_counts = nullptr; ctx->sort_starts = nullptr; ctx->sort_capacity = 0; ctx->sort_seeded = false;
	ctx->step_parity = 0;
	ctx->pair_capacity = 0;
	ctx->collide_seq = 0; ctx->setup_seq = 0;
	ctx->idle_bound = -1; ctx->idle_unknown = true; ctx->advance_count = 0; ctx->collide_mark = 0; ctx->idle_bound_mark = 0; ctx->islands_skipped = false;
	// Behaviour switches (A/B runs, tests): defaults here, changed through nh_set_option -- the library itself never reads the environment
	ctx->sort_tile = 0u; ctx->sort_target = 0u;
	ctx->env_sort_classic = false; ctx->env_sort_radix = false; ctx->env_colour_jp = false;
	ctx->env_solver_waves = 1; ctx->env_fp_sub = 0;
	ctx->env_no_sort_reuse = false;
	ctx->sort_keys_by_position = nullptr; ctx->sort_sorted_keys = nullptr; ctx->sort_sorted_idx = nullptr;
	ctx->bp_seen_rebuilds = 0; ctx->bp_rebuild_streak = 0; ctx->bp_direct_left = 0;
	ctx->env_measure_skip = false;
	ctx->env_no_fat = false;
	ctx->fat_pairs = nullptr; ctx->fat_pair_capacity = 0; ctx->fat_box = nullptr; ctx->fat_collider_capacity = 0; ctx->fat_nbox = ctx->fat_nsph = 0xffffffffu;
	ctx->grid_sbox = nullptr; ctx->grid_skeys = nullptr; ctx->grid_cstart = nullptr; ctx->grid_counts = nullptr; ctx->grid_large = nullptr; ctx->grid_cells = 0; ctx->grid_collider_capacity = 0;
	ctx->fat_gen = nullptr; ctx->fat_esc_mark = nullptr; ctx->fat_esc_list = nullptr; ctx->fat_moved_list = nullptr;
	ctx->env_blk_rows_global = false; ctx->sync_exports_views = false;
	ctx->env_blk_global_colours = false; ctx->last_general_contacts = 0;
	ctx->env_no_incremental = false;
	ctx->env_colour_check_seeds = false;
	ctx->env_no_resident = false;
	ctx->os_resident = 0; ctx->os_plain_launch = true;
	memset(&ctx->blk, 0, sizeof(ctx->blk));
	ctx->blk.scale = 1.0f;
	ctx->blk.disabled = false; ctx->blk.check = false;
	ctx->blk.chain = true;
	ctx->blk.min_contacts = 200000u;
	ctx->blk.target = 900u;
	ctx->raw_data = nullptr; ctx->raw_feature = nullptr; ctx->raw_slots = 0; ctx->rec = nullptr; ctx->lay_rank = nullptr; ctx->cnt_sorted = nullptr; ctx->start_sorted = nullptr; ctx->dense_slot = nullptr;
	ctx->sc_imp = nullptr; ctx->sc_feat = nullptr; ctx->sc_count = nullptr; ctx->sc_undo = nullptr; ctx->pair_mark = nullptr; ctx->pair_list = nullptr; ctx->pair_list_capacity = 0; ctx->exp_cnt = nullptr; ctx->exp_start = nullptr; ctx->exp_scan_tmp = nullptr; ctx->still_delta = nullptr; ctx->lay_capacity = 0; ctx->lay_contact_capacity = 0;
	ctx->own_xf = nullptr; ctx->own_aabb_min = nullptr; ctx->own_aabb_max = nullptr; ctx->own_ctag = nullptr; ctx->own_capacity = 0; ctx->body_col = nullptr; ctx->body_col_capacity = 0;
	ctx->still_awake = nullptr; ctx->still_awake_capacity = 0; ctx->exp_sleep_a = nullptr; ctx->exp_sleep_b = nullptr; ctx->exp_sleep_hist = nullptr; ctx->exp_flags = nullptr; ctx->exp_sleep_capacity = 0; ctx->exp_flags_capacity = 0;
	ctx->lay_class = nullptr; ctx->lay_simple = nullptr; ctx->body_rec = nullptr; ctx->body_pos = nullptr; ctx->lay_body_capacity = 0;
	memset(&ctx->still, 0, sizeof(ctx->still));
	memset(&ctx->fused, 0, sizeof(ctx->fused));
	ctx->step_hook = nullptr; ctx->step_hook_user = nullptr; ctx->halo_ghost_first = 0u;
	ctx->first_ghost = 0;
	memset(&ctx->asleep, 0, sizeof(ctx->asleep));
	memset(&ctx->stream_state, 0, sizeof(ctx->stream_state)); ctx->stream_state.last_slot = -1;
	{ int cus = 0; ctx->cu_count = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0) ? cus : 256; }
	ctx->d_state = nullptr;
	ctx->h_state = nullptr;
	if (hipMalloc((void**)&ctx->d_state, sizeof(nh_DevState)) != hipSuccess) { delete ctx; return NH_ERR_HIP; }
	if (hipHostMalloc((void**)&ctx->h_state, sizeof(nh_DevState), 0) != hipSuccess) { hipFree(ctx->d_state); delete ctx; return NH_ERR_HIP; }
	memset(ctx->h_state, 0, sizeof(nh_DevState));
	ctx->h_early = nullptr; ctx->no_early_counts = false; ctx->early_reads = 0; ctx->early_fallbacks = 0;
	if (hipHostMalloc((void**)&ctx->h_early, sizeof(uint32_t) * (NH_COUNTER_WORDS + 16u), 0) != hipSuccess) { ctx->h_early = nullptr; (void)hipGetLastError(); }
	else memset(ctx->h_early, 0, sizeof(uint32_t) * (NH_COUNTER_WORDS + 16u));
	ctx->side = nullptr; ctx->ev_fork = nullptr; ctx->ev_join = nullptr; ctx->ev_active = nullptr; ctx->ev_gravity = nullptr; ctx->gravity_may_overlap = false;
	if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_active, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_gravity, hipEventDisableTiming) != hipSuccess) { ctx->side = nullptr; }       // no side stream: everything stays on the caller's
	hipMemsetAsync(ctx->d_state, 0, sizeof(nh_DevState), ctx->stream);
	hipStreamSynchronize(ctx->stream);
	*out = ctx;
	return NH_OK;
}

extern "C" void nh_destroy(nh_context* ctx) {
	if (!ctx) return;
	hipSetDevice(ctx->device);
	hipStreamSynchronize(ctx->stream);
	for (auto& s : ctx->timer_slots) { ctx->event_pool.push_back(s.a); ctx->event_pool.push_back(s.b); }
	for (auto e : ctx->event_pool) hipEventDestroy(e);
	for (auto p : ctx->impulse_ring) delete p;
	for (auto p : ctx->constraint_ring) delete p;
	if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
	if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
	if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
	if (ctx->ev_active) hipEventDestroy(ctx->ev_active);
	if (ctx->ev_gravity) hipEventDestroy(ctx->ev_gravity);
	{
		nh_StateStream& ss = ctx->stream_state;
		if (ss.stage) hipFree(ss.stage);
		if (ss.staged) hipEventDestroy(ss.staged);
		for (int k = 0; k < NH_STREAM_MAX_SLOTS; ++k) if (ss.landed[k]) hipEventDestroy(ss.landed[k]);
	}
	if (ctx->asleep.aabb_min) hipFree(ctx->asleep.aabb_min);
	if (ctx->asleep.aabb_max) hipFree(ctx->asleep.aabb_max);
	if (ctx->asleep.tags) hipFree(ctx->asleep.tags);
	if (ctx->hint) hipFree(ctx->hint);
	if (ctx->deg) hipFree(ctx->deg);
	if (ctx->fat_pairs) hipFree(ctx->fat_pairs);
	if (ctx->fat_box) hipFree(ctx->fat_box);
	if (ctx->grid_sbox) hipFree(ctx->grid_sbox);
	if (ctx->grid_skeys) hipFree(ctx->grid_skeys);
	if (ctx->grid_cstart) hipFree(ctx->grid_cstart);
	if (ctx->grid_counts) hipFree(ctx->grid_counts);
	if (ctx->grid_large) hipFree(ctx->grid_large);
	if (ctx->fat_gen) hipFree(ctx->fat_gen);
	if (ctx->fat_esc_mark) hipFree(ctx->fat_esc_mark);
	if (ctx->fat_esc_list) hipFree(ctx->fat_esc_list);
	if (ctx->fat_moved_list) hipFree(ctx->fat_moved_list);
	if (ctx->sort_keys_by_position) hipFree(ctx->sort_keys_by_position);
	if (ctx->sort_sorted_keys) hipFree(ctx->sort_sorted_keys);
	if (ctx->sort_sorted_idx) hipFree(ctx->sort_sorted_idx);
	if (ctx->sort_splitters) hipFree(ctx->sort_splitters);
	if (ctx->sort_counts) hipFree(ctx->sort_counts);
	if (ctx->sort_starts) hipFree(ctx->sort_starts);
	for (int k = 0; k < 2; ++k) { if (ctx->still.h_ring[k]) hipHostFree(ctx->still.h_ring[k]); if (ctx->still.ev_ring[k]) hipEventDestroy(ctx->still.ev_ring[k]); }
	nh_fused_free(ctx);
	{
		void* bufs[] = { ctx->raw_data, ctx->raw_feature, ctx->rec, ctx->lay_rank, ctx->cnt_sorted, ctx->start_sorted, ctx->dense_slot, ctx->sc_imp, ctx->sc_feat, ctx->sc_count, ctx->sc_undo, ctx->pair_mark, ctx->pair_list, ctx->exp_cnt, ctx->exp_start,
		                 ctx->exp_scan_tmp, ctx->still_delta, ctx->lay_class, ctx->lay_simple, ctx->body_rec, ctx->body_pos, ctx->still_awake, ctx->exp_sleep_a, ctx->exp_sleep_b, ctx->exp_sleep_hist, ctx->exp_flags,
		                 ctx->own_xf, ctx->own_aabb_min, ctx->own_aabb_max, ctx->own_ctag, ctx->body_col };
		for (void* b : bufs) if (b) hipFree(b);
	}
	{
		nh_BlkBuffers& k = ctx->blk;
		void* bufs[] = { k.brank, k.bm, k.pos_body, k.touch, k.lpair, k.ghost_list, k.blk_count, k.own_base, k.bins, k.bin_off, k.ghost_base, k.ghost_count, k.phase_list, k.desc, k.scan_tmp, k.prof, k.chain_slot, k.chain_nbr, k.chain_done, k.chain_ticket };
		for (void* b : bufs) if (b) hipFree(b);
	}
	if (ctx->d_state) hipFree(ctx->d_state);
	if (ctx->h_state) hipHostFree(ctx->h_state);
	if (ctx->h_early) hipHostFree(ctx->h_early);
	delete ctx;
}

extern "C" int nh_set_flags(nh_context* ctx, uint32_t flags) {
	if (!ctx) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	ctx->asleep.streak = 0;
	ctx->flags = (flags & NH_FLAG_FUSED_STEP) ? (flags | NH_FLAG_SINGLE_APPLY) : flags;
	return NH_OK;
}
// Diagnostic switches (include/nudge_hip.h: nh_set_option).  None is needed in production: each selects an older, slower or more talkative path for A/B runs and
// tests.  Call right after nh_create (a switch changed between two steps takes effect at the next nh_collide; buffers already sized are kept).
extern "C" int nh_set_first_ghost_body(nh_context* ctx, uint32_t first_ghost) {
	if (!ctx) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	ctx->asleep.streak = 0;
	ctx->first_ghost = first_ghost;          // (still steps are not launched while the rule is on: nh_collide)
	return NH_OK;
}

extern "C" int nh_set_option(nh_context* ctx, const char* name, int value) {
	if (!ctx || !name) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx); if (rc) return rc; }
	const std::string n(name);
	const bool on = value != 0;
	ctx->asleep.streak = 0;
	if (n == "no_asleep") ctx->asleep.disabled = on;                      // nh_step runs every step of a world that is asleep in full (nh_internal.h: nh_AsleepState)
	else if (n == "sync_exports_views") ctx->sync_exports_views = on;          // legacy observers: nh_synchronize / nh_read_counts end with nh_export_views(NH_VIEW_ALL) like they did before round 5 (note 9)
	else if (n == "fused_fail_in") ctx->fused.debug_fail_in = (uint32_t)value;   // TEST HOOK: the value-th fused launch from now raises the failure flag at its very end (rollback exercised on purpose)
	else if (n == "fused_still") ctx->fused.enabled = on;                      // plain still steps inside nh_step as ONE out-of-place kernel with a look-back (nh_fused.hip): bit-exact, measured SLOWER, off by default
	else if (n == "no_early_counts") ctx->no_early_counts = on;                // a full step's counters by a copy behind its solver, not by the solver's first thread (nh_internal.h: early counters; A/B, tests)
	else if (n == "no_sleeper_ahead") ctx->still.no_sleeper_ahead = on;        // a still step in sleepers form always launches its three kernels (nh_internal.h: sleepers ahead; A/B, tests)
	else if (n == "no_sleeper_skip") ctx->still.no_sleeper_skip = on;          // the sleepers form of a still step does all its work for sleeping bodies too (nh_internal.h; A/B, tests)
	else if (n == "halo_overlap") ctx->halo_overlap = on;                      // nh_partition_step: the next sub-step's halo beside the interior's solver, on a second stream (nh_internal.h: halo split; measured: no gain on one GPU)
	else if (n == "no_pair_ahead") ctx->still.no_pair = on;                    // every still step launches its narrowphase (nh_internal.h: pair ahead; A/B, tests)
	else if (n == "no_xform_ahead") ctx->still.no_ahead = on;                  // every still step launches k_xform<true> itself (nh_internal.h: xform ahead)
	else if (n == "no_local_still") ctx->still.no_local = on;                  // a collider that leaves its inflated box fails a still step as a whole (round 4's behaviour; nh_internal.h "LOCAL speculation")
	else if (n == "no_still") ctx->still.disabled = on;                        // no step is launched speculatively as a still step (note 9)
	else if (n == "sort_classic") ctx->env_sort_classic = on;             // three-kernel radix passes
	else if (n == "sort_radix") ctx->env_sort_radix = on;                 // radix passes every step (no seeded bucket sort)
	else if (n == "no_sort_reuse") ctx->env_no_sort_reuse = on;           // the tag sort of the contact records runs every step even when nothing has changed
	else if (n == "bucket_tile") ctx->sort_tile = (uint32_t)value;
	else if (n == "bucket_target") ctx->sort_target = (uint32_t)value;
	else if (n == "colour_jp") ctx->env_colour_jp = on;                   // Jones-Plassmann colouring rounds
	else if (n == "colour_check_seeds") ctx->env_colour_check_seeds = on; // colours seeded from the cache go through a settle round instead of being final at once
	else if (n == "solver_waves") ctx->env_solver_waves = value;          // wavefronts per workgroup of the fused solver (1, 2, 4)
	else if (n == "fp_sub") ctx->env_fp_sub = value;                      // lanes per collider in the pair search (1, 2, 4, 16; 0 = by world size)
	else if (n == "no_kept_pairs") ctx->env_no_fat = on;                  // the broadphase regroups and searches every step instead of re-using its kept pair list
	else if (n == "no_incremental") ctx->env_no_incremental = on;         // a collider that leaves its inflated box forces a rebuild of the kept pair list
	else if (n == "no_resident") ctx->env_no_resident = on;               // small general sets also take the one-launch-per-level path
	else if (n == "cooperative") ctx->os_plain_launch = !on;              // the one-kernel radix passes go out as cooperative launches (nh_util.hip)
	else if (n == "no_blocks") ctx->blk.disabled = on;                    // large general sets take the per-colour global launches instead of the blocked solver
	else if (n == "blk_check") ctx->blk.check = on;                       // verify the block tables on the device every step
	else if (n == "blk_min") ctx->blk.min_contacts = (uint32_t)value;     // general contacts from which the blocked form is used
	else if (n == "blk_target") ctx->blk.target = value < 8 ? 8u : (uint32_t)value;      // bodies per average block
	else if (n == "blk_rows_global") ctx->env_blk_rows_global = on;       // rows of block-owned contacts by k_rows_general
	else if (n == "blk_global_colours") ctx->env_blk_global_colours = on; // the blocked solver takes the world-wide colouring
	else if (n == "no_blk_chain") ctx->blk.chain = !on;                   // blocked solver: eight launches per pass (one per block colour) instead of one launch with neighbour-to-neighbour waits
	else if (n == "blk_profile") {                                         // wall-clock readings of the sweep workgroups, printed now and then
		if (on && !ctx->blk.prof) NH_HIP_CHECK(ctx, hipMalloc((void**)&ctx->blk.prof, sizeof(unsigned long long) * 4u * 8u * NH_BLK_MAX_BLOCKS));
	}
#if defined(NH_MEASURE)
	else if (n == "measure_skip_chains") ctx->env_measure_skip = on;      // MEASUREMENT BUILDS ONLY (-DNH_MEASURE): wrong as soon as anything moves
#endif
	else return NH_ERR_INVALID;
	return NH_OK;
}

extern "C" int nh_set_tag_bits(nh_context* ctx, uint32_t bits) { if (!ctx || bits < 1 || bits > 32) return NH_ERR_INVALID; ctx->tag_bits = bits; return NH_OK; }

extern "C" int nh_bodies_changed(nh_context* ctx) {
	if (!ctx) return NH_ERR_INVALID;
	{ int rc = nh_still_abandon(ctx); if (rc) return rc; }
	ctx->idle_bound = -1; ctx->idle_unknown = true;
	ctx->asleep.streak = 0;
	ctx->still.ok_next = false;                  // body records rewritten from outside: the next step looks at everything again
	ctx->still.sleep_backoff = 0u; ctx->still.sleep_stable = 0u;          // (sleepers ahead: whoever sleeps now, it is another set)
	return NH_OK;
}

extern "C" int nh_set_pair_capacity(nh_context* ctx, uint32_t pairs) { if (!ctx) return NH_ERR_INVALID; ctx->pair_capacity = pairs; return NH_OK; }

extern "C" int nh_synchronize(nh_context* ctx) {
	if (!ctx) return NH_ERR_INVALID;
	int rc = nh_flush_pending(ctx);
	if (rc) return rc;
	// (waits only: the views of note 9 are exported by nh_export_views -- a caller that synchronises every step must not pay two passes over the contacts for it;
	// option "sync_exports_views" restores the older contract for hosts written against it)
	if (ctx->sync_exports_views) { rc = nh_still_sync_outputs(ctx, NH_VIEW_ALL); if (rc) return rc; }
	NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->timing) nh_timer_collect(ctx);
	return NH_OK;
}

// the mirrored counter block -> what callers see (and the host's sleep prediction)
void nh_counts_from_mirror(nh_context* ctx, nh_Counts* out) {
	const nh_DevState* h = ctx->h_state;
	// what the last nh_collide saw -- unless the counters have been rewritten since (nh_bodies_changed, nh_halo_unpack): then it describes another state
	if (ctx->collide_seq && !ctx->idle_unknown) { ctx->idle_bound = (int)h->max_idle[ctx->step_parity]; ctx->idle_bound_mark = ctx->collide_mark; }
	out->colliders = h->colliders; out->pairs = h->pairs + h->pairs_sph; out->contacts = h->contacts; out->sleeping_pairs = h->sleeping;
	out->active_bodies = h->active; out->cache = h->cache; out->culled = h->culled; out->large_colliders = h->large;
	// (still steps, note 9: the cache IS the contact list of the last step -- nothing is culled in a still step -- and its count is written to the device only by the
	// export of the view: until then the step's contact count stands for it)
	if (ctx->still.cache_stale) out->cache = h->contacts + h->culled;          // (+ the entries kept aside for sleeping pairs: a step in sleepers form)
	out->general_contacts = h->general_contacts; out->levels = h->levels; out->error = h->error;
	out->still_steps = (uint32_t)(ctx->still.launched - ctx->still.failed - ((ctx->still.active && !ctx->still.resolved) ? 1u : 0u)); out->still_replays = (uint32_t)ctx->still.failed;
	for (int k = 0; k < 4; ++k) out->still_diff[k] = h->still_diff[k];
	out->asleep_steps = (uint32_t)ctx->asleep.steps; out->ahead_steps = (uint32_t)ctx->still.ahead_steps; out->fused_steps = (uint32_t)ctx->fused.steps; out->pair_steps = (uint32_t)ctx->still.pair_steps;
	for (int k = 0; k < 4; ++k) out->pair_diag[k] = h->pair_diag[k];
	out->blk_blocks = h->blk.nblk; out->blk_bodies = h->blk.bodies; out->blk_ghosts = h->blk.ghost_cursor;
	out->has_other_bodies = h->has_other; out->broadphase_rebuilds = h->fat_rebuilds; out->sort_reuses = h->sort_reuses; out->broadphase_inserts = h->fat_inserts; out->unleveled = h->unleveled; out->raw_pairs = h->pairs + h->pairs_sph;
}

extern "C" int nh_read_counts(nh_context* ctx, nh_Counts* out) {
	if (!ctx || !out) return NH_ERR_INVALID;
	{ int rc = nh_flush_pending(ctx, true); if (rc) return rc; }      // (counters only: a pending NH_FLAG_FUSED_STEP gravity stays pending; no view is exported, note 9)
	if (ctx->sync_exports_views) { int rc = nh_still_sync_outputs(ctx, NH_VIEW_ALL); if (rc) return rc; }
	NH_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_state, ctx->d_state, NH_COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
	nh_counts_from_mirror(ctx, out);
	if (ctx->timing) nh_timer_collect(ctx);
	return NH_OK;
}

__global__ void k_set_cache_count(nh_DevState* st, uint32_t n) { st->cache = n; }

extern "C" int nh_set_cache_count(nh_context* ctx, uint32_t count) {
	if (!ctx) return NH_ERR_INVALID;
	{ int rc = nh_still_abandon(ctx); if (rc) return rc; }
	ctx->asleep.streak = 0;
	ctx->still.cache_ok = false; ctx->still.slots_current = false; ctx->still.cache_stale = false;      // the caller has put another cache in place: what the slots hold is void
	hipLaunchKernelGGL(k_set_cache_count, dim3(1), dim3(1), 0, ctx->stream, ctx->d_state, count);
	return NH_OK;
}

// ---- state streaming (include/nudge_hip.h: nh_stream_state) ---------------------------------------------------------------------------------------------------
extern "C" int nh_stream_state(nh_context* ctx, const nh_BodyData* bodies, uint32_t count, void* host_ring, uint32_t slots, uint32_t every) {
	if (!ctx) return NH_ERR_INVALID;
	NH_HIP_CHECK(ctx, hipSetDevice(ctx->device));
	nh_StateStream& ss = ctx->stream_state;
	// (frames still on their way belong to the old ring: wait for them before the ring changes hands)
	if (ctx->side) NH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->side));
	for (int k = 0; k < NH_STREAM_MAX_SLOTS; ++k) { ss.in_flight[k] = false; ss.slot_frame[k] = 0; ss.slot_step[k] = 0; }
	ss.every = 0; ss.last_slot = -1; ss.advances = 0; ss.frames = 0; ss.dropped = 0;
	if (every == 0u) return NH_OK;
	if (!bodies || !bodies->transforms || !host_ring || count == 0u || count > bodies->count || slots == 0u || slots > NH_STREAM_MAX_SLOTS || !ctx->side) return NH_ERR_INVALID;
	if (ss.stage_capacity < count) {
		if (ss.stage) NH_HIP_CHECK(ctx, hipFree(ss.stage));
		ss.stage = nullptr; ss.stage_capacity = 0;
		NH_HIP_CHECK(ctx, hipMalloc((void**)&ss.stage, sizeof(nh_Transform) * (size_t)count));
		ss.stage_capacity = count;
	}
	if (!ss.staged) NH_HIP_CHECK(ctx, hipEventCreateWithFlags(&ss.staged, hipEventDisableTiming));
	for (uint32_t k = 0; k < slots; ++k) if (!ss.landed[k]) NH_HIP_CHECK(ctx, hipEventCreateWithFlags(&ss.landed[k], hipEventDisableTiming));
	ss.src = bodies->transforms; ss.count = count; ss.ring = static_cast<char*>(host_ring); ss.slots = slots; ss.every = every;
	return NH_OK;
}

int nh_stream_after_advance(nh_context* ctx) {
	nh_StateStream& ss = ctx->stream_state;
	if (!ss.every) return NH_OK;
	ss.advances++;
	if (ss.advances % ss.every) return NH_OK;
	const uint32_t slot = (uint32_t)(ss.frames % ss.slots);
	// one staging buffer: the frame before this one must have left it (and a slot is rewritten only when its last frame has landed) -- otherwise this frame is dropped
	const int prev = ss.last_slot;
	if ((prev >= 0 && ss.in_flight[prev] && hipEventQuery(ss.landed[prev]) != hipSuccess) || (ss.in_flight[slot] && hipEventQuery(ss.landed[slot]) != hipSuccess)) { ss.dropped++; return NH_OK; }
	const size_t bytes = sizeof(nh_Transform) * (size_t)ss.count;
	NH_HIP_CHECK(ctx, hipMemcpyAsync(ss.stage, ss.src, bytes, hipMemcpyDeviceToDevice, ctx->stream));          // (behind the step's last kernel on the world's stream)
	NH_HIP_CHECK(ctx, hipEventRecord(ss.staged, ctx->stream));
	NH_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side, ss.staged, 0));
	NH_HIP_CHECK(ctx, hipMemcpyAsync(ss.ring + (size_t)slot * bytes, ss.stage, bytes, hipMemcpyDeviceToHost, ctx->side));
	NH_HIP_CHECK(ctx, hipEventRecord(ss.landed[slot], ctx->side));
	// (the next DEVICE copy into the staging buffer must not overtake this frame's way out: the world's stream waits for `landed` only if it gets there first --
	// which the in-flight test above has just ruled out for the frames that follow)
	ss.in_flight[slot] = true; ss.slot_step[slot] = ss.advances; ss.last_slot = (int)slot; ss.frames++; ss.slot_frame[slot] = ss.frames;
	return NH_OK;
}

// The last nh_advance counted did not happen (a still step that failed: nh_solve.hip, still_forget_failed).  Its solver left the world as it was, so a frame taken behind
// it shows the state BEFORE the step under the number the replayed step will give its own frame: the frame is withdrawn -- nh_stream_latest no longer names it -- while
// the slot stays marked in flight until its copy has landed (it must not be rewritten under the copy).
void nh_stream_void_advance(nh_context* ctx) {
	nh_StateStream& ss = ctx->stream_state;
	if (!ss.every || !ss.advances) return;
	for (uint32_t k = 0; k < ss.slots; ++k)
		if (ss.slot_frame[k] != 0 && ss.slot_step[k] == ss.advances) { ss.slot_frame[k] = 0; ss.slot_step[k] = 0; }
	ss.advances--;
}

extern "C" int nh_stream_latest(nh_context* ctx, nh_StreamInfo* out) {
	if (!ctx || !out) return NH_ERR_INVALID;
	nh_StateStream& ss = ctx->stream_state;
	out->slot = 0; out->valid = 0; out->step = 0; out->frames = ss.frames; out->dropped = ss.dropped;
	uint64_t best = 0;          // (frames are numbered from 1 in the order they were taken: the newest that has landed)
	for (uint32_t k = 0; k < ss.slots; ++k) {
		if (ss.slot_frame[k] == 0) continue;
		if (ss.in_flight[k]) { if (hipEventQuery(ss.landed[k]) != hipSuccess) continue; ss.in_flight[k] = false; }
		if (ss.slot_frame[k] > best) { best = ss.slot_frame[k]; out->slot = k; out->valid = 1; out->step = ss.slot_step[k]; }
	}
	return NH_OK;
}

extern "C" const char* nh_error_string(int code) {
	switch (code) {
	case NH_OK: return "ok";
	case NH_ERR_INVALID: return "invalid argument";
	case NH_ERR_NO_DEVICE: return "no HIP device";
	case NH_ERR_ARENA: return "arena exhausted";
	case NH_ERR_CONTACT_CAPACITY: return "contact capacity exceeded";
	case NH_ERR_CACHE_CAPACITY: return "contact cache capacity exceeded";
	case NH_ERR_ACTIVE_CAPACITY: return "active body capacity exceeded";
	case NH_ERR_PAIR_CAPACITY: return "broadphase pair capacity exceeded";
	case NH_ERR_HIP: return "HIP runtime error";
	case NH_ERR_SCHEDULER_CAPACITY: return "exact-order scheduler replay out of open batches";
	case NH_ERR_STALE_HINT: return "idle counters changed without nh_bodies_changed";
	case NH_ERR_STALE_SETUP: return "setup_contact_constraints does not match the last collide";
	default: return "unknown error";
	}
}

extern "C" int nh_last_hip_error(nh_context* ctx) { return ctx ? ctx->last_hip_error : 0; }

// ---- per-kernel timing with HIP events on the context's stream ----------------------------------------------
static hipEvent_t take_event(nh_context* ctx) {
	if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
	hipEvent_t e;
	hipEventCreate(&e);
	return e;
}

void nh_timer_begin(nh_context* ctx, const char* name) {
	ctx->timing_current = ctx->timing_filter.empty() || ctx->timing_filter == name;
	if (!ctx->timing_current) return;
	nh_TimerSlot s;
	s.name = name;
	s.a = take_event(ctx);
	s.b = take_event(ctx);
	hipEventRecord(s.a, ctx->stream);
	ctx->timer_slots.push_back(s);
}

void nh_timer_end(nh_context* ctx) {
	if (!ctx->timing_current) return;
	hipEventRecord(ctx->timer_slots.back().b, ctx->stream);
}

void nh_timer_collect(nh_context* ctx) {
	for (auto& s : ctx->timer_slots) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
			auto& acc = ctx->kernel_ms[s.name];
			acc.first += ms;
			acc.second += 1;
		}
		ctx->event_pool.push_back(s.a);
		ctx->event_pool.push_back(s.b);
	}
	ctx->timer_slots.clear();
}

extern "C" int nh_enable_timing(nh_context* ctx, int on) {
	if (!ctx) return NH_ERR_INVALID;
	hipStreamSynchronize(ctx->stream);
	if (ctx->timing) nh_timer_collect(ctx);
	ctx->timing = on != 0;
	return NH_OK;
}

extern "C" int nh_set_timing_filter(nh_context* ctx, const char* name) {
	if (!ctx) return NH_ERR_INVALID;
	ctx->timing_filter = name ? name : "";
	return NH_OK;
}

extern "C" int nh_kernel_times(nh_context* ctx, nh_KernelTime* out, int cap, int reset) {
	if (!ctx) return 0;
	hipStreamSynchronize(ctx->stream);
	nh_timer_collect(ctx);
	int n = 0;
	ctx->name_store.clear();
	ctx->name_store.reserve(ctx->kernel_ms.size());
	for (auto& kv : ctx->kernel_ms) {
		if (n >= cap) break;
		ctx->name_store.push_back(kv.first);
		out[n].name = ctx->name_store.back().c_str();
		out[n].ms = kv.second.first;
		out[n].launches = kv.second.second;
		out[n].reserved = 0;
		++n;
	}
	if (reset) ctx->kernel_ms.clear();
	return n;
}
