"""EXPERIMENT (round 5, not adopted): the still solver software-pipelined -- 2048 persistent waves, two per SIMD, each sweeping group k of 64 bodies while the loads of
group k + 1 are in flight (per-contact records by LDS-direct loads into the slots group k's rows were built from, per-body words into registers the sweeps do not use,
round 1 of group k + 2 behind them), results stored lane by lane.  Same arithmetic line by line: tests/test_gpu_still.py passes bit for bit.

Measured on an MI355X (c2, 1,004,400 boxes, xform ahead on): 177.3 us per launch against 171.3 us for k_solve_one_body<4, true, true, 1, true, true> -- the kernel is not
waiting for memory LATENCY (the VALU pipe is ~1.0 busy while the waves compute, and what the kernel moves per microsecond is what the other two
kernels of the step move), so hiding it buys nothing and the persistent grid's 7.66 -> 8 rounds cost 4 %.

    python tools/experiments/solver_pipelined.py <in nh_solve.hip> <out nh_solve.hip>
or through tools/build_variant.sh:
    NH_VARIANT_PATCH=tools/experiments/solver_pipelined.py tools/build_variant.sh pipe ""
    NUDGE_HIP_LIBRARY=$PWD/nudge_amd/_ab/libpipe.so python bench.py"""
import sys
s = open(sys.argv[1]).read()
KERNEL = r'''// ---- the still solver, software-pipelined (round 5) ---------------------------------------------------------------------------------------------------------------
// k_solve_one_body<4, true, true, 1, true> spends a third of its time with BOTH waves of a SIMD in their memory phases (two dependent round trips at the head of a wave,
// the stores at its tail: DESIGN 7.3) -- the sweeps in between are VALU-throughput bound, so more waves per SIMD do not help (measured: three waves, each 1.55 x
// slower).  This form keeps the arithmetic, line by line, and changes who waits for memory: 2048 PERSISTENT waves (two per SIMD), each taking every 2048th group of 64
// bodies, and while a wave sweeps group k the loads of group k + 1 are in flight --
//   * round 1 (record, tag-order position [, idle counter, collider]) of group k + 2 into three or four registers;
//   * round 2 of group k + 1: the per-contact records (contact, cached impulse: 12 KB per wave) by LDS-direct loads into the slots group k's rows were just built
//     from -- no staging registers --, the per-body words (counts, feature words, body state: 31 registers) into registers the sweeps do not use (the kernel is
//     compiled for two waves per SIMD: 256 registers a lane, the sweeps live in ~190);
//   * AHEAD: the collider's 80 bytes of group k, LDS-direct, for the tail of group k.
// Results go out lane by lane (16 bytes per contact at the slot the contact came from: no staging through LDS, whose slots by then hold the next group).
struct nh_PipeR1 { uint32_t br, bpos, idle0, col; };
struct nh_PipeR2 { uint32_t cnt, start, oc, nf[4], of[4]; float4 m0, m1, pr, t0, t1; uint32_t idle_early, br, bpos, base, rec, col; bool has, mine; };

template<bool AHEAD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_solve_still_pipe(uint32_t nbodies, const nh_BodyProperties* __restrict__ props, nh_BodyMomentum* __restrict__ momentum,
                                                        uint32_t iterations, const nh_Contact* __restrict__ contacts, nh_Transform* xf, nh_CacheView cv, nh_FusedStep fs, nh_StillView sv, nh_AheadView av) {
	constexpr int M = 4;
	if (nbodies == 0u) return;
	{
		// (every workgroup reads the same words and comes to the same verdict: k_solve_one_body<.., STILL>)
		const nh_BodyMomentum m0 = momentum[0];
		const nh_BodyProperties p0 = props[0];
		const bool inert = nh_is_inert(m0.velocity, m0.angular_velocity, p0.inertia_inverse, p0.mass_inverse);
		if (!inert && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&const_cast<nh_DevState*>(cv.st)->still_failed_seq, fs.seq);
		if (!inert || cv.st->still_failed_seq >= fs.guard_seq) return;
	}
	__shared__ float4 s_c0[64 * M], s_c1[64 * M], s_w[64 * M];          // slot e = 4 * (lane of the body) + its q-th raw slot: contact (two halves), cached impulse
	__shared__ float4 s_ah[AHEAD ? 5 * 64 : 1];
	__shared__ uint32_t s_first[64];
	__shared__ int2 s_delta[NH_DELTA_MAX];
	const uint32_t lane = threadIdx.x;
	uint32_t n_delta = cv.st->delta_overflow[sv.parity] ? 0u : min(cv.st->delta_count[sv.parity], NH_DELTA_MAX);
	for (uint32_t k = lane; k < n_delta; k += 64u) s_delta[k] = sv.delta[sv.parity * NH_DELTA_MAX + k];
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		// the contact count of the step = the total behind the last record, kept like the starts; the counters of a step in sleepers form (k_solve_one_body)
		const uint32_t nrec = cv.st->records;
		int shift = 0;
		for (uint32_t k = 0; k < n_delta; ++k) shift += sv.delta[sv.parity * NH_DELTA_MAX + k].y;
		const uint32_t total = sv.start_sorted[nrec] + (uint32_t)shift;
		if (shift) sv.start_sorted[nrec] = total;
		nh_DevState* const stw = const_cast<nh_DevState*>(cv.st);
		stw->contacts = total;
		stw->active = (nbodies - 1u) - min(stw->still_asleep[sv.parity], nbodies - 1u); stw->sleeping = stw->still_sleeping[sv.parity]; stw->culled = stw->still_culled[sv.parity];
	}
	__builtin_amdgcn_wave_barrier();
	const uint32_t groups = (nbodies + 63u) / 64u, G = gridDim.x;
	if (blockIdx.x >= groups) return;
	const uint32_t trips = (groups - blockIdx.x + G - 1u) / G;
	const uint32_t n_bb = min(cv.st->pairs, sv.pair_cap);
	nh_CachedContactImpulse* const out_impulses = const_cast<nh_CachedContactImpulse*>(cv.cdata);
	const bool sleepers = (fs.bits & 4u) != 0u;
	auto x_of = [&](uint32_t k) -> uint32_t { return (blockIdx.x + k * G) * 64u + lane; };
	// round 1 of a group: the body's record and tag-order position (+ its idle counter where somebody may be asleep, + its collider)
	auto round1 = [&](uint32_t k) -> nh_PipeR1 {
		nh_PipeR1 r = { NH_BODY_REC_NONE, 0u, 0u, 0xFFFFFFFFu };
		if (k >= trips) return r;
		const uint32_t xc = min(x_of(k), nbodies - 1u);
		r.br = sv.body_rec[xc]; r.bpos = sv.body_pos[xc];
		if (sleepers) r.idle0 = fs.idle[xc];
		if (AHEAD) r.col = av.body_col[xc];
		return r;
	};
	// round 2: everything else at once -- the record's count, start and cached count, the feature words of its four slots (this step's and the cached ones), the body's
	// state into registers; the four contacts and cached impulses of every body of the group, slot by slot, straight into LDS
	auto round2 = [&](uint32_t k, const nh_PipeR1& a) -> nh_PipeR2 {
		nh_PipeR2 r;
		const uint32_t x = x_of(k);
		r.mine = k < trips && x >= 1u && x < nbodies && a.idle0 != 0xffu;
		r.br = r.mine ? a.br : NH_BODY_REC_NONE; r.bpos = a.bpos; r.col = r.mine ? a.col : 0xFFFFFFFFu;
		r.has = r.br != NH_BODY_REC_NONE;
		r.rec = r.has ? (r.br & ~NH_BODY_REC_IS_A) : 0u;
		r.base = r.rec < n_bb ? 4u * r.rec : 4u * n_bb + (r.rec - n_bb);
		const uint32_t bp = r.has ? r.bpos : 0u;
		r.cnt = sv.cnt_sorted[bp]; r.start = sv.start_sorted[bp]; r.oc = sv.sc_count[r.rec];
#pragma unroll
		for (int q = 0; q < M; ++q) { r.nf[q] = sv.raw_feature[r.base + q]; r.of[q] = sv.sc_feat[r.base + q]; }
		s_first[lane] = r.has ? r.base : 0xFFFFFFFFu;
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int q = 0; q < M; ++q) {
			const uint32_t e = 64u * q + lane, j = e / M;
			const uint32_t bj = s_first[j];
			const uint32_t c = bj != 0xFFFFFFFFu ? bj + (e - j * M) : 0u;          // (a body without a record: slot 0 comes in and nobody looks at it)
			const float4* cp = reinterpret_cast<const float4*>(contacts + c);
			nh_glds16(cp, &s_c0[e]); nh_glds16(cp + 1, &s_c1[e]);
			nh_glds16(reinterpret_cast<const float4*>(cv.cdata + c), &s_w[e]);
		}
		const uint32_t xb = min(x, nbodies - 1u);
		r.m0 = reinterpret_cast<const float4*>(momentum + xb)[0]; r.m1 = reinterpret_cast<const float4*>(momentum + xb)[1];
		r.pr = *reinterpret_cast<const float4*>(props + xb);
		r.t0 = reinterpret_cast<const float4*>(xf + xb)[0]; r.t1 = reinterpret_cast<const float4*>(xf + xb)[1];
		r.idle_early = sleepers ? a.idle0 : fs.idle[xb];
		return r;
	};
	nh_PipeR1 a1 = round1(0u);
	nh_PipeR2 cur = round2(0u, a1);
	a1 = round1(1u);
	for (uint32_t k = 0; k < trips; ++k) {
		const uint32_t x = x_of(k);
		// this group's loads have landed (registers and LDS), and so has round 1 of the next group
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_wave_barrier();
		const bool mine = cur.mine, still_has = cur.has;
		const uint32_t br = cur.br, bpos = cur.bpos, still_base = cur.base, still_rec = cur.rec;
		float4 m0 = cur.m0, m1 = cur.m1;
		const float4 pr = cur.pr, t0 = cur.t0, t1 = cur.t1;
		if (mine) {          // NH_FLAG_FUSED_STEP: gravity / damping (k_gravity) on the way in
			float vel[3] = { m0.x, m0.y, m0.z }, ang[3] = { m1.x, m1.y, m1.z };
			nh_gravity_damping(vel, ang, fs.gx_dt, fs.gy_dt, fs.gz_dt, fs.damping);
			m0.x = vel[0]; m0.y = vel[1]; m0.z = vel[2]; m1.x = ang[0]; m1.y = ang[1]; m1.z = ang[2];
		}
		// which slot is solved when, and which cached impulse it starts from (k_solve_one_body<.., STILL>: the same lines)
		uint32_t cnt = 0, oc = 0, st_start = cur.start;
		uint32_t sl[M], jm[M], cid[M];
		if (still_has) {
			cnt = min(cur.cnt, 4u); oc = min(cur.oc, 4u);
			int shift = 0;
			for (uint32_t d = 0; d < n_delta; ++d) shift += (uint32_t)s_delta[d].x < bpos ? s_delta[d].y : 0;
			if (shift) { st_start += (uint32_t)shift; sv.start_sorted[bpos] = st_start; }
		}
		{
			uint32_t ci[M], key[M];
#pragma unroll
			for (int q = 0; q < M; ++q) {
				uint32_t r = 0;
#pragma unroll
				for (int j = 0; j < M; ++j) r += ((uint32_t)j < cnt && (uint32_t)q < cnt && (cur.nf[j] < cur.nf[q] || (cur.nf[j] == cur.nf[q] && j < q))) ? 1u : 0u;
				const bool on = (uint32_t)q < cnt;
				ci[q] = on ? st_start + r : 0xFFFFFFFFu;
				key[q] = on ? slot_key_of(ci[q], nullptr) : 0xFFFFFFFFu;
				sl[q] = (uint32_t)q;
			}
#pragma unroll
			for (int q = 1; q < M; ++q) {
#pragma unroll
				for (int j = q; j > 0; --j) {
					const bool sw2 = (key[j] < key[j - 1]) || (key[j] == key[j - 1] && ci[j] < ci[j - 1]);
					const uint32_t tc = sw2 ? ci[j - 1] : ci[j], tk = sw2 ? key[j - 1] : key[j], ts = sw2 ? sl[j - 1] : sl[j];
					ci[j - 1] = sw2 ? ci[j] : ci[j - 1]; key[j - 1] = sw2 ? key[j] : key[j - 1]; sl[j - 1] = sw2 ? sl[j] : sl[j - 1];
					ci[j] = tc; key[j] = tk; sl[j] = ts;
				}
			}
#pragma unroll
			for (int q = 0; q < M; ++q) {
				const bool on = (uint32_t)q < cnt;
				uint32_t f = 0, wj = 0xFFFFFFFFu;
#pragma unroll
				for (int j = 0; j < M; ++j) if (sl[q] == (uint32_t)j) f = cur.nf[j];
#pragma unroll
				for (int j = M - 1; j >= 0; --j) if ((uint32_t)j < oc && cur.of[j] == f) wj = (uint32_t)j;
				jm[q] = on ? wj : 0xFFFFFFFFu;
				cid[q] = on ? still_base + sl[q] : 0xFFFFFFFFu;
			}
			// this step's feature words and count become the slot cache's (the impulses follow at the end)
			if (still_has) {
#pragma unroll
				for (int q = 0; q < M; ++q) if ((uint32_t)q < cnt) sv.sc_feat[still_base + q] = cur.nf[q];
				sv.sc_count[still_rec] = cnt;
			}
		}
		nh_vel v = { m0.x, m0.y, m0.z, pr.w, m1.x, m1.y, m1.z, m1.w };
		const nh_quat qx = { t1.x, t1.y, t1.z, t1.w };
		const nh_inertia IX = nh_world_inertia(qx, pr.x, pr.y, pr.z);
		const nh_f3 posx = nh_make3(t0.x, t0.y, t0.z);
		nh_row1 r[M];
		nh_state s[M];
		const bool is_a = (br & NH_BODY_REC_IS_A) != 0 && still_has;
		// rows, and the warm start in slot order
#pragma unroll
		for (int q = 0; q < M; ++q) {
			if ((uint32_t)q < cnt) {
				const uint32_t e = lane * M + sl[q];
				const float4 c0 = s_c0[e], c1 = s_c1[e];
				nh_contact_in ci = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
				nh_build_row1(ci, posx, IX, v.mass_inverse, is_a, r[q]);
				const float4 wv = s_w[lane * M + (jm[q] != 0xFFFFFFFFu ? jm[q] : 0u)];
				const bool hit = jm[q] != 0xFFFFFFFFu;
				nh_warm_start1(r[q], hit ? wv.x : 0.0f, hit ? wv.y : 0.0f, hit ? wv.z : 0.0f, v, is_a, s[q]);
			}
		}
		const uint32_t idle_in = cur.idle_early, my_col = cur.col;
		// ---- the slots are free: the next group's round 2 goes out, the round 1 of the group after it, and (AHEAD) this group's collider data ----
		__builtin_amdgcn_wave_barrier();
		if (k + 1u < trips) cur = round2(k + 1u, a1);          // (block-uniform)
		a1 = round1(k + 2u);
		float ah_rad = 0.0f;
		if (AHEAD) {
			const uint32_t cc = my_col != 0xFFFFFFFFu ? my_col : 0u;          // (unconditional loads at a clamped index)
			const bool is_box = cc < av.nbox;
			const float4* lp = reinterpret_cast<const float4*>(is_box ? av.box_xf + cc : av.sph_xf + (cc - av.nbox));
			const float4* sp4 = is_box ? reinterpret_cast<const float4*>(av.box_data + cc) : lp;          // (a sphere's radius -- four bytes -- comes by register)
			if (!is_box) ah_rad = av.sph_data[cc - av.nbox].radius;
			nh_glds16(lp, &s_ah[lane]); nh_glds16(lp + 1, &s_ah[64u + lane]);
			nh_glds16(av.fat_box + 2u * (size_t)cc, &s_ah[128u + lane]); nh_glds16(av.fat_box + 2u * (size_t)cc + 1u, &s_ah[192u + lane]);
			nh_glds16(sp4, &s_ah[256u + lane]);
		}
		// (a body's role is one bit for all its contacts, and in most scenes the static world is always "a": when no lane of the wave has D = a the role selects drop out)
		if (__builtin_amdgcn_ballot_w64(mine && is_a) == 0) {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) nh_apply_one1(r[q], s[q], v, false);
			}
		} else {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) nh_apply_one1(r[q], s[q], v, is_a);
			}
		}
		float4 ah_l0 = make_float4(0, 0, 0, 0), ah_l1 = ah_l0, ah_fmn = ah_l0, ah_fmx = ah_l0, ah_sz = ah_l0;
		if (AHEAD) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (everything issued before the sweeps has had the sweeps to arrive)
			ah_l0 = s_ah[lane]; ah_l1 = s_ah[64u + lane]; ah_fmn = s_ah[128u + lane]; ah_fmx = s_ah[192u + lane]; ah_sz = s_ah[256u + lane];
		}
		// results: the exported impulse (update_cached_impulses, nudge.cpp:4872-4882) to the slot the contact came from
#pragma unroll
		for (int q = 0; q < M; ++q) {
			if ((uint32_t)q < cnt) {
				float ox, oy, oz;
				nh_export_impulse1(r[q], s[q], ox, oy, oz);
				*reinterpret_cast<float4*>(out_impulses + cid[q]) = make_float4(ox, oy, oz, 0.0f);
			}
		}
		uint32_t nmin[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, nmax[3] = { 0u, 0u, 0u }, ntop = 0u;
		bool nfail = false;
		if (mine) {
			reinterpret_cast<float4*>(momentum + x)[0] = make_float4(v.vx, v.vy, v.vz, v.mass_inverse);
			reinterpret_cast<float4*>(momentum + x)[1] = make_float4(v.wx, v.wy, v.wz, v.w_unused);
			// NH_FLAG_FUSED_STEP: the advance (k_advance) on the way out
			float pos[3] = { t0.x, t0.y, t0.z }, rot[4] = { t1.x, t1.y, t1.z, t1.w };
			const float vel[3] = { v.vx, v.vy, v.vz }, ang[3] = { v.wx, v.wy, v.wz };
			const uint8_t idle_out = nh_advance_body(pos, rot, vel, ang, (uint8_t)idle_in, fs.time_step);
			fs.idle[x] = idle_out;
			reinterpret_cast<float4*>(xf + x)[0] = make_float4(pos[0], pos[1], pos[2], t0.w);
			reinterpret_cast<float4*>(xf + x)[1] = make_float4(rot[0], rot[1], rot[2], rot[3]);
			if (AHEAD) {
				// the next step's k_xform<true> for this body's collider (k_solve_one_body<.., AHEAD>: the same lines)
				ntop = idle_out; nfail = idle_out == 0xffu;
				const uint32_t c = my_col;
				if (c != 0xFFFFFFFFu) {
					const bool is_box = c < av.nbox;
					const float4 l0 = ah_l0, l1 = ah_l1, fmn = ah_fmn, fmx = ah_fmx;
					float4 sz = ah_sz;
					if (!is_box) sz.x = ah_rad;
					const nh_quat bq = { rot[0], rot[1], rot[2], rot[3] };
					const nh_quat lq = { l1.x, l1.y, l1.z, l1.w };
					const nh_f3 p = nh_rotate(bq, nh_make3(l0.x, l0.y, l0.z)) + nh_make3(pos[0], pos[1], pos[2]);
					const nh_quat q = nh_qmul(bq, lq);
					float sx, sy, sz2;
					if (is_box) {
						const nh_m33 m = nh_matrix(q);
						const nh_f3 c0 = m.c0 * sz.x, c1 = m.c1 * sz.y, c2 = m.c2 * sz.z;
						sx = fabsf(c0.x) + fabsf(c1.x) + fabsf(c2.x);
						sy = fabsf(c0.y) + fabsf(c1.y) + fabsf(c2.y);
						sz2 = fabsf(c0.z) + fabsf(c1.z) + fabsf(c2.z);
					} else {
						sx = sy = sz2 = sz.x;
					}
					const float mnx = p.x - sx, mny = p.y - sy, mnz = p.z - sz2, mxx = p.x + sx, mxy = p.y + sy, mxz = p.z + sz2;
					float4* const wx = reinterpret_cast<float4*>(av.xf) + 2u * (size_t)c;
					wx[0] = make_float4(p.x, p.y, p.z, l0.w);
					wx[1] = make_float4(q.x, q.y, q.z, q.s);
					av.aabb_min[c] = make_float4(mnx, mny, mnz, l0.w);
					av.aabb_max[c] = make_float4(mxx, mxy, mxz, __uint_as_float(0u));
					nfail |= !(mnx >= fmn.x && mny >= fmn.y && mnz >= fmn.z && mxx <= fmx.x && mxy <= fmx.y && mxz <= fmx.z);       // (NaN: out)
					nmin[0] = nmax[0] = nh_float_flip(mnx); nmin[1] = nmax[1] = nh_float_flip(mny); nmin[2] = nmax[2] = nh_float_flip(mnz);
				}
			}
		}
		if (AHEAD) {
			nh_DevState* const stw = const_cast<nh_DevState*>(cv.st);
			const uint32_t np = sv.parity ^ 1u;
#pragma unroll
			for (int d3 = 0; d3 < 3; ++d3) {
				for (int d = 32; d >= 1; d >>= 1) { nmin[d3] = min(nmin[d3], (uint32_t)__shfl_xor((int)nmin[d3], d)); nmax[d3] = max(nmax[d3], (uint32_t)__shfl_xor((int)nmax[d3], d)); }
			}
			for (int d = 32; d >= 1; d >>= 1) ntop = max(ntop, (uint32_t)__shfl_xor((int)ntop, d));
			const unsigned long long bad = __ballot(nfail);
			uint32_t* const part = &stw->ahead_part[np][(blockIdx.x + k) & (NH_AHEAD_PARTS - 1u)][0];
			const uint32_t mine_min = lane == 0u ? nmin[0] : (lane == 1u ? nmin[1] : nmin[2]);
			const uint32_t mine_max = lane == 3u ? nmax[0] : (lane == 4u ? nmax[1] : (lane == 5u ? nmax[2] : ntop));
			if (lane < 3u) atomicMin(&part[lane], mine_min);
			else if (lane < 7u) atomicMax(&part[lane], mine_max);
			else if (lane == 7u && bad != 0ull) atomicMax(&stw->ahead_failed_seq, fs.seq + 1u);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (no LDS-direct load of this wave may land after the wave has gone)
}

'''
LAUNCH_OLD = r'''			if (ahead) {
				const nh_ColliderData& cd = ss.colliders;
				nh_AheadView av = { ctx->body_col, cd.boxes.transforms, cd.boxes.data, cd.spheres.transforms, cd.spheres.data, cd.boxes.count, ctx->own_xf, ctx->own_aabb_min, ctx->own_aabb_max, ctx->fat_box };
				NH_LAUNCH(ctx, "solve_still", (k_solve_one_body<4, true, true, 1, true, true>), (B + 63u) / 64u, 64, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
				          (float4*)d->states, iterations, ctx->raw_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, 2u, d->simple, fs, sv, av);
			} else
'''
LAUNCH_NEW = r'''			// (the software-pipelined form -- persistent waves, two per SIMD, the next group's loads in flight under this group's sweeps -- unless option "no_solver_pipe")
			const bool pipe = !ss.no_pipe && ctx->env_solver_waves != 4 && ctx->env_solver_waves != 2;
			const uint32_t pipe_grid = std::min((B + 63u) / 64u, 8u * (uint32_t)(ctx->cu_count > 0 ? ctx->cu_count : 256));
			if (ahead) {
				const nh_ColliderData& cd = ss.colliders;
				nh_AheadView av = { ctx->body_col, cd.boxes.transforms, cd.boxes.data, cd.spheres.transforms, cd.spheres.data, cd.boxes.count, ctx->own_xf, ctx->own_aabb_min, ctx->own_aabb_max, ctx->fat_box };
				if (pipe)
					NH_LAUNCH(ctx, "solve_still", (k_solve_still_pipe<true>), pipe_grid, 64, B, bodies->properties, bodies->momentum, iterations, ctx->raw_data, bodies->transforms, cv, fs, sv, av);
				else
				NH_LAUNCH(ctx, "solve_still", (k_solve_one_body<4, true, true, 1, true, true>), (B + 63u) / 64u, 64, B, d->body_class, d->body_off, d->adj, d->bodies, bodies->properties, bodies->momentum,
				          (float4*)d->states, iterations, ctx->raw_data, bodies->transforms, d->impulses, cv, ctx->hint, NH_CLS_STATIC4, NH_CLS_STATIC4, 2u, d->simple, fs, sv, av);
			} else if (pipe) {
				NH_LAUNCH(ctx, "solve_still", (k_solve_still_pipe<false>), pipe_grid, 64, B, bodies->properties, bodies->momentum, iterations, ctx->raw_data, bodies->transforms, cv, fs, sv, nh_AheadView());
			} else
'''
marker = "// ---- level-scheduled path ---"
assert marker in s and LAUNCH_OLD in s
s = s.replace(marker, KERNEL + marker, 1).replace(LAUNCH_OLD, LAUNCH_NEW, 1)
s = s.replace("!ss.no_pipe", "true")
open(sys.argv[2], "w").write(s)
