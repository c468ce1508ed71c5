"""Experiment kept for the record (round 5, measured and NOT adopted): the still form of k_solve_one_body at THREE waves per SIMD.
    python tools/experiments/solver_three_waves.py <in nh_solve.hip> <out nh_solve.hip>
What it changes: (1) a row's second tangent v = u x n is recomputed at every use (nh_build_row1's expression on the same operands: the same bits) instead of living in
12 registers; (2) the body's transform waits in the (idle) LDS exchange slots across the sweeps instead of in 8 registers; (3) the still form keeps nothing but the
exchange slots in LDS -- 12,288 bytes per one-wave workgroup, which is what lets TWELVE of them share a CU (tools/valu_rate.hip: 13,312 bytes do not fit twelve
times): the first-slot / solve-order tables travel by shuffle, the delta list is read with uniform loads; (4) amdgpu_waves_per_eu(3): 168 VGPRs, 96 bytes of scratch.
Result on an MI355X (profiles/r05_solver_three_waves_ab.log): solve_still 153 us (two waves, as shipped) -> 158 us with (1)-(3) at two waves -> 170 us at three waves.
Bit-identical in every variant (tests/test_gpu_still.py, tests/test_gpu_parity.py).
Applies to nh_solve.hip as of commit d5fd6d4 (before the solver learnt to do the next step's transforms): tools/build_variant.sh three "" d5fd6d4 with NH_VARIANT_PATCH set."""
import sys
s = open(sys.argv[1]).read()
def rep(old, new, count=1):
    global s
    assert old in s, old[:80]
    s = s.replace(old, new, count)

rep("""struct nh_StillView { const uint32_t* body_rec;""",
"""__device__ __forceinline__ nh_row1 nh_row1_with_v(const nh_row1& r) {
	nh_row1 f = r;
	f.v_x = r.u_y * r.n_z - r.u_z * r.n_y; f.v_y = r.u_z * r.n_x - r.u_x * r.n_z; f.v_z = r.u_x * r.n_y - r.u_y * r.n_x;
	return f;
}
#define NH_OPAQUE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
template<int NW, int M, bool STILL> struct nh_ob_shared { nh_ob_slot slot[NW][64 * M]; uint32_t cid[NW][64 * M]; uint32_t hint[NW][64], first[NW][64]; };
template<int NW, int M> struct nh_ob_shared<NW, M, true> { nh_ob_slot slot[NW][64 * M]; };
struct nh_StillView { const uint32_t* body_rec;""")
rep("""__attribute__((amdgpu_waves_per_eu(MAXC == 4 ? NH_SOLVER_WPE : 1, MAXC == 4 ? NH_SOLVER_WPE : 1)))""",
    """__attribute__((amdgpu_waves_per_eu(MAXC == 4 ? (STILL ? NH_SOLVER_WPE : 2) : 1, MAXC == 4 ? (STILL ? NH_SOLVER_WPE : 2) : 1)))""")
rep("""	__shared__ nh_ob_slot s_slot[NW][64 * M];
	__shared__ uint32_t s_cid[NW][64 * M];
	__shared__ uint32_t s_hint[NW][64], s_first[NW][64];
	__shared__ int2 s_delta[STILL ? NH_DELTA_MAX : 1];                   // STILL: this step's changes of contact counts (tag-order position, difference)
	uint32_t n_delta = 0;
	if (STILL) {""",
"""	__shared__ nh_ob_shared<NW, M, STILL> s_tab;
	uint32_t n_delta = 0;
	const int2* const delta_list = sv.delta + sv.parity * NH_DELTA_MAX;
	if (STILL) {""")
rep("""		for (uint32_t k = threadIdx.x; k < n_delta; k += blockDim.x) s_delta[k] = sv.delta[sv.parity * NH_DELTA_MAX + k];
""", "")
rep("""			const_cast<nh_DevState*>(cv.st)->contacts = total;
		}
		__syncthreads();
	}""", """			const_cast<nh_DevState*>(cv.st)->contacts = total;
		}
	}""")
rep("""	nh_ob_slot* slot = s_slot[wave];
	uint32_t* scid = s_cid[wave];
""", """	nh_ob_slot* slot = s_tab.slot[wave];
	uint32_t* scid = nullptr;
	if constexpr (!STILL) scid = s_tab.cid[wave];
""")
rep("""		uint32_t still_base = 0, still_rec = 0;
		bool still_has = false;""", """		uint32_t still_base = 0, still_rec = 0, still_perm = 0;
		bool still_has = false;""")
rep("""			s_first[wave][lane] = still_has ? still_base : 0xFFFFFFFFu;          // (where the wave-cooperative loads find this body's four slots)
""", "")
rep("""			s_hint[wave][lane] = h; s_first[wave][lane] = first;
""", """			if constexpr (!STILL) { s_tab.hint[wave][lane] = h; s_tab.first[wave][lane] = first; }
""")
rep("""			if (STILL) { const uint32_t bj = s_first[wave][j]; sc[k] = bj != 0xFFFFFFFFu ? bj + (e - j * M) : 0xFFFFFFFFu; }""",
    """			if (STILL) { const uint32_t bj = (uint32_t)__shfl((int)(still_has ? still_base : 0xFFFFFFFFu), (int)j); sc[k] = bj != 0xFFFFFFFFu ? bj + (e - j * M) : 0xFFFFFFFFu; }""")
rep("""				const uint32_t hj = s_hint[wave][j];
				sguess[k] = hj != NH_HINT_NONE ? hj + (sc[k] - s_first[wave][j]) : sc[k];""",
    """				uint32_t hj = NH_HINT_NONE, fj = 0u;
				if constexpr (!STILL) { hj = s_tab.hint[wave][j]; fj = s_tab.first[wave][j]; }
				sguess[k] = hj != NH_HINT_NONE ? hj + (sc[k] - fj) : sc[k];""")
rep("""				for (uint32_t k = 0; k < n_delta; ++k) shift += (uint32_t)s_delta[k].x < bpos ? s_delta[k].y : 0;""",
    """				for (uint32_t k = 0; k < n_delta; ++k) { const int2 dl = delta_list[k]; shift += (uint32_t)dl.x < bpos ? dl.y : 0; }""")
rep("""				scid[lane * M + q] = cid[q];          // (where the results go: read by the wave after the barrier before the result stores)
			}""", """			}
			still_perm = sl[0] | (sl[1] << 2) | (sl[2] << 4) | (sl[3] << 6) | (cnt << 8);""")
rep("""		if (CONTIG && __builtin_amdgcn_ballot_w64(mine && (STILL ? (still_has && (br & NH_BODY_REC_IS_A) != 0) : ((sp.y >> 28) & 1u) != 0)) == 0) {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) nh_apply_one1(r[q], s[q], v, false);
			}
		} else {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) nh_apply_one1(r[q], s[q], v, is_a[q]);
			}
		}
		// NH_FLAG_FUSED_STEP: the advance (k_advance) on the way out, from the transform held since phase A (7 registers; a reload would be a third exposed round trip)
		float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
		uint32_t idle_in = 0;
		const bool advance_here = CONTIG && (fs.bits & 2u) && mine;
		if (advance_here) { a0 = t0; a1 = t1; idle_in = idle_early; }""",
"""		const bool advance_here = CONTIG && (fs.bits & 2u) && mine;
		if (M == 4) {
			__builtin_amdgcn_wave_barrier();
			slot[lane * M].c0 = t0; slot[lane * M].c1 = t1;
			asm volatile("" ::: "memory");
		}
		if (CONTIG && __builtin_amdgcn_ballot_w64(mine && (STILL ? (still_has && (br & NH_BODY_REC_IS_A) != 0) : ((sp.y >> 28) & 1u) != 0)) == 0) {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) {
						if (M == 4) { NH_OPAQUE3(r[q].u_x, r[q].u_y, r[q].u_z); nh_apply_one1(nh_row1_with_v(r[q]), s[q], v, false); }
						else nh_apply_one1(r[q], s[q], v, false);
					}
			}
		} else {
			for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
				for (int q = 0; q < M; ++q)
					if ((uint32_t)q < cnt) {
						if (M == 4) { NH_OPAQUE3(r[q].u_x, r[q].u_y, r[q].u_z); nh_apply_one1(nh_row1_with_v(r[q]), s[q], v, is_a[q]); }
						else nh_apply_one1(r[q], s[q], v, is_a[q]);
					}
			}
		}
		float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
		uint32_t idle_in = 0;
		if (M == 4) { asm volatile("" ::: "memory"); const float4 p0 = slot[lane * M].c0, p1 = slot[lane * M].c1; if (advance_here) { a0 = p0; a1 = p1; idle_in = idle_early; } }
		else if (advance_here) { a0 = t0; a1 = t1; idle_in = idle_early; }""")
rep("""				nh_export_impulse1(r[q], s[q], ox, oy, oz);""", """				if (M == 4) nh_export_impulse1(nh_row1_with_v(r[q]), s[q], ox, oy, oz); else
				nh_export_impulse1(r[q], s[q], ox, oy, oz);""")
rep("""			const uint32_t c = scid[e];
			if (c != 0xFFFFFFFFu) {""", """			uint32_t c;
			if constexpr (STILL) {
				const uint32_t j = e / M, q = e - j * M;
				const uint32_t bj = (uint32_t)__shfl((int)(still_has ? still_base : 0xFFFFFFFFu), (int)j), pj = (uint32_t)__shfl((int)still_perm, (int)j);
				c = (bj != 0xFFFFFFFFu && q < (pj >> 8)) ? bj + ((pj >> (2u * q)) & 3u) : 0xFFFFFFFFu;
			} else c = scid[e];
			if (c != 0xFFFFFFFFu) {""")
open(sys.argv[2], "w").write(s)
