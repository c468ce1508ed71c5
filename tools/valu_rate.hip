// valu_rate.hip -- what a wave64 fp32 VALU instruction costs on this chip, per wave and per SIMD (measurement program; runs on the GPU).
// Decides the question round 4 left open for k_solve_one_body: SQ counts ~1 quad-cycle of a wave's time per VALU instruction; with two waves per
// SIMD each issuing VALU in 0.48 of its cycles, is the SIMD's VALU pipe saturated (an instruction holds it for the 4 cycles) or half idle (it holds it for 2, and a
// single wave simply cannot issue dependent instructions faster than one per 4)?
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Each kernel runs CHAINS independent dependent-FMA chains of N instructions per lane; WAVES waves per SIMD on every SIMD of the chip (256 CUs x 4).
// Printed: shader cycles (s_memtime) per instruction seen by ONE wave, and instructions per cycle per SIMD = waves x N x CHAINS / cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template<int CHAINS, bool PACKED>
__global__ __launch_bounds__(64) void k_chain(float* out, unsigned long long* cycles, int iters, float a, float b) {
	extern __shared__ float dyn_lds[];
	if (iters < 0) dyn_lds[threadIdx.x] = a;          // (never: keeps the allocation referenced)
	float x[CHAINS];
	typedef float v2 __attribute__((ext_vector_type(2)));
	v2 p[CHAINS];
#pragma unroll
	for (int c = 0; c < CHAINS; ++c) { x[c] = (float)threadIdx.x * 1e-3f + c; p[c] = v2{ x[c], x[c] + 1.0f }; }
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 32; ++u) {
#pragma unroll
			for (int c = 0; c < CHAINS; ++c) {
				if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(v2{ a, a }), "v"(v2{ b, b }));
				else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
			}
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float s = 0.0f;
#pragma unroll
	for (int c = 0; c < CHAINS; ++c) s += PACKED ? p[c].x + p[c].y : x[c];
	out[blockIdx.x * 64 + threadIdx.x] = s;
	if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

static size_t g_lds = 0;          // dynamic LDS per 64-thread workgroup (0: none): what LDS footprint still lets `waves_per_simd` waves share a SIMD?
// the same for other instructions of the solver's mix (static counts in the still solver: 3520 v_mul_f32, 1520 v_add_f32, 1248 v_fmac_f32, 908 v_fma_f32, 720 v_sub_f32,
// 860 v_cndmask_b32): OP 0 v_mul_f32 (VOP2), 1 v_add_f32 (VOP2), 2 v_fmac_f32 (VOP2, accumulates into its destination), 3 v_cndmask_b32 (VOP2 reading VCC), 4 v_fma_f32 with
// three DIFFERENT source registers besides the chain (VOP3), 5 an alternation mul / add / fmac / fma like the sweep's
template<int CHAINS, int OP>
__global__ __launch_bounds__(64) void k_mix(float* out, unsigned long long* cycles, int iters, float a, float b) {
	float x[CHAINS], y[CHAINS];
#pragma unroll
	for (int c = 0; c < CHAINS; ++c) { x[c] = (float)threadIdx.x * 1e-3f + c; y[c] = 1.0f + c; }
	float a2 = a * 1.0001f, b2 = b * 0.5f;
	unsigned long long smask = __ballot(a > b);
	asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 32; ++u) {
#pragma unroll
			for (int c = 0; c < CHAINS; ++c) {
				if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
				else if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
				else if (OP == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
				else if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a));
				else if (OP == 6) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a));
				else if (OP == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "s"(smask));
				else if (OP == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
				else if (OP == 9) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y[c]), "v"(a));
				else if (OP == 10) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x[c]), "v"(b) : "vcc"); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a) : "vcc"); }
				else if (OP == 11) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(x[c]) : "v"(a), "v"(b));
				else if (OP == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
				else if (OP == 13) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[c]));
				else if (OP == 14) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x[c]), "v"(b) : "vcc");
				else if (OP == 15) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(smask) : "v"(x[c]), "v"(b));
				else if (OP == 16) { asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(smask) : "v"(x[c]), "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "s"(smask)); }
				else if (OP == 17) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
				else if (OP == 18) asm volatile("v_mov_b32 %0, %1" : "=v"(x[c]) : "v"(y[c]));
				else if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"((u & 1) ? a : a2), "v"((u & 2) ? b : b2));
				else {
					if ((u & 3) == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
					else if ((u & 3) == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
					else if ((u & 3) == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(y[c]));
					else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a2), "v"(b2));
				}
			}
		}
	}
	float s = 0.0f;
#pragma unroll
	for (int c = 0; c < CHAINS; ++c) s += x[c];
	out[blockIdx.x * 64 + threadIdx.x] = s;
	if (threadIdx.x == 0) cycles[blockIdx.x] = 0;
}

template<int CHAINS, int OP>
static void run_mix(int waves_per_simd, int iters) {
	static const char* names[] = { "v_mul_f32   ", "v_add_f32   ", "v_fmac_f32  ", "v_cndmask   ", "v_fma 3 src ", "mul/add/fmac/fma", "v_cndmask_e32 vcc", "v_cndmask_e64 sgpr", "v_max_f32   ", "v_bfi_b32   ",
	                               "v_cmp + v_cndmask (2 instr)", "v_cndmask_e32, no dependence", "v_rcp_f32   ", "v_rsq_f32   ", "v_cmp -> vcc", "v_cmp_e64 -> sgpr", "v_cmp_e64 + v_cndmask_e64 via sgpr (2 instr)",
	                               "v_sub_f32   ", "v_mov_b32   " };
	const int simds = 256 * 4, blocks = simds * waves_per_simd;
	float* out; unsigned long long* cyc;
	hipMalloc((void**)&out, sizeof(float) * 64 * blocks); hipMalloc((void**)&cyc, sizeof(unsigned long long) * blocks);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k_mix<CHAINS, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, 16, 1.0000001f, 1e-7f);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((k_mix<CHAINS, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0000001f, 1e-7f);
	hipEventRecord(e1, 0);
	hipDeviceSynchronize();
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	const double n = (double)iters * 32.0 * CHAINS, us = ms * 1e3;
	printf("%s chains %d  waves/SIMD %d : %8.1f us  -> %7.1f wave-instr per us per SIMD\n", names[OP], CHAINS, waves_per_simd, us, n * waves_per_simd / us);
	hipFree(out); hipFree(cyc);
}

template<int CHAINS, bool PACKED>
static void run(int waves_per_simd, int iters) {
	const int simds = 256 * 4, blocks = simds * waves_per_simd;
	float* out; unsigned long long* cyc;
	hipMalloc((void**)&out, sizeof(float) * 64 * blocks); hipMalloc((void**)&cyc, sizeof(unsigned long long) * blocks);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k_chain<CHAINS, PACKED>), dim3(blocks), dim3(64), g_lds, 0, out, cyc, 16, 1.0000001f, 1e-7f);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((k_chain<CHAINS, PACKED>), dim3(blocks), dim3(64), g_lds, 0, out, cyc, iters, 1.0000001f, 1e-7f);
	hipEventRecord(e1, 0);
	hipDeviceSynchronize();
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(blocks);
	hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
	double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
	const double n = (double)iters * 32.0 * CHAINS;
	// (s_memtime runs at a constant 100 MHz on this family: cycles per instruction are taken from the event time and the instruction count, the counter only
	// shows that the waves did run side by side)
	const double us = ms * 1e3;
	const double inst_per_us_per_simd = n * waves_per_simd / us;
	printf("%s chains %d  waves/SIMD %d : %8.1f us  -> %7.1f wave-instr per us per SIMD  (at 2.4 GHz: %5.2f cycles per wave-instr per SIMD; one wave: %5.2f cycles per instr)   [timer ticks per wave %.0f]\n",
	       PACKED ? "v_pk_fma_f32" : "v_fma_f32   ", CHAINS, waves_per_simd, us, inst_per_us_per_simd, 2400.0 / inst_per_us_per_simd, 2400.0 * us / n, mean);
	hipFree(out); hipFree(cyc);
}

int main() {
	const int iters = 2048;
	for (int w : { 1, 2, 3, 4, 8 }) run<1, false>(w, iters);
	for (int w : { 1, 2, 4 }) run<2, false>(w, iters);
	for (int w : { 1, 2, 4 }) run<4, false>(w, iters);
	for (int w : { 1, 2, 4 }) run<8, false>(w, iters / 2);
	for (int w : { 1, 2, 4 }) run<1, true>(w, iters);
	for (int w : { 1, 2, 4 }) run<4, true>(w, iters);
	// other instructions of the solver's mix: issue limit (8 chains, 2 waves) and dependent rate (1 chain, 2 and 3 waves)
	run_mix<8, 0>(2, iters / 2); run_mix<8, 1>(2, iters / 2); run_mix<8, 2>(2, iters / 2); run_mix<8, 3>(2, iters / 8); run_mix<8, 4>(2, iters / 2); run_mix<8, 5>(2, iters / 2);
	run_mix<8, 6>(2, iters / 8); run_mix<8, 7>(2, iters / 8); run_mix<8, 8>(2, iters / 2); run_mix<8, 9>(2, iters / 2); run_mix<8, 10>(2, iters / 8); run_mix<8, 11>(2, iters / 8); run_mix<1, 6>(2, iters / 8); run_mix<1, 11>(1, iters / 8);
	run_mix<8, 12>(2, iters / 4); run_mix<8, 13>(2, iters / 4); run_mix<8, 14>(2, iters / 4); run_mix<8, 15>(2, iters / 4); run_mix<8, 16>(2, iters / 4); run_mix<8, 17>(2, iters / 2); run_mix<8, 18>(2, iters / 2);
	run_mix<1, 12>(2, iters / 4); run_mix<1, 16>(2, iters / 4);
	run_mix<1, 0>(2, iters); run_mix<1, 2>(2, iters); run_mix<1, 5>(2, iters); run_mix<1, 5>(3, iters); run_mix<1, 5>(4, iters);
	run_mix<2, 5>(2, iters); run_mix<4, 5>(2, iters);
	// LDS footprint against residency: three waves per SIMD = 12 workgroups of one wave per CU
	for (size_t lds : { (size_t)12288, (size_t)13312, (size_t)13568, (size_t)13824, (size_t)15616 }) {
		g_lds = lds;
		printf("dynamic LDS %zu bytes per one-wave workgroup, asked for 3 waves per SIMD: ", lds);
		run<1, false>(3, iters);
	}
	g_lds = 0;
	return 0;
}
