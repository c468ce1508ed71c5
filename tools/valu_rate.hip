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
	// LDS footprint against residency: three waves per SIMD = 12 workgroups of one wave per CU
	for (size_t lds : { (size_t)12288, (size_t)13312, (size_t)13568, (size_t)13824, (size_t)15616 }) {
		g_lds = lds;
		printf("dynamic LDS %zu bytes per one-wave workgroup, asked for 3 waves per SIMD: ", lds);
		run<1, false>(3, iters);
	}
	g_lds = 0;
	return 0;
}
