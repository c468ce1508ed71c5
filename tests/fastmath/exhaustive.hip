// tests/fastmath/exhaustive.hip -- exhaustive (all 2^32 bit patterns) comparison of the short reciprocal and reciprocal-square-
// root sequences of nh_math.h (nh_recip_fast / nh_rsqrt_fast and the zero-preserving nh_recip_fast_z / nh_rsqrt_fast_z) against the correctly rounded forms the kernels are specified with
// (nh_recip = 1.0f / x, nh_rsqrt = 1.0f / sqrtf(x)).
//
// Kernels use the short sequence and fall back to the exact form whenever any lane's short result fails a one-instruction
// predicate (nh_*_fast_ok).  That is sound iff   short(x) != exact(x)  implies  !ok(short(x))   for every x: this program counts
// the violations of that implication over all 2^32 inputs (must be 0) and, for information, how many inputs take the fallback.
//   make -C nudge_amd/csrc fastmath && tests/fastmath/exhaustive        (~1 s on an MI355X)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../nudge_amd/csrc/nh_math.h"

__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

// counters, three per form: [0] differs from the exact form, [1] differs AND the predicate holds (a violation), [2] predicate fails
#define NFORMS 4
__global__ void k_check(unsigned long long* count, uint32_t* example, uint32_t base) {
#if defined(__HIP_DEVICE_COMPILE__)       // (the short forms exist in device code only)
	uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
	float x = __uint_as_float(bits);
	auto tally = [&](int k, float fast, float exact, bool ok) {
		bool differs = !same(fast, exact);
		if (differs) atomicAdd(&count[3 * k], 1ull);
		if (differs && ok) { if (atomicAdd(&count[3 * k + 1], 1ull) == 0) example[k] = bits; }
		if (!ok) atomicAdd(&count[3 * k + 2], 1ull);
	};
	{ float f = nh_recip_fast(x); tally(0, f, 1.0f / x, nh_recip_fast_ok(f)); }
	{ float f = nh_rsqrt_fast(x); tally(1, f, 1.0f / sqrtf(x), nh_rsqrt_fast_ok(f)); }
	{ float f = nh_recip_fast_z(x); tally(2, f, 1.0f / x, !nh_recip_fast_z_bad(f)); }
	{ float f = nh_rsqrt_fast_z(x); tally(3, f, 1.0f / sqrtf(x), !(nh_rsqrt_fast_z_bad(f) || (nh_rsqrt_fast_z_small(x) && x != 0.0f))); }
#endif
}

int main() {
	unsigned long long* count; uint32_t* example;
	if (hipMalloc(&count, 8 * 3 * NFORMS) != hipSuccess || hipMalloc(&example, 4 * NFORMS) != hipSuccess) { printf("no device\n"); return 2; }
	if (hipMemset(count, 0, 8 * 3 * NFORMS) != hipSuccess || hipMemset(example, 0, 4 * NFORMS) != hipSuccess) return 2;
	for (uint32_t part = 0; part < 16; ++part) {                          // 16 launches of 2^20 blocks of 256
		hipLaunchKernelGGL(k_check, dim3(1u << 20), dim3(256), 0, 0, count, example, part << 28);
		if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 2; }
	}
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
	unsigned long long h[3 * NFORMS]; uint32_t he[NFORMS];
	if (hipMemcpy(h, count, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(he, example, sizeof(he), hipMemcpyDeviceToHost) != hipSuccess) return 2;
	const char* names[NFORMS] = { "recip", "rsqrt", "recip_z", "rsqrt_z" };
	bool good = true;
	for (int k = 0; k < NFORMS; ++k) {
		printf("%-8s differs %llu  violations %llu (e.g. 0x%08x)  fallback %llu\n", names[k], h[3 * k], h[3 * k + 1], he[k], h[3 * k + 2]);
		good = good && h[3 * k + 1] == 0;
	}
	printf("%s\n", good ? "EXHAUSTIVE OK" : "EXHAUSTIVE FAILED");
	return good ? 0 : 1;
}
