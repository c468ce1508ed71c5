// tools/stream_rate.hip -- what HBM delivers on an MI355X when ONE kernel moves the same bytes through FEW or MANY arrays of 16-byte records (one record per lane and array,
// consecutive lanes on consecutive records: every wave instruction is one contiguous kilobyte).  The still solver reads ~25 arrays and writes ~15 per body; is the number of
// streams what holds it at ~3.5 TB/s?      hipcc -O3 --offload-arch=gfx950 tools/stream_rate.hip -o /tmp/stream_rate && /tmp/stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Ptrs { float4* p[48]; };

template<int NR, int NW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_move(Ptrs a, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float4 s = make_float4(0, 0, 0, 0);
	float4 v[NR];
#pragma unroll
	for (int k = 0; k < NR; ++k) v[k] = a.p[k][i];
#pragma unroll
	for (int k = 0; k < NR; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
#pragma unroll
	for (int k = 0; k < NW; ++k) a.p[NR + k][i] = make_float4(s.x + k, s.y, s.z, s.w);
}

// the same bytes through ONE array read and ONE written: records of NR (NW) x 16 bytes per lane
template<int NR, int NW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_move_aos(Ptrs a, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float4 s = make_float4(0, 0, 0, 0);
	float4 v[NR];
	const float4* src = a.p[0] + (size_t)i * NR;
#pragma unroll
	for (int k = 0; k < NR; ++k) v[k] = src[k];
#pragma unroll
	for (int k = 0; k < NR; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
	float4* dst = a.p[1] + (size_t)i * NW;
#pragma unroll
	for (int k = 0; k < NW; ++k) dst[k] = make_float4(s.x + k, s.y, s.z, s.w);
}

template<typename F> static float time_ms(F launch, int reps) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	launch(); hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < reps; ++r) launch();
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	return ms / reps;
}

template<int NR, int NW, int WAVES> static int run(uint32_t n, Ptrs soa, Ptrs aos) {
	const int grid = (n + 64 * WAVES - 1) / (64 * WAVES);
	const float t1 = time_ms([&] { hipLaunchKernelGGL((k_move<NR, NW, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, soa, n); }, 20);
	const float t2 = time_ms([&] { hipLaunchKernelGGL((k_move_aos<NR, NW, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, aos, n); }, 20);
	const double bytes = (double)n * 16.0 * (NR + NW);
	printf("%2d arrays read + %2d written, %d wave(s) per workgroup: %7.1f us = %5.2f TB/s   |  the same bytes as one array of records each way: %7.1f us = %5.2f TB/s\n",
	       NR, NW, WAVES, t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9);
	return 0;
}

int main() {
	const uint32_t n = 1u << 20;          // a million lanes, like the still solver of config 2
	Ptrs soa, aos;
	for (int k = 0; k < 48; ++k) CHECK(hipMalloc((void**)&soa.p[k], (size_t)n * 16));
	CHECK(hipMalloc((void**)&aos.p[0], (size_t)n * 16 * 32)); CHECK(hipMalloc((void**)&aos.p[1], (size_t)n * 16 * 32));
	for (int k = 0; k < 48; ++k) CHECK(hipMemset(soa.p[k], 0, (size_t)n * 16));
	CHECK(hipMemset(aos.p[0], 0, (size_t)n * 16 * 32));
	run<2, 2, 1>(n, soa, aos); run<4, 4, 1>(n, soa, aos); run<8, 8, 1>(n, soa, aos); run<16, 8, 1>(n, soa, aos); run<24, 16, 1>(n, soa, aos); run<32, 16, 1>(n, soa, aos);
	run<24, 16, 4>(n, soa, aos); run<8, 8, 4>(n, soa, aos);
	return 0;
}
