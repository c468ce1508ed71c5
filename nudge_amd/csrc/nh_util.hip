// nh_util.hip -- device-count driven building blocks: stable LSD radix sort (8-bit digits) and exclusive scan.
//
// The reference's three radix sorts (reference nudge.cpp:2647-2836) are all "histogram, prefix, stable
// scatter" with 8-bit digits; this is the same algorithm laid out for a GPU: NH_SORT_GRID workgroups each
// own one contiguous chunk of the input, so a pass is  histogram[digit][block] -> one scan -> stable scatter.
// Element counts live in device memory (no host round trip); kernels are launched with a fixed grid.
#include "nh_internal.h"

#define RS_G NH_SORT_GRID
#define RS_TILE 256

__device__ __forceinline__ void rs_chunk(uint32_t n, uint32_t& begin, uint32_t& end) {
	uint32_t tiles = (n + RS_TILE - 1) / RS_TILE;
	uint32_t tiles_per_block = (tiles + RS_G - 1) / RS_G;
	uint32_t chunk = tiles_per_block * RS_TILE;
	uint64_t b = (uint64_t)blockIdx.x * chunk;
	begin = b < n ? (uint32_t)b : n;
	uint64_t e = b + chunk;
	end = e < n ? (uint32_t)e : n;
}

template<typename K>
__global__ __launch_bounds__(256) void rs_hist(const K* __restrict__ keys, const uint32_t* __restrict__ d_n, uint32_t* __restrict__ hist, int shift) {
	__shared__ uint32_t h[256];
	uint32_t n = *d_n;
	uint32_t begin, end;
	rs_chunk(n, begin, end);
	h[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t i = begin + threadIdx.x; i < end; i += 256)
		atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
	__syncthreads();
	hist[threadIdx.x * RS_G + blockIdx.x] = h[threadIdx.x];
}

// per-digit exclusive scan over the RS_G block counters (one workgroup per digit) + the digit's total
__global__ __launch_bounds__(RS_G) void rs_rowscan(uint32_t* __restrict__ hist) {
	__shared__ uint32_t wsum[RS_G / 64];
	uint32_t* row = hist + blockIdx.x * RS_G;
	uint32_t v = row[threadIdx.x];
	uint32_t x = v;
	for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
	if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
	__syncthreads();
	uint32_t woff = 0;
	for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
	row[threadIdx.x] = woff + x - v;
	if (threadIdx.x == RS_G - 1) hist[256 * RS_G + blockIdx.x] = woff + x;     // digit total
}

template<typename K, typename V, bool HAS_V>
__global__ __launch_bounds__(256) void rs_scatter(const K* __restrict__ keys, K* __restrict__ keys_out,
                                                  const V* __restrict__ vals, V* __restrict__ vals_out,
                                                  const uint32_t* __restrict__ d_n, const uint32_t* __restrict__ hist, int shift) {
	__shared__ uint32_t offs[256];
	__shared__ uint32_t whist[4][256];
	uint32_t n = *d_n;
	uint32_t begin, end;
	rs_chunk(n, begin, end);
	{
		// digit base = exclusive scan of the 256 digit totals (every workgroup redoes this tiny scan in LDS)
		uint32_t v = hist[256 * RS_G + threadIdx.x];
		uint32_t x = v;
		for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
		if ((threadIdx.x & 63) == 63) whist[0][threadIdx.x >> 6] = x;
		__syncthreads();
		uint32_t woff = 0;
		for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) woff += whist[0][w];
		offs[threadIdx.x] = woff + x - v + hist[threadIdx.x * RS_G + blockIdx.x];
		__syncthreads();
	}
	for (int w = 0; w < 4; ++w) whist[w][threadIdx.x] = 0;
	__syncthreads();
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (uint32_t tile = begin; tile < end; tile += RS_TILE) {
		uint32_t i = tile + threadIdx.x;
		bool valid = i < end;
		K key = valid ? keys[i] : (K)0;
		V val = V();
		if (HAS_V && valid) val = vals[i];
		uint32_t d = (uint32_t)(key >> shift) & 255u;
		unsigned long long mask = __ballot(valid);
		for (int b = 0; b < 8; ++b) {
			unsigned long long bal = __ballot((d >> b) & 1u);
			mask &= ((d >> b) & 1u) ? bal : ~bal;
		}
		uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
		uint32_t cnt = __popcll(mask);
		if (valid && rank == 0) whist[wave][d] = cnt;
		__syncthreads();
		uint32_t pos = 0;
		if (valid) {
			uint32_t pre = 0;
			for (unsigned w = 0; w < wave; ++w) pre += whist[w][d];
			pos = offs[d] + pre + rank;
		}
		__syncthreads();
		{
			uint32_t t = threadIdx.x;
			uint32_t tot = whist[0][t] + whist[1][t] + whist[2][t] + whist[3][t];
			offs[t] += tot;
			whist[0][t] = 0; whist[1][t] = 0; whist[2][t] = 0; whist[3][t] = 0;
		}
		if (valid) {
			keys_out[pos] = key;
			if (HAS_V) vals_out[pos] = val;
		}
		__syncthreads();
	}
}

template<typename K, typename V, bool HAS_V>
static int sort_impl(nh_context* ctx, K* keys_a, K* keys_b, V* vals_a, V* vals_b, const uint32_t* d_count, uint32_t* hist, int begin_bit, int end_bit) {
	K* kin = keys_a; K* kout = keys_b;
	V* vin = vals_a; V* vout = vals_b;
	int passes = 0;
	for (int shift = begin_bit; shift < end_bit; shift += 8) {
		NH_LAUNCH(ctx, "radix_hist", (rs_hist<K>), RS_G, 256, kin, d_count, hist, shift);
		NH_LAUNCH(ctx, "radix_scan", rs_rowscan, 256, RS_G, hist);
		NH_LAUNCH(ctx, "radix_scatter", (rs_scatter<K, V, HAS_V>), RS_G, 256, kin, kout, vin, vout, d_count, hist, shift);
		K* tk = kin; kin = kout; kout = tk;
		V* tv = vin; vin = vout; vout = tv;
		++passes;
	}
	return passes & 1;      // 1: the sorted data sits in the *_b buffers
}

__global__ __launch_bounds__(256) void k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const uint32_t* __restrict__ d_n, uint32_t words_per_item) {
	uint64_t n = (uint64_t)(*d_n) * words_per_item;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

void nh_copy_back_u32(nh_context* ctx, const uint32_t* src, uint32_t* dst, const uint32_t* d_count, uint32_t words_per_item) {
	NH_LAUNCH(ctx, "copy_words", k_copy_words, 1024, 256, src, dst, d_count, words_per_item);
}

int nh_sort_u32_u32(nh_context* ctx, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                     const uint32_t* d_count, uint32_t* hist, int begin_bit, int end_bit) {
	return sort_impl<uint32_t, uint32_t, true>(ctx, keys_a, keys_b, vals_a, vals_b, d_count, hist, begin_bit, end_bit);
}

int nh_sort_u64_u32(nh_context* ctx, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                     const uint32_t* d_count, uint32_t* hist, int begin_bit, int end_bit) {
	return sort_impl<uint64_t, uint32_t, true>(ctx, keys_a, keys_b, vals_a, vals_b, d_count, hist, begin_bit, end_bit);
}

int nh_sort_u64(nh_context* ctx, uint64_t* keys_a, uint64_t* keys_b, const uint32_t* d_count, uint32_t* hist, int begin_bit, int end_bit) {
	return sort_impl<uint64_t, uint32_t, false>(ctx, keys_a, keys_b, (uint32_t*)nullptr, (uint32_t*)nullptr, d_count, hist, begin_bit, end_bit);
}

// ---- exclusive scan -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* wsum /*4*/, uint32_t& block_total) {
	uint32_t x = v;
	for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
	__syncthreads();
	if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
	__syncthreads();
	uint32_t woff = 0;
	for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
	block_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
	return woff + x - v;
}

__global__ __launch_bounds__(256) void sc_sums(const uint32_t* __restrict__ in, const uint32_t* __restrict__ d_n, uint32_t extra, uint32_t* __restrict__ tmp, const uint32_t* __restrict__ d_enable) {
	__shared__ uint32_t wsum[4];
	if (d_enable && !*d_enable) return;
	uint32_t n = *d_n + extra;
	uint32_t begin, end;
	rs_chunk(n, begin, end);
	uint32_t s = 0;
	for (uint32_t i = begin + threadIdx.x; i < end; i += 256) s += in[i];
	for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) tmp[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(RS_G) void sc_top(uint32_t* __restrict__ tmp, uint32_t* __restrict__ d_total, const uint32_t* __restrict__ d_enable) {
	__shared__ uint32_t wsum[RS_G / 64];
	if (d_enable && !*d_enable) { if (threadIdx.x == 0 && d_total) *d_total = 0; return; }
	uint32_t v = tmp[threadIdx.x];
	uint32_t x = v;
	for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)(threadIdx.x & 63) >= d) x += y; }
	if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
	__syncthreads();
	uint32_t woff = 0;
	for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
	tmp[threadIdx.x] = woff + x - v;
	if (threadIdx.x == RS_G - 1) {
		uint32_t total = woff + x;
		tmp[RS_G] = total;
		if (d_total) *d_total = total;
	}
}

__global__ __launch_bounds__(256) void sc_final(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ d_n, uint32_t extra, const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ d_enable) {
	__shared__ uint32_t wsum[4];
	if (d_enable && !*d_enable) return;
	uint32_t n = *d_n + extra;
	uint32_t begin, end;
	rs_chunk(n, begin, end);
	uint32_t carry = tmp[blockIdx.x];
	for (uint32_t tile = begin; tile < end; tile += RS_TILE) {
		uint32_t i = tile + threadIdx.x;
		uint32_t v = i < end ? in[i] : 0u;
		uint32_t total;
		uint32_t ex = block_exclusive_scan_256(v, wsum, total);
		if (i < end) out[i] = carry + ex;
		carry += total;
		__syncthreads();
	}
}

void nh_scan_u32(nh_context* ctx, const uint32_t* in, uint32_t* out, const uint32_t* d_count, uint32_t extra, uint32_t* tmp, uint32_t* d_total, const uint32_t* d_enable) {
	NH_LAUNCH(ctx, "scan_sums", sc_sums, RS_G, 256, in, d_count, extra, tmp, d_enable);
	NH_LAUNCH(ctx, "scan_top", sc_top, 1, RS_G, tmp, d_total, d_enable);
	NH_LAUNCH(ctx, "scan_final", sc_final, RS_G, 256, in, out, d_count, extra, tmp, d_enable);
}
