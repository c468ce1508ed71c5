// nh_util.hip -- device-count driven building blocks: stable LSD radix sort (8-bit digits) and exclusive scan.
//
// The reference's three radix sorts (reference nudge.cpp:2647-2836) are all "histogram, prefix, stable
// scatter" with 8-bit digits; this is the same algorithm laid out for a GPU: NH_SORT_GRID workgroups each
// own one contiguous chunk of the input, so a pass is  histogram[digit][block] -> one scan -> stable scatter.
// Element counts live in device memory (no host round trip); kernels are launched with a fixed grid.
#include "nh_internal.h"
#include <mutex>

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

// ---- stable LSD radix sort, 8-bit digits, ONE kernel per pass ------------------------------------------------------------
// For the ~1 M-key sorts of this engine the three-kernel pass above is bound by launch latency and by its serial tile loops,
// not by bandwidth.  Here a pass is one kernel whose workgroups are all RESIDENT at once (<= OS_MAX_GRID) and talk through a
// control block:
//   1. workgroup w owns the contiguous chunk of m = ceil(tiles / workgroups) tiles (OS_TILE keys each); it counts its digits --
//      for m = 1 (<= 2 M keys) by ranking its keys, which then stay in registers -- and PUBLISHES its 256 counts;
//   2. the last workgroup of every group of OS_GROUP publishes the group's sums once its group has published;
//   3. every workgroup reads the earlier counts of its own group and ALL group sums (independent loads, two visibility hops --
//      no chained look-back: with every workgroup resident a chain is hundreds of serial hops) and so knows, per digit, the total
//      (-> the digit's base, by a block scan) and how many equal digits precede its chunk;
//   4. it scatters (m = 1: from registers; m > 1: a second sweep over its chunk, tile by tile).
// No histogram kernel, no ticket counter (one ticket per workgroup on one address would serialise ~12 ns each), no key is read twice
// for m = 1.  Control block per pass (words): counts[OS_MAX_GRID][256] | group sums [OS_MAX_GRID / OS_GROUP][256], zeroed per sort.
#define OS_KPT 8
#define OS_TILE (256 * OS_KPT)
#define OS_GROUP 32
#define OS_MAX_GRID 1024                  // rows of the control block; the launch is further bounded by what is resident (onesweep_impl)
#define OS_MAX_PASSES 8
#define OS_READY 0x80000000u
#define OS_VALUE_MASK 0x7FFFFFFFu
#define OS_PASS_WORDS ((OS_MAX_GRID + OS_MAX_GRID / OS_GROUP + 2) * 256)      // ... | digit histogram of the whole input [256] | ticket counter (+ padding) [256]
#define OS_HIST_OFFSET ((OS_MAX_GRID / OS_GROUP) * 256)                      // (from the group sums)

// a pass descriptor: digit = ((key >> s0) & (2^b0 - 1)) | ((key >> s1) << b0) & 255 -- s0 | b0 << 8 | s1 << 16.  b0 = 8 is the plain
// byte at s0; b0 < 8 glues the top bits of one key field to the low bits of the next (nh_onesweep_u64_u32_two_fields)
#define OS_DESC(s0, b0, s1) ((s0) | ((b0) << 8) | ((s1) << 16))
template<typename K>
__device__ __forceinline__ uint32_t os_digit(K key, int desc) {
	const int s0 = desc & 255, b0 = (desc >> 8) & 255, s1 = (desc >> 16) & 255;
	return ((uint32_t)(key >> s0) & ((1u << b0) - 1u)) | (((uint32_t)(key >> s1) << b0) & 255u);
}

struct os_shifts { int s[OS_MAX_PASSES]; int n; };

size_t nh_sort_scratch_words(uint32_t capacity) { (void)capacity; return (size_t)OS_MAX_PASSES * OS_PASS_WORDS; }

// sum of `count` published values p[0], p[stride], ... : sixteen independent loads in flight, re-polling only what is not there yet
__device__ __forceinline__ uint32_t os_sum_published(const uint32_t* p, uint32_t count, size_t stride) {
	uint32_t sum = 0;
	for (uint32_t t0 = 0; t0 < count; t0 += 16) {
		uint32_t v[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) v[k] = t0 + k < count ? __hip_atomic_load(p + (size_t)(t0 + k) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : OS_READY;
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			while (!(v[k] & OS_READY)) { __builtin_amdgcn_s_sleep(1); v[k] = __hip_atomic_load(p + (size_t)(t0 + k) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
			sum += v[k] & OS_VALUE_MASK;
		}
	}
	return sum;
}

// ranks the (up to) OS_TILE keys of one tile held by the workgroup: a wave owns 512 consecutive keys (8 rounds of 64); rank within
// the round by ballots (match-any on the digit bits), plus the wave's running count of the digit in LDS.  On return wh_all[w][d] =
// keys of digit d in wave w's part of the tile, local[q] = rank of key q among the wave's keys of its digit.
template<typename K>
__device__ __forceinline__ void os_rank_tile(const K (&key)[OS_KPT], uint32_t tile_begin, uint32_t n, int shift, uint32_t (*whist)[256], uint32_t (&dig)[OS_KPT], uint32_t (&local)[OS_KPT]) {
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t* wh = whist[wave];
#pragma unroll
	for (int q = 0; q < OS_KPT; ++q) {
		uint32_t i = tile_begin + wave * (64u * OS_KPT) + q * 64u + lane;
		bool valid = i < n;
		uint32_t d = os_digit<K>(key[q], shift);
		dig[q] = d;
		unsigned long long mask = __ballot(valid);
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			unsigned long long bal = __ballot((d >> b) & 1u);
			mask &= ((d >> b) & 1u) ? bal : ~bal;
		}
		uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
		uint32_t before = valid ? wh[d] : 0u;                    // keys of this digit seen by this wave in earlier rounds
		__builtin_amdgcn_wave_barrier();
		if (valid && rank == 0) wh[d] = before + (uint32_t)__popcll(mask);
		__builtin_amdgcn_wave_barrier();
		local[q] = before + rank;
	}
}

// Digit histograms of the whole input for every pass of a sort, in one read of the keys (a pass permutes the keys, its histogram does not care).  With the totals
// known up front a pass only ever waits for workgroups that STARTED BEFORE it (its chunk index is a ticket drawn at start, below): no workgroup depends on one that may
// not be resident yet, so the launch needs no co-residency guarantee -- neither an occupancy bound nor a cooperative launch -- to be free of deadlock.
template<typename K>
__global__ __launch_bounds__(256) void os_hist(const K* __restrict__ keys, const uint32_t* __restrict__ d_n, uint32_t* __restrict__ scratch, os_shifts sh) {
	__shared__ uint32_t h[OS_MAX_PASSES][256];
	for (int p = 0; p < sh.n; ++p) h[p][threadIdx.x] = 0;
	__syncthreads();
	const uint32_t n = *d_n;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const K key = keys[i];
		for (int p = 0; p < sh.n; ++p) atomicAdd(&h[p][os_digit<K>(key, sh.s[p])], 1u);
	}
	__syncthreads();
	for (int p = 0; p < sh.n; ++p) {
		const uint32_t c = h[p][threadIdx.x];
		if (c) atomicAdd(scratch + (size_t)p * OS_PASS_WORDS + (size_t)OS_MAX_GRID * 256 + OS_HIST_OFFSET + threadIdx.x, c);
	}
}

template<typename K, typename V, bool HAS_V>
__global__ __launch_bounds__(256) void os_pass(const K* __restrict__ keys, K* __restrict__ keys_out, const V* __restrict__ vals, V* __restrict__ vals_out,
                                               const uint32_t* __restrict__ d_n, uint32_t* __restrict__ counts, uint32_t* __restrict__ group_sum, int shift) {
	__shared__ uint32_t offs[256];
	__shared__ uint32_t whist[4][256];
	__shared__ uint32_t chunk_count[256];
	__shared__ uint32_t wtot[4];
	const uint32_t n = *d_n;
	const uint32_t ntiles = (n + OS_TILE - 1) / OS_TILE;
	const uint32_t m = (ntiles + gridDim.x - 1) / gridDim.x;               // tiles per workgroup (1 up to 2 M keys)
	const uint32_t nwg = m ? (ntiles + m - 1) / m : 0;                       // workgroups that own a chunk
	// the chunk a workgroup owns is a TICKET drawn when it starts, not its block index: whoever holds a lower ticket is already running (or done), whatever order the
	// hardware dispatches blocks in
	__shared__ uint32_t s_ticket;
	if (threadIdx.x == 0) s_ticket = atomicAdd(group_sum + OS_HIST_OFFSET + 256, 1u);
	__syncthreads();
	const uint32_t w = s_ticket;
	if (w >= nwg) return;
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t first_tile = w * m, last_tile = min(first_tile + m, ntiles);
	K key[OS_KPT];
	V val[OS_KPT];
	uint32_t dig[OS_KPT], local[OS_KPT];
	whist[0][threadIdx.x] = 0; whist[1][threadIdx.x] = 0; whist[2][threadIdx.x] = 0; whist[3][threadIdx.x] = 0;
	chunk_count[threadIdx.x] = 0;
	__syncthreads();
	// 1. count the chunk's digits
	for (uint32_t t = first_tile; t < last_tile; ++t) {
		const uint32_t tile_begin = t * OS_TILE;
#pragma unroll
		for (int q = 0; q < OS_KPT; ++q) {
			uint32_t i = tile_begin + wave * (64u * OS_KPT) + q * 64u + lane;
			key[q] = i < n ? keys[i] : (K)0;
			if (HAS_V && m == 1) val[q] = i < n ? vals[i] : V();
		}
		os_rank_tile<K>(key, tile_begin, n, shift, whist, dig, local);
		__syncthreads();
		if (m > 1) {
			chunk_count[threadIdx.x] += whist[0][threadIdx.x] + whist[1][threadIdx.x] + whist[2][threadIdx.x] + whist[3][threadIdx.x];
			whist[0][threadIdx.x] = 0; whist[1][threadIdx.x] = 0; whist[2][threadIdx.x] = 0; whist[3][threadIdx.x] = 0;
			__syncthreads();
		}
	}
	{
		// 2./3. thread d owns digit d: publish, collect
		const uint32_t d = threadIdx.x;
		const uint32_t cnt = m == 1 ? whist[0][d] + whist[1][d] + whist[2][d] + whist[3][d] : chunk_count[d];
		__hip_atomic_store(counts + (size_t)w * 256 + d, OS_READY | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t group = w / OS_GROUP, first = group * OS_GROUP, ngroups = (nwg + OS_GROUP - 1) / OS_GROUP;
		const uint32_t in_group = os_sum_published(counts + (size_t)first * 256 + d, w - first, 256);
		if ((w % OS_GROUP) == OS_GROUP - 1 || w == nwg - 1)
			__hip_atomic_store(group_sum + (size_t)group * 256 + d, OS_READY | (in_group + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t before_groups = os_sum_published(group_sum + d, group, 256);
		(void)ngroups;
		const uint32_t total = group_sum[OS_HIST_OFFSET + d];          // (os_hist: the whole input's count of this digit -- nobody waits for a LATER workgroup)
		// digit base: exclusive scan of the 256 digit totals
		uint32_t x = total;
		for (int k = 1; k < 64; k <<= 1) { uint32_t y = __shfl_up(x, k); if ((int)lane >= k) x += y; }
		if (lane == 63) wtot[wave] = x;
		__syncthreads();
		uint32_t base = x - total;
		for (unsigned v = 0; v < wave; ++v) base += wtot[v];
		offs[d] = base + before_groups + in_group;
	}
	__syncthreads();
	// 4. scatter
	if (m == 1) {
		const uint32_t tile_begin = first_tile * OS_TILE;
#pragma unroll
		for (int q = 0; q < OS_KPT; ++q) {
			uint32_t i = tile_begin + wave * (64u * OS_KPT) + q * 64u + lane;
			if (i < n) {
				uint32_t d = dig[q];
				uint32_t pos = offs[d] + local[q];
				for (unsigned v = 0; v < wave; ++v) pos += whist[v][d];
				keys_out[pos] = key[q];
				if (HAS_V) vals_out[pos] = val[q];
			}
		}
		return;
	}
	for (uint32_t t = first_tile; t < last_tile; ++t) {
		const uint32_t tile_begin = t * OS_TILE;
#pragma unroll
		for (int q = 0; q < OS_KPT; ++q) {
			uint32_t i = tile_begin + wave * (64u * OS_KPT) + q * 64u + lane;
			key[q] = i < n ? keys[i] : (K)0;
			if (HAS_V) val[q] = i < n ? vals[i] : V();
		}
		os_rank_tile<K>(key, tile_begin, n, shift, whist, dig, local);
		__syncthreads();
#pragma unroll
		for (int q = 0; q < OS_KPT; ++q) {
			uint32_t i = tile_begin + wave * (64u * OS_KPT) + q * 64u + lane;
			if (i < n) {
				uint32_t d = dig[q];
				uint32_t pos = offs[d] + local[q];
				for (unsigned v = 0; v < wave; ++v) pos += whist[v][d];
				keys_out[pos] = key[q];
				if (HAS_V) vals_out[pos] = val[q];
			}
		}
		__syncthreads();
		offs[threadIdx.x] += whist[0][threadIdx.x] + whist[1][threadIdx.x] + whist[2][threadIdx.x] + whist[3][threadIdx.x];
		whist[0][threadIdx.x] = 0; whist[1][threadIdx.x] = 0; whist[2][threadIdx.x] = 0; whist[3][threadIdx.x] = 0;
		__syncthreads();
	}
}

static std::mutex& nh_cooperative_launch_lock() { static std::mutex m; return m; }

template<typename K, typename V, bool HAS_V>
static int onesweep_impl(nh_context* ctx, K* keys_a, K* keys_b, V* vals_a, V* vals_b, const uint32_t* d_count, uint32_t capacity, uint32_t expected, uint32_t* scratch, const os_shifts& sh) {
	K* kin = keys_a; K* kout = keys_b;
	V* vin = vals_a; V* vout = vals_b;
	// one workgroup per expected tile (+12 %), never more than are resident at once; any count is still sorted correctly
	// (workgroups then own chunks of several tiles)
	const uint64_t cap_tiles = ((uint64_t)capacity + OS_TILE - 1) / OS_TILE + 1;
	uint64_t want = expected ? ((uint64_t)expected + expected / 8 + OS_TILE - 1) / OS_TILE + 2 : cap_tiles;
	if (want > cap_tiles) want = cap_tiles;
	// every workgroup of a pass must be resident at once (they wait for each other): bound the launch by 3/4 of what the
	// occupancy calculator says fits (kernels of the side stream may hold some slots for a while; they never wait for us)
	int& resident = ctx->os_resident;          // per context (= per device): the bound differs between devices
	if (!resident) {
		int per_cu = 0, cus = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, os_pass<K, V, HAS_V>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 64;
		resident = per_cu * cus * 3 / 4;
		if (resident > OS_MAX_GRID) resident = OS_MAX_GRID;
		if (resident < 16) resident = 16;
	}
	const uint32_t grid = want > (uint64_t)resident ? (uint32_t)resident : (uint32_t)want;
	if (ctx->timing) nh_timer_begin(ctx, "radix_clear");
	hipError_t e = hipMemsetAsync(scratch, 0, sizeof(uint32_t) * (size_t)sh.n * OS_PASS_WORDS, ctx->stream);
	if (ctx->timing) nh_timer_end(ctx);
	if (e != hipSuccess) ctx->last_hip_error = (int)e;
	// digit totals of every pass, up front (os_hist): a pass then only waits for workgroups holding LOWER tickets, which are running or done -- no deadlock whatever
	// else shares the device; the occupancy bound below is about speed (spinning workgroups that crowd out the ones they wait for), no longer about correctness
	NH_LAUNCH(ctx, "radix_hist", (os_hist<K>), (uint32_t)(cap_tiles < 512 ? cap_tiles : 512), 256, (const K*)kin, d_count, scratch, sh);
	// (history) The workgroups of a pass used to wait for each other in both directions (earlier counts, group sums of ALL groups): all of them had to be resident.  The launch is
	// bounded by 3/4 of what the occupancy query says is resident at once, and nothing this library runs beside it (side stream) ever waits for it, so
	// workgroups that find the compute units busy with somebody else's kernels are merely late.  NH_COOPERATIVE=1 turns the bound into the runtime's
	// promise (hipLaunchCooperativeKernel; a grid it finds too large is halved -- any grid sorts correctly).  It is not the default because of what it
	// costs: with two processes sharing one MI355X (the N = 2 rehearsal, whose refresh re-seeds the sort every 16 steps) every cooperative launch took
	// ~10 ms against ~0.05 ms for the plain one (profiles/r03_cooperative_launch_ab.txt).
	uint32_t g = grid;
	for (int p = 0; p < sh.n; ++p) {
		uint32_t* c = scratch + (size_t)p * OS_PASS_WORDS;
		uint32_t* c2 = c + (size_t)OS_MAX_GRID * 256;
		int shift = sh.s[p];
		void* args[] = { (void*)&kin, (void*)&kout, (void*)&vin, (void*)&vout, (void*)&d_count, (void*)&c, (void*)&c2, (void*)&shift };
		if (ctx->timing) nh_timer_begin(ctx, "radix_pass");
		hipError_t le = hipErrorUnknown;
		while (!ctx->os_plain_launch) {
			// (one cooperative launch at a time process-wide: the runtime sets up its cooperative queue on first use and two host threads racing
			// there -- two worlds stepped by two threads -- crashed its teardown; the lock costs nothing on the one-thread path)
			std::lock_guard<std::mutex> guard(nh_cooperative_launch_lock());
			le = hipLaunchCooperativeKernel((const void*)os_pass<K, V, HAS_V>, dim3(g), dim3(256), args, 0, ctx->stream);
			if (le == hipSuccess) break;
			(void)hipGetLastError();
			if (le == hipErrorCooperativeLaunchTooLarge && g > 16u) { g /= 2u; resident = (int)g; continue; }
			ctx->os_plain_launch = true;           // no cooperative launches on this device / runtime: the occupancy-bounded plain launch, as before
		}
		if (ctx->os_plain_launch) hipLaunchKernelGGL((os_pass<K, V, HAS_V>), dim3(g), dim3(256), 0, ctx->stream, kin, kout, vin, vout, d_count, c, c2, shift);
		if (ctx->timing) nh_timer_end(ctx);
		K* tk = kin; kin = kout; kout = tk;
		V* tv = vin; vin = vout; vout = tv;
	}
	return sh.n & 1;      // 1: the sorted data sits in the *_b buffers
}

static os_shifts shifts_for(int begin_bit, int end_bit, os_shifts sh = os_shifts{ {}, 0 }) {
	for (int s = begin_bit; s < end_bit && sh.n < OS_MAX_PASSES; s += 8) sh.s[sh.n++] = OS_DESC(s, 8, 0);
	return sh;
}

int nh_onesweep_u32_u32(nh_context* ctx, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                         const uint32_t* d_count, uint32_t capacity, uint32_t expected, uint32_t* scratch, int begin_bit, int end_bit) {
	return onesweep_impl<uint32_t, uint32_t, true>(ctx, keys_a, keys_b, vals_a, vals_b, d_count, capacity, expected, scratch, shifts_for(begin_bit, end_bit));
}

// sorts 64-bit keys made of two `bits`-wide fields, one at bit 0 and one at bit 32, by (high field, low field): the 2 x bits
// significant bits are consumed eight at a time as if the fields were adjacent, so 20-bit fields take 5 passes, not 6
static os_shifts two_field_shifts(int bits) {
	os_shifts sh = os_shifts{ {}, 0 };
	for (int lo = 0; lo < 2 * bits && sh.n < OS_MAX_PASSES; lo += 8) {
		if (lo + 8 <= bits) sh.s[sh.n++] = OS_DESC(lo, 8, 0);                    // inside the low field
		else if (lo >= bits) sh.s[sh.n++] = OS_DESC(32 + lo - bits, 8, 0);       // inside the high field
		else sh.s[sh.n++] = OS_DESC(lo, bits - lo, 32);                          // the last bits of the low field + the first of the high
	}
	return sh;
}

int nh_onesweep_u64_u32_two_fields(nh_context* ctx, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                                    const uint32_t* d_count, uint32_t capacity, uint32_t expected, uint32_t* scratch, int bits) {
	return onesweep_impl<uint64_t, uint32_t, true>(ctx, keys_a, keys_b, vals_a, vals_b, d_count, capacity, expected, scratch, two_field_shifts(bits));
}

// ---- tag sort seeded by the previous step's order ---------------------------------------------------------------------------
// Contact records barely change from one step to the next, so last step's sorted keys are near-perfect SPLITTERS for this
// step's: every BK_TARGET-th key of the previous result bounds a bucket that again receives about BK_TARGET keys.  Four
// kernels without any workgroup waiting for another (the radix passes above are chains of grid-wide hand-offs):
//   bk_count    bucket of every key (binary search over the splitters, held in LDS) + its arrival rank in the bucket
//   bk_scan     bucket starts (one workgroup)
//   bk_scatter  keys / values to their bucket
//   bk_sort     one workgroup per bucket: LSD radix sort in registers + LDS (a bucket of up to 2048 keys = one tile of the
//               ranking primitive above), written out in place; also emits the splitters of the next step
// Any distribution is sorted correctly: a bucket larger than a tile is sorted by its workgroup through global memory (slow,
// only seen when the world changes abruptly); with no history (first step, capacity change) the radix passes run instead and
// seed the splitters.  Keys are unique (one record per collider pair), so the arrival order inside a bucket never shows.
// Sort re-use (nh_collide): when this step's records sit at last step's positions with last step's keys, the order this sort left in the context's
// buffers still holds and its four kernels leave at once (they are launched either way: the decision is made on the device, by k_narrowphase).
__device__ __forceinline__ bool bk_order_still_holds(const nh_DevState* st) { return st->sort_valid != 0u && st->keys_changed == 0u && st->records == st->records_kept; }

#define BK_TARGET 1024u                 // default; NH_BUCKET_TARGET (read per context) changes it -- a test knob
#define BK_LDS_SPLITTERS 4096u

__global__ __launch_bounds__(256) void bk_count(const uint64_t* __restrict__ keys, const nh_DevState* __restrict__ st, const uint64_t* __restrict__ splitters,
                                                uint2* __restrict__ place, uint32_t* __restrict__ counts) {
	__shared__ uint64_t s_spl[BK_LDS_SPLITTERS];
	if (bk_order_still_holds(st)) { if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<nh_DevState*>(st)->sort_reuses += 1u; return; }
	const uint32_t n = st->records;
	const uint32_t nb = st->sort_buckets ? st->sort_buckets : 1u, ns = nb - 1u;
	// all splitters in LDS when they fit; otherwise every stride-th one (the last of each block of `stride`): the search first finds the block
	// in LDS, then finishes inside it in global memory (at most log2(stride) dependent loads instead of log2(ns))
	const uint32_t stride = (ns + BK_LDS_SPLITTERS - 1u) / BK_LDS_SPLITTERS > 1u ? (ns + BK_LDS_SPLITTERS - 1u) / BK_LDS_SPLITTERS : 1u;
	const uint32_t nblocks = (ns + stride - 1u) / stride;                   // <= BK_LDS_SPLITTERS
	for (uint32_t k = threadIdx.x; k < nblocks; k += 256) { const uint32_t last = (k + 1u) * stride - 1u; s_spl[k] = splitters[last < ns ? last : ns - 1u]; }
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t nbits = 32u - (uint32_t)__clz(nb);             // bucket ids are < 2^nbits
	// Arrival ranks.  Lanes of a wave that share a bucket find each other by ballots; the groups of the workgroup's four waves then meet in
	// a small LDS table (open addressing on the bucket id), so that ONE global atomic per bucket and 256-record slice hands out the ranks --
	// when records arrive in key order (a settled ball pit: tag order follows the lattice, and so do the broadphase cells) every wave
	// of the machine otherwise queues up on the same few counters, and same-address atomics are ~12 ns each, serialised.  A full table
	// (records arriving in scattered order: many buckets per slice) falls back to one global atomic per group.
	constexpr uint32_t TAB = 32u, EMPTY = 0xFFFFFFFFu;
	__shared__ uint32_t t_key[TAB], t_cnt[TAB], t_base[TAB];
	const uint32_t rounded = (n + 255u) & ~255u;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += gridDim.x * blockDim.x) {       // block-uniform trip count
		if (threadIdx.x < TAB) { t_key[threadIdx.x] = EMPTY; t_cnt[threadIdx.x] = 0u; }
		const bool valid = i < n;
		const uint64_t key = valid ? keys[i] : 0ull;
		uint32_t lo = 0, hi = nblocks;                     // bucket = number of splitters <= key
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (s_spl[mid] <= key) lo = mid + 1; else hi = mid; }
		if (stride > 1u) {
			// blocks 0 .. lo-1 lie entirely at or below the key; the answer is inside block lo (if there is one)
			uint32_t f0 = lo * stride, f1 = min(f0 + stride, ns);
			if (lo >= nblocks) { f0 = ns; f1 = ns; }
			while (f0 < f1) { uint32_t mid = (f0 + f1) >> 1; if (splitters[mid] <= key) f0 = mid + 1; else f1 = mid; }
			lo = f0;
		}
		const uint32_t b = lo;
		unsigned long long m = __ballot(valid);
		for (uint32_t bit = 0; bit < nbits; ++bit) {
			const unsigned long long bal = __ballot((b >> bit) & 1u);
			m &= ((b >> bit) & 1u) ? bal : ~bal;
		}
		const int leader = valid ? __ffsll((long long)m) - 1 : (int)lane;
		const bool is_leader = valid && (int)lane == leader;
		const uint32_t group = (uint32_t)__popcll(m);
		__syncthreads();                                   // the table is clear
		uint32_t slot = EMPTY, off = 0;
		if (is_leader) {
			uint32_t h = (b * 2654435761u) >> 27;          // 5 bits
			for (uint32_t probe = 0; probe < TAB; ++probe, h = (h + 1u) & (TAB - 1u)) {
				const uint32_t old = atomicCAS(&t_key[h], EMPTY, b);
				if (old == EMPTY || old == b) { slot = h; break; }
			}
			if (slot != EMPTY) off = atomicAdd(&t_cnt[slot], group);
			else off = atomicAdd(&counts[b], group);           // table full: straight to the global counter
		}
		__syncthreads();
		if (threadIdx.x < TAB && t_key[threadIdx.x] != EMPTY) t_base[threadIdx.x] = atomicAdd(&counts[t_key[threadIdx.x]], t_cnt[threadIdx.x]);
		__syncthreads();
		if (is_leader && slot != EMPTY) off += t_base[slot];
		const uint32_t rank = (uint32_t)__shfl((int)off, leader) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		if (valid) place[i] = make_uint2(b, rank);
	}
}

__global__ __launch_bounds__(1024) void bk_scan(const nh_DevState* __restrict__ st, uint32_t* __restrict__ counts, uint32_t* __restrict__ starts) {
	__shared__ uint32_t wsum[16];
	if (bk_order_still_holds(st)) return;
	const uint32_t nb = st->sort_buckets ? st->sort_buckets : 1u;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nb; base += 1024) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < nb ? counts[i] : 0u;
		if (i < nb) counts[i] = 0;                              // ready for the next step
		uint32_t x = v;
		for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
		if (lane == 63) wsum[wave] = x;
		__syncthreads();
		uint32_t woff = 0, total = 0;
		for (uint32_t w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
		if (i < nb) starts[i] = carry + woff + x - v;
		carry += total;
		__syncthreads();
	}
	if (threadIdx.x == 0) starts[nb] = carry;
}

__global__ __launch_bounds__(256) void bk_scatter(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const nh_DevState* __restrict__ st,
                                                  const uint2* __restrict__ place, const uint32_t* __restrict__ starts, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
	if (bk_order_still_holds(st)) return;
	const uint32_t n = st->records;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint2 p = place[i];
		const uint32_t pos = starts[p.x] + p.y;
		keys_out[pos] = keys[i];
		vals_out[pos] = vals[i];
	}
}

// digit totals of the workgroup (whist summed over its four waves) -> exclusive bases in offs[]
__device__ __forceinline__ void bk_digit_bases(const uint32_t total, uint32_t* offs, uint32_t* wtot) {
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t x = total;
	for (int k = 1; k < 64; k <<= 1) { uint32_t y = __shfl_up(x, k); if ((int)lane >= k) x += y; }
	if (lane == 63) wtot[wave] = x;
	__syncthreads();
	uint32_t base = x - total;
	for (unsigned v = 0; v < wave; ++v) base += wtot[v];
	offs[threadIdx.x] = base;
	__syncthreads();
}

// one LSD pass over `cnt` keys by ONE workgroup through global memory (buckets that do not fit a tile)
__device__ void bk_pass_global(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, uint32_t cnt, int desc,
                               uint32_t (*whist)[256], uint32_t* offs, uint32_t* wtot) {
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint64_t key[OS_KPT];
	uint32_t val[OS_KPT], dig[OS_KPT], local[OS_KPT];
	const uint32_t ntiles = (cnt + OS_TILE - 1) / OS_TILE;
	uint32_t mine = 0;                                          // keys of digit threadIdx.x
	for (uint32_t t = 0; t < ntiles; ++t) {
#pragma unroll
		for (int q = 0; q < OS_KPT; ++q) { uint32_t i = t * OS_TILE + wave * (64u * OS_KPT) + q * 64u + lane; key[q] = i < cnt ? kin[i] : 0ull; }
		os_rank_tile<uint64_t>(key, t * OS_TILE, cnt, desc, whist, dig, local);
		__syncthreads();
		mine += whist[0][threadIdx.x] + whist[1][threadIdx.x] + whist[2][threadIdx.x] + whist[3][threadIdx.x];
		whist[0][threadIdx.x] = 0; whist[1][threadIdx.x] = 0; whist[2][threadIdx.x] = 0; whist[3][threadIdx.x] = 0;
		__syncthreads();
	}
	bk_digit_bases(mine, offs, wtot);
	for (uint32_t t = 0; t < ntiles; ++t) {
#pragma unroll
		for (int q = 0; q < OS_KPT; ++q) {
			uint32_t i = t * OS_TILE + wave * (64u * OS_KPT) + q * 64u + lane;
			key[q] = i < cnt ? kin[i] : 0ull; val[q] = i < cnt ? vin[i] : 0u;
		}
		os_rank_tile<uint64_t>(key, t * OS_TILE, cnt, desc, whist, dig, local);
		__syncthreads();
#pragma unroll
		for (int q = 0; q < OS_KPT; ++q) {
			uint32_t i = t * OS_TILE + wave * (64u * OS_KPT) + q * 64u + lane;
			if (i < cnt) {
				uint32_t pos = offs[dig[q]] + local[q];
				for (unsigned v = 0; v < wave; ++v) pos += whist[v][dig[q]];
				kout[pos] = key[q]; vout[pos] = val[q];
			}
		}
		__syncthreads();
		offs[threadIdx.x] += whist[0][threadIdx.x] + whist[1][threadIdx.x] + whist[2][threadIdx.x] + whist[3][threadIdx.x];
		whist[0][threadIdx.x] = 0; whist[1][threadIdx.x] = 0; whist[2][threadIdx.x] = 0; whist[3][threadIdx.x] = 0;
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void bk_sort(uint64_t* kb, uint32_t* vb, uint64_t* ka, uint32_t* va,      // (no __restrict__: oversized buckets ping-pong between the two)
                                               const uint32_t* __restrict__ starts, nh_DevState* __restrict__ st, os_shifts sh, int field_bits, uint64_t* __restrict__ splitters, uint32_t lds_cap, uint32_t target) {
	__shared__ uint64_t s_key[OS_TILE];
	__shared__ uint32_t s_val[OS_TILE];
	__shared__ uint32_t whist[4][256];
	__shared__ uint32_t offs[256];
	__shared__ uint32_t wtot[4];
	__shared__ uint64_t s_ref, s_diff[4];
	if (bk_order_still_holds(st)) return;
	const uint32_t nb = st->sort_buckets ? st->sort_buckets : 1u;
	const uint32_t n = st->records;
	const uint64_t lomask = field_bits >= 32 ? 0xFFFFFFFFull : ((1ull << field_bits) - 1ull);
	if (blockIdx.x == 0 && threadIdx.x == 0) st->sort_buckets_next = n ? (n + target - 1u) / target : 1u;
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	whist[0][threadIdx.x] = 0; whist[1][threadIdx.x] = 0; whist[2][threadIdx.x] = 0; whist[3][threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
		const uint32_t s = starts[b], cnt = starts[b + 1] - s;
		if (cnt == 0) continue;
		if (cnt <= lds_cap) {
			uint64_t key[OS_KPT];
			uint32_t val[OS_KPT], dig[OS_KPT], local[OS_KPT];
			// the two tag fields packed side by side: one plain 2 x field_bits number.  The keys of a bucket lie between two neighbouring
			// splitters, so they agree in their high bits: only the bits in which any of them differs from the first are sorted on
#pragma unroll
			for (int q = 0; q < OS_KPT; ++q) {
				uint32_t i = wave * (64u * OS_KPT) + q * 64u + lane;
				const uint64_t k = i < cnt ? kb[s + i] : 0ull;
				key[q] = (k & lomask) | ((k >> 32) << field_bits);
				val[q] = i < cnt ? vb[s + i] : 0u;
			}
			if (threadIdx.x == 0) s_ref = key[0];
			__syncthreads();
			uint64_t diff = 0;
			{
				const uint64_t ref = s_ref;
#pragma unroll
				for (int q = 0; q < OS_KPT; ++q) { uint32_t i = wave * (64u * OS_KPT) + q * 64u + lane; if (i < cnt) diff |= key[q] ^ ref; }
				for (int d = 32; d >= 1; d >>= 1) diff |= (uint64_t)__shfl_xor((long long)diff, d);
				if (lane == 0) s_diff[wave] = diff;
				__syncthreads();
				diff = s_diff[0] | s_diff[1] | s_diff[2] | s_diff[3];
			}
			const int npass = diff ? (64 - __clzll((long long)diff) + 7) / 8 : 0;
			for (int p = 0; p < npass; ++p) {
				os_rank_tile<uint64_t>(key, 0u, cnt, OS_DESC(8 * p, 8, 0), whist, dig, local);
				__syncthreads();
				bk_digit_bases(whist[0][threadIdx.x] + whist[1][threadIdx.x] + whist[2][threadIdx.x] + whist[3][threadIdx.x], offs, wtot);
#pragma unroll
				for (int q = 0; q < OS_KPT; ++q) {
					uint32_t i = wave * (64u * OS_KPT) + q * 64u + lane;
					if (i < cnt) {
						uint32_t pos = offs[dig[q]] + local[q];
						for (unsigned v = 0; v < wave; ++v) pos += whist[v][dig[q]];
						s_key[pos] = key[q]; s_val[pos] = val[q];
					}
				}
				__syncthreads();
#pragma unroll
				for (int q = 0; q < OS_KPT; ++q) {
					uint32_t i = wave * (64u * OS_KPT) + q * 64u + lane;
					if (i < cnt) { key[q] = s_key[i]; val[q] = s_val[i]; }
				}
				whist[0][threadIdx.x] = 0; whist[1][threadIdx.x] = 0; whist[2][threadIdx.x] = 0; whist[3][threadIdx.x] = 0;
				__syncthreads();
			}
#pragma unroll
			for (int q = 0; q < OS_KPT; ++q) {
				uint32_t i = wave * (64u * OS_KPT) + q * 64u + lane;
				if (i < cnt) {
					const uint32_t g = s + i;
					const uint64_t k = (key[q] & lomask) | ((key[q] >> field_bits) << 32);
					ka[g] = k; va[g] = val[q];
					if (g && (g % target) == 0) splitters[g / target - 1u] = k;
				}
			}
		} else {
			uint64_t* ck = kb + s; uint32_t* cv = vb + s;
			uint64_t* ok = ka + s; uint32_t* ov = va + s;
			for (int p = 0; p < sh.n; ++p) {
				bk_pass_global(ck, cv, ok, ov, cnt, sh.s[p], whist, offs, wtot);
				uint64_t* tk = ck; ck = ok; ok = tk;
				uint32_t* tv = cv; cv = ov; ov = tv;
			}
			// the result sits in `ck`: ka + s after an odd number of passes
			for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
				const uint32_t g = s + i;
				const uint64_t k = ck[i];
				if (ck != ka + s) { ka[g] = k; va[g] = cv[i]; }
				if (g && (g % target) == 0) splitters[g / target - 1u] = k;
			}
			__syncthreads();
		}
	}
}

// splitters for the next step from a sorted key array (after the radix passes ran: first step, capacity change)
__global__ __launch_bounds__(256) void bk_seed(const uint64_t* __restrict__ sorted, nh_DevState* __restrict__ st, uint64_t* __restrict__ splitters, uint32_t target) {
	const uint32_t n = st->records;
	const uint32_t nb = n ? (n + target - 1u) / target : 1u;
	if (blockIdx.x == 0 && threadIdx.x == 0) st->sort_buckets_next = nb;
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j + 1 < nb; j += gridDim.x * blockDim.x) splitters[j] = sorted[(size_t)(j + 1u) * target];
}

static uint32_t bucket_target(const nh_context* ctx) { return ctx->sort_target >= 1u && ctx->sort_target <= (uint32_t)OS_TILE ? ctx->sort_target : BK_TARGET; }
uint32_t nh_bucket_sort_max_buckets(const nh_context* ctx, uint32_t capacity) { return capacity / bucket_target(ctx) + 2u; }

// keys_a/vals_a: in (left as they are); keys_b/vals_b: scratch of the same size; keys_out/vals_out: the sorted result.  st->records keys; splitters /
// counts / starts: the context's (nh_bucket_sort_max_buckets(capacity) + 1 entries each, counts zeroed once); place: capacity entries of scratch.
void nh_bucket_sort_u64_u32(nh_context* ctx, const uint64_t* keys_a, uint64_t* keys_b, const uint32_t* vals_a, uint32_t* vals_b, uint32_t capacity, uint2* place, int field_bits,
                            uint64_t* keys_out, uint32_t* vals_out) {
	nh_DevState* st = ctx->d_state;
	// NH_BUCKET_TILE (read per context): buckets above this size take the through-global-memory path -- a test knob, the default is the tile
	const uint32_t lds_cap = ctx->sort_tile >= 1u && ctx->sort_tile <= (uint32_t)OS_TILE ? ctx->sort_tile : (uint32_t)OS_TILE;
	const uint32_t max_buckets = nh_bucket_sort_max_buckets(ctx, capacity);
	NH_LAUNCH(ctx, "bucket_count", bk_count, nh_grid_for(capacity, 256, 2048), 256, keys_a, st, ctx->sort_splitters, place, ctx->sort_counts);
	NH_LAUNCH(ctx, "bucket_scan", bk_scan, 1, 1024, st, ctx->sort_counts, ctx->sort_starts);
	NH_LAUNCH(ctx, "bucket_scatter", bk_scatter, nh_grid_for(capacity, 256, 2048), 256, keys_a, vals_a, st, place, ctx->sort_starts, keys_b, vals_b);
	NH_LAUNCH(ctx, "bucket_sort", bk_sort, max_buckets < 4096u ? max_buckets : 4096u, 256, keys_b, vals_b, keys_out, vals_out, ctx->sort_starts, st, two_field_shifts(field_bits), field_bits, ctx->sort_splitters, lds_cap, bucket_target(ctx));
}

void nh_bucket_sort_seed(nh_context* ctx, const uint64_t* sorted_keys, uint32_t capacity) {
	NH_LAUNCH(ctx, "bucket_seed", bk_seed, nh_grid_for(nh_bucket_sort_max_buckets(ctx, capacity), 256, 64), 256, sorted_keys, ctx->d_state, ctx->sort_splitters, bucket_target(ctx));
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

// Two kernels per scan: per-chunk sums, then every workgroup adds up the sums of the chunks before its own (<= 511 values) and
// scans its chunk.  Up to two arrays of the same length are scanned by the same pair of launches.
__global__ __launch_bounds__(256) void sc_sums(const uint32_t* __restrict__ in_a, const uint32_t* __restrict__ in_b, const uint32_t* __restrict__ d_n, uint32_t extra,
                                               uint32_t* __restrict__ tmp, const uint32_t* __restrict__ d_enable) {
	__shared__ uint32_t wsum[2][4];
	if (d_enable && !*d_enable) return;
	uint32_t n = *d_n + extra;
	uint32_t begin, end;
	rs_chunk(n, begin, end);
	uint32_t sa = 0, sb = 0;
	for (uint32_t i = begin + threadIdx.x; i < end; i += 256) { sa += in_a[i]; if (in_b) sb += in_b[i]; }
	for (int d = 32; d >= 1; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
	if ((threadIdx.x & 63) == 0) { wsum[0][threadIdx.x >> 6] = sa; wsum[1][threadIdx.x >> 6] = sb; }
	__syncthreads();
	if (threadIdx.x == 0) {
		tmp[blockIdx.x] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
		tmp[RS_G + blockIdx.x] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
	}
}

__global__ __launch_bounds__(256) void sc_final(const uint32_t* __restrict__ in_a, uint32_t* __restrict__ out_a, const uint32_t* __restrict__ in_b, uint32_t* __restrict__ out_b,
                                                const uint32_t* __restrict__ d_n, uint32_t extra, const uint32_t* __restrict__ tmp,
                                                uint32_t* __restrict__ d_total_a, uint32_t* __restrict__ d_total_b, const uint32_t* __restrict__ d_enable) {
	__shared__ uint32_t wsum[4];
	__shared__ uint32_t red[2][4];
	if (d_enable && !*d_enable) {
		if (blockIdx.x == 0 && threadIdx.x == 0) { if (d_total_a) *d_total_a = 0; if (d_total_b) *d_total_b = 0; }
		return;
	}
	uint32_t n = *d_n + extra;
	uint32_t begin, end;
	rs_chunk(n, begin, end);
	// carry = sum of the chunk sums before this one
	uint32_t ca = 0, cb = 0;
	for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256) { ca += tmp[i]; cb += tmp[RS_G + i]; }
	for (int d = 32; d >= 1; d >>= 1) { ca += __shfl_xor(ca, d); cb += __shfl_xor(cb, d); }
	if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ca; red[1][threadIdx.x >> 6] = cb; }
	__syncthreads();
	uint32_t carry_a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
	uint32_t carry_b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
		if (d_total_a) *d_total_a = carry_a + tmp[blockIdx.x];
		if (d_total_b) *d_total_b = carry_b + tmp[RS_G + blockIdx.x];
	}
	// eight consecutive items per thread: local sums, ONE workgroup scan per 2048 items and array, eight outputs
	constexpr uint32_t SC_IPT = 8;
	for (uint32_t tile = begin; tile < end; tile += RS_TILE * SC_IPT) {
		const uint32_t i0 = tile + threadIdx.x * SC_IPT;
		uint32_t va[SC_IPT], vb[SC_IPT], sa = 0, sb = 0;
#pragma unroll
		for (uint32_t k = 0; k < SC_IPT; ++k) va[k] = vb[k] = 0u;
		const bool full = i0 + SC_IPT <= end;           // (chunks start at multiples of 256 items and the arrays are 16-byte aligned)
		if (full) {
			const uint4 x0 = *reinterpret_cast<const uint4*>(in_a + i0), x1 = *reinterpret_cast<const uint4*>(in_a + i0 + 4);
			va[0] = x0.x; va[1] = x0.y; va[2] = x0.z; va[3] = x0.w; va[4] = x1.x; va[5] = x1.y; va[6] = x1.z; va[7] = x1.w;
			if (in_b) {
				const uint4 y0 = *reinterpret_cast<const uint4*>(in_b + i0), y1 = *reinterpret_cast<const uint4*>(in_b + i0 + 4);
				vb[0] = y0.x; vb[1] = y0.y; vb[2] = y0.z; vb[3] = y0.w; vb[4] = y1.x; vb[5] = y1.y; vb[6] = y1.z; vb[7] = y1.w;
			}
		} else {
#pragma unroll
			for (uint32_t k = 0; k < SC_IPT; ++k) { if (i0 + k < end) { va[k] = in_a[i0 + k]; if (in_b) vb[k] = in_b[i0 + k]; } }
		}
#pragma unroll
		for (uint32_t k = 0; k < SC_IPT; ++k) { sa += va[k]; sb += vb[k]; }
		uint32_t total;
		uint32_t ex = carry_a + block_exclusive_scan_256(sa, wsum, total);
		carry_a += total;
#pragma unroll
		for (uint32_t k = 0; k < SC_IPT; ++k) { const uint32_t v = va[k]; va[k] = ex; ex += v; }
		if (full) {
			*reinterpret_cast<uint4*>(out_a + i0) = make_uint4(va[0], va[1], va[2], va[3]);
			*reinterpret_cast<uint4*>(out_a + i0 + 4) = make_uint4(va[4], va[5], va[6], va[7]);
		} else {
#pragma unroll
			for (uint32_t k = 0; k < SC_IPT; ++k) if (i0 + k < end) out_a[i0 + k] = va[k];
		}
		__syncthreads();
		if (in_b) {
			ex = carry_b + block_exclusive_scan_256(sb, wsum, total);
			carry_b += total;
#pragma unroll
			for (uint32_t k = 0; k < SC_IPT; ++k) { const uint32_t v = vb[k]; vb[k] = ex; ex += v; }
			if (full) {
				*reinterpret_cast<uint4*>(out_b + i0) = make_uint4(vb[0], vb[1], vb[2], vb[3]);
				*reinterpret_cast<uint4*>(out_b + i0 + 4) = make_uint4(vb[4], vb[5], vb[6], vb[7]);
			} else {
#pragma unroll
				for (uint32_t k = 0; k < SC_IPT; ++k) if (i0 + k < end) out_b[i0 + k] = vb[k];
			}
			__syncthreads();
		}
	}
}

// `tmp` needs 2 * NH_SORT_GRID words
void nh_scan_u32(nh_context* ctx, const uint32_t* in, uint32_t* out, const uint32_t* d_count, uint32_t extra, uint32_t* tmp, uint32_t* d_total, const uint32_t* d_enable) {
	NH_LAUNCH(ctx, "scan_sums", sc_sums, RS_G, 256, in, (const uint32_t*)nullptr, d_count, extra, tmp, d_enable);
	NH_LAUNCH(ctx, "scan_final", sc_final, RS_G, 256, in, out, (const uint32_t*)nullptr, (uint32_t*)nullptr, d_count, extra, tmp, d_total, (uint32_t*)nullptr, d_enable);
}

void nh_scan2_u32(nh_context* ctx, const uint32_t* in_a, uint32_t* out_a, uint32_t* d_total_a, const uint32_t* in_b, uint32_t* out_b, uint32_t* d_total_b,
                  const uint32_t* d_count, uint32_t extra, uint32_t* tmp) {
	NH_LAUNCH(ctx, "scan_sums", sc_sums, RS_G, 256, in_a, in_b, d_count, extra, tmp, (const uint32_t*)nullptr);
	NH_LAUNCH(ctx, "scan_final", sc_final, RS_G, 256, in_a, out_a, in_b, out_b, d_count, extra, tmp, d_total_a, d_total_b, (const uint32_t*)nullptr);
}
