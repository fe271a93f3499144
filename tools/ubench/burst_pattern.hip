// burst_pattern.hip -- does kernel A's load schedule matter?  1024 row streams (one 512-thread workgroup per row, four loader waves,
// 16 KB per step, two register sets, a barrier per step, rows 2 MiB apart) with some arithmetic per step standing for the
// discriminator; CONTINUOUS: a set is re-requested right after it is consumed (kernel A: two steps between request and use);
// BURST: both sets are consumed, then both are re-requested (no loads outstanding while a workgroup computes: at any time only part
// of the 1024 streams has requests in the memory system).  stride_pattern.hip showed 512 streams x 2 sets reaching 7.2 TB/s against
// 6.2 for 1024 x 2.  (Measurement aid, not product.)
// build: hipcc --offload-arch=gfx950 -O3 -o burst_pattern burst_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool BURST, int WORK>
__global__ __launch_bounds__(512, 8) void streams(const f4 *src, size_t ch_f4, int steps, float *sink)
{
	extern __shared__ float dyn_lds[];
	if (steps < 0) sink[1] = dyn_lds[threadIdx.x];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool loader = wave >= 4;
	const int kw = wave - 4;
	constexpr int NLD = 4;
	const f4 *p = src + (size_t)blockIdx.x * ch_f4;
	f4 acc = {0, 0, 0, 0};
	float w = (float)lane;
	auto work = [&](f4 x) {          // WORK dependent-ish FMAs per load: the discriminator's arithmetic
#pragma unroll
		for (int i = 0; i < WORK; i++) { w = __builtin_fmaf(w, 1.0001f, x.x); x.x = __builtin_fmaf(x.y, 0.5f, w); }
		acc += x;
	};
	if (loader) {
		f4 v[2][NLD];
		auto ld = [&](int d, int step) {
#pragma unroll
			for (int r = 0; r < NLD; r++) v[d][r] = __builtin_nontemporal_load(p + (size_t)step * 1024 + 64 * (NLD * kw + r) + lane);
		};
		ld(0, 0); ld(1, 1);
		for (int s = 0; s < steps; s += 2) {
#pragma unroll
			for (int r = 0; r < NLD; r++) work(v[0][r]);
			if (!BURST && s + 2 < steps) ld(0, s + 2);
			__syncthreads();
#pragma unroll
			for (int r = 0; r < NLD; r++) work(v[1][r]);
			if (s + 3 < steps) ld(1, s + 3);
			if (BURST && s + 2 < steps) ld(0, s + 2);
			__syncthreads();
		}
	} else {
		for (int s = 0; s < steps; s++) {
#pragma unroll
			for (int i = 0; i < 4 * WORK; i++) w = __builtin_fmaf(w, 1.0001f, 0.5f);      // the round waves' share
			__syncthreads();
		}
	}
	if (acc.x + acc.y + acc.z + acc.w + w == 1.2345f) sink[0] = acc.x;
}

template <class F> float timeit(F f)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 3; i++) f();
	float sum = 0.f;
	for (int i = 0; i < 20; i++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); sum += ms; }
	return sum / 20;
}

int main()
{
	const int C = 1024, steps = 96;
	const size_t st = (size_t)2048 * 1024 / 16;
	f4 *buf; float *sink; hipMalloc(&buf, C * st * 16); hipMalloc(&sink, 8); hipMemset(buf, 0, C * st * 16);
	const double gb = (double)C * steps * 1024 * 16 / 1e9;
	for (int i = 0; i < 200; i++) streams<false, 0><<<C, 512, 39 * 1024>>>(buf, st, steps, sink);
	hipDeviceSynchronize();
#define RUN(B, W) { hipFuncSetAttribute((const void *)streams<B, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 39 * 1024); \
	const float ms = timeit([&] { streams<B, W><<<C, 512, 39 * 1024>>>(buf, st, steps, sink); }); \
	printf("%-10s work %3d: %.4f ms  %.0f GB/s\n", B ? "burst" : "continuous", W, ms, gb / (ms * 1e-3)); }
	for (int rep = 0; rep < 2; rep++) {
		RUN(false, 0) RUN(true, 0) RUN(false, 8) RUN(true, 8) RUN(false, 16) RUN(true, 16) RUN(false, 24) RUN(true, 24) RUN(false, 32) RUN(true, 32)
	}
	return 0;
}
