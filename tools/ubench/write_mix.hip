// write_mix.hip -- do kernel A's few small writes (bit-ring words every tile: 1 % of its bytes) cost read bandwidth?  stride_pattern's
// continuous two-set read of 1024 row streams, plus W bytes written per workgroup and step by one lane-group of a non-loader wave
// (W = 0, 32, 64 as 16-byte stores; or one 4-byte store).  (Measurement aid, not product.)
// build: hipcc --offload-arch=gfx950 -O3 -o write_mix write_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W>      // bytes written per workgroup and step: 0, 4, 32, 64, 256
__global__ __launch_bounds__(512, 8) void streams(const f4 *src, size_t ch_f4, int steps, float *sink, float *dst)
{
	extern __shared__ float dyn_lds[];
	if (steps < 0) sink[1] = dyn_lds[threadIdx.x];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool loader = wave >= 4;
	const int kw = wave - 4;
	constexpr int NLD = 4;
	const f4 *p = src + (size_t)blockIdx.x * ch_f4;
	f4 acc = {0, 0, 0, 0};
	if (loader) {
		f4 v[2][NLD];
		auto ld = [&](int d, int step) {
#pragma unroll
			for (int r = 0; r < NLD; r++) v[d][r] = __builtin_nontemporal_load(p + (size_t)step * 1024 + 64 * (NLD * kw + r) + lane);
		};
		ld(0, 0); ld(1, 1);
		for (int s = 0; s < steps; s += 2) {
#pragma unroll
			for (int d = 0; d < 2; d++) {
#pragma unroll
				for (int r = 0; r < NLD; r++) acc += v[d][r];
				if (s + d + 2 < steps) ld(d, s + d + 2);
				__syncthreads();
			}
		}
	} else {
		// a ring of 512 bytes per workgroup, rewritten cyclically like the bit ring (stays in L2) ...
		float *ring = dst + (size_t)blockIdx.x * 128;
		for (int s = 0; s < steps; s++) {
			if (wave == 0) {
				if (W == 4 && lane == 0) ring[(s * 1) & 127] = (float)s;
				if (W >= 16 && lane < W / 16) reinterpret_cast<f4 *>(ring)[((s * (W / 16)) + lane) & 31] = f4{(float)s, 0, 0, 0};
			}
			__syncthreads();
		}
	}
	if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
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
	f4 *buf; float *sink, *dst; hipMalloc(&buf, C * st * 16); hipMalloc(&sink, 8); hipMalloc(&dst, (size_t)C * 512 + (size_t)C * steps * 1024); hipMemset(buf, 0, C * st * 16);
	const double gb = (double)C * steps * 1024 * 16 / 1e9;
	for (int i = 0; i < 3000; i++) streams<0><<<C, 512, 39 * 1024>>>(buf, st, steps, sink, dst);      // ~0.8 s: the clock ramps
	hipDeviceSynchronize();
#define RUN(W) { hipFuncSetAttribute((const void *)streams<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 39 * 1024); \
	const float ms = timeit([&] { streams<W><<<C, 512, 39 * 1024>>>(buf, st, steps, sink, dst); }); \
	printf("ring writes %3d B per workgroup and step: %.4f ms  %.0f GB/s\n", W, ms, gb / (ms * 1e-3)); }
	for (int rep = 0; rep < 3; rep++) { RUN(0) RUN(4) RUN(32) RUN(64) RUN(256) }
	return 0;
}
