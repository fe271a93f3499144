// read_plus_valu.hip -- does arithmetic on the CU cost read bandwidth?  1024 row streams, four 512-thread workgroups per CU, 16 KB per
// step, two register sets with kernel A's (conservative-wait) schedule: 6.9 TB/s as a pure read (hybrid_depth.hip).  Here the four
// NON-loading waves of every workgroup run W dependent-free fma per lane and step (kernel A's round waves: ~170 VALU per tile), the
// loader waves X more (kernel A's discriminator: ~80).  If the rate falls towards kernel A's 6.3 TB/s with W = 170-200, the bound is
// shared (power / issue), not the load schedule.  (Measurement aid, not product.)
// build: hipcc --offload-arch=gfx950 -O3 -o read_plus_valu read_plus_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void streams(const f4 *src, size_t ch_f4, int steps, float *sink, int W, int X)
{
	extern __shared__ __attribute__((aligned(16))) f4 pad[];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool loader = wave >= 4;
	const int kw = __builtin_amdgcn_readfirstlane(wave - 4);
	const f4 *p = src + (size_t)blockIdx.x * ch_f4;
	f4 acc = {0, 0, 0, 0};
	float w0 = 1.0f + tid, w1 = 2.0f, w2 = 3.0f, w3 = 4.0f;
	const float a = 0.999f, b = 0.001f;
	if (steps < 0) sink[1] = pad[tid].x;
	auto work = [&](int n) {
		for (int i = 0; i < n; i += 4) {
			w0 = __builtin_fmaf(w0, a, b); w1 = __builtin_fmaf(w1, a, b); w2 = __builtin_fmaf(w2, a, b); w3 = __builtin_fmaf(w3, a, b);
		}
	};
	if (loader) {
		f4 va[4], vb[4];
		auto ld = [&](int step, f4 (&v)[4]) {
#pragma unroll
			for (int r = 0; r < 4; r++) v[r] = __builtin_nontemporal_load(p + (size_t)step * 1024 + 64 * (4 * kw + r) + lane);
		};
		auto use = [&](const f4 (&v)[4]) {
#pragma unroll
			for (int r = 0; r < 4; r++) acc += v[r];
		};
		ld(0, va); ld(1, vb);
		for (int s = 0; s < steps; s += 2) {
			use(va); work(X); if (s + 2 < steps) ld(s + 2, va);
			__syncthreads();
			if (s + 1 < steps) { use(vb); work(X); if (s + 3 < steps) ld(s + 3, vb); }
			__syncthreads();
		}
	} else {
		for (int s = 0; s < steps; s += 2) { work(W); __syncthreads(); work(W); __syncthreads(); }
	}
	if (acc.x + acc.y + acc.z + acc.w + w0 + w1 + w2 + w3 == 1.2345f) sink[0] = acc.x;
}

template <class F> float timeit(F f)
{
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	for (int i = 0; i < 5; i++) f();
	float sum = 0.f;
	for (int i = 0; i < 30; i++) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); sum += ms; }
	return sum / 30;
}

int main()
{
	const int C = 1024, steps = 96;
	const size_t stride_f4 = (size_t)2048 * 1024 / 16;
	f4 *buf; float *sink; (void)hipMalloc(&buf, C * stride_f4 * 16); (void)hipMalloc(&sink, 8); (void)hipMemset(buf, 0, C * stride_f4 * 16);
	const double gb = (double)C * steps * 16384 / 1e9;
	const size_t lds = 39 * 1024;
	for (int i = 0; i < 2000; i++) streams<<<C, 512, lds>>>(buf, stride_f4, steps, sink, 200, 80);       // clocks AND power up to the steady state
	(void)hipDeviceSynchronize();
	for (int rep = 0; rep < 2; rep++)
		for (int X : {0, 80})
			for (int W : {0, 100, 200, 400, 800}) {
				const float ms = timeit([&] { streams<<<C, 512, lds>>>(buf, stride_f4, steps, sink, W, X); });
				printf("fma per lane and step: other waves %4d, loader waves %3d: %.4f ms %.0f GB/s\n", W, X, ms, gb / (ms * 1e-3));
			}
	printf("status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
	return 0;
}
