// stream_pattern.hip -- how fast can HBM be read with kernel A's access pattern?  (measurement aid, not product)
// 1024 concurrent channel streams (1.5 MB apart), 16 KB per stream per step, versus one grid-stride sweep.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip ; run: ./stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sweep(const f4 *src, size_t n16, float *sink)
{
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	f4 acc = {0, 0, 0, 0};
	for (; i + 7 * stride < n16; i += 8 * stride) {
		f4 v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
		for (int k = 0; k < 8; k++) acc += v[k];
	}
	if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
}

// one workgroup per channel; LOADERS of the WG's waves read, each step = STEP_F4 float4 per channel
// DEPTH register sets in flight; BAR: s_barrier per step; NT: nontemporal
template <int WG, int LOADERS, int DEPTH, bool BAR, bool NT, bool PAIR = false>
__global__ __launch_bounds__(WG) void streams(const f4 *src, size_t ch_f4, int steps, float *sink)
{
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool loader = wave >= (WG / 64 - LOADERS);
	const int kw = wave - (WG / 64 - LOADERS);
	constexpr int STEP_F4 = 1024;                       // 16 KB per step
	constexpr int NLD = STEP_F4 / (LOADERS * 64);
	const f4 *p = src + (size_t)blockIdx.x * ch_f4;
	f4 acc = {0, 0, 0, 0};
	if (loader) {
		f4 v[DEPTH][NLD];
#pragma unroll
		for (int d = 0; d < DEPTH; d++)
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const f4 *a = PAIR ? p + (size_t)d * STEP_F4 + 64 * NLD * kw + 128 * (r >> 1) + 2 * lane + (r & 1)
				                   : p + (size_t)d * STEP_F4 + 64 * (NLD * kw + r) + lane;
				v[d][r] = NT ? __builtin_nontemporal_load(a) : *a;
			}
		for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
			for (int d = 0; d < DEPTH; d++) {
#pragma unroll
				for (int r = 0; r < NLD; r++) acc += v[d][r];
				if (s + d + DEPTH < steps) {
#pragma unroll
					for (int r = 0; r < NLD; r++) {
						const f4 *a = PAIR ? p + (size_t)(s + d + DEPTH) * STEP_F4 + 64 * NLD * kw + 128 * (r >> 1) + 2 * lane + (r & 1)
						                   : p + (size_t)(s + d + DEPTH) * STEP_F4 + 64 * (NLD * kw + r) + lane;
						v[d][r] = NT ? __builtin_nontemporal_load(a) : *a;
					}
				}
				if (BAR) __syncthreads();
			}
		}
	} else {
		for (int s = 0; s < steps; s++) if (BAR) __syncthreads();
	}
	if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
}

template <class F> float timeit(F f)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	f();
	float best = 1e30f;
	for (int i = 0; i < 10; i++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
	return best;
}

int main()
{
	const int C = 1024, steps = 96; const size_t ch_f4 = (size_t)steps * 1024; const size_t n16 = C * ch_f4;
	f4 *buf; float *sink; hipMalloc(&buf, n16 * 16); hipMalloc(&sink, 4); hipMemset(buf, 0, n16 * 16);
	const double gb = n16 * 16 / 1e9;
	auto rep = [&](const char *name, float ms) { printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, gb / (ms * 1e-3)); };
	rep("sweep grid 4096", timeit([&] { sweep<<<4096, 256>>>(buf, n16, sink); }));
	rep("sweep grid 8192", timeit([&] { sweep<<<8192, 256>>>(buf, n16, sink); }));
	rep("streams wg256 load4 depth2 nobar", timeit([&] { streams<256, 4, 2, false, false><<<C, 256>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg256 load4 depth2 nobar nt", timeit([&] { streams<256, 4, 2, false, true><<<C, 256>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg256 load4 depth4 nobar nt", timeit([&] { streams<256, 4, 4, false, true><<<C, 256>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth2 bar", timeit([&] { streams<512, 4, 2, true, false><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth2 bar nt", timeit([&] { streams<512, 4, 2, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth2 nobar nt", timeit([&] { streams<512, 4, 2, false, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load8 depth2 bar nt", timeit([&] { streams<512, 8, 2, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load8 depth4 bar nt", timeit([&] { streams<512, 8, 4, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth4 bar nt", timeit([&] { streams<512, 4, 4, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg384 load4 depth2 bar nt", timeit([&] { streams<384, 4, 2, true, true><<<C, 384>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg384 load4 depth3 bar nt", timeit([&] { streams<384, 4, 3, true, true><<<C, 384>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg384 load4 depth4 bar nt", timeit([&] { streams<384, 4, 4, true, true><<<C, 384>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth3 bar nt", timeit([&] { streams<512, 4, 3, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth2 bar nt PAIR", timeit([&] { streams<512, 4, 2, true, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth2 bar    PAIR", timeit([&] { streams<512, 4, 2, true, false, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load4 depth2 bar nt (again)", timeit([&] { streams<512, 4, 2, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	rep("streams wg512 load2 depth2 bar nt", timeit([&] { streams<512, 2, 2, true, true><<<C, 512>>>(buf, ch_f4, steps, sink); }));
	return 0;
}
