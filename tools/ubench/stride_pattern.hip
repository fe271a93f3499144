// stride_pattern.hip -- 1024 concurrent row streams (kernel A's access pattern: one 512-thread workgroup per row, four loader
// waves, 16 KB per step, two register sets in flight, a barrier per step, nontemporal loads) against the distance between the
// rows (measurement aid, not product): is the channel-stride effect of profiles/r3_stride_sweep.txt a property of the memory
// system alone, and what does a pure read of this pattern reach?
// build: hipcc --offload-arch=gfx950 -O3 -o stride_pattern stride_pattern.hip ; run: ./stride_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

// TM: tile-major layout [step][row][16 KB] instead of row-major [row][step][16 KB]: ch_f4 is then the number of rows * 1024
// PF > 0: besides the DEPTH register sets, the lines of the tile PF steps ahead are touched by one discarded dword per 128-byte
// line (an L2 prefetch: no registers held)
template <int DEPTH, bool TM = false, int PF = 0, bool NT = true>
__global__ __launch_bounds__(512) void streams(const f4 *src, size_t ch_f4, int steps, float *sink)
{
	extern __shared__ float dyn_lds[];                  // only there to limit the workgroups per CU
	if (steps < 0) sink[1] = dyn_lds[threadIdx.x];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool loader = wave >= 4;
	const int kw = wave - 4;
	constexpr int STEP_F4 = 1024, NLD = 4;
	const f4 *p = TM ? src + (size_t)blockIdx.x * 1024 : src + (size_t)blockIdx.x * ch_f4;
	const size_t step_f4 = TM ? ch_f4 : (size_t)1024;
	f4 acc = {0, 0, 0, 0};
	if (loader) {
		f4 v[DEPTH][NLD];
#pragma unroll
		for (int d = 0; d < DEPTH; d++)
#pragma unroll
			for (int r = 0; r < NLD; r++) v[d][r] = NT ? __builtin_nontemporal_load(p + (size_t)d * step_f4 + 64 * (NLD * kw + r) + lane) : p[(size_t)d * step_f4 + 64 * (NLD * kw + r) + lane];
		for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
			for (int d = 0; d < DEPTH; d++) {
#pragma unroll
				for (int r = 0; r < NLD; r++) acc += v[d][r];
				if (s + d + DEPTH < steps) {
#pragma unroll
					for (int r = 0; r < NLD; r++) v[d][r] = NT ? __builtin_nontemporal_load(p + (size_t)(s + d + DEPTH) * step_f4 + 64 * (NLD * kw + r) + lane) : p[(size_t)(s + d + DEPTH) * step_f4 + 64 * (NLD * kw + r) + lane];
				}
				if (PF > 0 && s + d + PF < steps && lane < 32) {
					// this wave's 4 KB of the tile PF steps ahead: 32 lines of 128 bytes, one dword each
					const float *q = reinterpret_cast<const float *>(p + (size_t)(s + d + PF) * step_f4 + 64 * NLD * kw) + 32 * lane;
					float junk;
					asm volatile("global_load_dword %0, %1, off" : "=v"(junk) : "v"(q) : "memory");
				}
				__syncthreads();
			}
		}
	} else {
		for (int s = 0; s < steps; s++) __syncthreads();
	}
	if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
}

template <class F> float timeit(F f)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 3; i++) f();
	float best = 1e30f, sum = 0.f;
	for (int i = 0; i < 20; i++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; sum += ms; }
	printf("best %.4f ms avg %.4f ms  ", best, sum / 20);
	return sum / 20;
}

int main()
{
	const int C = 1024, steps = 96;
	const size_t row_f4 = (size_t)steps * 1024;                       // 1.5 MiB of payload per row
	const size_t max_stride_f4 = (size_t)8 * 1024 * 1024 / 16;         // up to 8 MiB apart
	f4 *buf; float *sink; hipMalloc(&buf, C * max_stride_f4 * 16); hipMalloc(&sink, 4); hipMemset(buf, 0, C * max_stride_f4 * 16);
	const double gb = (double)C * row_f4 * 16 / 1e9;
	// warm the clocks
	for (int i = 0; i < 200; i++) streams<2><<<C, 512>>>(buf, row_f4, steps, sink);
	hipDeviceSynchronize();
	const int kib[] = {1536, 2048, 2064, 4096, 1536, 2048};
	for (int k : kib) {
		const size_t st = (size_t)k * 1024 / 16;
		printf("stride %5d KiB depth 2: ", k);
		const float ms = timeit([&] { streams<2><<<C, 512>>>(buf, st, steps, sink); });
		printf("%.0f GB/s\n", gb / (ms * 1e-3));
	}
	for (int k : {1536, 2048}) {
		const size_t st = (size_t)k * 1024 / 16;
		printf("stride %5d KiB depth 3: ", k);
		float ms = timeit([&] { streams<3><<<C, 512>>>(buf, st, steps, sink); });
		printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("stride %5d KiB depth 4: ", k);
		ms = timeit([&] { streams<4><<<C, 512>>>(buf, st, steps, sink); });
		printf("%.0f GB/s\n", gb / (ms * 1e-3));
	}
	{
		printf("tile-major [step][row][16 KB] depth 2: "); float ms = timeit([&] { streams<2, true><<<C, 512>>>(buf, (size_t)C * 1024, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("tile-major [step][row][16 KB] depth 3: "); ms = timeit([&] { streams<3, true><<<C, 512>>>(buf, (size_t)C * 1024, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("tile-major [step][row][16 KB] depth 4: "); ms = timeit([&] { streams<4, true><<<C, 512>>>(buf, (size_t)C * 1024, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("row-major 2 MiB depth 2 (again):       "); ms = timeit([&] { streams<2><<<C, 512>>>(buf, (size_t)2048 * 1024 / 16, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
	}
	{
		const size_t st = (size_t)2048 * 1024 / 16;
		float ms;
		printf("2 MiB depth 2 + L2 prefetch 3 ahead: "); ms = timeit([&] { streams<2, false, 3><<<C, 512>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("2 MiB depth 2 + L2 prefetch 4 ahead: "); ms = timeit([&] { streams<2, false, 4><<<C, 512>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("2 MiB depth 2 + L2 prefetch 6 ahead: "); ms = timeit([&] { streams<2, false, 6><<<C, 512>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("2 MiB depth 2 plain loads + L2 prefetch 3 ahead: "); ms = timeit([&] { streams<2, false, 3, false><<<C, 512>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("2 MiB depth 2 plain loads, no prefetch:          "); ms = timeit([&] { streams<2, false, 0, false><<<C, 512>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("2 MiB depth 3 plain loads, no prefetch:          "); ms = timeit([&] { streams<3, false, 0, false><<<C, 512>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		printf("2 MiB depth 2 (again):               "); ms = timeit([&] { streams<2><<<C, 512>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
	}
	// bytes in flight per CU: workgroups per CU (limited by dynamic LDS) x depth x 16 KB; rows 2 MiB apart
	{
		const size_t st = (size_t)2048 * 1024 / 16;
		struct { int wgs; size_t lds; } occ[] = { {4, 39 * 1024}, {3, 52 * 1024}, {2, 79 * 1024} };
		for (auto o : occ) {
			hipFuncSetAttribute((const void *)streams<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o.lds);
			hipFuncSetAttribute((const void *)streams<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o.lds);
			hipFuncSetAttribute((const void *)streams<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o.lds);
			hipFuncSetAttribute((const void *)streams<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)o.lds);
			float ms;
			printf("%d workgroups per CU depth 2 (%3d KB in flight per CU): ", o.wgs, o.wgs * 2 * 16); ms = timeit([&] { streams<2><<<C, 512, o.lds>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
			printf("%d workgroups per CU depth 3 (%3d KB in flight per CU): ", o.wgs, o.wgs * 3 * 16); ms = timeit([&] { streams<3><<<C, 512, o.lds>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
			printf("%d workgroups per CU depth 4 (%3d KB in flight per CU): ", o.wgs, o.wgs * 4 * 16); ms = timeit([&] { streams<4><<<C, 512, o.lds>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
			printf("%d workgroups per CU depth 6 (%3d KB in flight per CU): ", o.wgs, o.wgs * 6 * 16); ms = timeit([&] { streams<6><<<C, 512, o.lds>>>(buf, st, steps, sink); }); printf("%.0f GB/s\n", gb / (ms * 1e-3));
		}
	}
	return 0;
}
