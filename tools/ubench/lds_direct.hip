// lds_direct.hip -- can direct-to-LDS loads (global_load_lds_dwordx4, gfx950) keep more bytes in flight than the
// register sets kernel A can afford (64 VGPRs)?  Same access pattern as stream_pattern.hip: 1024 channel streams,
// 512-thread workgroups with 4 loader waves, one s_barrier per 16 KB step; the data goes HBM -> LDS ring of DEPTH
// slots without touching VGPRs and is read back with ds_read_b128.  (Measurement aid, not product.)
// build: hipcc --offload-arch=gfx950 -O3 -o lds_direct lds_direct.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool BAR>
__global__ __launch_bounds__(512) void streams_lds(const f4 *src, size_t ch_f4, int steps, float *sink)
{
	extern __shared__ __attribute__((aligned(16))) f4 ring[];            // DEPTH slots of 1024 f4 (16 KB)
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool loader = wave >= 4;
	const int kw = __builtin_amdgcn_readfirstlane(wave - 4);
	const f4 *p = src + (size_t)blockIdx.x * ch_f4;
	f4 acc = {0, 0, 0, 0};
	if (loader) {
		auto issue = [&](int step) {
			const int slot = step % DEPTH;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const f4 *g = p + (size_t)step * 1024 + 64 * (4 * kw + r) + lane;
				// LDS destination: M0 holds the wave-uniform base (bytes); the hardware adds lane * 16
				// the builtin takes the wave-uniform LDS base; the hardware adds lane * 16 itself
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
					(__attribute__((address_space(3))) void *)(&ring[slot * 1024 + 64 * (4 * kw + r)]), 16, 0, 0);
			}
		};
		for (int d = 0; d < DEPTH - 1 && d < steps; d++) issue(d);
		for (int s = 0; s < steps; s++) {
			if (s + DEPTH - 1 < steps) issue(s + DEPTH - 1);
			// wait until the loads of step s have landed: at most 4 * (DEPTH - 1) newer ones may stay in flight
			if (s + DEPTH - 1 < steps) {
				if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
				if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
				if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
				if (DEPTH == 6) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
			} else {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			const int slot = s % DEPTH;
#pragma unroll
			for (int r = 0; r < 4; r++) acc += ring[slot * 1024 + 64 * (4 * kw + r) + lane];
			if (BAR) __syncthreads();
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
	f4 *buf; float *sink; hipMalloc(&buf, n16 * 16); hipMalloc(&sink, 4);
	// data = index pattern so that a wrong LDS mapping shows up in the checksum kernel below
	hipMemset(buf, 0, n16 * 16);
	const double gb = n16 * 16 / 1e9;
	auto rep = [&](const char *name, float ms) { printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, gb / (ms * 1e-3)); };
	rep("lds-direct depth2 bar (32 KB LDS/WG)", timeit([&] { streams_lds<2, true><<<C, 512, 2 * 16384>>>(buf, ch_f4, steps, sink); }));
	rep("lds-direct depth2 nobar", timeit([&] { streams_lds<2, false><<<C, 512, 2 * 16384>>>(buf, ch_f4, steps, sink); }));
	hipFuncSetAttribute((const void *)streams_lds<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384);
	rep("lds-direct depth3 bar (48 KB: 3 WG/CU)", timeit([&] { streams_lds<3, true><<<C, 512, 3 * 16384>>>(buf, ch_f4, steps, sink); }));
	hipError_t e = hipDeviceSynchronize();
	printf("status: %s\n", hipGetErrorString(e));
	return 0;
}
