// hybrid_depth.hip -- 1024 row streams, four 512-thread workgroups per CU (39 KB of LDS each), 16 KB per step, a barrier per step:
// two register sets (kernel A's form), three register sets (what 64 VGPRs cannot hold), and the HYBRID: two register sets + one
// direct-to-LDS set (global_load_lds_dwordx4 into a 16 KB staging area, read back with ds_read_b128) -- does a third tile in flight
// that bypasses the registers buy what the third register set buys?  (Measurement aid, not product.)
// build: hipcc --offload-arch=gfx950 -O3 -o hybrid_depth hybrid_depth.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: three register sets, the third loaded when it is needed (a burst: waits for all three); 1: three register sets, each re-requested
// when consumed; 2: two register sets + one LDS-direct set (tiles 3j + 2); 3: two register sets (kernel A's schedule)
template <int MODE>
__global__ __launch_bounds__(512) void streams(const f4 *src, size_t ch_f4, int steps, float *sink)
{
	extern __shared__ __attribute__((aligned(16))) f4 stage[];            // [1024] staging (MODE 2) + padding up to 39 KB
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool loader = wave >= 4;
	const int kw = __builtin_amdgcn_readfirstlane(wave - 4);
	const f4 *p = src + (size_t)blockIdx.x * ch_f4;
	f4 acc = {0, 0, 0, 0};
	if (steps < 0) sink[1] = stage[tid].x;
	if (loader) {
		f4 va[4], vb[4], vc[4];
		auto ld = [&](int step, f4 (&v)[4]) {
#pragma unroll
			for (int r = 0; r < 4; r++) v[r] = __builtin_nontemporal_load(p + (size_t)step * 1024 + 64 * (4 * kw + r) + lane);
		};
		auto ld_lds = [&](int step) {
#pragma unroll
			for (int r = 0; r < 4; r++)
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + (size_t)step * 1024 + 64 * (4 * kw + r) + lane),
					(__attribute__((address_space(3))) void *)(&stage[64 * (4 * kw + r)]), 16, 0, 0);
		};
		auto use = [&](const f4 (&v)[4]) {
#pragma unroll
			for (int r = 0; r < 4; r++) acc += v[r];
		};
		ld(0, va); ld(1, vb);
		if (MODE == 3) {
			for (int s = 0; s < steps; s += 2) {
				use(va); if (s + 2 < steps) ld(s + 2, va);
				__syncthreads();
				if (s + 1 < steps) { use(vb); if (s + 3 < steps) ld(s + 3, vb); }
				__syncthreads();
			}
			if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
			return;
		}
		if (MODE == 4) {
			// kernel A's schedule with every load of the steady state UNCONDITIONAL (tail peeled): the compiler then knows that four newer
			// loads are in flight behind the set it waits for (vmcnt(7..4)); with `if (s + 2 < steps) ld(...)` inside the loop it must
			// assume they may not have been issued and waits for vmcnt(3..0): for everything, the set requested last included
			int s = 0;
			for (; s + 4 <= steps - 2; s += 2) {
				use(va); ld(s + 2, va);
				__syncthreads();
				use(vb); ld(s + 3, vb);
				__syncthreads();
			}
			for (; s < steps; s += 2) {
				use(va); if (s + 2 < steps) ld(s + 2, va);
				__syncthreads();
				if (s + 1 < steps) { use(vb); if (s + 3 < steps) ld(s + 3, vb); }
				__syncthreads();
			}
			if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
			return;
		}
		if (MODE == 1) ld(2, vc);
		if (MODE == 2) ld_lds(2);
		for (int s = 0; s < steps; s += 3) {
			// tile s (set A)
			use(va);                                   // (the compiler's vmcnt: everything older than the newest 4 / 8 loads)
			if (s + 3 < steps) ld(s + 3, va);
			__syncthreads();
			// tile s + 1 (set B)
			if (s + 1 < steps) { use(vb); if (s + 4 < steps) ld(s + 4, vb); }
			__syncthreads();
			// tile s + 2 (set C: registers, LDS, or -- MODE 0 -- loaded only now)
			if (s + 2 < steps) {
				if (MODE == 0) { ld(s + 2, vc); use(vc); }
				else if (MODE == 1) { use(vc); if (s + 5 < steps) ld(s + 5, vc); }
				else {
					// wait until the staged tile has landed (it is the oldest of the loads in flight: two register sets issued after it)
					asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
					for (int r = 0; r < 4; r++) acc += stage[64 * (4 * kw + r) + lane];
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					if (s + 5 < steps) ld_lds(s + 5);
				}
			}
			__syncthreads();
		}
	} else if (MODE == 3 || MODE == 4) {
		for (int s = 0; s < steps; s += 2) { __syncthreads(); __syncthreads(); }
	} else {
		for (int s = 0; s < steps; s += 3) { __syncthreads(); __syncthreads(); __syncthreads(); }
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
	const size_t stride_f4 = (size_t)2048 * 1024 / 16;
	f4 *buf; float *sink; hipMalloc(&buf, C * stride_f4 * 16); hipMalloc(&sink, 8); hipMemset(buf, 0, C * stride_f4 * 16);
	const double gb = (double)C * steps * 16384 / 1e9;
	const size_t lds = 39 * 1024;
	for (int i = 0; i < 300; i++) streams<1><<<C, 512, lds>>>(buf, stride_f4, steps, sink);
	hipDeviceSynchronize();
	for (int rep = 0; rep < 3; rep++) {
		float ms;
		ms = timeit([&] { streams<3><<<C, 512, lds>>>(buf, stride_f4, steps, sink); }); printf("two register sets (kernel A)                 %.4f ms %.0f GB/s\n", ms, gb / (ms * 1e-3));
		ms = timeit([&] { streams<4><<<C, 512, lds>>>(buf, stride_f4, steps, sink); }); printf("two register sets, unconditional loads        %.4f ms %.0f GB/s\n", ms, gb / (ms * 1e-3));
		ms = timeit([&] { streams<0><<<C, 512, lds>>>(buf, stride_f4, steps, sink); }); printf("three register sets, third loaded late       %.4f ms %.0f GB/s\n", ms, gb / (ms * 1e-3));
		ms = timeit([&] { streams<1><<<C, 512, lds>>>(buf, stride_f4, steps, sink); }); printf("three register sets                          %.4f ms %.0f GB/s\n", ms, gb / (ms * 1e-3));
		ms = timeit([&] { streams<2><<<C, 512, lds>>>(buf, stride_f4, steps, sink); }); printf("two register sets + one LDS-direct set       %.4f ms %.0f GB/s\n", ms, gb / (ms * 1e-3));
	}
	printf("status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
	return 0;
}
