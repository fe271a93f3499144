// swar_rate.hip -- issue cost of the byte-sliced GF(2^8) multiply step of sd_rsdec.h (v_perm_b32 based) on gfx950:
// independent v_perm_b32, and the dependent step chain itself, at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o ab/swar_rate tools/ubench/swar_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N_ITERS 4096

template <int KIND>
__global__ void k(uint32_t *out, uint32_t a, uint32_t b)
{
	uint32_t x0 = threadIdx.x * 2654435761u, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	const uint32_t ta = a * 3 + threadIdx.x, tb = b * 5 + threadIdx.x, tc = a ^ b ^ threadIdx.x, td = a + 77u * threadIdx.x, te = b + 13u;
	for (int i = 0; i < N_ITERS; i++) {
		if (KIND == 0) {        // 8 independent v_perm_b32
			asm volatile("v_perm_b32 %0, %8, %9, %0\n v_perm_b32 %1, %8, %9, %1\n v_perm_b32 %2, %8, %9, %2\n v_perm_b32 %3, %8, %9, %3\n"
			             "v_perm_b32 %4, %8, %9, %4\n v_perm_b32 %5, %8, %9, %5\n v_perm_b32 %6, %8, %9, %6\n v_perm_b32 %7, %8, %9, %7\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
		} else if (KIND == 1) { // 8 independent v_and_b32 (VOP2)
			asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
			             "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
		} else if (KIND == 2) { // the dependent SWAR step, 4 times (C++, as sd_rsdec.h writes it)
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const uint32_t ia = x0 & 0x07070707u, ib = (x0 >> 3) & 0x07070707u, ic = (x0 >> 6) & 0x03030303u;
				x0 = __builtin_amdgcn_perm(tb, ta, ia) ^ __builtin_amdgcn_perm(td, tc, ib) ^ __builtin_amdgcn_perm(te, te, ic) ^ (x1 + q);
			}
		} else if (KIND == 3) { // two independent SWAR chains, 4 steps each
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const uint32_t ia = x0 & 0x07070707u, ib = (x0 >> 3) & 0x07070707u, ic = (x0 >> 6) & 0x03030303u;
				x0 = __builtin_amdgcn_perm(tb, ta, ia) ^ __builtin_amdgcn_perm(td, tc, ib) ^ __builtin_amdgcn_perm(te, te, ic) ^ (x1 + q);
				const uint32_t ja = x2 & 0x07070707u, jb = (x2 >> 3) & 0x07070707u, jc = (x2 >> 6) & 0x03030303u;
				x2 = __builtin_amdgcn_perm(tb, ta, ja) ^ __builtin_amdgcn_perm(td, tc, jb) ^ __builtin_amdgcn_perm(te, te, jc) ^ (x3 + q);
			}
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
static void run(const char *name, int units_per_iter, int waves_per_simd)
{
	uint32_t *out;
	const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD; one block per CU per unit
	hipMalloc(&out, (size_t)blocks * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	k<KIND><<<blocks, 256>>>(out, 0x04050607u, 0x00010203u);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<KIND><<<blocks, 256>>>(out, 0x04050607u, 0x00010203u);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double units_per_simd = (double)N_ITERS * units_per_iter * waves_per_simd;
	printf("%-34s waves/SIMD=%d  %.3f ms  cycles per unit per SIMD @2.4GHz: %.2f\n", name, waves_per_simd, ms, ms * 1e6 / units_per_simd * 2.4);
	hipFree(out);
}

int main()
{
	for (int w : {1, 2, 4, 8}) {
		run<0>("v_perm_b32 indep (unit = inst)", 8, w);
		run<1>("v_xor_b32 indep (unit = inst)", 8, w);
		run<2>("SWAR step, 1 chain (unit = step)", 4, w);
		run<3>("SWAR step, 2 chains (unit = step)", 8, w);
	}
	return 0;
}
