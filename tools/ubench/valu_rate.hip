// valu_rate.hip -- measures issue cost of the VALU instruction kinds kernel A is made of (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_ITERS 4096

template <int KIND>
__global__ void k(float *out, float a, float b)
{
	float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
	f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
	f32x2 pa = {a, a}, pb = {b, b};
	for (int i = 0; i < N_ITERS; i++) {
		if (KIND == 0) {        // 8 independent v_fma_f32
			asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
		} else if (KIND == 1) { // 8 dependent v_fma_f32
			asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
			             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
			             : "+v"(x0) : "v"(a), "v"(b));
		} else if (KIND == 2) { // 4 independent v_pk_fma_f32 (= 8 fma)
			asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
			             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
		} else if (KIND == 3) { // 4 dependent v_pk_fma_f32
			asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n"
			             : "+v"(p0) : "v"(pa), "v"(pb));
		} else if (KIND == 4) { // 8 independent v_max3_f32
			asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
			             "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
		} else if (KIND == 5) { // 8 independent v_bfi_b32
			asm volatile("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n"
			             "v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
		} else if (KIND == 6) { // 8 independent v_mul_f32 (VOP2)
			asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
			             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
		} else if (KIND == 7) { // 4x (v_cmp_gt_f32 e64 -> s_nop -> v_cndmask e64)
			asm volatile("v_cmp_gt_f32 s[20:21], %0, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %5, s[20:21]\n v_cmp_gt_f32 s[20:21], %1, %4\n s_nop 1\n v_cndmask_b32 %1, %1, %5, s[20:21]\n"
			             "v_cmp_gt_f32 s[20:21], %2, %4\n s_nop 1\n v_cndmask_b32 %2, %2, %5, s[20:21]\n v_cmp_gt_f32 s[20:21], %3, %4\n s_nop 1\n v_cndmask_b32 %3, %3, %5, s[20:21]\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "s20", "s21");
		} else if (KIND == 8) { // 8 independent v_pk_mul_f32... as 4 pk_mul + 4 pk_add
			asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
			             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
static void run(const char *name, int insts_per_iter, int waves_per_simd)
{
	float *out;
	const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD; one block per CU per unit
	hipMalloc(&out, (size_t)blocks * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	k<KIND><<<blocks, 256>>>(out, 1.0001f, 0.5f);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<KIND><<<blocks, 256>>>(out, 1.0001f, 0.5f);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double insts_per_simd = (double)N_ITERS * insts_per_iter * waves_per_simd;
	printf("%-28s waves/SIMD=%d  %.3f ms  ns/inst/SIMD=%.3f  (cycles @2.4GHz: %.2f)\n", name, waves_per_simd, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
	hipFree(out);
}

int main()
{
	for (int w : {1, 2, 4, 8}) {
		run<0>("v_fma_f32 indep", 8, w);
		run<1>("v_fma_f32 dependent", 8, w);
		run<2>("v_pk_fma_f32 indep", 4, w);
		run<3>("v_pk_fma_f32 dependent", 4, w);
		run<4>("v_max3_f32 indep", 8, w);
		run<5>("v_bfi_b32 indep", 8, w);
		run<6>("v_mul_f32 (VOP2) indep", 8, w);
		run<7>("cmp+nop+cndmask (x2 inst)", 8, w);
		run<8>("v_pk_mul/add_f32 indep", 4, w);
	}
	return 0;
}
