// permlane_probe.hip -- what v_permlane32_swap_b32 does on gfx950 (round 5: the half-wave symbol mapping of the 2.5-samples-per-symbol
// classes needs "lane l < 32 gets lane 32 + l - 1, lane l >= 32 gets lane l - 32" without the LDS pipe)
// build: hipcc --offload-arch=gfx950 -O2 -o permlane_probe permlane_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
	unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
	auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
	out[threadIdx.x] = r[0];
	out[64 + threadIdx.x] = r[1];
}
int main() {
	unsigned *d, h[128];
	hipMalloc(&d, sizeof(h));
	k<<<1, 64>>>(d);
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	printf("r0:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[i]); printf(" [63]=%u\n", h[63]);
	printf("r1:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 + i]); printf(" [63]=%u\n", h[127]);
	return 0;
}
