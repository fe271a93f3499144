// channelizer.hip -- wideband front-end (BASELINE config 4):
//   10 MS/s complex IQ -> 512-bin oversampled polyphase filter bank (decimation 250 -> 40 kS/s per bin)
//   -> per-bin FM discriminator at 40 kS/s -> real rational resampler 6/5 -> 48 kS/s -> kernel A (real input)
// i.e. the reference's ordering  VFO channeliser -> dsp::demod::FM -> RationalResampler -> decoder
// (/root/reference/src/main.cpp:55-60) for all 512 bins at once.  SPEC: DESIGN.md section 3.5; the CPU oracle is
// oracle/or_chan.c.  One object takes S wideband streams per submit (grid.y = stream): the filter bank, the per-bin
// discriminator + resampler and the decoders of all S x 512 bins are one launch each.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "sd_math.h"
#include "sonde_dev.h"
#include "../../include/sonde_abi.h"
#include "launch.h"

#define CH_M 512
#define CH_D 250
#define CH_T 16
#define CH_L (CH_M * CH_T)
#define CH_H (CH_L - CH_D)       // wideband samples of history in front of a block
#define RS_UP 6
#define RS_DN 5
#define RS_TAPS 16

// ---- PFB: one 1024-thread workgroup per PFB_S = 20 consecutive output time steps: a 1.28 M-sample block is 5120 steps =
// 256 workgroups = ONE round on the 256 CUs (the window needs 104 KB of LDS: one workgroup per CU; with 16 steps per
// workgroup the 320 workgroups took two rounds, the second a quarter full).
//   1. the 8192 + 19*250 wideband samples the 20 windows cover are staged in LDS ONCE (a workgroup per step
//      re-read a 64 KB window per 250 new samples: 32.8x through L2);
//   2. fold: thread (r, g) accumulates the 16 taps of bin residue r for the 10 steps of group g, taps in registers,
//      t ascending (SPEC 3.5: v[r] = sum_t fmaf(h[r+512t], x[r+512t], acc));
//   3. circular shift by (m*D mod 512) + bit reversal into a per-step buffer (aliasing the dead window);
//   4. 512-point radix-2 DIT FFT, one WAVE per time step (16 waves: steps 0..15, then waves 0..3 steps 16..19): every lane
//      keeps 8 points in registers and runs three stages on them, three times, with two transposes through the step's own
//      LDS buffer in between -- the butterflies, their operand order and the twiddles are exactly those of the
//      stage-by-stage form (oracle/or_chan.c or_fft512), only the 18 workgroup barriers are gone;
//   5. the 512 x 20 output tile is transposed through LDS so that every bin row receives one aligned 160-byte run
//      (it was an 8-byte store per bin per step, 40 KB apart).
#define PFB_S    20
#define PFB_NW   16                                    // waves per workgroup
#define PFB_NT   (64 * PFB_NW)
#define PFB_WIN  (CH_L + (PFB_S - 1) * CH_D)          // 12942 samples
#define PFB_FB   (CH_M + CH_M / 8)                    // FFT buffer per step: one pad element per 8 (bank spread)
#define PFB_OT   (PFB_S + 1)                          // output tile row stride (float2)
static_assert(PFB_S * PFB_FB <= PFB_WIN && CH_M * PFB_OT <= PFB_WIN, "the FFT buffers and the output tile alias the window");
static_assert(PFB_WIN % 2 == 0 && CH_H % 2 == 0 && (PFB_WIN - CH_H) % 2 == 0 && (CH_D * sizeof(float2)) % 16 == 0, "16-byte staging loads");
static_assert(PFB_S % 4 == 0 && PFB_S <= 2 * PFB_NW && PFB_NT == 2 * CH_M, "fold: two groups of 512 bins; FFT: at most two passes; stores: 16-byte runs");

__device__ __forceinline__ void pfb_bfly(float2 &a, float2 &b, const float2 w)
{
	const float tr = __builtin_fmaf(-b.y, w.y, b.x * w.x);
	const float ti = __builtin_fmaf(b.x, w.y, b.y * w.x);
	const float2 a0 = a;
	a = make_float2(a0.x + tr, a0.y + ti);
	b = make_float2(a0.x - tr, a0.y - ti);
}
__device__ __forceinline__ int pfb_pad(int i) { return i + (i >> 3); }

// 512-point FFT of the bit-reversed, padded buffer fb by one wave; the result stays in registers: e[j] = bin lane + 64 j
__device__ __forceinline__ void pfb_fft512(float2 *fb, const float2 *s_tw, int lane, float2 (&e)[8])
{
	// stages 1-3 on elements 8*lane + j
#pragma unroll
	for (int j = 0; j < 8; j++) e[j] = fb[pfb_pad(8 * lane + j)];
#pragma unroll
	for (int j = 0; j < 8; j += 2) pfb_bfly(e[j], e[j + 1], s_tw[0]);
#pragma unroll
	for (int j = 0; j < 8; j++) if ((j & 2) == 0) pfb_bfly(e[j], e[j + 2], s_tw[(j & 1) * 128]);
#pragma unroll
	for (int j = 0; j < 4; j++) pfb_bfly(e[j], e[j + 4], s_tw[j * 64]);
#pragma unroll
	for (int j = 0; j < 8; j++) fb[pfb_pad(8 * lane + j)] = e[j];
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
	// stages 4-6 on elements lo + 8 j + 64 hi
	{
		const int lo = lane & 7, hi = lane >> 3;
#pragma unroll
		for (int j = 0; j < 8; j++) e[j] = fb[pfb_pad(lo + 8 * j + 64 * hi)];
#pragma unroll
		for (int j = 0; j < 8; j += 2) pfb_bfly(e[j], e[j + 1], s_tw[lo * 32]);
#pragma unroll
		for (int j = 0; j < 8; j++) if ((j & 2) == 0) pfb_bfly(e[j], e[j + 2], s_tw[(lo + 8 * (j & 1)) * 16]);
#pragma unroll
		for (int j = 0; j < 4; j++) pfb_bfly(e[j], e[j + 4], s_tw[(lo + 8 * j) * 8]);
#pragma unroll
		for (int j = 0; j < 8; j++) fb[pfb_pad(lo + 8 * j + 64 * hi)] = e[j];
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
	// stages 7-9 on elements lane + 64 j
#pragma unroll
	for (int j = 0; j < 8; j++) e[j] = fb[pfb_pad(lane + 64 * j)];
#pragma unroll
	for (int j = 0; j < 8; j += 2) pfb_bfly(e[j], e[j + 1], s_tw[lane * 4]);
#pragma unroll
	for (int j = 0; j < 8; j++) if ((j & 2) == 0) pfb_bfly(e[j], e[j + 2], s_tw[(lane + 64 * (j & 1)) * 2]);
#pragma unroll
	for (int j = 0; j < 4; j++) pfb_bfly(e[j], e[j + 4], s_tw[lane + 64 * j]);
}

// The stream in front of the block (the last CH_H samples of the previous submit) comes from hist_in; the last workgroup
// saves this block's tail to hist_out (the other buffer of a ping-pong pair: the first workgroups of this launch still
// read hist_in).  No staging copy of the block and no history roll: a submit is three kernels (was: copy, PFB,
// discriminator + resampler, copy, decoder -- the two copies were 14 of 70 us).
__global__ __launch_bounds__(PFB_NT) void sd_pfb_kernel(const float2 *__restrict__ iq, const float2 *__restrict__ hist_in,
                                                         float2 *__restrict__ hist_out, const float *__restrict__ h,
                                                         const float2 *__restrict__ tw, float2 *__restrict__ bins, uint32_t n_steps)
{
	__shared__ __attribute__((aligned(16))) float2 s_x[PFB_WIN];
	__shared__ float2 s_tw[CH_M / 2];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t m0 = blockIdx.x * PFB_S;
	{	// 1. stage the window (16-byte loads; iq + m0*250 samples is 16-byte aligned, CH_H is even) and the twiddles:
		// window sample w is stream sample m0*250 + w - CH_H of this block, negative = history.
		// all loads first, then all LDS stores: one memory round trip per workgroup instead of one per loop iteration
		const long p0 = (long)m0 * CH_D - CH_H;                       // stream position of window sample 0 (even)
		const float4 *src_iq = reinterpret_cast<const float4 *>(iq) + p0 / 2;       // (p0 < 0: indexed only where p0/2 + i >= 0)
		const float4 *src_h = reinterpret_cast<const float4 *>(hist_in) + (CH_H + p0) / 2;
		float4 *dst = reinterpret_cast<float4 *>(s_x);
		constexpr int NQ = (PFB_WIN / 2 + PFB_NT - 1) / PFB_NT;
		float4 tmp[NQ];
#pragma unroll
		for (int q = 0; q < NQ; q++) {
			const int i = tid + PFB_NT * q;
			tmp[q] = i < PFB_WIN / 2 ? (p0 + 2 * (long)i < 0 ? src_h[i] : src_iq[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		const float2 twv = tid < CH_M / 2 ? tw[tid] : make_float2(0.f, 0.f);
#pragma unroll
		for (int q = 0; q < NQ; q++) {
			const int i = tid + PFB_NT * q;
			if (i < PFB_WIN / 2) dst[i] = tmp[q];
		}
		if (tid < CH_M / 2) s_tw[tid] = twv;
	}
	// 2. fold: r = tid & 511, steps (PFB_S/2) g .. with g = tid >> 9
	constexpr int SPT = PFB_S / 2;          // steps per thread
	const int r = tid & (CH_M - 1), g = tid >> 9;
	float hr[CH_T];
#pragma unroll
	for (int t = 0; t < CH_T; t++) hr[t] = h[r + t * CH_M];
	__syncthreads();
	if (blockIdx.x == gridDim.x - 1) {     // the last CH_H samples of [history | block] are the next submit's history
		const float4 *tail = reinterpret_cast<const float4 *>(s_x + (PFB_WIN - CH_H));
		float4 *ho = reinterpret_cast<float4 *>(hist_out);
		for (int i = tid; i < CH_H / 2; i += PFB_NT) ho[i] = tail[i];
	}
	float2 v[SPT];
#pragma unroll
	for (int q = 0; q < SPT; q++) {
		const float2 *xs = s_x + (SPT * g + q) * CH_D + r;
		float ar = 0.0f, ai = 0.0f;
#pragma unroll
		for (int t = 0; t < CH_T; t++) {
			const float2 xv = xs[t * CH_M];
			ar = __builtin_fmaf(hr[t], xv.x, ar);
			ai = __builtin_fmaf(hr[t], xv.y, ai);
		}
		v[q] = make_float2(ar, ai);
	}
	__syncthreads();                       // the window is dead from here on
	// 3. rotate + bit-reverse into the buffer of the step
#pragma unroll
	for (int q = 0; q < SPT; q++) {
		const int sidx = SPT * g + q;
		const uint32_t shift = ((m0 + (uint32_t)sidx) * CH_D) & (CH_M - 1);
		const uint32_t pos = ((uint32_t)r + shift) & (CH_M - 1);
		const int rev = (int)(__brev(pos) >> 23);               // 9-bit reversal
		s_x[sidx * PFB_FB + pfb_pad(rev)] = v[q];
	}
	__syncthreads();
	// 4. FFT of step m0 + wave (and of step m0 + 16 + wave on the first PFB_S - 16 waves), each by one wave alone
	float2 e0[8], e1[8];
	pfb_fft512(s_x + wave * PFB_FB, s_tw, lane, e0);
	const bool second = wave + PFB_NW < PFB_S;
	if (second) pfb_fft512(s_x + (wave + PFB_NW) * PFB_FB, s_tw, lane, e1);
	__syncthreads();                       // every wave has left its FFT buffers: the output tile aliases them
	// 5. tile[bin][step] (row stride 21), then one 80-byte run per thread: bin = tid >> 1, steps 10*(tid & 1) ..
#pragma unroll
	for (int j = 0; j < 8; j++) s_x[(lane + 64 * j) * PFB_OT + wave] = e0[j];
	if (second) {
#pragma unroll
		for (int j = 0; j < 8; j++) s_x[(lane + 64 * j) * PFB_OT + wave + PFB_NW] = e1[j];
	}
	__syncthreads();
	{
		const int k = tid >> 1, s0 = SPT * (tid & 1);
		float2 o[SPT];
#pragma unroll
		for (int j = 0; j < SPT; j++) o[j] = s_x[k * PFB_OT + s0 + j];
		float4 *dst = reinterpret_cast<float4 *>(bins + (size_t)k * n_steps + m0 + s0);
#pragma unroll
		for (int j = 0; j < SPT / 2; j++) dst[j] = make_float4(o[2 * j].x, o[2 * j].y, o[2 * j + 1].x, o[2 * j + 1].y);
	}
}

// ---- PFB, second form (round 3): 8 steps per 512-thread workgroup, the window (8192 + 7*250 samples = 79.5 KB) leaves room for
// TWO workgroups per CU, blockIdx.y = wideband stream.  The 20-step form above runs one workgroup per CU and all 256 of a
// block in lockstep, so its phases add up: staging (HBM / L2 bound, 5 us), fold (LDS bound: every step reads its 8192
// samples, 4.3 us), FFT (VALU, 3.6 us), stores (5 us) -- 24.7 us measured for a 1.28 M-sample block.  Here the phases of the
// two resident workgroups (and, with several streams or blocks in one launch, of successive generations) overlap; the
// window overlap between neighbouring workgroups (5x instead of 2.6x) is served by the L2.  Arithmetic, operand order and
// tables are those of the 20-step form (bit-identical bins: tests/test_channelizer.py).
#define P8_S    8
#define P8_NT   (64 * P8_S)
#define P8_WIN  (CH_L + (P8_S - 1) * CH_D)            // 9942 samples
#define P8_OT   (P8_S + 1)
static_assert(P8_NT == CH_M, "fold: one thread per bin residue; FFT: one wave per step; stores: one thread per bin");
static_assert(P8_S * PFB_FB + CH_M / 2 <= P8_WIN && CH_M * P8_OT <= P8_S * PFB_FB, "FFT buffers + twiddles / output tile alias the window");
static_assert(2 * P8_WIN * sizeof(float2) <= 160 * 1024, "two workgroups per CU");
static_assert(P8_WIN % 2 == 0 && (P8_S * CH_D) % 2 == 0, "16-byte staging loads");

__global__ __launch_bounds__(P8_NT, 4) void sd_pfb8_kernel(const float2 *__restrict__ iq_all, size_t stream_stride,
                                                            const float2 *__restrict__ hist_in_all, float2 *__restrict__ hist_out_all,
                                                            const float *__restrict__ h, const float2 *__restrict__ tw,
                                                            float2 *__restrict__ bins_all, uint32_t n_steps, uint32_t xcd_map)
{
	__shared__ __attribute__((aligned(16))) float2 s_x[P8_WIN];
	float2 *const s_tw = s_x + P8_S * PFB_FB;                 // written once the window is dead
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// Workgroups go to the 8 XCDs round robin (linear id mod 8; gridDim.x is a multiple of 8): XCD x takes the x-th eighth of the
	// block's step groups, so that the workgroups resident on one XCD are neighbours in time and the 5 x overlap of their
	// windows is served by that XCD's L2 (in launch order neighbours sat on 8 different XCDs and every window came from HBM / MALL)
	// Measured (tools/r3_pfb_xcd.sh): 8 streams 104 -> 97 us, one stream x 4 blocks 57 -> 51.5 us, but one stream x one block
	// (640 workgroups: 1.25 generations) 21.0 -> 22.9 us: the host asks for it from two generations on (xcd_map).
	const uint32_t grp = xcd_map ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
	const uint32_t m0 = grp * P8_S, sidx = blockIdx.y;
	const float2 *iq = iq_all + (size_t)sidx * stream_stride;
	const float2 *hist_in = hist_in_all + (size_t)sidx * CH_H;
	float2 *bins = bins_all + (size_t)sidx * CH_M * n_steps;
	{	// 1. stage the window: all loads first, then all LDS stores
		const long p0 = (long)m0 * CH_D - CH_H;                       // stream position of window sample 0 (even)
		const float4 *src_iq = reinterpret_cast<const float4 *>(iq) + p0 / 2;
		const float4 *src_h = reinterpret_cast<const float4 *>(hist_in) + (CH_H + p0) / 2;
		float4 *dst = reinterpret_cast<float4 *>(s_x);
		constexpr int NQ = (P8_WIN / 2 + P8_NT - 1) / P8_NT;
		float4 tmp[NQ];
#pragma unroll
		for (int q = 0; q < NQ; q++) {
			const int i = tid + P8_NT * q;
#ifdef PFB_AB_NOSTAGE       // A/B builds (profiles/r3_notes.md): what each phase of the kernel costs
			tmp[q] = make_float4((float)i, 0.f, 1.f, 0.f);
#else
			tmp[q] = i < P8_WIN / 2 ? (p0 + 2 * (long)i < 0 ? src_h[i] : src_iq[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
		}
#pragma unroll
		for (int q = 0; q < NQ; q++) {
			const int i = tid + P8_NT * q;
			if (i < P8_WIN / 2) dst[i] = tmp[q];
		}
	}
	const float2 twv = tid < CH_M / 2 ? tw[tid] : make_float2(0.f, 0.f);
	const int r = tid;
	float hr[CH_T];
#pragma unroll
	for (int t = 0; t < CH_T; t++) hr[t] = h[r + t * CH_M];
	__syncthreads();
	if (grp == gridDim.x - 1) {     // the last CH_H samples of [history | block] are the next submit's history
		const float4 *tail = reinterpret_cast<const float4 *>(s_x + (P8_WIN - CH_H));
		float4 *ho = reinterpret_cast<float4 *>(hist_out_all + (size_t)sidx * CH_H);
		for (int i = tid; i < CH_H / 2; i += P8_NT) ho[i] = tail[i];
	}
	// 2. fold (SPEC 3.5: v[r] = sum_t fmaf(h[r+512t], x[r+512t], acc), t ascending)
	float2 v[P8_S];
#pragma unroll
	for (int q = 0; q < P8_S; q++) {
		const float2 *xs = s_x + q * CH_D + r;
		float ar = 0.0f, ai = 0.0f;
#ifdef PFB_AB_NOFOLD
		constexpr int NTAP = 1;
#else
		constexpr int NTAP = CH_T;
#endif
#pragma unroll
		for (int t = 0; t < NTAP; t++) {
			const float2 xv = xs[t * CH_M];
			ar = __builtin_fmaf(hr[t], xv.x, ar);
			ai = __builtin_fmaf(hr[t], xv.y, ai);
		}
		v[q] = make_float2(ar, ai);
	}
	__syncthreads();                       // the window is dead from here on
	// 3. rotate + bit-reverse into the buffer of the step; twiddles next to the buffers
#pragma unroll
	for (int q = 0; q < P8_S; q++) {
		const uint32_t shift = ((m0 + (uint32_t)q) * CH_D) & (CH_M - 1);
		const uint32_t pos = ((uint32_t)r + shift) & (CH_M - 1);
		s_x[q * PFB_FB + pfb_pad((int)(__brev(pos) >> 23))] = v[q];
	}
	if (tid < CH_M / 2) s_tw[tid] = twv;
	__syncthreads();
	// 4. FFT of step m0 + wave by this wave alone
	float2 e0[8];
#ifdef PFB_AB_NOFFT
#pragma unroll
	for (int j = 0; j < 8; j++) e0[j] = s_x[wave * PFB_FB + pfb_pad(lane + 64 * j)];
#else
	pfb_fft512(s_x + wave * PFB_FB, s_tw, lane, e0);
#endif
	__syncthreads();                       // every wave has left its FFT buffer: the output tile aliases them
	// 5. tile[bin][step] (row stride 9), then one 64-byte run per thread (bin = tid)
#pragma unroll
	for (int j = 0; j < 8; j++) s_x[(lane + 64 * j) * P8_OT + wave] = e0[j];
	__syncthreads();
#ifdef PFB_AB_ROWSTORE      // the first form of this phase: thread = bin, four 16-byte stores into its own row: every store
	{                          // instruction touches 64 rows (40 KB apart), 16 bytes each
		float2 o[P8_S];
#pragma unroll
		for (int j = 0; j < P8_S; j++) o[j] = s_x[tid * P8_OT + j];
		float4 *dst = reinterpret_cast<float4 *>(bins + (size_t)tid * n_steps + m0);
#pragma unroll
		for (int j = 0; j < P8_S / 2; j++) dst[j] = make_float4(o[2 * j].x, o[2 * j].y, o[2 * j + 1].x, o[2 * j + 1].y);
	}
#else
	// four lanes per bin row: a store instruction writes 16 whole 64-byte runs
#pragma unroll
	for (int k = 0; k < P8_S / 2; k++) {
		const int idx = tid + P8_NT * k, bin = idx >> 2, part = idx & 3;
		const float2 o0 = s_x[bin * P8_OT + 2 * part], o1 = s_x[bin * P8_OT + 2 * part + 1];
#ifdef PFB_AB_NOSTORE
		if (o0.x == 1.2345e-30f)
#endif
		*reinterpret_cast<float4 *>(bins + (size_t)bin * n_steps + m0 + 2 * part) = make_float4(o0.x, o0.y, o1.x, o1.y);
	}
#endif
}

// ---- PFB, third form (round 3; NOT the default: measured slower than the 8-step form, kept for the record and the A/B):
// one 1024-thread workgroup per THREE consecutive 8-step groups of one stream, wave-specialised:
// waves 0-7 (thread = bin residue) fold group g + 1 out of the staged window while waves 8-15 (wave = time step) run the FFTs
// of group g, transpose them through the FFT buffer and store.  The window of the 24 steps (8192 + 23*250 samples, 111.5 KB)
// is staged once and never overwritten -- the FFT buffer is separate (36.9 KB) -- so the fold, which is bound by LDS
// bandwidth (every step reads its 8192 samples), and the FFT, which is bound by VALU issue, overlap inside the workgroup;
// window samples are re-read 2.3 x (8-step form: 5 x).  Four workgroup barriers per group hand the FFT buffer back and forth.
// Arithmetic, operand order and tables are those of the other two forms (bit-identical bins: tests/test_channelizer.py).
#define PP_G     3                                     // groups per workgroup
#define PP_NT    1024
#define PP_WIN   (CH_L + (PP_G * P8_S - 1) * CH_D)     // 13942 samples
static_assert((PP_WIN + P8_S * PFB_FB + CH_M / 2) * sizeof(float2) <= 160 * 1024, "window + FFT buffer + twiddles in LDS");
static_assert(PP_WIN % 2 == 0, "16-byte staging loads");

__global__ __launch_bounds__(PP_NT) void sd_pfbp_kernel(const float2 *__restrict__ iq_all, size_t stream_stride,
                                                         const float2 *__restrict__ hist_in_all, float2 *__restrict__ hist_out_all,
                                                         const float *__restrict__ h, const float2 *__restrict__ tw,
                                                         float2 *__restrict__ bins_all, uint32_t n_steps)
{
	__shared__ __attribute__((aligned(16))) float2 s_x[PP_WIN];
	__shared__ __attribute__((aligned(16))) float2 s_f[P8_S * PFB_FB];
	__shared__ float2 s_tw[CH_M / 2];
	const int tid = threadIdx.x, lane = tid & 63;
	const bool folder = tid < CH_M;                            // wave-uniform role: waves 0-7 fold, waves 8-15 transform and store
	const int r = tid & (CH_M - 1), fw = (tid >> 6) & 7;      // bin residue (fold) / bin (store); FFT wave = step inside the group
	const uint32_t n_groups = n_steps / P8_S, g0 = blockIdx.x * PP_G, sidx = blockIdx.y;
	const int ng = (int)min((uint32_t)PP_G, n_groups - g0);
	const uint32_t m0 = g0 * P8_S;
	const float2 *iq = iq_all + (size_t)sidx * stream_stride;
	const float2 *hist_in = hist_in_all + (size_t)sidx * CH_H;
	float2 *bins = bins_all + (size_t)sidx * CH_M * n_steps;
	{	// stage the window of the ng groups: all loads first, then all LDS stores
		const long p0 = (long)m0 * CH_D - CH_H;                       // stream position of window sample 0 (even)
		const int need = (CH_L + (ng * P8_S - 1) * CH_D) / 2;         // float4s this workgroup's steps read
		const float4 *src_iq = reinterpret_cast<const float4 *>(iq) + p0 / 2;
		const float4 *src_h = reinterpret_cast<const float4 *>(hist_in) + (CH_H + p0) / 2;
		float4 *dst = reinterpret_cast<float4 *>(s_x);
		constexpr int NQ = (PP_WIN / 2 + PP_NT - 1) / PP_NT;
		float4 tmp[NQ];
#pragma unroll
		for (int q = 0; q < NQ; q++) {
			const int i = tid + PP_NT * q;
			tmp[q] = i < need ? (p0 + 2 * (long)i < 0 ? src_h[i] : src_iq[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
#pragma unroll
		for (int q = 0; q < NQ; q++) {
			const int i = tid + PP_NT * q;
			if (i < need) dst[i] = tmp[q];
		}
	}
	float hr[CH_T];
	if (folder) {
#pragma unroll
		for (int t = 0; t < CH_T; t++) hr[t] = h[r + t * CH_M];
	} else if (tid - CH_M < CH_M / 2) {
		s_tw[tid - CH_M] = tw[tid - CH_M];
	}
	__syncthreads();
	if (blockIdx.x == gridDim.x - 1) {     // the last CH_H samples of [history | block] are the next submit's history
		const float4 *tail = reinterpret_cast<const float4 *>(s_x + ng * P8_S * CH_D);
		float4 *ho = reinterpret_cast<float4 *>(hist_out_all + (size_t)sidx * CH_H);
		for (int i = tid; i < CH_H / 2; i += PP_NT) ho[i] = tail[i];
	}
	// fold of one group (SPEC 3.5: v[r] = sum_t fmaf(h[r+512t], x[r+512t], acc), t ascending)
	float2 v[P8_S];
	auto fold = [&](int g) {
#pragma unroll
		for (int q = 0; q < P8_S; q++) {
			const float2 *xs = s_x + (g * P8_S + q) * CH_D + r;
			float ar = 0.0f, ai = 0.0f;
#pragma unroll
			for (int t = 0; t < CH_T; t++) {
				const float2 xv = xs[t * CH_M];
				ar = __builtin_fmaf(hr[t], xv.x, ar);
				ai = __builtin_fmaf(hr[t], xv.y, ai);
			}
			v[q] = make_float2(ar, ai);
		}
	};
	if (folder) fold(0);
	for (int g = 0; g < ng; g++) {
		__syncthreads();                   // (A) the FFT buffer is free: the store waves have read the previous group's tile
		if (folder) {
#pragma unroll
			for (int q = 0; q < P8_S; q++) {   // rotate + bit-reverse into the buffer of the step
				const uint32_t shift = ((m0 + (uint32_t)(g * P8_S + q)) * CH_D) & (CH_M - 1);
				const uint32_t pos = ((uint32_t)r + shift) & (CH_M - 1);
				s_f[q * PFB_FB + pfb_pad((int)(__brev(pos) >> 23))] = v[q];
			}
		}
		__syncthreads();                   // (B) the buffers hold group g
		float2 e0[8];
		if (folder) {
			if (g + 1 < ng) fold(g + 1);   // overlaps the other waves' FFTs
		} else {
			pfb_fft512(s_f + fw * PFB_FB, s_tw, lane, e0);
		}
		__syncthreads();                   // (C) every FFT wave has left its buffer: the output tile aliases them
		if (!folder) {
#pragma unroll
			for (int j = 0; j < 8; j++) s_f[(lane + 64 * j) * P8_OT + fw] = e0[j];
		}
		__syncthreads();                   // (D) tile[bin][step] complete
		if (!folder) {
			float2 o[P8_S];
#pragma unroll
			for (int j = 0; j < P8_S; j++) o[j] = s_f[r * P8_OT + j];
			float4 *dst = reinterpret_cast<float4 *>(bins + (size_t)r * n_steps + m0 + g * P8_S);
#pragma unroll
			for (int j = 0; j < P8_S / 2; j++) dst[j] = make_float4(o[2 * j].x, o[2 * j].y, o[2 * j + 1].x, o[2 * j + 1].y);
		}
	}
}

// ---- per bin: discriminator at 40 kS/s + 6/5 polyphase resampler to 48 kS/s.  One workgroup per bin.
__global__ __launch_bounds__(256) void sd_disc_resamp_kernel(const float2 *__restrict__ bins, uint32_t n_steps,
                                                              const float *__restrict__ g, float2 *__restrict__ iq_last,
                                                              float *__restrict__ dhist, float *__restrict__ out48)
{
	extern __shared__ float s_d[];            // [RS_TAPS history | n_steps]
	__shared__ float s_g[RS_UP * RS_TAPS];
	const uint32_t k = blockIdx.x;
	const int tid = threadIdx.x;
	const float2 *x = bins + (size_t)k * n_steps;
	if (tid < RS_UP * RS_TAPS) s_g[tid] = g[tid];
	if (tid < RS_TAPS) s_d[tid] = dhist[(size_t)k * RS_TAPS + tid];
	const float2 first_prev = iq_last[k];
	// chunks of 5120 samples: every thread issues its 10 sixteen-byte loads (two samples each) and the 10 predecessor
	// samples before the first result is needed -- one memory round trip per chunk instead of one per sample
	for (uint32_t c0 = 0; c0 < n_steps; c0 += 5120) {
		const float4 *x4 = reinterpret_cast<const float4 *>(x + c0);
		float4 cur[10];
		float2 prv[10];
#pragma unroll
		for (int q = 0; q < 10; q++) {
			const uint32_t i4 = (uint32_t)tid + 256u * q;            // float4 index inside the chunk: samples 2*i4, 2*i4 + 1
			cur[q] = x4[i4];
			prv[q] = (c0 + i4) ? x[c0 + 2 * i4 - 1] : first_prev;
		}
#pragma unroll
		for (int q = 0; q < 10; q++) {
			const uint32_t i = c0 + 2 * ((uint32_t)tid + 256u * q);
			s_d[RS_TAPS + i] = sd_disc(cur[q].x, cur[q].y, prv[q].x, prv[q].y);
			s_d[RS_TAPS + i + 1] = sd_disc(cur[q].z, cur[q].w, cur[q].x, cur[q].y);
		}
	}
	__syncthreads();
	const uint32_t n_out = n_steps * RS_UP / RS_DN;
	for (uint32_t j = tid; j < n_out; j += 256) {
		const uint32_t i0 = (j * RS_DN) / RS_UP, p = (j * RS_DN) % RS_UP;
		float acc = 0.0f;
#pragma unroll
		for (int t = 0; t < RS_TAPS; t++) acc = __builtin_fmaf(s_g[p * RS_TAPS + t], s_d[RS_TAPS + i0 - t], acc);
		out48[(size_t)k * n_out + j] = acc;
	}
	if (tid < RS_TAPS) dhist[(size_t)k * RS_TAPS + tid] = s_d[n_steps + tid];
	if (tid == 0) iq_last[k] = x[n_steps - 1];
}

// ---------------------------------------------------------------- host object
static thread_local std::string g_cerr;
extern "C" const char *sonde_last_error(void);

struct SondeChannelizer {
	int device = 0;
	uint32_t n_steps = 0, n_streams = 1;
	bool fused = false;                    // the decoder kernel takes the bins themselves (discriminator + resampler in its load path): two launches per submit
	SdBinsIn *d_bins_in = nullptr;
	// steps per filter-bank workgroup (SONDE_PFB_FORM): 8 (two workgroups per CU; the default: 8 streams 136 us per submit), 24
	// (wave-specialised, one workgroup per CU: 165 us -- its 111 KB staging burst is not overlapped with anything) or 20 (the
	// round-2 kernel, one stream only); measured in profiles/r3_notes.md
	int pfb_form = 8;
	hipStream_t last_stream = nullptr;     // a submit on another stream waits for the previous one (the state is carried)
	hipEvent_t ev_xs = nullptr;
	// overlapped form (an OPTION: SONDE_CHAN_OVERLAP / sonde_chan_set_overlap(c, 1); fused mode only): the filter bank runs on
	// s_pfb, the decoder on s_dec, the bins are double-buffered, so that the filter bank of submit k+1 may run beside the decoder
	// of submit k.  Measured (profiles/r3_notes.md): no gain at one stream (43.9 -> 48.9 us per block: the two kernels do not
	// co-reside, two filter-bank workgroups take a CU's LDS, and the event hand-overs cost), +1-2 % at 8 streams x 4-8 blocks.
	// The caller's stream only waits for the filter bank (the last reader of its block); results come with sonde_batch_sync.
	uint32_t xcd_map = 0;                  // 8-step filter bank: XCD-aware step-group order (from two generations of workgroups on)
	bool overlap = false;
	hipStream_t s_pfb = nullptr, s_dec = nullptr;
	hipEvent_t ev_in = nullptr, ev_pfb[2] = {}, ev_dec[2] = {};
	float2 *d_bins_b = nullptr;            // the second bins buffer (allocated at the first overlapped submit)
	float2 *d_bins_last = nullptr;         // the bins of the last submit (sonde_chan_read)
	SondeBatch *batch = nullptr;
	float2 *d_hist[2] = {}, *d_bins = nullptr, *d_tw = nullptr, *d_iqlast = nullptr;
	float *d_h = nullptr, *d_g = nullptr, *d_gc = nullptr, *d_dhist = nullptr, *d_out48 = nullptr;
	// kernel timing (HIP events on the submit stream), sampled: every 8th submit
	hipEvent_t ev[3] = {};
	unsigned long n_submits = 0, n_blocks = 0;
	double acc_ms[2] = { 0.0, 0.0 };
	int n_timed = 0;
	bool ev_pending = false;
};

// SPEC 3.5b: the 6/5 resampler and the decoder's boxcar decimator (dec = 2 or 4) as one polyphase filter: row phi = n mod 3 of
// decimated sample n, tap k against d[b(n) - k], b(n) = floor(5 (dec n + dec - 1) / 6): the mean of the dec resampler rows
// involved, each shifted by how much older its newest input is, summed in double (oracle/or_chan.c or_chan_composite_taps)
static void composite_rows(const std::vector<float> &g, int dec, float *G /* 3 * SD_RS_KT_LD */)
{
	for (int phi = 0; phi < 3; phi++) {
		const int newest = (5 * (dec * phi + dec - 1)) / 6;
		for (int k = 0; k < SD_RS_KT_LD; k++) {
			double acc = 0.0;
			for (int j = dec * phi; j < dec * phi + dec; j++) {
				const int t = k - (newest - (5 * j) / 6);
				if (t >= 0 && t < RS_TAPS) acc += (double)g[((5 * j) % RS_UP) * RS_TAPS + t];
			}
			G[phi * SD_RS_KT_LD + k] = (float)(acc / (double)dec);
		}
	}
}

static void make_tables(std::vector<float> &h, std::vector<float> &tw, std::vector<float> &g)
{
	const double PI = 3.14159265358979323846;
	h.resize(CH_L); tw.resize(CH_M); g.resize(RS_UP * RS_TAPS);
	{
		const double fc = 8000.0 / 10000000.0;
		std::vector<double> tmp(CH_L);
		double sum = 0.0;
		for (int i = 0; i < CH_L; i++) {
			const double t = (double)i - 0.5 * (double)(CH_L - 1);
			const double x = (double)i / (double)(CH_L - 1);
			const double w = 0.42 - 0.5 * cos(2.0 * PI * x) + 0.08 * cos(4.0 * PI * x);
			const double s = (t == 0.0) ? 2.0 * fc : sin(2.0 * PI * fc * t) / (PI * t);
			tmp[i] = s * w;
			sum += tmp[i];
		}
		for (int i = 0; i < CH_L; i++) h[i] = (float)(tmp[i] / sum);
	}
	for (int k = 0; k < CH_M / 2; k++) {
		tw[2 * k] = (float)cos(2.0 * PI * (double)k / (double)CH_M);
		tw[2 * k + 1] = (float)(-sin(2.0 * PI * (double)k / (double)CH_M));
	}
	{
		const int N = RS_UP * RS_TAPS;
		const double fc = 18000.0 / 240000.0;
		std::vector<double> tmp(N);
		for (int i = 0; i < N; i++) {
			const double t = (double)i - 0.5 * (double)(N - 1);
			const double x = (double)i / (double)(N - 1);
			const double w = 0.42 - 0.5 * cos(2.0 * PI * x) + 0.08 * cos(4.0 * PI * x);
			const double s = (t == 0.0) ? 2.0 * fc : sin(2.0 * PI * fc * t) / (PI * t);
			tmp[i] = s * w;
		}
		for (int p = 0; p < RS_UP; p++) {
			double sum = 0.0;
			for (int t = 0; t < RS_TAPS; t++) sum += tmp[t * RS_UP + p];
			for (int t = 0; t < RS_TAPS; t++) g[p * RS_TAPS + t] = (float)(tmp[t * RS_UP + p] / sum);
		}
	}
}

extern "C" void sonde_chan_destroy(SondeChannelizer *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	sonde_batch_destroy(c->batch);
	for (int i = 0; i < 3; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
	if (c->ev_xs) (void)hipEventDestroy(c->ev_xs);
	if (c->ev_in) (void)hipEventDestroy(c->ev_in);
	for (int i = 0; i < 2; i++) { if (c->ev_pfb[i]) (void)hipEventDestroy(c->ev_pfb[i]); if (c->ev_dec[i]) (void)hipEventDestroy(c->ev_dec[i]); }
	if (c->s_pfb) (void)hipStreamDestroy(c->s_pfb);
	if (c->s_dec) (void)hipStreamDestroy(c->s_dec);
	(void)hipFree(c->d_bins_b);
	(void)hipFree(c->d_hist[0]); (void)hipFree(c->d_hist[1]); (void)hipFree(c->d_bins); (void)hipFree(c->d_tw); (void)hipFree(c->d_iqlast);
	(void)hipFree(c->d_h); (void)hipFree(c->d_g); (void)hipFree(c->d_dhist); (void)hipFree(c->d_out48); (void)hipFree(c->d_bins_in); (void)hipFree(c->d_gc);
	delete c;
}

extern "C" int sonde_chan_create_multi(const uint8_t *types, uint32_t blocks_per_submit, uint32_t n_streams, int device, SondeChannelizer **out)
{
	// up to 8 blocks per submit when the decoder takes the bins itself; the stand-alone discriminator + resampler kernel stages
	// (16 + 5120 q) floats per bin in LDS: 1-2 blocks
	if (!out || blocks_per_submit == 0 || blocks_per_submit > 8 || n_streams == 0 || n_streams > 64) return -1;
	*out = nullptr;
	SondeChannelizer *c = new SondeChannelizer;
	c->device = device;
	c->n_streams = n_streams;
	c->n_steps = 5120u * blocks_per_submit;                  // 5120 steps = 1.28 M wideband samples = 6144 samples at 48 kS/s
	if (const char *e = getenv("SONDE_PFB_FORM")) c->pfb_form = (atoi(e) == 20 && n_streams == 1) ? 20 : (atoi(e) == 24 ? 24 : 8);
	c->xcd_map = (c->n_steps / P8_S) % 8 == 0 && (size_t)(c->n_steps / P8_S) * n_streams > 1024 && !getenv("SONDE_PFB_NOXCD");
	const uint32_t n_out = c->n_steps * RS_UP / RS_DN;
	const size_t nb = (size_t)n_streams * CH_M;              // bins of all streams: the decoder batch's channels, stream-major
	SondeBatchConfig cfg;
	memset(&cfg, 0, sizeof(cfg));
	cfg.n_channels = (uint32_t)nb;
	cfg.types = types;
	cfg.max_samples = n_out;
	cfg.input_kind = SONDE_INPUT_REAL;
	cfg.device = device;
	if (sonde_batch_create(&cfg, &c->batch) != 0) { delete c; return -1; }
	std::vector<float> h, tw, g;
	make_tables(h, tw, g);
	const size_t hist_bytes = (size_t)n_streams * CH_H * sizeof(float2);
	bool ok = hipMalloc((void **)&c->d_hist[0], hist_bytes) == hipSuccess && hipMalloc((void **)&c->d_hist[1], hist_bytes) == hipSuccess &&
	          hipMalloc((void **)&c->d_bins, nb * c->n_steps * sizeof(float2)) == hipSuccess &&
	          (blocks_per_submit > 2 || hipMalloc((void **)&c->d_out48, nb * n_out * sizeof(float)) == hipSuccess) &&
	          hipMalloc((void **)&c->d_h, CH_L * sizeof(float)) == hipSuccess &&
	          hipMalloc((void **)&c->d_tw, CH_M * sizeof(float)) == hipSuccess &&
	          hipMalloc((void **)&c->d_g, RS_UP * RS_TAPS * sizeof(float)) == hipSuccess &&
	          hipMalloc((void **)&c->d_iqlast, nb * sizeof(float2)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dhist, nb * RS_TAPS * sizeof(float)) == hipSuccess;
	ok = ok && hipMemset(c->d_hist[0], 0, hist_bytes) == hipSuccess && hipMemset(c->d_hist[1], 0, hist_bytes) == hipSuccess && hipMemset(c->d_iqlast, 0, nb * sizeof(float2)) == hipSuccess &&
	     hipMemset(c->d_dhist, 0, nb * RS_TAPS * sizeof(float)) == hipSuccess &&
	     hipMemcpy(c->d_h, h.data(), CH_L * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(c->d_tw, tw.data(), CH_M * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(c->d_g, g.data(), RS_UP * RS_TAPS * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
	for (int i = 0; i < 3 && ok; i++) ok = hipEventCreateWithFlags(&c->ev[i], hipEventDisableSystemFence) == hipSuccess;     // timing only, same device
	ok = ok && hipEventCreateWithFlags(&c->ev_xs, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
	if (ok) {
		float gc[2 * 3 * SD_RS_KT_LD];
		composite_rows(g, 2, gc);
		composite_rows(g, 4, gc + 3 * SD_RS_KT_LD);
		ok = hipMalloc((void **)&c->d_gc, sizeof(gc)) == hipSuccess && hipMemcpy(c->d_gc, gc, sizeof(gc), hipMemcpyHostToDevice) == hipSuccess;
		const SdBinsIn bi = { c->d_gc, reinterpret_cast<float *>(c->d_iqlast), c->d_dhist };
		ok = ok && hipMalloc((void **)&c->d_bins_in, sizeof(bi)) == hipSuccess && hipMemcpy(c->d_bins_in, &bi, sizeof(bi), hipMemcpyHostToDevice) == hipSuccess;
		// fused unless a bin's sonde type needs 48 kS/s rows (AFSK) or the host asks for the rows (sonde_chan_set_fused, SONDE_CHAN_UNFUSED)
		c->fused = sd_batch_bins_capable(c->batch) && !getenv("SONDE_CHAN_UNFUSED");
		if (!c->fused && blocks_per_submit > 2) ok = false;      // (AFSK bins: the three-kernel form, 1-2 blocks per submit)
		c->overlap = c->fused && getenv("SONDE_CHAN_OVERLAP");
	}
	if (!ok) { sonde_chan_destroy(c); return -1; }
	*out = c;
	return 0;
}

extern "C" int sonde_chan_create(const uint8_t *types, uint32_t blocks_per_submit, int device, SondeChannelizer **out)
{
	return sonde_chan_create_multi(types, blocks_per_submit, 1, device, out);
}
extern "C" uint32_t sonde_chan_streams(const SondeChannelizer *c) { return c ? c->n_streams : 0; }
// on = 0: keep the per-bin discriminator + resampler as a kernel of its own, so that the 48 kS/s rows exist (sonde_chan_read;
// parity tests); on = 1 (the default where every bin's sonde type allows it): they run inside the decoder kernel.  Before the
// first submit only.  Returns the mode in force.
extern "C" int sonde_chan_set_fused(SondeChannelizer *c, int on)
{
	if (!c) return -1;
	if (c->n_blocks == 0 && (on || c->d_out48)) c->fused = on && sd_batch_bins_capable(c->batch);      // (> 2 blocks per submit: fused only)
	return c->fused ? 1 : 0;
}

// on = 1: filter bank and decoder on two internal streams, bins double-buffered, so that consecutive submits may overlap
// (fused mode only); on = 0 (the default): both kernels in the caller's stream.  Before the first submit only.  Returns the mode.
extern "C" int sonde_chan_set_overlap(SondeChannelizer *c, int on)
{
	if (!c) return -1;
	if (c->n_blocks == 0) c->overlap = on && c->fused;
	return c->overlap ? 1 : 0;
}

static bool chan_overlap_setup(SondeChannelizer *c)
{
	if (c->s_pfb) return true;
	const size_t nb = (size_t)c->n_streams * CH_M;
	bool ok = hipStreamCreateWithFlags(&c->s_pfb, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c->s_dec, hipStreamNonBlocking) == hipSuccess &&
	          hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess &&
	          hipMalloc((void **)&c->d_bins_b, nb * c->n_steps * sizeof(float2)) == hipSuccess;
	for (int i = 0; i < 2 && ok; i++)
		ok = hipEventCreateWithFlags(&c->ev_pfb[i], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess &&
		     hipEventCreateWithFlags(&c->ev_dec[i], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
	return ok;
}

extern "C" uint32_t sonde_chan_samples_per_submit(const SondeChannelizer *c) { return c ? c->n_steps * CH_D : 0; }
extern "C" SondeBatch *sonde_chan_batch(SondeChannelizer *c) { return c ? c->batch : nullptr; }

extern "C" int sonde_chan_submit(SondeChannelizer *c, const void *iq_dev, size_t n_samples, void *stream_)
{
	if (!c || !iq_dev || n_samples != (size_t)c->n_steps * CH_D) return -1;
	hipStream_t stream = (hipStream_t)stream_;
	if (hipSetDevice(c->device) != hipSuccess) return -1;
	const uint32_t n_out = c->n_steps * RS_UP / RS_DN;
	// fold the previous timed submit's events into the running sums (they have long completed)
	if (c->ev_pending && hipEventQuery(c->ev[2]) == hipSuccess) {
		float a = 0.f, b = 0.f;
		if (hipEventElapsedTime(&a, c->ev[0], c->ev[1]) == hipSuccess && hipEventElapsedTime(&b, c->ev[1], c->ev[2]) == hipSuccess) {
			c->acc_ms[0] += a; c->acc_ms[1] += b; c->n_timed++;
		}
		c->ev_pending = false;
	}
	const bool timed = !c->ev_pending && ((c->n_submits % 8) == 7 || c->n_blocks == 0);      // (as sonde_batch_submit: never the first submit behind a synchronize)
	c->n_submits++;
	if ((uintptr_t)iq_dev & 15u) return -1;       // 16-byte loads straight from the caller's block(s)
	c->overlap = c->overlap && c->fused;
	if (c->overlap) {
		if (!chan_overlap_setup(c)) return -1;
		const int b = (int)(c->n_blocks & 1);
		float2 *bins = b ? c->d_bins_b : c->d_bins;
		// the block is ready where the caller's stream stands now; bins[b] is free once the decoder of two submits ago is done
		if (hipEventRecord(c->ev_in, stream) != hipSuccess || hipStreamWaitEvent(c->s_pfb, c->ev_in, 0) != hipSuccess) return -1;
		if (c->n_blocks >= 2 && hipStreamWaitEvent(c->s_pfb, c->ev_dec[b], 0) != hipSuccess) return -1;
		if (timed) (void)hipEventRecord(c->ev[0], c->s_pfb);
		if (c->pfb_form == 8)
			hipLaunchKernelGGL(sd_pfb8_kernel, dim3(c->n_steps / P8_S, c->n_streams), dim3(P8_NT), 0, c->s_pfb, (const float2 *)iq_dev, n_samples,
			                   c->d_hist[c->n_blocks & 1], c->d_hist[(c->n_blocks + 1) & 1], c->d_h, c->d_tw, bins, c->n_steps, c->xcd_map);
		else if (c->pfb_form == 24)
			hipLaunchKernelGGL(sd_pfbp_kernel, dim3((c->n_steps / P8_S + PP_G - 1) / PP_G, c->n_streams), dim3(PP_NT), 0, c->s_pfb, (const float2 *)iq_dev, n_samples,
			                   c->d_hist[c->n_blocks & 1], c->d_hist[(c->n_blocks + 1) & 1], c->d_h, c->d_tw, bins, c->n_steps);
		else
			hipLaunchKernelGGL(sd_pfb_kernel, dim3(c->n_steps / PFB_S), dim3(PFB_NT), 0, c->s_pfb, (const float2 *)iq_dev, c->d_hist[c->n_blocks & 1],
			                   c->d_hist[(c->n_blocks + 1) & 1], c->d_h, c->d_tw, bins, c->n_steps);
		c->n_blocks++;
		c->d_bins_last = bins;
		if (timed) { (void)hipEventRecord(c->ev[1], c->s_pfb); (void)hipEventRecord(c->ev[2], c->s_pfb); c->ev_pending = true; }
		if (hipEventRecord(c->ev_pfb[b], c->s_pfb) != hipSuccess) return -1;
		// the filter bank is the last reader of the caller's block: work queued on the caller's stream behind this submit may
		// overwrite it; the decoder waits for the bins on its own stream
		if (hipStreamWaitEvent(stream, c->ev_pfb[b], 0) != hipSuccess || hipStreamWaitEvent(c->s_dec, c->ev_pfb[b], 0) != hipSuccess) return -1;
		if (hipGetLastError() != hipSuccess) return -1;
		c->last_stream = stream;
		if (sd_batch_submit_bins(c->batch, bins, c->n_steps, c->n_steps, c->d_bins_in, (void *)c->s_dec) != 0) return -1;
		return hipEventRecord(c->ev_dec[b], c->s_dec) == hipSuccess ? 0 : -1;
	}
	// the front-end kernels carry state too (window history, discriminator history): a submit on another stream waits for
	// the previous one BEFORE the filter bank starts (sonde_batch_submit orders only the decoder behind them)
	if (c->n_blocks && stream != c->last_stream) {
		if (hipEventRecord(c->ev_xs, c->last_stream) != hipSuccess || hipStreamWaitEvent(stream, c->ev_xs, 0) != hipSuccess) return -1;
	}
	c->last_stream = stream;
	if (timed) (void)hipEventRecord(c->ev[0], stream);
	if (c->pfb_form == 20)
		hipLaunchKernelGGL(sd_pfb_kernel, dim3(c->n_steps / PFB_S), dim3(PFB_NT), 0, stream, (const float2 *)iq_dev, c->d_hist[c->n_blocks & 1],
		                   c->d_hist[(c->n_blocks + 1) & 1], c->d_h, c->d_tw, c->d_bins, c->n_steps);
	else if (c->pfb_form == 8)
		hipLaunchKernelGGL(sd_pfb8_kernel, dim3(c->n_steps / P8_S, c->n_streams), dim3(P8_NT), 0, stream, (const float2 *)iq_dev, n_samples,
		                   c->d_hist[c->n_blocks & 1], c->d_hist[(c->n_blocks + 1) & 1], c->d_h, c->d_tw, c->d_bins, c->n_steps, c->xcd_map);
	else
		hipLaunchKernelGGL(sd_pfbp_kernel, dim3((c->n_steps / P8_S + PP_G - 1) / PP_G, c->n_streams), dim3(PP_NT), 0, stream, (const float2 *)iq_dev, n_samples,
		                   c->d_hist[c->n_blocks & 1], c->d_hist[(c->n_blocks + 1) & 1], c->d_h, c->d_tw, c->d_bins, c->n_steps);
	c->n_blocks++;
	c->d_bins_last = c->d_bins;
	if (timed) (void)hipEventRecord(c->ev[1], stream);
	if (!c->fused)
		hipLaunchKernelGGL(sd_disc_resamp_kernel, dim3(CH_M * c->n_streams), dim3(256), (RS_TAPS + c->n_steps) * sizeof(float), stream,
		                   c->d_bins, c->n_steps, c->d_g, c->d_iqlast, c->d_dhist, c->d_out48);
	if (timed) { (void)hipEventRecord(c->ev[2], stream); c->ev_pending = true; }
	if (hipGetLastError() != hipSuccess) return -1;
	if (c->fused) return sd_batch_submit_bins(c->batch, c->d_bins, c->n_steps, c->n_steps, c->d_bins_in, stream_);
	return sonde_batch_submit(c->batch, c->d_out48, n_out, n_out, stream_);
}

// Average device time (ms) of the filter-bank kernel and of the discriminator + resampler kernel over the timed submits
// (every 8th) since the previous call, then of the decoder behind them (sonde_batch_kernel_ms of the embedded batch).
extern "C" int sonde_chan_kernel_ms(SondeChannelizer *c, float *pfb_ms, float *disc_resamp_ms, float *demod_ms, float *framer_ms)
{
	if (!c) return -1;
	if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
	if (c->ev_pending) {
		float a = 0.f, b = 0.f;
		if (hipEventElapsedTime(&a, c->ev[0], c->ev[1]) == hipSuccess && hipEventElapsedTime(&b, c->ev[1], c->ev[2]) == hipSuccess) {
			c->acc_ms[0] += a; c->acc_ms[1] += b; c->n_timed++;
		}
		c->ev_pending = false;
	}
	if (c->n_timed == 0) return -1;
	if (pfb_ms) *pfb_ms = (float)(c->acc_ms[0] / c->n_timed);
	if (disc_resamp_ms) *disc_resamp_ms = (float)(c->acc_ms[1] / c->n_timed);
	c->acc_ms[0] = c->acc_ms[1] = 0.0;
	c->n_timed = 0;
	c->n_submits = 0;
	return sonde_batch_kernel_ms(c->batch, demod_ms, framer_ms);
}

// introspection for the parity tests: copies of the intermediate products of the last submit
extern "C" int sonde_chan_read(SondeChannelizer *c, float *bins /* [512][n_steps][2] or NULL */, float *out48 /* [512][n_out] or NULL */)
{
	if (!c) return -1;
	if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
	const uint32_t n_out = c->n_steps * RS_UP / RS_DN;
	const size_t nb = (size_t)c->n_streams * CH_M;
	if (bins && hipMemcpy(bins, c->d_bins_last ? c->d_bins_last : c->d_bins, nb * c->n_steps * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	if (out48 && c->fused) return -1;      // the rows are never materialised in fused mode: sonde_chan_set_fused(c, 0) before the first submit
	if (out48 && hipMemcpy(out48, c->d_out48, nb * n_out * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return 0;
}

extern "C" int sonde_chan_tables(float *h /* 8192 */, float *tw /* 512 */, float *g /* 96 */)
{
	std::vector<float> vh, vt, vg;
	make_tables(vh, vt, vg);
	if (h) memcpy(h, vh.data(), vh.size() * sizeof(float));
	if (tw) memcpy(tw, vt.data(), vt.size() * sizeof(float));
	if (g) memcpy(g, vg.data(), vg.size() * sizeof(float));
	return 0;
}
