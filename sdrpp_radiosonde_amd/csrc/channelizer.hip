// channelizer.hip -- wideband front-end (BASELINE config 4):
//   10 MS/s complex IQ -> 512-bin polyphase filter bank (decimation 500 -> 20 kS/s per bin, 1.024 x oversampled) -> per-bin
//   instantaneous PHASE (a 16-bit fraction of a turn per bin and step, round 5) -> FM discriminator = wrapped 16-bit phase difference at 20 kS/s
//   -> real rational resampler 12/5 -> 48 kS/s -> kernel A (real input)
// (round 4: the bank ran at 40 kS/s per bin and stored complex bins in rounds 2-3 -- 16.4 B written and re-read per wideband
// sample; a bin now leaves the filter bank as 4 B per 500 wideband samples: 0.8 B per wideband sample)
// i.e. the reference's ordering  VFO channeliser -> dsp::demod::FM -> RationalResampler -> decoder
// (/root/reference/src/main.cpp:55-60) for all 512 bins at once.  SPEC: DESIGN.md section 3.5; the CPU oracle is
// oracle/or_chan.c.  One object takes S wideband streams per submit (grid.y = stream): the filter bank, the per-bin
// discriminator + resampler and the decoders of all S x 512 bins are one launch each.  By default the discriminator and the
// resampler run inside the bins decoder (bins_kernel.hip, one wave per bin): a submit is two launches.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <string>
#include <vector>
#include "sd_math.h"
#include "sonde_dev.h"
#include "../../include/sonde_abi.h"
#include "launch.h"

#define CH_M 512
#define CH_D 500
#define CH_T 16
#define CH_L (CH_M * CH_T)
#define CH_H (CH_L - CH_D)       // wideband samples of history in front of a block
#define RS_UP 12
#define RS_DN 5
#define RS_TAPS 16
#define PH_HEAD 16                  // phases of the previous block kept at the head of every bin's row (bins_kernel.hip)

// ---- the filter bank: 8 output steps per 512-thread workgroup, blockIdx.y = wideband stream.
// Round 5, the commutator form.  Step m0 + q, bin residue r, tap t reads window sample 500 q + r + 512 t (window sample 0 = stream
// position 500 m0 - CH_H).  Rounds 2-4 staged the window (11 692 samples, in two halves of 60.8 KB) in LDS and let thread = r read its
// 16 x 8 operands back: 128 8-byte LDS reads per thread, every window sample read 5.6 times -- the LDS pipe was busy for 29 of the
// kernel's 56 us (and without any global load the kernel still took 48 us: profiles/r5_notes.md).  But 500 = 512 - 12: a thread that owns
// window COLUMN c (the samples c + 512 s, s = 0 .. 22) and, at step q, works for bin r = (c + 12 q) mod 512 needs
//     500 q + r + 512 t  =  c + 512 (q + t)            [ - 512 where c + 12 q >= 512 ]
// i.e. only ITS OWN 23 samples, each used for up to 8 (q, t) pairs: they come from global memory straight into registers, each
// sample of the window once per workgroup, coalesced (consecutive columns, consecutive addresses); no window in LDS, no staging
// rounds, no barrier before the fold.  What varies per (q, t) now is the TAP h[r + 512 t]: the prototype (32 KB) sits in LDS and is read
// as 4-byte operands (half the bytes, and ds_read2st64 pairs).  The sums are the same products added in the same order (t ascending per
// (q, r)): bit for bit SPEC 3.5.
//   1. 23 global loads per thread (all in flight at once, issued and consumed in stream order), the first half of the (symmetric)
//      prototype into LDS meanwhile: 16.5 KB + the FFT buffers = 52.5 KB per workgroup, THREE workgroups per CU, 80 VGPRs;
//   2. fold: columns 0 .. 383 (waves 0-5) never wrap (c + 84 < 512): sample s = q + t; waves 6-7 hold the columns that wrap for some q
//      (there the operand is s = q + t - 1): the same walk plus the two edge terms of every q under a per-lane predicate;
//   3. the step's vector goes to its FFT buffer at (r + 500 (m0 + q)) mod 512 = (c + 500 m0) mod 512: the same rotation for all q;
//   4. 512-point radix-2 DIT FFT, one WAVE per time step, 8 points per lane, two transposes through LDS (pfb_fft512n);
//   5. phase = atan2q(bin) of the 8 points a lane holds, transposed through LDS into [bin][step], one aligned 16-byte store per bin.
// The stream in front of the block (the last CH_H samples of the previous submit) comes from hist_in; the last workgroup of a
// stream copies this block's tail to hist_out (the other buffer of a ping-pong pair: the first workgroups still read hist_in).
#define P_S    8
#define P_NT   (64 * P_S)
#define P_NS   (CH_T + P_S - 1)                        // samples per column: 23
#define P_WIN  (CH_L + (P_S - 1) * CH_D)              // window samples of a workgroup: 11 692
#define P_NOWRAP ((CH_M - (P_S - 1) * (CH_M - CH_D)) / 64)   // waves whose columns never wrap: 6
#define PFB_FB (CH_M + CH_M / 8)                      // FFT buffer per step: one pad element per 8 (bank spread)
#define P_OT   (P_S + 1)                               // phase tile row stride (floats)
static_assert(P_NT == CH_M, "fold: one thread per window column; FFT: one wave per step");
static_assert(CH_M * P_OT <= 2 * P_S * PFB_FB, "the phase tile aliases the FFT buffers");
static_assert(3 * ((CH_L / 2 + 128) * sizeof(float) + P_S * PFB_FB * sizeof(float2)) <= 160 * 1024, "three workgroups per CU");

#ifdef P_TS       // experiment: cycle stamps of one wave's phases (make EXTRA=-DP_TS; tools/pfb_ts.py reads them)
__device__ unsigned long long g_pfb_ts[128];
extern "C" int sonde_debug_pfb_ts(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pfb_ts), sizeof(g_pfb_ts)) == hipSuccess ? 0 : -1; }
#define P_STAMP(i) do { if (blockIdx.x == P_TS_WG && blockIdx.y == 0 && (threadIdx.x & 63) == 0) g_pfb_ts[8 * (i) + (threadIdx.x >> 6)] = __builtin_readcyclecounter(); } while (0)
#ifndef P_TS_WG
#define P_TS_WG 100
#endif
#else
#define P_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void pfb_bfly(float2 &a, float2 &b, const float2 w)
{
	const float tr = __builtin_fmaf(-b.y, w.y, b.x * w.x);
	const float ti = __builtin_fmaf(b.x, w.y, b.y * w.x);
	const float2 a0 = a;
	a = make_float2(a0.x + tr, a0.y + ti);
	b = make_float2(a0.x - tr, a0.y - ti);
}

// 512-point FFT by one wave of a step's folded window in NATURAL order (fb[n], n < 512; the buffer has PFB_FB = 576 elements).
// The arithmetic is the radix-2 decimation-in-time FFT of SPEC 3.5 / oracle or_fft512 butterfly for butterfly -- the same operands
// meet the same twiddles in the same order -- only who holds what differs: three passes of three stages over 8 values per lane,
//   pass 1 (stages 1-3: index bits n8 n7 n6):  lane = n mod 64,                 e[j] = x[lane + 64 brev3(j)]  (stride-1 across lanes)
//   pass 2 (stages 4-6: n5 n4 n3):             lane = (n mod 8) + 8 lo,         lo = the pass-1 register number (= position bits 0-2)
//   pass 3 (stages 7-9: n2 n1 n0):             lane = lo + 8 jj = position mod 64, jj the pass-2 register number; e[j] = bin lane + 64 j
// with two transposes through the step's own buffer whose layouts ([j][72] and [lane][9]) keep every access conflict-free.
// (Round 4 wrote the window bit-reversed -- a 4-way conflicting scatter that cost as much LDS time as the fold, r5_notes.md.)
// The twiddles of passes 2 and 3 depend on the lane alone: tw2 / tw3 hold them in registers (read from global memory at the head
// of the kernel, L1 hits), pass 1's are three wave-uniform values.
// The two TRIVIAL twiddles, exact (SPEC 3.5, round 6): t = b * (1, -0) = b and t = b * (0, -1) = (b.im, -b.re): additions only.  With the
// table's entries 0 and 128 exact these are what pfb_bfly computes, up to the sign of a zero (which cannot change a phase).
__device__ __forceinline__ void pfb_bfly_one(float2 &a, float2 &b)
{
	const float2 a0 = a, b0 = b;
	a = make_float2(a0.x + b0.x, a0.y + b0.y);
	b = make_float2(a0.x - b0.x, a0.y - b0.y);
}
__device__ __forceinline__ void pfb_bfly_mj(float2 &a, float2 &b)
{
	const float2 a0 = a, b0 = b;
	a = make_float2(a0.x + b0.y, a0.y - b0.x);
	b = make_float2(a0.x - b0.y, a0.y + b0.x);
}
struct PfbTw { float2 a2, b2[2], c2[4], a3, b3[2], c3[4]; };
__device__ __forceinline__ PfbTw pfb_load_tw(const float2 *__restrict__ tw, int lane)
{
	PfbTw t;
	const int lo = lane >> 3;
	t.a2 = tw[lo * 32];
#pragma unroll
	for (int k = 0; k < 2; k++) t.b2[k] = tw[(lo + 8 * k) * 16];
#pragma unroll
	for (int k = 0; k < 4; k++) t.c2[k] = tw[(lo + 8 * k) * 8];
	t.a3 = tw[lane * 4];
#pragma unroll
	for (int k = 0; k < 2; k++) t.b3[k] = tw[(lane + 64 * k) * 2];
#pragma unroll
	for (int k = 0; k < 4; k++) t.c3[k] = tw[lane + 64 * k];
	return t;
}
#define PFB_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ void pfb_fft512n(float2 *fb, const float2 w64 /* tw[64] */, const float2 w128 /* tw[128] */, const float2 w192 /* tw[192] */,
                                            const PfbTw &t, int lane, float2 (&e)[8])
{
	constexpr int BR3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
	(void)w128;                                                  // (exact since round 6: pfb_bfly_mj)
	// pass 1
#pragma unroll
	for (int j = 0; j < 8; j++) e[j] = fb[lane + 64 * BR3[j]];
#pragma unroll
	for (int j = 0; j < 8; j += 2) pfb_bfly_one(e[j], e[j + 1]);
#pragma unroll
	for (int j = 0; j < 8; j++) if ((j & 2) == 0) { if (j & 1) pfb_bfly_mj(e[j], e[j + 2]); else pfb_bfly_one(e[j], e[j + 2]); }
	pfb_bfly_one(e[0], e[4]); pfb_bfly(e[1], e[5], w64); pfb_bfly_mj(e[2], e[6]); pfb_bfly(e[3], e[7], w192);
	PFB_WSYNC();
#pragma unroll
	for (int j = 0; j < 8; j++) fb[72 * j + lane] = e[j];
	PFB_WSYNC();
	// pass 2
	{
		const int lo = lane >> 3, hp = lane & 7;
#pragma unroll
		for (int j = 0; j < 8; j++) e[j] = fb[72 * lo + 8 * BR3[j] + hp];
#pragma unroll
		for (int j = 0; j < 8; j += 2) pfb_bfly(e[j], e[j + 1], t.a2);
#pragma unroll
		for (int j = 0; j < 8; j++) if ((j & 2) == 0) pfb_bfly(e[j], e[j + 2], t.b2[j & 1]);
#pragma unroll
		for (int j = 0; j < 4; j++) pfb_bfly(e[j], e[j + 4], t.c2[j]);
		PFB_WSYNC();
#pragma unroll
		for (int j = 0; j < 8; j++) fb[9 * (lo + 8 * j) + hp] = e[j];
	}
	PFB_WSYNC();
	// pass 3
#pragma unroll
	for (int j = 0; j < 8; j++) e[j] = fb[9 * lane + BR3[j]];
#pragma unroll
	for (int j = 0; j < 8; j += 2) pfb_bfly(e[j], e[j + 1], t.a3);
#pragma unroll
	for (int j = 0; j < 8; j++) if ((j & 2) == 0) pfb_bfly(e[j], e[j + 2], t.b3[j & 1]);
#pragma unroll
	for (int j = 0; j < 4; j++) pfb_bfly(e[j], e[j + 4], t.c3[j]);
}

// I16: the wideband stream (and the carried history) as 16-bit integer I, Q pairs -- what a 10 MS/s receiver delivers -- converted on
// the way into LDS (exactly, no scaling: a phase does not see the amplitude); everything behind the window is the float path
// (IK: 0 complex64, 1 int16 pairs, 2 int8 pairs -- a 10 MS/s 8-bit receiver's format)
template <int IK>
__global__ __launch_bounds__(P_NT, 6) void sd_pfb_kernel(const void *__restrict__ iq_all_, size_t stream_stride,
                                                           const void *__restrict__ hist_in_all_, void *__restrict__ hist_out_all_,
                                                           const float *__restrict__ h_even, const float2 *__restrict__ tw,
                                                           int16_t *__restrict__ phi_all, uint32_t n_steps, uint32_t xcd_map,
                                                           uint32_t dual, const float *__restrict__ h_odd, const float2 *__restrict__ twist)
{
	// The prototype is symmetric, h[n] = h[8191 - n] bit for bit (the odd bank's, taps of odd t negated, antisymmetric): its FIRST HALF sits in
	// LDS behind a pad of 128 (the wrapping columns' edge taps are read unconditionally -- a negative index lands in the pad -- and used
	// under a predicate); tap u >= 4096 is read at 8191 - u (descending lanes: conflict-free) and negated for the odd bank.  16.5 KB instead
	// of 32: with the FFT buffers 52.5 KB per workgroup, THREE workgroups per CU (six waves per SIMD).
	__shared__ __attribute__((aligned(16))) float s_h_[CH_L / 2 + 128];
	float *const s_h = s_h_ + 128;
	__shared__ __attribute__((aligned(16))) float2 s_x[P_S * PFB_FB];     // the steps' FFT buffers; later the phase tile
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// Workgroups go to the 8 XCDs round robin (linear id mod 8; gridDim.x is a multiple of 8): XCD x takes the x-th eighth of the
	// block's step groups, so that the workgroups resident on one XCD are neighbours in time and the overlap of their windows is
	// served by that XCD's L2 (round 3: 8 streams 104 -> 97 us; one stream x one block loses: the host asks for it from two
	// generations of workgroups on, xcd_map)
	// xcd_map 2 (round 5; streams a multiple of 8, one bank): XCD x takes STREAMS x, x + 8, ... whole, their step groups in order -- no
	// window straddles two L2s any more (the eighths' 8 x 8 seams cost 3.9 MB of HBM fetches per 8-stream step: 86.3 -> 82.4 MB)
	uint32_t grp, sidx;
	if (xcd_map == 2u) {
		const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y, j = lin >> 3;
		sidx = (lin & 7u) + 8u * (j / gridDim.x);
		grp = j % gridDim.x;
	} else {
		grp = xcd_map ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
		sidx = blockIdx.y;
	}
	// blockIdx.y = LOGICAL stream.  dual (SPEC 3.5c): logical streams 2p and 2p + 1 are the even and the odd-stacked bank (bins centred
	// at k and k + 1/2 bin spacings) of physical stream p: the same samples, taps of odd t negated, a twist behind the fold, the
	// step's common phase taken off the phase samples
	const uint32_t m0 = grp * P_S;
	const bool odd = dual && (sidx & 1u);
	const uint32_t phys = dual ? sidx >> 1 : sidx;
	const float *h = odd ? h_odd : h_even;
	constexpr bool I16 = IK == 1, I8 = IK == 2;
	using ET = typename std::conditional<I8, uint16_t, typename std::conditional<I16, uint32_t, float2>::type>::type;       // one complex sample
	using PT = typename std::conditional<I8, uint32_t, typename std::conditional<I16, uint2, float4>::type>::type;          // a pair of them (the history copy)
	const ET *iq = reinterpret_cast<const ET *>(iq_all_) + (size_t)phys * stream_stride;
	const ET *hist_in = reinterpret_cast<const ET *>(hist_in_all_) + (size_t)phys * CH_H;
	ET *hist_out_all = reinterpret_cast<ET *>(hist_out_all_);
	const size_t prow = (size_t)n_steps + PH_HEAD;                     // a bin's row: [16 carried phases | n_steps]
	int16_t *phi = phi_all + (size_t)sidx * CH_M * prow + PH_HEAD;
	const long p0 = (long)m0 * CH_D - CH_H;                   // stream position of window sample 0
	const int c = tid;                                        // this thread's window column
	P_STAMP(0);
	// 1. the column's 23 samples (window sample c + 512 s exists for c + 512 s < P_WIN: s = 22 only for the columns that do not wrap,
	// the only ones that use it; positions in front of the block come from the carried history) and the prototype, which goes to LDS.
	// Loads return in order: the first 8 samples, then the 16 taps (L2 hits), then the other 15 samples -- the taps are in LDS and the
	// fold starts on the first samples while the rest of the window is still on its way
	ET xr[P_NS];
	auto stage = [&](auto get) {
#pragma unroll
		for (int k = 0; k < 8; k++) { xr[k] = get(k); __builtin_amdgcn_sched_barrier(0); }       // (issued in k order: they return in that order)
		float hreg[CH_T / 2];
#pragma unroll
		for (int t = 0; t < CH_T / 2; t++) hreg[t] = h[c + t * CH_M];
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int k = 8; k < P_NS; k++) { xr[k] = get(k); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
		for (int t = 0; t < CH_T / 2; t++) s_h[c + t * CH_M] = hreg[t];      // (coalesced, conflict-free)
	};
	if (p0 >= 0) {                                            // (workgroup-uniform) all but the first two groups of a block
		const ET *src = iq + p0 + c;
		stage([&](int k) -> ET { return (k < P_NS - 1 || c + CH_M * k < P_WIN) ? src[CH_M * k] : ET{}; });
	} else {
		stage([&](int k) -> ET {
			const long pos = p0 + c + CH_M * k;
			return (k < P_NS - 1 || c + CH_M * k < P_WIN) ? (pos < 0 ? hist_in[CH_H + pos] : iq[pos]) : ET{};
		});
	}
	P_STAMP(1);
	if (grp == gridDim.x - 1 && !odd) {     // the last CH_H samples of the block are the next submit's history (n_steps * 500 >= CH_H)
		const PT *tail = reinterpret_cast<const PT *>(iq + (size_t)n_steps * CH_D - CH_H);
		PT *ho = reinterpret_cast<PT *>(hist_out_all + (size_t)phys * CH_H);
		for (int i = tid; i < CH_H / 2; i += P_NT) ho[i] = tail[i];
	}
	__syncthreads();
	P_STAMP(2);
	// 2. fold (SPEC 3.5: v[r] = sum_t fmaf(h[r+512t], x[r+512t], acc), t ascending)
	auto sample = [&](int k) -> float2 {
		if constexpr (I8) return make_float2((float)(int8_t)(xr[k] & 0xffu), (float)(int8_t)(xr[k] >> 8));
		else if constexpr (I16) return make_float2((float)(int16_t)(xr[k] & 0xffffu), (float)((int32_t)xr[k] >> 16));
		else return xr[k];
	};
	float2 v[P_S];
#pragma unroll
	for (int q = 0; q < P_S; q++) v[q] = make_float2(0.0f, 0.0f);
	// The samples are consumed in the order they arrive (k ascending; a scheduling barrier per sample keeps the compiler from starting with
	// the last ones, which would wait for the whole window); the taps of sample k + 1 are read from LDS before the products of sample k.
	// Tap index of (q, k): u = c + 12 q + 512 d, d = k - q.  d = 0 .. 14: inside [0, 8192) for every column; d = 15 only where
	// c + 12 q < 512, d = -1 only where it is not.  Columns 0 .. 383 (waves 0-5) never wrap: d = 0 .. 15, no predicate; waves 6-7 carry
	// the two edge terms of every q under a per-lane predicate.
	auto fold = [&](auto edge_c, auto odd_c) {
		constexpr bool edge = decltype(edge_c)::value, ODD = decltype(odd_c)::value;
		const float *hq = s_h + c;                              // tap u = c + 12 q + 512 d < 4096 at hq[12 q + 512 d]
		const float *hm = s_h + (CH_L - 1) - c;                 // tap u >= 4096 at s_h[8191 - u] = hm[-(12 q + 512 d)]
		float hk[P_S], he = 0.0f;                               // he: the d = -1 tap (q = k + 1) of the wrapping columns
		auto mir = [&](int off) -> float { const float m = hm[-off]; return ODD ? -m : m; };
		auto taps_of = [&](int k, float (&hh)[P_S], float &hm1) {
#pragma unroll
			for (int q = 0; q < P_S; q++) {
				const int d = k - q, off = (CH_M - CH_D) * q + CH_M * d;
				if (d >= 0 && d < CH_T / 2 - 1) hh[q] = hq[off];
				else if (d == CH_T / 2 - 1) {                       // u = c + 12 q + 3584: the second half for a column that has wrapped
					if (edge) { const float a = hq[off], b = mir(off); hh[q] = c + (CH_M - CH_D) * q < CH_M ? a : b; }
					else hh[q] = hq[off];
				} else if (d >= CH_T / 2 && d < CH_T) hh[q] = mir(off);      // (d = 15 of a wrapping column: index < 0, the pad)
				else hh[q] = 0.0f;
			}
			hm1 = 0.0f;
			if (edge && k + 1 < P_S) hm1 = hq[(CH_M - CH_D) * (k + 1) - CH_M];      // q = k + 1, d = -1 (in front of the table for a column that does not wrap: the pad)
		};
#pragma unroll
		for (int k = 0; k < P_NS; k++) {
			taps_of(k, hk, he);                                  // (no software prefetch of the next sample's taps: at six waves per SIMD the
			const float2 xv = sample(k);                          // other waves cover the LDS latency, and the 9 registers are the difference to spilling)
			if (edge && k + 1 < P_S) {                           // d = -1 comes first in q = k + 1's sum
				const int q = k + 1;
				const bool ok = c + (CH_M - CH_D) * q >= CH_M;
				const float nx = __builtin_fmaf(he, xv.x, v[q].x), ny = __builtin_fmaf(he, xv.y, v[q].y);
				v[q].x = ok ? nx : v[q].x;
				v[q].y = ok ? ny : v[q].y;
			}
#pragma unroll
			for (int q = 0; q < P_S; q++) {
				const int d = k - q;
				if (d >= 0 && d < CH_T - 1) {
					v[q].x = __builtin_fmaf(hk[q], xv.x, v[q].x);
					v[q].y = __builtin_fmaf(hk[q], xv.y, v[q].y);
				} else if (d == CH_T - 1) {
					const float nx = __builtin_fmaf(hk[q], xv.x, v[q].x), ny = __builtin_fmaf(hk[q], xv.y, v[q].y);
					const bool ok = !edge || c + (CH_M - CH_D) * q < CH_M;
					v[q].x = ok ? nx : v[q].x;
					v[q].y = ok ? ny : v[q].y;
				}
			}
#ifndef P_NOSB
			__builtin_amdgcn_sched_barrier(0);
#endif
		}
	};
	if (wave < P_NOWRAP) { if (odd) fold(std::false_type{}, std::true_type{}); else fold(std::false_type{}, std::false_type{}); }
	else { if (odd) fold(std::true_type{}, std::true_type{}); else fold(std::true_type{}, std::false_type{}); }
	P_STAMP(3);
	// the FFT's twiddles: pass 1's three are wave-uniform; passes 2 and 3 per lane (requested now, in flight during the rotation)
	const float2 w64 = tw[64], w128 = tw[128], w192 = tw[192];
	const PfbTw twr = pfb_load_tw(tw, lane);
	if (odd) {                              // the twist W[r] = exp(-j pi r / 512), r = (c + 12 q) mod 512 (wave-uniform branch)
#pragma unroll
		for (int q = 0; q < P_S; q++) {
			const float2 w = twist[(c + (CH_M - CH_D) * q) & (CH_M - 1)];
			const float tr = __builtin_fmaf(-v[q].y, w.y, v[q].x * w.x), ti = __builtin_fmaf(v[q].x, w.y, v[q].y * w.x);
			v[q] = make_float2(tr, ti);
		}
	}
	// 3. rotate into the buffer of the step: (r + 500 (m0 + q)) mod 512 = (c + 500 m0) mod 512 for every q
	{
		const uint32_t pos = ((uint32_t)c + m0 * (uint32_t)CH_D) & (CH_M - 1);
#pragma unroll
		for (int q = 0; q < P_S; q++) s_x[q * PFB_FB + pos] = v[q];      // natural order: consecutive lanes, consecutive addresses
	}
	__syncthreads();
	P_STAMP(4);
	// 4. FFT of step m0 + wave by this wave alone; 5. the phases of the 8 bins this lane holds
	float2 e0[8];
	pfb_fft512n(s_x + wave * PFB_FB, w64, w128, w192, twr, lane, e0);
	P_STAMP(5);
	float ph[8];
#pragma unroll
	for (int j = 0; j < 8; j++) ph[j] = sd_atan2q(e0[j].y, e0[j].x);
	P_STAMP(6);
	__syncthreads();                       // every wave has left its FFT buffer: the phase tile aliases them
	P_STAMP(7);
	float *const s_t = reinterpret_cast<float *>(s_x);
#pragma unroll
	for (int j = 0; j < 8; j++) s_t[(lane + 64 * j) * P_OT + wave] = ph[j];
	__syncthreads();
	P_STAMP(8);
	// SPEC 3.5 (round 5): a phase leaves the bank as a 16-bit fraction of a turn, q = rint(16384 atan2q) mod 2^16; the odd bank takes
	// the step's common phase -(125 / 64) m quadrants = -32000 m (mod 2^16) off in integers (exact; 256 steps = whole turns).
	// Thread = bin: its 8 steps are one aligned 16-byte store.
	{
		const float *row = s_t + tid * P_OT;
		uint32_t q[P_S];
#pragma unroll
		for (int j = 0; j < P_S; j++) {
			q[j] = (uint32_t)__float2int_rn(row[j] * 16384.0f);
			if (odd) q[j] -= 32000u * (m0 + (uint32_t)j);
			q[j] &= 0xffffu;
		}
		*reinterpret_cast<uint4 *>(phi + (size_t)tid * prow + m0) = make_uint4(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16));
	}
	P_STAMP(9);
}

// ---- per bin: discriminator (wrapped difference of consecutive phases) at 20 kS/s + 12/5 polyphase resampler to 48 kS/s, as a
// kernel of its own (bins with AFSK sondes, which want 48 kS/s rows, and the parity tests' unfused mode).  One workgroup per bin.
__global__ __launch_bounds__(256) void sd_phase_resamp_kernel(const int16_t *__restrict__ phi, uint32_t n_steps,
                                                               const float *__restrict__ g, int32_t *__restrict__ phi_last,
                                                               float *__restrict__ dhist, float *__restrict__ out48)
{
	extern __shared__ float s_d[];            // [RS_TAPS history | n_steps]
	__shared__ float s_g[RS_UP * RS_TAPS];
	const uint32_t k = blockIdx.x;
	const int tid = threadIdx.x;
	const int16_t *x = phi + (size_t)k * (n_steps + PH_HEAD) + PH_HEAD;
	if (tid < RS_UP * RS_TAPS) s_g[tid] = g[tid];
	if (tid < RS_TAPS) s_d[tid] = dhist[(size_t)k * RS_TAPS + tid];
	const int first_prev = phi_last[k];
	// the discriminator: a wrapped 16-bit difference, in quadrants (SPEC 3.5)
	for (uint32_t i = tid; i < n_steps; i += 256) s_d[RS_TAPS + i] = (float)(int16_t)(uint16_t)((int)x[i] - (i ? (int)x[i - 1] : first_prev)) * (1.0f / 16384.0f);
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
	if (tid == 0) phi_last[k] = x[n_steps - 1];
}

// ---------------------------------------------------------------- host object
static thread_local std::string g_cerr;
extern "C" const char *sonde_last_error(void);

struct SondeChannelizer {
	int device = 0;
	uint32_t n_steps = 0, n_streams = 1;     // n_streams: LOGICAL streams (grid.y, 512 decoder channels each)
	int input_kind = SONDE_INPUT_IQ;       // what sonde_chan_submit's block holds: complex64 or (sonde_chan_set_input) 16-bit integer IQ
	uint32_t n_phys = 1, dual = 0;            // physical input streams; dual: every physical stream feeds an even and an odd-stacked bank (SPEC 3.5c)
	float *d_h_odd = nullptr; float2 *d_twist = nullptr;
	bool fused = false;                    // the decoder kernel takes the bins themselves (discriminator + resampler in its load path): two launches per submit
	hipStream_t last_stream = nullptr;     // a submit on another stream waits for the previous one (the state is carried)
	hipEvent_t ev_xs = nullptr;
	// overlapped form (an OPTION: sonde_chan_set_overlap(c, 1); fused mode only): the filter bank runs on
	// s_pfb, the decoder on s_dec, the bins are double-buffered, so that the filter bank of submit k+1 may run beside the decoder
	// of submit k.  Measured (profiles/r3_notes.md): no gain at one stream (43.9 -> 48.9 us per block: the two kernels do not
	// co-reside, two filter-bank workgroups take a CU's LDS, and the event hand-overs cost), +1-2 % at 8 streams x 4-8 blocks.
	// The caller's stream only waits for the filter bank (the last reader of its block); results come with sonde_batch_sync.
	uint32_t xcd_map = 0;                  // 8-step filter bank: XCD-aware step-group order (from two generations of workgroups on)
	bool overlap = false;
	hipStream_t s_pfb = nullptr, s_dec = nullptr;
	hipEvent_t ev_in = nullptr, ev_pfb[2] = {}, ev_dec[2] = {};
	int16_t *d_bins_b = nullptr;           // the second phase buffer (allocated at the first overlapped submit)
	int16_t *d_bins_last = nullptr;        // the phases of the last submit (sonde_chan_read)
	SondeBatch *batch = nullptr;
	float2 *d_hist[2] = {}, *d_tw = nullptr;
	int16_t *d_bins = nullptr;             // per bin: [16 phases carried from the previous block | n_steps phases], 16-bit fractions of a turn
	int32_t *d_philast = nullptr;          // unfused mode: per bin, the last phase of the previous block
	float *d_h = nullptr, *d_g = nullptr, *d_gc = nullptr, *d_dhist = nullptr, *d_out48 = nullptr;
	// kernel timing (HIP events on the submit stream), sampled: every 8th submit
	hipEvent_t ev[3] = {};
	unsigned long n_submits = 0, n_blocks = 0;
	double acc_ms[2] = { 0.0, 0.0 };
	int n_timed = 0;
	bool ev_pending = false;
};

// SPEC 3.5b: the 12/5 resampler and the decoder's 4:1 boxcar decimator as one polyphase filter: row phi = n mod 3 of decimated
// sample n, tap k against d[b(n) - k], b(n) = floor(5 (4 n + 3) / 12): the mean of the 4 resampler rows involved, each shifted by
// how much older its newest input is, summed in double (oracle/or_chan.c or_chan_composite_taps); 17 taps in use
static void composite_rows(const std::vector<float> &g, int dec, float *G /* 3 * SD_RS_KT_LD */)
{
	for (int phi = 0; phi < 3; phi++) {
		const int newest = (RS_DN * (dec * phi + dec - 1)) / RS_UP;
		for (int k = 0; k < SD_RS_KT_LD; k++) {
			double acc = 0.0;
			for (int j = dec * phi; j < dec * phi + dec; j++) {
				const int t = k - (newest - (RS_DN * j) / RS_UP);
				if (t >= 0 && t < RS_TAPS) acc += (double)g[((RS_DN * j) % RS_UP) * RS_TAPS + t];
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
	tw[2 * 128] = 0.0f;          // SPEC 3.5 (round 6): exp(-j pi / 2) = (0, -1) exactly (oracle or_chan_twiddles: the same line)
	{
		const int N = RS_UP * RS_TAPS;
		const double fc = 9000.0 / 240000.0;         // 0.45 x the 20 kS/s input rate: the VFO front-end's 20 kS/s taps (vfo.hip)
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
	(void)hipFree(c->d_hist[0]); (void)hipFree(c->d_hist[1]); (void)hipFree(c->d_bins); (void)hipFree(c->d_tw); (void)hipFree(c->d_philast);
	(void)hipFree(c->d_h); (void)hipFree(c->d_g); (void)hipFree(c->d_dhist); (void)hipFree(c->d_out48); (void)hipFree(c->d_gc); (void)hipFree(c->d_h_odd); (void)hipFree(c->d_twist);
	delete c;
}

static int chan_create(const uint8_t *types, uint32_t blocks_per_submit, uint32_t n_phys, uint32_t dual, int device, SondeChannelizer **out);
extern "C" int sonde_chan_create_multi(const uint8_t *types, uint32_t blocks_per_submit, uint32_t n_streams, int device, SondeChannelizer **out)
{
	return chan_create(types, blocks_per_submit, n_streams, 0, device, out);
}
// Both stackings of every stream (SPEC 3.5c): 1024 decoder channels per stream, channel 1024 p + k = even bin k (centre k x 19531.25 Hz),
// 1024 p + 512 + k = odd bin k (centre (k + 1/2) x 19531.25 Hz): every carrier lies within 4.9 kHz of a bin centre.
extern "C" int sonde_chan_create_dual(const uint8_t *types, uint32_t blocks_per_submit, uint32_t n_streams, int device, SondeChannelizer **out)
{
	return chan_create(types, blocks_per_submit, n_streams, 1, device, out);
}
static int chan_create(const uint8_t *types, uint32_t blocks_per_submit, uint32_t n_phys, uint32_t dual, int device, SondeChannelizer **out)
{
	const uint32_t n_streams = n_phys * (dual ? 2u : 1u);
	// up to 8 blocks per submit when the decoder takes the phases itself; the stand-alone discriminator + resampler kernel stages
	// (16 + 2560 q) floats per bin in LDS: 1-2 blocks
	if (!out || blocks_per_submit == 0 || blocks_per_submit > 8 || n_streams == 0 || n_streams > 64) return -1;
	*out = nullptr;
	// an M10 / M20 channel is 50 kHz wide in the reference (/root/reference/src/main.hpp:48): it does not fit a 19.5 kHz bin
	for (size_t i = 0; types && i < (size_t)n_streams * CH_M; i++)
		if (types[i] == SONDE_M10) return sd_fail("sonde_chan_create: an M10 / M20 channel (50 kHz) does not fit a 19.5 kHz channelizer bin; use sonde_vfo_* at 50 kS/s", hipSuccess);
	SondeChannelizer *c = new SondeChannelizer;
	c->device = device;
	c->n_streams = n_streams;
	c->n_phys = n_phys;
	c->dual = dual;
	c->n_steps = 2560u * blocks_per_submit;                  // 2560 steps = 1.28 M wideband samples = 6144 samples at 48 kS/s
	c->xcd_map = (c->n_steps / P_S) % 8 == 0 && (size_t)(c->n_steps / P_S) * n_streams > 1024;
	if (c->xcd_map && !dual && n_streams % 8 == 0) c->xcd_map = 2;      // a stream per XCD (the two banks of a dual object share their samples: they stay together)
	const uint32_t n_out = c->n_steps * RS_UP / RS_DN;
	const size_t nb = (size_t)n_streams * CH_M;              // bins of all streams: the decoder batch's channels, stream-major
	SondeBatchConfig cfg = SONDE_BATCH_CONFIG_INIT;
	cfg.n_channels = (uint32_t)nb;
	cfg.types = types;
	cfg.max_samples = n_out;
	cfg.input_kind = SONDE_INPUT_REAL;
	cfg.device = device;
	cfg.flags = 0;                                           // default completion: the bins decoder is one launch behind the filter bank, in the caller's stream
	if (sonde_batch_create(&cfg, &c->batch) != 0) { delete c; return -1; }
	std::vector<float> h, tw, g;
	make_tables(h, tw, g);
	const size_t hist_bytes = (size_t)n_phys * CH_H * sizeof(float2);
	bool ok = hipMalloc((void **)&c->d_hist[0], hist_bytes) == hipSuccess && hipMalloc((void **)&c->d_hist[1], hist_bytes) == hipSuccess &&
	          hipMalloc((void **)&c->d_bins, nb * ((size_t)c->n_steps + PH_HEAD) * sizeof(int16_t)) == hipSuccess &&
	          (blocks_per_submit > 2 || hipMalloc((void **)&c->d_out48, nb * n_out * sizeof(float)) == hipSuccess) &&
	          hipMalloc((void **)&c->d_h, CH_L * sizeof(float)) == hipSuccess &&
	          hipMalloc((void **)&c->d_tw, CH_M * sizeof(float)) == hipSuccess &&
	          hipMalloc((void **)&c->d_g, RS_UP * RS_TAPS * sizeof(float)) == hipSuccess &&
	          hipMalloc((void **)&c->d_philast, nb * sizeof(int32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dhist, nb * RS_TAPS * sizeof(float)) == hipSuccess;
	ok = ok && hipMemset(c->d_hist[0], 0, hist_bytes) == hipSuccess && hipMemset(c->d_hist[1], 0, hist_bytes) == hipSuccess && hipMemset(c->d_philast, 0, nb * sizeof(int32_t)) == hipSuccess && hipMemset(c->d_bins, 0, nb * ((size_t)c->n_steps + PH_HEAD) * sizeof(int16_t)) == hipSuccess &&
	     hipMemset(c->d_dhist, 0, nb * RS_TAPS * sizeof(float)) == hipSuccess &&
	     [&] { for (int i = 0; i < CH_L / 2; i++) if (memcmp(&h[i], &h[CH_L - 1 - i], sizeof(float))) return false; return true; }() &&      // the kernel keeps half of it
	     hipMemcpy(c->d_h, h.data(), CH_L * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(c->d_tw, tw.data(), CH_M * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(c->d_g, g.data(), RS_UP * RS_TAPS * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
	for (int i = 0; i < 3 && ok; i++) ok = hipEventCreateWithFlags(&c->ev[i], hipEventDisableSystemFence) == hipSuccess;     // timing only, same device
	ok = ok && hipEventCreateWithFlags(&c->ev_xs, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
	if (ok) {
		float gc[3 * SD_RS_KT_LD];
		composite_rows(g, 4, gc);
		for (float &v : gc) v *= 1.0f / 16384.0f;      // the bins decoder keeps the discriminator samples as 16-bit integers: the scale (a power of two: exact) sits in the taps
		ok = hipMalloc((void **)&c->d_gc, sizeof(gc)) == hipSuccess && hipMemcpy(c->d_gc, gc, sizeof(gc), hipMemcpyHostToDevice) == hipSuccess;
		if (ok && dual) {                     // the odd-stacked bank's tables: taps of odd t negated, twist exp(-j pi r / 512)
			const double PI = 3.14159265358979323846;
			std::vector<float> ho(h), wt(2 * CH_M);
			for (int i = 0; i < CH_L; i++) if ((i / CH_M) & 1) ho[i] = -ho[i];
			for (int r = 0; r < CH_M; r++) { wt[2 * r] = (float)cos(PI * (double)r / (double)CH_M); wt[2 * r + 1] = (float)(-sin(PI * (double)r / (double)CH_M)); }
			ok = hipMalloc((void **)&c->d_h_odd, CH_L * sizeof(float)) == hipSuccess && hipMalloc((void **)&c->d_twist, CH_M * sizeof(float2)) == hipSuccess &&
			     hipMemcpy(c->d_h_odd, ho.data(), CH_L * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
			     hipMemcpy(c->d_twist, wt.data(), CH_M * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess;
		}
		// fused unless a bin's sonde type needs 48 kS/s rows (AFSK) or the host asks for the rows (sonde_chan_set_fused)
		c->fused = sd_batch_bins_capable(c->batch);
		if (!c->fused && blocks_per_submit > 2) ok = false;      // (AFSK bins: the three-kernel form, 1-2 blocks per submit)
		c->overlap = false;
	}
	if (!ok) { sonde_chan_destroy(c); return -1; }
	*out = c;
	return 0;
}

extern "C" int sonde_chan_create(const uint8_t *types, uint32_t blocks_per_submit, int device, SondeChannelizer **out)
{
	return sonde_chan_create_multi(types, blocks_per_submit, 1, device, out);
}
extern "C" uint32_t sonde_chan_streams(const SondeChannelizer *c) { return c ? c->n_phys : 0; }          // physical input streams
extern "C" uint32_t sonde_chan_channels(const SondeChannelizer *c) { return c ? c->n_streams * CH_M : 0; }  // decoder channels: 512 (dual: 1024) per stream
// on = 0: keep the per-bin discriminator + resampler as a kernel of its own, so that the 48 kS/s rows exist (sonde_chan_read;
// parity tests); on = 1 (the default where every bin's sonde type allows it): they run inside the decoder kernel.  Before the
// first submit only.  Returns the mode in force.
extern "C" int sonde_chan_set_fused(SondeChannelizer *c, int on)
{
	if (!c) return -1;
	if (on < 0) return c->fused ? 1 : 0;       // query
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
	          hipMalloc((void **)&c->d_bins_b, nb * ((size_t)c->n_steps + PH_HEAD) * sizeof(int16_t)) == hipSuccess &&      // the layout of d_bins: [PH_HEAD carried | n_steps] 16-bit phases
	          hipMemset(c->d_bins_b, 0, nb * ((size_t)c->n_steps + PH_HEAD) * sizeof(int16_t)) == hipSuccess;
	for (int i = 0; i < 2 && ok; i++)
		ok = hipEventCreateWithFlags(&c->ev_pfb[i], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess &&
		     hipEventCreateWithFlags(&c->ev_dec[i], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
	return ok;
}

// What the wideband block holds: SONDE_INPUT_IQ (complex64, the default) or SONDE_INPUT_IQ16 (int16 I, int16 Q: what a 10 MS/s
// receiver delivers; half the bytes to move).  Before the first submit only (the carried window is kept in the input's format).
// Returns the kind in force, or -1.
extern "C" int sonde_chan_set_input(SondeChannelizer *c, int input_kind)
{
	if (!c) return -1;
	if (c->n_blocks == 0 && (input_kind == SONDE_INPUT_IQ || input_kind == SONDE_INPUT_IQ16 || input_kind == SONDE_INPUT_IQ8)) c->input_kind = input_kind;
	return c->input_kind;
}

static void launch_pfb(SondeChannelizer *c, hipStream_t st, const void *iq_dev, size_t n_samples, int16_t *bins)
{
	const dim3 g(c->n_steps / P_S, c->n_streams), blk(P_NT);
	const void *hin = c->d_hist[c->n_blocks & 1];
	void *hout = c->d_hist[(c->n_blocks + 1) & 1];
	if (c->input_kind == SONDE_INPUT_IQ8)
		hipLaunchKernelGGL(sd_pfb_kernel<2>, g, blk, 0, st, iq_dev, n_samples, hin, hout, c->d_h, c->d_tw, bins, c->n_steps, c->xcd_map, c->dual, c->d_h_odd, c->d_twist);
	else if (c->input_kind == SONDE_INPUT_IQ16)
		hipLaunchKernelGGL(sd_pfb_kernel<1>, g, blk, 0, st, iq_dev, n_samples, hin, hout, c->d_h, c->d_tw, bins, c->n_steps, c->xcd_map, c->dual, c->d_h_odd, c->d_twist);
	else
		hipLaunchKernelGGL(sd_pfb_kernel<0>, g, blk, 0, st, iq_dev, n_samples, hin, hout, c->d_h, c->d_tw, bins, c->n_steps, c->xcd_map, c->dual, c->d_h_odd, c->d_twist);
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
		int16_t *bins = b ? c->d_bins_b : c->d_bins;
		int16_t *bins_next = b ? c->d_bins : c->d_bins_b;      // the next submit's buffer takes this submit's last 16 phases
		// the block is ready where the caller's stream stands now; bins[b] is free once the decoder of two submits ago is done
		if (hipEventRecord(c->ev_in, stream) != hipSuccess || hipStreamWaitEvent(c->s_pfb, c->ev_in, 0) != hipSuccess) return -1;
		if (c->n_blocks >= 2 && hipStreamWaitEvent(c->s_pfb, c->ev_dec[b], 0) != hipSuccess) return -1;
		if (timed) (void)hipEventRecord(c->ev[0], c->s_pfb);
		launch_pfb(c, c->s_pfb, iq_dev, n_samples, bins);
		c->n_blocks++;
		c->d_bins_last = bins;
		if (timed) { (void)hipEventRecord(c->ev[1], c->s_pfb); (void)hipEventRecord(c->ev[2], c->s_pfb); c->ev_pending = true; }
		if (hipEventRecord(c->ev_pfb[b], c->s_pfb) != hipSuccess) return -1;
		// the filter bank is the last reader of the caller's block: work queued on the caller's stream behind this submit may
		// overwrite it; the decoder waits for the bins on its own stream
		if (hipStreamWaitEvent(stream, c->ev_pfb[b], 0) != hipSuccess || hipStreamWaitEvent(c->s_dec, c->ev_pfb[b], 0) != hipSuccess) return -1;
		if (hipGetLastError() != hipSuccess) return -1;
		c->last_stream = stream;
		const SdBinsArgs ba = { bins, (size_t)c->n_steps + PH_HEAD, bins_next, (size_t)c->n_steps + PH_HEAD, c->d_gc };
		if (sd_batch_submit_bins(c->batch, &ba, c->n_steps, (void *)c->s_dec) != 0) return -1;
		return hipEventRecord(c->ev_dec[b], c->s_dec) == hipSuccess ? 0 : -1;
	}
	// the front-end kernels carry state too (window history, discriminator history): a submit on another stream waits for
	// the previous one BEFORE the filter bank starts (sonde_batch_submit orders only the decoder behind them)
	if (c->n_blocks && stream != c->last_stream) {
		if (hipEventRecord(c->ev_xs, c->last_stream) != hipSuccess || hipStreamWaitEvent(stream, c->ev_xs, 0) != hipSuccess) return -1;
	}
	c->last_stream = stream;
	if (timed) (void)hipEventRecord(c->ev[0], stream);
	launch_pfb(c, stream, iq_dev, n_samples, c->d_bins);
	c->n_blocks++;
	c->d_bins_last = c->d_bins;
	if (timed) (void)hipEventRecord(c->ev[1], stream);
	if (!c->fused)
		hipLaunchKernelGGL(sd_phase_resamp_kernel, dim3(CH_M * c->n_streams), dim3(256), (RS_TAPS + c->n_steps) * sizeof(float), stream,
		                   c->d_bins, c->n_steps, c->d_g, c->d_philast, c->d_dhist, c->d_out48);
	if (timed) { (void)hipEventRecord(c->ev[2], stream); c->ev_pending = true; }
	if (hipGetLastError() != hipSuccess) return -1;
	if (c->fused) {
		const SdBinsArgs ba = { c->d_bins, (size_t)c->n_steps + PH_HEAD, c->d_bins, (size_t)c->n_steps + PH_HEAD, c->d_gc };
		return sd_batch_submit_bins(c->batch, &ba, c->n_steps, stream_);
	}
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
extern "C" int sonde_chan_read(SondeChannelizer *c, float *bins /* [512][n_steps] phases (quadrants) or NULL */, float *out48 /* [512][n_out] or NULL */)
{
	if (!c) return -1;
	if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
	const uint32_t n_out = c->n_steps * RS_UP / RS_DN;
	const size_t nb = (size_t)c->n_streams * CH_M;
	if (bins) {           // the 16-bit phases as quadrants (q / 16384: exact), without the carried head of each row
		const size_t prow = (size_t)c->n_steps + PH_HEAD;
		std::vector<int16_t> tmp(nb * prow);
		if (hipMemcpy(tmp.data(), c->d_bins_last ? c->d_bins_last : c->d_bins, tmp.size() * sizeof(int16_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
		for (size_t k = 0; k < nb; k++)
			for (size_t m = 0; m < c->n_steps; m++) bins[k * c->n_steps + m] = (float)tmp[k * prow + PH_HEAD + m] * (1.0f / 16384.0f);
	}
	if (out48 && c->fused) return -1;      // the rows are never materialised in fused mode: sonde_chan_set_fused(c, 0) before the first submit
	if (out48 && hipMemcpy(out48, c->d_out48, nb * n_out * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return 0;
}

extern "C" int sonde_chan_tables(float *h /* 8192 */, float *tw /* 512 */, float *g /* 192 */)
{
	std::vector<float> vh, vt, vg;
	make_tables(vh, vt, vg);
	if (h) memcpy(h, vh.data(), vh.size() * sizeof(float));
	if (tw) memcpy(tw, vt.data(), vt.size() * sizeof(float));
	if (g) memcpy(g, vg.data(), vg.size() * sizeof(float));
	return 0;
}
