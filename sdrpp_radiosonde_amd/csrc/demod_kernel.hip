// demod_kernel.hip -- kernel A: IQ (or discriminator samples) -> hard bits, one workgroup per channel.
//
// Stages, all inside one launch so that IQ is read from HBM exactly once and only bits
// (1/640 of the input bytes) are written:
//   K1  FM quadrature discriminator      (SDR++ dsp::demod::FM<float>, /root/reference/src/main.cpp:57)
//   K2  polyphase low-pass FIR, evaluated only at the timing loop's sampling instants
//   K3  Gardner timing-error detector + PI loop filter, updated once per round (<= 256 symbols)
//       and hard slicer -> bit ring in HBM
// (K2/K3 stand where sondedump's gfsk_demod sits behind X_decode, /root/reference/src/decode/decoder.hpp:22,61.)
//
// Bit-exactness contract: compiled with -ffp-contract=off, every fused op is an
// explicit __builtin_fmaf; each lane's FIR is a fixed-order fmaf chain; every cross-lane
// reduction (timing error, slicer level statistics) is an integer sum, so lane order is irrelevant.
#include <hip/hip_runtime.h>
#include "sonde_dev.h"

#define PI_F        3.14159274f
#define TWO_PI_F    6.28318548f
#define HALF_PI_F   1.57079637f
#define TWO_OVER_PI 0.636619747f
// Abramowitz & Stegun 4.4.47
#define AT_A1  0.9998660f
#define AT_A3 -0.3302995f
#define AT_A5  0.1801410f
#define AT_A7 -0.0851330f
#define AT_A9  0.0208351f

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Reciprocal by Newton-Raphson from an integer-subtract seed (SPEC 3.1): 1 integer op + 6 fma,
// identical sequence in the oracle.  Two lanes' worth at a time so the fmas issue as v_pk_fma_f32.
__device__ __forceinline__ f32x2 sd_recip2(f32x2 x)
{
	f32x2 r;
	r.x = __uint_as_float(0x7EF311C7u - __float_as_uint(x.x));
	r.y = __uint_as_float(0x7EF311C7u - __float_as_uint(x.y));
	const f32x2 one = {1.0f, 1.0f};
	f32x2 e = pk_fma(-x, r, one);
	r = pk_fma(r, e, r);
	e = pk_fma(-x, r, one);
	r = pk_fma(r, e, r);
	e = pk_fma(-x, r, one);
	r = pk_fma(r, e, r);
	return r;
}
__device__ __forceinline__ float sd_recip(float x)
{
	float r = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
	float e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(r, e, r);
	e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(r, e, r);
	e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(r, e, r);
	return r;
}

#define TINY_BITS 0x0DA24260u   // 1e-30f: floor of the divisor, so that atan2p(0,0) = 0 without a select

// atan2p of two samples (y0,x0), (y1,x1): SPEC 3.1.  max/min on the bit patterns, A&S 4.4.47
// polynomial of min/max (packed), octant fix-ups as |offset - p| and a final copysign.
__device__ __forceinline__ f32x2 sd_atan2x2(float y0, float x0, float y1, float x1)
{
	const uint32_t ax0 = __float_as_uint(x0) & 0x7FFFFFFFu, ay0 = __float_as_uint(y0) & 0x7FFFFFFFu;
	const uint32_t ax1 = __float_as_uint(x1) & 0x7FFFFFFFu, ay1 = __float_as_uint(y1) & 0x7FFFFFFFu;
	const f32x2 mx = {__uint_as_float(max(max(ax0, ay0), TINY_BITS)), __uint_as_float(max(max(ax1, ay1), TINY_BITS))};
	const f32x2 mn = {__uint_as_float(min(ax0, ay0)), __uint_as_float(min(ax1, ay1))};
	const f32x2 r = mn * sd_recip2(mx);
	const f32x2 sq = r * r;
	const f32x2 c9 = {AT_A9, AT_A9}, c7 = {AT_A7, AT_A7}, c5 = {AT_A5, AT_A5}, c3 = {AT_A3, AT_A3}, c1 = {AT_A1, AT_A1};
	f32x2 p = pk_fma(sq, c9, c7);
	p = pk_fma(sq, p, c5);
	p = pk_fma(sq, p, c3);
	p = pk_fma(sq, p, c1);
	p = p * r;
	const float q20 = ((ay0 > ax0) ? HALF_PI_F : 0.0f) - p.x;
	const float q21 = ((ay1 > ax1) ? HALF_PI_F : 0.0f) - p.y;
	const float q0 = __uint_as_float((uint32_t)((int32_t)__float_as_uint(x0) >> 31) & 0x40490FDBu) - __builtin_fabsf(q20);
	const float q1 = __uint_as_float((uint32_t)((int32_t)__float_as_uint(x1) >> 31) & 0x40490FDBu) - __builtin_fabsf(q21);
	f32x2 out;
	out.x = __builtin_copysignf(q0, y0);
	out.y = __builtin_copysignf(q1, y1);
	return out;
}

// wrap a phase difference into [-pi, pi] and scale by 2/pi (SPEC 3.1)
__device__ __forceinline__ float sd_wrap(float diff)
{
	const float w = diff - __builtin_copysignf(TWO_PI_F, diff);
	return (__builtin_fabsf(diff) > PI_F) ? w : diff;
}

__device__ __forceinline__ float sd_clamp(float v, float lo, float hi)
{
	return __builtin_fminf(__builtin_fmaxf(v, lo), hi);
}

// Integer wave reduction with DPP (VALU, no LDS crossbar): rows of 16, then row broadcasts; the total
// lands in lane 63.  Integer addition is associative, so the tree shape is free (SPEC 3).
__device__ __forceinline__ int wave_sum(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
	v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
	v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
	v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1,3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2,3
	return __builtin_amdgcn_readlane(v, 63);
}

#define SD_LH      64     // samples of history kept in front of the tile in LDS
#define SD_BUF     (SD_LH + SD_TILE + 4)

// LDS: the discriminator samples of [tile_start - 64, tile_end) twice, so that every (d[x], d[x+1])
// pair the FIR needs is one 8-byte-aligned ds_read_b64 with an immediate offset:
//   A[x] = d[x],  B[x] = d[x+1]   (x = index relative to tile_start - 64)
struct DemodLds {
	float A[SD_BUF];
	float B[SD_BUF];
	float taps[SD_NPHASE * SD_TAPS_LD];     // rows padded to 36 floats: 16-byte aligned ds_read_b128
	float y[SD_WG];
	float P[1 + SD_TILE / 2];               // P[1 + 256r + tid] = phase of the 2nd sample of load (r, tid); P[0] = previous tile's last
	int red[4][3];
	uint32_t chunk[10];
	uint32_t partial[2];                    // bits already in the ring word that wpos points into (ping-pong)
};

// samples (i, i+1) of the new tile, i even
__device__ __forceinline__ void store_pair(DemodLds &s, uint32_t i, float d0, float d1)
{
	*reinterpret_cast<float2 *>(&s.A[SD_LH + i]) = make_float2(d0, d1);
	s.B[SD_LH + i - 1] = d0;
	s.B[SD_LH + i] = d1;
}

// y(pos) = (sum_{j even} H[p][j] d[n+16-j]) + (sum_{j odd} H[p][j] d[n+16-j]), each an fmaf chain with j
// ascending (SPEC 3.2): one v_pk_fma_f32 per tap pair.  The tap rows are stored pair-swapped
// (T[2i] = H[2i+1], T[2i+1] = H[2i]) so that they line up with the (d[x], d[x+1]) pairs.
// rel = pos relative to A[0], Q16.
__device__ __forceinline__ float interp(const DemodLds &s, uint32_t rel)
{
	const uint32_t top = (rel >> 16) + SD_NTAPS / 2;                    // buffer index of d for j = 0
	const float *h = s.taps + ((rel >> 11) & (SD_NPHASE - 1)) * SD_TAPS_LD;
	// pair i holds (d[top-1-2i], d[top-2i]); it is 8-byte aligned in A when top is odd, in B otherwise
	const float *lo = (top & 1u) ? (s.A + (top - 31u)) : (s.B + (top - 32u));
	f32x2 acc = {0.0f, 0.0f};                                           // (odd chain, even chain)
#pragma unroll
	for (int q = 0; q < SD_NTAPS / 4; q++) {
		const float4 hv = *reinterpret_cast<const float4 *>(h + 4 * q);
		const float2 v0 = *reinterpret_cast<const float2 *>(lo + 30 - 4 * q);
		const float2 v1 = *reinterpret_cast<const float2 *>(lo + 28 - 4 * q);
		const f32x2 h0 = {hv.x, hv.y}, h1 = {hv.z, hv.w};
		const f32x2 d0 = {v0.x, v0.y}, d1 = {v1.x, v1.y};
		acc = pk_fma(h0, d0, acc);
		acc = pk_fma(h1, d1, acc);
	}
	return acc.y + acc.x;
}

template <bool IS_IQ>
__global__ __launch_bounds__(SD_WG, 4) void sd_demod_kernel(
	const float *__restrict__ in, size_t ch_stride, int n_tiles,
	SdChanState *__restrict__ states, float *__restrict__ hist,
	uint32_t *__restrict__ bitring, uint32_t ring_words,
	const float *__restrict__ taps_all, const SdModem *__restrict__ modems)
{
	__shared__ __attribute__((aligned(16))) DemodLds s;

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const uint32_t ch = blockIdx.x;

	SdChanState st = states[ch];
	const SdModem md = modems[st.type];
	const float *taps_g = taps_all + (size_t)st.type * SD_NPHASE * SD_NTAPS;
	for (int i = tid; i < SD_NPHASE * SD_NTAPS; i += SD_WG)
		s.taps[(i >> 5) * SD_TAPS_LD + ((i & 31) ^ 1)] = taps_g[i];     // pair-swapped rows, see interp()
	// restore the carried history in front of the first tile (both copies)
	if (tid < SD_LH) {
		const float hv = hist[(size_t)ch * SD_HIST + tid];
		s.A[tid] = hv;
		if (tid) s.B[tid - 1] = hv;
	}
	uint32_t *ring_g = bitring + (size_t)ch * ring_words;
	const uint32_t ring_mask = ring_words - 1;
	if (tid == 0) {
		s.chunk[0] = 0;
		s.chunk[9] = 0;
		s.partial[0] = ((uint32_t)st.wpos & 31u) ? ring_g[(uint32_t)(st.wpos >> 5) & ring_mask] : 0u;
	}
	int par = 0;
	float carry = st.phi_last;             // thread 255: phase of the last sample stored so far
	if (IS_IQ && tid == SD_WG - 1) s.P[0] = carry;

	constexpr int NLD = IS_IQ ? 4 : 2;     // float4 loads per thread per tile
	constexpr int TILE_F4 = (IS_IQ ? 2 : 1) * SD_TILE / 4;
	const float4 *src = reinterpret_cast<const float4 *>(in + (IS_IQ ? 2 : 1) * (size_t)ch * ch_stride);
	float4 v[NLD];
	f32x2 ph[NLD];                         // IQ: phases of the two samples of each load

	// K1 first half: raw samples -> phases (pure ALU); the 2nd phase of each load goes to P[] so that the
	// next sample's owner (tid+1, or tid 0 of the next load group) can read it after the barrier
	auto k1_phase = [&]() {
		if (IS_IQ) {
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				ph[r] = sd_atan2x2(v[r].y, v[r].x, v[r].w, v[r].z);
				s.P[1 + SD_WG * r + tid] = ph[r].y;
			}
		}
	};
	// K1 second half (after a barrier): phase differences -> both LDS copies
	auto k1_store = [&]() {
		if (IS_IQ) {
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const float prev = s.P[SD_WG * r + tid];
				const float d0 = sd_wrap(ph[r].x - prev) * TWO_OVER_PI;
				const float d1 = sd_wrap(ph[r].y - ph[r].x) * TWO_OVER_PI;
				store_pair(s, 2u * (uint32_t)(tid + SD_WG * r), d0, d1);
			}
			carry = ph[NLD - 1].y;               // meaningful in thread 255: phase of the tile's last sample
		} else {
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const uint32_t i = 4u * (uint32_t)(tid + SD_WG * r);
				store_pair(s, i, v[r].x, v[r].y);
				store_pair(s, i + 2u, v[r].z, v[r].w);
			}
		}
	};
	// after the barrier that follows k1_store: roll the last phase into P[0] for the next tile
	auto k1_carry = [&]() {
		if (IS_IQ && tid == SD_WG - 1) s.P[0] = carry;
	};
	auto load_tile = [&](int tile) {
#pragma unroll
		for (int r = 0; r < NLD; r++) v[r] = src[(size_t)tile * TILE_F4 + tid + SD_WG * r];
	};

	// timing-loop round, first part: both FIR evaluations of this lane's symbol
	float y = 0.0f, m = 0.0f;
	auto round_interp = [&](int K) {
		y = 0.0f; m = 0.0f;
		if (tid < K) {
			const int64_t base = (st.n0 - SD_TILE - SD_LH) << 16;
			const uint32_t rel = (uint32_t)(st.t_next - base) + (uint32_t)tid * (uint32_t)st.period;
			y = interp(s, rel);
			m = interp(s, rel - ((uint32_t)st.period >> 1));
		}
		s.y[tid] = y;
	};
	// second part (after a barrier): Gardner error, slice, integer statistics -> LDS
	float ylast = 0.0f;
	auto round_reduce = [&](int K) {
		int Ei = 0, S1i = 0, S0i = 0;
		bool bit = false;
		if (tid < K) {
			const float prev = tid ? s.y[tid - 1] : st.yprev;
			const float a = prev - y;
			const float b = m - st.bias;
			float e = a * b;
			e = sd_clamp(e * 1024.0f, -1.0e6f, 1.0e6f);
			Ei = __float2int_rn(e);
			bit = y > st.bias;
			const int Y = __float2int_rn(sd_clamp(y, -8.0f, 8.0f) * 4096.0f);
			if (bit) S1i = Y; else S0i = Y;
		}
		const unsigned long long bal = __ballot(bit);
		Ei = wave_sum(Ei);
		S1i = wave_sum(S1i);
		S0i = wave_sum(S0i);
		if (lane == 0) {
			s.red[wave][0] = Ei; s.red[wave][1] = S1i; s.red[wave][2] = S0i;
			s.chunk[1 + 2 * wave] = (uint32_t)bal;
			s.chunk[2 + 2 * wave] = (uint32_t)(bal >> 32);
		}
		ylast = K > 0 ? s.y[K - 1] : 0.0f;
	};
	// third part (after a barrier): append the bits, update slicer levels and the PI loop filter
	auto round_update = [&](int K) {
		if (K <= 0) return;
		const int E = s.red[0][0] + s.red[1][0] + s.red[2][0] + s.red[3][0];
		const int S1 = s.red[0][1] + s.red[1][1] + s.red[2][1] + s.red[3][1];
		const int S0 = s.red[0][2] + s.red[1][2] + s.red[2][2] + s.red[3][2];
		int C1 = 0;
#pragma unroll
		for (int w = 1; w <= 8; w++) C1 += __popc(s.chunk[w]);
		const int C0 = K - C1;
		if (tid < 9) {
			const uint32_t sh = (uint32_t)st.wpos & 31u;
			const uint32_t w0 = (uint32_t)(st.wpos >> 5);
			if ((uint32_t)(32 * tid) < sh + (uint32_t)K) {
				const uint32_t lo = s.chunk[tid + 1];
				const uint32_t pv = s.chunk[tid];
				uint32_t vv = sh ? ((lo << sh) | (pv >> (32u - sh))) : lo;
				const uint32_t idx = (w0 + tid) & ring_mask;
				if (tid == 0 && sh) vv |= s.partial[par] & ((1u << sh) - 1u);
				ring_g[idx] = vv;
				// whoever owns the word the next round starts in publishes it (read after a barrier)
				if ((uint32_t)tid == ((sh + (uint32_t)K) >> 5)) s.partial[par ^ 1] = vv;
			}
		}
		if (C1 > 0 && C0 > 0) {
			const f32x2 cnt = {(float)C1, (float)C0};
			const f32x2 rc = sd_recip2(cnt);
			const float hi = ((float)S1 * rc.x) * (1.0f / 4096.0f);
			const float lo = ((float)S0 * rc.y) * (1.0f / 4096.0f);
			const float c = 0.5f * (hi + lo);
			const float a = 0.5f * (hi - lo);
			if (st.nstat == 0) {
				st.bias = c;
				st.amp = a;
			} else {
				st.bias = st.bias + 0.5f * (c - st.bias);
				st.amp = st.amp + 0.5f * (a - st.amp);
			}
			if (!(st.amp >= 1.0e-3f)) st.amp = 1.0e-3f;
			st.nstat = 1;
		}
		const f32x2 den = {(float)K, st.amp * st.amp};
		const f32x2 rd = sd_recip2(den);
		float err = ((float)E * rd.x) * (1.0f / 1024.0f);
		err = err * rd.y;
		err = sd_clamp(err, -1.0f, 1.0f);
		const int dphase = __float2int_rn(err * md.kp);
		const int dper = __float2int_rn(err * md.ki);
		st.t_next += (int64_t)K * st.period + dphase;
		st.period += dper;
		if (st.period < md.pmin) st.period = md.pmin;
		if (st.period > md.pmax) st.period = md.pmax;
		st.yprev = ylast;
		st.wpos += (uint64_t)K;
		par ^= 1;
	};

	// ---- prologue: tile 0 into LDS, tile 1 in flight
	load_tile(0);
	k1_phase();
	__syncthreads();
	k1_store();
	if (n_tiles > 1) load_tile(1);
	st.n0 += SD_TILE;
	__syncthreads();
	k1_carry();

	for (int tile = 0; tile < n_tiles; tile++) {
		// LDS holds tile `tile` (+64 samples of history); v[] holds the raw samples of tile+1
		const bool has_next = tile + 1 < n_tiles;
		const int64_t limit = (((st.n0 - 1 - SD_NTAPS / 2 - SD_MARGIN) << 16) | 0xFFFF);
		int K_total = (st.t_next <= limit) ? (int)((uint32_t)(limit - st.t_next) / (uint32_t)st.period) + 1 : 0;
		// all but the last round of the tile (only sondes with > 256 symbols per tile get here)
		while (K_total > SD_ROUND_MAX) {
			round_interp(SD_ROUND_MAX);
			__syncthreads();
			round_reduce(SD_ROUND_MAX);
			__syncthreads();
			round_update(SD_ROUND_MAX);
			K_total -= SD_ROUND_MAX;
		}
		// last round, software-pipelined with K1 of the next tile: two barriers per tile
		const int K = K_total;
		round_interp(K);
		float rollA = 0.0f, rollB = 0.0f;
		if (has_next) {
			k1_phase();                                   // ALU work that covers the LDS latency above
			if (tid < SD_LH) rollA = s.A[SD_TILE + tid];
			else if (tid < 2 * SD_LH - 1) rollB = s.B[SD_TILE + tid - SD_LH];
		}
		__syncthreads();                                  // (1) all FIR reads of this tile are done
		round_reduce(K);
		if (has_next) {
			if (tid < SD_LH) s.A[tid] = rollA;            // history roll: last 64 samples to the front
			else if (tid < 2 * SD_LH - 1) s.B[tid - SD_LH] = rollB;
			k1_store();                                   // tile+1 replaces tile in LDS
			if (tile + 2 < n_tiles) load_tile(tile + 2);
		}
		__syncthreads();                                  // (2) statistics + next tile visible
		if (has_next) k1_carry();
		round_update(K);
		if (has_next) st.n0 += SD_TILE;
	}

	// carry the history and the scalar state to the next submit
	if (tid < SD_LH) hist[(size_t)ch * SD_HIST + tid] = s.A[SD_TILE + tid];
	if (tid == 0) {
		if (IS_IQ) st.phi_last = s.P[0];     // written by thread 255 behind the last barrier
		states[ch] = st;
	}
}

void sd_launch_demod(bool is_iq, uint32_t n_channels, hipStream_t stream,
	const float *in, size_t ch_stride, int n_tiles, SdChanState *states, float *hist,
	uint32_t *bitring, uint32_t ring_words, const float *taps_all, const SdModem *modems)
{
	if (is_iq)
		hipLaunchKernelGGL(sd_demod_kernel<true>, dim3(n_channels), dim3(SD_WG), 0, stream,
			in, ch_stride, n_tiles, states, hist, bitring, ring_words, taps_all, modems);
	else
		hipLaunchKernelGGL(sd_demod_kernel<false>, dim3(n_channels), dim3(SD_WG), 0, stream,
			in, ch_stride, n_tiles, states, hist, bitring, ring_words, taps_all, modems);
}
