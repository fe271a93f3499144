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

__device__ __forceinline__ float sd_atan2(float y, float x)
{
	const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
	const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
	const float r = (mx > 0.0f) ? mn / mx : 0.0f;     // IEEE correctly-rounded division
	const float s = r * r;
	float p = __builtin_fmaf(s, AT_A9, AT_A7);
	p = __builtin_fmaf(s, p, AT_A5);
	p = __builtin_fmaf(s, p, AT_A3);
	p = __builtin_fmaf(s, p, AT_A1);
	p = p * r;
	if (ay > ax) p = HALF_PI_F - p;
	if (x < 0.0f) p = PI_F - p;
	if (y < 0.0f) p = -p;
	return p;
}

__device__ __forceinline__ float sd_clamp(float v, float lo, float hi)
{
	return __builtin_fminf(__builtin_fmaxf(v, lo), hi);
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

template <bool IS_IQ>
__global__ __launch_bounds__(SD_WG) void sd_demod_kernel(
	const float *__restrict__ in, size_t ch_stride, int n_tiles,
	SdChanState *__restrict__ states, float *__restrict__ hist,
	uint32_t *__restrict__ bitring, uint32_t ring_words,
	const float *__restrict__ taps_all, const SdModem *__restrict__ modems)
{
	__shared__ float s_ring[SD_RING];
	__shared__ float s_taps[SD_NPHASE * SD_TAPS_LD];
	__shared__ float s_phi[IS_IQ ? SD_TILE + 1 : 1];
	__shared__ float s_y[SD_WG];
	__shared__ int s_red[4][3];
	__shared__ uint32_t s_chunk[10];
	__shared__ uint32_t s_partial[2];   // bits already in the ring word that wpos points into (ping-pong per round)

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const uint32_t ch = blockIdx.x;

	SdChanState st = states[ch];
	const SdModem md = modems[st.type];
	const float *taps_g = taps_all + (size_t)st.type * SD_NPHASE * SD_NTAPS;
	for (int i = tid; i < SD_NPHASE * SD_NTAPS; i += SD_WG)
		s_taps[(i >> 5) * SD_TAPS_LD + (i & 31)] = taps_g[i];
	// restore the carried tail of the ring
	{
		const float *h = hist + (size_t)ch * SD_HIST;
		for (int i = tid; i < SD_HIST; i += SD_WG)
			s_ring[(uint32_t)(st.n0 - SD_HIST + i) & (SD_RING - 1)] = h[i];
	}
	if (tid == 0) {
		s_chunk[0] = 0;
		s_chunk[9] = 0;
		if (IS_IQ) s_phi[0] = st.phi_last;
	}
	uint32_t *ring_g = bitring + (size_t)ch * ring_words;
	const uint32_t ring_mask = ring_words - 1;
	if (tid == 0) s_partial[0] = ((uint32_t)st.wpos & 31u) ? ring_g[(uint32_t)(st.wpos >> 5) & ring_mask] : 0u;
	int par = 0;
	__syncthreads();

	for (int tile = 0; tile < n_tiles; tile++) {
		const uint32_t nbase = (uint32_t)st.n0;
		if (IS_IQ) {
			const float4 *src = reinterpret_cast<const float4 *>(in + 2 * ((size_t)ch * ch_stride + (size_t)tile * SD_TILE));
			float4 v[4];
#pragma unroll
			for (int r = 0; r < 4; r++) v[r] = src[tid + SD_WG * r];
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int q = tid + SD_WG * r;
				s_phi[1 + 2 * q] = sd_atan2(v[r].y, v[r].x);
				s_phi[2 + 2 * q] = sd_atan2(v[r].w, v[r].z);
			}
			__syncthreads();
#pragma unroll
			for (int r = 0; r < SD_TILE / SD_WG; r++) {
				const int i = tid + SD_WG * r;
				float diff = s_phi[1 + i] - s_phi[i];
				if (diff > PI_F) diff = diff - TWO_PI_F;
				else if (diff <= -PI_F) diff = diff + TWO_PI_F;
				s_ring[(nbase + i) & (SD_RING - 1)] = diff * TWO_OVER_PI;
			}
			// only thread 0 reads s_phi[0] (above, i == 0), so it may already roll the carry over
			if (tid == 0) s_phi[0] = s_phi[SD_TILE];
			__syncthreads();
		} else {
			const float4 *src = reinterpret_cast<const float4 *>(in + (size_t)ch * ch_stride + (size_t)tile * SD_TILE);
#pragma unroll
			for (int r = 0; r < 2; r++) {
				const int q = tid + SD_WG * r;
				const float4 v = src[q];
				const uint32_t b = (nbase + 4 * q) & (SD_RING - 1);
				s_ring[b] = v.x; s_ring[b + 1] = v.y; s_ring[b + 2] = v.z; s_ring[b + 3] = v.w;
			}
			__syncthreads();
		}
		st.n0 += SD_TILE;

		const int64_t limit = (((st.n0 - 1 - SD_NTAPS / 2) << 16) | 0xFFFF);
		while (st.t_next <= limit) {
			const int64_t K64 = (limit - st.t_next) / st.period + 1;
			const int K = K64 > SD_ROUND_MAX ? SD_ROUND_MAX : (int)K64;
			const bool active = tid < K;
			float y = 0.0f, m = 0.0f;
			if (active) {
				const int64_t t = st.t_next + (int64_t)(tid * st.period);
				const int64_t tm = t - (st.period >> 1);
				const uint32_t n1 = (uint32_t)(t >> 16) + SD_NTAPS / 2;
				const uint32_t n2 = (uint32_t)(tm >> 16) + SD_NTAPS / 2;
				const float *h1 = s_taps + ((uint32_t)(t >> 11) & (SD_NPHASE - 1)) * SD_TAPS_LD;
				const float *h2 = s_taps + ((uint32_t)(tm >> 11) & (SD_NPHASE - 1)) * SD_TAPS_LD;
#pragma unroll
				for (int j = 0; j < SD_NTAPS; j++) {
					y = __builtin_fmaf(h1[j], s_ring[(n1 - j) & (SD_RING - 1)], y);
					m = __builtin_fmaf(h2[j], s_ring[(n2 - j) & (SD_RING - 1)], m);
				}
			}
			s_y[tid] = y;
			__syncthreads();
			int Ei = 0, S1i = 0, S0i = 0;
			bool bit = false;
			if (active) {
				const float prev = tid ? s_y[tid - 1] : st.yprev;
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
				s_red[wave][0] = Ei; s_red[wave][1] = S1i; s_red[wave][2] = S0i;
				s_chunk[1 + 2 * wave] = (uint32_t)bal;
				s_chunk[2 + 2 * wave] = (uint32_t)(bal >> 32);
			}
			__syncthreads();
			const int E = s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0];
			const int S1 = s_red[0][1] + s_red[1][1] + s_red[2][1] + s_red[3][1];
			const int S0 = s_red[0][2] + s_red[1][2] + s_red[2][2] + s_red[3][2];
			int C1 = 0;
#pragma unroll
			for (int w = 1; w <= 8; w++) C1 += __popc(s_chunk[w]);
			const int C0 = K - C1;

			// append K bits at bit position wpos of the channel's bit ring
			if (tid < 9) {
				const uint32_t sh = (uint32_t)st.wpos & 31u;
				const uint32_t w0 = (uint32_t)(st.wpos >> 5);
				if ((uint32_t)(32 * tid) < sh + (uint32_t)K) {
					const uint32_t lo = s_chunk[tid + 1];
					const uint32_t pv = s_chunk[tid];
					uint32_t v = sh ? ((lo << sh) | (pv >> (32u - sh))) : lo;
					const uint32_t idx = (w0 + tid) & ring_mask;
					if (tid == 0 && sh) v |= s_partial[par] & ((1u << sh) - 1u);
					ring_g[idx] = v;
					// whoever owns the word the next round starts in publishes it (read after the barrier)
					if ((uint32_t)tid == ((sh + (uint32_t)K) >> 5)) s_partial[par ^ 1] = v;
				}
			}

			if (C1 > 0 && C0 > 0) {
				const float hi = ((float)S1 / (float)C1) * (1.0f / 4096.0f);
				const float lo = ((float)S0 / (float)C0) * (1.0f / 4096.0f);
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
			float err = ((float)E / (float)K) * (1.0f / 1024.0f);
			err = err / (st.amp * st.amp);
			err = sd_clamp(err, -1.0f, 1.0f);
			const int dphase = __float2int_rn(err * md.kp);
			const int dper = __float2int_rn(err * md.ki);
			st.t_next += (int64_t)K * st.period + dphase;
			st.period += dper;
			if (st.period < md.pmin) st.period = md.pmin;
			if (st.period > md.pmax) st.period = md.pmax;
			st.yprev = s_y[K - 1];
			st.wpos += (uint64_t)K;
			par ^= 1;
			__syncthreads();   // s_y / s_red / s_chunk are rewritten by the next round; ring word hand-over
		}
	}

	// carry the ring tail and the scalar state to the next submit
	{
		float *h = hist + (size_t)ch * SD_HIST;
		for (int i = tid; i < SD_HIST; i += SD_WG)
			h[i] = s_ring[(uint32_t)(st.n0 - SD_HIST + i) & (SD_RING - 1)];
	}
	if (tid == 0) {
		if (IS_IQ) st.phi_last = s_phi[0];
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
