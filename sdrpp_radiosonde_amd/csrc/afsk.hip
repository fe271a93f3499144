// afsk.hip -- the AFSK sondes' two extra stages (iMet-1 / iMet-4 and SRS-C50):
//   sd_afsk_kernel  : discriminator output (the FM audio) -> complex mixer at the tone pair's centre (iMet 1700 Hz,
//                     C50 3800 Hz) -> boxcar (iMet: one 1200 Bd symbol = 5 blocks of 8; C50: 2 blocks) ->
//                     8:1 decimation -> second discriminator; its 6 kS/s output goes through kernel A's
//                     real-input path (timing loop, slicer) like any GFSK sonde            (SPEC 3.6)
//   sd_imet_kernel  : bit ring -> asynchronous 8N1 characters -> packets 01 <type> ... CRC16  (SPEC 3.3c)
//   sd_c50_kernel   : bit ring -> 8N1 characters -> packets 00 FF <type> <4 bytes> <2 check bytes>  (SPEC 3.3e)
// (imet4_decode / c50_decode slots, /root/reference/src/main.hpp:40-41, src/decode/decoder.hpp:7,9,61.)  Neither is on the
// benchmarked path: written for exact reproducibility of the oracle's arithmetic, not tuned.
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "sd_math.h"
#include "launch.h"

#define AF_T 256      // threads = block sums per 2048-sample tile

template <int KIND, int PER, int NW>    // KIND: 0 real (discriminator samples), 1 complex64, 2 int16 I, Q pairs, 3 int8 pairs (SD_IN_IQ16 / IQ8: converted exactly, no scaling); mixer table period; boxcar length in blocks of 8 samples (<= 5)
__global__ __launch_bounds__(AF_T) void sd_afsk_kernel(
	const float *__restrict__ in, size_t ch_stride, int n_tiles, const uint32_t *__restrict__ chlist,
	SdAfskState *__restrict__ astates, const float2 *__restrict__ wtab, float *__restrict__ out, size_t out_stride)
{
	__shared__ float2 bs[AF_T + 4];       // block sums: [0..3] the four before this tile, [4 + t] this tile's
	__shared__ float2 zs[AF_T + 1];       // boxcar outputs: [0] the one before this tile
	constexpr bool IS_IQ = KIND != 0;
	const int t = threadIdx.x;
	const uint32_t ch = chlist[blockIdx.x];
	SdAfskState st = astates[ch];
	const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(in) + (size_t)ch * ch_stride * (KIND == 1 ? 8 : (KIND == 3 ? 2 : 4)));
	float *dst = out + (size_t)blockIdx.x * out_stride;
	if (t < 4) bs[t] = make_float2(st.b[t][0], st.b[t][1]);
	if (t == 0) zs[0] = make_float2(st.z[0], st.z[1]);
	uint32_t ph = (uint32_t)(st.n % PER);                // mixer phase of the tile's first sample
	float2 last = make_float2(st.iq_last[0], st.iq_last[1]);
	__syncthreads();
	for (int tile = 0; tile < n_tiles; tile++) {
		const size_t s0 = (size_t)tile * SD_TILE + 8u * (size_t)t;        // this thread's first input sample
		float d[8];
		if (IS_IQ) {
			const float2 *x = reinterpret_cast<const float2 *>(src) + s0;
			const uint32_t *x16 = reinterpret_cast<const uint32_t *>(src) + s0;
			auto sample = [&](int i) -> float2 {
				if (KIND == 2) { const uint32_t q = x16[i]; return make_float2((float)(int16_t)(q & 0xffffu), (float)((int32_t)q >> 16)); }
				if (KIND == 3) { const uint16_t q = (reinterpret_cast<const uint16_t *>(src) + s0)[i]; return make_float2((float)(int8_t)(q & 0xffu), (float)(int8_t)(q >> 8)); }
				return x[i];
			};
			float2 p = s0 ? sample(-1) : last;
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const float2 c = sample(i);
				d[i] = sd_disc(c.x, c.y, p.x, p.y);
				p = c;
			}
			if (tile == n_tiles - 1 && t == AF_T - 1) last = p;
		} else {
#pragma unroll
			for (int i = 0; i < 8; i++) d[i] = src[s0 + i];
		}
		uint32_t k = (ph + 8u * (uint32_t)t) % PER;
		float br = 0.0f, bi = 0.0f;
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const float2 w = wtab[k];
			br = __builtin_fmaf(d[i], w.x, br);
			bi = __builtin_fmaf(d[i], w.y, bi);
			k = (k + 1 == PER) ? 0 : k + 1;
		}
		bs[4 + t] = make_float2(br, bi);
		__syncthreads();
		// boxcar over the newest NW block sums, oldest first
		float zr = bs[t + 5 - NW].x, zi = bs[t + 5 - NW].y;
#pragma unroll
		for (int q = 6 - NW; q <= 4; q++) { zr = zr + bs[t + q].x; zi = zi + bs[t + q].y; }
		zs[1 + t] = make_float2(zr, zi);
		__syncthreads();
		const float2 zp = zs[t];
		dst[(size_t)tile * AF_T + t] = sd_disc(zr, zi, zp.x, zp.y);
		__syncthreads();
		if (t < 4) bs[t] = bs[AF_T + t];
		if (t == 0) zs[0] = zs[AF_T];
		ph = (ph + SD_TILE) % PER;
		__syncthreads();
	}
	if (t < 4) { astates[ch].b[t][0] = bs[t].x; astates[ch].b[t][1] = bs[t].y; }
	if (t == 0) {
		astates[ch].z[0] = zs[0].x; astates[ch].z[1] = zs[0].y;
		astates[ch].n = st.n + (uint64_t)n_tiles * SD_TILE;
	}
	if (IS_IQ && t == AF_T - 1) { astates[ch].iq_last[0] = last.x; astates[ch].iq_last[1] = last.y; }
}

void sd_launch_afsk(int type, int kind /* 0 real, 1 complex64, 2 int16 IQ, 3 int8 IQ */, uint32_t n_list, hipStream_t stream, const float *in, size_t ch_stride, int n_tiles,
	const uint32_t *chlist, SdAfskState *astates, const float *wtab, float *out, size_t out_stride)
{
#define AF_GO(IQ, PER, NW) hipLaunchKernelGGL((sd_afsk_kernel<IQ, PER, NW>), dim3(n_list), dim3(AF_T), 0, stream, in, ch_stride, n_tiles, chlist, astates, \
		(const float2 *)wtab, out, out_stride)
	if (type == SONDE_C50) { if (kind == 3) AF_GO(3, SD_C50_PER, 2); else if (kind == 2) AF_GO(2, SD_C50_PER, 2); else if (kind == 1) AF_GO(1, SD_C50_PER, 2); else AF_GO(0, SD_C50_PER, 2); }
	else { if (kind == 3) AF_GO(3, SD_AF_PER, 5); else if (kind == 2) AF_GO(2, SD_AF_PER, 5); else if (kind == 1) AF_GO(1, SD_AF_PER, 5); else AF_GO(0, SD_AF_PER, 5); }
#undef AF_GO
}

// ---------------------------------------------------------------- iMet framer: one wave per channel
#define IMET_SYNC     0x405u    // bits 1,0,1,0,0,0,0,0,0,0,1,0 in stream order, first bit = bit 0
#define IMET_SYNC_INV 0xBFAu
#define IMET_MAXLEN   64

__device__ __forceinline__ uint32_t bits_at(const uint32_t *ring, uint32_t mask, uint64_t p, int nbits)
{
	const uint32_t w = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
	const uint64_t lo = (uint64_t)ring[w & mask] | ((uint64_t)ring[(w + 1) & mask] << 32);
	return (uint32_t)(lo >> sh) & ((1u << nbits) - 1u);
}

// character at bit position `at`: returns the byte, ok = start bit 0 and stop bit 1 (after polarity)
__device__ __forceinline__ uint32_t imet_char(const uint32_t *ring, uint32_t mask, uint64_t at, uint32_t xinv, bool &ok)
{
	const uint32_t c = bits_at(ring, mask, at, 10) ^ xinv;
	ok = (c & 1u) == 0u && (c >> 9) == 1u;
	return (c >> 1) & 0xFFu;
}

__global__ __launch_bounds__(64) void sd_imet_kernel(
	const SdChanState *__restrict__ states, SdFramerState *__restrict__ fstates,
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	SondeFrame *__restrict__ frames, uint32_t *__restrict__ counts, uint32_t max_frames,
	const uint32_t *__restrict__ chlist)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t s_ring[];
	__shared__ uint8_t s_pkt[IMET_MAXLEN];
	const int lane = threadIdx.x;
	const uint32_t ch = chlist[blockIdx.x];
	const uint32_t mask = ring_words - 1;
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(bitring + (size_t)ch * ring_words);
		uint4 *dst = reinterpret_cast<uint4 *>(s_ring);
		for (uint32_t i = lane; i < ring_words / 4; i += 64) dst[i] = src[i];
	}
	const uint64_t wpos = states[ch].wpos;
	uint64_t rpos = fstates[ch].rpos;
	uint32_t nout = 0;
	__syncthreads();
	while (rpos + 12 <= wpos) {
		const uint64_t p = rpos + (uint64_t)lane;
		uint32_t w = 0;
		const bool valid = p + 12 <= wpos;
		if (valid) w = bits_at(s_ring, mask, p, 12);
		const unsigned long long hm = __ballot(valid && (w == IMET_SYNC || w == IMET_SYNC_INV));
		if (!hm) {
			uint64_t next = rpos + 64;
			if (next > wpos - 11) next = wpos - 11;
			rpos = next;
			continue;
		}
		const int fl = __ffsll((long long)hm) - 1;
		const uint64_t hit = rpos + (uint64_t)fl;
		const bool inv = __shfl((int)w, fl, 64) == (int)IMET_SYNC_INV;
		const uint32_t xinv = inv ? 0x3FFu : 0u;
		const uint64_t c0 = hit + 1;
		if (c0 + 30 > wpos) { rpos = hit; break; }
		bool ok1, ok2;
		const uint32_t type = imet_char(s_ring, mask, c0 + 10, xinv, ok1);
		const uint32_t lenb = imet_char(s_ring, mask, c0 + 20, xinv, ok2);
		int len = 0;
		if (type == 1) len = 14;
		else if (type == 2) len = 18;
		else if (type == 3) len = 5 + (int)lenb;
		else if (type == 4) len = 20;
		if (!ok1 || !ok2 || len == 0 || len > IMET_MAXLEN) { rpos = hit + 1; continue; }
		if (c0 + 10 * (uint64_t)len > wpos) { rpos = hit; break; }
		bool okc = true;
		uint32_t byte = 0;
		if (lane < len) byte = imet_char(s_ring, mask, c0 + 10 * (uint64_t)lane, xinv, okc);
		if (__ballot(!okc) != 0ull) { rpos = hit + 1; continue; }
		if (lane < len) s_pkt[lane] = (uint8_t)byte;
		__syncthreads();
		if (nout < max_frames) {
			SondeFrame *fr = frames + (size_t)ch * max_frames + nout;
			if (lane == 0) {
				uint32_t crc = 0x1D0F;
				for (int i = 0; i < len - 2; i++) {
					crc ^= (uint32_t)s_pkt[i] << 8;
					for (int k = 0; k < 8; k++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) & 0xFFFFu : (crc << 1) & 0xFFFFu;
				}
				fr->channel = ch;
				fr->type = SONDE_IMET4;
				fr->len = len;
				fr->nerr[0] = (crc == (((uint32_t)s_pkt[len - 2] << 8) | s_pkt[len - 1])) ? 0 : -1;
				fr->nerr[1] = 0;
				fr->flags = inv ? 1u : 0u;
				fr->bitpos = c0;
			}
			for (int i = lane; i < SONDE_FRAME_MAX / 4; i += 64) {
				uint32_t wd = 0;
				for (int q = 0; q < 4; q++) wd |= (uint32_t)(4 * i + q < len ? s_pkt[4 * i + q] : 0) << (8 * q);
				reinterpret_cast<uint32_t *>(fr->data)[i] = wd;
			}
		}
		__syncthreads();
		nout++;
		rpos = c0 + 10 * (uint64_t)len - 1;
	}
	if (lane == 0) {
		fstates[ch].rpos = rpos;
		counts[ch] = nout;
	}
}

void sd_launch_framer_imet(uint32_t n_list, hipStream_t stream, const SdChanState *states, SdFramerState *fstates,
	const uint32_t *bitring, uint32_t ring_words, SondeFrame *frames, uint32_t *counts, uint32_t max_frames, const uint32_t *chlist)
{
	hipLaunchKernelGGL(sd_imet_kernel, dim3(n_list), dim3(64), ring_words * sizeof(uint32_t), stream,
		states, fstates, bitring, ring_words, frames, counts, max_frames, chlist);
}

// ---------------------------------------------------------------- SRS-C50 framer: one wave per channel (SPEC 3.3e)
// 8N1 characters at 2400 Bd, LSB first, idle = mark = 1.  Packet: 00 FF <type> <4 value bytes, big-endian> <c1> <c2>,
// c1 = sum of type and value bytes mod 256, c2 = sum of the running c1 values mod 256 (a Fletcher sum) [RECALL: public
// C34/C50 decoder notes].  Sync = the 21 bits  1 | 0 00000000 1 | 0 11111111 1  (stop/idle, the characters 00 and FF),
// exact match in either polarity.  A candidate whose characters have a wrong start/stop bit is dropped (search
// resumes one bit later); it waits while the packet is incomplete; checksum failures are recorded, not dropped.
#define C50_SYNC     0x1FF401u
#define C50_SYNC_INV 0x000BFEu
#define C50_LEN      9

__global__ __launch_bounds__(64) void sd_c50_kernel(
	const SdChanState *__restrict__ states, SdFramerState *__restrict__ fstates,
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	SondeFrame *__restrict__ frames, uint32_t *__restrict__ counts, uint32_t max_frames,
	const uint32_t *__restrict__ chlist)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t s_ring[];
	__shared__ uint8_t s_pkt[16];
	const int lane = threadIdx.x;
	const uint32_t ch = chlist[blockIdx.x];
	const uint32_t mask = ring_words - 1;
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(bitring + (size_t)ch * ring_words);
		uint4 *dst = reinterpret_cast<uint4 *>(s_ring);
		for (uint32_t i = lane; i < ring_words / 4; i += 64) dst[i] = src[i];
	}
	const uint64_t wpos = states[ch].wpos;
	uint64_t rpos = fstates[ch].rpos;
	uint32_t nout = 0;
	__syncthreads();
	while (rpos + 21 <= wpos) {
		const uint64_t p = rpos + (uint64_t)lane;
		uint32_t w = 0;
		const bool valid = p + 21 <= wpos;
		if (valid) w = bits_at(s_ring, mask, p, 21);
		const unsigned long long hm = __ballot(valid && (w == C50_SYNC || w == C50_SYNC_INV));
		if (!hm) {
			uint64_t next = rpos + 64;
			if (next > wpos - 20) next = wpos - 20;
			rpos = next;
			continue;
		}
		const int fl = __ffsll((long long)hm) - 1;
		const uint64_t hit = rpos + (uint64_t)fl;
		const bool inv = __shfl((int)w, fl, 64) == (int)C50_SYNC_INV;
		const uint32_t xinv = inv ? 0x3FFu : 0u;
		const uint64_t c0 = hit + 1;                      // start bit of the 00 character
		if (c0 + 10 * C50_LEN > wpos) { rpos = hit; break; }
		bool okc = true;
		uint32_t byte = 0;
		if (lane < C50_LEN) byte = imet_char(s_ring, mask, c0 + 10 * (uint64_t)lane, xinv, okc);
		if (__ballot(!okc) != 0ull) { rpos = hit + 1; continue; }
		if (lane < C50_LEN) s_pkt[lane] = (uint8_t)byte;
		__syncthreads();
		if (nout < max_frames) {
			SondeFrame *fr = frames + (size_t)ch * max_frames + nout;
			if (lane == 0) {
				uint32_t c1 = 0, c2 = 0;
				for (int i = 2; i < 7; i++) { c1 = (c1 + s_pkt[i]) & 0xFFu; c2 = (c2 + c1) & 0xFFu; }
				fr->channel = ch;
				fr->type = SONDE_C50;
				fr->len = C50_LEN;
				fr->nerr[0] = (c1 == s_pkt[7] && c2 == s_pkt[8]) ? 0 : -1;
				fr->nerr[1] = 0;
				fr->flags = inv ? 1u : 0u;
				fr->bitpos = c0;
			}
			for (int i = lane; i < SONDE_FRAME_MAX / 4; i += 64) {
				uint32_t wd = 0;
				for (int q = 0; q < 4; q++) wd |= (uint32_t)(4 * i + q < C50_LEN ? s_pkt[4 * i + q] : 0) << (8 * q);
				reinterpret_cast<uint32_t *>(fr->data)[i] = wd;
			}
		}
		__syncthreads();
		nout++;
		rpos = c0 + 10 * (uint64_t)C50_LEN - 1;          // the last stop bit may open the next sync
	}
	if (lane == 0) {
		fstates[ch].rpos = rpos;
		counts[ch] = nout;
	}
}

void sd_launch_framer_c50(uint32_t n_list, hipStream_t stream, const SdChanState *states, SdFramerState *fstates,
	const uint32_t *bitring, uint32_t ring_words, SondeFrame *frames, uint32_t *counts, uint32_t max_frames, const uint32_t *chlist)
{
	hipLaunchKernelGGL(sd_c50_kernel, dim3(n_list), dim3(64), ring_words * sizeof(uint32_t), stream,
		states, fstates, bitring, ring_words, frames, counts, max_frames, chlist);
}
