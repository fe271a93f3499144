// framer_kernel.hip -- kernel B: bit ring -> sync search -> de-whitening -> RS(255,231) -> frame records.
// One 64-lane wave per channel.
//
//   K4  frame-sync correlator: 64-bit window XOR sync word, popcount, both polarities; the 64
//       candidate offsets of one step are tested one per lane and reduced with a ballot
//   K5  XOR de-whitening with the 64-byte RS41 mask
//   K6  RS(255,231) over GF(2^8)/0x11D, roots alpha^0..alpha^23, two interleaved codewords:
//       syndromes one per lane (48 lanes), Berlekamp-Massey on one lane per codeword,
//       Chien search one position per lane, Forney one error per lane
// (stands where sondedump's framer/correlator/rs sit behind rs41_decode,
//  /root/reference/src/main.hpp:36, /root/reference/src/decode/decoder.hpp:61; protocol constants:
//  SURVEY.md Appendix B.2.)  All integer/byte work: bit-exact by construction.
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "../../include/sonde_abi.h"

#define RS_R 24
#define RS_T 12
#define RS41_SYNC_THR 6
#define RS41_LEN_STD  320
#define RS41_LEN_EXT  518
#define RS41_TYPE_POS 56

// on-air RS41 header 10 B6 CA 11 22 96 12 F8, LSB-first bit order => little-endian u64
#define RS41_SYNC64 0xF812962211CAB610ull

__constant__ uint8_t c_rs41_mask[64] = {
	0x96, 0x83, 0x3E, 0x51, 0xB1, 0x49, 0x08, 0x98, 0x32, 0x05, 0x59, 0x0E, 0xF9, 0x44, 0xC6, 0x26,
	0x21, 0x60, 0xC2, 0xEA, 0x79, 0x5D, 0x6D, 0xA1, 0x54, 0x69, 0x47, 0x0C, 0xDC, 0xE8, 0x5C, 0xF1,
	0xF7, 0x76, 0x82, 0x7F, 0x07, 0x99, 0xA2, 0x2C, 0x93, 0x7C, 0x30, 0x63, 0xF5, 0x10, 0x2E, 0x61,
	0xD0, 0xBC, 0xB4, 0xB6, 0x06, 0xAA, 0xF4, 0x23, 0x78, 0x6E, 0x3B, 0xAE, 0xBF, 0x7B, 0x4C, 0xC1,
};

struct FramerLds {
	uint8_t exp[512];
	uint8_t log[256];
	uint8_t frame[SONDE_FRAME_MAX];
	uint8_t cw[2][256];
	uint8_t S[2][RS_R];
	uint8_t lam[2][RS_R + 2];
	uint8_t B[2][RS_R + 2];
	uint8_t T[2][RS_R + 2];
	uint8_t om[2][RS_R];
	uint8_t ev[2][RS_T];
	int     pos[2][RS_T];
	int     L[2];
	int     status[2];     // 0 clean, >0 errors to fix, -1 fail
};

__device__ __forceinline__ uint8_t gmul(const FramerLds &s, uint8_t a, uint8_t b)
{
	return (a && b) ? s.exp[s.log[a] + s.log[b]] : 0;
}
__device__ __forceinline__ uint8_t gdiv(const FramerLds &s, uint8_t a, uint8_t b)
{
	return a ? s.exp[s.log[a] + 255 - s.log[b]] : 0;
}

// 64 stream bits starting at absolute bit index p (LSB = bit p)
__device__ __forceinline__ uint64_t window64(const uint32_t *ring, uint32_t mask, uint64_t p)
{
	const uint32_t w = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
	const uint64_t lo = (uint64_t)ring[w & mask] | ((uint64_t)ring[(w + 1) & mask] << 32);
	const uint64_t hi = ring[(w + 2) & mask];
	return sh ? ((lo >> sh) | (hi << (64u - sh))) : lo;
}
__device__ __forceinline__ uint8_t byte_at(const uint32_t *ring, uint32_t mask, uint64_t p)
{
	const uint32_t w = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
	const uint64_t lo = (uint64_t)ring[w & mask] | ((uint64_t)ring[(w + 1) & mask] << 32);
	return (uint8_t)(lo >> sh);
}

// Decode both codewords held in s.cw[c][0..n).  Wave-synchronous; 64 lanes.
__device__ void rs255_decode_pair(FramerLds &s, int n, int lane)
{
	// ---- syndromes: lane = 24*c + j, Horner from the highest position down
	uint8_t syn = 0;
	if (lane < 2 * RS_R) {
		const int c = lane / RS_R, j = lane % RS_R;
		for (int i = n - 1; i >= 0; i--)
			syn = (uint8_t)((syn ? s.exp[s.log[syn] + j] : 0) ^ s.cw[c][i]);
		s.S[c][j] = syn;
	}
	const unsigned long long nzm = __ballot(syn != 0);
	if (lane < 2) {
		const bool nz = (nzm >> (RS_R * lane)) & 0xFFFFFFull;
		s.status[lane] = nz ? 1 : 0;
		s.L[lane] = 0;
	}
		__syncthreads();

	// ---- Berlekamp-Massey: lane 0 -> codeword 0, lane 32 -> codeword 1
	if ((lane & 31) == 0) {
		const int c = lane >> 5;
		if (s.status[c] > 0) {
			uint8_t *lam = s.lam[c], *B = s.B[c], *T = s.T[c];
			const uint8_t *S = s.S[c];
			for (int i = 0; i < RS_R + 2; i++) { lam[i] = 0; B[i] = 0; }
			lam[0] = 1; B[0] = 1;
			int L = 0, m = 1;
			uint8_t b = 1;
			for (int r = 0; r < RS_R; r++) {
				uint8_t delta = S[r];
				for (int i = 1; i <= L; i++) delta ^= gmul(s, lam[i], S[r - i]);
				if (!delta) {
					m++;
				} else {
					const uint8_t f = gdiv(s, delta, b);
					if (2 * L <= r) {
						for (int i = 0; i < RS_R + 2; i++) T[i] = lam[i];
						for (int i = 0; i + m < RS_R + 2; i++) lam[i + m] ^= gmul(s, f, B[i]);
						L = r + 1 - L;
						for (int i = 0; i < RS_R + 2; i++) B[i] = T[i];
						b = delta;
						m = 1;
					} else {
						for (int i = 0; i + m < RS_R + 2; i++) lam[i + m] ^= gmul(s, f, B[i]);
						m++;
					}
				}
			}
			int deg = 0;
			for (int i = 0; i < RS_R + 2; i++) if (lam[i]) deg = i;
			s.L[c] = L;
			if (L > RS_T || deg != L) s.status[c] = -1;
		}
	}
		__syncthreads();

	for (int c = 0; c < 2; c++) {
		if (s.status[c] <= 0) continue;          // wave-uniform
		const int L = s.L[c];
		const uint8_t *lam = s.lam[c];
		// ---- Chien search: position i = lane + 64*it
		int npos = 0;
		bool fail = false;
		for (int it = 0; it < 4; it++) {
			const int i = lane + 64 * it;
			bool root = false;
			if (i < 255) {
				uint8_t v = 0;
				for (int k = 0; k <= L; k++)
					if (lam[k]) v ^= s.exp[(s.log[lam[k]] + (255 - i) * k) % 255];
				root = (v == 0);
			}
			const unsigned long long rm = __ballot(root);
			if (root) {
				const int idx = npos + __popcll(rm & ((1ull << lane) - 1ull));
				if (idx < RS_T) s.pos[c][idx] = i;
				if (i >= n) fail = true;
			}
			npos += __popcll(rm);
		}
		if (__ballot(fail) != 0ull || npos != L) {
			if (lane == 0) s.status[c] = -1;
						__syncthreads();
			continue;
		}
		// ---- omega = S*lam mod x^24
		if (lane < RS_R) {
			uint8_t v = 0;
			for (int k = 0; k <= lane && k <= L; k++) v ^= gmul(s, lam[k], s.S[c][lane - k]);
			s.om[c][lane] = v;
		}
				__syncthreads();
		// ---- Forney: e = X * omega(X^-1) / lam'(X^-1)
		bool bad = false;
		uint8_t ev = 0;
		int p = 0;
		if (lane < npos) {
			p = s.pos[c][lane];
			const int xi = (255 - p) % 255;
			uint8_t num = 0, den = 0;
			for (int k = 0; k < RS_R; k++)
				if (s.om[c][k]) num ^= s.exp[(s.log[s.om[c][k]] + xi * k) % 255];
			for (int k = 1; k <= L; k += 2)
				if (lam[k]) den ^= s.exp[(s.log[lam[k]] + xi * (k - 1)) % 255];
			if (!den) bad = true;
			else ev = gmul(s, s.exp[p], gdiv(s, num, den));
		}
		if (__ballot(bad) != 0ull) {
			if (lane == 0) s.status[c] = -1;
		} else {
			if (lane < npos) s.cw[c][p] ^= ev;
			if (lane == 0) s.status[c] = npos;
		}
				__syncthreads();
	}
}

__global__ __launch_bounds__(64) void sd_framer_rs41_kernel(
	const SdChanState *__restrict__ states, SdFramerState *__restrict__ fstates,
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	const uint8_t *__restrict__ gf_exp, const uint8_t *__restrict__ gf_log,
	SondeFrame *__restrict__ frames, uint32_t *__restrict__ counts, uint32_t max_frames,
	const uint32_t *__restrict__ chlist)
{
	__shared__ FramerLds s;
	const int lane = threadIdx.x;
	const uint32_t ch = chlist ? chlist[blockIdx.x] : blockIdx.x;
	const uint32_t *ring = bitring + (size_t)ch * ring_words;
	const uint32_t mask = ring_words - 1;

	for (int i = lane; i < 512; i += 64) s.exp[i] = gf_exp[i];
	for (int i = lane; i < 256; i += 64) s.log[i] = gf_log[i];
		__syncthreads();

	const uint64_t wpos = states[ch].wpos;
	SdFramerState fs = fstates[ch];
	uint32_t nout = 0;

	for (;;) {
		if (!fs.collecting) {
			bool found = false;
			while (fs.rpos + 64 <= wpos) {
				const uint64_t p = fs.rpos + (uint64_t)lane;
				const bool valid = p + 64 <= wpos;
				int hd = 32;
				if (valid) hd = __popcll(window64(ring, mask, p) ^ RS41_SYNC64);
				const bool hit = valid && (hd <= RS41_SYNC_THR || hd >= 64 - RS41_SYNC_THR);
				const unsigned long long hm = __ballot(hit);
				if (hm) {
					const int first = __ffsll((long long)hm) - 1;
					const int hd1 = __shfl(hd, first, 64);
					fs.fstart = fs.rpos + (uint64_t)first;
					fs.inv = hd1 >= 64 - RS41_SYNC_THR;
					fs.collecting = 1;
					found = true;
					break;
				}
				const uint64_t remain = wpos - 63 - fs.rpos;    // candidates left
				fs.rpos += remain < 64 ? remain : 64;
			}
			if (!found) break;
		}
		if (wpos < fs.fstart + 8 * (RS41_TYPE_POS + 1)) break;
		const uint8_t xinv = fs.inv ? 0xFF : 0x00;
		const uint8_t tb = (uint8_t)(byte_at(ring, mask, fs.fstart + 8 * RS41_TYPE_POS) ^ xinv ^ c_rs41_mask[RS41_TYPE_POS & 63]);
		const bool ext = __popc(tb ^ 0xF0u) < __popc(tb ^ 0x0Fu);
		const int flen = ext ? RS41_LEN_EXT : RS41_LEN_STD;
		if (wpos < fs.fstart + 8 * (uint64_t)flen) break;

		// K5: extract + de-whiten
		for (int i = lane; i < flen; i += 64)
			s.frame[i] = (uint8_t)(byte_at(ring, mask, fs.fstart + 8 * (uint64_t)i) ^ xinv ^ c_rs41_mask[i & 63]);
				__syncthreads();
		// K6: de-interleave into two shortened codewords
		const int msglen = (flen - 56) / 2;
		const int n = RS_R + msglen;
		for (int i = lane; i < 2 * 256; i += 64) {
			const int c = i >> 8, k = i & 255;
			uint8_t v = 0;
			if (k < RS_R) v = s.frame[8 + RS_R * c + k];
			else if (k < n) v = s.frame[56 + 2 * (k - RS_R) + c];
			s.cw[c][k] = v;
		}
				__syncthreads();
		rs255_decode_pair(s, n, lane);
		for (int c = 0; c < 2; c++) {
			if (s.status[c] > 0) {
				for (int k = lane; k < n; k += 64) {
					if (k < RS_R) s.frame[8 + RS_R * c + k] = s.cw[c][k];
					else s.frame[56 + 2 * (k - RS_R) + c] = s.cw[c][k];
				}
			}
		}
				__syncthreads();

		if (nout < max_frames) {
			SondeFrame *fr = frames + (size_t)ch * max_frames + nout;
			if (lane == 0) {
				fr->channel = ch;
				fr->type = SONDE_RS41;
				fr->len = flen;
				fr->nerr[0] = s.status[0];
				fr->nerr[1] = s.status[1];
				fr->flags = fs.inv ? 1u : 0u;
				fr->bitpos = fs.fstart;
			}
			for (int i = lane; i < SONDE_FRAME_MAX; i += 64) fr->data[i] = i < flen ? s.frame[i] : 0;
		}
		nout++;
		fs.rpos = fs.fstart + 8 * (uint64_t)flen;
		fs.collecting = 0;
				__syncthreads();
	}
	if (lane == 0) {
		fstates[ch] = fs;
		counts[ch] = nout;
	}
}

void sd_launch_framer_rs41(uint32_t n_list, hipStream_t stream,
	const SdChanState *states, SdFramerState *fstates, const uint32_t *bitring, uint32_t ring_words,
	const uint8_t *gf_exp, const uint8_t *gf_log, SondeFrame *frames, uint32_t *counts, uint32_t max_frames,
	const uint32_t *chlist)
{
	hipLaunchKernelGGL(sd_framer_rs41_kernel, dim3(n_list), dim3(64), 0, stream,
		states, fstates, bitring, ring_words, gf_exp, gf_log, frames, counts, max_frames, chlist);
}
