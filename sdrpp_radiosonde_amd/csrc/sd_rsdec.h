// sd_rsdec.h -- K5/K6 for RS41 as wave-level device code: one 64-lane wave extracts one listed frame from the bit ring,
// de-whitens it, RS(255,231)-decodes its two interleaved codewords and writes the frame record.  Used by the FEC
// epilogue of the demodulator kernel (demod_kernel.hip) and by the stand-alone FEC kernel (framer_kernel.hip).
//
//   K5  XOR de-whitening with the 64-byte RS41 mask
//   K6  RS(255,231) over GF(2^8)/0x11D, roots alpha^0..alpha^23, two interleaved codewords:
//       syndromes one per lane (48 lanes, four byte-sliced Horner chains per lane, no table memory),
//       Berlekamp-Massey one coefficient per lane, Chien search one position per lane, Forney one error per lane
// (stands where sondedump's rs/framer sit behind rs41_decode, /root/reference/src/main.hpp:36,
//  /root/reference/src/decode/decoder.hpp:61; protocol constants: SURVEY.md Appendix B.2.)
// All integer/byte work: bit-exact by construction.
#pragma once
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "sd_rs41.h"
#include "../../include/sonde_abi.h"

#define RS_R 24
#define RS_T 12

static __constant__ __attribute__((aligned(4))) uint8_t c_rs41_mask[64] = {
	0x96, 0x83, 0x3E, 0x51, 0xB1, 0x49, 0x08, 0x98, 0x32, 0x05, 0x59, 0x0E, 0xF9, 0x44, 0xC6, 0x26,
	0x21, 0x60, 0xC2, 0xEA, 0x79, 0x5D, 0x6D, 0xA1, 0x54, 0x69, 0x47, 0x0C, 0xDC, 0xE8, 0x5C, 0xF1,
	0xF7, 0x76, 0x82, 0x7F, 0x07, 0x99, 0xA2, 0x2C, 0x93, 0x7C, 0x30, 0x63, 0xF5, 0x10, 0x2E, 0x61,
	0xD0, 0xBC, 0xB4, 0xB6, 0x06, 0xAA, 0xF4, 0x23, 0x78, 0x6E, 0x3B, 0xAE, 0xBF, 0x7B, 0x4C, 0xC1,
};

// wave-scope ordering of LDS traffic: DS operations of one wave execute in order, so lanes of the same
// wave see each other's LDS writes once the compiler is kept from reordering across this point
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// GF(2^8) arithmetic in the log domain without zero tests: log of 0 is GF_LZ, larger than any sum of valid
// logarithms (<= 254 + 509), and the antilog table is periodic below GF_LZ and zero from there on, so a product
// with a zero factor reads a zero.  The kernel is bound by the CU's LDS pipe, so a product is one table read once
// the logarithms of its factors are at hand.
#define GF_LZ    768
#define GF_EXP2  (3 * GF_LZ)
struct FramerTabs {                // shared by the waves of a workgroup
	alignas(16) uint8_t  exp2[GF_EXP2];        // alpha^(i mod 255) for i < GF_LZ, 0 above
	alignas(16) uint16_t log2[256];            // log2[0] = GF_LZ
};
struct FramerLds {                 // one per wave (= per frame)
	alignas(4) uint8_t frame[SONDE_FRAME_MAX];
	alignas(4) uint8_t cw[2][256];
	uint16_t logS[2][RS_R];        // logarithms of the syndromes
	uint8_t  lam[2][RS_R + 2];
	uint16_t loglam[2][RS_R + 2];
	uint16_t logom[2][RS_R];
	int     pos[2][RS_T];
	int     L[2];
	int     status[2];     // 0 clean, >0 errors to fix, -1 fail
	int     done[2];       // 1: the single-error fast path has settled this codeword; 2: Lambda known in closed form (two errors)
};

__device__ __forceinline__ uint32_t gmul(const FramerTabs &s, uint32_t a, uint32_t b)
{
	return s.exp2[(uint32_t)s.log2[a] + (uint32_t)s.log2[b]];
}

__device__ __forceinline__ uint8_t byte_at(const uint32_t *ring, uint32_t mask, uint64_t p)
{
	const uint32_t w = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
	const uint64_t lo = (uint64_t)ring[w & mask] | ((uint64_t)ring[(w + 1) & mask] << 32);
	return (uint8_t)(lo >> sh);
}

// Byte-sliced multiplication by a per-lane constant c in GF(2^8): v -> v*c is GF(2)-linear, so
//   v*c = Ta[v & 7] ^ Tb[(v >> 3) & 7] ^ Tc[v >> 6]   with Ta[x] = c*x, Tb[x] = c*(x << 3), Tc[x] = c*(x << 6),
// and v_perm_b32 looks all four bytes of a register up in an 8-entry byte table (two registers) at once: four
// independent Horner chains advance with 2 shifts, 3 ands, 3 perms and 2 xors and no memory access at all
// (the LDS byte-table form cost one conflicting LDS read per multiplication: 65 % of the kernel's LDS cycles
// were bank conflicts, profiles/r1_v16_counters.csv).
struct GfSwar { uint32_t a_lo, a_hi, b_lo, b_hi, c; };       // the 20 table bytes of one multiplier

__device__ __forceinline__ uint32_t gf_swar_mul(uint32_t s, const GfSwar &t)
{
	const uint32_t ia = s & 0x07070707u;
	const uint32_t ib = (s >> 3) & 0x07070707u;
	const uint32_t ic = (s >> 6) & 0x03030303u;
	// v_perm_b32: selector bytes 0..3 pick bytes of the SECOND source, 4..7 of the first
	return __builtin_amdgcn_perm(t.a_hi, t.a_lo, ia) ^ __builtin_amdgcn_perm(t.b_hi, t.b_lo, ib) ^ __builtin_amdgcn_perm(t.c, t.c, ic);
}

// One syndrome S_j = r(alpha^j) of the codeword cw[0..4W) (zero-padded).  Chain k (byte k of the state) runs over
// the positions 4m + k with the multiplier alpha^(4j), so one aligned word of the codeword feeds all four chains
// (a broadcast LDS read: the lanes of a codeword share the address); then S = U0 + a^j U1 + a^2j U2 + a^3j U3.
// (in two pieces so that the in-loop decoder of the demod kernel can stop half way: the chain state u is all there is to carry)
__device__ __forceinline__ uint32_t syndrome_swar_part(const uint8_t *cw, int w_hi, int w_lo, uint32_t u, const GfSwar &t)
{
	const uint32_t *cww = reinterpret_cast<const uint32_t *>(cw);
#pragma unroll 4
	for (int w = w_hi - 1; w >= w_lo; w--) u = gf_swar_mul(u, t) ^ cww[w];
	return u;
}
__device__ __forceinline__ uint32_t syndrome_swar_finish(const FramerTabs &tb, uint32_t u, int j)
{
	uint32_t syn = u & 0xFFu;
	syn ^= (uint32_t)tb.exp2[(uint32_t)tb.log2[(u >> 8) & 0xFFu] + (uint32_t)j];
	syn ^= (uint32_t)tb.exp2[(uint32_t)tb.log2[(u >> 16) & 0xFFu] + 2u * (uint32_t)j];
	syn ^= (uint32_t)tb.exp2[(uint32_t)tb.log2[u >> 24] + 3u * (uint32_t)j];
	return syn;
}
__device__ __forceinline__ uint32_t syndrome_swar(const FramerTabs &tb, const uint8_t *cw, int W, int j, const GfSwar &t)
{
	return syndrome_swar_finish(tb, syndrome_swar_part(cw, W, 0, 0u, t), j);
}

#define SD_TSI(k)

// Decode both codewords held in s.cw[c][0..n) (zero-padded to 256).  Wave-synchronous; 64 lanes.
__device__ void rs255_decode_pair(const FramerTabs &tb, FramerLds &s, int n, int lane, const GfSwar &swar)
{
	// ---- syndromes: lane = 24*c + j, byte-sliced Horner from the highest position down (the codeword buffer is
	// zero-padded to 256 bytes, so the last word may be read whole)
	uint32_t syn = 0, lsyn = GF_LZ;
	SD_TSI(0);
	if (lane < 2 * RS_R) {
		const int c = lane / RS_R, j = lane % RS_R;
		syn = syndrome_swar(tb, s.cw[c], (n + 3) >> 2, j, swar);
		lsyn = tb.log2[syn];
		s.logS[c][j] = (uint16_t)lsyn;
	}
	const unsigned long long nzm = __ballot(syn != 0);
	SD_TSI(1);
	if (nzm == 0ull) {                          // both codewords clean (wave-uniform): nothing to correct
		if (lane < 2) { s.status[lane] = 0; s.L[lane] = 0; s.done[lane] = 0; }
		WAVE_SYNC();
		return;
	}
	// ---- one byte error (the usual case of a codeword that needs the corrector at all): S_j = e X^j for every j, i.e. all
	// 24 syndromes are non-zero and consecutive ones have the same ratio X = alpha^p.  Then Lambda = 1 + X x is the unique
	// shortest recurrence (2 L <= 24), its root is position p, Omega = S_0 and the Forney value e = S_0: exactly what
	// Berlekamp-Massey, Chien and Forney below arrive at after 24 dependent iterations -- including the verdict "uncorrectable"
	// when p lies in the padding of the shortened code (SPEC 3.3).
	bool settled[2];
	{
		const int c = lane >= RS_R ? 1 : 0, j = lane - RS_R * c;
		const int nx = __shfl_down((int)lsyn, 1, 64);
		int diff = nx - (int)lsyn;
		if (diff < 0) diff += 255;
		const int p0 = __builtin_amdgcn_readlane(diff, 0), p1 = __builtin_amdgcn_readlane(diff, RS_R);
		const bool good = lane < 2 * RS_R && lsyn < 255u && (j == RS_R - 1 || (nx < 255 && diff == (c ? p1 : p0)));
		const unsigned long long gm = __ballot(good);
		const bool one0 = (gm & 0xFFFFFFull) == 0xFFFFFFull, one1 = ((gm >> RS_R) & 0xFFFFFFull) == 0xFFFFFFull;
		const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)syn, 0), e1 = (uint32_t)__builtin_amdgcn_readlane((int)syn, RS_R);
		const bool dirty0 = (nzm & 0xFFFFFFull) != 0ull, dirty1 = ((nzm >> RS_R) & 0xFFFFFFull) != 0ull;
		settled[0] = !dirty0 || one0;
		settled[1] = !dirty1 || one1;
		if (lane < 2) {
			const bool one = lane ? one1 : one0, dirty = lane ? dirty1 : dirty0;
			const int p = lane ? p1 : p0;
			const uint32_t e = lane ? e1 : e0;
			int st = dirty ? 1 : 0;
			if (one) {
				if (p < n) s.cw[lane][p] ^= (uint8_t)e;
				else st = -1;
			}
			s.status[lane] = st;
			s.L[lane] = one ? 1 : 0;
			s.done[lane] = one ? 1 : 0;
		}
	}
	WAVE_SYNC();
	if (settled[0] && settled[1]) return;       // wave-uniform

	// ---- two byte errors: Lambda = 1 + L1 x + L2 x^2 in closed form from S0..S3 (Cramer over the first two recurrence
	// equations: D = S1^2 + S0 S2, L1 = (S1 S2 + S0 S3) / D, L2 = (S2^2 + S1 S3) / D), accepted when D != 0, L2 != 0 and all
	// 22 equations S_{j+2} = L1 S_{j+1} + L2 S_j hold (one per lane).  D != 0 rules out a recurrence of length 1, so this is
	// THE shortest recurrence (2 L <= 24: unique), i.e. the polynomial Berlekamp-Massey ends with; Chien, omega and Forney
	// below then run as for any other Lambda.  At the SNRs where the corrector works at all, three or more byte errors in one
	// 156-byte codeword are rare: the 24 dependent iterations of the general algorithm (~ 24 k cycles, the tail of the whole
	// launch whenever one frame of one workgroup needed them) are skipped for nearly every frame.
	bool lam_known[2] = { false, false };
	{
		const int c = lane >= RS_R ? 1 : 0, j = lane - RS_R * c;
		uint32_t l[4];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)lsyn, q), b = (uint32_t)__builtin_amdgcn_readlane((int)lsyn, RS_R + q);
			l[q] = c ? b : a;
		}
		const uint32_t D = (uint32_t)tb.exp2[2u * l[1]] ^ (uint32_t)tb.exp2[l[0] + l[2]];
		const uint32_t N1 = (uint32_t)tb.exp2[l[1] + l[2]] ^ (uint32_t)tb.exp2[l[0] + l[3]];
		const uint32_t N2 = (uint32_t)tb.exp2[2u * l[2]] ^ (uint32_t)tb.exp2[l[1] + l[3]];
		const uint32_t lD = D ? (uint32_t)tb.log2[D] : 0u;                       // (D == 0: nothing below is used)
		const uint32_t L1 = tb.exp2[(uint32_t)tb.log2[N1] + 255u - lD], L2 = tb.exp2[(uint32_t)tb.log2[N2] + 255u - lD];
		const uint32_t lL1 = tb.log2[L1], lL2 = tb.log2[L2];
		const uint32_t n1 = (uint32_t)__shfl_down((int)lsyn, 1, 64), s2 = (uint32_t)__shfl_down((int)syn, 2, 64);
		const uint32_t rhs = (uint32_t)tb.exp2[lL1 + n1] ^ (uint32_t)tb.exp2[lL2 + lsyn];
		const bool okj = lane < 2 * RS_R && D != 0u && L2 != 0u && (j >= RS_R - 2 || rhs == s2);
		const unsigned long long gm = __ballot(okj);
		lam_known[0] = !settled[0] && (gm & 0xFFFFFFull) == 0xFFFFFFull;
		lam_known[1] = !settled[1] && ((gm >> RS_R) & 0xFFFFFFull) == 0xFFFFFFull;
		// lanes c*24 + i, i < 26 would be needed for lam[0..25]: codeword 0 from lanes 0.., codeword 1 from lanes 32..
		const int h = lane >> 5, idx = lane & 31;
		const uint32_t L1h = (uint32_t)__builtin_amdgcn_readlane((int)L1, 0), L1g = (uint32_t)__builtin_amdgcn_readlane((int)L1, RS_R);
		const uint32_t L2h = (uint32_t)__builtin_amdgcn_readlane((int)L2, 0), L2g = (uint32_t)__builtin_amdgcn_readlane((int)L2, RS_R);
		const uint32_t lL1h = (uint32_t)__builtin_amdgcn_readlane((int)lL1, 0), lL1g = (uint32_t)__builtin_amdgcn_readlane((int)lL1, RS_R);
		const uint32_t lL2h = (uint32_t)__builtin_amdgcn_readlane((int)lL2, 0), lL2g = (uint32_t)__builtin_amdgcn_readlane((int)lL2, RS_R);
		if (lam_known[h] && idx < RS_R + 2) {
			const uint32_t v = idx == 0 ? 1u : idx == 1 ? (h ? L1g : L1h) : idx == 2 ? (h ? L2g : L2h) : 0u;
			const uint32_t lv = idx == 0 ? 0u : idx == 1 ? (h ? lL1g : lL1h) : idx == 2 ? (h ? lL2g : lL2h) : (uint32_t)GF_LZ;
			s.lam[h][idx] = (uint8_t)v;
			s.loglam[h][idx] = (uint16_t)lv;
			if (idx == 0) { s.L[h] = 2; s.done[h] = 2; }       // 2: Lambda is known, the root search is still to do
		}
	}
	WAVE_SYNC();
	const bool need_bm = !((settled[0] || lam_known[0]) && (settled[1] || lam_known[1]));      // wave-uniform
	SD_TSI(2);

	// ---- the general case: the error locator by the reformulated inversion-free Berlekamp-Massey recurrence (RiBM, Sarwate &
	// Shanbhag 2001) on lanes 0..36.  Lane i holds (the logarithms of) delta_i and theta_i;
	// initially delta_i = theta_i = S_i (i < 24), delta_36 = theta_36 = 1; every step
	//     delta_i <- gamma delta_{i+1} + delta_0 theta_i;    if delta_0 != 0 and 2L <= r: theta_i <- delta_{i+1}, gamma <- delta_0, L <- r+1-L
	// and after 24 steps lanes 12..24 hold Lambda scaled by a non-zero constant -- the same polynomial the textbook recurrence
	// (oracle/or_fec.c) ends with, so the same roots, the same omega = S Lambda mod x^24 up to that constant, which cancels in
	// Forney's quotient, and the same verdicts (L > 12, deg Lambda != L).  Why: the textbook form needs the discrepancy
	// sum_i Lambda_i S_{r-i} -- a product, a 32-lane reduction, a logarithm -- before it can start the update (a product and
	// another logarithm): four dependent LDS look-ups and seven cross-lane steps per iteration, ~1000 cycles x 24.  Here the
	// discrepancy IS delta_0 (one v_readlane), a step is two parallel antilog reads and one log read: ~ a third of that.
	// Whenever one frame of one workgroup takes this path it is the tail of the whole launch (0.288 against 0.271 ms per step
	// at Eb/N0 9 dB before).
	if (need_bm) {
		// both codewords in one loop (two independent dependency chains per step: each hides the other's LDS latency); a
		// codeword that needs no work runs as all zeros.  delta_{i+1} comes from the next lane by a DPP wave shift.
		const bool todo0 = __builtin_amdgcn_readfirstlane(s.status[0] > 0 && !s.done[0]);       // wave-uniform
		const bool todo1 = __builtin_amdgcn_readfirstlane(s.status[1] > 0 && !s.done[1]);
		const uint32_t one = lane == 3 * RS_T ? 0u : (uint32_t)GF_LZ;
		uint32_t ldl0 = todo0 ? (lane < RS_R ? (uint32_t)s.logS[0][lane] : one) : (uint32_t)GF_LZ;
		uint32_t ldl1 = todo1 ? (lane < RS_R ? (uint32_t)s.logS[1][lane] : one) : (uint32_t)GF_LZ;
		uint32_t lth0 = ldl0, lth1 = ldl1, lgam0 = 0u, lgam1 = 0u, dl0 = 0u, dl1 = 0u;
		int L0 = 0, L1 = 0;
#pragma unroll 2
		for (int r = 0; r < RS_R; r++) {
			const uint32_t ld00 = (uint32_t)__builtin_amdgcn_readlane((int)ldl0, 0), ld01 = (uint32_t)__builtin_amdgcn_readlane((int)ldl1, 0);
			// wave_shl:1 -- lane i reads lane i + 1; lane 63 keeps GF_LZ
			const uint32_t lup0 = (uint32_t)__builtin_amdgcn_update_dpp(GF_LZ, (int)ldl0, 0x130, 0xF, 0xF, false);
			const uint32_t lup1 = (uint32_t)__builtin_amdgcn_update_dpp(GF_LZ, (int)ldl1, 0x130, 0xF, 0xF, false);
			dl0 = (uint32_t)tb.exp2[lgam0 + lup0] ^ (uint32_t)tb.exp2[ld00 + lth0];
			dl1 = (uint32_t)tb.exp2[lgam1 + lup1] ^ (uint32_t)tb.exp2[ld01 + lth1];
			if (ld00 != (uint32_t)GF_LZ && 2 * L0 <= r) { lth0 = lup0; lgam0 = ld00; L0 = r + 1 - L0; }
			if (ld01 != (uint32_t)GF_LZ && 2 * L1 <= r) { lth1 = lup1; lgam1 = ld01; L1 = r + 1 - L1; }
			ldl0 = (uint32_t)tb.log2[dl0];                                                       // (lanes above 36 stay at zero: log = GF_LZ)
			ldl1 = (uint32_t)tb.log2[dl1];
		}
#pragma unroll
		for (int h = 0; h < 2; h++) {
			if (!(h ? todo1 : todo0)) continue;
			// Lambda_k = delta_{12+k}, k = 0..12
			const uint32_t lam = (uint32_t)__shfl_down((int)(h ? dl1 : dl0), RS_T, 64), ll = (uint32_t)__shfl_down((int)(h ? ldl1 : ldl0), RS_T, 64);
			const int L = h ? L1 : L0;
			const bool mine = lane <= RS_T;
			if (lane < RS_R + 2) { s.lam[h][lane] = mine ? (uint8_t)lam : (uint8_t)0; s.loglam[h][lane] = mine ? (uint16_t)ll : (uint16_t)GF_LZ; }
			const unsigned long long nzl = __ballot(mine && lam != 0u);
			const int deg = nzl ? 63 - __clzll((long long)nzl) : 0;
			if (lane == 0) {
				s.L[h] = L;
				if (L > RS_T || deg != L) s.status[h] = -1;
			}
		}
	}
	WAVE_SYNC();

	SD_TSI(3);
	// ---- Chien search, omega and Forney for BOTH codewords at once: half-wave h = lane >> 5 works on codeword h, its
	// 32 lanes on 32 positions (coefficients, errors) at a time.  The table look-ups of different k are independent, so
	// the loops are unrolled to keep several LDS reads in flight (they were one dependent read after the other, and one
	// codeword after the other: 7 of the 18 us this stage cost at the end of the demod kernel).
	{
		const int h = lane >> 5, idx = lane & 31;
		const bool live = s.status[h] > 0 && s.done[h] != 1;
		const int L = live ? s.L[h] : 0;
		const uint16_t *ll = s.loglam[h];
		// Chien search over the n positions of the shortened codeword, position i = idx + 32*it.
		// Lambda has degree L, hence at most L roots among the 255 candidates: "L roots inside [0, n)" is the
		// same condition as "L roots in all and none in the padding" (SPEC 3.3).  Term k at position i is
		// lam[k] * alpha^(-i*k): its logarithm advances by (255 - i) mod 255 per k.
		int npos = 0;
		const int nit = (n + 31) >> 5;
#pragma unroll 1
		for (int it = 0; it < nit; it++) {
			const uint32_t i = (uint32_t)(idx + 32 * it);
			const uint32_t st = (i && i < 255u) ? 255u - i : 0u;
			uint32_t v = 0, e = 0;
#pragma unroll 4
			for (int k = 0; k <= L; k++) {
				v ^= tb.exp2[(uint32_t)ll[k] + e];
				e += st;
				if (e >= 255u) e -= 255u;
			}
			const bool root = live && (int)i < n && v == 0u;
			const unsigned long long rm = __ballot(root);
			const uint32_t rmh = h ? (uint32_t)(rm >> 32) : (uint32_t)rm;
			if (root) {
				const int slot = npos + __popc(rmh & ((1u << idx) - 1u));
				if (slot < RS_T) s.pos[h][slot] = (int)i;
			}
			npos += __popc(rmh);
		}
		const bool ok = live && npos == L;
		if (live && !ok && idx == 0) s.status[h] = -1;
		// ---- omega = S*lam mod x^24
		if (ok && idx < RS_R) {
			uint32_t om = 0;
			const int kmax = idx < L ? idx : L;
#pragma unroll 4
			for (int k = 0; k <= kmax; k++) om ^= tb.exp2[(uint32_t)ll[k] + (uint32_t)s.logS[h][idx - k]];
			s.logom[h][idx] = tb.log2[om];
		}
		WAVE_SYNC();
		// ---- Forney: e = X * omega(X^-1) / lam'(X^-1), one error per lane
		bool bad = false;
		uint32_t ev = 0;
		int p = 0;
		if (ok && idx < npos) {
			p = s.pos[h][idx];
			const uint32_t xi = p ? 255u - (uint32_t)p : 0u;
			uint32_t num = 0, den = 0, ex = 0;
#pragma unroll 4
			for (int k = 0; k < RS_R; k++) {
				num ^= tb.exp2[(uint32_t)s.logom[h][k] + ex];
				ex += xi;
				if (ex >= 255u) ex -= 255u;
			}
			uint32_t xi2 = 2u * xi;
			if (xi2 >= 255u) xi2 -= 255u;
			ex = 0;
#pragma unroll 2
			for (int k = 1; k <= L; k += 2) {
				den ^= tb.exp2[(uint32_t)ll[k] + ex];
				ex += xi2;
				if (ex >= 255u) ex -= 255u;
			}
			if (!den) bad = true;
			else ev = tb.exp2[(uint32_t)p + (uint32_t)tb.log2[num] + 255u - (uint32_t)tb.log2[den]];
		}
		const unsigned long long bm = __ballot(bad);
		const bool anybad = (h ? (uint32_t)(bm >> 32) : (uint32_t)bm) != 0u;     // a zero denominator voids the whole codeword
		if (ok) {
			if (anybad) {
				if (idx == 0) s.status[h] = -1;
			} else {
				if (idx < npos) s.cw[h][p] ^= (uint8_t)ev;
				if (idx == 0) s.status[h] = npos;
			}
		}
		WAVE_SYNC();
	}
	SD_TSI(4);
}


// ---- the pieces of one frame's way through K5 / K6 (shared by the one-call decoder below and by the in-loop decoder)
// K5: extract + de-whiten, four bytes per lane and step (the frame lengths are even, the word past the end is written
// whole and never read beyond flen).  COHERENT: the ring words were written earlier in THIS launch by another wave of
// the workgroup: read them with agent-scope loads, which are served by the L2 and cannot hit a stale line of this CU's vector L1.
template <bool COHERENT>
__device__ __forceinline__ void sd_rs41_extract(FramerLds &s, const uint32_t *__restrict__ ring, uint32_t mask, const SdFrameDesc d, int lane)
{
	const int flen = d.flen;
	const uint32_t winv = d.inv ? 0xFFFFFFFFu : 0u;
	const uint32_t *mask32 = reinterpret_cast<const uint32_t *>(c_rs41_mask);
	uint32_t *frame32 = reinterpret_cast<uint32_t *>(s.frame);
	for (int i = lane; 4 * i < flen; i += 64) {
		const uint64_t p = d.fstart + 32ull * (uint64_t)i;
		const uint32_t w = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
		uint32_t r0, r1;
		if (COHERENT) {
			r0 = __hip_atomic_load(ring + (w & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			r1 = __hip_atomic_load(ring + ((w + 1) & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			r0 = ring[w & mask];
			r1 = ring[(w + 1) & mask];
		}
		const uint64_t lo = (uint64_t)r0 | ((uint64_t)r1 << 32);
		frame32[i] = (uint32_t)(lo >> sh) ^ winv ^ mask32[i & 15];
	}
}
// K6, first step: de-interleave into two shortened codewords, a word at a time: codeword c holds its 24 parity bytes (frame
// bytes 8 + 24c ..) at positions 0..23 and the message bytes frame[56 + 2i + c] at position 24 + i, i.e. word 6 + m of codeword c
// gathers bytes c, 2 + c of frame word 14 + 2m and of frame word 15 + 2m (one v_perm_b32 each; the byte-wise loop cost a
// quarter of this stage's instructions).  Bytes behind the message (the odd length of the extended frame) are zeroed;
// nothing reads a codeword beyond word (n + 3) / 4.  Returns n, the codeword length.
__device__ __forceinline__ int sd_rs41_deinterleave(FramerLds &s, int flen, int lane)
{
	const int msglen = (flen - 56) / 2;
	const uint32_t *frame32 = reinterpret_cast<const uint32_t *>(s.frame);
	uint32_t *cw0 = reinterpret_cast<uint32_t *>(s.cw[0]), *cw1 = reinterpret_cast<uint32_t *>(s.cw[1]);
	if (lane < RS_R / 4) { cw0[lane] = frame32[2 + lane]; cw1[lane] = frame32[2 + RS_R / 4 + lane]; }
	const int valid = msglen - 4 * lane;                      // message bytes in this lane's word
	if (valid > 0) {
		const uint32_t f0 = frame32[14 + 2 * lane], f1 = frame32[15 + 2 * lane];
		const uint32_t keep = valid >= 4 ? 0xFFFFFFFFu : (1u << (8 * valid)) - 1u;
		cw0[RS_R / 4 + lane] = __builtin_amdgcn_perm(f1, f0, 0x06040200u) & keep;
		cw1[RS_R / 4 + lane] = __builtin_amdgcn_perm(f1, f0, 0x07050301u) & keep;
	}
	return RS_R + msglen;
}
// the frame record: header fields from s.status, the frame bytes zero-padded to SONDE_FRAME_MAX
__device__ __forceinline__ void sd_rs41_write_record(const FramerLds &s, const SdFrameDesc d, SondeFrame *__restrict__ fr, uint32_t ch, int lane)
{
	const int flen = d.flen;
	if (lane == 0) {
		fr->channel = ch;
		fr->type = SONDE_RS41;
		fr->len = flen;
		fr->nerr[0] = s.status[0];
		fr->nerr[1] = s.status[1];
		fr->flags = d.inv ? 1u : 0u;
		fr->bitpos = d.fstart;
	}
	for (int i = lane; i < SONDE_FRAME_MAX / 4; i += 64) {
		const int rem = flen - 4 * i;                            // bytes of this word inside the frame
		uint32_t wd = rem > 0 ? reinterpret_cast<const uint32_t *>(s.frame)[i] : 0u;
		if (rem > 0 && rem < 4) wd &= (1u << (8 * rem)) - 1u;
		reinterpret_cast<uint32_t *>(fr->data)[i] = wd;
	}
}

// ---- the in-loop decoder of the demod kernel (round 3), used for launches of ONE generation of workgroups:
// CLEAN frames leave the kernel while it is still streaming.  Round wave 2 -- which otherwise only waits for the lead wave's loop filter -- takes one step of this state machine per
// timing-loop round: extract + de-whiten | de-interleave + first half of the 48 syndrome chains | second half + the all-zero
// test | write the record.  A frame whose syndromes are not all zero is left to the epilogue (the corrector's fast paths take
// 4-20 k cycles, its general path 25 k: several tile periods of one wave, which would hold up the workgroup's barrier), and so
// is whatever the loop did not get to.  At 14 dB 96 % of the frames are clean.
// What was measured (profiles/r3_notes.md, 1024 channels x 96 tiles, ms per step, one box per line):
//   * inlined into the round loop, on the K4 wave: 0.2834 against 0.2692 without (the loop's register allocation spills; even
//     NEVER run the inlined code costs 0.280 against 0.272); inlined on a discriminator wave: 0.2826 against 0.2683 (the
//     spill reloads are vector loads: waiting for them drains the tile prefetch);
//   * as a real function call on round wave 2 (this form): 0.2644 against 0.2697 inlined and 0.2641 without -- kernel 0.2625
//     against 0.2661 -- and 0.2743 against 0.2805 at 9 dB; on 24-tile launches 0.555 against 0.542 without: the steps of the
//     1.6 frames a 24-tile submit completes stall more rounds than the epilogue they save, so the kernel runs the loop decoder
//     only from 48 tiles per submit up;
//   * tracking the frame word by word as it arrives (chains in arrival order with the multipliers alpha^(-4j), no read-back from
//     HBM) needs a call EVERY round on the round's critical path: 0.3846 against 0.2686.  Dropped.
//   * single runs on one box scatter by 1-2 %, which is the size of the effect; interleaved, three runs each (tools/ab_repeat.sh),
//     without / with: 1024 x 96 tiles 0.2745 / 0.2700 (-1.6 %), at 9 dB 0.2847 / 0.2815 (-1.1 %), 8192 x 24 (gated off by the tile
//     count) 0.5513 / 0.5505, mix 4096 x 24 0.2888 / 0.2866, but 4096 x 96 tiles 1.0575 / 1.1317 (+7 %): where several generations of
//     workgroups share the GPU their epilogues overlap the others' streaming anyway and the steps only stall.  Hence the gate:
//     >= 48 tiles AND a launch that is resident all at once (SdFramerOut.loop_fec_max_wg).
struct SdFecJob {
	int32_t  frame;             // index (in K4's list) of the frame in work, or of the next one to look at when phase == 0
	int32_t  phase;             // 0 idle, 1 extracted, 2 half way through the syndromes, 3 clean: record to write
	uint32_t done_mask;         // bit k: frame k's record has been written
	uint32_t usave[2 * RS_R];   // the syndrome chains' states between phases 1 and 2
};
__device__ __attribute__((noinline)) void sd_rs41_loop_step(SdFecJob &job, const SdSyncRun &k4, const FramerTabs &tb, const uint32_t *__restrict__ swar_tab,
	FramerLds &wl, const uint32_t *__restrict__ ring, uint32_t mask, SondeFrame *__restrict__ fout, uint32_t ch, uint32_t max_frames, int lane)
{
	const int phase = __builtin_amdgcn_readfirstlane(job.phase), frame = __builtin_amdgcn_readfirstlane(job.frame);
	// K4 (round wave 3) may be appending to its list in this very round: the count is read with acquire semantics, behind K4's
	// release store, so that every descriptor below `listed` is complete
	const uint32_t nout_now = __hip_atomic_load(const_cast<uint32_t *>(&k4.nout), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
	const int listed = (int)min(min((uint32_t)__builtin_amdgcn_readfirstlane((int)nout_now), max_frames), (uint32_t)SD_K4_LIST);
	if (phase == 0 && frame >= listed) return;                   // nothing new (the usual step)
	SdFrameDesc d;
	d.fstart = sd_uniform64(reinterpret_cast<const unsigned long long *>(&k4.list[frame])[0]);
	d.flen = __builtin_amdgcn_readfirstlane(k4.list[frame].flen);
	d.flen = d.flen == 518 ? 518 : 320;                          // (a frame length is one of the two: the extract loop's bound never comes from anywhere else)
	d.inv = __builtin_amdgcn_readfirstlane(k4.list[frame].inv);
	const int W = (RS_R + (d.flen - 56) / 2 + 3) >> 2, Wh = W >> 1;
	const int c = lane >= RS_R ? 1 : 0, j = lane - RS_R * c;
	GfSwar t;
	int next_phase = 0, next_frame = frame;
	if (phase == 0) {
		sd_rs41_extract<true>(wl, ring, mask, d, lane);
		next_phase = 1;
	} else if (phase == 1) {
		(void)sd_rs41_deinterleave(wl, d.flen, lane);
		WAVE_SYNC();
		if (lane < 2 * RS_R) {
			const uint32_t *sw = swar_tab + 8 * j;
			t.a_lo = sw[0]; t.a_hi = sw[1]; t.b_lo = sw[2]; t.b_hi = sw[3]; t.c = sw[4];
			job.usave[lane] = syndrome_swar_part(wl.cw[c], W, Wh, 0u, t);
		}
		next_phase = 2;
	} else if (phase == 2) {
		uint32_t syn = 0;
		if (lane < 2 * RS_R) {
			const uint32_t *sw = swar_tab + 8 * j;
			t.a_lo = sw[0]; t.a_hi = sw[1]; t.b_lo = sw[2]; t.b_hi = sw[3]; t.c = sw[4];
			syn = syndrome_swar_finish(tb, syndrome_swar_part(wl.cw[c], Wh, 0, job.usave[lane], t), j);
		}
		if (__ballot(syn != 0) == 0ull) next_phase = 3;          // clean: the record is the de-whitened frame as it stands
		else next_frame = frame + 1;                             // dirty: the epilogue's corrector takes it
	} else {
		if (lane < 2) wl.status[lane] = 0;
		WAVE_SYNC();
		sd_rs41_write_record(wl, d, fout + frame, ch, lane);
		if (lane == 0) job.done_mask |= 1u << frame;
		next_frame = frame + 1;
	}
	if (lane == 0) { job.phase = next_phase; job.frame = next_frame; }
	WAVE_SYNC();
}

// One listed frame, start to finish, by one wave.
template <bool COHERENT>
__device__ __forceinline__ void sd_rs41_decode_frame(const FramerTabs &tabs, FramerLds &s, const GfSwar &swar,
	const uint32_t *__restrict__ ring, uint32_t mask, const SdFrameDesc d, SondeFrame *__restrict__ fr, uint32_t ch, int lane)
{
	const int flen = d.flen;
#define SD_TS(x)
	sd_rs41_extract<COHERENT>(s, ring, mask, d, lane);
	WAVE_SYNC();
	SD_TS(ts1);
	const int n = sd_rs41_deinterleave(s, flen, lane);
	WAVE_SYNC();
	SD_TS(ts2);
	rs255_decode_pair(tabs, s, n, lane, swar);
	SD_TS(ts3);
	for (int c = 0; c < 2; c++) {
		if (s.status[c] > 0) {
			for (int kk = lane; kk < n; kk += 64) {
				if (kk < RS_R) s.frame[8 + RS_R * c + kk] = s.cw[c][kk];
				else s.frame[56 + 2 * (kk - RS_R) + c] = s.cw[c][kk];
			}
		}
	}
	WAVE_SYNC();
	sd_rs41_write_record(s, d, fr, ch, lane);
	WAVE_SYNC();       // the next frame of this wave reuses s
}
