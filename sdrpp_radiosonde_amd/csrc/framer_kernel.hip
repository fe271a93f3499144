// framer_kernel.hip -- kernels B1/B2: bit ring -> sync search -> de-whitening -> RS(255,231) -> frame records.
// B1 (sd_sync_rs41_kernel): one workgroup per channel walks the new bits and lists frame starts.
// B2 (sd_rsdec_rs41_kernel): one 64-lane wave per listed frame extracts, de-whitens and RS-decodes it,
//     so the latency-bound GF(2^8) chains of thousands of frames overlap.
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

__constant__ __attribute__((aligned(4))) uint8_t c_rs41_mask[64] = {
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
	uint8_t  mulk[RS_R * 256];     // mulk[j][v] = v * alpha^j : one dependent lookup per Horner step
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

// One syndrome S_j = r(alpha^j) of the codeword cw[0..4Q) (zero-padded, Q a multiple of 4), as four interleaved
// Horner chains.  The kernel is bound by the CU's one LDS pipe, so the codeword is read a word at a time (one
// ds_read_b32 per chain per four steps; the byte select folds into the xor) and only the multiplication
// table costs one LDS access per step.
template <int Q>
__device__ __forceinline__ uint32_t syndrome4(const FramerTabs &tb, const uint8_t *cw, int j)
{
	static_assert(Q % 4 == 0, "chain length must be a whole number of words");
	const uint8_t *mj = tb.mulk + 256 * j;      // mj[v] = v * alpha^j
	const uint32_t *cww = reinterpret_cast<const uint32_t *>(cw);
	uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll 1
	for (int w = Q / 4 - 1; w >= 0; w--) {
		const uint32_t w0 = cww[w], w1 = cww[Q / 4 + w], w2 = cww[2 * (Q / 4) + w], w3 = cww[3 * (Q / 4) + w];
#pragma unroll
		for (int b = 3; b >= 0; b--) {
			p0 = (uint32_t)mj[p0] ^ ((w0 >> (8 * b)) & 0xFFu);
			p1 = (uint32_t)mj[p1] ^ ((w1 >> (8 * b)) & 0xFFu);
			p2 = (uint32_t)mj[p2] ^ ((w2 >> (8 * b)) & 0xFFu);
			p3 = (uint32_t)mj[p3] ^ ((w3 >> (8 * b)) & 0xFFu);
		}
	}
	const uint32_t la = (uint32_t)(j * Q) % 255u;          // log of A = alpha^(j*Q)
	uint32_t syn = (uint32_t)tb.exp2[tb.log2[p3] + la] ^ p2;
	syn = (uint32_t)tb.exp2[tb.log2[syn] + la] ^ p1;
	syn = (uint32_t)tb.exp2[tb.log2[syn] + la] ^ p0;
	return syn;
}

// Decode both codewords held in s.cw[c][0..n) (zero-padded to 256).  Wave-synchronous; 64 lanes.
__device__ void rs255_decode_pair(const FramerTabs &tb, FramerLds &s, int n, int lane)
{
	// ---- syndromes: lane = 24*c + j, Horner from the highest position down
	uint32_t syn = 0;
	if (lane < 2 * RS_R) {
		const int c = lane / RS_R, j = lane % RS_R;
		// Horner in four independent quarter-chains (4x shorter dependent LDS-lookup chain), then
		// S = ((S3*A + S2)*A + S1)*A + S0 with A = alpha^(j*q): same field element as one long chain.
		// The codeword buffer is zero-padded to 256 and 4q <= 256, so no chain needs a bounds test; each
		// chain keeps "table base + value" in one register, so a step is two LDS byte reads and one
		// xor-add.  The two RS41 frame lengths get their own instantiation (constant chain offsets).
		if (n == RS_R + (RS41_LEN_STD - 56) / 2) syn = syndrome4<((RS_R + (RS41_LEN_STD - 56) / 2 + 15) / 16) * 4>(tb, s.cw[c], j);
		else syn = syndrome4<64>(tb, s.cw[c], j);
		s.logS[c][j] = tb.log2[syn];
	}
	const unsigned long long nzm = __ballot(syn != 0);
	if (lane < 2) {
		const bool nz = (nzm >> (RS_R * lane)) & 0xFFFFFFull;
		s.status[lane] = nz ? 1 : 0;
		s.L[lane] = 0;
	}
	WAVE_SYNC();
	if (nzm == 0ull) return;                    // both codewords clean (wave-uniform): nothing to correct

	// ---- Berlekamp-Massey, one coefficient per lane: half-wave h handles codeword h, lane idx = lane&31 holds
	// lam[idx] and (the logarithm of) Bp[idx] where Bp = x^m * B.  Same recurrence as the sequential form (delta,
	// then lam -= delta/b * x^m B, length change iff 2L <= r), so the same Lambda comes out.  Log domain: the
	// discrepancy term is one antilog read, the update one more, plus the logarithm of the new coefficient.
	{
		const int h = lane >> 5, idx = lane & 31;
		const bool live = s.status[h] > 0;
		uint32_t lam = (idx == 0) ? 1u : 0u;
		uint32_t loglam = (idx == 0) ? 0u : (uint32_t)GF_LZ;
		uint32_t logBp = (idx == 1) ? 0u : (uint32_t)GF_LZ;
		int L = 0;
		uint32_t logbb = 0;                                     // b = 1
#pragma unroll 1
		for (int r = 0; r < RS_R; r++) {
			const uint32_t ls = (live && idx <= r && idx <= L) ? (uint32_t)s.logS[h][r - idx] : (uint32_t)GF_LZ;
			int t = tb.exp2[loglam + ls];
			// xor-reduce over the 32 lanes of this half (two DPP rows)
			t ^= __builtin_amdgcn_update_dpp(0, t, 0xB1, 0xF, 0xF, true);
			t ^= __builtin_amdgcn_update_dpp(0, t, 0x4E, 0xF, 0xF, true);
			t ^= __builtin_amdgcn_update_dpp(0, t, 0x141, 0xF, 0xF, true);
			t ^= __builtin_amdgcn_update_dpp(0, t, 0x140, 0xF, 0xF, true);
			t ^= __builtin_amdgcn_update_dpp(0, t, 0x142, 0xA, 0xF, true);   // row_bcast:15 into rows 1 and 3
			const int d0 = __builtin_amdgcn_readlane(t, 31), d1 = __builtin_amdgcn_readlane(t, 63);
			const uint32_t delta = (uint32_t)(h ? d1 : d0);
			const uint32_t logd = tb.log2[delta];
			// lam -= (delta / b) * Bp ; with delta = 0 the index lands in the zero part of the table
			const uint32_t upd = tb.exp2[logd + 255u - logbb + logBp];
			const bool change = delta != 0u && 2 * L <= r;
			const uint32_t loglam_old = loglam;
			lam ^= upd;
			loglam = tb.log2[lam];
			int up = __shfl_up((int)(change ? loglam_old : logBp), 1, 32);
			if (idx == 0) up = GF_LZ;
			logBp = (uint32_t)up;
			if (change) { L = r + 1 - L; logbb = logd; }
		}
		if (idx < RS_R + 2) { s.lam[h][idx] = (uint8_t)lam; s.loglam[h][idx] = (uint16_t)loglam; }
		const unsigned long long nzl = __ballot(lam != 0);
		const uint32_t halfmask = (uint32_t)(h ? (nzl >> 32) : nzl);
		const int deg = halfmask ? 31 - __clz(halfmask) : 0;
		if (idx == 0 && live) {
			s.L[h] = L;
			if (L > RS_T || deg != L) s.status[h] = -1;
		}
	}
	WAVE_SYNC();

	for (int c = 0; c < 2; c++) {
		if (s.status[c] <= 0) continue;          // wave-uniform
		const int L = s.L[c];
		// ---- Chien search over the n positions of the shortened codeword, position i = lane + 64*it.
		// Lambda has degree L, hence at most L roots among the 255 candidates: "L roots inside [0, n)" is the
		// same condition as "L roots in all and none in the padding" (SPEC 3.3).  Term k at position i is
		// lam[k] * alpha^(-i*k): its logarithm advances by (255 - i) mod 255 per k.
		const int nit = (n + 63) >> 6;
		uint32_t v[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0}, st[4];
#pragma unroll
		for (int it = 0; it < 4; it++) { const uint32_t i = (uint32_t)(lane + 64 * it); st[it] = i ? 255u - i : 0u; }
#pragma unroll 1
		for (int k = 0; k <= L; k++) {
			const uint32_t ll = s.loglam[c][k];
#pragma unroll
			for (int it = 0; it < 4; it++) {
				if (it < nit) {
					v[it] ^= tb.exp2[ll + e[it]];
					e[it] += st[it];
					if (e[it] >= 255u) e[it] -= 255u;
				}
			}
		}
		int npos = 0;
#pragma unroll
		for (int it = 0; it < 4; it++) {
			if (it < nit) {
				const int i = lane + 64 * it;
				const bool root = i < n && v[it] == 0u;
				const unsigned long long rm = __ballot(root);
				if (root) {
					const int idx = npos + __popcll(rm & ((1ull << lane) - 1ull));
					if (idx < RS_T) s.pos[c][idx] = i;
				}
				npos += __popcll(rm);
			}
		}
		if (npos != L) {
			if (lane == 0) s.status[c] = -1;
			WAVE_SYNC();
			continue;
		}
		// ---- omega = S*lam mod x^24
		if (lane < RS_R) {
			uint32_t om = 0;
#pragma unroll 1
			for (int k = 0; k <= lane && k <= L; k++) om ^= tb.exp2[(uint32_t)s.loglam[c][k] + (uint32_t)s.logS[c][lane - k]];
			s.logom[c][lane] = tb.log2[om];
		}
		WAVE_SYNC();
		// ---- Forney: e = X * omega(X^-1) / lam'(X^-1)
		bool bad = false;
		uint32_t ev = 0;
		int p = 0;
		if (lane < npos) {
			p = s.pos[c][lane];
			const uint32_t xi = p ? 255u - (uint32_t)p : 0u;
			uint32_t num = 0, den = 0, ex = 0;
#pragma unroll 1
			for (int k = 0; k < RS_R; k++) {
				num ^= tb.exp2[(uint32_t)s.logom[c][k] + ex];
				ex += xi;
				if (ex >= 255u) ex -= 255u;
			}
			uint32_t xi2 = 2u * xi;
			if (xi2 >= 255u) xi2 -= 255u;
			ex = 0;
#pragma unroll 1
			for (int k = 1; k <= L; k += 2) {
				den ^= tb.exp2[(uint32_t)s.loglam[c][k] + ex];
				ex += xi2;
				if (ex >= 255u) ex -= 255u;
			}
			if (!den) bad = true;
			else ev = tb.exp2[(uint32_t)p + (uint32_t)tb.log2[num] + 255u - (uint32_t)tb.log2[den]];
		}
		if (__ballot(bad) != 0ull) {
			if (lane == 0) s.status[c] = -1;
		} else {
			if (lane < npos) s.cw[c][p] ^= (uint8_t)ev;
			if (lane == 0) s.status[c] = npos;
		}
		WAVE_SYNC();
	}
}

struct SdFrameDesc {            // one frame located by B1, decoded by B2
	uint64_t fstart;            // absolute bit index of the first sync bit
	int32_t  flen;              // bytes
	int32_t  inv;               // polarity
};

#define RS41_SYNC_LO 0x11CAB610u
#define RS41_SYNC_HI 0xF8129622u

// ---------------------------------------------------------------- B1: sync search
// One 256-thread workgroup per channel.  The channel's bit ring is staged in LDS; a search step covers the 2048
// bit positions of 64 ring words, thread t testing the 8 positions  8*(t&3) .. 8*(t&3)+7  of word t>>2, so that
// position order = thread order and "the earliest hit wins" is: lowest wave with a hit, lowest lane in it,
// lowest offset in that lane.  The frame bookkeeping that follows a hit is replicated in every thread (the
// state is tiny), so all control flow is workgroup-uniform and one s_barrier per search step is enough
// (the per-wave results are double-buffered by step parity).
#define B1_WAVES 4
__global__ __launch_bounds__(64 * B1_WAVES) void sd_sync_rs41_kernel(
	const SdChanState *__restrict__ states, SdFramerState *__restrict__ fstates,
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	SdFrameDesc *__restrict__ descs, uint32_t *__restrict__ counts, uint32_t max_frames,
	const uint32_t *__restrict__ chlist)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t s_ring[];   // the channel's whole bit ring
	__shared__ int s_hit[2][B1_WAVES];       // [step parity][wave]: (bit offset inside the 2048-block) << 8 | hd, or -1
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const uint32_t ch = chlist ? chlist[blockIdx.x] : blockIdx.x;
	const uint32_t mask = ring_words - 1;
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(bitring + (size_t)ch * ring_words);
		uint4 *dst = reinterpret_cast<uint4 *>(s_ring);
		for (uint32_t i = tid; i < ring_words / 4; i += 64 * B1_WAVES) dst[i] = src[i];
	}
	const uint64_t wpos = states[ch].wpos;
	SdFramerState fs = fstates[ch];
	uint32_t nout = 0;
	int par = 0;
	__syncthreads();

	for (;;) {
		if (!fs.collecting) {
			bool found = false;
			while (fs.rpos + 64 <= wpos) {
				const uint64_t blk = fs.rpos & ~31ull;
				const uint64_t pos0 = blk + 32ull * (uint64_t)(tid >> 2);
				const uint32_t wi = (uint32_t)(pos0 >> 5);
				const uint32_t w0 = s_ring[wi & mask], w1 = s_ring[(wi + 1) & mask], w2 = s_ring[(wi + 2) & mask];
				// valid offsets s: pos0+s >= rpos and pos0+s+64 <= wpos
				const int s_lo = pos0 >= fs.rpos ? 0 : (int)(fs.rpos - pos0);
				const int64_t room = (int64_t)(wpos - 64) - (int64_t)pos0;
				const int s_hi = room < 0 ? -1 : (room > 31 ? 31 : (int)room);
				const int sub = 8 * (tid & 3);
				int first = 64, hd_first = 0;
#pragma unroll
				for (int q = 7; q >= 0; q--) {
					const int sft = sub + q;
					const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sft);
					const uint32_t c = __popc(lo ^ RS41_SYNC_LO);
					// necessary condition on the low half: c <= THR or c >= 32-THR
					if ((uint32_t)(c - (RS41_SYNC_THR + 1)) >= (uint32_t)(32 - 2 * RS41_SYNC_THR - 1) && sft >= s_lo && sft <= s_hi) {
						const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sft);
						const int hd = (int)c + __popc(hi ^ RS41_SYNC_HI);
						if (hd <= RS41_SYNC_THR || hd >= 64 - RS41_SYNC_THR) { first = sft; hd_first = hd; }
					}
				}
				const unsigned long long hm = __ballot(first < 64);
				if (lane == 0) s_hit[par][wave] = -1;
				if (hm) {
					const int fl = __ffsll((long long)hm) - 1;
					if (lane == fl) s_hit[par][wave] = ((32 * (tid >> 2) + first) << 8) | hd_first;
				}
				__syncthreads();
				int hit = -1;
#pragma unroll
				for (int w = B1_WAVES - 1; w >= 0; w--) {
					const int hw = s_hit[par][w];
					if (hw >= 0) hit = hw;
				}
				par ^= 1;
				if (hit >= 0) {
					fs.fstart = blk + (uint64_t)(hit >> 8);
					fs.inv = (hit & 0xFF) >= 64 - RS41_SYNC_THR;
					fs.collecting = 1;
					found = true;
					break;
				}
				uint64_t next = blk + 32ull * 64ull;
				if (next > wpos - 63) next = wpos - 63;
				fs.rpos = next;
			}
			if (!found) break;
		}
		if (wpos < fs.fstart + 8 * (RS41_TYPE_POS + 1)) break;
		const uint8_t xinv = fs.inv ? 0xFF : 0x00;
		const uint8_t tb = (uint8_t)(byte_at(s_ring, mask, fs.fstart + 8 * RS41_TYPE_POS) ^ xinv ^ c_rs41_mask[RS41_TYPE_POS & 63]);
		const bool ext = __popc(tb ^ 0xF0u) < __popc(tb ^ 0x0Fu);
		const int flen = ext ? RS41_LEN_EXT : RS41_LEN_STD;
		if (wpos < fs.fstart + 8 * (uint64_t)flen) break;
		if (nout < max_frames && tid == 0) {
			SdFrameDesc d;
			d.fstart = fs.fstart; d.flen = flen; d.inv = fs.inv;
			descs[(size_t)ch * max_frames + nout] = d;
		}
		nout++;
		fs.rpos = fs.fstart + 8 * (uint64_t)flen;
		fs.collecting = 0;
	}
	if (tid == 0) {
		fstates[ch] = fs;
		counts[ch] = nout;
	}
}

// ---------------------------------------------------------------- B2: per-frame de-whitening + RS
// 256 threads = 4 waves = 4 frames per workgroup: the 7 KB of GF tables are staged once per
// workgroup, each wave then works alone on its own frame (wave-scope synchronisation only), so that
// all frames of a step are resident at once and their latency-bound GF(2^8) chains overlap.
#define B2_WAVES 4
static_assert(64 * B2_WAVES == 256, "the table staging below writes one log2 entry per thread");
__global__ __launch_bounds__(64 * B2_WAVES) void sd_rsdec_rs41_kernel(
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	const uint8_t *__restrict__ gf_exp, const uint8_t *__restrict__ gf_log, const uint8_t *__restrict__ gf_mulk,
	const SdFrameDesc *__restrict__ descs, const uint32_t *__restrict__ counts, uint32_t max_frames,
	SondeFrame *__restrict__ frames, const uint32_t *__restrict__ chlist)
{
	__shared__ __attribute__((aligned(16))) FramerTabs tabs;
	__shared__ __attribute__((aligned(16))) FramerLds wl[B2_WAVES];
	const uint32_t ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
	const uint32_t nfr = min(counts[ch], max_frames);
	if (B2_WAVES * blockIdx.x >= nfr) return;                  // whole workgroup has nothing to do
	{
		const int tid = threadIdx.x;
		const uint4 *src = reinterpret_cast<const uint4 *>(gf_mulk);
		uint4 *dst = reinterpret_cast<uint4 *>(tabs.mulk);
		for (int i = tid; i < RS_R * 256 / 16; i += 64 * B2_WAVES) dst[i] = src[i];
		// antilog (zero-absorbing, GF_EXP2 bytes) and log (256 x u16) tables as the host laid them out: plain copies
		const uint4 *se = reinterpret_cast<const uint4 *>(gf_exp);
		uint4 *de = reinterpret_cast<uint4 *>(tabs.exp2);
		for (int i = tid; i < GF_EXP2 / 16; i += 64 * B2_WAVES) de[i] = se[i];
		if (tid < 512 / 16) reinterpret_cast<uint4 *>(tabs.log2)[tid] = reinterpret_cast<const uint4 *>(gf_log)[tid];
	}
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int w = threadIdx.x >> 6;
	const uint32_t k = B2_WAVES * blockIdx.x + (uint32_t)w;
	if (k >= nfr) return;
	FramerLds &s = wl[w];
	const uint32_t *ring = bitring + (size_t)ch * ring_words;
	const uint32_t mask = ring_words - 1;
	const SdFrameDesc d = descs[(size_t)ch * max_frames + k];
	const int flen = d.flen;
	const uint8_t xinv = d.inv ? 0xFF : 0x00;
	// K5: extract + de-whiten, four bytes per lane and step (the frame lengths are even, the word past the end
	// is written whole and never read beyond flen)
	{
		const uint32_t winv = d.inv ? 0xFFFFFFFFu : 0u;
		const uint32_t *mask32 = reinterpret_cast<const uint32_t *>(c_rs41_mask);
		uint32_t *frame32 = reinterpret_cast<uint32_t *>(s.frame);
		for (int i = lane; 4 * i < flen; i += 64) {
			const uint64_t p = d.fstart + 32ull * (uint64_t)i;
			const uint32_t w = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
			const uint64_t lo = (uint64_t)ring[w & mask] | ((uint64_t)ring[(w + 1) & mask] << 32);
			frame32[i] = (uint32_t)(lo >> sh) ^ winv ^ mask32[i & 15];
		}
	}
	WAVE_SYNC();
	// K6: de-interleave into two shortened codewords
	const int msglen = (flen - 56) / 2;
	const int n = RS_R + msglen;
	for (int i = lane; i < 2 * 256; i += 64) {
		const int c = i >> 8, kk = i & 255;
		uint8_t v = 0;
		if (kk < RS_R) v = s.frame[8 + RS_R * c + kk];
		else if (kk < n) v = s.frame[56 + 2 * (kk - RS_R) + c];
		s.cw[c][kk] = v;
	}
	WAVE_SYNC();
	rs255_decode_pair(tabs, s, n, lane);
	for (int c = 0; c < 2; c++) {
		if (s.status[c] > 0) {
			for (int kk = lane; kk < n; kk += 64) {
				if (kk < RS_R) s.frame[8 + RS_R * c + kk] = s.cw[c][kk];
				else s.frame[56 + 2 * (kk - RS_R) + c] = s.cw[c][kk];
			}
		}
	}
	WAVE_SYNC();
	SondeFrame *fr = frames + (size_t)ch * max_frames + k;
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

void sd_launch_framer_rs41(uint32_t n_list, hipStream_t stream,
	const SdChanState *states, SdFramerState *fstates, const uint32_t *bitring, uint32_t ring_words,
	const uint8_t *gf_exp, const uint8_t *gf_log, const uint8_t *gf_mulk, void *descs,
	SondeFrame *frames, uint32_t *counts, uint32_t max_frames, uint32_t grid_frames, const uint32_t *chlist)
{
	hipLaunchKernelGGL(sd_sync_rs41_kernel, dim3(n_list), dim3(64 * B1_WAVES), ring_words * sizeof(uint32_t), stream,
		states, fstates, bitring, ring_words, (SdFrameDesc *)descs, counts, max_frames, chlist);
	hipLaunchKernelGGL(sd_rsdec_rs41_kernel, dim3((grid_frames + B2_WAVES - 1) / B2_WAVES, n_list), dim3(64 * B2_WAVES), 0, stream,
		bitring, ring_words, gf_exp, gf_log, gf_mulk, (const SdFrameDesc *)descs, counts, max_frames, frames, chlist);
}
