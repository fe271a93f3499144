// sd_fixed.h -- the fixed-length framers (DFM06/09/17, M10/M20, iMS-100 / RS-11G) as wave-level device code:
//   * the frame-sync correlator (K4) that runs inside the demodulator kernel, like sd_rs41.h's, and
//   * one-wave-per-frame decoders (Manchester / biphase-S, de-interleaving, Hamming(8,4), Meteomodem checksum,
//     BCH(63,51)) for the demod kernel's epilogue and for the stand-alone kernels of framer2_kernel.hip.
// They stand where sondedump's per-sonde framers sit behind dfm09_decode / m10_decode / ims100_decode
// (/root/reference/src/main.hpp:37-39, src/decode/decoder.hpp:8,10,11,61); protocol constants: SURVEY.md Appendix
// B.3-B.5; semantics: SPEC 3.3b, oracle/or_framers.c.  The demodulator delivers on-air chips, two per data bit.
// All integer work: bit-exact by construction.
#pragma once
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "sd_rs41.h"
#include "../../include/sonde_abi.h"

// ---- per-type sync traits.  Ring bit order: stream chip i sits at bit (i & 31) of word i >> 5, so a
// 32-chip window read with alignbit has the FIRST chip in bit 0.
template <int T> struct SyncTraits;
template <> struct SyncTraits<SONDE_DFM09> {
	// Manchester(0x45CF), 1 -> 10, 0 -> 01, first chip in bit 0
	static constexpr uint32_t SYNC = 0x55A566A6u;
	static constexpr int WIN = 32, THR = 3, FRAME_CHIPS = 560;
};
template <> struct SyncTraits<SONDE_M10> {
	// "10011001100110010100110010011001", first chip in bit 0
	static constexpr uint32_t SYNC = 0x99329999u;
	static constexpr int WIN = 32, THR = 3, FRAME_CHIPS = 32 + 16 * 101;
};
template <> struct SyncTraits<SONDE_IMS100> {
	// 24-bit 0x049DCE, first bit at even position 0 of the 48-chip window (bit k at position 2k)
	static constexpr uint32_t SYNC_LO = 0x45410400u;   // bits 0..15 of the word spread to even positions
	static constexpr uint32_t SYNC_HI = 0x00001505u;   // bits 16..23
	static constexpr int WIN = 48, THR = 2, FRAME_CHIPS = 2 * (24 + 12 * 46);
};

template <> struct SyncTraits<SONDE_MRZN1> {
	// Manchester(AA BF 35), 48 chips, first chip in bit 0
	static constexpr uint32_t SYNC_LO = 0x55599999u, SYNC_HI = 0x0000665Au;
	static constexpr int WIN = 48, THR = 4, FRAME_CHIPS = 48 + 16 * 45;
};

template <int T>
__device__ __forceinline__ bool sync_match(uint32_t w0, uint32_t w1, uint32_t w2, int sft, int &inv)
{
	typedef SyncTraits<T> Tr;
	const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sft);
	if (T == SONDE_MRZN1) {
		const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sft);
		const int c = __popc(lo ^ SyncTraits<SONDE_MRZN1>::SYNC_LO) + __popc((hi ^ SyncTraits<SONDE_MRZN1>::SYNC_HI) & 0xFFFFu);
		inv = c >= 48 - Tr::THR;
		return c <= Tr::THR || c >= 48 - Tr::THR;
	} else if (T == SONDE_IMS100) {
		const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sft);
		// biphase-S: bit = 1 when both chips of the cell are equal
		const uint32_t tl = ~(lo ^ (lo >> 1)), th = ~(hi ^ (hi >> 1));
		const int hd = __popc((tl ^ SyncTraits<SONDE_IMS100>::SYNC_LO) & 0x55555555u) +
		               __popc((th ^ SyncTraits<SONDE_IMS100>::SYNC_HI) & 0x00005555u);
		inv = 0;
		return hd <= Tr::THR;
	} else {
		const int c = __popc(lo ^ SyncTraits<(T == SONDE_IMS100 || T == SONDE_MRZN1) ? SONDE_DFM09 : T>::SYNC);
		inv = c >= 32 - Tr::THR;
		return c <= Tr::THR || c >= 32 - Tr::THR;
	}
}

// K4 for one fixed-length sonde type: the state machine of sd_sync_fixed_kernel (framer2_kernel.hip), advanced over
// the bits [.., wp) with one candidate position per lane; `mirror` as in sd_rs41_sync_step.
template <int T, bool REG = false, int NCHUNK = (REG ? 4 : 1)>      // REG, NCHUNK: as sd_rs41_sync_step
__device__ __forceinline__ void sd_fixed_sync_step(SdSyncRun &lds_state, uint64_t wp, const uint32_t *mirror, int lane,
	SdFrameDesc *__restrict__ descs_ch, uint32_t max_frames, SdFrameDesc *list = nullptr)
{
	typedef SyncTraits<T> Tr;
	if (!REG) list = lds_state.list;
	SdSyncRun fs;
	fs.rpos = sd_uniform64(lds_state.rpos); fs.fstart = sd_uniform64(lds_state.fstart);
	fs.collecting = __builtin_amdgcn_readfirstlane(lds_state.collecting);
	fs.inv = __builtin_amdgcn_readfirstlane(lds_state.inv);
	fs.nout = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_state.nout);
	if (fs.collecting && wp < fs.fstart + (uint64_t)Tr::FRAME_CHIPS) return;
	for (;;) {
		if (!fs.collecting) {
			bool found = false;
			while (fs.rpos + Tr::WIN <= wp) {
				constexpr int NCH = NCHUNK;               // chunks of 64 candidate positions per trip (as sd_rs41_sync_step)
				unsigned long long hm[NCH];
				int invv[NCH];
#pragma unroll
				for (int j = 0; j < NCH; j++) {
					const uint64_t pos = fs.rpos + (uint64_t)(64 * j + lane);
					const uint32_t wi = (uint32_t)(pos >> 5);
					const int sh = (int)((uint32_t)pos & 31u);
					const uint32_t w0 = mirror[wi & (SD_MIRROR_WORDS - 1)], w1 = mirror[(wi + 1) & (SD_MIRROR_WORDS - 1)],
					               w2 = mirror[(wi + 2) & (SD_MIRROR_WORDS - 1)];
					int inv;
					const bool hit = sync_match<T>(w0, w1, w2, sh, inv) && pos + Tr::WIN <= wp;
					hm[j] = __ballot(hit);
					invv[j] = inv;
				}
#pragma unroll
				for (int j = 0; j < NCH; j++) {
					if (!found && hm[j]) {
						const int fl = __ffsll((long long)hm[j]) - 1;         // the earliest position wins
						fs.fstart = fs.rpos + (uint64_t)(64 * j + fl);
						fs.inv = __builtin_amdgcn_readlane(invv[j], fl);
						fs.collecting = 1;
						found = true;
					}
				}
				if (found) break;
				uint64_t next = fs.rpos + 64 * NCH;
				if (next > wp - (Tr::WIN - 1)) next = wp - (Tr::WIN - 1);
				fs.rpos = next;
			}
			if (!found) break;
		}
		if (wp < fs.fstart + (uint64_t)Tr::FRAME_CHIPS) break;
		if (fs.nout < max_frames && lane == 0) {
			SdFrameDesc d;
			d.fstart = fs.fstart; d.flen = Tr::FRAME_CHIPS; d.inv = fs.inv;
			descs_ch[fs.nout] = d;
			if (fs.nout < SD_K4_LIST) list[fs.nout] = d;
		}
		fs.nout++;
		fs.rpos = fs.fstart + (uint64_t)Tr::FRAME_CHIPS;
		fs.collecting = 0;
	}
	if (REG) {
		lds_state.rpos = fs.rpos; lds_state.fstart = fs.fstart;
		lds_state.collecting = fs.collecting; lds_state.inv = fs.inv; lds_state.flen = 0; lds_state.nout = fs.nout;
		return;
	}
	if (lane == 0) {
		lds_state.rpos = fs.rpos; lds_state.fstart = fs.fstart;
		lds_state.collecting = fs.collecting; lds_state.inv = fs.inv; lds_state.flen = 0; lds_state.nout = fs.nout;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- per-frame decoders (one wave each)
struct FixedLds {                   // per-wave work area
	uint32_t words[64];             // the frame's chips, first chip in bit 0 of words[0] (up to 2048 chips)
	uint8_t  bytes[416];            // DFM: 33 codewords; M10: 101 bytes; iMS-100: 12 x 34 data bits
	uint8_t  g64[192];              // iMS-100: GF(2^6) exp[128], log[64]
};

#define SD_FIX_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// the frame's chips from the bit ring into s.words, aligned to the frame start.  COHERENT: see sd_rs41_decode_frame.
template <bool COHERENT>
__device__ __forceinline__ void sd_fixed_fetch(FixedLds &s, const uint32_t *__restrict__ ring, uint32_t mask, uint64_t fstart, int nchips, int lane)
{
	for (int i = lane; 32 * i < nchips; i += 64) {
		const uint64_t p = fstart + 32ull * (uint64_t)i;
		const uint32_t w = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
		uint32_t r0, r1;
		if (COHERENT) {
			r0 = __hip_atomic_load(ring + (w & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			r1 = __hip_atomic_load(ring + ((w + 1) & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			r0 = ring[w & mask];
			r1 = ring[(w + 1) & mask];
		}
		s.words[i] = (uint32_t)((((uint64_t)r1 << 32) | r0) >> sh);
	}
	SD_FIX_SYNC();
}
__device__ __forceinline__ uint32_t sd_chip(const FixedLds &s, int i) { return (s.words[i >> 5] >> (i & 31)) & 1u; }

// DFM: Manchester + de-interleave + Hamming(8,4).  syndrome (row0..row3 = bits 3..0) -> code bit to flip (0..7),
// 8 = clean, 9 = uncorrectable.  Rows 01111000 / 10110100 / 11010010 / 11100001.
static __constant__ uint8_t c_dfm_fix[16] = { 8, 7, 6, 9, 5, 9, 9, 0, 4, 9, 9, 1, 9, 2, 3, 9 };

template <bool COHERENT>
__device__ __forceinline__ void sd_dfm_decode_frame(FixedLds &s, const uint32_t *__restrict__ ring, uint32_t mask, const SdFrameDesc d,
	SondeFrame *__restrict__ fr, uint32_t ch, int lane)
{
	sd_fixed_fetch<COHERENT>(s, ring, mask, d.fstart, SyncTraits<SONDE_DFM09>::FRAME_CHIPS, lane);
	int st = 0;
	uint32_t cw = 0;
	if (lane < 33) {
		const int blk = lane < 7 ? 0 : (lane < 20 ? 1 : 2);
		const int off = blk == 0 ? 0 : (blk == 1 ? 56 : 160), N = blk == 0 ? 7 : 13;
		const int i = lane - (blk == 0 ? 0 : (blk == 1 ? 7 : 20));
		for (int j = 0; j < 8; j++) {
			const uint32_t a = sd_chip(s, 32 + 2 * (off + j * N + i)) ^ (uint32_t)d.inv;
			cw |= a << (7 - j);
		}
		const uint32_t syn = ((uint32_t)__popc(cw & 0x78u) & 1u) << 3 | ((uint32_t)__popc(cw & 0xB4u) & 1u) << 2 |
		                     ((uint32_t)__popc(cw & 0xD2u) & 1u) << 1 | ((uint32_t)__popc(cw & 0xE1u) & 1u);
		const uint32_t fix = c_dfm_fix[syn];
		if (fix < 8) { cw ^= 0x80u >> fix; st = 1; }
		else if (fix == 9) st = -1;
	}
	const int ncorr = __popcll(__ballot(st > 0)), nbad = __popcll(__ballot(st < 0));
	if (lane == 0) {
		fr->channel = ch; fr->type = SONDE_DFM09; fr->len = 33;
		fr->nerr[0] = ncorr; fr->nerr[1] = nbad;
		fr->flags = d.inv ? 1u : 0u; fr->bitpos = d.fstart;
	}
	s.bytes[lane] = (uint8_t)cw;
	SD_FIX_SYNC();
	for (int i = lane; i < SONDE_FRAME_MAX / 4; i += 64) {
		uint32_t v = 0;
		if (4 * i < 36) v = (uint32_t)s.bytes[4 * i] | ((uint32_t)s.bytes[4 * i + 1] << 8) | ((uint32_t)s.bytes[4 * i + 2] << 16) | ((uint32_t)s.bytes[4 * i + 3] << 24);
		if (4 * i == 32) v &= 0xFFu;                          // byte 32 is the last codeword
		reinterpret_cast<uint32_t *>(fr->data)[i] = v;
	}
	SD_FIX_SYNC();
}

// M10 / M20: Manchester + Meteomodem's 16-bit rolling checksum; the first byte gives the length.
// The checksum recurrence c' = f(c, b) is linear over GF(2): c' = A c + B b (tests/test_oracle_kat.py checks it), so the
// checksum of n bytes is the XOR over i of A^(n-1-i) B b_i: one table row m10tab[k][j] = A^k B e_j per distance k (built
// on the host by running the recurrence on unit bytes), eight conditional XORs per byte, one XOR reduction over the wave --
// instead of 99 dependent steps on one lane (13 us per frame; the decode kernel took 0.1 ms for 16 384 frames).
template <bool COHERENT>
__device__ __forceinline__ void sd_m10_decode_frame(FixedLds &s, const uint16_t *__restrict__ m10tab /* [99][8] */,
	const uint32_t *__restrict__ ring, uint32_t mask, const SdFrameDesc d, SondeFrame *__restrict__ fr, uint32_t ch, int lane)
{
	sd_fixed_fetch<COHERENT>(s, ring, mask, d.fstart, SyncTraits<SONDE_M10>::FRAME_CHIPS, lane);
	int vb[2] = { 0, 0 };                            // Manchester violations of this lane's bytes (lane, lane + 64)
	for (int i = lane, q = 0; i < 101; i += 64, q++) {
		// 16 chips of byte i: the data bit is the first chip of each pair, MSB first
		const int c0 = 32 + 16 * i;
		const uint64_t pair = (uint64_t)s.words[c0 >> 5] | ((uint64_t)s.words[(c0 >> 5) + 1] << 32);
		uint32_t x = (uint32_t)(pair >> (c0 & 31)) & 0xFFFFu;
		if (d.inv) x ^= 0xFFFFu;
		uint32_t v = 0;
#pragma unroll
		for (int b = 0; b < 8; b++) {
			const uint32_t a = (x >> (2 * b)) & 1u, c = (x >> (2 * b + 1)) & 1u;
			v = (v << 1) | a;
			vb[q] += (a == c);
		}
		s.bytes[i] = (uint8_t)v;
	}
	SD_FIX_SYNC();
	// the first byte is the length of what follows: 0x64 = M10 (101 bytes in all), 0x45 = M20 (70)
	const int total = s.bytes[0] == 0x45 ? 70 : 101;
	int viol = (lane < total ? vb[0] : 0) + (lane + 64 < total ? vb[1] : 0);
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) viol += __shfl_xor(viol, off, 64);
	unsigned cs = 0;
	{
		const int n = total - 2;
#pragma unroll
		for (int q = 0; q < 2; q++) {
			const int i = lane + 64 * q;
			if (i < n) {
				const uint4 row = *reinterpret_cast<const uint4 *>(m10tab + 8 * (n - 1 - i));     // eight 16-bit columns
				const unsigned b = s.bytes[i];
				const uint32_t w[4] = { row.x, row.y, row.z, row.w };
#pragma unroll
				for (int j = 0; j < 8; j++) if ((b >> j) & 1u) cs ^= (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
			}
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) cs ^= (unsigned)__shfl_xor((int)cs, off, 64);
	}
	if (lane == 0) {
		fr->channel = ch; fr->type = SONDE_M10; fr->len = total;
		fr->nerr[0] = (cs == (((unsigned)s.bytes[total - 2] << 8) | s.bytes[total - 1])) ? 0 : -1;
		fr->nerr[1] = viol;
		fr->flags = d.inv ? 1u : 0u; fr->bitpos = d.fstart;
	}
	for (int i = lane; i < SONDE_FRAME_MAX / 4; i += 64) {
		uint32_t v = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) if (4 * i + k < total) v |= (uint32_t)s.bytes[4 * i + k] << (8 * k);
		reinterpret_cast<uint32_t *>(fr->data)[i] = v;
	}
	SD_FIX_SYNC();
}

// iMS-100: biphase-S + BCH(63,51) shortened to (46,34), t = 2, one block per lane
template <bool COHERENT>
__device__ __forceinline__ void sd_ims_decode_frame(FixedLds &s, const uint8_t *__restrict__ g64 /* exp[128], log[64] */,
	const uint32_t *__restrict__ ring, uint32_t mask, const SdFrameDesc d, SondeFrame *__restrict__ fr, uint32_t ch, int lane)
{
	sd_fixed_fetch<COHERENT>(s, ring, mask, d.fstart, SyncTraits<SONDE_IMS100>::FRAME_CHIPS, lane);
	for (int i = lane; i < 192; i += 64) s.g64[i] = g64[i];
	SD_FIX_SYNC();
	const uint8_t *s_exp = s.g64, *s_log = s.g64 + 128;
	int st = 0;
	if (lane < 12) {
		uint64_t blk = 0;
		for (int b = 0; b < 46; b++) {
			const int p = 48 + 2 * (lane * 46 + b);
			blk = (blk << 1) | (uint64_t)(sd_chip(s, p) == sd_chip(s, p + 1));
		}
		unsigned s1 = 0, s3 = 0;
		for (int i = 0; i < 46; i++) {
			if ((blk >> i) & 1) { s1 ^= s_exp[i % 63]; s3 ^= s_exp[(3 * i) % 63]; }
		}
		if (s1 || s3) {
			st = -1;
			if (s1) {
				const unsigned l1 = s_log[s1];
				const unsigned s1c = s_exp[(3 * l1) % 63];
				if (s3 == s1c) {
					if (l1 < 46) { blk ^= 1ull << l1; st = 1; }
				} else {
					const unsigned num = s3 ^ s1c;                    // != 0 here
					const unsigned prod = s_exp[s_log[num] + 63 - l1];
					int found = 0, p0 = 0, p1 = 0;
					for (int i = 0; i < 63; i++) {
						const unsigned x2 = s_exp[(2 * i) % 63];
						const unsigned sx = s_exp[l1 + i];
						if ((x2 ^ sx ^ prod) == 0) {
							if (found == 0) p0 = i; else if (found == 1) p1 = i;
							found++;
						}
					}
					if (found == 2 && p0 < 46 && p1 < 46) { blk ^= (1ull << p0) ^ (1ull << p1); st = 2; }
				}
			}
		}
		for (int b = 0; b < 34; b++) s.bytes[lane * 34 + b] = (uint8_t)((blk >> (45 - b)) & 1);
	}
	int ncorr = st > 0 ? st : 0;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) ncorr += __shfl_xor(ncorr, off, 64);
	const int nbad = __popcll(__ballot(st < 0));
	SD_FIX_SYNC();
	if (lane == 0) {
		fr->channel = ch; fr->type = SONDE_IMS100; fr->len = 51;
		fr->nerr[0] = ncorr; fr->nerr[1] = nbad;
		fr->flags = 0; fr->bitpos = d.fstart;
	}
	for (int i = lane; i < SONDE_FRAME_MAX / 4; i += 64) {
		uint32_t v = 0;
		for (int k = 0; k < 4; k++) {
			const int byte = 4 * i + k;
			if (byte < 51) {
				uint32_t bv = 0;
				for (int b = 0; b < 8; b++) bv = (bv << 1) | s.bytes[8 * byte + b];
				v |= bv << (8 * k);
			}
		}
		reinterpret_cast<uint32_t *>(fr->data)[i] = v;
	}
	SD_FIX_SYNC();
}

// MRZ-N1: Manchester, 45 bytes MSB first behind the 48-chip header; CRC16 (reflected 0xA001, init 0xFFFF) of the first 43
template <bool COHERENT>
__device__ __forceinline__ void sd_mrz_decode_frame(FixedLds &s, const uint32_t *__restrict__ ring, uint32_t mask, const SdFrameDesc d,
	SondeFrame *__restrict__ fr, uint32_t ch, int lane)
{
	sd_fixed_fetch<COHERENT>(s, ring, mask, d.fstart, SyncTraits<SONDE_MRZN1>::FRAME_CHIPS, lane);
	int viol = 0;
	if (lane < 45) {
		const int c0 = 48 + 16 * lane;
		const uint64_t pair = (uint64_t)s.words[c0 >> 5] | ((uint64_t)s.words[(c0 >> 5) + 1] << 32);
		uint32_t x = (uint32_t)(pair >> (c0 & 31)) & 0xFFFFu;
		if (d.inv) x ^= 0xFFFFu;
		uint32_t v = 0;
#pragma unroll
		for (int b = 0; b < 8; b++) {
			const uint32_t a = (x >> (2 * b)) & 1u, c = (x >> (2 * b + 1)) & 1u;
			v = (v << 1) | a;
			viol += (a == c);
		}
		s.bytes[lane] = (uint8_t)v;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) viol += __shfl_xor(viol, off, 64);
	if (lane >= 45) s.bytes[lane] = 0;
	SD_FIX_SYNC();
	if (lane == 0) {
		unsigned crc = 0xFFFFu;
		for (int i = 0; i < 43; i++) {
			crc ^= s.bytes[i];
#pragma unroll
			for (int k = 0; k < 8; k++) crc = (crc & 1u) ? ((crc >> 1) ^ 0xA001u) : (crc >> 1);
		}
		fr->channel = ch; fr->type = SONDE_MRZN1; fr->len = 45;
		fr->nerr[0] = (crc == ((unsigned)s.bytes[43] | ((unsigned)s.bytes[44] << 8))) ? 0 : -1;
		fr->nerr[1] = viol;
		fr->flags = d.inv ? 1u : 0u; fr->bitpos = d.fstart;
	}
	for (int i = lane; i < SONDE_FRAME_MAX / 4; i += 64) {
		uint32_t v = 0;
		if (4 * i < 48) v = (uint32_t)s.bytes[4 * i] | ((uint32_t)s.bytes[4 * i + 1] << 8) | ((uint32_t)s.bytes[4 * i + 2] << 16) | ((uint32_t)s.bytes[4 * i + 3] << 24);
		reinterpret_cast<uint32_t *>(fr->data)[i] = v;
	}
	SD_FIX_SYNC();
}
