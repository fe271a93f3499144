// framer2_kernel.hip -- kernels B1/B2 for the Manchester / biphase sondes: DFM06/09/17 (Hamming(8,4)),
// M10 (16-bit checksum), iMS-100 / RS-11G (BCH(63,51), t = 2).
//
// They stand where sondedump's per-sonde framers sit behind dfm09_decode / m10_decode / ims100_decode
// (/root/reference/src/main.hpp:37-39, src/decode/decoder.hpp:8,10,11,61); protocol constants: SURVEY.md
// Appendix B.3-B.5.  The demodulator (kernel A) delivers on-air chips; two chips make one data bit.
//   B1 sd_sync_fixed_kernel<T>: one wave per channel, bit ring staged in LDS, 32 candidate offsets per
//      lane per step (popcount correlator, both polarities), lists complete fixed-length frames.
//   B2 one wave per listed frame: Manchester/biphase decode + the per-sonde FEC.
// All integer work: bit-exact by construction.
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "../../include/sonde_abi.h"


__device__ __forceinline__ uint32_t chip_at(const uint32_t *ring, uint32_t mask, uint64_t p)
{
	return (ring[(uint32_t)(p >> 5) & mask] >> ((uint32_t)p & 31u)) & 1u;
}

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

template <int T>
__device__ __forceinline__ bool sync_match(uint32_t w0, uint32_t w1, uint32_t w2, int sft, int &inv)
{
	typedef SyncTraits<T> Tr;
	const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sft);
	if (T == SONDE_IMS100) {
		const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sft);
		// biphase-S: bit = 1 when both chips of the cell are equal
		const uint32_t tl = ~(lo ^ (lo >> 1)), th = ~(hi ^ (hi >> 1));
		const int hd = __popc((tl ^ SyncTraits<SONDE_IMS100>::SYNC_LO) & 0x55555555u) +
		               __popc((th ^ SyncTraits<SONDE_IMS100>::SYNC_HI) & 0x00005555u);
		inv = 0;
		return hd <= Tr::THR;
	} else {
		const int c = __popc(lo ^ SyncTraits<T == SONDE_IMS100 ? SONDE_DFM09 : T>::SYNC);
		inv = c >= 32 - Tr::THR;
		return c <= Tr::THR || c >= 32 - Tr::THR;
	}
}

template <int T>
__global__ __launch_bounds__(64) void sd_sync_fixed_kernel(
	const SdChanState *__restrict__ states, SdFramerState *__restrict__ fstates,
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	SdFrameDesc *__restrict__ descs, uint32_t *__restrict__ counts, uint32_t max_frames,
	const uint32_t *__restrict__ chlist)
{
	typedef SyncTraits<T> Tr;
	extern __shared__ __attribute__((aligned(16))) uint32_t s_ring[];
	const int lane = threadIdx.x;
	const uint32_t ch = chlist ? chlist[blockIdx.x] : blockIdx.x;
	const uint32_t mask = ring_words - 1;
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(bitring + (size_t)ch * ring_words);
		uint4 *dst = reinterpret_cast<uint4 *>(s_ring);
		for (uint32_t i = lane; i < ring_words / 4; i += 64) dst[i] = src[i];
	}
	const uint64_t wpos = states[ch].wpos;
	SdFramerState fs = fstates[ch];
	uint32_t nout = 0;
	__syncthreads();

	for (;;) {
		if (!fs.collecting) {
			bool found = false;
			while (fs.rpos + Tr::WIN <= wpos) {
				const uint64_t pos0 = (fs.rpos & ~31ull) + 32ull * (uint64_t)lane;
				const uint32_t wi = (uint32_t)(pos0 >> 5);
				const uint32_t w0 = s_ring[wi & mask], w1 = s_ring[(wi + 1) & mask], w2 = s_ring[(wi + 2) & mask];
				const int s_lo = pos0 >= fs.rpos ? 0 : (int)(fs.rpos - pos0);
				const int64_t room = (int64_t)(wpos - Tr::WIN) - (int64_t)pos0;
				const int s_hi = room < 0 ? -1 : (room > 31 ? 31 : (int)room);
				int first = 64, inv_first = 0;
#pragma unroll 4
				for (int sft = 31; sft >= 0; sft--) {
					int inv;
					if (sync_match<T>(w0, w1, w2, sft, inv) && sft >= s_lo && sft <= s_hi) { first = sft; inv_first = inv; }
				}
				const unsigned long long hm = __ballot(first < 64);
				if (hm) {
					const int fl = __ffsll((long long)hm) - 1;
					const int sf = __shfl(first, fl, 64);
					fs.inv = __shfl(inv_first, fl, 64);
					fs.fstart = (fs.rpos & ~31ull) + 32ull * (uint64_t)fl + (uint64_t)sf;
					fs.collecting = 1;
					found = true;
					break;
				}
				uint64_t next = (fs.rpos & ~31ull) + 32ull * 64ull;
				if (next > wpos - (Tr::WIN - 1)) next = wpos - (Tr::WIN - 1);
				fs.rpos = next;
			}
			if (!found) break;
		}
		if (wpos < fs.fstart + (uint64_t)Tr::FRAME_CHIPS) break;
		if (nout < max_frames && lane == 0) {
			SdFrameDesc d;
			d.fstart = fs.fstart; d.flen = Tr::FRAME_CHIPS; d.inv = fs.inv;
			descs[(size_t)ch * max_frames + nout] = d;
		}
		nout++;
		fs.rpos = fs.fstart + (uint64_t)Tr::FRAME_CHIPS;
		fs.collecting = 0;
	}
	if (lane == 0) {
		fstates[ch] = fs;
		counts[ch] = nout;
	}
}

// ---------------------------------------------------------------- DFM: Manchester + de-interleave + Hamming(8,4)
// syndrome (row0..row3 = bits 3..0) -> code bit to flip (0..7), 8 = clean, 9 = uncorrectable.
// Rows 01111000 / 10110100 / 11010010 / 11100001: column i of H is the syndrome of an error in c_i.
__constant__ uint8_t c_dfm_fix[16] = { 8, 7, 6, 9, 5, 9, 9, 0, 4, 9, 9, 1, 9, 2, 3, 9 };

__global__ __launch_bounds__(64) void sd_dec_dfm_kernel(
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	const SdFrameDesc *__restrict__ descs, const uint32_t *__restrict__ counts, uint32_t max_frames,
	SondeFrame *__restrict__ frames, const uint32_t *__restrict__ chlist)
{
	const uint32_t ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
	const uint32_t k = blockIdx.x;
	if (k >= counts[ch] || k >= max_frames) return;
	const int lane = threadIdx.x;
	const uint32_t *ring = bitring + (size_t)ch * ring_words;
	const uint32_t mask = ring_words - 1;
	const SdFrameDesc d = descs[(size_t)ch * max_frames + k];
	int st = 0;
	uint32_t cw = 0;
	if (lane < 33) {
		const int blk = lane < 7 ? 0 : (lane < 20 ? 1 : 2);
		const int off = blk == 0 ? 0 : (blk == 1 ? 56 : 160), N = blk == 0 ? 7 : 13;
		const int i = lane - (blk == 0 ? 0 : (blk == 1 ? 7 : 20));
		for (int j = 0; j < 8; j++) {
			const uint32_t a = chip_at(ring, mask, d.fstart + 32 + 2 * (uint64_t)(off + j * N + i)) ^ (uint32_t)d.inv;
			cw |= a << (7 - j);
		}
		const uint32_t syn = ((uint32_t)__popc(cw & 0x78u) & 1u) << 3 | ((uint32_t)__popc(cw & 0xB4u) & 1u) << 2 |
		                     ((uint32_t)__popc(cw & 0xD2u) & 1u) << 1 | ((uint32_t)__popc(cw & 0xE1u) & 1u);
		const uint32_t fix = c_dfm_fix[syn];
		if (fix < 8) { cw ^= 0x80u >> fix; st = 1; }
		else if (fix == 9) st = -1;
	}
	const int ncorr = __popcll(__ballot(st > 0)), nbad = __popcll(__ballot(st < 0));
	SondeFrame *fr = frames + (size_t)ch * max_frames + k;
	if (lane == 0) {
		fr->channel = ch; fr->type = SONDE_DFM09; fr->len = 33;
		fr->nerr[0] = ncorr; fr->nerr[1] = nbad;
		fr->flags = d.inv ? 1u : 0u; fr->bitpos = d.fstart;
	}
	__shared__ uint8_t s_cw[64];
	s_cw[lane] = (uint8_t)cw;
	__syncthreads();
	for (int i = lane; i < SONDE_FRAME_MAX; i += 64) fr->data[i] = i < 33 ? s_cw[i] : 0;
}

// ---------------------------------------------------------------- M10: Manchester + 16-bit checksum
__device__ __forceinline__ unsigned m10_check_step(unsigned c, unsigned b)
{
	const unsigned c1 = c & 0xFF;
	b = ((b >> 1) | ((b & 1) << 7)) & 0xFF;
	b ^= (b >> 2) & 0xFF;
	const unsigned t6 = (c & 1) ^ ((c >> 2) & 1) ^ ((c >> 4) & 1);
	const unsigned t7 = ((c >> 1) & 1) ^ ((c >> 3) & 1) ^ ((c >> 5) & 1);
	const unsigned t = (c & 0x3F) | (t6 << 6) | (t7 << 7);
	unsigned s = (c >> 7) & 0xFF;
	s ^= (s >> 2) & 0xFF;
	return ((c1 << 8) | (b ^ t ^ s)) & 0xFFFF;
}

__global__ __launch_bounds__(64) void sd_dec_m10_kernel(
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	const SdFrameDesc *__restrict__ descs, const uint32_t *__restrict__ counts, uint32_t max_frames,
	SondeFrame *__restrict__ frames, const uint32_t *__restrict__ chlist)
{
	const uint32_t ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
	const uint32_t k = blockIdx.x;
	if (k >= counts[ch] || k >= max_frames) return;
	__shared__ uint8_t s_fr[128];
	const int lane = threadIdx.x;
	const uint32_t *ring = bitring + (size_t)ch * ring_words;
	const uint32_t mask = ring_words - 1;
	const SdFrameDesc d = descs[(size_t)ch * max_frames + k];
	int vb[2] = { 0, 0 };                            // Manchester violations of this lane's bytes (lane, lane + 64)
	for (int i = lane, q = 0; i < 101; i += 64, q++) {
		uint32_t v = 0;
		for (int b = 0; b < 8; b++) {
			const uint64_t p = d.fstart + 32 + 16 * (uint64_t)i + 2 * (uint64_t)b;
			const uint32_t a = chip_at(ring, mask, p) ^ (uint32_t)d.inv, c = chip_at(ring, mask, p + 1) ^ (uint32_t)d.inv;
			v = (v << 1) | a;
			vb[q] += (a == c);
		}
		s_fr[i] = (uint8_t)v;
	}
	__syncthreads();
	// the first byte is the length of what follows: 0x64 = M10 (101 bytes in all), 0x45 = M20 (70)
	const int total = s_fr[0] == 0x45 ? 70 : 101;
	int viol = (lane < total ? vb[0] : 0) + (lane + 64 < total ? vb[1] : 0);
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) viol += __shfl_xor(viol, off, 64);
	SondeFrame *fr = frames + (size_t)ch * max_frames + k;
	if (lane == 0) {
		unsigned cs = 0;
		for (int i = 0; i < total - 2; i++) cs = m10_check_step(cs, s_fr[i]);
		fr->channel = ch; fr->type = SONDE_M10; fr->len = total;
		fr->nerr[0] = (cs == (((unsigned)s_fr[total - 2] << 8) | s_fr[total - 1])) ? 0 : -1;
		fr->nerr[1] = viol;
		fr->flags = d.inv ? 1u : 0u; fr->bitpos = d.fstart;
	}
	for (int i = lane; i < SONDE_FRAME_MAX; i += 64) fr->data[i] = i < total ? s_fr[i] : 0;
}

// ---------------------------------------------------------------- iMS-100: biphase-S + BCH(63,51) shortened to (46,34)
__global__ __launch_bounds__(64) void sd_dec_ims_kernel(
	const uint32_t *__restrict__ bitring, uint32_t ring_words, const uint8_t *__restrict__ g64 /* exp[128], log[64] */,
	const SdFrameDesc *__restrict__ descs, const uint32_t *__restrict__ counts, uint32_t max_frames,
	SondeFrame *__restrict__ frames, const uint32_t *__restrict__ chlist)
{
	const uint32_t ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
	const uint32_t k = blockIdx.x;
	if (k >= counts[ch] || k >= max_frames) return;
	__shared__ uint8_t s_exp[128], s_log[64], s_bits[12 * 34];
	const int lane = threadIdx.x;
	const uint32_t *ring = bitring + (size_t)ch * ring_words;
	const uint32_t mask = ring_words - 1;
	const SdFrameDesc d = descs[(size_t)ch * max_frames + k];
	for (int i = lane; i < 128; i += 64) s_exp[i] = g64[i];
	s_log[lane] = g64[128 + lane];
	__syncthreads();
	int st = 0;
	if (lane < 12) {
		uint64_t blk = 0;
		for (int b = 0; b < 46; b++) {
			const uint64_t p = d.fstart + 48 + 2 * (uint64_t)(lane * 46 + b);
			blk = (blk << 1) | (uint64_t)(chip_at(ring, mask, p) == chip_at(ring, mask, p + 1));
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
						const unsigned X = s_exp[i];
						const unsigned x2 = s_exp[(2 * i) % 63];
						const unsigned sx = s_exp[l1 + i];
						if ((x2 ^ sx ^ prod) == 0) {
							if (found == 0) p0 = i; else if (found == 1) p1 = i;
							found++;
						}
						(void)X;
					}
					if (found == 2 && p0 < 46 && p1 < 46) { blk ^= (1ull << p0) ^ (1ull << p1); st = 2; }
				}
			}
		}
		for (int b = 0; b < 34; b++) s_bits[lane * 34 + b] = (uint8_t)((blk >> (45 - b)) & 1);
	}
	int ncorr = st > 0 ? st : 0;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) ncorr += __shfl_xor(ncorr, off, 64);
	const int nbad = __popcll(__ballot(st < 0));
	__syncthreads();
	SondeFrame *fr = frames + (size_t)ch * max_frames + k;
	if (lane == 0) {
		fr->channel = ch; fr->type = SONDE_IMS100; fr->len = 51;
		fr->nerr[0] = ncorr; fr->nerr[1] = nbad;
		fr->flags = 0; fr->bitpos = d.fstart;
	}
	for (int i = lane; i < SONDE_FRAME_MAX; i += 64) {
		uint32_t v = 0;
		if (i < 51)
			for (int b = 0; b < 8; b++) v = (v << 1) | s_bits[8 * i + b];
		fr->data[i] = (uint8_t)v;
	}
}

// ---------------------------------------------------------------- host launcher
void sd_launch_framer_other(int type, uint32_t n_list, hipStream_t stream,
	const SdChanState *states, SdFramerState *fstates, const uint32_t *bitring, uint32_t ring_words,
	const uint8_t *g64, void *descs_, SondeFrame *frames, uint32_t *counts, uint32_t max_frames, uint32_t grid_frames, const uint32_t *chlist)
{
	SdFrameDesc *descs = (SdFrameDesc *)descs_;
	const size_t lds = ring_words * sizeof(uint32_t);
	const dim3 g2(grid_frames, n_list);
	switch (type) {
	case SONDE_DFM09:
		hipLaunchKernelGGL(sd_sync_fixed_kernel<SONDE_DFM09>, dim3(n_list), dim3(64), lds, stream, states, fstates, bitring, ring_words, descs, counts, max_frames, chlist);
		hipLaunchKernelGGL(sd_dec_dfm_kernel, g2, dim3(64), 0, stream, bitring, ring_words, descs, counts, max_frames, frames, chlist);
		break;
	case SONDE_M10:
		hipLaunchKernelGGL(sd_sync_fixed_kernel<SONDE_M10>, dim3(n_list), dim3(64), lds, stream, states, fstates, bitring, ring_words, descs, counts, max_frames, chlist);
		hipLaunchKernelGGL(sd_dec_m10_kernel, g2, dim3(64), 0, stream, bitring, ring_words, descs, counts, max_frames, frames, chlist);
		break;
	case SONDE_IMS100:
		hipLaunchKernelGGL(sd_sync_fixed_kernel<SONDE_IMS100>, dim3(n_list), dim3(64), lds, stream, states, fstates, bitring, ring_words, descs, counts, max_frames, chlist);
		hipLaunchKernelGGL(sd_dec_ims_kernel, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist);
		break;
	default:
		break;
	}
}
