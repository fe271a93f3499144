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
#include "sd_fixed.h"
#include "../../include/sonde_abi.h"


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

// ---------------------------------------------------------------- stand-alone decoders: one wave per listed frame (sd_fixed.h)
// `g64`: the type's constant table -- iMS-100: GF(2^6) exp[128] + log[64]; M10: the checksum matrix rows (uint16 [99][8])
template <int T>
__global__ __launch_bounds__(64) void sd_dec_fixed_kernel(
	const uint32_t *__restrict__ bitring, uint32_t ring_words, const uint8_t *__restrict__ g64,
	const SdFrameDesc *__restrict__ descs, const uint32_t *__restrict__ counts, uint32_t max_frames,
	SondeFrame *__restrict__ frames, const uint32_t *__restrict__ chlist)
{
	const uint32_t ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
	const uint32_t k = blockIdx.x;
	if (k >= counts[ch] || k >= max_frames) return;
	__shared__ FixedLds s;
	const int lane = threadIdx.x;
	const uint32_t *ring = bitring + (size_t)ch * ring_words;
	const SdFrameDesc d = descs[(size_t)ch * max_frames + k];
	SondeFrame *fr = frames + (size_t)ch * max_frames + k;
	if (T == SONDE_DFM09) sd_dfm_decode_frame<false>(s, ring, ring_words - 1, d, fr, ch, lane);
	else if (T == SONDE_M10) sd_m10_decode_frame<false>(s, reinterpret_cast<const uint16_t *>(g64), ring, ring_words - 1, d, fr, ch, lane);
	else if (T == SONDE_MRZN1) sd_mrz_decode_frame<false>(s, ring, ring_words - 1, d, fr, ch, lane);
	else sd_ims_decode_frame<false>(s, g64, ring, ring_words - 1, d, fr, ch, lane);
}

// ---------------------------------------------------------------- host launcher
void sd_launch_framer_other(int type, uint32_t n_list, hipStream_t stream,
	const SdChanState *states, SdFramerState *fstates, const uint32_t *bitring, uint32_t ring_words,
	const uint8_t *g64, void *descs_, SondeFrame *frames, uint32_t *counts, uint32_t max_frames, uint32_t grid_frames, const uint32_t *chlist, bool with_sync)
{
	SdFrameDesc *descs = (SdFrameDesc *)descs_;
	const size_t lds = ring_words * sizeof(uint32_t);
	const dim3 g2(grid_frames, n_list);
	if (!with_sync) {      // the demod kernel has listed the frames (K4 in-kernel): decode only
		switch (type) {
		case SONDE_DFM09: hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_DFM09>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist); break;
		case SONDE_M10: hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_M10>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist); break;
		case SONDE_IMS100: hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_IMS100>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist); break;
		case SONDE_MRZN1: hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_MRZN1>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist); break;
		default: break;
		}
		return;
	}
	switch (type) {
	case SONDE_DFM09:
		hipLaunchKernelGGL(sd_sync_fixed_kernel<SONDE_DFM09>, dim3(n_list), dim3(64), lds, stream, states, fstates, bitring, ring_words, descs, counts, max_frames, chlist);
		hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_DFM09>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist);
		break;
	case SONDE_M10:
		hipLaunchKernelGGL(sd_sync_fixed_kernel<SONDE_M10>, dim3(n_list), dim3(64), lds, stream, states, fstates, bitring, ring_words, descs, counts, max_frames, chlist);
		hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_M10>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist);
		break;
	case SONDE_IMS100:
		hipLaunchKernelGGL(sd_sync_fixed_kernel<SONDE_IMS100>, dim3(n_list), dim3(64), lds, stream, states, fstates, bitring, ring_words, descs, counts, max_frames, chlist);
		hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_IMS100>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist);
		break;
	case SONDE_MRZN1:
		hipLaunchKernelGGL(sd_sync_fixed_kernel<SONDE_MRZN1>, dim3(n_list), dim3(64), lds, stream, states, fstates, bitring, ring_words, descs, counts, max_frames, chlist);
		hipLaunchKernelGGL(sd_dec_fixed_kernel<SONDE_MRZN1>, g2, dim3(64), 0, stream, bitring, ring_words, g64, descs, counts, max_frames, frames, chlist);
		break;
	default:
		break;
	}
}
