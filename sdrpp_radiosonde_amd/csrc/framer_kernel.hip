// framer_kernel.hip -- stand-alone FEC kernel (sd_rsdec_rs41_kernel): listed RS41 frames -> de-whitening -> RS(255,231) ->
// frame records, one 64-lane wave per frame (sd_rsdec.h).  The frame-sync correlator (K4) runs inside the demodulator
// kernel, which lists the complete frames of a submit as descriptors; by default that kernel also runs the FEC in its
// epilogue (SONDE_FLAG_SPLIT_FEC selects this kernel instead: A/B measurements).
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "sd_rsdec.h"
#include "launch.h"

// 256 threads = 4 waves = 4 frames per workgroup: the 2.8 KB of GF tables are staged once per
// workgroup, each wave then works alone on its own frame (wave-scope synchronisation only), so that
// all frames of a step are resident at once and their latency-bound GF(2^8) chains overlap.
#define B2_WAVES 4
__global__ __launch_bounds__(64 * B2_WAVES) void sd_rsdec_rs41_kernel(
	const uint32_t *__restrict__ bitring, uint32_t ring_words,
	const uint8_t *__restrict__ gf_exp, const uint8_t *__restrict__ gf_log, const uint32_t *__restrict__ gf_swar /* [24][8] */,
	const SdFrameDesc *__restrict__ descs, const uint32_t *__restrict__ counts, uint32_t max_frames,
	SondeFrame *__restrict__ frames, const uint32_t *__restrict__ chlist)
{
	__shared__ __attribute__((aligned(16))) FramerTabs tabs;
	__shared__ __attribute__((aligned(16))) FramerLds wl[B2_WAVES];
	const uint32_t ch = chlist ? chlist[blockIdx.x] : blockIdx.x;
	const uint32_t nfr = min(counts[ch], max_frames);
	if (B2_WAVES * blockIdx.y >= nfr) return;                  // whole workgroup has nothing to do
	GfSwar swar;
	{
		const int tid = threadIdx.x;
		// this lane's multiplier alpha^(4j) as byte-slice tables (lane = 24 c + j of its wave)
		const uint32_t *sw = gf_swar + 8 * ((tid & 63) % RS_R);
		swar.a_lo = sw[0]; swar.a_hi = sw[1]; swar.b_lo = sw[2]; swar.b_hi = sw[3]; swar.c = sw[4];
		// antilog (zero-absorbing, GF_EXP2 bytes) and log (256 x u16) tables as the host laid them out: plain copies
		const uint4 *se = reinterpret_cast<const uint4 *>(gf_exp);
		uint4 *de = reinterpret_cast<uint4 *>(tabs.exp2);
		for (int i = tid; i < GF_EXP2 / 16; i += 64 * B2_WAVES) de[i] = se[i];
		if (tid < 512 / 16) reinterpret_cast<uint4 *>(tabs.log2)[tid] = reinterpret_cast<const uint4 *>(gf_log)[tid];
	}
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int w = threadIdx.x >> 6;
	const uint32_t k = B2_WAVES * blockIdx.y + (uint32_t)w;
	if (k >= nfr) return;
	sd_rs41_decode_frame<false>(tabs, wl[w], swar, bitring + (size_t)ch * ring_words, ring_words - 1,
		descs[(size_t)ch * max_frames + k], frames + (size_t)ch * max_frames + k, ch, lane);
}

void sd_launch_framer_rs41(uint32_t n_list, hipStream_t stream, const uint32_t *bitring, uint32_t ring_words,
	const uint8_t *gf_exp, const uint8_t *gf_log, const uint32_t *gf_swar, const void *descs,
	SondeFrame *frames, const uint32_t *counts, uint32_t max_frames, uint32_t grid_frames, const uint32_t *chlist)
{
	// channels on grid.x (no 65535 limit), frame groups on grid.y
	hipLaunchKernelGGL(sd_rsdec_rs41_kernel, dim3(n_list, (grid_frames + B2_WAVES - 1) / B2_WAVES), dim3(64 * B2_WAVES), 0, stream,
		bitring, ring_words, gf_exp, gf_log, gf_swar, (const SdFrameDesc *)descs, counts, max_frames, frames, chlist);
}

// ---- parity-test introspection: the corrector alone on caller-supplied codeword pairs (sonde_batch_test_rs255).  One wave
// per pair; cw_io holds [n_pairs][2][256] bytes (positions >= n must be zero, as the frame path builds them).
__global__ __launch_bounds__(64 * B2_WAVES) void sd_rs255_unit_kernel(uint8_t *__restrict__ cw_io, uint32_t n_pairs, int n, int32_t *__restrict__ status,
	const uint8_t *__restrict__ gf_exp, const uint8_t *__restrict__ gf_log, const uint32_t *__restrict__ gf_swar)
{
	__shared__ __attribute__((aligned(16))) FramerTabs tabs;
	__shared__ __attribute__((aligned(16))) FramerLds wl[B2_WAVES];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	GfSwar swar;
	const uint32_t *sw = gf_swar + 8 * (lane % RS_R);
	swar.a_lo = sw[0]; swar.a_hi = sw[1]; swar.b_lo = sw[2]; swar.b_hi = sw[3]; swar.c = sw[4];
	for (int i = tid; i < GF_EXP2 / 16; i += 64 * B2_WAVES) reinterpret_cast<uint4 *>(tabs.exp2)[i] = reinterpret_cast<const uint4 *>(gf_exp)[i];
	if (tid < 512 / 16) reinterpret_cast<uint4 *>(tabs.log2)[tid] = reinterpret_cast<const uint4 *>(gf_log)[tid];
	__syncthreads();
	const uint32_t k = B2_WAVES * blockIdx.x + (uint32_t)w;
	if (k >= n_pairs) return;
	FramerLds &s = wl[w];
	uint32_t *io = reinterpret_cast<uint32_t *>(cw_io + (size_t)k * 512);
	for (int i = lane; i < 128; i += 64) reinterpret_cast<uint32_t *>(s.cw[0])[i] = io[i];      // cw[0] and cw[1] are contiguous
	WAVE_SYNC();
	rs255_decode_pair(tabs, s, n, lane, swar);
	for (int i = lane; i < 128; i += 64) io[i] = reinterpret_cast<uint32_t *>(s.cw[0])[i];
	if (lane < 2) status[2 * k + lane] = s.status[lane];
}

void sd_launch_rs255_unit(uint8_t *cw_io, uint32_t n_pairs, int n, int32_t *status, const uint8_t *gf_exp, const uint8_t *gf_log,
	const uint32_t *gf_swar, hipStream_t stream)
{
	hipLaunchKernelGGL(sd_rs255_unit_kernel, dim3((n_pairs + B2_WAVES - 1) / B2_WAVES), dim3(64 * B2_WAVES), 0, stream,
		cw_io, n_pairs, n, status, gf_exp, gf_log, gf_swar);
}

