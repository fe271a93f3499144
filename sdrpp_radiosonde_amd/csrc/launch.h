// launch.h -- host-side launchers of the HIP kernels (one per kernel translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "../../include/sonde_abi.h"

// chlist: null = channels 0..n_channels-1 read their own row of `in`; else block i works on channel chlist[i] and reads
// row chlist[i] (compact_in = false: the caller's buffer) or row i (compact_in = true: a per-list scratch buffer)
// decim, nt: the decimation factor (1, 2, 4) and the taps per filter row (8, 16) of EVERY channel this launch works on: (4, 8), (2, 8), (2, 16), (1, 16)
// where the in-kernel sync search of the RS41 channels (sd_rs41.h) keeps its state and lists the complete frames
// and, for the FEC epilogue of the same kernel (sd_rsdec.h), the GF(2^8) tables and the frame slots
struct SdFramerOut {
	SdFramerState *fstates; void *descs; uint32_t *counts; uint32_t max_frames;
	uint32_t fuse_fec;                      // 1: the demod kernel decodes the listed frames in its epilogue; 0: sd_rsdec_rs41_kernel does
	uint32_t loop_fec_max_wg;               // launches of at most this many workgroups (= one residency of the GPU) and >= 48 tiles decode clean
	                                        // RS41 frames inside the tile loop (sd_rsdec.h sd_rs41_loop_step); 0: never
	const uint8_t *gf_exp, *gf_log; const uint32_t *gf_swar;
	const uint8_t *gf64;                    // GF(2^6) tables of the iMS-100 BCH decoder
	SondeFrame *frames;
	const uint16_t *m10tab;                 // Meteomodem checksum matrix rows [99][8] (sd_fixed.h sd_m10_decode_frame)
	uint32_t fixed_epi;                     // 1: the demod kernel also decodes the frames of the fixed-length framers (DFM / M10 / iMS-100 / MRZ-N1) in its
	                                        // epilogue, one frame per wave, like RS41's; 0: sd_dec_fixed_kernel does, one launch more per type
};
// TIME SLICES (round 6).  A demod workgroup lives for its channel's whole submit, so a launch whose workgroup count is not a multiple of
// one residency of the GPU ends with a part-filled generation running alone, and any launch ends with a ramp-down as long as a
// workgroup's life.  With n_seg > 1 a channel's submit is cut into n_seg consecutive SEGMENTS of seg_tiles tiles, each its own
// workgroup: grid = n_wg * n_seg, block b works on segment b / n_wg of list entry b % n_wg (segment-major: every channel's first
// segment is dispatched before any second one).  A segment is what a submit is to the arithmetic -- the channel's whole state is
// carried in HBM, the SPEC does not see where a submit is cut (tests: ragged submits) -- so the frames are those of the unsliced
// launch, bit for bit.  Segment s of a channel waits (one lane polls an agent-scope word, the others sit at a barrier) until segment
// s - 1 has published prog[channel] = seg_base + s; the predecessor has a lower block index, so it was dispatched earlier and is
// running or done: no deadlock as long as workgroups are dispatched in index order per XCD.  Should that ever not hold, the poll
// gives up after ~0.3 s, sets prog[err_index] and the workgroup leaves without touching the channel: the host reports it (never a hang).
struct SdSlice {
	int32_t   n_seg;        // segments per channel (1: unsliced: the fields below are not read)
	int32_t   seg_tiles;    // tiles per segment (the last one takes what is left)
	uint32_t  n_wg;         // list entries (channels) of the launch
	uint32_t  seg_base;     // prog[channel] == seg_base when the launch starts (the host advances it by n_seg per sliced launch)
	uint32_t *prog;         // [n_channels of the batch + 1]: per channel the segments completed so far; the last word: poll gave up
	uint32_t  err_index;    // = n_channels of the batch
};
// which launches have a time-sliced instantiation (the two default classes, IQ input of every kind); the host asks before it slices
bool sd_slices_supported(int in_kind, int decim, int nt);
// in_kind: what the 48 kS/s rows hold (SD_IN_REAL / SD_IN_IQ / SD_IN_IQ16 / SD_IN_IQ8)
void sd_launch_demod(int in_kind, int decim, int nt, uint32_t n_channels, hipStream_t stream,
	const float *in, size_t ch_stride, int n_tiles, SdChanState *states, float *hist,
	uint32_t *bitring, uint32_t ring_words, const float *taps_all, const SdModem *modems,
	const uint32_t *chlist, bool compact_in, const SdFramerOut *fo /* DEVICE memory: the kernel reads it on demand */,
	int utype /* >= 0: every channel of the launch is of this sonde type (taps / modem loads need not wait for the state); -1: per channel */,
	const SdSlice *slice = nullptr /* null or n_seg <= 1: one workgroup per channel for the whole submit */);
// One launch over the channels of the two default classes: list_a = the (4, 8) class's channels (RS41, DFM, iMS-100, MRZ-N1), list_b = the
// (2, 8) class's (M10); utype_x >= 0: every channel of that list is of this sonde type (demod_kernel.hip sd_demod_mixed_kernel)
void sd_launch_demod_mixed(int in_kind, uint32_t n_a, const uint32_t *list_a, int utype_a, uint32_t n_b, const uint32_t *list_b, int utype_b, hipStream_t stream,
	const float *in, size_t ch_stride, int n_tiles, SdChanState *states, float *hist, uint32_t *bitring, uint32_t ring_words,
	const float *taps_all, const SdModem *modems, const SdFramerOut *fo /* DEVICE memory */);
// The decoder behind the filter bank (bins_kernel.hip): one wave per bin; rows of 16-bit phases [16 carried | n_steps]; the last 16
// phases of the submit go to the head of carry_rows' rows (the buffer the next submit reads: the same one unless double-buffered)
struct SdBinsArgs { const int16_t *phases; size_t row_stride; int16_t *carry_rows; size_t carry_stride; const float *g_comp /* [3][SD_RS_KT_LD], device */; };
void sd_launch_bins(uint32_t n_channels, hipStream_t stream, const int16_t *phases, size_t row_stride, int n_blocks,
	int16_t *carry_rows, size_t carry_stride, SdChanState *states, float *hist, uint32_t *bitring, uint32_t ring_words,
	const float *taps_all, const SdModem *modems_host /* [SONDE_NTYPES], HOST memory: passed by value */, const SdFramerOut *fo_host /* HOST copy: by value */,
	const float *g_comp, int utype /* >= 0: every bin is of this sonde type; -1: per bin */);
// the batch object behind a channelizer takes its input as phase rows (channelizer.hip): 3 tiles per 2560 phase samples
int sd_batch_submit_bins(SondeBatch *b, const SdBinsArgs *ba, size_t n_steps, void *stream);
int sd_batch_bins_capable(const SondeBatch *b);      // 1: every channel's class has a bins instantiation (no AFSK sonde, no class without one)

void sd_launch_afsk(int type /* SONDE_IMET4 or SONDE_C50 */, int kind /* 0 real, 1 complex64, 2 int16 IQ pairs */, uint32_t n_list, hipStream_t stream, const float *in, size_t ch_stride, int n_tiles,
	const uint32_t *chlist, SdAfskState *astates, const float *wtab, float *out, size_t out_stride);
void sd_launch_framer_imet(uint32_t n_list, hipStream_t stream, const SdChanState *states, SdFramerState *fstates,
	const uint32_t *bitring, uint32_t ring_words, SondeFrame *frames, uint32_t *counts, uint32_t max_frames, const uint32_t *chlist);
void sd_launch_framer_c50(uint32_t n_list, hipStream_t stream, const SdChanState *states, SdFramerState *fstates,
	const uint32_t *bitring, uint32_t ring_words, SondeFrame *frames, uint32_t *counts, uint32_t max_frames, const uint32_t *chlist);

#define SD_DESC_BYTES 16   // sizeof(SdFrameDesc)
void sd_launch_framer_rs41(uint32_t n_list, hipStream_t stream, const uint32_t *bitring, uint32_t ring_words,
	const uint8_t *gf_exp, const uint8_t *gf_log, const uint32_t *gf_swar, const void *descs,
	SondeFrame *frames, const uint32_t *counts, uint32_t max_frames, uint32_t grid_frames, const uint32_t *chlist);

void sd_launch_framer_other(int type, uint32_t n_list, hipStream_t stream,
	const SdChanState *states, SdFramerState *fstates, const uint32_t *bitring, uint32_t ring_words,
	const uint8_t *g64, void *descs, SondeFrame *frames, uint32_t *counts, uint32_t max_frames, uint32_t grid_frames, const uint32_t *chlist,
	bool with_sync /* false: the demod kernel has listed the frames already */);

// parity-test introspection: the RS(255,231) corrector on caller-supplied codeword pairs ([n_pairs][2][256] bytes, device memory)
void sd_launch_rs255_unit(uint8_t *cw_io, uint32_t n_pairs, int n, int32_t *status, const uint8_t *gf_exp, const uint8_t *gf_log,
	const uint32_t *gf_swar, hipStream_t stream);

// sets the text sonde_last_error() returns; returns -1
int sd_fail(const char *what, hipError_t e = hipSuccess);
