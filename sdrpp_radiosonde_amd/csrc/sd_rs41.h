// sd_rs41.h -- RS41 framing constants and the frame-sync correlator (K4) that runs INSIDE
// the demodulator kernel: one round wave advances the channel's sync-search state machine once per tile over the
// newest bits, which the lead wave mirrors in LDS while it appends them to the bit ring in HBM.  Complete frames
// are listed as descriptors for the FEC kernel (framer_kernel.hip).  This replaces a separate sync-search kernel
// (12.8 us per step plus a launch boundary in round 1).
//
// It stands where sondedump's framer/correlator sits behind rs41_decode (/root/reference/src/main.hpp:36,
// /root/reference/src/decode/decoder.hpp:61).  Search semantics (SPEC 3.3, oracle/or_fec.c rs41_run): positions
// are tried in ascending order from `rpos`; the 64-bit window is compared with the on-air header, Hamming distance
// <= 6 (>= 58: inverted polarity) starts a frame; its length follows from the de-whitened type byte 56; once the
// last bit of the frame has arrived it is listed and the search resumes right behind it.  All integer work.
#pragma once
#include <hip/hip_runtime.h>
#include "sonde_dev.h"

#define SD_K4_LIST 8
#define RS41_SYNC_THR 6
#define RS41_LEN_STD  320
#define RS41_LEN_EXT  518
#define RS41_TYPE_POS 56
#define RS41_TYPE_MASK 0x78u        // whitening mask byte 56

// on-air RS41 header 10 B6 CA 11 22 96 12 F8, LSB-first bit order => little-endian words
#define RS41_SYNC_LO 0x11CAB610u
#define RS41_SYNC_HI 0xF8129622u

struct SdSyncRun {                  // working copy of SdFramerState + the frames listed so far; lives in LDS between steps
	uint64_t rpos, fstart;          // (scalar registers are scarce in the demod kernel: 80 at 8 waves per SIMD)
	int32_t collecting, inv, flen;
	uint32_t nout;
	uint64_t wp_seen;               // bits the lead wave has announced (and mirrored) so far
	SdFrameDesc list[SD_K4_LIST];   // the first frames listed in this launch, for the FEC epilogue (the full list is in HBM)
};

__device__ __forceinline__ uint64_t sd_uniform64(unsigned long long v)    // a value all lanes hold alike -> scalar registers
{
	return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
	       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// Advance the state machine over the bits [.., wp) of the channel.  `mirror` holds the ring words of the newest
// SD_MIRROR_WORDS * 32 bits (word w of the stream at mirror[w % SD_MIRROR_WORDS]); the caller guarantees that the
// search never trails wp by more than that (it is called at least once per tile).  Wave-synchronous, 64 lanes,
// all control flow wave-uniform.
// REG: `lds_state` is a register-resident copy owned by the calling wave alone (bins_kernel.hip: one wave per channel), its list of
// the first frames lives at `list` (LDS); else the state sits in LDS between steps (kernel A) and list = lds_state.list.
template <bool REG = false, int NCHUNK = (REG ? 4 : 1)>
__device__ __forceinline__ void sd_rs41_sync_step(SdSyncRun &lds_state, uint64_t wp, const uint32_t *mirror, int lane,
	SdFrameDesc *__restrict__ descs_ch, uint32_t max_frames, SdFrameDesc *list = nullptr)
{
	if (!REG) list = lds_state.list;
	SdSyncRun fs;
	fs.rpos = sd_uniform64(lds_state.rpos); fs.fstart = sd_uniform64(lds_state.fstart);
	fs.collecting = __builtin_amdgcn_readfirstlane(lds_state.collecting);
	fs.inv = __builtin_amdgcn_readfirstlane(lds_state.inv);
	fs.flen = __builtin_amdgcn_readfirstlane(lds_state.flen);
	fs.nout = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_state.nout);
	// nothing to do in the common case: a frame is being collected and its end has not arrived
	if (fs.collecting && fs.flen && wp < fs.fstart + 8 * (uint64_t)fs.flen) return;
	if (REG && !fs.collecting) {
		// The other common case, as straight-line code (round 5; tools/bk_ts.py: the general loop below costs a wave of the bins decoder
		// 2 000 of a tile's 6 000 ticks on a bin without a transmitter -- 1 100 for four chunks behind 64-bit position compares, 700 for
		// the scalar tail of hit tests and loop exits): searching, and ONE trip of NCHUNK chunks covers every candidate whose window is
		// complete.  Without a hit the loop would leave rpos = wp - 63 and nothing else changed: exactly that, behind one branch.
		const uint32_t avail = (uint32_t)(wp - fs.rpos);               // (the search never trails by more than the mirror: < 2^11)
		if (wp >= fs.rpos && avail < 64u) return;                          // no candidate with a complete window yet
		if (wp >= fs.rpos && avail <= 64u * NCHUNK + 63u) {
			const uint32_t b0 = (uint32_t)fs.rpos;
			unsigned long long any = 0;
#pragma unroll
			for (int j = 0; j < NCHUNK; j++) {
				const uint32_t pb = b0 + (uint32_t)(64 * j + lane);           // the position's low 32 bits: word index and shift are all it gives
				const uint32_t wi = (uint32_t)((fs.rpos + (uint64_t)(64 * j)) >> 5) + ((((uint32_t)fs.rpos & 31u) + (uint32_t)lane) >> 5), sh = pb & 31u;
				const uint32_t w0 = mirror[wi & (SD_MIRROR_WORDS - 1)], w1 = mirror[(wi + 1) & (SD_MIRROR_WORDS - 1)],
				               w2 = mirror[(wi + 2) & (SD_MIRROR_WORDS - 1)];
				const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
				const int hd = __popc(lo ^ RS41_SYNC_LO) + __popc(hi ^ RS41_SYNC_HI);
				const bool hit = (uint32_t)(64 * j + lane) + 64u <= avail && (hd <= RS41_SYNC_THR || hd >= 64 - RS41_SYNC_THR);
				any |= __ballot(hit);
			}
			if (!any) { lds_state.rpos = wp - 63; return; }
		}
	}
	for (;;) {
		if (!fs.collecting) {
			bool found = false;
			while (fs.rpos + 64 <= wp) {
				// NCH chunks of 64 candidate positions per trip (REG: a wave that sees a whole tile's bits at once: one LDS round trip
				// instead of four); positions are still tried in ascending order: the earliest hit of the earliest chunk wins
				constexpr int NCH = NCHUNK;
				unsigned long long hm[NCH];
				int hdv[NCH];
#pragma unroll
				for (int j = 0; j < NCH; j++) {
					const uint64_t pos = fs.rpos + (uint64_t)(64 * j + lane);
					const uint32_t wi = (uint32_t)(pos >> 5), sh = (uint32_t)pos & 31u;
					const uint32_t w0 = mirror[wi & (SD_MIRROR_WORDS - 1)], w1 = mirror[(wi + 1) & (SD_MIRROR_WORDS - 1)],
					               w2 = mirror[(wi + 2) & (SD_MIRROR_WORDS - 1)];
					const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
					const int hd = __popc(lo ^ RS41_SYNC_LO) + __popc(hi ^ RS41_SYNC_HI);
					const bool hit = pos + 64 <= wp && (hd <= RS41_SYNC_THR || hd >= 64 - RS41_SYNC_THR);
					hm[j] = __ballot(hit);
					hdv[j] = hd;
				}
#pragma unroll
				for (int j = 0; j < NCH; j++) {
					if (!found && hm[j]) {
						const int fl = __ffsll((long long)hm[j]) - 1;         // the earliest position wins
						fs.fstart = fs.rpos + (uint64_t)(64 * j + fl);
						fs.inv = __builtin_amdgcn_readlane(hdv[j], fl) >= 64 - RS41_SYNC_THR;
						fs.collecting = 1;
						fs.flen = 0;
						found = true;
					}
				}
				if (found) break;
				uint64_t next = fs.rpos + 64 * NCH;
				if (next > wp - 63) next = wp - 63;                       // first position whose window is not complete yet
				fs.rpos = next;
			}
			if (!found) break;
		}
		if (!fs.flen) {
			if (wp < fs.fstart + 8 * (RS41_TYPE_POS + 1)) break;
			const uint64_t p = fs.fstart + 8 * RS41_TYPE_POS;
			const uint32_t wi = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
			const uint32_t raw = __builtin_amdgcn_alignbit(mirror[(wi + 1) & (SD_MIRROR_WORDS - 1)], mirror[wi & (SD_MIRROR_WORDS - 1)], sh);
			const uint32_t tb = (raw ^ (fs.inv ? 0xFFu : 0u) ^ RS41_TYPE_MASK) & 0xFFu;
			fs.flen = (__popc(tb ^ 0xF0u) < __popc(tb ^ 0x0Fu)) ? RS41_LEN_EXT : RS41_LEN_STD;
		}
		if (wp < fs.fstart + 8 * (uint64_t)fs.flen) break;
		if (fs.nout < max_frames && lane == 0) {
			SdFrameDesc d;
			d.fstart = fs.fstart; d.flen = fs.flen; d.inv = fs.inv;
			descs_ch[fs.nout] = d;
			if (fs.nout < SD_K4_LIST) list[fs.nout] = d;
		}
		fs.nout++;
		fs.rpos = fs.fstart + 8 * (uint64_t)fs.flen;
		fs.collecting = 0;
		fs.flen = 0;
	}
	if (REG) {
		lds_state.rpos = fs.rpos; lds_state.fstart = fs.fstart;
		lds_state.collecting = fs.collecting; lds_state.inv = fs.inv; lds_state.flen = fs.flen; lds_state.nout = fs.nout;
		return;
	}
	if (lane == 0) {
		lds_state.rpos = fs.rpos; lds_state.fstart = fs.fstart;
		lds_state.collecting = fs.collecting; lds_state.inv = fs.inv; lds_state.flen = fs.flen;
		// nout is what ANOTHER wave (the in-loop clean-frame decoder, sd_rs41_loop_step on round wave 2) reads in the same round with
		// no barrier in between: published behind the descriptors it announces (workgroup-scope release)
		__hip_atomic_store(&lds_state.nout, fs.nout, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	// the next step (same wave) reads the state back: DS operations of a wave execute in order
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
