// sonde_dev.h -- device-side state layout and SPEC constants shared by the HIP kernels and the
// host engine of libsonde_mi355.so.  (Product code: never includes anything from oracle/.)
//
// The arithmetic contract these kernels implement is DESIGN.md section 3; it stands where sondedump's
// gfsk_demod/framer/rs sit behind X_decode (/root/reference/src/decode/decoder.hpp:22,61).
#pragma once
#include <stdint.h>

#define SD_FS         48000
#define SD_TILE       2048        // samples per tile
#define SD_RING       4096        // discriminator ring, floats (LDS)
#define SD_NTAPS      32          // row length of the polyphase tap table
// taps in use per row: 3.2 symbols (SPEC 3.2) = 8 below 3.5 samples per symbol, else 16; the 6 kS/s AFSK streams (pre = 8) keep 16
#define SD_NT_OF(period0, pre)  (((pre) == 1 && 2 * (int64_t)(period0) < 7 * 65536) ? 8 : 16)
#define SD_NPHASE     32
#define SD_TAPS_LD    36          // padded leading dimension of the tap table in LDS (16-B aligned rows)
#define SD_ROUND_MAX  256         // symbols per timing-loop round = round lanes of the demod kernel; streams whose tile holds more
// (M10 at 24 kS/s / 2.5 samples per chip, anything at 48 or 6 kS/s with 5 samples per symbol) take two symbols per lane: rounds of <= 512
#define SD_ROUND_SPL(decim, nt) (((SD_TILE / (decim)) / ((nt) == 8 ? 2 : 4)) > SD_ROUND_MAX ? 2 : 1)
#define SD_HIST       64          // discriminator samples carried between submits
#define SD_MARGIN     4           // look-ahead slack (samples) behind the newest sample, SPEC 3.2
#define SD_WG         256

// what the demod kernel's input rows hold
#define SD_IN_REAL 0        // 48 kS/s FM-discriminator samples (float)
#define SD_IN_IQ   1        // 48 kS/s complex samples
#define SD_IN_IQ16 3        // 48 kS/s complex samples as 16-bit integers (I, Q interleaved: what SDR hardware and WAV recordings hold): half the
                            // bytes of SD_IN_IQ per sample; converted exactly (int16 -> float, no scaling) in the load path, then SD_IN_IQ's arithmetic
#define SD_IN_IQ8  4        // the same as 8-bit integers (int8 I, int8 Q): 2 bytes per sample
#define SD_RS_KT_LD 20      // row stride of the composite 12/5 resampler - 4:1 decimator taps of SPEC 3.5b (17 in use; bins_kernel.hip)
#define SD_RS_KT    17
struct SdModem {            // per sonde type, built on the host
	int32_t period0;        // Q16 internal-rate samples per symbol
	float   kp;             // proportional gain, Q16 samples per unit error
	float   ki;             // integral gain
	int32_t pmin, pmax;     // period clamp
	int32_t rounds;         // timing-loop rounds per tile: 1; 2 for the SRS-C50 6 kS/s stream (822 symbols per tile, rounds of <= 512)
	int32_t decim;          // IQ boxcar-decimated decim:1 before the discriminator: 4 RS41 / DFM / iMS-100 / MRZ-N1 (12 kS/s), 2 M10 (SPEC 3.0)
	int32_t itile;          // internal samples per 2048-sample input tile = 2048 / decim
	int32_t nt;             // taps in use per polyphase row: 3.2 symbols = 8 (2.5 samples per symbol) or 16 (5): SD_NT_OF
};

#define SD_AF_DEC 8        // AFSK tone demodulator (SPEC 3.6): 48 kS/s -> 6 kS/s
#define SD_AF_PER 480      // iMet mixer table period: 17 cycles of 1700 Hz at 48 kS/s
#define SD_C50_PER 240     // SRS-C50 mixer table period: 19 cycles of 3800 Hz at 48 kS/s
struct SdAfskState {        // tone-demodulator state, one per channel (64 B)
	float    iq_last[2];    // previous IQ sample of the first discriminator
	float    b[4][2];       // the four 8-sample block sums before the next one, oldest first
	float    z[2];          // previous boxcar output
	uint64_t n;             // input samples consumed (mixer phase = n mod 480)
	uint32_t pad[2];
};

struct SdChanState {        // demodulator state, one per channel (64 B)
	int64_t  t_next;        // Q16 absolute on-time instant of the next symbol
	int64_t  n0;            // samples consumed
	uint64_t wpos;          // bits produced
	int32_t  period;        // Q16
	int32_t  nstat;
	float    bias;
	float    amp;
	float    iq_last[2];    // previous IQ sample (I, Q) of the discriminator
	int32_t  type;
	float    afc[3];        // SPEC 3.0b: the AFC states u the discriminator of the next three tiles uses, oldest first (IQ input)
};
static_assert(sizeof(SdChanState) == 64, "SdChanState is 64 bytes");
// AFC (SPEC 3.0b): rotation 2 atan(u) per internal sample; per tile u <- clamp(u - LEAK u + GAIN bias, +-MAX)
#define SD_AFC_GAIN 0.19634954f
#define SD_AFC_LEAK 0.0078125f
#define SD_AFC_MAX  0.8f

struct SdFramerState {      // framer state, one per channel (32 B)
	uint64_t rpos;          // the sync search resumes at this absolute bit index
	uint64_t fstart;        // collecting: absolute bit index of the first sync bit
	int32_t  collecting;
	int32_t  inv;
	int32_t  flen;          // RS41 (framed inside the demod kernel): frame length in bytes once the type byte has arrived, else 0
	int32_t  pad;
};

struct SdFrameDesc {        // one complete frame located by the sync search, decoded by the FEC kernel
	uint64_t fstart;        // absolute bit index of the first sync bit
	int32_t  flen;          // bytes (RS41) or chips (the fixed-length framers)
	int32_t  inv;           // polarity
};

// newest bits of a channel's bit ring kept in LDS by the demod kernel for its in-kernel sync search
#define SD_MIRROR_WORDS 64
