/*
 * sonde_abi.h -- C ABI of libsonde_mi355.so, the MI355X-native radiosonde demod+FEC path.
 *
 * Two layers, both plain C (no torch / HIP types in any signature):
 *
 *  B1  the per-sonde decoder triple that /root/reference/src/decode/decoder.hpp:22 fixes as
 *      template parameters and /root/reference/src/main.hpp:36-42 instantiates:
 *          T*           X_decoder_init(int samplerate);
 *          void         X_decoder_deinit(T*);
 *          ParserStatus X_decode(T*, SondeData *dst, const float *src, size_t len);
 *      for X in {rs41, dfm09, ims100, m10, imet4, c50, mrzn1}.  These replace the
 *      un-vendored sondedump library (CMake target `radiosonde`, /root/reference/CMakeLists.txt:18,25)
 *      and are what "drops in for src/decode/".
 *
 *  B0  the batch API the B1 shims sit on: many independent 48 kS/s channels, one HIP workgroup
 *      per channel, complex IQ (the dsp::stream<dsp::complex_t> of /root/reference/src/main.cpp:57)
 *      or real FM-discriminator samples (the dsp::stream<float> of decoder.hpp:35) in, corrected
 *      frames out.
 *
 * SondeData / ParserStatus / DATA_* are *defined here*: the reference only uses them
 * (/root/reference/src/decode/decoder.hpp:54,61,64-104); their definitions lived in the absent
 * sondedump headers.
 */
#ifndef SONDE_ABI_H
#define SONDE_ABI_H

#include <stddef.h>
#include <stdint.h>
#include <time.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ B1: sondedump surface */

/* decoder.hpp:61 tests `!= PROCEED`; PARSED = "one fragment is in *dst". */
typedef enum { PROCEED = 0, PARSED = 1 } ParserStatus;

/* validity bitmask of SondeData.fields -- names from decoder.hpp:64,68,74,80,84,93,97,102 */
#define DATA_SEQ      (1 << 0)
#define DATA_POS      (1 << 1)
#define DATA_SPEED    (1 << 2)
#define DATA_TIME     (1 << 3)
#define DATA_PTU      (1 << 4)
#define DATA_SERIAL   (1 << 5)
#define DATA_SHUTDOWN (1 << 6)
#define DATA_OZONE    (1 << 7)

/* members and types implied by the assignments at decoder.hpp:65-104 */
typedef struct {
	int    fields;
	int    seq;
	float  lat, lon, alt;          /* degrees, degrees, metres */
	float  speed, heading, climb;  /* m/s, degrees, m/s */
	time_t time;
	float  calib_percent;
	float  temp, rh, pressure;     /* degC, %, hPa (<= 0: unknown -> decoder.hpp:108 falls back to ISA) */
	char   serial[32];
	int    shutdown;               /* seconds to burst-kill shutdown, -1 inactive */
	float  o3_mpa;
} SondeData;

typedef struct SondeB1Decoder RS41Decoder;
typedef struct SondeB1Decoder DFM09Decoder;
typedef struct SondeB1Decoder IMS100Decoder;
typedef struct SondeB1Decoder M10Decoder;
typedef struct SondeB1Decoder IMET4Decoder;
typedef struct SondeB1Decoder C50Decoder;
typedef struct SondeB1Decoder MRZN1Decoder;

#define SONDE_B1_DECL(T, x) \
	T *x##_decoder_init(int samplerate); \
	void x##_decoder_deinit(T *d); \
	ParserStatus x##_decode(T *d, SondeData *dst, const float *src, size_t len);

SONDE_B1_DECL(RS41Decoder,   rs41)    /* main.hpp:36 */
SONDE_B1_DECL(DFM09Decoder,  dfm09)   /* main.hpp:37 */
SONDE_B1_DECL(IMS100Decoder, ims100)  /* main.hpp:38 -- EXPERIMENTAL field parser: demod + BCH(63,51) follow the public structure, the word OFFSETS
                                         inside the 408 data bits are this repo's own, not a recorded sonde's */
SONDE_B1_DECL(M10Decoder,    m10)     /* main.hpp:39 */
SONDE_B1_DECL(IMET4Decoder,  imet4)   /* main.hpp:40 */
SONDE_B1_DECL(C50Decoder,    c50)     /* main.hpp:41 -- EXPERIMENTAL field parser: AFSK demod, packet framing and checksums are the public structure, the value
                                         SCALINGS are this repo's own */
SONDE_B1_DECL(MRZN1Decoder,  mrzn1)   /* main.hpp:42 -- EXPERIMENTAL field parser: demod, framing and CRC are the public structure, the byte OFFSETS are this
                                         repo's own.  For these three the generator and the parser share the layout: their round-trip tests
                                         cannot catch a wrong one; RS41, DFM, M10 / M20 and iMet layouts follow the public decoders */

/* ------------------------------------------------------------------ B0: batch API */

/* sonde types, order of supportedTypes[] at main.hpp:44-52 */
enum { SONDE_RS41 = 0, SONDE_DFM09 = 1, SONDE_IMS100 = 2, SONDE_M10 = 3, SONDE_IMET4 = 4, SONDE_C50 = 5, SONDE_MRZN1 = 6, SONDE_NTYPES = 7 };

enum { SONDE_INPUT_IQ = 0,      /* complex64 interleaved I,Q at 48 kS/s (vfo->output level, main.cpp:57) */
       SONDE_INPUT_REAL = 1,    /* float FM-discriminator output at 48 kS/s (decoder.hpp:35 level) */
       SONDE_INPUT_IQ8 = 3,     /* the same as 8-bit integers: int8 I, int8 Q interleaved, 2 bytes per sample (signed; an offset-binary stream
                                 * such as RTL-SDR's cu8 becomes this by XOR 0x80 on every byte).  Enough for a sonde channel: with the signal
                                 * at 8 counts or more nothing is lost against float (profiles/r4_iq16_scale.md).  A quarter of the bytes. */
       SONDE_INPUT_IQ16 = 2 };  /* the vfo->output level as 16-bit integers: int16 I, int16 Q interleaved at 48 kS/s, 4 bytes per sample --
                                 * what SDR hardware and WAV recordings hold before SDR++'s sources convert to float.  Converted in the
                                 * kernel's load path (exactly; no scaling: the discriminator does not depend on the amplitude), so the
                                 * frames are those of SONDE_INPUT_IQ fed with the same integers as floats, for half the bytes over PCIe,
                                 * xGMI and HBM.  Batch / node API, all seven sonde types. */

#define SONDE_TILE       2048   /* samples; submit lengths are multiples of this */
#define SONDE_FRAME_MAX  528

typedef struct {
	uint32_t channel;
	uint32_t type;
	int32_t  len;            /* bytes valid in data[] */
	int32_t  nerr[2];        /* RS41: bytes corrected per RS codeword, -1 = uncorrectable */
	uint32_t flags;          /* bit0: signal polarity was inverted */
	uint64_t bitpos;         /* absolute index (since create) of the first sync bit */
	uint8_t  data[SONDE_FRAME_MAX];   /* de-whitened, error-corrected frame */
} SondeFrame;

typedef struct {
	uint32_t       n_channels;
	const uint8_t *types;            /* n_channels entries of SONDE_*; NULL = all SONDE_RS41 */
	uint32_t       max_samples;      /* largest samples-per-channel of one submit (multiple of SONDE_TILE) */
	int32_t        input_kind;       /* SONDE_INPUT_IQ, SONDE_INPUT_REAL, SONDE_INPUT_IQ16 or SONDE_INPUT_IQ8 */
	int32_t        device;           /* HIP device ordinal */
	uint32_t       flags;            /* SONDE_FLAG_*; 0 = defaults */
	uint32_t       launch_units;     /* 0 = the library's choice; else the number of launch units a batch of ONE sonde type is cut into when its
	                                    submits are not joined at every call (1..16; measurements: profiles/r4_units_sweep.txt) */
	uint32_t       struct_size;      /* sizeof(SondeBatchConfig) as the caller compiled it (SONDE_BATCH_CONFIG_INIT sets it).  sonde_batch_create
	                                    REFUSES any other value: a caller built against an older, shorter header, or one that did not zero the
	                                    struct, fails loudly instead of handing over garbage in the members it does not know */
	uint32_t       time_slices;      /* 0 = the library's choice; 1 = never; n = cut every channel's submit into n consecutive segments, each its own
	                                    workgroup (round 6: a launch whose workgroup count does not fill whole residencies of the GPU no longer ends with
	                                    a part-filled generation running alone; same frames, bit for bit: a segment is to the arithmetic what a submit is;
	                                    DESIGN 5).  Measurements and tests set it; hosts leave it 0 */
} SondeBatchConfig;
/* SondeBatchConfig cfg = SONDE_BATCH_CONFIG_INIT;  -- everything zero (= defaults), struct_size filled in */
#define SONDE_BATCH_CONFIG_INIT { 0, NULL, 0, 0, 0, 0, 0, (uint32_t)sizeof(SondeBatchConfig), 0 }

/* One decimation step less before the discriminator for every GFSK sonde (RS41 / DFM / iMS-100 / MRZ-N1 2:1 instead of 4:1:
 * 24 kS/s internally; M10 none instead of 2:1: 48 kS/s): tolerates about twice the carrier offset (+-5 kHz instead of +-2 kHz
 * for RS41, +-10 kHz instead of +-4.5 kHz for M10) at about 2 dB of sensitivity and twice the discriminator arithmetic.
 * IQ input only. */
#define SONDE_FLAG_WIDE      1u
#define SONDE_FLAG_RS41_WIDE SONDE_FLAG_WIDE     /* the flag's name in rounds 1-2, when it moved RS41 only */
/* RS41 channels: run the Reed-Solomon stage as a kernel of its own behind the demodulator instead of in the demodulator
 * kernel's epilogue (one launch more per submit; same frames).  Kept for A/B measurements. */
#define SONDE_FLAG_SPLIT_FEC 2u
/* HOW A SUBMIT COMPLETES ON THE CALLER'S STREAM.
 * DEFAULT (flags 0; rounds 1-4's behaviour, round 6 again): ORDINARY STREAM SEMANTICS.  When sonde_batch_submit returns, every kernel
 * of the submit is ordered into `stream`: work queued on that stream afterwards -- an asynchronous copy into the same sample buffer, a
 * kernel that refills it -- runs behind the submit's last reader.  This is what /root/reference/src/decode/decoder.hpp:59-117 assumes of
 * X_decode (the stream buffer is flushed right after the call returns).  Nothing to remember, nothing to get wrong.
 *
 * Internally a batch of several demodulator classes, or of one class whose channel count does not fill whole residencies of the GPU,
 * is cut into LAUNCH UNITS on the library's own streams, forked from the caller's stream at each submit (a workgroup lives for its
 * channel's whole submit: a launch whose workgroup count is not a multiple of one residency ends with a part-filled generation
 * running alone).  By default the units are joined back into the caller's stream before sonde_batch_submit returns.  Two OPT-IN modes
 * trade the stream guarantee for overlap BETWEEN submits (the tail of one unit's submit t beside the other units' submit t + 1):
 *   SONDE_FLAG_LATE_JOIN  one submit LATE: sonde_batch_submit t makes the caller's stream wait for the units of submit t - 1 only.
 *                         Work queued on the caller's stream after submit t returns is NOT ordered behind submit t's readers: the host
 *                         either double-buffers its sample blocks or calls sonde_batch_wait_input(b, stream) before it queues the
 *                         work that overwrites the block (a device-side wait, no host synchronisation).  (This was the default in
 *                         round 5: a host that refilled its one buffer on the same stream got corrupted input without any error --
 *                         VERDICT r5, ADVICE r5 -- hence opt-in now.)
 *   SONDE_FLAG_PIPELINE   never joined: the caller's stream orders the INPUT only; completion is observed through sonde_batch_sync /
 *                         sonde_batch_frames_of, the buffer is released by sonde_batch_wait_input or by one of those calls.
 * Frames are observed through sonde_batch_sync / frames / frames_of / poll in every mode.  The contract depends on the flags ONLY --
 * never on the device or on how many units the library chose (sonde_batch_launch_info reports both). */
#define SONDE_FLAG_PIPELINE  4u
/* SONDE_FLAG_WIDE for the sonde types whose channel is 20 kHz or wider in the reference only (iMS-100 / RS-11G and MRZ-N1: 20 kHz,
 * M10 / M20: 50 kHz; /root/reference/src/main.hpp:47-51); RS41 (10 kHz) and DFM (15 kHz) keep the default classes.  The per-type
 * choice sonde::IqStreamDecoder makes for its one channel, for a batch of mixed types.  IQ input only. */
#define SONDE_FLAG_WIDE_AUTO 8u
#define SONDE_FLAG_JOIN      16u     /* the default since round 6 (accepted and ignored; round 5: the opt-out of its lagging default) */
#define SONDE_FLAG_LATE_JOIN 32u     /* see SONDE_FLAG_PIPELINE above */

typedef struct SondeBatch SondeBatch;

/* All int-returning calls: 0 = ok, negative = error (text via sonde_last_error()). */
int  sonde_batch_create(const SondeBatchConfig *cfg, SondeBatch **out);
void sonde_batch_destroy(SondeBatch *b);

/* Demodulate + frame + FEC n_samples more samples of every channel.
 * samples: DEVICE pointer, channel-major; channel c starts at element c*channel_stride
 * (elements = complex samples for IQ, floats for REAL).  n_samples % SONDE_TILE == 0.
 * stream: hipStream_t (NULL = default stream).  Asynchronous.  Submits are ordered among themselves even across
 * streams (per-channel state is carried).  The frames of a submit live until the submit after the next one starts.
 * Layout advice (measured): the kernel streams every channel's row at once, and how far apart the rows lie decides how
 * they spread over the HBM channels: 1024 rows of 1.5 MiB run 2.3-5.5 % faster 2 MiB apart than back to back, while a stride a
 * little off a power of two (2064 KiB) is 7 % slower.  sonde_row_stride() returns the stride (in elements) the library uses for
 * its own staging buffer: the next power of two in bytes where that costs at most a third more memory (1.5 MiB rows -> 2 MiB),
 * else the next odd multiple of 64 KiB (rows of 1.01 MiB -> 1.0625 MiB, not 2 MiB); rows below 64 KiB stay back to back. */
size_t sonde_row_stride(size_t n_samples, int input_kind);
/* bytes per sample (element) of an input kind: 8 (IQ), 4 (REAL, IQ16), 2 (IQ8) */
size_t sonde_sample_bytes(int input_kind);
int  sonde_batch_submit(SondeBatch *b, const void *samples, size_t n_samples, size_t channel_stride, void *stream);
/* Same, from HOST memory (staged through an internal pinned/device buffer; PCIe-inclusive). */
int  sonde_batch_submit_host(SondeBatch *b, const void *samples, size_t n_samples, size_t channel_stride);
/* Make `stream` wait ON THE DEVICE (no host synchronisation) until every reader of the LAST submit's sample buffer is done: after this
 * call, work queued on `stream` may overwrite or free that buffer.  With default flags and `stream` = the submit's stream it adds
 * nothing (stream order already says so); it is what a SONDE_FLAG_LATE_JOIN / SONDE_FLAG_PIPELINE host calls before it refills a
 * single buffer, and what any host calls to release the buffer on ANOTHER stream (a copy engine's).  0 = ok. */
int  sonde_batch_wait_input(SondeBatch *b, void *stream);
/* Wait for the last submit; returns number of frames it produced (or negative error). */
long sonde_batch_sync(SondeBatch *b);
/* Copy the last submit's frames to host memory, ordered by (channel, bitpos).  Returns count copied. */
long sonde_batch_frames(SondeBatch *b, SondeFrame *out, size_t cap);
/* Pipelined hosts: frame slots exist twice, so the frames of submit t stay readable while submit t + 1 is queued or running.
 * sonde_batch_ticket: the number of the last submit (1-based; 0 = none yet).  sonde_batch_frames_of waits for THAT submit only
 * (not for the stream) and copies its frames (out == NULL or cap == 0: returns their number without copying, so that the
 * caller can size its buffer); valid for the last two tickets, a negative error for older ones.  Per-submit
 * completion events (a few microseconds of command-stream bubble each) are recorded from the first sonde_batch_ticket call on;
 * a submit queued before that call is waited for through its stream. */
uint64_t sonde_batch_ticket(SondeBatch *b);
long sonde_batch_frames_of(SondeBatch *b, uint64_t ticket, SondeFrame *out, size_t cap);
/* Frames of the last submit that found no slot (more than the per-channel maximum, which is sized from max_samples and the
 * shortest frame of the type, so this stays 0 unless the sizing rule is broken); they are dropped, never written out of bounds. */
long sonde_batch_overflow(SondeBatch *b);
/* The decoded telemetry of the last submit as SondeData fragments (what the reference's per-channel X_decode loops
 * would have returned, decoder.hpp:61), in (channel, time) order, with the channel each belongs to.  The engine keeps
 * one stateful parser per channel.  Call repeatedly until it returns 0; fragments not fetched before the next
 * submit's poll are kept and delivered first.  Every submit since the previous poll that is still resident is parsed, in
 * order (frame slots exist twice: poll at least every second submit; an overwritten, unpolled submit is a negative error,
 * once).  Returns the number written (<= cap) or a negative error. */
long sonde_batch_poll(SondeBatch *b, SondeData *out, uint32_t *channel, size_t cap);
/* Average device time (ms) of the demod kernel (for RS41 channels it includes the sync search and the FEC epilogue) and of
 * the framer/FEC kernels behind it (0 if there are none) over the TIMED submits since the previous call of this function
 * (at most the last 128), from HIP events recorded on the submit stream around each launch.  Synchronises.
 * Every event record is a bubble of a few microseconds in the command stream, so by default only every 8th submit is
 * timed (the last of each group of eight, plus the first submit of the batch's life); sonde_batch_set_timing changes that
 * (1 = every submit, 0 = none) and restarts the count: the every_n-th submit after it is the next one timed. */
int  sonde_batch_kernel_ms(SondeBatch *b, float *demod_ms, float *framer_ms);
int  sonde_batch_set_timing(SondeBatch *b, int every_n);
/* launch units per submit (1: one plain launch on the caller's stream) and the completion mode the FLAGS ask for: 0 at every submit
 * (the default), 1 one submit late (SONDE_FLAG_LATE_JOIN), 2 never (SONDE_FLAG_PIPELINE); see SONDE_FLAG_PIPELINE above */
int  sonde_batch_launch_info(const SondeBatch *b, uint32_t *n_units, int32_t *join_mode);
/* Mixed batches (several demodulator classes, one kernel each on the library's own streams): the average device time (ms) of
 * each class's demod kernel alone over the timed submits since the last call; index 0: no decimation / 16 taps (M10 wide,
 * the AFSK 6 kS/s streams), 1: 2:1 / 16 (RS41, DFM, iMS-100, MRZ-N1 wide), 2: 4:1 / 8 (the same four, default), 3: 2:1 / 8
 * (M10); -1 = class not in the batch.  Returns the number of submits averaged (0 for a one-class batch). */
int  sonde_batch_class_ms(SondeBatch *b, float out[4]);

/* introspection for staged parity tests */
int      sonde_batch_read_bits(SondeBatch *b, uint32_t channel, uint64_t from, size_t count, uint8_t *out /* one bit per byte */);
/* parity-test introspection: the RS(255,231) corrector alone on n_pairs codeword pairs of [2][256] bytes (positions >= n zero),
 * corrected in place; status[2 i + c] = 0 clean, > 0 corrected byte errors, -1 uncorrectable (word left as received) */
int      sonde_batch_test_rs255(SondeBatch *b, uint8_t *cw_pairs, size_t n_pairs, int n, int32_t *status);
uint64_t sonde_batch_nbits(SondeBatch *b, uint32_t channel);
int      sonde_batch_read_state(SondeBatch *b, uint32_t channel, int64_t *t_next, int32_t *period, float *bias, float *amp,
                                float *afc_u /* the newest AFC state u of SPEC 3.0b (the carrier offset the channel is following: 2 atan u per
                                                decimated sample); 0 for real input.  (The parameter was called yprev, reserved, in rounds 1-3.) */);
int      sonde_get_taps(int type, float *out /* 32*32 floats, [phase][tap] */);
int      sonde_get_afsk_table(float *out /* 480*2 floats: the iMet tone demodulator's mixer table (cos, -sin) */);

/* frame -> SondeData fragments (what one X_decode call sequence yields for this frame).
 * Returns number of fragments written (<= cap). */
int  sonde_parse_frame(const SondeFrame *f, SondeData *out, int cap);

/* The same with per-channel memory: RS41 temperature/humidity need the calibration table that arrives 16 bytes per
 * frame (51 frames), DFM positions arrive over three frames.  One handle per channel, frames fed in order. */
typedef struct SondeParserHandle SondeParserHandle;
SondeParserHandle *sonde_parser_create(int sonde_type);
int  sonde_parser_feed(SondeParserHandle *p, const SondeFrame *f, SondeData *out, int cap);
void sonde_parser_destroy(SondeParserHandle *p);

/* RS41 sensor conversions behind fragment.temp / fragment.rh (decoder.hpp:87-88; bodies in the absent sondedump):
 * counts f between the reference counts f1 < f2, calibration words from the sonde's table. */
float sonde_rs41_temp(uint32_t f, uint32_t f1, uint32_t f2, float rf1, float rf2, const float co[3], const float cal[3]);
float sonde_rs41_rh(uint32_t f, uint32_t f1, uint32_t f2, float calh0, float temp);
/* DFM thermistor: measurement channel 0 against the reference channels 3 and 4 (already converted from 24-bit floats) */
float sonde_dfm_temp(float f, float f1, float f2);
/* RS41-SGP pressure (hPa) from the sensor counts, the sensor temperature (deg C) and the 25-entry coefficient matrix */
float sonde_rs41_pressure(uint32_t f, uint32_t f1, uint32_t f2, float tpress, const float *cfP /* [25] */);
/* ozone partial pressure (mPa) of an ECC ozonesonde (RS41 / iMet XDATA) from cell current (uA) and pump temperature (deg C) */
float sonde_ozone_mpa(float cell_ua, float tpump_c);
/* M10 / M20 / iMS-100 sensor conversions behind fragment.temp / .rh */
float sonde_m10_temp(unsigned scale, unsigned adc);
float sonde_m10_rh(uint32_t cap_sensor, uint32_t cap_ref, float temp);
float sonde_m20_temp(unsigned adc);
float sonde_ims100_temp(uint32_t f, float c0, float c1, float c2);

/* ------------------------------------------------------------------ wideband front-end (BASELINE config 4)
 * 10 MS/s complex IQ -> 512-bin polyphase channelizer (19531.25 Hz spacing, 20 kS/s per bin; a bin leaves the filter bank as
 * one PHASE sample per step) -> per-bin FM discriminator (wrapped phase difference) -> 12/5 rational resampler -> 48 kS/s ->
 * the decoder of the bin's sonde type: the reference's VFO -> dsp::demod::FM -> RationalResampler -> Decoder chain
 * (/root/reference/src/main.cpp:55-68) for every bin at once.  A bin carries the sondes whose channel is 10-20 kHz wide in the
 * reference (RS41, DFM, iMS-100, MRZ-N1; iMet-4 and SRS-C50 in the unfused mode); an M10 / M20 channel is 50 kHz wide
 * (/root/reference/src/main.hpp:48) and does not fit: sonde_chan_create fails for SONDE_M10 bins (use sonde_vfo_* for those).
 * One submit takes blocks_per_submit * 1 280 000 wideband samples (device pointer, complex64, 16-byte
 * aligned); the filter bank reads the block in place, so it must stay untouched until the submit's kernels have run
 * (stream order, as for sonde_batch_submit). */
typedef struct SondeChannelizer SondeChannelizer;
int         sonde_chan_create(const uint8_t *types /* 512 entries or NULL = RS41 */, uint32_t blocks_per_submit /* 1..8; 1..2 with AFSK bins or unfused */,
                              int device, SondeChannelizer **out);
/* The same for n_streams wideband streams per submit (one launch of each stage over all streams: grid.y = stream): types has
 * n_streams * 512 entries (stream-major) or is NULL; sonde_chan_submit then takes n_streams blocks laid out back to back,
 * [n_streams][n_samples] complex64; channel s * 512 + k of sonde_chan_batch() is bin k of stream s. */
int         sonde_chan_create_multi(const uint8_t *types, uint32_t blocks_per_submit, uint32_t n_streams, int device, SondeChannelizer **out);
/* Both stackings of every stream (DESIGN SPEC 3.5c): besides the 512 bins centred at k x 19531.25 Hz a second, ODD-STACKED bank with
 * bins centred at (k + 1/2) x 19531.25 Hz runs over the same samples, so that every carrier on the reference's 1 kHz VFO raster
 * (/root/reference/src/main.cpp:14,55-56) lies within 4.9 kHz of a bin centre (a bin passes about +-5 kHz; the slicer follows the
 * offset).  1024 decoder channels per stream: channel 1024 p + k = even bin k of stream p, 1024 p + 512 + k = odd bin k; types has
 * n_streams * 1024 entries in that order (or is NULL).  Costs about twice the single bank (the same filter-bank kernel runs twice
 * over the input, the decoder over twice the channels). */
int         sonde_chan_create_dual(const uint8_t *types, uint32_t blocks_per_submit, uint32_t n_streams, int device, SondeChannelizer **out);
uint32_t    sonde_chan_streams(const SondeChannelizer *c);       /* input streams per submit */
uint32_t    sonde_chan_channels(const SondeChannelizer *c);      /* decoder channels: 512 per stream, 1024 with both stackings */
/* By default (where every bin's sonde type allows it: no AFSK sonde) the per-bin discriminator and the 12/5 resampler run in
 * the decoder kernel's load path: a submit is two launches and the 48 kS/s rows never exist in HBM.  on = 0 keeps them as a
 * kernel of their own (then sonde_chan_read can return the rows: parity tests).  Call before the first submit; returns the mode
 * in force (1 fused, 0 not); on < 0 only asks. */
int         sonde_chan_set_fused(SondeChannelizer *c, int on);
/* What sonde_chan_submit's block holds: SONDE_INPUT_IQ (complex64; the default), SONDE_INPUT_IQ16 (int16 I, int16 Q interleaved) or
 * SONDE_INPUT_IQ8 (int8 pairs): the formats 10 MS/s receivers deliver -- a half / a quarter of the bytes over PCIe and HBM; converted exactly, no scaling, on the way into the filter
 * bank's window, so phases and frames are those of the float block holding the same integers).  Call before the first submit;
 * returns the kind in force (or -1). */
int         sonde_chan_set_input(SondeChannelizer *c, int input_kind);
/* Option (fused mode only, off by default): the filter bank runs on an internal
 * stream, the decoder on another, the bins are double-buffered, so the filter bank of submit k+1 may run beside the decoder
 * of submit k.  The caller's stream is made to wait (on the device) for the filter bank only -- the last reader of the
 * caller's block -- and the frames come with sonde_batch_sync / sonde_batch_frames_of on sonde_chan_batch() as always; the
 * decoder is NOT ordered into the caller's stream.  Measured: +1-2 % at 8 streams x 4-8 blocks per submit, a loss at one
 * stream.  Call before the first submit; returns the mode in force. */
int         sonde_chan_set_overlap(SondeChannelizer *c, int on);
void        sonde_chan_destroy(SondeChannelizer *c);
uint32_t    sonde_chan_samples_per_submit(const SondeChannelizer *c);
int         sonde_chan_submit(SondeChannelizer *c, const void *iq_dev, size_t n_samples, void *stream);
SondeBatch *sonde_chan_batch(SondeChannelizer *c);     /* frames of the 512 bins: sonde_batch_sync / _frames on this */
/* parity-test introspection: the last submit's per-bin phases ([bins][n_steps] floats, quadrants) and, unfused mode only,
 * the 48 kS/s rows ([bins][n_steps * 12 / 5]); either pointer may be NULL (out48 != NULL in fused mode is an error) */
int         sonde_chan_read(SondeChannelizer *c, float *bins, float *out48);
/* average device time (ms) of the filter-bank kernel, the discriminator + resampler kernel and the decoder kernels over the
 * timed submits (every 8th) since the previous call; synchronises */
int         sonde_chan_kernel_ms(SondeChannelizer *c, float *pfb_ms, float *disc_resamp_ms, float *demod_ms, float *framer_ms);
int         sonde_chan_tables(float *h, float *tw, float *g);

/* ------------------------------------------------------------------ VFO front-end (SURVEY section 8 rows a1 + a2)
 * The stage between the SDR++ VFO and the decoder in /root/reference/src/main.cpp:55-60, for a batch of channels of one
 * rate: complex IQ at the sonde type's VFO bandwidth (supportedTypes[], main.hpp:44-52: 10 000 RS41, 15 000 DFM, 20 000
 * iMS-100 / iMet-4 / SRS-C50 / MRZ-N1, 50 000 M10/M20; 40 000 = a channelizer bin) -> dsp::demod::FM -> dsp::RationalResampler
 * (24/5, 16/5, 12/5, 24/25, 6/5) -> 48 kS/s FM audio rows, which are the rows sonde_batch_submit() takes from a batch
 * created with SONDE_INPUT_REAL.  n_in: complex samples per channel, a multiple of the ratio's denominator (5; 25 at
 * 50 kS/s); a row of n_in samples yields n_in * up / down output samples (sonde_vfo_out_samples). */
typedef struct SondeVfo SondeVfo;
int    sonde_vfo_create(uint32_t n_channels, int rate_in, size_t max_in, int device, SondeVfo **out);
void   sonde_vfo_destroy(SondeVfo *v);
int    sonde_vfo_ratio(int rate_in, int *up, int *down);
size_t sonde_vfo_out_samples(const SondeVfo *v, size_t n_in);
int    sonde_vfo_process(SondeVfo *v, const void *iq_dev, size_t n_in, size_t channel_stride /* complex samples */,
                         float *out48_dev, size_t out_stride /* floats */, void *stream);
/* host rows in; the 48 kS/s rows stay on the device (internal buffer, valid until the next call) */
int    sonde_vfo_process_host(SondeVfo *v, const void *iq_host, size_t n_in, size_t channel_stride,
                              const float **out48_dev, size_t *out_stride);
int    sonde_vfo_taps(int rate_in, float *g /* up * 16 */);      /* parity-test introspection */

/* post-FEC derived quantities, as /root/reference/src/decode/decoder.hpp:132-174 computes them */
float sonde_dewpt(float temp, float rh);
float sonde_altitude_to_pressure(float alt);

/* sinks (C faces of include/sonde_sinks.hpp): GPX 1.1 track + PTU CSV, byte-identical to the files
 * /root/reference/src/gpx.cpp and /root/reference/src/ptu.cpp write */
void *sonde_gpx_open(const char *path);
void  sonde_gpx_close(void *g);
void  sonde_gpx_start_track(void *g, const char *name);
void  sonde_gpx_stop_track(void *g);
void  sonde_gpx_add_point(void *g, long t, float lat, float lon, float alt, float spd, float hdg);
void *sonde_ptu_open(const char *path);
void  sonde_ptu_close(void *p);
void  sonde_ptu_add_point(void *p, long t, float temp, float rh, float dewpt, float pressure, float lat, float lon,
                          float alt, float spd, float hdg, float climb, const char *aux);

/* Measurement aid (bench.py): best-of-`reps` bandwidth, in GB/s, of a read-only streaming kernel over a device
 * buffer -- the HBM read rate this GPU can actually deliver, quoted beside the spec peak. */
int sonde_hbm_read_probe(const void *d_buf, size_t bytes, int reps, float *gbs_out);

const char *sonde_last_error(void);
const char *sonde_version(void);

#ifdef __cplusplus
}
#endif
#endif
