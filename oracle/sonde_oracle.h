/*
 * sonde_oracle.h -- CPU oracle for the radiosonde demod+FEC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sdrpp_radiosonde_amd/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED.  The reference's arithmetic for this path lives in the
 * un-vendored git submodule dbdexter-dev/sondedump (branch master, SHA
 * unrecoverable; /root/reference/.gitmodules:1-4, directory empty) and in SDR++
 * core (dsp::demod::FM, /root/reference/src/main.hpp:33).  Neither is present,
 * the reference ships no tests or golden vectors, so this
 * oracle restates the *published* algorithm structure of that path
 * (FM quadrature discriminator -> GFSK low-pass/matched FIR -> Gardner timing
 * recovery -> hard slicer -> sync-word correlator -> XOR de-whitening ->
 * RS(255,231) -> CRC16 per subframe) from the call sites
 * (/root/reference/src/decode/decoder.hpp:22,53-119, src/main.cpp:54-72) and
 * public protocol facts.  It is pinned only by the
 * known-answer tests in tests/ (RS41 header/mask KAT, CRC16 "123456789",
 * RS encode->corrupt->decode round trips, dewpt/altitude KATs from SURVEY.md).
 *
 * The arithmetic contract ("SPEC") that both this oracle and the HIP kernels
 * implement is written down in DESIGN.md section 3.  Every float operation is an
 * explicit IEEE-754 binary32 op (compile with -ffp-contract=off; fused
 * multiply-adds are spelled fmaf()), every cross-symbol reduction is done on
 * integers, so a sequential C loop and a 256-lane HIP workgroup produce the
 * same bits.
 */
#ifndef SONDE_ORACLE_H
#define SONDE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SPEC constants ---- */
#define OR_FS          48000      /* decoder input rate, src/main.cpp:16 OUT_SAMPLE_RATE */
#define OR_TILE        2048       /* samples per tile */
#define OR_RING        4096       /* discriminator ring (floats) */
#define OR_NTAPS       32         /* row length of the polyphase table */
#define OR_NT(m)       ((m)->nt)   /* taps in use per branch: 3.2 symbols = 8 at ~2.5 samples per symbol, 16 at ~5 (the 6 kS/s AFSK streams: 16) */
#define OR_NPHASE      32         /* polyphase branches (1/32 sample resolution) */
#define OR_ROUND_MAX   256        /* max symbols per timing-loop round */
#define OR_LOOKAHEAD_MARGIN 4     /* samples of slack behind the newest sample */
#define OR_FRAME_MAX   528        /* bytes reserved per frame record */

/* sonde types: order of supportedTypes[], /root/reference/src/main.hpp:44-52 */
enum { OR_RS41 = 0, OR_DFM09 = 1, OR_IMS100 = 2, OR_M10 = 3, OR_IMET4 = 4, OR_C50 = 5, OR_MRZN1 = 6, OR_NTYPES = 7 };

typedef struct {
	int    type;
	double baud;        /* on-air symbol (chip) rate */
	int    period0;     /* Q16 (internal-rate) samples per symbol = rint(65536*(FS/decim)/baud) */
	float  cutoff;      /* low-pass cutoff, in units of baud */
	int    decim;       /* IQ is decimated decim:1 (boxcar) before the discriminator: 4 RS41 / DFM / iMS-100 / MRZ-N1 (12 kS/s), 2 M10 (24 kS/s) */
	int    pre;         /* 8: AFSK sonde, the tone demodulator in front delivers FS/8 samples (SPEC 3.6); else 1 */
	int    nt;          /* taps in use per polyphase row (filled in by or_modem()): 8 below 3.5 samples per symbol, else 16; AFSK: 16 */
} OrModem;

typedef struct {
	uint32_t channel;
	uint32_t type;
	int32_t  len;            /* frame length in bytes */
	int32_t  nerr[2];        /* RS41: corrected byte errors per codeword, -1 = uncorrectable. Others: per-type error counts */
	uint32_t flags;          /* bit0: inverted polarity */
	uint64_t bitpos;         /* absolute bit index of the first sync bit */
	uint8_t  data[OR_FRAME_MAX];
} OrFrame;

/* ---- stage 1: FM discriminator (SDR++ dsp::demod::FM<float>, src/main.cpp:57) ---- */
float or_recip(float x);
float or_atan2(float y, float x);
/* d[n] = arg(x[n]*conj(x[n-1])) * 2/pi ; last[2] = previous (I,Q), carried state ((0,0) at init) */
void  or_discriminate(const float *iq, size_t n, float *d, float *last);

/* ---- stage 2: GFSK demod (sondedump gfsk.c equivalent; SPEC) ---- */
void  or_make_taps(const OrModem *m, float taps[OR_NPHASE][OR_NTAPS]);
const OrModem *or_modem(int type);
void  or_modem_set_decim(int type, int decim);   /* test hook: the product's SONDE_FLAG_WIDE (one decimation step less) */

typedef struct OrDemod OrDemod;
OrDemod *or_demod_new(int type);
void     or_demod_free(OrDemod *d);
/* n must be a multiple of OR_TILE.  is_iq: src is interleaved I,Q (2n floats); else n real discriminator samples */
void     or_demod_feed(OrDemod *d, const float *src, size_t n, int is_iq);
uint64_t or_demod_nbits(const OrDemod *d);
/* copy bits [from, from+count) one per byte */
void     or_demod_getbits(const OrDemod *d, uint64_t from, size_t count, uint8_t *out);
/* debug taps for staged parity tests */
void     or_demod_state(const OrDemod *d, int64_t *t_next, int32_t *period, float *bias, float *amp, float *yprev);

/* ---- stage 1b: AFSK tone demodulator (iMet-4; SPEC 3.6) ---- */
#define OR_AF_DEC  8          /* 48 kS/s -> 6 kS/s */
#define OR_AF_PER  480        /* period of the 1700 Hz mixer table at 48 kS/s (17 cycles) */
#define OR_AF_WIN  5          /* boxcar: 5 blocks of 8 samples = one 1200 Bd symbol */
void     or_afsk_table(float *w /* [480][2]: cos, -sin */);

/* ---- stage 3: framer + FEC ---- */
void     or_gf256_init(void);
uint8_t  or_gf256_mul(uint8_t a, uint8_t b);
/* RS(255,231), poly 0x11d, roots alpha^0..alpha^23.  cw[0..23]=parity, cw[24..]=message, n = used length (<=255).
 * returns number of corrected bytes or -1. */
int      or_rs255_decode(uint8_t *cw, int n);
void     or_rs255_encode(uint8_t *cw, int n);   /* fills cw[0..23] from cw[24..n) */
uint16_t or_crc16_ccitt(const uint8_t *p, size_t n);
uint16_t or_m10_checksum(const uint8_t *p, size_t n);
uint16_t or_imet_crc(const uint8_t *p, size_t n);        /* CRC16-CCITT, init 0x1D0F */
uint16_t or_crc16_modbus(const uint8_t *p, size_t n);   /* reflected 0xA001, init 0xFFFF (MRZ-N1) */
uint32_t or_bch_parity(uint64_t data34);                 /* BCH(63,51) shortened to (46,34) */
uint64_t or_bch_decode(uint64_t blk46, int *st);         /* st: errors corrected (0..2) or -1 */

typedef struct OrFramer OrFramer;
OrFramer *or_framer_new(int type, uint32_t channel);
void      or_framer_free(OrFramer *f);
/* consume bits from the demod up to its current write position; returns number of new frames appended */
int       or_framer_run(OrFramer *f, const OrDemod *d);
size_t    or_framer_nframes(const OrFramer *f);
const OrFrame *or_framer_frame(const OrFramer *f, size_t i);

/* ---- whole channel convenience (= what one GPU workgroup does) ---- */
typedef struct OrChannel OrChannel;
OrChannel *or_channel_new(int type, uint32_t channel);
void       or_channel_free(OrChannel *c);
void       or_channel_feed(OrChannel *c, const float *src, size_t n, int is_iq);
size_t     or_channel_nframes(const OrChannel *c);
const OrFrame *or_channel_frame(const OrChannel *c, size_t i);
OrDemod   *or_channel_demod(OrChannel *c);

/* batch helper for timing: channels [0,nch) each n samples, channel-major; returns total frames.
 * nthreads<=1: serial.  Used by bench.py cpu_baseline. */
size_t     or_batch_run(int type, const float *iq, size_t nch, size_t n, int nthreads, OrFrame *out, size_t cap);

/* ---- yardstick: a conventional per-sample GFSK receiver (or_yardstick.c) in front of the same framers / FEC.  Shares no
 * demodulator arithmetic with or_dsp.c (libm atan2f, AGC, per-symbol Gardner PI loop); GFSK sonde types only.
 * cutoff_rel: symbol-filter cutoff in symbol rates (<= 0: 1.0); loop_bw: loop noise bandwidth in symbol rates (<= 0: 0.01). */
typedef struct OrYard OrYard;
OrYard *or_yard_new(int type, uint32_t channel, float cutoff_rel, float loop_bw);
void    or_yard_free(OrYard *y);
void    or_yard_feed(OrYard *y, const float *iq, size_t n);          /* complex IQ at 48 kS/s, any n */
size_t  or_yard_nframes(const OrYard *y);
const OrFrame *or_yard_frame(const OrYard *y, size_t i);
uint64_t or_yard_nbits(const OrYard *y);
void    or_yard_getbits(const OrYard *y, uint64_t from, size_t count, uint8_t *out);
size_t  or_yard_batch_run(int type, const float *iq, size_t nch, size_t n, int nthreads, float cutoff_rel, float loop_bw, OrFrame *out, size_t cap);

/* ---- wideband front-end (config 4): 512-bin PFB channelizer + discriminator + 6/5 resampler ---- */
#define OR_CH_FS   10000000.0   /* wideband sample rate */
#define OR_CH_M    512          /* bins, spacing 19531.25 Hz */
#define OR_CH_D    500          /* decimation: 20 kS/s per bin (round 4; rounds 2-3: 250) */
#define OR_CH_T    16           /* prototype taps per bin */
#define OR_CH_L    (OR_CH_M * OR_CH_T)
#define OR_RS_L    12           /* resampler: up 12 */
#define OR_RS_M    5            /*            down 5 : 20 kS/s -> 48 kS/s */
#define OR_RS_T    16           /* taps per resampler phase */
#define OR_RS_KT_LD 20          /* row stride of the composite (resampler + boxcar) taps, SPEC 3.5b: 17 in use */
typedef struct OrChan OrChan;
void    or_chan_proto(float *h);
void    or_chan_twiddles(float *tw);
void    or_chan_resamp_taps(float *g);
void    or_resamp_taps(int up, double fs_up_hz, double cutoff_hz, float *g);
/* VFO front-end: IQ at the sonde type's VFO rate -> discriminator -> rational resampler -> 48 kS/s (main.cpp:55-60) */
typedef struct OrVfo OrVfo;
int     or_vfo_ratio(int rate_in, int *up, int *down, int *cutoff_hz);
OrVfo  *or_vfo_new(int rate_in);
void    or_vfo_free(OrVfo *v);
size_t  or_vfo_process(OrVfo *v, const float *iq, size_t n_in, float *out48);
void    or_fft512(float *re, float *im, const float *tw);
OrChan *or_chan_new(void);
OrChan *or_chan_new_odd(void);        /* the odd-stacked bank: bin k centred at (k + 1/2) bin spacings (SPEC 3.5c) */
void    or_chan_twist(float *w);
float   or_chan_ramp(size_t m);
void    or_chan_free(OrChan *c);
void    or_chan_block(OrChan *c, const float *iq, size_t n_steps, float *bins, float *out48);
void    or_chan_block2(OrChan *c, const float *iq, size_t n_steps, float *bins, float *out48, const uint8_t *decs, float *outdec);
int     or_chan_composite_kt(int dec);
void    or_chan_composite_taps(const float *g, int dec, float *G /* 3 * OR_RS_KT_LD */);

/* ---- post-FEC derived values, restating /root/reference/src/decode/decoder.hpp:132-174 ---- */
float or_dewpt(float temp, float rh);
float or_altitude_to_pressure(float alt);
float or_rs41_temp(uint32_t f, uint32_t f1, uint32_t f2, float rf1, float rf2, const float *co, const float *cal);
float or_rs41_rh(uint32_t f, uint32_t f1, uint32_t f2, float calh0, float T);
float or_dfm_temp(float f, float f1, float f2);
/* double-precision, differently factored restatements (compared within a tolerance, tests/test_parsers_cpu.py) */
double or_rs41_pressure_d(uint32_t f, uint32_t f1, uint32_t f2, double tpress, const float *cfP);
double or_ozone_mpa_d(double cell_ua, double tpump_c);
double or_m10_temp_d(unsigned scale, unsigned adc);
double or_m10_rh_d(uint32_t cap_sensor, uint32_t cap_ref, double T);
double or_m20_temp_d(unsigned adc);
double or_ims100_temp_d(uint32_t f, double c0, double c1, double c2);

#ifdef __cplusplus
}
#endif
#endif
