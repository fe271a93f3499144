/*
 * or_dsp.c -- oracle stages 1+2: FM discriminator and GFSK demodulator.
 * TEST INFRASTRUCTURE ONLY (see sonde_oracle.h).  PARITY UNPINNED.
 *
 * Stage 1 restates SDR++'s quadrature FM demodulator as wired at
 * /root/reference/src/main.cpp:57  fmDemod.init(vfo->output, bw, bw/2.0f, false)
 * (deviation = bandwidth/2 = samplerate/4  =>  gain = samplerate/(2*pi*dev) = 2/pi,
 *  SURVEY.md section 8a-1), with the libm atan2f replaced by an explicit
 * polynomial so that CPU and GPU agree bit for bit.
 *
 * Stage 2 stands where sondedump's gfsk_demod() sits behind
 * X_decode(T*, SondeData*, const float*, size_t)  (/root/reference/src/decode/decoder.hpp:22,61):
 * polyphase low-pass FIR evaluated at the timing loop's sampling instants,
 * Gardner timing-error detector, PI loop filter, hard slicer.  The loop filter is
 * updated once per "round" (all symbols that became decodable after one 2048-sample
 * tile, at most 256) instead of once per symbol; DESIGN.md section 3 gives the contract.
 *
 * Build with -ffp-contract=off: every float op below is one IEEE binary32 op.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "sonde_oracle.h"

#define OR_PI_D     3.14159265358979323846

/* atan2 in quadrants (SPEC 3.1, round 3).  The consumer of the discriminator is a low-pass FIR and a hard slicer, so the
 * arctangent needs about 1e-3 rad, not 1e-5: one division-free form for the whole first quadrant,
 *     r = (|x| - |y|) / (|x| + |y|)  in [-1, 1],     angle = pi/4 - atan(r),
 * (no |y| > |x| swap), a three-term odd minimax polynomial for (2/pi) atan(r) (max error 4.5e-4 quadrant = 7e-4 rad,
 * constrained to 0.5 at r = 1 so that the octants join), and ONE Newton step on the reciprocal (relative error < 0.26 %,
 * always from below, so |r| <= 1).  15 VALU operations per sample on the GPU instead of 24.
 * Coefficients already carry the discriminator gain 2/pi (angles in quadrants: pi/2 -> 1, pi -> 2). */
#define AT_C1  0.6332877278327942f     /* 0x3F221F25 */
#define AT_C3 -0.18171308934688568f    /* 0xBE3A12FF */
#define AT_C5  0.04842534288764f       /* 0x3D4659A7 */

static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline uint32_t f2u(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
static inline float u2f(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }

/* Reciprocal by Newton-Raphson from an integer-subtract seed: 1 integer op + 6 fmaf, the same
 * sequence on CPU and GPU (an IEEE divide costs the GPU ~13 issue slots).  Relative error ~1e-7 for normal x > 0.
 * Used by the loop filter (once per round). */
float or_recip(float x)
{
	float r = u2f(0x7EF311C7u - f2u(x));
	float e = fmaf(-x, r, 1.0f);
	r = fmaf(r, e, r);
	e = fmaf(-x, r, 1.0f);
	r = fmaf(r, e, r);
	e = fmaf(-x, r, 1.0f);
	r = fmaf(r, e, r);
	return r;
}

/* atan2q(y, x) = atan2(y, x) * 2/pi, in [-2, 2], |error| <= 2.5e-3 rad.  NaN / Inf inputs are outside the contract.
 * atan2q(0, 0) = 0 (round 4): the divisor s = |x| + |y| is floored at AT_FLOOR and the numerator is taken from
 * it, d = s - 2|y| (= |x| - |y| up to one rounding of s), so that at (0, 0) r = AT_FLOOR * rc(AT_FLOOR); AT_FLOOR is the
 * float near 1e-30 for which the seed + one Newton step return its reciprocal so that this product is EXACTLY 1, where the
 * polynomial is exactly 1/2: a squelched or zero-padded stream reads 0 like any FM demodulator's, not +-1/2. */
#define AT_FLOOR 6.9721523e-31f     /* 0x0D624260 */
float or_atan2(float y, float x)
{
	const float ax = u2f(f2u(x) & 0x7FFFFFFFu), ay = u2f(f2u(y) & 0x7FFFFFFFu);
	const float s = fmaxf(ax + ay, AT_FLOOR);
	const float d = fmaf(-2.0f, ay, s);
	float rc = u2f(0x7EF311C7u - f2u(s));            /* seed: relative error < 5.1 % */
	const float e = fmaf(-s, rc, 1.0f);
	rc = fmaf(rc, e, rc);                            /* one Newton step: < 0.26 % */
	const float r = d * rc;
	const float t = r * r;
	float p = fmaf(t, AT_C5, AT_C3);
	p = fmaf(t, p, AT_C1);
	const float q = fmaf(-p, r, 0.5f);               /* first-quadrant angle of (|x|, |y|), in [0, 1] */
	const float s2 = (f2u(x) >> 31) ? 2.0f : 0.0f;
	const float q2 = s2 - q;                         /* x < 0: 2 - q */
	return u2f((f2u(q2) & 0x7FFFFFFFu) | (f2u(y) & 0x80000000u));
}

/* Quadrature FM discriminator: d[n] = arg(x[n] * conj(x[n-1])) * 2/pi.  In real arithmetic this is
 * the wrapped phase difference SDR++'s dsp::demod::FM computes (main.cpp:57, gain 2/pi); the
 * product form needs neither a cross-lane phase exchange nor a wrap on the GPU.
 * last[2] = previous sample (I, Q), carried state, (0,0) at init. */
void or_discriminate(const float *iq, size_t n, float *d, float *last)
{
	float x0 = last[0], y0 = last[1];
	for (size_t i = 0; i < n; i++) {
		const float x1 = iq[2 * i], y1 = iq[2 * i + 1];
		const float cross = fmaf(-x1, y0, y1 * x0);
		const float dot = fmaf(y1, y0, x1 * x0);
		d[i] = or_atan2(cross, dot);
		x0 = x1;
		y0 = y1;
	}
	last[0] = x0;
	last[1] = y0;
}

/* SPEC 3.0e: (4 / pi) atan(u), |u| <= OR_AFC_MAX = 0.8: odd polynomial, max error 2.3e-5 quadrant */
static float afc_rot(float u)
{
	const float t = u * u;
	float p = fmaf(t, -0.073257752f, 0.21412420f);
	p = fmaf(t, p, -0.41814741f);
	p = fmaf(t, p, 1.2729679f);
	return u * p;
}

/* The same with the product x[n] conj(x[n-1]) turned back by the AFC phasor (c, s) = (1 - u^2, 2u) before the arctangent
 * (SPEC 3.0b): d[n] = arg(x[n] conj(x[n-1]) (c - j s)).  The phasor's length (1 + u^2) does not matter to an arctangent. */
static void discriminate_rot(const float *iq, size_t n, float *d, float *last, float u)
{
	const float c = fmaf(-u, u, 1.0f), sn = u + u;
	float x0 = last[0], y0 = last[1];
	for (size_t i = 0; i < n; i++) {
		const float x1 = iq[2 * i], y1 = iq[2 * i + 1];
		const float cross = fmaf(-x1, y0, y1 * x0);
		const float dot = fmaf(y1, y0, x1 * x0);
		const float cr = fmaf(-dot, sn, cross * c);
		const float dr = fmaf(cross, sn, dot * c);
		d[i] = or_atan2(cr, dr);
		x0 = x1;
		y0 = y1;
	}
	last[0] = x0;
	last[1] = y0;
}

/* ---- modem table.  Baud rates: SURVEY.md Appendix B [RECALL]; VFO bandwidths in
 * /root/reference/src/main.hpp:44-52 bound them from above. ---- */
static OrModem g_modems[OR_NTYPES] = {
	{ OR_RS41,   4800.0, 0, 0.65f, 4, 1, 0 },  /* RS41: 4800 Bd GFSK, NRZ; 12 kS/s internally (the reference's VFO is 10 kHz wide, main.hpp:45) */
	{ OR_DFM09,  5000.0, 0, 0.65f, 4, 1, 0 },  /* DFM: 2500 bit/s Manchester => 5000 chips/s; 12 kS/s (2.4 samples per chip) */
	{ OR_IMS100, 4800.0, 0, 0.65f, 4, 1, 0 },  /* iMS-100/RS-11G: 2400 bit/s biphase => 4800 chips/s; 12 kS/s */
	{ OR_M10,    9600.0, 0, 0.65f, 2, 1, 0 },  /* M10/M20: 9600 chips/s Manchester; 24 kS/s (2.5 samples per chip) */
	{ OR_IMET4,  1200.0, 0, 0.65f, 1, 8, 0 },  /* iMet-1/4: Bell-202 AFSK 1200 Bd; tone demodulator in front, 6 kS/s behind it */
	{ OR_C50,    2400.0, 0, 0.65f, 1, 8, 0 },  /* SRS-C50: AFSK 2400 Bd (2900 / 4700 Hz); tone demodulator in front, 6 kS/s behind it */
	{ OR_MRZN1,  4800.0, 0, 0.65f, 4, 1, 0 },  /* MRZ-N1: 2400 bit/s Manchester => 4800 chips/s; 12 kS/s */
};
static OrModem g_modem_rt[OR_NTYPES];

/* test hook for configuration flags of the product (SONDE_FLAG_RS41_WIDE: RS41 at 2:1): affects demodulators created afterwards */
void or_modem_set_decim(int type, int decim)
{
	if (type < 0 || type >= OR_NTYPES) return;
	g_modems[type].decim = decim;
	g_modem_rt[type].period0 = 0;
}

const OrModem *or_modem(int type)
{
	if (type < 0 || type >= OR_NTYPES) return NULL;
	if (g_modem_rt[type].period0 == 0) {
		g_modem_rt[type] = g_modems[type];
		g_modem_rt[type].period0 = (int)llrint(65536.0 * ((double)OR_FS / (g_modems[type].decim * g_modems[type].pre)) / g_modems[type].baud);
		/* 3.2 symbols of taps: 8 at ~2.5 samples per symbol, 16 at ~5; the AFSK streams keep 16-tap rows (SPEC 3.6) */
		g_modem_rt[type].nt = (g_modems[type].pre == 1 && 2 * g_modem_rt[type].period0 < 7 * 65536) ? 8 : 16;
	}
	return &g_modem_rt[type];
}

/* Blackman-windowed sinc, cutoff = m->cutoff * baud, one row per polyphase branch,
 * each row normalised to unit DC gain.  H[p][j] = f(j - N/2 + p/P). */
void or_make_taps(const OrModem *m, float taps[OR_NPHASE][OR_NTAPS])
{
	const int nt = OR_NT(m);                 /* taps in use: 3.2 symbols */
	memset(taps, 0, sizeof(float) * OR_NPHASE * OR_NTAPS);
	const double fc = (double)m->cutoff * m->baud / ((double)OR_FS / (m->decim * m->pre)); /* cycles per internal sample */
	for (int p = 0; p < OR_NPHASE; p++) {
		double h[OR_NTAPS], sum = 0.0;
		for (int j = 0; j < nt; j++) {
			const double t = (double)j - (double)(nt / 2) + (double)p / (double)OR_NPHASE;
			const double x = (t + (double)(nt / 2)) / (double)nt;
			const double w = 0.42 - 0.5 * cos(2.0 * OR_PI_D * x) + 0.08 * cos(4.0 * OR_PI_D * x);
			const double s = (t == 0.0) ? 2.0 * fc : sin(2.0 * OR_PI_D * fc * t) / (OR_PI_D * t);
			h[j] = s * w;
			sum += h[j];
		}
		for (int j = 0; j < nt; j++) taps[p][j] = (float)(h[j] / sum);
	}
}

/* AFC (SPEC 3.0b): state u, rotation angle 2 atan(u) per internal sample; gain per tile on the slicer bias (quadrants), leak
 * per tile, range +-0.8 (+-77 degrees per sample: +-2.6 kHz at 12 kS/s, +-5.2 kHz at 24 kS/s) */
#ifndef OR_AFC_GAIN
#define OR_AFC_GAIN 0.19634954f      /* 0.25 * pi / 4: a quarter of the measured offset per tile (0.125: slower to acquire, 0.4: the lag of 3-4 tiles rings) */
#endif
#define OR_AFC_LEAK 0.0078125f       /* 1/128 per tile: the state returns to 0 without a signal (time constant 5.5 s) */
#define OR_AFC_MAX  0.8f

struct OrDemod {
	const OrModem *m;
	float taps[OR_NPHASE][OR_NTAPS];
	float ring[OR_RING];
	int64_t n0;          /* samples consumed so far */
	float iq_last[2];
	/* AFSK front-end state (SPEC 3.6) */
	float af_b[OR_AF_WIN - 1][2];   /* the four block sums before the current one, oldest first */
	float af_z[2];                  /* previous boxcar output */
	uint64_t af_n;                  /* input samples consumed (mixer phase = af_n mod 480) */
	int64_t t_next;      /* Q16 absolute on-time instant of the next symbol */
	int32_t period;      /* Q16 samples per symbol */
	float bias, amp;
	int32_t nstat;
	float afc[3];        /* SPEC 3.0b: the AFC states u that the discriminator of the next three tiles will use, oldest first */
	uint8_t *bits;
	uint64_t nbits, cap;
};

OrDemod *or_demod_new(int type)
{
	OrDemod *d = calloc(1, sizeof(*d));
	d->m = or_modem(type);
	or_make_taps(d->m, d->taps);
	d->period = d->m->period0;
	d->t_next = ((int64_t)OR_NTAPS << 16) + d->period;
	d->amp = 0.25f;
	return d;
}

void or_demod_free(OrDemod *d) { if (d) { free(d->bits); free(d); } }

static inline float interp(const OrDemod *d, int64_t pos)
{
	const int64_t n = pos >> 16;
	const int p = (int)((pos >> 11) & (OR_NPHASE - 1));
	/* even and odd taps accumulate separately (one v_pk_fma_f32 per tap pair on the GPU) */
	const int nt = OR_NT(d->m);
	float acc_e = 0.0f, acc_o = 0.0f;
	for (int j = 0; j < nt; j += 2) {
		acc_e = fmaf(d->taps[p][j], d->ring[(n + nt / 2 - j) & (OR_RING - 1)], acc_e);
		acc_o = fmaf(d->taps[p][j + 1], d->ring[(n + nt / 2 - j - 1) & (OR_RING - 1)], acc_o);
	}
	return acc_e + acc_o;
}



static void push_bit(OrDemod *d, int b)
{
	if (d->nbits == d->cap) {
		d->cap = d->cap ? d->cap * 2 : 8192;
		d->bits = realloc(d->bits, d->cap);
	}
	d->bits[d->nbits++] = (uint8_t)b;
}

/* One tile's worth of symbols.  The number of symbols is fixed when the tile arrives
 * (K_total, from the timing state at that moment) and is split into rounds of at most 256 -- 512 for the
 * streams whose tile holds more than 256 symbols (M10, 410 chips per tile, and the 6 kS/s AFSK streams: up to 822), so
 * that M10 and iMet update the loop once per tile like every other sonde (every 42.7 ms of signal);
 * the loop filter is updated after every round.  OR_LOOKAHEAD_MARGIN samples of slack keep the
 * FIR support inside the data when a mid-tile correction moves the instants later. */
static void run_rounds(OrDemod *d)
{
	const int64_t limit = (((d->n0 - 1 - OR_NT(d->m) / 2 - OR_LOOKAHEAD_MARGIN) << 16) | 0xFFFF);
	float y[2 * OR_ROUND_MAX], m[2 * OR_ROUND_MAX];
	/* rounds of <= 256 symbols; 512 where a tile holds more than 256 (M10: 410 chips per tile; the 6 kS/s AFSK streams) */
	const int rmax = ((OR_TILE / d->m->decim) / (OR_NT(d->m) == 8 ? 2 : 4) > OR_ROUND_MAX) ? 2 * OR_ROUND_MAX : OR_ROUND_MAX;
	int64_t K_total = (d->t_next <= limit) ? (limit - d->t_next) / d->period + 1 : 0;

	while (K_total > 0) {
		const int K = K_total > rmax ? rmax : (int)K_total;
		int32_t E = 0, S1 = 0, S0 = 0, C1 = 0, SY = 0, SM = 0;
		K_total -= K;

		for (int k = 0; k < K; k++) {
			const int64_t t = d->t_next + (int64_t)k * d->period;
			y[k] = interp(d, t);
			m[k] = k < OR_ROUND_MAX ? interp(d, t - (d->period >> 1)) : 0.0f;
		}
		for (int k = 0; k < K; k++) {
			/* Gardner term of symbol k uses the previous symbol of the same round; the first symbol of
			 * every 64-symbol group contributes nothing (on the GPU a group is one wavefront, and this
			 * keeps the detector free of cross-wave traffic: 1.6 % fewer terms in a 200-term average) */
			/* rounds of more than 256 symbols (two symbols per lane on the GPU): only the first 256 feed the detector --
			 * 256 terms per 42.7 ms are plenty for a clock that drifts by ppm, and the second symbol of a lane then needs no
			 * mid-symbol FIR at all (round 3: a quarter of the M10 class's FIR work) */
			if ((k & 63) && k < OR_ROUND_MAX) {
				if (d->m->pre == 8) {       /* SPEC 3.6b: how open the eye is at the on-time and at the mid-symbol instants */
					SY += (int32_t)lrintf(clampf(fabsf(y[k] - d->bias), 0.0f, 8.0f) * 4096.0f);
					SM += (int32_t)lrintf(clampf(fabsf(m[k] - d->bias), 0.0f, 8.0f) * 4096.0f);
				}
				const float a = y[k - 1] - y[k];
				const float b = m[k] - d->bias;
				float e = a * b;
				e = clampf(e * 1024.0f, -1.0e6f, 1.0e6f);
				E += (int32_t)lrintf(e);
			}
			const int bit = y[k] > d->bias;
			const int32_t Y = (int32_t)lrintf(clampf(y[k], -8.0f, 8.0f) * 4096.0f);
			if (bit) { S1 += Y; C1++; } else { S0 += Y; }
			push_bit(d, bit);
		}
		const int32_t C0 = K - C1;
		if (C1 > 0 && C0 > 0) {
			const float hi = ((float)S1 * or_recip((float)C1)) * (1.0f / 4096.0f);
			const float lo = ((float)S0 * or_recip((float)C0)) * (1.0f / 4096.0f);
			const float c = 0.5f * (hi + lo);
			float a = 0.5f * (hi - lo);
			if (d->nstat == 0) {
				d->bias = c;
				d->amp = a;
			} else {
				d->bias = d->bias + 0.5f * (c - d->bias);
				d->amp = d->amp + 0.5f * (a - d->amp);
			}
			if (!(d->amp >= 1.0e-3f)) d->amp = 1.0e-3f;
			d->nstat = 1;
		} else {
			/* one-sided round (a carrier offset larger than the deviation puts every sample on one side of the
			 * threshold): no level estimate; move the threshold to the mean so that the next round sees both levels */
			d->bias = ((float)(S1 + S0) * or_recip((float)K)) * (1.0f / 4096.0f);
		}
		/* SPEC 3.2b (round 5), acquisition: during the first three tiles of a stream the threshold is the MEAN of the round.  With the
		 * carrier 2 kHz off, the first rounds slice most symbols to one side, the level estimate above is built from wrong decisions
		 * and creeps towards the offset over several tiles (0.20, 0.25, 0.36 ... of 0.67 quadrants): every frame the SPEC missed and a
		 * conventional receiver decoded at +-2 kHz lay in the first 700 chips (profiles/r5_notes.md section 7).  The sondes' line codes
		 * are balanced (Manchester, biphase, whitened NRZ): the mean of a round IS the offset.  Amplitude and `nstat` as above; the
		 * AFSK streams (threshold 0 by construction) keep the level estimate. */
		if (d->m->pre == 1 && d->n0 <= 3 * (int64_t)(OR_TILE / d->m->decim))
			d->bias = ((float)(S1 + S0) * or_recip((float)K)) * (1.0f / 4096.0f);
		float err = ((float)E * or_recip((float)(K > OR_ROUND_MAX ? OR_ROUND_MAX : K))) * (1.0f / 1024.0f);   /* the symbols that fed the detector */
		err = err * or_recip(d->amp * d->amp);
		err = clampf(err, -1.0f, 1.0f);
		const float kp = (float)d->m->period0 * 0.159154943f;   /* 0.5/pi of a symbol, Q16 samples */
		const float ki = kp * (1.0f / 4096.0f);
		const int32_t dphase = (int32_t)lrintf(err * kp);
		const int32_t dper = (int32_t)lrintf(err * ki);
		d->t_next += (int64_t)K * d->period + dphase;
		/* SPEC 3.6b (round 5), acquisition aid of the AFSK streams: a Gardner loop that starts half a symbol off sits on the detector's
		 * unstable zero, and a loop that is closed three (iMet) or six (C50) times a second takes many seconds to drift off it (9 % of random
		 * iMet channels decoded nothing in their first two seconds).  There the mid-symbol samples show the open eye and the on-time ones the
		 * transitions: when the mid-symbol samples lie further from the threshold (by 17/16) than the on-time ones, the instants move by half
		 * a symbol, once the slicer has seen both levels. */
		if (d->m->pre == 8 && d->nstat && 16 * (int64_t)SM > 17 * (int64_t)SY) d->t_next += d->period >> 1;
		d->period += dper;
		const int32_t pmin = d->m->period0 - (d->m->period0 >> 8);
		const int32_t pmax = d->m->period0 + (d->m->period0 >> 8);
		if (d->period < pmin) d->period = pmin;
		if (d->period > pmax) d->period = pmax;
	}
}

/* ---- AFSK tone demodulator (SPEC 3.6; iMet-1/4: Bell-202 tones 1200 Hz = mark, 2200 Hz = space at 1200 Bd on the
 * FM audio; SURVEY.md Appendix B [RECALL]).  The discriminator output d[n] (the audio) is mixed with a 1700 Hz
 * complex oscillator, so mark/space sit at -/+500 Hz, low-passed by a one-symbol boxcar (40 samples, nulls at
 * multiples of 1200 Hz: the images at -2900/-3900 Hz are down 18-23 dB), decimated 8:1 and FM-discriminated
 * again: q[m] ~ -/+ 1/3 quadrant per sample at 6 kS/s, which the same timing loop / slicer as for the GFSK
 * sondes turns into bits.  Mixer table: 480 entries = 17 cycles (1700/48000 = 17/480), (cos, -sin) from double. */
void or_afsk_table(float *w)
{
	for (int k = 0; k < OR_AF_PER; k++) {
		const double a = 2.0 * OR_PI_D * 17.0 * (double)k / (double)OR_AF_PER;
		w[2 * k] = (float)cos(a);
		w[2 * k + 1] = (float)(-sin(a));
	}
}

/* SRS-C50: tones 2900 / 4700 Hz at 2400 Bd: mixer at 3800 Hz (19 cycles in 240 samples), boxcar over the last two
 * 8-sample blocks (a symbol is 20 samples), the rest as for iMet.  [RECALL: public C34/C50 decoder notes.] */
#define OR_C50_PER 240
#define OR_C50_WIN 2
static void mixer_table(float *w, int cycles, int per)
{
	for (int k = 0; k < per; k++) {
		const double a = 2.0 * OR_PI_D * (double)cycles * (double)k / (double)per;
		w[2 * k] = (float)cos(a);
		w[2 * k + 1] = (float)(-sin(a));
	}
}

/* n_in input samples (a multiple of 8) -> n_in/8 samples of q */
static void afsk_front(OrDemod *d, const float *src, size_t n_in, int is_iq, float *q)
{
	static float W17[2 * OR_AF_PER], W19[2 * OR_C50_PER];
	static int have;
	if (!have) { or_afsk_table(W17); mixer_table(W19, 19, OR_C50_PER); have = 1; }
	const int c50 = d->m->type == OR_C50;
	const float *W = c50 ? W19 : W17;
	const unsigned per = c50 ? OR_C50_PER : OR_AF_PER;
	const int win = c50 ? OR_C50_WIN : OR_AF_WIN;
	for (size_t m = 0; m < n_in / OR_AF_DEC; m++) {
		float dv[OR_AF_DEC];
		if (is_iq) or_discriminate(src + 2 * OR_AF_DEC * m, OR_AF_DEC, dv, d->iq_last);
		else memcpy(dv, src + OR_AF_DEC * m, sizeof(dv));
		float br = 0.0f, bi = 0.0f;
		for (int i = 0; i < OR_AF_DEC; i++) {
			const unsigned k = (unsigned)((d->af_n + OR_AF_DEC * m + (size_t)i) % per);
			br = fmaf(dv[i], W[2 * k], br);
			bi = fmaf(dv[i], W[2 * k + 1], bi);
		}
		/* boxcar over the last `win` block sums (the current one included), oldest first */
		float zr, zi;
		if (win == OR_AF_WIN) {
			zr = (((d->af_b[0][0] + d->af_b[1][0]) + d->af_b[2][0]) + d->af_b[3][0]) + br;
			zi = (((d->af_b[0][1] + d->af_b[1][1]) + d->af_b[2][1]) + d->af_b[3][1]) + bi;
		} else {
			zr = d->af_b[3][0] + br;
			zi = d->af_b[3][1] + bi;
		}
		for (int h = 0; h < OR_AF_WIN - 2; h++) { d->af_b[h][0] = d->af_b[h + 1][0]; d->af_b[h][1] = d->af_b[h + 1][1]; }
		d->af_b[OR_AF_WIN - 2][0] = br;
		d->af_b[OR_AF_WIN - 2][1] = bi;
		/* second discriminator: the same product form as stage 1 */
		const float zz[2] = { zr, zi };
		or_discriminate(zz, 1, &q[m], d->af_z);
	}
	d->af_n += n_in;
}

/* One 2048-sample input tile at a time.  Stage K0 (SPEC 3.0): IQ is first decimated by a boxcar -- 4:1 for the sondes at
 * about 5000 chips/s (RS41, DFM, iMS-100, MRZ-N1: z[m] = (x[4m] + x[4m+1]) + (x[4m+2] + x[4m+3]), 12 kS/s behind it),
 * 2:1 for M10 (9600 chips/s: z[m] = x[2m] + x[2m+1], 24 kS/s); real discriminator input is averaged the same way --
 * so that the discriminator and everything behind it run at about 2.5 samples per symbol.  This is the reference's own
 * ordering (the VFO hands dsp::demod::FM a stream at the channel bandwidth, 10 kS/s for RS41:
 * /root/reference/src/main.cpp:55-57, src/main.hpp:45), cuts the arithmetic per input sample and lowers the FM threshold by
 * narrowing the pre-detection noise bandwidth (2-2.5 dB per halving, profiles/r3_sensitivity.md).  or_modem_set_decim()
 * moves a type one step wider (the product's SONDE_FLAG_WIDE). */
void or_demod_feed(OrDemod *d, const float *src, size_t n, int is_iq)
{
	{	/* one allocation per feed instead of doubling reallocs inside the symbol loop (many threads feed at once) */
		const int32_t pmin = d->m->period0 - (d->m->period0 >> 8);
		const uint64_t want = d->nbits + (((uint64_t)(n / (size_t)(d->m->decim * d->m->pre)) << 16) / (uint64_t)pmin) + 64;
		if (want > d->cap) {
			d->cap = want;
			d->bits = realloc(d->bits, d->cap);
		}
	}
	if (d->m->pre > 1) {
		/* AFSK: one tile of the demodulator = 2048 samples behind the tone demodulator = 16384 input samples
		 * (n must be a multiple of that) */
		const size_t blk = (size_t)OR_TILE * (size_t)d->m->pre;
		float q[OR_TILE];
		for (size_t off = 0; off + blk <= n; off += blk) {
			afsk_front(d, src + (is_iq ? 2 : 1) * off, blk, is_iq, q);
			for (int i = 0; i < OR_TILE; i++) d->ring[(d->n0 + i) & (OR_RING - 1)] = q[i];
			d->n0 += OR_TILE;
			run_rounds(d);
		}
		return;
	}
	const int dec = d->m->decim, it = OR_TILE / dec;
	float tile[OR_TILE], z[2 * OR_TILE];
	for (size_t off = 0; off + OR_TILE <= n; off += OR_TILE) {
		if (is_iq == 1) {
			const float *x = src + 2 * off;
			/* SPEC 3.0d (round 5), the carrier-following boxcar: with the carrier f off, consecutive input samples turn by
			 * w = 2 pi f / 48000, and a plain sum of 4 (2) of them sees the far half of the signal's band through the boxcar's
			 * droop (-3.6 dB at 5.8 kHz): RS41 at Eb/N0 10 dB and +-1 kHz lost 6 % of the frames a conventional receiver decodes
			 * (profiles/r4_yardstick.md).  The second half of every group is turned back before it is added,
			 *   4:1  z = (x0 + x1) + R (x2 + x3),   2:1  z = x0 + R x1,   R = (1 - u^2 / 2, -u),
			 * u the AFC state of the tile (SPEC 3.0b: 2 atan u per decimated sample, so atan u ~ u is the turn of half a group
			 * in either class); R's length does not matter.  R p = (fmaf(-p.im, R.im, p.re R.re), fmaf(p.re, R.im, p.im R.re)). */
			const float u0 = d->afc[0], rr = fmaf(-0.5f * u0, u0, 1.0f), ri = -u0;
			if (dec == 4) {
				for (int m = 0; m < it; m++) {
					const float p0r = x[8 * m] + x[8 * m + 2], p0i = x[8 * m + 1] + x[8 * m + 3];
					const float p1r = x[8 * m + 4] + x[8 * m + 6], p1i = x[8 * m + 5] + x[8 * m + 7];
					z[2 * m] = p0r + fmaf(-p1i, ri, p1r * rr);
					z[2 * m + 1] = p0i + fmaf(p1r, ri, p1i * rr);
				}
				discriminate_rot(z, (size_t)it, tile, d->iq_last, d->afc[0]);
			} else if (dec == 2) {
				for (int m = 0; m < it; m++) {
					const float p1r = x[4 * m + 2], p1i = x[4 * m + 3];
					z[2 * m] = x[4 * m] + fmaf(-p1i, ri, p1r * rr);
					z[2 * m + 1] = x[4 * m + 1] + fmaf(p1r, ri, p1i * rr);
				}
				discriminate_rot(z, (size_t)it, tile, d->iq_last, d->afc[0]);
			} else {
				discriminate_rot(x, (size_t)it, tile, d->iq_last, d->afc[0]);
			}
		} else if (is_iq == 2) {
			/* real input already decimated by this type's factor (SPEC 3.5b: the channelizer's composite filter): n still
			 * counts 48 kS/s samples, src holds n / dec */
			memcpy(tile, src + off / (size_t)dec, sizeof(float) * (size_t)it);
		} else {
			const float *x = src + off;
			if (dec == 4) {
				for (int m = 0; m < it; m++) tile[m] = ((x[4 * m] + x[4 * m + 1]) + (x[4 * m + 2] + x[4 * m + 3])) * 0.25f;
			} else if (dec == 2) {
				for (int m = 0; m < it; m++) tile[m] = (x[2 * m] + x[2 * m + 1]) * 0.5f;
			} else {
				memcpy(tile, x, sizeof(float) * (size_t)it);
			}
		}
		for (int i = 0; i < it; i++)
			d->ring[(d->n0 + i) & (OR_RING - 1)] = tile[i];
		d->n0 += it;
		run_rounds(d);
		if (is_iq == 1) {
			/* SPEC 3.0b, AFC: the slicer's threshold `bias` is the mean of the discriminator output, i.e. what is left of the
			 * carrier offset behind the rotation: a leaky integrator moves the rotation after every tile.  The discriminator
			 * runs ahead of the timing loop on the GPU (other waves of the workgroup), so the state after tile i is what the
			 * discriminator of tile i + 3 uses: a three-deep FIFO. */
			float u = d->afc[2];
			u = fmaf(-OR_AFC_LEAK, u, u);
			u = fmaf(OR_AFC_GAIN, d->bias, u);
			u = clampf(u, -OR_AFC_MAX, OR_AFC_MAX);
			/* SPEC 3.0e (round 5), the threshold follows the rotation: the next tile's discriminator turns the signal back by
			 * rot(afc[1]) instead of rot(afc[0]) quadrants per sample, rot(u) = (4 / pi) atan(u) (the phasor (1 - u^2, 2u) of
			 * discriminate_rot turns by 2 atan u) as an odd polynomial (afc_rot): its output drops by the difference, and so does
			 * the slicer's threshold, now, instead of following through its smoothing a few tiles later (MRZ-N1 at -2 kHz and
			 * Eb/N0 14 dB lost the first frame of 9 channels of 32 to that; profiles/r5_notes.md section 7).  The AFC's input stays
			 * the threshold: what is left of the offset behind the rotation the next tile will see. */
			d->bias -= afc_rot(d->afc[1]) - afc_rot(d->afc[0]);
			d->afc[0] = d->afc[1];
			d->afc[1] = d->afc[2];
			d->afc[2] = u;
		}
	}
}

uint64_t or_demod_nbits(const OrDemod *d) { return d->nbits; }

void or_demod_getbits(const OrDemod *d, uint64_t from, size_t count, uint8_t *out)
{
	for (size_t i = 0; i < count; i++)
		out[i] = (from + i < d->nbits) ? d->bits[from + i] : 0;
}

void or_demod_state(const OrDemod *d, int64_t *t_next, int32_t *period, float *bias, float *amp, float *yprev)
{
	if (t_next) *t_next = d->t_next;
	if (period) *period = d->period;
	if (bias) *bias = d->bias;
	if (amp) *amp = d->amp;
	if (yprev) *yprev = d->afc[2];      /* (round 4: the newest AFC state u, SPEC 3.0b; 0 for real input) */
}

/* bits produced elsewhere (the conventional yardstick demodulator, or_yardstick.c) into the container the framers read */
void or_demod_append_bits(OrDemod *d, const uint8_t *b, size_t n)
{
	for (size_t i = 0; i < n; i++) push_bit(d, b[i]);
}

/* internal accessor for the framer */
const uint8_t *or_demod_bitptr(const OrDemod *d) { return d->bits; }
