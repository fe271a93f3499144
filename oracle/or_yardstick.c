/*
 * or_yardstick.c -- a CONVENTIONAL GFSK demodulator, as a yardstick for the SPEC demodulator of or_dsp.c.
 * TEST INFRASTRUCTURE ONLY (see sonde_oracle.h): never linked into, loaded by or measured as the product.
 *
 * Why it exists ("What's missing" 1): the oracle's demodulator (or_dsp.c) is the GPU kernel's twin -- its
 * arithmetic contract (round-wise loop update, polynomial arctangent, boxcar decimation) was shaped so that 256 lanes and a
 * sequential loop agree bit for bit, and was re-tuned for kernel speed.  This file shares NONE of that: it is the textbook
 * per-sample receiver that SURVEY.md Appendix B.1 records for sondedump's gfsk_demod (the library behind
 * X_decode(), /root/reference/src/decode/decoder.hpp:22,61) and that SDR++ puts in front of it
 * (/root/reference/src/main.cpp:55-60: VFO channel filter of the sonde type's bandwidth, main.hpp:44-52 -> dsp::demod::FM):
 *
 *   complex IQ @ 48 kS/s
 *   -> channel filter: windowed-sinc low-pass, cutoff = VFO bandwidth / 2 (10 / 15 / 20 kHz channels; 50 kHz: none)
 *   -> FM discriminator with libm atan2f (arg of x[n] conj(x[n-1]))
 *   -> AGC: DC (carrier-offset) removal + amplitude normalisation, exponential averages
 *   -> polyphase low-pass FIR (16 phases, 4 symbols long, cutoff = cutoff_rel x symbol rate)
 *   -> PER-SYMBOL Gardner timing recovery: an NCO stepped through the 16 interpolation phases of every input sample
 *      picks the mid-symbol and the symbol instants; error (prev - cur) * mid through a PI loop filter, updated every symbol
 *   -> hard slicer -> bits
 *   -> the EXISTING oracle framers / FEC (or_framer_run: sync search, de-whitening, RS(255,231), Hamming, BCH, checksums).
 *
 * It answers one question: does the SPEC decode the frames a normal CPU decoder decodes?  (tests/test_yardstick.py,
 * profiles/r4_yardstick.md.)  It is not bit-exact to anything and is not meant to be.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "sonde_oracle.h"

#define YD_PI      3.14159265358979323846
#define YD_NCH     63            /* channel-filter taps */
#define YD_P       16            /* interpolation phases of the symbol filter */
#define YD_NT_MAX  96            /* symbol-filter taps per phase, upper bound */

void or_demod_append_bits(OrDemod *d, const uint8_t *b, size_t n);      /* or_dsp.c */

/* VFO bandwidth of each sonde type: supportedTypes[], /root/reference/src/main.hpp:44-52 */
static const double k_vfo_bw[OR_NTYPES] = { 10e3, 15e3, 20e3, 50e3, 20e3, 20e3, 20e3 };

struct OrYard {
	int type;
	double sps;                       /* input samples per symbol */
	/* channel filter */
	int use_chf;
	float chf[YD_NCH];
	float xi[YD_NCH], xq[YD_NCH];
	int xpos;
	float pi_, pq_;                   /* previous (filtered) sample */
	/* AGC */
	float bias, mag;
	float k_agc;
	uint64_t nseen;
	/* symbol filter */
	int nt;
	float lpf[YD_P][YD_NT_MAX];
	float dl[YD_NT_MAX];
	int dpos;
	/* timing loop */
	double phase, rel, alpha, beta, maxrel;
	int have_mid;
	float mid, prev;
	/* output */
	OrDemod *bits;
	OrFramer *fr;
};

static double blackman(int i, int n) { const double x = (double)i / (double)(n - 1); return 0.42 - 0.5 * cos(2.0 * YD_PI * x) + 0.08 * cos(4.0 * YD_PI * x); }

OrYard *or_yard_new(int type, uint32_t channel, float cutoff_rel, float loop_bw)
{
	const OrModem *m = or_modem(type);
	if (!m || m->pre != 1) return NULL;                    /* GFSK sondes only */
	OrYard *y = calloc(1, sizeof(*y));
	y->type = type;
	y->sps = (double)OR_FS / m->baud;
	if (cutoff_rel <= 0.0f) cutoff_rel = 1.0f;             /* "cut-off ~ symbol rate"  */
	if (loop_bw <= 0.0f) loop_bw = 0.01f;                  /* loop noise bandwidth, in symbol rates */
	/* channel filter: cutoff bw/2 */
	y->use_chf = k_vfo_bw[type] < (double)OR_FS;
	if (y->use_chf) {
		const double fc = 0.5 * k_vfo_bw[type] / (double)OR_FS;
		double sum = 0.0, h[YD_NCH];
		for (int i = 0; i < YD_NCH; i++) {
			const double t = (double)i - 0.5 * (YD_NCH - 1);
			h[i] = (t == 0.0 ? 2.0 * fc : sin(2.0 * YD_PI * fc * t) / (YD_PI * t)) * blackman(i, YD_NCH);
			sum += h[i];
		}
		for (int i = 0; i < YD_NCH; i++) y->chf[i] = (float)(h[i] / sum);
	}
	/* symbol filter: 4 symbols long at the input rate, 16 interpolation phases */
	y->nt = (int)ceil(4.0 * y->sps);
	if (y->nt > YD_NT_MAX) y->nt = YD_NT_MAX;
	{
		const int n = y->nt * YD_P;
		const double fc = (double)cutoff_rel * m->baud / ((double)OR_FS * YD_P);      /* cycles per up-sampled sample */
		double *h = malloc(sizeof(double) * (size_t)n), sum = 0.0;
		for (int i = 0; i < n; i++) {
			const double t = (double)i - 0.5 * (n - 1);
			h[i] = (t == 0.0 ? 2.0 * fc : sin(2.0 * YD_PI * fc * t) / (YD_PI * t)) * blackman(i, n);
			sum += h[i];
		}
		for (int p = 0; p < YD_P; p++)
			for (int j = 0; j < y->nt; j++) y->lpf[p][j] = (float)(h[p + YD_P * j] * YD_P / sum);      /* unit DC gain per phase */
		free(h);
	}
	y->k_agc = 1.0f / 4096.0f;                             /* 85 ms */
	y->mag = 0.0f;
	/* second-order loop, damping 0.707, detector gain ~ 2 (Gardner on a unit-amplitude, band-limited NRZ signal) */
	{
		const double zeta = 0.7071, kd = 2.0;
		const double th = (double)loop_bw / (zeta + 0.25 / zeta);
		const double den = 1.0 + 2.0 * zeta * th + th * th;
		y->alpha = 4.0 * zeta * th / den / kd;              /* symbols of phase per unit error */
		y->beta = 4.0 * th * th / den / kd;                 /* relative clock per unit error */
	}
	y->maxrel = 1.0 / 256.0;                               /* the range the SPEC loop allows as well */
	y->bits = or_demod_new(type);                          /* used as the bit container the framers read */
	y->fr = or_framer_new(type, channel);
	return y;
}

void or_yard_free(OrYard *y) { if (y) { or_demod_free(y->bits); or_framer_free(y->fr); free(y); } }

static inline float yd_symfilt(const OrYard *y, int p)
{
	/* y(n + p/16 - delay) = sum_j h[p + 16 j] x[n - j] */
	float acc = 0.0f;
	int k = y->dpos;
	for (int j = 0; j < y->nt; j++) {
		acc += y->lpf[p][j] * y->dl[k];
		k = k ? k - 1 : y->nt - 1;
	}
	return acc;
}

void or_yard_feed(OrYard *y, const float *iq, size_t n)
{
	uint8_t out[4096];
	size_t nout = 0;
	const double step = 1.0 / (y->sps * YD_P);             /* nominal symbols per interpolation step */
	for (size_t i = 0; i < n; i++) {
		float fi = iq[2 * i], fq = iq[2 * i + 1];
		if (y->use_chf) {
			y->xi[y->xpos] = fi; y->xq[y->xpos] = fq;
			float ai = 0.0f, aq = 0.0f;
			int k = y->xpos;
			for (int j = 0; j < YD_NCH; j++) {
				ai += y->chf[j] * y->xi[k]; aq += y->chf[j] * y->xq[k];
				k = k ? k - 1 : YD_NCH - 1;
			}
			y->xpos = (y->xpos + 1) % YD_NCH;
			fi = ai; fq = aq;
		}
		/* FM discriminator: arg(x[n] conj(x[n-1])), libm */
		const float re = fi * y->pi_ + fq * y->pq_, im = fq * y->pi_ - fi * y->pq_;
		float d = (re == 0.0f && im == 0.0f) ? 0.0f : atan2f(im, re);
		y->pi_ = fi; y->pq_ = fq;
		/* AGC: carrier offset (DC) and level */
		const float k = y->nseen < 4096 ? 1.0f / (float)(y->nseen + 1) : y->k_agc;       /* plain average while the window fills */
		y->nseen++;
		y->bias += k * (d - y->bias);
		d -= y->bias;
		y->mag += k * (fabsf(d) - y->mag);
		d *= 1.0f / fmaxf(y->mag, 1e-6f);
		y->dpos = (y->dpos + 1) % y->nt;
		y->dl[y->dpos] = d;
		/* timing NCO through the interpolation phases of this sample */
		for (int p = 0; p < YD_P; p++) {
			y->phase += step * (1.0 + y->rel);
			if (!y->have_mid && y->phase >= 0.5) {
				y->mid = yd_symfilt(y, p);
				y->have_mid = 1;
			} else if (y->phase >= 1.0) {
				const float cur = yd_symfilt(y, p);
				y->phase -= 1.0;
				y->have_mid = 0;
				double err = (double)(y->prev - cur) * (double)y->mid;
				if (err > 1.0) err = 1.0;
				if (err < -1.0) err = -1.0;
				/* sampling late -> err < 0 -> the next instant must come sooner */
				y->phase -= y->alpha * err;
				y->rel -= y->beta * err;
				if (y->rel > y->maxrel) y->rel = y->maxrel;
				if (y->rel < -y->maxrel) y->rel = -y->maxrel;
				y->prev = cur;
				out[nout++] = cur > 0.0f;
				if (nout == sizeof(out)) { or_demod_append_bits(y->bits, out, nout); nout = 0; }
			}
		}
	}
	if (nout) or_demod_append_bits(y->bits, out, nout);
	or_framer_run(y->fr, y->bits);
}

size_t or_yard_nframes(const OrYard *y) { return or_framer_nframes(y->fr); }
const OrFrame *or_yard_frame(const OrYard *y, size_t i) { return or_framer_frame(y->fr, i); }
uint64_t or_yard_nbits(const OrYard *y) { return or_demod_nbits(y->bits); }
void or_yard_getbits(const OrYard *y, uint64_t from, size_t count, uint8_t *out) { or_demod_getbits(y->bits, from, count, out); }

/* channels [0, nch) of n complex samples each, channel-major; frames in (channel, time) order; returns their number */
size_t or_yard_batch_run(int type, const float *iq, size_t nch, size_t n, int nthreads, float cutoff_rel, float loop_bw, OrFrame *out, size_t cap)
{
	size_t total = 0;
	size_t *counts = calloc(nch, sizeof(size_t));
	OrYard **ys = calloc(nch, sizeof(*ys));
	(void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
	for (long c = 0; c < (long)nch; c++) {
		ys[c] = or_yard_new(type, (uint32_t)c, cutoff_rel, loop_bw);
		if (!ys[c]) continue;
		or_yard_feed(ys[c], iq + 2 * (size_t)c * n, n);
		counts[c] = or_yard_nframes(ys[c]);
	}
	for (size_t c = 0; c < nch; c++) {
		for (size_t i = 0; i < counts[c]; i++) {
			if (out && total < cap) out[total] = *or_yard_frame(ys[c], i);
			total++;
		}
		or_yard_free(ys[c]);
	}
	free(ys);
	free(counts);
	return total;
}
