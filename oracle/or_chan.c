/*
 * or_chan.c -- oracle for the wideband front-end (BASELINE config 4):
 *   10 MS/s complex IQ -> 512-bin polyphase filter bank (decimation 500 -> 20 kS/s per bin: the bin spacing, 19.53 kHz,
 *   1.024 x oversampled) -> per-bin instantaneous PHASE phi = atan2q(bin sample) -> FM discriminator as the wrapped phase
 *   difference d[m] = wrap(phi[m] - phi[m-1]) at 20 kS/s -> real rational resampler 12/5 -> 48 kS/s
 * (round 4: rounds 2-3 ran the bank at 40 kS/s per bin and stored complex bins -- 16.4 B written and re-read per wideband
 * sample, 5.2 x the algorithmic traffic; a bin now leaves the filter bank as ONE float per 500 wideband samples)
 * which is the reference's own ordering  VFO channeliser -> dsp::demod::FM -> RationalResampler -> decoder
 * (/root/reference/src/main.cpp:55-60).  Those SDR++ blocks are absent from the reference tree, so the
 * arithmetic is this repo's SPEC.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * Bit-exactness contract as elsewhere: -ffp-contract=off, explicit fmaf, fixed summation orders.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "sonde_oracle.h"

#define CH_PI 3.14159265358979323846

/* prototype low-pass of the filter bank: Blackman-windowed sinc, cutoff 8 kHz at 10 MS/s, unit DC gain */
void or_chan_proto(float *h /* OR_CH_L */)
{
	const double fc = 8000.0 / OR_CH_FS;
	double sum = 0.0;
	static double tmp[OR_CH_L];
	for (int i = 0; i < OR_CH_L; i++) {
		const double t = (double)i - 0.5 * (double)(OR_CH_L - 1);
		const double x = (double)i / (double)(OR_CH_L - 1);
		const double w = 0.42 - 0.5 * cos(2.0 * CH_PI * x) + 0.08 * cos(4.0 * CH_PI * x);
		const double s = (t == 0.0) ? 2.0 * fc : sin(2.0 * CH_PI * fc * t) / (CH_PI * t);
		tmp[i] = s * w;
		sum += tmp[i];
	}
	for (int i = 0; i < OR_CH_L; i++) h[i] = (float)(tmp[i] / sum);
}

/* twiddles w[k] = exp(-2 pi i k / 512), k < 256 */
void or_chan_twiddles(float *tw /* 2*256, (re, im) */)
{
	for (int k = 0; k < OR_CH_M / 2; k++) {
		tw[2 * k] = (float)cos(2.0 * CH_PI * (double)k / (double)OR_CH_M);
		tw[2 * k + 1] = (float)(-sin(2.0 * CH_PI * (double)k / (double)OR_CH_M));
	}
	/* SPEC 3.5 (round 6): the TRIVIAL twiddles are exact.  tw[0] = (1, -0) already is; tw[128] = exp(-j pi / 2) is (0, -1), not the
	 * (6.1e-17, -1) that cos(pi / 2) in double rounds to: multiplying by it is then a swap with one sign change, exactly (up to the sign
	 * of a zero, which no later stage can turn into a different phase), and the GPU's wave-uniform butterflies of stages 1-3 use
	 * additions only (csrc/channelizer.hip pfb_fft512n: 40 of a wave's ~800 vector instructions per step). */
	tw[2 * 128] = 0.0f;
}

/* Rational resampler prototype (polyphase, `up` phases of OR_RS_T taps): Blackman-windowed sinc at up * rate_in with the
 * given cutoff; g[p][t] = proto[up t + p], every phase normalised to unit DC gain.  The channelizer's 6/5 stage is
 * (6, 240 kHz, 18 kHz); the VFO front-end's ratios follow below. */
void or_resamp_taps(int up, double fs_up_hz, double cutoff_hz, float *g /* up * OR_RS_T */)
{
	const int N = up * OR_RS_T;
	const double fc = cutoff_hz / fs_up_hz;
	double *tmp = malloc((size_t)N * sizeof(double));
	for (int i = 0; i < N; i++) {
		const double t = (double)i - 0.5 * (double)(N - 1);
		const double x = (double)i / (double)(N - 1);
		const double w = 0.42 - 0.5 * cos(2.0 * CH_PI * x) + 0.08 * cos(4.0 * CH_PI * x);
		const double s = (t == 0.0) ? 2.0 * fc : sin(2.0 * CH_PI * fc * t) / (CH_PI * t);
		tmp[i] = s * w;
	}
	for (int p = 0; p < up; p++) {
		double sum = 0.0;
		for (int t = 0; t < OR_RS_T; t++) sum += tmp[t * up + p];
		for (int t = 0; t < OR_RS_T; t++) g[p * OR_RS_T + t] = (float)(tmp[t * up + p] / sum);
	}
	free(tmp);
}
void or_chan_resamp_taps(float *g /* 12*16 */) { or_resamp_taps(OR_RS_L, 240000.0, 9000.0, g); }     /* = the VFO front-end's 20 kS/s taps */

/* SPEC 3.5b (round 3): behind the channelizer the 6/5 resampler and the boxcar decimator of SPEC 3.0 (4:1 or 2:1) are ONE
 * polyphase filter when the product runs them inside the decoder kernel (its default, "fused" mode): decimated sample n of a
 * block is z[n] = sum_k fmaf(G[n mod 3][k], d[b(n) - k], acc), k = 0 .. KT-1 ascending, b(n) = floor(5 (dec n + dec - 1) / 6)
 * the newest discriminator sample it reads, KT = 19 (4:1) or 17 (2:1), and the composite row
 * G[phi][k] = (float)((1/dec) sum_{i < dec} g[5 j mod 6][k - (b - floor(5 j / 6))], j = dec phi + i, terms outside 0..15 dropped)
 * summed in double from the float taps g.  (The rows repeat every 3 decimated samples: 12 or 6 resampler outputs, 10 or 5
 * discriminator samples.)  It is the same filter as "resample, then average" up to rounding -- 19 multiply-adds per decimated
 * sample instead of 64 -- and no 48 kS/s row exists. */
int or_chan_composite_kt(int dec) { (void)dec; return 17; }
/* round 4 (20 kS/s bins, 12/5 resampler): decimated sample n (12 kS/s) = the mean of the 48 kS/s outputs j = 4n .. 4n+3,
 * output j = sum_t g[5j mod 12][t] d[floor(5j/12) - t]; newest input b(n) = floor(5 (4n + 3) / 12); the rows repeat every 3
 * decimated samples (12 outputs, 5 inputs); 17 taps in use. */
void or_chan_composite_taps(const float *g /* 12*16 */, int dec, float *G /* 3 * OR_RS_KT_LD (20) */)
{
	for (int phi = 0; phi < 3; phi++) {
		const int j0 = dec * phi, b = (OR_RS_M * (j0 + dec - 1)) / OR_RS_L;
		for (int k = 0; k < OR_RS_KT_LD; k++) {
			double sum = 0.0;
			for (int i = 0; i < dec; i++) {
				const int j = j0 + i, t = k - (b - (OR_RS_M * j) / OR_RS_L);
				if (t >= 0 && t < OR_RS_T) sum += (double)g[((OR_RS_M * j) % OR_RS_L) * OR_RS_T + t];
			}
			G[phi * OR_RS_KT_LD + k] = (float)(sum / (double)dec);
		}
	}
}

/* ---- VFO front-end (SURVEY 8 rows a1 + a2 at the reference's own rates): what sits between the VFO and the decoder in
 * /root/reference/src/main.cpp:55-60 -- IQ at the sonde type's VFO bandwidth (supportedTypes[], main.hpp:44-52: 10, 15, 20
 * or 50 kS/s) -> dsp::demod::FM -> dsp::RationalResampler to 48 kS/s (24/5, 16/5, 12/5, 24/25).  Cutoff: 0.45 of the
 * lower of the two rates.  One channel; n_in % down == 0; returns the output samples written (n_in * up / down). */
struct OrVfo {
	int up, down;
	float *g;
	float iq_last[2];
	float dhist[OR_RS_T];
};
int or_vfo_ratio(int rate_in, int *up, int *down, int *cutoff_hz)
{
	switch (rate_in) {
	case 10000: *up = 24; *down = 5;  *cutoff_hz = 4500;  return 0;
	case 15000: *up = 16; *down = 5;  *cutoff_hz = 6750;  return 0;
	case 20000: *up = 12; *down = 5;  *cutoff_hz = 9000;  return 0;
	case 40000: *up = 6;  *down = 5;  *cutoff_hz = 18000; return 0;      /* a channelizer bin */
	case 50000: *up = 24; *down = 25; *cutoff_hz = 21600; return 0;
	}
	return -1;
}
OrVfo *or_vfo_new(int rate_in)
{
	int up, down, fc;
	if (or_vfo_ratio(rate_in, &up, &down, &fc)) return NULL;
	OrVfo *v = calloc(1, sizeof(*v));
	v->up = up; v->down = down;
	v->g = malloc((size_t)up * OR_RS_T * sizeof(float));
	or_resamp_taps(up, (double)rate_in * up, (double)fc, v->g);
	return v;
}
void or_vfo_free(OrVfo *v) { if (v) { free(v->g); free(v); } }
size_t or_vfo_process(OrVfo *v, const float *iq, size_t n_in, float *out48)
{
	const size_t n_out = n_in * (size_t)v->up / (size_t)v->down;
	float *d = malloc((OR_RS_T + n_in) * sizeof(float));
	memcpy(d, v->dhist, OR_RS_T * sizeof(float));
	or_discriminate(iq, n_in, d + OR_RS_T, v->iq_last);
	for (size_t j = 0; j < n_out; j++) {
		const size_t i0 = (j * (size_t)v->down) / (size_t)v->up;
		const int p = (int)((j * (size_t)v->down) % (size_t)v->up);
		float acc = 0.0f;
		for (int t = 0; t < OR_RS_T; t++) acc = fmaf(v->g[p * OR_RS_T + t], d[OR_RS_T + i0 - t], acc);
		out48[j] = acc;
	}
	memcpy(v->dhist, d + n_in, OR_RS_T * sizeof(float));
	free(d);
	return n_out;
}

/* in-place radix-2 decimation-in-time FFT, 512 points, bit-reversed load; each butterfly:
 *   t = b * w  with  t.re = fmaf(-b.im, w.im, b.re*w.re),  t.im = fmaf(b.re, w.im, b.im*w.re);  a' = a + t, b' = a - t */
void or_fft512(float *re, float *im, const float *tw)
{
	for (int i = 0; i < OR_CH_M; i++) {
		int r = 0;
		for (int b = 0; b < 9; b++) r |= ((i >> b) & 1) << (8 - b);
		if (r > i) {
			float t = re[i]; re[i] = re[r]; re[r] = t;
			t = im[i]; im[i] = im[r]; im[r] = t;
		}
	}
	for (int s = 1; s <= 9; s++) {
		const int half = 1 << (s - 1), step = OR_CH_M >> s;
		for (int g0 = 0; g0 < OR_CH_M; g0 += 2 * half) {
			for (int j = 0; j < half; j++) {
				const float wr = tw[2 * (j * step)], wi = tw[2 * (j * step) + 1];
				const int a = g0 + j, b = a + half;
				const float tr = fmaf(-im[b], wi, re[b] * wr);
				const float ti = fmaf(re[b], wi, im[b] * wr);
				const float ar = re[a], ai = im[a];
				re[a] = ar + tr; im[a] = ai + ti;
				re[b] = ar - tr; im[b] = ai - ti;
			}
		}
	}
}

struct OrChan {
	int odd;                     /* the odd-stacked bank (round 4): bin k centred at (k + 1/2) x 19531.25 Hz */
	float wtw[2 * OR_CH_M];      /* its twist W[r] = exp(-j pi r / 512) */
	float h[OR_CH_L], tw[OR_CH_M], g[OR_RS_L * OR_RS_T];
	float *hist;                 /* last L-D wideband samples (I,Q) */
	float phi_last[OR_CH_M];     /* per bin: the previous phase sample */
	float dhist[OR_CH_M][OR_RS_T];   /* per bin: last 16 discriminator samples (oldest first) */
};

/* The ODD-STACKED bank (SPEC 3.5c, round 4): the same bank applied to the stream shifted down by half a bin,
 * x'[n] = x[n] exp(-j pi n / 512), so that bin k is centred at (k + 1/2) bin spacings: together with the even bank every carrier
 * on the reference's 1 kHz VFO raster (/root/reference/src/main.cpp:14,55-56) lies within 4.9 kHz of a bin centre.  Without touching
 * the samples: window sample i = r + 512 t of step m is stream sample n0 + i, n0 = 500 m - 7692, so
 * x'[n0 + i] = x[n0 + i] (-1)^t W[r] exp(-j pi n0 / 512): the taps of odd t change sign (exact), the folded value is multiplied by
 * the twist W[r] = exp(-j pi r / 512) (t.re = fmaf(-v.im, W.im, v.re W.re), t.im = fmaf(v.re, W.im, v.im W.re)), and the step's
 * common phase -pi 500 m / 512 = -(125 / 64) m quadrants (the constant part is dropped: a discriminator does not see it) is taken
 * off the phase samples: phi = wrap(atan2q(...) - ramp(m)), ramp(m) = (125 (m mod 256)) / 64 wrapped into [-2, 2] (exact in
 * float; 256 steps are a whole number of turns, and a block is a multiple of 256 steps: no state). */
void or_chan_twist(float *w /* 2*512: (re, im) of exp(-j pi r / 512) */)
{
	for (int r = 0; r < OR_CH_M; r++) {
		w[2 * r] = (float)cos(CH_PI * (double)r / (double)OR_CH_M);
		w[2 * r + 1] = (float)(-sin(CH_PI * (double)r / (double)OR_CH_M));
	}
}
float or_chan_ramp(size_t m)
{
	const float t = (float)(125u * (unsigned)(m & 255u)) * (1.0f / 64.0f);      /* < 499: exact */
	return fmaf(-4.0f, rintf(0.25f * t), t);
}
OrChan *or_chan_new_odd(void)
{
	OrChan *c = or_chan_new();
	c->odd = 1;
	or_chan_twist(c->wtw);
	for (int i = 0; i < OR_CH_L; i++) if ((i / OR_CH_M) & 1) c->h[i] = -c->h[i];
	return c;
}

OrChan *or_chan_new(void)
{
	OrChan *c = calloc(1, sizeof(*c));
	or_chan_proto(c->h);
	or_chan_twiddles(c->tw);
	or_chan_resamp_taps(c->g);
	c->hist = calloc(2 * (OR_CH_L - OR_CH_D), sizeof(float));
	return c;
}
void or_chan_free(OrChan *c) { if (c) { free(c->hist); free(c); } }

/* One block: n_steps*500 wideband samples in, per bin n_steps phase samples at 20 kS/s (bins: [512][n_steps]) and,
 * if out48 != NULL, n_steps*12/5 real samples at 48 kS/s ([512][n_steps*12/5]); n_steps % 5 == 0. */
void or_chan_block(OrChan *c, const float *iq, size_t n_steps, float *bins, float *out48)
{
	or_chan_block2(c, iq, n_steps, bins, out48, NULL, NULL);
}

/* The same, plus (decs != NULL) the decimated rows of SPEC 3.5b: bin k with decs[k] = 4 gets n_steps*12/5/4 samples
 * at outdec[k * (n_steps*12/5/2)] (row stride as in round 3); decs[k] = 0 skips the bin.
 * bins: the per-bin PHASE samples, [512][n_steps] floats (quadrants). */
void or_chan_block2(OrChan *c, const float *iq, size_t n_steps, float *bins, float *out48, const uint8_t *decs, float *outdec)
{
	const size_t H = OR_CH_L - OR_CH_D, N = n_steps * OR_CH_D;
	float *buf = malloc(2 * (H + N) * sizeof(float));
	memcpy(buf, c->hist, 2 * H * sizeof(float));
	memcpy(buf + 2 * H, iq, 2 * N * sizeof(float));
	float re[OR_CH_M], im[OR_CH_M];
	float *bl = bins ? bins : malloc((size_t)OR_CH_M * n_steps * sizeof(float));
	for (size_t m = 0; m < n_steps; m++) {
		const float *x = buf + 2 * m * OR_CH_D;
		const int shift = (int)((m * OR_CH_D) % OR_CH_M);
		for (int r = 0; r < OR_CH_M; r++) {
			float ar = 0.0f, ai = 0.0f;
			for (int t = 0; t < OR_CH_T; t++) {
				const int i = r + t * OR_CH_M;
				ar = fmaf(c->h[i], x[2 * i], ar);
				ai = fmaf(c->h[i], x[2 * i + 1], ai);
			}
			if (c->odd) {
				const float wr = c->wtw[2 * r], wi = c->wtw[2 * r + 1];
				const float tr = fmaf(-ai, wi, ar * wr), ti = fmaf(ar, wi, ai * wr);
				ar = tr; ai = ti;
			}
			re[(r + shift) & (OR_CH_M - 1)] = ar;
			im[(r + shift) & (OR_CH_M - 1)] = ai;
		}
		or_fft512(re, im, c->tw);
		/* SPEC 3.5 (round 5): the phase leaves the bank as a 16-bit fraction of a turn, q = rint(16384 atan2q) mod 2^16 (int16: a full
		 * turn is 65536, so every later phase operation -- the odd bank's ramp, the discriminator's wrapped difference -- is EXACT
		 * 16-bit integer arithmetic); 2 bytes per bin and step in HBM instead of 4.  Resolution 9.6e-5 rad: a twentieth of atan2q's
		 * own error.  The odd bank takes the step's common phase -(125 / 64) m quadrants = -32000 m (mod 2^16) off in integers. */
		for (int k = 0; k < OR_CH_M; k++) {
			uint16_t q = (uint16_t)((long)lrintf(or_atan2(im[k], re[k]) * 16384.0f) & 0xFFFF);
			if (c->odd) q = (uint16_t)(q - (uint16_t)(32000u * (unsigned)(m & 0xFFFFu)));
			bl[(size_t)k * n_steps + m] = (float)(int16_t)q * (1.0f / 16384.0f);
		}
	}
	memcpy(c->hist, buf + 2 * N, 2 * H * sizeof(float));   /* last L-D samples of [hist|block] */
	free(buf);
	if (out48 || (decs && outdec)) {
		const size_t n_out = n_steps * OR_RS_L / OR_RS_M;
		float *d = malloc((OR_RS_T + n_steps) * sizeof(float));
		float G4[3 * OR_RS_KT_LD];
		or_chan_composite_taps(c->g, 4, G4);
		for (int k = 0; k < OR_CH_M; k++) {
			memcpy(d, c->dhist[k], OR_RS_T * sizeof(float));
			/* discriminator: the wrapped difference of consecutive phases as a 16-bit integer subtraction, quadrants in [-2, 2) */
			float prev = c->phi_last[k];
			for (size_t m = 0; m < n_steps; m++) {
				const float ph = bl[(size_t)k * n_steps + m];
				/* (both are multiples of 2^-14 in [-2, 2): back to the 16-bit integers, exactly) */
				const uint16_t dq = (uint16_t)((uint16_t)(int16_t)(ph * 16384.0f) - (uint16_t)(int16_t)(prev * 16384.0f));
				d[OR_RS_T + m] = (float)(int16_t)dq * (1.0f / 16384.0f);
				prev = ph;
			}
			c->phi_last[k] = prev;
			for (size_t j = 0; out48 && j < n_out; j++) {
				const size_t i0 = (j * OR_RS_M) / OR_RS_L;
				const int p = (int)((j * OR_RS_M) % OR_RS_L);
				float acc = 0.0f;
				for (int t = 0; t < OR_RS_T; t++) acc = fmaf(c->g[p * OR_RS_T + t], d[OR_RS_T + i0 - t], acc);
				out48[(size_t)k * n_out + j] = acc;
			}
			if (decs && outdec && decs[k] == 4) {
				const size_t dec = 4;
				const int kt = or_chan_composite_kt((int)dec);
				for (size_t n = 0; n < n_out / dec; n++) {
					const size_t b = (OR_RS_M * (dec * n + dec - 1)) / OR_RS_L;
					const float *row = G4 + (n % 3) * OR_RS_KT_LD;
					float acc = 0.0f;
					for (int t = 0; t < kt; t++) acc = fmaf(row[t], d[OR_RS_T + b - t], acc);      /* b - t >= -16: the carried history */
					outdec[(size_t)k * (n_out / 2) + n] = acc;
				}
			}
			memcpy(c->dhist[k], d + n_steps, OR_RS_T * sizeof(float));
		}
		free(d);
	}
	if (!bins) free(bl);
}
