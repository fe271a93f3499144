/*
 * or_fec.c -- oracle stage 3: GF(2^8) Reed-Solomon(255,231), CRC16, RS41 framer.
 * TEST INFRASTRUCTURE ONLY (see sonde_oracle.h).  PARITY UNPINNED.
 *
 * Stands where sondedump's framer/correlator + decode/ecc/rs.c + the RS41
 * subframe walker sit behind rs41_decode (/root/reference/src/main.hpp:36,
 * /root/reference/src/decode/decoder.hpp:13,61).  Protocol constants are public
 * RS41 facts restated in SURVEY.md Appendix B.2 and self-checked by
 * tests/test_oracle_kat.py (header ^ mask KAT, (518-56)/2 = 231, 48 = 2*24).
 */
#include <stdlib.h>
#include <string.h>
#include "sonde_oracle.h"

/* ---------------- GF(2^8), primitive polynomial 0x11D, alpha = 2 ---------------- */
static uint8_t gf_exp[512];
static uint8_t gf_log[256];
static int gf_ready;

void or_gf256_init(void)
{
	if (gf_ready) return;
	int x = 1;
	for (int i = 0; i < 255; i++) {
		gf_exp[i] = (uint8_t)x;
		gf_log[x] = (uint8_t)i;
		x <<= 1;
		if (x & 0x100) x ^= 0x11D;
	}
	for (int i = 255; i < 512; i++) gf_exp[i] = gf_exp[i - 255];
	gf_log[0] = 0;
	gf_ready = 1;
}

uint8_t or_gf256_mul(uint8_t a, uint8_t b)
{
	if (!a || !b) return 0;
	return gf_exp[gf_log[a] + gf_log[b]];
}

static inline uint8_t gf_div(uint8_t a, uint8_t b)   /* b != 0 */
{
	if (!a) return 0;
	return gf_exp[gf_log[a] + 255 - gf_log[b]];
}

#define RS_R 24   /* parity bytes, roots alpha^0 .. alpha^23 */
#define RS_T 12

void or_rs255_encode(uint8_t *cw, int n)
{
	or_gf256_init();
	/* g(x) = prod_{j=0}^{23} (x - alpha^j), little-endian coefficients */
	uint8_t g[RS_R + 1] = {1};
	for (int j = 0; j < RS_R; j++) {
		/* multiply by (x + alpha^j) */
		for (int i = j + 1; i > 0; i--)
			g[i] = g[i - 1] ^ or_gf256_mul(g[i], gf_exp[j]);
		g[0] = or_gf256_mul(g[0], gf_exp[j]);
	}
	/* remainder of msg(x)*x^24 mod g(x): LFSR over message bytes from highest degree down */
	uint8_t rem[RS_R] = {0};
	for (int i = n - 1; i >= RS_R; i--) {
		const uint8_t fb = cw[i] ^ rem[RS_R - 1];
		for (int k = RS_R - 1; k > 0; k--)
			rem[k] = rem[k - 1] ^ or_gf256_mul(fb, g[k]);
		rem[0] = or_gf256_mul(fb, g[0]);
	}
	memcpy(cw, rem, RS_R);
}

int or_rs255_decode(uint8_t *cw, int n)
{
	uint8_t S[RS_R], lam[RS_R + 2] = {1}, B[RS_R + 2] = {1}, T[RS_R + 2], om[RS_R];
	int L = 0, m = 1, nz = 0;
	uint8_t b = 1;

	or_gf256_init();
	for (int j = 0; j < RS_R; j++) {
		uint8_t s = 0;
		for (int i = n - 1; i >= 0; i--)
			s = (uint8_t)((s ? gf_exp[gf_log[s] + j] : 0) ^ cw[i]);
		S[j] = s;
		nz |= s;
	}
	if (!nz) return 0;

	/* Berlekamp-Massey */
	for (int r = 0; r < RS_R; r++) {
		uint8_t delta = S[r];
		for (int i = 1; i <= L; i++) delta ^= or_gf256_mul(lam[i], S[r - i]);
		if (!delta) {
			m++;
		} else {
			const uint8_t f = gf_div(delta, b);
			if (2 * L <= r) {
				memcpy(T, lam, sizeof(T));
				for (int i = 0; i + m < RS_R + 2; i++) lam[i + m] ^= or_gf256_mul(f, B[i]);
				L = r + 1 - L;
				memcpy(B, T, sizeof(B));
				b = delta;
				m = 1;
			} else {
				for (int i = 0; i + m < RS_R + 2; i++) lam[i + m] ^= or_gf256_mul(f, B[i]);
				m++;
			}
		}
	}
	if (L > RS_T) return -1;
	int deg = 0;
	for (int i = 0; i < RS_R + 2; i++) if (lam[i]) deg = i;
	if (deg != L) return -1;

	/* Chien search over all 255 positions */
	int pos[RS_T], npos = 0;
	for (int i = 0; i < 255; i++) {
		uint8_t v = 0;   /* lam(alpha^-i) */
		for (int k = 0; k <= L; k++)
			if (lam[k]) v ^= gf_exp[(gf_log[lam[k]] + (255 - i) * k) % 255];
		if (!v) {
			if (npos == RS_T || npos == L) return -1;
			pos[npos++] = i;
		}
	}
	if (npos != L) return -1;
	for (int k = 0; k < npos; k++) if (pos[k] >= n) return -1;

	/* omega = S*lam mod x^24 */
	for (int i = 0; i < RS_R; i++) {
		uint8_t v = 0;
		for (int k = 0; k <= i && k <= L; k++) v ^= or_gf256_mul(lam[k], S[i - k]);
		om[i] = v;
	}
	/* Forney, first root alpha^0: e = X * omega(X^-1) / lam'(X^-1) */
	uint8_t ev[RS_T];
	for (int e = 0; e < npos; e++) {
		const int xi = (255 - pos[e]) % 255;   /* log of X^-1 */
		uint8_t num = 0, den = 0;
		for (int k = 0; k < RS_R; k++)
			if (om[k]) num ^= gf_exp[(gf_log[om[k]] + xi * k) % 255];
		for (int k = 1; k <= L; k += 2)
			if (lam[k]) den ^= gf_exp[(gf_log[lam[k]] + xi * (k - 1)) % 255];
		if (!den) return -1;
		ev[e] = or_gf256_mul(gf_exp[pos[e]], gf_div(num, den));
	}
	for (int e = 0; e < npos; e++) cw[pos[e]] ^= ev[e];
	return npos;
}

/* CRC16-CCITT (poly 0x1021, init 0xFFFF, MSB first, no reflection, no final xor) */
uint16_t or_crc16_ccitt(const uint8_t *p, size_t n)
{
	uint16_t crc = 0xFFFF;
	for (size_t i = 0; i < n; i++) {
		crc ^= (uint16_t)p[i] << 8;
		for (int k = 0; k < 8; k++)
			crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
	}
	return crc;
}

/* ---------------- RS41 framing ---------------- */
static const uint8_t rs41_header[8] = { 0x10, 0xB6, 0xCA, 0x11, 0x22, 0x96, 0x12, 0xF8 };   /* on-air bytes */
static const uint8_t rs41_mask[64] = {
	0x96, 0x83, 0x3E, 0x51, 0xB1, 0x49, 0x08, 0x98, 0x32, 0x05, 0x59, 0x0E, 0xF9, 0x44, 0xC6, 0x26,
	0x21, 0x60, 0xC2, 0xEA, 0x79, 0x5D, 0x6D, 0xA1, 0x54, 0x69, 0x47, 0x0C, 0xDC, 0xE8, 0x5C, 0xF1,
	0xF7, 0x76, 0x82, 0x7F, 0x07, 0x99, 0xA2, 0x2C, 0x93, 0x7C, 0x30, 0x63, 0xF5, 0x10, 0x2E, 0x61,
	0xD0, 0xBC, 0xB4, 0xB6, 0x06, 0xAA, 0xF4, 0x23, 0x78, 0x6E, 0x3B, 0xAE, 0xBF, 0x7B, 0x4C, 0xC1,
};
#define RS41_SYNC_THR   6
#define RS41_LEN_STD    320
#define RS41_LEN_EXT    518
#define RS41_TYPE_POS   56

const uint8_t *or_demod_bitptr(const OrDemod *d);
int or_framer_run_other(void *f, const uint8_t *bits, uint64_t wpos);

struct OrFramer {
	int type;
	uint32_t channel;
	uint64_t rpos;      /* search resumes here */
	int collecting;
	uint64_t fstart;
	int inv;
	OrFrame *frames;
	size_t nframes, cap;
};

OrFramer *or_framer_new(int type, uint32_t channel)
{
	OrFramer *f = calloc(1, sizeof(*f));
	f->type = type;
	f->channel = channel;
	or_gf256_init();
	return f;
}

void or_framer_free(OrFramer *f) { if (f) { free(f->frames); free(f); } }
size_t or_framer_nframes(const OrFramer *f) { return f->nframes; }
const OrFrame *or_framer_frame(const OrFramer *f, size_t i) { return &f->frames[i]; }

static OrFrame *new_frame(OrFramer *f)
{
	if (f->nframes == f->cap) {
		f->cap = f->cap ? f->cap * 2 : 16;
		f->frames = realloc(f->frames, f->cap * sizeof(OrFrame));
	}
	OrFrame *fr = &f->frames[f->nframes++];
	memset(fr, 0, sizeof(*fr));
	fr->channel = f->channel;
	fr->type = (uint32_t)f->type;
	return fr;
}

static inline uint8_t lsb_byte(const uint8_t *bits, uint64_t at)
{
	uint8_t v = 0;
	for (int m = 0; m < 8; m++) v |= (uint8_t)(bits[at + m] << m);
	return v;
}

static int popcount8(unsigned v) { int c = 0; while (v) { c += v & 1; v >>= 1; } return c; }

static int rs41_run(OrFramer *f, const uint8_t *bits, uint64_t wpos)
{
	int produced = 0;
	for (;;) {
		if (!f->collecting) {
			int found = 0;
			while (f->rpos + 64 <= wpos) {
				int hd = 0;
				for (int i = 0; i < 64; i++)
					hd += bits[f->rpos + i] ^ ((rs41_header[i >> 3] >> (i & 7)) & 1);
				if (hd <= RS41_SYNC_THR || hd >= 64 - RS41_SYNC_THR) {
					f->fstart = f->rpos;
					f->inv = hd >= 64 - RS41_SYNC_THR;
					f->collecting = 1;
					found = 1;
					break;
				}
				f->rpos++;
			}
			if (!found) return produced;
		}
		if (wpos < f->fstart + 8 * (RS41_TYPE_POS + 1)) return produced;
		const uint8_t xinv = f->inv ? 0xFF : 0x00;
		const uint8_t tb = (uint8_t)(lsb_byte(bits, f->fstart + 8 * RS41_TYPE_POS) ^ xinv ^ rs41_mask[RS41_TYPE_POS & 63]);
		const int ext = popcount8(tb ^ 0xF0u) < popcount8(tb ^ 0x0Fu);
		const int flen = ext ? RS41_LEN_EXT : RS41_LEN_STD;
		if (wpos < f->fstart + 8 * (uint64_t)flen) return produced;

		OrFrame *fr = new_frame(f);
		fr->len = flen;
		fr->flags = f->inv ? 1u : 0u;
		fr->bitpos = f->fstart;
		for (int i = 0; i < flen; i++)
			fr->data[i] = (uint8_t)(lsb_byte(bits, f->fstart + 8 * (uint64_t)i) ^ xinv ^ rs41_mask[i & 63]);

		const int msglen = (flen - 56) / 2;
		for (int c = 0; c < 2; c++) {
			uint8_t cw[255];
			memset(cw, 0, sizeof(cw));
			for (int i = 0; i < RS_R; i++) cw[i] = fr->data[8 + RS_R * c + i];
			for (int i = 0; i < msglen; i++) cw[RS_R + i] = fr->data[56 + 2 * i + c];
			const int ne = or_rs255_decode(cw, RS_R + msglen);
			fr->nerr[c] = ne;
			if (ne > 0) {
				for (int i = 0; i < RS_R; i++) fr->data[8 + RS_R * c + i] = cw[i];
				for (int i = 0; i < msglen; i++) fr->data[56 + 2 * i + c] = cw[RS_R + i];
			}
		}
		produced++;
		f->rpos = f->fstart + 8 * (uint64_t)flen;
		f->collecting = 0;
	}
}

int or_framer_run(OrFramer *f, const OrDemod *d)
{
	const uint8_t *bits = or_demod_bitptr(d);
	const uint64_t wpos = or_demod_nbits(d);
	switch (f->type) {
	case OR_RS41: return rs41_run(f, bits, wpos);
	default: return or_framer_run_other(f, bits, wpos);   /* or_framers.c; same struct layout */
	}
}

/* ---------------- whole channel ---------------- */
struct OrChannel { OrDemod *d; OrFramer *f; };

OrChannel *or_channel_new(int type, uint32_t channel)
{
	OrChannel *c = calloc(1, sizeof(*c));
	c->d = or_demod_new(type);
	c->f = or_framer_new(type, channel);
	return c;
}
void or_channel_free(OrChannel *c) { if (c) { or_demod_free(c->d); or_framer_free(c->f); free(c); } }
void or_channel_feed(OrChannel *c, const float *src, size_t n, int is_iq)
{
	or_demod_feed(c->d, src, n, is_iq);
	or_framer_run(c->f, c->d);
}
size_t or_channel_nframes(const OrChannel *c) { return or_framer_nframes(c->f); }
const OrFrame *or_channel_frame(const OrChannel *c, size_t i) { return or_framer_frame(c->f, i); }
OrDemod *or_channel_demod(OrChannel *c) { return c->d; }

size_t or_batch_run(int type, const float *iq, size_t nch, size_t n, int nthreads, OrFrame *out, size_t cap)
{
	size_t total = 0;
	size_t *counts = calloc(nch, sizeof(size_t));
	OrChannel **chs = calloc(nch, sizeof(*chs));
	(void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
	for (long c = 0; c < (long)nch; c++) {
		chs[c] = or_channel_new(type, (uint32_t)c);
		or_channel_feed(chs[c], iq + 2 * (size_t)c * n, n, 1);
		counts[c] = or_channel_nframes(chs[c]);
	}
	for (size_t c = 0; c < nch; c++) {
		for (size_t i = 0; i < counts[c]; i++) {
			if (out && total < cap) out[total] = *or_channel_frame(chs[c], i);
			total++;
		}
		or_channel_free(chs[c]);
	}
	free(chs);
	free(counts);
	return total;
}
