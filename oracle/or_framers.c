/*
 * or_framers.c -- oracle stage 3 for the Manchester/biphase sondes: DFM06/09/17 (Hamming(8,4)),
 * M10 (16-bit checksum, no FEC), iMS-100 / RS-11G (BCH(63,51), t = 2).
 * TEST INFRASTRUCTURE ONLY (see sonde_oracle.h).  PARITY UNPINNED.
 *
 * These stand where sondedump's per-sonde framers sit behind dfm09_decode / m10_decode /
 * ims100_decode (/root/reference/src/main.hpp:37-39, src/decode/decoder.hpp:8,10,11,61).  The
 * protocol constants are public facts restated in SURVEY.md Appendix B.3-B.5, which marks them
 * [RECALL]: the generator (sdrpp_radiosonde_amd/synth.py) and these decoders share the tables, so
 * GPU-vs-oracle parity holds regardless; fidelity on real captures is unverified.
 *
 * The demodulator delivers on-air *chips* (DFM 5000/s, M10 9600/s, iMS-100 4800/s); two chips make
 * one data bit.
 */
#include <stdlib.h>
#include <string.h>
#include "sonde_oracle.h"

/* shared with or_fec.c */
typedef struct {
	int type;
	uint32_t channel;
	uint64_t rpos;
	int collecting;
	uint64_t fstart;
	int inv;
	OrFrame *frames;
	size_t nframes, cap;
} OrFramerPub;

static OrFrame *push_frame(OrFramerPub *f)
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

/* ------------------------------------------------------------------ generic fixed-length sync search
 * sync: `slen` chips, first chip first.  A window matches with Hamming distance <= thr (normal) or
 * >= slen - thr (inverted).  Returns 1 when a complete frame of `flen_chips` chips (sync included)
 * starts at f->fstart; the caller decodes it and continues at fstart + flen_chips. */
static int next_fixed(OrFramerPub *f, const uint8_t *bits, uint64_t wpos, const uint8_t *sync, int slen, int thr,
                      int allow_inv, uint64_t flen_chips)
{
	if (!f->collecting) {
		while (f->rpos + (uint64_t)slen <= wpos) {
			int hd = 0;
			for (int i = 0; i < slen; i++) hd += bits[f->rpos + i] ^ sync[i];
			if (hd <= thr || (allow_inv && hd >= slen - thr)) {
				f->fstart = f->rpos;
				f->inv = hd > thr;
				f->collecting = 1;
				break;
			}
			f->rpos++;
		}
		if (!f->collecting) return 0;
	}
	if (wpos < f->fstart + flen_chips) return 0;
	return 1;
}

static void done_fixed(OrFramerPub *f, uint64_t flen_chips)
{
	f->rpos = f->fstart + flen_chips;
	f->collecting = 0;
}

/* ------------------------------------------------------------------ DFM06/09/17 */
#define DFM_SYNC16      0x45CFu
#define DFM_FRAME_CHIPS 560          /* 280 bits: 16 sync + 56 conf + 104 + 104 */
#define DFM_SYNC_THR    3
#define DFM_NCW         33           /* 7 + 13 + 13 Hamming(8,4) codewords */

/* parity-check rows 01111000 / 10110100 / 11010010 / 11100001 over code bits c0..c7 (c0 first on air
 * inside a de-interleaved codeword); data = c0..c3 */
static const uint8_t dfm_H[4] = { 0x78, 0xB4, 0xD2, 0xE1 };   /* bit 7 = c0 */

static int parity8(unsigned v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return (int)(v & 1); }

/* returns corrected codeword; *st: 0 clean, 1 one bit corrected, -1 uncorrectable */
static uint8_t dfm_hamming_fix(uint8_t cw, int *st)
{
	unsigned syn = 0;
	for (int r = 0; r < 4; r++) syn |= (unsigned)parity8(cw & dfm_H[r]) << (3 - r);   /* syn bit3 = row 0 */
	*st = 0;
	if (!syn) return cw;
	for (int i = 0; i < 8; i++) {
		unsigned col = 0;
		for (int r = 0; r < 4; r++) col |= ((dfm_H[r] >> (7 - i)) & 1u) << (3 - r);
		if (col == syn) { *st = 1; return (uint8_t)(cw ^ (0x80u >> i)); }
	}
	*st = -1;
	return cw;
}

static int dfm09_run(OrFramerPub *f, const uint8_t *bits, uint64_t wpos)
{
	uint8_t sync[32];
	for (int i = 0; i < 16; i++) {
		const int b = (DFM_SYNC16 >> (15 - i)) & 1;
		sync[2 * i] = (uint8_t)b;
		sync[2 * i + 1] = (uint8_t)!b;          /* Manchester: 1 -> 10, 0 -> 01 */
	}
	int produced = 0;
	while (next_fixed(f, bits, wpos, sync, 32, DFM_SYNC_THR, 1, DFM_FRAME_CHIPS)) {
		OrFrame *fr = push_frame(f);
		fr->len = DFM_NCW;
		fr->flags = f->inv ? 1u : 0u;
		fr->bitpos = f->fstart;
		uint8_t db[264];
		int viol = 0;
		for (int k = 0; k < 264; k++) {
			const int a = bits[f->fstart + 32 + 2 * k] ^ f->inv, b = bits[f->fstart + 33 + 2 * k] ^ f->inv;
			db[k] = (uint8_t)a;
			viol += (a == b);
		}
		static const int blk_off[3] = { 0, 56, 160 }, blk_n[3] = { 7, 13, 13 };
		int ncorr = 0, nbad = 0, o = 0;
		for (int b = 0; b < 3; b++) {
			for (int i = 0; i < blk_n[b]; i++) {
				uint8_t cw = 0;
				for (int j = 0; j < 8; j++) cw |= (uint8_t)(db[blk_off[b] + j * blk_n[b] + i] << (7 - j));
				int st;
				cw = dfm_hamming_fix(cw, &st);
				ncorr += st > 0;
				nbad += st < 0;
				fr->data[o++] = cw;
			}
		}
		(void)viol;
		fr->nerr[0] = ncorr;
		fr->nerr[1] = nbad;
		produced++;
		done_fixed(f, DFM_FRAME_CHIPS);
	}
	return produced;
}

/* ------------------------------------------------------------------ M10 */
static const char m10_sync_str[] = "10011001100110010100110010011001";
#define M10_FRAME_BYTES 101
#define M10_FRAME_CHIPS (32 + 16 * M10_FRAME_BYTES)
#define M10_SYNC_THR    3

/* Meteomodem's 16-bit rolling checksum, one byte per step (public M10 decoders) */
static unsigned m10_check_step(unsigned c, unsigned b)
{
	const unsigned c1 = c & 0xFF;
	b = ((b >> 1) | ((b & 1) << 7)) & 0xFF;
	b ^= (b >> 2) & 0xFF;
	const unsigned t6 = (c & 1) ^ ((c >> 2) & 1) ^ ((c >> 4) & 1);
	const unsigned t7 = ((c >> 1) & 1) ^ ((c >> 3) & 1) ^ ((c >> 5) & 1);
	const unsigned t = (c & 0x3F) | (t6 << 6) | (t7 << 7);
	unsigned s = (c >> 7) & 0xFF;
	s ^= (s >> 2) & 0xFF;
	const unsigned c0 = b ^ t ^ s;
	return ((c1 << 8) | c0) & 0xFFFF;
}

uint16_t or_m10_checksum(const uint8_t *p, size_t n)
{
	unsigned cs = 0;
	for (size_t i = 0; i < n; i++) cs = m10_check_step(cs, p[i]);
	return (uint16_t)cs;
}

static int m10_run(OrFramerPub *f, const uint8_t *bits, uint64_t wpos)
{
	uint8_t sync[32];
	for (int i = 0; i < 32; i++) sync[i] = (uint8_t)(m10_sync_str[i] - '0');
	int produced = 0;
	while (next_fixed(f, bits, wpos, sync, 32, M10_SYNC_THR, 1, M10_FRAME_CHIPS)) {
		OrFrame *fr = push_frame(f);
		fr->flags = f->inv ? 1u : 0u;
		fr->bitpos = f->fstart;
		/* the first byte is the length of what follows: 0x64 = M10 (101 bytes in all), 0x45 = M20 (70); the window
		 * behind the sync is always 101 bytes long, whatever lies behind an M20 frame is dropped */
		int viol = 0, total = M10_FRAME_BYTES;
		for (int i = 0; i < total; i++) {
			uint8_t v = 0;
			int vi = 0;
			for (int k = 0; k < 8; k++) {
				const uint64_t p = f->fstart + 32 + 16 * (uint64_t)i + 2 * (uint64_t)k;
				const int a = bits[p] ^ f->inv, b = bits[p + 1] ^ f->inv;
				v = (uint8_t)((v << 1) | a);                 /* 10 -> 1, 01 -> 0, MSB first */
				vi += (a == b);
			}
			fr->data[i] = v;
			viol += vi;
			if (i == 0 && v == 0x45) total = 70;
		}
		fr->len = total;
		const unsigned cs = or_m10_checksum(fr->data, (size_t)total - 2);
		fr->nerr[0] = (cs == (((unsigned)fr->data[total - 2] << 8) | fr->data[total - 1])) ? 0 : -1;
		fr->nerr[1] = viol;
		produced++;
		done_fixed(f, M10_FRAME_CHIPS);
	}
	return produced;
}

/* ------------------------------------------------------------------ iMS-100 / RS-11G */
#define IMS_SYNC24       0x049DCEu
#define IMS_NBLK         12
#define IMS_BLK_BITS     46           /* shortened BCH(63,51): 34 data + 12 parity */
#define IMS_FRAME_BITS   (24 + IMS_NBLK * IMS_BLK_BITS)
#define IMS_FRAME_CHIPS  (2 * IMS_FRAME_BITS)
#define IMS_SYNC_THR     2
#define IMS_DATA_BYTES   51           /* 12 * 34 = 408 bits */
#define BCH_G            0x1539u      /* x^12+x^10+x^8+x^5+x^4+x^3+1 */

static uint8_t g64_exp[128], g64_log[64];
static int g64_ready;
static void g64_init(void)
{
	if (g64_ready) return;
	int x = 1;
	for (int i = 0; i < 63; i++) {
		g64_exp[i] = (uint8_t)x;
		g64_log[x] = (uint8_t)i;
		x <<= 1;
		if (x & 0x40) x ^= 0x43;       /* x^6 + x + 1 */
	}
	for (int i = 63; i < 128; i++) g64_exp[i] = g64_exp[i - 63];
	g64_ready = 1;
}
static inline unsigned g64_mul(unsigned a, unsigned b) { return (a && b) ? g64_exp[g64_log[a] + g64_log[b]] : 0; }

/* 12 parity bits of 34 data bits (d[0] = coefficient of x^45) */
uint32_t or_bch_parity(uint64_t data34)
{
	uint64_t r = data34 << 12;
	for (int i = 45; i >= 12; i--)
		if (r & (1ull << i)) r ^= (uint64_t)BCH_G << (i - 12);
	return (uint32_t)(r & 0xFFF);
}

/* block: 46 bits, bit 45 = first on air.  Returns corrected block; *st = errors corrected (0..2) or -1. */
uint64_t or_bch_decode(uint64_t blk, int *st)
{
	g64_init();
	unsigned s1 = 0, s3 = 0;
	for (int i = 0; i < IMS_BLK_BITS; i++) {
		if (blk & (1ull << i)) {
			s1 ^= g64_exp[i % 63];
			s3 ^= g64_exp[(3 * i) % 63];
		}
	}
	*st = 0;
	if (!s1 && !s3) return blk;
	if (s1) {
		const unsigned s1c = g64_mul(g64_mul(s1, s1), s1);
		if (s3 == s1c) {                          /* single error at log(S1) */
			const int p = g64_log[s1];
			if (p >= IMS_BLK_BITS) { *st = -1; return blk; }
			*st = 1;
			return blk ^ (1ull << p);
		}
		/* two errors: X1 + X2 = S1, X1 X2 = (S3 + S1^3) / S1 ; search the roots among valid positions */
		const unsigned prod = g64_mul(s3 ^ s1c, g64_exp[63 - g64_log[s1]]);
		int found = 0, pos[2] = { 0, 0 };
		for (int i = 0; i < 63; i++) {
			const unsigned X = g64_exp[i];
			if ((g64_mul(X, X) ^ g64_mul(s1, X) ^ prod) == 0) {
				if (found < 2) pos[found] = i;
				found++;
			}
		}
		if (found == 2 && pos[0] < IMS_BLK_BITS && pos[1] < IMS_BLK_BITS) {
			*st = 2;
			return blk ^ (1ull << pos[0]) ^ (1ull << pos[1]);
		}
	}
	*st = -1;
	return blk;
}

static int ims100_run(OrFramerPub *f, const uint8_t *bits, uint64_t wpos)
{
	int produced = 0;
	for (;;) {
		if (!f->collecting) {
			/* biphase-S: data bit = 1 when the two chips of a bit cell are equal (polarity-free) */
			while (f->rpos + 48 <= wpos) {
				int hd = 0;
				for (int k = 0; k < 24; k++) {
					const int b = bits[f->rpos + 2 * k] == bits[f->rpos + 2 * k + 1];
					hd += b ^ (int)((IMS_SYNC24 >> (23 - k)) & 1);
				}
				if (hd <= IMS_SYNC_THR) {
					f->fstart = f->rpos;
					f->inv = 0;
					f->collecting = 1;
					break;
				}
				f->rpos++;
			}
			if (!f->collecting) return produced;
		}
		if (wpos < f->fstart + IMS_FRAME_CHIPS) return produced;
		OrFrame *fr = push_frame(f);
		fr->len = IMS_DATA_BYTES;
		fr->bitpos = f->fstart;
		int ncorr = 0, nbad = 0, ob = 0;
		for (int b = 0; b < IMS_NBLK; b++) {
			uint64_t blk = 0;
			for (int k = 0; k < IMS_BLK_BITS; k++) {
				const uint64_t p = f->fstart + 48 + 2 * ((uint64_t)b * IMS_BLK_BITS + k);
				blk = (blk << 1) | (uint64_t)(bits[p] == bits[p + 1]);
			}
			int st;
			blk = or_bch_decode(blk, &st);
			if (st > 0) ncorr += st;
			if (st < 0) nbad++;
			for (int k = 0; k < 34; k++, ob++)
				fr->data[ob >> 3] |= (uint8_t)(((blk >> (45 - k)) & 1) << (7 - (ob & 7)));
		}
		fr->nerr[0] = ncorr;
		fr->nerr[1] = nbad;
		produced++;
		done_fixed(f, IMS_FRAME_CHIPS);
	}
}

/* ------------------------------------------------------------------ iMet-1 / iMet-4 (SPEC 3.3c)
 * Asynchronous characters at 1200 Bd: start bit 0, eight data bits LSB first, stop bit 1; idle = mark = 1.
 * A packet starts with 0x01, then the type: 1 PTU (14 bytes), 2 GPS (18), 3 XDATA (5 + byte 2), 4 PTUX (20);
 * the last two bytes are CRC16-CCITT (init 0x1D0F) big-endian over the rest.  [RECALL: public iMet notes.]
 * Sync = the 12 bits  1 | 0 1000 0000 1 | 0  (stop/idle, the character 0x01, the next start bit), exact match in
 * either polarity (the slicer's polarity depends on which side of the 1700 Hz mixer the mark tone falls).
 * A candidate is dropped (search resumes one bit later) if the type is unknown or any character of the packet
 * has a wrong start/stop bit; it waits if the packet is not complete yet.  CRC failures are recorded, not dropped. */
uint16_t or_imet_crc(const uint8_t *p, size_t n)
{
	uint16_t crc = 0x1D0F;
	for (size_t i = 0; i < n; i++) {
		crc ^= (uint16_t)((uint16_t)p[i] << 8);
		for (int k = 0; k < 8; k++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
	}
	return crc;
}

static const uint8_t imet_sync[12] = { 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0 };

static inline int imet_char(const uint8_t *bits, uint64_t at, int inv, uint8_t *out)
{
	uint8_t v = 0;
	for (int m = 0; m < 8; m++) v |= (uint8_t)((bits[at + 1 + m] ^ inv) << m);
	*out = v;
	return (bits[at] ^ inv) == 0 && (bits[at + 9] ^ inv) == 1;
}

static int imet4_run(OrFramerPub *f, const uint8_t *bits, uint64_t wpos)
{
	int added = 0;
	while (f->rpos + 12 <= wpos) {
		int hd = 0;
		for (int i = 0; i < 12; i++) hd += bits[f->rpos + i] ^ imet_sync[i];
		if (hd != 0 && hd != 12) { f->rpos++; continue; }
		const int inv = hd == 12;
		const uint64_t c0 = f->rpos + 1;                 /* start bit of the 0x01 character */
		if (c0 + 30 > wpos) break;                       /* need the type and (for XDATA) the length byte */
		uint8_t type, lenb;
		int ok = imet_char(bits, c0 + 10, inv, &type);
		ok &= imet_char(bits, c0 + 20, inv, &lenb);
		int len = 0;
		if (type == 1) len = 14;
		else if (type == 2) len = 18;
		else if (type == 3) len = 5 + lenb;
		else if (type == 4) len = 20;
		if (!ok || len == 0 || len > 64) { f->rpos++; continue; }
		if (c0 + 10 * (uint64_t)len > wpos) break;       /* wait for the rest */
		uint8_t pkt[64];
		for (int j = 0; j < len; j++) ok &= imet_char(bits, c0 + 10 * (uint64_t)j, inv, &pkt[j]);
		if (!ok) { f->rpos++; continue; }
		OrFrame *fr = push_frame(f);
		fr->len = len;
		memcpy(fr->data, pkt, (size_t)len);
		const uint16_t crc = or_imet_crc(pkt, (size_t)len - 2);
		fr->nerr[0] = (crc == (uint16_t)((pkt[len - 2] << 8) | pkt[len - 1])) ? 0 : -1;
		fr->nerr[1] = 0;
		fr->flags = inv ? 1u : 0u;
		fr->bitpos = c0;
		f->rpos = c0 + 10 * (uint64_t)len - 1;          /* the last stop bit may open the next sync */
		added++;
	}
	return added;
}

/* ------------------------------------------------------------------ MRZ-N1 (SPEC 3.3d; [RECALL]: 2400 bit/s GFSK,
 * Manchester, CRC16 reflected 0xA001 as the public MP3-H1 / MRZ decoders describe; header and field offsets: this repo's).
 * Frame: preamble/header AA BF 35 (24 bits, the sync window), 45 payload bytes MSB first, the last two = CRC16-MODBUS
 * (init 0xFFFF, little-endian) of the first 43. */
#define MRZ_FRAME_BYTES 45
#define MRZ_SYNC_CHIPS  48
#define MRZ_FRAME_CHIPS (MRZ_SYNC_CHIPS + 16 * MRZ_FRAME_BYTES)
#define MRZ_SYNC_THR    4

uint16_t or_crc16_modbus(const uint8_t *p, size_t n)
{
	uint16_t crc = 0xFFFF;
	for (size_t i = 0; i < n; i++) {
		crc ^= p[i];
		for (int k = 0; k < 8; k++) crc = (crc & 1) ? (uint16_t)((crc >> 1) ^ 0xA001) : (uint16_t)(crc >> 1);
	}
	return crc;
}

static int mrzn1_run(OrFramerPub *f, const uint8_t *bits, uint64_t wpos)
{
	static const uint8_t hdr[3] = { 0xAA, 0xBF, 0x35 };
	uint8_t sync[MRZ_SYNC_CHIPS];
	for (int i = 0; i < 24; i++) {
		const int b = (hdr[i >> 3] >> (7 - (i & 7))) & 1;
		sync[2 * i] = (uint8_t)b;
		sync[2 * i + 1] = (uint8_t)!b;          /* Manchester: 1 -> 10, 0 -> 01 */
	}
	int produced = 0;
	while (next_fixed(f, bits, wpos, sync, MRZ_SYNC_CHIPS, MRZ_SYNC_THR, 1, MRZ_FRAME_CHIPS)) {
		OrFrame *fr = push_frame(f);
		fr->len = MRZ_FRAME_BYTES;
		fr->flags = f->inv ? 1u : 0u;
		fr->bitpos = f->fstart;
		int viol = 0;
		for (int i = 0; i < MRZ_FRAME_BYTES; i++) {
			uint8_t v = 0;
			for (int k = 0; k < 8; k++) {
				const uint64_t p = f->fstart + MRZ_SYNC_CHIPS + 16 * (uint64_t)i + 2 * (uint64_t)k;
				const int a = bits[p] ^ f->inv, b = bits[p + 1] ^ f->inv;
				v = (uint8_t)((v << 1) | a);
				viol += (a == b);
			}
			fr->data[i] = v;
		}
		const unsigned crc = or_crc16_modbus(fr->data, MRZ_FRAME_BYTES - 2);
		fr->nerr[0] = (crc == ((unsigned)fr->data[43] | ((unsigned)fr->data[44] << 8))) ? 0 : -1;
		fr->nerr[1] = viol;
		produced++;
		done_fixed(f, MRZ_FRAME_CHIPS);
	}
	return produced;
}

/* ------------------------------------------------------------------ SRS-C50 (SPEC 3.3e)
 * 8N1 characters at 2400 Bd, LSB first, idle = mark = 1.  Packet: 00 FF <type> <4 value bytes, big-endian> <c1> <c2>,
 * c1 = sum of type and value bytes mod 256, c2 = sum of the running c1 values mod 256.  [RECALL: public C34/C50 notes.]
 * Sync = the 21 bits  1 | 0 00000000 1 | 0 11111111 1, exact match in either polarity; candidates with a wrong start or
 * stop bit are dropped (search resumes one bit later); checksum failures are recorded, not dropped. */
#define C50_LEN 9
static int c50_run(OrFramerPub *f, const uint8_t *bits, uint64_t wpos)
{
	static const uint8_t sync[21] = { 1, 0, 0,0,0,0,0,0,0,0, 1, 0, 1,1,1,1,1,1,1,1, 1 };
	int added = 0;
	while (f->rpos + 21 <= wpos) {
		int hd = 0;
		for (int i = 0; i < 21; i++) hd += bits[f->rpos + i] ^ sync[i];
		if (hd != 0 && hd != 21) { f->rpos++; continue; }
		const int inv = hd == 21;
		const uint64_t c0 = f->rpos + 1;                 /* start bit of the 00 character */
		if (c0 + 10 * C50_LEN > wpos) break;             /* wait for the rest */
		uint8_t pkt[C50_LEN];
		int ok = 1;
		for (int j = 0; j < C50_LEN; j++) ok &= imet_char(bits, c0 + 10 * (uint64_t)j, inv, &pkt[j]);
		if (!ok) { f->rpos++; continue; }
		OrFrame *fr = push_frame(f);
		fr->len = C50_LEN;
		memcpy(fr->data, pkt, C50_LEN);
		unsigned c1 = 0, c2 = 0;
		for (int j = 2; j < 7; j++) { c1 = (c1 + pkt[j]) & 0xFF; c2 = (c2 + c1) & 0xFF; }
		fr->nerr[0] = (c1 == pkt[7] && c2 == pkt[8]) ? 0 : -1;
		fr->nerr[1] = 0;
		fr->flags = inv ? 1u : 0u;
		fr->bitpos = c0;
		f->rpos = c0 + 10 * (uint64_t)C50_LEN - 1;      /* the last stop bit may open the next sync */
		added++;
	}
	return added;
}

int or_framer_run_other(void *fp, const uint8_t *bits, uint64_t wpos)
{
	OrFramerPub *f = fp;
	switch (f->type) {
	case OR_DFM09:  return dfm09_run(f, bits, wpos);
	case OR_M10:    return m10_run(f, bits, wpos);
	case OR_IMS100: return ims100_run(f, bits, wpos);
	case OR_IMET4:  return imet4_run(f, bits, wpos);
	case OR_MRZN1:  return mrzn1_run(f, bits, wpos);
	case OR_C50:    return c50_run(f, bits, wpos);
	default: return 0;
	}
}
