// b1.cpp -- the per-sonde C triple the reference binds to (B1 boundary):
//     T* X_decoder_init(int);  void X_decoder_deinit(T*);  ParserStatus X_decode(T*, SondeData*, const float*, size_t);
// fixed by /root/reference/src/decode/decoder.hpp:22 and instantiated at /root/reference/src/main.hpp:36-42.
// Each decoder is a one-channel batch (real 48 kS/s discriminator samples in, decoder.hpp:35) on the GPU.
//
// Re-entrancy contract honoured (decoder.hpp:59-117): the caller re-passes the same
// (src,len) until PROCEED; the first call of a buffer consumes all of it, later calls drain the
// fragment queue one fragment per call, then PROCEED is returned once and the next buffer is accepted.
#include <deque>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "parse.h"
#include "../../include/sonde_abi.h"

struct SondeB1Decoder {
	int type = 0;
	SondeBatch *batch = nullptr;      // null for sonde types that are not implemented (always PROCEED)
	SondeParser parser;
	std::vector<float> pending;       // samples waiting for a full tile (b1_granule)
	std::deque<SondeData> frags;
	std::vector<SondeFrame> frames;
	bool consumed = false;
	bool complained = false;          // a failed submit has been reported once
	explicit SondeB1Decoder(int t) : type(t), parser(t) {}
};

static const uint32_t kB1MaxSamples = 16 * SONDE_TILE;
// submit granule: one tile, or one tile behind the 8:1 tone demodulator for the AFSK sondes
static size_t b1_granule(int type) { return (type == SONDE_IMET4 || type == SONDE_C50) ? 8 * (size_t)SONDE_TILE : (size_t)SONDE_TILE; }

static SondeB1Decoder *b1_init(int type, int samplerate, bool implemented)
{
	if (samplerate != 48000) return nullptr;   // reference always passes OUT_SAMPLE_RATE, main.cpp:16,62-68
	SondeB1Decoder *d = new SondeB1Decoder(type);
	if (implemented) {
		SondeBatchConfig cfg = SONDE_BATCH_CONFIG_INIT;
		const uint8_t t = (uint8_t)type;
		cfg.n_channels = 1;
		cfg.types = &t;
		cfg.max_samples = kB1MaxSamples;
		cfg.input_kind = SONDE_INPUT_REAL;
		// HIP device ordinal: SONDE_B1_DEVICE (the reference's init(int samplerate) has no room for it, decoder.hpp:39)
		const char *dv = getenv("SONDE_B1_DEVICE");
		cfg.device = dv ? atoi(dv) : 0;
		if (sonde_batch_create(&cfg, &d->batch) != 0) { delete d; return nullptr; }
	}
	return d;
}

static void b1_deinit(SondeB1Decoder *d)
{
	if (!d) return;
	sonde_batch_destroy(d->batch);
	delete d;
}

static ParserStatus b1_decode(SondeB1Decoder *d, SondeData *dst, const float *src, size_t len)
{
	if (!d || !d->batch) return PROCEED;
	if (!d->consumed) {
		d->pending.insert(d->pending.end(), src, src + len);
		size_t off = 0;
		const size_t gran = b1_granule(d->type);
		while (d->pending.size() - off >= gran) {
			size_t n = ((d->pending.size() - off) / gran) * gran;
			if (n > kB1MaxSamples) n = kB1MaxSamples;
			if (sonde_batch_submit_host(d->batch, d->pending.data() + off, n, n) != 0) {
				// the GPU refused the block: say so once and drop the samples (keeping them would grow `pending`
				// without bound and answer PROCEED forever with no diagnostic)
				if (!d->complained) { fprintf(stderr, "sonde_mi355: decoder submit failed: %s\n", sonde_last_error()); d->complained = true; }
				off = d->pending.size();
				break;
			}
			const long nf = sonde_batch_sync(d->batch);
			if (nf > 0) {
				d->frames.resize((size_t)nf);
				const long got = sonde_batch_frames(d->batch, d->frames.data(), (size_t)nf);
				std::vector<SondeData> v;
				for (long i = 0; i < got; i++) d->parser.feed(d->frames[(size_t)i], v);
				d->frags.insert(d->frags.end(), v.begin(), v.end());
			}
			off += n;
		}
		d->pending.erase(d->pending.begin(), d->pending.begin() + (long)off);
		d->consumed = true;
	}
	if (!d->frags.empty()) {
		*dst = d->frags.front();
		d->frags.pop_front();
		return PARSED;
	}
	d->consumed = false;
	return PROCEED;
}

#define SONDE_B1_DEF(T, x, TYPE, IMPL) \
	extern "C" T *x##_decoder_init(int samplerate) { return b1_init(TYPE, samplerate, IMPL); } \
	extern "C" void x##_decoder_deinit(T *d) { b1_deinit(d); } \
	extern "C" ParserStatus x##_decode(T *d, SondeData *dst, const float *src, size_t len) { return b1_decode(d, dst, src, len); }

SONDE_B1_DEF(RS41Decoder,   rs41,   SONDE_RS41,   true)
SONDE_B1_DEF(DFM09Decoder,  dfm09,  SONDE_DFM09,  true)
SONDE_B1_DEF(IMS100Decoder, ims100, SONDE_IMS100, true)
SONDE_B1_DEF(M10Decoder,    m10,    SONDE_M10,    true)
SONDE_B1_DEF(IMET4Decoder,  imet4,  SONDE_IMET4,  true)    // Bell-202 AFSK: tone demodulator in front (SPEC 3.6)
SONDE_B1_DEF(C50Decoder,    c50,    SONDE_C50,    true)     // AFSK 2400 Bd: the same tone demodulator, other mixer (SPEC 3.6)
SONDE_B1_DEF(MRZN1Decoder,  mrzn1,  SONDE_MRZN1,  true)
