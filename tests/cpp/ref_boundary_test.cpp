// Boundary proof against the reference's OWN header: /root/reference/src/decode/decoder.hpp (+ common.hpp) is compiled
// unmodified, with include/compat/ standing where the absent sondedump headers were (decoder.hpp:6-14) and the
// test-only tests/cpp/refstub/dsp/block.h standing for SDR++'s; the seven instantiations below are those of
// /root/reference/src/main.hpp:36-42 and bind to the B1 triples libsonde_mi355.so exports.
//   ref_boundary_test link                 -- construct the seven blocks (no GPU needed): proves compile + link
//   ref_boundary_test pump <f32 file> <n>  -- GPU: pump an RS41 discriminator stream through the REFERENCE's run()
//                                             and through sonde::Decoder<> (include/sonde_decoder.hpp); print both
//   ref_boundary_test physics              -- no GPU: the header's OWN static dewpt() / altitude_to_pressure()
//                                             (decoder.hpp:132-174) over the ISA layer edges +- 1 ulp, 400 pseudo-random
//                                             altitudes and a T / RH grid, printed as hex floats: the generator of
//                                             tests/golden/physics_golden.json (tests/golden/make_physics_golden.py)
#include <dsp/block.h>
#include "decode/decoder.hpp"
#include <cmath>
#include <cstdio>
#include <vector>
#include "sonde_decoder.hpp"

// explicit instantiation: every member (init, deinit, run, destructor) of all seven is compiled and must link
template class radiosonde::Decoder<RS41Decoder, rs41_decoder_init, rs41_decoder_deinit, rs41_decode>;
template class radiosonde::Decoder<DFM09Decoder, dfm09_decoder_init, dfm09_decoder_deinit, dfm09_decode>;
template class radiosonde::Decoder<IMS100Decoder, ims100_decoder_init, ims100_decoder_deinit, ims100_decode>;
template class radiosonde::Decoder<M10Decoder, m10_decoder_init, m10_decoder_deinit, m10_decode>;
template class radiosonde::Decoder<IMET4Decoder, imet4_decoder_init, imet4_decoder_deinit, imet4_decode>;
template class radiosonde::Decoder<C50Decoder, c50_decoder_init, c50_decoder_deinit, c50_decode>;
template class radiosonde::Decoder<MRZN1Decoder, mrzn1_decoder_init, mrzn1_decoder_deinit, mrzn1_decode>;

static radiosonde::Decoder<RS41Decoder, rs41_decoder_init, rs41_decoder_deinit, rs41_decode> rs41decoder;
static radiosonde::Decoder<DFM09Decoder, dfm09_decoder_init, dfm09_decoder_deinit, dfm09_decode> dfm09decoder;
static radiosonde::Decoder<IMS100Decoder, ims100_decoder_init, ims100_decoder_deinit, ims100_decode> ims100decoder;
static radiosonde::Decoder<M10Decoder, m10_decoder_init, m10_decoder_deinit, m10_decode> m10decoder;
static radiosonde::Decoder<IMET4Decoder, imet4_decoder_init, imet4_decoder_deinit, imet4_decode> imet4decoder;
static radiosonde::Decoder<C50Decoder, c50_decoder_init, c50_decoder_deinit, c50_decode> c50decoder;
static radiosonde::Decoder<MRZN1Decoder, mrzn1_decoder_init, mrzn1_decoder_deinit, mrzn1_decode> mrzn1decoder;

template <class D> static void line(const char *tag, const D *d)
{
	printf("%s seq=%d serial=%s lat=%a lon=%a alt=%a spd=%a hdg=%a climb=%a time=%ld temp=%a rh=%a dewpt=%a pressure=%a cal=%d kill=%d aux=%s\n",
	       tag, d->seq, d->serial.c_str(), d->lat, d->lon, d->alt, d->spd, d->hdg, d->climb, (long)d->time, d->temp, d->rh, d->dewpt, d->pressure,
	       (int)d->calibrated, d->burstkill, d->auxData.c_str());
}
static void ref_cb(SondeFullData *d, void *) { line("REF", d); }
static void own_cb(sonde::FullData *d, void *) { line("OWN", d); }

int main(int argc, char **argv)
{
	if (argc >= 2 && std::string(argv[1]) == "link") {
		dsp::block *all[7] = { &rs41decoder, &dfm09decoder, &ims100decoder, &m10decoder, &imet4decoder, &c50decoder, &mrzn1decoder };
		for (dsp::block *b : all) if (b->_block_init) return 1;
		printf("LINK OK %zu\n", sizeof(all) / sizeof(*all));
		return 0;
	}
	if (argc >= 2 && std::string(argv[1]) == "physics") {
		// dewpt / altitude_to_pressure are statics of the reference's header: this calls the reference's own compiled bodies
		const float edges[] = { 0.0f, 11000.0f, 20000.0f, 32000.0f, 47000.0f, 51000.0f, 77000.0f };
		for (float e : edges)
			for (float a : { std::nextafterf(e, -1e9f), e, std::nextafterf(e, 1e9f) }) printf("ALT %a %a\n", a, altitude_to_pressure(a));
		uint32_t lcg = 0x5EED0003u;
		for (int i = 0; i < 400; i++) {
			lcg = lcg * 1664525u + 1013904223u;
			const float a = -500.0f + 90500.0f * (float)(lcg >> 8) * (1.0f / 16777216.0f);
			printf("ALT %a %a\n", a, altitude_to_pressure(a));
		}
		for (int ti = 0; ti <= 26; ti++)
			for (int ri = 0; ri <= 20; ri++) {
				const float t = -90.0f + 5.0f * (float)ti, rh = ri == 0 ? 0.5f : 5.0f * (float)ri;
				printf("DEW %a %a %a\n", t, rh, dewpt(t, rh));
			}
		return 0;
	}
	if (argc < 4) return 2;
	std::vector<float> x;
	{
		FILE *f = fopen(argv[2], "rb");
		if (!f) return 3;
		float tmp[4096];
		size_t n;
		while ((n = fread(tmp, sizeof(float), 4096, f)) > 0) x.insert(x.end(), tmp, tmp + n);
		fclose(f);
	}
	const size_t chunk = (size_t)atol(argv[3]);
	// ---- the reference's block: init(stream, 48000, cb, ctx) as main.cpp:62 does, then run() per buffer
	dsp::stream<float> in;
	rs41decoder.init(&in, 48000, ref_cb, nullptr);
	for (size_t off = 0; off < x.size(); off += chunk) {
		in.readBuf = x.data() + off;
		in.pending = (int)std::min(chunk, x.size() - off);
		if (rs41decoder.run() != 0) { printf("ERROR run\n"); return 4; }
	}
	if (rs41decoder.run() != -1) printf("ERROR run did not stop on a closed stream\n");     // read() < 0 ends the worker, decoder.hpp:59
	rs41decoder.deinit();
	rs41decoder._block_init = false;
	// ---- this repo's adaptor on a second decoder instance, same samples, same buffer sizes
	sonde::Decoder<RS41Decoder, rs41_decoder_init, rs41_decoder_deinit, rs41_decode> own;
	if (!own.init(48000, own_cb, nullptr)) { printf("ERROR own init\n"); return 5; }
	for (size_t off = 0; off < x.size(); off += chunk) own.process(x.data() + off, std::min(chunk, x.size() - off));
	printf("DONE\n");
	return 0;
}
