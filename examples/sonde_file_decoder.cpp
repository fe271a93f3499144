// sonde_file_decoder.cpp -- the reference module's whole data flow on a recorded IQ file, through this library:
//
//     VFO IQ stream  ->  [GPU: FM discriminator, (resampler,) timing recovery, framing, FEC]  ->  field parsers + fragment merge
//                    ->  sondeDataHandler: GPX track + PTU CSV                     (/root/reference/src/main.cpp:54-72, 320-331)
//
// usage: sonde_file_decoder <iq.cf32> <sonde_type 0..6> <rate> [out.gpx] [out.csv]
//   iq.cf32   interleaved float32 I/Q, at 48000 S/s or at the reference's VFO rate for the type (10000 RS41, 15000 DFM,
//             20000 iMS-100 / iMet-4 / SRS-C50 / MRZ-N1, 50000 M10/M20; supportedTypes[], main.hpp:44-52)
// Build: g++ -std=c++17 -Iinclude examples/sonde_file_decoder.cpp -Lsdrpp_radiosonde_amd -l:libsonde_mi355.so -Wl,-rpath,$PWD/sdrpp_radiosonde_amd
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sonde_decoder.hpp"
#include "sonde_sinks.hpp"

struct Sinks {
	sonde::GpxWriter gpx;
	sonde::PtuWriter ptu;
	long points = 0;
};

// the body of RadiosondeDecoderModule::sondeDataHandler (main.cpp:320-331), on this library's sinks
static void on_data(sonde::FullData *d, void *ctx)
{
	Sinks *s = (Sinks *)ctx;
	if (d->serial != "") s->gpx.startTrack(d->serial.c_str());
	s->gpx.addTrackPoint(d->time, d->lat, d->lon, d->alt, d->spd, d->hdg);
	s->ptu.addPoint(*d);
	s->points++;
}

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s <iq.cf32> <sonde_type 0..6> <rate> [out.gpx] [out.csv]\n", argv[0]); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	Sinks sinks;
	if (argc > 4 && !sinks.gpx.open(argv[4])) { perror(argv[4]); return 2; }
	if (argc > 5 && !sinks.ptu.open(argv[5])) { perror(argv[5]); return 2; }
	sonde::IqStreamDecoder dec;
	if (!dec.init(atoi(argv[2]), atoi(argv[3]), on_data, &sinks)) {
		fprintf(stderr, "init failed: %s\n", sonde_last_error());
		return 1;
	}
	std::vector<float> buf(2 * 4096);                 // what one dsp::stream<dsp::complex_t>::read() might hand over
	size_t n;
	long samples = 0;
	while ((n = fread(buf.data(), 2 * sizeof(float), 4096, f)) > 0) {
		if (dec.process(buf.data(), (int)n) < 0) { fprintf(stderr, "decode failed: %s\n", sonde_last_error()); return 1; }
		samples += (long)n;
	}
	fclose(f);
	const sonde::FullData &d = dec.data();
	printf("%ld samples, %ld callbacks; last: serial=%s seq=%d lat=%.5f lon=%.5f alt=%.1f temp=%.1f rh=%.1f\n", samples, sinks.points,
	       d.serial.c_str(), d.seq, (double)d.lat, (double)d.lon, (double)d.alt, (double)d.temp, (double)d.rh);
	return 0;
}
