// iq_stream_test.cpp -- drives sonde::IqStreamDecoder (B3: complex IQ stream in, FullData callback out) on a GPU:
//   iq_stream_test <iq.bin> <sonde_type> <chunk> [rate]     iq.bin: float32 [n][2] at `rate` (48 kS/s, or the reference's VFO
//   rate for the type: 10/15/20/50 kS/s); fed in buffers of <chunk> samples
//   iq_stream_test <a.bin> <typeA> <chunk> <rateA> <b.bin> <typeB> <rateB>     the type switch of the reference's
//   onTypeSelected (/root/reference/src/main.cpp:366-405): the SAME decoder object is re-initialised for another sonde type
//   at another VFO rate; "SWITCH" is printed between the two streams' callbacks
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sonde_decoder.hpp"

static void on_data(sonde::FullData *d, void *ctx)
{
	(*(long *)ctx)++;
	printf("CB seq=%d serial=%s lat=%.5f lon=%.5f alt=%.1f spd=%.2f temp=%.2f rh=%.2f pressure=%a time=%ld\n", d->seq, d->serial.c_str(),
	       (double)d->lat, (double)d->lon, (double)d->alt, (double)d->spd, (double)d->temp, (double)d->rh, (double)d->pressure, (long)d->time);
}

static bool load(const char *path, std::vector<float> &x)
{
	FILE *f = fopen(path, "rb");
	if (!f) return false;
	float tmp[4096];
	size_t n;
	x.clear();
	while ((n = fread(tmp, sizeof(float), 4096, f)) > 0) x.insert(x.end(), tmp, tmp + n);
	fclose(f);
	return true;
}

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	std::vector<float> x;
	if (!load(argv[1], x)) return 3;
	const int chunk = atoi(argv[3]);
	long calls = 0;
	sonde::IqStreamDecoder dec;
	if (dec.init(0, 44100, on_data, &calls)) { printf("ERROR accepted 44100\n"); return 1; }
	const int rate = argc > 4 ? atoi(argv[4]) : 48000;
	if (!dec.init(atoi(argv[2]), rate, on_data, &calls)) { printf("ERROR init: %s\n", sonde_last_error()); return 1; }
	long fired = 0;
	for (size_t off = 0; off < x.size() / 2; off += (size_t)chunk) {
		const int k = dec.process(x.data() + 2 * off, (int)std::min((size_t)chunk, x.size() / 2 - off));
		if (k < 0) { printf("ERROR process: %s\n", sonde_last_error()); return 1; }
		fired += k;
	}
	printf("DONE fired=%ld calls=%ld seq=%d\n", fired, calls, dec.data().seq);
	if (argc > 7) {
		printf("SWITCH\n");
		if (!load(argv[5], x)) return 3;
		if (!dec.init(atoi(argv[6]), atoi(argv[7]), on_data, &calls)) { printf("ERROR re-init: %s\n", sonde_last_error()); return 1; }
		if (dec.data().seq != 0 || !dec.data().serial.empty()) { printf("ERROR aggregate not reset by the type switch\n"); return 1; }
		fired = 0;
		for (size_t off = 0; off < x.size() / 2; off += (size_t)chunk) {
			const int k = dec.process(x.data() + 2 * off, (int)std::min((size_t)chunk, x.size() / 2 - off));
			if (k < 0) { printf("ERROR process: %s\n", sonde_last_error()); return 1; }
			fired += k;
		}
		printf("DONE fired=%ld calls=%ld seq=%d\n", fired, calls, dec.data().seq);
	}
	return 0;
}
