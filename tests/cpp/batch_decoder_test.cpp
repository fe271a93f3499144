// batch_decoder_test.cpp -- drives sonde::BatchDecoder (include/sonde_decoder.hpp) on a GPU:
//   batch_decoder_test <iq.bin> <n_channels> <n_samples> <n_submits>
// iq.bin: float32 [n_channels][n_samples][2].  Prints one line per callback and a summary.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "sonde_decoder.hpp"

static long g_calls = 0;
static void on_data(uint32_t ch, sonde::FullData *d, void *)
{
	g_calls++;
	printf("CB ch=%u seq=%d serial=%s lat=%.5f lon=%.5f alt=%.1f temp=%.2f rh=%.2f dewpt=%a pressure=%a cal=%.1f\n", ch, d->seq,
	       d->serial.c_str(), (double)d->lat, (double)d->lon, (double)d->alt, (double)d->temp, (double)d->rh, (double)d->dewpt,
	       (double)d->pressure, (double)d->calib_percent);
}

int main(int argc, char **argv)
{
	if (argc < 5) return 2;
	const uint32_t C = (uint32_t)atoi(argv[2]);
	const size_t n = (size_t)atol(argv[3]);
	const int parts = atoi(argv[4]);
	std::vector<float> iq((size_t)C * n * 2);
	FILE *f = fopen(argv[1], "rb");
	if (!f || fread(iq.data(), sizeof(float), iq.size(), f) != iq.size()) { printf("ERROR reading input\n"); return 1; }
	fclose(f);
	float *d_iq = nullptr;
	if (hipMalloc((void **)&d_iq, iq.size() * sizeof(float)) != hipSuccess) { printf("ERROR hipMalloc\n"); return 1; }
	hipMemcpy(d_iq, iq.data(), iq.size() * sizeof(float), hipMemcpyHostToDevice);
	SondeBatchConfig cfg = SONDE_BATCH_CONFIG_INIT;
	cfg.n_channels = C;
	cfg.max_samples = (uint32_t)(n / parts);
	cfg.input_kind = SONDE_INPUT_IQ;
	sonde::BatchDecoder dec;
	if (!dec.init(cfg, on_data, nullptr)) { printf("ERROR init: %s\n", sonde_last_error()); return 1; }
	long fired = 0;
	for (int p = 0; p < parts; p++) {
		// channel c of this submit starts at d_iq + 2*(c*n + p*n/parts): channel stride n, offset inside the row
		const long k = dec.process(d_iq + 2 * (size_t)p * (n / parts), n / parts, n, nullptr);
		if (k < 0) { printf("ERROR process: %s\n", sonde_last_error()); return 1; }
		fired += k;
	}
	printf("DONE fired=%ld calls=%ld seq0=%d\n", fired, g_calls, dec.data(0).seq);
	hipFree(d_iq);
	return 0;
}
