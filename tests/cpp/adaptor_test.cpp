// CPU test of include/sonde_decoder.hpp with a scripted decoder triple (no GPU): checks the merge order,
// the PROCEED re-entrancy loop, dew point / ISA fallback and the callback-iff-fields rule of
// /root/reference/src/decode/decoder.hpp:53-119.  Prints one line per callback for the pytest wrapper.
#include <cstdio>
#include <cstring>
#include <vector>
#include "sonde_decoder.hpp"

struct Fake { std::vector<SondeData> script; size_t next = 0; int calls = 0; const float *last = nullptr; };
static Fake *g_fake;

static Fake *fake_init(int sr) { return sr == 48000 ? g_fake : nullptr; }
static void fake_deinit(Fake *) {}
static ParserStatus fake_get(Fake *f, SondeData *dst, const float *src, size_t len)
{
	f->calls++;
	if (f->last && f->last != src) { printf("ERROR buffer changed before PROCEED\n"); }
	if (len != 480) printf("ERROR len\n");
	if (f->next < f->script.size() && f->script[f->next].fields != -1) { *dst = f->script[f->next++]; f->last = src; return PARSED; }
	if (f->next < f->script.size()) f->next++;     // -1 marks "buffer exhausted"
	f->last = nullptr;
	return PROCEED;
}

static void cb(sonde::FullData *d, void *ctx)
{
	(*(int *)ctx)++;
	printf("CB seq=%d serial=%s lat=%.6f lon=%.6f alt=%.1f spd=%.2f hdg=%.2f climb=%.2f time=%ld temp=%.2f rh=%.2f dewpt=%a pressure=%a cal=%d calp=%.1f kill=%d aux=%s\n",
	       d->seq, d->serial.c_str(), d->lat, d->lon, d->alt, d->spd, d->hdg, d->climb, (long)d->time, d->temp, d->rh, d->dewpt, d->pressure,
	       (int)d->calibrated, d->calib_percent, d->burstkill, d->auxData.c_str());
}

static SondeData frag(int fields) { SondeData s; memset(&s, 0, sizeof(s)); s.fields = fields; return s; }

int main()
{
	Fake fake;
	g_fake = &fake;
	SondeData a = frag(DATA_SEQ | DATA_SERIAL); a.seq = 1234; strcpy(a.serial, "S1234567");
	SondeData b = frag(DATA_POS | DATA_SPEED); b.lat = 47.5f; b.lon = 8.25f; b.alt = 12000.0f; b.speed = 12.0f; b.heading = 90.0f; b.climb = 5.0f;
	SondeData c = frag(0);                                    // fields == 0: merged (nothing), no callback
	SondeData end = frag(-1);
	SondeData d = frag(DATA_PTU); d.temp = -50.0f; d.rh = 30.0f; d.pressure = 0.0f; d.calib_percent = 100.0f;
	SondeData e = frag(DATA_TIME | DATA_SHUTDOWN); e.time = 1700000000; e.shutdown = 3600;
	SondeData f = frag(DATA_OZONE); f.o3_mpa = 3.14159f;
	SondeData g = frag(DATA_PTU); g.temp = 10.0f; g.rh = 50.0f; g.pressure = 900.0f; g.calib_percent = 40.0f;
	fake.script = { a, b, c, end, d, e, end, f, g, end };

	int fired = 0;
	sonde::Decoder<Fake, fake_init, fake_deinit, fake_get> dec;
	if (dec.init(44100, cb, &fired)) printf("ERROR init accepted 44100\n");
	if (!dec.init(48000, cb, &fired)) { printf("ERROR init\n"); return 1; }
	float buf1[480], buf2[480], buf3[480];
	int n1 = dec.process(buf1, 480);
	int n2 = dec.process(buf2, 480);
	int n3 = dec.process(buf3, 480);
	printf("DONE fired=%d per_buffer=%d,%d,%d get_calls=%d\n", fired, n1, n2, n3, fake.calls);
	return 0;
}
