// sinks.cpp -- C entry points for include/sonde_sinks.hpp (so that non-C++ hosts and the ctypes tests
// can drive the GPX / CSV writers).
#include "../../include/sonde_sinks.hpp"

extern "C" {
void *sonde_gpx_open(const char *path)
{
	sonde::GpxWriter *g = new sonde::GpxWriter();
	if (!g->open(path)) { delete g; return nullptr; }
	return g;
}
void sonde_gpx_close(void *g) { delete (sonde::GpxWriter *)g; }
void sonde_gpx_start_track(void *g, const char *name) { if (g) ((sonde::GpxWriter *)g)->startTrack(name); }
void sonde_gpx_stop_track(void *g) { if (g) ((sonde::GpxWriter *)g)->stopTrack(); }
void sonde_gpx_add_point(void *g, long t, float lat, float lon, float alt, float spd, float hdg)
{
	if (g) ((sonde::GpxWriter *)g)->addTrackPoint((time_t)t, lat, lon, alt, spd, hdg);
}

void *sonde_ptu_open(const char *path)
{
	sonde::PtuWriter *p = new sonde::PtuWriter();
	if (!p->open(path)) { delete p; return nullptr; }
	return p;
}
void sonde_ptu_close(void *p) { delete (sonde::PtuWriter *)p; }
void sonde_ptu_add_point(void *p, long t, float temp, float rh, float dewpt, float pressure, float lat, float lon,
                         float alt, float spd, float hdg, float climb, const char *aux)
{
	if (!p) return;
	sonde::FullData d;
	d.time = (time_t)t; d.temp = temp; d.rh = rh; d.dewpt = dewpt; d.pressure = pressure;
	d.lat = lat; d.lon = lon; d.alt = alt; d.spd = spd; d.hdg = hdg; d.climb = climb; d.auxData = aux ? aux : "";
	((sonde::PtuWriter *)p)->addPoint(d);
}
}
