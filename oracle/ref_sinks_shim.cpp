// ref_sinks_shim.cpp -- C entry points that drive the REFERENCE's own GPXWriter / PTUWriter classes
// (/root/reference/src/gpx.cpp, /root/reference/src/ptu.cpp, compiled from where they lie; nothing of
// theirs is copied into this repo).  TEST INFRASTRUCTURE ONLY: it produces the byte-exact GPX / CSV
// text the product's sinks (include/sonde_sinks.hpp) are compared with.  Built by oracle/Makefile
// target `ref` into oracle/_ref/libref_sinks.so when /root/reference is present.
#include <string>
#include "gpx.hpp"
#include "ptu.hpp"

extern "C" {
void *ref_gpx_new(void) { return new GPXWriter(); }
void ref_gpx_free(void *g) { delete (GPXWriter *)g; }
int ref_gpx_init(void *g, const char *fname) { return ((GPXWriter *)g)->init(fname) ? 1 : 0; }
void ref_gpx_deinit(void *g) { ((GPXWriter *)g)->deinit(); }
void ref_gpx_start_track(void *g, const char *name) { ((GPXWriter *)g)->startTrack(name); }
void ref_gpx_stop_track(void *g) { ((GPXWriter *)g)->stopTrack(); }
void ref_gpx_add_point(void *g, long t, float lat, float lon, float alt, float spd, float hdg)
{
	((GPXWriter *)g)->addTrackPoint((time_t)t, lat, lon, alt, spd, hdg);
}

void *ref_ptu_new(void) { return new PTUWriter(); }
void ref_ptu_free(void *p) { delete (PTUWriter *)p; }
int ref_ptu_init(void *p, const char *fname) { return ((PTUWriter *)p)->init(fname) ? 1 : 0; }
void ref_ptu_deinit(void *p) { ((PTUWriter *)p)->deinit(); }
void ref_ptu_add_point(void *p, long t, float temp, float rh, float dewpt, float pressure, float lat, float lon,
                       float alt, float spd, float hdg, float climb, const char *aux)
{
	SondeFullData d;
	d.time = (time_t)t; d.temp = temp; d.rh = rh; d.dewpt = dewpt; d.pressure = pressure;
	d.lat = lat; d.lon = lon; d.alt = alt; d.spd = spd; d.hdg = hdg; d.climb = climb; d.auxData = aux;
	((PTUWriter *)p)->addPoint(&d);
}
}
