// parse.cpp -- subframe walkers producing the SondeData fragments that
// /root/reference/src/decode/decoder.hpp:61-106 merges.  One fragment per CRC-valid subframe,
// as the run() loop there expects ("one decode call yields one fragment").
// RS41 subframe layout: SURVEY.md Appendix B.2.
#include <math.h>
#include <string.h>
#include "parse.h"

uint16_t sonde_crc16_ccitt(const uint8_t *p, size_t n)
{
	uint16_t crc = 0xFFFF;
	for (size_t i = 0; i < n; i++) {
		crc ^= (uint16_t)p[i] << 8;
		for (int k = 0; k < 8; k++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
	}
	return crc;
}

static inline uint32_t rd_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static inline uint32_t rd_u32(const uint8_t *p) { return rd_u16(p) | (rd_u16(p + 2) << 16); }
static inline int32_t rd_i32(const uint8_t *p) { return (int32_t)rd_u32(p); }
static inline int16_t rd_i16(const uint8_t *p) { return (int16_t)rd_u16(p); }

// WGS84 ECEF (metres) -> geodetic latitude/longitude (degrees) and height (metres)
static void ecef_to_lla(double x, double y, double z, double *lat, double *lon, double *alt)
{
	const double a = 6378137.0, e2 = 6.69437999014e-3;
	const double b = a * sqrt(1.0 - e2), ep2 = (a * a - b * b) / (b * b);
	const double p = sqrt(x * x + y * y);
	const double th = atan2(z * a, p * b);
	const double st = sin(th), ct = cos(th);
	const double phi = atan2(z + ep2 * b * st * st * st, p - e2 * a * ct * ct * ct);
	const double N = a / sqrt(1.0 - e2 * sin(phi) * sin(phi));
	*lat = phi * 180.0 / M_PI;
	*lon = atan2(y, x) * 180.0 / M_PI;
	*alt = p / cos(phi) - N;
}

void SondeParser::feed_rs41(const SondeFrame &f, std::vector<SondeData> &out)
{
	const uint8_t *d = f.data;
	int off = 57;
	while (off + 4 <= f.len) {
		const int type = d[off], len = d[off + 1];
		if (off + 2 + len + 2 > f.len) break;
		const uint8_t *body = d + off + 2;
		const uint16_t crc = (uint16_t)(body[len] | (body[len + 1] << 8));
		const bool ok = sonde_crc16_ccitt(body, (size_t)len) == crc;
		off += len + 4;
		if (!ok) continue;
		SondeData sd;
		memset(&sd, 0, sizeof(sd));
		switch (type) {
		case 0x79:   // status: frame number, serial, one of 51 calibration fragments
			if (len < 40) break;
			sd.fields = DATA_SEQ | DATA_SERIAL;
			sd.seq = (int)rd_u16(body);
			memcpy(sd.serial, body + 2, 8);
			sd.serial[8] = 0;
			if (body[23] < 51) {
				memcpy(m_calib + 16 * body[23], body + 24, 16);
				m_calib_mask |= 1ull << body[23];
			}
			break;
		case 0x7C: { // GPS info: week, milliseconds of week
			if (len < 6) break;
			const uint32_t week = rd_u16(body), ms = rd_u32(body + 2);
			sd.fields = DATA_TIME;
			// GPS epoch 1980-01-06T00:00:00Z = 315964800; GPS-UTC = 18 s
			sd.time = (time_t)(315964800LL + (long long)week * 604800LL + (long long)(ms / 1000) - 18LL);
			break;
		}
		case 0x7B: { // GPS position: ECEF cm, velocity cm/s
			if (len < 21) break;
			const double x = rd_i32(body) / 100.0, y = rd_i32(body + 4) / 100.0, z = rd_i32(body + 8) / 100.0;
			const double vx = rd_i16(body + 12) / 100.0, vy = rd_i16(body + 14) / 100.0, vz = rd_i16(body + 16) / 100.0;
			if (x == 0.0 && y == 0.0 && z == 0.0) break;
			double lat, lon, alt;
			ecef_to_lla(x, y, z, &lat, &lon, &alt);
			const double la = lat * M_PI / 180.0, lo = lon * M_PI / 180.0;
			const double ve = -vx * sin(lo) + vy * cos(lo);
			const double vn = -vx * sin(la) * cos(lo) - vy * sin(la) * sin(lo) + vz * cos(la);
			const double vu = vx * cos(la) * cos(lo) + vy * cos(la) * sin(lo) + vz * sin(la);
			double hdg = atan2(ve, vn) * 180.0 / M_PI;
			if (hdg < 0.0) hdg += 360.0;
			sd.fields = DATA_POS | DATA_SPEED;
			sd.lat = (float)lat; sd.lon = (float)lon; sd.alt = (float)alt;
			sd.speed = (float)sqrt(ve * ve + vn * vn);
			sd.heading = (float)hdg;
			sd.climb = (float)vu;
			break;
		}
		case 0x7A: { // measurements: calibration progress only (PTU physics: DESIGN.md "next")
			int have = 0;
			for (int i = 0; i < 51; i++) have += (int)((m_calib_mask >> i) & 1);
			sd.fields = 0;   // no DATA_PTU until the calibrated conversion lands
			sd.calib_percent = 100.0f * (float)have / 51.0f;
			break;
		}
		default:
			break;
		}
		if (sd.fields) out.push_back(sd);
	}
}

void SondeParser::feed(const SondeFrame &f, std::vector<SondeData> &out)
{
	switch (f.type) {
	case SONDE_RS41: feed_rs41(f, out); break;
	default: break;
	}
}

extern "C" int sonde_parse_frame(const SondeFrame *f, SondeData *out, int cap)
{
	if (!f || !out || cap <= 0) return 0;
	SondeParser p((int)f->type);
	std::vector<SondeData> v;
	p.feed(*f, v);
	int n = 0;
	for (; n < (int)v.size() && n < cap; n++) out[n] = v[n];
	return n;
}
