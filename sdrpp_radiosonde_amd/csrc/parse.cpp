// parse.cpp -- subframe walkers producing the SondeData fragments that
// /root/reference/src/decode/decoder.hpp:61-106 merges.  One fragment per CRC-valid subframe,
// as the run() loop there expects ("one decode call yields one fragment").
// RS41 subframe layout: SURVEY.md Appendix B.2.
#include <math.h>
#include <stdio.h>
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
static inline uint32_t rd_u24(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
static inline float rd_f32(const uint8_t *p) { const uint32_t u = rd_u32(p); float f; memcpy(&f, &u, 4); return f; }

// ---- RS41 PTU conversions.  The sensor maths live in the absent sondedump submodule (the reference reads only
// fragment.temp / .rh / .pressure, /root/reference/src/decode/decoder.hpp:84-91); these are the formulas of the
// public RS41 decoders ([RECALL]): the temperature sensor is a resistance measured as a
// frequency count between two reference resistors Rf1, Rf2; R -> T by the calibration polynomial.
// Single-precision throughout, one IEEE operation per written operator (-ffp-contract=off): the oracle
// (oracle/or_physics.c) computes the same bits.
extern "C" float sonde_rs41_temp(uint32_t f, uint32_t f1, uint32_t f2, float rf1, float rf2, const float *co, const float *cal)
{
	const float ff = (float)f, ff1 = (float)f1, ff2 = (float)f2;
	const float g = (ff2 - ff1) / (rf2 - rf1);                 // counts per ohm
	const float Rb = (ff1 * rf2 - ff2 * rf1) / (ff2 - ff1);    // offset, ohm
	const float Rc = ff / g - Rb;
	const float R = Rc * cal[0];
	return (co[0] + co[1] * R + co[2] * R * R + cal[1]) * (1.0f + cal[2]);
}

// Capacitive humidity sensor between two reference capacitors, empirical temperature compensation.
extern "C" float sonde_rs41_rh(uint32_t f, uint32_t f1, uint32_t f2, float calh0, float T)
{
	const float a0 = 7.5f, a1 = 350.0f / calh0;
	const float fh = ((float)f - (float)f1) / ((float)f2 - (float)f1);
	float rh = 100.0f * (a1 * fh - a0);
	rh = rh - T / 5.5f;
	if (T < -25.0f) rh = rh * (1.0f + (-25.0f - T) / 90.0f);
	if (rh < 0.0f) rh = 0.0f;
	if (rh > 100.0f) rh = 100.0f;
	if (T < -273.0f) rh = -1.0f;
	return rh;
}

// RS41-SGP pressure sensor: a capacitance measured like the other sensors (count f between two reference counts),
// mapped to hPa by a polynomial in the normalised count and the sensor's own temperature; 18 coefficients sit in the
// calibration memory at 0x25E.. and fill a 6 x 4 matrix (rows: powers of cfP[24]/fp, columns: powers of T) in the
// order the public RS41 decoders read them ([RECALL]; the reference only reads fragment.pressure,
// /root/reference/src/decode/decoder.hpp:89).  Single precision, row-major accumulation.
extern "C" float sonde_rs41_pressure(uint32_t f, uint32_t f1, uint32_t f2, float tpress, const float *cfP /* [25] */)
{
	if (f1 == f2 || f1 == f) return 0.0f;
	const float a0 = cfP[24] / (((float)f - (float)f1) / ((float)f2 - (float)f1));
	const float a1 = tpress;
	float p = 0.0f, a0j = 1.0f;
	for (int j = 0; j < 6; j++) {
		float a1k = 1.0f;
		for (int k = 0; k < 4; k++) {
			p = p + a0j * a1k * cfP[j * 4 + k];
			a1k = a1k * a1;
		}
		a0j = a0j * a0;
	}
	return p;
}

// calibration memory -> cfP[25]: 0x25E + 4i, i = 0..17, scattered into the matrix as the public decoders do ([RECALL])
static const uint8_t k_rs41_cfp_slot[18] = { 0, 4, 8, 12, 16, 20, 24, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11 };

// Ozone partial pressure (mPa) of an ECC ozonesonde from its cell current (uA) and pump temperature (deg C):
// P = 4.3085e-4 * (I - I_bg) * T_pump[K] * t100, t100 = seconds the pump needs for 100 ml (nominal 28 s; the
// per-instrument value and the background current are flight-preparation data the sonde does not transmit).
// Behind fragment.o3_mpa (decoder.hpp:102-106; the body is in the absent sondedump) [RECALL].
extern "C" float sonde_ozone_mpa(float cell_ua, float tpump_c)
{
	const float t100 = 28.0f, ibg = 0.0f;
	const float p = 4.3085e-4f * (cell_ua - ibg) * (tpump_c + 273.15f) * t100;
	return p > 0.0f ? p : 0.0f;
}

static int hexval(int c) { return (c >= '0' && c <= '9') ? c - '0' : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : -1; }
static bool hexfield(const uint8_t *p, int n, uint32_t *out)
{
	uint32_t v = 0;
	for (int i = 0; i < n; i++) { const int h = hexval(p[i]); if (h < 0) return false; v = (v << 4) | (uint32_t)h; }
	*out = v;
	return true;
}
// XDATA of an OIF411 ozone interface, ASCII hex: "05" instrument type, 2 instrument number, 4 pump temperature (0.01 C),
// 5 cell current (0.0001 uA), 2 battery (0.1 V), 3 pump current (mA), 2 external voltage (0.1 V) [RECALL: public XDATA notes]
static bool xdata_ozone_ascii(const uint8_t *p, int n, float *o3)
{
	for (int i = 0; i + 20 <= n; i++) {
		if (p[i] != '0' || p[i + 1] != '5') continue;
		uint32_t num, tp, cur;
		if (!hexfield(p + i + 2, 2, &num) || !hexfield(p + i + 4, 4, &tp) || !hexfield(p + i + 8, 5, &cur)) continue;
		*o3 = sonde_ozone_mpa((float)cur * 1.0e-4f, (float)(int16_t)tp * 0.01f);
		return true;
	}
	return false;
}

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
				// burst-kill countdown: calibration word 0x316 (seconds, 0xFFFF = timer not armed) [RECALL: public RS41 notes]
				if (body[23] == 0x31 && rd_u16(m_calib + 0x316) != 0xFFFFu) {
					sd.fields |= DATA_SHUTDOWN;
					sd.shutdown = (int)rd_u16(m_calib + 0x316);
				}
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
		case 0x7A: { // measurements: 12 x 24-bit counts (temperature, humidity, humidity-sensor temperature,
			// pressure; each: sensor, reference 1, reference 2), converted with the calibration memory that the
			// 0x79 blocks deliver 16 bytes at a time: Rf1 @0x3D, Rf2 @0x41, co1[3] @0x4D, calT1[3] @0x59, calH @0x75
			if (len < 36) break;
			int have = 0;
			for (int i = 0; i < 51; i++) have += (int)((m_calib_mask >> i) & 1);
			sd.calib_percent = 100.0f * (float)have / 51.0f;
			const uint64_t need = (1ull << 3) | (1ull << 4) | (1ull << 5) | (1ull << 6) | (1ull << 7);
			if ((m_calib_mask & need) != need) break;
			uint32_t m[12];
			for (int i = 0; i < 12; i++) m[i] = rd_u24(body + 3 * i);
			const float rf1 = rd_f32(m_calib + 0x3D), rf2 = rd_f32(m_calib + 0x41);
			float co1[3], calT1[3];
			for (int i = 0; i < 3; i++) { co1[i] = rd_f32(m_calib + 0x4D + 4 * i); calT1[i] = rd_f32(m_calib + 0x59 + 4 * i); }
			const float calh0 = rd_f32(m_calib + 0x75);
			if (m[2] <= m[1] || m[5] <= m[4] || !(rf2 > rf1) || !(calh0 > 0.0f)) break;
			const float T = sonde_rs41_temp(m[0], m[1], m[2], rf1, rf2, co1, calT1);
			if (!(T > -273.0f && T < 200.0f)) break;
			sd.fields = DATA_PTU;
			sd.temp = T;
			sd.rh = sonde_rs41_rh(m[3], m[4], m[5], calh0, T);
			// RS41-SGP: pressure sensor counts (measurement triple 4) + its temperature (i16, 0.01 C, body offset 38) and
			// the coefficient block at 0x25E..0x2A5 (calibration fragments 0x25..0x2A).  An RS41-SG sends zeros there:
			// pressure = 0 and the caller falls back to the ISA model (decoder.hpp:108-110)
			sd.pressure = 0.0f;
			if (m[9] != 0 && m[11] > m[10] && len >= 40 && ((m_calib_mask >> 0x25) & 0x3F) == 0x3F) {
				float cfP[25] = {};
				for (int i = 0; i < 18; i++) cfP[k_rs41_cfp_slot[i]] = rd_f32(m_calib + 0x25E + 4 * i);
				const float P = sonde_rs41_pressure(m[9], m[10], m[11], (float)rd_i16(body + 38) * 0.01f, cfP);
				if (P > 0.0f && P < 1200.0f) sd.pressure = P;
			}
			break;
		}
		case 0x7E: { // XDATA: one instrument-chain byte, then the ASCII hex string of the attached instrument(s)
			float o3;
			if (len >= 21 && xdata_ozone_ascii(body + 1, len - 1, &o3)) {
				sd.fields = DATA_OZONE;
				sd.o3_mpa = o3;
			}
			break;
		}
		default:
			break;
		}
		if (sd.fields) out.push_back(sd);
	}
}

// days since 1970-01-01 of a proleptic Gregorian date
static long long days_from_civil(int y, int m, int d)
{
	y -= m <= 2;
	const long long era = (y >= 0 ? y : y - 399) / 400;
	const unsigned yoe = (unsigned)(y - era * 400);
	const unsigned doy = (153u * (unsigned)(m + (m > 2 ? -3 : 9)) + 2u) / 5u + (unsigned)d - 1u;
	const unsigned doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
	return era * 146097LL + (long long)doe - 719468LL;
}

// DFM06/09/17: frame = 33 corrected Hamming codewords (data nibble = high nibble): 7 CONF, 13 DAT1, 13 DAT2.
// Each DAT block = 48 payload bits + sub-packet id nibble (ids per public DFM notes):
//   0 frame counter (8 bits at bit 24 of the payload), 1 UTC ms of minute (last 16 bits), 2 lat (1e-7 deg) + ground
//   speed (cm/s), 3 lon + heading (0.01 deg), 4 altitude (cm) + climb (cm/s), 8 date/time (12/4/5/5/6 bits from the top).
// DFM temperature (public DFM-09 decoders' formula, [RECALL]): the CONF block carries one measurement channel per
// frame as a 24-bit float (20-bit mantissa / 2^exponent); channel 0 is the NTC thermistor, 3 and 4 the references.
extern "C" float sonde_dfm_temp(float f, float f1, float f2)
{
	const float B0 = 3260.0f, T0 = 25.0f + 273.15f, R0 = 5.0e3f, Rf = 220.0e3f;
	if (f * f1 * f2 == 0.0f) return -273.15f;
	const float g = f2 / Rf;
	const float R = (f - f1) / g;
	if (!(R > 0.0f)) return -273.15f;
	return 1.0f / (1.0f / T0 + 1.0f / B0 * logf(R / R0)) - 273.15f;
}

void SondeParser::feed_dfm(const SondeFrame &f, std::vector<SondeData> &out)
{
	if (f.len != 33 || f.nerr[1] != 0) return;       // a codeword with a detected double error poisons the frame
	{	// CONF block: id nibble + 24-bit float
		const int id = f.data[0] >> 4;
		uint32_t v = 0;
		for (int k = 1; k < 7; k++) v = (v << 4) | (uint32_t)(f.data[k] >> 4);
		if (id <= 4) {
			m_dfm_meas[id] = (float)(v & 0xFFFFFu) / (float)(1u << ((v >> 20) & 0xFu));
			m_dfm_meas_mask |= 1u << id;
		}
		if (id == 0 && (m_dfm_meas_mask & 0x19u) == 0x19u) {
			const float T = sonde_dfm_temp(m_dfm_meas[0], m_dfm_meas[3], m_dfm_meas[4]);
			if (T > -270.0f) {
				SondeData sd;
				memset(&sd, 0, sizeof(sd));
				sd.fields = DATA_PTU;
				sd.temp = T;
				sd.calib_percent = 100.0f;
				out.push_back(sd);
			}
		}
	}
	for (int blk = 0; blk < 2; blk++) {
		const uint8_t *cw = f.data + 7 + 13 * blk;
		unsigned long long pay = 0;
		for (int k = 0; k < 12; k++) pay = (pay << 4) | (unsigned long long)(cw[k] >> 4);
		const int id = cw[12] >> 4;
		SondeData sd;
		memset(&sd, 0, sizeof(sd));
		const int32_t v32 = (int32_t)(uint32_t)(pay >> 16);
		const uint32_t v16 = (uint32_t)(pay & 0xFFFF);
		switch (id) {
		case 0:
			sd.fields = DATA_SEQ;
			sd.seq = (int)((pay >> 16) & 0xFF);
			break;
		case 1:
			if (m_dfm_date >= 0) {
				sd.fields = DATA_TIME;
				sd.time = (time_t)(m_dfm_date + (long long)(pay & 0xFFFF) / 1000);
			}
			break;
		case 2:
			m_dfm_lat = v32 * 1e-7; m_dfm_spd = v16 * 1e-2; m_dfm_have |= 1;
			break;
		case 3:
			m_dfm_lon = v32 * 1e-7; m_dfm_hdg = v16 * 1e-2; m_dfm_have |= 2;
			break;
		case 4:
			if (m_dfm_have == 3) {
				sd.fields = DATA_POS | DATA_SPEED;
				sd.lat = (float)m_dfm_lat; sd.lon = (float)m_dfm_lon; sd.alt = (float)(v32 * 1e-2);
				sd.speed = (float)m_dfm_spd; sd.heading = (float)m_dfm_hdg;
				sd.climb = (float)((int16_t)v16 * 1e-2);
			}
			break;
		case 8: {
			const int year = (int)((pay >> 36) & 0xFFF), mon = (int)((pay >> 32) & 0xF), day = (int)((pay >> 27) & 0x1F);
			const int hour = (int)((pay >> 22) & 0x1F), min = (int)((pay >> 16) & 0x3F);
			if (mon >= 1 && mon <= 12 && day >= 1) m_dfm_date = days_from_civil(year, mon, day) * 86400LL + hour * 3600LL + min * 60LL;
			break;
		}
		default:
			break;
		}
		if (sd.fields) out.push_back(sd);
	}
}

// M10 thermistor (public M10 decoders' model, [RECALL]): a 12-bit ADC reads an NTC behind one of three series/parallel
// resistor ranges; R -> T by a cubic in ln R.  frame: scale byte 0x3E, ADC word 0x3F (LE, offset 0xA000).
extern "C" float sonde_m10_temp(unsigned scale, unsigned adc)
{
	static const float Rs[3] = { 12.1e3f, 36.5e3f, 475.0e3f }, Rp[3] = { 1.0e20f, 330.0e3f, 2000.0e3f };
	const float p0 = 1.07303516e-03f, p1 = 2.41296733e-04f, p2 = 2.26744154e-06f, p3 = 6.52855181e-08f;
	if (scale > 2 || adc == 0 || adc >= 4095) return -273.15f;
	const float x = (4095.0f - (float)adc) / (float)adc;                  // (Vcc - Vout) / Vout
	const float R = Rs[scale] / (x - Rs[scale] / Rp[scale]);
	if (!(R > 0.0f)) return -273.15f;
	const float l = logf(R);
	return 1.0f / (p0 + p1 * l + p2 * l * l + p3 * l * l * l) - 273.15f;
}

// M10 humidity: ratio of two 24-bit timer captures (sensor against reference, frame 0x35 / 0x32, LE) mapped linearly,
// with a small temperature term ([RECALL], low confidence on the constants)
extern "C" float sonde_m10_rh(uint32_t cap_sensor, uint32_t cap_ref, float T)
{
	if (cap_ref == 0) return -1.0f;
	const float q = (float)cap_sensor / (float)cap_ref;
	float rh = (q - 0.8955f) / 0.002f;
	rh = rh + (20.0f - T) * 0.03f;
	if (rh < 0.0f) rh = 0.0f;
	if (rh > 100.0f) rh = 100.0f;
	return rh;
}

// M20 thermistor ([RECALL]): 12-bit ADC word at 0x04 (LE), fixed 22.1 k series resistor, Beta model (B = 3450 K, 15 k at 0 C)
extern "C" float sonde_m20_temp(unsigned adc)
{
	if (adc == 0 || adc >= 4095) return -273.15f;
	const float R = 22.1e3f * (float)adc / (4095.0f - (float)adc);
	return 1.0f / (1.0f / 273.15f + logf(R / 15.0e3f) / 3450.0f) - 273.15f;
}

// M10: 101-byte frame (length byte 0x64, type 0x9F), M20: 70-byte frame (0x45, 0x20); big-endian fields at fixed
// offsets (public M10/M20 notes [RECALL]).  The framer checks the 16-bit checksum at the
// position the length byte implies.
void SondeParser::feed_m10(const SondeFrame &f, std::vector<SondeData> &out)
{
	if (f.nerr[0] != 0) return;                      // checksum failed
	const uint8_t *d = f.data;
	auto be16 = [&](int o) { return (int16_t)((d[o] << 8) | d[o + 1]); };
	auto be24 = [&](int o) { return (int32_t)(((uint32_t)d[o] << 16) | ((uint32_t)d[o + 1] << 8) | d[o + 2]); };
	auto be32 = [&](int o) { return (int32_t)(((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]); };
	SondeData sd;
	memset(&sd, 0, sizeof(sd));
	if (f.len == 101 && d[0] == 0x64 && d[1] == 0x9F) {
		const double ve = be16(0x04) / 200.0, vn = be16(0x06) / 200.0, vu = be16(0x08) / 200.0;
		const uint32_t tow_ms = (uint32_t)be32(0x0A);
		const unsigned week = (unsigned)(uint16_t)be16(0x20);
		sd.fields = DATA_POS | DATA_SPEED | DATA_TIME;
		sd.lat = (float)(be32(0x0E) * (360.0 / 4294967296.0));
		sd.lon = (float)(be32(0x12) * (360.0 / 4294967296.0));
		sd.alt = (float)(be32(0x16) / 1000.0);
		sd.speed = (float)sqrt(ve * ve + vn * vn);
		double hdg = atan2(ve, vn) * 180.0 / M_PI;
		if (hdg < 0.0) hdg += 360.0;
		sd.heading = (float)hdg;
		sd.climb = (float)vu;
		sd.time = (time_t)(315964800LL + (long long)week * 604800LL + (long long)(tow_ms / 1000) - 18LL);
		out.push_back(sd);
		// temperature / humidity (README.md:12 ticks both for M10)
		const unsigned adc = (unsigned)(((d[0x40] << 8) | d[0x3F]) - 0xA000) & 0xFFFFu;
		const float T = sonde_m10_temp(d[0x3E], adc);
		if (T > -270.0f) {
			SondeData pt;
			memset(&pt, 0, sizeof(pt));
			pt.fields = DATA_PTU;
			pt.temp = T;
			const uint32_t ref = (uint32_t)d[0x32] | ((uint32_t)d[0x33] << 8) | ((uint32_t)d[0x34] << 16);
			const uint32_t sen = (uint32_t)d[0x35] | ((uint32_t)d[0x36] << 8) | ((uint32_t)d[0x37] << 16);
			const float rh = sonde_m10_rh(sen, ref, T);
			pt.rh = rh < 0.0f ? 0.0f : rh;
			pt.calib_percent = 100.0f;
			out.push_back(pt);
		}
	} else if (f.len == 70 && d[0] == 0x45 && d[1] == 0x20) {
		// M20: alt 0x08 (3 bytes, cm), vE 0x0B, vN 0x0D (0.01 m/s), time of week 0x0F (3 bytes, s), vU 0x18, week 0x1A,
		// lat 0x1C, lon 0x20 (1e-6 deg)
		const double ve = be16(0x0B) / 100.0, vn = be16(0x0D) / 100.0, vu = be16(0x18) / 100.0;
		const uint32_t tow_s = (uint32_t)be24(0x0F);
		const unsigned week = (unsigned)(uint16_t)be16(0x1A);
		sd.fields = DATA_POS | DATA_SPEED | DATA_TIME;
		sd.lat = (float)(be32(0x1C) * 1e-6);
		sd.lon = (float)(be32(0x20) * 1e-6);
		sd.alt = (float)(be24(0x08) / 100.0);
		sd.speed = (float)sqrt(ve * ve + vn * vn);
		double hdg = atan2(ve, vn) * 180.0 / M_PI;
		if (hdg < 0.0) hdg += 360.0;
		sd.heading = (float)hdg;
		sd.climb = (float)vu;
		sd.time = (time_t)(315964800LL + (long long)week * 604800LL + (long long)tow_s - 18LL);
		out.push_back(sd);
		const float T = sonde_m20_temp((unsigned)(d[0x04] | (d[0x05] << 8)) & 0xFFFu);     // README.md:13: M20 GPS + T
		if (T > -270.0f) {
			SondeData pt;
			memset(&pt, 0, sizeof(pt));
			pt.fields = DATA_PTU;
			pt.temp = T;
			pt.calib_percent = 100.0f;
			out.push_back(pt);
		}
	}
}

// iMet-1 / iMet-4 packets (SPEC 3.3c; field layout: public iMet notes, [RECALL]):
//   01 01 pkt#(u16) P(u24, 0.01 hPa) T(i16, 0.01 C) U(u16, 0.01 %) Vbat(u8) crc      PTU  (type 4 PTUX: the same, longer)
//   01 02 lat(f32 deg) lon(f32 deg) alt(u16, m + 5000) nsat(u8) hh mm ss crc          GPS
// little-endian fields, CRC big-endian.  Packets with a bad CRC are dropped.
void SondeParser::feed_imet(const SondeFrame &f, std::vector<SondeData> &out)
{
	if (f.nerr[0] != 0 || f.len < 7) return;
	const uint8_t *d = f.data;
	SondeData sd;
	memset(&sd, 0, sizeof(sd));
	if ((d[1] == 1 || d[1] == 4) && f.len >= 14) {
		sd.fields = DATA_SEQ | DATA_PTU;
		sd.seq = (int)rd_u16(d + 2);
		sd.pressure = (float)rd_u24(d + 4) / 100.0f;
		sd.temp = (float)rd_i16(d + 7) / 100.0f;
		sd.rh = (float)rd_u16(d + 9) / 100.0f;
		sd.calib_percent = 100.0f;
	} else if (d[1] == 2 && f.len >= 18) {
		sd.fields = DATA_POS | DATA_TIME;
		sd.lat = rd_f32(d + 2);
		sd.lon = rd_f32(d + 6);
		sd.alt = (float)rd_u16(d + 10) - 5000.0f;
		sd.time = (time_t)(3600 * (int)d[13] + 60 * (int)d[14] + (int)d[15]);      // time of day only: no date on air
	} else if (d[1] == 3 && f.len >= 5 + 8 && d[3] == 0x01) {
		// XDATA, binary: 01 03 len | instrument id (01 = ECC ozonesonde) index | cell current u16 BE (0.001 uA),
		// pump temperature i16 BE (0.01 C), pump current u8 (mA), battery u8 (0.1 V)  [RECALL: public iMet XDATA notes]
		const float cur = (float)((d[5] << 8) | d[6]) * 1.0e-3f;
		const float tp = (float)(int16_t)((d[7] << 8) | d[8]) * 0.01f;
		sd.fields = DATA_OZONE;
		sd.o3_mpa = sonde_ozone_mpa(cur, tp);
	}
	if (sd.fields) out.push_back(sd);
}

// iMS-100 / RS-11G (Meisei).  The FEC stage delivers the 408 data bits of a frame = 12 BCH blocks x 2 groups of 17 bits:
// a 16-bit word, MSB first, followed by its parity bit (odd overall parity: the 17 bits hold an odd number of ones).
// Words with a wrong parity bit, or from a block the BCH decoder gave up on, are unusable: a field is emitted only if
// all of its words are good.  Word layout ([RECALL]: the word/parity/BCH structure is the public description of the
// format; the offsets below are this repo's -- the reference's layout is in the absent sondedump):
//   w0 frame counter | w1:w2 calibration word #(counter mod 4): 0 serial number (u32), 1..3 float32 c0, c1, c2 of the
//   thermistor polynomial | w3 temperature count f (T = c0 + c1 f + c2 f^2) | w4 humidity, 0.01 % |
//   w5:w6 GPS time of week, ms | w7 GPS week | w8:w9 latitude, w10:w11 longitude (i32, 1e-6 deg) | w12:w13 altitude
//   (i32, cm) | w14 ground speed, w15 heading (u16, 0.01) | w16 climb (i16, 0.01 m/s) | w17..w23 spare
extern "C" float sonde_ims100_temp(uint32_t f, float c0, float c1, float c2)
{
	const float x = (float)f;
	return c0 + c1 * x + c2 * x * x;
}

void SondeParser::feed_ims100(const SondeFrame &f, std::vector<SondeData> &out)
{
	if (f.len != 51) return;
	uint16_t w[24];
	uint32_t good = 0;
	for (int g = 0; g < 24; g++) {
		uint32_t v = 0;
		for (int b = 0; b < 17; b++) {
			const int bit = 17 * g + b;
			v = (v << 1) | ((f.data[bit >> 3] >> (7 - (bit & 7))) & 1u);
		}
		w[g] = (uint16_t)(v >> 1);
		if (__builtin_popcount(v) & 1) good |= 1u << g;
	}
	if (f.nerr[1] != 0) good = 0;          // an uncorrectable BCH block: which one is not recorded, drop the frame
	auto ok = [&](int a, int b) { const uint32_t m = ((1u << (b - a + 1)) - 1u) << a; return (good & m) == m; };
	auto u32 = [&](int a) { return ((uint32_t)w[a] << 16) | w[a + 1]; };
	SondeData sd;
	if (ok(0, 0)) {
		memset(&sd, 0, sizeof(sd));
		sd.fields = DATA_SEQ;
		sd.seq = w[0];
		if (ok(1, 2)) {
			m_ims_cal[w[0] & 3] = u32(1);
			m_ims_cal_mask |= 1u << (w[0] & 3);
			if ((w[0] & 3) == 0) {
				sd.fields |= DATA_SERIAL;
				snprintf(sd.serial, sizeof(sd.serial), "%u", (unsigned)u32(1));
			}
		}
		out.push_back(sd);
	}
	if (ok(5, 7)) {
		memset(&sd, 0, sizeof(sd));
		sd.fields = DATA_TIME;
		sd.time = (time_t)(315964800LL + (long long)w[7] * 604800LL + (long long)(u32(5) / 1000u) - 18LL);
		out.push_back(sd);
	}
	if (ok(8, 16)) {
		memset(&sd, 0, sizeof(sd));
		sd.fields = DATA_POS | DATA_SPEED;
		sd.lat = (float)((int32_t)u32(8) * 1e-6);
		sd.lon = (float)((int32_t)u32(10) * 1e-6);
		sd.alt = (float)((int32_t)u32(12) * 1e-2);
		sd.speed = (float)(w[14] * 1e-2);
		sd.heading = (float)(w[15] * 1e-2);
		sd.climb = (float)((int16_t)w[16] * 1e-2);
		out.push_back(sd);
	}
	if (ok(3, 4)) {
		int have = 0;
		for (int i = 0; i < 4; i++) have += (m_ims_cal_mask >> i) & 1;
		if ((m_ims_cal_mask & 0xEu) == 0xEu) {
			float c[3];
			for (int i = 0; i < 3; i++) memcpy(&c[i], &m_ims_cal[1 + i], 4);
			memset(&sd, 0, sizeof(sd));
			sd.fields = DATA_PTU;
			sd.temp = sonde_ims100_temp(w[3], c[0], c[1], c[2]);
			sd.rh = (float)w[4] * 0.01f;
			sd.calib_percent = 100.0f * (float)have / 4.0f;
			if (sd.temp > -273.0f && sd.temp < 200.0f) out.push_back(sd);
		}
	}
}

// MRZ-N1 (README.md:19: GPS + temperature).  45-byte frame behind the AA BF 35 header, little-endian fields
// ([RECALL]: ECEF position/velocity, time of day + date, a rotating calibration/serial word and a CRC16 are what the
// public MP3-H1 / MRZ decoders describe; the offsets are this repo's, the reference's layout is in the absent sondedump):
//   0 frame counter u16 | 2 hh mm ss | 5 day month year-2000 | 8 X, 12 Y, 16 Z (i32, cm ECEF) | 20 vX, 22 vY, 24 vZ
//   (i16, cm/s) | 26 satellites | 27 temperature (i16, 0.01 C) | 29 fragment index, 30 fragment word u32 (index 0: serial)
//   | 34..42 spare | 43 CRC16 (reflected 0xA001, init 0xFFFF) of bytes 0..42
void SondeParser::feed_mrzn1(const SondeFrame &f, std::vector<SondeData> &out)
{
	if (f.len != 45 || f.nerr[0] != 0) return;
	const uint8_t *d = f.data;
	SondeData sd;
	memset(&sd, 0, sizeof(sd));
	sd.fields = DATA_SEQ;
	sd.seq = (int)rd_u16(d);
	if (d[29] == 0) {
		sd.fields |= DATA_SERIAL;
		snprintf(sd.serial, sizeof(sd.serial), "MRZ-%u", (unsigned)rd_u32(d + 30));
	}
	out.push_back(sd);
	if (d[6] >= 1 && d[6] <= 12 && d[5] >= 1 && d[5] <= 31 && d[2] < 24 && d[3] < 60 && d[4] < 61) {
		memset(&sd, 0, sizeof(sd));
		sd.fields = DATA_TIME;
		sd.time = (time_t)(days_from_civil(2000 + d[7], d[6], d[5]) * 86400LL + d[2] * 3600LL + d[3] * 60LL + d[4]);
		out.push_back(sd);
	}
	const double x = rd_i32(d + 8) / 100.0, y = rd_i32(d + 12) / 100.0, z = rd_i32(d + 16) / 100.0;
	if (d[26] >= 4 && !(x == 0.0 && y == 0.0 && z == 0.0)) {
		const double vx = rd_i16(d + 20) / 100.0, vy = rd_i16(d + 22) / 100.0, vz = rd_i16(d + 24) / 100.0;
		double lat, lon, alt;
		ecef_to_lla(x, y, z, &lat, &lon, &alt);
		const double la = lat * M_PI / 180.0, lo = lon * M_PI / 180.0;
		const double ve = -vx * sin(lo) + vy * cos(lo);
		const double vn = -vx * sin(la) * cos(lo) - vy * sin(la) * sin(lo) + vz * cos(la);
		const double vu = vx * cos(la) * cos(lo) + vy * cos(la) * sin(lo) + vz * sin(la);
		double hdg = atan2(ve, vn) * 180.0 / M_PI;
		if (hdg < 0.0) hdg += 360.0;
		memset(&sd, 0, sizeof(sd));
		sd.fields = DATA_POS | DATA_SPEED;
		sd.lat = (float)lat; sd.lon = (float)lon; sd.alt = (float)alt;
		sd.speed = (float)sqrt(ve * ve + vn * vn);
		sd.heading = (float)hdg;
		sd.climb = (float)vu;
		out.push_back(sd);
	}
	memset(&sd, 0, sizeof(sd));
	sd.fields = DATA_PTU;
	sd.temp = (float)rd_i16(d + 27) * 0.01f;
	sd.calib_percent = 100.0f;
	out.push_back(sd);
}

// SRS-C50 (README.md:17: GPS + temperature).  One value per 9-byte packet 00 FF <type> <value, 4 bytes big-endian> <c1> <c2>
// EXPERIMENTAL: [RECALL] the packet shape, the Fletcher sum and the type numbers (0x14 date, 0x15 time, 0x16 latitude,
// 0x17 longitude, 0x18 altitude; 0x03 temperature, 0x10 sonde number) are the public C34/C50 decoders' as recalled; the value
// SCALINGS are this repo's and unverified against a recorded sonde (the reference's are in the absent sondedump) -- the
// generator shares them, so the round-trip tests cannot see a wrong scaling:
//   0x03 air temperature (float32) | 0x10 sonde number | 0x14 date as the decimal number ddmmyy | 0x15 UTC time as the decimal
//   number hhmmss | 0x16 latitude, 0x17 longitude (i32, 1e-6 deg) | 0x18 altitude (i32, cm; emits DATA_POS)
void SondeParser::feed_c50(const SondeFrame &f, std::vector<SondeData> &out)
{
	if (f.len != 9 || f.nerr[0] != 0) return;
	const uint8_t *d = f.data;
	const uint32_t v = ((uint32_t)d[3] << 24) | ((uint32_t)d[4] << 16) | ((uint32_t)d[5] << 8) | d[6];
	SondeData sd;
	memset(&sd, 0, sizeof(sd));
	switch (d[2]) {
	case 0x03: {
		float t;
		memcpy(&t, &v, 4);
		if (t > -120.0f && t < 80.0f) { sd.fields = DATA_PTU; sd.temp = t; sd.calib_percent = 100.0f; }
		break;
	}
	case 0x10:
		sd.fields = DATA_SERIAL;
		snprintf(sd.serial, sizeof(sd.serial), "C50-%u", (unsigned)v);
		break;
	case 0x16: m_c50_lat = (int32_t)v * 1e-6; m_c50_have |= 1; break;
	case 0x17: m_c50_lon = (int32_t)v * 1e-6; m_c50_have |= 2; break;
	case 0x18:
		if ((m_c50_have & 3) == 3) {
			sd.fields = DATA_POS;
			sd.lat = (float)m_c50_lat; sd.lon = (float)m_c50_lon; sd.alt = (float)((int32_t)v * 1e-2);
		}
		break;
	case 0x15:
		if ((m_c50_have & 4) && v < 240000u && (v / 100) % 100 < 60 && v % 100 < 61) {
			sd.fields = DATA_TIME;
			sd.time = (time_t)(m_c50_date + (long long)(v / 10000) * 3600 + (long long)((v / 100) % 100) * 60 + (long long)(v % 100));
		}
		break;
	case 0x14: {
		const int day = (int)(v / 10000), mon = (int)((v / 100) % 100), yr = (int)(v % 100);
		if (day >= 1 && day <= 31 && mon >= 1 && mon <= 12) { m_c50_date = days_from_civil(2000 + yr, mon, day) * 86400LL; m_c50_have |= 4; }
		break;
	}
	default:
		break;
	}
	if (sd.fields) out.push_back(sd);
}

void SondeParser::feed(const SondeFrame &f, std::vector<SondeData> &out)
{
	switch (f.type) {
	case SONDE_RS41: feed_rs41(f, out); break;
	case SONDE_DFM09: feed_dfm(f, out); break;
	case SONDE_M10: feed_m10(f, out); break;
	case SONDE_IMET4: feed_imet(f, out); break;
	case SONDE_IMS100: feed_ims100(f, out); break;
	case SONDE_MRZN1: feed_mrzn1(f, out); break;
	case SONDE_C50: feed_c50(f, out); break;
	default: break;
	}
}

// Stateful parser handle: what a host keeps per channel (RS41 calibration and DFM date/position arrive
// spread over many frames).
struct SondeParserHandle { SondeParser p; explicit SondeParserHandle(int t) : p(t) {} };
extern "C" SondeParserHandle *sonde_parser_create(int type) { return new SondeParserHandle(type); }
extern "C" void sonde_parser_destroy(SondeParserHandle *h) { delete h; }
extern "C" int sonde_parser_feed(SondeParserHandle *h, const SondeFrame *f, SondeData *out, int cap)
{
	if (!h || !f || !out || cap <= 0) return 0;
	std::vector<SondeData> v;
	h->p.feed(*f, v);
	int n = 0;
	for (; n < (int)v.size() && n < cap; n++) out[n] = v[n];
	return n;
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

// ---- post-FEC derived quantities the reference's Decoder computes while merging fragments.
// Restated from /root/reference/src/decode/decoder.hpp:132-137 (Magnus dew point) and :138-174 (ISA
// barometric formula; note the double-precision 1e-2 factor and the table's 77 km last layer).
extern "C" float sonde_dewpt(float temp, float rh)
{
	const float g = (logf(rh / 100.0f) + (17.27f * temp / (237.3f + temp))) / 17.27f;
	return 237.3f * g / (1 - g);
}

extern "C" float sonde_altitude_to_pressure(float alt)
{
	struct Layer { float hb, Lb, Pb, Tb; };
	static const Layer isa[7] = {
		{ 0.0,     -0.0065, 101325.0, 288.15 }, { 11000.0, 0.0,    22632.1, 216.65 }, { 20000.0, 0.001, 5474.89, 216.65 },
		{ 32000.0,  0.0028, 868.02,   228.65 }, { 47000.0, 0.0,    110.91,  270.65 }, { 51000.0, -0.0028, 66.94, 270.65 },
		{ 77000.0, -0.002,  3.96,     214.65 },
	};
	const float g0 = 9.80665, M = 0.0289644, R_star = 8.3144598;
	int b = 6;
	for (int i = 0; i < 6; i++) if (alt < isa[i + 1].hb) { b = i; break; }
	const Layer &l = isa[b];
	if (l.Lb != 0) {
		const float base = (l.Tb + l.Lb * (alt - l.hb)) / l.Tb;
		const float expo = -(g0 * M) / (R_star * l.Lb);
		return (float)(1e-2 * l.Pb * powf(base, expo));
	}
	return (float)(1e-2 * l.Pb * expf(-g0 * M * (alt - l.hb) / (R_star * l.Tb)));
}
