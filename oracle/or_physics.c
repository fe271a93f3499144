/*
 * or_physics.c -- oracle for the post-FEC derived quantities.
 * TEST INFRASTRUCTURE ONLY (see sonde_oracle.h).
 *
 * Restates the arithmetic of
 *   dewpt()                 /root/reference/src/decode/decoder.hpp:132-137
 *   altitude_to_pressure()  /root/reference/src/decode/decoder.hpp:138-174
 * Pinned by the two known answers recorded in SURVEY.md section 8a (a11, a12), which the
 * surveyor obtained by running the reference bodies: dewpt(-50, 30) = -59.7688,
 * altitude_to_pressure(12000) = 193.3049 hPa.
 */
#include <math.h>
#include "sonde_oracle.h"

float or_dewpt(float temp, float rh)
{
	/* Magnus form with a = 17.27, b = 237.3 (decoder.hpp:135-136) */
	const float g = (logf(rh / 100.0f) + (17.27f * temp / (237.3f + temp))) / 17.27f;
	return 237.3f * g / (1 - g);
}

/* ISA layers, decoder.hpp:145-148; values are double literals narrowed to float there too */
typedef struct { float hb, Lb, Pb, Tb; } IsaLayer;
static const IsaLayer isa[7] = {
	{ 0.0,     -0.0065, 101325.0, 288.15 },
	{ 11000.0,  0.0,    22632.1,  216.65 },
	{ 20000.0,  0.001,  5474.89,  216.65 },
	{ 32000.0,  0.0028, 868.02,   228.65 },
	{ 47000.0,  0.0,    110.91,   270.65 },
	{ 51000.0, -0.0028, 66.94,    270.65 },
	{ 77000.0, -0.002,  3.96,     214.65 },
};

float or_altitude_to_pressure(float alt)
{
	const float g0 = 9.80665, M = 0.0289644, R_star = 8.3144598;   /* decoder.hpp:141-143 */
	int b = 6;
	for (int i = 0; i < 6; i++) {
		if (alt < isa[i + 1].hb) { b = i; break; }
	}
	const IsaLayer *l = &isa[b];
	/* The leading 1e-2 is a double literal in the reference (decoder.hpp:171,173), so the
	 * product is formed in double and narrowed on return. */
	if (l->Lb != 0) {
		const float base = (l->Tb + l->Lb * (alt - l->hb)) / l->Tb;
		const float expo = -(g0 * M) / (R_star * l->Lb);
		return (float)(1e-2 * l->Pb * powf(base, expo));
	}
	return (float)(1e-2 * l->Pb * expf(-g0 * M * (alt - l->hb) / (R_star * l->Tb)));
}

/* RS41 sensor conversions (public RS41 decoder formulas, SURVEY.md Appendix B.2 [RECALL]; the reference only
 * reads the results, /root/reference/src/decode/decoder.hpp:87-88).  Single precision, one operation per operator. */
float or_rs41_temp(uint32_t f, uint32_t f1, uint32_t f2, float rf1, float rf2, const float *co, const float *cal)
{
	const float ff = (float)f, ff1 = (float)f1, ff2 = (float)f2;
	const float gain = (ff2 - ff1) / (rf2 - rf1);
	const float ofs = (ff1 * rf2 - ff2 * rf1) / (ff2 - ff1);
	const float rc = ff / gain - ofs;
	const float r = rc * cal[0];
	return (co[0] + co[1] * r + co[2] * r * r + cal[1]) * (1.0f + cal[2]);
}

float or_rs41_rh(uint32_t f, uint32_t f1, uint32_t f2, float calh0, float T)
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

/* DFM thermistor temperature (public DFM-09 decoder formula, [RECALL]) */
float or_dfm_temp(float f, float f1, float f2)
{
	const float B0 = 3260.0f, T0 = 25.0f + 273.15f, R0 = 5.0e3f, Rf = 220.0e3f;
	if (f * f1 * f2 == 0.0f) return -273.15f;
	const float g = f2 / Rf;
	const float R = (f - f1) / g;
	if (!(R > 0.0f)) return -273.15f;
	return 1.0f / (1.0f / T0 + 1.0f / B0 * logf(R / R0)) - 273.15f;
}
