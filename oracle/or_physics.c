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

/* ---- independent double-precision restatements of the sensor conversions added in round 2 (product: single precision,
 * csrc/parse.cpp).  Written in a different algebraic form on purpose (Horner instead of running powers, exp/log
 * rearranged): tests compare within a stated tolerance, not bit for bit, so that they are not a self-comparison.
 * Formulas: public decoders' models [RECALL]; the reference only reads fragment.temp/.rh/.pressure/.o3_mpa
 * (/root/reference/src/decode/decoder.hpp:87-89,104). */
double or_rs41_pressure_d(uint32_t f, uint32_t f1, uint32_t f2, double tpress, const float *cfP)
{
	if (f1 == f2 || f1 == f) return 0.0;
	const double fp = ((double)f - (double)f1) / ((double)f2 - (double)f1);
	const double a0 = (double)cfP[24] / fp;
	double p = 0.0;
	for (int j = 5; j >= 0; j--) {                 /* Horner in a0, inner Horner in the sensor temperature */
		double row = 0.0;
		for (int k = 3; k >= 0; k--) row = row * tpress + (double)cfP[4 * j + k];
		p = p * a0 + row;
	}
	return p;
}

double or_ozone_mpa_d(double cell_ua, double tpump_c)
{
	/* P[mPa] = 0.043085 * T[K] * I[uA] / flow[ml/s], flow = 100 ml / 28 s */
	const double flow = 100.0 / 28.0;
	const double p = 0.043085 * (tpump_c + 273.15) * cell_ua / flow;
	return p > 0.0 ? p : 0.0;
}

double or_m10_temp_d(unsigned scale, unsigned adc)
{
	static const double Rs[3] = { 12.1e3, 36.5e3, 475.0e3 }, Gp[3] = { 0.0, 1.0 / 330.0e3, 1.0 / 2000.0e3 };
	if (scale > 2 || adc == 0 || adc >= 4095) return -273.15;
	/* divider: Vout/Vcc = adc/4095 across Rs, the thermistor in parallel with Rp on top.  In conductances:
	 * 1/R = (Vcc - Vout)/(Vout Rs) - 1/Rp */
	const double g = ((4095.0 - (double)adc) / (double)adc) / Rs[scale] - Gp[scale];
	if (!(g > 0.0)) return -273.15;
	const double l = -log(g);
	return 1.0 / (1.07303516e-03 + l * (2.41296733e-04 + l * (2.26744154e-06 + l * 6.52855181e-08))) - 273.15;
}

double or_m10_rh_d(uint32_t cap_sensor, uint32_t cap_ref, double T)
{
	if (cap_ref == 0) return -1.0;
	double rh = ((double)cap_sensor - 0.8955 * (double)cap_ref) / (0.002 * (double)cap_ref) + 0.03 * (20.0 - T);
	return rh < 0.0 ? 0.0 : (rh > 100.0 ? 100.0 : rh);
}

double or_m20_temp_d(unsigned adc)
{
	if (adc == 0 || adc >= 4095) return -273.15;
	const double lr = log(22.1e3 / 15.0e3) + log((double)adc) - log(4095.0 - (double)adc);
	return 3450.0 * 273.15 / (3450.0 + 273.15 * lr) - 273.15;
}

double or_ims100_temp_d(uint32_t f, double c0, double c1, double c2)
{
	return c0 + (double)f * (c1 + (double)f * c2);
}
