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
