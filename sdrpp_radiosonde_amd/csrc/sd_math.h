// sd_math.h -- device-side arithmetic primitives shared by the demodulator and the channelizer front-end
// (SPEC).  Compiled with -ffp-contract=off: every fused op is an explicit fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Three-term odd minimax polynomial for (2/pi) atan(r) on [-1, 1] (max error 4.5e-4 quadrant, 0.5 at r = 1), discriminator
// gain 2/pi folded in (angles in quadrants: pi/2 -> 1, pi -> 2); same binary32 constants as the oracle (SPEC 3.1)
#define AT_C1  0.6332877278327942f     // 0x3F221F25
#define AT_C3 -0.18171308934688568f    // 0xBE3A12FF
#define AT_C5  0.04842534288764f       // 0x3D4659A7
#define AT_FLOOR 6.9721523e-31f        // 0x0D624260

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Reciprocal by Newton-Raphson from an integer-subtract seed (SPEC 3.1): 1 integer op + 6 fma,
// identical sequence in the oracle.  Two lanes' worth at a time so the fmas issue as v_pk_fma_f32.
__device__ __forceinline__ f32x2 sd_recip2(f32x2 x)
{
	f32x2 r;
	r.x = __uint_as_float(0x7EF311C7u - __float_as_uint(x.x));
	r.y = __uint_as_float(0x7EF311C7u - __float_as_uint(x.y));
	const f32x2 one = {1.0f, 1.0f};
	f32x2 e = pk_fma(-x, r, one);
	r = pk_fma(r, e, r);
	e = pk_fma(-x, r, one);
	r = pk_fma(r, e, r);
	e = pk_fma(-x, r, one);
	r = pk_fma(r, e, r);
	return r;
}
__device__ __forceinline__ float sd_recip(float x)
{
	float r = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
	float e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(r, e, r);
	e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(r, e, r);
	e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(r, e, r);
	return r;
}

// atan2q(y, x) in quadrants, [-2, 2]: SPEC 3.1 (round 3).  The consumer is a low-pass FIR and a hard slicer: 2e-3 rad is
// plenty.  r = (|x| - |y|) / (|x| + |y|) covers the whole first quadrant with one polynomial (angle = 1/2 - f(r), no
// |y| > |x| swap), the reciprocal takes ONE Newton step from the integer-subtract seed (relative error < 0.26 %, from below,
// so |r| <= 1): 15 VALU operations, all 2-operand or inline-constant forms, against 24 for the round-2 version (A&S 4.4.47
// with three Newton steps and two octant fix-ups); the M10 class was 85 % VALU-issue bound on that (profiles/r2_v4_class_counters.csv).
// Scalar on purpose: on gfx950 a v_pk_fma_f32 costs as much VALU time as two v_fmac_f32 (tools/ubench/valu_rate.hip).
__device__ __forceinline__ float sd_atan2q(float y, float x)
{
	// atan2q(0, 0) = 0 (round 4): the divisor is floored at the float for which seed + one Newton step give r = d * rc = exactly 1 at
	// (0, 0), where the polynomial is exactly 1/2; the numerator comes from the floored sum, d = s - 2|y| (SPEC 3.1, oracle or_atan2)
	const float s = __builtin_fmaxf(__builtin_fabsf(x) + __builtin_fabsf(y), AT_FLOOR);
	const float d = __builtin_fmaf(-2.0f, __builtin_fabsf(y), s);
	float rc = __uint_as_float(0x7EF311C7u - __float_as_uint(s));
	const float e = __builtin_fmaf(-s, rc, 1.0f);
	rc = __builtin_fmaf(rc, e, rc);
	const float r = d * rc;
	const float t = r * r;
	float p = __builtin_fmaf(t, AT_C5, AT_C3);
	p = __builtin_fmaf(t, p, AT_C1);
	const float q = __builtin_fmaf(-p, r, 0.5f);                                                   // angle of (|x|, |y|), [0, 1]
	const float s2 = __uint_as_float((uint32_t)((int32_t)__float_as_uint(x) >> 31) & 0x40000000u);     // 2.0 or 0.0
	const float q2 = s2 - q;                                                                       // x < 0: 2 - q
	return __builtin_copysignf(q2, y);
}

// discriminator behind the filter bank (SPEC 3.5, round 4): the wrapped difference of two phases given in quadrants, in [-2, 2]
__device__ __forceinline__ float sd_phase_diff(float ph, float prev)
{
	const float t = ph - prev;
	return __builtin_fmaf(-4.0f, __builtin_rintf(0.25f * t), t);
}

// SPEC 3.0e: the rotation the AFC state u stands for, (4 / pi) atan(u) quadrants per internal sample (the phasor (1 - u^2, 2u) of
// sd_disc_rot turns by 2 atan u), as an odd polynomial on |u| <= SD_AFC_MAX = 0.8: max error 2.3e-5 quadrant; 5 operations
#define SD_ROT_C1  1.2729679f
#define SD_ROT_C3 -0.41814741f
#define SD_ROT_C5  0.21412420f
#define SD_ROT_C7 -0.073257752f
__device__ __forceinline__ float sd_afc_rot(float u)
{
	const float t = u * u;
	float p = __builtin_fmaf(t, SD_ROT_C7, SD_ROT_C5);
	p = __builtin_fmaf(t, p, SD_ROT_C3);
	p = __builtin_fmaf(t, p, SD_ROT_C1);
	return u * p;
}

// AFC (SPEC 3.0b): the product x1 conj(x0) turned back by the phasor (c, sn) = (1 - u^2, 2u) before the arctangent; the phasor's
// length does not matter to an arctangent
__device__ __forceinline__ float sd_disc_rot(float x1, float y1, float x0, float y0, float c, float sn)
{
	const float cross = __builtin_fmaf(-x1, y0, y1 * x0);
	const float dot = __builtin_fmaf(y1, y0, x1 * x0);
	const float cr = __builtin_fmaf(-dot, sn, cross * c);
	const float dr = __builtin_fmaf(cross, sn, dot * c);
	return sd_atan2q(cr, dr);
}

// (cross, dot) of x1 * conj(x0): cross = fmaf(-x1, y0, y1*x0), dot = fmaf(y1, y0, x1*x0)  (SPEC 3.1)
__device__ __forceinline__ float sd_disc(float x1, float y1, float x0, float y0)
{
	const float cross = __builtin_fmaf(-x1, y0, y1 * x0);
	const float dot = __builtin_fmaf(y1, y0, x1 * x0);
	return sd_atan2q(cross, dot);
}

