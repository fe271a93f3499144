// sd_math.h -- device-side arithmetic primitives shared by the demodulator and the channelizer front-end
// (SPEC).  Compiled with -ffp-contract=off: every fused op is an explicit fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Abramowitz & Stegun 4.4.47 with the discriminator gain 2/pi folded in (angles in quadrants:
// pi/2 -> 1, pi -> 2); same binary32 constants as the oracle
#define AT_A1  0.636534452f
#define AT_A3 -0.210275188f
#define AT_A5  0.114681326f
#define AT_A7 -0.0541973524f
#define AT_A9  0.0132640367f

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

// atan2q(y, x) in quadrants, [-2, 2]: SPEC 3.1.  Scalar on purpose: on gfx950 a v_pk_fma_f32 costs as
// much VALU time as two v_fmac_f32 (measured, tools/ubench/valu_rate.hip) and packing needs operand
// shuffles; the eight samples of a lane give the scheduler independent chains instead.
// max/min of the magnitudes are single VOP3 instructions with |.| source modifiers (equal to the
// oracle's integer max/min of the bit patterns for every non-NaN input); 1e-30 floors the divisor
// so atan2q(0,0) = 0 with no select.
__device__ __forceinline__ float sd_atan2q(float y, float x)
{
	const float tiny = 1.0e-30f;
	float mx, mn;
	asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(mx) : "v"(y), "v"(x), "v"(tiny));
	asm("v_min_f32 %0, |%1|, |%2|" : "=v"(mn) : "v"(y), "v"(x));
	const float r = mn * sd_recip(mx);
	const float sq = r * r;
	float p = __builtin_fmaf(sq, AT_A9, AT_A7);
	p = __builtin_fmaf(sq, p, AT_A5);
	p = __builtin_fmaf(sq, p, AT_A3);
	p = __builtin_fmaf(sq, p, AT_A1);
	p = p * r;
	// octant fix-ups, arithmetic form (cheap 2-operand ALU ops instead of compare+select pairs):
	//   |y|>|x|: p = 1 - p      x<0: p = 2 - p      sign from y
	const float dxy = __builtin_fabsf(x) - __builtin_fabsf(y);                                   // < 0 iff |y| > |x|
	const float s1 = __uint_as_float((uint32_t)((int32_t)__float_as_uint(dxy) >> 31) & 0x3F800000u);   // 1.0 or 0.0
	const float q1 = s1 - p;                                                                     // |q1| = 1-p or p
	const float s2 = __uint_as_float((uint32_t)((int32_t)__float_as_uint(x) >> 31) & 0x40000000u);     // 2.0 or 0.0
	const float q2 = s2 - __builtin_fabsf(q1);                                                   // |q2| = 2-|q1| or |q1|
	return __builtin_copysignf(q2, y);
}

// (cross, dot) of x1 * conj(x0): cross = fmaf(-x1, y0, y1*x0), dot = fmaf(y1, y0, x1*x0)  (SPEC 3.1)
__device__ __forceinline__ float sd_disc(float x1, float y1, float x0, float y0)
{
	const float cross = __builtin_fmaf(-x1, y0, y1 * x0);
	const float dot = __builtin_fmaf(y1, y0, x1 * x0);
	return sd_atan2q(cross, dot);
}

