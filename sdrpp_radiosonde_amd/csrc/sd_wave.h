// sd_wave.h -- wave-level primitives shared by the demodulator kernels (demod_kernel.hip: one workgroup per channel;
// bins_kernel.hip: one wave per channelizer bin): DPP reductions and shifts, the clamp, the polyphase interpolator of SPEC 3.2.
#pragma once
#include <hip/hip_runtime.h>
#include "sonde_dev.h"
#include "sd_math.h"
#include "sd_rsdec.h"

#define SD_LH      64     // samples of history kept in front of the tile in LDS
// the GF(2^8) tables of the RS41 FEC epilogue, one copy per workgroup
struct EpiTabs { FramerTabs tabs; alignas(16) uint32_t swar[RS_R * 8]; };

// min(max(v, lo), hi) as ONE v_med3_f32 (equal for every non-NaN v; the fminf / fmaxf pair costs a canonicalising v_max_f32 more)
__device__ __forceinline__ float sd_clamp(float v, float lo, float hi)
{
	return __builtin_amdgcn_fmed3f(v, lo, hi);
}

// Integer wave reduction with DPP (VALU, no LDS crossbar): rows of 16, then row broadcasts; the total
// lands in lane 63.  Integer addition is associative, so the tree shape is free (SPEC 3).
__device__ __forceinline__ int wave_sum(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
	v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
	v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
	v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1,3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2,3
	return __builtin_amdgcn_readlane(v, 63);
}

// v of the lane below, lane 0: `first` (DPP wave_shr:1, GFX9; lanes without a source keep the old value)
__device__ __forceinline__ float sd_wave_shr1(float v, float first)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}

// THE HALF-WAVE SYMBOL MAPPING (round 5) of the classes at 2.5 samples per symbol (8-tap rows).  Lanes <-> consecutive symbols put
// consecutive lanes 2.5 dwords apart in the tile: of a ds_read_b32's 32 lanes, lanes 13 apart meet on a bank (32.5 dwords), the two tap
// rows in use (phases 0 and 1/2) lie 576 dwords apart = on the same banks: 23 % (RS41 class) to 38 % (M10 class) of the LDS pipe's
// cycles were bank conflicts (profiles/r3_class_counters.csv).  Instead lanes 0-31 take the EVEN symbols of the wave's 64 and lanes
// 32-63 the odd ones: within each half (the unit a ds_read_b32 is served in) the addresses are 5 dwords apart -- co-prime with the 32
// banks -- and share one tap row.  Nothing arithmetic changes: the Gardner term of symbol k still takes y[k - 1] (now from the other
// half-wave: a DPP shift and v_permlane32_swap, no LDS), the integer sums do not care, and the bits are put back in symbol order
// before the ballot (one ds_bpermute_b32).
__device__ __forceinline__ int sd_hw_symbol(int lane) { return 2 * (lane & 31) + (lane >> 5); }      // the symbol (of the wave's 64) a lane owns
// y of the symbol before this lane's: lane l < 32 (symbol 2l) <- lane 32 + l - 1 (symbol 2l - 1; l = 0: not used), lane l >= 32 <- lane l - 32
__device__ __forceinline__ float sd_hw_prev(float y, int lane)
{
	const int sh = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x138, 0xF, 0xF, false);      // sh[l] = y[l - 1]
	const auto r = __builtin_amdgcn_permlane32_swap((unsigned)sh, __builtin_bit_cast(unsigned, y), false, false);
	// v_permlane32_swap exchanges the first operand's lanes 32-63 with the second's lanes 0-31 (tools/ubench/permlane_probe.hip):
	// r[0] = (sh[0..31], y[0..31]), r[1] = (sh[32..63], y[32..63])
	return __builtin_bit_cast(float, lane < 32 ? r[1] : r[0]);
}
// a per-lane flag (0 / 1) from the half-wave mapping back into symbol order: lane p receives the flag of the lane that owns symbol p
__device__ __forceinline__ int sd_hw_unpermute(int flag, int lane)
{
	return __builtin_amdgcn_ds_bpermute(4 * ((lane >> 1) + 32 * (lane & 1)), flag);
}

// y(pos) = (sum_{j even} H[p][j] d[n+16-j]) + (sum_{j odd} H[p][j] d[n+16-j]), each an fmaf chain with j
// ascending (SPEC 3.2): one v_pk_fma_f32 per tap pair.  The tap rows are stored pair-swapped
// (T[2i] = H[2i+1], T[2i+1] = H[2i]) so that they line up with the (d[x], d[x+1]) pairs.
// rel = pos relative to A[0], Q16.
template <int NT>   // taps in use
__device__ __forceinline__ float interp(const float *A, const float *taps, uint32_t rel)
{
	const uint32_t top = (rel >> 16) + NT / 2;                          // buffer index of d for j = 0
	const float *h = taps + ((rel >> 11) & (SD_NPHASE - 1)) * SD_TAPS_LD;
	// pair i holds (d[top-1-2i], d[top-2i]): pairs at any alignment, read as two dwords each (ds_read2_b32)
	const float *lo = A + (top - (NT - 1));
	f32x2 acc = {0.0f, 0.0f};                                           // (odd chain, even chain)
#pragma unroll
	for (int q = 0; q < NT / 4; q++) {
		const float4 hv = *reinterpret_cast<const float4 *>(h + 4 * q);
		const float2 v0 = make_float2(lo[(NT - 2) - 4 * q], lo[(NT - 1) - 4 * q]);
		const float2 v1 = make_float2(lo[(NT - 4) - 4 * q], lo[(NT - 3) - 4 * q]);
		const f32x2 h0 = {hv.x, hv.y}, h1 = {hv.z, hv.w};
		const f32x2 d0 = {v0.x, v0.y}, d1 = {v1.x, v1.y};
		acc = pk_fma(h0, d0, acc);
		acc = pk_fma(h1, d1, acc);
	}
	return acc.y + acc.x;
}

