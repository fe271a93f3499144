// demod_kernel.hip -- kernel A: IQ (or discriminator samples) -> hard bits, one workgroup per channel.
//
// Stages, all inside one launch so that IQ is read from HBM exactly once and only bits
// (1/640 of the input bytes) are written:
//   K1  FM quadrature discriminator      (SDR++ dsp::demod::FM<float>, /root/reference/src/main.cpp:57)
//   K2  polyphase low-pass FIR, evaluated only at the timing loop's sampling instants
//   K3  Gardner timing-error detector + PI loop filter, updated once per round (<= 256 symbols)
//       and hard slicer -> bit ring in HBM
//   K4  RS41 channels: frame-sync correlator over the newest bits (sd_rs41.h), on a round wave that would
//       otherwise spin while the lead wave runs the loop filter -> frame descriptors
//   K5/K6  RS41 channels, epilogue: the frames completed in this submit are de-whitened and RS(255,231)-decoded
//       by the eight waves of the workgroup, one frame per wave (sd_rsdec.h) -> frame records in HBM.
//       One launch per step instead of three (round 1: sync 12.8 us + FEC 38.2 us + two launch boundaries).
// (K2/K3 stand where sondedump's gfsk_demod sits behind X_decode, /root/reference/src/decode/decoder.hpp:22,61.)
//
// Bit-exactness contract: compiled with -ffp-contract=off, every fused op is an
// explicit __builtin_fmaf; each lane's FIR is a fixed-order fmaf chain; every cross-lane
// reduction (timing error, slicer level statistics) is an integer sum, so lane order is irrelevant.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "sonde_dev.h"

#include "sd_math.h"
#include "sd_wave.h"
#include "sd_rs41.h"
#include "sd_rsdec.h"
#include "sd_fixed.h"
#include "launch.h"

typedef float sd_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sd_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned sd_u32x4 __attribute__((ext_vector_type(4)));
typedef short sd_i16x2 __attribute__((ext_vector_type(2)));
// SD_IN_IQ8: two complex samples of 8-bit integers (I0 Q0 I1 Q1) -> the float4 of the float path (exact, no scaling)
static __device__ __forceinline__ float4 sd_cs8_f4(uint32_t q)
{
	return make_float4((float)(int8_t)(q & 0xffu), (float)(int8_t)((q >> 8) & 0xffu), (float)(int8_t)((q >> 16) & 0xffu), (float)((int32_t)q >> 24));
}
// SPEC 3.0d: the two half-sums of a group of four, (samples 0 + 1, samples 2 + 3), each exact in integers: (P0.re, P0.im, P1.re, P1.im)
static __device__ __forceinline__ float4 sd_cs16_halves(uint4 q)
{
	const sd_i16x2 lo = {1, 0}, hi = {0, 1};
	int a = 0, b = 0, c = 0, d = 0;
	a = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.x), lo, a, false); b = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.x), hi, b, false);
	a = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.y), lo, a, false); b = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.y), hi, b, false);
	c = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.z), lo, c, false); d = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.z), hi, d, false);
	c = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.w), lo, c, false); d = __builtin_amdgcn_sdot2(__builtin_bit_cast(sd_i16x2, q.w), hi, d, false);
	return make_float4((float)a, (float)b, (float)c, (float)d);
}
static __device__ __forceinline__ float4 sd_cs8_halves(uint2 q)
{
	int a = 0, b = 0, c = 0, d = 0;
	a = __builtin_amdgcn_sdot4((int)q.x, 0x00010001, a, false); b = __builtin_amdgcn_sdot4((int)q.x, 0x01000100, b, false);
	c = __builtin_amdgcn_sdot4((int)q.y, 0x00010001, c, false); d = __builtin_amdgcn_sdot4((int)q.y, 0x01000100, d, false);
	return make_float4((float)a, (float)b, (float)c, (float)d);
}
static __device__ __forceinline__ float4 sd_cs16_halves_uniform(uint4 q)
{
	return make_float4((float)((int)(int16_t)(q.x & 0xffffu) + (int)(int16_t)(q.y & 0xffffu)), (float)(((int32_t)q.x >> 16) + ((int32_t)q.y >> 16)),
	                   (float)((int)(int16_t)(q.z & 0xffffu) + (int)(int16_t)(q.w & 0xffffu)), (float)(((int32_t)q.z >> 16) + ((int32_t)q.w >> 16)));
}
static __device__ __forceinline__ float4 sd_cs8_halves_uniform(uint2 q)
{
	return make_float4((float)((int)(int8_t)(q.x & 0xffu) + (int)(int8_t)((q.x >> 16) & 0xffu)), (float)((int)(int8_t)((q.x >> 8) & 0xffu) + ((int32_t)q.x >> 24)),
	                   (float)((int)(int8_t)(q.y & 0xffu) + (int)(int8_t)((q.y >> 16) & 0xffu)), (float)((int)(int8_t)((q.y >> 8) & 0xffu) + ((int32_t)q.y >> 24)));
}
// SPEC 3.0d, the carrier-following boxcar: z = P0 + R P1, R = (rr, ri) = (1 - u^2 / 2, -u): the second half of a group turned back by
// about atan u before it is added (oracle or_demod_feed)
static __device__ __forceinline__ float2 sd_follow(float p0r, float p0i, float p1r, float p1i, float rr, float ri)
{
	return make_float2(p0r + __builtin_fmaf(-p1i, ri, p1r * rr), p0i + __builtin_fmaf(p1r, ri, p1i * rr));
}
// SD_IN_IQ16: two complex samples of 16-bit integers (I0 Q0 I1 Q1, little endian) -> the float4 the float path would have loaded
// (int16 -> float is exact; no scaling: the discriminator's output does not depend on the amplitude)
static __device__ __forceinline__ float4 sd_cs16_f4(uint2 q)
{
	return make_float4((float)(int16_t)(q.x & 0xffffu), (float)((int32_t)q.x >> 16), (float)(int16_t)(q.y & 0xffffu), (float)((int32_t)q.y >> 16));
}

// ---- carried per-channel data (state, history, bit-ring words, framer state, frame count).  In a time-sliced launch (launch.h SdSlice)
// it crosses from one workgroup to the next of the same channel INSIDE the launch, possibly from one XCD's L2 to another's.  Fencing that
// hand-over at agent scope costs an L2-wide write-back per release and an invalidate per acquire -- measured: 8 fences per workgroup made a
// sliced launch 3-10 x slower (profiles/r6_notes.md).  So the data itself is accessed coherently instead: agent-scope relaxed atomic loads
// and stores (the sc1 bit: served at the device's coherence point, never from a stale L1 / L2 line; stores write through), the few hundred
// bytes per workgroup that they are.  ONLY sliced launches do it (the `coh` argument, launch-uniform): a write-through store is
// acknowledged later than a write-back one, and the round wave that appends the bits waits for its stores at every tile's barrier --
// with coherent stores everywhere the unsliced headline lost 2.5 % and 8192 x 24 tiles 4.8 % (interleaved A/B against round 5's
// library on one box, profiles/r6_ab_r5_vs_r6.txt).
template <typename T> static __device__ __forceinline__ T sd_ld_coh(const T *p, bool coh = true) { return coh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
template <typename T> static __device__ __forceinline__ void sd_st_coh(T *p, T v, bool coh = true) { if (coh) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; }
static __device__ __forceinline__ float sd_ld_coh_f(const float *p, bool coh) { return coh ? __builtin_bit_cast(float, sd_ld_coh(reinterpret_cast<const uint32_t *>(p))) : *p; }
static __device__ __forceinline__ void sd_st_coh_f(float *p, float v, bool coh) { if (coh) sd_st_coh(reinterpret_cast<uint32_t *>(p), __builtin_bit_cast(uint32_t, v)); else *p = v; }
template <typename S> static __device__ __forceinline__ S sd_ld_coh_struct(const S *p, bool coh)
{
	if (!coh) return *p;
	static_assert(sizeof(S) % 8 == 0, "whole qwords");
	unsigned long long w[sizeof(S) / 8];
#pragma unroll
	for (unsigned i = 0; i < sizeof(S) / 8; i++) w[i] = sd_ld_coh(reinterpret_cast<const unsigned long long *>(p) + i);
	S s;
	__builtin_memcpy(&s, w, sizeof(S));
	return s;
}
template <typename S> static __device__ __forceinline__ void sd_st_coh_struct(S *p, const S &s, bool coh)
{
	if (!coh) { *p = s; return; }
	static_assert(sizeof(S) % 8 == 0, "whole qwords");
	unsigned long long w[sizeof(S) / 8];
	__builtin_memcpy(w, &s, sizeof(S));
#pragma unroll
	for (unsigned i = 0; i < sizeof(S) / 8; i++) sd_st_coh(reinterpret_cast<unsigned long long *>(p) + i, w[i]);
}

#define SD_BUF     (SD_LH + SD_TILE + 4)
#define SD_WGT     512    // workgroup: waves 0-3 run the timing-loop rounds, waves 4-7 the discriminator
// K4 (sync search) runs on round wave 3; on a discriminator wave it measured equal for RS41 and 40 % slower for DFM (profiles/r2_notes.md)

// LDS: A = the discriminator samples of [tile_start - 64, tile_end), double-buffered over tiles: while the round waves read tile i
// from buffer i & 1, the discriminator waves fill buffer (i + 1) & 1 with tile i + 1.  One barrier per tile.  (Rounds 1-2 kept
// a second copy of every tile shifted by one sample so that every FIR operand pair was one aligned 8-byte read; one copy and
// two-dword reads measured faster, profiles/r3_notes.md.)
// X = two more buffers of the same size that never hold a tile: X[1] keeps the GF(2^8) tables of the FEC epilogue for the whole
// launch (from SD_EPI_TAB_OFF), X[0] and X[1] give the channelizer-input path its wave-private scratch, and A[0], A[1], X[0]
// together are the eight per-wave work areas of the epilogue once the tiles are dead.  The workgroup's 39.7 KB are what four
// workgroups per CU can have (160 KB): nothing to gain from a smaller struct, 8 waves per SIMD is the cap either way.
struct DemodLds {
	float A[2][SD_BUF];
	float X[2][SD_BUF];
	float taps[SD_NPHASE * SD_TAPS_LD];     // rows padded to 36 floats: 16-byte aligned ds_read_b128
	int4 red[2][4];                         // [round parity][wave]: (E, S1, S0, C1)
	int2 red2[2][4];                        // AFSK streams (SPEC 3.6b): (SY, SM), the eye opening at the on-time / the mid-symbol instants
	uint32_t chunk[2][18];                  // [round parity]: the round's bits, one ballot per wave (and half), zero-padded
	uint32_t partial[2];                    // bits already in the ring word that wpos points into (ping-pong)
	float iq_last[2];
	float afc_u[4];                         // SPEC 3.0b: AFC state for the discriminator of tile T at [T & 3] (written by the lead wave three tiles earlier)
	float afc_dq[4];                        // SPEC 3.0e: by how much tile T's rotation exceeds tile T - 1's, quadrants per sample: sd_afc_rot(u[T]) - sd_afc_rot(u[T - 1])
	// what the lead round wave (wave 0) computes once per round and the other round waves pick up:
	// the PI loop filter runs on one wave instead of four (it is ~35 % of a round wave's VALU work)
	struct { long long t_next; int period; float bias; int K; unsigned flag; unsigned long long wpos; } pub;
	uint32_t mirror[SD_MIRROR_WORDS];       // the newest 2048 bits of the bit ring, for the in-kernel sync search (K4)
	SdSyncRun k4;                           // K4's state between steps (wave 3 only)
	SdFecJob fec;                           // the in-loop decoder of clean RS41 frames (wave 2 only, sd_rsdec.h)
	uint32_t nout0;                         // time slices: frames the channel's earlier segments listed in this submit (0 in an unsliced launch)
};

// samples (i, i+1) of a tile, i even, into buffer b
__device__ __forceinline__ void store_pair(DemodLds &s, int b, uint32_t i, float d0, float d1)
{
	*reinterpret_cast<float2 *>(&s.A[b][SD_LH + i]) = make_float2(d0, d1);
}

// one sample i of a tile into buffer b
__device__ __forceinline__ void store_one(DemodLds &s, int b, uint32_t i, float d0)
{
	s.A[b][SD_LH + i] = d0;
}

// FEC epilogue (RS41 channels): the GF tables sit, for the whole launch, in X[1] (placed as if behind the part of a buffer that a decimated
// tile uses (decimation 4 or 2: <= 64 + 1024 + 4 floats of 2116), the per-wave work areas alias the tile buffers,
// which are dead by then.
#define SD_EPI_TAB_OFF 1100                 // floats into X[1]
static_assert(SD_LH + SD_TILE / 2 + 4 <= SD_EPI_TAB_OFF, "the tables must stay clear of a 2:1 tile");
static_assert(sizeof(EpiTabs) <= (SD_BUF - SD_EPI_TAB_OFF) * sizeof(float), "GF tables do not fit behind the tile");
static_assert(8 * sizeof(FramerLds) <= (2 * SD_BUF + SD_BUF) * sizeof(float), "per-wave FEC work areas do not fit into the tile buffers");
static_assert(8 * sizeof(FixedLds) <= (2 * SD_BUF + SD_BUF) * sizeof(float), "per-wave work areas of the fixed-length decoders do not fit into the tile buffers");

// The epilogue's RS41 frame decoder as a REAL call (round 6): their own register allocation instead of the kernel's 64-register squeeze -- the
// inlined RS41 decoder was what put three hoisted LDS addresses into scratch (16 B per lane: the only private memory of this kernel;
// VERDICT r5 item 7) -- and one copy of the code per instantiation less: 8192 x 24 tiles -3.6 %, mixed 4096 x 24 -1.8 %, the headline
// -0.3 % (interleaved A/B, profiles/r6_ab_epicall.txt).
static __device__ __attribute__((noinline)) void sd_rs41_decode_frame_call(const FramerTabs &tabs, FramerLds &s, const uint32_t *__restrict__ swar_row,
	const uint32_t *__restrict__ ring, uint32_t mask, unsigned long long d0, unsigned long long d1, SondeFrame *__restrict__ fr, uint32_t ch, int lane)
{
	GfSwar swar;
	swar.a_lo = swar_row[0]; swar.a_hi = swar_row[1]; swar.b_lo = swar_row[2]; swar.b_hi = swar_row[3]; swar.c = swar_row[4];
	SdFrameDesc d;
	d.fstart = sd_uniform64(d0);
	d.flen = __builtin_amdgcn_readfirstlane((int)(uint32_t)d1);
	d.inv = __builtin_amdgcn_readfirstlane((int)(d1 >> 32));
	sd_rs41_decode_frame<true>(tabs, s, swar, ring, mask, d, fr, ch, lane);
}

// ---------------------------------------------------------------- the kernel
// Wave specialisation: the discriminator (K1) is pure per-sample ALU work, the rounds (K2/K3) are a
// latency-bound chain (loop update -> FIR reads -> reduction).  Running them as different waves of the
// same workgroup doubles the waves per SIMD and lets the hardware overlap them; the only hand-over is
// the double-buffered LDS tile, one s_barrier per tile (two for sondes with > 256 symbols per tile).
// LIST: work on the channels of `chlist` (mixed batches); the plain instantiation ignores it.  DEC, NT: the decimation
// factor and the taps per filter row of every channel of this launch (the host launches once per class), so that the
// discriminator and FIR variants do not share one register allocation.  Classes in use: (4, 8) RS41 / DFM / iMS-100 / MRZ-N1,
// (2, 8) M10, (2, 16) and (1, 16) the same two groups under SONDE_FLAG_WIDE, (1, 16) also the 6 kS/s AFSK streams.
// IN: what `in` holds per channel: SD_IN_REAL 48 kS/s discriminator samples, SD_IN_IQ 48 kS/s complex samples, SD_IN_IQ16 / SD_IN_IQ8 the
// same as 16- / 8-bit integer pairs.  (Channelizer bins have their own kernel: bins_kernel.hip, one wave per bin.)
// The kernel's BODY, a device function so that two classes can share one launch (sd_demod_mixed_kernel below): bidx = the workgroup's
// index in ITS launch (or in its class's share of a mixed launch), grid_wg = the workgroups that share the GPU with it (the in-loop FEC's
// one-residency test), s = the workgroup's LDS (declared once, by the __global__ wrapper).
// SLICED: the time-sliced instantiation (launch.h SdSlice): a compile-time flag, so that an unsliced launch carries none of it -- with a
// launch-uniform runtime flag the one-residency headline was 2.2 % slower than round 5's kernel (profiles/r6_ab_r5_vs_r6.txt).
template <int IN, bool LIST, int DEC, int NT, bool SLICED = false>
__device__ __forceinline__ void sd_demod_body(DemodLds &s, const uint32_t bidx, const uint32_t grid_wg,
	const float *__restrict__ in, size_t ch_stride, int n_tiles_all,
	SdChanState *__restrict__ states, float *__restrict__ hist,
	uint32_t *__restrict__ bitring, uint32_t ring_words,
	const float *__restrict__ taps_all, const SdModem *__restrict__ modems,
	const uint32_t *__restrict__ chlist, int compact_in, const SdFramerOut *__restrict__ fo, int utype, const SdSlice sl)
{
	// integer IQ rows: IQ16 = int16 pairs (SD_IN_IQ16) OR int8 pairs (SD_IN_IQ8): one code path, IQ8 picks the element size
	constexpr bool IQ8 = IN == SD_IN_IQ8, IQ16 = IN == SD_IN_IQ16 || IQ8, IS_IQ = IN == SD_IN_IQ || IQ16;
	constexpr bool IQ16D4 = IQ16 && DEC == 4;                                 // one load of four samples = one decimated sample each (16 bytes; int8: 8)
	// what one lane holds per load: two input samples (IQ; four in the integer 4:1 classes), four (real)
	using LoadT = typename std::conditional<IQ8, typename std::conditional<IQ16D4, uint2, uint32_t>::type,
	              typename std::conditional<IQ16, typename std::conditional<IQ16D4, uint4, uint2>::type, float4>::type>::type;

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const bool is_k = wave >= 4;           // wave-uniform role
	const int t = tid & 255;               // index inside the role group
	const int rwave = wave & 3;
	// time slices (launch.h SdSlice): block b works on segment b / n_wg of list entry b % n_wg; a segment is a submit of its own to
	// everything below (its tiles start at `src`, n_tiles of them; the state comes from and goes back to HBM)
	constexpr bool sliced = SLICED;
	const int seg = sliced ? (int)(bidx / sl.n_wg) : 0;
	const uint32_t wgi = sliced ? bidx - (uint32_t)seg * sl.n_wg : bidx;
	const int n_tiles = sliced ? min(sl.seg_tiles, n_tiles_all - seg * sl.seg_tiles) : n_tiles_all;
	const uint32_t ch = LIST ? chlist[wgi] : wgi;                      // channel (state, bit ring)
	const uint32_t row = (LIST && compact_in) ? wgi : ch;              // row of `in`

	// ---- discriminator waves: the first two tiles' loads go out before anything else (see the prologue note below)
	constexpr int NLD = (IS_IQ && !IQ16D4) ? 4 : 2;     // loads per thread per tile (float4; SD_IN_IQ16: 8 bytes, the same two samples, or 16 bytes, four)
	constexpr int TILE_F4 = ((IS_IQ && !IQ16D4) ? 2 : 1) * SD_TILE / 4;      // load units (LoadT) per tile
	// (ch_stride counts samples: 8 bytes each for complex64, 4 for real input and for 16-bit IQ)
	const LoadT *src = reinterpret_cast<const LoadT *>(reinterpret_cast<const char *>(in) + (size_t)row * ch_stride * (IQ8 ? 2 : ((IS_IQ && !IQ16) ? 8 : 4)))
	                   + (sliced ? (size_t)seg * (size_t)sl.seg_tiles * TILE_F4 : 0);
	LoadT va[NLD], vb[NLD];                // two register sets: tiles are prefetched two phases ahead
	// Work split: wave kw of the four owns 256 consecutive float4s of the tile, load r covers 64 of them, so
	// every load instruction is one contiguous 1 KB and the predecessor sample of lane 0 at r > 0 is lane 63
	// of the same wave at r - 1 (no extra load); only each wave's very first sample needs the float4 before it.
	const int kw = wave & 3;
	auto f4_index = [&](int r) { return SD_WG / 4 * (NLD * kw + r) + lane; };      // float4 index inside the tile
	auto load_vec = [&](int tile, LoadT (&v)[NLD]) {
#pragma unroll
		for (int r = 0; r < NLD; r++) {                    // read-once data: streaming (nontemporal) loads
			if constexpr (IQ8 && IQ16D4) {
				const sd_u32x2 q = __builtin_nontemporal_load(reinterpret_cast<const sd_u32x2 *>(src + (size_t)tile * TILE_F4 + f4_index(r)));
				v[r] = make_uint2(q.x, q.y);
			} else if constexpr (IQ8) {
				v[r] = __builtin_nontemporal_load(src + (size_t)tile * TILE_F4 + f4_index(r));
			} else if constexpr (IQ16D4) {
				const sd_u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const sd_u32x4 *>(src + (size_t)tile * TILE_F4 + f4_index(r)));
				v[r] = make_uint4(q.x, q.y, q.z, q.w);
			} else if constexpr (IQ16) {
				const sd_u32x2 q = __builtin_nontemporal_load(reinterpret_cast<const sd_u32x2 *>(src + (size_t)tile * TILE_F4 + f4_index(r)));
				v[r] = make_uint2(q.x, q.y);
			} else {
				const sd_f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const sd_f32x4 *>(src + (size_t)tile * TILE_F4 + f4_index(r)));
				v[r] = make_float4(q.x, q.y, q.z, q.w);
			}
		}
	};
	if (is_k) {
		load_vec(0, va);
		if (n_tiles > 1) load_vec(1, vb);
	}

	// time slices: segment s > 0 starts where segment s - 1 of the same channel stopped: wait until that workgroup has published its
	// state (its block index is lower: it was dispatched earlier).  One lane polls, the rest wait at the barrier.
	if (sliced && seg > 0) {
		if (tid == 0) {
			const uint32_t want = sl.seg_base + (uint32_t)seg;
			unsigned spins = 0;
			bool ok = true;
			while (__hip_atomic_load(&sl.prog[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
				__builtin_amdgcn_s_sleep(8);
				if ((++spins & 1023u) == 0 && (spins >= (1u << 20) || __hip_atomic_load(&sl.prog[sl.err_index], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { ok = false; break; }
			}
			if (!ok) __hip_atomic_store(&sl.prog[sl.err_index], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s.pub.flag = ok ? 0u : 0xDEADu;
		}
		__syncthreads();
		if (s.pub.flag == 0xDEADu) return;         // (workgroup-uniform; the host finds the error word: never a hang)
		__syncthreads();                           // (pub.flag is rewritten by the prologue below)
		// no acquire fence: everything the predecessor handed over is read through coherent loads (sd_ld_coh) below
	}
	SdChanState st = sd_ld_coh_struct(states + ch, sliced);
	// utype >= 0: every channel of this launch is of that sonde type (the host knows: one-type batches, per-type launch units), so
	// the taps and modem parameters do not have to wait for the state
	int stype;
	if (utype >= 0) stype = utype;
	else stype = __builtin_amdgcn_readfirstlane(st.type);
	const SdModem md = modems[stype];
	constexpr bool AF = IN == SD_IN_REAL && DEC == 1;                  // the instantiation the 6 kS/s AFSK streams run through
	const bool afsk = AF && (stype == SONDE_IMET4 || stype == SONDE_C50);
	const int rounds = md.rounds;          // sub-phases (= barriers) per tile: 1, or 2 for the SRS-C50 6 kS/s stream
	constexpr int IT = SD_TILE / DEC;      // internal samples per input tile (= md.itile)
	constexpr bool dec2 = DEC == 2, dec4 = DEC == 4;
	// ---- prologue, round waves only (tid < 256): taps, carried history, bit-ring words, GF tables -> LDS.  The discriminator
	// waves issue the first two tiles' loads right away instead (below): a wave's vector loads retire in order (vmcnt), so a
	// wave that has to wait for its share of the taps cannot have tile loads in flight behind them; with the state -> taps
	// chain (two dependent global loads) and the first tile's HBM round trip in parallel a workgroup reaches its first round
	// in ~4 us instead of ~7 (profiles/r3_notes.md).
	uint32_t *ring_g = bitring + (size_t)ch * ring_words;
	const uint32_t ring_mask = ring_words - 1;
	// Round 4, the (4:1, 8 taps) class: the history and the ring words are requested TOGETHER with the taps (one HBM round trip less at
	// the head of every workgroup; the loads the history / open ring word / mirror used to wait for one after the other).  The other
	// classes have no registers to spare while the tile loads are in flight (their first tile's loads spill) and keep the old order.
	constexpr bool JOIN = DEC == 4 && NT == 8;
	if (!is_k) {
		const float4 *taps_g = reinterpret_cast<const float4 *>(taps_all + (size_t)stype * SD_NPHASE * SD_NTAPS);
		const float4 tv = taps_g[tid];                                   // 32 rows x 8 float4: one 16-byte load per lane
		if constexpr (JOIN) {
			// the carried history (stored by wave 0) and the bit ring's newest words, the ones up to and including wpos's (lane 63's), for
			// K4's mirror and the open word (stored by wave 3) -- requested together with the taps, not after the taps have arrived, and
			// by every round wave: unconditional loads (256 bytes each, L2 hits for three of the four waves) keep the three requests of a
			// wave back to back; predicated ones made the compiler wait for the taps before the next request (register reuse)
			const uint32_t w = (uint32_t)(st.wpos >> 5) - (uint32_t)(63 - lane);
			const float hv = sd_ld_coh_f(hist + (size_t)ch * SD_HIST + lane, sliced);
			const uint32_t xw = sd_ld_coh(ring_g + (w & ring_mask), sliced);
			asm volatile("" :: "v"(hv), "v"(xw));      // (a use right here: without it the compiler sinks each load into the branch that stores it)
			// pair-swapped rows (T[2i] = H[2i+1], T[2i+1] = H[2i]), see interp()
			*reinterpret_cast<float4 *>(&s.taps[(tid >> 3) * SD_TAPS_LD + 4 * (tid & 7)]) = make_float4(tv.y, tv.x, tv.w, tv.z);
			if (rwave == 0) s.A[0][lane] = hv;                               // in front of the first tile
			if (rwave == 3) {
				s.mirror[w & (SD_MIRROR_WORDS - 1)] = xw;
				if (lane == 63) {
					s.chunk[0][0] = 0; s.chunk[0][9] = 0; s.chunk[1][0] = 0; s.chunk[1][9] = 0; s.chunk[0][17] = 0; s.chunk[1][17] = 0;
					s.partial[0] = ((uint32_t)st.wpos & 31u) ? xw : 0u;
					s.pub.flag = 0;
					s.pub.wpos = st.wpos;
					s.afc_u[0] = st.afc[0]; s.afc_u[1] = st.afc[1]; s.afc_u[2] = st.afc[2];
					if (IS_IQ) { const float q1 = sd_afc_rot(st.afc[1]); s.afc_dq[1] = q1 - sd_afc_rot(st.afc[0]); s.afc_dq[2] = sd_afc_rot(st.afc[2]) - q1; }
				}
			}
		} else {
			*reinterpret_cast<float4 *>(&s.taps[(tid >> 3) * SD_TAPS_LD + 4 * (tid & 7)]) = make_float4(tv.y, tv.x, tv.w, tv.z);
			// restore the carried history in front of the first tile
			if (tid < SD_LH) {
				const float hv = sd_ld_coh_f(hist + (size_t)ch * SD_HIST + tid, sliced);
				s.A[0][tid] = hv;
			}
			if (tid == 0) {
				s.chunk[0][0] = 0; s.chunk[0][9] = 0; s.chunk[1][0] = 0; s.chunk[1][9] = 0; s.chunk[0][17] = 0; s.chunk[1][17] = 0;
				s.partial[0] = ((uint32_t)st.wpos & 31u) ? sd_ld_coh(ring_g + ((uint32_t)(st.wpos >> 5) & ring_mask), sliced) : 0u;
				s.pub.flag = 0;
				s.pub.wpos = st.wpos;
				s.afc_u[0] = st.afc[0]; s.afc_u[1] = st.afc[1]; s.afc_u[2] = st.afc[2];
					if (IS_IQ) { const float q1 = sd_afc_rot(st.afc[1]); s.afc_dq[1] = q1 - sd_afc_rot(st.afc[0]); s.afc_dq[2] = sd_afc_rot(st.afc[2]) - q1; }
			}
		}
	}
	// K4: the sync search of the framed sondes runs in here, on round wave 3, over an LDS mirror of the newest ring words
	// (RS41 always; DFM / iMS-100 / M10 unless the batch asked for the stand-alone framer kernels)
	// Which sonde types an instantiation can meet follows from its class (batch.hip k_modems): (4, 8) and, wide, (2, 16) ->
	// RS41, DFM, iMS-100, MRZ-N1; (2, 8) and, wide, (1, 16) -> M10 (and the AFSK 6 kS/s streams, which are framed elsewhere).
	// Testing the class first lets the compiler drop the other types' code from each instantiation.
	constexpr bool cls_slow = DEC == 4 || (DEC == 2 && NT == 16);     // the ~5000 chips/s sondes
	const bool is_rs41 = cls_slow && stype == SONDE_RS41, is_dfm = cls_slow && stype == SONDE_DFM09,
	           is_ims = cls_slow && stype == SONDE_IMS100, is_m10 = !cls_slow && stype == SONDE_M10,
	           is_mrz = cls_slow && stype == SONDE_MRZN1;
	const bool fuse = fo->fuse_fec != 0;
	const bool framing = is_rs41 || (fuse && (is_dfm || is_ims || is_m10 || is_mrz));   // workgroup-uniform
	if (!JOIN && framing && !is_k && tid >= SD_WG - SD_MIRROR_WORDS) {                      // round wave 3, the wave that runs K4
		const uint32_t w = (uint32_t)(st.wpos >> 5) - (uint32_t)(SD_WG - 1 - tid);       // the words up to and including wpos's
		s.mirror[w & (SD_MIRROR_WORDS - 1)] = sd_ld_coh(ring_g + (w & ring_mask), sliced);
	}

	// K5/K6 in this kernel's epilogue: RS41 only (few, heavy frames per submit).  The short frames of the fixed-length
	// framers (a DFM frame every 2.7 tiles) decode faster as one wave per frame across the whole GPU (framer2_kernel.hip)
	// than eight at a time at the end of each workgroup: measured 0.358 vs 0.324 ms per step for 4096 DFM channels.
	const bool fec_here = is_rs41 && fuse;     // workgroup-uniform
	EpiTabs &et = *reinterpret_cast<EpiTabs *>(&s.X[1][SD_EPI_TAB_OFF]);
	// Round 3: clean frames can be decoded INSIDE the tile loop, by round wave 2, a step per round (sd_rs41_loop_step); its
	// work area sits in the part of A[1] no decimated tile uses.
	// Only where it pays (interleaved A/B, tools/ab_repeat.sh, profiles/r3_notes.md): launches whose workgroups are ALL resident at
	// once -- their epilogues coincide, nothing else streams meanwhile: 1024 x 96 tiles -1.6 % at 14 dB, -1.1 % at 9 dB -- and long
	// enough (>= 48 tiles; at 24 the steps of the 1.6 frames a submit completes stall as much as they save).  In launches of several
	// generations the epilogues overlap other workgroups' streaming for free and the steps only stall: 4096 x 96 tiles +7 %.
	const bool fec_loop = fec_here && n_tiles >= 48 && grid_wg <= fo->loop_fec_max_wg && !sliced;
	FramerLds &loop_wl = *reinterpret_cast<FramerLds *>(&s.A[1][1100]);
	static_assert(sizeof(FramerLds) <= (SD_BUF - 1100) * sizeof(float), "in-loop FEC work area");
	if (tid == 3 * 64 - 1) { s.fec.frame = 0; s.fec.phase = 0; s.fec.done_mask = 0; }      // (wave 2: the wave that uses it)
	if (fec_here && is_rs41 && !is_k) {
		// GF(2^8) tables for the epilogue: one 16-byte global load per round-wave thread now, hidden behind the first tile's loads
		static_assert(GF_EXP2 / 16 + 512 / 16 + RS_R * 8 * 4 / 16 <= SD_WG, "the round waves load the GF tables");
		if (tid < GF_EXP2 / 16) reinterpret_cast<uint4 *>(et.tabs.exp2)[tid] = reinterpret_cast<const uint4 *>(fo->gf_exp)[tid];
		else if (tid < GF_EXP2 / 16 + 512 / 16) reinterpret_cast<uint4 *>(et.tabs.log2)[tid - GF_EXP2 / 16] = reinterpret_cast<const uint4 *>(fo->gf_log)[tid - GF_EXP2 / 16];
		else if (tid < GF_EXP2 / 16 + 512 / 16 + RS_R * 8 * 4 / 16)
			reinterpret_cast<uint4 *>(et.swar)[tid - (GF_EXP2 / 16 + 512 / 16)] = reinterpret_cast<const uint4 *>(fo->gf_swar)[tid - (GF_EXP2 / 16 + 512 / 16)];
	}

	// ================================================================ discriminator role (waves 4-7)
	LoadT pa, pb, qa, qb;                  // the float4s (two input samples each) just before the wave's first one: -1 (pa, pb), -2 (qa, qb)
	if constexpr (IQ8 && IQ16D4) { pa = pb = qa = qb = make_uint2(0u, 0u); }
	else if constexpr (IQ8) { pa = pb = qa = qb = 0u; }
	else if constexpr (IQ16D4) { pa = pb = qa = qb = make_uint4(0u, 0u, 0u, 0u); }
	else if constexpr (IQ16) { pa = pb = qa = qb = make_uint2(0u, 0u); }
	else { qa = qb = make_float4(-0.0f, -0.0f, -0.0f, -0.0f); }
	float2 last_iq = make_float2(st.iq_last[0], st.iq_last[1]);
	auto load_prev = [&](int tile, LoadT &pv, LoadT &pw) {
		if constexpr (IQ16) {
			// (16-bit input: the raw pairs; at the very start of the stream k1_tile takes the carried sample from the state itself)
			const long f4 = (long)tile * TILE_F4 + SD_WG / 4 * NLD * kw;
			if (f4 > 0) {
				pv = src[f4 - 1];
				if (dec4 && !IQ16D4) pw = src[f4 - 2];
			}
		} else if constexpr (IS_IQ) {
			// the two float4s just before this wave's first one: wave-uniform addresses, so scalar loads (no VGPRs,
			// not counted by vmcnt); at the very start of the stream they stand for the carried (decimated) sample.
			// -0.0f is the additive identity for every float (+0 and -0 included), so k1_tile needs no case split
			const long f4 = (long)tile * TILE_F4 + SD_WG / 4 * NLD * kw;
			const float4 nz = make_float4(-0.0f, -0.0f, -0.0f, -0.0f);
			if (f4 > 0) {
				pv = src[f4 - 1];
				if (dec4) pw = src[f4 - 2];
			} else {
				// (the carried sample is a finished z: it stands for the first half P0, the half behind it, which SPEC 3.0d turns, is empty)
				pv = dec4 ? nz : (dec2 ? make_float4(st.iq_last[0], st.iq_last[1], -0.0f, -0.0f) : make_float4(0.0f, 0.0f, st.iq_last[0], st.iq_last[1]));
				pw = dec4 ? make_float4(st.iq_last[0], st.iq_last[1], -0.0f, -0.0f) : nz;
			}
		}
	};
	auto load_tile = [&](int tile, LoadT (&v)[NLD], LoadT &pv, LoadT &pw) { load_vec(tile, v); load_prev(tile, pv, pw); };
	// K0+K1: (2:1 boxcar decimation,) d[n] = atan2q(z[n] * conj(z[n-1])), straight into buffer b
	// AFC (SPEC 3.0b, IQ input): tile T's products are turned back by the phasor (1 - u^2, 2u), u = the state the lead wave published
	// three tiles earlier (LDS, written two barriers ago) or, for the first three tiles of a submit, carried in the channel state
	auto afc_of = [&](int tile) -> float {
		const float u = tile == 0 ? st.afc[0] : (tile == 1 ? st.afc[1] : (tile == 2 ? st.afc[2] : s.afc_u[tile & 3]));
		return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, u)));
	};
	auto k1_tile = [&](int b, int tile, const LoadT (&vraw)[NLD], const LoadT &pvraw, const LoadT &pwraw) {
		const float afc_u = IS_IQ ? afc_of(tile) : 0.0f;
		const float rc = __builtin_fmaf(-afc_u, afc_u, 1.0f), rs = afc_u + afc_u;
		// SPEC 3.0d: the boxcar's second half is turned back by R = (1 - u^2 / 2, -u) of its OWN tile; the sample in front of the tile's
		// first wave belongs to the tile before
		const float fr = __builtin_fmaf(-0.5f * afc_u, afc_u, 1.0f), fi_ = -afc_u;
		const float pu = (IS_IQ && kw == 0 && tile > 0) ? afc_of(tile - 1) : afc_u;
		const float pfr = __builtin_fmaf(-0.5f * pu, pu, 1.0f), pfi = -pu;
		if constexpr (IQ16D4) {
			// 16-bit input, 4:1: load g of lane l IS decimated sample 128 kw + 64 g + l (no exchange between lanes as in the float path)
			float4 ch;
			if constexpr (IQ8) ch = sd_cs8_halves_uniform(pvraw); else ch = sd_cs16_halves_uniform(pvraw);
			const float2 c = sd_follow(ch.x, ch.y, ch.z, ch.w, pfr, pfi);
			float cx = c.x, cy = c.y;
			if (tile == 0 && kw == 0) { cx = st.iq_last[0]; cy = st.iq_last[1]; }       // the carried (decimated) sample
#pragma unroll
			for (int g = 0; g < NLD; g++) {
				float4 hh;
				if constexpr (IQ8) hh = sd_cs8_halves(vraw[g]); else hh = sd_cs16_halves(vraw[g]);
				const float2 z = sd_follow(hh.x, hh.y, hh.z, hh.w, fr, fi_);
				const float px = sd_wave_shr1(z.x, cx), py = sd_wave_shr1(z.y, cy);
				store_one(s, b, (uint32_t)(128 * kw + 64 * g + lane), sd_disc_rot(z.x, z.y, px, py, rc, rs));
				cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z.x), 63));
				cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z.y), 63));
			}
			last_iq = make_float2(cx, cy);
			return;
		}
		float4 v[NLD], pv, pw;
		if constexpr (IQ16D4) {
		} else if constexpr (IQ8) {
#pragma unroll
			for (int r = 0; r < NLD; r++) v[r] = sd_cs8_f4(vraw[r]);
			pv = sd_cs8_f4(pvraw); pw = sd_cs8_f4(pwraw);
		} else if constexpr (IQ16) {
#pragma unroll
			for (int r = 0; r < NLD; r++) v[r] = sd_cs16_f4(vraw[r]);
			pv = sd_cs16_f4(pvraw); pw = sd_cs16_f4(pwraw);
		} else {
#pragma unroll
			for (int r = 0; r < NLD; r++) v[r] = vraw[r];
			pv = pvraw; pw = pwraw;
		}
		// lane 0's predecessor (decimated) sample; after each load: lane 63's last sample
		float cx = pv.z, cy = pv.w;
		if (IS_IQ && (dec4 || dec2)) {
			const float2 c = dec4 ? sd_follow(pw.x + pw.z, pw.y + pw.w, pv.x + pv.z, pv.y + pv.w, pfr, pfi) : sd_follow(pv.x, pv.y, pv.z, pv.w, pfr, pfi);
			cx = c.x; cy = c.y;
		}
		if (IQ16 && tile == 0 && kw == 0) { cx = st.iq_last[0]; cy = st.iq_last[1]; }       // the carried (decimated) sample
		if (IS_IQ && dec4) {
			// 4:1: a decimated sample spans two adjacent float4s, which the coalesced loads put into adjacent LANES.
			// The per-float4 partial sums of two loads go through a wave-private LDS scratch (128 float2) and come back
			// as one 16-byte read per lane: lane l gets the partials of float4s 2l and 2l+1, i.e. decimated sample l of
			// the 64 this pair of loads holds.  (The scratch sits in the part of A[0] a 512-sample tile never uses.)
			float2 *scr = reinterpret_cast<float2 *>(&s.A[0][1024]) + 128 * kw;
#pragma unroll
			for (int g = 0; g < NLD / 2; g++) {
				scr[lane] = make_float2(v[2 * g].x + v[2 * g].z, v[2 * g].y + v[2 * g].w);
				scr[64 + lane] = make_float2(v[2 * g + 1].x + v[2 * g + 1].z, v[2 * g + 1].y + v[2 * g + 1].w);
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
				const float4 q = *reinterpret_cast<const float4 *>(&scr[2 * lane]);
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
				const float2 z = sd_follow(q.x, q.y, q.z, q.w, fr, fi_);
				const float zx = z.x, zy = z.y;
				// (lane l takes lane l - 1's sample, lane 0 the carry: one DPP move each; __shfl_up is a ds_bpermute on the LDS pipe)
				const float px = sd_wave_shr1(zx, cx), py = sd_wave_shr1(zy, cy);
				store_one(s, b, (uint32_t)(128 * kw + 64 * g + lane), sd_disc_rot(zx, zy, px, py, rc, rs));
				cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zx), 63));
				cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zy), 63));
			}
			last_iq = make_float2(cx, cy);
			return;
		}
#pragma unroll
		for (int r = 0; r < NLD; r++) {
			const uint32_t fi = (uint32_t)f4_index(r);
			if (IS_IQ) {
				if (dec2) {
					// one float4 = two input samples = one decimated sample z, index fi
					const float2 z = sd_follow(v[r].x, v[r].y, v[r].z, v[r].w, fr, fi_);
					const float zx = z.x, zy = z.y;
					const float px = sd_wave_shr1(zx, cx), py = sd_wave_shr1(zy, cy);
					store_one(s, b, fi, sd_disc_rot(zx, zy, px, py, rc, rs));
					cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zx), 63));
					cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zy), 63));
				} else {
					const float px = sd_wave_shr1(v[r].z, cx), py = sd_wave_shr1(v[r].w, cy);
					const float d0 = sd_disc_rot(v[r].x, v[r].y, px, py, rc, rs);
					const float d1 = sd_disc_rot(v[r].z, v[r].w, v[r].x, v[r].y, rc, rs);
					store_pair(s, b, 2u * fi, d0, d1);
					cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[r].z), 63));
					cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[r].w), 63));
				}
			} else if (dec4) {
				// real input, 4:1: one float4 = one decimated sample
				store_one(s, b, fi, ((v[r].x + v[r].y) + (v[r].z + v[r].w)) * 0.25f);
			} else if (dec2) {
				// real input: average pairs, 4 inputs -> 2 consecutive decimated samples
				store_pair(s, b, 2u * fi, (v[r].x + v[r].y) * 0.5f, (v[r].z + v[r].w) * 0.5f);
			} else {
				store_pair(s, b, 4u * fi, v[r].x, v[r].y);
				store_pair(s, b, 4u * fi + 2u, v[r].z, v[r].w);
			}
		}
		if (IS_IQ) last_iq = make_float2(cx, cy);      // wave 7: the last (decimated) sample of the tile
	};

	// ================================================================ round role (waves 0-3)
	const bool lead = rwave == 0;          // wave 0 owns the loop filter, the bit ring and the state
	unsigned seq = 0;                      // round counter; its parity selects the red/chunk/partial slots
	int pendK = -1;                        // symbols of the round whose statistics are pending, -1: none
	int64_t n0 = st.n0;                    // samples consumed up to and including the tile in LDS
	int64_t t_next = st.t_next;            // per-round values every round wave needs
	int period = st.period;
	float bias = st.bias;

	// FIR at this lane's symbol + Gardner term + slicer, integer statistics of the round -> LDS
	// SPL symbols per lane: symbol k = 256 h + t of the round, h < SPL (the undecimated streams hold up to 410 / 822 symbols per tile)
	constexpr int SPL = SD_ROUND_SPL(DEC, NT);
	auto round_front = [&](int K, int b, int par) {
		int Ei = 0, S1i = 0, S0i = 0, C1 = 0, SYi = 0, SMi = 0;
		constexpr bool HW = NT == 8;             // the half-wave symbol mapping (sd_wave.h): 2.5 samples per symbol
#pragma unroll
		for (int h = 0; h < SPL; h++) {
			const int k = (HW ? 64 * rwave + sd_hw_symbol(lane) : t) + SD_WG * h;
			const bool act = k < K;
			float y = 0.0f, m = 0.0f;
			if (act) {
				const int64_t base = (n0 - IT - SD_LH) << 16;
				const uint32_t rel = (uint32_t)(t_next - base) + __umul24((unsigned)k, (unsigned)period);      // (k < 2^10, period < 2^20: one full-rate v_mad_u32_u24, not a quarter-rate v_mul_lo_u32)
				y = interp<NT>(s.A[b], s.taps, rel);          // 3.2 symbols of taps (8 at 2.5 samples per symbol)
				// only the first 256 symbols of a round feed the timing detector (SPEC 3.2): the second symbol of a lane needs no
				// mid-symbol FIR -- a quarter of the FIR work of the two-symbols-per-lane classes (M10: 117.7 M -> 108.5 M VALU instructions, 297 -> 277 us)
				if (h == 0) m = interp<NT>(s.A[b], s.taps, rel - ((uint32_t)period >> 1));
			}
			if (h == 0) {
				const float yprev = HW ? sd_hw_prev(y, lane) : sd_wave_shr1(y, 0.0f);            // (the 64-group's first symbol's term is not used)
				float e = (yprev - y) * (m - bias);
				e = sd_clamp(e * 1024.0f, -1.0e6f, 1.0e6f);
				Ei += (act && lane != 0) ? __float2int_rn(e) : 0;    // first symbol of a 64-group (lane 0 in either mapping): no term (SPEC 3.2)
				if (AF) {                                            // SPEC 3.6b: same symbols as the detector's
					SYi += (act && lane != 0) ? __float2int_rn(sd_clamp(fabsf(y - bias), 0.0f, 8.0f) * 4096.0f) : 0;
					SMi += (act && lane != 0) ? __float2int_rn(sd_clamp(fabsf(m - bias), 0.0f, 8.0f) * 4096.0f) : 0;
				}
			}
			const bool bit = act && (y > bias);
			const int Y = __float2int_rn(sd_clamp(y, -8.0f, 8.0f) * 4096.0f);
			S1i += bit ? Y : 0;
			S0i += (act && !bit) ? Y : 0;
			const unsigned long long bal = HW ? __ballot(sd_hw_unpermute(bit ? 1 : 0, lane) != 0) : __ballot(bit);      // bits in symbol order
			C1 += __popcll(bal);
			if (lane == 0) {
				s.chunk[par][1 + 2 * rwave + 8 * h] = (uint32_t)bal;
				s.chunk[par][2 + 2 * rwave + 8 * h] = (uint32_t)(bal >> 32);
			}
		}
		Ei = wave_sum(Ei);
		S1i = wave_sum(S1i);
		S0i = wave_sum(S0i);
		if (lane == 0) s.red[par][rwave] = make_int4(Ei, S1i, S0i, C1);
		if (AF) {
			SYi = wave_sum(SYi);
			SMi = wave_sum(SMi);
			if (lane == 0) s.red2[par][rwave] = make_int2(SYi, SMi);
		}
	};
	// round wave 1, after the barrier: append the previous round's bits to the bit ring (HBM) and its LDS mirror, beside the lead wave's
	// loop filter (round 4: it was the first quarter of the lead wave's chain) and tell K4 how far the mirror is valid
	uint64_t wpos_w = st.wpos;
	auto ring_append = [&](int K, int par) {
		if (K <= 0) {                      // nothing appended: carry the partial word to the next slot
			if (lane == 0) s.partial[par ^ 1] = s.partial[par];
			return;
		}
		if (lane < 8 * SPL + 1) {
			const uint32_t sh = (uint32_t)wpos_w & 31u;
			const uint32_t w0 = (uint32_t)(wpos_w >> 5);
			uint32_t vv = 0;
			const bool touched = (uint32_t)(32 * lane) < sh + (uint32_t)K;
			if (touched) {
				const uint32_t lo = s.chunk[par][lane + 1];
				const uint32_t pvw = s.chunk[par][lane];
				vv = sh ? ((lo << sh) | (pvw >> (32u - sh))) : lo;
				const uint32_t idx = (w0 + lane) & ring_mask;
				if (lane == 0 && sh) vv |= s.partial[par] & ((1u << sh) - 1u);
				sd_st_coh(ring_g + idx, vv, sliced);
				s.mirror[idx & (SD_MIRROR_WORDS - 1)] = vv;
			}
			// whoever owns the word the next round starts in publishes it (read after a barrier)
			if ((uint32_t)lane == ((sh + (uint32_t)K) >> 5)) s.partial[par ^ 1] = vv;
		}
		wpos_w += (uint64_t)K;
		// every bit below it is in the ring and in the mirror (this wave's LDS stores execute in order; K4 reads it whenever it looks:
		// how far the search has come by a given round may vary, what it finds does not)
		if (lane == 0) __hip_atomic_store(&s.pub.wpos, (unsigned long long)wpos_w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	// lead wave, after the barrier: update slicer levels and the PI loop filter
	auto round_back = [&](int K, int par, int64_t n_done /* samples consumed up to and including that round's tile */) {
		if (K <= 0) return;
		const int4 r0 = s.red[par][0], r1 = s.red[par][1], r2 = s.red[par][2], r3 = s.red[par][3];
		const int E = r0.x + r1.x + r2.x + r3.x;
		const int S1 = r0.y + r1.y + r2.y + r3.y;
		const int S0 = r0.z + r1.z + r2.z + r3.z;
		const int C1 = r0.w + r1.w + r2.w + r3.w;
		const int C0 = K - C1;
		if (C1 > 0 && C0 > 0) {
			const f32x2 cnt = {(float)C1, (float)C0};
			const f32x2 rc = sd_recip2(cnt);
			const float hi = ((float)S1 * rc.x) * (1.0f / 4096.0f);
			const float lo = ((float)S0 * rc.y) * (1.0f / 4096.0f);
			const float c = 0.5f * (hi + lo);
			const float a = 0.5f * (hi - lo);
			if (st.nstat == 0) {
				st.bias = c;
				st.amp = a;
			} else {
				st.bias = st.bias + 0.5f * (c - st.bias);
				st.amp = st.amp + 0.5f * (a - st.amp);
			}
			if (!(st.amp >= 1.0e-3f)) st.amp = 1.0e-3f;
			st.nstat = 1;
		} else {
			// one-sided round (carrier offset larger than the deviation): no level estimate; move the threshold to the
			// mean of the round so that the next one sees both levels (SPEC 3.2)
			st.bias = ((float)(S1 + S0) * sd_recip((float)K)) * (1.0f / 4096.0f);
		}
		// SPEC 3.2b (round 5), acquisition: during a stream's first three tiles the threshold is the mean of the round (the level estimate
		// above is built from the slicer's own decisions: with the carrier 2 kHz off it needs several tiles to find the offset)
		if (!afsk && n_done <= 3 * (int64_t)IT) st.bias = ((float)(S1 + S0) * sd_recip((float)K)) * (1.0f / 4096.0f);
		const f32x2 den = {(float)(K > SD_ROUND_MAX ? SD_ROUND_MAX : K), st.amp * st.amp};      // the symbols that fed the detector
		const f32x2 rd = sd_recip2(den);
		float err = ((float)E * rd.x) * (1.0f / 1024.0f);
		err = err * rd.y;
		err = sd_clamp(err, -1.0f, 1.0f);
		const int dphase = __float2int_rn(err * md.kp);
		const int dper = __float2int_rn(err * md.ki);
		st.t_next += (int64_t)K * st.period + dphase;
		if (AF && afsk && st.nstat) {
			// SPEC 3.6b, acquisition aid of the AFSK streams: half a symbol off, the Gardner detector sits on its unstable zero and a loop
			// closed 3-6 times a second takes seconds to leave it; there the mid-symbol samples show the open eye.  Jump half a symbol.
			const int2 q0 = s.red2[par][0], q1 = s.red2[par][1], q2 = s.red2[par][2], q3 = s.red2[par][3];
			const int64_t SY = (int64_t)q0.x + q1.x + q2.x + q3.x, SM = (int64_t)q0.y + q1.y + q2.y + q3.y;
			if (16 * SM > 17 * SY) st.t_next += st.period >> 1;
		}
		st.period += dper;
		if (st.period < md.pmin) st.period = md.pmin;
		if (st.period > md.pmax) st.period = md.pmax;
		st.wpos += (uint64_t)K;
	};

	// AFC (SPEC 3.0b), lead wave, after the rounds of tile j: the slicer threshold is what is left of the carrier offset behind the
	// rotation; a leaky integrator moves the state, which the discriminator of tile j + 3 will use
	float afc_last = st.afc[2];
	// SPEC 3.0e (round 5): the next tile's discriminator turns the signal back by rot(u[j + 1]) instead of rot(u[j]) quadrants per sample:
	// the slicer threshold drops by the difference now, instead of following through its smoothing some tiles later.  The difference
	// was formed when u[j + 1] was (two tiles ago): one LDS read and one subtraction on the lead wave's chain.
	float afc_rq_last = IS_IQ ? sd_afc_rot(afc_last) : 0.0f;
	auto afc_step = [&](int j) {
		const float dq = s.afc_dq[(j + 1) & 3];
		float u = __builtin_fmaf(-SD_AFC_LEAK, afc_last, afc_last);
		u = __builtin_fmaf(SD_AFC_GAIN, st.bias, u);
		u = sd_clamp(u, -SD_AFC_MAX, SD_AFC_MAX);
		afc_last = u;
		st.bias -= dq;
		const float rq = sd_afc_rot(u);
		if (lane == 0) { s.afc_u[(j + 3) & 3] = u; s.afc_dq[(j + 3) & 3] = rq - afc_rq_last; }
		afc_rq_last = rq;
	};

	// ---- the two roles run separate loops (so that neither carries the other's live registers) with the
	// same number of s_barriers: one after the prologue, then `rounds` per tile.  is_k is wave-uniform.
	auto k4_load = [&]() {         // one lane: the channel's search state into LDS
		const SdFramerState f0 = sd_ld_coh_struct(fo->fstates + ch, sliced);
		s.k4.rpos = f0.rpos; s.k4.fstart = f0.fstart; s.k4.collecting = f0.collecting; s.k4.inv = f0.inv; s.k4.flen = f0.flen;
		// (time slices: a later segment appends to the frames its predecessors listed in this submit)
		const uint32_t nout0 = (sliced && seg > 0) ? sd_ld_coh(fo->counts + ch) : 0u;
		s.k4.nout = nout0; s.nout0 = nout0; s.k4.wp_seen = st.wpos;
	};
	auto k4_run = [&](uint64_t wp) {   // one step of the channel's sync-search state machine (wave-uniform type dispatch)
		SdFrameDesc *dch = (SdFrameDesc *)fo->descs + (size_t)ch * fo->max_frames;
		const uint32_t mf = fo->max_frames;
		if (is_rs41) sd_rs41_sync_step(s.k4, wp, s.mirror, lane, dch, mf);
		else if (is_dfm) sd_fixed_sync_step<SONDE_DFM09>(s.k4, wp, s.mirror, lane, dch, mf);
		else if (is_m10) sd_fixed_sync_step<SONDE_M10>(s.k4, wp, s.mirror, lane, dch, mf);
		else if (is_ims) sd_fixed_sync_step<SONDE_IMS100>(s.k4, wp, s.mirror, lane, dch, mf);
		else if (is_mrz) sd_fixed_sync_step<SONDE_MRZN1>(s.k4, wp, s.mirror, lane, dch, mf);
	};
	auto k4_finish = [&]() {       // K4's wave, after barrier E: catch up with the last rounds' bits, state and frame count back to HBM
		k4_run(sd_uniform64(s.pub.wpos));
		if (lane == 0) {
			SdFramerState f1;
			f1.rpos = s.k4.rpos; f1.fstart = s.k4.fstart; f1.collecting = s.k4.collecting; f1.inv = s.k4.inv; f1.flen = s.k4.flen; f1.pad = 0;
			sd_st_coh_struct(fo->fstates + ch, f1, sliced);
			sd_st_coh(fo->counts + ch, s.k4.nout, sliced);
		}
	};
	if (is_k) {
		// register set A holds the even tiles, set B the odd ones (the arguments are literals at every call: static register sets)
		auto k1A = [&](int b, int tile) { k1_tile(b, tile, va, pa, qa); };
		auto k1B = [&](int b, int tile) { k1_tile(b, tile, vb, pb, qb); };
		auto ldA = [&](int tile) { load_tile(tile, va, pa, qa); };
		auto ldB = [&](int tile) { load_tile(tile, vb, pb, qb); };
		load_prev(0, pa, qa);          // (the vector loads of tiles 0 and 1 went out at the top of the kernel)
		if (n_tiles > 1) load_prev(1, pb, qb);
		k1A(0, 0);
		if (n_tiles > 2) ldA(2);
		__syncthreads();
		// phase `tile`: tile+1 goes from registers into the other LDS buffer, tile+3 is requested from HBM
		// (two phases of latency budget); the loop is unrolled by two so the register sets are static
		int tile = 0;
		// Steady state with every load of an iteration issued UNCONDITIONALLY (16-bit input only).  The compiler inserts the s_waitcnt in
		// front of a register set's first use; with `if (tile + 3 < n_tiles) ldB(...)` in the loop it cannot know whether the four
		// loads of the OTHER set are in flight behind the set it waits for, must assume they are not, and waits for vmcnt(3..0): for
		// everything, the tile requested a moment ago included -- the second register set then prefetches less than a tile ahead.
		// With unconditional loads it counts the newer ones and waits for vmcnt(6), (4).  Measured, interleaved
		// (profiles/r4_ab_uncond.txt): 16-bit rows 1024 x 96 tiles -2.5 %, 4096 x 96 -1.7 %; FLOAT rows: 8192 x 24 -1.5 % but the
		// one-residency headline +4 % (0.2548-0.2620 -> 0.2685-0.2707 ms) and 4096 x 96 +1.5 %: with 16 KB tiles the launch does
		// better with FEWER bytes in flight per workgroup (the memory system's latency grows faster than the bytes: r4_notes.md),
		// so the float classes keep the conditional loop and its conservative waits.  The last iterations run the general loop below.
		if constexpr (IQ16) {
			for (; tile + 4 < n_tiles; tile += 2) {
				if (t < SD_LH) s.A[1][t] = s.A[0][IT + t];                // history roll into the other buffer
				k1B(1, tile + 1);
				ldB(tile + 3);
				for (int r = 0; r < rounds; r++) __syncthreads();
				if (t < SD_LH) s.A[0][t] = s.A[1][IT + t];
				k1A(0, tile + 2);
				ldA(tile + 4);
				for (int r = 0; r < rounds; r++) __syncthreads();
			}
		}
		for (; tile < n_tiles; tile += 2) {
			if (tile + 1 < n_tiles) {
				if (t < SD_LH) s.A[1][t] = s.A[0][IT + t];            // history roll into the other buffer
				k1B(1, tile + 1);
				if (tile + 3 < n_tiles) ldB(tile + 3);
			}
			for (int r = 0; r < rounds; r++) __syncthreads();
			if (tile + 1 >= n_tiles) break;
			if (tile + 2 < n_tiles) {
				if (t < SD_LH) s.A[0][t] = s.A[1][IT + t];
				k1A(0, tile + 2);
				if (tile + 4 < n_tiles) ldA(tile + 4);
			}
			for (int r = 0; r < rounds; r++) __syncthreads();
		}
		if (IS_IQ && t == SD_WG - 1) { s.iq_last[0] = last_iq.x; s.iq_last[1] = last_iq.y; }
		__syncthreads();                                   // (E)
	} else {
		// the round waves are the critical path of a tile (update -> FIR -> reduction, all dependent);
		// the discriminator waves only have to be done by the next barrier: let the round waves win
		// every issue arbitration (static priority, T5 in the CDNA guide)
		if (lead) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
		// K4 (wave 3 of an RS41 channel only): its state sits in LDS between steps, the output pointers in a
		// descriptor in HBM -- scalar registers are the scarce resource of this kernel
		const bool k4 = framing && rwave == 3;
		if (k4 && lane == 0) k4_load();
		__syncthreads();
		int K_total = 0;
		for (int tile = 0; tile < n_tiles; tile++) {
			const int b = tile & 1;
			for (int r = 0; r < rounds; r++) {
				const int par = (int)(seq & 1u);                      // slots of the round about to run
				int K;
				if (r == 0) n0 += IT;                                 // the tile in buffer b is now counted
				if (lead) {
					if (pendK >= 0) {
						round_back(pendK, par ^ 1, r == 0 ? n0 - IT : n0);      // the previous round's update (n0 already counts this round's tile)
						if (IS_IQ) afc_step(tile - 1);                // (IQ classes run one round per tile)
					}
					if (r == 0) {
						const int64_t limit = (((n0 - 1 - NT / 2 - SD_MARGIN) << 16) | 0xFFFF);
						K_total = (st.t_next <= limit) ? (int)((uint32_t)(limit - st.t_next) / (uint32_t)st.period) + 1 : 0;
					}
					K = K_total > SPL * SD_ROUND_MAX ? SPL * SD_ROUND_MAX : K_total;
					K_total -= K;
					t_next = st.t_next; period = st.period; bias = st.bias;
					if (lane == 0) {
						s.pub.t_next = t_next; s.pub.period = period; s.pub.bias = bias; s.pub.K = K;
						__hip_atomic_store(&s.pub.flag, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				} else {
					// K4 instead of spinning while the lead wave runs the loop filter: the search works on the bits the
					// PREVIOUS publish announced (one round behind; the epilogue catches up)
					if (rwave == 1 && pendK >= 0) ring_append(pendK, par ^ 1);      // the previous round's bits
					if (k4) k4_run(sd_uniform64(s.k4.wp_seen));
					// the in-loop decoder of clean RS41 frames: one step per round on round wave 2, which would otherwise only wait
					// for the lead wave's loop filter (a real call: sd_rsdec.h says why)
					if (fec_loop && rwave == 2) sd_rs41_loop_step(s.fec, s.k4, et.tabs, et.swar, loop_wl, ring_g, ring_mask,
					                                              fo->frames + (size_t)ch * fo->max_frames, ch, fo->max_frames, lane);
					while (__hip_atomic_load(&s.pub.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq + 1u)
						__builtin_amdgcn_s_sleep(2);
					t_next = s.pub.t_next; period = s.pub.period; bias = s.pub.bias; K = s.pub.K;
					if (k4 && lane == 0) s.k4.wp_seen = __hip_atomic_load(&s.pub.wpos, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
				round_front(K, b, par);
				pendK = K;
				seq++;
				__syncthreads();
			}
		}
		// ---- epilogue of the round role: the last round's update ...
		if (lead && pendK >= 0) {
			round_back(pendK, (int)((seq & 1u) ^ 1u), n0);
			if (IS_IQ) afc_step(n_tiles - 1);
		}
		if (rwave == 1 && pendK >= 0) ring_append(pendK, (int)((seq & 1u) ^ 1u));
		__syncthreads();                                   // (E) matched by the discriminator role's last barrier
		if (k4) k4_finish();      // ... and K4's catch-up over the last rounds' bits
	}

	// ---- common epilogue (after barrier E): carry history and state to the next submit
	const int bl = (n_tiles - 1) & 1;
	if (tid < SD_LH) sd_st_coh_f(hist + (size_t)ch * SD_HIST + tid, s.A[bl][IT + tid], sliced);
	if (tid == 0) {
		st.n0 = n0;
		if (IS_IQ) { st.iq_last[0] = s.iq_last[0]; st.iq_last[1] = s.iq_last[1]; }
		if (IS_IQ) { st.afc[0] = s.afc_u[n_tiles & 3]; st.afc[1] = s.afc_u[(n_tiles + 1) & 3]; st.afc[2] = s.afc_u[(n_tiles + 2) & 3]; }
		sd_st_coh_struct(states + ch, st, sliced);
	}

	// ---- K5/K6 (RS41 channels): the frames K4 listed in this submit, one per wave
	if (fec_here) {
		__syncthreads();           // (F) K4's catch-up is done (nout in LDS, descriptors in HBM); the tile buffers are dead
		const uint32_t max_frames = fo->max_frames;
		const uint32_t nfr = min((uint32_t)__builtin_amdgcn_readfirstlane((int)s.k4.nout), max_frames);
		const uint32_t nf0 = sliced ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s.nout0) : 0u;      // (the earlier segments decoded theirs)
		if (nf0 + (uint32_t)wave < nfr) {
			FramerLds &wl = reinterpret_cast<FramerLds *>(&s.A[0][0])[wave];
			GfSwar swar;
			const uint32_t *sw = et.swar + 8 * (lane % RS_R);
			swar.a_lo = sw[0]; swar.a_hi = sw[1]; swar.b_lo = sw[2]; swar.b_hi = sw[3]; swar.c = sw[4];
			const unsigned long long *dg = reinterpret_cast<const unsigned long long *>((const SdFrameDesc *)fo->descs + (size_t)ch * max_frames);
			SondeFrame *fout = fo->frames + (size_t)ch * max_frames;
			const uint32_t done_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.fec.done_mask);      // records the loop has written
			for (uint32_t k = nf0 + (uint32_t)wave; k < nfr; k += SD_WGT / 64) {
				if (k < 32u && ((done_mask >> k) & 1u)) continue;
				// the descriptor: from K4's list in LDS, or (beyond its first entries) from HBM, where wave 3 of this workgroup
				// stored it a moment ago: agent-scope loads (L2), like the ring words
				unsigned long long d0, d1;
				if (k < SD_K4_LIST) {
					const unsigned long long *dl = reinterpret_cast<const unsigned long long *>(&s.k4.list[k]);
					d0 = dl[0]; d1 = dl[1];
				} else {
					d0 = __hip_atomic_load(dg + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					d1 = __hip_atomic_load(dg + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				sd_rs41_decode_frame_call(et.tabs, wl, sw, ring_g, ring_mask, d0, d1, fout + k, ch, lane);
			}
		}
	}

	// ---- the fixed-length framers' frames (DFM / M10 / iMS-100 / MRZ-N1), round 6: decoded HERE as well, one frame per wave, instead of
	// by a kernel of their own behind this one (sd_dec_fixed_kernel: one launch more per type and submit, 14-20 us that run alone at
	// the end of a joined submit).  Manchester / biphase decode, de-interleaving, Hamming(8,4) / checksum / BCH(63,51): sd_fixed.h, the
	// same device functions the stand-alone kernel calls, reading the bit ring this workgroup has just written (agent-scope loads).
	const bool fixed_here = fuse && fo->fixed_epi != 0 && (is_dfm || is_ims || is_m10 || is_mrz);       // workgroup-uniform
	if (fixed_here) {
		__syncthreads();           // (F) K4's catch-up is done (nout in LDS, descriptors in HBM); the tile buffers are dead
		const uint32_t max_frames = fo->max_frames;
		const uint32_t nfr = min((uint32_t)__builtin_amdgcn_readfirstlane((int)s.k4.nout), max_frames);
		const uint32_t nf0 = sliced ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s.nout0) : 0u;
		if (nf0 + (uint32_t)wave < nfr) {
			FixedLds &wl = reinterpret_cast<FixedLds *>(&s.A[0][0])[wave];
			const unsigned long long *dg = reinterpret_cast<const unsigned long long *>((const SdFrameDesc *)fo->descs + (size_t)ch * max_frames);
			SondeFrame *fout = fo->frames + (size_t)ch * max_frames;
			for (uint32_t k = nf0 + (uint32_t)wave; k < nfr; k += SD_WGT / 64) {
				unsigned long long d0, d1;
				if (k < SD_K4_LIST) {
					const unsigned long long *dl = reinterpret_cast<const unsigned long long *>(&s.k4.list[k]);
					d0 = dl[0]; d1 = dl[1];
				} else {
					d0 = __hip_atomic_load(dg + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					d1 = __hip_atomic_load(dg + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				SdFrameDesc d;
				d.fstart = sd_uniform64(d0);
				d.flen = __builtin_amdgcn_readfirstlane((int)(uint32_t)d1);
				d.inv = __builtin_amdgcn_readfirstlane((int)(d1 >> 32));
				// (inlined: as a real call like RS41's these short decoders keep their small arrays on the stack and cost 7 %: DFM 4096 x 24
				// 0.2886 -> 0.3122 ms, mixed 0.298 -> 0.319)
				if (is_dfm) sd_dfm_decode_frame<true>(wl, ring_g, ring_mask, d, fout + k, ch, lane);
				else if (is_m10) sd_m10_decode_frame<true>(wl, fo->m10tab, ring_g, ring_mask, d, fout + k, ch, lane);
				else if (is_mrz) sd_mrz_decode_frame<true>(wl, ring_g, ring_mask, d, fout + k, ch, lane);
				else sd_ims_decode_frame<true>(wl, fo->gf64, ring_g, ring_mask, d, fout + k, ch, lane);
			}
		}
	}

	// ---- time slices: hand the channel to its next segment.  What it will read (state, history, ring words, framer state, frame
	// count) went out through write-through stores (sd_st_coh); every wave waits until its own have been acknowledged, the barrier
	// collects the waves, one store publishes.  No release fence (an L2-wide write-back per workgroup: see sd_ld_coh above).
	if (sliced) {
		asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		__syncthreads();
		if (tid == 0) sd_st_coh(&sl.prog[ch], sl.seg_base + (uint32_t)seg + 1u);
	}
}

// (every instantiation: 8 waves per SIMD = 64 VGPRs, four workgroups per CU)
template <int IN, bool LIST, int DEC, int NT, bool SLICED = false>
__global__ __launch_bounds__(SD_WGT, 8) void sd_demod_kernel(
	const float *__restrict__ in, size_t ch_stride, int n_tiles_all,
	SdChanState *__restrict__ states, float *__restrict__ hist,
	uint32_t *__restrict__ bitring, uint32_t ring_words,
	const float *__restrict__ taps_all, const SdModem *__restrict__ modems,
	const uint32_t *__restrict__ chlist, int compact_in, const SdFramerOut *__restrict__ fo, int utype, const SdSlice sl)
{
	__shared__ __attribute__((aligned(16))) DemodLds s;
	sd_demod_body<IN, LIST, DEC, NT, SLICED>(s, blockIdx.x, gridDim.x, in, ch_stride, n_tiles_all, states, hist, bitring, ring_words, taps_all, modems, chlist, compact_in, fo, utype, sl);
}

// ONE launch for a batch of the two default classes (round 6): (4, 8) -- RS41, DFM, iMS-100, MRZ-N1 -- and (2, 8) -- M10 --, i.e. BASELINE
// config 3's mixed batch.  Rounds 2-5 launched one kernel per class (or per sonde type) on streams of their own, forked from and joined
// back into the caller's stream: two cross-stream event hops per submit (tens of microseconds of command-processor latency at the
// default, joined completion mode) and two tails.  Here block b takes the next channel of class B whenever floor((b + 1) nB / N) steps
// (a Bresenham walk: the classes are interleaved in proportion, every CU holds both kinds of workgroup -- the M10 class is bound by the
// vector ALU, the other by HBM --), else the next channel of class A.  The body of each class is the very code of its own kernel.
template <int IN>
__global__ __launch_bounds__(SD_WGT, 8) void sd_demod_mixed_kernel(
	const float *__restrict__ in, size_t ch_stride, int n_tiles_all,
	SdChanState *__restrict__ states, float *__restrict__ hist,
	uint32_t *__restrict__ bitring, uint32_t ring_words,
	const float *__restrict__ taps_all, const SdModem *__restrict__ modems,
	const uint32_t *__restrict__ list_a, uint32_t n_a, int utype_a, const uint32_t *__restrict__ list_b, uint32_t n_b, int utype_b,
	const SdFramerOut *__restrict__ fo)
{
	__shared__ __attribute__((aligned(16))) DemodLds s;
	const uint32_t b = blockIdx.x, N = n_a + n_b;
	const uint32_t ib = (uint32_t)(((unsigned long long)b * n_b) / N);                 // class-B blocks in front of this one
	const bool is_b = (uint32_t)(((unsigned long long)(b + 1u) * n_b) / N) > ib;
	const SdSlice sl = { 1, n_tiles_all, N, 0u, nullptr, 0u };
	if (is_b) sd_demod_body<IN, true, 2, 8>(s, ib, N, in, ch_stride, n_tiles_all, states, hist, bitring, ring_words, taps_all, modems, list_b, 0, fo, utype_b, sl);
	else sd_demod_body<IN, true, 4, 8>(s, b - ib, N, in, ch_stride, n_tiles_all, states, hist, bitring, ring_words, taps_all, modems, list_a, 0, fo, utype_a, sl);
}

void sd_launch_demod_mixed(int in_kind, uint32_t n_a, const uint32_t *list_a, int utype_a, uint32_t n_b, const uint32_t *list_b, int utype_b, hipStream_t stream,
	const float *in, size_t ch_stride, int n_tiles, SdChanState *states, float *hist, uint32_t *bitring, uint32_t ring_words,
	const float *taps_all, const SdModem *modems, const SdFramerOut *fo)
{
	const dim3 g(n_a + n_b), blk(SD_WGT);
#define SD_MIXED_ARGS in, ch_stride, n_tiles, states, hist, bitring, ring_words, taps_all, modems, list_a, n_a, utype_a, list_b, n_b, utype_b, fo
	if (in_kind == SD_IN_IQ8) hipLaunchKernelGGL((sd_demod_mixed_kernel<SD_IN_IQ8>), g, blk, 0, stream, SD_MIXED_ARGS);
	else if (in_kind == SD_IN_IQ16) hipLaunchKernelGGL((sd_demod_mixed_kernel<SD_IN_IQ16>), g, blk, 0, stream, SD_MIXED_ARGS);
	else if (in_kind == SD_IN_IQ) hipLaunchKernelGGL((sd_demod_mixed_kernel<SD_IN_IQ>), g, blk, 0, stream, SD_MIXED_ARGS);
	else hipLaunchKernelGGL((sd_demod_mixed_kernel<SD_IN_REAL>), g, blk, 0, stream, SD_MIXED_ARGS);
#undef SD_MIXED_ARGS
}

bool sd_slices_supported(int in_kind, int decim, int nt)
{
	return (in_kind == SD_IN_IQ || in_kind == SD_IN_IQ16 || in_kind == SD_IN_IQ8) && nt == 8 && (decim == 4 || decim == 2);
}

void sd_launch_demod(int in_kind, int decim, int nt, uint32_t n_channels, hipStream_t stream,
	const float *in, size_t ch_stride, int n_tiles, SdChanState *states, float *hist,
	uint32_t *bitring, uint32_t ring_words, const float *taps_all, const SdModem *modems,
	const uint32_t *chlist, bool compact_in, const SdFramerOut *fo /* device memory */, int utype, const SdSlice *slice)
{
	SdSlice sl = { 1, n_tiles, n_channels, 0u, nullptr, 0u };
	const int ci = compact_in ? 1 : 0;
#define SD_DEMOD_ARGS in, ch_stride, n_tiles, states, hist, bitring, ring_words, taps_all, modems, chlist, ci, fo, utype, sl
	if (slice && slice->n_seg > 1 && sd_slices_supported(in_kind, decim, nt)) {
		// the time-sliced instantiations: the two default classes, IQ input of every kind (the classes BASELINE's configurations run)
		sl = *slice;
		const dim3 gs(n_channels * (uint32_t)sl.n_seg), blks(SD_WGT);
#define SD_SLICED_LAUNCH(KIND) do { \
		if (decim == 4 && chlist) hipLaunchKernelGGL((sd_demod_kernel<KIND, true, 4, 8, true>), gs, blks, 0, stream, SD_DEMOD_ARGS); \
		else if (decim == 4) hipLaunchKernelGGL((sd_demod_kernel<KIND, false, 4, 8, true>), gs, blks, 0, stream, SD_DEMOD_ARGS); \
		else if (chlist) hipLaunchKernelGGL((sd_demod_kernel<KIND, true, 2, 8, true>), gs, blks, 0, stream, SD_DEMOD_ARGS); \
		else hipLaunchKernelGGL((sd_demod_kernel<KIND, false, 2, 8, true>), gs, blks, 0, stream, SD_DEMOD_ARGS); } while (0)
		if (in_kind == SD_IN_IQ8) SD_SLICED_LAUNCH(SD_IN_IQ8);
		else if (in_kind == SD_IN_IQ16) SD_SLICED_LAUNCH(SD_IN_IQ16);
		else SD_SLICED_LAUNCH(SD_IN_IQ);
#undef SD_SLICED_LAUNCH
		return;
	}
	const dim3 g(n_channels), blk(SD_WGT);
#define SD_DEMOD_LAUNCH(KIND, LS) do { \
		if (decim == 4) hipLaunchKernelGGL((sd_demod_kernel<KIND, LS, 4, 8>), g, blk, 0, stream, SD_DEMOD_ARGS); \
		else if (decim == 2 && nt == 8) hipLaunchKernelGGL((sd_demod_kernel<KIND, LS, 2, 8>), g, blk, 0, stream, SD_DEMOD_ARGS); \
		else if (decim == 2) hipLaunchKernelGGL((sd_demod_kernel<KIND, LS, 2, 16>), g, blk, 0, stream, SD_DEMOD_ARGS); \
		else hipLaunchKernelGGL((sd_demod_kernel<KIND, LS, 1, 16>), g, blk, 0, stream, SD_DEMOD_ARGS); } while (0)
	if (in_kind == SD_IN_IQ8 && !chlist) SD_DEMOD_LAUNCH(SD_IN_IQ8, false);
	else if (in_kind == SD_IN_IQ8) SD_DEMOD_LAUNCH(SD_IN_IQ8, true);
	else if (in_kind == SD_IN_IQ16 && !chlist) SD_DEMOD_LAUNCH(SD_IN_IQ16, false);
	else if (in_kind == SD_IN_IQ16) SD_DEMOD_LAUNCH(SD_IN_IQ16, true);
	else if (in_kind == SD_IN_IQ && !chlist) SD_DEMOD_LAUNCH(SD_IN_IQ, false);
	else if (in_kind == SD_IN_IQ) SD_DEMOD_LAUNCH(SD_IN_IQ, true);
	else if (!chlist) SD_DEMOD_LAUNCH(SD_IN_REAL, false);
	else SD_DEMOD_LAUNCH(SD_IN_REAL, true);
#undef SD_DEMOD_LAUNCH
#undef SD_DEMOD_ARGS
}
