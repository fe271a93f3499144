// bins_kernel.hip -- the decoder behind the wideband filter bank (BASELINE config 4; the reference's per-VFO chain
// dsp::demod::FM -> dsp::RationalResampler -> decoder, /root/reference/src/main.cpp:55-60, /root/reference/src/decode/decoder.hpp:22,61,
// for every channelizer bin): ONE WAVE PER BIN, eight bins per workgroup.
//
// A bin's block is 2560 phase samples (20 kS/s; 16-bit fractions of a turn, SPEC 3.5) = 1536 samples at 12 kS/s = 3 tiles of the
// timing loop -- about 615 symbols, one wave's worth of work.  Kernel A (demod_kernel.hip) spends a 512-thread, wave-specialised
// workgroup on a channel because its channels stream megabytes; run over bins (round 4) its 4096 workgroups lived for three tiles
// each: eight waves of prologue, a lead-wave publish / spin hand-over per round and an eight-wave epilogue per 615 symbols, 50 us
// per 8-stream step with the SQ waiting 70 % of the wave-cycles (profiles/r4_wb_counters.csv).  Here everything a bin needs happens
// in one wave, in program order, with no hand-over: discriminator (16-bit wrapped phase difference) + composite 12/5 resampler -
// 4:1 decimator (SPEC 3.5b) -> LDS tile -> low-pass at the timing loop's instants, Gardner detector, slicer (SPEC 3.2, four symbols
// per lane) -> PI loop filter -> bit ring -> sync search (K4) -> RS41: de-whitening + RS(255,231) of the frames completed in
// this submit (K5 / K6).  The arithmetic is SPEC 3.2 / 3.3 / 3.5b to the bit: same fmaf chains, same integer statistics; frames,
// bits and loop state equal the oracle's (tests/test_channelizer.py).
#include <hip/hip_runtime.h>
#include <atomic>
#include "sonde_dev.h"
#include "sd_math.h"
#include "sd_wave.h"
#include "sd_rs41.h"
#include "sd_rsdec.h"
#include "sd_fixed.h"
#include "launch.h"

#define BK_WAVES 8                         // bins per workgroup
#define BK_IT    (SD_TILE / 4)             // decimated samples per tile: 512
#define BK_SPILL 128                       // a pass of 192 outputs may run past the tile it completes
#define BK_A     (SD_LH + BK_IT + BK_SPILL + 4)
#define BK_PASS_PH 336                     // phases a pass reads: 320 new + the 16 before them
#define BK_NT    8                         // taps per polyphase row (class (4, 8): 2.4-2.5 samples per symbol)
#define BK_SLOTS 4                         // tap tables per workgroup: one per sonde type a bin can carry

struct BinsWaveLds {
	float A[BK_A];                          // [64 history | the tile | the next tile's first samples]
	alignas(16) float d[BK_PASS_PH + 8];    // the pass's discriminator samples (16-bit differences as floats), index = phase element
	uint32_t chunk[10];                     // the round's bits: [0] = 0, [1..8] one ballot half each, [9] = 0
	uint32_t mirror[SD_MIRROR_WORDS];       // the newest 2048 bits of the bit ring (K4 reads them)
	SdFrameDesc k4list[SD_K4_LIST];         // the first frames K4 listed in this launch (the FEC epilogue reads them)
};
struct BinsLds {
	BinsWaveLds w[BK_WAVES];
	alignas(16) float taps[BK_SLOTS][SD_NPHASE * SD_TAPS_LD];
	EpiTabs et;
};
static_assert(sizeof(FramerLds) <= sizeof(float) * BK_A, "the FEC work area aliases the tile buffer");
static_assert(sizeof(BinsLds) <= 80 * 1024, "two workgroups per CU: 16 bins in flight per CU, 4096 bins in one generation");

__device__ __forceinline__ int bk_slot(int type) { return type == SONDE_RS41 ? 0 : type == SONDE_DFM09 ? 1 : type == SONDE_IMS100 ? 2 : 3; }

#ifdef BK_TS      // experiment: cycle stamps of one wave's phases (make EXTRA=-DBK_TS; tools/bk_ts.py reads them)
__device__ unsigned long long g_bk_ts[64];
extern "C" int sonde_debug_bins_ts(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bk_ts), sizeof(g_bk_ts)) == hipSuccess ? 0 : -1; }
#define BK_STAMP(i) do { if (ch == 1000u && lane == 0) g_bk_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define BK_STAMP(i) do { } while (0)
#endif

// What a wave would otherwise fetch through a chain of dependent loads at the head of its life (descriptor -> table pointer -> table,
// state -> type -> modem): by value in the kernel arguments.  Measured (tools/bk_ts.py, 8 streams): the head of a wave's life
// 17 000 cycles -> see profiles/r5_notes.md.
struct SdBinsParams {
	SdFramerOut fo;
	SdModem modems[SONDE_NTYPES];
	int utype;                              // >= 0: every bin is of this sonde type (taps and modem do not wait for the state); -1: per bin
};

// No workgroup barrier anywhere: a wave shares nothing with its neighbours but read-only tables, which every wave that needs one
// writes itself (identical values: the stores may race) and reads behind its own stores.
__global__ __launch_bounds__(64 * BK_WAVES, 4) void sd_bins_kernel(
	const int16_t *__restrict__ phases, size_t row_stride /* int16 elements per bin row: [16 carried | n_steps] */, int n_blocks,
	int16_t *__restrict__ carry_rows, size_t carry_stride /* where the last 16 phases go: the head of the row the next submit reads */,
	SdChanState *__restrict__ states, float *__restrict__ hist, uint32_t *__restrict__ bitring, uint32_t ring_words,
	const float *__restrict__ taps_all, const float *__restrict__ g_comp /* [3][SD_RS_KT_LD], x 2^-14: the discriminator samples stay integers */, uint32_t n_channels, const SdBinsParams P)
{
	__shared__ __attribute__((aligned(16))) BinsLds s;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t ch = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave;      // (the launcher picks the waves per workgroup: sd_launch_bins)
	if (ch >= n_channels) return;
	BinsWaveLds &w = s.w[wave];
	BK_STAMP(0);

	// ---- every load that does not depend on another goes out now, in the order of first use: the composite taps, the block's
	// phases (24 dwords per lane: a block is 8 passes of 168 dwords), the state, the history, K4's state, the tap table
	const uint32_t *row32 = reinterpret_cast<const uint32_t *>(phases + (size_t)ch * row_stride);
	// a pass reads 336 phases = 42 lanes x 16 bytes; four register sets: the loads of pass g + 3 go out when pass g starts, so that the
	// requests of the 4096 waves of a step spread over the kernel instead of arriving together (all 24 MB at once took the memory
	// system 15 000 cycles during which no wave could start: tools/bk_ts.py, profiles/r5_notes.md)
	const uint4 *row128 = reinterpret_cast<const uint4 *>(row32);
	uint4 ph[4];
	const int n_pass = 8 * n_blocks;
	auto load_pass = [&](int gp, uint4 &p) {
		// pass gp reads phase elements [320 gp - 16, 320 gp + 320) of the submit = row elements [320 gp, 320 gp + 336)
		p = lane < 42 ? row128[40 * (size_t)gp + lane] : make_uint4(0u, 0u, 0u, 0u);
	};
	load_pass(0, ph[0]); load_pass(1, ph[1]); load_pass(2, ph[2]);
	// the channel's state and K4's state come through VECTOR loads (a dword per lane, fields by v_readlane): 4096 waves asking the
	// scalar data cache for 64 private bytes each at the same moment waited 5 800 cycles for them (tools/bk_ts.py); a vector load 500
	const uint32_t sv = lane < 16 ? reinterpret_cast<const uint32_t *>(states + ch)[lane] : (lane < 24 ? reinterpret_cast<const uint32_t *>(P.fo.fstates + ch)[lane - 16] : 0u);
	// (of the 64 carried samples the filters of a tile's first symbols reach back 13 at most -- t_next lies behind the previous tile's
	// limit, 8 samples before its end, the 8-tap rows and the mid-symbol instant take 5 more: only the newest 16 travel, 1.5 MB less HBM
	// traffic per 8-stream step)
	const float hv = lane >= SD_HIST - 16 ? hist[(size_t)ch * SD_HIST + lane] : 0.0f;
	auto svw = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)sv, i); };
	auto sv64 = [&](int i) { return (uint64_t)svw(i) | ((uint64_t)svw(i + 1) << 32); };
	SdChanState st;
	st.t_next = (int64_t)sv64(0); st.n0 = (int64_t)sv64(2); st.wpos = sv64(4); st.period = (int32_t)svw(6); st.nstat = (int32_t)svw(7);
	st.bias = __uint_as_float(svw(8)); st.amp = __uint_as_float(svw(9)); st.iq_last[0] = __uint_as_float(svw(10)); st.iq_last[1] = __uint_as_float(svw(11));
	st.type = (int32_t)svw(12); st.afc[0] = __uint_as_float(svw(13)); st.afc[1] = __uint_as_float(svw(14)); st.afc[2] = __uint_as_float(svw(15));
	SdFramerState f0;
	f0.rpos = sv64(16); f0.fstart = sv64(18); f0.collecting = (int32_t)svw(20); f0.inv = (int32_t)svw(21); f0.flen = (int32_t)svw(22); f0.pad = 0;
	const int stype = P.utype >= 0 ? P.utype : st.type;
	const SdModem md = P.modems[stype];
	const bool is_rs41 = stype == SONDE_RS41, is_dfm = stype == SONDE_DFM09, is_ims = stype == SONDE_IMS100, is_mrz = stype == SONDE_MRZN1;
	const bool fuse = P.fo.fuse_fec != 0;
	const bool framing = is_rs41 || (fuse && (is_dfm || is_ims || is_mrz));
	const bool fec_here = is_rs41 && fuse;
	uint32_t *ring_g = bitring + (size_t)ch * ring_words;
	const uint32_t ring_mask = ring_words - 1;
	// the ring's newest words (K4's window into the past): the address needs the state, which has arrived; requested now, used before tile 0
	// (round 5: requested there it cost a full memory latency, 1 200 of the wave's 34 600 ticks: tools/bk_ts.py)
	const uint32_t wi = (uint32_t)(st.wpos >> 5) - (uint32_t)(63 - lane);
	const uint32_t xw = __hip_atomic_load(ring_g + (wi & ring_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	float *const taps = s.taps[bk_slot(stype)];
	{
		// the type's tap table into its slot, pair-swapped rows as interp() wants them (bins of one type write identical values)
		const float4 *taps_g = reinterpret_cast<const float4 *>(taps_all + (size_t)stype * SD_NPHASE * SD_NTAPS);
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int i = lane + 64 * q, rowi = i >> 3, c4 = i & 7;      // 32 rows x 8 float4
			const float4 tv = taps_g[i];
			*reinterpret_cast<float4 *>(&taps[rowi * SD_TAPS_LD + 4 * c4]) = make_float4(tv.y, tv.x, tv.w, tv.z);
		}
	}
	if (lane < 10) w.chunk[lane] = 0u;
	WAVE_SYNC();
	BK_STAMP(1);

	// ---- one pass: 336 phases -> 335 wrapped differences -> 192 decimated samples n = 192 gp .. 192 gp + 191 (SPEC 3.5b)
	auto pass = [&](const uint4 p, int out0 /* A index of the pass's first output */) {
		{
			// lane l < 42 holds phase elements 8 l .. 8 l + 7 (two per dword); the element before its first is the high half of the lane
			// below's last dword (DPP wave_shr:1; lane 0's is never used).  v_pk_sub_u16 wraps: the FM discriminator of SPEC 3.5
			typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
			const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p.w, 0x138, 0xF, 0xF, false);
			const uint32_t x0 = __builtin_amdgcn_alignbit(p.x, below, 16), x1 = __builtin_amdgcn_alignbit(p.y, p.x, 16),
			               x2 = __builtin_amdgcn_alignbit(p.z, p.y, 16), x3 = __builtin_amdgcn_alignbit(p.w, p.z, 16);
			const uint32_t e0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, p.x) - __builtin_bit_cast(u16x2, x0));
			const uint32_t e1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, p.y) - __builtin_bit_cast(u16x2, x1));
			const uint32_t e2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, p.z) - __builtin_bit_cast(u16x2, x2));
			const uint32_t e3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, p.w) - __builtin_bit_cast(u16x2, x3));
			if (lane < 42) {
				float4 *dst = reinterpret_cast<float4 *>(&w.d[8 * lane]);
				dst[0] = make_float4((float)(int16_t)(e0 & 0xffffu), (float)((int32_t)e0 >> 16), (float)(int16_t)(e1 & 0xffffu), (float)((int32_t)e1 >> 16));
				dst[1] = make_float4((float)(int16_t)(e2 & 0xffffu), (float)((int32_t)e2 >> 16), (float)(int16_t)(e3 & 0xffffu), (float)((int32_t)e3 >> 16));
			}
		}
		WAVE_SYNC();
		// lane <-> group u = 64 gp + lane: the three decimated samples that d[5u .. 5u + 4] complete read d[5u - 15 .. 5u + 4] =
		// elements 5 lane + 1 .. 5 lane + 20 (lanes 20 bytes apart: conflict-free), statically indexed; tap rows broadcast
		// (the table's address through an opaque copy per pass: left to itself the compiler loads the 51 taps ONCE and keeps them in scalar
		// registers for the whole kernel -- half of the wave's 102 -- and spills 118 other scalars into VGPR lanes: 530 v_readlane /
		// v_writelane in the tile code; reloaded per pass they are scalar-cache hits)
		typedef const __attribute__((address_space(4))) float *CTab;      // (constant address space: scalar loads)
		unsigned long long g_addr = (unsigned long long)g_comp;
		asm volatile("" : "+s"(g_addr));
		const CTab g_pass = (CTab)g_addr;
		float dv[20];
		const float *dp = &w.d[5 * lane + 1];
#pragma unroll
		for (int k = 0; k < 20; k++) dv[k] = dp[k];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			constexpr int BO[3] = {1, 2, 4};
			const int bo = BO[c] + 15;
			// the composite tap row: wave-uniform addresses of a read-only table = scalar loads, the taps SGPR operands of the multiply-adds
			// (as broadcast LDS reads they were 15 of a pass's 31 LDS instructions, and the LDS pipe is this kernel's busiest unit)
			const CTab gt = g_pass + SD_RS_KT_LD * c;
			float acc = 0.0f;
#pragma unroll
			for (int t = 0; t < SD_RS_KT; t++) acc = __builtin_fmaf(gt[t], dv[bo - t], acc);
			w.A[out0 + 3 * lane + c] = acc;
		}
		WAVE_SYNC();
	};

	// ---- one tile of the timing loop (SPEC 3.2): lane <-> symbols lane + 64 h, h < 4
	int64_t n0 = st.n0;
	uint64_t wpos = st.wpos;
	uint32_t partial = 0;
	SdSyncRun k4;                              // K4's state: registers of this wave (the list of frames: LDS)
	auto tile_rounds = [&]() {
		n0 += BK_IT;
		const int64_t limit = (((n0 - 1 - BK_NT / 2 - SD_MARGIN) << 16) | 0xFFFF);
		int K = (st.t_next <= limit) ? (int)((uint32_t)(limit - st.t_next) / (uint32_t)st.period) + 1 : 0;
		if (K > SD_ROUND_MAX) K = SD_ROUND_MAX;
		const int64_t base = (n0 - BK_IT - SD_LH) << 16;
		const uint32_t rel0 = (uint32_t)(st.t_next - base);
		const float bias = st.bias;
		// every lane evaluates its four symbols, wanted or not (a symbol beyond K reads in-bounds LDS behind the tile and is masked
		// below): no branch between the eight filter evaluations, their LDS reads overlap
		int Ei = 0, S1i = 0, S0i = 0, C1 = 0;
#pragma unroll
		for (int hh = 0; hh < 4; hh += 2) {
			float y[2], m[2];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const uint32_t rel = rel0 + __umul24((unsigned)(sd_hw_symbol(lane) + 64 * (hh + h)), (unsigned)st.period);      // (the half-wave symbol mapping, sd_wave.h)
				y[h] = interp<BK_NT>(w.A, taps, rel);
				m[h] = interp<BK_NT>(w.A, taps, rel - ((uint32_t)st.period >> 1));
			}
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const bool act = sd_hw_symbol(lane) + 64 * (hh + h) < K;
				const float yy = act ? y[h] : 0.0f;
				const float yprev = sd_hw_prev(yy, lane);
				float e = (yprev - yy) * (m[h] - bias);
				e = sd_clamp(e * 1024.0f, -1.0e6f, 1.0e6f);
				Ei += (act && lane != 0) ? __float2int_rn(e) : 0;           // the first symbol of a 64-group carries no term
				const bool bit = act && (yy > bias);
				const int Y = __float2int_rn(sd_clamp(yy, -8.0f, 8.0f) * 4096.0f);
				S1i += bit ? Y : 0;
				S0i += (act && !bit) ? Y : 0;
				const unsigned long long bal = __ballot(sd_hw_unpermute(bit ? 1 : 0, lane) != 0);      // bits in symbol order
				C1 += __popcll(bal);
				if (lane == 0) { w.chunk[1 + 2 * (hh + h)] = (uint32_t)bal; w.chunk[2 + 2 * (hh + h)] = (uint32_t)(bal >> 32); }
			}
		}
		const int E = wave_sum(Ei), S1 = wave_sum(S1i), S0 = wave_sum(S0i);
		WAVE_SYNC();
		if (K <= 0) return;
		// ---- the round's bits into the ring (HBM) and its mirror
		if (lane < 9) {
			const uint32_t sh = (uint32_t)wpos & 31u, w0 = (uint32_t)(wpos >> 5);
			uint32_t vv = 0;
			if ((uint32_t)(32 * lane) < sh + (uint32_t)K) {
				const uint32_t lo = w.chunk[lane + 1], pvw = w.chunk[lane];
				vv = sh ? ((lo << sh) | (pvw >> (32u - sh))) : lo;
				if (lane == 0 && sh) vv |= partial & ((1u << sh) - 1u);
				const uint32_t idx = (w0 + (uint32_t)lane) & ring_mask;
				ring_g[idx] = vv;
				w.mirror[idx & (SD_MIRROR_WORDS - 1)] = vv;
			}
			partial = vv;
		}
		partial = (uint32_t)__builtin_amdgcn_readlane((int)partial, (int)((((uint32_t)wpos & 31u) + (uint32_t)K) >> 5));
		wpos += (uint64_t)K;
		// ---- slicer levels and the PI loop filter (the lead wave's round_back of kernel A, here in line)
		const int C0 = K - C1;
		if (C1 > 0 && C0 > 0) {
			const f32x2 cnt = {(float)C1, (float)C0};
			const f32x2 rc = sd_recip2(cnt);
			const float hi = ((float)S1 * rc.x) * (1.0f / 4096.0f);
			const float lo = ((float)S0 * rc.y) * (1.0f / 4096.0f);
			const float c = 0.5f * (hi + lo), a = 0.5f * (hi - lo);
			if (st.nstat == 0) { st.bias = c; st.amp = a; }
			else { st.bias = st.bias + 0.5f * (c - st.bias); st.amp = st.amp + 0.5f * (a - st.amp); }
			if (!(st.amp >= 1.0e-3f)) st.amp = 1.0e-3f;
			st.nstat = 1;
		} else {
			st.bias = ((float)(S1 + S0) * sd_recip((float)K)) * (1.0f / 4096.0f);
		}
		if (n0 <= 3 * (int64_t)BK_IT) st.bias = ((float)(S1 + S0) * sd_recip((float)K)) * (1.0f / 4096.0f);      // SPEC 3.2b: acquisition, the mean of the round
		const f32x2 den = {(float)K, st.amp * st.amp};
		const f32x2 rd = sd_recip2(den);
		float err = ((float)E * rd.x) * (1.0f / 1024.0f);
		err = err * rd.y;
		err = sd_clamp(err, -1.0f, 1.0f);
		const int dphase = __float2int_rn(err * md.kp), dper = __float2int_rn(err * md.ki);
		st.t_next += (int64_t)K * st.period + dphase;
		st.period += dper;
		if (st.period < md.pmin) st.period = md.pmin;
		if (st.period > md.pmax) st.period = md.pmax;
	};
	// ---- K4: the sync search over the bits that are now in the mirror.  ONCE PER BLOCK (round 6; per tile before): a block adds at most
	// 3 x 256 bits, the mirror holds the newest 2048, and what the search finds does not depend on how its calls are cut -- three calls
	// of ~2 000 ns each (tools/bk_ts.py: mostly the latency of the state machine's scalar control flow, not its 64 positions per lane)
	// were a fifth of a wave's life
	auto k4_block = [&]() {
		if (!framing) return;
		WAVE_SYNC();
		SdFrameDesc *dch = (SdFrameDesc *)P.fo.descs + (size_t)ch * P.fo.max_frames;
		const uint32_t mf = P.fo.max_frames;
		if (is_rs41) sd_rs41_sync_step<true>(k4, wpos, w.mirror, lane, dch, mf, w.k4list);
		else if (is_dfm) sd_fixed_sync_step<SONDE_DFM09, true>(k4, wpos, w.mirror, lane, dch, mf, w.k4list);
		else if (is_ims) sd_fixed_sync_step<SONDE_IMS100, true>(k4, wpos, w.mirror, lane, dch, mf, w.k4list);
		else sd_fixed_sync_step<SONDE_MRZN1, true>(k4, wpos, w.mirror, lane, dch, mf, w.k4list);
	};
	// the tile is consumed: its last 64 samples become the history, what the last pass produced beyond it the head of the next tile
	auto roll = [&]() {
		const float h0 = w.A[BK_IT + lane], h1 = w.A[BK_IT + 64 + lane], h2 = w.A[BK_IT + 128 + lane];
		WAVE_SYNC();
		w.A[lane] = h0; w.A[64 + lane] = h1; w.A[128 + lane] = h2;
		WAVE_SYNC();
	};

	// ---- a block = 8 passes = 3 tiles: passes 0-2 complete tile 0 (64 samples beyond it), 3-5 tile 1 (128 beyond), 6-7 tile 2
	for (int b = 0; b < n_blocks; b++) {
		const int gp = 8 * b;
		load_pass(gp + 3, ph[3]);                         pass(ph[0], SD_LH + 0);     
		load_pass(gp + 4, ph[0]);                         pass(ph[1], SD_LH + 192);   
		load_pass(gp + 5, ph[1]);                         pass(ph[2], SD_LH + 384);   BK_STAMP(5);
		if (b == 0) {
			// what the timing loop needs from the head of the wave's life has arrived by now: history, the ring's newest words, K4's state
			w.A[lane] = hv;
			w.mirror[wi & (SD_MIRROR_WORDS - 1)] = xw;
			partial = ((uint32_t)st.wpos & 31u) ? (uint32_t)__builtin_amdgcn_readlane((int)xw, 63) : 0u;
			k4.rpos = f0.rpos; k4.fstart = f0.fstart; k4.collecting = f0.collecting; k4.inv = f0.inv; k4.flen = f0.flen; k4.nout = 0; k4.wp_seen = 0;
			WAVE_SYNC();
			BK_STAMP(2);
		}
		tile_rounds(); BK_STAMP(6); roll(); 
		load_pass(gp + 6, ph[2]);                         pass(ph[3], SD_LH + 64);    
		load_pass(gp + 7, ph[3]);                         pass(ph[0], SD_LH + 256);   
		if (gp + 8 < n_pass) load_pass(gp + 8, ph[0]);    pass(ph[1], SD_LH + 448);   BK_STAMP(10);
		tile_rounds(); BK_STAMP(11); roll();
		if (gp + 9 < n_pass) load_pass(gp + 9, ph[1]);    pass(ph[2], SD_LH + 128);   
		if (gp + 10 < n_pass) load_pass(gp + 10, ph[2]);  pass(ph[3], SD_LH + 320);   BK_STAMP(13);
		tile_rounds(); BK_STAMP(14); roll();
		k4_block();
	}

	// ---- epilogue: history, state, the carried phases, K4's state; RS41: the frames listed in this submit
	if (lane >= SD_HIST - 16) hist[(size_t)ch * SD_HIST + lane] = w.A[lane];
	if (lane < 8) {
		const size_t n_ph = 2560 * (size_t)n_blocks;
		reinterpret_cast<uint32_t *>(carry_rows + (size_t)ch * carry_stride)[lane] = row32[n_ph / 2 + lane];      // row elements [n_ph, n_ph + 16) = the last 16 phases
	}
	if (lane == 0) {
		st.n0 = n0;
		st.wpos = wpos;
		states[ch] = st;
		if (framing) {
			SdFramerState f1;
			f1.rpos = k4.rpos; f1.fstart = k4.fstart; f1.collecting = k4.collecting; f1.inv = k4.inv; f1.flen = k4.flen; f1.pad = 0;
			P.fo.fstates[ch] = f1;
			P.fo.counts[ch] = k4.nout;
		}
	}
	BK_STAMP(15);
	if (fec_here) {
		const uint32_t max_frames = P.fo.max_frames;
		const uint32_t nfr = min((uint32_t)__builtin_amdgcn_readfirstlane((int)k4.nout), max_frames);
		if (nfr) {
			// the GF(2^8) tables: whoever has a frame writes them (identical values), then reads behind its own stores
			for (int i = lane; i < GF_EXP2 / 16 + 512 / 16 + RS_R * 8 * 4 / 16; i += 64) {
				if (i < GF_EXP2 / 16) reinterpret_cast<uint4 *>(s.et.tabs.exp2)[i] = reinterpret_cast<const uint4 *>(P.fo.gf_exp)[i];
				else if (i < GF_EXP2 / 16 + 512 / 16) reinterpret_cast<uint4 *>(s.et.tabs.log2)[i - GF_EXP2 / 16] = reinterpret_cast<const uint4 *>(P.fo.gf_log)[i - GF_EXP2 / 16];
				else reinterpret_cast<uint4 *>(s.et.swar)[i - (GF_EXP2 / 16 + 512 / 16)] = reinterpret_cast<const uint4 *>(P.fo.gf_swar)[i - (GF_EXP2 / 16 + 512 / 16)];
			}
			WAVE_SYNC();
			FramerLds &wl = *reinterpret_cast<FramerLds *>(&w.A[0]);
			GfSwar swar;
			const uint32_t *sw = s.et.swar + 8 * (lane % RS_R);
			swar.a_lo = sw[0]; swar.a_hi = sw[1]; swar.b_lo = sw[2]; swar.b_hi = sw[3]; swar.c = sw[4];
			const unsigned long long *dg = reinterpret_cast<const unsigned long long *>((const SdFrameDesc *)P.fo.descs + (size_t)ch * max_frames);
			SondeFrame *fout = P.fo.frames + (size_t)ch * max_frames;
			for (uint32_t k = 0; k < nfr; k++) {
				unsigned long long d0, d1;
				if (k < SD_K4_LIST) {
					const unsigned long long *dl = reinterpret_cast<const unsigned long long *>(&w.k4list[k]);
					d0 = dl[0]; d1 = dl[1];
				} else {
					d0 = __hip_atomic_load(dg + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					d1 = __hip_atomic_load(dg + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				SdFrameDesc d;
				d.fstart = sd_uniform64(d0);
				d.flen = __builtin_amdgcn_readfirstlane((int)(uint32_t)d1);
				d.inv = __builtin_amdgcn_readfirstlane((int)(d1 >> 32));
				sd_rs41_decode_frame<true>(s.et.tabs, wl, swar, ring_g, ring_mask, d, fout + k, ch, lane);
			}
		}
	}
	BK_STAMP(16);
}

void sd_launch_bins(uint32_t n_channels, hipStream_t stream, const int16_t *phases, size_t row_stride, int n_blocks,
	int16_t *carry_rows, size_t carry_stride, SdChanState *states, float *hist, uint32_t *bitring, uint32_t ring_words,
	const float *taps_all, const SdModem *modems_host, const SdFramerOut *fo_host, const float *g_comp, int utype)
{
	// Waves (= bins) per workgroup: 8 for launches that fill the GPU anyway; small launches (one stream = 512 bins) spread over all CUs -- a
	// tile's rounds are bound by the VALU share a wave gets on its SIMD (tools/bk_ts.py): 512 waves as 64 workgroups sit two to a SIMD
	// on a quarter of the CUs, as 256 workgroups of two waves each has a SIMD to itself
	int wpw = BK_WAVES;
	{
		static std::atomic<int> cus_of[64];                      // (per device, asked once: 0 = not asked yet; atomics: hosts submit from several threads)
		int dev = 0, cus = 256;
		if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
			int known = cus_of[dev].load(std::memory_order_relaxed);
			if (!known && hipDeviceGetAttribute(&known, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && known > 0) cus_of[dev].store(known, std::memory_order_relaxed);
			if (known > 0) cus = known;
		}
		while (wpw > 1 && (n_channels + wpw - 1) / wpw < (uint32_t)cus) wpw >>= 1;
	}
	const dim3 g((n_channels + wpw - 1) / wpw), blk(64 * wpw);
	SdBinsParams P;
	P.fo = *fo_host;
	for (int t = 0; t < SONDE_NTYPES; t++) P.modems[t] = modems_host[t];
	P.utype = utype;
	hipLaunchKernelGGL(sd_bins_kernel, g, blk, 0, stream, phases, row_stride, n_blocks, carry_rows, carry_stride, states, hist, bitring,
	                   ring_words, taps_all, g_comp, n_channels, P);
}
