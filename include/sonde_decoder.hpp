// sonde_decoder.hpp -- the B2 boundary: a C++ adaptor with the behaviour of radiosonde::Decoder<>
// (/root/reference/src/decode/decoder.hpp:21-130) minus the SDR++ dsp::block plumbing, so that it can be
// dropped into a dsp::block::run() or driven directly.
//
// Behaviour mirrored, by reference line:
//   * the same (buf, count) is offered to `get` until it answers PROCEED            decoder.hpp:61
//   * every fragment is merged into one sticky aggregate, field group by field group  decoder.hpp:64-106
//   * PTU fragments also refresh `calibrated` (>= 100 %) and the Magnus dew point     decoder.hpp:84-91
//   * an ozone fragment rewrites auxData as "O3=<2 decimals>mPa", nothing else does   decoder.hpp:102-106
//   * a non-positive pressure is replaced by the ISA value for the current altitude   decoder.hpp:108-110
//   * the callback fires once per fragment whose field mask is non-zero                decoder.hpp:112-114
// The aggregate has the members of SondeFullData (/root/reference/src/decode/common.hpp:4-28); unlike the
// reference, calib_percent is initialised.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include "sonde_abi.h"

namespace sonde {

struct FullData {
	std::string serial;
	int seq = 0;
	time_t time = 0;
	int burstkill = 0;
	float lat = 0, lon = 0, alt = 0;
	float spd = 0, hdg = 0, climb = 0;
	float temp = 0, rh = 0;
	float dewpt = 0, pressure = 0;
	bool calibrated = false;
	float calib_percent = 0;
	std::string auxData;
};

// One fragment into the sticky aggregate, field group by field group (decoder.hpp:64-106), then the pressure
// fall-back (decoder.hpp:108-110).  Returns whether the caller's callback is due (decoder.hpp:112).
inline bool merge_fragment(FullData &m, const SondeData &f)
{
	if (f.fields & DATA_SEQ) m.seq = f.seq;
	if (f.fields & DATA_POS) { m.lat = f.lat; m.lon = f.lon; m.alt = f.alt; }
	if (f.fields & DATA_SPEED) { m.spd = f.speed; m.hdg = f.heading; m.climb = f.climb; }
	if (f.fields & DATA_TIME) m.time = f.time;
	if (f.fields & DATA_PTU) {
		m.calib_percent = f.calib_percent;
		m.calibrated = f.calib_percent >= 100.0f;
		m.temp = f.temp;
		m.rh = f.rh;
		m.pressure = f.pressure;
		m.dewpt = sonde_dewpt(f.temp, f.rh);
	}
	if (f.fields & DATA_SERIAL) m.serial = f.serial;
	if (f.fields & DATA_SHUTDOWN) m.burstkill = f.shutdown;
	if (f.fields & DATA_OZONE) {
		char tmp[64];
		snprintf(tmp, sizeof(tmp), "O3=%.2fmPa", (double)f.o3_mpa);
		m.auxData = tmp;
	}
	if (m.pressure <= 0) m.pressure = sonde_altitude_to_pressure(m.alt);
	return f.fields != 0;
}

template <typename T, T *(*decoder_init)(int), void (*decoder_deinit)(T *),
          ParserStatus (*decoder_get)(T *, SondeData *, const float *, size_t)>
class Decoder {
public:
	typedef void (*Callback)(FullData *data, void *ctx);

	Decoder() = default;
	Decoder(const Decoder &) = delete;
	Decoder &operator=(const Decoder &) = delete;
	~Decoder() { deinit(); }

	// returns false when the underlying decoder could not be created (the reference does not check, decoder.hpp:39)
	bool init(int samplerate, Callback cb, void *ctx)
	{
		deinit();
		m_cb = cb;
		m_ctx = ctx;
		m_dec = decoder_init(samplerate);
		return m_dec != nullptr;
	}

	void deinit()
	{
		if (m_dec) decoder_deinit(m_dec);
		m_dec = nullptr;
	}

	// One stream buffer (what dsp::stream<float>::read() handed out).  Returns the number of callbacks made.
	int process(const float *buf, int count)
	{
		int fired = 0;
		SondeData frag;
		while (decoder_get(m_dec, &frag, buf, (size_t)count) != PROCEED) {
			if (merge_fragment(m_data, frag)) {
				if (m_cb) m_cb(&m_data, m_ctx);
				fired++;
			}
		}
		return fired;
	}

	const FullData &data() const { return m_data; }

private:
	T *m_dec = nullptr;
	Callback m_cb = nullptr;
	void *m_ctx = nullptr;
	FullData m_data;
};

// The same for many channels at once, on top of the batch API: one sticky aggregate per channel, the callback gets
// the channel index as well.  What a host with N narrowband channels on the GPU runs instead of N Decoder<> blocks.
class BatchDecoder {
public:
	typedef void (*Callback)(uint32_t channel, FullData *data, void *ctx);

	BatchDecoder() = default;
	BatchDecoder(const BatchDecoder &) = delete;
	BatchDecoder &operator=(const BatchDecoder &) = delete;
	~BatchDecoder() { deinit(); }

	bool init(const SondeBatchConfig &cfg, Callback cb, void *ctx)
	{
		deinit();
		m_cb = cb;
		m_ctx = ctx;
		if (sonde_batch_create(&cfg, &m_batch) != 0) return false;
		m_data = new FullData[cfg.n_channels];
		return true;
	}

	void deinit()
	{
		if (m_batch) sonde_batch_destroy(m_batch);
		m_batch = nullptr;
		delete[] m_data;
		m_data = nullptr;
	}

	// n_samples more samples of every channel (device pointer, channel-major).  Returns the callbacks made, < 0 on error.
	long process(const void *d_samples, size_t n_samples, size_t channel_stride, void *stream = nullptr)
	{
		if (sonde_batch_submit(m_batch, d_samples, n_samples, channel_stride, stream) != 0) return -1;
		long fired = 0, k;
		SondeData frag[64];
		uint32_t chan[64];
		while ((k = sonde_batch_poll(m_batch, frag, chan, 64)) > 0) {
			for (long i = 0; i < k; i++) {
				if (merge_fragment(m_data[chan[i]], frag[i])) {
					if (m_cb) m_cb(chan[i], &m_data[chan[i]], m_ctx);
					fired++;
				}
			}
		}
		return k < 0 ? k : fired;
	}

	const FullData &data(uint32_t channel) const { return m_data[channel]; }
	SondeBatch *batch() { return m_batch; }

private:
	SondeBatch *m_batch = nullptr;
	FullData *m_data = nullptr;
	Callback m_cb = nullptr;
	void *m_ctx = nullptr;
};

// B3 -- the module's IQ input as a stream pump: complex baseband in, FullData callback out.  This is the level
// BASELINE.json's north_star names ("dsp::stream<dsp::complex_t> in, SondeData callback out"): in the reference the
// VFO hands complex IQ to dsp::demod::FM, whose output goes through the resampler into the decoder block
// (/root/reference/src/main.cpp:55-68); here FM discriminator, timing recovery, framing and FEC all run on the GPU, so
// the host only forwards the VFO's samples.  Two input rates are taken: 48 kS/s (createVFO(..., bandwidth = bw,
// sampleRate = 48000, ...), INTEGRATION.md: the batch API's native input, the FM discriminator runs inside the demod
// kernel), or the reference's own VFO rate for the sonde type (sampleRate = supportedTypes[i].bandwidth, main.hpp:44-52:
// 10/15/20/50 kS/s), in which case the VFO front-end (sonde_vfo_*, vfo.hip: discriminator + rational resampler,
// main.cpp:57-60) turns it into the 48 kS/s FM audio the decoder takes.  Buffers may have any length, samples are
// interleaved I, Q.
class IqStreamDecoder {
public:
	typedef void (*Callback)(FullData *data, void *ctx);

	IqStreamDecoder() = default;
	IqStreamDecoder(const IqStreamDecoder &) = delete;
	IqStreamDecoder &operator=(const IqStreamDecoder &) = delete;
	~IqStreamDecoder() { deinit(); }

	// What a plugin-equivalent host should tolerate is the reference's VFO width (main.hpp:44-52): 20 kHz for iMS-100 and MRZ-N1,
	// 50 kHz for M10.  At 48 kS/s IQ input the library's default classes decimate those three to 12 / 12 / 24 kS/s (SPEC 3.0) and
	// the AFC of SPEC 3.0b pulls in +-2.6 / +-2.6 / +-5.2 kHz; SONDE_FLAG_WIDE doubles the rates and the pull-in range at about 2 dB
	// of sensitivity.  One channel costs nothing either way, so this adaptor defaults to the wide classes for those three types
	// (ADVICE r3); kFlagsAuto = that choice, any explicit value (0 included) is taken as given.
	static constexpr uint32_t kFlagsAuto = 0x80000000u;
	static uint32_t default_flags(int sonde_type)
	{
		return (sonde_type == SONDE_IMS100 || sonde_type == SONDE_MRZN1 || sonde_type == SONDE_M10) ? SONDE_FLAG_WIDE : 0u;
	}

	// sonde_type: SONDE_* (the index into supportedTypes[], main.hpp:44-52).  false: wrong rate or no GPU decoder.
	bool init(int sonde_type, int samplerate, Callback cb, void *ctx, int device = 0, uint32_t flags = kFlagsAuto)
	{
		deinit();
		if (flags == kFlagsAuto) flags = default_flags(sonde_type);
		const uint8_t t = (uint8_t)sonde_type;
		const bool afsk = sonde_type == SONDE_IMET4 || sonde_type == SONDE_C50;
		SondeBatchConfig cfg = SONDE_BATCH_CONFIG_INIT;
		cfg.n_channels = 1;
		cfg.types = &t;
		cfg.device = device;
		cfg.flags = flags;
		m_up = m_down = 1;
		if (samplerate == 48000) {
			cfg.max_samples = kMaxSamples;
			cfg.input_kind = SONDE_INPUT_IQ;
			m_granule = afsk ? 8 * (size_t)SONDE_TILE : (size_t)SONDE_TILE;
			m_max_in = kMaxSamples;
		} else {
			// VFO-rate input: whole decoder tiles come out of input blocks of (3 tiles | 24 tiles with AFSK) * down / up samples
			if (sonde_vfo_ratio(samplerate, &m_up, &m_down) != 0) return false;
			const size_t out_granule = (afsk ? 24 : 3) * (size_t)SONDE_TILE;
			m_granule = out_granule * (size_t)m_down / (size_t)m_up;
			m_max_in = ((size_t)kMaxSamples / m_granule) * m_granule;
			if (m_max_in == 0) m_max_in = m_granule;
			if (m_max_in > kMaxSamples) return false;
			cfg.max_samples = (uint32_t)(m_max_in / (size_t)m_down * (size_t)m_up);
			cfg.input_kind = SONDE_INPUT_REAL;
			if (sonde_vfo_create(1, samplerate, m_max_in, device, &m_vfo) != 0) return false;
		}
		if (sonde_batch_create(&cfg, &m_batch) != 0) { deinit(); return false; }
		m_cb = cb;
		m_ctx = ctx;
		// one object serves every sonde type here (the reference keeps one Decoder<> per type, whose m_data is never reset,
		// decoder.hpp:127, and clears the module's lastData on a type switch, main.cpp:376): a re-init starts an empty aggregate
		m_data = FullData();
		m_n = 0;
		return true;
	}

	void deinit()
	{
		if (m_batch) sonde_batch_destroy(m_batch);
		m_batch = nullptr;
		if (m_vfo) sonde_vfo_destroy(m_vfo);
		m_vfo = nullptr;
	}

	// One stream buffer of `count` complex samples (what dsp::stream<dsp::complex_t>::read() handed out; dsp::complex_t
	// is {float re, im}).  Returns the number of callbacks made, < 0 on error (text: sonde_last_error()).
	int process(const float *iq, int count)
	{
		int fired = 0;
		while (count > 0) {
			const size_t take = std::min((size_t)count, m_max_in - m_n);
			std::memcpy(m_buf + 2 * m_n, iq, take * 2 * sizeof(float));
			m_n += take;
			iq += 2 * take;
			count -= (int)take;
			const size_t n = (m_n / m_granule) * m_granule;
			if (n == 0) continue;
			if (m_vfo) {
				const float *rows = nullptr;
				size_t stride = 0;
				if (sonde_vfo_process_host(m_vfo, m_buf, n, n, &rows, &stride) != 0 ||
				    sonde_batch_submit(m_batch, rows, n / (size_t)m_down * (size_t)m_up, stride, nullptr) != 0) { m_n = 0; return -1; }
				if (sonde_batch_sync(m_batch) < 0) { m_n = 0; return -1; }     // the rows are reused by the next block
			} else if (sonde_batch_submit_host(m_batch, m_buf, n, n) != 0) { m_n = 0; return -1; }
			std::memmove(m_buf, m_buf + 2 * n, (m_n - n) * 2 * sizeof(float));
			m_n -= n;
			long k;
			SondeData frag[32];
			uint32_t chan[32];
			while ((k = sonde_batch_poll(m_batch, frag, chan, 32)) > 0) {
				for (long i = 0; i < k; i++) {
					if (merge_fragment(m_data, frag[i])) {
						if (m_cb) m_cb(&m_data, m_ctx);
						fired++;
					}
				}
			}
			if (k < 0) return -1;
		}
		return fired;
	}

	const FullData &data() const { return m_data; }

private:
	static const uint32_t kMaxSamples = 16 * SONDE_TILE;
	SondeBatch *m_batch = nullptr;
	SondeVfo *m_vfo = nullptr;          // VFO-rate input only
	int m_up = 1, m_down = 1;
	size_t m_max_in = kMaxSamples;
	Callback m_cb = nullptr;
	void *m_ctx = nullptr;
	FullData m_data;
	size_t m_granule = SONDE_TILE, m_n = 0;
	float m_buf[2 * 16 * SONDE_TILE];
};

}  // namespace sonde
