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
#include <cstdio>
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

}  // namespace sonde
