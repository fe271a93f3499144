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
			merge(frag);
			if (m_data.pressure <= 0) m_data.pressure = sonde_altitude_to_pressure(m_data.alt);
			if (frag.fields) {
				if (m_cb) m_cb(&m_data, m_ctx);
				fired++;
			}
		}
		return fired;
	}

	const FullData &data() const { return m_data; }

private:
	void merge(const SondeData &f)
	{
		if (f.fields & DATA_SEQ) m_data.seq = f.seq;
		if (f.fields & DATA_POS) { m_data.lat = f.lat; m_data.lon = f.lon; m_data.alt = f.alt; }
		if (f.fields & DATA_SPEED) { m_data.spd = f.speed; m_data.hdg = f.heading; m_data.climb = f.climb; }
		if (f.fields & DATA_TIME) m_data.time = f.time;
		if (f.fields & DATA_PTU) {
			m_data.calib_percent = f.calib_percent;
			m_data.calibrated = f.calib_percent >= 100.0f;
			m_data.temp = f.temp;
			m_data.rh = f.rh;
			m_data.pressure = f.pressure;
			m_data.dewpt = sonde_dewpt(f.temp, f.rh);
		}
		if (f.fields & DATA_SERIAL) m_data.serial = f.serial;
		if (f.fields & DATA_SHUTDOWN) m_data.burstkill = f.shutdown;
		if (f.fields & DATA_OZONE) {
			char tmp[64];
			snprintf(tmp, sizeof(tmp), "O3=%.2fmPa", (double)f.o3_mpa);
			m_data.auxData = tmp;
		}
	}

	T *m_dec = nullptr;
	Callback m_cb = nullptr;
	void *m_ctx = nullptr;
	FullData m_data;
};

}  // namespace sonde
