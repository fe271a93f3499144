// parse.h -- frame -> SondeData fragments (host side, after FEC; tiny per-frame work).
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/sonde_abi.h"

// Stateful per-channel parser: RS41 calibration data arrives as 51 fragments spread over 51 frames.
class SondeParser {
public:
	explicit SondeParser(int type) : m_type(type) {}
	// Appends the fragments of one frame (one per valid subframe, in frame order).
	void feed(const SondeFrame &f, std::vector<SondeData> &out);
private:
	void feed_rs41(const SondeFrame &f, std::vector<SondeData> &out);
	void feed_dfm(const SondeFrame &f, std::vector<SondeData> &out);
	void feed_m10(const SondeFrame &f, std::vector<SondeData> &out);
	void feed_imet(const SondeFrame &f, std::vector<SondeData> &out);
	void feed_ims100(const SondeFrame &f, std::vector<SondeData> &out);
	void feed_mrzn1(const SondeFrame &f, std::vector<SondeData> &out);
	void feed_c50(const SondeFrame &f, std::vector<SondeData> &out);
	int m_type;
	uint64_t m_calib_mask = 0;         // RS41: which of the 51 calibration fragments have been seen
	uint8_t m_calib[51 * 16] = {};
	// DFM: position/time arrive as separate sub-packets spread over three frames
	double m_dfm_lat = 0, m_dfm_lon = 0, m_dfm_spd = 0, m_dfm_hdg = 0;
	long long m_dfm_date = -1;        // seconds since epoch of the date+hh:mm part, -1 unknown
	int m_dfm_have = 0;               // bit0 lat, bit1 lon
	float m_dfm_meas[5] = {};         // CONF measurement channels 0..4
	unsigned m_dfm_meas_mask = 0;
	// iMS-100 / RS-11G: the calibration words arrive one per frame (frame counter mod 4)
	// SRS-C50: one value per packet
	double m_c50_lat = 0, m_c50_lon = 0;
	int m_c50_have = 0;               // bit0 lat, bit1 lon, bit2 date
	long long m_c50_date = 0;         // seconds since epoch of the date
	uint32_t m_ims_cal[4] = {};
	unsigned m_ims_cal_mask = 0;
};

uint16_t sonde_crc16_ccitt(const uint8_t *p, size_t n);
