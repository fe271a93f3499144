// sonde_sinks.hpp -- GPX 1.1 track writer and PTU CSV logger producing byte-for-byte the files of the
// reference's sinks (/root/reference/src/gpx.cpp, /root/reference/src/ptu.cpp), which
// sondeDataHandler feeds (/root/reference/src/main.cpp:320-331).
//
// Behaviour kept, by reference line:
//   GPX  * the file is a complete document after every call: the closing tags are rewritten behind the
//          body each time and the body end is remembered                      gpx.cpp:97-121
//        * startTrack: ignored for a name with non-printing characters, a repeat of the current name
//          is a no-op, another name closes the open track first               gpx.cpp:41-61
//        * addTrackPoint: dropped without an open track, for NaN or all-zero positions, for a repeated
//          timestamp or a repeated position                                   gpx.cpp:73-95
//        * number formats "%f", time "%Y-%m-%dT%H:%M:%SZ" (UTC)               gpx.cpp:7,86-91
//   CSV  * header line and "%ld,%.1f,...,%s" row, flushed per point           ptu.cpp:11,28-34
// Checked against the reference's own classes (oracle/_ref) by tests/test_sinks.py.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include "sonde_decoder.hpp"

namespace sonde {

class GpxWriter {
public:
	GpxWriter() = default;
	GpxWriter(const GpxWriter &) = delete;
	GpxWriter &operator=(const GpxWriter &) = delete;
	~GpxWriter() { close(); }

	bool open(const char *path)
	{
		close();
		m_f = fopen(path, "wb");
		if (!m_f) return false;
		m_open_track = false;
		m_last_lat = m_last_lon = m_last_alt = 0;
		m_last_time = 0;
		fputs("<?xml version=\"1.0\" encoding=\"UTF-8\" standalone=\"no\" ?>\n"
		      "<gpx xmlns=\"http://www.topografix.com/GPX/1/1\" version=\"1.1\" creator=\"SDR++\">\n", m_f);
		m_body_end = ftell(m_f);
		seal();
		return true;
	}

	void close()
	{
		if (!m_f) return;
		seal();
		fclose(m_f);
		m_f = nullptr;
	}

	void startTrack(const char *name)
	{
		if (!m_f) return;
		if (m_open_track && m_track == name) return;
		for (const char *p = name; *p; ++p)
			if (!isgraph((unsigned char)*p)) return;
		if (m_open_track) stopTrack();
		m_track.assign(name, strnlen(name, 63));
		append("<trk>\n<name>" + std::string(name) + "</name>\n<trkseg>\n");
		m_open_track = true;
		seal();
	}

	void stopTrack()
	{
		if (!m_f || !m_open_track) return;
		append("</trkseg>\n</trk>\n");
		m_open_track = false;
		seal();
	}

	void addTrackPoint(time_t t, float lat, float lon, float alt, float spd, float hdg)
	{
		if (!m_f || !m_open_track) return;
		if (std::isnan(lat) || std::isnan(lon) || std::isnan(alt)) return;
		if (lat == 0 && lon == 0 && alt == 0) return;
		if (t == m_last_time || (lat == m_last_lat && lon == m_last_lon && alt == m_last_alt)) return;
		m_last_lat = lat; m_last_lon = lon; m_last_alt = alt; m_last_time = t;
		char stamp[32], buf[256];
		struct tm tmv;
		gmtime_r(&t, &tmv);
		strftime(stamp, sizeof(stamp), "%Y-%m-%dT%H:%M:%SZ", &tmv);
		snprintf(buf, sizeof(buf),
		         "<trkpt lat=\"%f\" lon=\"%f\">\n<time>%s</time>\n<ele>%f</ele>\n<speed>%f</speed>\n<course>%f</course>\n</trkpt>\n",
		         lat, lon, stamp, alt, spd, hdg);
		append(buf);
		seal();
	}

private:
	// put text at the end of the body and move the body end behind it
	void append(const std::string &text)
	{
		fseek(m_f, m_body_end, SEEK_SET);
		fwrite(text.data(), 1, text.size(), m_f);
		m_body_end = ftell(m_f);
	}
	// closing tags behind the body; the body end stays where it is so the next append overwrites them
	void seal()
	{
		fseek(m_f, m_body_end, SEEK_SET);
		if (m_open_track) fputs("</trkseg>\n</trk>\n", m_f);
		fputs("</gpx>\n", m_f);
		fflush(m_f);
	}

	FILE *m_f = nullptr;
	long m_body_end = 0;
	bool m_open_track = false;
	std::string m_track;
	float m_last_lat = 0, m_last_lon = 0, m_last_alt = 0;
	time_t m_last_time = 0;
};

class PtuWriter {
public:
	PtuWriter() = default;
	PtuWriter(const PtuWriter &) = delete;
	PtuWriter &operator=(const PtuWriter &) = delete;
	~PtuWriter() { close(); }

	bool open(const char *path)
	{
		close();
		m_f = fopen(path, "wb");
		if (!m_f) return false;
		fputs("Epoch,Temperature,Relative humidity,Dew point,Pressure,Latitude,Longitude,Altitude,Speed,Heading,Climb,XDATA\n", m_f);
		return true;
	}

	void close()
	{
		if (m_f) fclose(m_f);
		m_f = nullptr;
	}

	void addPoint(const FullData &d)
	{
		if (!m_f) return;
		fprintf(m_f, "%ld,%.1f,%.1f,%.1f,%.1f,%.6f,%.6f,%.1f,%.1f,%.1f,%.1f,%s\n", (long)d.time, d.temp, d.rh, d.dewpt, d.pressure,
		        d.lat, d.lon, d.alt, d.spd, d.hdg, d.climb, d.auxData.c_str());
		fflush(m_f);
	}

private:
	FILE *m_f = nullptr;
};

}  // namespace sonde
