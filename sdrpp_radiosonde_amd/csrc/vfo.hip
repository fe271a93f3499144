// vfo.hip -- VFO front-end (SURVEY.md section 8 rows a1 + a2 at the reference's own rates): what the reference puts
// between the SDR++ VFO and the decoder, /root/reference/src/main.cpp:55-60 --
//     VFO at supportedTypes[i].bandwidth (main.hpp:44-52: 10, 15, 20 or 50 kS/s complex)
//       -> dsp::demod::FM<float>                      (K1, sd_math.h)
//       -> dsp::RationalResampler to 48 kHz           (24/5, 16/5, 12/5, 24/25; 6/5 for a 40 kS/s channelizer bin)
// for a batch of channels of one rate.  The 48 kS/s rows it writes are the real-input rows of sonde_batch_submit()
// (SONDE_INPUT_REAL).  A host that already has 48 kS/s IQ (the batch API's native input) never needs this; the
// sonde::IqStreamDecoder adaptor uses it when SDR++ hands it the VFO-rate stream.
//
// One workgroup per (channel, chunk of 1600 input samples): the chunk and the 16 samples in front of it are
// discriminated into LDS, the polyphase FIR (16 taps per phase, fixed-order fmaf chain, SPEC 3.7) reads them back.
// Chunks are independent (the 16 samples of filter history are recomputed from the IQ in front of the chunk); only the
// first chunk of a submit needs the carried state, which is double-buffered so that the last chunk's update cannot
// race with the first chunk's read.  HBM traffic: 8 B in + (up/down) * 4 B out per input sample, read once.
// Bit-exact against oracle/or_chan.c or_vfo_process.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string>
#include <vector>
#include "sd_math.h"
#include "launch.h"
#include "../../include/sonde_abi.h"

#define VF_T      16          // taps per phase
#define VF_CHUNK  1600        // input samples per workgroup (a multiple of every `down`)
#define VF_MAXUP  24

struct SdVfoState {           // per channel, 128 B
	float iq_last[2];
	float dhist[VF_T];        // the last 16 discriminator samples, oldest first
	float pad[14];
};

__global__ __launch_bounds__(256) void sd_vfo_kernel(const float2 *__restrict__ in, size_t ch_stride, uint32_t n_in, int up, int down,
	const float *__restrict__ g, const SdVfoState *__restrict__ st_in, SdVfoState *__restrict__ st_out,
	float *__restrict__ out, size_t out_stride)
{
	__shared__ float s_d[VF_T + VF_CHUNK];
	__shared__ float s_g[VF_MAXUP * VF_T];
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t ch = blockIdx.y, c0 = blockIdx.x * VF_CHUNK;
	const uint32_t n = min((uint32_t)VF_CHUNK, n_in - c0);
	const float2 *x = in + (size_t)ch * ch_stride;
	for (int i = tid; i < up * VF_T; i += 256) s_g[i] = g[i];
	// discriminator samples c0 - 16 .. c0 + n - 1 -> s_d[0 .. 16 + n): one 8-byte load per lane (512 B per wave), the
	// predecessor from the neighbouring lane; lane 0 of a wave fetches its own (or takes the carried sample)
	const uint32_t total = VF_T + n, trips = (total + 255u) & ~255u;     // whole waves take part in every trip (the shuffle)
	for (uint32_t i = tid; i < trips; i += 256) {
		const long idx = (long)c0 - VF_T + (long)i;
		const bool live = i < total && idx >= 0;
		float2 cur = make_float2(0.0f, 0.0f);
		if (live) cur = x[idx];
		float px = __shfl_up(cur.x, 1, 64), py = __shfl_up(cur.y, 1, 64);
		if (live && (lane == 0 || idx == 0)) {
			if (idx == 0) { px = st_in[ch].iq_last[0]; py = st_in[ch].iq_last[1]; }
			else { const float2 pv = x[idx - 1]; px = pv.x; py = pv.y; }
		}
		if (i < total) s_d[i] = live ? sd_disc(cur.x, cur.y, px, py) : st_in[ch].dhist[i];     // idx < 0: the first chunk's carried history
	}
	__syncthreads();
	const uint32_t nj = n / (uint32_t)down * (uint32_t)up;
	float *o = out + (size_t)ch * out_stride + (size_t)(c0 / (uint32_t)down) * (uint32_t)up;
	for (uint32_t jj = tid; jj < nj; jj += 256) {
		const uint32_t q = jj * (uint32_t)down;
		const uint32_t i0 = q / (uint32_t)up, p = q - i0 * (uint32_t)up;
		const float *gp = s_g + p * VF_T, *dp = s_d + VF_T + i0;
		float acc = 0.0f;
#pragma unroll
		for (int t = 0; t < VF_T; t++) acc = __builtin_fmaf(gp[t], dp[-t], acc);
		o[jj] = acc;
	}
	if (c0 + n == n_in) {      // the last chunk carries the state to the next submit
		if (tid < VF_T) st_out[ch].dhist[tid] = s_d[n + tid];
		if (tid == 0) { const float2 l = x[n_in - 1]; st_out[ch].iq_last[0] = l.x; st_out[ch].iq_last[1] = l.y; }
	}
}

// ---------------------------------------------------------------- host object
struct SondeVfo {
	int device = 0;
	uint32_t n_channels = 0;
	int rate_in = 0, up = 0, down = 0;
	size_t max_in = 0;
	float *d_g = nullptr;
	SdVfoState *d_state[2] = {};
	unsigned parity = 0;
	float2 *d_stage = nullptr;      // sonde_vfo_process_host: staged input and the output rows
	float *d_out = nullptr;
};

static int vfo_ratio(int rate_in, int *up, int *down, double *cutoff_hz)
{
	switch (rate_in) {          // 48000 / rate_in in lowest terms; cutoff 0.45 of the lower rate (SPEC 3.7)
	case 10000: *up = 24; *down = 5;  *cutoff_hz = 4500.0;  return 0;      // RS41            (main.hpp:45)
	case 15000: *up = 16; *down = 5;  *cutoff_hz = 6750.0;  return 0;      // DFM06/09        (main.hpp:46)
	case 20000: *up = 12; *down = 5;  *cutoff_hz = 9000.0;  return 0;      // iMS-100, iMet-4, SRS-C50, MRZ-N1 (main.hpp:47,49-51)
	case 40000: *up = 6;  *down = 5;  *cutoff_hz = 18000.0; return 0;      // a channelizer bin (channelizer.hip)
	case 50000: *up = 24; *down = 25; *cutoff_hz = 21600.0; return 0;      // M10/M20         (main.hpp:48)
	}
	return -1;
}

static void vfo_taps(int up, double fs_up, double cutoff_hz, std::vector<float> &g)
{
	const double PI = 3.14159265358979323846;
	const int N = up * VF_T;
	const double fc = cutoff_hz / fs_up;
	std::vector<double> tmp(N);
	for (int i = 0; i < N; i++) {
		const double t = (double)i - 0.5 * (double)(N - 1);
		const double x = (double)i / (double)(N - 1);
		const double w = 0.42 - 0.5 * cos(2.0 * PI * x) + 0.08 * cos(4.0 * PI * x);
		const double s = (t == 0.0) ? 2.0 * fc : sin(2.0 * PI * fc * t) / (PI * t);
		tmp[i] = s * w;
	}
	g.resize(N);
	for (int p = 0; p < up; p++) {
		double sum = 0.0;
		for (int t = 0; t < VF_T; t++) sum += tmp[t * up + p];
		for (int t = 0; t < VF_T; t++) g[p * VF_T + t] = (float)(tmp[t * up + p] / sum);
	}
}

extern "C" int sonde_vfo_ratio(int rate_in, int *up, int *down)
{
	double fc;
	int u, d;
	if (vfo_ratio(rate_in, &u, &d, &fc)) return sd_fail("sonde_vfo_ratio: rate_in must be 10000, 15000, 20000, 40000 or 50000");
	if (up) *up = u;
	if (down) *down = d;
	return 0;
}

extern "C" int sonde_vfo_taps(int rate_in, float *g /* up * 16 */)
{
	double fc;
	int u, d;
	if (!g || vfo_ratio(rate_in, &u, &d, &fc)) return sd_fail("sonde_vfo_taps: bad argument");
	std::vector<float> v;
	vfo_taps(u, (double)rate_in * u, fc, v);
	for (size_t i = 0; i < v.size(); i++) g[i] = v[i];
	return 0;
}

extern "C" void sonde_vfo_destroy(SondeVfo *v)
{
	if (!v) return;
	(void)hipSetDevice(v->device);
	(void)hipFree(v->d_g); (void)hipFree(v->d_state[0]); (void)hipFree(v->d_state[1]); (void)hipFree(v->d_stage); (void)hipFree(v->d_out);
	delete v;
}

extern "C" int sonde_vfo_create(uint32_t n_channels, int rate_in, size_t max_in, int device, SondeVfo **out)
{
	if (!out || !n_channels || !max_in) return sd_fail("sonde_vfo_create: bad argument");
	int up, down;
	double fc;
	if (vfo_ratio(rate_in, &up, &down, &fc)) return sd_fail("sonde_vfo_create: rate_in must be 10000, 15000, 20000, 40000 or 50000");
	if (max_in % (size_t)down) return sd_fail("sonde_vfo_create: max_in must be a multiple of the ratio's denominator (5; 25 at 50 kS/s)");
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || device < 0 || device >= ndev) return sd_fail("sonde_vfo_create: no such HIP device (this library has no CPU path)", e);
	if ((e = hipSetDevice(device)) != hipSuccess) return sd_fail("hipSetDevice", e);
	SondeVfo *v = new SondeVfo;
	v->device = device; v->n_channels = n_channels; v->rate_in = rate_in; v->up = up; v->down = down; v->max_in = max_in;
	std::vector<float> g;
	vfo_taps(up, (double)rate_in * up, fc, g);
	bool ok = hipMalloc((void **)&v->d_g, g.size() * sizeof(float)) == hipSuccess &&
	          hipMalloc((void **)&v->d_state[0], n_channels * sizeof(SdVfoState)) == hipSuccess &&
	          hipMalloc((void **)&v->d_state[1], n_channels * sizeof(SdVfoState)) == hipSuccess;
	ok = ok && hipMemcpy(v->d_g, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemset(v->d_state[0], 0, n_channels * sizeof(SdVfoState)) == hipSuccess &&
	     hipMemset(v->d_state[1], 0, n_channels * sizeof(SdVfoState)) == hipSuccess;
	if (!ok) { sonde_vfo_destroy(v); return sd_fail("sonde_vfo_create: device allocation failed"); }
	*out = v;
	return 0;
}

extern "C" size_t sonde_vfo_out_samples(const SondeVfo *v, size_t n_in) { return v ? n_in / (size_t)v->down * (size_t)v->up : 0; }

extern "C" int sonde_vfo_process(SondeVfo *v, const void *iq_dev, size_t n_in, size_t channel_stride, float *out48_dev, size_t out_stride, void *stream)
{
	if (!v || !iq_dev || !out48_dev) return sd_fail("sonde_vfo_process: null argument");
	if (!n_in || n_in > v->max_in || n_in % (size_t)v->down) return sd_fail("sonde_vfo_process: n_in must be a multiple of the ratio's denominator and <= max_in");
	if (channel_stride < n_in || out_stride < sonde_vfo_out_samples(v, n_in)) return sd_fail("sonde_vfo_process: stride shorter than the row");
	if ((uintptr_t)iq_dev & 7u) return sd_fail("sonde_vfo_process: iq must be 8-byte aligned");
	hipError_t e = hipSetDevice(v->device);
	if (e != hipSuccess) return sd_fail("hipSetDevice", e);
	const dim3 grid((unsigned)((n_in + VF_CHUNK - 1) / VF_CHUNK), v->n_channels);
	hipLaunchKernelGGL(sd_vfo_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float2 *)iq_dev, channel_stride, (uint32_t)n_in, v->up, v->down,
		v->d_g, v->d_state[v->parity & 1], v->d_state[(v->parity + 1) & 1], out48_dev, out_stride);
	if ((e = hipGetLastError()) != hipSuccess) return sd_fail("sd_vfo_kernel launch", e);
	v->parity++;
	return 0;
}

// Host buffers in, device rows out (for hosts that hold the VFO stream in host memory, like an SDR++ module): the rows
// stay in an internal buffer whose address is returned; they are valid until the next call (stream-ordered on the null stream).
extern "C" int sonde_vfo_process_host(SondeVfo *v, const void *iq_host, size_t n_in, size_t channel_stride, const float **out48_dev, size_t *out_stride)
{
	if (!v || !iq_host || !out48_dev) return sd_fail("sonde_vfo_process_host: null argument");
	hipError_t e = hipSetDevice(v->device);
	if (e != hipSuccess) return sd_fail("hipSetDevice", e);
	const size_t n_out_max = sonde_vfo_out_samples(v, v->max_in);
	if (!v->d_stage) {
		if ((e = hipMalloc((void **)&v->d_stage, (size_t)v->n_channels * v->max_in * sizeof(float2))) != hipSuccess ||
		    (e = hipMalloc((void **)&v->d_out, (size_t)v->n_channels * n_out_max * sizeof(float))) != hipSuccess)
			return sd_fail("sonde_vfo_process_host: device allocation failed", e);
	}
	if (!n_in || n_in > v->max_in || channel_stride < n_in) return sd_fail("sonde_vfo_process_host: bad n_in / stride");
	e = hipMemcpy2DAsync(v->d_stage, v->max_in * sizeof(float2), iq_host, channel_stride * sizeof(float2), n_in * sizeof(float2), v->n_channels,
		hipMemcpyHostToDevice, nullptr);
	if (e != hipSuccess) return sd_fail("sonde_vfo_process_host: upload", e);
	if (sonde_vfo_process(v, v->d_stage, n_in, v->max_in, v->d_out, n_out_max, nullptr)) return -1;
	*out48_dev = v->d_out;
	if (out_stride) *out_stride = n_out_max;
	return 0;
}
