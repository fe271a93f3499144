// node.cpp -- libsonde_rccl.so: the node-level host of include/sonde_node.h: ONE process, the GPUs of one node, one decoder
// batch per GPU (north_star: "C++ host code ... sharded across the 8 GPUs of one node with an RCCL scatter of IQ blocks over
// xGMI"; the reference's unit is one module instance per channel, any number of them, /root/reference/src/main.cpp:18-24).
//
// Channels shard in contiguous ranges (sonde_node_shard_range).  The only exchange is ingest: the IQ of all channels arrives on one
// device and every other device receives its shard over xGMI STRAIGHT INTO THE ROWS ITS DECODER READS, as groups of ncclSend
// (ingest device) / ncclRecv (peers) -- RCCL has no scatter primitive, and point-to-point transfers let the ingest device drive
// all its xGMI links at once.  ONLY THE ROWS' BYTES TRAVEL (round 6): xGMI is the slowest link of the system (7 x ~153 GB/s against
// 8 TB/s of HBM), so ingest rows that do not lie back to back are PACKED on the ingest device first (a strided device copy into a
// staging buffer, chunk by chunk, the next chunk packed while the current one is on the links) and the peers decode from
// back-to-back rows (a decoder takes any stride; back-to-back rows cost it ~3-5 %, the padding of the recommended stride would cost
// the links 33 %).  The ingest device's own shard is a strided device copy, not a send to itself.
// Frames come back per device straight to host memory (one process: nothing travels back over xGMI).
// (Round 5 had a second, rank-per-GPU stack beside this one -- shard_rccl.cpp / sonde_shard_* -- with the same transfer logic
// written twice; round 6 retired it: VERDICT r5 item 5d.  A rank-per-GPU host shards with plain torch.distributed / MPI and runs one
// sonde_batch per rank: sdrpp_radiosonde_amd/shard.py.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>
#include "../../include/sonde_node.h"

static thread_local std::string g_nerr;
static int nfail(const char *what, const char *detail = nullptr)
{
	g_nerr = what;
	if (detail && *detail) { g_nerr += ": "; g_nerr += detail; }
	return -1;
}
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return nfail(#x, ncclGetErrorString(r_)); } while (0)
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return nfail(#x, hipGetErrorString(e_)); } while (0)
#define BCHK(x) do { if ((x) < 0) return nfail(#x, sonde_last_error()); } while (0)

extern "C" const char *sonde_node_last_error(void) { return g_nerr.c_str(); }

struct SondeNode {
	uint32_t nd = 0, ingest = 0, n_channels = 0, max_samples = 0;
	int kind = SONDE_INPUT_IQ;
	size_t elem = 8;
	std::vector<int> dev;
	std::vector<uint32_t> first, count;
	std::vector<SondeBatch *> batch;
	std::vector<ncclComm_t> comm;
	std::vector<hipStream_t> st;
	// per device: its shard's rows, max stride apart -- TWICE: submit t scatters into set t & 1 while the decoders of submit t - 1 may
	// still read the other set (the per-device batches join the caller's stream one submit late, include/sonde_abi.h: when the
	// scatter of submit t is queued on a device's stream, that stream is already ordered behind submit t - 2, the last reader of set t & 1)
	std::vector<void *> rows, rows_b;
	// packing (ingest rows not back to back): staging on the ingest device, two buffers (chunk c is on the links while c + 1 is packed on
	// the pack stream); events: pack of buffer j done / the group that sent buffer j done
	void *stage[2] = { nullptr, nullptr };
	size_t stage_bytes = 0;
	hipStream_t s_pack = nullptr;
	hipEvent_t ev_pack[2] = { nullptr, nullptr }, ev_sent[2] = { nullptr, nullptr };
	bool sent_valid[2] = { false, false };
	bool shared_devices = false;           // TEST HOOK (SONDE_NODE_TEST_SHARED_DEVICES): a HIP device may be listed more than once
	uint64_t n_submits = 0;
	size_t rows_stride_max = 0;            // elements between rows at max_samples
	uint32_t scatter_mode = 0;
	hipEvent_t ev_in = nullptr;            // the caller's stream at sonde_node_submit_on: the ingest buffer is complete there
	double gather_ms = 0.0;                // host time of the last sonde_node_frames (device -> host copies of every device), and its bytes
	uint64_t gather_bytes = 0;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool have_scatter = false;
	uint64_t last_bytes = 0;
	uint32_t last_sends = 0;
	std::vector<SondeFrame> tmp;
};

extern "C" void sonde_node_destroy(SondeNode *n)
{
	if (!n) return;
	for (uint32_t d = 0; d < n->nd; d++) {
		(void)hipSetDevice(n->dev[d]);
		if (d < n->batch.size() && n->batch[d]) sonde_batch_destroy(n->batch[d]);
		if (d < n->st.size() && n->st[d]) { (void)hipStreamSynchronize(n->st[d]); (void)hipStreamDestroy(n->st[d]); }
		if (d < n->rows.size() && n->rows[d]) (void)hipFree(n->rows[d]);
		if (d < n->rows_b.size() && n->rows_b[d]) (void)hipFree(n->rows_b[d]);
		if (d < n->comm.size() && n->comm[d]) (void)ncclCommDestroy(n->comm[d]);
	}
	if (n->nd) (void)hipSetDevice(n->dev[n->ingest]);
	if (n->s_pack) { (void)hipStreamSynchronize(n->s_pack); (void)hipStreamDestroy(n->s_pack); }
	for (int j = 0; j < 2; j++) { if (n->stage[j]) (void)hipFree(n->stage[j]); if (n->ev_pack[j]) (void)hipEventDestroy(n->ev_pack[j]); if (n->ev_sent[j]) (void)hipEventDestroy(n->ev_sent[j]); }
	if (n->ev0) (void)hipEventDestroy(n->ev0);
	if (n->ev1) (void)hipEventDestroy(n->ev1);
	if (n->ev_in) (void)hipEventDestroy(n->ev_in);
	delete n;
}

extern "C" int sonde_node_create(const SondeNodeConfig *cfg, SondeNode **out)
{
	if (!cfg || !out) return nfail("sonde_node_create: null argument");
	*out = nullptr;
	if (cfg->n_devices < 1 || cfg->n_devices > 16 || cfg->ingest >= cfg->n_devices) return nfail("sonde_node_create: n_devices must be 1..16 and ingest one of them");
	if (cfg->n_channels < cfg->n_devices) return nfail("sonde_node_create: fewer channels than devices");
	if (cfg->input_kind != SONDE_INPUT_IQ && cfg->input_kind != SONDE_INPUT_REAL && cfg->input_kind != SONDE_INPUT_IQ16 && cfg->input_kind != SONDE_INPUT_IQ8)
		return nfail("sonde_node_create: bad input_kind");
	int ndev = 0;
	HCHK(hipGetDeviceCount(&ndev));
	SondeNode *n = new SondeNode;
	n->nd = cfg->n_devices; n->ingest = cfg->ingest; n->n_channels = cfg->n_channels; n->max_samples = cfg->max_samples;
	n->kind = cfg->input_kind;
	n->elem = sonde_sample_bytes(cfg->input_kind);
	n->dev.resize(n->nd); n->first.resize(n->nd); n->count.resize(n->nd);
	n->batch.assign(n->nd, nullptr); n->st.assign(n->nd, nullptr); n->rows.assign(n->nd, nullptr); n->rows_b.assign(n->nd, nullptr); n->comm.assign(n->nd, nullptr);
	n->scatter_mode = cfg->scatter_mode & 0xFFu;
	n->shared_devices = (cfg->scatter_mode & SONDE_NODE_TEST_SHARED_DEVICES) != 0;
	if (n->scatter_mode > 1) { delete n; return nfail("sonde_node_create: scatter_mode must be 0 or 1"); }
	for (uint32_t d = 0; d < n->nd; d++) {
		n->dev[d] = cfg->devices ? cfg->devices[d] : (int)d;
		if (n->dev[d] < 0 || n->dev[d] >= ndev) { delete n; return nfail("sonde_node_create: no such HIP device"); }
		for (uint32_t e = 0; e < d && !n->shared_devices; e++) if (n->dev[e] == n->dev[d]) { delete n; return nfail("sonde_node_create: a device is listed twice"); }
		sonde_node_shard_range(cfg->n_channels, n->nd, d, &n->first[d], &n->count[d]);
	}
	n->rows_stride_max = sonde_row_stride(cfg->max_samples, cfg->input_kind);
	for (uint32_t d = 0; d < n->nd; d++) {
		hipError_t e = hipSetDevice(n->dev[d]);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&n->st[d], hipStreamNonBlocking);
		if (e == hipSuccess) e = hipMalloc(&n->rows[d], (size_t)n->count[d] * n->rows_stride_max * n->elem);
		if (e == hipSuccess) e = hipMalloc(&n->rows_b[d], (size_t)n->count[d] * n->rows_stride_max * n->elem);
		if (e != hipSuccess) { sonde_node_destroy(n); return nfail("sonde_node_create: stream / rows", hipGetErrorString(e)); }
		SondeBatchConfig bc = SONDE_BATCH_CONFIG_INIT;
		bc.n_channels = n->count[d];
		bc.types = cfg->types ? cfg->types + n->first[d] : nullptr;
		bc.max_samples = cfg->max_samples;
		bc.input_kind = cfg->input_kind;
		bc.device = n->dev[d];
		// (SONDE_FLAG_PIPELINE never joins a device's stream with its decoders: the node could not tell when a row set may be
		// rewritten; masked.  SONDE_FLAG_LATE_JOIN is fine: the row sets exist twice, and when the scatter of submit t is queued on a
		// device's stream that stream is already ordered behind submit t - 2, the last reader of set t & 1)
		bc.flags = cfg->flags & ~SONDE_FLAG_PIPELINE;
		if (sonde_batch_create(&bc, &n->batch[d]) != 0) { sonde_node_destroy(n); return nfail("sonde_batch_create", sonde_last_error()); }
	}
	if (n->nd > 1) {
		ncclResult_t r = ncclCommInitAll(n->comm.data(), (int)n->nd, n->dev.data());
		if (r != ncclSuccess) { for (auto &c : n->comm) c = nullptr; sonde_node_destroy(n); return nfail("ncclCommInitAll", ncclGetErrorString(r)); }
	}
	hipError_t e = hipSetDevice(n->dev[n->ingest]);
	if (e == hipSuccess) e = hipEventCreate(&n->ev0);
	if (e == hipSuccess) e = hipEventCreate(&n->ev1);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_in, hipEventDisableTiming);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&n->s_pack, hipStreamNonBlocking);
	for (int j = 0; j < 2 && e == hipSuccess; j++) {
		e = hipEventCreateWithFlags(&n->ev_pack[j], hipEventDisableTiming);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_sent[j], hipEventDisableTiming);
	}
	if (e != hipSuccess) { sonde_node_destroy(n); return nfail("sonde_node_create: events", hipGetErrorString(e)); }
	*out = n;
	return 0;
}

// The channel range of device index d of nd: contiguous, the remainder spread over the first devices.
extern "C" void sonde_node_shard_range(uint32_t n_channels, uint32_t nd, uint32_t d, uint32_t *first, uint32_t *count)
{
	if (nd == 0) nd = 1;
	const uint32_t base = n_channels / nd, rem = n_channels % nd;
	if (first) *first = d * base + (d < rem ? d : rem);
	if (count) *count = base + (d < rem ? 1u : 0u);
}

extern "C" uint32_t sonde_node_devices(const SondeNode *n) { return n ? n->nd : 0; }
extern "C" int sonde_node_range(const SondeNode *n, uint32_t d, uint32_t *first, uint32_t *count)
{
	if (!n || d >= n->nd) return nfail("sonde_node_range: bad argument");
	if (first) *first = n->first[d];
	if (count) *count = n->count[d];
	return 0;
}
extern "C" SondeBatch *sonde_node_batch(SondeNode *n, uint32_t d) { return (n && d < n->nd) ? n->batch[d] : nullptr; }

static int check_submit(const SondeNode *n, size_t n_samples, size_t channel_stride)
{
	if (n_samples == 0 || n_samples % SONDE_TILE || n_samples > n->max_samples) return nfail("sonde_node_submit: n_samples must be a multiple of SONDE_TILE and <= max_samples");
	if (channel_stride < n_samples) return nfail("sonde_node_submit: channel_stride < n_samples");
	return 0;
}

extern "C" int sonde_node_submit(SondeNode *n, const void *samples, size_t n_samples, size_t channel_stride)
{
	return sonde_node_submit_on(n, samples, n_samples, channel_stride, nullptr);
}

extern "C" int sonde_node_submit_on(SondeNode *n, const void *samples, size_t n_samples, size_t channel_stride, void *stream)
{
	if (!n || !samples) return nfail("sonde_node_submit: null argument");
	if (check_submit(n, n_samples, channel_stride)) return -1;
	const size_t rs_reco = sonde_row_stride(n_samples, n->kind);     // the stride the decoders like best (elements)
	const size_t row_bytes = n_samples * n->elem;
	const char *src = (const char *)samples;
	const uint32_t gi = n->ingest;
	hipStream_t si = n->st[gi];
	const std::vector<void *> &rows = (n->n_submits & 1) ? n->rows_b : n->rows;
	n->n_submits++;
	HCHK(hipSetDevice(n->dev[gi]));
	// the ingest buffer is produced by work on the CALLER's stream (NULL: the legacy default stream): the scatter, which runs on the
	// node's own non-blocking streams, starts behind it (it used to start unordered)
	HCHK(hipEventRecord(n->ev_in, (hipStream_t)stream));
	HCHK(hipStreamWaitEvent(si, n->ev_in, 0));
	HCHK(hipEventRecord(n->ev0, si));
	n->last_bytes = 0; n->last_sends = 0;
	// Shape of the transfer.  ONLY THE ROWS' BYTES CROSS xGMI (round 6; round 5 sent the padding of strided ingest rows too: 1.33 x):
	//   ingest rows back to back      -> ONE send per peer of exactly the shard's bytes, straight from the ingest buffer;
	//   any other ingest stride       -> PACK: N_CHUNKS chunks of rows; chunk c of every peer is copied (strided -> back to back) into
	//                                    staging buffer c & 1 on the pack stream while chunk c - 1 is on the links; one send per peer and
	//                                    chunk of exactly the rows' bytes;
	//   scatter_mode 1 (no staging)   -> one send per row, 256 rows of every peer per group, rows land on the recommended stride.
	// In the first two shapes a peer's rows land BACK TO BACK and are decoded from there (the decoder takes any stride; it costs it
	// ~3-5 % (bench.py contiguous_layout), the padding would cost the slowest link of the system a third).
	const bool per_row = n->scatter_mode == 1;
	const bool direct = !per_row && channel_stride == n_samples;
	const size_t rs_peer = per_row ? rs_reco : n_samples;               // destination stride (elements) on the peers
	if (n->nd > 1) {
		uint32_t max_rows = 0;
		for (uint32_t d = 0; d < n->nd; d++) if (d != gi) max_rows = std::max(max_rows, n->count[d]);
		if (direct) {
			NCHK(ncclGroupStart());
			for (uint32_t d = 0; d < n->nd; d++) {
				if (d == gi) continue;
				const size_t bytes = (size_t)n->count[d] * row_bytes;
				ncclResult_t r = ncclSend(src + (size_t)n->first[d] * row_bytes, bytes, ncclChar, (int)d, n->comm[gi], si);
				if (r == ncclSuccess) r = ncclRecv(rows[d], bytes, ncclChar, (int)gi, n->comm[d], n->st[d]);
				if (r != ncclSuccess) { (void)ncclGroupEnd(); return nfail("ncclSend/ncclRecv", ncclGetErrorString(r)); }
				n->last_bytes += bytes; n->last_sends++;
			}
			NCHK(ncclGroupEnd());
		} else if (per_row) {
			for (uint32_t done_rows = 0; done_rows < max_rows; done_rows += 256u) {
				NCHK(ncclGroupStart());
				for (uint32_t d = 0; d < n->nd; d++) {
					if (d == gi) continue;
					const uint32_t hi = std::min(n->count[d], done_rows + 256u);
					for (uint32_t row = done_rows; row < hi; row++) {
						ncclResult_t r = ncclSend(src + ((size_t)n->first[d] + row) * channel_stride * n->elem, row_bytes, ncclChar, (int)d, n->comm[gi], si);
						if (r == ncclSuccess) r = ncclRecv((char *)rows[d] + (size_t)row * rs_peer * n->elem, row_bytes, ncclChar, (int)gi, n->comm[d], n->st[d]);
						if (r != ncclSuccess) { (void)ncclGroupEnd(); return nfail("ncclSend/ncclRecv", ncclGetErrorString(r)); }
						n->last_bytes += row_bytes; n->last_sends++;
					}
				}
				NCHK(ncclGroupEnd());
			}
		} else {
			const uint32_t N_CHUNKS = 4;
			const uint32_t chunk_rows = (max_rows + N_CHUNKS - 1) / N_CHUNKS;
			// staging: the chunk of every peer back to back, sized for max_samples rows (allocated at the first packed submit)
			size_t peers_rows = 0;
			for (uint32_t d = 0; d < n->nd; d++) if (d != gi) peers_rows += std::min(chunk_rows, n->count[d]);
			const size_t need = peers_rows * (size_t)n->max_samples * n->elem;
			if (n->stage_bytes < need) {
				HCHK(hipStreamSynchronize(n->s_pack)); HCHK(hipStreamSynchronize(si));
				for (int j = 0; j < 2; j++) { if (n->stage[j]) (void)hipFree(n->stage[j]); n->stage[j] = nullptr; }
				n->stage_bytes = 0;
				for (int j = 0; j < 2; j++) HCHK(hipMalloc(&n->stage[j], need));
				n->stage_bytes = need;
				n->sent_valid[0] = n->sent_valid[1] = false;
			}
			HCHK(hipStreamWaitEvent(n->s_pack, n->ev_in, 0));
			uint32_t c = 0;
			for (uint32_t r0 = 0; r0 < max_rows; r0 += chunk_rows, c++) {
				const int j = (int)(c & 1u);
				if (n->sent_valid[j]) HCHK(hipStreamWaitEvent(n->s_pack, n->ev_sent[j], 0));      // the group that last read this buffer is done
				size_t off = 0;
				for (uint32_t d = 0; d < n->nd; d++) {
					if (d == gi || r0 >= n->count[d]) continue;
					const uint32_t nr = std::min(chunk_rows, n->count[d] - r0);
					HCHK(hipMemcpy2DAsync((char *)n->stage[j] + off, row_bytes, src + ((size_t)n->first[d] + r0) * channel_stride * n->elem,
					                      channel_stride * n->elem, row_bytes, nr, hipMemcpyDeviceToDevice, n->s_pack));
					off += (size_t)nr * row_bytes;
				}
				HCHK(hipEventRecord(n->ev_pack[j], n->s_pack));
				HCHK(hipStreamWaitEvent(si, n->ev_pack[j], 0));
				NCHK(ncclGroupStart());
				off = 0;
				for (uint32_t d = 0; d < n->nd; d++) {
					if (d == gi || r0 >= n->count[d]) continue;
					const uint32_t nr = std::min(chunk_rows, n->count[d] - r0);
					const size_t bytes = (size_t)nr * row_bytes;
					ncclResult_t r = ncclSend((const char *)n->stage[j] + off, bytes, ncclChar, (int)d, n->comm[gi], si);
					if (r == ncclSuccess) r = ncclRecv((char *)rows[d] + (size_t)r0 * row_bytes, bytes, ncclChar, (int)gi, n->comm[d], n->st[d]);
					if (r != ncclSuccess) { (void)ncclGroupEnd(); return nfail("ncclSend/ncclRecv", ncclGetErrorString(r)); }
					n->last_bytes += bytes; n->last_sends++;
					off += bytes;
				}
				NCHK(ncclGroupEnd());
				HCHK(hipSetDevice(n->dev[gi]));
				HCHK(hipEventRecord(n->ev_sent[j], si));
				n->sent_valid[j] = true;
			}
		}
		HCHK(hipSetDevice(n->dev[gi]));
	}
	// the ingest device's own shard: a strided device copy on its stream (not a send to itself), onto the recommended stride
	HCHK(hipMemcpy2DAsync(rows[gi], rs_reco * n->elem, src + (size_t)n->first[gi] * channel_stride * n->elem, channel_stride * n->elem,
	                      row_bytes, n->count[gi], hipMemcpyDeviceToDevice, si));
	HCHK(hipEventRecord(n->ev1, si));
	n->have_scatter = true;
	for (uint32_t d = 0; d < n->nd; d++) {
		HCHK(hipSetDevice(n->dev[d]));
		BCHK(sonde_batch_submit(n->batch[d], rows[d], n_samples, d == gi ? rs_reco : rs_peer, (void *)n->st[d]));
	}
	return 0;
}

extern "C" int sonde_node_submit_local(SondeNode *n, const void *const *rows, size_t n_samples, size_t channel_stride)
{
	if (!n || !rows) return nfail("sonde_node_submit_local: null argument");
	if (check_submit(n, n_samples, channel_stride)) return -1;
	n->have_scatter = false;
	n->last_bytes = 0; n->last_sends = 0;
	for (uint32_t d = 0; d < n->nd; d++) {
		if (!rows[d]) return nfail("sonde_node_submit_local: null rows pointer");
		HCHK(hipSetDevice(n->dev[d]));
		BCHK(sonde_batch_submit(n->batch[d], rows[d], n_samples, channel_stride, (void *)n->st[d]));
	}
	return 0;
}

extern "C" int sonde_node_scatter_done(SondeNode *n)
{
	if (!n) return nfail("sonde_node_scatter_done: null argument");
	if (!n->have_scatter) return 0;
	HCHK(hipSetDevice(n->dev[n->ingest]));
	HCHK(hipEventSynchronize(n->ev1));
	return 0;
}

extern "C" long sonde_node_sync(SondeNode *n)
{
	if (!n) return nfail("sonde_node_sync: null argument");
	long total = 0;
	for (uint32_t d = 0; d < n->nd; d++) {
		const long k = sonde_batch_sync(n->batch[d]);
		if (k < 0) return nfail("sonde_batch_sync", sonde_last_error());
		total += k;
	}
	return total;
}

extern "C" long sonde_node_frames(SondeNode *n, SondeFrame *out, size_t cap)
{
	if (!n) return nfail("sonde_node_frames: null argument");
	size_t k = 0;
	for (uint32_t d = 0; d < n->nd; d++)            // every device done first: what is timed below is the gather alone
		if (sonde_batch_sync(n->batch[d]) < 0) return nfail("sonde_batch_sync", sonde_last_error());
	const auto t0 = std::chrono::steady_clock::now();
	for (uint32_t d = 0; d < n->nd; d++) {          // contiguous ascending ranges: device order is node-wide channel order
		const long m = sonde_batch_sync(n->batch[d]);
		if (m < 0) return nfail("sonde_batch_sync", sonde_last_error());
		if (!out || cap == 0) { k += (size_t)m; continue; }        // count only
		const size_t take = std::min((size_t)m, cap - k);
		if (take == 0) continue;
		const long got = sonde_batch_frames(n->batch[d], out + k, take);
		if (got < 0) return nfail("sonde_batch_frames", sonde_last_error());
		for (long i = 0; i < got; i++) out[k + (size_t)i].channel += n->first[d];
		k += (size_t)got;
	}
	if (out && cap) {
		n->gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		n->gather_bytes = (uint64_t)k * sizeof(SondeFrame);
	}
	return (long)k;
}

extern "C" int sonde_node_gather_stats(SondeNode *n, double *ms, uint64_t *bytes)
{
	if (!n) return nfail("sonde_node_gather_stats: null argument");
	if (ms) *ms = n->gather_ms;
	if (bytes) *bytes = n->gather_bytes;
	return 0;
}

extern "C" long sonde_node_poll(SondeNode *n, SondeData *out, uint32_t *channel, size_t cap)
{
	if (!n || !out || !channel) return nfail("sonde_node_poll: null argument");
	size_t k = 0;
	for (uint32_t d = 0; d < n->nd && k < cap; d++) {
		const long got = sonde_batch_poll(n->batch[d], out + k, channel + k, cap - k);
		if (got < 0) return nfail("sonde_batch_poll", sonde_last_error());
		for (long i = 0; i < got; i++) channel[k + (size_t)i] += n->first[d];
		k += (size_t)got;
	}
	return (long)k;
}

extern "C" int sonde_node_scatter_stats(SondeNode *n, float *ms, uint64_t *bytes_out, uint32_t *n_sends)
{
	if (!n) return nfail("sonde_node_scatter_stats: null argument");
	float t = 0.0f;
	if (n->have_scatter) {
		HCHK(hipSetDevice(n->dev[n->ingest]));
		HCHK(hipEventSynchronize(n->ev1));
		HCHK(hipEventElapsedTime(&t, n->ev0, n->ev1));
	}
	if (ms) *ms = t;
	if (bytes_out) *bytes_out = n->last_bytes;
	if (n_sends) *n_sends = n->last_sends;
	return 0;
}
