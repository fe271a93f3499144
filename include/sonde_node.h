/* sonde_node.h -- C ABI of the node-level host in libsonde_rccl.so: ONE process, the GPUs of one node, one decoder batch per GPU.
 *
 * north_star: "C++ host code ... independent narrowband channels are batched one-per-workgroup and sharded across the 8 GPUs of
 * one node with an RCCL scatter of IQ blocks over xGMI".  The reference's unit is one module instance per channel, any number
 * of instances, no shared state (SDRPP_MOD_INFO max instances -1, /root/reference/src/main.cpp:18-24): a node-level host is an
 * object that owns ALL channels of the node, shards them in contiguous ranges (sonde_node_shard_range: device d of D gets
 * [d C / D, (d + 1) C / D), remainders to the first devices), and hands back SondeData fragments with node-wide channel numbers
 * (the callback of /root/reference/src/main.cpp:320-331, once per channel).
 *
 * Data path of one sonde_node_submit(): the IQ of all channels sits on the INGEST device (an SDR front-end attached to one
 * GPU); every other device receives its shard over xGMI straight into the rows its decoder reads, as groups of ncclSend (ingest
 * device) / ncclRecv (peers); the ingest device's own shard is a strided device copy, not a send to itself; then every device runs
 * its own sonde_batch_submit.  There is no other collective: channels are independent.  Frames come back per device straight to
 * host memory (one process: the host is shared; nothing travels back over xGMI).
 *
 * ONLY THE ROWS' BYTES CROSS xGMI (round 6), whatever the layout of the ingest rows:
 *   back to back (channel_stride == n_samples): ONE send per peer of exactly the shard's bytes, straight from the ingest buffer.
 *   RECOMMENDED FOR INGEST: it needs no staging and no packing pass.
 *   any other stride (sonde_row_stride's included): the rows are packed on the ingest device, four chunks per submit, chunk c + 1
 *   copied into a staging buffer while chunk c is on the links; one send per peer and chunk of exactly the rows' bytes.
 *   (Round 5 sent a strided shard as one run, padding included: 1.33 x the bytes over the slowest link of the system.)
 * In both, a peer's rows land back to back and are decoded from there (the decoder takes any stride; back-to-back rows cost it
 * 3-5 %).  scatter_mode 1: one send per row into rows on the recommended stride (no staging memory; 57 344 sends per block of
 * BASELINE config 5: a fallback, not a recommendation).
 * The rows exist twice per device: submit t + 1 scatters while the decoders of submit t still read.
 *
 * All int / long calls: >= 0 ok, negative = error (text: sonde_node_last_error()).  Not thread-safe per object. */
#ifndef SONDE_NODE_H
#define SONDE_NODE_H
#include <stddef.h>
#include <stdint.h>
#include "sonde_abi.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct SondeNode SondeNode;

typedef struct {
	uint32_t       n_devices;     /* 1 .. 16 */
	const int32_t *devices;       /* HIP ordinals, n_devices entries; NULL = 0 .. n_devices - 1 */
	uint32_t       ingest;        /* index into devices[] of the device that receives the IQ of all channels */
	uint32_t       n_channels;    /* all channels of the node */
	const uint8_t *types;         /* n_channels entries of SONDE_*; NULL = all SONDE_RS41 */
	uint32_t       max_samples;   /* largest samples-per-channel of one submit (multiple of SONDE_TILE) */
	int32_t        input_kind;    /* SONDE_INPUT_IQ, SONDE_INPUT_REAL, SONDE_INPUT_IQ16 or SONDE_INPUT_IQ8 (a half / a quarter of the bytes over xGMI) */
	uint32_t       flags;         /* SONDE_FLAG_* of the per-device batches (SONDE_FLAG_PIPELINE is ignored: the node's row sets need the
	                                 per-device streams joined with their decoders; SONDE_FLAG_LATE_JOIN is honoured) */
	uint32_t       scatter_mode;  /* 0: by the ingest layout (above); 1: always one send per row into rows on the recommended stride.
	                                 | SONDE_NODE_TEST_SHARED_DEVICES: test hook, see below */
} SondeNodeConfig;

/* TEST HOOK (tests/test_node_fake.py): a HIP device may be listed more than once in `devices`, so that a node of N > 1 shards can run on
 * a box with one GPU against tests/cpp/fake_rccl.cpp.  The real RCCL refuses such a communicator: never set it in production. */
#define SONDE_NODE_TEST_SHARED_DEVICES 0x100u

int    sonde_node_create(const SondeNodeConfig *cfg, SondeNode **out);
/* the contiguous channel range of device index d of nd (pure arithmetic): [first, first + count), remainders to the first devices */
void   sonde_node_shard_range(uint32_t n_channels, uint32_t nd, uint32_t d, uint32_t *first, uint32_t *count);
void   sonde_node_destroy(SondeNode *n);
uint32_t sonde_node_devices(const SondeNode *n);
/* channel range of device index d */
int    sonde_node_range(const SondeNode *n, uint32_t d, uint32_t *first, uint32_t *count);
/* the batch of device index d (sonde_batch_* introspection: kernel times, bits, state); owned by the node */
SondeBatch *sonde_node_batch(SondeNode *n, uint32_t d);

/* samples: DEVICE pointer on the ingest device, channel-major, channel c at element c * channel_stride (as
 * sonde_batch_submit).  Scatters, then submits on every device.  Asynchronous; the ingest buffer may be reused once
 * sonde_node_scatter_done() / sonde_node_sync() has returned.
 * The scatter runs on the node's own streams.  sonde_node_submit_on: it starts behind the work queued so far on `stream`
 * (hipStream_t on the ingest device; NULL = the legacy default stream) -- the stream that produced the ingest buffer;
 * sonde_node_submit = sonde_node_submit_on(..., NULL). */
int    sonde_node_submit(SondeNode *n, const void *samples, size_t n_samples, size_t channel_stride);
int    sonde_node_submit_on(SondeNode *n, const void *samples, size_t n_samples, size_t channel_stride, void *stream);
/* Every device's shard is already resident on that device (rank-local ingest: no scatter): rows[d] = device pointer on device d,
 * channel-major, channel_stride elements apart.  The decoders run on the node's own streams: the rows must be COMPLETE when this
 * is called (synchronise the streams that wrote them), and stay untouched until the submit after the next one (or sonde_node_sync). */
int    sonde_node_submit_local(SondeNode *n, const void *const *rows, size_t n_samples, size_t channel_stride);
int    sonde_node_scatter_done(SondeNode *n);
/* wait for every device; returns the frames of the last submit, all devices (or negative) */
long   sonde_node_sync(SondeNode *n);
/* the last submit's frames in (node-wide channel, time) order, channel numbers node-wide */
long   sonde_node_frames(SondeNode *n, SondeFrame *out, size_t cap);
/* the last submit's telemetry as SondeData fragments with node-wide channel numbers (sonde_batch_poll of every device, merged in
 * device order); call until it returns 0 */
long   sonde_node_poll(SondeNode *n, SondeData *out, uint32_t *channel, size_t cap);
/* device time of the last submit's scatter (ms, on the ingest device's stream; 0 for one device / submit_local), the bytes that
 * left the ingest device, and the number of ncclSend calls they took */
int    sonde_node_scatter_stats(SondeNode *n, float *ms, uint64_t *bytes_out, uint32_t *n_sends);
/* host time (ms) the last sonde_node_frames spent copying frame records device -> host (every device done before the clock starts),
 * and their bytes: the return path of a step (nothing travels back over xGMI: one process, one host) */
int    sonde_node_gather_stats(SondeNode *n, double *ms, uint64_t *bytes);
const char *sonde_node_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
