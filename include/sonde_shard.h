/* sonde_shard.h -- C ABI of libsonde_rccl.so: sharding the channels of one node over its GPUs.
 *
 * The reference runs one channel per module instance with no shared state (SDRPP_MOD_INFO max instances -1,
 * /root/reference/src/main.cpp:18-24): channels shard with no data-path collective.  The only exchange is the input
 * scatter when a single GPU ingests all channels, and the mirrored gather of decoded frames; both are groups of
 * ncclSend / ncclRecv over xGMI (RCCL has no scatter primitive).  libsonde_mi355.so does not depend on this library.
 *
 * Bootstrap: one rank calls sonde_shard_unique_id(), hands the 128 bytes to the others (MPI, torch.distributed, a file),
 * every rank calls sonde_shard_create(id, world, rank, device).  All calls return 0 on success, -1 on error with the text
 * in sonde_shard_last_error().  Buffers are device pointers; transfers are ordered on `stream`. */
#ifndef SONDE_SHARD_H
#define SONDE_SHARD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SONDE_SHARD_ID_BYTES 128
typedef struct SondeShard SondeShard;

int         sonde_shard_unique_id(void *id128);
int         sonde_shard_create(const void *id128, int world, int rank, int device, SondeShard **out);
void        sonde_shard_destroy(SondeShard *s);
int         sonde_shard_rank(const SondeShard *s);
int         sonde_shard_world(const SondeShard *s);
/* contiguous channel range of `rank` out of `world` (pure arithmetic, no communicator needed) */
void        sonde_shard_range(uint32_t n_channels, int world, int rank, uint32_t *first, uint32_t *count);
/* root holds `world` consecutive blocks of `bytes`; rank r receives block r */
int         sonde_shard_scatter(SondeShard *s, const void *full_dev /* root only */, void *shard_dev, size_t bytes, int root, void *stream);
/* Rows into a strided destination: n_rows_total rows of row_bytes on the root (row k at full_dev + k * src_stride_bytes); rank r
 * receives the rows of sonde_shard_range(n_rows_total, world, r) (unequal shards allowed) at shard_dev + row * dst_stride_bytes,
 * i.e. straight into rows on the decoder's recommended channel stride (sonde_row_stride): no re-stride copy.  The root's own
 * shard is a device copy, not a send to itself.  Equal strides make a peer's shard one contiguous run: ONE send / receive pair per
 * peer; else one pair per row (256 rows of every peer per group).  EVERY RANK MUST PASS THE SAME src_stride_bytes AND
 * dst_stride_bytes (the root's source layout is an argument of the collective, not a local property: a rank that guesses another
 * value posts receives of the other shape and the transfer hangs). */
int         sonde_shard_scatter_rows(SondeShard *s, const void *full_dev /* root only */, size_t src_stride_bytes, void *shard_dev,
                                     size_t dst_stride_bytes, size_t row_bytes, uint32_t n_rows_total, int root, void *stream);
/* every rank sends `bytes`; root receives block r from rank r */
int         sonde_shard_gather(SondeShard *s, const void *part_dev, size_t bytes, void *all_dev /* root only */, int root, void *stream);
const char *sonde_shard_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
