// shard_rccl.cpp -- libsonde_rccl.so: the one collective of this path, native.
//
// Channels are independent, so a node shards them in contiguous ranges (GPU g of G gets [g*C/G, (g+1)*C/G)) and the
// decoders never talk to each other.  What moves is the INPUT, once: when one GPU ingests the IQ of all channels (an SDR
// front-end attached to one device), its blocks are scattered to the other GPUs over xGMI.  RCCL has no scatter
// primitive: it is a group of point-to-point transfers -- the root posts one ncclSend per peer, every rank one ncclRecv
// -- which lets the root drive all seven xGMI links at once (7 x ~153 GB/s; a ring would be bound by one link).  The
// return path (decoded frames, <= 0.2 % of the input bytes) is the mirrored gather.
//
// A separate library so that libsonde_mi355.so does not depend on librccl: single-GPU hosts never load this.
// The communicator is bootstrapped from a 128-byte id (ncclGetUniqueId on one rank, handed to the others by whatever
// the host has: torch.distributed in bench.py, MPI, a file).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string>
#include <string.h>
#include "../../include/sonde_shard.h"

static thread_local std::string g_serr;
static int sfail(const char *what, const char *detail = nullptr)
{
	g_serr = what;
	if (detail) { g_serr += ": "; g_serr += detail; }
	return -1;
}
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return sfail(#x, ncclGetErrorString(r_)); } while (0)
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return sfail(#x, hipGetErrorString(e_)); } while (0)

struct SondeShard {
	ncclComm_t comm = nullptr;
	int world = 0, rank = 0, device = 0;
};

extern "C" const char *sonde_shard_last_error(void) { return g_serr.c_str(); }

extern "C" int sonde_shard_unique_id(void *id128)
{
	if (!id128) return sfail("sonde_shard_unique_id: null argument");
	static_assert(sizeof(ncclUniqueId) == SONDE_SHARD_ID_BYTES, "id size");
	NCHK(ncclGetUniqueId((ncclUniqueId *)id128));
	return 0;
}

extern "C" int sonde_shard_create(const void *id128, int world, int rank, int device, SondeShard **out)
{
	if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return sfail("sonde_shard_create: bad argument");
	HCHK(hipSetDevice(device));
	SondeShard *s = new SondeShard;
	s->world = world; s->rank = rank; s->device = device;
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	ncclResult_t r = ncclCommInitRank(&s->comm, world, id, rank);
	if (r != ncclSuccess) { delete s; return sfail("ncclCommInitRank", ncclGetErrorString(r)); }
	*out = s;
	return 0;
}

extern "C" void sonde_shard_destroy(SondeShard *s)
{
	if (!s) return;
	(void)hipSetDevice(s->device);
	if (s->comm) (void)ncclCommDestroy(s->comm);
	delete s;
}

extern "C" int sonde_shard_rank(const SondeShard *s) { return s ? s->rank : -1; }
extern "C" int sonde_shard_world(const SondeShard *s) { return s ? s->world : -1; }

// The channel range of a rank: contiguous, the remainder spread over the first ranks.
extern "C" void sonde_shard_range(uint32_t n_channels, int world, int rank, uint32_t *first, uint32_t *count)
{
	const uint32_t base = n_channels / (uint32_t)world, rem = n_channels % (uint32_t)world;
	const uint32_t r = (uint32_t)rank;
	if (first) *first = r * base + (r < rem ? r : rem);
	if (count) *count = base + (r < rem ? 1u : 0u);
}

// Scatter: rank `root` holds world consecutive blocks of `bytes` each; rank r receives block r into shard_dev.
extern "C" int sonde_shard_scatter(SondeShard *s, const void *full_dev, void *shard_dev, size_t bytes, int root, void *stream_)
{
	if (!s || !shard_dev || root < 0 || root >= s->world || (s->rank == root && !full_dev)) return sfail("sonde_shard_scatter: bad argument");
	hipStream_t stream = (hipStream_t)stream_;
	HCHK(hipSetDevice(s->device));
	NCHK(ncclGroupStart());
	if (s->rank == root)
		for (int p = 0; p < s->world; p++) {
			ncclResult_t r = ncclSend((const char *)full_dev + (size_t)p * bytes, bytes, ncclChar, p, s->comm, stream);
			if (r != ncclSuccess) { (void)ncclGroupEnd(); return sfail("ncclSend", ncclGetErrorString(r)); }
		}
	ncclResult_t r = ncclRecv(shard_dev, bytes, ncclChar, root, s->comm, stream);
	if (r != ncclSuccess) { (void)ncclGroupEnd(); return sfail("ncclRecv", ncclGetErrorString(r)); }
	NCHK(ncclGroupEnd());
	return 0;
}

// Gather (frames back to the ingest rank): every rank sends `bytes` from part_dev; root receives block r from rank r.
extern "C" int sonde_shard_gather(SondeShard *s, const void *part_dev, size_t bytes, void *all_dev, int root, void *stream_)
{
	if (!s || !part_dev || root < 0 || root >= s->world || (s->rank == root && !all_dev)) return sfail("sonde_shard_gather: bad argument");
	hipStream_t stream = (hipStream_t)stream_;
	HCHK(hipSetDevice(s->device));
	NCHK(ncclGroupStart());
	if (s->rank == root)
		for (int p = 0; p < s->world; p++) {
			ncclResult_t r = ncclRecv((char *)all_dev + (size_t)p * bytes, bytes, ncclChar, p, s->comm, stream);
			if (r != ncclSuccess) { (void)ncclGroupEnd(); return sfail("ncclRecv", ncclGetErrorString(r)); }
		}
	ncclResult_t r = ncclSend(part_dev, bytes, ncclChar, root, s->comm, stream);
	if (r != ncclSuccess) { (void)ncclGroupEnd(); return sfail("ncclSend", ncclGetErrorString(r)); }
	NCHK(ncclGroupEnd());
	return 0;
}

// Scatter of ROWS into a strided destination (round 4): n_rows_total rows of row_bytes, root holds row k at
// full_dev + k * src_stride_bytes; rank r owns the rows of sonde_shard_range(n_rows_total, world, r) -- unequal shards when the
// count does not divide -- and receives them at shard_dev + row * dst_stride_bytes: straight into the rows its decoder reads
// (sonde_row_stride), no re-stride copy behind the collective.  One ncclSend / ncclRecv pair per row, 256 rows of every peer
// per group; with equal strides a peer's shard is one contiguous run (padding included): one pair per peer.  The root's own
// shard is a strided device copy, not a send to itself.
extern "C" int sonde_shard_scatter_rows(SondeShard *s, const void *full_dev, size_t src_stride_bytes, void *shard_dev, size_t dst_stride_bytes,
                                        size_t row_bytes, uint32_t n_rows_total, int root, void *stream_)
{
	if (!s || !shard_dev || root < 0 || root >= s->world || (s->rank == root && !full_dev) || row_bytes == 0 ||
	    src_stride_bytes < row_bytes || dst_stride_bytes < row_bytes || n_rows_total < (uint32_t)s->world)
		return sfail("sonde_shard_scatter_rows: bad argument");
	hipStream_t stream = (hipStream_t)stream_;
	HCHK(hipSetDevice(s->device));
	uint32_t my_first = 0, my_count = 0;
	sonde_shard_range(n_rows_total, s->world, s->rank, &my_first, &my_count);
	const bool same = src_stride_bytes == dst_stride_bytes;
	const uint32_t max_rows = n_rows_total / (uint32_t)s->world + 1;
	const uint32_t per_group = same ? max_rows : 256u;
	for (uint32_t r0 = 0; r0 < max_rows; r0 += per_group) {
		NCHK(ncclGroupStart());
		ncclResult_t r = ncclSuccess;
		if (s->rank == root) {
			for (int p = 0; p < s->world && r == ncclSuccess; p++) {
				if (p == root) continue;
				uint32_t f = 0, c = 0;
				sonde_shard_range(n_rows_total, s->world, p, &f, &c);
				if (same) {
					if (r0 == 0) r = ncclSend((const char *)full_dev + (size_t)f * src_stride_bytes, (size_t)(c - 1) * src_stride_bytes + row_bytes, ncclChar, p, s->comm, stream);
				} else {
					for (uint32_t row = r0; row < c && row < r0 + per_group && r == ncclSuccess; row++)
						r = ncclSend((const char *)full_dev + ((size_t)f + row) * src_stride_bytes, row_bytes, ncclChar, p, s->comm, stream);
				}
			}
		} else if (same) {
			if (r0 == 0) r = ncclRecv(shard_dev, (size_t)(my_count - 1) * dst_stride_bytes + row_bytes, ncclChar, root, s->comm, stream);
		} else {
			for (uint32_t row = r0; row < my_count && row < r0 + per_group && r == ncclSuccess; row++)
				r = ncclRecv((char *)shard_dev + (size_t)row * dst_stride_bytes, row_bytes, ncclChar, root, s->comm, stream);
		}
		if (r != ncclSuccess) { (void)ncclGroupEnd(); return sfail("ncclSend/ncclRecv", ncclGetErrorString(r)); }
		NCHK(ncclGroupEnd());
	}
	if (s->rank == root)
		HCHK(hipMemcpy2DAsync(shard_dev, dst_stride_bytes, (const char *)full_dev + (size_t)my_first * src_stride_bytes, src_stride_bytes,
		                      row_bytes, my_count, hipMemcpyDeviceToDevice, stream));
	return 0;
}
