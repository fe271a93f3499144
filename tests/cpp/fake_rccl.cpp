// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the handful of RCCL entry points csrc/node.cpp calls, so that the node host's
// N > 1 logic (shard ranges, transfer shapes, staging, stream ordering, statistics) can EXECUTE on a box with one GPU
// ("ncclCommInitAll + grouped ncclSend / ncclRecv have never run with nd > 1").
// Linked INSTEAD of librccl into tests/cpp/_build/libsonde_rccl_fake.so (tests/test_node_fake.py builds it); the product library
// libsonde_rccl.so links the real RCCL and never sees this file.
//
// Semantics kept from RCCL: communicators of one process (ncclCommInitAll), point-to-point operations collected between
// ncclGroupStart / ncclGroupEnd, a send on rank r to peer p pairs with the receive on rank p from peer r in posting order, byte counts
// must agree, an operation is ordered on the stream it was posted with (the data moves once both sides' streams have reached the
// group; both streams then wait for the copy).  Not kept: any notion of links -- the "transfer" is a device copy.
// Several communicators MAY sit on the same HIP device (the test hook of sonde_node_create lists device 0 more than once).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>
#include <deque>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

struct FakeComm { int rank, nranks, dev; uint64_t world_id; };
struct FakeOp { bool send; const void *src; void *dst; size_t bytes; int peer; FakeComm *comm; hipStream_t st; };
static std::mutex g_mu;
static int g_depth = 0;
static std::vector<FakeOp> g_ops;
static uint64_t g_next_world = 1;
static uint64_t g_sends = 0, g_bytes = 0, g_groups = 0;

extern "C" {
// test introspection (not RCCL): totals since the last call
void fake_rccl_stats(uint64_t *sends, uint64_t *bytes, uint64_t *groups)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (sends) *sends = g_sends;
	if (bytes) *bytes = g_bytes;
	if (groups) *groups = g_groups;
	g_sends = g_bytes = g_groups = 0;
}

const char *ncclGetErrorString(ncclResult_t r)
{
	switch (r) {
	case ncclSuccess: return "no error";
	case ncclInvalidArgument: return "invalid argument (fake RCCL)";
	case ncclInvalidUsage: return "invalid usage (fake RCCL: unmatched or mismatched send / receive)";
	case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
	default: return "error (fake RCCL)";
	}
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
	if (!comms || ndev < 1) return ncclInvalidArgument;
	std::lock_guard<std::mutex> lk(g_mu);
	const uint64_t w = g_next_world++;
	for (int i = 0; i < ndev; i++) {
		FakeComm *c = new FakeComm{ i, ndev, devlist ? devlist[i] : i, w };
		comms[i] = reinterpret_cast<ncclComm_t>(c);
	}
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	delete reinterpret_cast<FakeComm *>(comm);
	return ncclSuccess;
}

static ncclResult_t flush_locked()
{
	// pair every send with the matching receive, in posting order per (world, source rank, destination rank)
	typedef std::pair<uint64_t, std::pair<int, int>> Key;
	std::map<Key, std::deque<size_t>> sends, recvs;
	for (size_t i = 0; i < g_ops.size(); i++) {
		const FakeOp &o = g_ops[i];
		if (o.peer < 0 || o.peer >= o.comm->nranks) { g_ops.clear(); return ncclInvalidArgument; }
		if (o.send) sends[Key(o.comm->world_id, std::make_pair(o.comm->rank, o.peer))].push_back(i);
		else recvs[Key(o.comm->world_id, std::make_pair(o.peer, o.comm->rank))].push_back(i);
	}
	ncclResult_t res = ncclSuccess;
	int dev0 = 0;
	(void)hipGetDevice(&dev0);
	for (auto &kv : sends) {
		std::deque<size_t> &sq = kv.second, &rq = recvs[kv.first];
		if (sq.size() != rq.size()) { res = ncclInvalidUsage; break; }
		for (size_t k = 0; k < sq.size() && res == ncclSuccess; k++) {
			const FakeOp &s = g_ops[sq[k]], &r = g_ops[rq[k]];
			if (s.bytes != r.bytes) { res = ncclInvalidUsage; break; }
			hipEvent_t es = nullptr, er = nullptr;
			bool ok = hipSetDevice(s.comm->dev) == hipSuccess && hipEventCreateWithFlags(&es, hipEventDisableTiming) == hipSuccess &&
			          hipEventRecord(es, s.st) == hipSuccess;                     // the sender's stream has produced the data ...
			ok = ok && hipSetDevice(r.comm->dev) == hipSuccess && hipStreamWaitEvent(r.st, es, 0) == hipSuccess &&
			     hipMemcpyAsync(r.dst, s.src, s.bytes, hipMemcpyDeviceToDevice, r.st) == hipSuccess &&      // ... the receiver's stream moves it ...
			     hipEventCreateWithFlags(&er, hipEventDisableTiming) == hipSuccess && hipEventRecord(er, r.st) == hipSuccess;
			ok = ok && hipSetDevice(s.comm->dev) == hipSuccess && hipStreamWaitEvent(s.st, er, 0) == hipSuccess;      // ... and the send buffer is free behind it
			if (es) (void)hipEventDestroy(es);
			if (er) (void)hipEventDestroy(er);
			if (!ok) res = ncclUnhandledCudaError;
			g_sends++; g_bytes += s.bytes;
		}
		rq.clear();
		if (res != ncclSuccess) break;
	}
	if (res == ncclSuccess)
		for (auto &kv : recvs) if (!kv.second.empty()) { res = ncclInvalidUsage; break; }      // a receive nobody sends to
	(void)hipSetDevice(dev0);
	g_ops.clear();
	g_groups++;
	return res;
}

ncclResult_t ncclGroupStart(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_depth++;
	return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (g_depth <= 0) return ncclInvalidUsage;
	if (--g_depth > 0) return ncclSuccess;
	return flush_locked();
}

static ncclResult_t post(bool send, const void *src, void *dst, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
	if (!comm || (!src && !dst) || dt != ncclChar) return ncclInvalidArgument;       // (node.cpp moves bytes)
	std::lock_guard<std::mutex> lk(g_mu);
	g_ops.push_back(FakeOp{ send, src, dst, count, peer, reinterpret_cast<FakeComm *>(comm), st });
	if (g_depth == 0) return flush_locked();          // outside a group an operation cannot find its partner in ONE process
	return ncclSuccess;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
	return post(true, sendbuff, nullptr, count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
	return post(false, nullptr, recvbuff, count, datatype, peer, comm, stream);
}
}
