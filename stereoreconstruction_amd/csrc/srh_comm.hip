// srh_comm.hip -- the exchange step of the path over RCCL: per-view depth maps
// are gathered to one rank (TwoView pairs, final result) or to every rank (MVS cross-check reads
// every other view's map).  xGMI is a point-to-point mesh (7 links per GPU), so the gather is
// written as direct sends to the root inside one group instead of a ring collective.
//
// librccl is loaded on first use (dlopen): single-GPU users never need it, and a missing
// library is reported as SRH_E_UNSUPPORTED instead of failing the load of this library.
#include "srh_internal.hpp"
#include "srh_wait.hpp"

#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
// Build dependency: <rccl/rccl.h> of the ROCm the library is built against -- for ncclConfig_t / NCCL_CONFIG_INITIALIZER (a
// versioned struct: redeclaring it here would pin one RCCL's layout) and the result codes.  Nothing is LINKED: the library
// itself is dlopen'ed on first use, and a machine without it gets SRH_E_UNSUPPORTED from srh_comm_*.
#include <rccl/rccl.h>

namespace srh {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rccl_comm;
enum { RCCL_FLOAT64 = 8 };                                   // ncclFloat64

struct RcclApi {
	void *lib = nullptr;
	int (*GetUniqueId)(rccl_unique_id *) = nullptr;
	int (*CommInitRankConfig)(rccl_comm *, int, rccl_unique_id, int, ncclConfig_t *) = nullptr;   // optional (older RCCL: blocking ncclCommInitRank)
	int (*CommInitRank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
	int (*CommFinalize)(rccl_comm) = nullptr;                                                     // optional
	int (*CommGetAsyncError)(rccl_comm, int *) = nullptr;
	int (*CommAbort)(rccl_comm) = nullptr;
	int (*CommCount)(rccl_comm, int *) = nullptr;
	int (*GetVersion)(int *) = nullptr;
	int (*CommDestroy)(rccl_comm) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, rccl_comm, hipStream_t) = nullptr;
	int (*Send)(const void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
	int (*Recv)(void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
};

static RcclApi g_rccl;

const char *rccl_load() {
	if (g_rccl.lib) return nullptr;
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
	if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!h) return "librccl.so not found";
#define SRH_SYM(field, name) \
	*(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) { dlclose(h); return "librccl lacks " name; }
	SRH_SYM(GetUniqueId, "ncclGetUniqueId")
	SRH_SYM(CommInitRank, "ncclCommInitRank")
	*(void **)(&g_rccl.CommInitRankConfig) = dlsym(h, "ncclCommInitRankConfig");   // (absent in old RCCL builds: the blocking rendezvous then)
	*(void **)(&g_rccl.CommFinalize) = dlsym(h, "ncclCommFinalize");
	SRH_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
	SRH_SYM(CommAbort, "ncclCommAbort")
	SRH_SYM(CommCount, "ncclCommCount")
	SRH_SYM(GetVersion, "ncclGetVersion")
	SRH_SYM(CommDestroy, "ncclCommDestroy")
	SRH_SYM(AllGather, "ncclAllGather")
	SRH_SYM(Send, "ncclSend")
	SRH_SYM(Recv, "ncclRecv")
	SRH_SYM(GroupStart, "ncclGroupStart")
	SRH_SYM(GroupEnd, "ncclGroupEnd")
	SRH_SYM(GetErrorString, "ncclGetErrorString")
#undef SRH_SYM
	g_rccl.lib = h;
	return nullptr;
}

const char *rccl_unique_id_get(void *out128) {
	if (const char *e = rccl_load()) return e;
	rccl_unique_id id;
	const int rc = g_rccl.GetUniqueId(&id);
	if (rc) return g_rccl.GetErrorString(rc);
	memcpy(out128, &id, sizeof(id));
	return nullptr;
}

// The communicator is NON-BLOCKING (ncclConfig_t::blocking = 0): no call into RCCL can hold a rank for ever.  A call
// that answers ncclInProgress is polled (ncclCommGetAsyncError) until it settles or `timeout_ms` has passed; a rank that
// never arrives at the rendezvous, or a peer that dies inside a collective, then becomes an error string here -- the
// caller returns SRH_E_DEVICE and the process can exit non-zero -- after the communicator has been aborted.
static int g_timeout_ms = 120000;
void rccl_set_timeout_ms(int ms) { g_timeout_ms = ms > 0 ? ms : 120000; }

static const char *rccl_settle(rccl_comm c, int rc, const char *what) {
	static thread_local char msg[200];
	if (rc == ncclSuccess) return nullptr;
	if (rc != ncclInProgress) return g_rccl.GetErrorString(rc);
	int last = ncclSuccess;
	return bounded_wait([&]() -> int {
		int st = ncclSuccess;
		const int q = g_rccl.CommGetAsyncError(c, &st);
		last = q != ncclSuccess ? q : st;
		return last == ncclSuccess ? 0 : last == ncclInProgress ? 1 : 2;
	}, [&]() { return g_rccl.GetErrorString(last); }, g_timeout_ms, what, msg, sizeof(msg));
}

// The completion of whatever `st` holds up to now -- a collective included -- with the same bound: an event behind it is
// polled (hipEventQuery) together with the communicator's asynchronous error state.  A peer that died inside the
// collective leaves the event unreached: after the timeout the caller aborts the communicator (comm_failed, srh_api.hip)
// and returns SRH_E_DEVICE; it never sits in hipStreamSynchronize.
const char *rccl_wait_stream(void *comm, hipStream_t st, const char *what) {
	static thread_local char msg[200];
	hipEvent_t ev;
	if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
	if (hipEventRecord(ev, st) != hipSuccess) { (void)hipEventDestroy(ev); return "hipEventRecord failed"; }
	int last = ncclSuccess;
	hipError_t herr = hipSuccess;
	const char *e = bounded_wait([&]() -> int {
		const hipError_t q = hipEventQuery(ev);
		if (q == hipSuccess) return 0;
		if (q != hipErrorNotReady) { herr = q; return 2; }
		if (comm && g_rccl.CommGetAsyncError) {
			int a = ncclSuccess;
			const int r = g_rccl.CommGetAsyncError((rccl_comm)comm, &a);
			last = r != ncclSuccess ? r : a;
			if (last != ncclSuccess && last != ncclInProgress) return 2;
		}
		return 1;
	}, [&]() -> const char * { return herr != hipSuccess ? hipGetErrorString(herr) : g_rccl.GetErrorString(last); },
	g_timeout_ms, what, msg, sizeof(msg));
	(void)hipEventDestroy(ev);                                    // (an event that was never reached is released with the aborted work)
	return e;
}

const char *rccl_comm_init(void **comm, int nranks, int rank, const void *id128) {
	if (const char *e = rccl_load()) return e;
	rccl_unique_id id;
	memcpy(&id, id128, sizeof(id));
	rccl_comm c = nullptr;
	if (g_rccl.CommInitRankConfig) {
		ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
		cfg.blocking = 0;
		const int rc = g_rccl.CommInitRankConfig(&c, nranks, id, rank, &cfg);
		if (const char *e = rccl_settle(c, rc, "ncclCommInitRank")) {
			if (c) g_rccl.CommAbort(c);
			return e;
		}
	} else {
		// an RCCL without ncclCommInitRankConfig: the blocking rendezvous (what round 4 used) -- collectives are still waited
		// for with a bound (rccl_wait_stream), only a rank missing at THIS call can hold the others
		const int rc = g_rccl.CommInitRank(&c, nranks, id, rank);
		if (rc != ncclSuccess) return g_rccl.GetErrorString(rc);
	}
	int n = 0;
	if (g_rccl.CommCount(c, &n) != ncclSuccess || n != nranks) { g_rccl.CommAbort(c); return "communicator does not span the ranks asked for"; }
	*comm = c;
	return nullptr;
}

// NCCL_VERSION_CODE of the loaded library (e.g. 22703), 0 if it cannot be loaded
int rccl_version() {
	if (rccl_load()) return 0;
	int v = 0;
	return g_rccl.GetVersion(&v) == ncclSuccess ? v : 0;
}

void rccl_comm_abort(void *comm) { if (comm && g_rccl.CommAbort) g_rccl.CommAbort((rccl_comm)comm); }

// A non-blocking communicator is finalized first (ncclCommFinalize, settled with the same bound) and destroyed then; one
// that does not settle is aborted.  Returns an error text for the caller's log, the communicator is gone either way.
const char *rccl_comm_destroy(void *comm) {
	if (!comm || !g_rccl.CommDestroy) return nullptr;
	rccl_comm c = (rccl_comm)comm;
	if (g_rccl.CommFinalize) {
		if (const char *e = rccl_settle(c, g_rccl.CommFinalize(c), "ncclCommFinalize")) { g_rccl.CommAbort(c); return e; }
	}
	const int rc = g_rccl.CommDestroy(c);
	if (rc == ncclSuccess) return nullptr;
	if (rc == ncclInProgress) { if (const char *e = rccl_settle(c, rc, "ncclCommDestroy")) { g_rccl.CommAbort(c); return e; } return nullptr; }
	return g_rccl.GetErrorString(rc);
}

bool rccl_available() { return rccl_load() == nullptr; }

// every rank contributes `count` doubles; the root receives nranks*count (rank order)
const char *rccl_gather_f64(void *comm, int nranks, int rank, int root, const double *send, double *recv,
                            size_t count, hipStream_t st)
{
	int rc = g_rccl.GroupStart();
	if (rc && rc != ncclInProgress) return g_rccl.GetErrorString(rc);
	rc = 0;
	if (rank == root) {
		for (int r = 0; r < nranks && (!rc || rc == ncclInProgress); ++r) {
			if (r == rank) continue;
			rc = g_rccl.Recv(recv + (size_t)r*count, count, RCCL_FLOAT64, r, (rccl_comm)comm, st);
		}
	} else {
		rc = g_rccl.Send(send, count, RCCL_FLOAT64, root, (rccl_comm)comm, st);
	}
	const int rc2 = g_rccl.GroupEnd();
	if (rc && rc != ncclInProgress) return g_rccl.GetErrorString(rc);
	if (const char *e = rccl_settle((rccl_comm)comm, rc2, "the gather's ncclGroupEnd")) return e;
	if (rank == root) {
		if (hipMemcpyAsync(recv + (size_t)rank*count, send, count*sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
			return "hipMemcpyAsync of the root's own map failed";
	}
	return nullptr;
}

const char *rccl_allgather_f64(void *comm, const double *send, double *recv, size_t count, hipStream_t st) {
	const int rc = g_rccl.AllGather(send, recv, count, RCCL_FLOAT64, (rccl_comm)comm, st);
	return rccl_settle((rccl_comm)comm, rc, "ncclAllGather");
}

} // namespace srh
