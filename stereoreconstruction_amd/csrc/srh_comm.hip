// srh_comm.hip -- the exchange step of the path over RCCL: per-view depth maps
// are gathered to one rank (TwoView pairs, final result) or to every rank (MVS cross-check reads
// every other view's map).  xGMI is a point-to-point mesh (7 links per GPU), so the gather is
// written as direct sends to the root inside one group instead of a ring collective.
//
// librccl is loaded on first use (dlopen): single-GPU users never need it, and a missing
// library is reported as SRH_E_UNSUPPORTED instead of failing the load of this library.
#include "srh_internal.hpp"

#include <dlfcn.h>
#include <cstdio>
#include <cstring>

namespace srh {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rccl_comm;
enum { RCCL_FLOAT64 = 8 };                                   // ncclFloat64

struct RcclApi {
	void *lib = nullptr;
	int (*GetUniqueId)(rccl_unique_id *) = nullptr;
	int (*CommInitRank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
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

const char *rccl_comm_init(void **comm, int nranks, int rank, const void *id128) {
	if (const char *e = rccl_load()) return e;
	rccl_unique_id id;
	memcpy(&id, id128, sizeof(id));
	rccl_comm c = nullptr;
	const int rc = g_rccl.CommInitRank(&c, nranks, id, rank);
	if (rc) return g_rccl.GetErrorString(rc);
	*comm = c;
	return nullptr;
}

void rccl_comm_destroy(void *comm) { if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy((rccl_comm)comm); }

// every rank contributes `count` doubles; the root receives nranks*count (rank order)
const char *rccl_gather_f64(void *comm, int nranks, int rank, int root, const double *send, double *recv,
                            size_t count, hipStream_t st)
{
	int rc = g_rccl.GroupStart();
	if (rc) return g_rccl.GetErrorString(rc);
	if (rank == root) {
		for (int r = 0; r < nranks && !rc; ++r) {
			if (r == rank) continue;
			rc = g_rccl.Recv(recv + (size_t)r*count, count, RCCL_FLOAT64, r, (rccl_comm)comm, st);
		}
	} else {
		rc = g_rccl.Send(send, count, RCCL_FLOAT64, root, (rccl_comm)comm, st);
	}
	const int rc2 = g_rccl.GroupEnd();
	if (rc) return g_rccl.GetErrorString(rc);
	if (rc2) return g_rccl.GetErrorString(rc2);
	if (rank == root) {
		if (hipMemcpyAsync(recv + (size_t)rank*count, send, count*sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
			return "hipMemcpyAsync of the root's own map failed";
	}
	return nullptr;
}

const char *rccl_allgather_f64(void *comm, const double *send, double *recv, size_t count, hipStream_t st) {
	const int rc = g_rccl.AllGather(send, recv, count, RCCL_FLOAT64, (rccl_comm)comm, st);
	return rc ? g_rccl.GetErrorString(rc) : nullptr;
}

} // namespace srh
