// sharded.hpp -- MultiViewStereo::runTask (multiviewstereo.cpp:325-475) over several srh_contexts: the C++ side of
// SURVEY.md 8(e).  Views are independent until the cross-check, so they are dealt out in contiguous shards
// (shardUnits); each shard runs computeInitialEstimate for its views on its own context (its own GPU); ONE
// all-gather moves every shard's depth maps to every shard (the cross-check of a view reads all other views' maps,
// multiviewstereo.cpp:694-719); then the sequential, order-dependent cross-check chain (:427-431) runs on every
// shard, so the final maps are resident everywhere, shard 0 included.  Same algorithm as
// stereoreconstruction_amd/distributed.py::multiview_sharded (which the multi-GPU bench drives through
// torch.distributed / RCCL).
//
// A shard is (engine, transport rank).  Engines: HipViewEngine (an srh_context) below; anything with the same five
// members for tests.  Transports: LoopbackTransport (shards are threads of one process: N contexts on N GPUs, or a
// rehearsal of N ranks on one), RcclTransport (one process per GPU, srh_comm_allgather_depth over xGMI).
#pragma once

#include <condition_variable>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "stereo_recon_hip.h"

namespace sharded {

// contiguous, balanced split of units 0..n-1; the first (n % world) ranks take one extra unit
inline void shardUnits(int n, int world, int rank, int &lo, int &hi) {
	const int base = n / world, extra = n % world;
	lo = rank*base + (rank < extra ? rank : extra);
	hi = lo + base + (rank < extra ? 1 : 0);
}

// how depth maps travel between shards: every rank contributes `count` doubles and receives ranks()*count
struct Transport {
	virtual ~Transport() { }
	virtual int ranks() const = 0;
	virtual bool allGather(int rank, const double *send, size_t count, double *recv) = 0;
};

// shards = threads of this process; allGather is a rendezvous on shared host memory
class LoopbackTransport : public Transport {
public:
	explicit LoopbackTransport(int n) : n_(n), arrived_(0), generation_(0) { }
	int ranks() const { return n_; }
	bool allGather(int rank, const double *send, size_t count, double *recv) {
		std::unique_lock<std::mutex> lk(m_);
		if (arrived_ == 0) buf_.assign(static_cast<size_t>(n_)*count, 0.0);
		if (buf_.size() != static_cast<size_t>(n_)*count) return false;          // ranks disagree on the size
		std::memcpy(&buf_[static_cast<size_t>(rank)*count], send, count*sizeof(double));
		const unsigned gen = generation_;
		if (++arrived_ == n_) { out_ = buf_; arrived_ = 0; ++generation_; cv_.notify_all(); }
		else cv_.wait(lk, [&] { return generation_ != gen; });
		std::memcpy(recv, out_.data(), out_.size()*sizeof(double));
		return true;
	}
private:
	int n_, arrived_;
	unsigned generation_;
	std::mutex m_;
	std::condition_variable cv_;
	std::vector<double> buf_, out_;
};

// MultiViewStereo::runTask for the shard `rank` of `t.ranks()`.  Engine: viewSize(v) -> pixels, initialEstimate(v),
// getDepth(v, double*), setDepth(v, const double*), crossCheck(v); all return false on error.  Views may differ
// in size: maps travel padded with NaN to the largest view.  Returns false on an engine / transport error.
template <class Engine>
bool runMultiView(Engine &e, int nviews, Transport *t, int rank, std::vector<int> *mine = nullptr) {
	const int world = t ? t->ranks() : 1;
	int lo, hi;
	shardUnits(nviews, world, rank, lo, hi);
	if (mine) { mine->clear(); for (int v = lo; v < hi; ++v) mine->push_back(v); }
	for (int v = lo; v < hi; ++v) if (!e.initialEstimate(v)) return false;        // multiviewstereo.cpp:365-376
	if (world > 1) {
		size_t npix = 0;
		for (int v = 0; v < nviews; ++v) if (e.viewSize(v) > npix) npix = e.viewSize(v);
		const int per = (nviews + world - 1)/world;                               // equal contributions: short ranks pad
		std::vector<double> send(static_cast<size_t>(per)*npix, std::numeric_limits<double>::quiet_NaN());
		std::vector<double> recv(static_cast<size_t>(world)*per*npix);
		for (int v = lo; v < hi; ++v) if (!e.getDepth(v, &send[static_cast<size_t>(v - lo)*npix])) return false;
		if (!t->allGather(rank, send.data(), send.size(), recv.data())) return false;
		for (int r = 0; r < world; ++r) {
			if (r == rank) continue;
			int rlo, rhi;
			shardUnits(nviews, world, r, rlo, rhi);
			for (int v = rlo; v < rhi; ++v)
				if (!e.setDepth(v, &recv[(static_cast<size_t>(r)*per + (v - rlo))*npix])) return false;
		}
	}
	for (int v = 0; v < nviews; ++v) if (!e.crossCheck(v)) return false;          // in view order (:427-431)
	return true;
}

// one srh_context holding ALL views (images and cameras are a few MB: every shard can match its views against any
// neighbour); slot v = view v
class HipViewEngine {
public:
	HipViewEngine(srh_context *ctx, const std::vector<srh_camera> &cams, const srh_params &p)
		: ctx_(ctx), cams_(cams), p_(p), nn_(p.num_neighbours > 0 ? p.num_neighbours : 1)
	{
		neigh_.assign(cams.size()*nn_, -1); count_.assign(cams.size(), 0);
		ok_ = srh_mvs_neighbours(static_cast<int>(cams.size()), cams_.data(), &p_, neigh_.data(), count_.data()) == SRH_OK;
		slots_.resize(cams.size());
		for (size_t v = 0; v < cams.size(); ++v) slots_[v] = static_cast<int32_t>(v);
	}
	bool ok() const { return ok_; }
	size_t viewSize(int v) const { int w = 0, h = 0; srh_view_size(ctx_, v, &w, &h); return static_cast<size_t>(w)*h; }
	bool initialEstimate(int v) { return srh_mvs_initial_estimate(ctx_, v, &neigh_[static_cast<size_t>(v)*nn_], count_[v], &p_, 0, 0, nullptr) == SRH_OK; }
	bool getDepth(int v, double *out) { return srh_view_depth_download(ctx_, v, out) == SRH_OK; }
	bool setDepth(int v, const double *in) { return srh_view_depth_upload(ctx_, v, in) == SRH_OK; }
	bool crossCheck(int v) { return srh_mvs_cross_check(ctx_, slots_.data(), static_cast<int>(slots_.size()), v, &p_) == SRH_OK; }
private:
	srh_context *ctx_;
	std::vector<srh_camera> cams_;
	srh_params p_;
	int nn_;
	std::vector<int32_t> neigh_, count_, slots_;
	bool ok_;
};

// one process per GPU: the context's RCCL communicator (srh_comm_init) carries the all-gather over xGMI
class RcclTransport : public Transport {
public:
	RcclTransport(srh_context *ctx, int nranks) : ctx_(ctx), n_(nranks) { }
	int ranks() const { return n_; }
	bool allGather(int, const double *send, size_t count, double *recv) {
		return srh_comm_allgather_host(ctx_, send, count, recv) == SRH_OK;
	}
private:
	srh_context *ctx_; int n_;
};

} // namespace sharded
