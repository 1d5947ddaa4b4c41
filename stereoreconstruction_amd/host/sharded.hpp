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
	// a shard that cannot go on says so: waiting shards wake up and their allGather returns false (no hang)
	virtual void abort() { }
	// device-resident exchange of the views' depth maps (the engine's context holds them): false = not offered,
	// runMultiView then moves the maps through host memory with allGather
	virtual bool deviceResident() const { return false; }
	virtual bool allGatherViews(int /*nviews*/) { return false; }
};

// shards = threads of this process; allGather is a rendezvous on shared host memory
class LoopbackTransport : public Transport {
public:
	explicit LoopbackTransport(int n) : n_(n), arrived_(0), generation_(0) { }
	int ranks() const { return n_; }
	bool allGather(int rank, const double *send, size_t count, double *recv) {
		std::unique_lock<std::mutex> lk(m_);
		if (aborted_) return false;
		if (arrived_ == 0) buf_.assign(static_cast<size_t>(n_)*count, 0.0);
		if (buf_.size() != static_cast<size_t>(n_)*count) {                      // ranks disagree on the size
			aborted_ = true; cv_.notify_all();
			return false;
		}
		std::memcpy(&buf_[static_cast<size_t>(rank)*count], send, count*sizeof(double));
		const unsigned gen = generation_;
		if (++arrived_ == n_) { out_ = buf_; arrived_ = 0; ++generation_; cv_.notify_all(); }
		else cv_.wait(lk, [&] { return generation_ != gen || aborted_; });
		if (aborted_) return false;
		std::memcpy(recv, out_.data(), out_.size()*sizeof(double));
		return true;
	}
	void abort() { std::lock_guard<std::mutex> lk(m_); aborted_ = true; cv_.notify_all(); }
private:
	int n_, arrived_;
	bool aborted_ = false;
	unsigned generation_;
	std::mutex m_;
	std::condition_variable cv_;
	std::vector<double> buf_, out_;
};

// Engines whose initialEstimate only QUEUES a view (HipViewEngine: two estimates in flight) offer finish(): wait for the
// queued estimates and say whether all of them succeeded.  Engines without the member are complete when the call returns.
template <class Engine>
auto finishEstimates(Engine &e, int) -> decltype(e.finish()) { return e.finish(); }
template <class Engine>
bool finishEstimates(Engine &, long) { return true; }

// MultiViewStereo::runTask for the shard `rank` of `t.ranks()`.  Engine: viewSize(v) -> pixels, initialEstimate(v),
// getDepth(v, double*), setDepth(v, const double*), crossCheck(v); all return false on error.  Views may differ
// in size: maps travel padded with NaN to the largest view.  Returns false on an engine / transport error.
template <class Engine>
bool runMultiView(Engine &e, int nviews, Transport *t, int rank, std::vector<int> *mine = nullptr) {
	const int world = t ? t->ranks() : 1;
	int lo, hi;
	shardUnits(nviews, world, rank, lo, hi);
	if (mine) { mine->clear(); for (int v = lo; v < hi; ++v) mine->push_back(v); }
	bool ok = true;
	for (int v = lo; v < hi && ok; ++v) ok = e.initialEstimate(v);               // multiviewstereo.cpp:365-376
	// a queued estimate reports a device error, or the failure of its cut-list redo, only when it is waited for: the status
	// below must cover that -- a shard that learnt of it inside the map exchange would leave the others in the collective
	ok = finishEstimates(e, 0) && ok;
	if (world > 1) {
		// every shard says whether its estimates succeeded BEFORE the maps travel: a shard that failed must not leave
		// the others waiting in the collective for ever -- all of them return false together
		double st = ok ? 0.0 : 1.0;
		std::vector<double> all(static_cast<size_t>(world), 1.0);
		if (!t->allGather(rank, &st, 1, all.data())) { t->abort(); return false; }
		for (int r = 0; r < world; ++r) if (all[r] != 0.0) return false;
		if (t->deviceResident()) {
			if (!t->allGatherViews(nviews)) { t->abort(); return false; }
		} else {
			size_t npix = 0;
			for (int v = 0; v < nviews; ++v) if (e.viewSize(v) > npix) npix = e.viewSize(v);
			const int per = (nviews + world - 1)/world;                           // equal contributions: short ranks pad
			std::vector<double> send(static_cast<size_t>(per)*npix, std::numeric_limits<double>::quiet_NaN());
			std::vector<double> recv(static_cast<size_t>(world)*per*npix);
			for (int v = lo; v < hi && ok; ++v) ok = e.getDepth(v, &send[static_cast<size_t>(v - lo)*npix]);
			if (!ok) { t->abort(); return false; }
			if (!t->allGather(rank, send.data(), send.size(), recv.data())) { t->abort(); return false; }
			for (int r = 0; r < world; ++r) {
				if (r == rank) continue;
				int rlo, rhi;
				shardUnits(nviews, world, r, rlo, rhi);
				for (int v = rlo; v < rhi; ++v)
					if (!e.setDepth(v, &recv[(static_cast<size_t>(r)*per + (v - rlo))*npix])) { t->abort(); return false; }
			}
		}
	} else if (!ok) return false;
	for (int v = 0; v < nviews; ++v) if (!e.crossCheck(v)) return false;          // in view order (:427-431)
	return true;
}

// ONE TwoViewStereo pair by row bands (BASELINE.md's C3 row "+ row-band split for 2/4/8"): computeCostVolumes is
// independent per reference row (twoviewstereo.cpp:265-332, 436-500), so shard r computes rows shardUnits(height)
// of both maps; the bands are gathered, shard 0 stitches them and runs the order-dependent cross-check
// (twoviewstereo.cpp:596-672) on the whole maps.  Engine: width(), wtaRows(y0, y1), getRows(view, y0, y1, double*),
// setRows(view, y0, y1, const double*), crossCheck(); all bool.  Every shard returns the same verdict.
template <class Engine>
bool runTwoViewRowBands(Engine &e, int height, Transport *t, int rank, int *band_lo = nullptr, int *band_hi = nullptr) {
	const int world = t ? t->ranks() : 1;
	int y0, y1;
	shardUnits(height, world, rank, y0, y1);
	if (band_lo) *band_lo = y0;
	if (band_hi) *band_hi = y1;
	bool ok = y1 > y0 ? e.wtaRows(y0, y1) : true;
	if (world > 1) {
		const size_t W = e.width(), tallest = static_cast<size_t>((height + world - 1)/world);
		std::vector<double> send(2*tallest*W, std::numeric_limits<double>::quiet_NaN()), recv(static_cast<size_t>(world)*2*tallest*W);
		for (int view = 0; view < 2 && ok && y1 > y0; ++view) ok = e.getRows(view, y0, y1, &send[view*tallest*W]);
		// the status travels apart from the maps (NaN is a legal map value)
		double st = ok ? 0.0 : 1.0;
		std::vector<double> all(static_cast<size_t>(world), 1.0);
		if (!t->allGather(rank, &st, 1, all.data())) { t->abort(); return false; }
		for (int r = 0; r < world; ++r) if (all[r] != 0.0) return false;
		if (!t->allGather(rank, send.data(), send.size(), recv.data())) { t->abort(); return false; }
		if (rank == 0) {
			for (int r = 1; r < world && ok; ++r) {
				int r0, r1;
				shardUnits(height, world, r, r0, r1);
				for (int view = 0; view < 2 && ok && r1 > r0; ++view)
					ok = e.setRows(view, r0, r1, &recv[(static_cast<size_t>(r)*2 + view)*tallest*W]);
			}
		}
	}
	if (rank == 0 && ok) ok = e.crossCheck();
	if (world > 1) {
		double st = ok ? 0.0 : 1.0;
		std::vector<double> all(static_cast<size_t>(world), 1.0);
		if (!t->allGather(rank, &st, 1, all.data())) { t->abort(); return false; }
		return all[0] == 0.0;
	}
	return ok;
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
	bool finish() { return srh_synchronize(ctx_) == SRH_OK; }                     // settles the estimates in flight (capacity check, redo)
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

// one process per GPU: the context's RCCL communicator (srh_comm_init) carries the exchange over xGMI.  The views'
// depth maps go device to device (srh_comm_allgather_views: pack, one ncclAllGather, unpack, all on the context's
// stream, nothing staged through host memory); allGather (host buffers) carries the status words.
class RcclTransport : public Transport {
public:
	RcclTransport(srh_context *ctx, int nranks) : ctx_(ctx), n_(nranks) { }
	int ranks() const { return n_; }
	bool allGather(int, const double *send, size_t count, double *recv) {
		return srh_comm_allgather_host(ctx_, send, count, recv) == SRH_OK;
	}
	bool deviceResident() const { return true; }
	bool allGatherViews(int nviews) {                                             // HipViewEngine: slot v = view v
		std::vector<int32_t> slots(static_cast<size_t>(nviews));
		for (int v = 0; v < nviews; ++v) slots[static_cast<size_t>(v)] = v;
		return srh_comm_allgather_views(ctx_, slots.data(), nviews) == SRH_OK;
	}
private:
	srh_context *ctx_; int n_;
};

// both views of one pair on one srh_context (slots 0, 1): the engine of runTwoViewRowBands
class HipPairEngine {
public:
	HipPairEngine(srh_context *ctx, const srh_params &p) : ctx_(ctx), p_(p) { srh_view_size(ctx_, 0, &w_, &h_); buf_.resize(static_cast<size_t>(w_)*h_); }
	size_t width() const { return static_cast<size_t>(w_); }
	int height() const { return h_; }
	bool wtaRows(int y0, int y1) { return srh_twoview_wta(ctx_, 0, 1, &p_, y0, y1) == SRH_OK && srh_twoview_wta(ctx_, 1, 0, &p_, y0, y1) == SRH_OK; }
	bool getRows(int view, int y0, int y1, double *out) {
		if (srh_view_depth_download(ctx_, view, buf_.data()) != SRH_OK) return false;
		std::memcpy(out, &buf_[static_cast<size_t>(y0)*w_], static_cast<size_t>(y1 - y0)*w_*sizeof(double));
		return true;
	}
	bool setRows(int view, int y0, int y1, const double *in) {
		if (srh_view_depth_download(ctx_, view, buf_.data()) != SRH_OK) return false;
		std::memcpy(&buf_[static_cast<size_t>(y0)*w_], in, static_cast<size_t>(y1 - y0)*w_*sizeof(double));
		return srh_view_depth_upload(ctx_, view, buf_.data()) == SRH_OK;
	}
	bool crossCheck() { return srh_twoview_cross_check(ctx_, 0, 1, &p_) == SRH_OK; }
private:
	srh_context *ctx_; srh_params p_; int w_ = 0, h_ = 0;
	std::vector<double> buf_;
};

} // namespace sharded
