// sharded_host_test.cpp -- host/sharded.hpp: MultiViewStereo::runTask over several shards.
//
//   sharded_host_test cpu                    (no GPU) the sharding / padding / ordering logic on a deterministic stand-in
//                                            engine: 2 and 3 ranks as threads over a LoopbackTransport == 1 rank
//   sharded_host_test gpu in.bin out.bin     (GPU) the same scene on ONE context and on 2 and 3 contexts (threads,
//                                            LoopbackTransport; every context on device (k mod device count)), and on
//                                            one context through the RCCL transport (1 rank): all bit-identical.
//                                            in.bin as host_api_test's "mvs" input; out.bin = per view double depth[w*h]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "sharded.hpp"
#include "../stereoreconstruction_amd/csrc/srh_wait.hpp"   // the bound behind every wait of the exchange (plain C++)

// ---- a deterministic stand-in: "depth maps" are small vectors; crossCheck(v) folds every other view's current map
// into view v (order dependent, like multiviewstereo.cpp:694-719 reading already-filtered maps)
struct FakeEngine {
	std::vector<std::vector<double> > maps;
	std::vector<std::pair<char, int> > log;
	bool fail_estimate = false;
	explicit FakeEngine(int nviews) {
		for (int v = 0; v < nviews; ++v) maps.push_back(std::vector<double>(5 + 3*(v % 3), std::nan("")));   // views differ in size
	}
	size_t viewSize(int v) const { return maps[v].size(); }
	bool initialEstimate(int v) { if (fail_estimate) return false; log.push_back({'e', v}); for (size_t i = 0; i < maps[v].size(); ++i) maps[v][i] = 10.0*v + 0.25*i; return true; }
	bool getDepth(int v, double *out) { std::memcpy(out, maps[v].data(), maps[v].size()*sizeof(double)); return true; }
	bool setDepth(int v, const double *in) { std::memcpy(maps[v].data(), in, maps[v].size()*sizeof(double)); return true; }
	bool crossCheck(int v) {
		log.push_back({'c', v});
		for (size_t u = 0; u < maps.size(); ++u) if ((int)u != v)
			for (size_t i = 0; i < maps[v].size(); ++i) maps[v][i] = 0.5*maps[v][i] + 0.125*maps[u][i % maps[u].size()];
		return true;
	}
};

// a stand-in pair: row y of map `view` is a function of (view, y, x); the "cross-check" folds the OTHER map into each
// map, left first, then right reading the already-filtered left map (twoviewstereo.cpp:596-672's order dependence)
struct FakePair {
	size_t w; int h;
	std::vector<double> maps[2];
	bool checked = false;
	FakePair(size_t w_, int h_) : w(w_), h(h_) { for (auto &m : maps) m.assign(w*h, std::nan("")); }
	size_t width() const { return w; }
	bool wtaRows(int y0, int y1) {
		for (int view = 0; view < 2; ++view) for (int y = y0; y < y1; ++y) for (size_t x = 0; x < w; ++x)
			maps[view][y*w + x] = (x + y) % 7 == 3 ? std::nan("") : 100.0*view + y + 0.125*x;
		return true;
	}
	bool getRows(int view, int y0, int y1, double *out) { std::memcpy(out, &maps[view][y0*w], (y1 - y0)*w*sizeof(double)); return true; }
	bool setRows(int view, int y0, int y1, const double *in) { std::memcpy(&maps[view][y0*w], in, (y1 - y0)*w*sizeof(double)); return true; }
	bool crossCheck() {
		checked = true;
		for (int view = 0; view < 2; ++view) for (int y = 0; y < h; ++y) for (size_t x = 0; x < w; ++x) {
			const double o = maps[1 - view][((y*5 + 3) % h)*w + x];                   // reads a row of (usually) another band
			if (o == o) maps[view][y*w + x] = 0.5*maps[view][y*w + x] + 0.25*o;
		}
		return true;
	}
};

static bool sameBits(const std::vector<double> &a, const std::vector<double> &b) {
	return a.size() == b.size() && !std::memcmp(a.data(), b.data(), a.size()*sizeof(double));
}

static int cpuTest() {
	for (int nviews : {1, 3, 5, 8}) {
		FakeEngine ref(nviews);
		std::vector<int> mine;
		if (!sharded::runMultiView(ref, nviews, nullptr, 0, &mine) || (int)mine.size() != nviews) return 1;
		for (int world : {2, 3}) {
			sharded::LoopbackTransport t(world);
			std::vector<FakeEngine> eng(world, FakeEngine(nviews));
			std::vector<std::vector<int> > own(world);
			std::vector<int> ok(world, 0);
			std::vector<std::thread> th;
			for (int r = 0; r < world; ++r) th.emplace_back([&, r] { ok[r] = sharded::runMultiView(eng[r], nviews, &t, r, &own[r]) ? 1 : 0; });
			for (auto &x : th) x.join();
			int covered = 0;
			for (int r = 0; r < world; ++r) {
				if (!ok[r]) { fprintf(stderr, "rank %d of %d failed\n", r, world); return 2; }
				for (int v = 0; v < nviews; ++v) if (!sameBits(eng[r].maps[v], ref.maps[v])) { fprintf(stderr, "views %d world %d rank %d view %d differs\n", nviews, world, r, v); return 3; }
				// estimates only for own views, then the full chain in view order
				size_t k = 0;
				for (int v : own[r]) { if (eng[r].log[k].first != 'e' || eng[r].log[k].second != v) return 4; ++k; }
				for (int v = 0; v < nviews; ++v) { if (eng[r].log[k].first != 'c' || eng[r].log[k].second != v) return 5; ++k; }
				if (k != eng[r].log.size()) return 6;
				covered += (int)own[r].size();
			}
			if (covered != nviews) return 7;
		}
	}
	int lo, hi;
	sharded::shardUnits(8, 3, 0, lo, hi); if (lo != 0 || hi != 3) return 8;
	sharded::shardUnits(8, 3, 2, lo, hi); if (lo != 6 || hi != 8) return 8;
	// ---- a shard whose estimate fails: every shard returns false, none is left waiting in the exchange
	for (int bad = 0; bad < 3; ++bad) {
		const int world = 3, nviews = 7;
		sharded::LoopbackTransport t(world);
		std::vector<FakeEngine> eng(world, FakeEngine(nviews));
		eng[bad].fail_estimate = true;
		std::vector<int> ok(world, 1);
		std::vector<std::thread> th;
		for (int r = 0; r < world; ++r) th.emplace_back([&, r] { ok[r] = sharded::runMultiView(eng[r], nviews, &t, r) ? 1 : 0; });
		for (auto &x : th) x.join();
		for (int r = 0; r < world; ++r) if (ok[r]) { fprintf(stderr, "failing shard %d: shard %d reported success\n", bad, r); return 9; }
	}
	// ---- one pair by row bands: 2, 3 and 5 shards (uneven bands, one of them empty when rows < shards) == 1 shard
	for (int height : {3, 16, 37}) {
		FakePair ref(11, height);
		if (!sharded::runTwoViewRowBands(ref, height, nullptr, 0)) return 10;
		for (int world : {2, 3, 5}) {
			sharded::LoopbackTransport t(world);
			std::vector<FakePair> eng(world, FakePair(11, height));
			std::vector<int> ok(world, 0);
			std::vector<std::thread> th;
			for (int r = 0; r < world; ++r) th.emplace_back([&, r] { ok[r] = sharded::runTwoViewRowBands(eng[r], height, &t, r) ? 1 : 0; });
			for (auto &x : th) x.join();
			for (int r = 0; r < world; ++r) if (!ok[r]) { fprintf(stderr, "row bands: rank %d of %d failed\n", r, world); return 11; }
			for (int view = 0; view < 2; ++view)
				if (!sameBits(eng[0].maps[view], ref.maps[view])) { fprintf(stderr, "row bands: height %d world %d view %d differs\n", height, world, view); return 12; }
			for (int r = 1; r < world; ++r) if (eng[r].checked) return 13;            // the cross-check runs on shard 0 only
		}
	}
	// ---- the wait behind every collective is bounded (csrc/srh_wait.hpp): a transport that never completes
	// -- a peer that died inside the collective -- ends in an error text after the timeout, one that fails says so at once,
	// one that completes returns in time
	{
		char msg[200];
		const auto t0 = std::chrono::steady_clock::now();
		int polls = 0;
		const char *e = srh::bounded_wait([&] { ++polls; return 1; }, nullptr, 150, "a collective that never completes", msg, sizeof(msg));
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		if (!e || !strstr(e, "still in progress after 150 ms") || ms < 140 || ms > 2000 || polls < 10) { fprintf(stderr, "bounded wait: '%s' after %.1f ms, %d polls\n", e ? e : "(null)", ms, polls); return 14; }
		polls = 0;
		e = srh::bounded_wait([&] { return ++polls < 5 ? 1 : 2; }, [] { return "peer gone"; }, 60000, "a failing collective", msg, sizeof(msg));
		if (!e || strcmp(e, "peer gone") || polls != 5) return 15;
		polls = 0;
		e = srh::bounded_wait([&] { return ++polls < 7 ? 1 : 0; }, [] { return "unused"; }, 60000, "a collective", msg, sizeof(msg));
		if (e || polls != 7) return 16;
		e = srh::bounded_wait([] { return 3; }, nullptr, 1000, "a call", msg, sizeof(msg));          // failed without a text of its own
		if (!e || strcmp(e, "a call failed")) return 17;
	}
	printf("cpu ok\n");
	return 0;
}

template <class T> static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

static int gpuTest(const char *in, const char *out) {
	FILE *f = fopen(in, "rb");
	if (!f) { perror(in); return 2; }
	int32_t hdr[6]; double dh[4];
	rd(f, hdr, 6); rd(f, dh, 4);
	const int nviews = hdr[0], w = hdr[1], h = hdr[2];
	srh_params p;
	srh_params_mvs_defaults(&p);
	p.num_depth_levels = hdr[3]; p.window_radius = hdr[4]; p.weight_kind = hdr[5];
	p.min_depth = dh[0]; p.max_depth = dh[1]; p.image_scale = dh[2]; p.cross_check_threshold = dh[3];
	std::vector<srh_camera> cams(nviews);
	std::vector<std::vector<uint8_t> > rgba(nviews), mask(nviews);
	for (int v = 0; v < nviews; ++v) {
		double K[9], R[9], t[3], dist[5];
		rd(f, K, 9); rd(f, R, 9); rd(f, t, 3); rd(f, dist, 5);
		if (srh_camera_from_krt(K, R, t, dist, nullptr, 0.0, 1.0, &cams[v]) != SRH_OK) return 3;
		rgba[v].resize((size_t)w*h*4); rd(f, rgba[v].data(), rgba[v].size());
		mask[v].resize((size_t)w*h);
		for (size_t i = 0; i < mask[v].size(); ++i) mask[v][i] = rgba[v][i*4 + 3] == 255 ? 1 : 0;
	}
	fclose(f);
	int ndev = 0;
	srh_device_count(&ndev);
	if (ndev < 1) { fprintf(stderr, "no device\n"); return 3; }
	auto make = [&](int dev) -> srh_context * {
		srh_context *c = nullptr;
		if (srh_create(dev, &c) != SRH_OK) { fprintf(stderr, "create: %s\n", srh_last_error()); return nullptr; }
		for (int v = 0; v < nviews; ++v)
			if (srh_view_upload(c, v, w, h, rgba[v].data(), mask[v].data(), &cams[v]) != SRH_OK) { fprintf(stderr, "upload: %s\n", srh_last_error()); return nullptr; }
		return c;
	};
	auto maps = [&](srh_context *c) {
		std::vector<std::vector<double> > m(nviews, std::vector<double>((size_t)w*h));
		for (int v = 0; v < nviews; ++v) srh_view_depth_download(c, v, m[v].data());
		return m;
	};
	// single context: the reference's runTask order
	srh_context *c0 = make(0);
	if (!c0) return 3;
	sharded::HipViewEngine e0(c0, cams, p);
	if (!e0.ok() || !sharded::runMultiView(e0, nviews, nullptr, 0)) { fprintf(stderr, "single: %s\n", srh_last_error()); return 4; }
	const std::vector<std::vector<double> > want = maps(c0);
	// 2 and 3 contexts in one process (several GPUs when there are several; the dynamic-LDS attribute is per device)
	for (int world : {2, 3}) {
		sharded::LoopbackTransport t(world);
		std::vector<srh_context *> ctx(world);
		for (int r = 0; r < world; ++r) if (!(ctx[r] = make(r % ndev))) return 3;
		std::vector<int> ok(world, 0);
		std::vector<std::thread> th;
		for (int r = 0; r < world; ++r) th.emplace_back([&, r] {
			sharded::HipViewEngine e(ctx[r], cams, p);
			ok[r] = (e.ok() && sharded::runMultiView(e, nviews, &t, r)) ? 1 : 0;
		});
		for (auto &x : th) x.join();
		for (int r = 0; r < world; ++r) {
			if (!ok[r]) { fprintf(stderr, "world %d rank %d failed: %s\n", world, r, srh_last_error()); return 5; }
			const std::vector<std::vector<double> > got = maps(ctx[r]);
			for (int v = 0; v < nviews; ++v) if (!sameBits(got[v], want[v])) { fprintf(stderr, "world %d rank %d view %d differs\n", world, r, v); return 6; }
			srh_destroy(ctx[r]);
		}
	}
	// the RCCL transport on a one-rank communicator (all a one-GPU box allows)
	{
		srh_context *c = make(0);
		if (!c) return 3;
		unsigned char id[SRH_COMM_ID_BYTES];
		if (srh_comm_unique_id(id) != SRH_OK || srh_comm_init(c, 1, 0, id) != SRH_OK) { fprintf(stderr, "rccl: %s\n", srh_last_error()); return 7; }
		sharded::RcclTransport t(c, 1);
		std::vector<double> a(1000), b(1000, -1.0);
		for (size_t i = 0; i < a.size(); ++i) a[i] = 0.5*i;
		if (!t.allGather(0, a.data(), a.size(), b.data()) || !sameBits(a, b)) { fprintf(stderr, "rccl all-gather: %s\n", srh_last_error()); return 8; }
		sharded::HipViewEngine e(c, cams, p);
		if (!e.ok() || !sharded::runMultiView(e, nviews, &t, 0)) return 9;
		const std::vector<std::vector<double> > got = maps(c);
		for (int v = 0; v < nviews; ++v) if (!sameBits(got[v], want[v])) return 10;
		srh_comm_destroy(c);
		srh_destroy(c);
	}
	FILE *o = fopen(out, "wb");
	if (!o) { perror(out); return 2; }
	for (int v = 0; v < nviews; ++v) fwrite(want[v].data(), sizeof(double), want[v].size(), o);
	fclose(o);
	srh_destroy(c0);
	printf("gpu ok: %d views on 1, 2 and 3 contexts (%d device%s) and through RCCL\n", nviews, ndev, ndev == 1 ? "" : "s");
	return 0;
}

int main(int argc, char **argv) {
	if (argc == 2 && !strcmp(argv[1], "cpu")) return cpuTest();
	if (argc == 4 && !strcmp(argv[1], "gpu")) return gpuTest(argv[2], argv[3]);
	fprintf(stderr, "usage: %s cpu | gpu in.bin out.bin\n", argv[0]);
	return 64;
}
