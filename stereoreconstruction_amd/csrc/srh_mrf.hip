// srh_mrf.hip -- the MRF branch of MultiViewStereo::computeInitialEstimate (multiviewstereo.cpp:481-516, 610-652;
// SURVEY 8(f) rank 2): sequential TRW-S over the W x H grid with K + 1 labels (K collected peaks + "unknown").
//
// PARITY UNPINNED: the reference links the third-party -lMRF library (StereoReconstruction.pro:100-103), which is
// neither in its tree nor in this image; the algorithm is the published one (Kolmogorov, PAMI 2006) in the order
// oracle/sr_oracle.c (sro_mvs_mrf) writes down, and the kernels are checked bit for bit against that.
//
// TRW-S visits pixels in scan order and each pixel needs the fresh messages of its left and upper neighbours, so the
// only parallelism inside a sweep is along anti-diagonals.  One workgroup owns a band of 16 image rows and walks it
// diagonal by diagonal: 16 pixels per step, 16 lanes per pixel (one per label).  The message to the right-hand
// neighbour never leaves its lanes' registers, the message to the pixel below crosses to the next 16 lanes through
// LDS, and the message out of a band's last row goes to the next band's workgroup through device memory as
// self-validating granules (the HIP guide's R2 hand-off): every double travels as two 8-byte words {pass tag, half},
// written through (sc1) and read with sc1 loads, so the data is its own flag -- no progress word, no release, no
// drain of the storing wave.  A consumer fetches its granules a chunk ahead; only when a tag is not yet the current
// pass's does it wait, and then until the band above is MRF_LAG columns ahead, so that the following prefetches are
// valid at first read.  Bands therefore run as a pipeline, each ~36 columns behind the one above, and a pass takes
// about W + 36 * H/16 steps instead of W * H (measured: 1.3 us per step averaged over the three passes of a sweep,
// whatever the number of bands -- the dependent chain LDS -> adds -> DPP min-reductions -> LDS -> barrier of one step).
// The four computing waves of a workgroup only load; a fifth wave writes each step's results from LDS to device
// memory one step later (a wave that mixes prefetching loads with stores must drain both to use a prefetched value).
//
// Within a pixel the label vectors are exchanged with DPP row broadcasts (v_mov_b64_dpp row_newbcast: no LDS).
// Per-step inputs (data costs, stored messages, peak depths) are fetched a chunk of 4 steps ahead into registers.
// The smoothness cost 2|z1 - z2| / (z1 + z2) is evaluated where it is used (an IEEE division per label pair): the
// table the CPU library caches would be 1.6 KB per pixel.
#include "srh_internal.hpp"

// (the any-K variant keeps a run-time exit in its 16-trip label loops; clang then reports the unroll request as not honoured)
#pragma clang diagnostic ignored "-Wpass-failed"

namespace srh {

namespace {

constexpr int MRF_ROWS = 16;            // rows per band = pixel groups per workgroup
constexpr int MRF_LANES = 16;           // lanes per pixel = label slots (K + 1 <= 16)
constexpr int MRF_CH = 4;               // steps per prefetched chunk
constexpr int MRF_THREADS = 320;        // 4 waves that compute (16 rows x 16 label lanes) + 1 that stores
constexpr int MRF_LAG = 12;             // columns a band drops behind the one above once it had to wait for it
constexpr unsigned MRF_SPIN_LIMIT = 1u << 21;
constexpr size_t MRF_LDS_BYTES = 12*1024;   // two hand-over buffers of 2 x 2 KB, labels, control words (several bands may share a CU:
                                            // bands of other views' sweeps running at the same time fill each other's waits)

// sync block (4 unsigned words, zeroed before every pass): [0] ticket
// status block (zeroed once per run): [0] abort, [1] first band that gave up + 1, [2] pass it gave up in + 1
constexpr int SY_TICKET = 0;
constexpr int ST_ABORT = 0, ST_WHO = 1, ST_PASS = 2;

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// The per-step barrier.  __syncthreads() carries a workgroup-scope release, which makes a wave wait for its own
// device-memory operations before it may arrive; what the step barrier has to order is LDS only (the two hand-over
// buffers): the device-memory results go to other workgroups as granules or to the next launch.
__device__ __forceinline__ void mrf_step_barrier() {
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int I> __device__ __forceinline__ double bc16(double v) {
	return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + I, 0xf, 0xf, false);      // row_newbcast:I
}

typedef unsigned long long u64;
// a double as two granules {tag, low half}, {tag, high half}: each one aligned 8-byte write-through store
__device__ __forceinline__ void put_granules(u64 *g, unsigned tag, double v) {
	const u64 bits = (u64)__double_as_longlong(v), t = (u64)tag << 32;
	__hip_atomic_store(g, t | (bits & 0xffffffffull), RLX_AGENT);
	__hip_atomic_store(g + 1, t | (bits >> 32), RLX_AGENT);
}
__device__ __forceinline__ u64 get_granule(const u64 *g) { return __hip_atomic_load(g, RLX_AGENT); }
__device__ __forceinline__ bool granules_ok(u64 g0, u64 g1, unsigned tag) { return (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag; }
__device__ __forceinline__ double granules_value(u64 g0, u64 g1) { return __longlong_as_double((long long)((g0 & 0xffffffffull) | (g1 << 32))); }

// smoothnessCost (multiviewstereo.cpp:499-514) between a peak label of one pixel (depth z1) and label kd of another (z2)
// (written as selects over an unconditional quotient: the divisions of one step are independent of each other and of
// the messages, and only straight-line code lets the scheduler overlap them)
__device__ __forceinline__ double smooth_peak(double z1, double z2, bool kd_unknown, double psi, double psi2) {
	const double q = 2.0 * fabs(z1 - z2) / (z1 + z2);
	const double r = (z1 < 0 || z2 < 0) ? psi2 : q;
	return kd_unknown ? psi : r;
}
// the same with the first label given as a carried value: NaN = "unknown", otherwise that label's depth
__device__ __forceinline__ double smooth_carried(double c, double z2, bool kd_unknown, double psi, double psi2) {
	if (c != c) return kd_unknown ? 0.0 : psi;
	return smooth_peak(c, z2, kd_unknown, psi, psi2);
}

struct Chunk {
	double D[MRF_CH], oH[MRF_CH], oV[MRF_CH], zs[MRF_CH + 1], zV[MRF_CH];
	u64 g0[MRF_CH], g1[MRF_CH];          // first row of a band: what the band above handed down (granules)
};

// runs its body once per label KS = 0 .. L-1, KS a compile-time constant (a DPP lane select is an immediate)
#define MRF_FOR_LABELS(...) \
	_Pragma("unroll") for (int ks_ = 0; ks_ < (LT ? LT : 16); ++ks_) { if (!LT && ks_ >= L) break; \
		switch (ks_) { \
		case 0: { constexpr int KS = 0; __VA_ARGS__ } break;  case 1: { constexpr int KS = 1; __VA_ARGS__ } break; \
		case 2: { constexpr int KS = 2; __VA_ARGS__ } break;  case 3: { constexpr int KS = 3; __VA_ARGS__ } break; \
		case 4: { constexpr int KS = 4; __VA_ARGS__ } break;  case 5: { constexpr int KS = 5; __VA_ARGS__ } break; \
		case 6: { constexpr int KS = 6; __VA_ARGS__ } break;  case 7: { constexpr int KS = 7; __VA_ARGS__ } break; \
		case 8: { constexpr int KS = 8; __VA_ARGS__ } break;  case 9: { constexpr int KS = 9; __VA_ARGS__ } break; \
		case 10: { constexpr int KS = 10; __VA_ARGS__ } break; case 11: { constexpr int KS = 11; __VA_ARGS__ } break; \
		case 12: { constexpr int KS = 12; __VA_ARGS__ } break; case 13: { constexpr int KS = 13; __VA_ARGS__ } break; \
		case 14: { constexpr int KS = 14; __VA_ARGS__ } break; default: { constexpr int KS = 15; __VA_ARGS__ } break; } }

} // namespace

struct MrfPassArgs {
	int W, H, K;
	double psi, psi2;
	const double *pz;        // [n][16] peak depths (label K and above: 0)
	const double *D;         // [n][16] data costs
	double *Mh, *Mv;         // [n][16] message stored on the edge (n, n+1) / (n, n+W)
	int32_t *ans;            // [n]
	unsigned long long *hand; // [band][logical column][16 lanes][2] granules out of each band's last row
	unsigned epoch;          // tag of this pass: unique within a run, never 0
	unsigned *sync, *status;
};

// MODE 0: forward sweep, 1: backward sweep (logical coordinates mirrored), 2: labels read off (forward order)
// LT: the label count K + 1 when known at compile time (10 for the reference's K = 9: straight-line label loops), 0: any
template <int MODE, int LT>
__global__ __launch_bounds__(MRF_THREADS, 1) void mrf_pass_kernel(const MrfPassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	double (*down)[MRF_ROWS][MRF_LANES] = reinterpret_cast<double (*)[MRF_ROWS][MRF_LANES]>(smem);   // [2][row][label]: message to the row below
	double (*outH)[MRF_ROWS][MRF_LANES] = down + 2;                                                   // [2][row][label]: message along the row
	int (*labs)[MRF_ROWS] = reinterpret_cast<int (*)[MRF_ROWS]>(smem + 4*MRF_ROWS*MRF_LANES*sizeof(double));   // [2][row]: chosen label
	int *s_ctl = reinterpret_cast<int *>(smem + 4*MRF_ROWS*MRF_LANES*sizeof(double) + 2*MRF_ROWS*sizeof(int));   // [0] band, [1] abort

	const int tid = threadIdx.x, r = tid >> 4, kd = tid & 15, wave = tid >> 6;
	const int W = a.W, H = a.H, K = LT ? LT - 1 : a.K, L = K + 1;
	const double psi = a.psi, psi2 = a.psi2;
	if (tid == 0) { s_ctl[0] = (int)atomicAdd(&a.sync[SY_TICKET], 1u); s_ctl[1] = 0; }
	__syncthreads();
	const int b = s_ctl[0];
	const int v = b*MRF_ROWS + r;                                    // logical row
	const bool rowok = v < H;
	const int y = MODE == 1 ? H - 1 - v : v;
	const bool hasV = v < H - 1;                                     // an edge to the next logical row
	const bool unknown = kd == K;
	const int rlast = min(MRF_ROWS, H - b*MRF_ROWS) - 1;             // the band's last row
	const int dstep = MODE == 1 ? -1 : 1;                            // physical step to the next logical column / row
	const int nsteps = W + MRF_ROWS - 1;
	const int nchunks = (nsteps + MRF_CH - 1)/MRF_CH;

	// ---- the storing wave.  A wave that both prefetches (loads) and stores has to wait for ALL its memory operations
	// whenever it needs a prefetched value (loads and stores share one counter and complete out of order against each
	// other), i.e. for the round trip of the stores of the step just finished, every chunk.  So the four computing waves
	// only load; what a step produced sits in LDS (the row-below hand-over is there anyway) and this wave writes the
	// previous step's values to device memory while the others compute the next: it never waits for anything.
	if (wave == 4) {
		const int lane = tid & 63;
		for (int s = 0; s <= nchunks*MRF_CH; ++s) {
			const int t = s - 1, par = t & 1;                          // the step whose outputs are written now
			if (t >= 0) {
				if (MODE == 2) {
					if (lane < MRF_ROWS) {
						const int rr = lane, vv = b*MRF_ROWS + rr, u = t - rr;
						if (vv < H && u >= 0 && u < W) {
							const long n = (long)vv*W + u;
							a.ans[n] = labs[par][rr];
							if (rr == rlast) put_granules(a.hand + (((size_t)b*W + u)*16)*2, a.epoch, down[par][rr][0]);
						}
					}
				} else {
#pragma unroll
					for (int k = 0; k < 2*MRF_ROWS*MRF_LANES/64; ++k) {
						const int idx = k*64 + lane;
						const int which = idx >> 8, rr = (idx >> 4) & 15, kk = idx & 15;
						const int vv = b*MRF_ROWS + rr, u = t - rr;
						if (vv < H && u >= 0 && u < W && kk < L) {
							const int yy = MODE == 1 ? H - 1 - vv : vv;
							const long n = (long)yy*W + (MODE == 1 ? W - 1 - u : u);
							if (which == 0) {
								if (u < W - 1) a.Mh[(MODE == 1 ? n - 1 : n)*16 + kk] = outH[par][rr][kk];
							} else if (vv < H - 1) {
								const double m = down[par][rr][kk];
								a.Mv[(MODE == 1 ? n - W : n)*16 + kk] = m;
								if (rr == rlast) put_granules(a.hand + (((size_t)b*W + u)*16 + kk)*2, a.epoch, m);
							}
						}
					}
				}
			}
			if (s == nchunks*MRF_CH) break;
			mrf_step_barrier();                                        // the computing waves' barrier of step s
			if ((s % MRF_CH) == MRF_CH - 1 && s_ctl[1]) break;         // they leave here too
		}
		return;
	}

	const unsigned epoch = a.epoch;
	const bool takes = r == 0 && b > 0;                              // this lane's row is fed by the band above
	const bool hlane = MODE == 2 ? true : kd < L;                    // lanes whose granules exist (solve pass: slot 0, read by all)

	// physical pixel index of logical column u in this lane's row
	auto pix = [&](int u) -> long { return (long)y*W + (MODE == 1 ? W - 1 - u : u); };
	// granules the band above wrote for logical column u (this lane's label; solve pass: the one value of the pixel)
	auto hand_at = [&](int u) -> const u64 * { return a.hand + (((size_t)(b - 1)*W + u)*16 + (MODE == 2 ? 0 : kd))*2; };

	auto load_chunk = [&](int c, Chunk &B) {
#pragma unroll
		for (int j = 0; j <= MRF_CH; ++j) {
			const int u = c*MRF_CH + j - r;
			const bool ok = rowok && u >= 0 && u < W;
			const long n = pix(u);
			B.zs[j] = ok ? a.pz[n*16 + kd] : 0.0;
			if (j == MRF_CH) break;
			B.D[j] = ok ? a.D[n*16 + kd] : 0.0;
			// the stored message on the edge towards the next logical column / row (written by the other sweep)
			const long eh = MODE == 1 ? n - 1 : n, ev = MODE == 1 ? n - W : n;
			B.oH[j] = (ok && u < W - 1) ? a.Mh[eh*16 + kd] : 0.0;
			B.oV[j] = (ok && hasV) ? a.Mv[ev*16 + kd] : 0.0;
			B.zV[j] = (ok && hasV) ? a.pz[(n + (long)dstep*W)*16 + kd] : 0.0;
			B.g0[j] = 0; B.g1[j] = 0;
			if (takes && ok && hlane) { const u64 *g = hand_at(u); B.g0[j] = get_granule(g); B.g1[j] = get_granule(g + 1); }
		}
	};

	// wave 0, before a chunk is used: are the granules of its MRF_CH columns this pass's?  If not: wait until the band
	// above is MRF_LAG columns further, fetch them again (until they are: stores of one wave need not land in order).
	auto settle = [&](int c, Chunk &B) {
		if (b == 0) return;
		bool good = true;
#pragma unroll
		for (int j = 0; j < MRF_CH; ++j) {
			const int u = c*MRF_CH + j;
			if (takes && hlane && u < W) good = good && granules_ok(B.g0[j], B.g1[j], epoch);
		}
		if (__all(good)) return;
		const int ufar = min(c*MRF_CH + MRF_CH - 1 + MRF_LAG, W - 1);
		unsigned spins = 0;
		int ok = 1;
		for (;;) {
			bool far = true;
			if (takes && hlane) { const u64 *g = hand_at(ufar); far = granules_ok(get_granule(g), get_granule(g + 1), epoch); }
			good = true;
#pragma unroll
			for (int j = 0; j < MRF_CH; ++j) {
				const int u = c*MRF_CH + j;
				if (takes && hlane && u < W) {
					const u64 *g = hand_at(u);
					B.g0[j] = get_granule(g); B.g1[j] = get_granule(g + 1);
					good = good && granules_ok(B.g0[j], B.g1[j], epoch);
				}
			}
			if (__all(far && good)) break;
			if (++spins > MRF_SPIN_LIMIT || __hip_atomic_load(&a.status[ST_ABORT], RLX_AGENT)) { ok = 0; break; }   // uniform
			__builtin_amdgcn_s_sleep(4);
		}
		if (!ok && tid == 0) {
			__hip_atomic_store(&a.status[ST_ABORT], 1u, RLX_AGENT);
			if (atomicCAS(&a.status[ST_WHO], 0u, (unsigned)b + 1u) == 0u) a.status[ST_PASS] = MODE + 1;
			s_ctl[1] = 1;
		}
	};

	Chunk cur, nxt;
	load_chunk(0, nxt);
	double carryL = 0.0;                                             // message (or chosen depth) handed along the row

	for (int c = 0; c < nchunks; ++c) {
		cur = nxt;
		if (wave == 0) settle(c, cur);
		if (c + 1 < nchunks) load_chunk(c + 1, nxt);
#pragma unroll
		for (int j = 0; j < MRF_CH; ++j) {
			const int s = c*MRF_CH + j;
			const int u = s - r;
			const double fromL = u > 0 ? carryL : 0.0;
			const double lds_top = down[(s + 1) & 1][(r + MRF_ROWS - 1) & (MRF_ROWS - 1)][MODE == 2 ? 0 : kd];
			const double fromT = r == 0 ? (b > 0 ? granules_value(cur.g0[j], cur.g1[j]) : 0.0) : lds_top;
			const double zs = cur.zs[j], zH = cur.zs[j + 1], zV = cur.zV[j];

			if (MODE == 2) {
				// Di = D + V(left's label, .) + V(upper label, .) + message from the right + message from below
				double Di = cur.D[j];
				if (u > 0) Di += smooth_carried(fromL, zs, unknown, psi, psi2);
				if (v > 0) Di += smooth_carried(fromT, zs, unknown, psi, psi2);
				Di += cur.oH[j];
				Di += cur.oV[j];
				double best = 0.0, cbest = 0.0;
				int lab = 0;
				MRF_FOR_LABELS(
					const double dk = bc16<KS>(Di); const double zk = bc16<KS>(zs);
					if (KS == 0 || best > dk) { best = dk; lab = KS; cbest = zk; }
				)
				if (lab == K) cbest = __builtin_nan("");
				carryL = cbest;
				if (kd == 0) { down[s & 1][r][0] = cbest; labs[s & 1][r] = lab; }
			} else {
				double Di = cur.D[j];
				if (MODE == 0) { Di += fromL; Di += fromT; Di += cur.oH[j]; Di += cur.oV[j]; }   // left, up, right, down
				else           { Di += cur.oH[j]; Di += cur.oV[j]; Di += fromL; Di += fromT; }
				if (MODE == 1) {
					double vmin = 0.0;
					MRF_FOR_LABELS( const double dk = bc16<KS>(Di); if (KS == 0 || vmin > dk) vmin = dk; )
					Di -= vmin;
				}
				const double bufH = 0.5*Di - cur.oH[j], bufV = 0.5*Di - cur.oV[j];
				double mH = 0.0, mV = 0.0;
				MRF_FOR_LABELS(
					const double bh = bc16<KS>(bufH); const double bv = bc16<KS>(bufV);
					double vh, vv;
					if (KS == K) { vh = vv = unknown ? 0.0 : psi; }
					else {
						const double z1 = bc16<KS>(zs);
						vh = smooth_peak(z1, zH, unknown, psi, psi2);
						vv = smooth_peak(z1, zV, unknown, psi, psi2);
					}
					const double th = bh + vh; const double tv = bv + vv;
					if (KS == 0 || mH > th) mH = th;
					if (KS == 0 || mV > tv) mV = tv;
				)
				double dH = 0.0, dV = 0.0;
				MRF_FOR_LABELS(
					const double h = bc16<KS>(mH); const double w = bc16<KS>(mV);
					if (KS == 0 || dH > h) dH = h;
					if (KS == 0 || dV > w) dV = w;
				)
				mH -= dH; mV -= dV;
				carryL = mH;
				down[s & 1][r][kd] = mV;
				outH[s & 1][r][kd] = mH;
			}
			mrf_step_barrier();
		}
		if (s_ctl[1]) break;                                         // a wait gave up: leave (results are reported invalid)
	}
}

// dataCost (multiviewstereo.cpp:485-497) for every pixel and label, and the peak depths in the lane layout
__global__ void mrf_data_kernel(long n, int K, double beta, double lambda, double phiu, const double *__restrict__ peaks,
                                double *__restrict__ D, double *__restrict__ pz)
{
	const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= n*16) return;
	const long p = i >> 4;
	const int l = (int)(i & 15);
	double d = 0.0, z = 0.0;
	if (l < K) {
		const double cost = peaks[(p*K + l)*2], depth = peaks[(p*K + l)*2 + 1];
		z = depth;
		d = depth < 0 ? lambda : lambda*exp(-beta * cost);
	} else if (l == K) d = phiu;
	D[i] = d;
	pz[i] = z;
}

// totalEnergy() of the current labels: one term set per pixel (its data cost, its edges to the left and up), summed per block
__global__ __launch_bounds__(256) void mrf_energy_kernel(int W, int H, int K, double psi, double psi2, const double *__restrict__ D,
                                                         const double *__restrict__ pz, const int32_t *__restrict__ ans,
                                                         double *__restrict__ partial)
{
	__shared__ double red[256];
	const long n = (long)blockIdx.x*256 + threadIdx.x;
	double e = 0.0;
	if (n < (long)W*H) {
		const int x = (int)(n % W), y = (int)(n / W);
		const int a = ans[n];
		e = D[n*16 + a];
		const double c = a == K ? __builtin_nan("") : pz[n*16 + a];
		if (x > 0) { const int o = ans[n - 1]; e += smooth_carried(c, o == K ? 0.0 : pz[(n - 1)*16 + o], o == K, psi, psi2); }
		if (y > 0) { const int o = ans[n - W]; e += smooth_carried(c, o == K ? 0.0 : pz[(n - W)*16 + o], o == K, psi, psi2); }
	}
	red[threadIdx.x] = e;
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
		__syncthreads();
	}
	if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// fixed-shape sum of the per-block partials (one block): same bits on every run
__global__ __launch_bounds__(1024) void mrf_energy_sum_kernel(int nparts, const double *__restrict__ partial, double *__restrict__ out)
{
	__shared__ double red[1024];
	double e = 0.0;
	for (int i = threadIdx.x; i < nparts; i += 1024) e += partial[i];
	red[threadIdx.x] = e;
	__syncthreads();
	for (int s = 512; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
		__syncthreads();
	}
	if (threadIdx.x == 0) *out = red[0];
}

// labels -> depths where the mask is WHITE (multiviewstereo.cpp:645-652)
__global__ void mrf_depth_kernel(const ViewDev *__restrict__ views, int slot, int K, const double *__restrict__ pz,
                                 const int32_t *__restrict__ ans)
{
	const ViewDev &V = views[slot];
	const long n = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (n >= (long)V.w*V.h) return;
	if (!V.mask[n]) return;
	const int label = ans[n];
	const double d = label == K ? __builtin_inf() : pz[n*16 + label];
	V.depth[n] = d > 0 ? d : __builtin_inf();
}

static MrfPassArgs carve(double *buf, int w, int h, int K, double psi, MrfLayout &lay)
{
	const size_t n = (size_t)w*h, nr = (n + 1) & ~(size_t)1;           // every block starts 16-byte aligned
	lay.pz = buf; lay.D = buf + n*16; lay.Mh = buf + n*32; lay.Mv = buf + n*48;
	lay.nbands = (h + MRF_ROWS - 1)/MRF_ROWS;
	lay.hand = reinterpret_cast<unsigned long long *>(buf + n*64);
	lay.hand_words = (size_t)lay.nbands*w*32;
	lay.partial = buf + n*64 + lay.hand_words;
	lay.nparts = (int)((n + 255)/256);
	lay.energy = lay.partial + ((lay.nparts + 1) & ~1);                // [0] energy, [1] unused
	lay.status = reinterpret_cast<unsigned *>(lay.energy + 2);         // 4 words
	lay.ans = reinterpret_cast<int32_t *>(lay.energy + 4);
	lay.sync = reinterpret_cast<unsigned *>(lay.ans + nr);
	lay.sync_words = 4;
	lay.total_doubles = (size_t)(reinterpret_cast<double *>(lay.sync) - buf) + lay.sync_words/2;
	MrfPassArgs a;
	a.W = w; a.H = h; a.K = K; a.psi = psi; a.psi2 = 2*psi;
	a.pz = lay.pz; a.D = lay.D; a.Mh = lay.Mh; a.Mv = lay.Mv; a.ans = lay.ans; a.hand = lay.hand; a.epoch = 0;
	a.sync = lay.sync; a.status = lay.status;
	return a;
}

void launch_mrf_layout(double *buf, int w, int h, MrfLayout &lay) { carve(buf, w, h, 1, 0.0, lay); }

size_t mrf_scratch_doubles(int w, int h)
{
	MrfLayout lay;
	carve(nullptr, w, h, 1, 0.0, lay);
	return lay.total_doubles;
}

hipError_t launch_mrf_setup(hipStream_t st, double *buf, int w, int h, int K, double beta, double lambda, double phiu,
                            const double *peaks, MrfLayout &lay)
{
	carve(buf, w, h, K, 0.0, lay);
	const size_t n = (size_t)w*h;
	hipError_t e;
	if ((e = hipMemsetAsync(lay.Mh, 0, n*32*sizeof(double), st)) != hipSuccess) return e;        // initialize(): messages 0
	if ((e = hipMemsetAsync(lay.ans, 0, n*sizeof(int32_t), st)) != hipSuccess) return e;         // clearAnswer(): label 0
	if ((e = hipMemsetAsync(lay.energy, 0, 4*sizeof(double), st)) != hipSuccess) return e;       // energy + status words
	if ((e = hipMemsetAsync(lay.hand, 0, lay.hand_words*sizeof(unsigned long long), st)) != hipSuccess) return e;   // no tag is 0
	hipLaunchKernelGGL(mrf_data_kernel, dim3((unsigned)((n*16 + 255)/256)), dim3(256), 0, st, (long)n, K, beta, lambda, phiu, peaks, lay.D, lay.pz);
	return hipGetLastError();
}

template <int MODE, int LT> static hipError_t launch_pass_lt(hipStream_t st, const MrfPassArgs &a, const MrfLayout &lay)
{
	hipError_t e;
	if ((e = hipMemsetAsync(lay.sync, 0, lay.sync_words*sizeof(unsigned), st)) != hipSuccess) return e;
	if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mrf_pass_kernel<MODE, LT>), hipFuncAttributeMaxDynamicSharedMemorySize,
	                             (int)MRF_LDS_BYTES)) != hipSuccess) return e;
	hipLaunchKernelGGL((mrf_pass_kernel<MODE, LT>), dim3((unsigned)lay.nbands), dim3(MRF_THREADS), MRF_LDS_BYTES, st, a);
	return hipGetLastError();
}
template <int MODE> static hipError_t launch_pass(hipStream_t st, const MrfPassArgs &a, const MrfLayout &lay)
{
	return a.K == 9 ? launch_pass_lt<MODE, 10>(st, a, lay) : launch_pass_lt<MODE, 0>(st, a, lay);
}

// one optimize(1): forward sweep, backward sweep, labels read off.  `sweep` (0, 1, ...) numbers the calls of one run:
// every pass gets a granule tag of its own
hipError_t launch_mrf_sweep(hipStream_t st, double *buf, int w, int h, int K, double psiu, int sweep)
{
	MrfLayout lay;
	MrfPassArgs a = carve(buf, w, h, K, psiu, lay);
	hipError_t e;
	a.epoch = 1u + 3u*(unsigned)sweep;
	if ((e = launch_pass<0>(st, a, lay)) != hipSuccess) return e;
	a.epoch += 1;
	if ((e = launch_pass<1>(st, a, lay)) != hipSuccess) return e;
	a.epoch += 1;
	return launch_pass<2>(st, a, lay);
}

// totalEnergy() into lay.energy[0]; the status words follow it (lay.status)
hipError_t launch_mrf_energy(hipStream_t st, double *buf, int w, int h, int K, double psiu)
{
	MrfLayout lay;
	carve(buf, w, h, K, psiu, lay);
	hipLaunchKernelGGL(mrf_energy_kernel, dim3((unsigned)lay.nparts), dim3(256), 0, st, w, h, K, psiu, 2*psiu, lay.D, lay.pz, lay.ans, lay.partial);
	hipLaunchKernelGGL(mrf_energy_sum_kernel, dim3(1), dim3(1024), 0, st, lay.nparts, lay.partial, lay.energy);
	return hipGetLastError();
}

hipError_t launch_mrf_depth(hipStream_t st, const ViewDev *views, int slot, double *buf, int w, int h, int K)
{
	MrfLayout lay;
	carve(buf, w, h, K, 0.0, lay);
	const size_t n = (size_t)w*h;
	hipLaunchKernelGGL(mrf_depth_kernel, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, views, slot, K, lay.pz, lay.ans);
	return hipGetLastError();
}

} // namespace srh
