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
// LDS, and the message out of a band's last row goes to the next band's workgroup through device memory: write-through
// (sc1) stores, one progress word per band, sc1 loads by the polling wave (the producer -> consumer form of the HIP
// guide's inter-workgroup hand-off).  Bands therefore run as a pipeline, each ~6 chunks of columns behind the one
// above, and a sweep takes about W + 26 * H/16 steps instead of W * H.
//
// Within a pixel the label vectors are exchanged with DPP row broadcasts (v_mov_b64_dpp row_newbcast: no LDS).
// Per-step inputs (data costs, stored messages, peak depths) are fetched a chunk of 4 steps ahead into registers.
// The smoothness cost 2|z1 - z2| / (z1 + z2) is evaluated where it is used (an IEEE division per label pair): the
// table the CPU library caches would be 1.6 KB per pixel.
#include "srh_internal.hpp"

namespace srh {

namespace {

constexpr int MRF_ROWS = 16;            // rows per band = pixel groups per workgroup
constexpr int MRF_LANES = 16;           // lanes per pixel = label slots (K + 1 <= 16)
constexpr int MRF_CH = 4;               // steps per prefetched chunk
constexpr unsigned MRF_SPIN_LIMIT = 1u << 21;
constexpr size_t MRF_LDS_BYTES = 96*1024;   // more than half of a CU's LDS: one band per CU (hand-off form, and 1 wave per SIMD)

// sync block (unsigned words, zeroed before every pass): [0] ticket, [4 + b] progress of band b
// status block (zeroed once per run): [0] abort, [1] first band that gave up + 1, [2] pass it gave up in + 1
constexpr int SY_TICKET = 0, SY_PROGRESS = 4;
constexpr int ST_ABORT = 0, ST_WHO = 1, ST_PASS = 2;

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int I> __device__ __forceinline__ double bc16(double v) {
	return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + I, 0xf, 0xf, false);      // row_newbcast:I
}

__device__ __forceinline__ void st_sc1(double *p, double v) {
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), RLX_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
	return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), RLX_AGENT));
}

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
	double D[MRF_CH], oH[MRF_CH], oV[MRF_CH], zs[MRF_CH + 1], zV[MRF_CH], T[MRF_CH];
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
	double *carry;           // [n] solve pass: the chosen label's depth (NaN: unknown), for the band below
	unsigned *sync, *status;
};

// MODE 0: forward sweep, 1: backward sweep (logical coordinates mirrored), 2: labels read off (forward order)
// LT: the label count K + 1 when known at compile time (10 for the reference's K = 9: straight-line label loops), 0: any
template <int MODE, int LT>
__global__ __launch_bounds__(256, 1) void mrf_pass_kernel(const MrfPassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	double (*down)[MRF_ROWS][MRF_LANES] = reinterpret_cast<double (*)[MRF_ROWS][MRF_LANES]>(smem);   // [2][row][label]
	int *s_ctl = reinterpret_cast<int *>(smem + 2*MRF_ROWS*MRF_LANES*sizeof(double));                  // [0] band, [1] abort

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
	unsigned known = 0;                                              // wave 0: columns the band above is known to have finished

	// physical pixel index of logical column u in this lane's row
	auto pix = [&](int u) -> long { return (long)y*W + (MODE == 1 ? W - 1 - u : u); };

	// Wave 0 learns how far the band above has come.  The progress word is read one chunk AHEAD of its use (`early`,
	// an sc1 load whose answer is only looked at when the next chunk starts), so that in the steady state -- this band
	// a chunk or two further behind than it strictly has to be -- no step ever waits for the ~2 us round trip of a poll;
	// only when that early answer is not enough does lane 0 spin, which also puts the band that much further behind.
	unsigned early = 0;
	auto wait_above = [&](unsigned need) -> bool {                    // wave 0 only; uniform result
		if (b == 0) return true;
		known = max(known, early);
		int ok = 1;
		if (known < need) {
			if (tid == 0) {
				unsigned spins = 0;
				for (;;) {
					known = __hip_atomic_load(&a.sync[SY_PROGRESS + b - 1], RLX_AGENT);
					if (known >= need) break;
					if (++spins > MRF_SPIN_LIMIT || __hip_atomic_load(&a.status[ST_ABORT], RLX_AGENT)) { ok = 0; break; }
					__builtin_amdgcn_s_sleep(8);
				}
				if (!ok) {
					__hip_atomic_store(&a.status[ST_ABORT], 1u, RLX_AGENT);
					if (atomicCAS(&a.status[ST_WHO], 0u, (unsigned)b + 1u) == 0u) a.status[ST_PASS] = MODE + 1;
					s_ctl[1] = 1;
				}
			}
			known = __shfl(known, 0);
			ok = __shfl(ok, 0);
		}
		early = __hip_atomic_load(&a.sync[SY_PROGRESS + b - 1], RLX_AGENT);   // for the next call; not waited for here
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // keeps the sc1 loads below the poll
		return ok != 0;
	};

	auto load_chunk = [&](int c, Chunk &B, bool top_ready) {
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
			B.T[j] = 0.0;
			if (r == 0 && b > 0 && ok && top_ready) {
				// what the band above handed down for this column: its edge towards us
				if (MODE == 2)      B.T[j] = ld_sc1(&a.carry[n - W]);
				else if (MODE == 0) B.T[j] = ld_sc1(&a.Mv[(n - W)*16 + kd]);
				else                B.T[j] = ld_sc1(&a.Mv[n*16 + kd]);
			}
		}
	};

	const int nsteps = W + MRF_ROWS - 1;
	const int nchunks = (nsteps + MRF_CH - 1)/MRF_CH;
	Chunk cur, nxt;
	{
		bool ok = true;
		if (wave == 0) ok = wait_above((unsigned)min(MRF_CH, W));
		load_chunk(0, nxt, ok);
	}
	double carryL = 0.0;                                             // message (or chosen depth) handed along the row

	for (int c = 0; c < nchunks; ++c) {
		cur = nxt;
		if (c + 1 < nchunks) {
			bool ok = true;
			if (wave == 0) ok = wait_above((unsigned)min((c + 2)*MRF_CH, W));
			load_chunk(c + 1, nxt, ok);
		}
#pragma unroll
		for (int j = 0; j < MRF_CH; ++j) {
			const int s = c*MRF_CH + j;
			const int u = s - r;
			const bool act = rowok && u >= 0 && u < W;
			const bool hasH = u < W - 1;
			const long n = pix(u);
			const double fromL = u > 0 ? carryL : 0.0;
			const double lds_top = down[(s + 1) & 1][(r + MRF_ROWS - 1) & (MRF_ROWS - 1)][MODE == 2 ? 0 : kd];
			const double fromT = r == 0 ? cur.T[j] : lds_top;       // v == 0: T is 0
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
				if (kd == 0) {
					down[s & 1][r][0] = cbest;
					if (act) {
						a.ans[n] = lab;
						if (r == rlast) st_sc1(&a.carry[n], cbest);
					}
				}
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
				if (act && kd < L) {
					if (hasH) a.Mh[(MODE == 1 ? n - 1 : n)*16 + kd] = mH;
					if (hasV) {
						double *dst = &a.Mv[(MODE == 1 ? n - W : n)*16 + kd];
						if (r == rlast) st_sc1(dst, mV); else *dst = mV;
					}
				}
			}
			__syncthreads();
		}
		// the band below may now read what the last row has finished: every sc1 store of this wave has landed first
		if (wave == (rlast >> 2)) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const int done = min(max(c*MRF_CH + MRF_CH - rlast, 0), W);
			if (done > 0 && tid == rlast*MRF_LANES)
				__hip_atomic_store(&a.sync[SY_PROGRESS + b], (unsigned)done, RLX_AGENT);
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
	lay.carry = buf + n*64;
	lay.partial = lay.carry + nr;
	lay.nparts = (int)((n + 255)/256);
	lay.energy = lay.partial + ((lay.nparts + 1) & ~1);                // [0] energy, [1] unused
	lay.status = reinterpret_cast<unsigned *>(lay.energy + 2);         // 4 words
	lay.ans = reinterpret_cast<int32_t *>(lay.energy + 4);
	lay.sync = reinterpret_cast<unsigned *>(lay.ans + nr);
	lay.nbands = (h + MRF_ROWS - 1)/MRF_ROWS;
	lay.sync_words = (size_t)((SY_PROGRESS + lay.nbands + 3) & ~3);
	lay.total_doubles = (size_t)(reinterpret_cast<double *>(lay.sync) - buf) + lay.sync_words/2;
	MrfPassArgs a;
	a.W = w; a.H = h; a.K = K; a.psi = psi; a.psi2 = 2*psi;
	a.pz = lay.pz; a.D = lay.D; a.Mh = lay.Mh; a.Mv = lay.Mv; a.ans = lay.ans; a.carry = lay.carry;
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
	hipLaunchKernelGGL(mrf_data_kernel, dim3((unsigned)((n*16 + 255)/256)), dim3(256), 0, st, (long)n, K, beta, lambda, phiu, peaks, lay.D, lay.pz);
	return hipGetLastError();
}

template <int MODE, int LT> static hipError_t launch_pass_lt(hipStream_t st, const MrfPassArgs &a, const MrfLayout &lay)
{
	hipError_t e;
	if ((e = hipMemsetAsync(lay.sync, 0, lay.sync_words*sizeof(unsigned), st)) != hipSuccess) return e;
	if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mrf_pass_kernel<MODE, LT>), hipFuncAttributeMaxDynamicSharedMemorySize,
	                             (int)MRF_LDS_BYTES)) != hipSuccess) return e;
	hipLaunchKernelGGL((mrf_pass_kernel<MODE, LT>), dim3((unsigned)lay.nbands), dim3(256), MRF_LDS_BYTES, st, a);
	return hipGetLastError();
}
template <int MODE> static hipError_t launch_pass(hipStream_t st, const MrfPassArgs &a, const MrfLayout &lay)
{
	return a.K == 9 ? launch_pass_lt<MODE, 10>(st, a, lay) : launch_pass_lt<MODE, 0>(st, a, lay);
}

// one optimize(1): forward sweep, backward sweep, labels read off
hipError_t launch_mrf_sweep(hipStream_t st, double *buf, int w, int h, int K, double psiu)
{
	MrfLayout lay;
	const MrfPassArgs a = carve(buf, w, h, K, psiu, lay);
	hipError_t e;
	if ((e = launch_pass<0>(st, a, lay)) != hipSuccess) return e;
	if ((e = launch_pass<1>(st, a, lay)) != hipSuccess) return e;
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
