// mfma_f64_rate.hip -- does v_mfma_f64_16x16x4_f64 buy the TwoView cost loops anything on one MI355X?
//
// The certified one-pass cost form is three dot products per (reference pixel, candidate column) over the
// 121 taps of the window: P = sum w r, Q = sum c r, U = sum d r^2.  For the 16 pixels of a row piece the other
// view's window at a column is the same, so the sums are a banded product [pixels x taps] x [taps x columns]:
// A = a pixel group's weights (registers), B = the other view's row ring in LDS read in a Toeplitz pattern
// (lane (k, n) reads column c0 + n + dx_k of ring row dy_k).  This file measures what the instruction sustains
//   (1) from registers, 1 and 2 waves per SIMD;
//   (2) with A and B fetched from LDS by ds_read_b64 per instruction, and in the kernel-like step
//       {1 ds_read_b64 of B, q = r*r, 3 MFMAs with A resident};
//   (3) beside a co-resident wave of v_fma_f64, of v_mul/v_add_f64, and of an integer / scalar / LDS "set-up" mix
//       (does the vector pipe keep issuing under an f64 MFMA, as it does under bf16?);
// and (4) what it computes: the lane layout and whether D equals the chain fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,c))))
//       bit for bit (what the error bound of DESIGN section 2b has to be derived for).
// A workgroup = 8 waves = 1 per CU (LDS request); waves w and w+4 share a SIMD (checked through HW_REG_HW_ID).
// The partner role runs until the measured role raises a flag in LDS, so both are measured under full overlap.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

enum Role { IDLE = 0, MFMA_REG = 1, MFMA_LDS2 = 2, MFMA_STEP = 3, VFMA = 4, VMULADD = 5, SETUP = 6, MFMA_STEP_EPI = 7 };

struct Stamp { unsigned long long cycles, real, units; unsigned hwid, role; };

constexpr int RING_PITCH = 352;            // doubles per ring row (320-column chunk + 32), as the strip kernel's rows
constexpr int RING_ROWS = 12;

// One "unit" of every role is sized so that the rates are easy to state: MFMA roles count MFMAs, VALU roles count
// wave-instructions.
template <int ROLE>
__device__ __forceinline__ unsigned long long run_role(const double *__restrict__ seed, double *ring, volatile int *flag_mine,
                                                       volatile int *flag_partner, bool measured, int iters, double &sink) {
	const int lane = threadIdx.x & 63;
	unsigned long long units = 0;
	if constexpr (ROLE == MFMA_REG) {
		v4d acc[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) acc[j] = (v4d){0, 0, 0, 0};
		double a = seed[lane], b = seed[lane + 64];
		for (int it = 0; measured ? it < iters : *flag_partner == 0; ++it) {
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
			units += 32;
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) sink += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
	} else if constexpr (ROLE == MFMA_LDS2) {
		// A and B both from LDS for every MFMA: lane (k = lane>>4, n = lane&15) reads ring[row][c0 + n + k + 4s]
		v4d acc[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) acc[j] = (v4d){0, 0, 0, 0};
		const double *pb = ring + (lane & 15) + (lane >> 4);
		const double *pa = ring + 6*RING_PITCH + lane;
		for (int it = 0; measured ? it < iters : *flag_partner == 0; ++it) {
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const double b = pb[(j >> 1)*RING_PITCH + (j & 1)*4 + u*16];
					const double a = pa[j*64 + u*8];
					acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
				}
			units += 32;
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) sink += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
	} else if constexpr (ROLE == MFMA_STEP || ROLE == MFMA_STEP_EPI) {
		// the kernel-like step: 33 K-steps (11 window rows x 12 padded taps / 4), A resident (3 x 33 doubles per lane),
		// per K-step one ds_read_b64 of B, max(r, 0) (NaN pad guard), q = r*r, three MFMAs; one N-tile = 16 columns.
		double aw[33], ac[33], ad[33];
#pragma unroll
		for (int s = 0; s < 33; ++s) { aw[s] = seed[(lane + s*7) & 2047]; ac[s] = seed[(lane + s*13 + 5) & 2047]; ad[s] = aw[s]*aw[s]; }
		const double *pb = ring + (lane & 15) + (lane >> 4);
		for (int it = 0; measured ? it < iters : *flag_partner == 0; ++it) {
			v4d P = {0, 0, 0, 0}, Q = {0, 0, 0, 0}, U = {0, 0, 0, 0};
			const double *pbt = pb + (it & 15)*16;
#pragma unroll
			for (int s = 0; s < 33; ++s) {
				double r = pbt[(s/3)*RING_PITCH + (s % 3)*4];
				r = __builtin_fmax(r, 0.0);
				const double q = r*r;
				P = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[s], r, P, 0, 0, 0);
				Q = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[s], r, Q, 0, 0, 0);
				U = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[s], q, U, 0, 0, 0);
			}
			if constexpr (ROLE == MFMA_STEP_EPI) {
				// the epilogue's arithmetic per output entry (4 per lane): m = P/tw, sum3, sum1, Q3, cost = 255(1 - |s1|/sqrt(s2 s3))
				const double tw = seed[lane + 128] + 50.0, s2 = seed[lane + 192] + 900.0, SA = seed[lane + 256];
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const double m = P[i]/tw;
					const double s3 = __builtin_fma(-m, __builtin_fma(-121.0, m, 2.0*P[i]), U[i]);
					const double s1 = __builtin_fma(-m, SA, Q[i]);
					const double q3 = __builtin_fma(121.0*m, m, __builtin_fma(2.0*m, P[i], U[i]));
					const double c = 255.0*(1.0 - __builtin_fabs(s1)/__builtin_sqrt(s2*s3));
					sink += (s3 >= 1e-3 && q3 <= 994.0*s3) ? c : 0.0;
				}
			} else {
				sink += P[0] + Q[1] + U[2] + P[3];
			}
			units += 99;
		}
	} else if constexpr (ROLE == VFMA) {
		double acc[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) acc[j] = seed[lane + j*64];
		const double a = seed[lane + 512], b = seed[lane + 576];
		for (int it = 0; measured ? it < iters : *flag_partner == 0; ++it) {
#pragma unroll
			for (int u = 0; u < 8; ++u)
#pragma unroll
				for (int j = 0; j < 8; ++j) acc[j] = __builtin_fma(acc[j], a, b);
			units += 64;
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) sink += acc[j];
	} else if constexpr (ROLE == VMULADD) {
		double acc[8], m[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) { acc[j] = seed[lane + j*64]; m[j] = seed[lane + j*64 + 977]; }
		for (int it = 0; measured ? it < iters : *flag_partner == 0; ++it) {
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				double p[8];
#pragma unroll
				for (int j = 0; j < 8; ++j) p[j] = m[j]*acc[(j + 1) & 7];
#pragma unroll
				for (int j = 0; j < 8; ++j) acc[j] += p[j];
#pragma unroll
				for (int j = 0; j < 8; ++j) m[j] = -m[j];
			}
			units += 64;
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) sink += acc[j];
	} else if constexpr (ROLE == SETUP) {
		// the strip kernel's set-up flavour: 32-bit integer vector work, compares + selects, LDS reads, scalar work
		unsigned x = lane*2654435761u, y = lane + 17, z = 0;
		const int *li = (const int *)ring;
		for (int it = 0; measured ? it < iters : *flag_partner == 0; ++it) {
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				x = x*1664525u + 1013904223u;             // v_mul_lo + v_add
				y = (y ^ (x >> 7)) + u;                   // v_lshr, v_xor, v_add
				const int v = li[(x >> 20) & 1023];       // ds_read_b32
				z += (v > (int)y) ? x : y;                // v_cmp + v_cndmask + v_add
				z = __builtin_amdgcn_readfirstlane(z) + (z & 0xffff);   // scalar round trip
			}
			units += 64;                                  // nominal: ~8 vector/scalar/LDS instructions x 8
		}
		sink += (double)z + (double)x + (double)y;
	}
	return units;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(double *out, Stamp *stamps, const double *__restrict__ seed, int iters, int measuredB) {
	extern __shared__ double ring[];                     // ring rows + an A area; the size request also keeps 1 WG per CU
	__shared__ int flags[8];
	const int wave = threadIdx.x >> 6;
	for (int i = threadIdx.x; i < RING_ROWS*RING_PITCH + 2048; i += 512) ring[i] = seed[i & 2047];
	if (threadIdx.x < 8) flags[threadIdx.x] = 0;
	__syncthreads();
	const bool isA = wave < 4;
	const bool measured = isA ? !measuredB : measuredB;   // the measured role runs `iters`, the other until the flag
	volatile int *mine = &flags[wave], *partner = &flags[wave ^ 4];
	double sink = 0;
	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long units;
	if (isA) units = run_role<RA>(seed, ring, mine, partner, measured || RB == IDLE, iters, sink);
	else     units = run_role<RB>(seed, ring, mine, partner, measured || RA == IDLE, iters, sink);
	const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	if ((threadIdx.x & 63) == 0) *mine = 1;
	out[blockIdx.x*512 + threadIdx.x] = sink;
	if ((threadIdx.x & 63) == 0) {
		Stamp s; s.cycles = c1 - c0; s.real = r1 - r0; s.units = units;
		s.hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
		s.role = isA ? RA : RB;
		stamps[blockIdx.x*8 + wave] = s;
	}
}

static const char *role_name(int r) {
	switch (r) {
		case IDLE: return "idle"; case MFMA_REG: return "mfma(reg)"; case MFMA_LDS2: return "mfma(A,B from LDS)";
		case MFMA_STEP: return "step{ds_read B, max, r*r, 3 mfma}"; case MFMA_STEP_EPI: return "step + epilogue";
		case VFMA: return "v_fma_f64"; case VMULADD: return "v_mul+v_add_f64"; case SETUP: return "set-up mix";
	}
	return "?";
}

// per unit: MFMA roles 2048 flop; VFMA 64 lanes x 2; VMULADD 64 lanes x 1 (a unit is one wave-instruction)
static double unit_flops(int r) {
	if (r == MFMA_REG || r == MFMA_LDS2 || r == MFMA_STEP || r == MFMA_STEP_EPI) return 2048.0;
	if (r == VFMA) return 128.0;
	if (r == VMULADD) return 64.0;
	return 0.0;
}

template <int RA, int RB>
static void run(const double *seed, double *out, Stamp *st, int iters, bool measuredB, bool simd_report) {
	const int blocks = 256;
	const size_t lds = (RING_ROWS*RING_PITCH + 2048)*sizeof(double) + 48*1024;   // > half of 160 KB: one workgroup per CU
	hipFuncSetAttribute((const void *)k<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<RA, RB>), dim3(blocks), dim3(512), lds, 0, out, st, seed, iters, (int)measuredB);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL((k<RA, RB>), dim3(blocks), dim3(512), lds, 0, out, st, seed, iters, (int)measuredB);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<Stamp> h(blocks*8);
	hipMemcpy(h.data(), st, sizeof(Stamp)*blocks*8, hipMemcpyDeviceToHost);
	if (simd_report) {
		// HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13] (gfx9 layout)
		int same = 0, tot = 0;
		for (int b = 0; b < blocks; ++b)
			for (int w = 0; w < 4; ++w) { tot++; same += ((h[b*8 + w].hwid >> 4) & 3) == ((h[b*8 + w + 4].hwid >> 4) & 3); }
		printf("# SIMD pairing: waves w and w+4 of a 512-thread workgroup share a SIMD in %d of %d pairs\n", same, tot);
	}
	for (int half = 0; half < 2; ++half) {
		const int role = half ? RB : RA;
		if (role == IDLE) continue;
		std::vector<double> cyc_per_unit, mhz; double units_total = 0, t_us_max = 0;
		for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) {
			const Stamp &s = h[b*8 + half*4 + w];
			if (!s.units) continue;
			cyc_per_unit.push_back((double)s.cycles/(double)s.units);
			mhz.push_back((double)s.cycles/(double)s.real*100.0);
			units_total += (double)s.units; t_us_max = std::max(t_us_max, (double)s.real/100.0);
		}
		std::sort(cyc_per_unit.begin(), cyc_per_unit.end()); std::sort(mhz.begin(), mhz.end());
		const double cpu_med = cyc_per_unit[cyc_per_unit.size()/2], clk = mhz[mhz.size()/2];
		// chip rate from the median wave: 1024 SIMDs, one such wave per SIMD
		const double tflops = unit_flops(role) ? 1024.0*unit_flops(role)/cpu_med*clk*1e6/1e12 : 0.0;
		printf("  %-36s %s: %8.2f cycles per %s (median of %zu waves), clock %.0f MHz", role_name(role),
		       ((half == 1) == measuredB) ? "[fixed work]" : "[until flag]", cpu_med,
		       unit_flops(role) == 2048.0 ? "MFMA" : "wave-instr", cyc_per_unit.size(), clk);
		if (tflops > 0) printf(", %.1f TFLOP/s chip-wide at one such wave per SIMD", tflops);
		printf("\n");
	}
	printf("  kernel wall %.3f ms\n", ms);
}

// ---- numerics: lane layout and the accumulation order of one instruction ----
__global__ void probe(const double *a, const double *b, const double *c, double *d) {
	const int lane = threadIdx.x;
	v4d acc = {c[lane*4 + 0], c[lane*4 + 1], c[lane*4 + 2], c[lane*4 + 3]};
	acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], acc, 0, 0, 0);
	for (int i = 0; i < 4; ++i) d[lane*4 + i] = acc[i];
}

static void numerics() {
	std::vector<double> a(64), b(64), c(256), d(256);
	unsigned long long s = 0xF64F64;
	auto rnd = [&]() { s = s*6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11)/9007199254740992.0; };
	double *da, *db, *dc, *dd;
	hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 2048); hipMalloc(&dd, 2048);
	long n_fwd = 0, n_rev = 0, n_tot = 0, n_pair = 0; double max_rel_fwd = 0;
	for (int trial = 0; trial < 200; ++trial) {
		const double scale = trial < 100 ? 255.0 : 1.0;
		for (auto &v : a) v = (rnd() - (trial & 1 ? 0.5 : 0.0))*scale;
		for (auto &v : b) v = (rnd() - (trial & 2 ? 0.5 : 0.0))*scale;
		for (auto &v : c) v = (rnd() - 0.5)*scale*scale*(trial % 5 == 0 ? 0.0 : 1.0);
		hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
		hipMemcpy(dc, c.data(), 2048, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
		hipMemcpy(d.data(), dd, 2048, hipMemcpyDeviceToHost);
		// layout under test: A lane l = A[m = l & 15][k = l >> 4]; B lane l = B[k = l >> 4][n = l & 15];
		// D register i of lane l = D[row = (l >> 4) + 4 i][col = l & 15]; C likewise.
		for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
			const int row = (l >> 4) + 4*i, col = l & 15;
			double f = c[l*4 + i], r = c[l*4 + i];
			for (int kk = 0; kk < 4; ++kk) f = std::fma(a[kk*16 + row], b[kk*16 + col], f);
			for (int kk = 3; kk >= 0; --kk) r = std::fma(a[kk*16 + row], b[kk*16 + col], r);
			const double p01 = std::fma(a[16 + row], b[16 + col], a[row]*b[col]);
			const double p23 = std::fma(a[48 + row], b[48 + col], a[32 + row]*b[32 + col]);
			const double pr = (p01 + p23) + c[l*4 + i];
			const double got = d[l*4 + i];
			n_tot++; n_fwd += !std::memcmp(&got, &f, 8); n_rev += !std::memcmp(&got, &r, 8); n_pair += !std::memcmp(&got, &pr, 8);
			if (f != 0) max_rel_fwd = std::max(max_rel_fwd, std::fabs(got - f)/std::fabs(f));
		}
	}
	printf("# numerics of one v_mfma_f64_16x16x4_f64 (200 random operand sets, layout A[l&15][l>>4], B[l>>4][l&15], D reg i -> row (l>>4)+4i, col l&15):\n");
	printf("  == fma chain k = 0,1,2,3 onto c : %ld of %ld bit-identical (max relative difference %.3g)\n", n_fwd, n_tot, max_rel_fwd);
	printf("  == fma chain k = 3,2,1,0 onto c : %ld of %ld\n", n_rev, n_tot);
	printf("  == pairwise (p01 + p23) + c      : %ld of %ld\n", n_pair, n_tot);
}

int main() {
	std::vector<double> h(2048);
	unsigned long long s = 0x5EED;
	for (auto &v : h) { s = s*6364136223846793005ull + 1442695040888963407ull; v = 0.5 + (double)(s >> 11)/9007199254740992.0*1e-3 - 5e-4; }
	double *out, *seed; Stamp *st;
	hipMalloc(&out, sizeof(double)*512*256); hipMalloc(&seed, sizeof(double)*2048); hipMalloc(&st, sizeof(Stamp)*8*256);
	hipMemcpy(seed, h.data(), sizeof(double)*2048, hipMemcpyHostToDevice);
	numerics();
	const int it = 4000;
	printf("# (1) registers only\n");
	printf("one wave per SIMD:\n");            run<MFMA_REG, IDLE>(seed, out, st, it, false, true);
	printf("two waves per SIMD, both MFMA:\n"); run<MFMA_REG, MFMA_REG>(seed, out, st, it, false, false);
	printf("# (2) fed from LDS\n");
	printf("A and B by ds_read_b64 per MFMA, one wave per SIMD:\n");  run<MFMA_LDS2, IDLE>(seed, out, st, it, false, false);
	printf("A and B by ds_read_b64 per MFMA, two waves per SIMD:\n"); run<MFMA_LDS2, MFMA_LDS2>(seed, out, st, it, false, false);
	printf("kernel-like step, one wave per SIMD:\n");  run<MFMA_STEP, IDLE>(seed, out, st, it/4, false, false);
	printf("kernel-like step, two waves per SIMD:\n"); run<MFMA_STEP, MFMA_STEP>(seed, out, st, it/4, false, false);
	printf("kernel-like step + per-entry epilogue (division, sqrt, certification), one wave per SIMD:\n");
	run<MFMA_STEP_EPI, IDLE>(seed, out, st, it/4, false, false);
	printf("kernel-like step + epilogue, two waves per SIMD:\n"); run<MFMA_STEP_EPI, MFMA_STEP_EPI>(seed, out, st, it/4, false, false);
	printf("# (3) references: the vector roles alone (one wave per SIMD, then two)\n");
	run<VFMA, IDLE>(seed, out, st, it*4, false, false);    run<VFMA, VFMA>(seed, out, st, it*4, false, false);
	run<VMULADD, IDLE>(seed, out, st, it*4, false, false); run<SETUP, IDLE>(seed, out, st, it*4, false, false);
	printf("# (3) MFMA beside a co-resident vector wave on the same SIMD (MFMA wave: fixed work; partner: until the flag)\n");
	printf("mfma(reg) + v_fma_f64:\n");        run<MFMA_REG, VFMA>(seed, out, st, it, false, false);
	printf("mfma(reg) + v_mul/v_add_f64:\n");  run<MFMA_REG, VMULADD>(seed, out, st, it, false, false);
	printf("mfma(reg) + set-up mix:\n");       run<MFMA_REG, SETUP>(seed, out, st, it, false, false);
	printf("step + v_fma_f64:\n");             run<MFMA_STEP, VFMA>(seed, out, st, it/4, false, false);
	printf("step + set-up mix:\n");            run<MFMA_STEP, SETUP>(seed, out, st, it/4, false, false);
	printf("# (3') the other way round (vector wave: fixed work; MFMA partner until the flag)\n");
	printf("v_fma_f64 + mfma(reg):\n");        run<MFMA_REG, VFMA>(seed, out, st, it*4, true, false);
	return 0;
}
