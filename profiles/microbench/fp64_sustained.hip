// fp64_sustained.hip -- which clock, and which separate-multiply/add FP64 rate, does one MI355X hold when
// the FP64 pipe is kept busy for SECONDS on non-trivial operands (the C3 step keeps it busy for ~40 ms per
// step, back to back)?  MI355X_MICROARCH.md "DVFS give-back": the chip lowers its clock under load, short
// runs on trivial data read high.  Method of that guide, item 6: in-kernel clock =
// d(s_memtime) / d(s_memrealtime) * 100 MHz, stamped once around the loop, median over workgroups, after
// >= 2 s of back-to-back launches.  The stamps go to a buffer nothing else reads.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_sustained.hip -o fp64_sustained
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(double *out, unsigned long long *stamps, const double *seed, int iters) {
	double acc[8], m[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) { acc[j] = seed[(threadIdx.x*8 + j) & 2047]; m[j] = seed[(threadIdx.x*8 + j + 977) & 2047]; }
	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			double p[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) p[j] = m[j]*acc[(j + 1) & 7];     // 8 independent v_mul_f64
#pragma unroll
			for (int j = 0; j < 8; ++j) acc[j] += p[j];                  // 8 independent v_add_f64
#pragma unroll
			for (int j = 0; j < 8; ++j) m[j] = -m[j];                    // sign flip folds into the next multiply's modifier
		}
	}
	const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	double s = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) s += acc[j];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) { stamps[2*blockIdx.x] = c1 - c0; stamps[2*blockIdx.x + 1] = r1 - r0; }
}

int main() {
	const int iters = 20000;
	std::vector<double> h(2048);
	unsigned long long s = 0x5EED;
	for (auto &v : h) { s = s*6364136223846793005ull + 1442695040888963407ull; v = 0.5 + (double)(s >> 11)/9007199254740992.0*1e-3 - 5e-4; }
	double *d, *seed; unsigned long long *st;
	hipMalloc(&d, sizeof(double)*256*256*16); hipMalloc(&seed, sizeof(double)*2048); hipMalloc(&st, 16*256*16);
	hipMemcpy(seed, h.data(), sizeof(double)*2048, hipMemcpyHostToDevice);
	for (int wpc : {1, 2, 4}) {
		const int blocks = 256*wpc;
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		// >= 2 s of back-to-back launches first, then the measured launch
		double warm_ms = 0; int launches = 0;
		while (warm_ms < 2500) {
			hipEventRecord(e0);
			for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, st, seed, iters);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1); warm_ms += ms; launches += 20;
		}
		hipEventRecord(e0);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, st, seed, iters);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> hs(2*blocks);
		hipMemcpy(hs.data(), st, sizeof(unsigned long long)*2*blocks, hipMemcpyDeviceToHost);
		std::vector<double> mhz(blocks);
		for (int b = 0; b < blocks; ++b) mhz[b] = (double)hs[2*b]/(double)hs[2*b + 1]*100.0;
		std::sort(mhz.begin(), mhz.end());
		const double lane_instr = (double)blocks*256*iters*128;          // 64 mul + 64 add per iteration
		printf("waves/SIMD %d after %.1f s (%d launches) of load: launch %.2f ms = %.2f T lane-instr/s (mul+add separate); "
		       "in-kernel clock median %.0f MHz (min %.0f, max %.0f); ceiling at that clock 256*4*16*clk = %.2f T\n",
		       wpc, warm_ms/1e3, launches, ms, lane_instr/ms/1e9, mhz[blocks/2], mhz[0], mhz[blocks - 1],
		       256.0*4*16*mhz[blocks/2]*1e6/1e12);
	}
	return 0;
}
