// fp64_rate.hip -- what vector FP64 rate does one MI355X sustain for (a) fused multiply-add,
// (b) separate multiply + add (the form the stereo kernels must use: contraction off)?
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
	double acc[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) acc[j] = threadIdx.x*1e-9 + j;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				if (MODE == 0) acc[j] = __builtin_fma(acc[j], a, b);          // 1 instr, 2 flop
				else if (MODE == 1) acc[j] = acc[j]*a + b;                    // mul + add (2 instr)
				else { const double p = a*acc[(j+1)&7]; acc[j] += p; }         // independent mul, then add
			}
		}
	}
	double s = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) s += acc[j];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int blocks, int iters, double *d) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 10);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main() {
	const int iters = 20000;
	double *d; hipMalloc(&d, sizeof(double)*256*256*16);
	for (int wpc : {1, 2, 4, 8}) {           // blocks of 256 threads per CU -> waves per SIMD
		const int blocks = 256*wpc;
		const double n_instr_lane = (double)blocks*256*iters*64;      // per mode: 64 acc updates per iter
		double ms0 = run<0>(blocks, iters, d), ms1 = run<1>(blocks, iters, d), ms2 = run<2>(blocks, iters, d);
		printf("waves/SIMD %d: fma %.2f ms = %.1f TFLOP/s (%.2f Tinstr-lane/s) | mul+add dependent %.2f ms = %.2f Tinstr-lane/s | mul,add interleaved %.2f ms = %.2f Tinstr-lane/s\n",
		       wpc, ms0, 2*n_instr_lane/ms0/1e9, n_instr_lane/ms0/1e9, ms1, 2*n_instr_lane/ms1/1e9, ms2, 2*n_instr_lane/ms2/1e9);
	}
	return 0;
}
