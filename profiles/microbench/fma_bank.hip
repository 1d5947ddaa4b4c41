// fma_bank.hip -- why does a three-operand v_fma_f64 issue every ~5 cycles at two waves per SIMD where v_mul_f64 / v_add_f64
// (two register operands) issue every ~4.2 (fp64_rate_mi355x.txt)?  If the fifth cycle is a VGPR bank conflict between the three
// 64-bit source operands (4 banks, register number mod 4; a 64-bit operand is an even-aligned pair and takes banks {0,1} or
// {2,3}: two of three operands always share a pair), the rate must depend on WHICH operands share -- and a hand-allocated loop
// could choose.  Eight accumulators, the instruction pattern of the strip kernel's block loops:
//     v_fma_f64 acc_j, w, r_j, acc_j      (acc = dst = src2, w the same for 8 instructions, r_j varies)
// with the bank pairs of (acc_j, w, r_j) chosen by hand.  Registers are named explicitly in one asm statement.
// build: hipcc --offload-arch=gfx950 -O3 fma_bank.hip -o fma_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

// pattern p: bank pair (0 -> v[4k:4k+1], 1 -> v[4k+2:4k+3]) of acc_j, w, r_j
//   0: acc 0, w 0, r 0 (all three share)     1: acc 0, w 0, r 1     2: acc 0, w 1, r 0     3: acc 0, w 1, r 1
//   4: acc alternates 0/1 with j, w 0, r alternates 0/1 (what an allocator that packs arrays gives)
//   5: acc alternates, w 0, r alternates the other way
//   6: two-operand reference: v_mul_f64 t_j, w, r_j ; (no accumulate)   7: v_add_f64 acc_j, acc_j, r_j
#define ACC(j, b) "v[" #j "*4+" #b "*2+32:" #j "*4+" #b "*2+33]"

template <int P>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
	// registers: w at v[100+2b : 101+2b] (b = bank pair), r_j at v[64 + 4j + 2b ...], acc_j at v[32 + 4j + 2b ...]
	asm volatile(
		"v_mov_b32 v100, 0\n v_mov_b32 v101, 0x3ff00000\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0x3ff00000\n"
		::: "v100", "v101", "v102", "v103");
#define INIT(j) asm volatile("v_mov_b32 v%c0, 0\n v_mov_b32 v%c1, 0x3ff00000\n v_mov_b32 v%c2, 0\n v_mov_b32 v%c3, 0x3ff00000\n" \
	"v_mov_b32 v%c4, 0\n v_mov_b32 v%c5, 0x3ff00000\n v_mov_b32 v%c6, 0\n v_mov_b32 v%c7, 0x3ff00000\n" \
	:: "n"(32 + 4*j), "n"(33 + 4*j), "n"(34 + 4*j), "n"(35 + 4*j), "n"(64 + 4*j), "n"(65 + 4*j), "n"(66 + 4*j), "n"(67 + 4*j));
	INIT(0) INIT(1) INIT(2) INIT(3) INIT(4) INIT(5) INIT(6) INIT(7)
	for (int it = 0; it < iters; ++it) {
		// 8 groups of 8 instructions
#define F(ab, wb, rb, j) "v_fma_f64 v[%c[a" #j "]+" #ab ":%c[a" #j "]+" #ab "+1], v[%c[w]+" #wb ":%c[w]+" #wb "+1], v[%c[r" #j "]+" #rb ":%c[r" #j "]+" #rb "+1], v[%c[a" #j "]+" #ab ":%c[a" #j "]+" #ab "+1]\n"
#define M(wb, rb, j) "v_mul_f64 v[%c[a" #j "]:%c[a" #j "]+1], v[%c[w]+" #wb ":%c[w]+" #wb "+1], v[%c[r" #j "]+" #rb ":%c[r" #j "]+" #rb "+1]\n"
#define AD(rb, j) "v_add_f64 v[%c[a" #j "]:%c[a" #j "]+1], v[%c[a" #j "]:%c[a" #j "]+1], v[%c[r" #j "]+" #rb ":%c[r" #j "]+" #rb "+1]\n"
#define OPS : : [w] "n"(100), [a0] "n"(32), [a1] "n"(36), [a2] "n"(40), [a3] "n"(44), [a4] "n"(48), [a5] "n"(52), [a6] "n"(56), [a7] "n"(60), \
	[r0] "n"(64), [r1] "n"(68), [r2] "n"(72), [r3] "n"(76), [r4] "n"(80), [r5] "n"(84), [r6] "n"(88), [r7] "n"(92)
#define G8(X) X X X X X X X X
		if (P == 0) asm volatile(G8(F(0,0,0,0) F(0,0,0,1) F(0,0,0,2) F(0,0,0,3) F(0,0,0,4) F(0,0,0,5) F(0,0,0,6) F(0,0,0,7)) OPS);
		if (P == 1) asm volatile(G8(F(0,0,2,0) F(0,0,2,1) F(0,0,2,2) F(0,0,2,3) F(0,0,2,4) F(0,0,2,5) F(0,0,2,6) F(0,0,2,7)) OPS);
		if (P == 2) asm volatile(G8(F(0,2,0,0) F(0,2,0,1) F(0,2,0,2) F(0,2,0,3) F(0,2,0,4) F(0,2,0,5) F(0,2,0,6) F(0,2,0,7)) OPS);
		if (P == 3) asm volatile(G8(F(0,2,2,0) F(0,2,2,1) F(0,2,2,2) F(0,2,2,3) F(0,2,2,4) F(0,2,2,5) F(0,2,2,6) F(0,2,2,7)) OPS);
		if (P == 4) asm volatile(G8(F(0,0,0,0) F(2,0,2,1) F(0,0,0,2) F(2,0,2,3) F(0,0,0,4) F(2,0,2,5) F(0,0,0,6) F(2,0,2,7)) OPS);
		if (P == 5) asm volatile(G8(F(0,0,2,0) F(2,0,0,1) F(0,0,2,2) F(2,0,0,3) F(0,0,2,4) F(2,0,0,5) F(0,0,2,6) F(2,0,0,7)) OPS);
		if (P == 6) asm volatile(G8(M(0,2,0) M(0,2,1) M(0,2,2) M(0,2,3) M(0,2,4) M(0,2,5) M(0,2,6) M(0,2,7)) OPS);
		if (P == 7) asm volatile(G8(AD(2,0) AD(2,1) AD(2,2) AD(2,3) AD(2,4) AD(2,5) AD(2,6) AD(2,7)) OPS);
	}
	double s;
	asm volatile("v_add_f64 %0, v[32:33], v[34:35]" : "=v"(s));
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <int P> static double run(int blocks, int iters, double *d) {
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, d, 10);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, d, iters);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main() {
	const int iters = 20000;
	double *d; (void)hipMalloc(&d, sizeof(double)*256*256*16);
	const char *names[8] = {"fma acc0 w0 r0 (all share a bank pair)", "fma acc0 w0 r1", "fma acc0 w1 r0", "fma acc0 w1 r1",
	                        "fma acc alt, w0, r alt same as acc", "fma acc alt, w0, r alt opposite", "v_mul_f64 t, w0, r1", "v_add_f64 acc0, acc0, r1"};
	for (int wpc : {1, 2, 4}) {
		const int blocks = 256*wpc;
		const double n = (double)blocks*4*iters*64;      // wave-instructions
		double ms[8] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d), run<3>(blocks, iters, d),
		                run<4>(blocks, iters, d), run<5>(blocks, iters, d), run<6>(blocks, iters, d), run<7>(blocks, iters, d)};
		for (int p = 0; p < 8; ++p)
			printf("waves/SIMD %d  %-44s %7.3f ms  %5.2f cycles per instruction per SIMD at 2.4 GHz  (%.2f T lane-instr/s)\n",
			       wpc, names[p], ms[p], ms[p]*1e-3*2.4e9/(n/1024.0), n*64/ms[p]/1e9);
	}
	return 0;
}
