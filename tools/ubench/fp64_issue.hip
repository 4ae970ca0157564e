// fp64_issue.hip -- micro-benchmarks that price the instruction mix of the transit kernels on gfx950:
// cycles per wave-instruction (s_memtime around an unrolled body) of dependent / independent chains, at
// 1..4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_build/fp64_issue tools/ubench/fp64_issue.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

enum Kind { FMA_DEP = 0, FMA_IND4, FMA_IND2, RCP_DEP, RCP_IND4, RSQ_DEP, MUL_DEP, ADD_DEP, FMA32_DEP, FMA32_IND4, FMA_DEP_SMOV,
            FMA_DEP_READLANE, LDS_ACC, CNDMASK_DEP, FMA_DEP_VMOV, SQRT32_DEP, EXP32_DEP, FMA_IND8, KIND_N };
static const char* kNames[KIND_N] = {"v_fma_f64 dependent", "v_fma_f64 4 chains", "v_fma_f64 2 chains", "v_rcp_f64 dependent", "v_rcp_f64 4 chains",
                                     "v_rsq_f64 dependent", "v_mul_f64 dependent", "v_add_f64 dependent", "v_fma_f32 dependent", "v_fma_f32 4 chains",
                                     "v_fma_f64 dep + 2 s_mov each", "v_fma_f64 dep + v_readlane each", "LDS read-add-write (ds_read_b64, v_add_f64, ds_write_b64)",
                                     "v_cndmask_b32 x2 dependent", "v_fma_f64 dep + 2 v_mov each", "v_sqrt_f32 dependent", "v_exp_f32 dependent", "v_fma_f64 8 chains"};
static const int kPerRep[KIND_N] = {1, 4, 2, 1, 4, 1, 1, 1, 1, 4, 1, 1, 1, 2, 1, 1, 1, 8};

template <int KIND>
__global__ __launch_bounds__(256) void bench(double* out, uint64_t* cyc, double a, double b, int iters) {
  __shared__ double lds[256];
  double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1e-3, x2 = x0 + 2e-3, x3 = x0 + 3e-3, x4 = x0 + 4e-3, x5 = x0 + 5e-3, x6 = x0 + 6e-3, x7 = x0 + 7e-3;
  float f0 = (float)x0, f1 = (float)x1, f2 = (float)x2, f3 = (float)x3;
  const float fa = (float)a, fb = (float)b;
  lds[threadIdx.x] = x0;
  double* col = &lds[threadIdx.x];
  __syncthreads();
  const uint64_t t0 = now();
  for (int it = 0; it < iters; ++it) {
    if (KIND == FMA_DEP) { REP256(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) }
    if (KIND == FMA_IND4) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
    if (KIND == FMA_IND8) { REP64(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));) }
    if (KIND == FMA_IND2) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));) }
    if (KIND == RCP_DEP) { REP256(asm volatile("v_rcp_f64 %0, %0" : "+v"(x0));) }
    if (KIND == RCP_IND4) { REP64(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
    if (KIND == RSQ_DEP) { REP256(asm volatile("v_rsq_f64 %0, %0" : "+v"(x0));) }
    if (KIND == MUL_DEP) { REP256(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x0) : "v"(a));) }
    if (KIND == ADD_DEP) { REP256(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(b));) }
    if (KIND == FMA32_DEP) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(fa), "v"(fb));) }
    if (KIND == FMA32_IND4) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));) }
    if (KIND == FMA_DEP_SMOV) { REP256(asm volatile("s_mov_b32 s20, 0x3ff00000\n s_mov_b32 s21, 0x3fe00000\n v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b) : "s20", "s21");) }
    if (KIND == FMA_DEP_VMOV) { REP256(asm volatile("v_mov_b32 %3, 0x3ff00000\n v_mov_b32 %4, 0x3fe00000\n v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b), "v"(f1), "v"(f2));) }
    if (KIND == FMA_DEP_READLANE) { REP256(asm volatile("v_readlane_b32 s20, %3, 3\n v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b), "v"(f1) : "s20");) }
    if (KIND == LDS_ACC) { REP64(col[0] += x1; asm volatile("" ::: "memory");) }
    if (KIND == CNDMASK_DEP) { REP256(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(f0), "+v"(f1));) }
    if (KIND == SQRT32_DEP) { REP256(asm volatile("v_sqrt_f32 %0, %0" : "+v"(f0));) }
    if (KIND == EXP32_DEP) { REP256(asm volatile("v_exp_f32 %0, %0" : "+v"(f0));) }
  }
  const uint64_t t1 = now();
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1 + f2 + f3 + lds[threadIdx.x];
}

template <int KIND>
static void run(double* out, uint64_t* cyc, int waves_per_simd, int clk_khz) {
  const int iters = 20, n_cu = 256, blocks = n_cu * waves_per_simd;
  const int reps = (KIND == LDS_ACC) ? 64 : 256;
  hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double n_inst = (double)iters * reps * ((KIND == FMA_IND4 || KIND == RCP_IND4 || KIND == FMA32_IND4) ? 4 : (KIND == FMA_IND2 || KIND == CNDMASK_DEP) ? 2 : (KIND == FMA_IND8) ? 8 : 1);
  const double med = (double)h[h.size() / 2];
  // wall-clock view: instructions per SIMD / (ms * clock)
  const double wall_cyc_per_inst = ms * 1e-3 * clk_khz * 1e3 / (n_inst * waves_per_simd);
  printf("%-58s waves/SIMD=%d  counter ticks/inst/wave=%7.2f  wall cyc/inst/SIMD=%6.2f  (%.3f ms)\n", kNames[KIND], waves_per_simd, med / n_inst, wall_cyc_per_inst, ms);
}

int main() {
  double* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 8 * 256 * 8); hipMalloc(&cyc, 256 * 8 * 4 * 8);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("clock rate attribute: %d kHz\n", clk);
  for (int w : {1, 2, 3, 4}) {
    run<FMA_DEP>(out, cyc, w, clk); run<FMA_IND2>(out, cyc, w, clk); run<FMA_IND4>(out, cyc, w, clk); run<FMA_IND8>(out, cyc, w, clk);
    run<MUL_DEP>(out, cyc, w, clk); run<ADD_DEP>(out, cyc, w, clk);
    run<RCP_DEP>(out, cyc, w, clk); run<RCP_IND4>(out, cyc, w, clk); run<RSQ_DEP>(out, cyc, w, clk);
    run<FMA32_DEP>(out, cyc, w, clk); run<FMA32_IND4>(out, cyc, w, clk); run<SQRT32_DEP>(out, cyc, w, clk); run<EXP32_DEP>(out, cyc, w, clk);
    run<FMA_DEP_SMOV>(out, cyc, w, clk); run<FMA_DEP_VMOV>(out, cyc, w, clk); run<FMA_DEP_READLANE>(out, cyc, w, clk);
    run<CNDMASK_DEP>(out, cyc, w, clk); run<LDS_ACC>(out, cyc, w, clk);
    printf("\n");
  }
  return 0;
}
