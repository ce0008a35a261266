// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 under the register/occupancy shapes the GMM scoring
// kernel uses.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bench tools/mfma_f32_microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void mfma_loop(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[n][r] = (float)(threadIdx.x + n + r);
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 40; k++) {
#pragma unroll
      for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
      a += 1e-6f;
    }
  }
  float s = 0;
#pragma unroll
  for (int n = 0; n < NACC; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int WPS>
void run(const char *name, int blocks) {
  float *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  int iters = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC, WPS><<<blocks, 256>>>(d, 10, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC, WPS><<<blocks, 256>>>(d, iters, 1.0f, 2.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 /*waves*/ * iters * 40.0 * NACC * 4096.0;
  printf("%-28s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  // 256 CUs; a 256-thread block puts one wavefront on each SIMD of a CU
  run<4, 1>("1 wave/SIMD, 4 accumulators", 256 * 8);
  run<2, 1>("1 wave/SIMD, 2 accumulators", 256 * 8);
  run<1, 1>("1 wave/SIMD, 1 accumulator", 256 * 8);
  run<2, 2>("2 waves/SIMD, 2 accumulators", 256 * 16);
  run<4, 2>("2 waves/SIMD, 4 accumulators", 256 * 16);
  run<2, 4>("4 waves/SIMD, 2 accumulators", 256 * 32);
  return 0;
}
