// Microbenchmark: sustained rate of v_mfma_f32_32x32x16_bf16 on gfx950 with operands resident in registers, random
// (non-zero) data, for tens of milliseconds — long enough for the clock to settle at whatever the power limit allows.
// This is the ceiling the bf16×3 scoring kernel is priced against in DESIGN.md besides the 2.5 PFLOP/s datasheet peak.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_bf16_bench tools/mfma_bf16_microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void mfma_loop(float *out, int iters, float seed) {
  f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[n][r] = 0.0f;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      a[k][e] = (__bf16)(seed * (float)((threadIdx.x * 7 + k * 13 + e * 29) % 97 - 48) * 0.01f);
      b[k][e] = (__bf16)(seed * (float)((threadIdx.x * 11 + k * 17 + e * 31) % 89 - 44) * 0.01f);
    }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 32; k++) {
#pragma unroll
      for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k & 3], b[(k + n) & 3], acc[n], 0, 0, 0);
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
void run(const char *name, int blocks, int iters) {
  float *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC, WPS><<<blocks, 256>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC, WPS><<<blocks, 256>>>(d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 * NACC * 32768.0;
  printf("%-30s blocks %5d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  run<2, 2>("2 waves/SIMD, 2 accumulators", 256 * 8, 6000);
  run<4, 1>("1 wave/SIMD, 4 accumulators", 256 * 4, 6000);
  run<2, 2>("2 waves/SIMD, 2 acc (again)", 256 * 8, 12000);
  return 0;
}
