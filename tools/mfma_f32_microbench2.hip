// Microbenchmark 2: the GMM kernel's MFMA stream shape — 40 k-steps x 2 accumulators with DISTINCT A/B registers per
// step, (a) operands resident, (b) A operands re-loaded from global memory every block (10 x 16-byte loads per lane),
// (c) as (b) but prefetched one block ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float *w, float *out, int blocks_per_wave, int rows_total) {
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  float b[2][40];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int s = 0; s < 40; s++) b[n][s] = (float)(lane + n + s) * 1e-3f;
  f32x16 acc[2];
  f32x4 a[10], an[10];
  float total = 0;
  int row = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 37) % (rows_total - 64);
  auto load = [&](f32x4 (&dst)[10], int r0) {
#pragma unroll
    for (int m = 0; m < 10; m++) dst[m] = *reinterpret_cast<const f32x4 *>(w + (size_t)(r0 + col) * 80 + 8 * m + 4 * h);
  };
  load(a, row);
  if (MODE == 2) load(an, (row + 32) % (rows_total - 64));
  for (int j = 0; j < blocks_per_wave; j++) {
    if (MODE == 1) load(a, row);
    if (MODE == 2) {
#pragma unroll
      for (int m = 0; m < 10; m++) a[m] = an[m];
      load(an, (row + 32) % (rows_total - 64));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[n][r] = (float)(r + j);
#pragma unroll
    for (int m = 0; m < 10; m++)
#pragma unroll
      for (int cc = 0; cc < 4; cc++)
#pragma unroll
        for (int n = 0; n < 2; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][cc], b[n][4 * m + cc], acc[n], 0, 0, 0);
    total += acc[0][0] + acc[1][15] + acc[0][7] + acc[1][3];
    row = (row + 32) % (rows_total - 64);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

template <int MODE>
void run(const char *name, const float *w, int rows_total) {
  int blocks = 256 * 2 * 8, bpw = 200;
  float *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(w, d, 10, rows_total); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(w, d, bpw, rows_total);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * bpw * 80.0 * 4096.0;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  int rows = 160000;  // 51 MB of model rows, as the benchmark model
  float *w; hipMalloc(&w, (size_t)rows * 80 * 4); hipMemset(w, 0, (size_t)rows * 80 * 4);
  run<0>("operands resident (distinct regs)", w, rows);
  run<1>("A re-loaded every block (just in time)", w, rows);
  run<2>("A prefetched one block ahead", w, rows);
  return 0;
}
