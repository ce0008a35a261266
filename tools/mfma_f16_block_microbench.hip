// Microbenchmark: the block loop of the f16×2 scoring kernel rebuilt piece by piece, to see which ingredient costs the
// matrix pipe its duty cycle.  Each variant runs 2 workgroups of 4 wavefronts per CU (2 wavefronts per SIMD), 30 MFMAs
// (v_mfma_f32_32x32x16_f16) per "block" into two accumulators, like gmm_split_single_kernel<5, 2>.
//   bit 0: accumulators re-initialised per block (32 v_mov)      bit 1: 20 distinct B operands per tile (registers)
//   bit 2: A operands read from LDS (10 ds_read_b128 per block)  bit 3: s_barrier per block
//   bit 4: a reduction of the accumulators per block (stand-in epilogue: 32 exp2 + adds)
//   bit 5: the next block's A operands copied global → LDS (global_load_lds, 10 KB per block from a 50 MB table)
//   bit 6: one score per (frame, block) staged in LDS and flushed to HBM every 32 blocks (nontemporal, 128-byte rows)
//   bit 7: the scoring kernel's own epilogue (permlane32 swaps, hardware log) instead of the stand-in
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f16_block_bench tools/mfma_f16_block_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float swap32(float v, int h) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(h ? r[0] : r[1]);
}

template <int kFlags>
__global__ __launch_bounds__(256, 2) void block_loop(float *out, int blocks, float seed, const uint4 *table, int table_blocks,
                                                      float *scores) {
  __shared__ uint4 a_lds[2][640];
  __shared__ float stage_all[4][64 * 33];
  float *stage = stage_all[threadIdx.x >> 6];
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned rng = blockIdx.x * 2654435761u + 12345u;
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 1280; i += 256) {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (_Float16)(seed * (float)((i * 7 + e * 29) % 97 - 48) * 0.01f);
    (&a_lds[0][0])[i] = __builtin_bit_cast(uint4, v);
  }
  __syncthreads();
  f16x8 b[2][5][2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int e = 0; e < 8; e++)
          b[n][s][q][e] = (_Float16)(seed * (float)((threadIdx.x * 11 + (n * 10 + s * 2 + q) * 17 + e * 31) % 89 - 44) * 0.01f);
  f32x16 init;
#pragma unroll
  for (int r = 0; r < 16; r++) init[r] = seed * (float)(r + lane);
  f32x16 acc[2] = {init, init};
  float total = 0.0f;
  for (int j = 0; j < blocks; j++) {
    const int buf = j & 1;
    if (kFlags & 32) {
      rng = rng * 1664525u + 1013904223u;
      const uint4 *src = table + (size_t)((rng >> 8) % (unsigned)table_blocks) * 640;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int u0 = 64 * wave_u + 256 * i;
        if (u0 < 640) __builtin_amdgcn_global_load_lds((gptr_t)(src + u0 + lane), (lptr_t)&a_lds[buf ^ 1][u0], 16, 0, 0);
      }
    }
    if (kFlags & 1) { acc[0] = init; acc[1] = init; asm volatile("" : "+v"(init)); }
    f16x8 a_cur[2], a_nxt[2];
    auto read_a = [&](int s, f16x8 (&a)[2]) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        if (kFlags & 4) a[q] = __builtin_bit_cast(f16x8, a_lds[buf][((s * 2 + q) * 2 + h) * 32 + col]);
        else a[q] = b[0][s][q];
      }
    };
    read_a(0, a_cur);
#pragma unroll
    for (int s = 0; s < 5; s++) {
      if (s + 1 < 5) read_a(s + 1, a_nxt);
      constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
          const int ss = (kFlags & 2) ? s : 0;
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[pa[t]], b[n][ss][pb[t]], acc[n], 0, 0, 0);
        }
      a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
    }
    if (kFlags & 128) {
      float mx[2], sum[2];
#pragma unroll
      for (int n = 0; n < 2; n++) {
        float m = acc[n][0];
#pragma unroll
        for (int r = 1; r < 16; r++) m = fmaxf(m, acc[n][r]);
        m = fmaxf(m, swap32(m, h));
        float sv = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) sv += __builtin_amdgcn_exp2f((acc[n][r] - m) * 1.44269504f);
        sv += swap32(sv, h);
        mx[n] = m; sum[n] = sv;
      }
      const float v = fmaf(__builtin_amdgcn_logf(h ? sum[1] : sum[0]), 0.6931472f, h ? mx[1] : mx[0]);
      if (kFlags & 64) stage[(32 * h + col) * 33 + (j & 31)] = v; else total += v;
    } else if (kFlags & 16) {
#pragma unroll
      for (int n = 0; n < 2; n++) {
        float m = acc[n][0];
#pragma unroll
        for (int r = 1; r < 16; r++) m = fmaxf(m, acc[n][r]);
        float sv = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) sv += __builtin_amdgcn_exp2f((acc[n][r] - m) * 1.44f);
        total += sv;
      }
    } else if (!(kFlags & 1)) {
      // accumulators carry over: nothing to do
    } else {
      total += acc[0][0] + acc[1][15];
    }
    if ((kFlags & 64) && (j & 31) == 31) {
      __builtin_amdgcn_wave_barrier();
      const size_t row0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
#pragma unroll 4
      for (int i = 0; i < 32; i++) {
        const int r = h + 2 * i;
        __builtin_nontemporal_store(stage[r * 33 + col], &scores[(row0 + r) * 512 + ((j - 31) & 511) + col]);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (kFlags & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kFlags & 8) __syncthreads();
  }
  float s = total;
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kFlags>
void run(const char *name) {
  const int wgs = 512, blocks = 20000;
  float *d; hipMalloc(&d, (size_t)wgs * 256 * 4);
  static uint4 *table = nullptr; static float *scores = nullptr;
  const int table_blocks = 5000;
  if (!table) {
    hipMalloc(&table, (size_t)table_blocks * 640 * 16);
    hipMemset(table, 0x3c, (size_t)table_blocks * 640 * 16);   // f16 0x3c3c ≈ 1.06
    hipMalloc(&scores, (size_t)wgs * 4 * 64 * 512 * 4);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  block_loop<kFlags><<<wgs, 256>>>(d, 100, 1.0f, table, table_blocks, scores);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  block_loop<kFlags><<<wgs, 256>>>(d, blocks, 1.0f, table, table_blocks, scores);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)wgs * 4 * blocks * 30.0 * 32768.0;
  printf("%-64s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  run<0>("MFMA only (accumulators carried, one B set, A in registers)");
  run<1>("+ accumulators re-initialised per block");
  run<3>("+ 20 distinct B operands per tile");
  run<7>("+ A operands from LDS");
  run<15>("+ s_barrier per block");
  run<31>("+ max/exp2/sum reduction per block");
  run<16 + 1 + 2>("re-init + distinct B + reduction, A in registers, no barrier");
  run<15 + 128>("re-init, B, A from LDS, barrier + real epilogue");
  run<15 + 128 + 64>("  + staged scores flushed to HBM");
  run<15 + 128 + 32>("  + global_load_lds copy of the next block");
  run<15 + 128 + 64 + 32>("  + both");
  run<15 + 32>("re-init, B, A from LDS, barrier + copy, no epilogue");
  run<0>("MFMA only (again)");
  return 0;
}
