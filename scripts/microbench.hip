// scripts/microbench.hip -- issue-rate probes that decide how the Viterbi kernels are shaped.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mb scripts/microbench.hip && /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(64) void probe(float *out, int iters, float seed)
{
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float c = seed * 0.5f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i)
  {
    if (MODE == 0) // 8 independent v_add_f32
    {
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
    if (MODE == 1) // 4 independent v_pk_add_f32 (same 8 adds)
    {
      float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, cc = {c, c};
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    }
    if (MODE == 2) // 8 independent v_min3_f32
    {
      asm volatile("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n"
                   "v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
    if (MODE == 3) // dependent chain of 8 v_add_f32
    {
      asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                   "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                   : "+v"(a0) : "v"(c));
    }
    if (MODE == 4) // wave min reduction (6 DPP steps + readlane), as in lane_ops_gpu.h
    {
      float v = a0;
      asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                   "s_nop 1" : "+v"(v));
      a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)) + a1;
    }
    if (MODE == 5) // 8 independent SALU ops
    {
      int s0 = i, s1 = i + 1;
      asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n"
                   "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n"
                   : "+s"(s0), "+s"(s1));
      a0 += __int_as_float(s0 & 1);
    }
    if (MODE == 6) // ballot + scalar branch on it (the lazy D->D vote)
    {
      if (__builtin_amdgcn_ballot_w64(a0 < a1 - 1e30f)) a2 += 1.0f;
      a0 += c;
    }
  }
  long long t1 = clock64();
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0)
  {
    out[blockIdx.x * 2] = (float)(t1 - t0);
    out[blockIdx.x * 2 + 1] = r;
  }
}

template <int MODE> void run(char const *name, int ops_per_iter)
{
  int const iters = 20000;
  float *d;
  hipMalloc(&d, sizeof(float) * 2 * 65536);
  int dev_cus = 256;
  for (int wps : {1, 2, 4, 8})
  {
    int blocks = dev_cus * 4 * wps; // one wave per block; wps waves per SIMD when evenly spread
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(2 * blocks);
    hipMemcpy(h.data(), d, sizeof(float) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) cyc += h[2 * b];
    cyc /= blocks;
    printf("%-28s waves/SIMD=%d  clock64 ticks/iter=%8.2f  ticks/op=%6.2f  wall=%7.3f ms  ns/op/wave=%6.3f\n", name, wps,
           cyc / iters, cyc / iters / ops_per_iter, ms, ms * 1e6 / iters / ops_per_iter);
  }
  hipFree(d);
}

int main()
{
  run<0>("8x v_add_f32 (indep)", 8);
  run<1>("4x v_pk_add_f32 (indep)", 4);
  run<2>("8x v_min3_f32 (indep)", 8);
  run<3>("8x v_add_f32 (dependent)", 8);
  run<4>("wave_min (6 dpp + readlane)", 1);
  run<5>("8x s_add_u32", 8);
  run<6>("ballot + branch", 1);
  return 0;
}
