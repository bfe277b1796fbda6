// scripts/valu_sustained.hip -- what a pure VALU stream sustains for seconds (not the ~1 ms of microbench.hip,
// which is over before the clocks settle): 8 independent v_add_f32 (or 4 v_min3_f32 + 4 v_add_f32) per iteration,
// W wavefronts per SIMD on every CU, launched back to back for about SECONDS; prints wave-instructions per second.
// Run beside `rocm-smi --showclocks --showpower` (scripts/valu_sustained.sh).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/vs scripts/valu_sustained.hip && /tmp/vs 8 6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

template <int MODE> __global__ __launch_bounds__(64) void stream(float *out, int iters, float seed)
{
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float c = seed * 0.5f;
  for (int i = 0; i < iters; ++i)
  {
    if (MODE == 0)
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    else // the Viterbi mix: two adds per min3
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_min3_f32 %2, %0, %1, %2\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_min3_f32 %5, %3, %4, %5\n v_add_f32 %6, %6, %8\n v_min_f32 %7, %7, %6\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0f) out[0] = a0;
}

int main(int argc, char **argv)
{
  int const wps = argc > 1 ? atoi(argv[1]) : 8;
  double const seconds = argc > 2 ? atof(argv[2]) : 6.0;
  int const mode = argc > 3 ? atoi(argv[3]) : 0;
  int const iters = 200000, blocks = 256 * 4 * wps;
  float *d;
  hipMalloc(&d, 64);
  auto const t0 = std::chrono::steady_clock::now();
  double ops = 0;
  int launches = 0;
  for (;;)
  {
    for (int k = 0; k < 8; ++k)
    {
      if (mode == 0)
        hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
      else
        hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
    }
    hipDeviceSynchronize();
    launches += 8;
    ops += 8.0 * blocks * (double)iters * 8;
    double const s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s >= seconds)
    {
      printf("mode %d, %d wavefronts per SIMD, %d launches in %.2f s: %.1f G wave-instr/s (nominal 1228.8 at 2.4 GHz, 2 cycles each)\n",
             mode, wps, launches, s, ops / s / 1e9);
      break;
    }
  }
  return 0;
}
