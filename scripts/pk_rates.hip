// scripts/pk_rates.hip -- does v_pk_add_f32 (two fp32 adds per lane in one instruction) buy the Viterbi kernels anything?
// The DP's work unit is (2 adds + 1 min3).  Plain: 3 instructions.  Packed: v_pk_add_f32 + v_min3_f32 = 2 instructions.
// A wavefront issues one VALU every ~7.5 cycles whatever it is (profiles/r02_valu_rates.txt), and the cost kernels run
// 2-4 wavefronts per SIMD, so fewer instructions per unit may pay even if the packed add takes twice the ALU time.
// Prints instruction rates and UNITS per second for both forms at 1..8 wavefronts per SIMD.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/pk scripts/pk_rates.hip && /tmp/pk
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(64) void stream(float *out, int iters, float seed)
{
  float a[8];
  f2 p[4], c2;
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
  for (int i = 0; i < 4; ++i) p[i] = f2{seed + i, seed - i};
  float c = seed * 0.5f, one = seed / seed;
  c2 = f2{c, one};
  for (int i = 0; i < iters; ++i)
  {
    if (MODE == 0) // plain mix: 9 instructions = 3 units
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_min3_f32 %2, %0, %1, %2\n"
                   "v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_min3_f32 %5, %3, %4, %5\n"
                   "v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_min3_f32 %2, %6, %7, %2\n"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(c), "v"(one));
    if (MODE == 1) // packed adds alone: 8 instructions
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                   : "v"(c2));
    if (MODE == 2) // packed mix: 8 instructions = 4 units
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_min3_f32 %4, %4, %5, %9\n v_pk_add_f32 %1, %1, %8\n v_min3_f32 %5, %5, %6, %9\n"
                   "v_pk_add_f32 %2, %2, %8\n v_min3_f32 %6, %6, %7, %9\n v_pk_add_f32 %3, %3, %8\n v_min3_f32 %7, %7, %4, %9\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                   : "v"(c2), "v"(c));
    if (MODE == 3) // packed adds with a broadcast low half of the second source (op_sel_hi): what "+ bg" needs
      asm volatile("v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %4 op_sel_hi:[1,0]\n"
                   "v_pk_add_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
                   "v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %4 op_sel_hi:[1,0]\n"
                   "v_pk_add_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                   : "v"(c2));
    if (MODE == 4) // packed mix 2: the real ratio of the DP, 18 adds : 9 min-type per cell = 1 pk_add : 1 min3, but with
                   // v_min_f32 for a third of the mins (8 instructions = 4 units)
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_min3_f32 %4, %4, %5, %9\n v_pk_add_f32 %1, %1, %8\n v_min_f32 %5, %5, %6\n"
                   "v_pk_add_f32 %2, %2, %8\n v_min3_f32 %6, %6, %7, %9\n v_pk_add_f32 %3, %3, %8\n v_min3_f32 %7, %7, %4, %9\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                   : "v"(c2), "v"(c));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
  if (s == 12345.0f) out[0] = s;
}

template <int MODE> static void run(char const *name, int per_iter, double units_per_iter, int wps)
{
  int const iters = 100000, blocks = 256 * 4 * wps;
  float *d;
  (void)hipMalloc(&d, 64);
  auto const t0 = std::chrono::steady_clock::now();
  double launches = 0;
  for (;;)
  {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(stream<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
    (void)hipDeviceSynchronize();
    launches += 4.0;
    double const s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s >= 1.5)
    {
      double const waves = launches * blocks * (double)iters;
      printf("%-44s %d waves/SIMD  %7.1f G wave-instr/s  %7.1f G units/s (unit = 2 adds + 1 min3)\n", name, wps,
             waves * per_iter / s / 1e9, waves * units_per_iter / s / 1e9);
      break;
    }
  }
  (void)hipFree(d);
}

int main()
{
  for (int w : {1, 2, 3, 4, 6, 8})
  {
    run<0>("plain: 2 v_add_f32 + v_min3_f32", 9, 3, w);
    run<2>("packed: v_pk_add_f32 + v_min3_f32", 8, 4, w);
    run<4>("packed: v_pk_add_f32 + v_min3/v_min", 8, 4, w);
  }
  for (int w : {1, 2, 4, 8})
  {
    run<1>("v_pk_add_f32 alone", 8, 0, w);
    run<3>("v_pk_add_f32 op_sel_hi:[1,0] alone", 8, 0, w);
  }
  return 0;
}
