// scripts/valu_rates.hip -- sustained issue rate of single VALU instructions (8 independent copies per iteration,
// 8 wavefronts per SIMD on every CU, ~2 s each): which of the instructions the Viterbi kernels are made of run at
// 2 cycles per wave64 and which at 4.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/vr scripts/valu_rates.hip && /tmp/vr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

#define REP8(OP)                                                                                                      \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                        \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])       \
               : "v"(c), "v"(one), "s"(sc))
#define ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %9, %8\n"
#define FMAS(i) "v_fma_f32 %" #i ", %" #i ", 1.0, %8\n"
#define MUL(i) "v_mul_f32 %" #i ", %" #i ", %9\n"
#define MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define MOV(i) "v_mov_b32 %" #i ", %8\n"
#define ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define ADDS(i) "v_add_f32 %" #i ", %10, %" #i "\n"
#define MINI(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define MIN3I(i) "v_min3_i32 %" #i ", %" #i ", %8, %9\n"
#define MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
// the Viterbi mix: per three instructions two adds and a min3 (a0 += c; a1 += c; a2 = min3(a0, a1, a2) ...)
#define MIXADD "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_min3_f32 %2, %0, %1, %2\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_min3_f32 %5, %3, %4, %5\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_min3_f32 %2, %6, %7, %2\n"
#define MIXFMA "v_fma_f32 %0, %0, 1.0, %8\n v_fma_f32 %1, %1, 1.0, %8\n v_min3_f32 %2, %0, %1, %2\n v_fma_f32 %3, %3, 1.0, %8\n v_fma_f32 %4, %4, 1.0, %8\n v_min3_f32 %5, %3, %4, %5\n v_fma_f32 %6, %6, 1.0, %8\n v_fma_f32 %7, %7, 1.0, %8\n v_min3_f32 %2, %6, %7, %2\n"
#define MIX1(OPS) asm volatile(OPS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "v"(one), "s"(sc))
#define DPP(i) "v_min_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

template <int MODE> __global__ __launch_bounds__(64) void stream(float *out, int iters, float seed)
{
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
  float c = seed * 0.5f, one = seed / seed;
  float sc = seed * 0.25f;
  for (int i = 0; i < iters; ++i)
  {
    if (MODE == 0) REP8(ADD);
    if (MODE == 1) REP8(FMA);
    if (MODE == 2) REP8(FMAS);
    if (MODE == 3) REP8(MUL);
    if (MODE == 4) REP8(MIN);
    if (MODE == 5) REP8(MIN3);
    if (MODE == 6) REP8(MOV);
    if (MODE == 7) REP8(ADDU);
    if (MODE == 8) REP8(ADDS);
    if (MODE == 9) REP8(DPP);
    if (MODE == 15) MIX1(MIXADD);
    if (MODE == 16) MIX1(MIXFMA);
    if (MODE == 10) REP8(MINI);
    if (MODE == 11) REP8(MINU);
    if (MODE == 12) REP8(MIN3I);
    if (MODE == 13) REP8(MAXF);
    if (MODE == 14) REP8(MED3);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.0f) out[0] = s;
}

template <int MODE> static void run(char const *name, int per_iter = 8, int wps = 8)
{
  int const iters = 100000, blocks = 256 * 4 * wps;
  float *d;
  (void)hipMalloc(&d, 64);
  auto const t0 = std::chrono::steady_clock::now();
  double ops = 0;
  for (;;)
  {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(stream<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
    (void)hipDeviceSynchronize();
    ops += 4.0 * blocks * (double)iters * per_iter;
    double const s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s >= 2.0)
    {
      printf("%-34s %7.1f G wave-instr/s  = %.2f cycles per wave64 instruction at 2.35 GHz\n", name, ops / s / 1e9,
             1024 * 2.35e9 / (ops / s));
      break;
    }
  }
  (void)hipFree(d);
}

int main()
{
  run<0>("v_add_f32 v, v, v");
  run<1>("v_fma_f32 v, v, v(1.0), v");
  run<2>("v_fma_f32 v, v, 1.0, v");
  run<3>("v_mul_f32 v, v, v");
  run<4>("v_min_f32 v, v, v");
  run<5>("v_min3_f32 v, v, v, v");
  run<6>("v_mov_b32 v, v");
  run<7>("v_add_u32 v, v, v");
  run<8>("v_add_f32 v, s, v");
  run<9>("v_min_f32_dpp quad_perm");
  run<15>("mix: 2 v_add_f32 + v_min3_f32", 9);
  run<16>("mix: 2 v_fma_f32(x,1.0,y) + v_min3_f32", 9);
  run<15>("mix: 2 v_add_f32 + v_min3_f32", 9);
  run<16>("mix: 2 v_fma_f32(x,1.0,y) + v_min3_f32", 9);
  for (int w : {1, 2, 3, 4, 6})
  {
    char nm[64];
    snprintf(nm, sizeof nm, "mix 2 add + min3, %d waves/SIMD", w);
    run<15>(nm, 9, w);
  }
  run<10>("v_min_i32 v, v, v");
  run<11>("v_min_u32 v, v, v");
  run<12>("v_min3_i32 v, v, v, v");
  run<13>("v_max_f32 v, v, v");
  run<14>("v_med3_f32 v, v, v, v");
  return 0;
}
