// GPU box: the VALU issue ceiling FOR THE INSTRUCTION MIX of the trace machine and of k_shade (round-5 review, item 2: "a tools/valu_mix.hip that replays
// k_closest_p's instruction mix -- cndmask / cmp / fma with its dependency depth, 96 VGPRs, 5 waves -- to get the issue ceiling for that mix").
//   hipcc --offload-arch=gfx950 -O2 tools/valu_mix.hip -o /tmp/valu_mix && /tmp/valu_mix > profiles/rNN_valu_mix.txt
// tools/valu_peak.hip measures independent instructions of ONE form (v_fmac_f32: 870-970 G wave-instructions/s at 8 waves per SIMD, the ceiling
// bench.py calibrates per run); the kernels issue a mix in which 57 % of the vector instructions are selects, compares, min / max, moves and bit
// operations, most of them fed by the instruction before (profiles/r06a_binders.json: k_closest_p 76.1 VALU per sample = 6.7 fma + 6.9 mul + 6.1 add
// + 12.1 int32 + 0.5 cvt + 0.8 int64 + 43 others, next to 29.0 SALU).  The blocks below replay those proportions with a dependency depth of 3-4, no
// memory instruction at all, at the occupancy the kernels run at.  What they reach is the ceiling an instruction-count argument may use.
//   MIX 0  trace machine (k_closest_p / k_shadow_p): per block 2 fma, 2 mul, 2 add, 2 min / max, 3 compares into SGPR pairs, 4 selects on them,
//          3 integer, 1 bit op, 1 move = 20 VALU + 7 SALU (4 blocks per loop iteration = 80 + 28; the kernels: 76 + 29)
//   MIX 1  k_shade: per block 4 fma, 4 mul, 2 add, 1 min / max, 2 compares, 2 selects, 2 integer, 1 cvt, 1 move + the IEEE division's shape
//          (v_div_scale x2, v_rcp, v_div_fmas, v_div_fixup around the fmas) = 28 VALU + 5 SALU (3 blocks per loop iteration = 84 + 15; the kernel: 78.8 + 19.0 per sample)
//   MIX 2  independent v_fmac_f32 (the form bench.py calibrates with), as the cross-check against tools/valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define TRACE_BLOCK_ASM                                                                                                                      \
  "v_fma_f32 %0, %8, %21, %22\n v_fma_f32 %1, %9, %21, %22\n v_mul_f32 %2, %0, %23\n v_mul_f32 %3, %1, %23\n v_add_f32 %4, %2, %3\n"          \
  "v_sub_f32 %5, %2, %3\n v_max_f32 %6, %0, %1\n v_min_f32 %7, %4, %5\n v_cmp_le_f32 %16, %6, %7\n v_cmp_lt_f32 %17, %2, %3\n"             \
  "v_cmp_ne_u32 %18, %10, %11\n s_and_b64 %16, %16, %17\n s_and_b64 %16, %16, %18\n v_cndmask_b32 %12, %10, %11, %16\n"                     \
  "v_cndmask_b32 %8, %6, %7, %17\n v_cndmask_b32 %9, %4, %5, %16\n v_cndmask_b32 %13, %12, %10, %18\n v_add_u32 %10, %10, %13\n"           \
  "v_lshl_add_u32 %11, %12, 2, %11\n v_and_b32 %12, 0x3fffffff, %13\n v_mov_b32 %14, %6\n v_add_u32 %13, %11, %14\n"                        \
  "s_add_u32 %19, %19, 1\n s_cmp_lt_u32 %19, 77\n s_cselect_b32 %20, %19, 3\n s_lshl_b32 %20, %20, 2\n s_xor_b64 %18, %18, %17\n"

#define SHADE_BLOCK_ASM                                                                                                                      \
  "v_fma_f32 %0, %8, %21, %22\n v_fma_f32 %1, %9, %21, %22\n v_fma_f32 %2, %0, %1, %22\n v_fma_f32 %3, %1, %0, %21\n"                        \
  "v_mul_f32 %4, %2, %23\n v_mul_f32 %5, %3, %23\n v_mul_f32 %6, %4, %5\n v_mul_f32 %7, %5, %2\n v_add_f32 %8, %6, %7\n v_sub_f32 %9, %6, %7\n" \
  "v_max_f32 %14, %8, %9\n v_cmp_lt_f32 %16, %4, %5\n v_cmp_gt_f32 %17, %6, %7\n v_cndmask_b32 %8, %8, %14, %16\n v_cndmask_b32 %9, %9, %14, %17\n" \
  "v_add_u32 %10, %10, %11\n v_mul_lo_u32 %11, %10, %12\n v_cvt_f32_u32 %15, %11\n v_mov_b32 %12, %13\n"                                      \
  "v_div_scale_f32 %0, vcc, %8, %8, %9\n v_div_scale_f32 %1, vcc, %9, %8, %9\n v_rcp_f32 %2, %0\n v_fma_f32 %3, -%0, %2, 1.0\n v_fma_f32 %2, %3, %2, %2\n" \
  "v_mul_f32 %3, %1, %2\n v_fma_f32 %4, -%0, %3, %1\n v_div_fmas_f32 %4, %4, %2, %3\n v_div_fixup_f32 %9, %4, %8, %9\n"                       \
  "s_add_u32 %19, %19, 1\n s_cmp_lt_u32 %19, 77\n s_cselect_b32 %20, %19, 3\n s_and_b64 %18, %16, %17\n s_lshl_b32 %20, %20, 2\n"

template <int MIX, int WAVES>
__global__ void __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) __launch_bounds__(256) k(float* out, int iters)
{
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float p0 = 0.5f + a0, p1 = 0.25f + a0;
  unsigned i0 = threadIdx.x, i1 = i0 * 3u + 1u, i2 = 5u, i3 = 7u;
  float t0 = 0.f, t1 = 0.f;
  const float q = 0.999f, r = 0.001f, m = 1.0001f;
  unsigned long long s0 = 0, s1 = 0, s2 = 0;
  unsigned u0 = 0, u1 = 0;
  for(int i = 0; i < iters; ++i)
  {
    if(MIX == 0)
      asm volatile(TRACE_BLOCK_ASM TRACE_BLOCK_ASM TRACE_BLOCK_ASM TRACE_BLOCK_ASM
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(t0), "+v"(t1),
                     "+s"(s0), "+s"(s1), "+s"(s2), "+s"(u0), "+s"(u1)
                   : "v"(q), "v"(r), "v"(m)
                   : "vcc", "scc");
    if(MIX == 1)
      asm volatile(SHADE_BLOCK_ASM SHADE_BLOCK_ASM SHADE_BLOCK_ASM
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(t0), "+v"(t1),
                     "+s"(s0), "+s"(s1), "+s"(s2), "+s"(u0), "+s"(u1)
                   : "v"(q), "v"(r), "v"(m)
                   : "vcc", "scc");
#define FMAC8 "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
    if(MIX == 2)  // 64 per loop iteration (a loop of 8 pays a branch per 8 instructions: 620 G/s instead of the ceiling)
      asm volatile(FMAC8 FMAC8 FMAC8 FMAC8 FMAC8 FMAC8 FMAC8 FMAC8
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q), "v"(r));
  }
  float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + p0 + p1 + t0 + t1 + float(i0 + i1 + i2 + i3);
  if(s == 12345.678f)
    out[0] = s;
}

template <int MIX, int WAVES>
void run(const char* name, int cus, int valuPerIter, int saluPerIter)
{
  const int ITERS = 2400000 / valuPerIter;  // ~2.4 M VALU per wave and launch: a launch lasts milliseconds, its start-up does not show
  float* out;
  (void)hipMalloc(&out, 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const unsigned blocks = cus * WAVES;  // one 256-thread block = one wave per SIMD of a CU
  k<MIX, WAVES><<<blocks, 256>>>(out, ITERS);
  (void)hipEventRecord(e0);
  for(int rr = 0; rr < 5; ++rr)
    k<MIX, WAVES><<<blocks, 256>>>(out, ITERS);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double t     = ms * 1e-3 / 5;
  const double valu  = double(blocks) * 4 * ITERS * valuPerIter / t;
  const double simds = cus * 4.0;
  printf("%-34s waves/SIMD %d: %7.1f G VALU wave-instr/s (+ %6.1f G SALU/s)   = one VALU per %.2f ns per SIMD\n", name, WAVES, valu / 1e9,
         double(blocks) * 4 * ITERS * saluPerIter / t / 1e9, 1e9 * simds / valu);
  (void)hipFree(out);
}
int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int cus = p.multiProcessorCount;
  run<2, 8>("independent v_fmac_f32", cus, 64, 0);
  run<2, 5>("independent v_fmac_f32", cus, 64, 0);
  run<0, 8>("trace-machine mix (80 VALU + 28 SALU)", cus, 80, 28);
  run<0, 5>("trace-machine mix (k_closest_p: 5)", cus, 80, 28);
  run<0, 4>("trace-machine mix (two-level: 4)", cus, 80, 28);
  run<0, 2>("trace-machine mix", cus, 80, 28);
  run<0, 1>("trace-machine mix", cus, 80, 28);
  run<1, 8>("k_shade mix (84 VALU + 15 SALU)", cus, 84, 15);
  run<1, 4>("k_shade mix (k_shade: 4)", cus, 84, 15);
  run<1, 2>("k_shade mix", cus, 84, 15);
  run<1, 1>("k_shade mix", cus, 84, 15);
  return 0;
}
