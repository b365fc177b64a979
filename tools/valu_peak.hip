// GPU box: what is the VALU issue ceiling of gfx950?  Independent wave64 VALU instructions, 1..8 waves per SIMD on every CU, several
// instruction forms; prints wave-instructions/s, and cycles per instruction per SIMD from the shader clock (s_memtime) so that the
// answer does not depend on the clock the chip sustains.   hipcc --offload-arch=gfx950 -O2 tools/valu_peak.hip -o /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int FORM> __global__ void __launch_bounds__(256) k(float* out, long long* cyc)
{
  float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float m = 0.999f, c = 0.001f;
  long long t0 = clock64();
  for(int i = 0; i < ITERS; ++i)
  {
    if(FORM == 0)  // v_fma_f32, three VGPR sources
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    if(FORM == 1)  // v_mul_f32 e32, one SGPR source
      asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                   "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
    if(FORM == 2)  // v_fmac_f32 (2 VGPR sources + accumulate)
      asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                   "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    if(FORM == 3)  // integer: v_add_u32
      asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                   "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
  }
  long long t1 = clock64();
  float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if(s == 12345.678f) out[0] = s;
  if(blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int FORM> void run(const char* name, int cus)
{
  float* out; long long* cyc;
  hipMalloc(&out, 64); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int wavesPerSimd : {1, 2, 4, 8})
  {
    unsigned blocks = cus * wavesPerSimd;  // one 256-thread block = 4 waves = 1 wave per SIMD of a CU
    k<FORM><<<blocks, 256>>>(out, cyc);
    hipEventRecord(e0);
    for(int r = 0; r < 5; ++r) k<FORM><<<blocks, 256>>>(out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double instr = double(blocks) * 4 * ITERS * 8;
    double rate = instr / (ms * 1e-3 / 5);
    printf("%-28s waves/SIMD %d: %8.1f G wave-instr/s   s_memtime ticks per instr per SIMD %.3f  (ticks per wave-loop %lld)\n", name, wavesPerSimd, rate / 1e9,
           double(c) / (double(ITERS) * 8 * wavesPerSimd), c);
  }
}
int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  run<0>("v_fma_f32 (3 VGPR)", p.multiProcessorCount);
  run<1>("v_mul_f32 e32 (SGPR, VGPR)", p.multiProcessorCount);
  run<2>("v_fmac_f32 (2 VGPR + acc)", p.multiProcessorCount);
  run<3>("v_add_u32 (SGPR, VGPR)", p.multiProcessorCount);
  return 0;
}
