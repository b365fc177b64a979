// GPU box diagnostic: ulp error of the device math functions (compiled with the product's flags) against
// double-precision libm rounded to fp32.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/math_ulps.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(int fn, const float* a, const float* b, float* out, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  float x = a[i], y = b[i], r = 0;
  switch(fn)
  {
    case 0: r = sinf(x); break;
    case 1: r = cosf(x); break;
    case 2: r = tanf(x); break;
    case 3: r = acosf(x); break;
    case 4: r = asinf(x); break;
    case 5: r = expf(x); break;
    case 6: r = logf(x); break;
    case 7: r = powf(x, y); break;
    case 8: r = atan2f(y, x); break;
    case 9: r = sqrtf(x); break;
    case 10: r = 1.0f / x; break;
    case 11: r = x / y; break;
    case 12: r = powf(x, 2.2f); break;
    case 13: r = powf(x, 5.0f); break;
  }
  out[i] = r;
}
static double ref(int fn, double x, double y)
{
  switch(fn)
  {
    case 0: return sin(x); case 1: return cos(x); case 2: return tan(x); case 3: return acos(x); case 4: return asin(x); case 5: return exp(x);
    case 6: return log(x); case 7: return pow(x, y); case 8: return atan2(y, x); case 9: return sqrt(x); case 10: return 1.0 / x; case 11: return x / y;
    case 12: return pow(x, (double)2.2f); case 13: return pow(x, 5.0);
  }
  return 0;
}
int main()
{
  const int n = 1 << 20;
  const char* names[] = {"sinf[-2pi,2pi]", "cosf[-2pi,2pi]", "tanf[-1.5,1.5]", "acosf[-1,1]", "asinf[-1,1]", "expf[-20,5]", "logf(0,1]", "powf(x in (0,1], y in [0,3])", "atan2f", "sqrtf", "1/x", "x/y", "powf(x,2.2)", "powf(x,5)"};
  std::vector<float> a(n), b(n), o(n);
  float *da, *db, *dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * 4);
  for(int fn = 0; fn < 14; ++fn)
  {
    srand(123 + fn);
    for(int i = 0; i < n; ++i)
    {
      double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
      float x, y = (float)(v * 3.0);
      switch(fn)
      {
        case 0: case 1: x = (float)((u * 2 - 1) * 6.2831853); break;
        case 2: x = (float)((u * 2 - 1) * 1.5); break;
        case 3: case 4: x = (float)(u * 2 - 1); break;
        case 5: x = (float)(u * 25 - 20); break;
        case 8: x = (float)(u * 2 - 1); y = (float)(v * 2 - 1); break;
        case 11: x = (float)(u * 10 - 5); y = (float)(v * 10 - 5 + 1e-3); break;
        default: x = (float)(u + 1e-6); break;
      }
      a[i] = x; b[i] = y;
    }
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(fn, da, db, dout, n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    double maxulp = 0, sum = 0; long exact = 0;
    for(int i = 0; i < n; ++i)
    {
      double r = ref(fn, a[i], b[i]);
      float rf = (float)r;
      float nx = nextafterf(fabsf(rf), INFINITY) - fabsf(rf);
      double ulp = fabs((double)o[i] - r) / (nx > 0 ? nx : 1e-45);
      if(ulp > maxulp) maxulp = ulp;
      sum += ulp;
      exact += (o[i] == rf);
    }
    printf("%-32s max %.3f ulp  mean %.3f ulp  correctly-rounded %.1f%%\n", names[fn], maxulp, sum / n, 100.0 * exact / n);
  }
  return 0;
}
