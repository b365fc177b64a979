/* Test helper: array front-ends of include/pt_fpmath.h for tests/test_fpmath.py (host side of the contract). */
#include "../../include/pt_fpmath.h"
#define F1(name) void probe_##name(const float* x, float* out, long n) { for(long i = 0; i < n; ++i) out[i] = pt_##name(x[i]); }
F1(sin) F1(cos) F1(tan) F1(asin) F1(acos) F1(atan) F1(exp) F1(log)
void probe_atan2(const float* y, const float* x, float* out, long n) { for(long i = 0; i < n; ++i) out[i] = pt_atan2(y[i], x[i]); }
void probe_pow(const float* x, const float* y, float* out, long n) { for(long i = 0; i < n; ++i) out[i] = pt_pow(x[i], y[i]); }
