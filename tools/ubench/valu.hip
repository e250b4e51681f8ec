// microbenchmark: VALU issue rate of scalar vs packed fp32 mul/add on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
   float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
   f2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7}, y4 = {x1, x2}, y5 = {x3, x4}, y6 = {x5, x6}, y7 = {x7, x0};
   f2 aa = {a, a}, bb = {b, b};
   for (int i = 0; i < iters; i++) {
      if (MODE == 0) {  // 16 scalar ops (8 mul + 8 add), independent chains
         x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
         x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
      } else {          // 16 packed ops
         y0 = y0 * aa; y1 = y1 * aa; y2 = y2 * aa; y3 = y3 * aa; y4 = y4 * aa; y5 = y5 * aa; y6 = y6 * aa; y7 = y7 * aa;
         y0 = y0 + bb; y1 = y1 + bb; y2 = y2 + bb; y3 = y3 + bb; y4 = y4 + bb; y5 = y5 + bb; y6 = y6 + bb; y7 = y7 + bb;
      }
   }
   float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0.x + y0.y + y1.x + y1.y + y2.x + y2.y + y3.x + y3.y + y4.x + y4.y + y5.x + y5.y + y6.x + y6.y + y7.x + y7.y;
   if (r == 12345.678f) out[0] = r;
}
int main()
{
   float* d; hipMalloc(&d, 4);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const int iters = 20000;
   for (int wavesPerSimd : {1, 2, 4, 8}) {
      int blocks = 256 * wavesPerSimd;  // 256 CUs x (wavesPerSimd) blocks of 4 waves
      for (int mode = 0; mode < 2; mode++) {
         for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
         }
         float ms; hipEventElapsedTime(&ms, e0, e1);
         double instr_per_simd = (double)iters * 16 * wavesPerSimd;  // wave-instructions per SIMD
         printf("waves/SIMD %d %s: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", wavesPerSimd,
                mode ? "packed" : "scalar", ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
      }
   }
   return 0;
}
