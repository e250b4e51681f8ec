// microbenchmark: the density pass's TEST step in isolation - six ds_read_b128 of an SoA tile,
// 24 packed fp32 operations, 8 v_alignbit - at the occupancy the kernel runs with (6 workgroups of
// 256 threads per CU, 25 KB of LDS each).  ns per 8-slot step per SIMD, to compare with the ≈85 ns
// the step costs inside k_full_density_tiled (229 us of TEST / (64 waves per SIMD x 42 steps)).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef float __attribute__((ext_vector_type(4))) f32x4;
extern __shared__ __attribute__((aligned(16))) float lds[];
__device__ __forceinline__ f32x4 rd4(const float* b, int i) { return *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(b + i, 16)); }
__device__ __forceinline__ f32x2 scr(f32x2 px, f32x2 py, f32x2 pz, f32x2 cx, f32x2 cy, f32x2 cz, f32x2 mh)
{
   const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
   return __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, mh)));
}
__device__ __forceinline__ uint32_t test8(const float* X, const float* Y, const float* Z, int t, f32x2 px, f32x2 py, f32x2 pz, float h2)
{
   const f32x4 X0 = rd4(X, t), X1 = rd4(X, t + 4), Y0 = rd4(Y, t), Y1 = rd4(Y, t + 4), Z0 = rd4(Z, t), Z1 = rd4(Z, t + 4);
   const f32x2 mh = {-h2, -h2};
   const f32x2 da = scr(px, py, pz, f32x2{X0.x, X0.y}, f32x2{Y0.x, Y0.y}, f32x2{Z0.x, Z0.y}, mh);
   const f32x2 db = scr(px, py, pz, f32x2{X0.z, X0.w}, f32x2{Y0.z, Y0.w}, f32x2{Z0.z, Z0.w}, mh);
   const f32x2 dc = scr(px, py, pz, f32x2{X1.x, X1.y}, f32x2{Y1.x, Y1.y}, f32x2{Z1.x, Z1.y}, mh);
   const f32x2 dd = scr(px, py, pz, f32x2{X1.z, X1.w}, f32x2{Y1.z, Y1.w}, f32x2{Z1.z, Z1.w}, mh);
   uint32_t m = 0;
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dd.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dd.x), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dc.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dc.x), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(db.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(db.x), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(da.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(da.x), 31);
   return m;
}
template <int MODE>
__global__ void __launch_bounds__(256, 6) k(unsigned* out, int iters, int cap, float h2, int pattern)
{
   float* X = lds; float* Y = X + cap; float* Z = Y + cap;
   for (int i = threadIdx.x; i < cap; i += 256) { X[i] = i * 0.001f; Y[i] = i * 0.002f; Z[i] = i * 0.003f; }
   __syncthreads();
   const float p = threadIdx.x * 0.0013f;
   const f32x2 px = {p, p}, py = {2 * p, 2 * p}, pz = {3 * p, 3 * p};
   unsigned acc = 0;
   // pattern 0: every lane its own start; 1: all lanes the same slots (pure broadcast reads);
   // 2: eight lanes per start, 7.6 apart - the lanes of a cell share their ranges, as in the kernel
   int t = pattern == 0 ? (((threadIdx.x * 7) & 1023) & ~3) : pattern == 1 ? 0 : ((((threadIdx.x & 63) >> 3) * 8 + (threadIdx.x >> 6) * 64) & ~3);
   for (int it = 0; it < iters; it++) {
      if (MODE == 0) {                            // one step per trip, result consumed at once
         acc += __builtin_popcount(test8(X, Y, Z, t, px, py, pz, h2));
         t = (t + 8) & 1023;
      } else {                                    // four steps per trip (a 32-slot chunk)
         uint32_t m = test8(X, Y, Z, t, px, py, pz, h2);
         m |= test8(X, Y, Z, t + 8, px, py, pz, h2) << 8;
         m |= test8(X, Y, Z, t + 16, px, py, pz, h2) << 16;
         m |= test8(X, Y, Z, t + 24, px, py, pz, h2) << 24;
         acc += __builtin_popcount(m);
         t = (t + 32) & 1023;
      }
   }
   if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
   unsigned* d; (void)hipMalloc(&d, 4);
   hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
   const int cap = 2048 + 64;
   const size_t ldsb = (size_t)cap * 12;
   for (int pattern = 0; pattern < 3; pattern++)
   for (int mode = 0; mode < 2; mode++) {
      for (int wgs : {2, 6}) {
         const int iters = mode == 1 ? 2000 : 8000;
         float ms = 0;
         for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wgs), dim3(256), ldsb, 0, d, iters, cap, 0.01f, pattern);
            else hipLaunchKernelGGL(k<1>, dim3(256 * wgs), dim3(256), ldsb, 0, d, iters, cap, 0.01f, pattern);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
         }
         const double steps_per_simd = (double)iters * (mode == 1 ? 4 : 1) * wgs;   // one wave of each workgroup per SIMD
         const char* names[2] = {"1 step per trip ", "4 steps per trip"};
         const char* pats[3] = {"own start per lane", "all lanes same slots", "8 lanes per start"};
         printf("%-20s %s, %d workgroups/CU: %6.1f ns per 8-slot step per SIMD\n", pats[pattern], names[mode], wgs, ms * 1e6 / steps_per_simd);
      }
   }
   return 0;
}
