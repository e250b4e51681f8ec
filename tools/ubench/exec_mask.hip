// microbenchmark: does the issue cost of a VALU instruction depend on how many lanes EXEC enables?
// (the append loop of the density pass runs with a handful of active lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
#define S_IADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define S_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define S_ADD(i) "v_add_f32 %" #i ", %" #i ", %9\n"
#define S_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define OPS8(S)                                                                                  \
   asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)                                          \
                : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),       \
                  "+v"(r[6]), "+v"(r[7])                                                         \
                : "v"(a), "v"(b))
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float fa, float fb, unsigned long long lanes)
{
   float r[8];
   for (int i = 0; i < 8; i++) r[i] = threadIdx.x + i;
   float a = fa, b = fb;
   if ((lanes >> (threadIdx.x & 63)) & 1ull) {
      for (int it = 0; it < iters; it++) {
         if (MODE == 0) { OPS8(S_IADD); OPS8(S_IADD); }
         if (MODE == 1) { OPS8(S_FMA); OPS8(S_FMA); }
         if (MODE == 2) { OPS8(S_MUL); OPS8(S_ADD); }
         if (MODE == 3) { OPS8(S_ALIGN); OPS8(S_ALIGN); }
      }
   }
   float s = 0;
   for (int i = 0; i < 8; i++) s += r[i];
   if (s == 12345.678f) out[0] = s;
}
template <int MODE>
void run(const char* name, float* d, unsigned long long lanes)
{
   hipEvent_t e0, e1;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   const int iters = 8000, w = 6;
   float ms = 0;
   for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, lanes);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
   }
   printf("%-10s lanes %016llx (%2d active): %5.2f ns per wave-instruction per SIMD (6 waves/SIMD)\n", name, lanes,
          __builtin_popcountll(lanes), ms * 1e6 / ((double)iters * 16 * w));
}
int main()
{
   float* d;
   (void)hipMalloc(&d, 4);
   const unsigned long long masks[] = {~0ull, 0x000000000000ffffull, 0x00000000000000ffull, 0x000000000000003full,
                                       0x000000000000001full, 0x000000000000000full, 0x0101010101010101ull,
                                       0x0001000100010001ull, 0x1111111111111111ull, 1ull};
   for (unsigned long long m : masks) run<1>("v_fma_f32", d, m);
   for (unsigned long long m : masks) run<2>("mul+add", d, m);
   for (unsigned long long m : masks) run<3>("alignbit", d, m);
   return 0;
}
