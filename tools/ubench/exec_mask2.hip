// microbenchmark: issue cost of VALU instruction forms under full and sparse EXEC masks
// (finding: several VOP3-encoded forms cost 3-4x more once few lanes are enabled; VOP2 forms do not)
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS8(S)                                                                                  \
   asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)                                          \
                : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),       \
                  "+v"(r[6]), "+v"(r[7])                                                         \
                : "v"(a), "v"(b) : "vcc")
#define OPS8L(S)                                                                                 \
   asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)                                          \
                : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]),       \
                  "+v"(q[6]), "+v"(q[7])                                                         \
                : "v"(qa), "v"(qb) : "vcc")
#define T0(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T1(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define T2(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define T3(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define T4(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define T5(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T6(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define T7(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define T8(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define T9(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %8\n"
#define T10(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T11(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define T12(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define T13(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define T14(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define T15(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define T16(i) "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8\n"
#define T17(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define T18(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float fa, float fb, unsigned long long lanes)
{
   float r[8];
   double q[8];
   for (int i = 0; i < 8; i++) { r[i] = threadIdx.x + i + 1; q[i] = threadIdx.x + i + 1; }
   float a = fa, b = fb;
   double qa = fa, qb = fb;
   if ((lanes >> (threadIdx.x & 63)) & 1ull) {
      for (int it = 0; it < iters; it++) {
#define CASE(n, T) if (MODE == n) { OPS8(T); OPS8(T); }
         CASE(0, T0) CASE(1, T1) CASE(2, T2) CASE(3, T3) CASE(4, T4) CASE(5, T5) CASE(6, T6) CASE(7, T7)
         CASE(8, T8) CASE(11, T11) CASE(12, T12) CASE(13, T13) CASE(14, T14) CASE(15, T15) CASE(16, T16)
         if (MODE == 9) { OPS8L(T9); OPS8L(T9); }
         if (MODE == 10) { OPS8L(T10); OPS8L(T10); }
         if (MODE == 17) { OPS8L(T17); OPS8L(T17); }
         if (MODE == 18) { OPS8L(T18); OPS8L(T18); }
      }
   }
   float s = 0;
   for (int i = 0; i < 8; i++) s += r[i] + (float)q[i];
   if (s == 12345.678f) out[0] = s;
}
template <int MODE>
void run(const char* name, float* d)
{
   hipEvent_t e0, e1;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   const int iters = 4000, w = 6;
   const unsigned long long masks[] = {~0ull, 0x0000ffffffffffffull, 0x00000000ffffffffull, 0x5555555555555555ull,
                                       0x0000000000ffffffull, 0x000000000000ffffull, 0x1111111111111111ull,
                                       0x00000000000000ffull, 0x0101010101010101ull, 0x0001000100010001ull, 1ull};
   printf("%-18s", name);
   for (unsigned long long m : masks) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
         (void)hipEventRecord(e0);
         hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, m);
         (void)hipEventRecord(e1);
         (void)hipEventSynchronize(e1);
         (void)hipEventElapsedTime(&ms, e0, e1);
      }
      printf(" %5.2f", ms * 1e6 / ((double)iters * 16 * w));
   }
   printf("\n");
}
int main()
{
   float* d;
   (void)hipMalloc(&d, 4);
   printf("ns per wave-instruction per SIMD at 6 waves/SIMD; columns = EXEC masks:\n");
   printf("%-18s  all64  lo48  lo32 alt32  lo24  lo16 ev16   lo8   ev8   ev4  one\n", "");
   run<0>("v_fma_f32", d); run<1>("v_fmac_f32 (VOP2)", d); run<11>("v_sub_f32 (VOP2)", d); run<10>("v_pk_fma_f32", d);
   run<12>("v_rsq_f32", d); run<17>("v_add_f64", d); run<18>("v_fma_f64", d);
   run<2>("v_alignbit_b32", d); run<3>("v_lshl_or_b32", d); run<15>("v_bfe_u32", d); run<13>("v_mad_u32_u24", d);
   run<14>("v_add3_u32", d); run<9>("v_lshl_add_u64", d);
   run<4>("v_and_b32 (VOP2)", d); run<5>("v_lshlrev (VOP2)", d); run<6>("v_ffbl_b32 (VOP1)", d);
   run<7>("v_cndmask (VOP2)", d); run<8>("v_cmp_lt_u32 VOPC", d); run<16>("v_cmp_lt_u32 e64", d);
   return 0;
}
