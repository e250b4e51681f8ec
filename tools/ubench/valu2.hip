// microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU forms the SPH
// kernels are built from, at 1 / 2 / 4 / 6 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS8(S)                                                                                  \
   asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)                                          \
                : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),       \
                  "+v"(r[6]), "+v"(r[7])                                                         \
                : "v"(a), "v"(b))
#define OPS8D(S)                                                                                 \
   asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)                                          \
                : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),       \
                  "+v"(d[6]), "+v"(d[7])                                                         \
                : "v"(da), "v"(db))

#define S_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_PKSUB16(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define S_DOT2(i) "v_dot2_i32_i16 %" #i ", %8, %9, %" #i "\n"
#define S_MAD16(i) "v_mad_i32_i16 %" #i ", %8, %9, %" #i " op_sel:[1,1,0,0]\n"
#define S_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define S_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define S_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define S_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define S_CVT64(i) "v_cvt_f32_f64 %8, %" #i "\n"
#define S_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define S_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define S_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define S_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define S_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float fa, float fb)
{
   float r[8];
   double d[8];
   typedef float f2 __attribute__((ext_vector_type(2)));
   f2 p[8];
   for (int i = 0; i < 8; i++) {
      r[i] = threadIdx.x + i;
      d[i] = threadIdx.x + i + 1.0;
      p[i] = f2{(float)threadIdx.x + i, (float)i};
   }
   float a = fa, b = fb;
   double da = fa, db = fb;
   f2 pa = {fa, fa}, pb = {fb, fb};
   for (int it = 0; it < iters; it++) {
      if (MODE == 0) { OPS8(S_FMA); OPS8(S_FMA); }
      if (MODE == 1) {
         asm volatile(S_PKFMA(0) S_PKFMA(1) S_PKFMA(2) S_PKFMA(3) S_PKFMA(4) S_PKFMA(5) S_PKFMA(6) S_PKFMA(7)
                      S_PKFMA(0) S_PKFMA(1) S_PKFMA(2) S_PKFMA(3) S_PKFMA(4) S_PKFMA(5) S_PKFMA(6) S_PKFMA(7)
                      : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                      : "v"(pa), "v"(pb));
      }
      if (MODE == 2) { OPS8(S_PKSUB16); OPS8(S_PKSUB16); }
      if (MODE == 3) { OPS8(S_DOT2); OPS8(S_DOT2); }
      if (MODE == 4) { OPS8(S_MAD16); OPS8(S_MAD16); }
      if (MODE == 5) { OPS8(S_ALIGN); OPS8(S_ALIGN); }
      if (MODE == 6) { OPS8(S_SQRT); OPS8(S_SQRT); }
      if (MODE == 7) { OPS8D(S_FMA64); OPS8D(S_FMA64); }
      if (MODE == 8) { OPS8D(S_RCP64); OPS8D(S_RCP64); }
      if (MODE == 9) { OPS8(S_CNDMASK); OPS8(S_CNDMASK); }
      if (MODE == 10) {
         asm volatile(S_PKADD(0) S_PKADD(1) S_PKADD(2) S_PKADD(3) S_PKADD(4) S_PKADD(5) S_PKADD(6) S_PKADD(7)
                      S_PKMUL(0) S_PKMUL(1) S_PKMUL(2) S_PKMUL(3) S_PKMUL(4) S_PKMUL(5) S_PKMUL(6) S_PKMUL(7)
                      : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                      : "v"(pa), "v"(pb));
      }
      if (MODE == 11) { OPS8(S_LSHLOR); OPS8(S_BFE); }
      if (MODE == 12) {   // the 16-bit screening mix: 3 pk_sub, 2 dot2, 2 mad16, 2 alignbit per candidate pair (x ~1.8)
         OPS8(S_PKSUB16); OPS8(S_DOT2);
      }
   }
   float s = 0;
   for (int i = 0; i < 8; i++) s += r[i] + (float)d[i] + p[i].x + p[i].y;
   if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* dptr)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0);
   hipEventCreate(&e1);
   const int iters = 8000;
   printf("%-28s", name);
   for (int w : {1, 2, 4, 6}) {
      const int blocks = 256 * w;
      float ms = 0;
      for (int rep = 0; rep < 3; rep++) {
         hipEventRecord(e0);
         hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dptr, iters, 1.0001f, 0.5f);
         hipEventRecord(e1);
         hipEventSynchronize(e1);
         hipEventElapsedTime(&ms, e0, e1);
      }
      const double instr_per_simd = (double)iters * 16 * w;
      printf("  %dw: %5.2f ns", w, ms * 1e6 / instr_per_simd);
   }
   printf("   (ns per wave-instruction per SIMD; x clock GHz = cycles)\n");
}

int main()
{
   float* d;
   hipMalloc(&d, 4);
   run<0>("v_fma_f32", d);
   run<1>("v_pk_fma_f32", d);
   run<10>("v_pk_add/mul_f32", d);
   run<2>("v_pk_sub_i16", d);
   run<3>("v_dot2_i32_i16", d);
   run<4>("v_mad_i32_i16 op_sel", d);
   run<5>("v_alignbit_b32", d);
   run<9>("v_cndmask_b32", d);
   run<11>("v_lshl_or / v_bfe", d);
   run<6>("v_sqrt_f32", d);
   run<7>("v_fma_f64", d);
   run<8>("v_rcp_f64", d);
   run<12>("pk_sub_i16 + dot2 mix", d);
   return 0;
}
