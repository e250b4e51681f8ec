// Issue price of the packed 16-bit and bit-gathering instructions a half-precision TEST screen
// would be made of (same method as valu3.hip: every SIMD busy, 8 waves per SIMD, eight independent
// chains per wave, cycles = wall ns x the shader clock held).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu4 tools/ubench/valu4.hip && ./valu4
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define PER_IT 64
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

#define KERNEL(NAME, S)                                                                                    \
   __global__ void __launch_bounds__(256) NAME(unsigned long long* stamps, unsigned* out, int iters,       \
                                               unsigned ua, unsigned ub)                                   \
   {                                                                                                       \
      unsigned u[8];                                                                                       \
      for (int i = 0; i < 8; i++) u[i] = 0x3c003c00u + ((threadIdx.x + i) & 7);                            \
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
      const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                      \
      for (int it = 0; it < iters; it++)                                                                   \
         asm volatile(REP64(S)                                                                             \
                      : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]),            \
                        "+v"(u[6]), "+v"(u[7])                                                             \
                      : "v"(ua), "v"(ub));                                                                 \
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                      \
      if ((threadIdx.x & 63) == 0) {                                                                       \
         const int w = blockIdx.x * 4 + threadIdx.x / 64;                                                  \
         stamps[2 * w + 0] = t1 - t0;                                                                      \
         stamps[2 * w + 1] = r1 - r0;                                                                      \
      }                                                                                                    \
      unsigned s = 0;                                                                                      \
      for (int i = 0; i < 8; i++) s += u[i];                                                               \
      if (s == 0x12345678u) out[0] = s;                                                                    \
   }

#define S_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_PKFMA16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define S_PKFMA16N(i) "v_pk_fma_f16 %" #i ", %8, %8, %" #i " neg_lo:[1,0,0] neg_hi:[1,0,0] clamp\n"
#define S_PKADD16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define S_PKADD16N(i) "v_pk_add_f16 %" #i ", %" #i ", %8 neg_lo:[0,1] neg_hi:[0,1]\n"
#define S_PKMUL16(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
#define S_PKMIN16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define S_FMA16(i) "v_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define S_PKMADU16(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define S_PKMADI16(i) "v_pk_mad_i16 %" #i ", %" #i ", %8, %9\n"
#define S_PKADDU16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define S_PKSUBI16(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define S_PKMULLO16(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define S_PKLSHR16(i) "v_pk_lshrrev_b16 %" #i ", 1, %" #i "\n"
#define S_PKLSHL16(i) "v_pk_lshlrev_b16 %" #i ", 1, %" #i "\n"
#define S_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define S_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define S_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define S_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define S_LSHLREV(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define S_LSHRREV(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define S_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define S_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define S_DOT2F(i) "v_dot2_f32_f16 %" #i ", %8, %9, %" #i "\n"
#define S_DOT2I(i) "v_dot2_i32_i16 %" #i ", %8, %9, %" #i "\n"
#define S_DOT4I(i) "v_dot4_i32_i8 %" #i ", %8, %9, %" #i "\n"
#define S_CVTPK(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8\n"
#define S_CVTPKI(i) "v_cvt_pk_i16_i32 %" #i ", %" #i ", %8\n"
#define S_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define S_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define S_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define S_SUBREV(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define S_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"

KERNEL(k_fma32, S_FMA32)
KERNEL(k_pkfma16, S_PKFMA16)
KERNEL(k_pkfma16n, S_PKFMA16N)
KERNEL(k_pkadd16, S_PKADD16)
KERNEL(k_pkadd16n, S_PKADD16N)
KERNEL(k_pkmul16, S_PKMUL16)
KERNEL(k_pkmin16, S_PKMIN16)
KERNEL(k_fma16, S_FMA16)
KERNEL(k_pkmadu16, S_PKMADU16)
KERNEL(k_pkmadi16, S_PKMADI16)
KERNEL(k_pkaddu16, S_PKADDU16)
KERNEL(k_pksubi16, S_PKSUBI16)
KERNEL(k_pkmullo16, S_PKMULLO16)
KERNEL(k_pklshr16, S_PKLSHR16)
KERNEL(k_pklshl16, S_PKLSHL16)
KERNEL(k_perm, S_PERM)
KERNEL(k_bfi, S_BFI)
KERNEL(k_andor, S_ANDOR)
KERNEL(k_or3, S_OR3)
KERNEL(k_lshladd, S_LSHLADD)
KERNEL(k_lshlrev, S_LSHLREV)
KERNEL(k_lshrrev, S_LSHRREV)
KERNEL(k_or, S_OR)
KERNEL(k_align, S_ALIGN)
KERNEL(k_dot2f, S_DOT2F)
KERNEL(k_dot2i, S_DOT2I)
KERNEL(k_dot4i, S_DOT4I)
KERNEL(k_cvtpk, S_CVTPK)
KERNEL(k_cvtpki, S_CVTPKI)
KERNEL(k_rndne, S_RNDNE)
KERNEL(k_med3, S_MED3)
KERNEL(k_max, S_MAX)
KERNEL(k_bcnt, S_BCNT)

typedef void (*kern_t)(unsigned long long*, unsigned*, int, unsigned, unsigned);

static void run(const char* name, kern_t kern, unsigned long long* dstamps, unsigned* dout)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0);
   hipEventCreate(&e1);
   const int iters = 2000;
   printf("%-34s", name);
   const int WAVES[2] = {4, 8};
   for (int wi = 0; wi < 2; wi++) {
      const int w = WAVES[wi];
      const int blocks = 256 * w;
      float ms = 0;
      std::vector<unsigned long long> st(2 * (size_t)blocks * 4);
      for (int rep = 0; rep < 3; rep++) {
         hipEventRecord(e0);
         hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dstamps, dout, iters, 0x3c003c00u, 0x38003800u);
         hipEventRecord(e1);
         hipEventSynchronize(e1);
         hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(st.data(), dstamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> ghz;
      for (size_t q = 0; q < st.size() / 2; q++)
         if (st[2 * q + 1]) ghz.push_back((double)st[2 * q] / (double)st[2 * q + 1] * 0.1);
      std::sort(ghz.begin(), ghz.end());
      const double ns = ms * 1e6 / ((double)iters * PER_IT * w);
      const double g = ghz[ghz.size() / 2];
      printf(" | %dw %5.2f ns %4.2f cyc @%4.2f GHz", w, ns, ns * g, g);
   }
   printf("\n");
   fflush(stdout);
}

int main()
{
   unsigned long long* dst;
   unsigned* dout;
   hipMalloc(&dst, sizeof(unsigned long long) * 2 * 256 * 8 * 4);
   hipMalloc(&dout, 4);
#define RUN(K) run(#K, K, dst, dout)
   RUN(k_fma32);
   RUN(k_pkfma16);
   RUN(k_pkfma16n);
   RUN(k_pkadd16);
   RUN(k_pkadd16n);
   RUN(k_pkmul16);
   RUN(k_pkmin16);
   RUN(k_fma16);
   RUN(k_pkmadu16);
   RUN(k_pkmadi16);
   RUN(k_pkaddu16);
   RUN(k_pksubi16);
   RUN(k_pkmullo16);
   RUN(k_pklshr16);
   RUN(k_pklshl16);
   RUN(k_perm);
   RUN(k_bfi);
   RUN(k_andor);
   RUN(k_or3);
   RUN(k_lshladd);
   RUN(k_lshlrev);
   RUN(k_lshrrev);
   RUN(k_or);
   RUN(k_align);
   RUN(k_dot2f);
   RUN(k_dot2i);
   RUN(k_dot4i);
   RUN(k_cvtpk);
   RUN(k_cvtpki);
   RUN(k_rndne);
   RUN(k_med3);
   RUN(k_max);
   RUN(k_bcnt);
   return 0;
}
