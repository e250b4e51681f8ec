// Do the two price classes of tools/ubench/valu3.hip / valu5.hip add up?  Mixed streams: the same
// harness (every SIMD busy, 8 waves per SIMD, eight independent chains per wave, 64 instruction slots per
// loop iteration), but the eight chains of a wave carry DIFFERENT opcodes - plain-class (v_fma_f32,
// v_add_u32: 2.2 cycles alone) and wide-class (v_alignbit_b32, v_lshlrev_b32, v_max_f32, v_pk_fma_f32:
// 4.2 alone) in the ratios 8:0, 6:2, 4:4, 2:6, 0:8.  If the classes shared one port the cost of a mix
// would be the sum of its parts (the additive price of tools/valu_census.py); what is measured is
// printed next to that sum and next to max(plain part, wide part).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu6 tools/ubench/valu6.hip && ./valu6 [json]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define PER_IT 64
#define P_FMA(i) "v_fma_f32 %" #i ", %" #i ", %12, %13\n"
#define P_ADD(i) "v_add_u32 %" #i ", %" #i ", %12\n"
#define W_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %12, 7\n"
#define W_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define W_MAX(i) "v_max_f32 %" #i ", %" #i ", %12\n"
#define W_PK(i) "v_pk_fma_f32 %" #i ", %" #i ", %14, %" #i "\n"
// chains: %0..%7 = u[0..7] (32-bit), %8..%11 = q[0..3] (64-bit pairs); %12 = ua, %13 = ub, %14 = qa
#define ROW8(A, B, C, D, E, F, G, H) A(0) B(1) C(2) D(3) E(4) F(5) G(6) H(7)
#define ASM64(ROW)                                                                                          \
   asm volatile(ROW ROW ROW ROW ROW ROW ROW ROW                                                             \
                : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]),      \
                  "+v"(u[7]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3])                               \
                : "v"(ua), "v"(ub), "v"(qa)                                                                \
                : "vcc")
// pk rows: the packed op runs on the q chains (slots 8..11)
#define PKROW_4_4 P_FMA(0) W_PK(8) P_FMA(1) W_PK(9) P_FMA(2) W_PK(10) P_FMA(3) W_PK(11)
#define PKROW_0_8 W_PK(8) W_PK(9) W_PK(10) W_PK(11) W_PK(8) W_PK(9) W_PK(10) W_PK(11)
#define PKROW_6_2 P_FMA(0) P_FMA(1) P_FMA(2) W_PK(8) P_FMA(3) P_FMA(4) P_FMA(5) W_PK(9)
#define PKROW_2_6 P_FMA(0) W_PK(8) W_PK(9) W_PK(10) P_FMA(1) W_PK(11) W_PK(8) W_PK(9)

// 4:4 mixes of one partner (v_fma_f32 or v_add_u32, chains 0 2 4 6) with one other opcode (chains 1 3 5 7)
#define X_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %12, vcc\n"
#define X_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %12\n"
#define X_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %12\n"
#define X_BFM(i) "v_bfm_b32 %" #i ", %" #i ", %12\n"
#define X_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %12\n"
#define X_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %12, %13\n"
#define X_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %12, %13\n"
#define X_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %12\n"
#define X_AND_OR(i) "v_and_or_b32 %" #i ", %" #i ", %12, %13\n"
#define X_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define X_PERM(i) "v_perm_b32 %" #i ", %" #i ", %12, %13\n"
#define X_CVT(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define X_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define X_MAX_I32(i) "v_max_i32 %" #i ", %" #i ", %12\n"
#define X_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define X_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define X_BFI(i) "v_bfi_b32 %" #i ", %12, %" #i ", %13\n"
#define X_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define X_SUB_F32(i) "v_sub_f32 %" #i ", %" #i ", %12\n"
#define X_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %12\n"
#define PAIRROW(P, X) P(0) X(1) P(2) X(3) P(4) X(5) P(6) X(7)
#define PKPAIR(P, OP) P(0) OP(8) P(1) OP(9) P(2) OP(10) P(3) OP(11)
#define W_PK_ADD(i) "v_pk_add_f32 %" #i ", %" #i ", %14\n"
#define W_PK_MUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %14\n"
#define PAIR_LIST(F)                                                                                         \
   F(CNDMASK, X_CNDMASK) F(CMP, X_CMP) F(BCNT, X_BCNT) F(BFM, X_BFM) F(LSHL_ADD, X_LSHL_ADD) F(ADD3, X_ADD3)      \
   F(MAD24, X_MAD24) F(MUL_LO, X_MUL_LO) F(AND_OR, X_AND_OR) F(FFBL, X_FFBL) F(PERM, X_PERM) F(CVT, X_CVT)       \
   F(FLOOR, X_FLOOR) F(MAX_I32, X_MAX_I32) F(RSQ, X_RSQ) F(RCP, X_RCP) F(BFI, X_BFI) F(LSHR, X_LSHR)            \
   F(SUB_F32, X_SUB_F32) F(MUL_F32, X_MUL_F32) F(ALIGN, W_ALIGN) F(LSHL, W_LSHL) F(MAXF, W_MAX)
#define ENUM_F(N, X) M_F_##N,
#define ENUM_A(N, X) M_A_##N,
enum { M_P8, M_W8, M_44, M_62, M_26, M_44_ADD_LSHL, M_44_FMA_MAX, M_PP, M_WW, M_PK08, M_PK44, M_PK62, M_PK26,
       PAIR_LIST(ENUM_F) PAIR_LIST(ENUM_A) M_F_PK_ADD, M_F_PK_MUL, M_A_PK_FMA, M_COUNT };

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* stamps, unsigned* out, int iters, unsigned a, unsigned b)
{
   unsigned u[8];
   unsigned long long q[4];
   for (int i = 0; i < 8; i++) u[i] = __float_as_uint(1.0f + 0.001f * (threadIdx.x + i));
   for (int i = 0; i < 4; i++) q[i] = ((unsigned long long)u[i] << 32) | u[i + 4];
   unsigned ua = a, ub = b;
   unsigned long long qa = ((unsigned long long)a << 32) | a;
   const unsigned long long t0 = __builtin_amdgcn_s_memtime();
   const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
   for (int it = 0; it < iters; it++) {
      if (MODE == M_P8) ASM64(ROW8(P_FMA, P_FMA, P_FMA, P_FMA, P_FMA, P_FMA, P_FMA, P_FMA));
      if (MODE == M_W8) ASM64(ROW8(W_ALIGN, W_ALIGN, W_ALIGN, W_ALIGN, W_ALIGN, W_ALIGN, W_ALIGN, W_ALIGN));
      if (MODE == M_44) ASM64(ROW8(P_FMA, W_ALIGN, P_FMA, W_ALIGN, P_FMA, W_ALIGN, P_FMA, W_ALIGN));
      if (MODE == M_62) ASM64(ROW8(P_FMA, P_FMA, P_FMA, W_ALIGN, P_FMA, P_FMA, P_FMA, W_ALIGN));
      if (MODE == M_26) ASM64(ROW8(P_FMA, W_ALIGN, W_ALIGN, W_ALIGN, P_FMA, W_ALIGN, W_ALIGN, W_ALIGN));
      if (MODE == M_44_ADD_LSHL) ASM64(ROW8(P_ADD, W_LSHL, P_ADD, W_LSHL, P_ADD, W_LSHL, P_ADD, W_LSHL));
      if (MODE == M_44_FMA_MAX) ASM64(ROW8(P_FMA, W_MAX, P_FMA, W_MAX, P_FMA, W_MAX, P_FMA, W_MAX));
      if (MODE == M_PP) ASM64(ROW8(P_FMA, P_ADD, P_FMA, P_ADD, P_FMA, P_ADD, P_FMA, P_ADD));
      if (MODE == M_WW) ASM64(ROW8(W_ALIGN, W_LSHL, W_ALIGN, W_LSHL, W_ALIGN, W_LSHL, W_ALIGN, W_LSHL));
      if (MODE == M_PK08) ASM64(PKROW_0_8);
      if (MODE == M_PK44) ASM64(PKROW_4_4);
      if (MODE == M_PK62) ASM64(PKROW_6_2);
      if (MODE == M_PK26) ASM64(PKROW_2_6);
#define RUN_F(N, X) if (MODE == M_F_##N) ASM64(PAIRROW(P_FMA, X));
#define RUN_A(N, X) if (MODE == M_A_##N) ASM64(PAIRROW(P_ADD, X));
      PAIR_LIST(RUN_F)
      PAIR_LIST(RUN_A)
      if (MODE == M_F_PK_ADD) ASM64(PKPAIR(P_FMA, W_PK_ADD));
      if (MODE == M_F_PK_MUL) ASM64(PKPAIR(P_FMA, W_PK_MUL));
      if (MODE == M_A_PK_FMA) ASM64(PKPAIR(P_ADD, W_PK));
   }
   const unsigned long long t1 = __builtin_amdgcn_s_memtime();
   const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
   if ((threadIdx.x & 63) == 0) {
      const int w = blockIdx.x * 4 + threadIdx.x / 64;
      stamps[2 * w + 0] = t1 - t0;
      stamps[2 * w + 1] = r1 - r0;
   }
   unsigned s = 0;
   for (int i = 0; i < 8; i++) s += u[i];
   for (int i = 0; i < 4; i++) s += (unsigned)q[i] + (unsigned)(q[i] >> 32);
   if (s == 0x12345678u) out[0] = s;
}

struct Row {
   const char* name;
   int plain, wide;
   double cyc;
};
static std::vector<Row> rows;

template <int MODE>
void run(const char* name, int plain, int wide, unsigned long long* dstamps, unsigned* dout)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0);
   hipEventCreate(&e1);
   const int iters = 1000, w = 8, blocks = 256 * w;
   float ms = 0;
   std::vector<unsigned long long> st(2 * (size_t)blocks * 4);
   for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dstamps, dout, iters, 0x3f800100u, 0x3f000000u);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
   }
   hipMemcpy(st.data(), dstamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
   std::vector<double> ghz;
   for (size_t i = 0; i < st.size() / 2; i++)
      if (st[2 * i + 1]) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);
   std::sort(ghz.begin(), ghz.end());
   const double g = ghz[ghz.size() / 2];
   Row r = {name, plain, wide, ms * 1e6 / ((double)iters * PER_IT * w) * g};
   rows.push_back(r);
}

int main(int argc, char** argv)
{
   unsigned long long* dst;
   unsigned* dout;
   hipMalloc(&dst, sizeof(unsigned long long) * 2 * 256 * 8 * 4);
   hipMalloc(&dout, 4);
   run<M_P8>("v_fma_f32 alone", 8, 0, dst, dout);
   run<M_W8>("v_alignbit_b32 alone", 0, 8, dst, dout);
   run<M_PK08>("v_pk_fma_f32 alone", 0, 8, dst, dout);
   run<M_62>("6 v_fma_f32 : 2 v_alignbit_b32", 6, 2, dst, dout);
   run<M_44>("4 v_fma_f32 : 4 v_alignbit_b32", 4, 4, dst, dout);
   run<M_26>("2 v_fma_f32 : 6 v_alignbit_b32", 2, 6, dst, dout);
   run<M_44_ADD_LSHL>("4 v_add_u32 : 4 v_lshlrev_b32", 4, 4, dst, dout);
   run<M_44_FMA_MAX>("4 v_fma_f32 : 4 v_max_f32", 4, 4, dst, dout);
   run<M_PK62>("6 v_fma_f32 : 2 v_pk_fma_f32", 6, 2, dst, dout);
   run<M_PK44>("4 v_fma_f32 : 4 v_pk_fma_f32", 4, 4, dst, dout);
   run<M_PK26>("2 v_fma_f32 : 6 v_pk_fma_f32", 2, 6, dst, dout);
   run<M_PP>("4 v_fma_f32 : 4 v_add_u32 (both plain)", 8, 0, dst, dout);
   run<M_WW>("4 v_alignbit_b32 : 4 v_lshlrev_b32 (both wide)", 0, 8, dst, dout);
   const size_t n_mix = rows.size();
#define CALL_F(N, X) run<M_F_##N>("4 v_fma_f32 : 4 " #N, 4, 4, dst, dout);
#define CALL_A(N, X) run<M_A_##N>("4 v_add_u32 : 4 " #N, 4, 4, dst, dout);
   PAIR_LIST(CALL_F)
   run<M_F_PK_ADD>("4 v_fma_f32 : 4 PK_ADD", 4, 4, dst, dout);
   run<M_F_PK_MUL>("4 v_fma_f32 : 4 PK_MUL", 4, 4, dst, dout);
   PAIR_LIST(CALL_A)
   run<M_A_PK_FMA>("4 v_add_u32 : 4 PK_FMA", 4, 4, dst, dout);
   const double p = rows[0].cyc, w = rows[1].cyc;
   printf("cycles per wave-instruction per SIMD, 8 waves per SIMD, every SIMD busy (plain alone %.2f, wide alone %.2f)\n", p, w);
   printf("%-48s %9s %9s %9s %9s\n", "stream", "measured", "additive", "max", "max+k*min");
   // k fitted on the 4:4 fma/alignbit row
   const double k = (rows[4].cyc * 8 - 4 * w) / (4 * p);
   for (size_t i = 0; i < n_mix; i++) {
      const Row& r = rows[i];
      const double P = r.plain * p, W = r.wide * w;
      printf("%-48s %9.2f %9.2f %9.2f %9.2f\n", r.name, r.cyc, (P + W) / 8, std::max(P, W) / 8,
             (std::max(P, W) + k * std::min(P, W)) / 8);
   }
   printf("\npairs, four chains each (cycles per wave-instruction of the MIX; the partner alone costs %.2f):\n", p);
   for (size_t i = n_mix; i < rows.size(); i++) printf("%-48s %9.2f   => the other opcode in this mix: %5.2f\n", rows[i].name, rows[i].cyc, 2 * rows[i].cyc - p);
   printf("k (share of the smaller class that does not overlap, from the 4:4 v_fma_f32 / v_alignbit_b32 row) = %.3f\n", k);
   if (argc > 1) {
      FILE* f = fopen(argv[1], "w");
      fprintf(f, "{\"source\": \"tools/ubench/valu6.hip\", \"plain_alone\": %.4f, \"wide_alone\": %.4f, \"overlap_k\": %.4f, \"streams\": {", p, w, k);
      for (size_t i = 0; i < rows.size(); i++)
         fprintf(f, "%s\"%s\": {\"plain\": %d, \"wide\": %d, \"cycles\": %.4f}", i ? ", " : "", rows[i].name, rows[i].plain, rows[i].wide, rows[i].cyc);
      fprintf(f, "}}\n");
      fclose(f);
   }
   return 0;
}
