// Issue price of the remaining VALU opcodes of the SPH pair kernels (everything tools/ubench/valu3.hip
// did not cover), measured the same way: every SIMD of the chip busy, 8 waves per SIMD, eight
// independent chains per wave, 64 instruction slots per loop iteration, shader clock from
// s_memtime / s_memrealtime.  Table-driven: one asm template per opcode.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu5 tools/ubench/valu5.hip && ./valu5 [json]
//
// Feeds tools/valu_census.py (profiles/r4_valu_prices_more.json).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#define PER_IT 64

// 32-bit chains u[0..7], operands ua, ub; 64-bit chains q[0..7] (register pairs), operand qa
#define OPSU(S)                                                                                   \
   asm volatile(REP64(S)                                                                          \
                : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]),        \
                  "+v"(u[6]), "+v"(u[7])                                                         \
                : "v"(ua), "v"(ub)                                                               \
                : "vcc")
#define OPSQ(S)                                                                                   \
   asm volatile(REP64(S)                                                                          \
                : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]),        \
                  "+v"(q[6]), "+v"(q[7])                                                         \
                : "v"(ua), "v"(qa)                                                               \
                : "vcc")

#define T_LSHL_ADD_U64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %9\n"
#define T_MAD_U64_U32(i) "v_mad_u64_u32 %" #i ", vcc, %8, %8, %" #i "\n"
#define T_LSHLREV_B64(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define T_ASHRREV(i) "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define T_LSHRREV(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define T_LSHLREV(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T_SUB_U32(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define T_SUB_F32(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define T_MAX_F32(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define T_MIN_I32(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define T_MAX_I32(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define T_MUL_LO_U32(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define T_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define T_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define T_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T_AND_OR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define T_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x80\n"
#define T_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define T_BFM(i) "v_bfm_b32 %" #i ", %" #i ", %8\n"
#define T_LSHL_ADD_U32(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define T_ADD_LSHL_U32(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 1\n"
#define T_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define T_MAD_U32_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define T_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define T_CVT_I32_F32(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define T_CVT_F32_I32(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define T_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define T_CMP_ONLY(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define T_CMP_F32_ONLY(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define T_CNDMASK_ONLY(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define T_CMP_CLASS(i) "v_cmp_class_f32 vcc, %" #i ", %8\n"
#define T_DIV_SCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define T_DIV_FMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define T_DIV_FIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define T_MBCNT_LO(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define T_READFIRSTLANE(i) "v_readfirstlane_b32 s4, %" #i "\n"
#define T_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define T_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define T_LOG(i) "v_log_f32 %" #i ", %" #i "\n"

enum {
   M_LSHL_ADD_U64, M_MAD_U64_U32, M_LSHLREV_B64, M_ASHRREV, M_LSHRREV, M_LSHLREV, M_SUB_U32, M_SUB_F32, M_MAX_F32,
   M_MIN_I32, M_MAX_I32, M_MUL_LO_U32, M_NOT, M_OR, M_XOR, M_AND_OR, M_BITOP3, M_BCNT, M_BFM, M_LSHL_ADD_U32,
   M_ADD_LSHL_U32, M_ADD3, M_MAD_U32_U24, M_FLOOR, M_CVT_I32_F32, M_CVT_F32_I32, M_FMAC, M_CMP_ONLY, M_CMP_F32_ONLY,
   M_CNDMASK_ONLY, M_CMP_CLASS, M_DIV_SCALE, M_DIV_FMAS, M_DIV_FIXUP, M_MBCNT_LO, M_READFIRSTLANE, M_MED3, M_PERM, M_LOG,
   M_COUNT
};

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* stamps, unsigned* out, int iters, unsigned a, unsigned b)
{
   unsigned u[8];
   unsigned long long q[8];
   for (int i = 0; i < 8; i++) {
      u[i] = __float_as_uint(1.0f + 0.001f * (threadIdx.x + i));
      q[i] = threadIdx.x * 2654435761ull + i;
   }
   unsigned ua = a, ub = b;
   unsigned long long qa = ((unsigned long long)a << 32) | b;
   const unsigned long long t0 = __builtin_amdgcn_s_memtime();
   const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
   for (int it = 0; it < iters; it++) {
      if (MODE == M_LSHL_ADD_U64) OPSQ(T_LSHL_ADD_U64);
      if (MODE == M_MAD_U64_U32) OPSQ(T_MAD_U64_U32);
      if (MODE == M_LSHLREV_B64) OPSQ(T_LSHLREV_B64);
      if (MODE == M_ASHRREV) OPSU(T_ASHRREV);
      if (MODE == M_LSHRREV) OPSU(T_LSHRREV);
      if (MODE == M_LSHLREV) OPSU(T_LSHLREV);
      if (MODE == M_SUB_U32) OPSU(T_SUB_U32);
      if (MODE == M_SUB_F32) OPSU(T_SUB_F32);
      if (MODE == M_MAX_F32) OPSU(T_MAX_F32);
      if (MODE == M_MIN_I32) OPSU(T_MIN_I32);
      if (MODE == M_MAX_I32) OPSU(T_MAX_I32);
      if (MODE == M_MUL_LO_U32) OPSU(T_MUL_LO_U32);
      if (MODE == M_NOT) OPSU(T_NOT);
      if (MODE == M_OR) OPSU(T_OR);
      if (MODE == M_XOR) OPSU(T_XOR);
      if (MODE == M_AND_OR) OPSU(T_AND_OR);
      if (MODE == M_BITOP3) OPSU(T_BITOP3);
      if (MODE == M_BCNT) OPSU(T_BCNT);
      if (MODE == M_BFM) OPSU(T_BFM);
      if (MODE == M_LSHL_ADD_U32) OPSU(T_LSHL_ADD_U32);
      if (MODE == M_ADD_LSHL_U32) OPSU(T_ADD_LSHL_U32);
      if (MODE == M_ADD3) OPSU(T_ADD3);
      if (MODE == M_MAD_U32_U24) OPSU(T_MAD_U32_U24);
      if (MODE == M_FLOOR) OPSU(T_FLOOR);
      if (MODE == M_CVT_I32_F32) OPSU(T_CVT_I32_F32);
      if (MODE == M_CVT_F32_I32) OPSU(T_CVT_F32_I32);
      if (MODE == M_FMAC) OPSU(T_FMAC);
      if (MODE == M_CMP_ONLY) OPSU(T_CMP_ONLY);
      if (MODE == M_CMP_F32_ONLY) OPSU(T_CMP_F32_ONLY);
      if (MODE == M_CNDMASK_ONLY) OPSU(T_CNDMASK_ONLY);
      if (MODE == M_CMP_CLASS) OPSU(T_CMP_CLASS);
      if (MODE == M_DIV_SCALE) OPSU(T_DIV_SCALE);
      if (MODE == M_DIV_FMAS) OPSU(T_DIV_FMAS);
      if (MODE == M_DIV_FIXUP) OPSU(T_DIV_FIXUP);
      if (MODE == M_MBCNT_LO) OPSU(T_MBCNT_LO);
      if (MODE == M_READFIRSTLANE) {
         asm volatile(REP64(T_READFIRSTLANE) : : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]),
                      "v"(u[6]), "v"(u[7]), "v"(ua), "v"(ub) : "s4");
      }
      if (MODE == M_MED3) OPSU(T_MED3);
      if (MODE == M_PERM) OPSU(T_PERM);
      if (MODE == M_LOG) OPSU(T_LOG);
   }
   const unsigned long long t1 = __builtin_amdgcn_s_memtime();
   const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
   if ((threadIdx.x & 63) == 0) {
      const int w = blockIdx.x * 4 + threadIdx.x / 64;
      stamps[2 * w + 0] = t1 - t0;
      stamps[2 * w + 1] = r1 - r0;
   }
   unsigned s = 0;
   for (int i = 0; i < 8; i++) s += u[i] + (unsigned)q[i] + (unsigned)(q[i] >> 32);
   if (s == 0x12345678u) out[0] = s;
}

struct Row {
   const char* name;
   double ns, cyc, ghz;
};
static std::vector<Row> rows;

template <int MODE>
void run(const char* name, unsigned long long* dstamps, unsigned* dout)
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
   Row r;
   r.name = name;
   r.ns = ms * 1e6 / ((double)iters * PER_IT * w);
   r.ghz = ghz[ghz.size() / 2];
   r.cyc = r.ns * r.ghz;
   rows.push_back(r);
   printf("%-24s %5.2f ns  %5.2f cycles  @ %4.2f GHz\n", name, r.ns, r.cyc, r.ghz);
   fflush(stdout);
}

int main(int argc, char** argv)
{
   unsigned long long* dst;
   unsigned* dout;
   hipMalloc(&dst, sizeof(unsigned long long) * 2 * 256 * 8 * 4);
   hipMalloc(&dout, 4);
   printf("8 waves per SIMD, every SIMD busy: wall ns and SIMD cycles per wave-instruction per SIMD\n");
   run<M_LSHL_ADD_U64>("v_lshl_add_u64", dst, dout);
   run<M_MAD_U64_U32>("v_mad_u64_u32", dst, dout);
   run<M_LSHLREV_B64>("v_lshlrev_b64", dst, dout);
   run<M_ASHRREV>("v_ashrrev_i32", dst, dout);
   run<M_LSHRREV>("v_lshrrev_b32", dst, dout);
   run<M_LSHLREV>("v_lshlrev_b32", dst, dout);
   run<M_SUB_U32>("v_sub_u32", dst, dout);
   run<M_SUB_F32>("v_sub_f32", dst, dout);
   run<M_MAX_F32>("v_max_f32", dst, dout);
   run<M_MIN_I32>("v_min_i32", dst, dout);
   run<M_MAX_I32>("v_max_i32", dst, dout);
   run<M_MUL_LO_U32>("v_mul_lo_u32", dst, dout);
   run<M_NOT>("v_not_b32", dst, dout);
   run<M_OR>("v_or_b32", dst, dout);
   run<M_XOR>("v_xor_b32", dst, dout);
   run<M_AND_OR>("v_and_or_b32", dst, dout);
   run<M_BITOP3>("v_bitop3_b32", dst, dout);
   run<M_BCNT>("v_bcnt_u32_b32", dst, dout);
   run<M_BFM>("v_bfm_b32", dst, dout);
   run<M_LSHL_ADD_U32>("v_lshl_add_u32", dst, dout);
   run<M_ADD_LSHL_U32>("v_add_lshl_u32", dst, dout);
   run<M_ADD3>("v_add3_u32", dst, dout);
   run<M_MAD_U32_U24>("v_mad_u32_u24", dst, dout);
   run<M_FLOOR>("v_floor_f32", dst, dout);
   run<M_CVT_I32_F32>("v_cvt_i32_f32", dst, dout);
   run<M_CVT_F32_I32>("v_cvt_f32_i32", dst, dout);
   run<M_FMAC>("v_fmac_f32", dst, dout);
   run<M_CMP_ONLY>("v_cmp_lt_u32", dst, dout);
   run<M_CMP_F32_ONLY>("v_cmp_gt_f32", dst, dout);
   run<M_CNDMASK_ONLY>("v_cndmask_b32", dst, dout);
   run<M_CMP_CLASS>("v_cmp_class_f32", dst, dout);
   run<M_DIV_SCALE>("v_div_scale_f32", dst, dout);
   run<M_DIV_FMAS>("v_div_fmas_f32", dst, dout);
   run<M_DIV_FIXUP>("v_div_fixup_f32", dst, dout);
   run<M_MBCNT_LO>("v_mbcnt_lo_u32_b32", dst, dout);
   run<M_READFIRSTLANE>("v_readfirstlane_b32", dst, dout);
   run<M_MED3>("v_med3_i32", dst, dout);
   run<M_PERM>("v_perm_b32", dst, dout);
   run<M_LOG>("v_log_f32", dst, dout);
   if (argc > 1) {
      FILE* f = fopen(argv[1], "w");
      fprintf(f, "{\n \"source\": \"tools/ubench/valu5.hip on MI355X: 8 waves per SIMD, every SIMD busy, 8 independent chains "
                 "per wave; cycles = wall ns per wave-instruction per SIMD x the shader clock held\",\n \"instructions\": {\n");
      for (size_t i = 0; i < rows.size(); i++)
         fprintf(f, "  \"%s\": {\"ns\": %.3f, \"cycles\": %.3f, \"ghz\": %.3f}%s\n", rows[i].name, rows[i].ns, rows[i].cyc,
                 rows[i].ghz, i + 1 < rows.size() ? "," : "");
      fprintf(f, " }\n}\n");
      fclose(f);
   }
   return 0;
}
