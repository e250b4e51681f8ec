// Issue price of the VALU instruction classes the SPH pair kernels are made of, SATURATED:
// 1 / 2 / 4 / 8 waves per SIMD on every CU, eight independent dependency chains per wave, and the
// shader clock the chip actually held during each run (s_memtime ticks per s_memrealtime tick,
// 100 MHz), so that the figures are cycles and not nanoseconds at an assumed clock.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu3 tools/ubench/valu3.hip && ./valu3 [json]
//
// Round-2 verdict, item 6: roofline.valu charged a flat 4 cycles per VALU wave-instruction, while
// MI355X_MICROARCH.md gives v_fma_f32 2 cycles once two or more waves share a SIMD, and
// tools/ubench/valu2.hip had not saturated (still falling at 6 waves).  This table replaces the
// flat figure (bench.py reads profiles/r3_valu_prices.json).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define PER_IT 64   // instruction slots per loop iteration (the loop's own 3 scalar instructions then weigh < 2 %)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#define OPS(S)                                                                                   \
   asm volatile(REP64(S)                                                                          \
                : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]),       \
                  "+v"(r[6]), "+v"(r[7])                                                         \
                : "v"(a), "v"(b))
#define OPSU(S)                                                                                  \
   asm volatile(REP64(S)                                                                          \
                : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]),       \
                  "+v"(u[6]), "+v"(u[7])                                                         \
                : "v"(ua), "v"(ub))
#define OPSD(S)                                                                                  \
   asm volatile(REP64(S)                                                                          \
                : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),       \
                  "+v"(d[6]), "+v"(d[7])                                                         \
                : "v"(da), "v"(db))
#define OPSP(S)                                                                                  \
   asm volatile(REP64(S)                                                                          \
                : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]),       \
                  "+v"(p[6]), "+v"(p[7])                                                         \
                : "v"(pa), "v"(pb))

#define S_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define S_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define S_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define S_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define S_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define S_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define S_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define S_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 29\n"
#define S_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define S_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define S_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define S_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define S_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define S_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define S_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define S_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define S_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define S_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define S_MOV(i) "v_mov_b32 %" #i ", %8\n"

enum { M_FMA, M_MUL, M_ADD, M_PKFMA, M_PKADD, M_PKMUL, M_ADDU, M_AND, M_LSHLOR, M_BFE, M_ALIGN, M_FFBL,
       M_CMPCND, M_RCP, M_RSQ, M_SQRT, M_FMA64, M_MUL64, M_ADD64, M_RCP64, M_MOV, M_MIX, M_COUNT };

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* stamps, float* out, int iters, float fa, float fb)
{
   float r[8];
   unsigned u[8];
   double d[8];
   typedef float f2 __attribute__((ext_vector_type(2)));
   f2 p[8];
   for (int i = 0; i < 8; i++) {
      r[i] = 1.0f + 0.001f * (threadIdx.x + i);
      u[i] = threadIdx.x * 2654435761u + i;
      d[i] = 1.0 + 0.001 * (threadIdx.x + i);
      p[i] = f2{1.0f + 0.001f * threadIdx.x, 1.0f + 0.002f * i};
   }
   float a = fa, b = fb;
   unsigned ua = __float_as_uint(fa), ub = __float_as_uint(fb);
   double da = fa, db = fb;
   f2 pa = {fa, fa}, pb = {fb, fb};
   const unsigned long long t0 = __builtin_amdgcn_s_memtime();
   const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
   for (int it = 0; it < iters; it++) {
      if (MODE == M_FMA) OPS(S_FMA);
      if (MODE == M_MUL) OPS(S_MUL);
      if (MODE == M_ADD) OPS(S_ADD);
      if (MODE == M_PKFMA) OPSP(S_PKFMA);
      if (MODE == M_PKADD) OPSP(S_PKADD);
      if (MODE == M_PKMUL) OPSP(S_PKMUL);
      if (MODE == M_ADDU) OPSU(S_ADDU);
      if (MODE == M_AND) OPSU(S_AND);
      if (MODE == M_LSHLOR) OPSU(S_LSHLOR);
      if (MODE == M_BFE) OPSU(S_BFE);
      if (MODE == M_ALIGN) OPSU(S_ALIGN);
      if (MODE == M_FFBL) OPSU(S_FFBL);
      if (MODE == M_CMPCND) OPSU(S_CMPCND);     // 2 instructions per slot
      if (MODE == M_RCP) OPS(S_RCP);
      if (MODE == M_RSQ) OPS(S_RSQ);
      if (MODE == M_SQRT) OPS(S_SQRT);
      if (MODE == M_FMA64) OPSD(S_FMA64);
      if (MODE == M_MUL64) OPSD(S_MUL64);
      if (MODE == M_ADD64) OPSD(S_ADD64);
      if (MODE == M_RCP64) OPSD(S_RCP64);
      if (MODE == M_MOV) OPS(S_MOV);
      if (MODE == M_MIX) {   // alternating fp32 / packed fp32 / integer, as the TEST step interleaves them
         for (int q = 0; q < 4; q++) {
            asm volatile(S_FMA(0) S_FMA(1) S_FMA(2) S_FMA(3) S_FMA(4) S_FMA(5) S_FMA(6) S_FMA(7)
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                         : "v"(a), "v"(b));
            asm volatile(S_ALIGN(0) S_ALIGN(1) S_ALIGN(2) S_ALIGN(3) S_ALIGN(4) S_ALIGN(5) S_ALIGN(6) S_ALIGN(7)
                         : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                         : "v"(ua), "v"(ub));
         }
      }
   }
   const unsigned long long t1 = __builtin_amdgcn_s_memtime();
   const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
   if ((threadIdx.x & 63) == 0) {
      const int w = blockIdx.x * 4 + threadIdx.x / 64;
      stamps[2 * w + 0] = t1 - t0;
      stamps[2 * w + 1] = r1 - r0;
   }
   float s = 0;
   for (int i = 0; i < 8; i++) s += r[i] + (float)d[i] + p[i].x + p[i].y + __uint_as_float(u[i]);
   if (s == 12345.678f) out[0] = s;
}

struct Row {
   const char* name;
   const char* cls;
   double ns[4], cyc[4], ghz[4];
};
static std::vector<Row> rows;
static const int WAVES[4] = {1, 2, 4, 8};

template <int MODE>
void run(const char* name, const char* cls, unsigned long long* dstamps, float* dout, int per_slot = 1)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0);
   hipEventCreate(&e1);
   const int iters = 2000;
   Row row;
   row.name = name;
   row.cls = cls;
   for (int wi = 0; wi < 4; wi++) {
      const int w = WAVES[wi];
      const int blocks = 256 * w;      // 256-thread workgroups: one wave per SIMD each, w per CU
      float ms = 0;
      std::vector<unsigned long long> st(2 * (size_t)blocks * 4);
      for (int rep = 0; rep < 3; rep++) {
         hipEventRecord(e0);
         hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dstamps, dout, iters, 1.0001f, 0.5f);
         hipEventRecord(e1);
         hipEventSynchronize(e1);
         hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(st.data(), dstamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> ghz, cyc;
      for (size_t q = 0; q < st.size() / 2; q++) {
         if (st[2 * q + 1] == 0) continue;
         ghz.push_back((double)st[2 * q] / (double)st[2 * q + 1] * 0.1);
         // cycles this wave lived / instructions it issued, x waves sharing the SIMD = SIMD cycles per instruction
         cyc.push_back((double)st[2 * q] / ((double)iters * PER_IT * per_slot) / w);
      }
      std::sort(ghz.begin(), ghz.end());
      std::sort(cyc.begin(), cyc.end());
      row.ns[wi] = ms * 1e6 / ((double)iters * PER_IT * per_slot * w);
      row.ghz[wi] = ghz[ghz.size() / 2];
      row.cyc[wi] = row.ns[wi] * row.ghz[wi];      // wall time per instruction x the clock held
      (void)cyc;
   }
   rows.push_back(row);
   printf("%-26s", name);
   for (int wi = 0; wi < 4; wi++) printf(" | %dw %5.2f ns %4.2f cyc @%4.2f GHz", WAVES[wi], row.ns[wi], row.cyc[wi], row.ghz[wi]);
   printf("\n");
   fflush(stdout);
}

int main(int argc, char** argv)
{
   unsigned long long* dst;
   float* dout;
   hipMalloc(&dst, sizeof(unsigned long long) * 2 * 256 * 8 * 4);
   hipMalloc(&dout, 4);
   printf("ns = wall time per wave-instruction per SIMD (all 1024 SIMDs busy); cyc = ns x the shader clock held\n");
   run<M_FMA>("v_fma_f32", "fma_f32", dst, dout);
   run<M_MUL>("v_mul_f32", "mul_f32", dst, dout);
   run<M_ADD>("v_add_f32", "add_f32", dst, dout);
   run<M_PKFMA>("v_pk_fma_f32", "pk_f32", dst, dout);
   run<M_PKADD>("v_pk_add_f32", "pk_f32", dst, dout);
   run<M_PKMUL>("v_pk_mul_f32", "pk_f32", dst, dout);
   run<M_ADDU>("v_add_u32", "int32_simple", dst, dout);
   run<M_AND>("v_and_b32", "int32_simple", dst, dout);
   run<M_MOV>("v_mov_b32", "int32_simple", dst, dout);
   run<M_LSHLOR>("v_lshl_or_b32", "int32_vop3", dst, dout);
   run<M_BFE>("v_bfe_u32", "int32_vop3", dst, dout);
   run<M_ALIGN>("v_alignbit_b32", "int32_vop3", dst, dout);
   run<M_FFBL>("v_ffbl_b32", "int32_vop3", dst, dout);
   run<M_CMPCND>("v_cmp + v_cndmask", "int32_vop3", dst, dout, 2);
   run<M_RCP>("v_rcp_f32", "trans_f32", dst, dout);
   run<M_RSQ>("v_rsq_f32", "trans_f32", dst, dout);
   run<M_SQRT>("v_sqrt_f32", "trans_f32", dst, dout);
   run<M_FMA64>("v_fma_f64", "fma_f64", dst, dout);
   run<M_MUL64>("v_mul_f64", "mul_f64", dst, dout);
   run<M_ADD64>("v_add_f64", "add_f64", dst, dout);
   run<M_RCP64>("v_rcp_f64", "trans_f64", dst, dout);
   run<M_MIX>("v_fma_f32 + v_alignbit mix", "mix", dst, dout);
   if (argc > 1) {
      FILE* f = fopen(argv[1], "w");
      fprintf(f, "{\n \"source\": \"tools/ubench/valu3.hip on MI355X: wall ns and SIMD cycles per wave-instruction per SIMD, "
                 "every SIMD of the chip busy, 8 independent chains per wave, at 1/2/4/8 waves per SIMD; cycles = ns x the "
                 "shader clock held during the run (s_memtime / s_memrealtime)\",\n \"waves_per_simd\": [1, 2, 4, 8],\n \"instructions\": {\n");
      for (size_t i = 0; i < rows.size(); i++) {
         const Row& r = rows[i];
         fprintf(f, "  \"%s\": {\"class\": \"%s\", \"ns\": [%.3f, %.3f, %.3f, %.3f], \"cycles\": [%.3f, %.3f, %.3f, %.3f], "
                    "\"ghz\": [%.3f, %.3f, %.3f, %.3f]}%s\n",
                 r.name, r.cls, r.ns[0], r.ns[1], r.ns[2], r.ns[3], r.cyc[0], r.cyc[1], r.cyc[2], r.cyc[3], r.ghz[0],
                 r.ghz[1], r.ghz[2], r.ghz[3], i + 1 < rows.size() ? "," : "");
      }
      fprintf(f, " }\n}\n");
      fclose(f);
   }
   return 0;
}
