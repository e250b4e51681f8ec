// FULL mode: all in-radius neighbours on the cell-sorted SoA state.
//
// Particle state is {x,y,z,m} / {vx,vy,vz,id} float4 pairs in cell-sorted order (ascending
// cell id, ascending persistent id inside a cell), so the canonical neighbour order of
// particle p is simply "ascending sorted position": for each of the 9 (dz,dy) rows around p's
// cell, the three cells x-1..x+1 are one contiguous range of the sorted arrays.
//
// The per-pair arithmetic is the reference's computeDensity / computeAcceleration
// (src/sph.cpp:721-766, 778-934) via pair_math.h; the acceptance test is the one in
// findNeighbors (src/sph.cpp:641,653): (dx*dx + dy*dy) + dz*dz < mH2, self excluded.
#pragma once

#include "pair_math.h"

// The 9 contiguous candidate ranges of a particle in cell (cx,cy,cz).
struct RowRanges {
   uint32_t s[9], e[9];
};

__device__ __forceinline__ void row_ranges(const CellGrid& g, const uint32_t* __restrict__ cell_start,
                                           int cx, int cy, int cz, RowRanges& r)
{
   const int x0 = cx - 1 < 0 ? 0 : cx - 1;
   const int x1 = cx + 1 >= g.nx ? g.nx - 1 : cx + 1;
#pragma unroll
   for (int k = 0; k < 9; k++) {
      const int z = cz + k / 3 - 1;
      const int y = cy + k % 3 - 1;
      if (z < 0 || z >= g.nz || y < 0 || y >= g.ny) {
         r.s[k] = 0;
         r.e[k] = 0;
      } else {
         const int row = (z * g.ny + y) * g.nx;
         r.s[k] = cell_start[row + x0];
         r.e[k] = cell_start[row + x1 + 1];
      }
   }
}

// ---- density, untiled: one thread per particle, candidates read through L1/L2 -------------------
// Used for workgroups whose LDS tile or neighbour lists would overflow (called inline from the
// tiled kernels of full_tiled.h), for SPH_HIP_UNTILED=1, and as an independent cross-check.
template <bool UNIT_SCALE, bool FAST = false>
__device__ __forceinline__ void density_untiled(int p, const float4* __restrict__ posm,
                                                const uint32_t* __restrict__ cell_start,
                                                const float4* __restrict__ velp, const CellGrid& g,
                                                const PairConsts& k, float* __restrict__ rho,
                                                float4* __restrict__ velB, float* __restrict__ auxc,
                                                int32_t* __restrict__ ncount)
{
   const float4 pi = posm[p];
   int cx, cy, cz;
   cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
   RowRanges r;
   row_ranges(g, cell_start, cx, cy, cz, r);

   float density = 0.0f;
   int count = 0;
#pragma unroll 1
   for (int row = 0; row < 9; row++) {
      const uint32_t s = row == 0 ? r.s[0] : row == 1 ? r.s[1] : row == 2 ? r.s[2] : row == 3 ? r.s[3]
                       : row == 4 ? r.s[4] : row == 5 ? r.s[5] : row == 6 ? r.s[6] : row == 7 ? r.s[7] : r.s[8];
      const uint32_t e = row == 0 ? r.e[0] : row == 1 ? r.e[1] : row == 2 ? r.e[2] : row == 3 ? r.e[3]
                       : row == 4 ? r.e[4] : row == 5 ? r.e[5] : row == 6 ? r.e[6] : row == 7 ? r.e[7] : r.e[8];
      for (uint32_t q = s; q < e; q++) {
         if (q == (uint32_t)p) continue;
         const float4 pj = posm[q];
         float dx, dy, dz;
         const float d2 = dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx, dy, dz);
         if (d2 < k.h2) {
            {
               float d = sqrtf(d2);
               if (!UNIT_SCALE) d *= k.sim_scale;
               density_accumulate<UNIT_SCALE>(k, pj.w, d, density);
            }
            count++;
         }
      }
   }
   rho[p] = density;
   const float2 bc = FAST ? neighbor_terms_fast(k, density, pi.w) : neighbor_terms(k, density, pi.w);
   const float4 v = velp[p];
   // (FAST: the two factors change places - {v, C} is what only the viscous sum's last few
   // neighbours need, m B what every pair needs: see visc_keep)
   velB[p] = make_float4(v.x, v.y, v.z, FAST ? bc.y : bc.x);
   auxc[p] = FAST ? bc.x : bc.y;
   ncount[p] = count;
}

template <bool UNIT_SCALE, bool FAST>
__global__ void __launch_bounds__(256)
k_full_density(const float4* __restrict__ posm, const uint32_t* __restrict__ cell_start,
               const float4* __restrict__ velp, const int32_t* __restrict__ meta, CellGrid g,
               PairConsts k, float* __restrict__ rho, float4* __restrict__ velB,
               float* __restrict__ auxc, int32_t* __restrict__ ncount)
{
   const int p = meta[META_SUM_BEGIN] + blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= meta[META_SUM_END]) return;
   density_untiled<UNIT_SCALE, FAST>(p, posm, cell_start, velp, g, k, rho, velB, auxc, ncount);
}

// ---- acceleration, untiled ------------------------------------------------------------------------
template <bool UNIT_SCALE, bool FAST = false>
__device__ __forceinline__ void accel_untiled(int p, const float4* __restrict__ posm,
                                              const float4* __restrict__ velB,
                                              const float* __restrict__ rho,
                                              const float* __restrict__ auxc,
                                              const uint32_t* __restrict__ cell_start,
                                              const CellGrid& g, const PairConsts& k,
                                              float4* __restrict__ acc,
                                              const int32_t* __restrict__ ncount = nullptr)
{
   const float4 pi = posm[p];
   int cx, cy, cz;
   cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
   RowRanges r;
   row_ranges(g, cell_start, cx, cy, cz, r);

   AccelState s;
   accel_begin(k, s, pi, velB[p], rho[p]);
   // FAST: the viscous sum visits the last visc_keep() neighbours only (ncount from the density pass)
   int first_v = 0, j = 0;
   if (FAST) {
      const int cnt = ncount[p], keep = visc_keep(s.visc_scale);
      first_v = keep < cnt ? cnt - keep : 0;
   }
#pragma unroll 1
   for (int row = 0; row < 9; row++) {
      const uint32_t b = row == 0 ? r.s[0] : row == 1 ? r.s[1] : row == 2 ? r.s[2] : row == 3 ? r.s[3]
                       : row == 4 ? r.s[4] : row == 5 ? r.s[5] : row == 6 ? r.s[6] : row == 7 ? r.s[7] : r.s[8];
      const uint32_t e = row == 0 ? r.e[0] : row == 1 ? r.e[1] : row == 2 ? r.e[2] : row == 3 ? r.e[3]
                       : row == 4 ? r.e[4] : row == 5 ? r.e[5] : row == 6 ? r.e[6] : row == 7 ? r.e[7] : r.e[8];
      for (uint32_t q = b; q < e; q++) {
         if (q == (uint32_t)p) continue;
         const float4 pj = posm[q];
         float dx, dy, dz;
         const float d2 = dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx, dy, dz);
         if (d2 < k.h2) {
            float d = sqrtf(d2);
            if (!UNIT_SCALE) d *= k.sim_scale;
            if (FAST) {
               accel_pair_fast_pressure<UNIT_SCALE>(k, s, dx, dy, dz, d, auxc[q]);
               if (j >= first_v) {
                  const float4 vj = velB[q];
                  accel_pair_fast_viscous(k, s, d, vj.x, vj.y, vj.z, vj.w);
               }
               j++;
            } else {
               const float4 vj = velB[q];
               accel_pair<UNIT_SCALE>(k, s, dx, dy, dz, d, pj.w, vj.x, vj.y, vj.z, vj.w, auxc[q]);
            }
         }
      }
   }
   if (FAST) accel_fast_finish(k, s);
   acc[p] = accel_end<UNIT_SCALE>(k, s);
}

template <bool UNIT_SCALE, bool FAST>
__global__ void __launch_bounds__(256)
k_full_accel(const float4* __restrict__ posm, const float4* __restrict__ velB,
             const float* __restrict__ rho, const float* __restrict__ auxc,
             const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta, CellGrid g,
             PairConsts k, float4* __restrict__ acc, const int32_t* __restrict__ ncount)
{
   // same workgroup -> particle mapping as the tiled kernels (from the density range); the
   // acceleration is only needed for owned particles
   const int p = meta[META_SUM_BEGIN] + blockIdx.x * blockDim.x + threadIdx.x;
   if (p < meta[META_OWN_BEGIN] || p >= meta[META_OWN_END]) return;
   accel_untiled<UNIT_SCALE, FAST>(p, posm, velB, rho, auxc, cell_start, g, k, acc, ncount);
}
