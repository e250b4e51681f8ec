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

// ---- density, untiled: one thread per particle, candidates read through L1/L2.  Used for
// workgroups whose LDS tile would overflow (full_tiled.h) and as an independent cross-check.
template <bool UNIT_SCALE>
__global__ void __launch_bounds__(256)
k_full_density(const float4* __restrict__ posm, const uint32_t* __restrict__ cell_start,
               const float4* __restrict__ velp, const int32_t* __restrict__ meta, CellGrid g,
               PairConsts k, float* __restrict__ rho, float4* __restrict__ velB,
               float* __restrict__ auxc, int32_t* __restrict__ ncount,
               const uint32_t* __restrict__ redo)
{
   // redo == nullptr: every workgroup (SPH_HIP_UNTILED=1).  Otherwise the fallback of the tiled
   // kernel: redo[0] = number of 256-particle workgroups it gave up on (tile or neighbour list
   // did not fit), redo[1..] = their indices; a small grid walks that list.
   const int nwork = redo ? (int)redo[0] : (int)gridDim.x;
   for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
   const int wg = redo ? (int)redo[1 + w] : w;
   const int p = meta[META_SUM_BEGIN] + wg * blockDim.x + threadIdx.x;
   if (p >= meta[META_SUM_END]) continue;
   const float4 pi = posm[p];
   int cx, cy, cz;
   cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
   RowRanges r;
   row_ranges(g, cell_start, cx, cy, cz, r);

   float density = 0.0f;
   int count = 0;
#pragma unroll
   for (int row = 0; row < 9; row++) {
      for (uint32_t q = r.s[row]; q < r.e[row]; q++) {
         if (q == (uint32_t)p) continue;
         const float4 pj = posm[q];
         float dx, dy, dz;
         const float d2 = dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx, dy, dz);
         if (d2 < k.h2) {
            float d = sqrtf(d2);
            if (!UNIT_SCALE) d *= k.sim_scale;
            density_accumulate(k, pj.w, d, density);
            count++;
         }
      }
   }
   rho[p] = density;
   const float2 bc = neighbor_terms(k, density, pi.w);
   const float4 v = velp[p];
   velB[p] = make_float4(v.x, v.y, v.z, bc.x);
   auxc[p] = bc.y;
   ncount[p] = count;
   }
}

// ---- acceleration, untiled ------------------------------------------------------------------------
template <bool UNIT_SCALE>
__global__ void __launch_bounds__(256)
k_full_accel(const float4* __restrict__ posm, const float4* __restrict__ velB,
             const float* __restrict__ rho, const float* __restrict__ auxc,
             const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta, CellGrid g,
             PairConsts k, float4* __restrict__ acc, const uint32_t* __restrict__ redo)
{
   // same scheme as k_full_density: all workgroups, or the tiled pass's give-up list
   const int nwork = redo ? (int)redo[0] : (int)gridDim.x;
   for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
   const int wg = redo ? (int)redo[1 + w] : w;
   // same workgroup -> particle mapping as the tiled kernels (from the density range); the
   // acceleration is only needed for owned particles
   const int p = meta[META_SUM_BEGIN] + wg * blockDim.x + threadIdx.x;
   if (p < meta[META_OWN_BEGIN] || p >= meta[META_OWN_END]) continue;
   const float4 pi = posm[p];
   int cx, cy, cz;
   cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
   RowRanges r;
   row_ranges(g, cell_start, cx, cy, cz, r);

   AccelState s;
   accel_begin(k, s, pi, velB[p], rho[p]);
#pragma unroll
   for (int row = 0; row < 9; row++) {
      for (uint32_t q = r.s[row]; q < r.e[row]; q++) {
         if (q == (uint32_t)p) continue;
         const float4 pj = posm[q];
         float dx, dy, dz;
         const float d2 = dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx, dy, dz);
         if (d2 < k.h2) {
            float d = sqrtf(d2);
            if (!UNIT_SCALE) d *= k.sim_scale;
            const float4 vj = velB[q];
            accel_pair<UNIT_SCALE>(k, s, dx, dy, dz, d, pj.w, vj.x, vj.y, vj.z, vj.w, auxc[q]);
         }
      }
   }
   acc[p] = accel_end<UNIT_SCALE>(k, s);
   }
}
