// REF mode: the shipped neighbour search and the list-driven sums, one thread per particle.
//
// State stays in particle-index order ({x,y,z,m} / {vx,vy,vz,id} float4 pairs), because the
// stored lists hold particle indices exactly like the reference's mNeighbors.  The work is
// integer-heavy and divergent (data-dependent early exits); it is the parity gate against the
// compiled reference, not the throughput path.
#pragma once

#include "pair_math.h"

#define REF_CHUNK 8 // `K`, reference src/sph.cpp:32

// findNeighbors (reference src/sph.cpp:484-692), with every shipped behaviour kept:
//   - octant slots 0,1,2,3',5,6,7 (slot 3 overwritten :536-543, slot 4 never assigned);
//   - a slot is used only if 0 < v < cells on all axes (:578-582) and its list is non-empty;
//   - 32-bit wrapping LCG on (index + slots used) (:590), offset = lcg % len, C truncation (:591);
//   - direction from index parity (:593); chunks of 8 positions, slot abandoned when any of the
//     8 is outside the list (:598-620);
//   - only the first 4 of each 8 are distance tested (:651-671);
//   - stop once count > examine_count - 8 (:679).
__global__ void __launch_bounds__(256)
k_ref_find_neighbors(const float4* __restrict__ posm, const int32_t* __restrict__ vox,
                     const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ order,
                     int n, int nx, int ny, int nz, float h, float htimes2, float h2,
                     float sim_scale, int cap, uint32_t* __restrict__ nb, float* __restrict__ nd,
                     int32_t* __restrict__ ncount)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const float4 pi = posm[i];
   const int X = vox[3 * i + 0], Y = vox[3 * i + 1], Z = vox[3 * i + 2];
   const float ox = pi.x - (X * htimes2);
   const float oy = pi.y - (Y * htimes2);
   const float oz = pi.z - (Z * htimes2);
   const int sx = (ox > h) ? 1 : -1;
   const int sy = (oy > h) ? 1 : -1;
   const int sz = (oz > h) ? 1 : -1;

   uint32_t* my_nb = nb + (size_t)i * cap;
   float* my_nd = nd + (size_t)i * cap;

   int count = 0;
   int used = 0;
   bool enough = false;
   const int dir = (i % 2) ? -1 : 1;

   for (int s = 0; s < 8 && !enough; s++) {
      if (s == 4) continue;
      // slot -> voxel offset; bit0: x, bit1: y, bit2: z — except slot 3 = (x,y) and there is no
      // pure-z slot (the reference overwrites it)
      const int bx = (s == 1 || s == 3 || s == 5 || s == 7) ? sx : 0;
      const int by = (s == 2 || s == 3 || s == 6 || s == 7) ? sy : 0;
      const int bz = (s >= 5) ? sz : 0;
      const int vx = X + bx, vy = Y + by, vz = Z + bz;
      if (!(vx > 0 && vx < nx && vy > 0 && vy < ny && vz > 0 && vz < nz)) continue;
      const int id = (vz * ny + vy) * nx + vx;
      const uint32_t start = cell_start[id];
      const int len = (int)(cell_start[id + 1] - start);
      if (len == 0) continue;

      const int32_t lcg = (int32_t)(1664525u * (uint32_t)(i + used) + 1013904223u);
      const int offset = lcg % len;
      used++;

      int ii = 0;
      const int max_steps = (len + REF_CHUNK - 1) / REF_CHUNK;
      for (int step = 0; step < max_steps; ++step) {
         const int first = offset + ii * dir;
         // positions first .. first+7 must all lie in [0, len)
         if (first < 0 || first + (REF_CHUNK - 1) >= len) break;
         ii += REF_CHUNK;
         // the four tested positions: indices first, then the four gathers, then the tests in
         // order - two dependent round trips per chunk instead of eight
         uint32_t q[4];
         float4 pj[4];
#pragma unroll
         for (int j = 0; j < 4; j++) q[j] = order[start + (uint32_t)(first + j)];
#pragma unroll
         for (int j = 0; j < 4; j++) pj[j] = posm[q[j]];
#pragma unroll
         for (int j = 0; j < 4; j++) {
            if (q[j] == (uint32_t)i) continue;
            float dx, dy, dz;
            const float d2 = dist2(pi.x, pi.y, pi.z, pj[j].x, pj[j].y, pj[j].z, dx, dy, dz);
            if (d2 < h2) {
               my_nb[count] = q[j];
               my_nd[count] = sqrtf(d2) * sim_scale;
               count++;
            }
         }
         enough = (count > cap - REF_CHUNK);
         if (enough) break;
      }
   }
   ncount[i] = count;
}

// computeDensity over stored lists (reference src/sph.cpp:721-766)
__global__ void __launch_bounds__(256)
k_ref_density(const float4* __restrict__ posm, const uint32_t* __restrict__ nb,
              const float* __restrict__ nd, const int32_t* __restrict__ ncount, int n, int cap,
              PairConsts k, float* __restrict__ rho)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const uint32_t* my_nb = nb + (size_t)i * cap;
   const float* my_nd = nd + (size_t)i * cap;
   const int cnt = ncount[i];
   float density = 0.0f;
   // four list entries per trip: their index loads, then their gathers, are issued together
   // (positions past the count re-read the last entry; unused)
   bool stop = false;
   for (int k0 = 0; k0 < cnt && !stop; k0 += 4) {
      uint32_t q[4];
      float dj[4], mj[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const int kk = min(k0 + u, cnt - 1);
         q[u] = my_nb[kk];
         dj[u] = my_nd[kk];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) mj[u] = posm[min(q[u], (uint32_t)(n - 1))].w;
#pragma unroll
      for (int u = 0; u < 4; u++) {
         if (k0 + u < cnt && !stop) {
            if (q[u] >= (uint32_t)n) stop = true;
            else if (q[u] != (uint32_t)i) density_accumulate(k, mj[u], dj[u], density);
         }
      }
   }
   rho[i] = density;
}

// computeAcceleration over stored lists (reference src/sph.cpp:778-934)
__global__ void __launch_bounds__(256)
k_ref_accel(const float4* __restrict__ posm, const float4* __restrict__ velp,
            const float* __restrict__ rho, const uint32_t* __restrict__ nb,
            const float* __restrict__ nd, const int32_t* __restrict__ ncount, int n, int cap,
            PairConsts k, float4* __restrict__ acc)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const uint32_t* my_nb = nb + (size_t)i * cap;
   const float* my_nd = nd + (size_t)i * cap;
   const int cnt = ncount[i];
   const float4 pi = posm[i];
   AccelState s;
   accel_begin(k, s, pi, velp[i], rho[i]);
   for (int k0 = 0; k0 < cnt; k0 += 4) {   // as above: four entries' loads in flight together
      uint32_t q[4];
      float dj[4], rj[4];
      float4 pj[4], vj[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const int kk = min(k0 + u, cnt - 1);
         q[u] = my_nb[kk];
         dj[u] = my_nd[kk];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         pj[u] = posm[q[u]];
         vj[u] = velp[q[u]];
         rj[u] = rho[q[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         if (k0 + u < cnt) {
            const float2 bc = neighbor_terms(k, rj[u], pj[u].w);
            accel_pair<false>(k, s, pi.x - pj[u].x, pi.y - pj[u].y, pi.z - pj[u].z, dj[u], pj[u].w,
                              vj[u].x, vj[u].y, vj[u].z, bc.x, bc.y);
         }
      }
   }
   acc[i] = accel_end<false>(k, s);
}
