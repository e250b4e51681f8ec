// Kernels shared by both modes: integrate (+ energy totals), host-layout import/export,
// neighbour-count statistics.
#pragma once

#include "cell_build.h"
#include "pair_math.h"

#define RED_THREADS 256

// SPH::applyBoundary (reference src/sph.cpp:1124-1148): reflect at a wall with unit normal
// along `axis` (sign sgn), continue for the rest of the step scaled by mDamping.  vec3
// operators of the reference are component-wise fp32 operations.
__device__ __forceinline__ void apply_boundary(const PairConsts& k, const float pos[3], float dt,
                                               float np[3], float dist, int axis, float sgn,
                                               float nv[3])
{
   float normal[3] = {0.0f, 0.0f, 0.0f};
   normal[axis] = sgn;
   float inter[3], refl[3];
#pragma unroll
   for (int c = 0; c < 3; c++) inter[c] = pos[c] + (nv[c] * dist);
   const float dot = nv[0] * normal[0] + nv[1] * normal[1] + nv[2] * normal[2];
#pragma unroll
   for (int c = 0; c < 3; c++) refl[c] = nv[c] - ((normal[c] * dot) * 2.0f);
   const float remaining = dt - dist;
#pragma unroll
   for (int c = 0; c < 3; c++) {
      nv[c] = refl[c];
      np[c] = inter[c] + refl[c] * (remaining * k.damping);
   }
}

// SPH::handleBoundaryConditions (reference src/sph.cpp:1025-1121): x, then y, then z.
__device__ __forceinline__ void handle_boundaries(const PairConsts& k, const float pos[3],
                                                  float nv[3], float dt, float np[3])
{
   const float maxv[3] = {k.max_x, k.max_y, k.max_z};
#pragma unroll
   for (int axis = 0; axis < 3; axis++) {
      if (np[axis] < 0.0f)
         apply_boundary(k, pos, dt, np, -pos[axis] / nv[axis], axis, 1.0f, nv);
      else if (np[axis] > maxv[axis])
         apply_boundary(k, pos, dt, np, (maxv[axis] - pos[axis]) / nv[axis], axis, -1.0f, nv);
   }
}

// ---- integrate (reference src/sph.cpp:937-1022) -------------------------------------------------
// "KDK as coded": half kick with the SPH acceleration, drift, then a FULL-dt kick with the
// point-mass gravity only, evaluated at the new position (reference src/sph.cpp:937-1022).
// Updates x (position, mass kept) and v (velocity, id kept); ke/pe = the particle's energy terms.
template <bool UNIT_SCALE>
__device__ __forceinline__ void integrate_particle(const PairConsts& k, float4& x, float4& v,
                                                   const float4 a, double& ke, double& pe)
{
   const float dt = k.dt;
   const float pos_dt = dt * k.sim_scale_inv;

   const float vhx = v.x + (a.x * dt * 0.5f);
   const float vhy = v.y + (a.y * dt * 0.5f);
   const float vhz = v.z + (a.z * dt * 0.5f);
   const float nx0 = x.x + (vhx * pos_dt);
   const float ny0 = x.y + (vhy * pos_dt);
   const float nz0 = x.z + (vhz * pos_dt);

   // (k.skip_point_mass: tolerance mode without a point mass - the term is +-0, see accel_end; d3 then
   // only divides a potential energy of exactly zero)
   float agx = 0.0f, agy = 0.0f, agz = 0.0f, d3 = 1.0f;
   if (!k.skip_point_mass) {
      float rsx = (nx0 - k.cx), rsy = (ny0 - k.cy), rsz = (nz0 - k.cz);
      if (!UNIT_SCALE) {
         rsx *= k.sim_scale;
         rsy *= k.sim_scale;
         rsz *= k.sim_scale;
      }
      float dot = rsx * rsx + rsy * rsy + rsz * rsz;
      dot = sqrtf(dot);
      const float ds = dot + k.softening;
      d3 = ds * ds * ds;
      const float gm = -k.grav_const * k.central_mass;
      agx = gm * (rsx / d3);
      agy = gm * (rsy / d3);
      agz = gm * (rsz / d3);
   } else {
      agx = agy = agz = (nx0 - nx0) + (ny0 - ny0) + (nz0 - nz0);   // 0, or NaN where the term is: see accel_end
   }
   if (k.apply_gravity) { // extension, as in accel_end
      agx += k.gx;
      agy += k.gy;
      agz += k.gz;
   }
   float nvx = vhx + (agx * dt);
   float nvy = vhy + (agy * dt);
   float nvz = vhz + (agz * dt);
   float nx = nx0, ny = ny0, nz = nz0;
   if (k.apply_walls) { // extension: the reference's own (unwired) wall handling
      const float pos[3] = {x.x, x.y, x.z};
      float nv[3] = {nvx, nvy, nvz}, np[3] = {nx, ny, nz};
      handle_boundaries(k, pos, nv, dt, np);
      nvx = nv[0]; nvy = nv[1]; nvz = nv[2];
      nx = np[0]; ny = np[1]; nz = np[2];
   }

   const float dot = nvx * nvx + nvy * nvy + nvz * nvz;
   ke = 0.0;
   pe = 0.0;
   if (dot > 0) {
      ke = (double)(0.5f * x.w * dot);
      pe = -(double)(k.grav_const * k.central_mass * x.w / d3);
   }
   x.x = nx; x.y = ny; x.z = nz;
   v.x = nvx; v.y = nvy; v.z = nvz;
}

// KE/PE contributions are reduced per block in double (the reference's serial fp32 running sum
// is order dependent).
// HASH: the context holds the whole grid and exchanges with nobody, so the sorted state this
// kernel leaves is exactly the input of the next cell build: the build's first step (cell id,
// counting atomics - k_hash_count) is done here, on the position just computed, and the next
// build starts at its scan.  One launch and one read of the positions less per step.
template <bool UNIT_SCALE, bool HASH>
__global__ void __launch_bounds__(RED_THREADS)
k_integrate(float4* __restrict__ posm, float4* __restrict__ velp, const float4* __restrict__ acc,
            const int32_t* __restrict__ meta, PairConsts k, double* __restrict__ epart, CellGrid g,
            uint32_t* __restrict__ key, uint32_t* __restrict__ slot,
            uint32_t* __restrict__ cell_count)
{
   // owned particles only: ghosts are integrated by the slab that owns them
   const int p = meta[META_OWN_BEGIN] + blockIdx.x * blockDim.x + threadIdx.x;
   double ke = 0.0, pe = 0.0;
   const bool live = p < meta[META_OWN_END];
   uint32_t c = 0xffffffffu;
   if (live) {
      float4 x = posm[p];
      float4 v = velp[p];
      integrate_particle<UNIT_SCALE>(k, x, v, acc[p], ke, pe);
      posm[p] = x;
      velp[p] = v;
      if (HASH) {
         int cx, cy, cz;
         c = cell_of(g, x.x, x.y, x.z, cx, cy, cz);
         key[p] = c;
      }
   }
   if (HASH) count_cell_runs(c, live, p, cell_count, slot, (uint32_t)g.ncells);
   // block reduction, fixed order
   __shared__ double s_ke[RED_THREADS / SPH_WAVE], s_pe[RED_THREADS / SPH_WAVE];
#pragma unroll
   for (int d = SPH_WAVE / 2; d > 0; d >>= 1) {
      ke += __shfl_down(ke, d);
      pe += __shfl_down(pe, d);
   }
   const int lane = threadIdx.x & (SPH_WAVE - 1), w = threadIdx.x / SPH_WAVE;
   if (lane == 0) {
      s_ke[w] = ke;
      s_pe[w] = pe;
   }
   __syncthreads();
   if (threadIdx.x == 0) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int q = 0; q < RED_THREADS / SPH_WAVE; q++) {
         a += s_ke[q];
         b += s_pe[q];
      }
      epart[2 * blockIdx.x + 0] = a;
      epart[2 * blockIdx.x + 1] = b;
   }
}

// one block: totals of the per-block partials, written to out[0..1]
__global__ void __launch_bounds__(RED_THREADS)
k_energy_total(const double* __restrict__ epart, int nblocks, double* __restrict__ out)
{
   double ke = 0.0, pe = 0.0;
   for (int b = threadIdx.x; b < nblocks; b += RED_THREADS) {
      ke += epart[2 * b + 0];
      pe += epart[2 * b + 1];
   }
   __shared__ double s_ke[RED_THREADS], s_pe[RED_THREADS];
   s_ke[threadIdx.x] = ke;
   s_pe[threadIdx.x] = pe;
   __syncthreads();
   for (int d = RED_THREADS / 2; d > 0; d >>= 1) {
      if ((int)threadIdx.x < d) {
         s_ke[threadIdx.x] += s_ke[threadIdx.x + d];
         s_pe[threadIdx.x] += s_pe[threadIdx.x + d];
      }
      __syncthreads();
   }
   if (threadIdx.x == 0) {
      out[0] = s_ke[0];
      out[1] = s_pe[0];
   }
}

// ---- host layout <-> device layout --------------------------------------------------------------
// stage holds the reference's arrays back to back: pos[3n] vel[3n] mass[n]
__global__ void __launch_bounds__(256)
k_import(const float* __restrict__ pos, const float* __restrict__ vel,
         const float* __restrict__ mass, const uint32_t* __restrict__ ids, int n,
         float4* __restrict__ posm, float4* __restrict__ velp)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const uint32_t id = ids ? ids[i] : (uint32_t)i;
   posm[i] = make_float4(pos[3 * i + 0], pos[3 * i + 1], pos[3 * i + 2], mass[i]);
   velp[i] = make_float4(vel[3 * i + 0], vel[3 * i + 1], vel[3 * i + 2], __uint_as_float(id));
}

// scatter by persistent id into pos[3n] vel[3n] rho[n] acc[3n] ncount[n] (any may be null)
// `compact`: write row (p - own_begin) instead of row id, and the ids to oid (slab download)
__global__ void __launch_bounds__(256)
k_export(const float4* __restrict__ posm, const float4* __restrict__ velp,
         const float* __restrict__ rho, const float4* __restrict__ acc,
         const int32_t* __restrict__ ncount, const int32_t* __restrict__ meta, int compact,
         float* __restrict__ pos, float* __restrict__ vel, float* __restrict__ orho,
         float* __restrict__ oacc, int32_t* __restrict__ ocount, uint32_t* __restrict__ oid)
{
   const int p = meta[META_OWN_BEGIN] + blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= meta[META_OWN_END]) return;
   const float4 v = velp[p];
   uint32_t id = __float_as_uint(v.w);
   if (compact) {
      if (oid) oid[p - meta[META_OWN_BEGIN]] = id;
      id = (uint32_t)(p - meta[META_OWN_BEGIN]);
   }
   if (pos) {
      const float4 x = posm[p];
      pos[3 * id + 0] = x.x;
      pos[3 * id + 1] = x.y;
      pos[3 * id + 2] = x.z;
   }
   if (vel) {
      vel[3 * id + 0] = v.x;
      vel[3 * id + 1] = v.y;
      vel[3 * id + 2] = v.z;
   }
   if (orho) orho[id] = rho[p];
   if (oacc) {
      const float4 a = acc[p];
      oacc[3 * id + 0] = a.x;
      oacc[3 * id + 1] = a.y;
      oacc[3 * id + 2] = a.z;
   }
   if (ocount) ocount[id] = ncount[p];
}

// masses of the owned particles, row p - own_begin (the row order of a compact k_export)
__global__ void __launch_bounds__(256)
k_export_mass(const float4* __restrict__ posm, const int32_t* __restrict__ meta, float* __restrict__ mass)
{
   const int p = meta[META_OWN_BEGIN] + blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= meta[META_OWN_END]) return;
   mass[p - meta[META_OWN_BEGIN]] = posm[p].w;
}

// per-voxel occupancy on the REFERENCE voxel grid (edge mCellSize = 2h, reference
// src/sph.cpp:438-481) whatever grid the context sorts by: what getGrid()[i].count() readers get
__global__ void __launch_bounds__(256)
k_voxel_counts(const float4* __restrict__ posm, const int32_t* __restrict__ meta, float inv, int nx,
               int ny, int nz, int32_t* __restrict__ counts)
{
   const int p = meta[META_OWN_BEGIN] + blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= meta[META_OWN_END]) return;
   const float4 x = posm[p];
   const int cx = cell_coord(x.x, inv, nx), cy = cell_coord(x.y, inv, ny), cz = cell_coord(x.z, inv, nz);
   atomicAdd(&counts[(cz * ny + cy) * nx + cx], 1);
}

// ---- neighbour statistics (reference src/sph.cpp:204-232) -----------------------------------------
// out: [0] = sum low 32, [1] = sum high 32 (as one 64-bit add), [2] = max, [3] = min (from 34)
__global__ void __launch_bounds__(RED_THREADS)
k_neighbor_stats(const int32_t* __restrict__ ncount, const int32_t* __restrict__ meta,
                 int32_t* __restrict__ out)
{
   long long sum = 0;
   int mx = -1, mn = 34;
   const int begin = meta[META_OWN_BEGIN], end = meta[META_OWN_END];
   for (int i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
      const int c = ncount[i];
      sum += c;
      mx = c > mx ? c : mx;
      mn = c < mn ? c : mn;
   }
#pragma unroll
   for (int d = SPH_WAVE / 2; d > 0; d >>= 1) {
      sum += __shfl_down(sum, d);
      const int omx = __shfl_down(mx, d), omn = __shfl_down(mn, d);
      mx = omx > mx ? omx : mx;
      mn = omn < mn ? omn : mn;
   }
   if ((threadIdx.x & (SPH_WAVE - 1)) == 0) {
      atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)sum);
      atomicMax(out + 2, mx);
      atomicMin(out + 3, mn);
   }
}
