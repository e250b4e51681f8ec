// FULL mode, LDS-tiled: the throughput kernels for the density and acceleration sums.
//
// One workgroup = 256 consecutive particles of the cell-sorted state (about 34 cells of one
// grid row).  Work per workgroup:
//   1. TILE    the candidate positions of the 9 neighbouring row segments (cells
//              c_first-1 .. c_last+1 of each (dz,dy) row, contiguous in the sorted arrays) are
//              copied once into LDS as three float arrays x[], y[], z[] (coalesced 16-byte
//              global loads);
//   2. TEST    each lane walks its own 9 candidate ranges inside the tile, eight aligned slots
//              per step: six ds_read_b128 fetch x/y/z of eight consecutive candidates, the
//              distance test (dx*dx + dy*dy) + dz*dz < h^2 runs as branch-free packed fp32
//              math and yields one acceptance bit per slot; the set bits of a 32-slot chunk are
//              then expanded, ascending, into the lane's private queue in LDS (lane-contiguous
//              layout, conflict free);
//   3. SUM     when a queue is nearly full (and at the end) the wave drains the queues:
//              the expensive per-pair arithmetic only ever runs on accepted neighbours, in the
//              canonical order (ascending sorted position).
// The reference visits a particle's neighbours sequentially and its viscous term is
// rescaled inside that loop (src/sph.cpp:880-882), so each particle's sum stays on one lane;
// nothing is reduced across lanes.
//
// Workgroups whose tile would not fit (very dense regions) are flagged and redone by the
// untiled kernels of full_kernels.h — same results, slower.
#pragma once

#include "full_kernels.h"

#define TILE_THREADS 256
#define TILE_CAP 3008   // candidate positions per workgroup tile (3 x 11.9 KiB; 3 workgroups per CU)
#define QUEUE_DEPTH 32  // accepted-neighbour slots per lane between two SUM phases
// Neighbour lists handed from the density pass to the acceleration pass: per workgroup
// NLIST_CAP rows of 256 u16 queue entries (row j = every lane's j-th accepted neighbour, so a
// wave reads/writes 128 contiguous bytes).  Only rows in use are ever touched.
#define NLIST_CAP 96
// queue entry: segment id << 12 | tile index (TILE_CAP + 32 <= 4096)
#define QUEUE_TBITS 12
#define QUEUE_TMASK 0xfffu

typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef float __attribute__((ext_vector_type(4))) f32x4;

// 16-byte LDS read of four consecutive floats; i must be a multiple of 4
__device__ __forceinline__ f32x4 lds_read4(const float* base, int i)
{
   return *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(base + i, 16));
}

static_assert(TILE_CAP + 32 <= (1 << QUEUE_TBITS), "tile index must fit the queue entry");
#define TILE_ROUNDS ((TILE_CAP + TILE_THREADS - 1) / TILE_THREADS)

// Per-workgroup tile layout, computed by k_tile_desc before the sums run.
struct TileDesc {
   int D[9];      // tile index = sorted index + D[k] inside segment k
   int B[9];      // first tile index of segment k (B[0] = 0)
   int total;     // tile entries; > TILE_CAP => the workgroup takes the untiled kernel
   int pad;
};

struct TileLds {
   __attribute__((aligned(16))) float x[TILE_CAP + 32];
   __attribute__((aligned(16))) float y[TILE_CAP + 32];
   __attribute__((aligned(16))) float z[TILE_CAP + 32];
   uint16_t queue[QUEUE_DEPTH * TILE_THREADS];
   TileDesc desc;
};

// One thread per workgroup-to-be: the 9 row segments (cells c_first-1 .. c_last+1 of every
// (dz,dy) row, as linear cell-id ranges) of the 256 particles starting at tile*256.
__global__ void __launch_bounds__(256)
k_tile_desc(const float4* __restrict__ posm, const uint32_t* __restrict__ cell_start,
            const int32_t* __restrict__ meta, int range, CellGrid g, int ntiles,
            TileDesc* __restrict__ desc)
{
   const int tile = blockIdx.x * blockDim.x + threadIdx.x;
   if (tile >= ntiles) return;
   const int begin = meta[range], end = meta[range + 1];
   const int p0 = begin + tile * TILE_THREADS;
   if (p0 >= end) return;
   const int plast = min(p0 + TILE_THREADS - 1, end - 1);
   const float4 a = posm[p0], b = posm[plast];
   int cx, cy, cz;
   const int c_first = (int)cell_of(g, a.x, a.y, a.z, cx, cy, cz);
   const int c_last = (int)cell_of(g, b.x, b.y, b.z, cx, cy, cz);
   TileDesc d;
   int G[9], len[9];
#pragma unroll
   for (int k = 0; k < 9; k++) {
      const int off = ((k / 3 - 1) * g.ny + (k % 3 - 1)) * g.nx;
      int lo = c_first + off - 1, hi = c_last + off + 1;
      lo = lo < 0 ? 0 : lo;
      hi = hi > g.ncells - 1 ? g.ncells - 1 : hi;
      G[k] = 0;
      len[k] = 0;
      if (hi >= lo) {
         G[k] = (int)cell_start[lo];
         len[k] = (int)cell_start[hi + 1] - G[k];
      }
   }
   int run = 0;
#pragma unroll
   for (int k = 0; k < 9; k++) {
      d.B[k] = run;
      d.D[k] = run - G[k];
      run += len[k];
   }
   d.total = run;
   d.pad = 0;
   desc[tile] = d;
}

// Copies the candidate positions of the workgroup's tile into LDS: every thread issues all of
// its (at most TILE_ROUNDS) 16-byte loads before the first LDS store.
__device__ __forceinline__ void tile_load(const float4* __restrict__ posm,
                                          const TileDesc* __restrict__ desc, TileLds& L)
{
   const int tid = threadIdx.x;
   if (tid < (int)(sizeof(TileDesc) / sizeof(int)))
      reinterpret_cast<int*>(&L.desc)[tid] = reinterpret_cast<const int*>(&desc[blockIdx.x])[tid];
   __syncthreads();
   const int total = L.desc.total;
#if defined(SPH_ABLATE) && (SPH_ABLATE == 3 || SPH_ABLATE == 5)
   if (false) {
#else
   if (total <= TILE_CAP) {
#endif
      int B[9], D[9];
#pragma unroll
      for (int k = 0; k < 9; k++) {
         B[k] = L.desc.B[k];
         D[k] = L.desc.D[k];
      }
      float4 buf[TILE_ROUNDS];
#pragma unroll
      for (int r = 0; r < TILE_ROUNDS; r++) {
         const int idx = tid + r * TILE_THREADS;
         if (idx < total) {
            int d = D[0];
#pragma unroll
            for (int k = 1; k < 9; k++) d = (idx >= B[k]) ? D[k] : d;
            buf[r] = posm[idx - d];
         }
      }
#pragma unroll
      for (int r = 0; r < TILE_ROUNDS; r++) {
         const int idx = tid + r * TILE_THREADS;
         if (idx < total) {
            L.x[idx] = buf[r].x;
            L.y[idx] = buf[r].y;
            L.z[idx] = buf[r].z;
         }
      }
   }
   __syncthreads();
}

// fp32 squared distances of two candidates at once: ((dx*dx) + (dy*dy)) + (dz*dz) per lane
// element, same association and rounding as dist2() — packed ops round each half like the
// scalar ops do.
__device__ __forceinline__ f32x2 dist2_pair(f32x2 px, f32x2 py, f32x2 pz, f32x2 cx, f32x2 cy,
                                            f32x2 cz)
{
   const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
   return dx * dx + dy * dy + dz * dz;
}

template <bool UNIT_SCALE, bool UNIFORM_MASS, int PASS>
struct TiledSum;

// What the SUM phase needs of one queued neighbour.
struct Staged {
   uint32_t entry;
   float x, y, z, m;
   float4 v;
   float2 bc;
};

// ---- density -------------------------------------------------------------------------------
template <bool UNIT_SCALE, bool UNIFORM_MASS>
struct TiledSum<UNIT_SCALE, UNIFORM_MASS, 0> {
   float density = 0.0f;
   int count = 0;
   uint16_t* nlist = nullptr;  // this lane's column of the workgroup's list block
   int overflow = 0;

   __device__ __forceinline__ void stage(Staged& s, const TileLds& L, int tid, int j,
                                         const float4& pi, const float4* __restrict__ posm,
                                         const float4* __restrict__ velp,
                                         const float2* __restrict__ aux) const
   {
      s.entry = L.queue[j * TILE_THREADS + tid];
      const int t = (int)(s.entry & QUEUE_TMASK);
      s.x = L.x[t];
      s.y = L.y[t];
      s.z = L.z[t];
      s.m = pi.w;
      if (!UNIFORM_MASS) s.m = posm[t - L.desc.D[s.entry >> QUEUE_TBITS]].w;
   }

   __device__ __forceinline__ void pair(const PairConsts& k, const Staged& s, const float4& pi,
                                        uint32_t self_entry)
   {
#if defined(SPH_ABLATE) && SPH_ABLATE == 1
      if (s.entry != self_entry) count++;
      return;
#endif
      if (s.entry != self_entry) {
         float dx, dy, dz;
         const float d2 = dist2(pi.x, pi.y, pi.z, s.x, s.y, s.z, dx, dy, dz);
         float d = sqrtf(d2);
         if (!UNIT_SCALE) d *= k.sim_scale;
         density_accumulate(k, s.m, d, density);
         if (count < NLIST_CAP) nlist[count * TILE_THREADS] = (uint16_t)s.entry;
         else overflow = 1;
         count++;
      }
   }
};

// ---- acceleration ----------------------------------------------------------------------------
template <bool UNIT_SCALE, bool UNIFORM_MASS>
struct TiledSum<UNIT_SCALE, UNIFORM_MASS, 1> {
   AccelState s;

   __device__ __forceinline__ void stage(Staged& g, const TileLds& L, int tid, int j,
                                         const float4& pi, const float4* __restrict__ posm,
                                         const float4* __restrict__ velp,
                                         const float2* __restrict__ aux) const
   {
      stage_entry(g, L.queue[j * TILE_THREADS + tid], L, pi, posm, velp, aux);
   }

   __device__ __forceinline__ void stage_entry(Staged& g, uint32_t entry, const TileLds& L,
                                               const float4& pi, const float4* __restrict__ posm,
                                               const float4* __restrict__ velp,
                                               const float2* __restrict__ aux) const
   {
      g.entry = entry;
      const int t = (int)(g.entry & QUEUE_TMASK);
      const int q = t - L.desc.D[g.entry >> QUEUE_TBITS];
      g.v = velp[q];
      g.bc = aux[q];
      g.m = pi.w;
      if (!UNIFORM_MASS) g.m = posm[q].w;
      g.x = L.x[t];
      g.y = L.y[t];
      g.z = L.z[t];
   }

   __device__ __forceinline__ void pair(const PairConsts& k, const Staged& g, const float4& pi,
                                        uint32_t self_entry)
   {
#if defined(SPH_ABLATE) && SPH_ABLATE == 1
      if (g.entry != self_entry) s.pgx += __uint_as_float(g.entry);
      return;
#endif
      if (g.entry != self_entry) {
         float dx, dy, dz;
         const float d2 = dist2(pi.x, pi.y, pi.z, g.x, g.y, g.z, dx, dy, dz);
         float d = sqrtf(d2);
         if (!UNIT_SCALE) d *= k.sim_scale;
         accel_pair<UNIT_SCALE>(k, s, dx, dy, dz, d, g.m, g.v.x, g.v.y, g.v.z, g.bc.x, g.bc.y);
      }
   }
};

// SUM phase: every lane walks its queue [0, qn) in order.  (Fetching iteration j+1's operands
// ahead of iteration j's arithmetic was measured slower: the extra register copies and branch
// cost more than the latency they hide at 3 waves per SIMD.)
template <class Sum>
__device__ __forceinline__ void drain_queue(Sum& sum, const PairConsts& k, const TileLds& L,
                                            int tid, int qn, const float4& pi,
                                            uint32_t self_entry, const float4* __restrict__ posm,
                                            const float4* __restrict__ velp,
                                            const float2* __restrict__ aux)
{
   for (int j = 0; __any(j < qn); ++j) {
      if (j < qn) {
         Staged s;
         sum.stage(s, L, tid, j, pi, posm, velp, aux);
         sum.pair(k, s, pi, self_entry);
      }
   }
}

// TEST step: eight consecutive, 32-byte aligned tile slots t..t+7 -> 8 acceptance bits.
// Six independent ds_read_b128 and branch-free packed math; slots outside the lane's range
// are masked by the caller.
__device__ __forceinline__ uint32_t test8(const TileLds& L, int t, f32x2 px, f32x2 py, f32x2 pz,
                                         float h2)
{
   const f32x4 X0 = lds_read4(L.x, t);
   const f32x4 X1 = lds_read4(L.x, t + 4);
   const f32x4 Y0 = lds_read4(L.y, t);
   const f32x4 Y1 = lds_read4(L.y, t + 4);
   const f32x4 Z0 = lds_read4(L.z, t);
   const f32x4 Z1 = lds_read4(L.z, t + 4);
   const f32x2 a = dist2_pair(px, py, pz, f32x2{X0.x, X0.y}, f32x2{Y0.x, Y0.y}, f32x2{Z0.x, Z0.y});
   const f32x2 b = dist2_pair(px, py, pz, f32x2{X0.z, X0.w}, f32x2{Y0.z, Y0.w}, f32x2{Z0.z, Z0.w});
   const f32x2 c = dist2_pair(px, py, pz, f32x2{X1.x, X1.y}, f32x2{Y1.x, Y1.y}, f32x2{Z1.x, Z1.y});
   const f32x2 d = dist2_pair(px, py, pz, f32x2{X1.z, X1.w}, f32x2{Y1.z, Y1.w}, f32x2{Z1.z, Z1.w});
   uint32_t m = 0;
   m |= (a.x < h2) ? 1u : 0u;
   m |= (a.y < h2) ? 2u : 0u;
   m |= (b.x < h2) ? 4u : 0u;
   m |= (b.y < h2) ? 8u : 0u;
   m |= (c.x < h2) ? 16u : 0u;
   m |= (c.y < h2) ? 32u : 0u;
   m |= (d.x < h2) ? 64u : 0u;
   m |= (d.y < h2) ? 128u : 0u;
   return m;
}

template <bool UNIT_SCALE, bool UNIFORM_MASS, int PASS>
__global__ void __launch_bounds__(TILE_THREADS, 3)
k_full_tiled(const float4* __restrict__ posm, const float4* __restrict__ velp,
             const float* __restrict__ rho_in, const float2* __restrict__ aux_in,
             const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta, CellGrid g,
             PairConsts k, float* __restrict__ rho_out, float2* __restrict__ aux_out,
             int32_t* __restrict__ ncount, float4* __restrict__ acc,
             const TileDesc* __restrict__ desc, uint16_t* __restrict__ nlist,
             uint32_t* __restrict__ nlist_overflow)
{
   __shared__ __attribute__((aligned(16))) TileLds L;
   __shared__ int list_overflow;

   // Both passes tile the range whose density is needed, planes [lo-1, hi+1): the same tiling
   // lets the acceleration pass reuse the density pass's neighbour lists.  The acceleration
   // is only computed for the owned planes [lo, hi), a sub-range.
   const int begin = meta[META_SUM_BEGIN];
   const int end = meta[META_SUM_END];
   const int tid = threadIdx.x;
   const int p0 = begin + blockIdx.x * TILE_THREADS;
   if (p0 >= end) return;
   const int p = p0 + tid;
   bool live = p < end;
   if (PASS == 1) {
      const int ob = meta[META_OWN_BEGIN], oe = meta[META_OWN_END];
      if (p0 + TILE_THREADS <= ob || p0 >= oe) return;  // a workgroup of ghosts only
      live = live && p >= ob && p < oe;
   }
   tile_load(posm, desc, L);
   if (L.desc.total > TILE_CAP) return;  // tile does not fit: the untiled kernel redoes this workgroup
   uint16_t* my_list = nlist + (size_t)blockIdx.x * (NLIST_CAP * TILE_THREADS) + tid;

   float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
   RowRanges r;
#pragma unroll
   for (int kk = 0; kk < 9; kk++) r.s[kk] = r.e[kk] = 0;
   if (live) {
      pi = posm[p];
#if !(defined(SPH_ABLATE) && (SPH_ABLATE == 4 || SPH_ABLATE == 5))
      int cx, cy, cz;
      cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
      row_ranges(g, cell_start, cx, cy, cz, r);
#endif
   }
   const uint32_t self_entry = (4u << QUEUE_TBITS) | (uint32_t)(p + L.desc.D[4]);
   const f32x2 px = {pi.x, pi.x}, py = {pi.y, pi.y}, pz = {pi.z, pi.z};

   TiledSum<UNIT_SCALE, UNIFORM_MASS, PASS> sum;
   if constexpr (PASS == 0) {
      sum.nlist = my_list;
      if (tid == 0) list_overflow = 0;
      __syncthreads();
   }
   if constexpr (PASS == 1) {
      accel_begin(k, sum.s, pi, live ? velp[p] : pi, live ? rho_in[p] : 0.0f);
      if (!nlist_overflow[blockIdx.x]) {
         // LIST path: the density pass left every lane's accepted neighbours (self excluded),
         // in canonical order, in this workgroup's list block
         const int cnt = live ? ncount[p] : 0;
         for (int j = 0; __any(j < cnt); ++j) {
            if (j < cnt) {
               Staged g;
               sum.stage_entry(g, my_list[j * TILE_THREADS], L, pi, posm, velp, aux_in);
               sum.pair(k, g, pi, 0xffffffffu);
            }
         }
         if (live) acc[p] = accel_end<UNIT_SCALE>(k, sum.s);
         return;
      }
   }

   int qn = 0;
#pragma unroll
   for (int kk = 0; kk < 9; kk++) {
      const int D = L.desc.D[kk];
      const uint32_t kbits = (uint32_t)kk << QUEUE_TBITS;
      const int ts = (int)r.s[kk] + D;
      const int te = (int)r.e[kk] + D;
      // chunks of 32 tile slots starting at an 8-aligned slot; one acceptance bit per slot
      for (int t0 = (ts < te) ? (ts & ~7) : te; __any(t0 < te); t0 += 32) {
         uint32_t mask = 0;
#if defined(SPH_ABLATE) && SPH_ABLATE == 2
         if (false) {
#else
         if (t0 < te) {
#endif
            mask = test8(L, t0, px, py, pz, k.h2);
            if (t0 + 8 < te) mask |= test8(L, t0 + 8, px, py, pz, k.h2) << 8;
            if (t0 + 16 < te) mask |= test8(L, t0 + 16, px, py, pz, k.h2) << 16;
            if (t0 + 24 < te) mask |= test8(L, t0 + 24, px, py, pz, k.h2) << 24;
            // keep only slots inside [ts, te)
            const int lo = ts - t0, hi = te - t0;
            if (lo > 0) mask &= ~0u << lo;
            if (hi < 32) mask &= ~(~0u << hi);
         }
         // expand the bits, ascending, into the lane's queue; drain whenever a queue fills
         while (__any(mask != 0u)) {
            if (mask != 0u && qn < QUEUE_DEPTH) {
               const int bit = __builtin_ctz(mask);
               mask &= mask - 1u;
               L.queue[qn * TILE_THREADS + tid] = (uint16_t)(kbits | (uint32_t)(t0 + bit));
               qn++;
            }
            if (__any(qn == QUEUE_DEPTH)) {
               drain_queue(sum, k, L, tid, qn, pi, self_entry, posm, velp, aux_in);
               qn = 0;
            }
         }
      }
   }
   drain_queue(sum, k, L, tid, qn, pi, self_entry, posm, velp, aux_in);

   if constexpr (PASS == 0) {
      if (sum.overflow) list_overflow = 1;
      __syncthreads();
      if (tid == 0) nlist_overflow[blockIdx.x] = (uint32_t)list_overflow;
   }
   if (live) {
      if constexpr (PASS == 0) {
         rho_out[p] = sum.density;
         aux_out[p] = neighbor_terms(k, sum.density, pi.w);
         ncount[p] = sum.count;
      } else {
         acc[p] = accel_end<UNIT_SCALE>(k, sum.s);
      }
   }
}
