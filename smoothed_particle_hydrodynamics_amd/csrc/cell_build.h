// Uniform-grid cell build for gfx950: cell hash + counting sort + in-cell ordering.
//
// Replaces SPH::clearGrid() / SPH::voxelizeParticles() (reference src/sph.cpp:429-481).  The
// reference appends particle indices to one QList per voxel in a serial loop, so every voxel
// list is in ascending particle index.  Here:
//   1. k_hash_count   cell id per particle, arrival slot from one counting atomic per cell run
//   2. k_scan_*       exclusive scan of the per-cell counts (wave64 shuffles + LDS carries; two
//                     launches: tile totals, then the scan with each tile's carry summed on the fly)
//   3. k_scatter      cell-sorted permutation (arbitrary order inside a cell)
//   4. k_rank_*       order inside each cell fixed to ascending particle index — the same
//                     lists the reference builds, independent of atomic arrival order
#pragma once

#include "sph_device.h"

#define SCAN_THREADS 256
#ifndef RANK_UNROLL
#define RANK_UNROLL 4   // cell members compared per trip of the in-cell ranking (FULL mode)
#endif
// Cells with more members than this are ranked by k_rank_big (sorted in LDS) instead of by every
// member scanning the whole cell: the scan is O(m^2) per cell, fine at the 8 particles per cell
// of a fluid at rest, ruinous for a cell that holds a large part of the scene - out-of-box
// particles are clamped into the edge cells exactly as the reference does (src/sph.cpp:456-463).
#ifndef RANK_BIG
#define RANK_BIG 512
#endif
#define RANK_CHUNK 4096   // members sorted at a time in LDS (key + source index: 32 KiB)
#define RANK_BIG_BLOCKS 64
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

// ---- 1. hash + count ------------------------------------------------------------------
// Lanes of a wave that hold consecutive particles of the same cell (the common case once the
// state is cell-sorted) share ONE global atomic: the run's first lane adds the run length and
// the others take base + their rank in the run.
// Counting step of the sort for entry i of cell c (all lanes of the wave call it; lanes that
// are not live pass c = 0xffffffff): run detection across the wave, one atomic per run.
// Entries of the trash cell (`trash` = its id: last step's ghosts, dead and departed entries of a
// slab) are not counted: nothing ever places them, and their thousands of runs would all add to
// one address.
__device__ __forceinline__ void count_cell_runs(uint32_t c, bool live, int i,
                                                uint32_t* __restrict__ cell_count,
                                                uint32_t* __restrict__ slot, uint32_t trash)
{
   live = live && c != trash;
   const int lane = threadIdx.x & (SPH_WAVE - 1);
   const uint32_t prev = __shfl_up(c, 1);
   const bool head = (lane == 0) || (prev != c);
   const unsigned long long heads = __ballot(head);
   // position of my run's head = highest set bit of heads at or below my lane
   const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
   const int head_lane = 63 - __clzll(below);
   // run length = distance to the next head above head_lane (or wave end)
   const unsigned long long above = (head_lane == 63) ? 0ull : (heads >> (head_lane + 1));
   const int run_len = above ? (__ffsll((long long)above)) : (SPH_WAVE - head_lane);
   uint32_t base = 0;
   if (live && head) base = atomicAdd(&cell_count[c], (uint32_t)run_len);
   base = __shfl(base, head_lane);
   if (live) slot[i] = base + (uint32_t)(lane - head_lane);
}

// CHECK_DEAD: entries may carry the dead id (only sph_hip_slab_pack writes it); otherwise the
// velocity/id array is not read at all.
template <bool WRITE_VOX, bool CHECK_DEAD>
__global__ void __launch_bounds__(256)
k_hash_count(const float4* __restrict__ posm, const float4* __restrict__ velp,
             int32_t* __restrict__ meta, CellGrid g, SlabZone zone, uint32_t* __restrict__ key,
             uint32_t* __restrict__ slot, uint32_t* __restrict__ cell_count,
             int32_t* __restrict__ vox)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   const bool live = i < meta[META_N_IN];
   uint32_t c = 0xffffffffu;
   if (live) {
      const float4 p = posm[i];
      int cx, cy, cz;
      c = cell_of(g, p.x, p.y, p.z, cx, cy, cz);
      // entries [0, n_live) are the previous sorted order (its owned range still in meta),
      // entries behind them were received from the neighbours since
      const bool carried = zone.drop_ghosts && i < meta[META_N_LIVE];
      const int ob = meta[META_OWN_BEGIN], oe = meta[META_OWN_END];
      const bool was_owned = i >= ob && i < oe;
      if (CHECK_DEAD && __float_as_uint(velp[i].w) == SPH_DEAD_ID) {
         c = (uint32_t)g.ncells;  // dropped ghost / departed particle
      } else if (carried && !was_owned) {
         c = (uint32_t)g.ncells;  // last step's ghost: its owner sends it again every step
      } else if (c == (uint32_t)g.ncells) {
         // an owned particle that left the slab and its halo was sent to its new owner (a
         // migrant); anything else out here is lost: a received entry that does not belong
         if (!carried) atomicOr(&meta[META_ERRORS], 1);
      } else if (zone.early && carried) {
         // The early exchange packed the planes next to the borders before the interior was
         // integrated.  An interior particle that ended up where it should have been sent
         // moved more than a cell plane in one step: nobody has it as a ghost - say so.
         const int lo_end = meta[META_BND_LO_END];
         const int hi_begin = max(meta[META_BND_HI_BEGIN], lo_end);
         if (i >= lo_end && i < hi_begin) {
            const int plane = cell_coord(p.z, g.inv, g.nz_global);
            if ((zone.have_left && plane < zone.lo + zone.halo) ||
                (zone.have_right && plane >= zone.hi - zone.halo))
               atomicOr(&meta[META_ERRORS], 8);
         }
      }
      key[i] = c;
      if (WRITE_VOX) {
         vox[3 * i + 0] = cx;
         vox[3 * i + 1] = cy;
         vox[3 * i + 2] = cz;
      }
   }
   count_cell_runs(c, live, i, cell_count, slot, (uint32_t)g.ncells);
}

// ---- 2. exclusive scan of cell counts ------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
   for (int d = 1; d < SPH_WAVE; d <<= 1) {
      uint32_t t = __shfl_up(v, d);
      if (lane >= d) v += t;
   }
   return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, total in *total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total)
{
   __shared__ uint32_t wave_sums[SCAN_THREADS / SPH_WAVE];
   const int lane = threadIdx.x & (SPH_WAVE - 1);
   const int w = threadIdx.x / SPH_WAVE;
   const uint32_t inc = wave_inclusive_scan(v, lane);
   if (lane == SPH_WAVE - 1) wave_sums[w] = inc;
   __syncthreads();
   uint32_t carry = 0, tot = 0;
#pragma unroll
   for (int k = 0; k < SCAN_THREADS / SPH_WAVE; k++) {
      const uint32_t s = wave_sums[k];
      if (k < w) carry += s;
      tot += s;
   }
   __syncthreads();
   *total = tot;
   return carry + inc - v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_reduce(const uint32_t* __restrict__ count, int ncells, uint32_t* __restrict__ part)
{
   const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
   uint32_t s = 0;
   if (base + SCAN_ITEMS <= ncells) {
      const uint4* p = reinterpret_cast<const uint4*>(count + base);
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS / 4; k++) {
         const uint4 v = p[k];
         s += v.x + v.y + v.z + v.w;
      }
   } else {
      for (int k = 0; k < SCAN_ITEMS; k++)
         if (base + k < ncells) s += count[base + k];
   }
   uint32_t total;
   block_exclusive_scan(s, &total);
   if (threadIdx.x == 0) part[blockIdx.x] = total;
}

// writes cell_start[0..ncells] and clears the counts for the next build.  The carry into a tile is
// the sum of the totals of the tiles before it (k_scan_reduce wrote them): every workgroup adds
// them up itself - a few KB out of L2 - instead of waiting for a one-workgroup scan launch of its
// own between the two kernels.
// limit = the context's capacity in entries: no cell_start value exceeds it, whatever the counts say.
// Counts that add up to more than the arrays hold (a slab exchange that delivered records twice, a
// step that counted its particles twice) then raise error bit 4 - sph_hip_slab_poll_errors /
// sph_hip_synchronize report SPH_HIP_ERR_EXCHANGE - instead of sending the scatter, the gather and
// the tile loads of this very step past the end of their arrays.
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_final(uint32_t* __restrict__ count, int ncells, const uint32_t* __restrict__ part,
             uint32_t* __restrict__ cell_start, uint32_t* __restrict__ big_cells, uint32_t limit,
             int32_t* __restrict__ meta)
{
   if (blockIdx.x == 0 && threadIdx.x == 0) big_cells[0] = 0u;   // k_scatter lists this build's big cells
   const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
   uint32_t v[SCAN_ITEMS];
   uint32_t s = 0;
   // A tile without particles (most of them when the fluid fills a corner of the box): its counts
   // are known - zero - and stay as they are; only cell_start = the carry is written.
   const bool empty_tile = part[blockIdx.x] == 0u;
   if (empty_tile) {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; k++) v[k] = 0u;
   } else if (base + SCAN_ITEMS <= ncells) {
      const uint4* p = reinterpret_cast<const uint4*>(count + base);
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS / 4; k++) {
         const uint4 q = p[k];
         v[4 * k + 0] = q.x;
         v[4 * k + 1] = q.y;
         v[4 * k + 2] = q.z;
         v[4 * k + 3] = q.w;
      }
   } else {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; k++) v[k] = (base + k < ncells) ? count[base + k] : 0u;
   }
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) s += v[k];
   uint32_t before = 0;
   for (int t = threadIdx.x; t < (int)blockIdx.x; t += SCAN_THREADS) before += part[t];
   uint32_t carry;
   block_exclusive_scan(before, &carry);   // carry = sum over the workgroup = totals of all earlier tiles
   uint32_t total;
   uint32_t run = carry + block_exclusive_scan(s, &total);
   if (base + SCAN_ITEMS <= ncells) {
      uint4* o = reinterpret_cast<uint4*>(cell_start + base);
      uint4* z = reinterpret_cast<uint4*>(count + base);
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS / 4; k++) {
         uint4 q;
         q.x = min(run, limit); run += v[4 * k + 0];
         q.y = min(run, limit); run += v[4 * k + 1];
         q.z = min(run, limit); run += v[4 * k + 2];
         q.w = min(run, limit); run += v[4 * k + 3];
         o[k] = q;
         // (counts that are zero already - nine cells in ten of a column in a corner of the box -
         // are not written again)
         if ((v[4 * k + 0] | v[4 * k + 1] | v[4 * k + 2] | v[4 * k + 3]) != 0u) z[k] = make_uint4(0, 0, 0, 0);
      }
   } else {
      for (int k = 0; k < SCAN_ITEMS; k++) {
         if (base + k < ncells) {
            cell_start[base + k] = min(run, limit);
            if (v[k] != 0u) count[base + k] = 0;
            run += v[k];
         }
      }
   }
   // the thread that owns the last cell also writes the end sentinel
   if (base <= ncells - 1 && ncells - 1 < base + SCAN_ITEMS) {
      cell_start[ncells] = min(run, limit);
      if (run > limit) atomicOr(&meta[META_ERRORS], 4);
   }
}

// ---- 3. scatter -----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_scatter(const uint32_t* __restrict__ key, const uint32_t* __restrict__ slot,
          const uint32_t* __restrict__ cell_start, int32_t* __restrict__ meta,
          uint32_t* __restrict__ perm, int cells_per_plane, int ncells, int own_lo, int own_hi,
          int sum_lo, int sum_hi, int bnd_lo, int bnd_hi, int32_t* __restrict__ tile_stats,
          int32_t* __restrict__ clear_a, int32_t* __restrict__ clear_b,
          uint32_t* __restrict__ big_cells, uint32_t limit)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   // tile statistics of the step: the descriptor pass accumulates into them
   if (tile_stats && i < TSTAT_COUNT && i != TSTAT_BLOCKS) tile_stats[i] = 0;
   const int n_in = meta[META_N_IN];
   if (i < n_in) {
      const uint32_t c = key[i];
      if (c != (uint32_t)ncells) {   // (the trash cell's entries are dropped: no place, no rank)
         const uint32_t first = cell_start[c], sl = slot[i];
         // (limit: k_scan_final clamped the cell ranges to the arrays' capacity and raised error bit
         // 4 when the counts exceed it; what does not fit is dropped here, not written past the end)
         if (first + sl < limit) perm[first + sl] = (uint32_t)i;
         // the cell's first arrival lists it when it is too crowded for the per-member ranking scan
         if (sl == 0u && cell_start[c + 1] - first > (uint32_t)RANK_BIG)
            big_cells[1u + atomicAdd(&big_cells[0], 1u)] = c;
      }
   }
   if (i == 0) {
      // Sorted ranges of the slab (kept out of a launch of their own: a kernel boundary costs
      // more than this does): live entries, owned planes [lo, hi), density planes one wider.
      // Planes are LOCAL indices.  Nothing in this launch reads these words.
      meta[META_N_LIVE] = (int)cell_start[ncells];
      meta[META_OWN_BEGIN] = (int)cell_start[own_lo * cells_per_plane];
      meta[META_OWN_END] = (int)cell_start[own_hi * cells_per_plane];
      meta[META_SUM_BEGIN] = (int)cell_start[sum_lo * cells_per_plane];
      meta[META_SUM_END] = (int)cell_start[sum_hi * cells_per_plane];
      meta[META_BND_LO_END] = (int)cell_start[bnd_lo * cells_per_plane];
      meta[META_BND_HI_BEGIN] = (int)cell_start[bnd_hi * cells_per_plane];
      // record counters of the messages this step will pack early (their previous contents
      // have been sent: the stream waited for that transfer before the last unpack)
      if (clear_a) *clear_a = 0;
      if (clear_b) *clear_b = 0;
      if (tile_stats)   // workgroups of the density range
         tile_stats[TSTAT_BLOCKS] =
            (meta[META_SUM_END] - meta[META_SUM_BEGIN] + TILE_THREADS - 1) / TILE_THREADS;
   }
}

// ---- 4a. REF: ascending-index order inside each cell ---------------------------------------
__global__ void __launch_bounds__(256)
k_rank_order(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ key,
             const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta,
             uint32_t* __restrict__ order)
{
   const int p = blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= meta[META_N_IN]) return;
   const uint32_t i = perm[p];
   const uint32_t c = key[i];
   const uint32_t s = cell_start[c], e = cell_start[c + 1];
   if (e - s > (uint32_t)RANK_BIG) return;   // a crowded cell: k_rank_big sorts it
   uint32_t rank = 0;
   // eight cell members per trip (independent loads; positions past the end re-read the last)
   for (uint32_t q0 = s; q0 < e; q0 += 8) {
      uint32_t other[8];
#pragma unroll
      for (int u = 0; u < 8; u++) other[u] = perm[q0 + u < e ? q0 + u : e - 1];
#pragma unroll
      for (int u = 0; u < 8; u++) rank += (q0 + u < e && other[u] < i) ? 1u : 0u;
   }
   order[s + rank] = i;
}

// ---- 4b. FULL: gather the state into cell-sorted order, ascending persistent id in a cell ----
// One thread per position p of the counting sort's order; rank = the members of p's cell with a
// smaller persistent id.  The members of a cell are consecutive positions, i.e. consecutive
// threads: every thread puts its own id into LDS and reads its cell mates' ids from there; only
// the part of a cell that lies in a neighbouring workgroup's range (the first and the last cell
// of a workgroup) is fetched through perm -> velp with two dependent loads per member, as all
// of it used to be (45 of the kernel's 100 us at 4M particles; staging a halo of ids on both
// sides as well measured 3-6 us slower than leaving those two cells to the loads).
// All threads of the workgroup must call this (barrier inside); lds_id holds 256 words.
__device__ __forceinline__ void
rank_gather(int p, int block_p0, uint32_t* __restrict__ lds_id, const uint32_t* __restrict__ perm,
            const uint32_t* __restrict__ key, const uint32_t* __restrict__ cell_start,
            const int32_t* __restrict__ meta, int trash, const float4* __restrict__ posm_in,
            const float4* __restrict__ velp_in, float4* __restrict__ posm_out,
            float4* __restrict__ velp_out, uint32_t* __restrict__ remap = nullptr)
{
   // positions [0, cell_start[trash]) hold the live entries; the trash cell's are not placed
   const int n_in = (int)cell_start[trash];
   bool active = p < n_in;
   uint32_t i = 0, c = 0, s = 0, e = 0, id = 0;
   float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
   if (active) {
      i = perm[p];
      c = key[i];
      v = velp_in[i];
      id = __float_as_uint(v.w);
   }
   lds_id[threadIdx.x] = id;
   if (active) {
      if (c == (uint32_t)trash) {
         active = false;  // dead entries are dropped: the live set is compacted
      } else {
         s = cell_start[c];
         e = cell_start[c + 1];
         if (e - s > (uint32_t)RANK_BIG) active = false;   // a crowded cell: k_rank_big sorts it
      }
   }
   __syncthreads();
   if (!active) return;
   const uint32_t b0 = (uint32_t)block_p0;
   const uint32_t b1 = min(b0 + blockDim.x, (uint32_t)n_in);
   const uint32_t in_lo = max(s, b0), in_hi = min(e, b1);   // p itself lies in [in_lo, in_hi)
   uint32_t rank = 0, same = 0;   // same: members of the cell with this very id (the entry itself: 1)
   for (uint32_t q0 = in_lo; q0 < in_hi; q0 += 4) {
      uint32_t other[4];
#pragma unroll
      for (int u = 0; u < 4; u++) other[u] = lds_id[(q0 + u < in_hi ? q0 + u : in_hi - 1) - b0];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         rank += (q0 + u < in_hi && other[u] < id) ? 1u : 0u;
         same += (q0 + u < in_hi && other[u] == id) ? 1u : 0u;
      }
   }
   // what is left of the cell outside this workgroup's positions: [s, in_lo) and [in_hi, e)
#pragma unroll
   for (int side = 0; side < 2; side++) {
      const uint32_t lo = side == 0 ? s : in_hi, hi = side == 0 ? in_lo : e;
      // RANK_UNROLL cell members per trip: their two dependent loads (perm, then id) overlap
      for (uint32_t q0 = lo; q0 < hi; q0 += RANK_UNROLL) {
         uint32_t other[RANK_UNROLL];
#pragma unroll
         for (int u = 0; u < RANK_UNROLL; u++) other[u] = perm[q0 + u < hi ? q0 + u : hi - 1];
#pragma unroll
         for (int u = 0; u < RANK_UNROLL; u++) other[u] = __float_as_uint(velp_in[other[u]].w);
#pragma unroll
         for (int u = 0; u < RANK_UNROLL; u++) {
            rank += (q0 + u < hi && other[u] < id) ? 1u : 0u;
            same += (q0 + u < hi && other[u] == id) ? 1u : 0u;
         }
      }
   }
   // A persistent id held twice by one cell (a halo record delivered twice, a particle two slabs both
   // think they own): the two would take the same place in the canonical order - one overwriting the
   // other, one position of the cell keeping whatever an earlier step left there.  Error bit 16.
   if (same > 1u) atomicOr(const_cast<int32_t*>(&meta[META_ERRORS]), 16);
   posm_out[s + rank] = posm_in[i];
   velp_out[s + rank] = v;
   if (remap) remap[i] = s + rank;   // stand-alone sph_hip_voxelize: where entry i went
}

__global__ void __launch_bounds__(256)
k_rank_gather(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ key,
              const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta, int trash,
              const float4* __restrict__ posm_in, const float4* __restrict__ velp_in,
              float4* __restrict__ posm_out, float4* __restrict__ velp_out,
              uint32_t* __restrict__ remap)
{
   __shared__ uint32_t lds_id[256];
   rank_gather(blockIdx.x * blockDim.x + threadIdx.x, blockIdx.x * blockDim.x, lds_id, perm, key,
               cell_start, meta, trash, posm_in, velp_in, posm_out, velp_out, remap);
}

// ---- 4c. crowded cells: sort by key in LDS, O(m log^2 m) instead of O(m^2) ----------------------
// One workgroup per listed cell (grid-stride over the list k_scatter made).  The cell's members are
// sorted RANK_CHUNK at a time by a bitonic network in LDS on (key, source index) - key = the
// persistent id (FULL) or the particle index itself (REF), both unique.  A cell of one chunk is
// placed straight from LDS.  A larger one writes its sorted chunks to a global scratch; every
// member's rank is then its index in its own chunk plus, for every other chunk, the number of
// smaller keys there (binary search): O(m * m / RANK_CHUNK * log RANK_CHUNK) loads, by the one
// workgroup that wrote the chunks (its own global writes are visible to it after a barrier).
template <bool FULL>
__global__ void __launch_bounds__(256)
k_rank_big(const uint32_t* __restrict__ big_cells, const uint32_t* __restrict__ perm,
           const uint32_t* __restrict__ cell_start, const float4* __restrict__ posm_in,
           const float4* __restrict__ velp_in, float4* __restrict__ posm_out,
           float4* __restrict__ velp_out, uint32_t* __restrict__ order,
           uint32_t* __restrict__ scratch_key, uint32_t* __restrict__ scratch_src,
           uint32_t* __restrict__ remap)
{
   __shared__ uint32_t skey[RANK_CHUNK], ssrc[RANK_CHUNK];
   const int tid = threadIdx.x;
   const uint32_t nbig = big_cells[0];
   for (uint32_t b = blockIdx.x; b < nbig; b += gridDim.x) {
      const uint32_t c = big_cells[1u + b];
      const uint32_t s = cell_start[c], m = cell_start[c + 1] - s;
      const uint32_t nchunks = (m + RANK_CHUNK - 1) / RANK_CHUNK;
      for (uint32_t ch = 0; ch < nchunks; ch++) {
         const uint32_t base = ch * RANK_CHUNK;
         const uint32_t len = min((uint32_t)RANK_CHUNK, m - base);
         uint32_t padded = 256;                        // power of two >= len
         while (padded < len) padded <<= 1;
         for (uint32_t t = tid; t < padded; t += 256) {
            uint32_t key = 0xffffffffu, src = 0u;      // padding sorts to the end
            if (t < len) {
               src = perm[s + base + t];
               key = FULL ? __float_as_uint(velp_in[src].w) : src;
            }
            skey[t] = key;
            ssrc[t] = src;
         }
         __syncthreads();
         for (uint32_t k2 = 2; k2 <= padded; k2 <<= 1) {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
               for (uint32_t t = tid; t < padded; t += 256) {
                  const uint32_t partner = t ^ j;
                  if (partner > t) {
                     const bool up = (t & k2) == 0;
                     const uint32_t a = skey[t], bb = skey[partner];
                     if ((a > bb) == up) {
                        skey[t] = bb;
                        skey[partner] = a;
                        const uint32_t sa = ssrc[t];
                        ssrc[t] = ssrc[partner];
                        ssrc[partner] = sa;
                     }
                  }
               }
               __syncthreads();
            }
         }
         for (uint32_t t = tid; t < len; t += 256) {
            if (nchunks == 1) {                        // ascending key = final order of the cell
               const uint32_t src = ssrc[t];
               if (FULL) {
                  posm_out[s + t] = posm_in[src];
                  velp_out[s + t] = velp_in[src];
                  if (remap) remap[src] = s + t;
               } else {
                  order[s + t] = src;
               }
            } else {
               scratch_key[s + base + t] = skey[t];
               scratch_src[s + base + t] = ssrc[t];
            }
         }
         __syncthreads();
      }
      if (nchunks > 1) {
         for (uint32_t idx = tid; idx < m; idx += 256) {
            const uint32_t key = scratch_key[s + idx], mine = idx / RANK_CHUNK;
            uint32_t rank = idx % RANK_CHUNK;
            for (uint32_t ch = 0; ch < nchunks; ch++) {
               if (ch == mine) continue;
               const uint32_t* keys = scratch_key + s + ch * RANK_CHUNK;
               uint32_t lo = 0, hi = min((uint32_t)RANK_CHUNK, m - ch * RANK_CHUNK);
               while (lo < hi) {                       // first position with keys[pos] >= key
                  const uint32_t mid = (lo + hi) >> 1;
                  if (keys[mid] < key) lo = mid + 1;
                  else hi = mid;
               }
               rank += lo;
            }
            const uint32_t src = scratch_src[s + idx];
            if (FULL) {
               posm_out[s + rank] = posm_in[src];
               velp_out[s + rank] = velp_in[src];
               if (remap) remap[src] = s + rank;
            } else {
               order[s + rank] = src;
            }
         }
         __syncthreads();
      }
   }
}

// Stand-alone sph_hip_voxelize (FULL mode re-sorts the state): the per-particle results of the
// last sums follow their particles into the new order (remap[i] = new position of entry i), as
// SPH::voxelizeParticles() leaves Particle::mDensity / mAcceleration / mNeighborCount valid.
__global__ void __launch_bounds__(256)
k_permute_sums(const uint32_t* __restrict__ remap, const uint32_t* __restrict__ key,
               const int32_t* __restrict__ meta, uint32_t trash, const float* __restrict__ rho,
               const float4* __restrict__ acc, const int32_t* __restrict__ ncount,
               float* __restrict__ rho_out, float4* __restrict__ acc_out,
               int32_t* __restrict__ ncount_out)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= meta[META_N_IN] || key[i] == trash) return;
   const uint32_t j = remap[i];
   rho_out[j] = rho[i];
   acc_out[j] = acc[i];
   ncount_out[j] = ncount[i];
}
