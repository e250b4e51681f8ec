// Slab exchange kernels (FULL mode, multi-GPU): what crosses xGMI between neighbouring slabs.
//
// A message is a fixed-capacity device buffer: 8 int32 header words (word 0 = record count,
// word 1 = capacity it was packed for) followed by 32-byte records {x,y,z,m | vx,vy,vz,id}.
// Each step a slab sends, to each neighbour, every owned particle that now lies within
// `halo` planes of that neighbour's territory or beyond it: the same record serves as ghost
// (still ours) or as migrant (now theirs) — the receiver decides by the particle's plane.
#pragma once

#include "common_kernels.h"
#include "sph_device.h"

#define SLAB_HEADER_INTS 8
#define SLAB_PACK_BLOCKS 512   // fixed grid of the early pack (grid-stride over the border planes)

struct SlabMsg {
   int32_t header[SLAB_HEADER_INTS];
   float4 rec[1];  // [2 * capacity]: posm, velp pairs
};

// After integrate: classify every live entry of the sorted state.
//   ghosts (outside the owned range)         -> dropped (re-sent by their owner every step)
//   owned, now in plane <  lo + halo          -> copied to the left message
//   owned, now in plane >= hi - halo          -> copied to the right message
//   owned, now outside [lo, hi)               -> it is in one of the messages (a migrant); it stays
//                                                here as a ghost for one step if it is still
//                                                inside the planes this slab holds — its new
//                                                owner cannot send it back before the next step
// Planes are GLOBAL indices; have_left/right say whether that neighbour exists.
__global__ void __launch_bounds__(256)
k_slab_pack(const float4* __restrict__ posm, float4* __restrict__ velp, int32_t* __restrict__ meta,
            CellGrid g, int lo, int hi, int halo, int have_left, int have_right,
            SlabMsg* __restrict__ left, SlabMsg* __restrict__ right, int capacity)
{
   const int p = blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= meta[META_N_LIVE]) return;
   float4 v = velp[p];
   if (__float_as_uint(v.w) == SPH_DEAD_ID) return;
   const bool owned = p >= meta[META_OWN_BEGIN] && p < meta[META_OWN_END];
   bool drop = !owned;
   if (owned) {
      const float4 x = posm[p];
      const int plane = cell_coord(x.z, g.inv, g.nz_global);
      if (have_left && plane < lo + halo) {
         const int s = atomicAdd(&left->header[0], 1);
         if (s < capacity) {
            left->rec[2 * s] = x;
            left->rec[2 * s + 1] = v;
         } else {
            atomicOr(&meta[META_ERRORS], 2);
         }
      }
      if (have_right && plane >= hi - halo) {
         const int s = atomicAdd(&right->header[0], 1);
         if (s < capacity) {
            right->rec[2 * s] = x;
            right->rec[2 * s + 1] = v;
         } else {
            atomicOr(&meta[META_ERRORS], 2);
         }
      }
      drop = plane < g.z0 || plane >= g.z0 + g.nz;  // migrated beyond the halo
   }
   if (drop) {
      v.w = __uint_as_float(SPH_DEAD_ID);
      velp[p] = v;
   }
}

// Early exchange: the same messages, built BEFORE the step's integrate has run, so that they
// travel while the interior's acceleration is computed.  Covers only the owned planes next to a
// neighbouring slab (sorted ranges [OWN_BEGIN, BND_LO_END) and [BND_HI_BEGIN, OWN_END), one plane
// wider than the halo), whose acceleration is already known; each of their particles is
// integrated here on the fly - the same arithmetic k_integrate applies to the state later - and
// classified by its NEW plane exactly as k_slab_pack does.  The state is not touched: last step's
// ghosts and departed particles are recognised by the next cell build from their sorted position
// (k_hash_count), which also flags an interior particle that should have been sent.
// Fixed grid, grid-stride.  The record counters are the headers' count words themselves, zeroed
// by this step's k_scatter (a "last workgroup publishes the count" scheme needs a device-scope
// fence per workgroup, which on this multi-L2 chip is an L2 write-back: it cost 60 us).
template <bool UNIT_SCALE>
__global__ void __launch_bounds__(256)
k_slab_pack_early(const float4* __restrict__ posm, const float4* __restrict__ velp,
                  const float4* __restrict__ acc, int32_t* __restrict__ meta, PairConsts k,
                  CellGrid g, SlabZone zone, SlabMsg* __restrict__ left,
                  SlabMsg* __restrict__ right, int capacity)
{
   const int ob = meta[META_OWN_BEGIN], oe = meta[META_OWN_END];
   const int lo_end = min(meta[META_BND_LO_END], oe);
   const int hi_begin = min(max(meta[META_BND_HI_BEGIN], lo_end), oe);
   const int n_left = lo_end - ob, n_right = oe - hi_begin;
   for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n_left + n_right;
        q += gridDim.x * blockDim.x) {
      const int p = q < n_left ? ob + q : hi_begin + (q - n_left);
      float4 x = posm[p];
      float4 v = velp[p];
      if (__float_as_uint(v.w) == SPH_DEAD_ID) continue;
      double ke, pe;
      integrate_particle<UNIT_SCALE>(k, x, v, acc[p], ke, pe);
      const int plane = cell_coord(x.z, g.inv, g.nz_global);
      if (zone.have_left && plane < zone.lo + zone.halo) {
         const int s = atomicAdd(&left->header[0], 1);
         if (s < capacity) {
            left->rec[2 * s] = x;
            left->rec[2 * s + 1] = v;
         } else {
            atomicOr(&meta[META_ERRORS], 2);
         }
      }
      if (zone.have_right && plane >= zone.hi - zone.halo) {
         const int s = atomicAdd(&right->header[0], 1);
         if (s < capacity) {
            right->rec[2 * s] = x;
            right->rec[2 * s + 1] = v;
         } else {
            atomicOr(&meta[META_ERRORS], 2);
         }
      }
   }
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (left) left->header[1] = capacity;
      if (right) right->header[1] = capacity;
   }
}

// Appends the records of the received messages behind the live entries [0, n_live) and sets
// n_in, the entry count of the next cell build.  One launch for both messages.
__global__ void __launch_bounds__(256)
k_slab_unpack(const SlabMsg* __restrict__ left, const SlabMsg* __restrict__ right,
              float4* __restrict__ posm, float4* __restrict__ velp, int32_t* __restrict__ meta,
              int capacity_entries, int msg_capacity)
{
   const int base = meta[META_N_LIVE];
   const int cl = left ? min(left->header[0], msg_capacity) : 0;   // senders flag overflows
   const int cr = right ? min(right->header[0], msg_capacity) : 0;
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < cl + cr) {
      const SlabMsg* m = i < cl ? left : right;
      const int r = i < cl ? i : i - cl;
      if (base + i >= capacity_entries) {
         atomicOr(&meta[META_ERRORS], 4);
      } else {
         posm[base + i] = m->rec[2 * r];
         velp[base + i] = m->rec[2 * r + 1];
      }
   }
   if (i == 0) meta[META_N_IN] = min(base + cl + cr, capacity_entries);
}
