// Slab exchange kernels (FULL mode, multi-GPU): what crosses xGMI between neighbouring slabs.
//
// A message is a fixed-capacity device buffer: 8 int32 header words (word 0 = record count,
// word 1 = capacity it was packed for) followed by 32-byte records {x,y,z,m | vx,vy,vz,id}.
// Each step a slab sends, to each neighbour, every owned particle that now lies within
// `halo` planes of that neighbour's territory or beyond it: the same record serves as ghost
// (still ours) or as migrant (now theirs) — the receiver decides by the particle's plane.
#pragma once

#include "sph_device.h"

#define SLAB_HEADER_INTS 8

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

// Appends the records of one received message behind the current entries.  offset_slot:
// meta word holding the first free entry; it is advanced by the message's count.
__global__ void __launch_bounds__(256)
k_slab_unpack(const SlabMsg* __restrict__ msg, float4* __restrict__ posm,
              float4* __restrict__ velp, int32_t* __restrict__ meta, int first_free_word,
              int capacity_entries, int msg_capacity)
{
   int count = msg->header[0];
   if (count > msg_capacity) count = msg_capacity;  // sender flagged the overflow on its side
   const int base = meta[first_free_word];
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= count) return;
   if (base + i >= capacity_entries) {
      atomicOr(&meta[META_ERRORS], 4);
      return;
   }
   posm[base + i] = msg->rec[2 * i];
   velp[base + i] = msg->rec[2 * i + 1];
}

// n_in = n_live (+ counts of the messages just appended); one thread.
__global__ void k_slab_set_n_in(int32_t* __restrict__ meta, const SlabMsg* __restrict__ a,
                                const SlabMsg* __restrict__ b, int capacity_entries,
                                int msg_capacity, int stage)
{
   if (threadIdx.x != 0 || blockIdx.x != 0) return;
   // stage 0: n_in = n_live; stage 1: += count(a); stage 2: += count(b)
   if (stage == 0) meta[META_N_IN] = meta[META_N_LIVE];
   const SlabMsg* m = stage == 1 ? a : (stage == 2 ? b : nullptr);
   if (m) {
      int c = m->header[0];
      if (c > msg_capacity) c = msg_capacity;
      int v = meta[META_N_IN] + c;
      if (v > capacity_entries) v = capacity_entries;
      meta[META_N_IN] = v;
   }
}
