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
#define SLAB_PACK_BLOCKS 4096  // fixed grid of the early pack (grid-stride over the border planes)

struct SlabMsg {
   int32_t header[SLAB_HEADER_INTS];
   float4 rec[1];  // [2 * capacity]: posm, velp pairs
};

// Record slot in a message for every lane that wants one: the wave's first such lane reserves the
// wave's records with ONE atomic on the message's count word and the others take their place
// behind it.  (One atomic per record - a hundred thousand of them on one address per message and
// step - was most of both pack kernels' time.)  All lanes of the wave must call it.
__device__ __forceinline__ int msg_reserve(int32_t* __restrict__ counter, bool want)
{
   const unsigned long long m = __ballot(want);
   if (m == 0ull) return -1;
   const int lane = threadIdx.x & (SPH_WAVE - 1);
   const int leader = __ffsll((long long)m) - 1;
   int base = 0;
   if (lane == leader) base = atomicAdd(counter, __popcll(m));
   base = __shfl(base, leader);
   return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

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
   const bool live = p < meta[META_N_LIVE];
   float4 v = make_float4(0.f, 0.f, 0.f, 0.f), x = v;
   if (live) v = velp[p];
   const bool alive = live && __float_as_uint(v.w) != SPH_DEAD_ID;
   const bool owned = alive && p >= meta[META_OWN_BEGIN] && p < meta[META_OWN_END];
   bool drop = alive && !owned;
   bool to_left = false, to_right = false;
   if (owned) {
      x = posm[p];
      const int plane = cell_coord(x.z, g.inv, g.nz_global);
      to_left = have_left && plane < lo + halo;
      to_right = have_right && plane >= hi - halo;
      drop = plane < g.z0 || plane >= g.z0 + g.nz;  // migrated beyond the halo
   }
   if (have_left) {
      const int s = msg_reserve(&left->header[0], to_left);
      if (to_left) {
         if (s < capacity) {
            left->rec[2 * s] = x;
            left->rec[2 * s + 1] = v;
         } else {
            atomicOr(&meta[META_ERRORS], 2);
         }
      }
   }
   if (have_right) {
      const int s = msg_reserve(&right->header[0], to_right);
      if (to_right) {
         if (s < capacity) {
            right->rec[2 * s] = x;
            right->rec[2 * s + 1] = v;
         } else {
            atomicOr(&meta[META_ERRORS], 2);
         }
      }
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
   // (whole waves stay in the loop together: msg_reserve is a wave-wide operation)
   for (int q0 = blockIdx.x * blockDim.x; q0 < n_left + n_right; q0 += gridDim.x * blockDim.x) {
      const int q = q0 + threadIdx.x;
      bool to_left = false, to_right = false;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f), v = x;
      if (q < n_left + n_right) {
         const int p = q < n_left ? ob + q : hi_begin + (q - n_left);
         x = posm[p];
         v = velp[p];
         if (__float_as_uint(v.w) != SPH_DEAD_ID) {
            double ke, pe;
            integrate_particle<UNIT_SCALE>(k, x, v, acc[p], ke, pe);
            const int plane = cell_coord(x.z, g.inv, g.nz_global);
            to_left = zone.have_left && plane < zone.lo + zone.halo;
            to_right = zone.have_right && plane >= zone.hi - zone.halo;
         }
      }
      if (zone.have_left) {
         const int s = msg_reserve(&left->header[0], to_left);
         if (to_left) {
            if (s < capacity) {
               left->rec[2 * s] = x;
               left->rec[2 * s + 1] = v;
            } else {
               atomicOr(&meta[META_ERRORS], 2);
            }
         }
      }
      if (zone.have_right) {
         const int s = msg_reserve(&right->header[0], to_right);
         if (to_right) {
            if (s < capacity) {
               right->rec[2 * s] = x;
               right->rec[2 * s + 1] = v;
            } else {
               atomicOr(&meta[META_ERRORS], 2);
            }
         }
      }
   }
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (left) left->header[1] = capacity;
      if (right) right->header[1] = capacity;
   }
}

// Cell build of a slab whose last step was integrated and hashed by its acceleration pass
// (FusedStep): key, slot and the cells' counts of the owned entries of the previous sorted order
// [OWN_BEGIN, OWN_END) are already there.  What is left: that order's other entries - last step's
// ghosts, re-sent by their owner every step - go to the trash cell, and the records received since
// (entries [N_LIVE, N_IN)) are hashed and counted like k_hash_count does.  Grid-stride over the
// entries outside the owned range.
__global__ void __launch_bounds__(256)
k_hash_tail(const float4* __restrict__ posm, int32_t* __restrict__ meta, CellGrid g,
            uint32_t* __restrict__ key, uint32_t* __restrict__ slot, uint32_t* __restrict__ cell_count)
{
   const int ob = meta[META_OWN_BEGIN], oe = meta[META_OWN_END];
   const int n_live = meta[META_N_LIVE], n_in = meta[META_N_IN];
   const int todo = ob + (n_in - oe);
   // (whole waves stay in the loop together: count_cell_runs is a wave-wide operation)
   for (int q0 = blockIdx.x * blockDim.x; q0 < todo; q0 += gridDim.x * blockDim.x) {
      const int q = q0 + threadIdx.x;
      const bool live = q < todo;
      const int i = q < ob ? q : oe + (q - ob);
      uint32_t c = 0xffffffffu;
      if (live) {
         if (i < n_live) {
            c = (uint32_t)g.ncells;          // last step's ghost
         } else {
            const float4 p = posm[i];
            int cx, cy, cz;
            c = cell_of(g, p.x, p.y, p.z, cx, cy, cz);
            if (c == (uint32_t)g.ncells) atomicOr(&meta[META_ERRORS], 1);   // received, but not ours to hold
         }
         key[i] = c;
      }
      count_cell_runs(c, live, i, cell_count, slot, (uint32_t)g.ncells);
   }
}

// Device-to-device re-partitioning (slab.py rebalance()): the owned particles as message records
// {x,y,z,m | vx,vy,vz,id} in cell-sorted order, and a slab's state from such records.
__global__ void __launch_bounds__(256)
k_export_records(const float4* __restrict__ posm, const float4* __restrict__ velp,
                 const int32_t* __restrict__ meta, float4* __restrict__ rec)
{
   const int ob = meta[META_OWN_BEGIN];
   const int p = ob + blockIdx.x * blockDim.x + threadIdx.x;
   if (p >= meta[META_OWN_END]) return;
   rec[2 * (p - ob)] = posm[p];
   rec[2 * (p - ob) + 1] = velp[p];
}

__global__ void __launch_bounds__(256)
k_import_records(const float4* __restrict__ rec, int n, float4* __restrict__ posm,
                 float4* __restrict__ velp)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   posm[i] = rec[2 * i];
   velp[i] = rec[2 * i + 1];
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
