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
// Workgroups whose tile would not fit (very dense regions) are flagged and run the untiled
// code of full_kernels.h inline — same results, slower, no extra launch.
#pragma once

#include "common_kernels.h"
#include "full_kernels.h"
#include "slab_kernels.h"

// SPH_ABLATE=N cuts a piece out of a kernel so that tools/ablate.py can price it: such a build
// computes garbage.  It only compiles as a declared diagnostic build (tools/build_variant.sh adds
// -DSPH_DIAGNOSTIC_BUILD), and a diagnostic build reports itself through sph_hip_abi_version(), which
// the Python binding refuses unless SPH_HIP_ALLOW_DIAGNOSTIC=1: no test or bench loads one by accident.
#if defined(SPH_ABLATE) && !defined(SPH_DIAGNOSTIC_BUILD)
#error "SPH_ABLATE is for diagnostic builds only: add -DSPH_DIAGNOSTIC_BUILD (tools/build_variant.sh does)"
#endif

// SPH_TRIPCOUNT (diagnostic builds only, tools/trip_counts.py): counts how often the loops of the two
// pair kernels run - per wave ("W": once per wave whenever ANY lane takes the trip: what the SIMD
// issues) and per lane ("L": what the particles need) - into g_trip[], read by sph_hip_diag_trips().
// Results stay correct; the timing of such a build means nothing.
#ifdef SPH_TRIPCOUNT
#ifndef SPH_DIAGNOSTIC_BUILD
#error "SPH_TRIPCOUNT is for diagnostic builds only: add -DSPH_DIAGNOSTIC_BUILD"
#endif
enum {
   TRIP_D_WAVES = 0, TRIP_D_CHUNKS_W, TRIP_D_TEST8_W, TRIP_D_TEST8_L, TRIP_D_SLOTS_L, TRIP_D_POPS_W,
   TRIP_D_POPS_L, TRIP_D_POPLOOPS_W, TRIP_D_SUMTRIPS_W, TRIP_D_SUMSLOTS_W, TRIP_D_LISTED_L, TRIP_D_LANES_L,
   TRIP_A_WAVES = 16, TRIP_A_PTRIPS_W, TRIP_A_CNT_L, TRIP_A_VTRIPS_W, TRIP_A_NV_L, TRIP_A_LANES_L,
   TRIP_COUNT = 32
};
__device__ unsigned long long g_trip[TRIP_COUNT];
struct TripCounters {
   unsigned int v[16];
   __device__ TripCounters() { for (int i = 0; i < 16; i++) v[i] = 0; }
   // per-wave counter: every lane adds the same value; flush takes lane 0's
   __device__ void wave(int i, bool any_lane) { v[i] += any_lane ? 1u : 0u; }
   __device__ void lane(int i, unsigned int n) { v[i] += n; }
   __device__ void flush(int base, unsigned int lane_mask /* bit i: v[i] is a per-lane sum */)
   {
      for (int i = 0; i < 16; i++) {
         unsigned int x = v[i];
         if (lane_mask >> i & 1u)
            for (int o = SPH_WAVE / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
         if ((threadIdx.x & (SPH_WAVE - 1)) == 0 && x) atomicAdd(&g_trip[base + i], (unsigned long long)x);
      }
   }
};
#define TRIP(x) x
#else
#define TRIP(x)
#endif

// SPH_PHASECLOCK (diagnostic builds only, tools/phase_clock.py): where a workgroup of the two tiled
// pair kernels spends its life - thread 0 reads the constant-rate clock (100 MHz) at the marks and adds
// the time since the last mark to g_phase[kernel][workgroup][phase] (marks 0.. = density, 16.. =
// acceleration; no atomics: a first version with six contended atomicAdd per workgroup took the
// density pass from 0.55 to 0.92 ms).  What is left of a workgroup's residency (launch duration x workgroups per CU / workgroups
// of a CU) after the marked phases is the drain of its stores and the dispatch of its successor.
#ifdef SPH_PHASECLOCK
#ifndef SPH_DIAGNOSTIC_BUILD
#error "SPH_PHASECLOCK is for diagnostic builds only: add -DSPH_DIAGNOSTIC_BUILD"
#endif
#define PHASE_WGS 65536
__device__ unsigned int g_phase[2][PHASE_WGS][8];   // [kernel][workgroup][phase]: 10 ns ticks, [7] = 1 when written
#define PHASE_BEGIN() unsigned long long ph_t = threadIdx.x == 0 ? wall_clock64() : 0ull
#define PHASE_MARK(i)                                                                               \
   if (threadIdx.x == 0 && blockIdx.x < PHASE_WGS) {                                                \
      const unsigned long long ph_now = wall_clock64();                                             \
      g_phase[(i) >> 4][blockIdx.x][(i) & 7] = (unsigned int)(ph_now - ph_t);                       \
      ph_t = ph_now;                                                                                \
   }
#define PHASE_COUNT(i) if (threadIdx.x == 0 && blockIdx.x < PHASE_WGS) g_phase[(i) >> 4][blockIdx.x][7] = 1u
#else
#define PHASE_BEGIN()
#define PHASE_MARK(i)
#define PHASE_COUNT(i)
#endif

// The tile lives in dynamic LDS: its capacity (candidate positions per workgroup) is a launch
// parameter, chosen by the host from the tile sizes the previous steps needed, because the
// workgroups a CU can hold (and with them the latency hiding of both passes) is set by the LDS
// a workgroup asks for: 12 B (density) / 16 B (acceleration) per tile entry.
#define TILE_PAD 32                      // slots past the capacity that aligned 8-slot reads may touch
#define TILE_CAP_MAX (4096 - TILE_PAD)        // tile indices are 12-bit in narrow list entries
#define TILE_CAP_MAX_WIDE (16384 - TILE_PAD)  // ... 14-bit in wide ones
#ifndef TILE_BATCH
#define TILE_BATCH 8
#endif
// (TILE_BATCH: 16-byte loads a thread keeps in flight while filling the tile)
#define DENSITY_TILE_BYTES 12
#define ACCEL_TILE_BYTES 16
// launch bounds = the most workgroups per CU the register budget should allow
#ifndef DENSITY_BLOCKS
#define DENSITY_BLOCKS (6 * 256 / TILE_THREADS)
#endif
#ifndef ACCEL_BLOCKS
#define ACCEL_BLOCKS (TILE_THREADS == 256 ? 5 : 2)
#endif
// Neighbour lists handed from the density pass to the acceleration pass, 16-bit entries, per
// workgroup a block of list_rows(list_cap) * TILE_THREADS 32-bit words laid out in BLOCKS OF EIGHT
// ENTRIES: entries 8b .. 8b+7 of lane t are the 16 bytes at b * LIST_BLOCK_BYTES + 16 t.  A consumer
// fetches a trip's eight entries with one 16-byte load per lane (a wave: 1 KB contiguous); the
// producer fills a 16-byte slot with eight consecutive 2-byte stores of ONE lane, so a 128-byte
// line is complete after eight lanes have taken eight pops each and leaves the L2 whole.  (Until
// round 4 entry j sat in half (j & 1) of word (j >> 1) * TILE_THREADS + t: a line was shared by 32
// lanes x 2 entries and stayed open until the slowest of them got there - in a compressed scene,
// 150 entries per particle and three workgroups per CU, the open lines outgrew the L2 several
// times over and the density pass WROTE 9 GB for 1.3 GB of entries: profiles/r4_notes.md.)
// Only blocks in use are ever touched.  list_cap - the neighbours per particle the lists hold - is a
// launch argument: a context starts with NLIST_CAP and the host enlarges it (up to NLIST_CAP_MAX,
// memory permitting) when the density pass reports particles that went without a list.
#ifndef NLIST_CAP
#define NLIST_CAP 254
#endif
#define NLIST_CAP_MAX 1022
#define LIST_BLOCK_ENTRIES 8
#define LIST_BLOCK_BYTES (16 * TILE_THREADS)
// rows of TILE_THREADS words a workgroup's list block takes: whole blocks for entries 0 .. list_cap
__host__ __device__ __forceinline__ constexpr int list_rows(int list_cap)
{
   return 4 * ((list_cap + LIST_BLOCK_ENTRIES) / LIST_BLOCK_ENTRIES);
}
// byte offset of entry j of the lane whose slots start at lane_off (= 16 * lane) in the block
__device__ __forceinline__ uint32_t list_entry_off(uint32_t j, uint32_t lane_off)
{
   return (j >> 3) * (uint32_t)LIST_BLOCK_BYTES + lane_off + (j & 7u) * 2u;
}
// The append's running position.  pos holds block and slot of the next entry with the lane field
// (bits 4 .. 4 + log2(TILE_THREADS)) ALL ONES: pos += 2 then carries from the slot field straight
// into the block field when a 16-byte slot is full (and leaves the lane field zero: or it back).
// The address puts the lane in: one v_bfi.  Three instructions per entry, none of them a shift.
#define LIST_LANE_FIELD ((uint32_t)(LIST_BLOCK_BYTES - 16))
__device__ __forceinline__ uint32_t list_pos_of(uint32_t j)
{
   return ((j >> 3) * (uint32_t)LIST_BLOCK_BYTES + (j & 7u) * 2u) | LIST_LANE_FIELD;
}
__device__ __forceinline__ uint32_t list_pos_off(uint32_t pos, uint32_t lane_off)
{
   return (LIST_LANE_FIELD & lane_off) | (~LIST_LANE_FIELD & pos);      // v_bfi_b32
}
__device__ __forceinline__ uint32_t list_pos_next(uint32_t pos) { return (pos + 2u) | LIST_LANE_FIELD; }
__device__ __forceinline__ uint32_t list_entry_load(const char* __restrict__ lists, uint32_t j, uint32_t lane_off)
{
   return *reinterpret_cast<const uint16_t*>(lists + list_entry_off(j, lane_off));
}
// the eight entries of block b of that lane
__device__ __forceinline__ uint4 list_block_load(const char* __restrict__ lists, int b, uint32_t lane_off)
{
   return *reinterpret_cast<const uint4*>(lists + (uint32_t)b * (uint32_t)LIST_BLOCK_BYTES + lane_off);
}
__device__ __forceinline__ void list_block_entries(const uint4& blk, uint32_t (&entry)[8])
{
   entry[0] = blk.x & 0xffffu; entry[1] = blk.x >> 16;
   entry[2] = blk.y & 0xffffu; entry[3] = blk.y >> 16;
   entry[4] = blk.z & 0xffffu; entry[5] = blk.z >> 16;
   entry[6] = blk.w & 0xffffu; entry[7] = blk.w >> 16;
}
// Zeroes the entries between a list's end and the end of its last block (0 is a valid tile index):
// the bit-exact acceleration loop gathers by every entry of a fetched block before it looks at the
// count, and what an earlier step left there need not be an index of this step's tile.
__device__ __forceinline__ void list_pad(char* __restrict__ lists, uint32_t lane_off, int count, int list_cap)
{
   if (count >= list_cap + 1) return;
   uint32_t c = (uint32_t)count;
   if (c & 1u) { *reinterpret_cast<uint16_t*>(lists + list_entry_off(c, lane_off)) = (uint16_t)0; c += 1u; }
   if (c & 2u) { *reinterpret_cast<uint32_t*>(lists + list_entry_off(c, lane_off)) = 0u; c += 2u; }
   if (c & 4u) { *reinterpret_cast<uint2*>(lists + list_entry_off(c, lane_off)) = make_uint2(0u, 0u); }
}
// first list word of a particle that has no list (more neighbours than list_cap): no valid
// entry has segment id 15
#define NLIST_NO_LIST 0xffffffffu
static_assert(NLIST_CAP <= NLIST_CAP_MAX, "initial list capacity");
// list entries fetched per trip of the SUM loops (measured best on MI355X: 8 / 8; the density
// pass ran best with 6 until the end of round 2, now 8 or 10 are 5-10 us ahead of it at 4M)
#ifndef DENSITY_UNROLL
#define DENSITY_UNROLL 8
#endif
#ifndef APPEND_POPS
#define APPEND_POPS 4     // accepted candidates appended per trip of the append loop
#endif
// The tiled density pass appends through LDS: a lane collects the eight entries of its current list
// block in a 16-byte slot of its own and writes the block with ONE 16-byte store when it is full (and
// the last, partial one once at the end).  The texture addresser was busy 80 % of that pass with the
// 2-byte stores of the append - one per accepted neighbour, each to a 16-byte slot of its own, 13 L2
// requests per wave-instruction - and every other load of the CU queued behind them: with seven stores in
// eight left out (a probe, lists wrong) a workgroup's life went from 55 to 45 us, its prologue from 13 to
// 9 us (tools/phase_clock.py, tools/pmc_latency.sh, profiles/r4_notes.md 4c).
#ifndef APPEND_STAGED
#define APPEND_STAGED 1
#endif
#ifndef ACCEL_UNROLL
#define ACCEL_UNROLL 8
#endif
#ifndef VISC_UNROLL
#define VISC_UNROLL 4     // neighbours per trip of the tolerance-mode viscous sum (the list's last few)
#endif
// A 16-bit list entry is a tile index plus what it takes to get back from it to the neighbour's
// sorted position (tile index - D[segment]).  Narrow (tiles up to 4064 entries): segment id << 12 |
// 12-bit index.  Wide (scenes several times denser, tiles up to 16352 entries, chosen per step by
// the host): plane dz + 1 << 14 | 14-bit index; the row inside the plane follows from the index
// and the descriptor's segment starts B - two more LDS reads and compares per neighbour, which is
// why it is not the only format.
template <bool WIDE>
struct ListEntry {
   static constexpr int TBITS = WIDE ? 14 : 12;
   static constexpr uint32_t TMASK = (1u << TBITS) - 1u;
   __device__ static __forceinline__ uint32_t tag(int segment)
   {
      return (uint32_t)(WIDE ? segment / 3 : segment) << TBITS;
   }
   __device__ static __forceinline__ int tile(uint32_t e) { return (int)(e & TMASK); }
   // D of the entry's segment.  Segments of a plane that share storage have equal D, and an
   // empty segment starts where the next begins, so comparing with the starts finds a valid one.
   template <class Desc>
   __device__ static __forceinline__ int shift(const Desc& d, uint32_t e)
   {
      if (!WIDE) return d.D[e >> TBITS];
      const int z = (int)(e >> TBITS), t = (int)(e & TMASK);
      const int k = 3 * z + (t >= d.B[3 * z + 1] ? 1 : 0) + (t >= d.B[3 * z + 2] ? 1 : 0);
      return d.D[k];
   }
};

typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef float __attribute__((ext_vector_type(4))) f32x4;

// 16-byte LDS read of four consecutive floats; i must be a multiple of 4
__device__ __forceinline__ f32x4 lds_read4(const float* base, int i)
{
   return *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(base + i, 16));
}

static_assert(DENSITY_UNROLL == LIST_BLOCK_ENTRIES && ACCEL_UNROLL == LIST_BLOCK_ENTRIES, "a trip of the list-driven loops is one block of the lists");
static_assert(TILE_CAP_MAX + TILE_PAD <= (1 << ListEntry<false>::TBITS) &&
                 TILE_CAP_MAX_WIDE + TILE_PAD <= (1 << ListEntry<true>::TBITS),
              "tile index must fit the list entry");

// Per-workgroup tile layout, computed by k_tile_desc before the sums run.
struct TileDesc {
   int D[9];      // tile index = sorted index + D[k] inside segment k
   int B[9];      // first tile index of segment k (B[0] = 0)
   int total;     // tile entries; > the launch's tile capacity => the workgroup runs untiled
   int pad;
};

// SoA tile of the density pass inside the dynamic LDS block: three arrays of cap + TILE_PAD floats
struct TileLds {
   float* x;
   float* y;
   float* z;
};

extern __shared__ __attribute__((aligned(16))) float tile_lds_dynamic[];

// One thread per workgroup-to-be: the 9 row segments (cells c_first-1 .. c_last+1 of every
// (dz,dy) row, as linear cell-id ranges) of the 256 particles starting at tile*256, as ranges
// of the sorted arrays plus their placement in the LDS tile.  Needs only the counting sort's
// key/perm/cell_start (not the sorted state), so it shares a launch with the gather below.
__device__ __forceinline__ void
tile_desc(int tile, int ntiles, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ key,
          const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta, CellGrid g,
          TileDesc* __restrict__ desc, const TileCaps& caps, int32_t* __restrict__ stats,
          uint32_t* __restrict__ giveup_density, uint32_t* __restrict__ giveup_accel)
{
   const int begin = meta[META_SUM_BEGIN], end = meta[META_SUM_END];
   const int p0 = begin + tile * TILE_THREADS;
   int most = -1;   // -1: no workgroup here
   if (tile < ntiles && p0 < end) {
      const int plast = min(p0 + TILE_THREADS - 1, end - 1);
      // position p of the sorted order lies in the cell of perm[p] (ordered by id inside the
      // cell only after the gather, but the cell is the same)
      const int c_first = (int)key[perm[p0]];
      const int c_last = (int)key[perm[plast]];
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
      // Segments come in ascending sorted position.  Where consecutive ones overlap or touch
      // (short rows: a 256-particle span then covers several rows, and the dy = -1, 0, +1
      // segments of a plane are nearly the same range) they share tile storage: same index
      // shift D, and only the part past the previous segment's end is new.  B[k] = first tile
      // index of segment k's new part, so "tile index t in [B[k], B[k+1])  <->  sorted index
      // t - D[k]" holds for the loader.
      int run = 0, chain_end = -1, chain_d = 0;
#pragma unroll
      for (int k = 0; k < 9; k++) {
         d.B[k] = run;
         if (len[k] == 0) {          // nothing to load, no lane range refers to it
            d.D[k] = run - G[k];
         } else if (G[k] <= chain_end) {
            const int seg_end = G[k] + len[k];
            d.D[k] = chain_d;
            if (seg_end > chain_end) {
               run += seg_end - chain_end;
               chain_end = seg_end;
            }
         } else {
            d.D[k] = chain_d = run - G[k];
            run += len[k];
            chain_end = G[k] + len[k];
         }
      }
      d.total = run;
      d.pad = 0;
      desc[tile] = d;
      most = run;
   }
   // Statistics for the host's next choice of capacity (k_scatter zeroed them): one atomic per
   // wave and counter, and only for counters that move.
   const bool lead = (threadIdx.x & (SPH_WAVE - 1)) == 0;
#pragma unroll
   for (int c = 0; c < TILE_CANDS; c++) {
      const int over = __popcll(__ballot(c < caps.n_cand && most > caps.cand[c]));
      if (lead && over) atomicAdd(&stats[TSTAT_OVER + c], over);
   }
   int wave_most = most;
#pragma unroll
   for (int o = SPH_WAVE / 2; o > 0; o >>= 1) wave_most = max(wave_most, __shfl_xor(wave_most, o));
   if (lead && wave_most > stats[TSTAT_MAX]) atomicMax(&stats[TSTAT_MAX], wave_most);
   // Workgroups whose tile does not fit this step's capacities: listed, so that the first
   // workgroups of the two passes compute them (untiled) before anything else - their long
   // latency then overlaps the rest of the launch instead of trailing it.
   if (most > caps.cap_density) giveup_density[atomicAdd(&stats[TSTAT_GIVEUP_DENSITY], 1)] = (uint32_t)tile;
   if (most > caps.cap_density || most > caps.cap_accel)
      giveup_accel[atomicAdd(&stats[TSTAT_GIVEUP_ACCEL], 1)] = (uint32_t)tile;
}

// Last launch of the FULL-mode cell build: the first `desc_blocks` workgroups lay out the tiles
// (a short chain of dependent loads, hidden behind the gather), the others gather the state
// into cell-sorted order.
__global__ void __launch_bounds__(256)
k_rank_gather_tile_desc(int desc_blocks, int ntiles, const uint32_t* __restrict__ perm,
                        const uint32_t* __restrict__ key, const uint32_t* __restrict__ cell_start,
                        const int32_t* __restrict__ meta, CellGrid g,
                        const float4* __restrict__ posm_in, const float4* __restrict__ velp_in,
                        float4* __restrict__ posm_out, float4* __restrict__ velp_out,
                        TileDesc* __restrict__ desc, TileCaps caps, int32_t* __restrict__ stats,
                        uint32_t* __restrict__ giveup_density, uint32_t* __restrict__ giveup_accel,
                        uint32_t* __restrict__ remap)
{
   if ((int)blockIdx.x < desc_blocks) {
      tile_desc(blockIdx.x * blockDim.x + threadIdx.x, ntiles, perm, key, cell_start, meta, g, desc,
                caps, stats, giveup_density, giveup_accel);
      return;
   }
   __shared__ uint32_t lds_id[256];
   rank_gather((blockIdx.x - desc_blocks) * blockDim.x + threadIdx.x,
               (blockIdx.x - desc_blocks) * blockDim.x, lds_id, perm, key, cell_start, meta,
               g.ncells, posm_in, velp_in, posm_out, velp_out, remap);
}

__device__ __forceinline__ void tile_desc_load(const TileDesc* __restrict__ desc, int wg, TileDesc& sd)
{
   const int tid = threadIdx.x;
   if (tid < (int)(sizeof(TileDesc) / sizeof(int)))
      reinterpret_cast<int*>(&sd)[tid] = reinterpret_cast<const int*>(&desc[wg])[tid];
   __syncthreads();
}

// Which 256-particle workgroup a hardware workgroup computes.  The dispatcher deals consecutive
// blockIdx round-robin over the 8 XCDs, each with its own L2, while neighbouring workgroups share
// most of their tile (each sorted particle is staged by ~6 workgroups): give every residue class
// of blockIdx mod 8 a contiguous eighth of the workgroups, so the shared rows hit one L2.
// Bijective for any grid size.  Only a placement hint: results do not depend on it.
__device__ __forceinline__ int xcd_workgroup(int block, int nblocks)
{
   const int q = nblocks / 8, r = nblocks % 8, x = block % 8;
   return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + block / 8;
}

// Copies the candidate positions of the workgroup's tile into LDS, TILE_BATCH 16-byte loads per
// thread in flight before the first LDS store of a batch.  The descriptor is already in LDS.
__device__ __forceinline__ void tile_load(const float4* __restrict__ posm, const TileDesc& sd,
                                          const TileLds& L)
{
   const int tid = threadIdx.x;
   const int total = sd.total;
   int B[9], D[9];
#pragma unroll
   for (int k = 0; k < 9; k++) {
      B[k] = sd.B[k];
      D[k] = sd.D[k];
   }
#if defined(SPH_ABLATE) && (SPH_ABLATE == 3 || SPH_ABLATE == 5)
   for (int base = 0; base < 0; base += TILE_BATCH * TILE_THREADS) {
#else
   for (int base = 0; base < total; base += TILE_BATCH * TILE_THREADS) {
#endif
      // unconditional loads (index clamped into the tile): a load inside a divergent branch gets
      // its own s_waitcnt from the compiler, which serialises the whole batch
      float4 buf[TILE_BATCH];
#pragma unroll
      for (int r = 0; r < TILE_BATCH; r++) {
         const int idx = min(base + tid + r * TILE_THREADS, total - 1);
         int d = D[0];
#pragma unroll
         for (int k = 1; k < 9; k++) d = (idx >= B[k]) ? D[k] : d;
         buf[r] = posm[idx - d];
      }
#pragma unroll
      for (int r = 0; r < TILE_BATCH; r++) {
         const int idx = base + tid + r * TILE_THREADS;
         if (idx < total) {
            L.x[idx] = buf[r].x;
            L.y[idx] = buf[r].y;
            L.z[idx] = buf[r].z;
         }
      }
   }
   __syncthreads();
}

// TEST screens, SUM confirms.  The reference accepts a neighbour iff the fp32 value
// ((dx*dx) + (dy*dy)) + (dz*dz), five separate roundings, is < h2.  TEST evaluates the sum with
// two fused multiply-adds instead (6 packed ops per candidate pair instead of 8; the 8 ops were a
// sixth of this pass's arithmetic), with -h2 * (1 + 2e-6) as the innermost addend, and takes the
// sign.  Near the threshold the two evaluations are within 2.5 + 1.5 ulp(h2) < 5e-7 * h2 of the
// true value, so every neighbour the exact test accepts passes the screen; a candidate that passes it wrongly (about one in 10^6) is
// caught by SUM, which needs the exact value of every listed pair for the distance anyway: it
// leaves the pair out of the sum and out of the list (the entries behind it move up), so neighbour
// lists and counts are exactly the reference's.

// d2 - h2_screen of two candidates at once, three fused multiply-adds: for screening only (its
// sign is the screening bit; the subtraction rides in the first FMA's addend)
__device__ __forceinline__ f32x2 screen_pair(f32x2 px, f32x2 py, f32x2 pz, f32x2 cx, f32x2 cy,
                                             f32x2 cz, f32x2 minus_h2)
{
   const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
   return __builtin_elementwise_fma(
      dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, minus_h2)));
}

// TEST step: eight consecutive tile slots t..t+7 (t a multiple of 4: two 16-byte reads per
// coordinate) -> 8 screening bits (h2 is the
// widened threshold).  Six independent ds_read_b128 and branch-free packed math; slots outside
// the lane's range are masked by the caller.
__device__ __forceinline__ uint32_t test8(const TileLds& L, int t, f32x2 px, f32x2 py, f32x2 pz,
                                         float h2)
{
   const f32x4 X0 = lds_read4(L.x, t);
   const f32x4 X1 = lds_read4(L.x, t + 4);
#if defined(SPH_ABLATE) && SPH_ABLATE == 14
   const f32x4 Y0 = lds_read4(L.y, t);           // timing only: three LDS reads, no sign gathering
   const f32x4 Y1 = X0;
#else
   const f32x4 Y0 = lds_read4(L.y, t);
   const f32x4 Y1 = lds_read4(L.y, t + 4);
#endif
#if defined(SPH_ABLATE) && SPH_ABLATE == 14
   const f32x4 Z0 = X1, Z1 = Y0;
#elif defined(SPH_ABLATE) && SPH_ABLATE == 12
   const f32x4 Z0 = Y0 + X1, Z1 = Y1 + X0;   // timing only: the same arithmetic on four LDS reads instead of six
#else
   const f32x4 Z0 = lds_read4(L.z, t);
   const f32x4 Z1 = lds_read4(L.z, t + 4);
#endif
   const f32x2 mh = {-h2, -h2};
   const f32x2 da = screen_pair(px, py, pz, f32x2{X0.x, X0.y}, f32x2{Y0.x, Y0.y}, f32x2{Z0.x, Z0.y}, mh);
   const f32x2 db = screen_pair(px, py, pz, f32x2{X0.z, X0.w}, f32x2{Y0.z, Y0.w}, f32x2{Z0.z, Z0.w}, mh);
   const f32x2 dc = screen_pair(px, py, pz, f32x2{X1.x, X1.y}, f32x2{Y1.x, Y1.y}, f32x2{Z1.x, Z1.y}, mh);
   const f32x2 dd = screen_pair(px, py, pz, f32x2{X1.z, X1.w}, f32x2{Y1.z, Y1.w}, f32x2{Z1.z, Z1.w}, mh);
   // inside the screen  <=>  sign bit of (d2 - h2); v_alignbit shifts the mask left and brings the
   // next sign in at bit 0 - slot 7 first, so that slot 0 ends in bit 0
#if defined(SPH_ABLATE) && SPH_ABLATE == 14
   {  // four packed operations and a byte permute in place of the eight v_alignbit
      f32x2 acc = {4.0f, 4.0f};
      acc = __builtin_elementwise_fma(acc, f32x2{2.f, 2.f}, da);
      acc = __builtin_elementwise_fma(acc, f32x2{2.f, 2.f}, db);
      acc = __builtin_elementwise_fma(acc, f32x2{2.f, 2.f}, dc);
      acc = __builtin_elementwise_fma(acc, f32x2{2.f, 2.f}, dd);
      return __builtin_amdgcn_perm(__float_as_uint(acc.x), __float_as_uint(acc.y), 0x07050301u) & 0x00010001u;
   }
#endif
   uint32_t m = 0;
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dd.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dd.x), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dc.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(dc.x), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(db.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(db.x), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(da.y), 31);
   m = __builtin_amdgcn_alignbit(m, __float_as_uint(da.x), 31);
   return m;
}

// ---- density pass: TILE + TEST (-> neighbour list) + SUM (<- neighbour list) --------------------
// TEST appends every accepted neighbour (self excluded), in canonical order, to the lane's
// column of the workgroup's list block in global memory; SUM then walks that list once.  The
// list doubles as the input of the acceleration pass.  A workgroup in which some particle has
// more than NLIST_CAP neighbours gives up (flag) and runs the untiled code inline instead.
// (launch bounds: the wide-entry instantiations run at four workgroups per CU at most - LDS - and may
// take the registers that leaves them: 128)
template <bool UNIT_SCALE, bool UNIFORM_MASS, bool WIDE, bool FAST>
__global__ void __launch_bounds__(TILE_THREADS, WIDE ? (DENSITY_BLOCKS < 4 ? DENSITY_BLOCKS : 4) : DENSITY_BLOCKS)
k_full_density_tiled(const float4* __restrict__ posm, const float4* __restrict__ velp,
                     const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta,
                     CellGrid g, PairConsts k, float* __restrict__ rho_out,
                     float4* __restrict__ velB_out, float* __restrict__ auxc_out,
                     int32_t* __restrict__ ncount, const TileDesc* __restrict__ desc,
                     uint32_t* __restrict__ nlist, uint32_t* __restrict__ nlist_overflow,
                     int tile_cap, int32_t* __restrict__ tile_stats,
                     const uint32_t* __restrict__ giveup, int* __restrict__ tile_feedback, int list_cap,
                     double* __restrict__ epart_clear, int inline_giveups)
{
   // A slab whose acceleration pass integrates (FusedStep) writes one pair of energy partial sums
   // per workgroup that owns particles; the others' must read zero whatever an earlier step left
   PHASE_BEGIN();
   if (epart_clear && threadIdx.x < 2) epart_clear[2 * blockIdx.x + threadIdx.x] = 0.0;
   __shared__ TileDesc sd;
   __shared__ int list_overflow;
   TileLds L;
   L.x = tile_lds_dynamic;
   L.y = L.x + (tile_cap + TILE_PAD);
   L.z = L.y + (tile_cap + TILE_PAD);
   // this step's tile statistics, for the host's next choice of capacity (plain stores to pinned
   // host memory; read there without synchronisation, only a hint)
   // (TSTAT_NO_LIST is counted by this launch: the acceleration pass hands it over)
   if (blockIdx.x == 0 && threadIdx.x < TSTAT_NO_LIST) tile_feedback[threadIdx.x] = tile_stats[threadIdx.x];

   const int begin = meta[META_SUM_BEGIN];
   const int end = meta[META_SUM_END];
   const int tid = threadIdx.x;
   // the first workgroups of the launch start with the workgroups whose tile does not fit
   // (give-up list), untiled: their long latency then overlaps the rest of the launch
   // (inline_giveups = 0: k_full_density_chunked, launched before this kernel, has done them)
   if (inline_giveups && (int)blockIdx.x < tile_stats[TSTAT_GIVEUP_DENSITY]) {
      const int gp = begin + (int)giveup[blockIdx.x] * TILE_THREADS + tid;
      if (gp < end)
         density_untiled<UNIT_SCALE, FAST>(gp, posm, cell_start, velp, g, k, rho_out, velB_out,
                                           auxc_out, ncount);
   }
   // (the mapping spreads the workgroups that exist over the XCDs: a slab context launches one
   // workgroup per 256 particles of its CAPACITY, and dealing eighths of that grid left the XCDs
   // of the empty tail idle - 3 of 8 with the usual 1.5x head room)
   const int wgs = tile_stats[TSTAT_BLOCKS];
   const int wg = (int)blockIdx.x < wgs ? xcd_workgroup(blockIdx.x, wgs) : (int)blockIdx.x;
   const int p0 = begin + wg * TILE_THREADS;
   if (p0 >= end) return;
   const int p = p0 + tid;
   const bool live = p < end;
   if (tid == 0) list_overflow = 0;
   // Prologue ordered to shorten the dependent-latency chain: the particle's own position is
   // requested first, then the tile descriptor; the 18 cell_start lookups of the lane's ranges go
   // out before the tile's loads, so that both are in flight together.
   float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
   if (live) pi = posm[p];
   tile_desc_load(desc, wg, sd);
   if (sd.total > tile_cap) {
      // tile does not fit: on the give-up lists, computed by the first workgroups of both passes
      // (or, with its lists, by k_full_density_chunked, which then has set the flag itself)
      if (inline_giveups && tid == 0) nlist_overflow[wg] = 1u;
      return;
   }
#if APPEND_STAGED
   // The three arrays are packed to the tile's own size; what the launch's capacity leaves free behind
   // them holds the append's staging slots (16 bytes per lane).  A tile that leaves less than that - one
   // workgroup in a hundred at the benchmark's density - appends with 2-byte stores as before round 4.
   char* stage_lane = nullptr;
   // (workgroup-uniform, and said so: a scalar branch around each pop's store, not an exec mask)
   const int packed = __builtin_amdgcn_readfirstlane(((sd.total + 31) & ~31) + TILE_PAD);
   const bool staged = (tile_cap + TILE_PAD - packed) * (int)DENSITY_TILE_BYTES >= 16 * TILE_THREADS;
   if (staged) {
      L.y = L.x + packed;
      L.z = L.y + packed;
      stage_lane = reinterpret_cast<char*>(L.z + packed) + 16 * tid;
      *reinterpret_cast<uint4*>(stage_lane) = make_uint4(0u, 0u, 0u, 0u);
   }
#endif
   RowRanges r;
#pragma unroll
   for (int kk = 0; kk < 9; kk++) r.s[kk] = r.e[kk] = 0;
   if (live) {
#if !(defined(SPH_ABLATE) && (SPH_ABLATE == 4 || SPH_ABLATE == 5))
      int cx, cy, cz;
      cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
      row_ranges(g, cell_start, cx, cy, cz, r);
#endif
   }
   PHASE_MARK(0);   // descriptor, own position, row ranges
   tile_load(posm, sd, L);
   PHASE_MARK(1);   // tile
   const int self_t = p + sd.D[4];
   const f32x2 px = {pi.x, pi.x}, py = {pi.y, pi.y}, pz = {pi.z, pi.z};
   // uniform base of the workgroup's list block; lanes address it with 32-bit offsets
   uint32_t* list_block = nlist + (size_t)wg * (size_t)(list_rows(list_cap) * TILE_THREADS);

   const float h2_screen = k.h2_screen;
   TRIP(TripCounters trips; trips.wave(TRIP_D_WAVES, true); trips.lane(TRIP_D_LANES_L, live ? 1u : 0u);)
   int count = 0;
   // TEST stores every accepted neighbour with one 2-byte store (no pairing of entries in registers:
   // the append loop runs to the largest popcount among the wave's lanes for every chunk, so what
   // counts is instructions per trip); pos = where this lane's next entry goes (list_pos_of).
   char* const lists = reinterpret_cast<char*>(list_block);
   const uint32_t lane_off = 16u * (uint32_t)tid;
   uint32_t pos = list_pos_of(0u);
   // TEST + append over the nine rows, compiled twice: for a workgroup that stages its appends in LDS
   // (nearly all) and for one whose tile leaves no room (a branch on that in every pop cost 10 us at 4M)
   auto test_and_append = [&](auto staged_c) {
      constexpr bool STAGED = decltype(staged_c)::value;
#pragma unroll
      for (int kk = 0; kk < 9; kk++) {
         const int D = sd.D[kk];
         const uint32_t kbits = ListEntry<WIDE>::tag(kk);
         const int ts = (int)r.s[kk] + D;
         const int te = (int)r.e[kk] + D;
         // chunks of 32 tile slots starting at a 4-aligned slot; one acceptance bit per slot
         TRIP(trips.lane(TRIP_D_SLOTS_L, te > ts ? (unsigned)(te - ts) : 0u);)
         for (int t0 = (ts < te) ? (ts & ~3) : te; __any(t0 < te); t0 += 32) {
            uint32_t mask = 0;
            TRIP(trips.wave(TRIP_D_CHUNKS_W, true);
                 for (int q8 = 0; q8 < 4; q8++) {
                    trips.wave(TRIP_D_TEST8_W, __any(t0 + 8 * q8 < te));
                    trips.lane(TRIP_D_TEST8_L, t0 + 8 * q8 < te ? 1u : 0u);
                 })
#if defined(SPH_ABLATE) && SPH_ABLATE == 2
            if (false) {
#else
            if (t0 < te) {
#endif
               mask = test8(L, t0, px, py, pz, h2_screen);
               if (t0 + 8 < te) mask |= test8(L, t0 + 8, px, py, pz, h2_screen) << 8;
               if (t0 + 16 < te) mask |= test8(L, t0 + 16, px, py, pz, h2_screen) << 16;
               if (t0 + 24 < te) mask |= test8(L, t0 + 24, px, py, pz, h2_screen) << 24;
               // keep only slots inside [ts, te), and not the particle itself
               const int lo = ts - t0, hi = te - t0;
               if (lo > 0) mask &= ~0u << lo;
               if (hi < 32) mask &= ~(~0u << hi);
               if (kk == 4) {
                  const int sb = self_t - t0;
                  if (sb >= 0 && sb < 32) mask &= ~(1u << sb);
               }
               // a list that would overflow stops growing here (checked per chunk, not per entry):
               // the particle goes without a list anyway
               count += __builtin_popcount(mask);
               if (count > list_cap) {
                  mask = 0u;
                  count = list_cap + 1;
               }
            }
#if defined(SPH_ABLATE) && (SPH_ABLATE == 9 || SPH_ABLATE == 12 || SPH_ABLATE == 14 || SPH_ABLATE == 15)
            mask = 0u;   // timing only: no lists (15: and no SUM either - tools/valu_census.py)
#endif
            // append the set bits, ascending, to the lane's neighbour list
            const uint32_t ebase = kbits | (uint32_t)t0;
            while (__any(mask != 0u)) {
               TRIP(trips.wave(TRIP_D_POPLOOPS_W, true);)
#pragma unroll
               for (int rep = 0; rep < APPEND_POPS; rep++) {   // several pops per trip: less loop control
                  TRIP(trips.wave(TRIP_D_POPS_W, __any(mask != 0u)); trips.lane(TRIP_D_POPS_L, mask != 0u ? 1u : 0u);)
                  if (mask != 0u) {
                     const uint32_t bit = (uint32_t)__builtin_ctz(mask);
                     mask &= mask - 1u;
#if APPEND_STAGED
                     const uint32_t in_block = pos & 14u;   // the entry's two bytes inside its 16-byte block
                     if constexpr (STAGED) {
                        *reinterpret_cast<uint16_t*>(stage_lane + in_block) = (uint16_t)(ebase + bit);
                        if (in_block == 14u)                // the block is complete: one 16-byte store
                           *reinterpret_cast<uint4*>(lists + list_pos_off(pos & ~14u, lane_off)) =
                              *reinterpret_cast<const uint4*>(stage_lane);
                     } else {
                        *reinterpret_cast<uint16_t*>(lists + list_pos_off(pos, lane_off)) = (uint16_t)(ebase + bit);
                     }
#else
                     *reinterpret_cast<uint16_t*>(lists + list_pos_off(pos, lane_off)) = (uint16_t)(ebase + bit);
#endif
                     pos = list_pos_next(pos);
                  }
               }
            }
         }
      }
   };
#if APPEND_STAGED
   if (staged) test_and_append(std::true_type());
   else test_and_append(std::false_type());
#else
   test_and_append(std::false_type());
#endif
   PHASE_MARK(2);   // TEST + append
#if APPEND_STAGED
   // The last block, partly filled (or block 0 of a lane without neighbours): the slot as it is - what
   // lies behind the list's end are entries of the lane's previous block or the zeros the slot started
   // with, valid indices of this tile either way, which is all list_pad's zeros are there for.
   if (!staged)
      list_pad(lists, lane_off, count, list_cap);
   else if (count <= list_cap && ((count & 7) != 0 || count == 0))
      *reinterpret_cast<uint4*>(lists + list_pos_off(pos & ~14u, lane_off)) = *reinterpret_cast<const uint4*>(stage_lane);
#else
   // (the rest of the list's last block: zeros)
   list_pad(lists, lane_off, count, list_cap);
#endif
   // A particle with more neighbours than its list holds (a scene many times denser than the
   // benchmark's) goes without a list: its lane walks its candidate ranges in the tile one by one
   // here - canonical order, small code - and again in the acceleration pass, which recognises it
   // by the marker in the list's first word; the other lanes of the workgroup keep their lists.
   const bool overflowed = count > list_cap;
   if (overflowed) {
      list_overflow = 1;
      *reinterpret_cast<uint32_t*>(lists + lane_off) = NLIST_NO_LIST;
   }
   if (__any(overflowed)) {
      // reported to the host (through the acceleration pass), which then enlarges the lists
      const int without = __popcll(__ballot(overflowed));
      if ((tid & (SPH_WAVE - 1)) == 0) atomicAdd(&tile_stats[TSTAT_NO_LIST], without);
   }
   __syncthreads();
   if (tid == 0) nlist_overflow[wg] = list_overflow ? 2u : 0u;
   float density = 0.0f;
   if (__any(overflowed) && overflowed) {
      count = 0;
      RowRanges rr;   // looked up again rather than kept alive through TEST in every workgroup
#pragma unroll
      for (int kk = 0; kk < 9; kk++) rr.s[kk] = rr.e[kk] = 0;
      if (live) {
         int cx, cy, cz;
         cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
         row_ranges(g, cell_start, cx, cy, cz, rr);
      }
#pragma unroll
      for (int kk = 0; kk < 9; kk++) {
         const int D = sd.D[kk];
         const int te = (int)rr.e[kk] + D;
         for (int t = (int)rr.s[kk] + D; t < te; t++) {
            float dx, dy, dz;
            const float d2 = dist2(pi.x, pi.y, pi.z, L.x[t], L.y[t], L.z[t], dx, dy, dz);
            if (d2 < k.h2 && !(kk == 4 && t == self_t)) {
               float mj = pi.w;
               if (!UNIFORM_MASS) mj = posm[t - D].w;
               {
                  float d = sqrt_rn(d2);
                  if (!UNIT_SCALE) d *= k.sim_scale;
                  density_accumulate<UNIT_SCALE>(k, mj, d, density);
               }
               count++;
            }
         }
      }
   }

   // SUM: one pass over the list, in canonical order.  TEST only screened: SUM applies the
   // reference's own test to the reference's own value of every listed pair, leaves a pair that
   // fails it out of the sum and moves the entries behind it up, so that lists and counts are exactly
   // the reference's (the write position never passes the read position, and a trip's words are in
   // registers before its first store).
   PHASE_MARK(3);   // pad, particles without a list
   const int listed = overflowed ? 0 : count;   // entries to sum from the list
   const int lastb = listed > 0 ? (listed - 1) >> 3 : 0;
   int kept = 0;
   TRIP(trips.lane(TRIP_D_LISTED_L, (unsigned)listed);)
   // (the next trip's block travels while this trip's entries are summed; the compaction below only
   // ever writes behind the current trip's read position, never into the block already fetched)
   uint4 sum_blk = list_block_load(lists, 0, lane_off);
   for (int j0 = 0; __any(j0 < listed); j0 += DENSITY_UNROLL) {
      TRIP(trips.wave(TRIP_D_SUMTRIPS_W, true);
           for (int u = 0; u < DENSITY_UNROLL; u++) trips.wave(TRIP_D_SUMSLOTS_W, __any(j0 + u < listed));)
#if !(defined(SPH_ABLATE) && (SPH_ABLATE == 1 || SPH_ABLATE == 15))
      uint32_t entry[DENSITY_UNROLL];
      // (the trip's eight entries: one 16-byte load; a lane past its last block takes that one again)
      list_block_entries(sum_blk, entry);
      sum_blk = list_block_load(lists, min((j0 >> 3) + 1, lastb), lane_off);
      if constexpr (WIDE) {
         // Scenes several times denser than the benchmark's (wide entries = tiles beyond 4064
         // positions = at most four workgroups per CU): few waves share a SIMD, so the latency of a
         // neighbour's chain - three LDS reads, the distance, the root, the term - is hidden by the
         // trip's other neighbours or not at all.  The eight entries as ONE basic block: all reads
         // first, the roots together, the terms added in list order (a neighbour past the count or
         // failing the reference's test adds +0.0f); the registers are there at this occupancy.
         float d2[DENSITY_UNROLL], mj[DENSITY_UNROLL];
         bool ok[DENSITY_UNROLL];
#pragma unroll
         for (int u = 0; u < DENSITY_UNROLL; u++) {
            const bool valid = j0 + u < listed;
            const int t = valid ? ListEntry<WIDE>::tile(entry[u]) : 0;
            mj[u] = pi.w;
            if (!UNIFORM_MASS) mj[u] = posm[valid ? t - ListEntry<WIDE>::shift(sd, entry[u]) : p0].w;
            float dx, dy, dz;
            d2[u] = dist2(pi.x, pi.y, pi.z, L.x[t], L.y[t], L.z[t], dx, dy, dz);
            ok[u] = valid && d2[u] < k.h2;      // the reference's own test, on the reference's own value
         }
         float dd[DENSITY_UNROLL];
#pragma unroll
         for (int u = 0; u < DENSITY_UNROLL; u++) dd[u] = ok[u] ? d2[u] : 0.0f;
         sqrt_rn_batch(dd);
#pragma unroll
         for (int u = 0; u < DENSITY_UNROLL; u++) {
            float d = dd[u];
            if (!UNIT_SCALE) d *= k.sim_scale;
            const float term = density_term<UNIT_SCALE>(k, mj[u], d);
            density += ok[u] ? term : 0.0f;
            if (ok[u] && kept != j0 + u)
               *reinterpret_cast<uint16_t*>(lists + list_entry_off((uint32_t)kept, lane_off)) = (uint16_t)entry[u];
            kept += ok[u] ? 1 : 0;
         }
      } else {
#pragma unroll
      for (int u = 0; u < DENSITY_UNROLL; u++) {
         if (j0 + u < listed) {
            const int t = ListEntry<WIDE>::tile(entry[u]);
            float mj = pi.w;
            if (!UNIFORM_MASS) mj = posm[t - ListEntry<WIDE>::shift(sd, entry[u])].w;
            float dx, dy, dz;
            const float d2 = dist2(pi.x, pi.y, pi.z, L.x[t], L.y[t], L.z[t], dx, dy, dz);
            if (d2 < k.h2) {             // the reference's own test, on the reference's own value
               {
                  float d = sqrt_rn(d2);
                  if (!UNIT_SCALE) d *= k.sim_scale;
                  density_accumulate<UNIT_SCALE>(k, mj, d, density);
               }
               if (kept != j0 + u) *reinterpret_cast<uint16_t*>(lists + list_entry_off((uint32_t)kept, lane_off)) = (uint16_t)entry[u];
               kept++;
            }
         }
      }
      }
#else
      kept = listed;
#endif
   }
   if (!overflowed && kept != count) {
      count = kept;
      list_pad(lists, lane_off, count, list_cap);
   }
   TRIP(trips.flush(0, 1u << TRIP_D_TEST8_L | 1u << TRIP_D_SLOTS_L | 1u << TRIP_D_POPS_L | 1u << TRIP_D_LISTED_L |
                       1u << TRIP_D_LANES_L);)
   PHASE_MARK(4);   // SUM
   if (live) {
      rho_out[p] = density;
      const float2 bc = FAST ? neighbor_terms_fast(k, density, pi.w) : neighbor_terms(k, density, pi.w);
      const float4 v = velp[p];
      // what the acceleration pass gathers, and what it stages in its tile (FAST: the two factors
      // change places - only the viscous sum's last few neighbours are gathered: visc_keep)
      velB_out[p] = make_float4(v.x, v.y, v.z, FAST ? bc.y : bc.x);
      auxc_out[p] = FAST ? bc.x : bc.y;
      ncount[p] = count;
   }
   PHASE_MARK(5);   // results issued
   PHASE_COUNT(15);
}

// ---- density pass of the workgroups whose tile fits no capacity: one row segment at a time ----------
// A scene several times denser than the benchmark's (a dam that really breaks compresses to 5x)
// has workgroups whose nine row segments together exceed every LDS capacity the pass can run
// with.  Inline, the tiled kernel computes such a workgroup "untiled": every lane walks its
// candidates through L1/L2 (6-8x a tiled workgroup), writes no lists, and the acceleration pass
// has to search again the same way.  This kernel is launched BEFORE the tiled one when the last
// step reported many of them, and stages the candidates through the same LDS tile in pieces:
// segment after segment (ascending = canonical order), chunk after chunk of at most tile_cap
// sorted positions.  Per chunk: TEST (the same screen), and every screened candidate confirmed, summed
// and appended where it is popped (round 4; appends staged through LDS as in the tiled kernel).  List entries are the indices of the workgroup's
// VIRTUAL tile (the layout its descriptor describes, wherever it would have been): the
// acceleration pass - for which the tile is too large as well - walks them with operands from
// global memory (accel_from_lists).  Same neighbours, same order, same arithmetic: same bits.
// A workgroup whose virtual tile does not fit the entry format (> 4064 / 16352 indices) and a
// particle with more neighbours than its list holds keep the untiled walk.
// (launch bounds: 4 workgroups per CU - the tiles of these workgroups allow fewer anyway, and at the
// tiled kernel's 6 the uniform-mass instantiations spilled two registers to scratch)
template <bool UNIT_SCALE, bool UNIFORM_MASS, bool WIDE, bool FAST>
__global__ void __launch_bounds__(TILE_THREADS, 4)
k_full_density_chunked(const float4* __restrict__ posm, const float4* __restrict__ velp,
                       const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ meta,
                       CellGrid g, PairConsts k, float* __restrict__ rho_out,
                       float4* __restrict__ velB_out, float* __restrict__ auxc_out,
                       int32_t* __restrict__ ncount, const TileDesc* __restrict__ desc,
                       uint32_t* __restrict__ nlist, uint32_t* __restrict__ nlist_overflow,
                       int tile_cap, int32_t* __restrict__ tile_stats,
                       const uint32_t* __restrict__ giveup, int list_cap)
{
   __shared__ TileDesc sd;
   __shared__ int list_overflow;
   // Chunks are a little shorter than the launch's capacity: what that leaves free behind the three arrays
   // holds the append's staging slots (16 bytes per lane, as in the tiled kernel).  A capacity too small to
   // spare them (pinned by a test) appends with 2-byte stores.
   const int tid = threadIdx.x;
   const bool staged = tile_cap >= 1024;
   const int chunk_cap = staged ? tile_cap - 352 : tile_cap;      // (352 entries x 12 bytes >= 16 bytes x 256 lanes)
   TileLds L;
   L.x = tile_lds_dynamic;
   L.y = L.x + (chunk_cap + TILE_PAD);
   L.z = L.y + (chunk_cap + TILE_PAD);
   char* const stage_lane = reinterpret_cast<char*>(L.z + (chunk_cap + TILE_PAD)) + 16 * tid;
   const int begin = meta[META_SUM_BEGIN], end = meta[META_SUM_END];
   const int n_giveup = tile_stats[TSTAT_GIVEUP_DENSITY];
   for (int gi = blockIdx.x; gi < n_giveup; gi += gridDim.x) {
      const int wg = (int)giveup[gi];
      const int p0 = begin + wg * TILE_THREADS;
      const int p = p0 + tid;
      const bool live = p < end;
      __syncthreads();                      // the previous workgroup of this block is done with sd / the tile
      if (tid == 0) list_overflow = 0;
      tile_desc_load(desc, wg, sd);
      if (sd.total + TILE_PAD > (1 << ListEntry<WIDE>::TBITS)) {
         // its virtual tile cannot be indexed by a list entry: the untiled walk, no lists
         if (live) density_untiled<UNIT_SCALE, FAST>(p, posm, cell_start, velp, g, k, rho_out, velB_out, auxc_out, ncount);
         if (tid == 0) nlist_overflow[wg] = 1u;
         continue;
      }
      float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
      RowRanges r;
#pragma unroll
      for (int kk = 0; kk < 9; kk++) r.s[kk] = r.e[kk] = 0;
      if (live) {
         pi = posm[p];
         int cx, cy, cz;
         cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
         row_ranges(g, cell_start, cx, cy, cz, r);
      }
      // the workgroup's nine row segments as sorted ranges (as tile_desc works them out)
      int cxa, cya, cza, cxb, cyb, czb;
      const float4 pa = posm[p0], pb = posm[min(p0 + TILE_THREADS - 1, end - 1)];
      const int c_first = (int)cell_of(g, pa.x, pa.y, pa.z, cxa, cya, cza);
      const int c_last = (int)cell_of(g, pb.x, pb.y, pb.z, cxb, cyb, czb);
      const f32x2 px = {pi.x, pi.x}, py = {pi.y, pi.y}, pz = {pi.z, pi.z};
      uint32_t* list_block = nlist + (size_t)wg * (size_t)(list_rows(list_cap) * TILE_THREADS);
      char* const lists = reinterpret_cast<char*>(list_block);
      const uint32_t lane_off = 16u * (uint32_t)tid;
      uint32_t pos = list_pos_of(0u);   // the append position (list_pos_of(entries this lane's list holds))
      int count = 0;
      float density = 0.0f;
      if (staged) *reinterpret_cast<uint4*>(stage_lane) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
      for (int kk = 0; kk < 9; kk++) {
         const int off = ((kk / 3 - 1) * g.ny + (kk % 3 - 1)) * g.nx;
         int lo = c_first + off - 1, hi = c_last + off + 1;
         lo = lo < 0 ? 0 : lo;
         hi = hi > g.ncells - 1 ? g.ncells - 1 : hi;
         int G = 0, len = 0;
         if (hi >= lo) {
            G = (int)cell_start[lo];
            len = (int)cell_start[hi + 1] - G;
         }
         const int D = sd.D[kk];
         const uint32_t kbits = ListEntry<WIDE>::tag(kk);
         const int rs = kk == 0 ? r.s[0] : kk == 1 ? r.s[1] : kk == 2 ? r.s[2] : kk == 3 ? r.s[3] : kk == 4 ? r.s[4]
                      : kk == 5 ? r.s[5] : kk == 6 ? r.s[6] : kk == 7 ? r.s[7] : r.s[8];
         const int re = kk == 0 ? r.e[0] : kk == 1 ? r.e[1] : kk == 2 ? r.e[2] : kk == 3 ? r.e[3] : kk == 4 ? r.e[4]
                      : kk == 5 ? r.e[5] : kk == 6 ? r.e[6] : kk == 7 ? r.e[7] : r.e[8];
         for (int c0 = G; c0 < G + len; c0 += chunk_cap) {
            const int have = min(chunk_cap, G + len - c0);
            __syncthreads();                 // the previous chunk's readers are done
            // (TILE_BATCH loads in flight before the first LDS store, as tile_load does: one load
            // waited for per trip made a chunk of 4 400 entries seventeen round trips long)
            for (int base = 0; base < have; base += TILE_BATCH * TILE_THREADS) {
               float4 buf[TILE_BATCH];
#pragma unroll
               for (int rr = 0; rr < TILE_BATCH; rr++) buf[rr] = posm[c0 + min(base + tid + rr * TILE_THREADS, have - 1)];
#pragma unroll
               for (int rr = 0; rr < TILE_BATCH; rr++) {
                  const int i = base + tid + rr * TILE_THREADS;
                  if (i < have) {
                     L.x[i] = buf[rr].x;
                     L.y[i] = buf[rr].y;
                     L.z[i] = buf[rr].z;
                  }
               }
            }
            __syncthreads();
            // this lane's candidates inside the chunk, as chunk-local slots
            const int ts = max(rs - c0, 0), te = min(re - c0, have);
            for (int t0 = (ts < te) ? (ts & ~3) : te; __any(t0 < te); t0 += 32) {
               uint32_t mask = 0;
               if (t0 < te && count <= list_cap) {      // (a list that has overflowed stops here: the lane walks)
#pragma unroll
                  for (int q8 = 0; q8 < 4; q8++) {
                     const int t = t0 + 8 * q8;
                     // (the tiled kernel's screen: the pop below confirms on the reference's own value)
                     if (t < te) mask |= test8(L, t, px, py, pz, k.h2_screen) << (8 * q8);
                  }
                  const int lo_b = ts - t0, hi_b = te - t0;
                  if (lo_b > 0) mask &= ~0u << lo_b;
                  if (hi_b < 32) mask &= ~(~0u << hi_b);
                  if (kk == 4) {
                     const int sb = (p - c0) - t0;      // the particle itself
                     if (sb >= 0 && sb < 32) mask &= ~(1u << sb);
                  }
               }
               // Every screened candidate is confirmed where it is popped - the chunk is in LDS now, and
               // only now: the reference's own test on the reference's own value, the density term, and
               // the entry appended if (and only if) it counts.  (Until round 4 the entries were appended
               // first and a second pass over the chunk's part of the list summed and compacted them:
               // a 2-byte store and a 2-byte load per entry, and the lists read back from memory.  These
               // workgroups are the heaviest of a compressed scene, where the vector ALUs idle and the
               // memory path does not: the sum's arithmetic at the append's lane use is the cheaper side.)
               const uint32_t ebase = kbits | (uint32_t)(c0 + D + t0);
               while (__any(mask != 0u)) {
                  if (mask != 0u) {
                     const uint32_t bit = (uint32_t)__builtin_ctz(mask);
                     mask &= mask - 1u;
                     const int t = t0 + (int)bit;
                     float mj = pi.w;
                     if (!UNIFORM_MASS) mj = posm[c0 + t].w;
                     float dx, dy, dz;
                     const float d2 = dist2(pi.x, pi.y, pi.z, L.x[t], L.y[t], L.z[t], dx, dy, dz);
                     if (d2 < k.h2) {
                        float d = sqrt_rn(d2);
                        if (!UNIT_SCALE) d *= k.sim_scale;
                        density_accumulate<UNIT_SCALE>(k, mj, d, density);
                        if (count >= list_cap) {        // one more than the list holds: no list for this particle
                           count = list_cap + 1;
                           mask = 0u;
                        } else {
                           count++;
                           const uint32_t in_block = pos & 14u;
                           if (staged) {
                              *reinterpret_cast<uint16_t*>(stage_lane + in_block) = (uint16_t)(ebase + bit);
                              if (in_block == 14u)
                                 *reinterpret_cast<uint4*>(lists + list_pos_off(pos & ~14u, lane_off)) =
                                    *reinterpret_cast<const uint4*>(stage_lane);
                           } else {
                              *reinterpret_cast<uint16_t*>(lists + list_pos_off(pos, lane_off)) = (uint16_t)(ebase + bit);
                           }
                           pos = list_pos_next(pos);
                        }
                     }
                  }
               }
            }
         }
      }
      if (!staged)
         list_pad(lists, lane_off, count, list_cap);
      else if (count <= list_cap && ((count & 7) != 0 || count == 0))
         *reinterpret_cast<uint4*>(lists + list_pos_off(pos & ~14u, lane_off)) = *reinterpret_cast<const uint4*>(stage_lane);
      const bool overflowed = count > list_cap;
      if (overflowed) {
         list_overflow = 1;
         *reinterpret_cast<uint32_t*>(lists + lane_off) = NLIST_NO_LIST;
      }
      if (__any(overflowed)) {
         const int without = __popcll(__ballot(overflowed));
         if ((tid & (SPH_WAVE - 1)) == 0) atomicAdd(&tile_stats[TSTAT_NO_LIST], without);
      }
      __syncthreads();
      if (tid == 0) nlist_overflow[wg] = list_overflow ? 2u : 0u;
      if (live) {
         if (overflowed) {
            // more neighbours than the list holds: this lane alone walks its candidates through L1/L2
            density_untiled<UNIT_SCALE, FAST>(p, posm, cell_start, velp, g, k, rho_out, velB_out, auxc_out, ncount);
         } else {
            rho_out[p] = density;
            const float2 bc = FAST ? neighbor_terms_fast(k, density, pi.w) : neighbor_terms(k, density, pi.w);
            const float4 v = velp[p];
            velB_out[p] = make_float4(v.x, v.y, v.z, FAST ? bc.y : bc.x);
            auxc_out[p] = FAST ? bc.x : bc.y;
            ncount[p] = count;
         }
      }
   }
}

// ---- acceleration pass: list-driven, no TEST ------------------------------------------------------
// Same workgroups, same tile layout as the density pass (so its u16 list entries are valid
// tile indices).  The pass is bound by the per-neighbour gathers, so the tile also holds the
// neighbour's viscosity coefficient C_j next to x/y/z and the only global gather left per
// neighbour is one 16-byte {vx, vy, vz, B_j}.
// The tile (dynamic LDS) is an array of {x, y, z, C}: this pass never scans it, it only looks
// single neighbours up, so one ds_read_b128 per neighbour beats four scattered ds_read_b32.
struct AccelLds {
   TileDesc desc;
};

// The acceleration pass can be launched in two parts (early exchange): part 1 = the workgroups
// that hold a particle of the owned planes next to a neighbouring slab (sorted ranges
// [OWN_BEGIN, BND_LO_END) and [BND_HI_BEGIN, OWN_END)), part 2 = all others, part 0 = everything.
__device__ __forceinline__ bool accel_part_has(int part, int p0, const int32_t* __restrict__ meta)
{
   if (part == 0) return true;
   const int ob = meta[META_OWN_BEGIN], oe = meta[META_OWN_END];
   const int lo_end = min(meta[META_BND_LO_END], oe);
   const int hi_begin = min(max(meta[META_BND_HI_BEGIN], lo_end), oe);
   const int a = max(p0, ob), b = min(p0 + TILE_THREADS, oe);   // owned particles of the workgroup
   const bool border = (a < lo_end && b > ob) || (b > hi_begin && a < oe);
   return border == (part == 1);
}

// Which workgroup hardware workgroup `block` of a launch of part `part` computes, or -1.  The
// workgroups of a part are index ranges - border = the two ends of the owned range, interior = what
// lies between - and each part spreads ITS workgroups over the XCDs (xcd_workgroup over the part's
// count): a border launch that took its workgroups where the whole grid's mapping puts them ran
// on two of the eight XCDs, the ones holding the ends of the sorted range.
__device__ __forceinline__ int accel_part_workgroup(int part, int block, int wgs, int begin,
                                                    const int32_t* __restrict__ meta)
{
   if (part == 0) return block < wgs ? xcd_workgroup(block, wgs) : -1;
   const int ob = meta[META_OWN_BEGIN], oe = meta[META_OWN_END];
   const int lo_end = min(meta[META_BND_LO_END], oe);
   const int hi_begin = min(max(meta[META_BND_HI_BEGIN], lo_end), oe);
   // workgroup w holds sorted positions [begin + 256 w, + 256); see accel_part_has
   const int first = min(max((ob - begin) / TILE_THREADS, 0), wgs);                        // first with owned particles
   const int last = min(max((oe - begin + TILE_THREADS - 1) / TILE_THREADS, first), wgs);   // one past the last
   const int b1 = lo_end > ob ? min(max((lo_end - begin + TILE_THREADS - 1) / TILE_THREADS, first), last) : first;
   const int a2 = hi_begin < oe ? min(max((hi_begin - begin) / TILE_THREADS, b1), last) : last;
   if (part == 1) {
      const int n1 = b1 - first, count = n1 + (last - a2);
      if (block >= count) return -1;
      const int j = xcd_workgroup(block, count);
      return j < n1 ? first + j : a2 + (j - n1);
   }
   const int count = a2 - b1;
   return block < count ? b1 + xcd_workgroup(block, count) : -1;
}

// FusedStep: what k_integrate<., HASH = true> does for particle p (position x, acceleration a), by
// the workgroup that computed a.  Called by all threads of the workgroup; wg_index = the
// workgroup of 256 particles p belongs to (its slot in the energy partial sums).
template <bool UNIT_SCALE>
__device__ __forceinline__ void fused_integrate(const FusedStep& fs, const PairConsts& k,
                                                const CellGrid& g, int p, bool live, float4 x,
                                                float4 a, int wg_index)
{
   double ke = 0.0, pe = 0.0;
   uint32_t c = 0xffffffffu;
   float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
   if (live) {
      v = fs.velp_in[p];
      integrate_particle<UNIT_SCALE>(k, x, v, a, ke, pe);
      fs.posm_out[p] = x;
      fs.velp_out[p] = v;
      int cx, cy, cz;
      c = cell_of(g, x.x, x.y, x.z, cx, cy, cz);
      fs.key[p] = c;
   }
   count_cell_runs(c, live, p, fs.cell_count, fs.slot, (uint32_t)g.ncells);
   if (fs.slab) {
      // the halo messages, by the particle's NEW plane exactly as k_slab_pack_early classifies:
      // particles of the owned planes next to a neighbouring slab (sorted ranges
      // [OWN_BEGIN, BND_LO_END) and [BND_HI_BEGIN, OWN_END)) go into that neighbour's message when
      // they are within the halo of its territory or beyond; a particle further inside that ended
      // up there moved more than a cell plane in one step - nobody has it as a ghost: say so
      bool to_left = false, to_right = false;
      if (live) {
         const int ob = fs.meta[META_OWN_BEGIN], oe = fs.meta[META_OWN_END];
         const int lo_end = min(fs.meta[META_BND_LO_END], oe);
         const int hi_begin = min(max(fs.meta[META_BND_HI_BEGIN], lo_end), oe);
         const int plane = cell_coord(x.z, g.inv, g.nz_global);
         const bool near_left = fs.zone.have_left && plane < fs.zone.lo + fs.zone.halo;
         const bool near_right = fs.zone.have_right && plane >= fs.zone.hi - fs.zone.halo;
         if ((p >= ob && p < lo_end) || (p >= hi_begin && p < oe)) {
            to_left = near_left;
            to_right = near_right;
         } else if (near_left || near_right) {
            atomicOr(&fs.meta[META_ERRORS], 8);
         }
      }
      if (fs.zone.have_left) {
         const int s = msg_reserve(&fs.left->header[0], to_left);
         if (to_left) {
            if (s < fs.msg_capacity) {
               fs.left->rec[2 * s] = x;
               fs.left->rec[2 * s + 1] = v;
            } else {
               atomicOr(&fs.meta[META_ERRORS], 2);
            }
         }
      }
      if (fs.zone.have_right) {
         const int s = msg_reserve(&fs.right->header[0], to_right);
         if (to_right) {
            if (s < fs.msg_capacity) {
               fs.right->rec[2 * s] = x;
               fs.right->rec[2 * s + 1] = v;
            } else {
               atomicOr(&fs.meta[META_ERRORS], 2);
            }
         }
      }
   }
   // block reduction, fixed order (the same as k_integrate's: same partial sums)
   __shared__ double s_ke[TILE_THREADS / SPH_WAVE], s_pe[TILE_THREADS / SPH_WAVE];
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
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int q = 0; q < TILE_THREADS / SPH_WAVE; q++) {
         sa += s_ke[q];
         sb += s_pe[q];
      }
      fs.epart[2 * wg_index + 0] = sa;
      fs.epart[2 * wg_index + 1] = sb;
   }
   __syncthreads();
}

// A workgroup whose tile fitted the density pass (12 B per entry) but not this pass's capacity
// (16 B per entry): its neighbour lists exist, so it does not have to search again as the untiled
// route does (every candidate of 27 cells through L1/L2, ~8x a tiled workgroup) - one lane walks
// its list and gathers {x, y, z, m} and C of each neighbour from global memory instead of the LDS
// tile (~2.5x a tiled workgroup).  Same pairs, same order, same arithmetic: same bits.  d = the
// workgroup's tile descriptor (in LDS: the entries' segment -> sorted position shift).
template <bool UNIT_SCALE, bool UNIFORM_MASS, bool WIDE, bool FAST>
__device__ __forceinline__ float4
accel_from_lists(int p, int cnt, const char* __restrict__ lists, uint32_t lane_off, const TileDesc& d,
                 const float4* __restrict__ posm, const float4* __restrict__ velB,
                 const float* __restrict__ rho, const float* __restrict__ auxc, const PairConsts& k)
{
   const float4 pi = posm[p];
   AccelState s;
   accel_begin(k, s, pi, velB[p], rho[p]);
   const bool in_range = accel_operands_in_range(k);
   int first_v = 0;
   if (FAST) {
      const int keep = visc_keep(s.visc_scale);
      first_v = keep < cnt ? cnt - keep : 0;
   }
   if constexpr (FAST) {
      // Tolerance mode (the arithmetic the compressed dam is benchmarked in, where a tenth of the workgroups
      // come here): the pressure sum takes a list block of eight entries per trip - ONE 16-byte load, the
      // next block requested before this one's gathers - and gathers {x, y, z} and m B only; the viscous
      // sum's last few neighbours follow in a loop of their own, as in the tiled route (the two sums meet in
      // accel_end: same bits).  Four entries per trip, a 2-byte load each and {v, C} gathered for all of
      // them, made a particle with 170 neighbours wait for 85 dependent round trips.
      const int lastb = cnt > 0 ? (cnt - 1) >> 3 : 0;
      uint4 blk = list_block_load(lists, 0, lane_off);
      for (int j0 = 0; j0 < cnt; j0 += 8) {
         uint32_t entry[8];
         list_block_entries(blk, entry);
         blk = list_block_load(lists, min((j0 >> 3) + 1, lastb), lane_off);
         float4 pj[8];
         float cj[8];
#pragma unroll
         for (int u = 0; u < 8; u++) {
            const uint32_t e = j0 + u < cnt ? entry[u] : entry[0];      // (entry 0 of a block in use is a neighbour)
            const int q = ListEntry<WIDE>::tile(e) - ListEntry<WIDE>::shift(d, e);
            pj[u] = posm[q];
            cj[u] = auxc[q];
         }
#pragma unroll
         for (int u = 0; u < 8; u++) {
            if (j0 + u < cnt) {
               float dx, dy, dz;
               float dd = sqrt_rn(dist2(pi.x, pi.y, pi.z, pj[u].x, pj[u].y, pj[u].z, dx, dy, dz));
               if (!UNIT_SCALE) dd *= k.sim_scale;
               accel_pair_fast_pressure<UNIT_SCALE>(k, s, dx, dy, dz, dd, cj[u]);
            }
         }
      }
      for (int j0 = first_v; j0 < cnt; j0 += VISC_UNROLL) {
         float4 pj[VISC_UNROLL], vj[VISC_UNROLL];
#pragma unroll
         for (int u = 0; u < VISC_UNROLL; u++) {
            const uint32_t e = list_entry_load(lists, (uint32_t)min(j0 + u, cnt - 1), lane_off);
            const int q = ListEntry<WIDE>::tile(e) - ListEntry<WIDE>::shift(d, e);
            pj[u] = posm[q];
            vj[u] = velB[q];
         }
#pragma unroll
         for (int u = 0; u < VISC_UNROLL; u++) {
            if (j0 + u < cnt) {
               float dx, dy, dz;
               float dd = sqrt_rn(dist2(pi.x, pi.y, pi.z, pj[u].x, pj[u].y, pj[u].z, dx, dy, dz));
               if (!UNIT_SCALE) dd *= k.sim_scale;
               accel_pair_fast_viscous(k, s, dd, vj[u].x, vj[u].y, vj[u].z, vj[u].w);
            }
         }
      }
      accel_fast_finish(k, s);
      return accel_end<UNIT_SCALE>(k, s);
   }
   constexpr int U = 4;
   for (int j0 = 0; j0 < cnt; j0 += U) {
      float4 pj[U], vj[U];
      float cj[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
         const int j = min(j0 + u, cnt - 1);
         const uint32_t e = list_entry_load(lists, (uint32_t)j, lane_off);
         const int q = ListEntry<WIDE>::tile(e) - ListEntry<WIDE>::shift(d, e);
         pj[u] = posm[q];
         cj[u] = auxc[q];
         vj[u] = make_float4(0.f, 0.f, 0.f, 0.f);
         if (!FAST || j >= first_v) vj[u] = velB[q];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
         if (j0 + u < cnt) {
            float dx, dy, dz;
            const float d2 = dist2(pi.x, pi.y, pi.z, pj[u].x, pj[u].y, pj[u].z, dx, dy, dz);
            float dd = sqrt_rn(d2);
            if (!UNIT_SCALE) dd *= k.sim_scale;
            if (FAST) {
               accel_pair_fast_pressure<UNIT_SCALE>(k, s, dx, dy, dz, dd, cj[u]);
               if (j0 + u >= first_v) accel_pair_fast_viscous(k, s, dd, vj[u].x, vj[u].y, vj[u].z, vj[u].w);
            } else {
               accel_pair<UNIT_SCALE>(k, s, dx, dy, dz, dd, UNIFORM_MASS ? pi.w : pj[u].w, vj[u].x, vj[u].y,
                                      vj[u].z, vj[u].w, cj[u], in_range);
            }
         }
      }
   }
   if (FAST) accel_fast_finish(k, s);
   return accel_end<UNIT_SCALE>(k, s);
}

template <bool UNIT_SCALE, bool UNIFORM_MASS, bool WIDE, bool FAST>
__global__ void __launch_bounds__(TILE_THREADS, ACCEL_BLOCKS)
k_full_accel_lists(const float4* __restrict__ posm, const float4* __restrict__ velB,
                   const float* __restrict__ rho, const float* __restrict__ auxc,
                   const int32_t* __restrict__ ncount, const uint32_t* __restrict__ cell_start,
                   const int32_t* __restrict__ meta, CellGrid g, PairConsts k,
                   float4* __restrict__ acc, const TileDesc* __restrict__ desc,
                   const uint32_t* __restrict__ nlist, const uint32_t* __restrict__ nlist_overflow,
                   int tile_cap, const int32_t* __restrict__ tile_stats,
                   const uint32_t* __restrict__ giveup, int part, int list_cap,
                   int* __restrict__ tile_feedback, FusedStep fs, int tile_cap_density)
{
   PHASE_BEGIN();
   __shared__ AccelLds L;
   float4* xyzc = reinterpret_cast<float4*>(tile_lds_dynamic);
   constexpr int BATCH = TILE_BATCH;

   const int begin = meta[META_SUM_BEGIN], end = meta[META_SUM_END];
   const int ob = meta[META_OWN_BEGIN], oe = meta[META_OWN_END];
   const int tid = threadIdx.x;
   // (the workgroups that exist - a slab context launches one per 256 particles of its CAPACITY -
   // and, of those, the ones of this launch's part, spread over the XCDs)
   const int wgs = tile_stats[TSTAT_BLOCKS];
   const int mapped = accel_part_workgroup(part, blockIdx.x, wgs, begin, meta);
   const int wg = mapped < 0 ? 0 : mapped;
   const int p0 = begin + wg * TILE_THREADS;
   // (the density pass has finished: what it counted goes to the host's pinned copy)
   if (blockIdx.x == 0 && tid == 0) {
      tile_feedback[TSTAT_NO_LIST] = tile_stats[TSTAT_NO_LIST];
      if (fs.slab && part == 1) {   // the border part packs the messages: the size they are packed for
         if (fs.left) fs.left->header[1] = fs.msg_capacity;
         if (fs.right) fs.right->header[1] = fs.msg_capacity;
      }
   }
   // nothing of its own to do for workgroups past the range or made of ghosts only - nor, when
   // the pass is launched in two parts (early exchange), for those of the other part
   const bool own = mapped >= 0 && !(p0 >= end || p0 + TILE_THREADS <= ob || p0 >= oe) &&
                    accel_part_has(part, p0, meta);
   // 1: tile did not fit the density pass (on the give-up list), 2: some particle of the
   // workgroup has no list (more neighbours than list_cap)
   const uint32_t gave_up = own ? nlist_overflow[wg] : 1u;
   // (requested together with the flag it would otherwise wait for - one round trip less in front
   // of the tile loads; kept in a register while a give-up workgroup uses L.desc)
   int desc_word = 0;
   if (tid < (int)(sizeof(TileDesc) / sizeof(int))) desc_word = reinterpret_cast<const int*>(&desc[wg])[tid];
   // the first workgroups of the launch start with the workgroups whose tile does not fit
   // (give-up list), untiled, so that their long latency overlaps the rest of the launch
   if ((int)blockIdx.x < tile_stats[TSTAT_GIVEUP_ACCEL]) {
      const int g0 = begin + (int)giveup[blockIdx.x] * TILE_THREADS;
      const int gp = g0 + tid;
      const bool mine = gp < end && gp >= ob && gp < oe && accel_part_has(part, g0, meta);
      const int gwg = (int)giveup[blockIdx.x];
      const uint32_t glists = nlist_overflow[gwg];   // 1: its tile did not fit the density pass either (no lists)
      if (glists != 1u) {
         // the lists are there: walk them, operands from global memory (accel_from_lists)
         tile_desc_load(desc, gwg, L.desc);
         if (mine) {
            const int gcnt = ncount[gp];
            const char* glists_base = reinterpret_cast<const char*>(
               nlist + (size_t)gwg * (size_t)(list_rows(list_cap) * TILE_THREADS));
            const uint32_t goff = 16u * (uint32_t)tid;
            if (glists == 2u && gcnt > 0 && *reinterpret_cast<const uint32_t*>(glists_base + goff) == NLIST_NO_LIST)
               accel_untiled<UNIT_SCALE, FAST>(gp, posm, velB, rho, auxc, cell_start, g, k, acc, ncount);
            else
               acc[gp] = accel_from_lists<UNIT_SCALE, UNIFORM_MASS, WIDE, FAST>(gp, gcnt, glists_base, goff, L.desc, posm,
                                                                                velB, rho, auxc, k);
         }
         __syncthreads();   // L.desc is loaded again below, for this workgroup's own tile
      } else if (mine) {
         accel_untiled<UNIT_SCALE, FAST>(gp, posm, velB, rho, auxc, cell_start, g, k, acc, ncount);
      }
      // (a two-part launch walks the give-up list in BOTH parts: only the part that owns the listed
      // workgroup integrates it and writes its energy partial sums - the other part's zeros would race
      // with them on the unordered streams; workgroup-uniform, so the barriers inside stay legal)
      if (fs.on && accel_part_has(part, g0, meta)) {
         float4 gx = make_float4(0.f, 0.f, 0.f, 0.f), ga = gx;
         if (mine) {
            gx = posm[gp];
            ga = acc[gp];   // (written by this thread just above)
         }
         fused_integrate<UNIT_SCALE>(fs, k, g, gp, mine, gx, ga, (int)giveup[blockIdx.x]);
      }
   }
   if (gave_up == 1u) return;
   if (tid < (int)(sizeof(TileDesc) / sizeof(int))) reinterpret_cast<int*>(&L.desc)[tid] = desc_word;
   __syncthreads();
   PHASE_MARK(16);   // ranges, flags, descriptor
   const int total = L.desc.total;
   // does not fit this pass's wider entries, or did not fit the density pass's capacity (then it
   // may fit this one and have lists all the same - k_full_density_chunked writes them): either way
   // it is on the give-up list and one of the first workgroups computes it.  Computing it here as
   // well would integrate and count its particles twice.
   if (total > tile_cap || total > tile_cap_density) return;
   int B[9], D[9];
#pragma unroll
   for (int kk = 0; kk < 9; kk++) {
      B[kk] = L.desc.B[kk];
      D[kk] = L.desc.D[kk];
   }
   // Prologue order: the first batch of tile loads is issued, then - while it is in flight - the
   // lanes' own loads go out; only then are the tile's entries stored to LDS.  Unconditional loads, index clamped into the tile (see
   // tile_load).
   float4 buf[BATCH];
   float cbuf[BATCH];
#if defined(SPH_ABLATE) && SPH_ABLATE == 23
   // timing only (with the pair loop cut as in 21): no tile
#pragma unroll
   for (int r = 0; r < BATCH; r++) {
      buf[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      cbuf[r] = 0.f;
   }
   const int total_staged = 0;
#define total total_staged
#else
#pragma unroll
   for (int r = 0; r < BATCH; r++) {
      const int idx = min(tid + r * TILE_THREADS, total - 1);
      int d = D[0];
#pragma unroll
      for (int kk = 1; kk < 9; kk++) d = (idx >= B[kk]) ? D[kk] : d;
      buf[r] = posm[idx - d];
      cbuf[r] = auxc[idx - d];
   }
#endif

   // Lanes keep their own particle.  (Rounds 1-2 dealt the workgroup's particles to the lanes in
   // order of neighbour count - counting sort through LDS, three barriers - so that a wave's loop
   // length was uniform; with the pass as it is now, not doing it measures 10 us faster at 4M and
   // 110 us at 16M: consecutive particles share most of their neighbours, dealt lanes gather from
   // all over the tile, and the 1.5 KB of LDS are better spent on tile capacity.)
   const int col = tid;  // the particle (column of the list block) this lane works on
   const int p = p0 + col;
   const bool live = p < end && p >= ob && p < oe;
   float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
   float4 vi = pi;
   float rho_i = 0.0f;
   int cnt = 0;
   if (live) {
      pi = posm[p];
      vi = velB[p];
      rho_i = rho[p];
      cnt = ncount[p];
   }

   // the lane's first list words travel with the tile (requested before the count is known: what
   // a lane with fewer entries reads in the rows of its block is never used)
   const char* lists = reinterpret_cast<const char*>(nlist + (size_t)wg * (size_t)(list_rows(list_cap) * TILE_THREADS));
   const uint32_t lane_off = 16u * (uint32_t)col;
   uint32_t entry[ACCEL_UNROLL];
   uint4 next_blk = list_block_load(lists, 0, lane_off);      // (block 0 of every lane exists)

   // the tile: first batch from the registers, then whatever is left
#pragma unroll
   for (int r = 0; r < BATCH; r++) {
      const int idx = tid + r * TILE_THREADS;
      if (idx < total) xyzc[idx] = make_float4(buf[r].x, buf[r].y, buf[r].z, cbuf[r]);
   }
   for (int base = BATCH * TILE_THREADS; base < total; base += BATCH * TILE_THREADS) {
#pragma unroll
      for (int r = 0; r < BATCH; r++) {
         const int idx = min(base + tid + r * TILE_THREADS, total - 1);
         int d = D[0];
#pragma unroll
         for (int kk = 1; kk < 9; kk++) d = (idx >= B[kk]) ? D[kk] : d;
         buf[r] = posm[idx - d];
         cbuf[r] = auxc[idx - d];
      }
#pragma unroll
      for (int r = 0; r < BATCH; r++) {
         const int idx = base + tid + r * TILE_THREADS;
         if (idx < total) xyzc[idx] = make_float4(buf[r].x, buf[r].y, buf[r].z, cbuf[r]);
      }
   }
   __syncthreads();
#if defined(SPH_ABLATE) && SPH_ABLATE == 23
#undef total
#endif

   PHASE_MARK(17);   // tile, own loads, first list block
   AccelState s;
   accel_begin(k, s, pi, vi, rho_i);
   // every listed pair passed the exact d2 < h2 test: the division's range checks are uniform
   const bool in_range = accel_operands_in_range(k);
   // a particle without a list (marker in its first word; only in workgroups flagged 2): its lane
   // skips the list loop and walks its candidate ranges in the tile afterwards
   bool no_list = false;
   if (gave_up == 2u && cnt > 0) no_list = next_blk.x == NLIST_NO_LIST;
   // FAST: only the last visc_keep() neighbours take part in the viscous sum - and only they are
   // gathered ({v, C}; what every pair needs, m B, is in the tile)
   int first_v = 0;
   if (FAST) {
      const int keep = visc_keep(s.visc_scale);
      first_v = keep < cnt ? cnt - keep : 0;
   }
   if (no_list) cnt = 0;
#if defined(SPH_ABLATE) && (SPH_ABLATE == 21 || SPH_ABLATE == 23)
   cnt = 0;   // timing only: prologue and epilogue
#endif
   // (a lane past its last block takes that one again; the entries between a list's end and the end
   // of its last block are zeros - list_pad - because the exact loop gathers by every entry it holds)
   const int lastb = cnt > 0 ? (cnt - 1) >> 3 : 0;
   if constexpr (FAST) {
      const int self_tile = live ? p + L.desc.D[4] : 0;   // (the lane's own entry of the tile)
      // The pressure sum over the whole list: everything it needs of a neighbour is in the tile
      // ({x, y, z, m B}), no gather.  The list entries of the NEXT trip are requested before this
      // trip's arithmetic.  Lanes past their count re-read their last entry (result unused).
      // (A lane past its count relies on 0 * A = 0.  A = p_i * rhoiInv^2 is not finite when p_i is a
      // positive subnormal (1 / p_i = inf), and 0 * inf would poison a sum the list-walking routes leave
      // untouched: a wave holding such a lane - never seen outside a test - selects the factor instead.)
      // (... and the lane itself as a neighbour must yield a FINITE factor, so that its r = 0 makes the
      // term vanish whatever the factor is: then a lane past its count needs no select on m B - a
      // compare and two selects per pair were 12 of the pair's ~140 issue cycles)
      const float self_c = (k.hscaled * k.hscaled) * (s.pi_div_rhoi2 * xyzc[self_tile].w);
      const bool odd_lane = __any(!__builtin_isfinite(s.pi_div_rhoi2) || !__builtin_isfinite(self_c));
      TRIP(TripCounters trips; trips.wave(TRIP_A_WAVES - 16, true); trips.lane(TRIP_A_CNT_L - 16, (unsigned)cnt);
           trips.lane(TRIP_A_LANES_L - 16, live ? 1u : 0u); trips.lane(TRIP_A_NV_L - 16, (unsigned)(cnt - first_v));)
      for (int j0 = 0; __any(j0 < cnt); j0 += ACCEL_UNROLL) {
         TRIP(trips.wave(TRIP_A_PTRIPS_W - 16, true);)
         list_block_entries(next_blk, entry);
         next_blk = list_block_load(lists, min((j0 >> 3) + 1, lastb), lane_off);
         // (the trip's roots taken together - sqrt_rn_batch - and no branch on j0 + u < cnt: a lane
         // past its count takes ITSELF as the neighbour with m B = 0 - r = 0 and a factor of zero:
         // the sum does not move - so that the eight pairs of a trip are one basic block)
         float dx[ACCEL_UNROLL], dy[ACCEL_UNROLL], dz[ACCEL_UNROLL], dd[ACCEL_UNROLL], bm[ACCEL_UNROLL];
#pragma unroll
         for (int u = 0; u < ACCEL_UNROLL; u++) {
            const bool valid = j0 + u < cnt;
            const float4 pj = xyzc[valid ? ListEntry<WIDE>::tile(entry[u]) : self_tile];
            dd[u] = dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx[u], dy[u], dz[u]);
            bm[u] = pj.w;      // (a lane past its count: its own m B, with r = 0 exactly - see odd_lane)
         }
#if defined(SPH_ABLATE) && SPH_ABLATE == 22
#pragma unroll
         for (int u = 0; u < ACCEL_UNROLL; u++) s.pgx += dd[u] + bm[u];   // timing only: no pair arithmetic
#else
         sqrt_rn_batch(dd);
         if (!odd_lane) {
#pragma unroll
            for (int u = 0; u < ACCEL_UNROLL; u++) {
               if (!UNIT_SCALE) dd[u] *= k.sim_scale;
               accel_pair_fast_pressure<UNIT_SCALE>(k, s, dx[u], dy[u], dz[u], dd[u], bm[u]);
            }
         } else {
#pragma unroll
            for (int u = 0; u < ACCEL_UNROLL; u++) {
               if (!UNIT_SCALE) dd[u] *= k.sim_scale;
               if (j0 + u < cnt) accel_pair_fast_pressure<UNIT_SCALE>(k, s, dx[u], dy[u], dz[u], dd[u], bm[u]);
            }
         }
#endif
      }
      // The viscous sum over the list's last visc_keep() entries (ascending, as everywhere): the
      // only neighbours whose {v, C} is gathered, with the reference's stored distance.
      PHASE_MARK(18);   // pressure loop
      const int nv = cnt - first_v;
      TRIP(for (int m0 = 0; __any(m0 < nv); m0 += VISC_UNROLL) trips.wave(TRIP_A_VTRIPS_W - 16, true);
           trips.flush(16, 1u << (TRIP_A_CNT_L - 16) | 1u << (TRIP_A_NV_L - 16) | 1u << (TRIP_A_LANES_L - 16));)
      for (int m0 = 0; __any(m0 < nv); m0 += VISC_UNROLL) {
         float4 vj[VISC_UNROLL];
         int tj[VISC_UNROLL];
#pragma unroll
         for (int u = 0; u < VISC_UNROLL; u++) {
            const int j = min(first_v + m0 + u, cnt > 0 ? cnt - 1 : 0);
            const uint32_t e = list_entry_load(lists, (uint32_t)j, lane_off);
            tj[u] = ListEntry<WIDE>::tile(e);
            vj[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + u < nv) vj[u] = velB[tj[u] - ListEntry<WIDE>::shift(L.desc, e)];
         }
#pragma unroll
         for (int u = 0; u < VISC_UNROLL; u++) {
            if (m0 + u < nv) {
               const float4 pj = xyzc[tj[u]];
               float dx, dy, dz;
               float d = sqrt_rn(dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx, dy, dz));
               if (!UNIT_SCALE) d *= k.sim_scale;
               accel_pair_fast_viscous(k, s, d, vj[u].x, vj[u].y, vj[u].z, vj[u].w);
            }
         }
      }
   } else {
   // ACCEL_UNROLL neighbours per trip: their {v,B} gathers are issued back to back before the
   // first pair's arithmetic, and the list entries of the NEXT trip are requested before it too,
   // so neither of a neighbour's two dependent memory round trips is waited for in isolation.
   // Lanes past their count re-read their last entry (valid address, result unused).
   for (int j0 = 0; __any(j0 < cnt); j0 += ACCEL_UNROLL) {
      float4 vj[ACCEL_UNROLL];
      float mj[ACCEL_UNROLL];
      list_block_entries(next_blk, entry);
#pragma unroll
      for (int u = 0; u < ACCEL_UNROLL; u++) {
         mj[u] = pi.w;
         const int q = ListEntry<WIDE>::tile(entry[u]) - ListEntry<WIDE>::shift(L.desc, entry[u]);
         const int qq = cnt > 0 ? q : p0;  // lanes without neighbours hold no valid entry
#if defined(SPH_ABLATE) && SPH_ABLATE == 7
         vj[u] = make_float4(1.f, 2.f, 3.f, 4.f);  // timing only: no gather
#else
         vj[u] = velB[qq];
#endif
         if (!UNIFORM_MASS) mj[u] = posm[qq].w;
      }
      // the next trip's list entries travel while this trip's pairs are computed
      next_blk = list_block_load(lists, min((j0 >> 3) + 1, lastb), lane_off);
#pragma unroll
      for (int u = 0; u < ACCEL_UNROLL; u++) {
         if (j0 + u < cnt) {
            const float4 pj = xyzc[ListEntry<WIDE>::tile(entry[u])];
            float dx, dy, dz;
            const float d2 = dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx, dy, dz);
            float d = sqrt_rn(d2);
            if (!UNIT_SCALE) d *= k.sim_scale;
#if defined(SPH_ABLATE) && SPH_ABLATE == 22
            s.pgx += d + vj[u].x + pj.w;   // timing only: no pair arithmetic
#else
            accel_pair<UNIT_SCALE>(k, s, dx, dy, dz, d, mj[u], vj[u].x, vj[u].y, vj[u].z, vj[u].w,
                                   pj.w, in_range);
#endif
         }
      }
   }
   }
   if (gave_up == 2u && __any(no_list) && no_list) {
      // canonical order: the 9 rows ascending, positions ascending inside a row
      RowRanges rr;
      int cx, cy, cz;
      cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
      row_ranges(g, cell_start, cx, cy, cz, rr);
      const int self_t = p + L.desc.D[4];
      int jw = 0;   // neighbours visited so far (FAST: first_v was taken from the full count above)
#pragma unroll
      for (int kk = 0; kk < 9; kk++) {
         const int D = L.desc.D[kk];
         const int te = (int)rr.e[kk] + D;
         for (int t = (int)rr.s[kk] + D; t < te; t++) {
            const float4 pj = xyzc[t];
            float dx, dy, dz;
            const float d2 = dist2(pi.x, pi.y, pi.z, pj.x, pj.y, pj.z, dx, dy, dz);
            if (d2 < k.h2 && !(kk == 4 && t == self_t)) {
               float d = sqrt_rn(d2);
               if (!UNIT_SCALE) d *= k.sim_scale;
               if (FAST) {
                  accel_pair_fast_pressure<UNIT_SCALE>(k, s, dx, dy, dz, d, pj.w);
                  if (jw >= first_v) {
                     const float4 vj = velB[t - D];
                     accel_pair_fast_viscous(k, s, d, vj.x, vj.y, vj.z, vj.w);
                  }
                  jw++;
               } else {
                  const float4 vj = velB[t - D];
                  float mj = pi.w;
                  if (!UNIFORM_MASS) mj = posm[t - D].w;
                  accel_pair<UNIT_SCALE>(k, s, dx, dy, dz, d, mj, vj.x, vj.y, vj.z, vj.w, pj.w, in_range);
               }
            }
         }
      }
   }
   PHASE_MARK(19);   // viscous loop (exact arithmetic: the one pair loop)
   if (FAST) accel_fast_finish(k, s);
   const float4 a_i = accel_end<UNIT_SCALE>(k, s);
   if (live) acc[p] = a_i;
   PHASE_MARK(20);   // the sum's end, acceleration issued
   if (fs.on) fused_integrate<UNIT_SCALE>(fs, k, g, p, live, pi, a_i, wg);
   PHASE_MARK(21);   // integrate, hash, count, energy sums
   PHASE_COUNT(31);
}
