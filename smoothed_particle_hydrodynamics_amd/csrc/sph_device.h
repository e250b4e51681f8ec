// Shared device-side helpers and the context layout of the gfx950 SPH step.
// Written for MI355X only: wave = 64 lanes, no other targets, no compatibility paths.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/sph_hip.h"

#define SPH_WAVE 64
#define EV_RING 128
#define PACE_STEPS 16   // the host enqueues at most 2 * PACE_STEPS steps ahead of the device

// ---- error plumbing -------------------------------------------------------------------
#define SPH_TRY(expr)                                                                \
   do {                                                                              \
      hipError_t e_ = (expr);                                                        \
      if (e_ != hipSuccess) {                                                        \
         ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);               \
         return SPH_HIP_ERR_DEVICE;                                                  \
      }                                                                              \
   } while (0)

// Grid description handed to kernels by value.
struct CellGrid {
   int nx, ny, nz;  // cells held by this context (nz = local planes for a slab)
   float inv;       // cells per unit length
   int ncells;      // nx*ny*nz; cell id `ncells` is the trash cell (dead / out-of-slab entries)
   int z0;          // global z-plane of local plane 0
   int nz_global;   // planes of the whole grid (positions are clamped against this)
};

// Device-side counts and ranges of a context (int32 each); kernels read them so that a step
// never needs a host round trip.
enum {
   META_N_IN = 0,      // entries in the input arrays of the next cell build
   META_N_LIVE = 1,    // entries in real cells after the build (sorted order [0, n_live))
   META_OWN_BEGIN = 2, // sorted range of owned particles (planes [lo, hi))
   META_OWN_END = 3,
   META_SUM_BEGIN = 4, // sorted range whose density is needed (planes [lo-1, hi+1))
   META_SUM_END = 5,
   META_ERRORS = 6,    // bit 0: entry outside slab+halo, bit 1: message overflow, bit 2: capacity
   // early exchange (sph_hip_slab_step_begin): sorted ranges [OWN_BEGIN, BND_LO_END) and
   // [BND_HI_BEGIN, OWN_END) hold the owned planes next to a neighbouring slab, whose particles are
   // integrated and packed before the interior's acceleration runs
   META_BND_LO_END = 7,
   META_BND_HI_BEGIN = 8,
   META_COUNT = 12
};

// What a slab's cell build needs to know about the exchange (by value in kernarg).
struct SlabZone {
   int lo, hi;          // owned global planes [lo, hi)
   int halo;            // planes sent to / held from a neighbour
   int have_left, have_right;
   int drop_ghosts;     // FULL mode: entries of the previous sorted order outside its owned range
                        // are last step's ghosts - dropped, their owner re-sends them
   int early;           // the previous step packed its messages early: check nothing was missed
};

// Per-step statistics of the LDS tiles (k_tile_desc), fed back to the host's choice of tile
// capacity: how many workgroups would not fit each candidate capacity.
#define TILE_CANDS 12
// density pass: widening of h2 for the fused screening test (csrc/full_tiled.h, "TEST screens")
#define TEST_SCREEN_FACTOR 1.000002f
// particles (= threads) of one workgroup of the tiled FULL-mode passes
#ifndef TILE_THREADS
#define TILE_THREADS 256
#endif
enum {
   TSTAT_OVER = 0,            // [TILE_CANDS] workgroups whose tile exceeds candidate i
   TSTAT_BLOCKS = 12,         // workgroups counted
   TSTAT_MAX = 13,            // largest tile
   TSTAT_GIVEUP_DENSITY = 14, // entries of the give-up lists of the current step
   TSTAT_GIVEUP_ACCEL = 15,
   TSTAT_NO_LIST = 16,        // particles with more neighbours than their list holds (density pass)
   TSTAT_COUNT = 17
};
// The rest of the step, done by the acceleration pass itself (a context that holds the whole
// grid and exchanges with nobody): every particle is integrated where its acceleration was just
// computed - into the state buffers the cell build left free, the neighbours still read the old
// ones - and hashed and counted for the next cell build; no k_integrate launch, no second trip of
// positions, velocities and accelerations through HBM.
struct SlabMsg;
struct FusedStep {
   int on;
   const float4* velp_in;   // velocity + id of the state the sums read
   float4* posm_out;        // the other pair of state buffers
   float4* velp_out;
   double* epart;           // energy partial sums, one pair per workgroup (as k_integrate)
   uint32_t* key;           // next build: cell id, slot inside the cell, the cells' counts
   uint32_t* slot;
   uint32_t* cell_count;
   // A slab that exchanges early (sph_hip_slab_step_begin / _end): the two parts of the
   // acceleration launch do the rest of the step for the particles they own - integrate, hash for
   // the next build, and, in the border part, the halo messages (what k_slab_pack_early did from
   // a second read of the state).  The next build only has to hash what arrives from the
   // neighbours (k_hash_tail).
   int slab;
   SlabZone zone;
   SlabMsg* left;
   SlabMsg* right;
   int msg_capacity;
   int32_t* meta;           // error bits
};
struct TileCaps {
   int cand[TILE_CANDS];    // ascending candidate capacities (the occupancy levels of both kernels)
   int n_cand;
   int cap_density;         // capacities the current step's launches use
   int cap_accel;
   int wide;                // list entries carry a 14-bit tile index (a capacity above 4064)
};

#define SPH_DEAD_ID 0xffffffffu

// Constants of the per-pair arithmetic, by value in kernarg (scalar registers).
struct PairConsts {
   float h2, hscaled, hscaled2, sim_scale;
   float h2_screen;   // h2 widened for the density pass's fused-multiply-add screening test
   // FAST pressure sum: kernel2 * sim_scale times a power of two 2^-s chosen from ITS exponent so that
   // the product is a normal number of magnitude [2^-8, 2^-7) whatever h and sim_scale are (a fixed
   // 2^-64 went subnormal for h_scaled ~ 1e3 and up), and 2^s, which takes it out of the finished sum
   float fast_k2s, fast_unscale;
   float kernel1, kernel2, kernel3;
   float rho0, stiffness, viscosity;
   float grav_const, central_mass, cx, cy, cz, softening;
   float cfl_limit, cfl_limit2;
   float dt, sim_scale_inv;
   // dam-break extensions (0 = shipped behaviour)
   float gx, gy, gz;         // mGravity
   float damping;            // mDamping
   float max_x, max_y, max_z; // mMaxX/Y/Z
   int apply_gravity, apply_walls;
   // tolerance mode, no point mass (central_mass == 0, softening > 0): the point-mass terms of
   // computeAcceleration / integrate (src/sph.cpp:892-915, 966-989) are +-0 and are skipped
   int skip_point_mass;
};

struct sph_hip_context {
   sph_hip_params prm;
   int mode = 0;
   int device = 0;
   int capacity = 0;
   int n = 0;       // host upper bound of resident entries (owned + ghosts + dead)
   int n_owned = 0; // owned particles at the last upload / count query
   int32_t* meta = nullptr; // META_* (device)
   // slab (FULL mode): owned global z-planes [plane_lo, plane_hi), halo planes on each side
   int plane_lo = 0, plane_hi = 0, halo = 0;
   hipStream_t stream = nullptr;     // the stream every launch goes to
   hipStream_t own_stream = nullptr; // created with the context; `stream` may be redirected
   // per-phase event ring: EV_RING steps x 7 events; `ev_steps` counts timed steps since the
   // last reset (phase totals cover the last min(ev_steps, EV_RING) of them)
   hipEvent_t* ev = nullptr;
   long long ev_steps = 0;
   std::string err;

   CellGrid grid;

   // particle state: {x,y,z,m} and {vx,vy,vz,id-bits}.  FULL mode keeps it cell-sorted and
   // ping-pongs between the two buffers at every cell build; REF mode keeps it in index order
   // in buffer 0.
   float4* posm[2] = {nullptr, nullptr};
   float4* velp[2] = {nullptr, nullptr};
   int cur = 0;

   // cell build
   uint32_t* key = nullptr;        // cell id per particle
   uint32_t* slot = nullptr;       // arrival rank inside the cell (from the counting atomic)
   uint32_t* perm = nullptr;       // cell-sorted, arbitrary order inside a cell
   uint32_t* order = nullptr;      // REF: cell-sorted, ascending index inside a cell
   uint32_t* cell_count = nullptr; // ncells
   uint32_t* cell_start = nullptr; // ncells + 1
   uint32_t* scan_part = nullptr;  // per-tile partial sums of the scan
   uint32_t* big_cells = nullptr;  // [0] = count, then the cells with more than RANK_BIG members
   int scan_tiles = 0;

   // sums
   float* rho = nullptr;
   float4* velB = nullptr; // per particle {vx, vy, vz, B = p_j * rhojInv^2}: the acceleration gather
   float* auxc = nullptr;  // per particle C = (rhojInv * m_j) * k3 (FAST: m_j * B): staged in the acceleration tile
   float4* acc = nullptr; // {ax, ay, az, unused}
   int32_t* ncount = nullptr;
   struct TileDesc* tile_desc = nullptr; // per 256-particle workgroup: LDS tile layout
   uint32_t* nlist = nullptr;            // neighbour lists density pass -> acceleration pass
   uint32_t* nlist_overflow = nullptr;   // per workgroup: 1 = tile or a list did not fit
   int fast = 0;                   // tolerance-mode pair arithmetic (SPH_HIP_MODE_FULL_FAST / sph_hip_set_arithmetic)
   int uniform_mass = 0;           // every resident particle has bit-identical mass
   int use_tiled = 1;              // FULL mode: LDS-tiled kernels (0 = untiled everywhere)
   int prehashed = 0;              // the last integrate also did the next build's cell hash + counts
                                   // (2: a slab's fused step - owned entries only, see k_hash_tail)
   int no_prehash = 0, no_fused_integrate = 0, no_fused_slab = 0;   // SPH_HIP_NO_* switches, read at creation
   hipStream_t chunk_stream = nullptr;          // k_full_density_chunked runs beside the tiled density pass
   hipEvent_t ev_chunk_fork = nullptr, ev_chunk_join = nullptr;
   int chunked_giveups = -1;       // SPH_HIP_CHUNKED: 1 always / 0 never launch k_full_density_chunked (-1: by count)
   int slab_fused = 0;             // the step in progress (step_begin .. step_end) is fused
   void* slab_msgs[2] = {nullptr, nullptr};   // its message buffers
   int slab_msg_capacity = 0;
   int had_exchange = 0;           // pack/unpack/step_begin were used on this context: never prehash
   int may_hold_dead = 0;          // sph_hip_slab_pack has marked entries dead since the last cell build
   int early_exchange = 0;         // the last step packed its messages early (sph_hip_slab_step_begin)
   struct SlabComm* comm = nullptr;     // native RCCL exchange (csrc/slab_rccl.h), or null
   hipStream_t border_stream = nullptr; // stream the last step_begin put the border work on
   hipEvent_t ev_density = nullptr; // early exchange: density done (main stream) -> border work may start
   hipEvent_t ev_border = nullptr;  //                 border acceleration done (exchange stream) -> integrate may run
   hipEvent_t ev_pace[2] = {nullptr, nullptr};  // recorded every PACE_STEPS steps (see pace_host)
   long long steps_enqueued = 0;
   int timing_level = 2;           // SPH_HIP_TIMING_*: which events sph_hip_step() records
   int timing_stride = 1;          // ... on every timing_stride-th step only (sph_hip_set_timing_stride)
   long long timing_seen = 0;      // timed steps since the stride was set
   int slab_step_level = 0;        // level sph_hip_slab_step_begin chose for the step in progress
   // LDS tile capacity of the two tiled kernels: chosen per launch among the largest tiles that
   // still allow B workgroups per CU (levels, ascending), from the tile size recent steps needed
   // (tile_feedback: pinned host word the density kernel stores into; 0 = nothing known yet)
   int* tile_feedback = nullptr;   // TSTAT_COUNT ints, pinned host memory
   int32_t* tile_stats = nullptr;  // TSTAT_* of the current step (device)
   uint32_t* giveup_density = nullptr; // workgroups whose tile exceeds the density capacity
   uint32_t* giveup_accel = nullptr;   // ... or the acceleration capacity
   int tile_cap_forced = 0;        // SPH_HIP_TILE_CAP: fixed capacity for both kernels (tests)
   int list_cap = 0;               // neighbours per particle the lists hold (even)
   int list_cap_max = 0;           // how far the host may enlarge them (SPH_HIP_LIST_CAP pins both)
   size_t list_blocks = 0;         // workgroup blocks the list allocation covers
   int density_levels[TILE_CANDS] = {0}, n_density_levels = 0;
   int accel_levels[TILE_CANDS] = {0}, n_accel_levels = 0;
   int density_per_cu[TILE_CANDS] = {0}, accel_per_cu[TILE_CANDS] = {0};  // workgroups per CU at each level
   TileCaps caps = {};             // candidate capacities + the two chosen for the current step
   int cand_kept[TILE_CANDS] = {0}, n_cand_kept = 0;   // the candidate list tile_feedback's counts belong to

   // REF-mode lists
   int32_t* vox = nullptr; // 3 ints per particle
   uint32_t* nb = nullptr;
   float* nd = nullptr;

   // reductions
   double* epart = nullptr; // 2 * blocks partial sums, then [0],[1] totals
   int eblocks = 0;
   int energy_blocks = 0;   // partials written by the last integrate (0 = none yet)
   int32_t* stats = nullptr; // sum(lo,hi), max, min

   // error word of the slab exchange as last copied to the host (pinned; sph_hip_slab_poll_errors)
   volatile int32_t* err_watch = nullptr;
   hipEvent_t watch_event = nullptr;  // behind the last requested copy of the error word
   int watch_pending = 0;

   // asynchronous host mirror (sph_hip_download_async): its own device staging, copy stream and events
   float* mirror_stage = nullptr;     // capacity * 11 floats + voxel counts
   hipStream_t copy_stream = nullptr;
   hipEvent_t ev_exported = nullptr;  // compute stream: the mirror staging is complete
   hipEvent_t ev_copied = nullptr;    // copy stream: it has reached the host
   int mirror_busy = 0;               // a copy has been started and not yet been seen complete

   // staging for host <-> device in the reference's interleaved layouts
   float* stage = nullptr; // capacity * 11 floats
};

// ---- arithmetic helpers -----------------------------------------------------------------

// (int)floor(x) the way the x86-64 reference build does it (cvttsd2si): out-of-range and NaN
// give INT_MIN, which the clamp that follows turns into cell 0.
__device__ __forceinline__ int floor_to_int_x86(float f)
{
   float fl = floorf(f);
   return (fl >= -2147483648.0f && fl < 2147483648.0f) ? (int)fl : (int)0x80000000;
}

__device__ __forceinline__ int cell_coord(float x, float inv, int ncell)
{
   int c = floor_to_int_x86(x * inv);
   c = c < 0 ? 0 : c;
   c = c >= ncell ? ncell - 1 : c;
   return c;
}

// Cell of a position.  cz is the LOCAL plane (global plane - g.z0); a position whose plane lies
// outside the planes this context holds gets the trash cell id g.ncells.
__device__ __forceinline__ uint32_t cell_of(const CellGrid& g, float x, float y, float z, int& cx,
                                           int& cy, int& cz)
{
   cx = cell_coord(x, g.inv, g.nx);
   cy = cell_coord(y, g.inv, g.ny);
   cz = cell_coord(z, g.inv, g.nz_global) - g.z0;
   if (cz < 0 || cz >= g.nz) return (uint32_t)g.ncells;
   return (uint32_t)((cz * g.ny + cy) * g.nx + cx);
}

// Correctly rounded fp32 square root for the hot pair loops: the same value sqrtf() returns
// (IEEE round-to-nearest-even; checked against it for EVERY non-negative finite float by
// sph_hip_selftest_sqrt / tests/test_gpu_full_mode.py), in 10 instructions instead of the 15 of
// the compiler's expansion (v_sqrt_f32 + two +-1 ulp residual checks + denormal rescaling).
// Reciprocal-square-root seed, one coupled Goldschmidt step for g ~ sqrt(x) and h ~ 1/(2 sqrt(x)),
// then Markstein's correction g + (x - g*g) * h, whose fused residual makes the last rounding the
// only one that matters.  x = 0 gives 0 (the seed is taken of max(x, FLT_MIN), so g = 0 * finite).
// Below 2^-102 the residual x - g*g leaves the normal range and the sweep finds 1.8 million
// inputs that round the other way (none above): those, and denormals, take sqrtf() - a squared
// distance that small needs two fp32 positions 2^-51 apart, so the branch is there for
// correctness, not for speed.
// Non-finite inputs are outside the sweep on purpose: sqrt_rn(+inf) would be NaN (inf * rsq(inf) =
// inf * 0) where sqrtf gives inf, but every call site takes the root of a squared distance that
// has just passed "d2 < h2" - the membership test of src/sph.cpp:641,653, repeated bit for bit by
// the acceleration pass on the same positions - and neither +inf nor NaN passes that test.
#ifndef SPH_SQRT_GENERIC
__device__ __forceinline__ float sqrt_rn(float x)
{
   // bits in [1, 0x0c7fffff]  <=>  0 < x < 2^-102  (x is never negative here; -0 and 0 pass)
   if (__builtin_expect(__any(__float_as_uint(x) - 1u < 0x0c7fffffu), 0)) return sqrtf(x);
   // (seed of x + FLT_MIN, one plain addition, instead of max(x, FLT_MIN), which costs two wide-class
   // v_max_f32 - profiles/r4_valu_prices_more.txt: here x is 0 or at least 2^-102, so the sum is
   // FLT_MIN for 0 and x itself otherwise - except for odd mantissas in [2^-102, 2^-101), a tie that
   // moves the SEED's argument by one ulp; the result is checked for every float all the same:
   // sph_hip_selftest_sqrt)
   const float y = __builtin_amdgcn_rsqf(x + 1.17549435e-38f);
   float g = x * y;
   float h = 0.5f * y;
   const float r = __builtin_fmaf(-h, g, 0.5f);
   g = __builtin_fmaf(g, r, g);
   h = __builtin_fmaf(h, r, h);
   const float d = __builtin_fmaf(-g, g, x);
   return __builtin_fmaf(d, h, g);
}

// The same root for N values at once, with ONE uniform branch for the lot: a pair loop that takes
// the roots of the N neighbours of a trip one by one gets a scalar branch per neighbour from
// sqrt_rn's test for tiny arguments, which fences every neighbour's dependent chain (reciprocal
// square root, three fused steps, the pair arithmetic behind them) off from the next one's; taken
// together the N chains are one basic block and overlap.  Same values as sqrt_rn, bit for bit.
template <int N>
__device__ __forceinline__ void sqrt_rn_batch(float (&x)[N])
{
   bool tiny = false;
#pragma unroll
   for (int u = 0; u < N; u++) tiny |= __float_as_uint(x[u]) - 1u < 0x0c7fffffu;
   if (__builtin_expect(__any(tiny), 0)) {
#pragma unroll
      for (int u = 0; u < N; u++) x[u] = sqrtf(x[u]);
      return;
   }
#pragma unroll
   for (int u = 0; u < N; u++) {
      const float y = __builtin_amdgcn_rsqf(x[u] + 1.17549435e-38f);      // (as in sqrt_rn)
      float g = x[u] * y;
      float h = 0.5f * y;
      const float r = __builtin_fmaf(-h, g, 0.5f);
      g = __builtin_fmaf(g, r, g);
      h = __builtin_fmaf(h, r, h);
      const float d = __builtin_fmaf(-g, g, x[u]);
      x[u] = __builtin_fmaf(d, h, g);
   }
}
#else
__device__ __forceinline__ float sqrt_rn(float x) { return sqrtf(x); }
template <int N>
__device__ __forceinline__ void sqrt_rn_batch(float (&x)[N])
{
#pragma unroll
   for (int u = 0; u < N; u++) x[u] = sqrtf(x[u]);
}
#endif

// fp32 squared distance with the reference's association: (dx*dx + dy*dy) + dz*dz, no FMA
// (the translation unit is compiled with -ffp-contract=off).
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz,
                                       float& dx, float& dy, float& dz)
{
   dx = ax - bx;
   dy = ay - by;
   dz = az - bz;
   return dx * dx + dy * dy + dz * dz;
}
