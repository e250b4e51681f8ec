// C ABI of the gfx950 SPH step (include/sph_hip.h) — context management and phase launches.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py).

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <type_traits>

#include "cell_build.h"
#include "common_kernels.h"
#include "full_kernels.h"
#include "full_tiled.h"
#include "ref_kernels.h"
#include "slab_kernels.h"
#include "slab_rccl.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

std::string g_create_error;

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// run-time flags -> template arguments: f(std::bool_constant..., one per flag)
template <class F>
void bind_flags(F&& f) { f(); }
template <class F, class... Rest>
void bind_flags(F&& f, bool flag, Rest... rest)
{
   if (flag) bind_flags([&](auto... later) { f(std::true_type{}, later...); }, rest...);
   else bind_flags([&](auto... later) { f(std::false_type{}, later...); }, rest...);
}

PairConsts pair_consts(const sph_hip_params& p, bool fast)
{
   PairConsts k;
   k.h2 = p.h2;
   // SPH_HIP_TEST_SCREEN widens the screen (tests: many candidates then reach the exact
   // confirmation and the list rewrite; the results must not change)
   static const float screen = getenv("SPH_HIP_TEST_SCREEN") ? (float)atof(getenv("SPH_HIP_TEST_SCREEN"))
                                                             : TEST_SCREEN_FACTOR;
   k.h2_screen = p.h2 * (screen >= TEST_SCREEN_FACTOR ? screen : TEST_SCREEN_FACTOR);
   k.hscaled = p.hscaled;
   k.hscaled2 = p.hscaled2;
   k.sim_scale = p.sim_scale;
   k.kernel1 = p.kernel1;
   k.kernel2 = p.kernel2;
   k.kernel3 = p.kernel3;
   {
      // (pair_math.h: accel_pair_fast_pressure) |k2 s| * 2^-shift in [2^-8, 2^-7): times 1 / (d + 0.01)
      // <= 100 the per-pair factor stays below 1, so it cannot overflow unless the reference's own
      // term has; a zero or non-finite product keeps shift 0
      const float k2s = p.kernel2 * p.sim_scale;
      int e = 0, shift = 0;
      if (std::isfinite(k2s) && k2s != 0.0f) {
         (void)frexpf(k2s, &e);            // |k2s| = m * 2^e, m in [0.5, 1)
         shift = e + 7;
         shift = shift < -120 ? -120 : shift > 120 ? 120 : shift;
      }
      k.fast_k2s = ldexpf(k2s, -shift);
      k.fast_unscale = ldexpf(1.0f, shift);
   }
   k.rho0 = p.rho0;
   k.stiffness = p.stiffness;
   k.viscosity = p.viscosity;
   k.grav_const = p.grav_const;
   k.central_mass = p.central_mass;
   k.cx = p.central_pos[0];
   k.cy = p.central_pos[1];
   k.cz = p.central_pos[2];
   k.softening = p.softening;
   k.cfl_limit = p.cfl_limit;
   k.cfl_limit2 = p.cfl_limit2;
   k.dt = p.time_step;
   k.sim_scale_inv = p.sim_scale_inv;
   k.gx = p.gravity[0];
   k.gy = p.gravity[1];
   k.gz = p.gravity[2];
   k.damping = p.damping;
   k.max_x = p.max_x;
   k.max_y = p.max_y;
   k.max_z = p.max_z;
   k.apply_gravity = p.apply_gravity;
   k.apply_walls = p.apply_walls;
   k.skip_point_mass = fast && p.central_mass == 0.0f && p.softening > 0.0f && std::isfinite(p.grav_const) ? 1 : 0;
   return k;
}

// The kernels' UNIT_SCALE instantiations: mSimulationScale = 1 (no multiplication by it) and - what
// lets the FULL-mode density sums drop the reference's "d > hscaled" test for pairs that passed
// d2 < h2 (pair_math.h: density_accumulate<INSIDE>) - a smoothing length whose constants agree:
// sqrtf(h2) <= hscaled.  Parameters that do not (a caller may set any) take the general
// instantiations, which multiply by a scale of 1.0: the same bits.
bool unit_scale(const sph_hip_params& p)
{
   return p.sim_scale == 1.0f && p.sim_scale_inv == 1.0f && sqrtf(p.h2) <= p.hscaled;
}

// environment switch "NAME=1", read once per name
bool getenv_flag(const char* name)
{
   const char* v = getenv(name);
   return v && v[0] == '1';
}

template <typename T>
hipError_t dev_alloc(T** ptr, size_t count)
{
   return hipMalloc(reinterpret_cast<void**>(ptr), count * sizeof(T));
}

void free_all(sph_hip_context* ctx)
{
   for (int b = 0; b < 2; b++) {
      if (ctx->posm[b]) (void)hipFree(ctx->posm[b]);
      if (ctx->velp[b]) (void)hipFree(ctx->velp[b]);
   }
   void* ptrs[] = {ctx->key, ctx->slot, ctx->perm, ctx->order, ctx->cell_count, ctx->cell_start,
                   ctx->scan_part, ctx->big_cells, ctx->rho, ctx->velB, ctx->auxc, ctx->acc, ctx->ncount, ctx->vox, ctx->nb,
                   ctx->nd, ctx->epart, ctx->stats, ctx->stage, ctx->tile_desc, ctx->meta, ctx->nlist,
                   ctx->nlist_overflow, ctx->tile_stats, ctx->giveup_density, ctx->giveup_accel};
   for (void* q : ptrs)
      if (q) (void)hipFree(q);
   if (ctx->ev) {
      for (int k = 0; k < EV_RING * 7; k++)
         if (ctx->ev[k]) (void)hipEventDestroy(ctx->ev[k]);
      delete[] ctx->ev;
   }
   if (ctx->comm) {
      SlabComm* c = ctx->comm;
      if (c->stream) (void)hipStreamSynchronize(c->stream);
      if (c->comm) {
         const RcclApi* api = rccl_api(nullptr);
         if (api) (void)api->CommDestroy(c->comm);
      }
      for (void* q : {c->send_left, c->send_right, c->recv_left, c->recv_right, (void*)c->trim_word,
                      (void*)c->fill_word})
         if (q) (void)hipFree(q);
      if (c->fill_host) (void)hipHostFree(c->fill_host);
      if (c->fill_arrived) (void)hipEventDestroy(c->fill_arrived);
      if (c->packed) (void)hipEventDestroy(c->packed);
      if (c->arrived) (void)hipEventDestroy(c->arrived);
      if (c->stream) (void)hipStreamDestroy(c->stream);
      delete c;
      ctx->comm = nullptr;
   }
   for (int k = 0; k < 2; k++)
      if (ctx->ev_pace[k]) (void)hipEventDestroy(ctx->ev_pace[k]);
   if (ctx->chunk_stream) {
      (void)hipStreamSynchronize(ctx->chunk_stream);
      (void)hipStreamDestroy(ctx->chunk_stream);
      if (ctx->ev_chunk_fork) (void)hipEventDestroy(ctx->ev_chunk_fork);
      if (ctx->ev_chunk_join) (void)hipEventDestroy(ctx->ev_chunk_join);
   }
   if (ctx->ev_density) (void)hipEventDestroy(ctx->ev_density);
   if (ctx->ev_border) (void)hipEventDestroy(ctx->ev_border);
   if (ctx->tile_feedback) (void)hipHostFree(ctx->tile_feedback);
   if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
   if (ctx->mirror_stage) (void)hipFree(ctx->mirror_stage);
   if (ctx->ev_exported) (void)hipEventDestroy(ctx->ev_exported);
   if (ctx->ev_copied) (void)hipEventDestroy(ctx->ev_copied);
   if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
   if (ctx->err_watch) (void)hipHostFree((void*)ctx->err_watch);
   if (ctx->watch_event) (void)hipEventDestroy(ctx->watch_event);
   if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
}

int check_ctx(sph_hip_context* ctx)
{
   if (!ctx) return SPH_HIP_ERR_INVALID;
   hipError_t e = hipSetDevice(ctx->device);
   if (e != hipSuccess) {
      ctx->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
      return SPH_HIP_ERR_DEVICE;
   }
   return SPH_HIP_OK;
}

// ---- phase launches (no event recording, no host sync) ---------------------------------------

// ---- LDS tile capacity ---------------------------------------------------------------------
// For every workgroups-per-CU count B a tiled kernel can reach, the largest tile (multiple of
// 32 entries) that still lets B workgroups share a CU.  Registers and waves: the runtime's
// occupancy calculator.  LDS: the MI355X hands a workgroup its LDS (static + dynamic) in units of
// 1280 bytes out of 160 KiB per CU - measured with a sweep of pinned capacities (tools/cap_sweep.py:
// the density pass drops from 6 to 5 workgroups per CU between 2176 and 2208 entries and from 5 to
// 4 between 2624 and 2656, the acceleration pass from 4 to 3 between 2496 and 2528; the
// calculator's own rounding is finer, and a size it rated 3/CU ran at 2/CU, which rounds 1-2 used
// to cover with 2 KiB of slack per workgroup at the price of 130-190 entries per level).
#define LDS_PER_CU (160 * 1024)
#define LDS_GRANULE 1280
template <class Kernel>
int tile_levels(Kernel kernel, int bytes_per_entry, int* levels, int* per_cu)
{
   hipFuncAttributes attr;
   size_t static_lds = 2048;   // (no answer: the old slack)
   if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)) == hipSuccess)
      static_lds = attr.sharedSizeBytes;
   else
      (void)hipGetLastError();
   auto blocks_at = [&](int cap) {
      int nb = 0;
      const size_t bytes = (size_t)(cap + TILE_PAD) * bytes_per_entry;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, TILE_THREADS, bytes) != hipSuccess) {
         (void)hipGetLastError();
         return 0;
      }
      const size_t granules = (static_lds + bytes + LDS_GRANULE - 1) / LDS_GRANULE;
      const int by_lds = (int)(LDS_PER_CU / (granules * LDS_GRANULE));
      return nb < by_lds ? nb : by_lds;
   };
   // a workgroup may take the whole LDS of a CU (160 KiB); the 14-bit tile index of wide list
   // entries stops a little earlier for 12-byte entries
   int cap_max = TILE_CAP_MAX_WIDE;
   while (cap_max > 256 && (size_t)(cap_max + TILE_PAD) * bytes_per_entry > 156 * 1024) cap_max -= 32;
   const int cap_min = 1024 - TILE_PAD;
   int n = 0, prev = 0;
   for (int want = blocks_at(cap_min); want >= 1 && n < TILE_CANDS / 2; want--) {
      int lo = cap_min, hi = cap_max;            // largest cap with blocks_at(cap) >= want
      while (lo < hi) {
         const int mid = lo + ((hi - lo) / 32 + 1) / 2 * 32;
         if (blocks_at(mid) >= want) lo = mid;
         else hi = mid - 32;
      }
      if (lo > prev) {
         per_cu[n] = want;
         levels[n++] = prev = lo;
      }
      if (lo >= cap_max) break;
   }
   if (n == 0) {                                 // no answer from the runtime: a size that fits
      per_cu[n] = 3;
      levels[n++] = 3008;
   }
   if (getenv("SPH_HIP_DEBUG")) {
      fprintf(stderr, "sph_hip: tile capacity levels (%d B/entry):", bytes_per_entry);
      for (int l = 0; l < n; l++) fprintf(stderr, " %d (%d/CU)", levels[l], blocks_at(levels[l]));
      fprintf(stderr, "\n");
   }
   return n;
}

// the tiled kernels may ask for all of a CU's LDS as dynamic shared memory (set per context: the
// attribute belongs to the function on the current device)
void allow_large_tiles()
{
   const int most = 160 * 1024;
   for (int m = 0; m < 16; m++) {
      bind_flags([&](auto U, auto M, auto W, auto F) {
         (void)hipFuncSetAttribute((const void*)(k_full_density_tiled<U.value, M.value, W.value, F.value>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, most);
         (void)hipFuncSetAttribute((const void*)(k_full_density_chunked<U.value, M.value, W.value, F.value>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, most);
         if constexpr (M.value || !F.value)   // (FAST never gathers masses: only its M = true form exists)
            (void)hipFuncSetAttribute((const void*)(k_full_accel_lists<U.value, M.value, W.value, F.value>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, most);
      }, (m & 1) != 0, (m & 2) != 0, (m & 4) != 0, (m & 8) != 0);
   }
   (void)hipGetLastError();
}

// Level with the least expected cost for the workgroups of the latest reported step.  A larger
// tile means fewer workgroups per CU (relative throughput thr, measured on the 4M dam-break and
// its breaking variant), a smaller one sends the workgroups that do not fit down the untiled
// route (several times the work, and ~100 us from start to end however little else there is to
// do - launches too short to hide that must not have any).  Nothing reported yet: the level next
// to 3008 entries.
int pick_level(const sph_hip_context* ctx, const int* fb, const int* levels, const int* per_cu, int n,
               const float* thr, float untiled_cost, int over_other = -1, float listed_cost = 0.0f)
{
   const int blocks = fb[TSTAT_BLOCKS];
   if (blocks <= 0) {
      for (int l = 0; l < n; l++)
         if (levels[l] >= 3008) return levels[l];
      return levels[n - 1];
   }
   const bool hides_untiled = blocks >= 8192 * 256 / TILE_THREADS;
   int best = levels[n - 1];
   float best_cost = 1e30f;
   for (int l = 0; l < n; l++) {
      int over = blocks;
      for (int c = 0; c < ctx->caps.n_cand; c++)
         if (ctx->caps.cand[c] == levels[l]) over = fb[TSTAT_OVER + c];
      if (over > 0 && !hides_untiled && l + 1 < n) continue;
      const float f = (float)over / (float)blocks;
      const int b = per_cu[l] < 1 ? 1 : (per_cu[l] > 6 ? 6 : per_cu[l]);
      // (acceleration pass: of the workgroups that do not fit, those that fitted the density pass
      // have their lists and take the cheaper list-driven route without a tile)
      float f_search = f;
      if (over_other >= 0) f_search = (float)(over_other < over ? over_other : over) / (float)blocks;
      const float cost = (1.0f - f) / thr[b] + untiled_cost * f_search + listed_cost * (f - f_search);
      if (cost < best_cost) {
         best_cost = cost;
         best = levels[l];
      }
   }
   return best;
}

// Capacities for the step about to be launched (before its k_tile_desc, which lists the
// workgroups that will not fit them).
// The density pass of an earlier step reported particles with more neighbours than their lists
// hold (those lanes walk their candidates one by one in both passes, an order of magnitude
// slower per particle): enlarge the lists for the steps from here on.  The device has
// to be idle for the exchange of the allocation - once or twice in a run that compresses.
// Never changes results, only which route a particle takes.
void grow_lists(sph_hip_context* ctx)
{
   // the largest capacity the device has room for, at once (a second reallocation later would be
   // a second stall); allocated while the device still works through the steps already enqueued
   uint32_t* bigger = nullptr;
   int want = ctx->list_cap_max;
   while (want > ctx->list_cap) {
      const size_t words = ctx->list_blocks * list_rows(want) * TILE_THREADS;
      if (dev_alloc(&bigger, words) == hipSuccess) break;
      (void)hipGetLastError();
      bigger = nullptr;
      want = (want / 2 - 1) & ~1;          // 1022 -> 510 -> 254
   }
   if (!bigger || want <= ctx->list_cap) {
      if (bigger) (void)hipFree(bigger);
      ctx->list_cap_max = ctx->list_cap;   // no memory for it: stay, and do not ask again
      return;
   }
   if (hipStreamSynchronize(ctx->stream) != hipSuccess) {   // (an error is reported by the step itself)
      (void)hipFree(bigger);
      return;
   }
   (void)hipFree(ctx->nlist);
   ctx->nlist = bigger;
   ctx->list_cap = want;
   ctx->list_cap_max = want < ctx->list_cap_max ? want : ctx->list_cap_max;
   ((volatile int*)ctx->tile_feedback)[TSTAT_NO_LIST] = 0;
   static const bool debug = getenv("SPH_HIP_DEBUG") != nullptr;
   if (debug) fprintf(stderr, "sph_hip: neighbour lists enlarged to %d entries\n", want);
}

// what a workgroup of the acceleration pass costs on the list-driven route without a tile (accel_from_lists),
// in units of a tiled one
// ... and what a workgroup whose tile fits no capacity costs the density pass (k_full_density_chunked).
// (Round 4: 3 since that kernel confirms at the pop and stages its appends; the 600-step transient of the
// breaking 4M dam - tools/dam_windows.py - takes 1712 ms with 6, 1690 with 3, 1697 with 2, 1799 with 1.5.)
#ifndef DENSITY_GIVEUP_COST
#define DENSITY_GIVEUP_COST 3.0f
#endif
#ifndef ACCEL_LISTED_COST
#define ACCEL_LISTED_COST 2.5f
#endif
void pick_tile_caps(sph_hip_context* ctx)
{
   TileCaps& caps = ctx->caps;
   if (ctx->list_cap < ctx->list_cap_max) {
      const int without = ((volatile int*)ctx->tile_feedback)[TSTAT_NO_LIST];
      const int blocks = ((volatile int*)ctx->tile_feedback)[TSTAT_BLOCKS];
      if (without > 64 && without > blocks * (TILE_THREADS / 256)) grow_lists(ctx);   // > 0.4 %
   }
   if (caps.n_cand == 0) {
      allow_large_tiles();
      if (ctx->fast)
         ctx->n_density_levels = tile_levels(k_full_density_tiled<true, true, false, true>, DENSITY_TILE_BYTES,
                                             ctx->density_levels, ctx->density_per_cu);
      else
         ctx->n_density_levels = tile_levels(k_full_density_tiled<true, true, false, false>, DENSITY_TILE_BYTES,
                                             ctx->density_levels, ctx->density_per_cu);
      if (ctx->fast)
         ctx->n_accel_levels = tile_levels(k_full_accel_lists<true, true, false, true>, ACCEL_TILE_BYTES,
                                           ctx->accel_levels, ctx->accel_per_cu);
      else
         ctx->n_accel_levels = tile_levels(k_full_accel_lists<true, true, false, false>, ACCEL_TILE_BYTES,
                                           ctx->accel_levels, ctx->accel_per_cu);
      // candidates = ascending union of both kernels' levels
      int nd = 0, na = 0;
      while ((nd < ctx->n_density_levels || na < ctx->n_accel_levels) && caps.n_cand < TILE_CANDS) {
         const int d = nd < ctx->n_density_levels ? ctx->density_levels[nd] : INT32_MAX;
         const int a = na < ctx->n_accel_levels ? ctx->accel_levels[na] : INT32_MAX;
         const int v = d < a ? d : a;
         if (d == v) nd++;
         if (a == v) na++;
         caps.cand[caps.n_cand++] = v;
      }
      // The arithmetic was switched (sph_hip_set_arithmetic): other kernels, possibly other levels.
      // The statistics the host holds were counted against the old candidate list: they stay valid
      // when the list is the same, and mean nothing otherwise.
      bool same = ctx->n_cand_kept == caps.n_cand;
      for (int c = 0; same && c < caps.n_cand; c++) same = ctx->cand_kept[c] == caps.cand[c];
      if (!same && ctx->n_cand_kept > 0 && ctx->tile_feedback) memset(ctx->tile_feedback, 0, TSTAT_COUNT * sizeof(int));
      ctx->n_cand_kept = caps.n_cand;
      for (int c = 0; c < caps.n_cand; c++) ctx->cand_kept[c] = caps.cand[c];
   }
   if (ctx->tile_cap_forced > 0) {
      caps.cap_density = caps.cap_accel = ctx->tile_cap_forced;
      // (tests: a smaller capacity for the acceleration pass alone sends the workgroups in between
      // down its list-driven route without a tile)
      if (const char* v = getenv("SPH_HIP_TILE_CAP_ACCEL")) {
         const int c = atoi(v) / 32 * 32;
         if (c >= 256 && c < caps.cap_accel) caps.cap_accel = c;
      }
      // (... and a smaller one for the density pass alone: workgroups that fit the acceleration
      // pass's capacity but are on the give-up lists all the same)
      if (const char* v = getenv("SPH_HIP_TILE_CAP_DENSITY")) {
         const int c = atoi(v) / 32 * 32;
         if (c >= 256 && c < caps.cap_density) caps.cap_density = c;
      }
      caps.wide = ctx->tile_cap_forced > TILE_CAP_MAX;
      return;
   }
   int fb[TSTAT_COUNT];
   for (int i = 0; i < TSTAT_COUNT; i++) fb[i] = ((volatile int*)ctx->tile_feedback)[i];
   // relative throughput by workgroups per CU (index 1..6), and what an untiled workgroup costs
   // in units of a tiled one
   // (On the 4M column at rest, with one pass pinned to each level - tools/occupancy_prices.py,
   // round 3 - the passes lose more than this below 5 per CU: density 1 / 0.97 / 0.87 / 0.74 / 0.54 at
   // 6 .. 2, acceleration 1 / 0.935 / 0.82 / 0.60 at 5 .. 2.  With those figures the breaking dam,
   // whose large tiles also hold more work per workgroup, ran 2-10 % slower in five of its sixteen
   // windows and faster in none: the tables stay as the breaking dam tuned them.)
   static const float density_thr[7] = {0.0f, 0.33f, 0.62f, 0.85f, 0.93f, 0.97f, 1.0f};
   static const float accel_thr[7] = {0.0f, 0.40f, 0.68f, 0.87f, 0.98f, 1.0f, 1.0f};
   caps.cap_density = pick_level(ctx, fb, ctx->density_levels, ctx->density_per_cu,
                                 ctx->n_density_levels, density_thr, DENSITY_GIVEUP_COST);
   int over_density = fb[TSTAT_BLOCKS];
   for (int c = 0; c < caps.n_cand; c++)
      if (caps.cand[c] == caps.cap_density) over_density = fb[TSTAT_OVER + c];
   caps.cap_accel = pick_level(ctx, fb, ctx->accel_levels, ctx->accel_per_cu, ctx->n_accel_levels,
                               accel_thr, 8.0f, over_density, ACCEL_LISTED_COST);
   // both passes of a step read and write the same lists: one entry format for the two
   caps.wide = caps.cap_density > TILE_CAP_MAX || caps.cap_accel > TILE_CAP_MAX;
   static int debug_left = getenv("SPH_HIP_DEBUG") ? 6 : 0;
   if (debug_left > 0 && debug_left--)
      fprintf(stderr, "sph_hip: %d workgroups, largest tile %d -> capacities %d / %d\n",
              fb[TSTAT_BLOCKS], fb[TSTAT_MAX], caps.cap_density, caps.cap_accel);
}

// what the cell build has to know about the slab's neighbours
SlabZone slab_zone(const sph_hip_context* ctx)
{
   SlabZone z;
   z.lo = ctx->plane_lo;
   z.hi = ctx->plane_hi;
   z.halo = ctx->halo;
   z.have_left = ctx->plane_lo > 0;
   z.have_right = ctx->plane_hi < ctx->grid.nz_global;
   z.drop_ghosts = ctx->mode == SPH_HIP_MODE_FULL;
   z.early = ctx->early_exchange;
   return z;
}

// clear_left/right: message buffers whose record counters this build zeroes (early exchange)
int launch_cell_build(sph_hip_context* ctx, void* clear_left = nullptr, void* clear_right = nullptr,
                      bool keep_sums = false)
{
   const int n = ctx->n;  // host upper bound of entries; the exact count is meta[META_N_IN]
   if (n == 0) return SPH_HIP_OK;
   const int blocks = div_up(n, 256);
   const CellGrid g = ctx->grid;
   hipStream_t st = ctx->stream;
   const int cur = ctx->cur;
   const SlabZone zone = slab_zone(ctx);
   if (ctx->prehashed == 2) {
      // a slab whose last step was integrated and hashed by its acceleration pass: only last
      // step's ghosts (to the trash cell) and the records received since are left
      ctx->prehashed = 0;
      hipLaunchKernelGGL(k_hash_tail, dim3(SLAB_PACK_BLOCKS), dim3(256), 0, st, ctx->posm[cur], ctx->meta, g,
                         ctx->key, ctx->slot, ctx->cell_count);
   } else if (ctx->prehashed) {
      ctx->prehashed = 0;   // the last integrate hashed and counted this very state already
   } else if (ctx->mode == SPH_HIP_MODE_REF)
      hipLaunchKernelGGL((k_hash_count<true, false>), dim3(blocks), dim3(256), 0, st, ctx->posm[cur],
                         ctx->velp[cur], ctx->meta, g, zone, ctx->key, ctx->slot, ctx->cell_count,
                         ctx->vox);
   else if (ctx->may_hold_dead)
      hipLaunchKernelGGL((k_hash_count<false, true>), dim3(blocks), dim3(256), 0, st, ctx->posm[cur],
                         ctx->velp[cur], ctx->meta, g, zone, ctx->key, ctx->slot, ctx->cell_count,
                         (int32_t*)nullptr);
   else
      hipLaunchKernelGGL((k_hash_count<false, false>), dim3(blocks), dim3(256), 0, st, ctx->posm[cur],
                         ctx->velp[cur], ctx->meta, g, zone, ctx->key, ctx->slot, ctx->cell_count,
                         (int32_t*)nullptr);
   ctx->early_exchange = 0;  // consumed: it described the step before this build
   ctx->may_hold_dead = 0;   // the build drops dead entries
   // the scan covers the real cells plus the trash cell, so cell_start[ncells] = live entries
   const int tiles = ctx->scan_tiles;
   const int ncells_scan = g.ncells + 1;
   hipLaunchKernelGGL(k_scan_reduce, dim3(tiles), dim3(SCAN_THREADS), 0, st, ctx->cell_count,
                      ncells_scan, ctx->scan_part);
   hipLaunchKernelGGL(k_scan_final, dim3(tiles), dim3(SCAN_THREADS), 0, st, ctx->cell_count,
                      ncells_scan, ctx->scan_part, ctx->cell_start, ctx->big_cells, (uint32_t)ctx->capacity,
                      ctx->meta);
   // sorted ranges: owned planes [lo, hi), density planes one wider (clipped to what is held)
   const int own_lo = ctx->plane_lo - g.z0, own_hi = ctx->plane_hi - g.z0;
   const int sum_lo = own_lo - 1 < 0 ? 0 : own_lo - 1;
   const int sum_hi = own_hi + 1 > g.nz ? g.nz : own_hi + 1;
   // owned planes next to a neighbouring slab, one wider than the halo (early exchange): a
   // particle further inside cannot reach the planes that are sent within one step
   const int border = ctx->halo + 1;
   const int bnd_lo = !zone.have_left ? own_lo : (own_lo + border < own_hi ? own_lo + border : own_hi);
   const int bnd_hi = !zone.have_right ? own_hi : (own_hi - border > own_lo ? own_hi - border : own_lo);
   hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(256), 0, st, ctx->key, ctx->slot,
                      ctx->cell_start, ctx->meta, ctx->perm, g.nx * g.ny, g.ncells, own_lo, own_hi,
                      sum_lo, sum_hi, bnd_lo, bnd_hi, ctx->tile_stats, (int32_t*)clear_left,
                      (int32_t*)clear_right, ctx->big_cells, (uint32_t)ctx->capacity);
   // crowded cells (listed by k_scatter; none in an ordinary scene: the workgroups then leave at
   // once) are ranked by sorting, behind the per-member scan that skips them; scratch = the
   // staging buffer, idle during a step
   uint32_t* scratch_key = reinterpret_cast<uint32_t*>(ctx->stage);
   uint32_t* scratch_src = scratch_key + ctx->capacity;
   // keep_sums (stand-alone voxelize of a FULL-mode context that holds the whole grid): note where
   // every entry goes (in `slot`, free once k_scatter has run) and move rho / acc / ncount along
   uint32_t* remap = (keep_sums && ctx->mode == SPH_HIP_MODE_FULL && !ctx->had_exchange) ? ctx->slot : nullptr;
   if (ctx->mode == SPH_HIP_MODE_REF) {
      hipLaunchKernelGGL(k_rank_order, dim3(blocks), dim3(256), 0, st, ctx->perm, ctx->key,
                         ctx->cell_start, ctx->meta, ctx->order);
      hipLaunchKernelGGL(k_rank_big<false>, dim3(RANK_BIG_BLOCKS), dim3(256), 0, st, ctx->big_cells,
                         ctx->perm, ctx->cell_start, (const float4*)nullptr, (const float4*)nullptr,
                         (float4*)nullptr, (float4*)nullptr, ctx->order, scratch_key, scratch_src,
                         (uint32_t*)nullptr);
   } else {
      const int nxt = cur ^ 1;
      if (ctx->use_tiled) {
         // + the LDS tile layout of every 256-particle workgroup of the density range, with the
         // statistics and give-up lists for the capacities chosen here for this step's sums
         static_assert(sizeof(TileDesc) == 20 * sizeof(int), "TileDesc is 20 ints");
         pick_tile_caps(ctx);
         const int ntiles = div_up(n, TILE_THREADS), desc_blocks = div_up(ntiles, 256);
         hipLaunchKernelGGL(k_rank_gather_tile_desc, dim3(desc_blocks + blocks), dim3(256), 0, st,
                            desc_blocks, ntiles, ctx->perm, ctx->key, ctx->cell_start, ctx->meta,
                            g, ctx->posm[cur], ctx->velp[cur], ctx->posm[nxt], ctx->velp[nxt],
                            ctx->tile_desc, ctx->caps, ctx->tile_stats, ctx->giveup_density,
                            ctx->giveup_accel, remap);
      } else {
         hipLaunchKernelGGL(k_rank_gather, dim3(blocks), dim3(256), 0, st, ctx->perm, ctx->key,
                            ctx->cell_start, ctx->meta, g.ncells, ctx->posm[cur], ctx->velp[cur],
                            ctx->posm[nxt], ctx->velp[nxt], remap);
      }
      hipLaunchKernelGGL(k_rank_big<true>, dim3(RANK_BIG_BLOCKS), dim3(256), 0, st, ctx->big_cells,
                         ctx->perm, ctx->cell_start, ctx->posm[cur], ctx->velp[cur], ctx->posm[nxt],
                         ctx->velp[nxt], (uint32_t*)nullptr, scratch_key, scratch_src, remap);
      if (remap) {
         // a build that is not followed by the sums: their last results move with the particles
         // (temporaries in the staging buffer behind k_rank_big's scratch; the float4 part 16-byte aligned)
         float4* acc_t = reinterpret_cast<float4*>(ctx->stage + ((2 * (size_t)ctx->capacity + 3) & ~(size_t)3));
         float* rho_t = reinterpret_cast<float*>(acc_t + (size_t)ctx->capacity);
         int32_t* cnt_t = reinterpret_cast<int32_t*>(rho_t + (size_t)ctx->capacity);
         hipLaunchKernelGGL(k_permute_sums, dim3(blocks), dim3(256), 0, st, remap, ctx->key, ctx->meta,
                            (uint32_t)g.ncells, ctx->rho, ctx->acc, ctx->ncount, rho_t, acc_t, cnt_t);
         SPH_TRY(hipMemcpyAsync(ctx->rho, rho_t, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
         SPH_TRY(hipMemcpyAsync(ctx->acc, acc_t, sizeof(float4) * n, hipMemcpyDeviceToDevice, st));
         SPH_TRY(hipMemcpyAsync(ctx->ncount, cnt_t, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, st));
      }
      ctx->cur = nxt;
      // The live set is now compacted at the front of the new buffers.  meta[N_IN] still holds
      // this build's input count: without an exchange nothing was dropped (n_live == n_in), and
      // with one, sph_hip_slab_unpack resets it to n_live before appending.
   }
   SPH_TRY(hipGetLastError());
   return SPH_HIP_OK;
}

int launch_find_neighbors(sph_hip_context* ctx)
{
   if (ctx->mode != SPH_HIP_MODE_REF || ctx->n == 0) return SPH_HIP_OK;
   const sph_hip_params& p = ctx->prm;
   hipLaunchKernelGGL(k_ref_find_neighbors, dim3(div_up(ctx->n, 256)), dim3(256), 0, ctx->stream,
                      ctx->posm[0], ctx->vox, ctx->cell_start, ctx->order, ctx->n, p.cells_x,
                      p.cells_y, p.cells_z, p.h, p.htimes2, p.h2, p.sim_scale, p.examine_count,
                      ctx->nb, ctx->nd, ctx->ncount);
   SPH_TRY(hipGetLastError());
   return SPH_HIP_OK;
}

// message buffers of a slab whose acceleration pass does the rest of the step (FusedStep.slab)
struct SlabFused {
   void* left;
   void* right;
   int capacity;
};

// tiled kernels of the two sums, specialised on (unit simulation scale, uniform mass)
void launch_density_tiled(sph_hip_context* ctx, bool unit, int blocks, const PairConsts& k)
{
   const int cap = ctx->caps.cap_density;
   const size_t lds = (size_t)(cap + TILE_PAD) * DENSITY_TILE_BYTES;
   // (a slab: the fused acceleration pass writes energy partials only for workgroups that own
   // particles; this launch zeroes the others' - same grid, one pair per workgroup)
   const bool whole = ctx->plane_lo == 0 && ctx->plane_hi == ctx->grid.nz_global;
   double* epart_clear = whole ? nullptr : ctx->epart + 2;
   // Many workgroups whose tile fits no capacity (a scene several times denser than the
   // benchmark's): a launch of its own stages their candidates through LDS piece by piece and
   // writes their lists (k_full_density_chunked) instead of the tiled kernel's first workgroups
   // walking them untiled.  Decided from what the last step reported; both kernels read the same
   // device-side list, the flag only says who works it off.
   const int reported = ((volatile int*)ctx->tile_feedback)[TSTAT_GIVEUP_DENSITY];
   const bool chunked = ctx->chunked_giveups == 1 || (ctx->chunked_giveups < 0 && reported >= 32);
   // The two kernels work on disjoint workgroups: the chunked one runs beside the tiled one on a
   // stream of its own (forked here, joined before anything else is enqueued) - alone it would
   // leave the device to ~1 000 long workgroups while the other 15 000 wait.
   hipStream_t side = ctx->stream;
   if (chunked) {
      if (!ctx->chunk_stream) {
         if (hipStreamCreateWithFlags(&ctx->chunk_stream, hipStreamNonBlocking) != hipSuccess ||
             hipEventCreateWithFlags(&ctx->ev_chunk_fork, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&ctx->ev_chunk_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ctx->chunk_stream = nullptr;
         }
      }
      if (ctx->chunk_stream && hipEventRecord(ctx->ev_chunk_fork, ctx->stream) == hipSuccess &&
          hipStreamWaitEvent(ctx->chunk_stream, ctx->ev_chunk_fork, 0) == hipSuccess)
         side = ctx->chunk_stream;
   }
   bind_flags([&](auto U, auto M, auto W, auto F) {
      if (chunked)
         hipLaunchKernelGGL((k_full_density_chunked<U.value, M.value, W.value, F.value>), dim3(1024),
                            dim3(TILE_THREADS), lds, side, ctx->posm[ctx->cur], ctx->velp[ctx->cur],
                            ctx->cell_start, ctx->meta, ctx->grid, k, ctx->rho, ctx->velB, ctx->auxc,
                            ctx->ncount, ctx->tile_desc, ctx->nlist, ctx->nlist_overflow, cap,
                            ctx->tile_stats, ctx->giveup_density, ctx->list_cap);
      hipLaunchKernelGGL((k_full_density_tiled<U.value, M.value, W.value, F.value>), dim3(blocks),
                         dim3(TILE_THREADS), lds, ctx->stream, ctx->posm[ctx->cur], ctx->velp[ctx->cur],
                         ctx->cell_start, ctx->meta, ctx->grid, k, ctx->rho, ctx->velB, ctx->auxc,
                         ctx->ncount, ctx->tile_desc, ctx->nlist, ctx->nlist_overflow, cap,
                         ctx->tile_stats, ctx->giveup_density, ctx->tile_feedback, ctx->list_cap,
                         epart_clear, chunked ? 0 : 1);
   }, unit, ctx->uniform_mass != 0, ctx->caps.wide != 0, ctx->fast != 0);
   if (side != ctx->stream) {
      // (a failure here would leave the streams unordered: drain the side stream the hard way)
      if (hipEventRecord(ctx->ev_chunk_join, side) != hipSuccess ||
          hipStreamWaitEvent(ctx->stream, ctx->ev_chunk_join, 0) != hipSuccess) {
         (void)hipGetLastError();
         (void)hipStreamSynchronize(side);
      }
   }
}

void launch_accel_lists(sph_hip_context* ctx, bool unit, int blocks, const PairConsts& k, int part,
                        hipStream_t st, bool fused = false, const SlabFused* slab = nullptr)
{
   FusedStep fs;
   memset(&fs, 0, sizeof(fs));
   if (fused) {
      fs.on = 1;
      fs.velp_in = ctx->velp[ctx->cur];
      fs.posm_out = ctx->posm[ctx->cur ^ 1];
      fs.velp_out = ctx->velp[ctx->cur ^ 1];
      fs.epart = ctx->epart + 2;
      fs.key = ctx->key;
      fs.slot = ctx->slot;
      fs.cell_count = ctx->cell_count;
      if (slab) {
         fs.slab = 1;
         fs.zone = slab_zone(ctx);
         fs.left = (SlabMsg*)slab->left;
         fs.right = (SlabMsg*)slab->right;
         fs.msg_capacity = slab->capacity;
         fs.meta = ctx->meta;
      }
   }
   const int cap = ctx->caps.cap_accel;
   const size_t lds = (size_t)(cap + TILE_PAD) * ACCEL_TILE_BYTES;
   // (a FAST context's density pass folds the mass into B: its acceleration pass never gathers masses)
   bind_flags([&](auto U, auto M, auto W, auto F) {
      if constexpr (M.value || !F.value)
      hipLaunchKernelGGL((k_full_accel_lists<U.value, M.value, W.value, F.value>), dim3(blocks),
                         dim3(TILE_THREADS), lds, st, ctx->posm[ctx->cur], ctx->velB, ctx->rho, ctx->auxc,
                         ctx->ncount, ctx->cell_start, ctx->meta, ctx->grid, k, ctx->acc, ctx->tile_desc,
                         ctx->nlist, ctx->nlist_overflow, cap, ctx->tile_stats, ctx->giveup_accel, part,
                         ctx->list_cap, ctx->tile_feedback, fs, ctx->caps.cap_density);
   }, unit, ctx->uniform_mass != 0 || ctx->fast != 0, ctx->caps.wide != 0, ctx->fast != 0);
}

int launch_density(sph_hip_context* ctx)
{
   const int n = ctx->n;
   if (n == 0) return SPH_HIP_OK;
   const PairConsts k = pair_consts(ctx->prm, ctx->fast != 0);
   const int blocks = div_up(n, 256);
   if (ctx->mode == SPH_HIP_MODE_REF) {
      hipLaunchKernelGGL(k_ref_density, dim3(blocks), dim3(256), 0, ctx->stream, ctx->posm[0],
                         ctx->nb, ctx->nd, ctx->ncount, n, ctx->prm.examine_count, k, ctx->rho);
   } else {
      const bool unit = unit_scale(ctx->prm);
      if (ctx->use_tiled) {
         launch_density_tiled(ctx, unit, div_up(n, TILE_THREADS), k);  // give-up workgroups fall back inline
      } else {                                         // SPH_HIP_UNTILED=1: untiled everywhere
         bind_flags([&](auto U, auto F) {
            hipLaunchKernelGGL((k_full_density<U.value, F.value>), dim3(blocks), dim3(256), 0, ctx->stream,
                               ctx->posm[ctx->cur], ctx->cell_start, ctx->velp[ctx->cur], ctx->meta,
                               ctx->grid, k, ctx->rho, ctx->velB, ctx->auxc, ctx->ncount);
         }, unit, ctx->fast != 0);
      }
   }
   SPH_TRY(hipGetLastError());
   return SPH_HIP_OK;
}

// part: 0 = all workgroups; 1 / 2 = those with / without particles of the owned planes next to
// a neighbouring slab (early exchange; tiled FULL mode only)
int launch_accel(sph_hip_context* ctx, int part = 0, hipStream_t part_stream = nullptr, bool fused = false,
                 const SlabFused* slab = nullptr)
{
   const int n = ctx->n;
   if (n == 0) return SPH_HIP_OK;
   const PairConsts k = pair_consts(ctx->prm, ctx->fast != 0);
   const int blocks = div_up(n, 256);
   if (ctx->mode == SPH_HIP_MODE_REF) {
      hipLaunchKernelGGL(k_ref_accel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->posm[0],
                         ctx->velp[0], ctx->rho, ctx->nb, ctx->nd, ctx->ncount, n,
                         ctx->prm.examine_count, k, ctx->acc);
   } else {
      const bool unit = unit_scale(ctx->prm);
      if (ctx->use_tiled) {
         // same tiling (and tile descriptors) as the density pass of this step
         launch_accel_lists(ctx, unit, div_up(n, TILE_THREADS), k, part, part ? part_stream : ctx->stream, fused, slab);
      } else {
         bind_flags([&](auto U, auto F) {
            hipLaunchKernelGGL((k_full_accel<U.value, F.value>), dim3(blocks), dim3(256), 0, ctx->stream,
                               ctx->posm[ctx->cur], ctx->velB, ctx->rho, ctx->auxc, ctx->cell_start,
                               ctx->meta, ctx->grid, k, ctx->acc, ctx->ncount);
         }, unit, ctx->fast != 0);
      }
   }
   SPH_TRY(hipGetLastError());
   return SPH_HIP_OK;
}

// with_hash: the kernel also does the first step of the next cell build (see k_integrate)
int launch_integrate(sph_hip_context* ctx, bool with_hash = false)
{
   const int n = ctx->n;
   if (n == 0) return SPH_HIP_OK;
   const PairConsts k = pair_consts(ctx->prm, ctx->fast != 0);
   const int blocks = div_up(n, RED_THREADS);
#define SPH_GO(U, H)                                                                             \
   hipLaunchKernelGGL((k_integrate<U, H>), dim3(blocks), dim3(RED_THREADS), 0, ctx->stream,       \
                      ctx->posm[ctx->cur], ctx->velp[ctx->cur], ctx->acc, ctx->meta, k,           \
                      ctx->epart + 2, ctx->grid, ctx->key, ctx->slot, ctx->cell_count)
   const bool unit = unit_scale(ctx->prm);
   if (unit && with_hash) SPH_GO(true, true);
   else if (unit) SPH_GO(true, false);
   else if (with_hash) SPH_GO(false, true);
   else SPH_GO(false, false);
#undef SPH_GO
   ctx->energy_blocks = blocks;  // totals are formed on demand (sph_hip_get_energy)
   ctx->prehashed = with_hash ? 1 : 0;
   SPH_TRY(hipGetLastError());
   return SPH_HIP_OK;
}

// The state is about to change behind the back of a prehash (upload, exchange, stand-alone
// integrate): forget it, and clear the counts it left in the histogram.
int drop_prehash(sph_hip_context* ctx)
{
   if (!ctx->prehashed) return SPH_HIP_OK;
   ctx->prehashed = 0;
   SPH_TRY(hipMemsetAsync(ctx->cell_count, 0, ((size_t)ctx->scan_tiles * SCAN_TILE + 16) * sizeof(uint32_t),
                          ctx->stream));
   return SPH_HIP_OK;
}

// Phase boundary k of a timed step is marked by event phase_event(ctx, k) of the step's ring
// slot.  An event record is a barrier packet (several microseconds on the stream), so a boundary
// with no launch before it shares the previous boundary's event: FULL mode has no separate
// neighbour search, and computePressure is a no-op in the reference (src/sph.cpp:253-263).
inline int phase_event(const sph_hip_context* ctx, int k)
{
   if (k == 4) return 3;
   if (k == 2 && ctx->mode == SPH_HIP_MODE_FULL) return 1;
   return k;
}

// Keeps the host from running arbitrarily far ahead of the device.  Launch parameters that follow
// the scene - the LDS tile capacities, chosen from statistics the device writes into pinned memory
// (tile_feedback) - are fixed when a step is ENQUEUED: a host that enqueues hundreds of steps at
// once (sph_hip_run(500)) would pick them all from the state before the first one, and a scene that
// compresses meanwhile ends up with nearly every workgroup on the untiled route.  Every PACE_STEPS
// steps an event is recorded and the event of PACE_STEPS steps ago waited for: the device always has
// at least PACE_STEPS steps queued (no bubble), the statistics are at most 2 * PACE_STEPS steps old.
int pace_host(sph_hip_context* ctx)
{
   const long long k = ctx->steps_enqueued++;
   if (k % PACE_STEPS != 0) return SPH_HIP_OK;
   const int slot = (int)((k / PACE_STEPS) & 1);
   if (k >= 2 * PACE_STEPS) SPH_TRY(hipEventSynchronize(ctx->ev_pace[slot]));   // recorded 2 * PACE_STEPS steps ago
   SPH_TRY(hipEventRecord(ctx->ev_pace[slot], ctx->stream));
   return SPH_HIP_OK;
}

// Which events the step about to be enqueued records: the context's level on every
// timing_stride-th timed step, nothing on the others (an event record is a barrier packet of
// ~10 us on the stream: sampling keeps the measurement from weighing on what it measures).
int next_step_level(sph_hip_context* ctx, bool timed)
{
   if (!timed || ctx->timing_level == SPH_HIP_TIMING_OFF) return SPH_HIP_TIMING_OFF;
   const bool sample = (ctx->timing_seen++ % ctx->timing_stride) == 0;
   return sample ? ctx->timing_level : SPH_HIP_TIMING_OFF;
}

int step_impl(sph_hip_context* ctx, bool timed)
{
   int rc;
   hipStream_t st = ctx->stream;
   hipEvent_t* ev = ctx->ev + 7 * (ctx->ev_steps % EV_RING);
   if ((rc = pace_host(ctx))) return rc;
   const int level = next_step_level(ctx, timed);
   const bool phases = level == SPH_HIP_TIMING_PHASES, sums = level == SPH_HIP_TIMING_SUMS;
   if (phases) SPH_TRY(hipEventRecord(ev[0], st));
   if ((rc = launch_cell_build(ctx))) return rc;
   if (phases || sums) SPH_TRY(hipEventRecord(ev[1], st));
   if ((rc = launch_find_neighbors(ctx))) return rc;
   if (phases && phase_event(ctx, 2) == 2) SPH_TRY(hipEventRecord(ev[2], st));
   if ((rc = launch_density(ctx))) return rc;
   if (phases) SPH_TRY(hipEventRecord(ev[3], st));
   // a context that holds the whole grid and has never exchanged anything: the integrate also
   // hashes and counts for the next cell build - and the tiled acceleration pass does both itself
   const bool hash_too = ctx->mode == SPH_HIP_MODE_FULL && !ctx->had_exchange && ctx->plane_lo == 0 &&
                         ctx->plane_hi == ctx->grid.nz_global && !ctx->no_prehash;
   const bool fused = hash_too && ctx->use_tiled && ctx->n > 0 && !ctx->no_fused_integrate;
   if ((rc = launch_accel(ctx, 0, nullptr, fused))) return rc;
   if (phases || sums) SPH_TRY(hipEventRecord(ev[5], st));
   if (fused) {
      ctx->cur ^= 1;                                    // the new state is in the other buffers
      ctx->energy_blocks = div_up(ctx->n, TILE_THREADS);
      ctx->prehashed = 1;
   } else if ((rc = launch_integrate(ctx, hash_too))) return rc;
   if (phases) SPH_TRY(hipEventRecord(ev[6], st));
   if (level != SPH_HIP_TIMING_OFF) ctx->ev_steps++;
   return SPH_HIP_OK;
}

// the six phase times of the step in ring slot `ev` (see sph_hip_set_timing)
int read_phases(sph_hip_context* ctx, hipEvent_t* ev, float ms[6])
{
   for (int k = 0; k < 6; k++) ms[k] = 0.0f;
   if (ctx->timing_level == SPH_HIP_TIMING_SUMS) {
      SPH_TRY(hipEventSynchronize(ev[5]));
      SPH_TRY(hipEventElapsedTime(&ms[2], ev[1], ev[5]));
      return SPH_HIP_OK;
   }
   SPH_TRY(hipEventSynchronize(ev[6]));
   for (int k = 0; k < 6; k++)
      SPH_TRY(hipEventElapsedTime(&ms[k], ev[phase_event(ctx, k)], ev[phase_event(ctx, k + 1)]));
   return SPH_HIP_OK;
}

// ---- error word of the slab exchange, watched without draining the stream ------------------------
// Request a copy of meta[META_ERRORS] into the pinned watch word.  Before that, wait for the
// PREVIOUS request (made one polling interval ago): normally long done; when the host has run far
// ahead of the device it holds the host back to at most two intervals of queued steps, which is
// what makes "reported within two intervals" true.
int watch_enqueue(sph_hip_context* ctx)
{
   if (ctx->watch_pending) SPH_TRY(hipEventSynchronize(ctx->watch_event));
   SPH_TRY(hipMemcpyAsync((void*)ctx->err_watch, ctx->meta + META_ERRORS, sizeof(int32_t),
                          hipMemcpyDeviceToHost, ctx->stream));
   SPH_TRY(hipEventRecord(ctx->watch_event, ctx->stream));
   ctx->watch_pending = 1;
   return SPH_HIP_OK;
}

// what the last arrived copy said
int watch_check(sph_hip_context* ctx, const char* who)
{
   const int32_t bits = ctx->err_watch[0];
   if (bits == 0) return SPH_HIP_OK;
   char text[256];
   snprintf(text, sizeof(text),
            "%s: the slab exchange lost particles, error bits %d (1 entry outside slab and halo, "
            "2 message overflow, 4 context capacity, 8 missed by the early exchange, 16 a particle id "
            "held twice)", who, (int)bits);
   ctx->err = text;
   return SPH_HIP_ERR_EXCHANGE;
}

__global__ void k_selftest_sqrt(unsigned long long* __restrict__ out)
{
   // every non-negative finite float: bit patterns 0 .. 0x7f7fffff
   unsigned long long bad = 0;
   uint32_t first = 0xffffffffu, last = 0u;
   const uint32_t stride = gridDim.x * blockDim.x;
   for (uint64_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= 0x7f7fffffull; b += stride) {
      const float x = __uint_as_float((uint32_t)b);
      const float want = sqrtf(x), got = sqrt_rn(x);
      if (__float_as_uint(want) != __float_as_uint(got)) {
         bad++;
         first = min(first, (uint32_t)b);
         last = max(last, (uint32_t)b);
      }
   }
   if (bad) {
      atomicAdd(&out[0], bad);
      atomicMin(&out[1], (unsigned long long)first);
      atomicMax(&out[2], (unsigned long long)last);
   }
}

} // namespace

extern "C" {

// (a diagnostic build - SPH_ABLATE hooks compiled in, garbage by design - says so here: the
// Python binding refuses it unless SPH_HIP_ALLOW_DIAGNOSTIC=1)
#ifdef SPH_DIAGNOSTIC_BUILD
int sph_hip_abi_version(void) { return SPH_HIP_ABI_VERSION | SPH_HIP_ABI_DIAGNOSTIC; }
#else
int sph_hip_abi_version(void) { return SPH_HIP_ABI_VERSION; }
#endif

#ifdef SPH_TRIPCOUNT
// diagnostic builds with -DSPH_TRIPCOUNT only (tools/trip_counts.py): the loop trip counters of the
// pair kernels (csrc/full_tiled.h: TRIP_*), accumulated since the last reset
extern "C" int sph_hip_diag_trips(unsigned long long* out, int n, int reset)
{
   unsigned long long h[TRIP_COUNT];
   if (hipDeviceSynchronize() != hipSuccess ||
       hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trip), sizeof(h)) != hipSuccess) return SPH_HIP_ERR_DEVICE;
   for (int i = 0; i < n && i < TRIP_COUNT; i++) out[i] = h[i];
   if (reset) {
      memset(h, 0, sizeof(h));
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_trip), h, sizeof(h)) != hipSuccess) return SPH_HIP_ERR_DEVICE;
   }
   return SPH_HIP_OK;
}
#endif

#ifdef SPH_PHASECLOCK
// diagnostic builds with -DSPH_PHASECLOCK only (tools/phase_clock.py): csrc/full_tiled.h, g_phase
extern "C" int sph_hip_diag_phases(unsigned long long* out, int n, int reset)
{
   // out[0..7] / out[16..23]: sums over the workgroups that wrote (density / acceleration), [7] / [23] their number
   static unsigned int h[2][PHASE_WGS][8];
   if (hipDeviceSynchronize() != hipSuccess ||
       hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) != hipSuccess) return SPH_HIP_ERR_DEVICE;
   for (int i = 0; i < n && i < 32; i++) out[i] = 0;
   for (int kq = 0; kq < 2; kq++)
      for (int w = 0; w < PHASE_WGS; w++)
         if (h[kq][w][7])
            for (int i = 0; i < 8; i++)
               if (16 * kq + i < n) out[16 * kq + i] += h[kq][w][i];
   if (reset) {
      memset(h, 0, sizeof(h));
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof(h)) != hipSuccess) return SPH_HIP_ERR_DEVICE;
   }
   return SPH_HIP_OK;
}
#endif

int sph_hip_selftest_sqrt(int device, uint64_t* mismatches, uint32_t* first_bad)
{
   if (hipSetDevice(device) != hipSuccess) {
      g_create_error = "sph_hip_selftest_sqrt: no such device";
      return SPH_HIP_ERR_NO_DEVICE;
   }
   unsigned long long* d = nullptr;
   unsigned long long h[3] = {0ull, 0xffffffffull, 0ull};
   if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess ||
       hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
      g_create_error = "sph_hip_selftest_sqrt: device memory";
      return SPH_HIP_ERR_DEVICE;
   }
   hipLaunchKernelGGL(k_selftest_sqrt, dim3(256 * 16), dim3(256), 0, 0, d);
   const hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
   (void)hipFree(d);
   if (e != hipSuccess) {
      g_create_error = std::string("sph_hip_selftest_sqrt: ") + hipGetErrorString(e);
      return SPH_HIP_ERR_DEVICE;
   }
   if (mismatches) *mismatches = h[0];
   if (first_bad) *first_bad = (uint32_t)h[1];
   if (getenv("SPH_HIP_DEBUG")) fprintf(stderr, "sph_hip_selftest_sqrt: %llu differ, bits 0x%08llx .. 0x%08llx\n", h[0], h[1], h[2]);
   return SPH_HIP_OK;
}

int sph_hip_params_default(sph_hip_params* p, float h, int cells_x, int cells_y, int cells_z)
{
   if (!p || !(h > 0.0f) || cells_x < 1 || cells_y < 1 || cells_z < 1) return SPH_HIP_ERR_INVALID;
   memset(p, 0, sizeof(*p));
   // reference src/sph.cpp:46-98, same order, same rounding points
   p->sim_scale = 1.0f;
   p->sim_scale_inv = 1.0f / p->sim_scale;
   p->h = h;
   p->h2 = (float)pow((double)h, 2.0);
   p->htimes2 = h * 2.0f;
   p->htimes2inv = 1.0f / p->htimes2;
   p->hscaled = h * p->sim_scale;
   p->hscaled2 = (float)pow((double)(h * p->sim_scale), 2.0);
   p->hscaled6 = (float)pow((double)(h * p->sim_scale), 6.0);
   p->hscaled9 = (float)pow((double)(h * p->sim_scale), 9.0);
   p->cells_x = cells_x;
   p->cells_y = cells_y;
   p->cells_z = cells_z;
   p->cell_size = 2.0f * h;
   p->max_x = p->cell_size * (float)cells_x;
   p->max_y = p->cell_size * (float)cells_y;
   p->max_z = p->cell_size * (float)cells_z;
   p->time_step = 0.001f;
   p->rho0 = 0.1f;
   p->stiffness = 0.001f;
   p->viscosity = 0.01f;
   p->damping = 0.001f;
   p->grav_const = 4.3009e-3f;
   p->central_mass = 1e+5f;
   p->central_pos[0] = p->max_x * 0.5f;
   p->central_pos[1] = p->max_y * 0.5f;
   p->central_pos[2] = p->max_z * 0.5f;
   p->softening = p->hscaled;
   p->cfl_limit = 10000.0f;
   p->cfl_limit2 = p->cfl_limit * p->cfl_limit;
   p->kernel1 = 315.0f / (64.0f * (float)(M_PI)*p->hscaled9);
   p->kernel2 = -45.0f / ((float)(M_PI)*p->hscaled6);
   p->kernel3 = -p->kernel2;
   p->examine_count = 32;
   // FULL grid: cell edge h*(1+1e-4) >= h, covering the same box
   const double edge = (double)h * 1.0001;
   p->full_cell_inv = (float)(1.0 / edge);
   p->full_cells_x = (int)ceil((double)p->max_x / edge);
   p->full_cells_y = (int)ceil((double)p->max_y / edge);
   p->full_cells_z = (int)ceil((double)p->max_z / edge);
   return SPH_HIP_OK;
}

static int create_impl(sph_hip_context** out, const sph_hip_params* params, int capacity, int mode,
                       int device, int plane_lo, int plane_hi, int halo)
{
   if (!out || !params || capacity < 1 ||
       (mode != SPH_HIP_MODE_REF && mode != SPH_HIP_MODE_FULL && mode != SPH_HIP_MODE_FULL_FAST)) {
      g_create_error = "sph_hip_create: invalid argument";
      return SPH_HIP_ERR_INVALID;
   }
   // FULL with the tolerance-mode pair arithmetic (SPH_HIP_ARITH=fast: experiments run the tools
   // that create plain FULL contexts - A/B, ablation, slab cost - in that mode)
   // - only together with SPH_HIP_ALLOW_DIAGNOSTIC=1, which those tools set: a variable left over
   // in a shell must not turn the bit-exact gates and bench.py's exact record into FAST runs)
   const char* arith_env = getenv_flag("SPH_HIP_ALLOW_DIAGNOSTIC") ? getenv("SPH_HIP_ARITH") : nullptr;
   const bool fast = mode == SPH_HIP_MODE_FULL_FAST ||
                     (mode == SPH_HIP_MODE_FULL && arith_env && strcmp(arith_env, "fast") == 0);
   if (fast) mode = SPH_HIP_MODE_FULL;
   *out = nullptr;
   int ndev = 0;
   hipError_t e = hipGetDeviceCount(&ndev);
   if (e != hipSuccess || ndev == 0 || device < 0 || device >= ndev) {
      g_create_error = "sph_hip_create: no usable HIP device (" +
                       std::string(e != hipSuccess ? hipGetErrorString(e) : "device index out of range") + ")";
      return SPH_HIP_ERR_NO_DEVICE;
   }
   sph_hip_context* ctx = new (std::nothrow) sph_hip_context();
   if (!ctx) return SPH_HIP_ERR_INVALID;
   ctx->prm = *params;
   ctx->mode = mode;
   ctx->fast = fast ? 1 : 0;
   // diagnostic switches, read once per context (not once per step)
   ctx->no_prehash = getenv_flag("SPH_HIP_NO_PREHASH");
   ctx->no_fused_integrate = getenv_flag("SPH_HIP_NO_FUSED_INTEGRATE");
   ctx->no_fused_slab = getenv_flag("SPH_HIP_NO_FUSED_SLAB");
   if (const char* v = getenv("SPH_HIP_CHUNKED")) ctx->chunked_giveups = v[0] == '1' ? 1 : 0;   // default: by count
   ctx->device = device;
   ctx->capacity = capacity;

   CellGrid& g = ctx->grid;
   if (mode == SPH_HIP_MODE_REF) {
      g.nx = params->cells_x; g.ny = params->cells_y; g.nz_global = params->cells_z;
      g.inv = params->htimes2inv;
   } else {
      g.nx = params->full_cells_x; g.ny = params->full_cells_y; g.nz_global = params->full_cells_z;
      g.inv = params->full_cell_inv;
   }
   if (plane_hi < 0) plane_hi = g.nz_global;  // whole grid
   if (g.nx < 1 || g.ny < 1 || g.nz_global < 1 || plane_lo < 0 || plane_hi > g.nz_global ||
       plane_lo >= plane_hi || (mode == SPH_HIP_MODE_REF && (plane_lo != 0 || plane_hi != g.nz_global))) {
      g_create_error = "sph_hip_create: bad grid shape or slab range";
      delete ctx;
      return SPH_HIP_ERR_INVALID;
   }
   // A slab with a neighbour feeds that neighbour's `halo` ghost planes from its own planes, and
   // the ghost planes of the two sides must not overlap in what they send: 2 * halo planes at
   // least (slab.plan_cuts plans with the same minimum).  A thinner slab would leave its
   // neighbour's ghosts incomplete without any error bit being raised.
   if (halo > 0 && (plane_lo > 0 || plane_hi < g.nz_global) && plane_hi - plane_lo < 2 * halo) {
      g_create_error = "sph_hip_create_slab: a slab next to another needs at least 2 * SPH_HIP_SLAB_HALO planes";
      delete ctx;
      return SPH_HIP_ERR_INVALID;
   }
   ctx->plane_lo = plane_lo;
   ctx->plane_hi = plane_hi;
   ctx->halo = halo;
   // planes held: the owned ones plus `halo` ghost planes on each side, clipped to the grid
   g.z0 = plane_lo - halo < 0 ? 0 : plane_lo - halo;
   const int z1 = plane_hi + halo > g.nz_global ? g.nz_global : plane_hi + halo;
   g.nz = z1 - g.z0;
   const long long ncells = (long long)g.nx * g.ny * g.nz;
   if (ncells > 0x7fff0000ll) {
      g_create_error = "sph_hip_create: grid too large";
      delete ctx;
      return SPH_HIP_ERR_INVALID;
   }
   g.ncells = (int)ncells;
   ctx->scan_tiles = div_up(g.ncells + 1, SCAN_TILE);
   ctx->eblocks = div_up(capacity, RED_THREADS);

   auto fail = [&](const char* what, hipError_t err) {
      g_create_error = std::string(what) + ": " + hipGetErrorString(err);
      free_all(ctx);
      delete ctx;
      return SPH_HIP_ERR_DEVICE;
   };
#define CREATE_TRY(expr)                                   \
   do {                                                    \
      hipError_t e_ = (expr);                              \
      if (e_ != hipSuccess) return fail(#expr, e_);        \
   } while (0)

   CREATE_TRY(hipSetDevice(device));
   CREATE_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
   ctx->stream = ctx->own_stream;
   ctx->ev = new hipEvent_t[EV_RING * 7]();
   for (int k = 0; k < EV_RING * 7; k++) CREATE_TRY(hipEventCreate(&ctx->ev[k]));
   for (int k = 0; k < 2; k++) CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_pace[k], hipEventDisableTiming));
   const size_t cap = (size_t)capacity;
   const int nbuf = (mode == SPH_HIP_MODE_FULL) ? 2 : 1;
   for (int b = 0; b < nbuf; b++) {
      CREATE_TRY(dev_alloc(&ctx->posm[b], cap));
      CREATE_TRY(dev_alloc(&ctx->velp[b], cap));
   }
   CREATE_TRY(dev_alloc(&ctx->key, cap));
   CREATE_TRY(dev_alloc(&ctx->slot, cap));
   CREATE_TRY(dev_alloc(&ctx->perm, cap));
   // cell arrays padded to whole scan tiles so vector accesses never run off the end
   const size_t cells_padded = (size_t)ctx->scan_tiles * SCAN_TILE + 16;
   CREATE_TRY(dev_alloc(&ctx->cell_count, cells_padded));
   CREATE_TRY(dev_alloc(&ctx->cell_start, cells_padded));
   CREATE_TRY(dev_alloc(&ctx->scan_part, (size_t)ctx->scan_tiles + 1));
   CREATE_TRY(dev_alloc(&ctx->big_cells, cap / RANK_BIG + 2));
   CREATE_TRY(hipMemsetAsync(ctx->big_cells, 0, sizeof(uint32_t), ctx->stream));
   CREATE_TRY(hipMemsetAsync(ctx->cell_count, 0, cells_padded * sizeof(uint32_t), ctx->stream));
   CREATE_TRY(hipMemsetAsync(ctx->cell_start, 0, cells_padded * sizeof(uint32_t), ctx->stream));
   CREATE_TRY(dev_alloc(&ctx->rho, cap));
   CREATE_TRY(dev_alloc(&ctx->acc, cap));
   CREATE_TRY(dev_alloc(&ctx->ncount, cap));
   CREATE_TRY(hipMemsetAsync(ctx->rho, 0, cap * sizeof(float), ctx->stream));
   CREATE_TRY(hipMemsetAsync(ctx->acc, 0, cap * sizeof(float4), ctx->stream));
   CREATE_TRY(hipMemsetAsync(ctx->ncount, 0, cap * sizeof(int32_t), ctx->stream));
   CREATE_TRY(dev_alloc(&ctx->meta, META_COUNT));
   CREATE_TRY(hipMemsetAsync(ctx->meta, 0, META_COUNT * sizeof(int32_t), ctx->stream));
   if (mode == SPH_HIP_MODE_FULL) {
      CREATE_TRY(dev_alloc(&ctx->velB, cap));
      CREATE_TRY(dev_alloc(&ctx->auxc, cap));
      CREATE_TRY(dev_alloc(&ctx->tile_desc, (size_t)div_up(capacity, TILE_THREADS) + 1));
      ctx->list_cap = NLIST_CAP;
      ctx->list_cap_max = NLIST_CAP_MAX;
      if (const char* v = getenv("SPH_HIP_LIST_CAP")) {
         const int c = atoi(v) / 2 * 2;
         if (c > 0) ctx->list_cap = ctx->list_cap_max = c < 2 ? 2 : (c > NLIST_CAP_MAX ? NLIST_CAP_MAX : c);
      }
      ctx->list_blocks = (size_t)div_up(capacity, TILE_THREADS) + 1;
      const size_t nlist_words = ctx->list_blocks * list_rows(ctx->list_cap) * TILE_THREADS;
      CREATE_TRY(dev_alloc(&ctx->nlist, nlist_words));
      // touched once here, so that the first step does not pay for mapping the pages
      CREATE_TRY(hipMemsetAsync(ctx->nlist, 0, nlist_words * sizeof(uint32_t), ctx->stream));
      CREATE_TRY(dev_alloc(&ctx->nlist_overflow, (size_t)div_up(capacity, TILE_THREADS) + 1));
      if (const char* v = getenv("SPH_HIP_UNTILED")) ctx->use_tiled = (v[0] == '1') ? 0 : 1;
      CREATE_TRY(hipHostMalloc((void**)&ctx->tile_feedback, TSTAT_COUNT * sizeof(int), hipHostMallocDefault));
      memset(ctx->tile_feedback, 0, TSTAT_COUNT * sizeof(int));
      CREATE_TRY(dev_alloc(&ctx->tile_stats, TSTAT_COUNT));
      CREATE_TRY(hipMemsetAsync(ctx->tile_stats, 0, TSTAT_COUNT * sizeof(int32_t), ctx->stream));
      CREATE_TRY(dev_alloc(&ctx->giveup_density, (size_t)div_up(capacity, TILE_THREADS) + 1));
      CREATE_TRY(dev_alloc(&ctx->giveup_accel, (size_t)div_up(capacity, TILE_THREADS) + 1));
      CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_density, hipEventDisableTiming));
      CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_border, hipEventDisableTiming));
      if (const char* v = getenv("SPH_HIP_TILE_CAP")) {
         const int c = atoi(v);
         if (c > 0) ctx->tile_cap_forced = c < 256 ? 256 : (c > 8000 ? 8000 : c / 32 * 32);  // 128 KiB at most
      }
   } else {
      CREATE_TRY(dev_alloc(&ctx->order, cap));
      CREATE_TRY(dev_alloc(&ctx->vox, cap * 3));
      CREATE_TRY(dev_alloc(&ctx->nb, cap * (size_t)params->examine_count));
      CREATE_TRY(dev_alloc(&ctx->nd, cap * (size_t)params->examine_count));
   }
   CREATE_TRY(dev_alloc(&ctx->epart, (size_t)2 * ctx->eblocks + 2));
   CREATE_TRY(hipMemsetAsync(ctx->epart, 0, sizeof(double) * 2, ctx->stream));
   CREATE_TRY(dev_alloc(&ctx->stats, 4));
   CREATE_TRY(hipHostMalloc((void**)&ctx->err_watch, 4 * sizeof(int32_t), hipHostMallocDefault));
   for (int i = 0; i < 4; i++) ctx->err_watch[i] = 0;
   CREATE_TRY(hipEventCreateWithFlags(&ctx->watch_event, hipEventDisableTiming));
   CREATE_TRY(dev_alloc(&ctx->stage, cap * 12));
   CREATE_TRY(hipStreamSynchronize(ctx->stream));
#undef CREATE_TRY
   *out = ctx;
   return SPH_HIP_OK;
}

int sph_hip_create(sph_hip_context** out, const sph_hip_params* params, int capacity, int mode,
                   int device)
{
   return create_impl(out, params, capacity, mode, device, 0, -1, 0);
}

int sph_hip_create_slab(sph_hip_context** out, const sph_hip_params* params, int capacity,
                        int device, int plane_lo, int plane_hi)
{
   return create_impl(out, params, capacity, SPH_HIP_MODE_FULL, device, plane_lo, plane_hi,
                      SPH_HIP_SLAB_HALO);
}

void sph_hip_destroy(sph_hip_context* ctx)
{
   if (!ctx) return;
   (void)hipSetDevice(ctx->device);
   if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
   free_all(ctx);
   delete ctx;
}

const char* sph_hip_last_error(const sph_hip_context* ctx)
{
   return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int sph_hip_set_stream(sph_hip_context* ctx, void* hip_stream)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
   return SPH_HIP_OK;
}

int sph_hip_set_params(sph_hip_context* ctx, const sph_hip_params* p)
{
   if (!ctx || !p) return SPH_HIP_ERR_INVALID;
   const sph_hip_params& o = ctx->prm;
   if (p->cells_x != o.cells_x || p->cells_y != o.cells_y || p->cells_z != o.cells_z ||
       p->full_cells_x != o.full_cells_x || p->full_cells_y != o.full_cells_y ||
       p->full_cells_z != o.full_cells_z || p->h != o.h || p->htimes2inv != o.htimes2inv ||
       p->full_cell_inv != o.full_cell_inv || p->examine_count != o.examine_count) {
      ctx->err = "sph_hip_set_params: grid shape, h and examine_count are fixed at creation";
      return SPH_HIP_ERR_INVALID;
   }
   ctx->prm = *p;
   return SPH_HIP_OK;
}

int sph_hip_set_arithmetic(sph_hip_context* ctx, int arithmetic)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || (arithmetic != SPH_HIP_ARITH_EXACT && arithmetic != SPH_HIP_ARITH_FAST)) {
      ctx->err = "sph_hip_set_arithmetic: FULL-mode contexts; SPH_HIP_ARITH_EXACT or SPH_HIP_ARITH_FAST";
      return SPH_HIP_ERR_INVALID;
   }
   if (ctx->fast == arithmetic) return SPH_HIP_OK;
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   ctx->fast = arithmetic;
   // the capacity levels belong to the kernels that run: worked out again at the next cell build,
   // and what the old kernels' levels reported means nothing for the new ones
   ctx->caps.n_cand = 0;
   return SPH_HIP_OK;
}

int sph_hip_get_arithmetic(const sph_hip_context* ctx) { return ctx ? ctx->fast : SPH_HIP_ERR_INVALID; }

int sph_hip_get_params(const sph_hip_context* ctx, sph_hip_params* out)
{
   if (!ctx || !out) return SPH_HIP_ERR_INVALID;
   *out = ctx->prm;
   return SPH_HIP_OK;
}

// shared by sph_hip_upload (ids = 0..n-1) and sph_hip_slab_upload (caller's global ids)
static int upload_impl(sph_hip_context* ctx, int n, const float* pos, const float* vel,
                       const float* mass, const uint32_t* ids, int uniform_mass,
                       const void* device_records = nullptr)
{
   if (n < 0 || (n > 0 && !device_records && (!pos || !vel || !mass))) {
      ctx->err = "upload: null array or negative count";
      return SPH_HIP_ERR_INVALID;
   }
   if (n > ctx->capacity) {
      ctx->err = "upload: more particles than the context capacity";
      return SPH_HIP_ERR_CAPACITY;
   }
   int rc_prehash = drop_prehash(ctx);
   if (rc_prehash) return rc_prehash;
   // a slab's entry count changes every step, so its launches are sized by the capacity
   const bool whole = ctx->plane_lo == 0 && ctx->plane_hi == ctx->grid.nz_global;
   ctx->n = whole ? n : ctx->capacity;
   ctx->n_owned = n;
   ctx->cur = 0;
   ctx->uniform_mass = uniform_mass;
   ctx->ev_steps = 0;
   ctx->err_watch[0] = 0;   // (the upload clears the device's error word below)
   ctx->watch_pending = 0;
   // before the first cell build everything uploaded is live and owned, in upload order
   const int32_t meta[META_COUNT] = {n, n, 0, n, 0, n, 0, 0};
   SPH_TRY(hipMemcpyAsync(ctx->meta, meta, sizeof(meta), hipMemcpyHostToDevice, ctx->stream));
   if (n > 0 && device_records) {
      // the state is on the device already, as message records (sph_hip_slab_export_records)
      hipLaunchKernelGGL(k_import_records, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream,
                         (const float4*)device_records, n, ctx->posm[0], ctx->velp[0]);
      SPH_TRY(hipGetLastError());
   } else if (n > 0) {
      float* spos = ctx->stage;
      float* svel = spos + 3 * (size_t)n;
      float* smass = svel + 3 * (size_t)n;
      uint32_t* sids = reinterpret_cast<uint32_t*>(smass + (size_t)n);
      hipStream_t st = ctx->stream;
      SPH_TRY(hipMemcpyAsync(spos, pos, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
      SPH_TRY(hipMemcpyAsync(svel, vel, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st));
      SPH_TRY(hipMemcpyAsync(smass, mass, sizeof(float) * n, hipMemcpyHostToDevice, st));
      if (ids) SPH_TRY(hipMemcpyAsync(sids, ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_import, dim3(div_up(n, 256)), dim3(256), 0, st, spos, svel, smass,
                         ids ? sids : (const uint32_t*)nullptr, n, ctx->posm[0], ctx->velp[0]);
      SPH_TRY(hipGetLastError());
   }
   if (ctx->tile_feedback) {
      // Tile capacity of the first steps: nothing has run yet to report what the scene needs,
      // and the host may enqueue many steps before the first one finishes, so sort the upload
      // once here (the first step's cell build then finds it already in canonical order) and
      // read the largest tile back.
      memset(ctx->tile_feedback, 0, TSTAT_COUNT * sizeof(int));
      if (n > 0 && ctx->use_tiled) {
         int rc = launch_cell_build(ctx);
         if (rc) return rc;
         int stats[TSTAT_COUNT];
         SPH_TRY(hipMemcpyAsync(stats, ctx->tile_stats, sizeof(stats), hipMemcpyDeviceToHost,
                                ctx->stream));
         SPH_TRY(hipStreamSynchronize(ctx->stream));
         memcpy(ctx->tile_feedback, stats, sizeof(stats));
      }
   }
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   return SPH_HIP_OK;
}

static int all_same_mass(int n, const float* mass)
{
   for (int i = 1; i < n; i++)
      if (memcmp(&mass[i], &mass[0], sizeof(float)) != 0) return 0;
   return 1;
}

int sph_hip_upload(sph_hip_context* ctx, int n, const float* pos, const float* vel,
                   const float* mass)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   // the reference gives every particle the same mass (src/sph.cpp:105-108); when the upload
   // does too, the tiled kernels skip the per-neighbour mass gather (bit-identical results)
   return upload_impl(ctx, n, pos, vel, mass, nullptr, (n > 0 && mass) ? all_same_mass(n, mass) : 0);
}

int sph_hip_slab_upload(sph_hip_context* ctx, int n, const float* pos, const float* vel,
                        const float* mass, const uint32_t* ids, int all_masses_equal)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || (n > 0 && !ids)) {
      ctx->err = "sph_hip_slab_upload: FULL-mode contexts only, ids required";
      return SPH_HIP_ERR_INVALID;
   }
   return upload_impl(ctx, n, pos, vel, mass, ids, all_masses_equal ? 1 : 0);
}

// owned count right now (device value); synchronises the stream
static int owned_count(sph_hip_context* ctx, int32_t* meta_out)
{
   int32_t meta[META_COUNT];
   SPH_TRY(hipMemcpyAsync(meta, ctx->meta, sizeof(meta), hipMemcpyDeviceToHost, ctx->stream));
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   if (meta_out) memcpy(meta_out, meta, sizeof(meta));
   ctx->n_owned = meta[META_OWN_END] - meta[META_OWN_BEGIN];
   return SPH_HIP_OK;
}

static int download_impl(sph_hip_context* ctx, int compact, int n, uint32_t* ids, float* pos,
                         float* vel, float* density, float* acc, int32_t* neighbor_count)
{
   if (n == 0) return SPH_HIP_OK;
   float* spos = ctx->stage;
   float* svel = spos + 3 * (size_t)n;
   float* srho = svel + 3 * (size_t)n;
   float* sacc = srho + (size_t)n;
   int32_t* scnt = reinterpret_cast<int32_t*>(sacc + 3 * (size_t)n);
   uint32_t* sids = reinterpret_cast<uint32_t*>(scnt + (size_t)n);
   hipStream_t st = ctx->stream;
   hipLaunchKernelGGL(k_export, dim3(div_up(ctx->n, 256)), dim3(256), 0, st, ctx->posm[ctx->cur],
                      ctx->velp[ctx->cur], ctx->rho, ctx->acc, ctx->ncount, ctx->meta, compact,
                      pos ? spos : nullptr, vel ? svel : nullptr, density ? srho : nullptr,
                      acc ? sacc : nullptr, neighbor_count ? scnt : nullptr, ids ? sids : nullptr);
   SPH_TRY(hipGetLastError());
   if (pos) SPH_TRY(hipMemcpyAsync(pos, spos, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, st));
   if (vel) SPH_TRY(hipMemcpyAsync(vel, svel, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, st));
   if (density) SPH_TRY(hipMemcpyAsync(density, srho, sizeof(float) * n, hipMemcpyDeviceToHost, st));
   if (acc) SPH_TRY(hipMemcpyAsync(acc, sacc, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, st));
   if (neighbor_count)
      SPH_TRY(hipMemcpyAsync(neighbor_count, scnt, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
   if (ids) SPH_TRY(hipMemcpyAsync(ids, sids, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
   SPH_TRY(hipStreamSynchronize(st));
   return SPH_HIP_OK;
}

int sph_hip_download(sph_hip_context* ctx, float* pos, float* vel, float* density, float* acc,
                     int32_t* neighbor_count)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (!(ctx->plane_lo == 0 && ctx->plane_hi == ctx->grid.nz_global)) {
      ctx->err = "sph_hip_download: slab contexts use sph_hip_slab_download";
      return SPH_HIP_ERR_INVALID;
   }
   return download_impl(ctx, 0, ctx->n_owned, nullptr, pos, vel, density, acc, neighbor_count);
}

// ---- asynchronous host mirror ---------------------------------------------------------------

int sph_hip_host_register(void* ptr, size_t bytes)
{
   if (!ptr || bytes == 0) return SPH_HIP_ERR_INVALID;
   const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
   if (e != hipSuccess) {
      (void)hipGetLastError();
      g_create_error = std::string("sph_hip_host_register: ") + hipGetErrorString(e);
      return SPH_HIP_ERR_DEVICE;
   }
   return SPH_HIP_OK;
}

int sph_hip_host_unregister(void* ptr)
{
   if (!ptr) return SPH_HIP_ERR_INVALID;
   if (hipHostUnregister(ptr) != hipSuccess) {
      (void)hipGetLastError();
      return SPH_HIP_ERR_DEVICE;
   }
   return SPH_HIP_OK;
}

int sph_hip_download_done(sph_hip_context* ctx, int wait)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (!ctx->mirror_busy) return 1;
   if (wait) {
      SPH_TRY(hipEventSynchronize(ctx->ev_copied));
   } else {
      const hipError_t e = hipEventQuery(ctx->ev_copied);
      if (e == hipErrorNotReady) {
         (void)hipGetLastError();
         return 0;
      }
      SPH_TRY(e);
   }
   ctx->mirror_busy = 0;
   return 1;
}

int sph_hip_download_async(sph_hip_context* ctx, float* pos, float* vel, float* density, float* acc,
                           int32_t* neighbor_count, int32_t* voxel_counts, int* started)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (started) *started = 0;
   if (!(ctx->plane_lo == 0 && ctx->plane_hi == ctx->grid.nz_global)) {
      ctx->err = "sph_hip_download_async: whole-grid contexts only";
      return SPH_HIP_ERR_INVALID;
   }
   // the previous mirror is still on its way: this request is dropped (a mirror is a picture of
   // the latest state, not a queue) - its staging must not be overwritten under the copy
   rc = sph_hip_download_done(ctx, 0);
   if (rc < 0) return rc;
   if (rc == 0) return SPH_HIP_OK;
   const int n = ctx->n_owned;
   const sph_hip_params& prm = ctx->prm;
   const size_t cells = (size_t)prm.cells_x * prm.cells_y * prm.cells_z;
   if (!ctx->mirror_stage) {
      SPH_TRY(hipMalloc((void**)&ctx->mirror_stage, sizeof(float) * ((size_t)ctx->capacity * 11 + cells)));
      int least = 0, greatest = 0;
      SPH_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
      SPH_TRY(hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, least));
      SPH_TRY(hipEventCreateWithFlags(&ctx->ev_exported, hipEventDisableTiming));
      SPH_TRY(hipEventCreateWithFlags(&ctx->ev_copied, hipEventDisableTiming));
   }
   float* spos = ctx->mirror_stage;
   float* svel = spos + 3 * (size_t)n;
   float* srho = svel + 3 * (size_t)n;
   float* sacc = srho + (size_t)n;
   int32_t* scnt = reinterpret_cast<int32_t*>(sacc + 3 * (size_t)n);
   int32_t* svox = reinterpret_cast<int32_t*>(ctx->mirror_stage + (size_t)ctx->capacity * 11);
   hipStream_t st = ctx->stream;
   if (n > 0)
      hipLaunchKernelGGL(k_export, dim3(div_up(ctx->n, 256)), dim3(256), 0, st, ctx->posm[ctx->cur],
                         ctx->velp[ctx->cur], ctx->rho, ctx->acc, ctx->ncount, ctx->meta, 0,
                         pos ? spos : nullptr, vel ? svel : nullptr, density ? srho : nullptr,
                         acc ? sacc : nullptr, neighbor_count ? scnt : nullptr, (uint32_t*)nullptr);
   if (voxel_counts) {
      SPH_TRY(hipMemsetAsync(svox, 0, cells * sizeof(int32_t), st));
      if (n > 0)
         hipLaunchKernelGGL(k_voxel_counts, dim3(div_up(ctx->n, 256)), dim3(256), 0, st,
                            ctx->posm[ctx->cur], ctx->meta, prm.htimes2inv, prm.cells_x, prm.cells_y,
                            prm.cells_z, svox);
   }
   SPH_TRY(hipGetLastError());
   // the copies run on their own low-priority stream: the next steps' kernels do not wait for PCIe
   SPH_TRY(hipEventRecord(ctx->ev_exported, st));
   hipStream_t cs = ctx->copy_stream;
   SPH_TRY(hipStreamWaitEvent(cs, ctx->ev_exported, 0));
   if (pos) SPH_TRY(hipMemcpyAsync(pos, spos, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, cs));
   if (vel) SPH_TRY(hipMemcpyAsync(vel, svel, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, cs));
   if (density) SPH_TRY(hipMemcpyAsync(density, srho, sizeof(float) * n, hipMemcpyDeviceToHost, cs));
   if (acc) SPH_TRY(hipMemcpyAsync(acc, sacc, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, cs));
   if (neighbor_count)
      SPH_TRY(hipMemcpyAsync(neighbor_count, scnt, sizeof(int32_t) * n, hipMemcpyDeviceToHost, cs));
   if (voxel_counts)
      SPH_TRY(hipMemcpyAsync(voxel_counts, svox, sizeof(int32_t) * cells, hipMemcpyDeviceToHost, cs));
   SPH_TRY(hipEventRecord(ctx->ev_copied, cs));
   ctx->mirror_busy = 1;
   if (started) *started = 1;
   return SPH_HIP_OK;
}

int sph_hip_slab_download(sph_hip_context* ctx, int max_rows, int32_t* rows, uint32_t* ids,
                          float* pos, float* vel, float* density, float* acc,
                          int32_t* neighbor_count)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   int32_t meta[META_COUNT];
   if ((rc = owned_count(ctx, meta))) return rc;
   ctx->err_watch[0] = meta[META_ERRORS];
   if ((rc = watch_check(ctx, "sph_hip_slab_download"))) return rc;
   if (rows) *rows = ctx->n_owned;
   if (ctx->n_owned > max_rows) {
      ctx->err = "sph_hip_slab_download: caller's arrays are too small";
      return SPH_HIP_ERR_CAPACITY;
   }
   return download_impl(ctx, 1, ctx->n_owned, ids, pos, vel, density, acc, neighbor_count);
}

int sph_hip_slab_download_mass(sph_hip_context* ctx, int max_rows, int32_t* rows, float* mass)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if ((rc = owned_count(ctx, nullptr))) return rc;
   if (rows) *rows = ctx->n_owned;
   if (!mass || ctx->n_owned > max_rows) {
      ctx->err = "sph_hip_slab_download_mass: caller's array is missing or too small";
      return SPH_HIP_ERR_CAPACITY;
   }
   if (ctx->n_owned == 0) return SPH_HIP_OK;
   hipLaunchKernelGGL(k_export_mass, dim3(div_up(ctx->n, 256)), dim3(256), 0, ctx->stream,
                      ctx->posm[ctx->cur], ctx->meta, ctx->stage);
   SPH_TRY(hipGetLastError());
   SPH_TRY(hipMemcpyAsync(mass, ctx->stage, sizeof(float) * ctx->n_owned, hipMemcpyDeviceToHost,
                          ctx->stream));
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   return SPH_HIP_OK;
}

int sph_hip_slab_export_records(sph_hip_context* ctx, void* device_records, int max_records, int32_t* rows)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   int32_t meta[META_COUNT];
   if ((rc = owned_count(ctx, meta))) return rc;
   ctx->err_watch[0] = meta[META_ERRORS];
   if ((rc = watch_check(ctx, "sph_hip_slab_export_records"))) return rc;
   if (rows) *rows = ctx->n_owned;
   if (!device_records || ctx->n_owned > max_records) {
      ctx->err = "sph_hip_slab_export_records: caller's buffer is missing or too small";
      return SPH_HIP_ERR_CAPACITY;
   }
   if (ctx->n_owned == 0) return SPH_HIP_OK;
   hipLaunchKernelGGL(k_export_records, dim3(div_up(ctx->n_owned, 256)), dim3(256), 0, ctx->stream,
                      ctx->posm[ctx->cur], ctx->velp[ctx->cur], ctx->meta, (float4*)device_records);
   SPH_TRY(hipGetLastError());
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   return SPH_HIP_OK;
}

int sph_hip_slab_upload_records(sph_hip_context* ctx, const void* device_records, int n, int all_masses_equal)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || n < 0 || (n > 0 && !device_records)) {
      ctx->err = "sph_hip_slab_upload_records: FULL-mode contexts only, records required";
      return SPH_HIP_ERR_INVALID;
   }
   return upload_impl(ctx, n, nullptr, nullptr, nullptr, nullptr, all_masses_equal ? 1 : 0, device_records);
}

int sph_hip_slab_status(sph_hip_context* ctx, int32_t* live, int32_t* owned, int32_t* errors)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   int32_t meta[META_COUNT];
   if ((rc = owned_count(ctx, meta))) return rc;
   if (live) *live = meta[META_N_LIVE];
   if (owned) *owned = meta[META_OWN_END] - meta[META_OWN_BEGIN];
   if (errors) *errors = meta[META_ERRORS];
   return SPH_HIP_OK;
}

int sph_hip_slab_poll_errors(sph_hip_context* ctx, int32_t* errors)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->watch_pending) SPH_TRY(hipEventSynchronize(ctx->watch_event));
   ctx->watch_pending = 0;
   if (errors) *errors = ctx->err_watch[0];
   if ((rc = watch_check(ctx, "sph_hip_slab_poll_errors"))) return rc;
   return watch_enqueue(ctx);
}

size_t sph_hip_slab_message_bytes(int capacity_records)
{
   return sizeof(int32_t) * SLAB_HEADER_INTS + (size_t)capacity_records * 2 * sizeof(float4);
}

int sph_hip_slab_pack(sph_hip_context* ctx, void* left_device, void* right_device,
                      int capacity_records)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || capacity_records < 0) return SPH_HIP_ERR_INVALID;
   hipStream_t st = ctx->stream;
   ctx->had_exchange = 1;
   if ((rc = drop_prehash(ctx))) return rc;
   ctx->early_exchange = 0;  // this pack sees every particle after the integrate
   ctx->may_hold_dead = 1;   // ... and marks the ones to drop with the dead id
   if (left_device) SPH_TRY(hipMemsetAsync(left_device, 0, sizeof(int32_t) * SLAB_HEADER_INTS, st));
   if (right_device) SPH_TRY(hipMemsetAsync(right_device, 0, sizeof(int32_t) * SLAB_HEADER_INTS, st));
   hipLaunchKernelGGL(k_slab_pack, dim3(div_up(ctx->n, 256)), dim3(256), 0, st, ctx->posm[ctx->cur],
                      ctx->velp[ctx->cur], ctx->meta, ctx->grid, ctx->plane_lo, ctx->plane_hi,
                      ctx->halo, left_device ? 1 : 0, right_device ? 1 : 0,
                      (SlabMsg*)left_device, (SlabMsg*)right_device, capacity_records);
   SPH_TRY(hipGetLastError());
   return SPH_HIP_OK;
}

int sph_hip_slab_unpack(sph_hip_context* ctx, const void* left_device, const void* right_device,
                        int capacity_records)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || capacity_records < 0) return SPH_HIP_ERR_INVALID;
   ctx->had_exchange = 1;
   // (a slab's fused step has hashed its owned entries for the next build, which hashes what is
   // unpacked here: that stays; a whole-grid prehash knows nothing of new entries)
   if (ctx->prehashed != 2 && (rc = drop_prehash(ctx))) return rc;
   // entries behind the live ones; n_in = n_live + what the messages hold
   hipLaunchKernelGGL(k_slab_unpack, dim3(div_up(2 * capacity_records, 256) + 1), dim3(256), 0,
                      ctx->stream, (const SlabMsg*)left_device, (const SlabMsg*)right_device,
                      ctx->posm[ctx->cur], ctx->velp[ctx->cur], ctx->meta, ctx->capacity,
                      capacity_records);
   SPH_TRY(hipGetLastError());
   return SPH_HIP_OK;
}

int sph_hip_slab_step_begin(sph_hip_context* ctx, void* left_device, void* right_device,
                            int capacity_records, void* exchange_stream)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || capacity_records < 0) return SPH_HIP_ERR_INVALID;
   if (!ctx->use_tiled) {
      ctx->err = "sph_hip_slab_step_begin: needs the tiled kernels (SPH_HIP_UNTILED is set)";
      return SPH_HIP_ERR_INVALID;
   }
   if ((left_device != nullptr) != (ctx->plane_lo > 0) ||
       (right_device != nullptr) != (ctx->plane_hi < ctx->grid.nz_global)) {
      ctx->err = "sph_hip_slab_step_begin: one message buffer per existing neighbour, no other";
      return SPH_HIP_ERR_INVALID;
   }
   ctx->had_exchange = 1;
   if (ctx->prehashed != 2 && (rc = drop_prehash(ctx))) return rc;
   hipStream_t st = ctx->stream;
   hipStream_t side = exchange_stream ? (hipStream_t)exchange_stream : st;
   hipEvent_t* ev = ctx->ev + 7 * (ctx->ev_steps % EV_RING);
   if ((rc = pace_host(ctx))) return rc;
   const int level = ctx->slab_step_level = next_step_level(ctx, true);
   const bool phases = level == SPH_HIP_TIMING_PHASES, sums = level == SPH_HIP_TIMING_SUMS;
   if (phases) SPH_TRY(hipEventRecord(ev[0], st));
   if ((rc = launch_cell_build(ctx, left_device, right_device))) return rc;
   if (phases || sums) SPH_TRY(hipEventRecord(ev[1], st));
   if ((rc = launch_density(ctx))) return rc;
   if (phases) SPH_TRY(hipEventRecord(ev[3], st));
   ctx->early_exchange = 1;
   if (ctx->n == 0) return SPH_HIP_OK;
   // border work on the exchange stream, behind the density pass: it runs next to the interior's
   // acceleration (sph_hip_slab_step_end, main stream) and is short, so the messages leave early
   if (side != st) {
      SPH_TRY(hipEventRecord(ctx->ev_density, st));
      SPH_TRY(hipStreamWaitEvent(side, ctx->ev_density, 0));
   }
   // The two parts of the acceleration launch do the rest of the step themselves (FusedStep):
   // integrate into the other pair of state buffers, hash for the next build, and - the border
   // part - the messages.  SPH_HIP_NO_FUSED_SLAB=1 keeps k_slab_pack_early + k_integrate.
   ctx->slab_fused = ctx->no_fused_slab ? 0 : 1;
   ctx->slab_msgs[0] = left_device;
   ctx->slab_msgs[1] = right_device;
   ctx->slab_msg_capacity = capacity_records;
   if (ctx->slab_fused) {
      const SlabFused sf = {left_device, right_device, capacity_records};
      if ((rc = launch_accel(ctx, 1, side, true, &sf))) return rc;
      if (side != st) SPH_TRY(hipEventRecord(ctx->ev_border, side));
      ctx->border_stream = side;
      return SPH_HIP_OK;
   }
   if ((rc = launch_accel(ctx, 1, side))) return rc;
   const PairConsts k = pair_consts(ctx->prm, ctx->fast != 0);
   const SlabZone zone = slab_zone(ctx);
   if (unit_scale(ctx->prm))
      hipLaunchKernelGGL(k_slab_pack_early<true>, dim3(SLAB_PACK_BLOCKS), dim3(256), 0, side,
                         ctx->posm[ctx->cur], ctx->velp[ctx->cur], ctx->acc, ctx->meta, k, ctx->grid,
                         zone, (SlabMsg*)left_device, (SlabMsg*)right_device, capacity_records);
   else
      hipLaunchKernelGGL(k_slab_pack_early<false>, dim3(SLAB_PACK_BLOCKS), dim3(256), 0, side,
                         ctx->posm[ctx->cur], ctx->velp[ctx->cur], ctx->acc, ctx->meta, k, ctx->grid,
                         zone, (SlabMsg*)left_device, (SlabMsg*)right_device, capacity_records);
   SPH_TRY(hipGetLastError());
   if (side != st) SPH_TRY(hipEventRecord(ctx->ev_border, side));
   ctx->border_stream = side;
   return SPH_HIP_OK;
}

int sph_hip_slab_step_end(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || !ctx->early_exchange) {
      ctx->err = "sph_hip_slab_step_end: no sph_hip_slab_step_begin before it";
      return SPH_HIP_ERR_INVALID;
   }
   hipStream_t st = ctx->stream;
   hipEvent_t* ev = ctx->ev + 7 * (ctx->ev_steps % EV_RING);
   const int level = ctx->slab_step_level;
   const bool phases = level == SPH_HIP_TIMING_PHASES, sums = level == SPH_HIP_TIMING_SUMS;
   if (ctx->slab_fused && ctx->n > 0) {
      const SlabFused sf = {ctx->slab_msgs[0], ctx->slab_msgs[1], ctx->slab_msg_capacity};
      if ((rc = launch_accel(ctx, 2, st, true, &sf))) return rc;
      if (phases || sums) SPH_TRY(hipEventRecord(ev[5], st));
      // what follows on this stream (unpack, the next build) reads what the border part wrote
      if (ctx->border_stream != st) SPH_TRY(hipStreamWaitEvent(st, ctx->ev_border, 0));
      ctx->cur ^= 1;                                    // the new state is in the other buffers
      ctx->energy_blocks = div_up(ctx->n, TILE_THREADS);
      ctx->prehashed = 2;
      if (phases) SPH_TRY(hipEventRecord(ev[6], st));
      if (level != SPH_HIP_TIMING_OFF) ctx->ev_steps++;
      return SPH_HIP_OK;
   }
   if ((rc = launch_accel(ctx, 2, st))) return rc;
   if (phases || sums) SPH_TRY(hipEventRecord(ev[5], st));
   // the integrate needs the border planes' acceleration (and must not move them under the pack)
   if (ctx->n > 0 && ctx->border_stream != st) SPH_TRY(hipStreamWaitEvent(st, ctx->ev_border, 0));
   if ((rc = launch_integrate(ctx))) return rc;
   if (phases) SPH_TRY(hipEventRecord(ev[6], st));
   if (level != SPH_HIP_TIMING_OFF) ctx->ev_steps++;
   return SPH_HIP_OK;
}

// ---- native RCCL exchange -------------------------------------------------------------------

#define SPH_NCCL_TRY(call)                                                                    \
   do {                                                                                       \
      const ncclResult_t r_ = (call);                                                         \
      if (r_ != ncclSuccess) {                                                                \
         ctx->err = std::string(#call " failed: ") + api->GetErrorString(r_);                 \
         return SPH_HIP_ERR_DEVICE;                                                          \
      }                                                                                       \
   } while (0)

int sph_hip_rccl_unique_id(void* id_out, int id_bytes)
{
   std::string why;
   const RcclApi* api = rccl_api(&why);
   if (!api || !id_out || id_bytes < (int)sizeof(ncclUniqueId)) {
      g_create_error = api ? "sph_hip_rccl_unique_id: buffer too small (128 bytes needed)" : why;
      return SPH_HIP_ERR_INVALID;
   }
   ncclUniqueId id;
   if (api->GetUniqueId(&id) != ncclSuccess) {
      g_create_error = "ncclGetUniqueId failed";
      return SPH_HIP_ERR_DEVICE;
   }
   memcpy(id_out, &id, sizeof(id));
   return SPH_HIP_OK;
}

int sph_hip_slab_comm_init(sph_hip_context* ctx, const void* id, int id_bytes, int rank, int nranks,
                           int capacity_records)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_FULL || !id || id_bytes < (int)sizeof(ncclUniqueId) || rank < 0 ||
       rank >= nranks || capacity_records < 1) {
      ctx->err = "sph_hip_slab_comm_init: bad arguments";
      return SPH_HIP_ERR_INVALID;
   }
   // slabs are ordered by rank along z: the neighbours of rank r are r - 1 and r + 1
   if ((rank > 0) != (ctx->plane_lo > 0) || (rank + 1 < nranks) != (ctx->plane_hi < ctx->grid.nz_global)) {
      ctx->err = "sph_hip_slab_comm_init: the slab's planes do not match its rank (rank 0 owns "
                 "plane 0, the last rank the last plane)";
      return SPH_HIP_ERR_INVALID;
   }
   if (ctx->comm) {
      ctx->err = "sph_hip_slab_comm_init: already initialised";
      return SPH_HIP_ERR_INVALID;
   }
   std::string why;
   const RcclApi* api = rccl_api(&why);
   if (!api) {
      ctx->err = why;
      return SPH_HIP_ERR_DEVICE;
   }
   SPH_TRY(hipSetDevice(ctx->device));
   SlabComm* c = new (std::nothrow) SlabComm();
   if (!c) return SPH_HIP_ERR_DEVICE;
   ctx->comm = c;   // from here on sph_hip_destroy cleans up
   c->rank = rank;
   c->nranks = nranks;
   c->capacity_records = c->active_records = capacity_records;
   c->bytes = sph_hip_slab_message_bytes(capacity_records);
   int least = 0, greatest = 0;
   SPH_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
   SPH_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
   SPH_TRY(hipEventCreateWithFlags(&c->packed, hipEventDisableTiming));
   SPH_TRY(hipEventCreateWithFlags(&c->arrived, hipEventDisableTiming));
   void** bufs[4] = {&c->send_left, &c->recv_left, &c->send_right, &c->recv_right};
   for (int b = 0; b < 4; b++) {
      if (b < 2 ? rank == 0 : rank + 1 == nranks) continue;   // no neighbour on that side
      SPH_TRY(hipMalloc(bufs[b], c->bytes));
      SPH_TRY(hipMemsetAsync(*bufs[b], 0, c->bytes, ctx->stream));
   }
   SPH_TRY(hipMalloc((void**)&c->trim_word, sizeof(int32_t)));
   SPH_TRY(hipMalloc((void**)&c->fill_word, sizeof(int32_t)));
   SPH_TRY(hipHostMalloc((void**)&c->fill_host, sizeof(int32_t), hipHostMallocDefault));
   c->fill_host[0] = 0;
   SPH_TRY(hipEventCreateWithFlags(&c->fill_arrived, hipEventDisableTiming));
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   ncclUniqueId uid;
   memcpy(&uid, id, sizeof(uid));
   SPH_NCCL_TRY(api->CommInitRank(&c->comm, nranks, uid, rank));
   return SPH_HIP_OK;
}

// record count of the fuller of a slab's two send messages (their headers' first word)
__global__ void k_msg_fill(const SlabMsg* __restrict__ left, const SlabMsg* __restrict__ right,
                           int32_t* __restrict__ out)
{
   const int a = left ? left->header[0] : 0, b = right ? right->header[0] : 0;
   out[0] = a > b ? a : b;
}

namespace {
// Trimmed messages (sph_hip_slab_comm_trim) grow before they overflow - an overflow drops records
// and the run is lost.  Called by every rank at the same steps (every SLAB_GROW_EVERY-th of
// sph_hip_slab_comm_run, while active < capacity - the same on every rank): look at the reduced
// fill the PREVIOUS call requested (it waits for that one copy: the exchange it rode behind is
// SLAB_GROW_EVERY steps old), go back to the allocated size when any rank's message was more than
// 4/5 full, and request the next one: max over the two send headers -> ncclAllReduce(max) on the
// exchange stream -> asynchronous copy to pinned memory.  Every rank sees the same number at the
// same step, so all of them switch together and sender and receiver keep agreeing on the size.
int comm_grow_if_needed(sph_hip_context* ctx)
{
   SlabComm* c = ctx->comm;
   const RcclApi* api = rccl_api(nullptr);
   if (c->fill_pending) {
      SPH_TRY(hipEventSynchronize(c->fill_arrived));
      c->fill_pending = false;
      const long long most = c->fill_host[0];
      if (most * SLAB_GROW_FILL_DEN > (long long)c->active_records * SLAB_GROW_FILL_NUM &&
          c->active_records < c->capacity_records) {
         c->active_records = c->capacity_records;
         c->bytes = sph_hip_slab_message_bytes(c->active_records);
         c->growths++;
      }
   }
   if (c->active_records >= c->capacity_records || c->nranks < 2) return SPH_HIP_OK;
   // (the headers are read on the context's stream, where this step's cell build will zero them;
   // the reduction rides on the exchange stream, in the same place between two exchanges on every rank)
   hipLaunchKernelGGL(k_msg_fill, dim3(1), dim3(1), 0, ctx->stream, (const SlabMsg*)c->send_left,
                      (const SlabMsg*)c->send_right, c->fill_word);
   SPH_TRY(hipGetLastError());
   SPH_TRY(hipEventRecord(c->packed, ctx->stream));
   SPH_TRY(hipStreamWaitEvent(c->stream, c->packed, 0));
   SPH_NCCL_TRY(api->AllReduce(c->fill_word, c->fill_word, 1, ncclInt32, ncclMax, c->comm, c->stream));
   SPH_TRY(hipMemcpyAsync(c->fill_host, c->fill_word, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
   SPH_TRY(hipEventRecord(c->fill_arrived, c->stream));
   c->fill_pending = true;
   return SPH_HIP_OK;
}

// both directions in one group on the exchange stream
int comm_send_recv(sph_hip_context* ctx)
{
   SlabComm* c = ctx->comm;
   const RcclApi* api = rccl_api(nullptr);
   if (c->rank == 0 && c->rank + 1 == c->nranks) return SPH_HIP_OK;
   SPH_NCCL_TRY(api->GroupStart());
   if (c->rank > 0) {
      SPH_NCCL_TRY(api->Send(c->send_left, c->bytes, ncclChar, c->rank - 1, c->comm, c->stream));
      SPH_NCCL_TRY(api->Recv(c->recv_left, c->bytes, ncclChar, c->rank - 1, c->comm, c->stream));
   }
   if (c->rank + 1 < c->nranks) {
      SPH_NCCL_TRY(api->Send(c->send_right, c->bytes, ncclChar, c->rank + 1, c->comm, c->stream));
      SPH_NCCL_TRY(api->Recv(c->recv_right, c->bytes, ncclChar, c->rank + 1, c->comm, c->stream));
   }
   SPH_NCCL_TRY(api->GroupEnd());
   return SPH_HIP_OK;
}
} // namespace

int sph_hip_slab_comm_run(sph_hip_context* ctx, int steps)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   SlabComm* c = ctx->comm;
   if (!c || !c->comm || steps < 0) {
      ctx->err = "sph_hip_slab_comm_run: sph_hip_slab_comm_init first";
      return SPH_HIP_ERR_INVALID;
   }
   hipStream_t st = ctx->stream;
   if (!c->primed) {
      // the first ghosts: pack -> send/recv -> unpack, serially
      if ((rc = sph_hip_slab_pack(ctx, c->send_left, c->send_right, c->active_records))) return rc;
      SPH_TRY(hipEventRecord(c->packed, st));
      SPH_TRY(hipStreamWaitEvent(c->stream, c->packed, 0));
      if ((rc = comm_send_recv(ctx))) return rc;
      SPH_TRY(hipEventRecord(c->arrived, c->stream));
      SPH_TRY(hipStreamWaitEvent(st, c->arrived, 0));
      if ((rc = sph_hip_slab_unpack(ctx, c->recv_left, c->recv_right, c->active_records))) return rc;
      c->primed = true;
   }
   for (int s = 0; s < steps; s++) {
      // every 16 steps: ask for the device's error word (asynchronous copy) and look at what the
      // previous request brought - a run that lost particles stops within 32 steps, with no
      // synchronisation anywhere
      // (counted over all calls: a caller that steps one at a time does not wait for a copy per step)
      if (c->steps_run % 16 == 0) {
         if (ctx->watch_pending) SPH_TRY(hipEventSynchronize(ctx->watch_event));
         ctx->watch_pending = 0;
         if ((rc = watch_check(ctx, "sph_hip_slab_comm_run"))) return rc;
         if ((rc = watch_enqueue(ctx))) return rc;
      }
      // (the messages packed by the previous step have been sent: their counts decide about growth)
      if (c->steps_run % SLAB_GROW_EVERY == 0 && (rc = comm_grow_if_needed(ctx))) return rc;
      c->steps_run++;
      // border planes + messages on the exchange stream, transfer behind them; the interior's
      // acceleration and the integrate meanwhile on the context's stream
      if ((rc = sph_hip_slab_step_begin(ctx, c->send_left, c->send_right, c->active_records, c->stream)))
         return rc;
      if ((rc = comm_send_recv(ctx))) return rc;
      SPH_TRY(hipEventRecord(c->arrived, c->stream));
      if ((rc = sph_hip_slab_step_end(ctx))) return rc;
      SPH_TRY(hipStreamWaitEvent(st, c->arrived, 0));
      if ((rc = sph_hip_slab_unpack(ctx, c->recv_left, c->recv_right, c->active_records))) return rc;
   }
   // the word as it stands after the last step travels behind the loop - unless a copy is on its
   // way already (a caller stepping one at a time): the caller's sph_hip_synchronize, or the next
   // call of this function, reports it
   return (steps > 1 || !ctx->watch_pending) ? watch_enqueue(ctx) : SPH_HIP_OK;
}

int sph_hip_slab_comm_trim(sph_hip_context* ctx, float slack, int extra_records, int32_t* active_records)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   SlabComm* c = ctx->comm;
   if (!c || !c->comm || !(slack >= 1.0f) || extra_records < 0) {
      ctx->err = "sph_hip_slab_comm_trim: sph_hip_slab_comm_init first; slack >= 1, extra >= 0";
      return SPH_HIP_ERR_INVALID;
   }
   const RcclApi* api = rccl_api(nullptr);
   // what this rank packed last (the headers' record counts), with head room
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   SPH_TRY(hipStreamSynchronize(c->stream));
   int32_t most = 0;
   for (void* msg : {c->send_left, c->send_right}) {
      if (!msg) continue;
      int32_t n = 0;
      SPH_TRY(hipMemcpy(&n, msg, sizeof(n), hipMemcpyDeviceToHost));
      most = n > most ? n : most;
   }
   double want_d = (double)most * (double)slack + (double)extra_records;
   int32_t want = want_d > (double)c->capacity_records ? c->capacity_records : (int32_t)want_d;
   // every message of the run has one size: the largest wish of any rank
   SPH_TRY(hipMemcpy(c->trim_word, &want, sizeof(want), hipMemcpyHostToDevice));
   SPH_NCCL_TRY(api->AllReduce(c->trim_word, c->trim_word, 1, ncclInt32, ncclMax, c->comm, c->stream));
   SPH_TRY(hipStreamSynchronize(c->stream));
   SPH_TRY(hipMemcpy(&want, c->trim_word, sizeof(want), hipMemcpyDeviceToHost));
   if (want < 1) want = 1;
   c->active_records = want;
   c->bytes = sph_hip_slab_message_bytes(want);
   c->fill_pending = false;      // (both streams were drained above: a request made for the old size is void)
   if (active_records) *active_records = want;
   return SPH_HIP_OK;
}

int sph_hip_slab_comm_stats(sph_hip_context* ctx, int32_t out[4])
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   SlabComm* c = ctx->comm;
   if (!c || !out) {
      ctx->err = "sph_hip_slab_comm_stats: sph_hip_slab_comm_init first";
      return SPH_HIP_ERR_INVALID;
   }
   out[0] = c->active_records;
   out[1] = c->capacity_records;
   out[2] = c->growths;
   out[3] = (int32_t)(c->steps_run > 0x7fffffffLL ? 0x7fffffffLL : c->steps_run);
   return SPH_HIP_OK;
}

// One checked message to and from each neighbour through the calls, the stream and the group shape
// the exchange uses (before the first step: the message buffers serve as scratch).  Rank r sends
// bytes of value r + 1 and expects r from the left, r + 2 from the right.
int sph_hip_slab_comm_exchange_check(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   SlabComm* c = ctx->comm;
   if (!c || !c->comm || c->primed) {
      ctx->err = "sph_hip_slab_comm_exchange_check: after sph_hip_slab_comm_init, before the first step";
      return SPH_HIP_ERR_INVALID;
   }
   for (void* q : {c->send_left, c->send_right}) if (q) SPH_TRY(hipMemsetAsync(q, (c->rank + 1) & 0xff, c->bytes, c->stream));
   for (void* q : {c->recv_left, c->recv_right}) if (q) SPH_TRY(hipMemsetAsync(q, 0, c->bytes, c->stream));
   if ((rc = comm_send_recv(ctx))) return rc;
   SPH_TRY(hipStreamSynchronize(c->stream));
   std::string got(c->bytes, '\0');
   bool ok = true;
   for (int side = 0; side < 2; side++) {
      void* q = side == 0 ? c->recv_left : c->recv_right;
      if (!q) continue;
      SPH_TRY(hipMemcpy(&got[0], q, c->bytes, hipMemcpyDeviceToHost));
      const char want = (char)((side == 0 ? c->rank : c->rank + 2) & 0xff);
      for (size_t i = 0; i < c->bytes; i++) ok = ok && got[i] == want;
   }
   // leave the buffers as sph_hip_slab_comm_init left them
   for (void* q : {c->send_left, c->send_right, c->recv_left, c->recv_right}) if (q) SPH_TRY(hipMemsetAsync(q, 0, c->bytes, c->stream));
   SPH_TRY(hipStreamSynchronize(c->stream));
   if (!ok) {
      ctx->err = "sph_hip_slab_comm_exchange_check: a neighbour's message arrived with the wrong content";
      return SPH_HIP_ERR_DEVICE;
   }
   return SPH_HIP_OK;
}

int sph_hip_slab_comm_selftest(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   SlabComm* c = ctx->comm;
   if (!c || !c->comm) {
      ctx->err = "sph_hip_slab_comm_selftest: sph_hip_slab_comm_init first";
      return SPH_HIP_ERR_INVALID;
   }
   const RcclApi* api = rccl_api(nullptr);
   // a message to oneself through the same calls, stream and group shape the exchange uses
   const size_t n = 1 << 20;
   unsigned char *a = nullptr, *b = nullptr;
   SPH_TRY(hipMalloc((void**)&a, n));
   SPH_TRY(hipMalloc((void**)&b, n));
   std::string host(n, '\0'), back(n, '\0');
   for (size_t i = 0; i < n; i++) host[i] = (char)((i * 2654435761u) >> 13);
   SPH_TRY(hipMemcpy(a, host.data(), n, hipMemcpyHostToDevice));
   SPH_TRY(hipMemset(b, 0, n));
   SPH_NCCL_TRY(api->GroupStart());
   SPH_NCCL_TRY(api->Send(a, n, ncclChar, c->rank, c->comm, c->stream));
   SPH_NCCL_TRY(api->Recv(b, n, ncclChar, c->rank, c->comm, c->stream));
   SPH_NCCL_TRY(api->GroupEnd());
   SPH_TRY(hipStreamSynchronize(c->stream));
   SPH_TRY(hipMemcpy(&back[0], b, n, hipMemcpyDeviceToHost));
   (void)hipFree(a);
   (void)hipFree(b);
   if (back != host) {
      ctx->err = "sph_hip_slab_comm_selftest: the message came back different";
      return SPH_HIP_ERR_DEVICE;
   }
   return SPH_HIP_OK;
}

int sph_hip_particle_count(const sph_hip_context* ctx) { return ctx ? ctx->n_owned : 0; }

int sph_hip_step(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   return step_impl(ctx, true);
}

int sph_hip_run(sph_hip_context* ctx, int steps)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   for (int s = 0; s < steps; s++)
      if ((rc = step_impl(ctx, false))) return rc;
   return SPH_HIP_OK;
}

int sph_hip_voxelize(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   // FULL mode keeps the state cell-sorted, so this call moves the particles in device memory:
   // density, acceleration and neighbour counts of the last sums move with them
   return rc ? rc : launch_cell_build(ctx, nullptr, nullptr, true);
}

int sph_hip_find_neighbors(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   return rc ? rc : launch_find_neighbors(ctx);
}

int sph_hip_compute_density(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   return rc ? rc : launch_density(ctx);
}

int sph_hip_compute_acceleration(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   return rc ? rc : launch_accel(ctx);
}

int sph_hip_integrate(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if ((rc = drop_prehash(ctx))) return rc;   // the state moves on without a new hash
   return launch_integrate(ctx);
}

int sph_hip_synchronize(sph_hip_context* ctx)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   if (ctx->mode == SPH_HIP_MODE_FULL) {
      // say so here, where every host waits before it reads results: a slab whose exchange lost or
      // duplicated particles, and any context whose cell build found more entries than it has room
      // for (bit 4) or an id twice in one cell (bit 16)
      int32_t bits = 0;
      SPH_TRY(hipMemcpy(&bits, ctx->meta + META_ERRORS, sizeof(bits), hipMemcpyDeviceToHost));
      ctx->err_watch[0] = bits;
      return watch_check(ctx, "sph_hip_synchronize");
   }
   return SPH_HIP_OK;
}

int sph_hip_get_timings(sph_hip_context* ctx, float ms[6])
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (!ms) return SPH_HIP_ERR_INVALID;
   if (ctx->ev_steps == 0) {
      ctx->err = "sph_hip_get_timings: no sph_hip_step() has run since the last upload/reset";
      return SPH_HIP_ERR_INVALID;
   }
   return read_phases(ctx, ctx->ev + 7 * ((ctx->ev_steps - 1) % EV_RING), ms);
}

int sph_hip_set_timing(sph_hip_context* ctx, int level)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (level < SPH_HIP_TIMING_OFF || level > SPH_HIP_TIMING_PHASES) {
      ctx->err = "sph_hip_set_timing: level must be SPH_HIP_TIMING_OFF, _SUMS or _PHASES";
      return SPH_HIP_ERR_INVALID;
   }
   SPH_TRY(hipStreamSynchronize(ctx->stream));  // events of the old level are not read any more
   ctx->timing_level = level;
   ctx->ev_steps = 0;
   ctx->timing_seen = 0;
   return SPH_HIP_OK;
}

int sph_hip_set_timing_stride(sph_hip_context* ctx, int every)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (every < 1) {
      ctx->err = "sph_hip_set_timing_stride: every >= 1";
      return SPH_HIP_ERR_INVALID;
   }
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   ctx->timing_stride = every;
   ctx->ev_steps = 0;
   ctx->timing_seen = 0;
   return SPH_HIP_OK;
}

int sph_hip_reset_timings(sph_hip_context* ctx)
{
   if (!ctx) return SPH_HIP_ERR_INVALID;
   ctx->ev_steps = 0;
   return SPH_HIP_OK;
}

int sph_hip_get_phase_totals(sph_hip_context* ctx, double ms[6], int32_t* steps)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (!ms || !steps) return SPH_HIP_ERR_INVALID;
   const long long have = ctx->ev_steps < EV_RING ? ctx->ev_steps : EV_RING;
   for (int k = 0; k < 6; k++) ms[k] = 0.0;
   for (long long s = ctx->ev_steps - have; s < ctx->ev_steps; s++) {
      float one[6];
      if ((rc = read_phases(ctx, ctx->ev + 7 * (s % EV_RING), one))) return rc;
      for (int k = 0; k < 6; k++) ms[k] += (double)one[k];
   }
   *steps = (int32_t)have;
   return SPH_HIP_OK;
}

int sph_hip_get_energy(sph_hip_context* ctx, float* kinetic, float* potential)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   double e[2] = {0.0, 0.0};
   if (ctx->energy_blocks > 0) {
      // per-workgroup partial sums of the last integrate -> totals, fixed order
      hipLaunchKernelGGL(k_energy_total, dim3(1), dim3(RED_THREADS), 0, ctx->stream,
                         ctx->epart + 2, ctx->energy_blocks, ctx->epart);
      SPH_TRY(hipGetLastError());
   }
   SPH_TRY(hipMemcpyAsync(e, ctx->epart, sizeof(e), hipMemcpyDeviceToHost, ctx->stream));
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   if (kinetic) *kinetic = (float)e[0];
   if (potential) *potential = (float)e[1];
   return SPH_HIP_OK;
}

int sph_hip_get_tile_stats(sph_hip_context* ctx, int32_t out[20])
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (!out || !ctx->tile_feedback) return SPH_HIP_ERR_INVALID;
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   for (int i = 0; i < 16; i++) out[i] = ctx->tile_feedback[i];
   out[16] = ctx->caps.cap_density;
   out[17] = ctx->caps.cap_accel;
   out[18] = ctx->caps.wide;
   out[19] = ctx->list_cap;
   return SPH_HIP_OK;
}

int sph_hip_get_neighbor_stats(sph_hip_context* ctx, int32_t* avg, int32_t* mx, int32_t* mn)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if ((rc = owned_count(ctx, nullptr))) return rc;
   const int n = ctx->n_owned;
   if (n == 0) return SPH_HIP_ERR_INVALID;
   const int32_t init[4] = {0, 0, -1, 34};
   SPH_TRY(hipMemcpyAsync(ctx->stats, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
   int blocks = div_up(n, RED_THREADS);
   if (blocks > 1024) blocks = 1024;
   hipLaunchKernelGGL(k_neighbor_stats, dim3(blocks), dim3(RED_THREADS), 0, ctx->stream,
                      ctx->ncount, ctx->meta, ctx->stats);
   SPH_TRY(hipGetLastError());
   int32_t out[4];
   SPH_TRY(hipMemcpyAsync(out, ctx->stats, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   const long long sum = ((long long)(uint32_t)out[1] << 32) | (uint32_t)out[0];
   if (avg) *avg = (int32_t)(sum / n);
   if (mx) *mx = out[2];
   if (mn) *mn = out[3];
   return SPH_HIP_OK;
}

int sph_hip_download_voxels(sph_hip_context* ctx, int32_t* coords_xyz, int32_t* ids)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_REF) {
      ctx->err = "sph_hip_download_voxels: REF-mode contexts only";
      return SPH_HIP_ERR_INVALID;
   }
   const int n = ctx->n_owned;
   if (coords_xyz)
      SPH_TRY(hipMemcpyAsync(coords_xyz, ctx->vox, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost,
                             ctx->stream));
   if (ids)
      SPH_TRY(hipMemcpyAsync(ids, ctx->key, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   return SPH_HIP_OK;
}

int sph_hip_download_grid_counts(sph_hip_context* ctx, int32_t* counts)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (!counts) return SPH_HIP_ERR_INVALID;
   const int nc = ctx->grid.ncells;
   // counts are kept as the exclusive scan; difference them on the host
   uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)nc + 1));
   if (!tmp) return SPH_HIP_ERR_INVALID;
   hipError_t e = hipMemcpyAsync(tmp, ctx->cell_start, sizeof(uint32_t) * ((size_t)nc + 1),
                                 hipMemcpyDeviceToHost, ctx->stream);
   if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
   if (e != hipSuccess) {
      free(tmp);
      ctx->err = std::string("sph_hip_download_grid_counts: ") + hipGetErrorString(e);
      return SPH_HIP_ERR_DEVICE;
   }
   for (int c = 0; c < nc; c++) counts[c] = (int32_t)(tmp[c + 1] - tmp[c]);
   free(tmp);
   return SPH_HIP_OK;
}

int sph_hip_download_neighbor_lists(sph_hip_context* ctx, uint32_t* neighbors, float* distances)
{
   int rc = check_ctx(ctx);
   if (rc) return rc;
   if (ctx->mode != SPH_HIP_MODE_REF) {
      ctx->err = "sph_hip_download_neighbor_lists: REF-mode contexts only (FULL mode stores no lists)";
      return SPH_HIP_ERR_INVALID;
   }
   const size_t m = (size_t)ctx->n_owned * ctx->prm.examine_count;
   if (neighbors)
      SPH_TRY(hipMemcpyAsync(neighbors, ctx->nb, sizeof(uint32_t) * m, hipMemcpyDeviceToHost,
                             ctx->stream));
   if (distances)
      SPH_TRY(hipMemcpyAsync(distances, ctx->nd, sizeof(float) * m, hipMemcpyDeviceToHost,
                             ctx->stream));
   SPH_TRY(hipStreamSynchronize(ctx->stream));
   return SPH_HIP_OK;
}

void* sph_hip_stream(sph_hip_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

} // extern "C"
