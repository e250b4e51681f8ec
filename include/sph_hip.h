/* sph_hip.h — C ABI of the MI355X (gfx950) SPH step.
 *
 * This is the drop-in boundary for the hot path of
 * DanielaCourel/smoothed_particle_hydrodynamics: everything SPH::step()
 * (reference src/sph.cpp:190-304) does between "particles in" and "particles out".
 * The reference has no FFI layer; its seam is the C++ class `SPH` (reference
 * src/sph.h:15-216).  Each entry point below names the member(s) of that class it
 * replaces, so a maintainer can keep sph.h unchanged and forward the bodies in sph.cpp
 * to this library.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary;
 *   - every call returns SPH_HIP_OK (0) or a negative sph_hip_status; the message for the
 *     last failure on a context is available from sph_hip_last_error();
 *   - host arrays use the reference's layouts: positions / velocities / accelerations
 *     interleaved xyz (`mPosition[3*i+c]`, reference src/particle.h:13-18), one float or
 *     int per particle otherwise, all indexed by the particle's persistent index;
 *   - device memory, streams and events are owned by the context.
 */
#ifndef SPH_HIP_H
#define SPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_HIP_ABI_VERSION 6
/* The ABI version the loaded library was built with (compare with SPH_HIP_ABI_VERSION of the
 * header the host was compiled against before calling anything else).  A library built with
 * profiling hooks that cut pieces out of the kernels (diagnostic builds: results are garbage by
 * design) ORs SPH_HIP_ABI_DIAGNOSTIC into the value, so that no host takes it for the product. */
#define SPH_HIP_ABI_DIAGNOSTIC 0x4000
int sph_hip_abi_version(void);

typedef enum sph_hip_status {
   SPH_HIP_OK = 0,
   SPH_HIP_ERR_INVALID = -1,   /* bad argument / bad state */
   SPH_HIP_ERR_DEVICE = -2,    /* a HIP runtime call failed */
   SPH_HIP_ERR_CAPACITY = -3,  /* more particles than the context was created for */
   SPH_HIP_ERR_NO_DEVICE = -4, /* no usable gfx950 device */
   SPH_HIP_ERR_EXCHANGE = -5   /* a slab exchange lost particles (sph_hip_slab_status bits) */
} sph_hip_status;

/* Neighbour semantics of a context.
 *   REF  — the shipped search: octant of 2h-voxels, LCG-sampled chunks of 8 of which 4 are
 *          tested, at most 28 stored neighbours (reference src/sph.cpp:484-692), then the
 *          list-driven sums.  Integer outputs are identical to the reference's.
 *   FULL — every neighbour inside the interaction radius, found on a grid of cell edge
 *          >= h (27 cells), visited in ascending (cell id, particle index); per-pair
 *          arithmetic is the reference's (src/sph.cpp:737-761, 825-884).
 *   FULL_FAST — FULL with tolerance-mode pair arithmetic (SPH_HIP_ARITH_FAST below): the same
 *          neighbour sets (the exact fp32 test (dx*dx + dy*dy) + dz*dz < h^2 of
 *          src/sph.cpp:641,653), the same canonical order and the viscous rescale inside the
 *          neighbour loop (src/sph.cpp:880-882) - but the per-pair arithmetic evaluated the way
 *          the reference's own shipped build may evaluate it (reference CMakeLists.txt:21:
 *          -O3 -ffast-math -funsafe-math-optimizations -mfma), in the ACCELERATION sum only: an
 *          fp32 reciprocal in place of the fp64 quotient of src/sph.cpp:854-856, fused
 *          accumulation, and a viscous sum that leaves out the neighbours whose weight - the
 *          rescale of src/sph.cpp:880-882 applied once per later neighbour - is below 1e-20;
 *          both sums on the reference's stored distance.  Neighbour counts AND densities are
 *          identical to FULL; the acceleration of EVERY particle agrees with the IEEE evaluation of
 *          src/sph.cpp to 1e-4 relative, |a - a_ref| <= 1e-4 * max(|a|, |a_ref|) (vector norm) -
 *          asserted as written, without escape clauses, on every BASELINE configuration and every
 *          committed scene (tests/test_gpu_full_fast.py, test_gpu_full_size.py, test_gpu_c4_c5.py;
 *          measured: 1.4e-5 at worst on the 4M column at rest, 6e-6 moving; bench.py re-checks the
 *          state it timed: `parity` in its JSON line).  ONE clause exists, for adversarial inputs only
 *          (tests/test_gpu_random_scenes.py, which prints every particle that takes it): where a
 *          particle's ~30 pair terms cancel to less than a hundredth of their magnitude sum T, any
 *          evaluation that is not the reference's bit for bit - its own -ffast-math build included -
 *          differs by rounding errors of the terms, and such a particle is held to 1e-6 * T instead
 *          (at most 0.5 % of a scene's particles).  Deterministic, the same for any route and slab
 *          count, not bit-reproducible against the CPU. */
typedef enum sph_hip_mode {
   SPH_HIP_MODE_REF = 0,
   SPH_HIP_MODE_FULL = 1,
   SPH_HIP_MODE_FULL_FAST = 2
} sph_hip_mode;

/* The constants the hot path reads — the protected members SPH::SPH() initialises
 * (reference src/sph.h:149-210, src/sph.cpp:46-98).  Field order is ABI. */
typedef struct sph_hip_params {
   int32_t cells_x, cells_y, cells_z; /* mGridCellsX/Y/Z   (voxel grid, edge 2h)        */
   float cell_size;                   /* mCellSize                                      */
   float max_x, max_y, max_z;         /* mMaxX/Y/Z                                      */
   float h;                           /* mH                                             */
   float h2;                          /* mH2                                            */
   float hscaled;                     /* mHScaled                                       */
   float hscaled2;                    /* mHScaled2                                      */
   float hscaled6;                    /* mHScaled6                                      */
   float hscaled9;                    /* mHScaled9                                      */
   float htimes2;                     /* mHTimes2                                       */
   float htimes2inv;                  /* mHTimes2Inv                                    */
   float sim_scale;                   /* mSimulationScale                               */
   float sim_scale_inv;               /* mSimulationScaleInverse                        */
   float kernel1, kernel2, kernel3;   /* mKernel1Scaled, mKernel2Scaled, mKernel3Scaled */
   float rho0;                        /* mRho0                                          */
   float stiffness;                   /* mStiffness        (SPH::setStiffness)          */
   float viscosity;                   /* mViscosityScalar  (SPH::setViscosityScalar)    */
   float time_step;                   /* mTimeStep         (SPH::setTimeStep)           */
   float damping;                     /* mDamping          (SPH::setDamping)            */
   float cfl_limit, cfl_limit2;       /* mCflLimit, mCflLimit2 (SPH::setCflLimit)       */
   float gravity[3];                  /* mGravity          (SPH::setGravity)            */
   float grav_const;                  /* mGravConstant                                  */
   float central_mass;                /* mCentralMass                                   */
   float central_pos[3];              /* mCentralPos                                    */
   float softening;                   /* mSoftening                                     */
   int32_t examine_count;             /* mExamineCount (32)                             */
   /* FULL-mode grid (no reference counterpart): cell edge >= h */
   int32_t full_cells_x, full_cells_y, full_cells_z;
   float full_cell_inv;
   /* Dam-break physics the reference defines but never wires; both 0 =
    * shipped behaviour.
    *   apply_gravity: mGravity is added wherever the reference adds its point-mass gravity
    *                  (computeAcceleration src/sph.cpp:913-915 and integrate :987-989).
    *   apply_walls:   integrate passes (old position, new velocity, dt, new position) through
    *                  SPH::handleBoundaryConditions / applyBoundary (src/sph.cpp:1025-1148):
    *                  per-axis reflection at 0 / mMax*, remaining path scaled by mDamping. */
   int32_t apply_gravity;
   int32_t apply_walls;
} sph_hip_params;

typedef struct sph_hip_context sph_hip_context;

/* ---- construction -------------------------------------------------------------------- */

/* Constants for smoothing length h and a voxel grid of the given shape, derived exactly as
 * SPH::SPH() derives them (reference src/sph.cpp:46-98: double pow() narrowed to float,
 * float kernel normalisations).  h = 0.1f, cells = 32^3 gives the reference's defaults. */
int sph_hip_params_default(sph_hip_params* out, float h, int cells_x, int cells_y, int cells_z);

/* Replaces the allocations in SPH::SPH() (reference src/sph.cpp:100-113): device storage for
 * up to `capacity` particles on HIP device `device`. */
int sph_hip_create(sph_hip_context** out, const sph_hip_params* params, int capacity, int mode,
                   int device);
void sph_hip_destroy(sph_hip_context* ctx);

/* Last error text for ctx (or for the failed create when ctx is NULL). Never NULL. */
const char* sph_hip_last_error(const sph_hip_context* ctx);

/* Replaces the six GUI setters + the constructor constants (reference src/sph.cpp:1219-1289).
 * Takes effect at the start of the next phase call; grid shape and h may not change. */
int sph_hip_set_params(sph_hip_context* ctx, const sph_hip_params* params);
int sph_hip_get_params(const sph_hip_context* ctx, sph_hip_params* out);

/* Pair arithmetic of a FULL-mode context (sph_hip_create with SPH_HIP_MODE_FULL_FAST starts in
 * SPH_HIP_ARITH_FAST; slab contexts start exact).  May be changed between steps (synchronises);
 * takes effect with the next cell build.  No counterpart in the reference's API: its counterpart
 * is the compiler flags the reference is built with (CMakeLists.txt:21). */
#define SPH_HIP_ARITH_EXACT 0
#define SPH_HIP_ARITH_FAST 1
int sph_hip_set_arithmetic(sph_hip_context* ctx, int arithmetic);
int sph_hip_get_arithmetic(const sph_hip_context* ctx);

/* ---- particle state ------------------------------------------------------------------- */

/* Host -> device.  Replaces the fill of Particle::mPosition/mVelocity/mMass done by
 * initParticlePolitionsSphere() and the constructor (reference src/sph.cpp:105-108,
 * 361-425).  pos/vel: 3*n floats interleaved; mass: n floats.  Sets the live count. */
int sph_hip_upload(sph_hip_context* ctx, int n, const float* pos, const float* vel,
                   const float* mass);

/* Device -> host mirror of `Particle` (reference src/particle.h:13-18), any pointer may be
 * NULL: mPosition, mVelocity, mDensity, mAcceleration, mNeighborCount, indexed by the
 * particle's persistent index.  This is what SPH::getParticles() consumers read
 * (reference src/visualization.cpp:144-158). */
int sph_hip_download(sph_hip_context* ctx, float* pos, float* vel, float* density, float* acc,
                     int32_t* neighbor_count);

/* The same mirror without stopping the solver thread - what the reference's GUI gets when it
 * reads SPH::getParticles() / getGrid() at 60 Hz from another thread, without locks
 * (reference src/visualization.cpp:144-158, 178-193).
 *   sph_hip_download_async  enqueues, behind the work queued so far, a snapshot of the
 *       per-particle arrays (any pointer may be NULL) and of the per-voxel occupancy on the
 *       REFERENCE voxel grid (cells_x*cells_y*cells_z ints, edge mCellSize - in FULL mode too),
 *       and their copy to the host on a separate low-priority stream; returns at once.  The
 *       later steps' kernels do not wait for the copy.  *started = 0 (nothing done) while the
 *       previous request is still on its way: a mirror is a picture, not a queue.
 *   sph_hip_download_done   1 = the last request has arrived in the host arrays, 0 = not yet
 *       (wait != 0: blocks until it has).  Negative = error.
 *   sph_hip_host_register   page-locks host memory (e.g. the storage of Particle's vectors) so
 *       that the copy really is asynchronous; pageable memory works, slower and less overlapped.
 * Double-buffer on the host: request into the back set, swap when done (integration/). */
int sph_hip_download_async(sph_hip_context* ctx, float* pos, float* vel, float* density, float* acc,
                           int32_t* neighbor_count, int32_t* voxel_counts, int* started);
int sph_hip_download_done(sph_hip_context* ctx, int wait);
int sph_hip_host_register(void* ptr, size_t bytes);
int sph_hip_host_unregister(void* ptr);

int sph_hip_particle_count(const sph_hip_context* ctx);

/* ---- the step -------------------------------------------------------------------------- */

/* SPH::step() (reference src/sph.cpp:190-304): the five phases below, in order. */
int sph_hip_step(sph_hip_context* ctx);
/* `steps` back-to-back steps with no host synchronisation in between. */
int sph_hip_run(sph_hip_context* ctx, int steps);

/* SPH::voxelizeParticles() + clearGrid() (reference src/sph.cpp:429-481): cell ids, per-cell
 * counts, cell-sorted order (ascending particle index inside a cell). */
int sph_hip_voxelize(sph_hip_context* ctx);
/* The findNeighbors() loop (reference src/sph.cpp:216-231, 484-692).  REF: builds
 * mNeighbors / mNeighborDistancesScaled / mNeighborCount.  FULL: no stored lists — the
 * neighbour walk is fused into the two sums; this call is a no-op kept for phase timing. */
int sph_hip_find_neighbors(sph_hip_context* ctx);
/* The computeDensity() loop (reference src/sph.cpp:242-249, 721-766). */
int sph_hip_compute_density(sph_hip_context* ctx);
/* The computeAcceleration() loop (reference src/sph.cpp:270-277, 778-934). */
int sph_hip_compute_acceleration(sph_hip_context* ctx);
/* The integrate() loop (reference src/sph.cpp:285-289, 937-1022) incl. KE/PE totals. */
int sph_hip_integrate(sph_hip_context* ctx);

/* Wait for all queued work of ctx. */
int sph_hip_synchronize(sph_hip_context* ctx);

/* ---- diagnostics ------------------------------------------------------------------------ */

/* Milliseconds (fractional, unlike the reference's truncated ints) spent in the six phases of
 * the last sph_hip_step(): voxelize, findNeighbors, density, pressure(=0), acceleration,
 * integrate — the arguments of SPH::updateElapsed (reference src/sph.cpp:292-299). */
int sph_hip_get_timings(sph_hip_context* ctx, float ms[6]);
/* Sums of the same six phase times over the sph_hip_step() calls since the last
 * sph_hip_reset_timings() (at most the most recent 128 steps are kept); *steps = how many
 * steps the sums cover.  Measured with HIP events on the context's own stream. */
int sph_hip_get_phase_totals(sph_hip_context* ctx, double ms[6], int32_t* steps);
int sph_hip_reset_timings(sph_hip_context* ctx);
/* What sph_hip_step() times (an event record is a barrier packet: ~10 us each on the stream).
 * SPH_HIP_TIMING_PHASES (default): every phase boundary, as SPH::updateElapsed wants.
 * SPH_HIP_TIMING_SUMS: only the density + acceleration pair, as one interval - reported in
 * slot 2 (density) of the two calls above, the other slots 0.  SPH_HIP_TIMING_OFF: nothing
 * (sph_hip_get_timings fails, the totals cover 0 steps).  Resets the collected timings. */
#define SPH_HIP_TIMING_OFF 0
#define SPH_HIP_TIMING_SUMS 1
#define SPH_HIP_TIMING_PHASES 2
int sph_hip_set_timing(sph_hip_context* ctx, int level);
/* Record the events of the chosen level on every `every`-th timed step only (default 1 = every
 * step); the totals and their step count then cover the sampled steps.  For measurements that
 * must not weigh on what they measure (bench.py).  Resets the collected timings. */
int sph_hip_set_timing_stride(sph_hip_context* ctx, int every);

/* FULL mode, tiled kernels: statistics of the LDS tiles of the last step (synchronises).
 * out[0..11]: workgroups whose tile exceeds capacity level i (the levels are the largest tiles
 * that allow a given number of workgroups per CU), out[12]: workgroups, out[13]: largest tile
 * (entries), out[14], out[15]: workgroups computed untiled in the density / acceleration pass,
 * out[16], out[17]: tile capacities the two passes were launched with, out[18]: 1 if the list
 * entries were in their wide format (a capacity above 4064), out[19]: neighbours per particle the
 * lists currently hold (starts at 254; enlarged to 1022 - or to 510, if that is all the device has
 * room for - when a step reports particles with more: those are computed without a list, slower,
 * same results). */
int sph_hip_get_tile_stats(sph_hip_context* ctx, int32_t out[20]);

/* mKineticEnergyTotal / mPotentialEnergyTotal of the last integrate
 * (reference src/sph.cpp:1001-1013).  Summed in double in a fixed tree order; the
 * reference's serial fp32 sum is order-dependent, so compare with a tolerance. */
int sph_hip_get_energy(sph_hip_context* ctx, float* kinetic, float* potential);

/* The three numbers the reference appends to out/neighbors.txt each step
 * (reference src/sph.cpp:204-232): sum/N (integer division), max, min(<=34). */
int sph_hip_get_neighbor_stats(sph_hip_context* ctx, int32_t* avg, int32_t* max, int32_t* min);

/* mVoxelCoords / mVoxelIds (reference src/sph.cpp:466-472); coords: 3*n ints. */
int sph_hip_download_voxels(sph_hip_context* ctx, int32_t* coords_xyz, int32_t* ids);
/* Per-voxel occupancy, what callers get from SPH::getGrid()[i].count()
 * (reference src/visualization.cpp:178-193). `counts` has cells_x*cells_y*cells_z entries
 * (REF) or full_cells_x*full_cells_y*full_cells_z (FULL). */
int sph_hip_download_grid_counts(sph_hip_context* ctx, int32_t* counts);
/* REF mode only: mNeighbors / mNeighborDistancesScaled, n*examine_count entries each
 * (reference src/sph.cpp:112-113). */
int sph_hip_download_neighbor_lists(sph_hip_context* ctx, uint32_t* neighbors, float* distances);

/* ---- multi-GPU: 1-D slab decomposition of the FULL-mode cell grid ------------------------- *
 *
 * No counterpart in the reference (one process, one thread).  One context per GPU owns the
 * global z-planes [plane_lo, plane_hi) of the FULL grid and additionally holds
 * SPH_HIP_SLAB_HALO (= 2) ghost planes on each side: with two planes the densities of the
 * ghosts next to the slab are recomputed locally from complete neighbourhoods, in the same
 * canonical order (cell id, particle id) as on their owner, so ONE exchange per step is enough
 * and results do not depend on the number of slabs.
 *
 * Per step, on every rank:
 *     sph_hip_slab_pack   -> two device messages (left / right neighbour)
 *     <transport>            RCCL send/recv over xGMI (torch.distributed), or a device copy
 *     sph_hip_slab_unpack <- the two messages received
 *     sph_hip_step
 * A message is sph_hip_slab_message_bytes(capacity) bytes of DEVICE memory owned by the
 * caller: 8 int32 header words (word 0 = record count) + 32-byte records
 * {x,y,z,m,vx,vy,vz,id}.  Every step a slab re-sends each owned particle that lies within the
 * halo width of (or beyond) a neighbour's border: the receiver treats records inside its own
 * planes as migrants (now owned) and the rest as ghosts; ghosts are dropped and re-sent every
 * step.  All counts stay on the device — none of these calls synchronises with the host.
 * sph_hip_create() is the special case plane_lo = 0, plane_hi = all planes, no neighbours.
 *
 * Exchange overlapped with the interior's force computation (after one pack/transport/unpack
 * round has delivered the first ghosts), per step:
 *     sph_hip_slab_step_begin   cell build, density, acceleration of the owned planes next to a
 *                               neighbour (halo + 1 planes), and the two messages: those
 *                               particles are integrated on the fly, the state is not touched
 *     <transport>               on the exchange stream handed to step_begin
 *     sph_hip_slab_step_end     acceleration of all other workgroups, integrate - concurrent
 *                               with the transport
 *     sph_hip_slab_unpack       after the context's stream has waited for the transport
 * Same results as the serial sequence.  Last step's ghosts and departed particles are recognised
 * by the next cell build from their position in the sorted order; error bit 8 reports an
 * interior particle that crossed more than one cell plane in a step and so missed its message. */
#define SPH_HIP_SLAB_HALO 2

int sph_hip_create_slab(sph_hip_context** out, const sph_hip_params* params, int capacity,
                        int device, int plane_lo, int plane_hi);
/* Owned particles of this slab with their GLOBAL persistent ids (the canonical order inside
 * a cell is by id).  all_masses_equal: non-zero iff every particle of the WHOLE system has
 * the same mass (enables the same fast path as sph_hip_upload detects by itself). */
int sph_hip_slab_upload(sph_hip_context* ctx, int n, const float* pos, const float* vel,
                        const float* mass, const uint32_t* ids, int all_masses_equal);
/* Owned particles in cell-sorted order: *rows receives their number (synchronises). */
int sph_hip_slab_download(sph_hip_context* ctx, int max_rows, int32_t* rows, uint32_t* ids,
                          float* pos, float* vel, float* density, float* acc,
                          int32_t* neighbor_count);
/* Masses of the owned particles in the row order of sph_hip_slab_download (what a host needs,
 * with that call's arrays, to move particles to another slab: slab.py rebalance()). */
int sph_hip_slab_download_mass(sph_hip_context* ctx, int max_rows, int32_t* rows, float* mass);
/* Device-to-device re-partitioning (no counterpart in the reference): the owned particles as the
 * 32-byte records of a halo message, {x,y,z,m | vx,vy,vz,id}, in cell-sorted order, written to
 * DEVICE memory of the caller (*rows = their number; synchronises) - and a slab's owned particles
 * from n such records in device memory (what sph_hip_slab_upload does from host arrays).  The rows
 * that change owner when the cut planes move (slab.py rebalance()) can then travel over RCCL
 * without touching the host. */
int sph_hip_slab_export_records(sph_hip_context* ctx, void* device_records, int max_records,
                                int32_t* rows);
int sph_hip_slab_upload_records(sph_hip_context* ctx, const void* device_records, int n,
                                int all_masses_equal);
size_t sph_hip_slab_message_bytes(int capacity_records);
/* NULL for a side without neighbour.  Call after sph_hip_step()/upload, before the transport. */
int sph_hip_slab_pack(sph_hip_context* ctx, void* left_device, void* right_device,
                      int capacity_records);
int sph_hip_slab_unpack(sph_hip_context* ctx, const void* left_device, const void* right_device,
                        int capacity_records);
/* See above.  Message buffers: exactly one per existing neighbour (NULL otherwise).
 * exchange_stream: the HIP stream the transport will be issued on (NULL = the context's stream,
 * no overlap).  The border planes' acceleration and the packing are enqueued on it, behind the
 * density pass, so the messages are complete in that stream's order and the work runs next to
 * the interior's acceleration; sph_hip_slab_step_end makes the integrate wait for it. */
int sph_hip_slab_step_begin(sph_hip_context* ctx, void* left_device, void* right_device,
                            int capacity_records, void* exchange_stream);
int sph_hip_slab_step_end(sph_hip_context* ctx);
/* ---- native RCCL exchange (no Python, no torch) ----
 * librccl is opened with dlopen on first use.  One rank (any) calls sph_hip_rccl_unique_id and
 * hands the 128 bytes to every rank (MPI, a file, a socket, torch.distributed ...); every rank
 * then calls sph_hip_slab_comm_init on its slab context - ranks are ordered along z, rank r's
 * neighbours are r - 1 and r + 1 - which creates the communicator, a high-priority exchange
 * stream and the four message buffers (capacity_records must be the same on all ranks).
 * sph_hip_slab_comm_run(steps) is then the whole loop: a first serial exchange, and per step
 * sph_hip_slab_step_begin -> ncclSend/ncclRecv of both directions in one group on the exchange
 * stream -> sph_hip_slab_step_end -> sph_hip_slab_unpack.  Asynchronous like sph_hip_run.
 * sph_hip_slab_comm_selftest sends a message to itself through the same calls (synchronises). */
int sph_hip_rccl_unique_id(void* id_out, int id_bytes);
int sph_hip_slab_comm_init(sph_hip_context* ctx, const void* id, int id_bytes, int rank, int nranks,
                           int capacity_records);
int sph_hip_slab_comm_run(sph_hip_context* ctx, int steps);
/* Only the used part of a message needs to cross the link.  Collective over the communicator
 * (synchronises): every rank looks at the record counts of the messages it packed last, the
 * largest count * slack + extra_records (at most the capacity given to sph_hip_slab_comm_init)
 * becomes the size every message is packed for, sent and received with from now on
 * (*active_records).  sph_hip_slab_comm_run puts the messages back to the allocated size before
 * they overflow (sph_hip_slab_comm_stats); a message that outgrows even that - or grows by more
 * than a quarter within 32 steps - raises error bit 2, which stops the run.  Call after a few
 * steps, and now and then in a long run. */
int sph_hip_slab_comm_trim(sph_hip_context* ctx, float slack, int extra_records,
                           int32_t* active_records);
int sph_hip_slab_comm_selftest(sph_hip_context* ctx);
/* One checked message to and from each neighbour through the same calls (after
 * sph_hip_slab_comm_init, before the first step; collective over the neighbours; synchronises): a
 * pre-flight of the device-to-device path that a launcher can run in a helper process with a time
 * limit before it commits a run to it (bench.py does). */
int sph_hip_slab_comm_exchange_check(sph_hip_context* ctx);
/* out[0]: records the messages are currently packed for and transferred with, out[1]: records the
 * buffers hold, out[2]: how often sph_hip_slab_comm_run put trimmed messages back to the allocated
 * size (they grow BEFORE they overflow: every 16 steps the ranks reduce the record counts of the
 * messages they packed last - ncclAllReduce(max) behind the exchange, result read 16 steps later,
 * at the same step on every rank - and all of them switch together when any message was more than
 * 4/5 full), out[3]: steps enqueued by sph_hip_slab_comm_run so far. */
int sph_hip_slab_comm_stats(sph_hip_context* ctx, int32_t out[4]);
/* Diagnostics (synchronises): live entries, owned particles, error bits (1: a received entry
 * lies outside the slab and its halo, 2: a message overflowed, 4: context capacity exceeded - by
 * received records or by the cell counts of a build: the build clamps its ranges to the capacity
 * and drops what does not fit instead of writing past its arrays, 8: a particle missed the early
 * exchange, 16: one cell holds a persistent id twice - a record delivered twice, a particle owned
 * by two slabs). */
int sph_hip_slab_status(sph_hip_context* ctx, int32_t* live, int32_t* owned, int32_t* errors);
/* The same error bits without draining the stream: reports what the copy requested by the
 * PREVIOUS call brought (*errors, may be NULL; waits for that one copy if the host has run more
 * than one polling interval ahead of the device) and requests the next copy of the device's
 * error word into pinned host memory.  Returns SPH_HIP_ERR_EXCHANGE (message in
 * sph_hip_last_error) once a non-zero word has arrived, so a loop that calls this every K steps
 * stops within 2 K steps of the exchange going wrong instead of running on with missing particles.
 * sph_hip_slab_comm_run polls every 16 steps itself; sph_hip_synchronize and
 * sph_hip_slab_download check after waiting and return the same status. */
int sph_hip_slab_poll_errors(sph_hip_context* ctx, int32_t* errors);

/* ---- self tests -------------------------------------------------------------------------- */

/* The pair loops take square roots with a shorter instruction sequence than the compiler's
 * sqrtf (csrc/sph_device.h: sqrt_rn).  This compares the two for EVERY non-negative finite
 * float on `device` (2^31 inputs, well under a second): *mismatches = how many differ,
 * *first_bad = bit pattern of the smallest one that does (0xffffffff if none). */
int sph_hip_selftest_sqrt(int device, uint64_t* mismatches, uint32_t* first_bad);

/* ---- streams ----------------------------------------------------------------------------- */

/* Run all of ctx's work on the caller's HIP stream (e.g. the stream torch.distributed orders
 * its RCCL calls against); NULL returns to the context's own stream. */
int sph_hip_set_stream(sph_hip_context* ctx, void* hip_stream);
/* The HIP stream all of ctx's kernels currently run on. */
void* sph_hip_stream(sph_hip_context* ctx);

#ifdef __cplusplus
}
#endif

#endif /* SPH_HIP_H */
