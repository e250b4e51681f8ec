// Drop-in body for the reference's solver class over the HIP C ABI.
//
// This file REPLACES the reference's src/sph.cpp in its CMake source list; src/sph.h,
// src/particle.h, src/sphconfig.*, src/visualization.*, src/widget.*, src/main.cpp stay
// untouched (moc still runs over the unchanged sph.h — no HIP type appears in it).  Link with
// -lsph_hip.  It is compile-checked against the reference's header by
// tests/test_integration_build.py wherever /root/reference exists.
//
// Every member keeps its meaning (reference src/sph.h:20-139):
//   * the constructor derives the constants through sph_hip_params_default (same arithmetic as
//     src/sph.cpp:46-98), builds the default scene and uploads it;
//   * step() runs the five phases on the GPU and emits the same two signals.  The host mirrors
//     the GUI reads at frame rate without locks (Particle arrays through getParticles(),
//     per-voxel QList sizes through getGrid(); reference src/visualization.cpp:144-158, 178-193)
//     are TRIPLE-BUFFERED (front / retired / back: a frame that still holds the previous pointer
//     never reads a buffer being filled): step() asks the library for an asynchronous snapshot
//     into the back `Particle` (page-locked, copied on a separate stream: sph_hip_download_async)
//     and returns without waiting for it; a later step() finds it complete and swaps it in.  The solver never
//     stalls on PCIe, the GUI always sees one complete state at most a frame or two old - in both
//     neighbour modes (the occupancy comes on the reference's voxel grid in FULL mode too).
//     SPH_DROPIN_SYNC_MIRROR=1 (or sph_dropin_sync_mirror() before reading) gives the blocking
//     behaviour: the mirror then shows the step just taken;
//   * the six setters are called from the GUI thread (reference src/sphconfig.cpp:89-94) while
//     the solver thread steps: they write the parameter block under a mutex, step() takes a copy
//     under the same mutex and hands it to the library, which applies it at that step.
// The per-particle protected methods (findNeighbors(i, ...), computeDensity(i, ...), ...) have
// no per-particle GPU meaning; they stay as no-ops because nothing outside step() calls them.
#include "sph.h"

#include "particle.h"

#include <QElapsedTimer>
#include <math.h>
#include <stdlib.h>
#include <sys/stat.h>

#include <fstream>
#include <iostream>
#include <mutex>
#include <vector>

#include "sph_hip.h"
#include <string.h>

#ifndef M
#define M 32
#endif

namespace {
sph_hip_context* g_ctx = nullptr; // one solver per process, like the reference's single SPH object
sph_hip_params g_prm;             // written by the GUI's setters, read by step(): under g_prm_mutex
std::mutex g_prm_mutex;
int g_mode = SPH_HIP_MODE_REF;    // SPH_HIP_MODE_FULL for complete neighbourhoods

// The two host mirrors: `front` is what getParticles() hands out, `back` is being filled by the
// library (or waits to be).  Both are the reference's own `Particle` (std::vector storage,
// page-locked once so that the copy into it is a DMA the solver thread does not wait for).
Particle* g_back = nullptr;      // being filled by the copy stream
Particle* g_retired = nullptr;   // the front of one swap ago: a GUI frame may still be reading it
std::vector<int32_t> g_counts_front, g_counts_back;   // per-voxel occupancy on the reference grid
bool g_copy_in_flight = false;
bool g_sync_mirror = false;       // SPH_DROPIN_SYNC_MIRROR: wait for every step's snapshot
long long g_steps_taken = 0, g_step_mirrored = -1, g_step_in_flight = -1;
long long g_returned_before_copy = 0;   // step() calls that returned while their snapshot was still travelling
// where the solver object keeps what the mirror swap touches (protected members; noted by the
// constructor so that the free function below can publish a mirror too)
Particle** g_front_slot = nullptr;
QList<uint32_t>* g_grid = nullptr;
int g_cells = 0;

void check(int rc, const char* what)
{
   if (rc != SPH_HIP_OK) {
      std::cerr << what << " failed: " << sph_hip_last_error(g_ctx) << std::endl;
      abort(); // the reference has no error channel; a dead GPU path must not go unnoticed
   }
}
} // namespace

SPH::SPH()
 : mParticleCount(0), mGridCellCount(0), mRho0(0.0f), mStopped(false), mPaused(false),
   mKineticEnergyTotal(0.0f), mPotentialEnergyTotal(0.0f),
   mAngularMomentumTotal(vec3(0.0f, 0.0f, 0.0f))
{
   if (const char* full = getenv("SPH_HIP_FULL"))   // "fast": FULL with the tolerance-mode pair arithmetic
      g_mode = strcmp(full, "fast") == 0 ? SPH_HIP_MODE_FULL_FAST : SPH_HIP_MODE_FULL;
   check(sph_hip_params_default(&g_prm, 0.1f, 32, 32, 32), "sph_hip_params_default");
   // mirror the constants the GUI getters hand out
   mSimulationScale = g_prm.sim_scale;
   mSimulationScaleInverse = g_prm.sim_scale_inv;
   mH = g_prm.h; mH2 = g_prm.h2; mHTimes2 = g_prm.htimes2; mHTimes2Inv = g_prm.htimes2inv;
   mHScaled = g_prm.hscaled; mHScaled2 = g_prm.hscaled2; mHScaled6 = g_prm.hscaled6;
   mHScaled9 = g_prm.hscaled9;
   mParticleCount = M * 1024;
   mGridCellsX = g_prm.cells_x; mGridCellsY = g_prm.cells_y; mGridCellsZ = g_prm.cells_z;
   mGridCellCount = mGridCellsX * mGridCellsY * mGridCellsZ;
   mCellSize = g_prm.cell_size;
   mMaxX = g_prm.max_x; mMaxY = g_prm.max_y; mMaxZ = g_prm.max_z;
   mTimeStep = g_prm.time_step;
   totalSteps = (int)round(1.0f / mTimeStep);
   mRho0 = g_prm.rho0; mStiffness = g_prm.stiffness;
   mGravity = vec3(g_prm.gravity[0], g_prm.gravity[1], g_prm.gravity[2]);
   mViscosityScalar = g_prm.viscosity; mDamping = g_prm.damping;
   mGravConstant = g_prm.grav_const; mCentralMass = g_prm.central_mass;
   for (int c = 0; c < 3; c++) mCentralPos[c] = g_prm.central_pos[c];
   mSoftening = g_prm.softening;
   mCflLimit = g_prm.cfl_limit; mCflLimit2 = g_prm.cfl_limit2;
   mKernel1Scaled = g_prm.kernel1; mKernel2Scaled = g_prm.kernel2; mKernel3Scaled = g_prm.kernel3;
   mExamineCount = g_prm.examine_count;

   mSrcParticles = new Particle(mParticleCount);
   g_back = new Particle(mParticleCount);
   g_retired = new Particle(mParticleCount);
   for (int i = 0; i < mParticleCount; i++)
      mSrcParticles->mMass[i] = g_back->mMass[i] = g_retired->mMass[i] = 1.0f;
   g_sync_mirror = getenv("SPH_DROPIN_SYNC_MIRROR") != nullptr;
   mVoxelIds = new int[mParticleCount];
   mVoxelCoords = new vec3i[mParticleCount];
   mGrid = new QList<uint32_t>[mGridCellCount];
   mNeighbors = nullptr;               // lists live on the device
   mNeighborDistancesScaled = nullptr;

   initParticlePolitionsSphere();
   check(sph_hip_create(&g_ctx, &g_prm, mParticleCount, g_mode, 0), "sph_hip_create");
   check(sph_hip_upload(g_ctx, mParticleCount, mSrcParticles->mPosition.data(),
                        mSrcParticles->mVelocity.data(), mSrcParticles->mMass.data()),
         "sph_hip_upload");
   // before the first snapshot arrives the GUI sees the initial condition in both mirrors
   g_back->mPosition = g_retired->mPosition = mSrcParticles->mPosition;
   g_back->mVelocity = g_retired->mVelocity = mSrcParticles->mVelocity;
   g_counts_front.assign(mGridCellCount, 0);
   g_counts_back.assign(mGridCellCount, 0);
   for (Particle* p : {mSrcParticles, g_back, g_retired}) {   // page-lock the mirrors (a failure only costs overlap)
      (void)sph_hip_host_register(p->mPosition.data(), p->mPosition.size() * sizeof(float));
      (void)sph_hip_host_register(p->mVelocity.data(), p->mVelocity.size() * sizeof(float));
      (void)sph_hip_host_register(p->mDensity.data(), p->mDensity.size() * sizeof(float));
      (void)sph_hip_host_register(p->mAcceleration.data(), p->mAcceleration.size() * sizeof(float));
      (void)sph_hip_host_register(p->mNeighborCount.data(), p->mNeighborCount.size() * sizeof(int));
   }
   (void)sph_hip_host_register(g_counts_front.data(), g_counts_front.size() * sizeof(int32_t));
   (void)sph_hip_host_register(g_counts_back.data(), g_counts_back.size() * sizeof(int32_t));
   g_front_slot = &mSrcParticles;
   g_grid = mGrid;
   g_cells = mGridCellCount;
}

// The snapshot that was travelling has arrived: it becomes what getParticles() / getGrid() show.
// The GUI fetches the pointer every frame (src/visualization.cpp:145, 178).  Three buffers: the
// old front is RETIRED for one swap before it becomes the target of a copy again, so a frame that
// fetched the pointer just before this swap goes on reading a complete older state - with two
// buffers the same step() that swapped would at once start a DMA into what that frame reads.
// (The per-voxel lists below are resized here, on the solver thread, while the GUI may call
// count() on them: the reference's own clearGrid()/push_back race, src/sph.cpp:429-481.)
static void publish_mirror()
{
   Particle* filled = g_back;
   g_back = g_retired;
   g_retired = *g_front_slot;
   *g_front_slot = filled;
   g_counts_front.swap(g_counts_back);
   for (int c = 0; c < g_cells; c++) {   // only count() is ever read from these lists
      QList<uint32_t>& l = g_grid[c];
      const int want = g_counts_front[c];
      if (l.size() == want) continue;
      while (l.size() > want) l.removeLast();
      while (l.size() < want) l.append(0u);
   }
   g_step_mirrored = g_step_in_flight;
   g_copy_in_flight = false;
}

static void request_mirror()
{
   Particle* p = g_back;
   int started = 0;
   check(sph_hip_download_async(g_ctx, p->mPosition.data(), p->mVelocity.data(), p->mDensity.data(),
                                p->mAcceleration.data(), p->mNeighborCount.data(),
                                g_counts_back.data(), &started), "sph_hip_download_async");
   if (started) {
      g_copy_in_flight = true;
      g_step_in_flight = g_steps_taken;
   }
}

// Free functions for hosts that need the mirror to show the LAST step (file writers, tests);
// the GUI never calls them.  Declared by their users (the reference's headers stay unchanged).
void sph_dropin_sync_mirror()
{
   if (g_copy_in_flight) {
      if (sph_hip_download_done(g_ctx, 1) < 0) check(SPH_HIP_ERR_DEVICE, "sph_hip_download_done");
      publish_mirror();
   }
   if (g_step_mirrored != g_steps_taken) {   // the snapshot that arrived was of an earlier step
      request_mirror();
      if (sph_hip_download_done(g_ctx, 1) < 0) check(SPH_HIP_ERR_DEVICE, "sph_hip_download_done");
      publish_mirror();
   }
}
long long sph_dropin_steps_returned_before_copy() { return g_returned_before_copy; }

SPH::~SPH()
{
   stopSimulation();
   quit();
   wait();
   sph_hip_destroy(g_ctx);
   g_ctx = nullptr;
}

bool SPH::isStopped() const { mMutex.lock(); bool s = mStopped; mMutex.unlock(); return s; }
bool SPH::isPaused() const { mMutex.lock(); bool p = mPaused; mMutex.unlock(); return p; }
void SPH::pauseResume() { mMutex.lock(); mPaused = !mPaused; mMutex.unlock(); }
void SPH::stopSimulation() { mMutex.lock(); mStopped = true; mMutex.unlock(); }

void SPH::run()
{
   int stepCount = 0;
   mkdir("out", 0777);
   std::ofstream energy("out/energy.txt"), timing("out/timing.txt"), momentum("out/angularmomentum.txt");
   momentum << "Step, Angular Momentum" << std::endl;
   energy << "Step, Kinetic Energy, Potential Energy, Total Energy" << std::endl;
   timing << "Step, Voxelize, Find Neighbors, Compute Density, Compute Pressure, "
             "Compute Acceleration, Integrate" << std::endl;
   while (!isStopped() && stepCount <= totalSteps) {
      if (isPaused()) continue;
      step();
      energy << stepCount << ", " << mKineticEnergyTotal << ", " << mPotentialEnergyTotal << ", "
             << mKineticEnergyTotal + mPotentialEnergyTotal << std::endl;
      timing << stepCount << ", " << timeVoxelize << ", " << timeFindNeighbors << ", "
             << timeComputeDensity << ", " << timeComputePressure << ", " << timeComputeAcceleration
             << ", " << timeIntegrate << std::endl;
      momentum << stepCount << ", " << mAngularMomentumTotal.length() << std::endl;
      stepCount++;
   }
}

void SPH::step()
{
   // parameters edited by the GUI since the last step (SphConfig::writeValuesToSimulation runs on
   // the GUI thread): one consistent copy per step
   sph_hip_params prm;
   {
      std::lock_guard<std::mutex> lock(g_prm_mutex);
      prm = g_prm;
   }
   check(sph_hip_set_params(g_ctx, &prm), "sph_hip_set_params");
   // a snapshot requested by an earlier step has arrived meanwhile: show it
   if (g_copy_in_flight && sph_hip_download_done(g_ctx, 0) == 1) publish_mirror();
   check(sph_hip_step(g_ctx), "sph_hip_step");
   g_steps_taken++;
   {
      // the line the reference's step() appends to out/neighbors.txt (src/sph.cpp:203, 232, 301):
      // opened in append mode every step, silently dropped when ./out does not exist - in step(), not
      // in run(), so that a host that drives step() itself (the GUI's single-step button) gets it too
      std::ofstream neighbors("out/neighbors.txt", std::ios_base::app);
      if (neighbors) {
         int32_t avg = 0, mx = 0, mn = 0;
         check(sph_hip_get_neighbor_stats(g_ctx, &avg, &mx, &mn), "sph_hip_get_neighbor_stats");
         neighbors << avg << ", " << mx << ", " << mn << std::endl;
      }
   }

   float ms[6];
   check(sph_hip_get_timings(g_ctx, ms), "sph_hip_get_timings");
   timeVoxelize = (int)ms[0]; timeFindNeighbors = (int)ms[1]; timeComputeDensity = (int)ms[2];
   timeComputePressure = (int)ms[3]; timeComputeAcceleration = (int)ms[4];
   timeIntegrate = (int)ms[5];
   check(sph_hip_get_energy(g_ctx, &mKineticEnergyTotal, &mPotentialEnergyTotal),
         "sph_hip_get_energy");

   // ask for this step's state into the back mirror; the copy travels while the next steps run
   if (!g_copy_in_flight) request_mirror();
   if (g_sync_mirror) {
      sph_dropin_sync_mirror();
   } else if (g_copy_in_flight && sph_hip_download_done(g_ctx, 0) == 0) {
      g_returned_before_copy++;
   }
   emit updateElapsed(timeVoxelize, timeFindNeighbors, timeComputeDensity, timeComputePressure,
                      timeComputeAcceleration, timeIntegrate);
   emit stepFinished();
}

// ---- default scene: same sequence of rand() calls and float/double arithmetic as the
// ---- reference's initParticlePolitionsSphere (src/sph.cpp:361-425)
void SPH::initParticlePolitionsSphere()
{
   srand(42);
   const float cx = mMaxX * 0.5f, cy = mMaxY * 0.5f, cz = mMaxZ * 0.5f, radius = 2.0f;
   for (int i = 0; i < mParticleCount; i++) {
      float x, y, z, dist;
      do {
         x = rand() / (float)RAND_MAX; y = rand() / (float)RAND_MAX; z = rand() / (float)RAND_MAX;
         x *= mGridCellsX * mHTimes2; y *= mGridCellsY * mHTimes2; z *= mGridCellsZ * mHTimes2;
         if (x == (float)mGridCellsX) x -= 0.00001f;
         if (y == (float)mGridCellsY) y -= 0.00001f;
         if (z == (float)mGridCellsZ) z -= 0.00001f;
         dist = sqrtf((x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz));
      } while (dist > radius);
      mSrcParticles->mPosition[3 * i] = x;
      mSrcParticles->mPosition[3 * i + 1] = y;
      mSrcParticles->mPosition[3 * i + 2] = z;
      const float phi = atan2f(z - cz, x - cx);
      const double amp = 20.0f * pow((double)dist + (double)mHScaled * 0.5, -0.5);
      mSrcParticles->mVelocity[3 * i] = (float)(amp * -sinf(phi));
      mSrcParticles->mVelocity[3 * i + 2] = (float)(amp * cosf(phi));
      mSrcParticles->mVelocity[3 * i + 1] = ((rand() / (float)RAND_MAX) * 0.5f) - 0.25f;
   }
}
void SPH::initParticlePositionsRandom() {}

// ---- per-particle pipeline members (protected; nothing outside step() calls them in the reference).
// The GPU runs each phase for ALL particles at once, so a subclass that drives the phases itself gets
// the whole loop from the call for particle 0 and nothing from the others - the same state after
// "for (i = 0; i < N; i++) phase(i, ...)" as in the reference (src/sph.cpp:216-289), instead of silence.
// The list arguments are the reference's own members; the lists live on the device
// (sph_hip_download_neighbor_lists brings them over).
void SPH::clearGrid() {}
void SPH::voxelizeParticles() { check(sph_hip_voxelize(g_ctx), "sph_hip_voxelize"); }
void SPH::findNeighbors(int particleIndex, uint32_t*, int, int, int, float*)
{
   if (particleIndex == 0) check(sph_hip_find_neighbors(g_ctx), "sph_hip_find_neighbors");
}
void SPH::computeDensity(int particleIndex, uint32_t*, float*)
{
   if (particleIndex == 0) check(sph_hip_compute_density(g_ctx), "sph_hip_compute_density");
}
void SPH::computePressure(int) {}      // (a no-op in the reference too: src/sph.cpp:769-775)
void SPH::computeAcceleration(int particleIndex, uint32_t*, float*)
{
   if (particleIndex == 0) check(sph_hip_compute_acceleration(g_ctx), "sph_hip_compute_acceleration");
}
void SPH::integrate(int particleIndex)
{
   if (particleIndex != 0) return;
   check(sph_hip_integrate(g_ctx), "sph_hip_integrate");
   g_steps_taken++;      // (the last phase of a step: sph_dropin_sync_mirror() then fetches this state)
   check(sph_hip_get_energy(g_ctx, &mKineticEnergyTotal, &mPotentialEnergyTotal), "sph_hip_get_energy");
}
int SPH::evaluateNeighbor(int, int) { return 0; }
int SPH::computeVoxelId(int x, int y, int z) { return (z * mGridCellsY + y) * mGridCellsX + x; }
void SPH::applyBoundary(vec3, float, vec3*, float, vec3, vec3*) {}
void SPH::handleBoundaryConditions(vec3, vec3*, float, vec3*) {}
void SPH::clearNeighbors() {}
void SPH::memClear32(void*, int) {}

// ---- getters / setters (reference src/sph.cpp:1172-1289) -----------------------------------------
float SPH::getCellSize() const { return mCellSize; }
Particle* SPH::getParticles() { return mSrcParticles; }
int SPH::getParticleCount() const { return mParticleCount; }
void SPH::getGridCellCounts(int& x, int& y, int& z) { x = mGridCellsX; y = mGridCellsY; z = mGridCellsZ; }
void SPH::getParticleBounds(float& x, float& y, float& z) { x = mMaxX; y = mMaxY; z = mMaxZ; }
float SPH::getInteractionRadius2() const { return mHScaled2; }
QList<uint32_t>* SPH::getGrid() { return mGrid; }
vec3 SPH::getGravity() const { return mGravity; }
void SPH::setGravity(const vec3& g)
{
   std::lock_guard<std::mutex> lock(g_prm_mutex);
   mGravity = g;
   g_prm.gravity[0] = g.x; g_prm.gravity[1] = g.y; g_prm.gravity[2] = g.z;
}
float SPH::getCflLimit() const { return mCflLimit; }
void SPH::setCflLimit(float v)
{
   std::lock_guard<std::mutex> lock(g_prm_mutex);
   mCflLimit = v; mCflLimit2 = v * v;
   g_prm.cfl_limit = mCflLimit; g_prm.cfl_limit2 = mCflLimit2;
}
float SPH::getDamping() const { return mDamping; }
void SPH::setDamping(float v)
{
   std::lock_guard<std::mutex> lock(g_prm_mutex);
   mDamping = v; g_prm.damping = v;
}
float SPH::getTimeStep() const { return mTimeStep; }
void SPH::setTimeStep(float v)
{
   std::lock_guard<std::mutex> lock(g_prm_mutex);
   mTimeStep = v; g_prm.time_step = v;
}
float SPH::getViscosityScalar() const { return mViscosityScalar; }
void SPH::setViscosityScalar(float v)
{
   std::lock_guard<std::mutex> lock(g_prm_mutex);
   mViscosityScalar = v; g_prm.viscosity = v;
}
float SPH::getStiffness() const { return mStiffness; }
void SPH::setStiffness(float v)
{
   std::lock_guard<std::mutex> lock(g_prm_mutex);
   mStiffness = v; g_prm.stiffness = v;
}
