// TEST INFRASTRUCTURE ONLY — not part of the product path.
//
// C-ABI driver around the *unmodified* reference solver.  It is compiled (by
// oracle/Makefile) together with the reference's own src/sph.cpp, src/particle.cpp,
// src/vec3.cpp and the moc output for src/sph.h, straight from /root/reference, into
// oracle/_ref/libsphref.so.  Nothing of the reference is copied into this repository:
// this file only *calls* the reference's methods and pokes its protected members
// (the `#define protected public` trick recommended in SURVEY.md §8(c)).
//
// What it exposes:
//   * REF mode  — the reference pipeline exactly as SPH::step() drives it
//                 (src/sph.cpp:190-304): voxelizeParticles, findNeighbors,
//                 computeDensity, computeAcceleration, integrate.
//   * FULL mode — the caller supplies complete in-radius neighbour lists in the
//                 canonical order; the reference's own computeDensity /
//                 computeAcceleration (src/sph.cpp:721-766, 778-934) do the sums.
//   * geometry / constant overrides so that scenes other than the compiled-in
//     `M*1024` sphere (dam-break, other grid sizes, other h) can be run.
//
// Built with -DM=1 so the constructor's own allocation is small; ref_resize()
// then re-allocates every per-particle array for the requested count.

#define protected public
#define private public
#include "sph.h"
#include "particle.h"
#undef protected
#undef private

#include <cmath>
#include <cstdint>
#include <cstring>
#include <sys/stat.h>

namespace {

SPH* g_sph = nullptr;

SPH* S()
{
   if (!g_sph)
      g_sph = new SPH();
   return g_sph;
}

}  // namespace

extern "C" {

// ---- lifetime / sizing -------------------------------------------------------------

// Re-allocate all per-particle storage for n particles (mirrors the allocations at
// src/sph.cpp:100-113).  Old storage is leaked exactly like the reference leaks it.
void ref_resize(int n)
{
   SPH* s = S();
   s->mParticleCount = n;
   s->mSrcParticles = new Particle(n);
   s->mVoxelIds = new int[n];
   s->mVoxelCoords = new vec3i[n];
   for (int i = 0; i < n; i++)
      s->mSrcParticles->mMass[i] = 1.0f;
   s->mNeighbors = new uint32_t[(size_t)n * s->mExamineCount];
   s->mNeighborDistancesScaled = new float[(size_t)n * s->mExamineCount];
}

// Re-run the reference's own default initial condition (src/sph.cpp:361-425) for the
// current particle count; equals what a `-DM=<n/1024>` build constructs.
void ref_init_sphere() { S()->initParticlePolitionsSphere(); }

int ref_particle_count() { return S()->mParticleCount; }

// ---- geometry / constants ----------------------------------------------------------

// Change the grid (cells per axis and voxel edge).  The reference hard-codes 32^3 voxels
// of edge 2h (src/sph.cpp:60-67); the oracle needs other shapes for dam-break scenes.
void ref_set_grid(int nx, int ny, int nz, float cell_size)
{
   SPH* s = S();
   s->mGridCellsX = nx;
   s->mGridCellsY = ny;
   s->mGridCellsZ = nz;
   s->mGridCellCount = nx * ny * nz;
   s->mCellSize = cell_size;
   s->mMaxX = cell_size * nx;
   s->mMaxY = cell_size * ny;
   s->mMaxZ = cell_size * nz;
   s->mGrid = new QList<uint32_t>[s->mGridCellCount];
}

// Raw setters for every constant the hot path reads.  Values are computed by the caller
// (oracle/sph_oracle.c: oracle_params_for_h) so that nothing is re-derived here.
void ref_set_h(float h, float h2, float hscaled, float hscaled2, float hscaled6, float hscaled9,
               float htimes2, float htimes2inv, float k1, float k2, float k3, float softening)
{
   SPH* s = S();
   s->mH = h;
   s->mH2 = h2;
   s->mHScaled = hscaled;
   s->mHScaled2 = hscaled2;
   s->mHScaled6 = hscaled6;
   s->mHScaled9 = hscaled9;
   s->mHTimes2 = htimes2;
   s->mHTimes2Inv = htimes2inv;
   s->mKernel1Scaled = k1;
   s->mKernel2Scaled = k2;
   s->mKernel3Scaled = k3;
   s->mSoftening = softening;
}

// mSimulationScale / mSimulationScaleInverse (reference src/sph.cpp:48-49, fixed at 1 there)
void ref_set_scale(float scale, float scale_inv)
{
   S()->mSimulationScale = scale;
   S()->mSimulationScaleInverse = scale_inv;
}

void ref_set_physics(float rho0, float stiffness, float viscosity, float dt, float cfl,
                     float grav_const, float central_mass, const float* central_pos)
{
   SPH* s = S();
   s->mRho0 = rho0;
   s->mStiffness = stiffness;
   s->mViscosityScalar = viscosity;
   s->mTimeStep = dt;
   s->setCflLimit(cfl);
   s->mGravConstant = grav_const;
   s->mCentralMass = central_mass;
   s->mCentralPos[0] = central_pos[0];
   s->mCentralPos[1] = central_pos[1];
   s->mCentralPos[2] = central_pos[2];
}

// Dump the constants the constructor derived (src/sph.cpp:46-98) so the restatement can be
// checked against them.  out[] needs 32 floats.
// fresh != 0: read them from a newly constructed object (what SPH::SPH() derives), independent
// of anything the harness has overridden on the shared one
void ref_get_constants2(float* out, int fresh);
void ref_get_constants(float* out) { ref_get_constants2(out, 0); }

void ref_get_constants2(float* out, int fresh)
{
   SPH* s = fresh ? new SPH() : S();
   int k = 0;
   out[k++] = s->mH;
   out[k++] = s->mH2;
   out[k++] = s->mHScaled;
   out[k++] = s->mHScaled2;
   out[k++] = s->mHScaled6;
   out[k++] = s->mHScaled9;
   out[k++] = s->mHTimes2;
   out[k++] = s->mHTimes2Inv;
   out[k++] = s->mKernel1Scaled;
   out[k++] = s->mKernel2Scaled;
   out[k++] = s->mKernel3Scaled;
   out[k++] = s->mSoftening;
   out[k++] = s->mRho0;
   out[k++] = s->mStiffness;
   out[k++] = s->mViscosityScalar;
   out[k++] = s->mTimeStep;
   out[k++] = s->mCflLimit;
   out[k++] = s->mCflLimit2;
   out[k++] = s->mGravConstant;
   out[k++] = s->mCentralMass;
   out[k++] = s->mCentralPos[0];
   out[k++] = s->mCentralPos[1];
   out[k++] = s->mCentralPos[2];
   out[k++] = s->mCellSize;
   out[k++] = s->mMaxX;
   out[k++] = s->mMaxY;
   out[k++] = s->mMaxZ;
   out[k++] = s->mSimulationScale;
   out[k++] = s->mDamping;
   out[k++] = (float)s->mGridCellsX;
   out[k++] = (float)s->mGridCellsY;
   out[k++] = (float)s->mGridCellsZ;
   if (fresh) delete s;
}

// ---- state in / out ----------------------------------------------------------------

void ref_set_state(const float* pos, const float* vel, const float* mass)
{
   SPH* s = S();
   int n = s->mParticleCount;
   if (pos) std::memcpy(s->mSrcParticles->mPosition.data(), pos, sizeof(float) * 3 * n);
   if (vel) std::memcpy(s->mSrcParticles->mVelocity.data(), vel, sizeof(float) * 3 * n);
   if (mass) std::memcpy(s->mSrcParticles->mMass.data(), mass, sizeof(float) * n);
}

void ref_set_density(const float* rho)
{
   SPH* s = S();
   std::memcpy(s->mSrcParticles->mDensity.data(), rho, sizeof(float) * s->mParticleCount);
}

void ref_get_state(float* pos, float* vel, float* mass, float* rho, float* acc, int* ncount)
{
   SPH* s = S();
   int n = s->mParticleCount;
   Particle* p = s->mSrcParticles;
   if (pos) std::memcpy(pos, p->mPosition.data(), sizeof(float) * 3 * n);
   if (vel) std::memcpy(vel, p->mVelocity.data(), sizeof(float) * 3 * n);
   if (mass) std::memcpy(mass, p->mMass.data(), sizeof(float) * n);
   if (rho) std::memcpy(rho, p->mDensity.data(), sizeof(float) * n);
   if (acc) std::memcpy(acc, p->mAcceleration.data(), sizeof(float) * 3 * n);
   if (ncount) std::memcpy(ncount, p->mNeighborCount.data(), sizeof(int) * n);
}

void ref_get_voxels(int* coords_xyz, int* ids)
{
   SPH* s = S();
   int n = s->mParticleCount;
   for (int i = 0; i < n; i++) {
      if (coords_xyz) {
         coords_xyz[3 * i + 0] = s->mVoxelCoords[i].x;
         coords_xyz[3 * i + 1] = s->mVoxelCoords[i].y;
         coords_xyz[3 * i + 2] = s->mVoxelCoords[i].z;
      }
      if (ids) ids[i] = s->mVoxelIds[i];
   }
}

// per-voxel occupancy, what Visualization reads through getGrid()[i].count()
void ref_get_grid_counts(int* counts)
{
   SPH* s = S();
   for (int c = 0; c < s->mGridCellCount; c++)
      counts[c] = s->mGrid[c].count();
}

int ref_examine_count() { return S()->mExamineCount; }

// list capacity used by the next ref_resize() (the reference fixes it at 32, src/sph.cpp:98)
void ref_set_examine_count(int cap) { S()->mExamineCount = cap; }

void ref_get_lists(uint32_t* neighbors, float* dists)
{
   SPH* s = S();
   size_t m = (size_t)s->mParticleCount * s->mExamineCount;
   if (neighbors) std::memcpy(neighbors, s->mNeighbors, sizeof(uint32_t) * m);
   if (dists) std::memcpy(dists, s->mNeighborDistancesScaled, sizeof(float) * m);
}

// FULL mode: install caller-built lists (row stride = cap) and their counts.
void ref_set_lists(int cap, const uint32_t* neighbors, const float* dists, const int* counts)
{
   SPH* s = S();
   int n = s->mParticleCount;
   s->mExamineCount = cap;
   s->mNeighbors = new uint32_t[(size_t)n * cap];
   s->mNeighborDistancesScaled = new float[(size_t)n * cap];
   std::memcpy(s->mNeighbors, neighbors, sizeof(uint32_t) * (size_t)n * cap);
   std::memcpy(s->mNeighborDistancesScaled, dists, sizeof(float) * (size_t)n * cap);
   std::memcpy(s->mSrcParticles->mNeighborCount.data(), counts, sizeof(int) * n);
}

// The reference's wall handling (src/sph.cpp:1025-1148), defined but never called by step():
// run it for n particles.  position: old positions; vel / newpos are updated in place.
void ref_boundary(int n, const float* position, float* vel, float time_step, float* newpos)
{
   SPH* s = S();
   for (int i = 0; i < n; i++) {
      vec3 p(position[3 * i], position[3 * i + 1], position[3 * i + 2]);
      vec3 v(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]);
      vec3 np(newpos[3 * i], newpos[3 * i + 1], newpos[3 * i + 2]);
      s->handleBoundaryConditions(p, &v, time_step, &np);
      vel[3 * i] = v.x; vel[3 * i + 1] = v.y; vel[3 * i + 2] = v.z;
      newpos[3 * i] = np.x; newpos[3 * i + 1] = np.y; newpos[3 * i + 2] = np.z;
   }
}

void ref_set_damping(float damping) { S()->setDamping(damping); }

void ref_get_energy(float* ke, float* pe)
{
   *ke = S()->mKineticEnergyTotal;
   *pe = S()->mPotentialEnergyTotal;
}

// ---- the pipeline, phase by phase (same call pattern as src/sph.cpp:208-289) --------

void ref_voxelize() { S()->voxelizeParticles(); }

void ref_find_neighbors()
{
   SPH* s = S();
   for (int i = 0; i < s->mParticleCount; i++) {
      const vec3i& v = s->mVoxelCoords[i];
      s->findNeighbors(i, &s->mNeighbors[(size_t)i * s->mExamineCount], v.x, v.y, v.z,
                       &s->mNeighborDistancesScaled[(size_t)i * s->mExamineCount]);
   }
}

void ref_compute_density()
{
   SPH* s = S();
   for (int i = 0; i < s->mParticleCount; i++)
      s->computeDensity(i, &s->mNeighbors[(size_t)i * s->mExamineCount],
                        &s->mNeighborDistancesScaled[(size_t)i * s->mExamineCount]);
}

void ref_compute_acceleration()
{
   SPH* s = S();
   for (int i = 0; i < s->mParticleCount; i++)
      s->computeAcceleration(i, &s->mNeighbors[(size_t)i * s->mExamineCount],
                             &s->mNeighborDistancesScaled[(size_t)i * s->mExamineCount]);
}

void ref_integrate()
{
   SPH* s = S();
   s->mKineticEnergyTotal = 0.0f;
   s->mPotentialEnergyTotal = 0.0f;
   for (int i = 0; i < s->mParticleCount; i++)
      s->integrate(i);
}

// The reference's own step() (src/sph.cpp:190-304), untouched.  It appends to
// out/neighbors.txt when ./out exists; callers that want that file mkdir it first.
void ref_step() { S()->step(); }

}  // extern "C"
