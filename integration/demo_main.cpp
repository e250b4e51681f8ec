// Headless driver for the drop-in: constructs the reference's `SPH` (body from sph_dropin.cpp,
// header from the reference), calls step() a few times like `./sph r` would, and prints SHA-free
// raw sums so the test can compare the host mirrors with the reference goldens.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define protected public
#include "sph.h"
#undef protected
#include "particle.h"

// free functions of sph_dropin.cpp for hosts that read the mirror right after stepping
void sph_dropin_sync_mirror();
long long sph_dropin_steps_returned_before_copy();

int main(int argc, char** argv)
{
   const int steps = argc > 1 ? atoi(argv[1]) : 1;
   const char* out = argc > 2 ? argv[2] : "dropin_state.bin";
   // third argument "phases": drive the protected per-particle members in the reference's own call
   // pattern (src/sph.cpp:208-289) instead of step() - what a subclass of SPH could do
   const bool phases = argc > 3 && strcmp(argv[3], "phases") == 0;
   SPH sph;
   for (int s = 0; s < steps; s++) {
      if (!phases) {
         sph.step();
         continue;
      }
      const int n = sph.getParticleCount();
      sph.voxelizeParticles();
      for (int i = 0; i < n; i++) sph.findNeighbors(i, nullptr, 0, 0, 0, nullptr);
      for (int i = 0; i < n; i++) sph.computeDensity(i, nullptr, nullptr);
      for (int i = 0; i < n; i++) sph.computeAcceleration(i, nullptr, nullptr);
      for (int i = 0; i < n; i++) sph.integrate(i);
   }
   // step() returns while the snapshot of its state is still on its way to the host mirror (the
   // GUI reads whatever complete mirror is current); a program that wants the LAST step waits
   printf("dropin: %lld of %d step() calls returned before their snapshot had arrived\n",
          sph_dropin_steps_returned_before_copy(), steps);
   sph_dropin_sync_mirror();
   Particle* p = sph.getParticles();
   const int n = sph.getParticleCount();
   FILE* f = fopen(out, "wb");
   if (!f) return 2;
   fwrite(&n, sizeof(int), 1, f);
   fwrite(p->mPosition.data(), sizeof(float), 3 * n, f);
   fwrite(p->mVelocity.data(), sizeof(float), 3 * n, f);
   fwrite(p->mDensity.data(), sizeof(float), n, f);
   fwrite(p->mAcceleration.data(), sizeof(float), 3 * n, f);
   fwrite(p->mNeighborCount.data(), sizeof(int), n, f);
   int gx, gy, gz;
   sph.getGridCellCounts(gx, gy, gz);
   long long occupied = 0, total = 0;
   for (int c = 0; c < gx * gy * gz; c++) {
      total += sph.getGrid()[c].count();
      occupied += sph.getGrid()[c].count() > 0;
   }
   fwrite(&total, sizeof(long long), 1, f);
   fclose(f);
   printf("dropin: %d particles, %d steps, grid holds %lld in %lld voxels, KE %.9g PE %.9g\n", n,
          steps, total, occupied, (double)sph.mKineticEnergyTotal, (double)sph.mPotentialEnergyTotal);
   return 0;
}
