/* TEST INFRASTRUCTURE ONLY — see sph_oracle.h.
 *
 * CPU restatement of the reference hot path, written from the behaviour documented in
 * SURVEY.md §8(a) rows A0-A8 and checked bit-for-bit against the compiled reference
 * (oracle/_ref/libsphref.so).  Compile with -O2 -ffp-contract=off and no fast-math: every
 * fp32 operation below is meant to round exactly once, in the order written.
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define CHUNK 8 /* `K` in the reference, src/sph.cpp:32 */

/* ------------------------------------------------------------------------------------
 * A0 — constants.  reference src/sph.cpp:46-98.  pow() results are doubles narrowed to
 * float on assignment; kernel normalisations are float arithmetic on (float)M_PI. */
void oracle_params_for_h(sph_oracle_params* p, float h, int cells_x, int cells_y, int cells_z)
{
   memset(p, 0, sizeof(*p));
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
   p->gravity[0] = p->gravity[1] = p->gravity[2] = 0.0f;
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
   /* FULL grid: cells of edge h*(1+1e-4) (>= h so that the 27-cell neighbourhood is
    * complete) covering the same box. */
   {
      double edge = (double)h * 1.0001;
      p->full_cell_inv = (float)(1.0 / edge);
      p->full_cells_x = (int)ceil((double)p->max_x / edge);
      p->full_cells_y = (int)ceil((double)p->max_y / edge);
      p->full_cells_z = (int)ceil((double)p->max_z / edge);
   }
}

/* ------------------------------------------------------------------------------------
 * A0' — default scene.  reference src/sph.cpp:361-425.  glibc rand(), srand(42). */
void oracle_init_sphere(const sph_oracle_params* p, int n, float* pos, float* vel)
{
   srand(42);
   float dist = 0.0f;
   float x = 0.0f, y = 0.0f, z = 0.0f;
   float cx = p->max_x * 0.5f, cy = p->max_y * 0.5f, cz = p->max_z * 0.5f;
   float radius = 2.0f;
   for (int i = 0; i < n; i++) {
      do {
         x = rand() / (float)RAND_MAX;
         y = rand() / (float)RAND_MAX;
         z = rand() / (float)RAND_MAX;
         x *= p->cells_x * p->htimes2;
         y *= p->cells_y * p->htimes2;
         z *= p->cells_z * p->htimes2;
         if (x == (float)p->cells_x) x -= 0.00001f;
         if (y == (float)p->cells_y) y -= 0.00001f;
         if (z == (float)p->cells_z) z -= 0.00001f;
         dist = (x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz);
         dist = sqrtf(dist);
      } while (dist > radius);
      pos[3 * i + 0] = x;
      pos[3 * i + 1] = y;
      pos[3 * i + 2] = z;
      /* <math.h> in C++ resolves atan2/sin/cos on float arguments to the float overloads;
       * pow(double, double) stays double and the products are double, narrowed on store */
      float phi = atan2f(z - p->max_z * 0.5f, x - p->max_x * 0.5f);
      float vx = (float)(20.0f * pow((double)dist + (double)p->hscaled * 0.5, -0.5) * -sinf(phi));
      float vz = (float)(20.0f * pow((double)dist + (double)p->hscaled * 0.5, -0.5) * cosf(phi));
      float vy = ((rand() / (float)RAND_MAX) * 0.5f) - 0.25f;
      vel[3 * i + 0] = vx;
      vel[3 * i + 1] = vy;
      vel[3 * i + 2] = vz;
   }
}

/* ------------------------------------------------------------------------------------
 * A1 — voxelize.  reference src/sph.cpp:438-481 (+ computeVoxelId :1151-1154).
 * The reference keeps one QList per voxel filled by a serial push_back loop, so each
 * voxel's list is in ascending particle index; a stable counting sort gives the same
 * lists as CSR. */
static void build_csr(int n, int ncells, const int32_t* ids, int32_t* cell_start,
                      int32_t* cell_items)
{
   memset(cell_start, 0, sizeof(int32_t) * (size_t)(ncells + 1));
   for (int i = 0; i < n; i++) cell_start[ids[i] + 1]++;
   for (int c = 0; c < ncells; c++) cell_start[c + 1] += cell_start[c];
   int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)ncells);
   memcpy(fill, cell_start, sizeof(int32_t) * (size_t)ncells);
   for (int i = 0; i < n; i++) cell_items[fill[ids[i]]++] = i;
   free(fill);
}

static inline int clampi(int v, int hi)
{
   if (v < 0) v = 0;
   if (v >= hi) v = hi - 1;
   return v;
}

void oracle_voxelize(const sph_oracle_params* p, int n, const float* pos, int32_t* coords_xyz,
                     int32_t* ids, int32_t* cell_start, int32_t* cell_items)
{
   for (int i = 0; i < n; i++) {
      int vx = (int)floor((double)(pos[3 * i + 0] * p->htimes2inv));
      int vy = (int)floor((double)(pos[3 * i + 1] * p->htimes2inv));
      int vz = (int)floor((double)(pos[3 * i + 2] * p->htimes2inv));
      vx = clampi(vx, p->cells_x);
      vy = clampi(vy, p->cells_y);
      vz = clampi(vz, p->cells_z);
      coords_xyz[3 * i + 0] = vx;
      coords_xyz[3 * i + 1] = vy;
      coords_xyz[3 * i + 2] = vz;
      ids[i] = (vz * p->cells_y + vy) * p->cells_x + vx;
   }
   build_csr(n, p->cells_x * p->cells_y * p->cells_z, ids, cell_start, cell_items);
}

/* ------------------------------------------------------------------------------------
 * A2 — the shipped neighbour search.  reference src/sph.cpp:484-692.
 * Behaviours restated deliberately:
 *   - octant slots 0,1,2,3',5,6,7; the intended slot 3 is overwritten (:536-543) and slot 4
 *     is never assigned — it behaves as skipped;
 *   - a slot is used only if 0 < v < cells on every axis (:578-582);
 *   - 32-bit wrapping LCG seeded by particle index + number of slots used so far (:590);
 *   - offset = lcg % len with C truncation (may be negative) (:591);
 *   - chunks of 8 consecutive list positions; the slot is abandoned as soon as any of the
 *     8 falls outside the list (:598-620);
 *   - only the first 4 of each 8 candidates are distance-tested (:651-671);
 *   - stop once more than examine_count-8 neighbours are stored (:679). */
void oracle_find_neighbors(const sph_oracle_params* p, int n, const float* pos,
                           const int32_t* coords_xyz, const int32_t* cell_start,
                           const int32_t* cell_items, uint32_t* neighbors, float* dists,
                           int32_t* counts)
{
   const int cap = p->examine_count;
   for (int i = 0; i < n; i++) {
      uint32_t* nb = neighbors + (size_t)i * cap;
      float* nd = dists + (size_t)i * cap;
      const float px = pos[3 * i + 0], py = pos[3 * i + 1], pz = pos[3 * i + 2];
      const int X = coords_xyz[3 * i + 0], Y = coords_xyz[3 * i + 1], Z = coords_xyz[3 * i + 2];

      float ox = px - (X * p->htimes2);
      float oy = py - (Y * p->htimes2);
      float oz = pz - (Z * p->htimes2);
      int sx = (ox > p->h) ? 1 : -1;
      int sy = (oy > p->h) ? 1 : -1;
      int sz = (oz > p->h) ? 1 : -1;

      int vx[8], vy[8], vz[8], live[8];
      for (int s = 0; s < 8; s++) live[s] = 1;
      vx[0] = X;      vy[0] = Y;      vz[0] = Z;
      vx[1] = X + sx; vy[1] = Y;      vz[1] = Z;
      vx[2] = X;      vy[2] = Y + sy; vz[2] = Z;
      vx[3] = X + sx; vy[3] = Y + sy; vz[3] = Z;
      live[4] = 0; vx[4] = vy[4] = vz[4] = 0;
      vx[5] = X + sx; vy[5] = Y;      vz[5] = Z + sz;
      vx[6] = X;      vy[6] = Y + sy; vz[6] = Z + sz;
      vx[7] = X + sx; vy[7] = Y + sy; vz[7] = Z + sz;

      int count = 0;
      int used = 0; /* almost_a_random */
      int enough = 0;

      for (int s = 0; s < 8 && !enough; s++) {
         if (!live[s]) continue;
         const int cx = vx[s], cy = vy[s], cz = vz[s];
         if (!(cx > 0 && cx < p->cells_x && cy > 0 && cy < p->cells_y && cz > 0 &&
               cz < p->cells_z))
            continue;
         const int id = (cz * p->cells_y + cy) * p->cells_x + cx;
         const int len = cell_start[id + 1] - cell_start[id];
         if (len == 0) continue;
         const int32_t* list = cell_items + cell_start[id];

         const int32_t lcg =
            (int32_t)(1664525u * (uint32_t)(i + used) + 1013904223u); /* wraps mod 2^32 */
         const int offset = lcg % len;
         used++;
         const int dir = (i % 2) ? -1 : 1;

         int ii = 0;
         const int max_steps = (len + CHUNK - 1) / CHUNK;
         for (int step = 0; step < max_steps; ++step) {
            int idx[CHUNK];
            int oob = 0;
            for (int j = 0; j < CHUNK; j++) {
               idx[j] = (offset + j) + ii * dir;
               if (idx[j] < 0 || idx[j] >= len) oob = 1;
            }
            if (oob) break;
            ii += CHUNK;

            for (int j = 0; j < 4; j++) { /* lanes 4..7 are gathered but never tested */
               const int q = list[idx[j]];
               if (q == i) continue;
               float dx = px - pos[3 * q + 0];
               float dy = py - pos[3 * q + 1];
               float dz = pz - pos[3 * q + 2];
               float dot = dx * dx + dy * dy + dz * dz;
               if (dot < p->h2) {
                  nb[count] = (uint32_t)q;
                  nd[count] = sqrtf(dot) * p->sim_scale;
                  count++;
               }
            }
            enough = (count > cap - CHUNK);
            if (enough) break;
         }
      }
      counts[i] = count;
   }
}

/* A3 — reference src/sph.cpp:204-232: integer-division average, max from -1, min from 34 */
void oracle_neighbor_stats(int n, const int32_t* counts, int32_t* avg, int32_t* mx, int32_t* mn)
{
   int sum = 0, hi = -1, lo = 34;
   for (int i = 0; i < n; i++) {
      sum += counts[i];
      if (counts[i] > hi) hi = counts[i];
      if (counts[i] < lo) lo = counts[i];
   }
   *avg = sum / n;
   *mx = hi;
   *mn = lo;
}

/* ------------------------------------------------------------------------------------
 * A4 — per-pair density term.  reference src/sph.cpp:737-761 */
static inline float density_term(const sph_oracle_params* p, float mass, float d)
{
   if (d > p->hscaled) return 0.0f;
   float t = (p->hscaled2 - (d * d));
   t = (t * t * t);
   float w = p->kernel1 * t;
   return (mass * w);
}

void oracle_density_lists(const sph_oracle_params* p, int n, int cap, const uint32_t* neighbors,
                          const float* dists, const int32_t* counts, const float* mass,
                          float* rho)
{
   for (int i = 0; i < n; i++) {
      const uint32_t* nb = neighbors + (size_t)i * cap;
      const float* nd = dists + (size_t)i * cap;
      float density = 0.0f;
      for (int k = 0; k < counts[i]; k++) {
         uint32_t q = nb[k];
         if (q >= (uint32_t)n) break;
         if (q != (uint32_t)i) {
            float d = nd[k];
            if (!(d > p->hscaled)) density += density_term(p, mass[q], d);
         }
      }
      rho[i] = density;
   }
}

/* ------------------------------------------------------------------------------------
 * A5 — acceleration.  reference src/sph.cpp:778-934.  State carried across one
 * particle's neighbour loop, so the list-driven and the cell-driven drivers share the
 * exact per-pair arithmetic. */
typedef struct {
   float pi, rhoi_inv, pi_div_rhoi2;
   float r[3], vi[3];
   float pg[3], vt[3];
} accel_state;

static inline void accel_begin(const sph_oracle_params* p, accel_state* s, const float* pos,
                               const float* vel, float rho_i)
{
   s->pi = (rho_i - p->rho0) * p->stiffness;
   s->rhoi_inv = ((s->pi > 0.0f) ? (1.0f / s->pi) : 1.0f); /* from pressure, :786 */
   float rhoi_inv2 = s->rhoi_inv * s->rhoi_inv;
   s->pi_div_rhoi2 = s->pi * rhoi_inv2;
   for (int c = 0; c < 3; c++) {
      s->r[c] = pos[c];
      s->vi[c] = vel[c];
      s->pg[c] = 0.0f;
      s->vt[c] = 0.0f;
   }
}

static inline void accel_pair(const sph_oracle_params* p, accel_state* s, const float* posj,
                              const float* velj, float mj, float rhoj, float d)
{
   float pj = (rhoj - p->rho0) * p->stiffness;
   float rhoj_inv = ((rhoj > 0.0f) ? (1.0f / rhoj) : 1.0f);
   float rhoj_inv2 = rhoj_inv * rhoj_inv;
   float rs[3], pgc[3];
   for (int c = 0; c < 3; c++) rs[c] = (s->r[c] - posj[c]) * p->sim_scale;
   /* float product, then a double add and a double divide, narrowed to float (:854-856) */
   for (int c = 0; c < 3; c++)
      pgc[c] = (float)((double)(p->kernel2 * rs[c]) / ((double)d + 0.01));

   float center = (p->hscaled - d);
   center *= center;
   center *= mj * s->pi_div_rhoi2 * (pj * rhoj_inv2); /* (mj*A) * (pj*rhojInv2), :860 */
   for (int c = 0; c < 3; c++) s->pg[c] += pgc[c] * center;

   center = (p->hscaled - d);
   center *= rhoj_inv * mj * p->kernel3;
   for (int c = 0; c < 3; c++) s->vt[c] += (velj[c] - s->vi[c]) * center;
   /* the rescale sits inside the neighbour loop (:880-882) */
   for (int c = 0; c < 3; c++) s->vt[c] *= p->viscosity * s->rhoi_inv;
}

static inline void accel_end(const sph_oracle_params* p, accel_state* s, float* acc)
{
   float a[3];
   for (int c = 0; c < 3; c++) a[c] = s->vt[c] - s->pg[c];

   float rs[3];
   for (int c = 0; c < 3; c++) rs[c] = (s->r[c] - p->central_pos[c]) * p->sim_scale;
   float dot = (rs[0] * rs[0]) + (rs[1] * rs[1]) + (rs[2] * rs[2]);
   dot = sqrtf(dot);
   float d3 = (dot + p->softening) * (dot + p->softening) * (dot + p->softening);
   for (int c = 0; c < 3; c++) {
      float g = rs[c] / d3;
      a[c] += -p->grav_const * p->central_mass * g;
   }
   if (p->apply_gravity) /* extension: uniform gravity enters next to the point-mass term */
      for (int c = 0; c < 3; c++) a[c] += p->gravity[c];
   dot = (a[0] * a[0]) + (a[1] * a[1]) + (a[2] * a[2]);
   if (dot > p->cfl_limit2) {
      float length = sqrtf(dot);
      float scale = p->cfl_limit / length;
      for (int c = 0; c < 3; c++) a[c] *= scale;
   }
   for (int c = 0; c < 3; c++) acc[c] = a[c];
}

void oracle_accel_lists(const sph_oracle_params* p, int n, int cap, const uint32_t* neighbors,
                        const float* dists, const int32_t* counts, const float* pos,
                        const float* vel, const float* mass, const float* rho, float* acc)
{
   for (int i = 0; i < n; i++) {
      const uint32_t* nb = neighbors + (size_t)i * cap;
      const float* nd = dists + (size_t)i * cap;
      accel_state s;
      accel_begin(p, &s, pos + 3 * i, vel + 3 * i, rho[i]);
      for (int k = 0; k < counts[i]; k++) {
         uint32_t q = nb[k];
         accel_pair(p, &s, pos + 3 * (size_t)q, vel + 3 * (size_t)q, mass[q], rho[q], nd[k]);
      }
      accel_end(p, &s, acc + 3 * i);
   }
}

/* ------------------------------------------------------------------------------------
 * Wall reflection.  reference src/sph.cpp:1124-1148 (applyBoundary) and :1025-1121
 * (handleBoundaryConditions).  vec3 operators are component-wise float ops; `*` between
 * vec3 and float scales, the dot product is written out (x*nx + y*ny + z*nz). */
static void apply_boundary(const sph_oracle_params* p, const float* position, float time_step,
                           float* new_position, float intersection_distance, const float* normal,
                           float* new_velocity)
{
   float intersection[3], reflection[3];
   for (int c = 0; c < 3; c++)
      intersection[c] = position[c] + (new_velocity[c] * intersection_distance);
   float dot = new_velocity[0] * normal[0] + new_velocity[1] * normal[1] +
               new_velocity[2] * normal[2];
   for (int c = 0; c < 3; c++) reflection[c] = new_velocity[c] - ((normal[c] * dot) * 2.0f);
   float remaining = time_step - intersection_distance;
   for (int c = 0; c < 3; c++) {
      new_velocity[c] = reflection[c];
      new_position[c] = intersection[c] + reflection[c] * (remaining * p->damping);
   }
}

void oracle_boundary(const sph_oracle_params* p, const float* position, float* new_velocity,
                     float time_step, float* new_position)
{
   const float maxv[3] = {p->max_x, p->max_y, p->max_z};
   for (int axis = 0; axis < 3; axis++) { /* x, then y, then z; `else if` per axis */
      float normal[3] = {0.0f, 0.0f, 0.0f};
      if (new_position[axis] < 0.0f) {
         normal[axis] = 1.0f;
         float dist = -position[axis] / new_velocity[axis];
         apply_boundary(p, position, time_step, new_position, dist, normal, new_velocity);
      } else if (new_position[axis] > maxv[axis]) {
         normal[axis] = -1.0f;
         float dist = (maxv[axis] - position[axis]) / new_velocity[axis];
         apply_boundary(p, position, time_step, new_position, dist, normal, new_velocity);
      }
   }
}

/* ------------------------------------------------------------------------------------
 * A6 — integrate.  reference src/sph.cpp:937-1022 */
void oracle_integrate(const sph_oracle_params* p, int n, float* pos, float* vel, const float* acc,
                      const float* mass, float* ke_out, float* pe_out)
{
   float ke = 0.0f, pe = 0.0f;
   const float dt = p->time_step;
   const float pos_dt = dt * p->sim_scale_inv;
   for (int i = 0; i < n; i++) {
      float vh[3], np[3], rs[3], nv[3];
      for (int c = 0; c < 3; c++) vh[c] = vel[3 * i + c] + (acc[3 * i + c] * dt * 0.5f);
      for (int c = 0; c < 3; c++) np[c] = pos[3 * i + c] + (vh[c] * pos_dt);
      for (int c = 0; c < 3; c++) rs[c] = (np[c] - p->central_pos[c]) * p->sim_scale;
      float dot = rs[0] * rs[0] + rs[1] * rs[1] + rs[2] * rs[2];
      dot = sqrtf(dot);
      float d3 = (dot + p->softening) * (dot + p->softening) * (dot + p->softening);
      for (int c = 0; c < 3; c++) {
         float a = -p->grav_const * p->central_mass * (rs[c] / d3);
         if (p->apply_gravity) a += p->gravity[c]; /* extension, see accel_end */
         nv[c] = vh[c] + (a * dt);
      }
      if (p->apply_walls) /* extension: the reference's own (unwired) wall handling */
         oracle_boundary(p, pos + 3 * i, nv, dt, np);
      dot = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
      if (dot > 0) {
         ke += 0.5f * mass[i] * dot;
         pe -= p->grav_const * p->central_mass * mass[i] / d3;
      }
      for (int c = 0; c < 3; c++) {
         pos[3 * i + c] = np[c];
         vel[3 * i + c] = nv[c];
      }
   }
   *ke_out = ke;
   *pe_out = pe;
}

/* A8 — one REF-mode step in the order of reference src/sph.cpp:208-289 */
void oracle_step_ref(const sph_oracle_params* p, int n, float* pos, float* vel, const float* mass,
                     float* rho, float* acc, int32_t* counts, float* ke, float* pe)
{
   const int ncells = p->cells_x * p->cells_y * p->cells_z;
   const int cap = p->examine_count;
   int32_t* coords = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)n);
   int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
   int32_t* cs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncells + 1));
   int32_t* ci = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
   uint32_t* nb = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n * cap);
   float* nd = (float*)malloc(sizeof(float) * (size_t)n * cap);
   oracle_voxelize(p, n, pos, coords, ids, cs, ci);
   oracle_find_neighbors(p, n, pos, coords, cs, ci, nb, nd, counts);
   oracle_density_lists(p, n, cap, nb, nd, counts, mass, rho);
   oracle_accel_lists(p, n, cap, nb, nd, counts, pos, vel, mass, rho, acc);
   oracle_integrate(p, n, pos, vel, acc, mass, ke, pe);
   free(coords); free(ids); free(cs); free(ci); free(nb); free(nd);
}

/* ====================================================================================
 * FULL mode: complete in-radius neighbourhoods on a cell-edge>=h grid.
 * No counterpart in the reference's search; the per-pair arithmetic is A4/A5 above and
 * the acceptance test is the reference's (dx*dx + dy*dy + dz*dz < mH2, src/sph.cpp:641,653).
 */
void oracle_full_cells(const sph_oracle_params* p, int n, const float* pos, int32_t* ids,
                       int32_t* cell_start, int32_t* cell_items)
{
   for (int i = 0; i < n; i++) {
      int cx = clampi((int)floor((double)(pos[3 * i + 0] * p->full_cell_inv)), p->full_cells_x);
      int cy = clampi((int)floor((double)(pos[3 * i + 1] * p->full_cell_inv)), p->full_cells_y);
      int cz = clampi((int)floor((double)(pos[3 * i + 2] * p->full_cell_inv)), p->full_cells_z);
      ids[i] = (cz * p->full_cells_y + cy) * p->full_cells_x + cx;
   }
   build_csr(n, p->full_cells_x * p->full_cells_y * p->full_cells_z, ids, cell_start, cell_items);
}

/* Visit i's neighbours in canonical order and call body(q, d) for each accepted one. */
#define FULL_FOR_EACH_NEIGHBOR(BODY)                                                        \
   do {                                                                                     \
      const float px = pos[3 * i + 0], py = pos[3 * i + 1], pz = pos[3 * i + 2];            \
      const int cx = clampi((int)floor((double)(px * p->full_cell_inv)), p->full_cells_x);  \
      const int cy = clampi((int)floor((double)(py * p->full_cell_inv)), p->full_cells_y);  \
      const int cz = clampi((int)floor((double)(pz * p->full_cell_inv)), p->full_cells_z);  \
      for (int dz = -1; dz <= 1; dz++) {                                                    \
         int z = cz + dz;                                                                   \
         if (z < 0 || z >= p->full_cells_z) continue;                                       \
         for (int dy = -1; dy <= 1; dy++) {                                                 \
            int y = cy + dy;                                                                \
            if (y < 0 || y >= p->full_cells_y) continue;                                    \
            int x0 = cx - 1 < 0 ? 0 : cx - 1;                                               \
            int x1 = cx + 1 >= p->full_cells_x ? p->full_cells_x - 1 : cx + 1;              \
            int row = (z * p->full_cells_y + y) * p->full_cells_x;                          \
            /* the 3 cells x0..x1 are contiguous in the CSR */                              \
            for (int s_ = cell_start[row + x0]; s_ < cell_start[row + x1 + 1]; s_++) {         \
               const int q = cell_items[s_];                                                \
               if (q == i) continue;                                                        \
               float dx_ = px - pos[3 * q + 0];                                             \
               float dy_ = py - pos[3 * q + 1];                                             \
               float dz_ = pz - pos[3 * q + 2];                                             \
               float dot = dx_ * dx_ + dy_ * dy_ + dz_ * dz_;                               \
               if (dot < p->h2) {                                                           \
                  const float d = sqrtf(dot) * p->sim_scale;                                \
                  BODY                                                                      \
               }                                                                            \
            }                                                                               \
         }                                                                                  \
      }                                                                                     \
   } while (0)

int oracle_full_build_lists(const sph_oracle_params* p, int n, const float* pos, int cap,
                            uint32_t* neighbors, float* dists, int32_t* counts)
{
   const int ncells = p->full_cells_x * p->full_cells_y * p->full_cells_z;
   int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
   int32_t* cell_start = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncells + 1));
   int32_t* cell_items = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
   oracle_full_cells(p, n, pos, ids, cell_start, cell_items);
   int worst = 0;
   for (int i = 0; i < n; i++) {
      int count = 0;
      FULL_FOR_EACH_NEIGHBOR({
         if (count < cap) {
            neighbors[(size_t)i * cap + count] = (uint32_t)q;
            dists[(size_t)i * cap + count] = d;
         }
         count++;
      });
      counts[i] = count;
      if (count > worst) worst = count;
   }
   free(ids); free(cell_start); free(cell_items);
   return worst;
}

void oracle_full_density(const sph_oracle_params* p, int n, const float* pos, const float* mass,
                         const int32_t* cell_start, const int32_t* cell_items, float* rho,
                         int32_t* counts)
{
   for (int i = 0; i < n; i++) {
      float density = 0.0f;
      int count = 0;
      FULL_FOR_EACH_NEIGHBOR({
         if (!(d > p->hscaled)) density += density_term(p, mass[q], d);
         count++;
      });
      rho[i] = density;
      if (counts) counts[i] = count;
   }
}

void oracle_full_accel(const sph_oracle_params* p, int n, const float* pos, const float* vel,
                       const float* mass, const float* rho, const int32_t* cell_start,
                       const int32_t* cell_items, float* acc)
{
   for (int i = 0; i < n; i++) {
      accel_state s;
      accel_begin(p, &s, pos + 3 * i, vel + 3 * i, rho[i]);
      FULL_FOR_EACH_NEIGHBOR({
         accel_pair(p, &s, pos + 3 * (size_t)q, vel + 3 * (size_t)q, mass[q], rho[q], d);
      });
      accel_end(p, &s, acc + 3 * i);
   }
}

/* How large the terms are that a particle's acceleration is the sum of (test infrastructure for
 * the tolerance-mode kernels; no counterpart in the reference).  computeAcceleration adds and
 * subtracts ~30 pair terms of either sign per particle (src/sph.cpp:866-882); where they cancel,
 * ANY evaluation that is not bit-for-bit the reference's - its own -ffast-math build included -
 * differs from it by rounding errors of the TERMS, not of the small sum.  scale[i] = sum over the
 * neighbours of |pressure term| + the viscous terms' magnitudes carried through the same in-loop
 * rescale + |point-mass gravity| (+ |uniform gravity|), in double: what a per-particle relative
 * error has to be read against. */
void oracle_full_accel_scale(const sph_oracle_params* p, int n, const float* pos, const float* vel,
                             const float* mass, const float* rho, const int32_t* cell_start,
                             const int32_t* cell_items, double* scale)
{
   for (int i = 0; i < n; i++) {
      accel_state s;
      accel_begin(p, &s, pos + 3 * i, vel + 3 * i, rho[i]);
      double pg = 0.0, vt = 0.0;
      FULL_FOR_EACH_NEIGHBOR({
         const float mj = mass[q];
         const float rhoj = rho[q];
         const double pj = (double)((rhoj - p->rho0) * p->stiffness);
         const double rhoj_inv = (rhoj > 0.0f) ? 1.0 / (double)rhoj : 1.0;
         const double r = sqrt((double)dot) * (double)p->sim_scale;
         const double hd = (double)p->hscaled - (double)d;
         pg += fabs((double)p->kernel2 * r / ((double)d + 0.01) * hd * hd * (double)mj *
                    (double)s.pi_div_rhoi2 * pj * rhoj_inv * rhoj_inv);
         double dv = 0.0;
         for (int c = 0; c < 3; c++) {
            const double t = (double)vel[3 * (size_t)q + c] - (double)s.vi[c];
            dv += t * t;
         }
         vt += sqrt(dv) * fabs(hd * rhoj_inv * (double)mj * (double)p->kernel3);
         vt *= fabs((double)p->viscosity * (double)s.rhoi_inv);
      });
      double g = 0.0;
      for (int c = 0; c < 3; c++) {
         const double t = ((double)s.r[c] - (double)p->central_pos[c]) * (double)p->sim_scale;
         g += t * t;
      }
      g = sqrt(g);
      const double d3 = (g + (double)p->softening) * (g + (double)p->softening) * (g + (double)p->softening);
      double t = pg + vt + fabs((double)p->grav_const * (double)p->central_mass) * g / d3;
      if (p->apply_gravity)
         t += sqrt((double)p->gravity[0] * p->gravity[0] + (double)p->gravity[1] * p->gravity[1] +
                   (double)p->gravity[2] * p->gravity[2]);
      scale[i] = t;
   }
}

void oracle_step_full(const sph_oracle_params* p, int n, float* pos, float* vel, const float* mass,
                      float* rho, float* acc, int32_t* counts, float* ke, float* pe)
{
   const int ncells = p->full_cells_x * p->full_cells_y * p->full_cells_z;
   int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
   int32_t* cell_start = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncells + 1));
   int32_t* cell_items = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
   oracle_full_cells(p, n, pos, ids, cell_start, cell_items);
   oracle_full_density(p, n, pos, mass, cell_start, cell_items, rho, counts);
   oracle_full_accel(p, n, pos, vel, mass, rho, cell_start, cell_items, acc);
   oracle_integrate(p, n, pos, vel, acc, mass, ke, pe);
   free(ids); free(cell_start); free(cell_items);
}
