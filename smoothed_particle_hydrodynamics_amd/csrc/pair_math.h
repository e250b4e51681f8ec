// Per-pair and per-particle arithmetic of the reference's sums, as device inlines.
//
// Every expression keeps the reference's association and rounding points so that results are
// bit-identical to the x86-64 IEEE build of src/sph.cpp (this translation unit is compiled with
// -ffp-contract=off; sqrtf and '/' are correctly rounded under hipcc's defaults).
#pragma once

#include "sph_device.h"

// computeDensity's inner term (reference src/sph.cpp:744-761).  d is the stored distance.
// INSIDE: the caller has established d <= hscaled, so the reference's range test (:746) cannot fail
// and is not evaluated - the FULL-mode sums with unit simulation scale: d = sqrt_rn(d2) of a pair
// that passed d2 < h2, the root is monotone and the host selects those instantiations only when
// sqrtf(h2) <= hscaled (unit_scale() in sph_hip.hip).  A compare and a select per neighbour less.
template <bool INSIDE = false>
__device__ __forceinline__ void density_accumulate(const PairConsts& k, float mass, float d,
                                                   float& density)
{
   if (INSIDE || !(d > k.hscaled)) {
      float t = (k.hscaled2 - (d * d));
      t = (t * t * t);
      const float w = k.kernel1 * t;
      density += (mass * w);
   }
}

// the same term as a value: m * (k1 * t^3), rounded as density_accumulate rounds it (callers that
// sum a batch of neighbours add the terms in list order; a neighbour that does not count adds +0.0f,
// which leaves a density - never negative, never -0 - unchanged bit for bit)
template <bool INSIDE = false>
__device__ __forceinline__ float density_term(const PairConsts& k, float mass, float d)
{
   float t = (k.hscaled2 - (d * d));
   t = (t * t * t);
   const float w = k.kernel1 * t;
   const float term = (mass * w);
   return (INSIDE || !(d > k.hscaled)) ? term : 0.0f;
}

// ---- tolerance mode (SPH_HIP_MODE_FULL_FAST) ----------------------------------------------------
// The reference ships with -O3 -ffast-math -funsafe-math-optimizations -mfma
// (reference CMakeLists.txt:21): its own binary is not the IEEE evaluation of src/sph.cpp - the
// survey measured 2.1e-5 relative on accelerations between the two builds - and the north star's
// bar is 1e-4 relative on forces.  The FAST variants keep what decides WHICH pairs are summed and
// in WHICH order (the exact unfused d2 < h2 membership test, the canonical order, the rescale of the
// viscous sum inside the neighbour loop) and evaluate the per-pair arithmetic the way such a build
// may - in the ACCELERATION sum only: fp32 reciprocal instead of the fp64 quotient of
// src/sph.cpp:854-856, fused accumulation, a viscous sum that leaves out terms below 1e-20.  The
// density sum stays the reference's arithmetic bit for bit: the pressure p = (rho - rho0) * k
// amplifies a relative error of rho by rho / (rho - rho0), and a first version that summed
// h^2 - d^2 without the square root (2e-6 off in rho) was 4e-4 off in the acceleration of
// particles next to one whose density is within a per cent of rho0.  Every fused operation is written out (the translation unit is compiled with
// -ffp-contract=off), so all routes - tiled, untiled, the walk of a particle without a list, any
// slab count - produce the same bits as each other.

// Quantities of neighbour j that computeAcceleration re-derives for every pair
// (reference src/sph.cpp:829-834, 860, 871).  They depend on j only, so they are computed once
// per particle after the density pass: B = p_j * rhojInv^2, C = (rhojInv * m_j) * k3.
__device__ __forceinline__ float2 neighbor_terms(const PairConsts& k, float rho_j, float m_j)
{
   const float pj = (rho_j - k.rho0) * k.stiffness;
   const float rhoj_inv = ((rho_j > 0.0f) ? (1.0f / rho_j) : 1.0f);
   const float rhoj_inv2 = rhoj_inv * rhoj_inv;
   return make_float2(pj * rhoj_inv2, rhoj_inv * m_j * k.kernel3);
}

// (float)((double)n / den) for three numerators over one denominator.
//
// hipcc expands a correctly rounded f64 division to: rcp, two Newton steps on the reciprocal,
// q = n*r, one residual step q' = fma(fma(-den, q, n), r, q) (plus v_div_scale / v_div_fixup,
// which only act on operands far outside the range seen here).  The reciprocal part depends on
// the denominator alone, so it is done once and each quotient keeps its own last two steps:
// the same operations on the same values, hence the same bits as three separate divisions.
// den is in [0.01, 0.01 + h]; a non-finite numerator (positions that already blew up) takes the
// plain division so that infinities propagate exactly like on the CPU.  in_range (uniform): the
// caller has established that no operand can leave the range (see accel_operands_in_range), so
// the per-pair checks are skipped.
__device__ __forceinline__ void div3_shared_den(float nx, float ny, float nz, double den,
                                                float& qx, float& qy, float& qz, bool in_range)
{
   if (in_range || (__builtin_isfinite(nx) && __builtin_isfinite(ny) && __builtin_isfinite(nz) &&
                    den >= 0.0078125 && den <= 1.0e6)) {
      double r = __builtin_amdgcn_rcp(den);
      double e = __builtin_fma(-den, r, 1.0);
      r = __builtin_fma(r, e, r);
      e = __builtin_fma(-den, r, 1.0);
      r = __builtin_fma(r, e, r);
      const double dx = (double)nx, dy = (double)ny, dz = (double)nz;
      double q = dx * r;
      qx = (float)__builtin_fma(__builtin_fma(-den, q, dx), r, q);
      q = dy * r;
      qy = (float)__builtin_fma(__builtin_fma(-den, q, dy), r, q);
      q = dz * r;
      qz = (float)__builtin_fma(__builtin_fma(-den, q, dz), r, q);
   } else {
      qx = (float)((double)nx / den);
      qy = (float)((double)ny / den);
      qz = (float)((double)nz / den);
   }
}

// FAST contexts: {m_j * B, C} (the mass rides in both factors)
__device__ __forceinline__ float2 neighbor_terms_fast(const PairConsts& k, float rho_j, float m_j)
{
   const float2 bc = neighbor_terms(k, rho_j, m_j);
   return make_float2(m_j * bc.x, bc.y);
}

struct AccelState {
   float rhoi_inv, pi_div_rhoi2, visc_scale;
   float k2s;   // FAST: kernel2 * sim_scale * 2^-shift (PairConsts::fast_k2s): what multiplies r / den in the pressure term
   float rx, ry, rz, vx, vy, vz;
   float pgx, pgy, pgz, vtx, vty, vtz;
};

// reference src/sph.cpp:785-798
__device__ __forceinline__ void accel_begin(const PairConsts& k, AccelState& s, float4 posm,
                                            float4 velp, float rho_i)
{
   const float pi = (rho_i - k.rho0) * k.stiffness;
   s.rhoi_inv = ((pi > 0.0f) ? (1.0f / pi) : 1.0f); // derived from the PRESSURE, as shipped
   const float rhoi_inv2 = s.rhoi_inv * s.rhoi_inv;
   s.pi_div_rhoi2 = pi * rhoi_inv2;
   s.visc_scale = k.viscosity * s.rhoi_inv;
   s.k2s = k.fast_k2s;
   s.rx = posm.x; s.ry = posm.y; s.rz = posm.z;
   s.vx = velp.x; s.vy = velp.y; s.vz = velp.z;
   s.pgx = s.pgy = s.pgz = 0.0f;
   s.vtx = s.vty = s.vtz = 0.0f;
}

// For a pair that passed the exact test d2 < h2, with d2 computed from (dx,dy,dz): |dx|, |dy|, |dz|
// and the distance are at most sqrt(h2); after scaling, the division's numerators are bounded by
// |kernel2| * reach and its denominator lies in [0.01, 0.01 + reach] - the per-pair range checks of
// div3_shared_den are decided by the constants alone.
__device__ __forceinline__ bool accel_operands_in_range(const PairConsts& k)
{
   const float reach = sqrtf(k.h2) * k.sim_scale * 1.01f;
   return k.sim_scale > 0.0f && __builtin_isfinite(reach) && reach <= 9.0e5f &&
          __builtin_isfinite(k.kernel2 * reach);
}

// One neighbour (reference src/sph.cpp:846-882).  (dx,dy,dz) = r_i - r_j, d = stored distance,
// B/C from neighbor_terms().  in_range: see accel_operands_in_range (false = check every pair).
template <bool UNIT_SCALE>
__device__ __forceinline__ void accel_pair(const PairConsts& k, AccelState& s, float dx, float dy,
                                           float dz, float d, float mj, float vjx, float vjy,
                                           float vjz, float B, float C, bool in_range = false)
{
   const float rsx = UNIT_SCALE ? dx : dx * k.sim_scale;
   const float rsy = UNIT_SCALE ? dy : dy * k.sim_scale;
   const float rsz = UNIT_SCALE ? dz : dz * k.sim_scale;
   // float product, double add, double divide, narrowed to float (:854-856)
   const double den = (double)d + 0.01;
   float gx, gy, gz;
   div3_shared_den(k.kernel2 * rsx, k.kernel2 * rsy, k.kernel2 * rsz, den, gx, gy, gz, in_range);

   float center = (k.hscaled - d);
   center *= center;
   center *= mj * s.pi_div_rhoi2 * B; // (m_j * A) * (p_j * rhojInv^2), :860
   s.pgx += gx * center;
   s.pgy += gy * center;
   s.pgz += gz * center;

   center = (k.hscaled - d);
   center *= C;
   s.vtx += (vjx - s.vx) * center;
   s.vty += (vjy - s.vy) * center;
   s.vtz += (vjz - s.vz) * center;
   // the rescale sits inside the neighbour loop (:880-882)
   s.vtx *= s.visc_scale;
   s.vty *= s.visc_scale;
   s.vtz *= s.visc_scale;
}

// One neighbour, tolerance mode (see the top of this file): same terms, same place of the viscous
// rescale; fp32 reciprocal for g = (k2 * r) / (d + 0.01) in place of the fp64 quotient, fused
// accumulation.  (dx,dy,dz) = r_i - r_j and d = the stored distance exactly as in accel_pair - the
// correctly rounded root of the unfused d2.  Both sums hang on h - d, a difference of nearly equal
// numbers for a neighbour near the rim of the kernel, where an ulp of d is a relative error of
// ulp * d / (h - d) in the pair's term - and a rim term is not a small term: the reference's
// B_j = p_j / rho_j^2 of a neighbour j whose own density is next to nothing (a particle of the
// spray: its few neighbours all at its rim) grows like (h - d)^-6, so that the pair's pressure term
// grows like (h - d)^-4 towards the rim and can be the whole force on i.  A version with the
// hardware root of the fused d2 in the pressure sum (13 instructions fewer per neighbour, 25 us
// at 4M) was within 8e-6 on every committed scene and failed seeded random scene 219 - one
// neighbour at d = 0.998 h_scaled carrying 6809 of a force of 6811 - by 1.09e-4.  A tolerance that
// holds for any scene needs the reference's own h - d, bit for bit.
// The two sums are separate functions because the tiled kernel runs them as separate loops: the
// pressure sum over the whole list (everything it needs of a neighbour is in the tile), the
// viscous sum over the list's last visc_keep() entries (the only gathers of the pass).  They
// accumulate into different registers and meet in accel_end, so a route that evaluates both per
// neighbour in one loop produces the same bits.
// Bm = m_j * B: the density pass of a FAST context stores the product (neighbor_terms_fast), so no
// route of the acceleration pass gathers masses.
template <bool UNIT_SCALE>
__device__ __forceinline__ void accel_pair_fast_pressure(const PairConsts& k, AccelState& s, float dx,
                                                         float dy, float dz, float d, float Bm)
{
   const float hd = k.hscaled - d;
   // pressure: (k2 * r / den) * ((h - d)^2 * ((m_j * A) * B)).  Where the numbers stop being finite
   // the grouping matters: B of a neighbour with next to no density is ~1e29.  A first version
   // multiplied (k2 A) by (m B) before (h - d)^2, overflowed five orders of magnitude before the
   // reference does and met 0 * inf where the reference's clamp of an overflowed |a|^2 returns zeros
   // (seeded random scene 1751 of the soak).  Now: c = (h - d)^2 * (A * (m B)) as the reference
   // forms it - it overflows when the reference's does - and the scalar k2 / den it is multiplied
   // with carries a power of two (2^-shift, taken from the exponent of k2 * sim_scale so that the
   // scaled constant is a normal number of magnitude ~2^-7 for any h and sim_scale: PairConsts::fast_k2s)
   // that accel_fast_finish() takes out of the finished sum: the
   // per-pair factor f cannot overflow unless c has, whatever the pair's r is, and a sum the
   // reference overflows overflows here when the factor is taken out (5 multiplications and 3 fused
   // multiply-adds per neighbour; the reference's own grouping, g = k2 r / den per component, costs 7 + 3
   // and measured 22 us more at 4M).
   const float c = (hd * hd) * (s.pi_div_rhoi2 * Bm);
   const float f = c * (s.k2s * __builtin_amdgcn_rcpf(d + 0.01f));
   s.pgx = __builtin_fmaf(dx, f, s.pgx);
   s.pgy = __builtin_fmaf(dy, f, s.pgy);
   s.pgz = __builtin_fmaf(dz, f, s.pgz);
}

// the pressure sum of a FAST context is accumulated times 2^-shift (accel_pair_fast_pressure): before accel_end
__device__ __forceinline__ void accel_fast_finish(const PairConsts& k, AccelState& s)
{
   s.pgx *= k.fast_unscale;
   s.pgy *= k.fast_unscale;
   s.pgz *= k.fast_unscale;
}

// viscosity, rescaled inside the neighbour loop (:880-882)
__device__ __forceinline__ void accel_pair_fast_viscous(const PairConsts& k, AccelState& s, float d,
                                                        float vjx, float vjy, float vjz, float C)
{
   const float c2 = (k.hscaled - d) * C;
   s.vtx = __builtin_fmaf(vjx - s.vx, c2, s.vtx) * s.visc_scale;
   s.vty = __builtin_fmaf(vjy - s.vy, c2, s.vty) * s.visc_scale;
   s.vtz = __builtin_fmaf(vjz - s.vz, c2, s.vtz) * s.visc_scale;
}

// How many of a particle's LAST neighbours the viscous sum of a FAST context has to visit.  The
// reference rescales the running sum by s = mu * rhoiInv after every neighbour (:880-882), so the
// neighbour that is m-th from the end enters with the weight s^m: with the reference's constants
// |s| is of the order 1e-7 in a fluid under pressure and 1e-2 (= mu) where the pressure is not
// positive, and all but the last few terms are below anything fp32 resolves.  A term is left out
// when its weight is below 1e-20 (2^-66.4): the last ceil(66.4 / -log2 |s|) neighbours are kept, all
// of them when |s| >= 1/2.  What is left out is at most 1e-20 of the largest viscous term of the
// particle - against a tolerance of 1e-4 on the acceleration - and the neighbours that are left
// out need neither their velocity nor their C: the only per-neighbour gather of the acceleration
// pass shrinks from ~31 to ~4 per particle (it was the pass's floor: profiles/design_history_r1_r3.md 3.2).
__device__ __forceinline__ int visc_keep(float visc_scale)
{
   const float a = __builtin_fabsf(visc_scale);
   if (!(a < 0.5f)) return 0x7fffffff;            // (NaN too)
   if (a < 1.0e-30f) return 1;
   const float m = __builtin_ceilf(66.4386f / -__builtin_amdgcn_logf(a));
   return m < 1.0f ? 1 : (int)m;
}

// reference src/sph.cpp:888-933.  k.skip_point_mass (tolerance mode only, set by the host): a scene
// without a point mass (the dam-break: central_mass = 0, so every term of the block is +-0) skips
// the term's square root and three divisions behind a uniform branch; the exact mode evaluates it
// always (x + -0 keeps a -0 that x + +0 does not).
template <bool UNIT_SCALE>
__device__ __forceinline__ float4 accel_end(const PairConsts& k, const AccelState& s)
{
   float ax = s.vtx - s.pgx;
   float ay = s.vty - s.pgy;
   float az = s.vtz - s.pgz;

   if (!k.skip_point_mass) {
   float rsx = (s.rx - k.cx), rsy = (s.ry - k.cy), rsz = (s.rz - k.cz);
   if (!UNIT_SCALE) {
      rsx *= k.sim_scale;
      rsy *= k.sim_scale;
      rsz *= k.sim_scale;
   }
   float dot = (rsx * rsx) + (rsy * rsy) + (rsz * rsz);
   dot = sqrtf(dot);
   const float ds = dot + k.softening;
   const float d3 = ds * ds * ds;
   const float gm = -k.grav_const * k.central_mass;
   ax += gm * (rsx / d3);
   ay += gm * (rsy / d3);
   az += gm * (rsz / d3);
   } else {
      // (... +-0 for a particle whose position is finite.  One that is not - a particle the reference has
      // lost to a NaN two steps ago, seeded random scene 594 of the round-4 soak - gets NaN from the term,
      // in all three components: x - x is 0 or NaN exactly when the term is.)
      const float lost = (s.rx - s.rx) + (s.ry - s.ry) + (s.rz - s.rz);
      ax += lost;
      ay += lost;
      az += lost;
   }
   if (k.apply_gravity) { // extension: uniform gravity enters next to the point-mass term
      ax += k.gx;
      ay += k.gy;
      az += k.gz;
   }

   const float dot = (ax * ax) + (ay * ay) + (az * az);
   if (dot > k.cfl_limit2) {
      const float length = sqrtf(dot);
      const float scale = k.cfl_limit / length;
      ax *= scale;
      ay *= scale;
      az *= scale;
   }
   return make_float4(ax, ay, az, 0.0f);
}
