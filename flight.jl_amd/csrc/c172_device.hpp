// c172_device.hpp — device-side physics of one Cessna172Sv0 in SimpleWorld, one aircraft per lane.
//
// Written for gfx950 (MI355X): fp64 VALU work, one thread = one aircraft, all per-aircraft state in
// registers, the small lookup tables (aero / piston / propeller, 25 KB) staged in LDS per workgroup,
// the EGM96 geoid (4.2 MB float32) gathered from global memory (L2 / Infinity-Cache resident).
//
// What it computes is the reference's f_ode!(world) / f_step!(world); how it computes it is laid out
// for the GPU (file:line = reference, relative to lib/):
//   * identity frame rotations of the C172 (every t_b* is a pure translation, FlightApps/src/c172/
//     c172.jl:32,455-471,514-518,628-629; c172s.jl:30) are elided — exact, x∘1 = x in floating point;
//   * half-angle sin/cos pairs are produced by one sincos();
//   * every rotation that involves a STATE quaternion keeps the reference's composition order: the
//     states q_wb, q_ew are only renormalised in f_step! (kinematics.jl:226-229), so at RK stages they
//     are off unit norm by ~1e-8, and the reference's rotate formula is not associative for non-unit
//     quaternions (measured: re-associating the gravity rotation moves v̇ by 5e-7 m/s²);
//   * ECEF->geodetic (Fukushima) is evaluated once per point, not once per consumer
//     (dynamics.jl:476,487 call it twice on the same Oc);
//   * total mass properties are accumulated as Σm, Σm r, ΣJ with one division instead of the
//     reference's chain of pairwise `+` (dynamics.jl:262-272), which divides at every addition.
// These change results at rounding level only (tests hold GPU vs CPU oracle to 1e-9 relative on ẋ).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "tables.h"
#include "../../include/flightbatch.h"

// the one libm entry point without a float overload in HIP's headers
__device__ __forceinline__ void fb_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void fb_sincos(float x, float* s, float* c) { sincosf(x, s, c); }

#define FBL(x) ((real)(x))
#define FB_NS fbd
#define FB_REAL double
#include "c172_device_impl.inc"
#undef FB_NS
#undef FB_REAL
#define FB_NS fbf
#define FB_REAL float
#include "c172_device_impl.inc"
#undef FB_NS
#undef FB_REAL
