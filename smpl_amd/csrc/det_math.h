// smpl_amd/csrc/det_math.h -- deterministic double arithmetic shared by the host
// compiler (model_compile.cpp, g++) and the gfx950 kernels (hipcc).
//
// Arithmetic contract: every value that decides a
// discrete outcome on the path (a cell index, a waypoint count, a coordinate, a
// validity bit) is produced by the same sequence of IEEE-754 binary64
// +,-,*,/ and sqrt operations on the host and on the device.  Both compilers run
// with -ffp-contract=off, so no multiply-add is ever fused, and sin/cos are the
// fixed polynomial below instead of libm / ocml (whose last bits differ).
#pragma once

#ifndef __HIPCC_RTC__   // hiprtc (per-robot specialisation, specialize.cpp) brings its own runtime declarations
#include <math.h>
#endif

#if defined(__HIPCC__)
#define SMPLX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SMPLX_HD inline
#endif

#define SMPLX_PI 3.14159265358979323846
#define SMPLX_2PI (2.0 * SMPLX_PI)

// sin(x), cos(x): 3-term Cody-Waite reduction by pi/2 + degree-13/14 minimax
// kernels on [-pi/4, pi/4]; < 1 ulp from correctly rounded for |x| < 1e5.
// Stands in for the reference's libm calls
// (sbpl_collision_checking/src/transform_functions.h:112-113,147-148,182-183;
//  smpl/src/graph/manip_lattice_action_space.cpp:591-594).
SMPLX_HD void smplx_sincos(double x, double* s_out, double* c_out)
{
    const double INVPIO2 = 6.36619772367581382433e-01;
    const double P1 = 0x1.921fb54400000p+0;
    const double P2 = 0x1.0b4611a600000p-34;
    const double P3 = 0x1.3198a2e000000p-69;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
#ifdef ABL_NO_SINCOS
    *s_out = x * 0.5; *c_out = 1.0 - x * 0.25; return;
#endif
    const double fn = rint(x * INVPIO2);
    const int n = (int)fn;
    const double r = ((x - fn * P1) - fn * P2) - fn * P3;
    const double z = r * r;
    const double v = z * r;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double ks = r + v * (S1 + z * rs);
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double kc = w + (((1.0 - w) - hz) + z * rc);
    const int k = n & 3;
    *s_out = (k == 0) ? ks : (k == 1) ? kc : (k == 2) ? -ks : -kc;
    *c_out = (k == 0) ? kc : (k == 1) ? -ks : (k == 2) ? -kc : ks;
}

// smpl/include/smpl/angles.h:45-62
SMPLX_HD double smplx_normalize_angle(double angle)
{
    if (fabs(angle) > SMPLX_2PI) angle = fmod(angle, SMPLX_2PI);
    if (angle < -SMPLX_PI) angle += SMPLX_2PI;
    if (angle > SMPLX_PI) angle -= SMPLX_2PI;
    return angle;
}

// smpl/include/smpl/angles.h:64-71
SMPLX_HD double smplx_normalize_angle_positive(double angle)
{
    angle = smplx_normalize_angle(angle);
    if (angle < 0.0) angle += SMPLX_2PI;
    return angle;
}

// smpl/include/smpl/angles.h:88-93
SMPLX_HD double smplx_shortest_angle_diff(double af, double ai) { return smplx_normalize_angle(af - ai); }
