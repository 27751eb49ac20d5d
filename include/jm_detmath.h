/*
 * jm_detmath.h — deterministic sin/cos/atan2 used by the box-geometry ops
 * (roipool3d point-in-box, rotated BEV overlap / NMS).
 *
 * Why this exists: the reference calls the toolchain's `cos/sin/atan2` float overloads
 * inside its kernels (jmodt/ops/roipool3d/src/roipool3d_kernel.cu:22,
 * jmodt/ops/iou3d/src/iou3d_kernel.cu:55,105,141-142).  CUDA libdevice, ROCm ocml and glibc
 * disagree in the last ulp, which flips in-box flags / NMS keep bits for inputs that sit
 * on a boundary.  Both the HIP kernels and the CPU oracle therefore evaluate the three
 * functions through the SAME arithmetic below: double-precision Cody–Waite reduction +
 * fixed polynomials, IEEE add/mul/div only (no FMA contraction), rounded once to float.
 * The float result is within 1 ulp of the correctly rounded value (tests/test_detmath.py
 * pins it against libm), i.e. inside the error band of every libm the reference could have
 * been built with, and it is bit-identical between gcc-x86 and hipcc-gfx950.
 *
 * Plain C99 / HIP-C++ compatible, header only.  Valid for |angle| < ~1e5 rad (box headings).
 */
#ifndef JM_DETMATH_H
#define JM_DETMATH_H

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define JM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define JM_HD static inline
#endif

#if defined(__clang__)
#define JM_NOCONTRACT _Pragma("clang fp contract(off)")
#else
#define JM_NOCONTRACT /* gcc: build with -ffp-contract=off */
#endif

/* round-half-even to integer: exact on both targets (x86 roundsd/libm rint in the default
   rounding mode, gfx950 v_rndne_f64) */
JM_HD double jm_rint(double x) { return __builtin_rint(x); }

/* sin and cos of `a` (radians), each rounded once to float. */
JM_HD void jm_sincosf(float a, float* s_out, float* c_out) {
    JM_NOCONTRACT
    const double x = (double)a;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double pio2_2 = 6.07710050630396597660e-11;  /* next 33 bits */
    const double pio2_3 = 2.02226624879595063154e-21;  /* remainder */
    const double kd = jm_rint(x * two_over_pi);
    double r = x - kd * pio2_1;
    r = r - kd * pio2_2;
    r = r - kd * pio2_3;
    const double z = r * r;
    /* Taylor on |r| <= pi/4: truncation < 1e-17 */
    double ps = -1.0 / 121645100408832000.0;            /* -1/19! */
    ps = ps * z + 1.0 / 355687428096000.0;              /*  1/17! */
    ps = ps * z - 1.0 / 1307674368000.0;                /* -1/15! */
    ps = ps * z + 1.0 / 6227020800.0;                   /*  1/13! */
    ps = ps * z - 1.0 / 39916800.0;                     /* -1/11! */
    ps = ps * z + 1.0 / 362880.0;                       /*  1/9!  */
    ps = ps * z - 1.0 / 5040.0;                         /* -1/7!  */
    ps = ps * z + 1.0 / 120.0;                          /*  1/5!  */
    ps = ps * z - 1.0 / 6.0;                            /* -1/3!  */
    const double sr = r + r * (z * ps);
    double pc = 1.0 / 2432902008176640000.0;            /*  1/20! */
    pc = pc * z - 1.0 / 6402373705728000.0;             /* -1/18! */
    pc = pc * z + 1.0 / 20922789888000.0;               /*  1/16! */
    pc = pc * z - 1.0 / 87178291200.0;                  /* -1/14! */
    pc = pc * z + 1.0 / 479001600.0;                    /*  1/12! */
    pc = pc * z - 1.0 / 3628800.0;                      /* -1/10! */
    pc = pc * z + 1.0 / 40320.0;                        /*  1/8!  */
    pc = pc * z - 1.0 / 720.0;                          /* -1/6!  */
    pc = pc * z + 1.0 / 24.0;                           /*  1/4!  */
    pc = pc * z - 0.5;                                  /* -1/2!  */
    const double cr = 1.0 + z * pc;
    /* quadrant = kd mod 4, computed in floating point (kd may exceed int range only for
       absurd angles, which are out of contract) */
    const long long ki = (long long)kd;
    const int q = (int)(ki & 3LL);
    double s, c;
    if (q == 0)      { s = sr;  c = cr;  }
    else if (q == 1) { s = cr;  c = -sr; }
    else if (q == 2) { s = -sr; c = -cr; }
    else             { s = -cr; c = sr;  }
    *s_out = (float)s;
    *c_out = (float)c;
}

/* atan2(y, x) rounded once to float; atan2(0,0) = 0 like libm. */
JM_HD float jm_atan2f(float yf, float xf) {
    JM_NOCONTRACT
    const double y = (double)yf, x = (double)xf;
    const double ax = x < 0.0 ? -x : x;
    const double ay = y < 0.0 ? -y : y;
    const double mx = ax > ay ? ax : ay;
    const double mn = ax > ay ? ay : ax;
    double r;
    if (mx == 0.0) {
        r = 0.0;
    } else if (mx != mx || mn != mn) {
        r = mx + mn; /* NaN in -> NaN out */
    } else {
        const double a = mn / mx; /* in [0,1] */
        /* atan(a) = pi/4 + atan((a-1)/(a+1)) for a > tan(pi/8) */
        double t, base;
        if (a > 0.41421356237309503) { t = (a - 1.0) / (a + 1.0); base = 7.85398163397448278999e-01; }
        else                         { t = a;                     base = 0.0; }
        const double z = t * t;
        /* odd Taylor to t^27, |t| <= 0.4143: truncation < 3e-13 */
        double p = 1.0 / 27.0;
        p = -1.0 / 25.0 + z * p;
        p =  1.0 / 23.0 + z * p;
        p = -1.0 / 21.0 + z * p;
        p =  1.0 / 19.0 + z * p;
        p = -1.0 / 17.0 + z * p;
        p =  1.0 / 15.0 + z * p;
        p = -1.0 / 13.0 + z * p;
        p =  1.0 / 11.0 + z * p;
        p = -1.0 / 9.0  + z * p;
        p =  1.0 / 7.0  + z * p;
        p = -1.0 / 5.0  + z * p;
        p =  1.0 / 3.0  + z * p;
        r = base + (t - t * (z * p));
        if (ay > ax) r = 1.57079632679489655800e+00 - r;      /* swap: pi/2 - r */
        if (x < 0.0) r = 3.14159265358979311600e+00 - r;      /* left half plane */
    }
    if (y < 0.0 || (y == 0.0 && (1.0 / (double)yf) < 0.0)) r = -r; /* sign of y incl. -0 */
    return (float)r;
}

#endif /* JM_DETMATH_H */
