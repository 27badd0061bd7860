/*
 * CPU restatement of the reference FastS explicit time step (plain C, scalar, one thread).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under fast_amd/ links,
 * loads or calls it.  It is pinned against the compiled reference Fortran
 * (oracle/_ref/libfast_ref.so, built by oracle/build_ref.py) in tests/test_oracle_vs_ref.py
 * and against the fixtures under tests/golden/ that were generated from that library.
 *
 * Scope: zone types 0 (curvilinear), 1 (k-homogeneous), 2 (Cartesian, constant step) and 3 (two-dimensional);
 * 5 equations or 6 (Spalart-Allmaras); schemes ausmpred (1) / senseur (2) / roe (5), slope o3 (roe also o1 and
 * minmod) or, with 5 equations, o3sc (the o3 states weighted by a shock sensor); explicit RK1/2/3 with a global or a local time step; implicit BDF with LU-SGS, LU-SSOR, the overlap
 * sweeps of lu_match and selective frequency damping; ghost fill for the boundary conditions cited at fo_bc_*,
 * roe wall-flux correction,
 * edge/corner fix.  Array layout, loop ranges and
 * the order of floating-point operations follow the reference routines cited at each
 * function, so that with one cache block the results agree with the reference to rounding.
 *
 * Indices into Parameter_int / Parameter_real are those of
 * Fast/FastS/FastS/param_solver.h:77-306.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NIJK 0
#define IJKV 20
#define ITYPZONE 25
/* zone types 0 (curvilinear) and 1 (k-homogeneous, an extruded mesh): metric arrays with three components per
 * direction; the caller pads those of a type-1 zone with zeros and gives them a zero stride along k.  The reference's
 * *_3dhomo_* fragments are the general ones without the terms the missing components multiply (the same bits),
 * except for the z derivatives of the viscous flux and of the Spalart-Allmaras source: see `homo` below. */
#define FULL3D(t) ((t) == 0 || (t) == 1)
#define IFLOW 27
#define ITYPCP 29
#define KFLUDOM 33
#define NEQ 36
#define NDIMDX 41
#define RK 52
#define NSSITER 58
#define SLOPE 64
#define PT_BC 70
#define SENSORTYPE 85

#define DTC 0
#define GAMMA 1
#define CVINF 2
#define ROINF 3
#define VINF 5
#define TINF 6
#define PRANDT 10
#define XMUL0 11
#define TEMP0 12
#define CS 13
#define PSIROE 34

typedef int32_t i32;

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dsign(double a, double b) { return copysign(fabs(a), b); }

/* 0-based linear index of cell (i,j,k), interior starting at 1
 * (Fast/FastS/FastS/formule_param.h:4-7). */
#define INDDM(i, j, k) ((long)((i) + ific - 1) + (long)((j) + ific - 1) * ni + (long)((k) + ifick - 1) * ninj)

#define GEOM(p)                                                                      \
  const long ni = (p)[NIJK], nj = (p)[NIJK + 1];                                     \
  const long ninj = ni * nj;                                                         \
  const int ific = (p)[NIJK + 3], ifick = (p)[NIJK + 4];                             \
  const long nd = (p)[NDIMDX];                                                       \
  (void)nj; (void)nd; (void)ifick; (void)ific; (void)ninj;

/* Sutherland viscosity, Fast/FastS/FastS/Compute/invist.for:49-61 */
void fo_invist(const i32 *p, const double *pr, const i32 *ind, const double *rop, double *xmut)
{
  GEOM(p)
  const double cmus1 = pr[CS];
  const double temp01 = 1. / pr[TEMP0];
  const double coesut = pr[XMUL0] * (1. + cmus1 * temp01) * sqrt(temp01);
  const double *T = rop + 4 * nd;
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        long l = INDDM(i, j, k);
        double t1 = T[l];
        xmut[l] = coesut * sqrt(t1) * t1 / (t1 + cmus1);
      }
}

/* drodm = 0 at the first stage, drodm *= xi afterwards
 * (Fast/FastS/FastS/Compute/init_rhs.for:31-32,58-129) */
void fo_init_rhs(const i32 *p, int nitcfg, const i32 *ind, double *drodm)
{
  GEOM(p)
  const double cxsi1 = -0.6830127018922193, cxsi2 = -4. / 3.;
  double coefH = 0.;
  int b = p[RK];
  if (b == 3) coefH = nitcfg == 2 ? cxsi1 : (nitcfg == 3 ? cxsi2 : 0.);
  else if (b == 2) coefH = nitcfg == 2 ? -1. : 0.;
  int zero = (p[ITYPCP] <= 1 || nitcfg == 1);
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        long l = INDDM(i, j, k);
        for (int e = 0; e < p[NEQ]; e++) drodm[l + e * nd] = zero ? 0. : drodm[l + e * nd] * coefH;
      }
}

/* one sensor value: 1 when some variable shows two consecutive slope sign changes
 * (Fast/FastS/FastS/Compute/SENSOR/wiggle.for:1-31, wiggle_i.for:3-8) */
static inline double wiggle1(const double *rop, long nd, long l, long inc)
{
  const double souszero = -1e-12;
  long nm = l - inc, nm2 = l - 2 * inc, np = l + inc;
  double test = 1e30;
  for (int e = 0; e < 5; e++) {
    const double *q = rop + e * nd;
    double qm = (q[nm] - q[nm2]) * (q[l] - q[nm]);
    double qp = (q[l] - q[nm]) * (q[np] - q[l]);
    double f = dsign(1., qm) + dsign(1., qp);
    test = e == 0 ? f : dmin(test, f);
  }
  return test < souszero ? 1. : 0.;
}

/* wiggle sensor on all faces of the range, incl. the max-side faces
 * (Fast/FastS/FastS/Compute/SENSOR/wiggle2001.for:50-142) */
void fo_wiggle(const i32 *p, const i32 *ind, const double *rop, double *wig)
{
  GEOM(p)
  const long inci = 1, incj = ni, inck = ninj;
  const int twod = p[ITYPZONE] == 3;   /* wiggle2001.for:61-93: i and j faces only */
  int icorr = ind[1] == p[IJKV], jcorr = ind[3] == p[IJKV + 1], kcorr = ind[5] == p[IJKV + 2] && !twod;
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        long l = INDDM(i, j, k);
        if (!twod) wig[l + 2 * nd] = wiggle1(rop, nd, l, inck);
        wig[l + nd] = wiggle1(rop, nd, l, incj);
        wig[l] = wiggle1(rop, nd, l, inci);
      }
  if (icorr)
    for (int k = ind[4]; k <= ind[5]; k++)
      for (int j = ind[2]; j <= ind[3]; j++) {
        long l = INDDM(ind[1] + 1, j, k);
        wig[l] = wiggle1(rop, nd, l, inci);
      }
  if (jcorr)
    for (int k = ind[4]; k <= ind[5]; k++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        long l = INDDM(i, ind[3] + 1, k);
        wig[l + nd] = wiggle1(rop, nd, l, incj);
      }
  if (kcorr)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        long l = INDDM(i, j, ind[5] + 1);
        wig[l + 2 * nd] = wiggle1(rop, nd, l, inck);
      }
}

/* Directional shock sensor of the o3sc slope (Fast/FastS/FastS/Compute/SENSOR/sciacovelli.for, shock_sensor.for; velocity
 * divergence and squared vorticity from SA/div_rot_{3dfull,3dhomo,3dcart,2d}.for, face sizes from SA/metric_*.for):
 * for every cell of the range and, on the max sides of the zone, the plane of ghost cells whose lower face is the
 * boundary face, wig(l, dir) = 1 - min(2 div^2 / (div^2 + rot + 1e-15) * |p- - 2 p + p+| / |p- + 2 p + p+| *
 * 0.5 (1 - tanh(2.5 + 10 div / (tc a))), 1), written after the three wiggle arrays when the scheme is senseur.
 * ti / tj / tk of a k-homogeneous zone padded to three components as for the fluxes. */
#define NIJK_MTR_ 5
#define NDIMDX_MTR_ 43
static void shock_cell(const i32 *p, const double *pr, const double *rop, const double *ti, const double *tj,
                       const double *tk, const double *vol, double *wsc, int i, int j, int k, int dirs, const int *mcell)
{
  const long ni = p[NIJK], nj = p[NIJK + 1], nd = p[NDIMDX];
  const int ific = p[NIJK + 3], ifick = p[NIJK + 4];
  const long inc[3] = {1, ni, ni * nj};
  const long l = (long)(i + ific - 1) + (long)(j + ific - 1) * ni + (long)(k + ifick - 1) * ni * nj;
  const int typ = p[ITYPZONE];
  const long l1 = l + 1, l2 = l - 1, l3 = l + inc[1], l4 = l - inc[1], l5 = l + inc[2], l6 = l - inc[2];
  const double gam = pr[GAMMA], rgp = pr[CVINF] * (gam - 1.);
  const double *U = rop + nd, *V = rop + 2 * nd, *W = rop + 3 * nd;
  double div, rot, tc[3];
  if (typ == 2) {
    const double tix = ti[0], tjy = tj[0], tkz = tk[0];
    tc[0] = fabs(tix); tc[1] = fabs(tjy); tc[2] = fabs(tkz);
    const double xvol = 0.5 / vol[0];
    double u1, u2, u3, u4, u5, u6, dudx, rotx, roty, rotz;
    u1 = U[l] + U[l2]; u2 = U[l] + U[l3]; u3 = U[l] + U[l1]; u4 = U[l] + U[l4]; u5 = U[l] + U[l5]; u6 = U[l] + U[l6];
    dudx = (u3 - u1) * tix; rotz = -(u2 - u4) * tjy; roty = (u5 - u6) * tkz; div = dudx * xvol;
    u1 = V[l] + V[l2]; u2 = V[l] + V[l3]; u3 = V[l] + V[l1]; u4 = V[l] + V[l4]; u5 = V[l] + V[l5]; u6 = V[l] + V[l6];
    dudx = (u3 - u1) * tix; rotz = (rotz + dudx) * xvol;
    dudx = (u2 - u4) * tjy; div = div + dudx * xvol;
    dudx = (u5 - u6) * tkz; rotx = -dudx;
    u1 = W[l] + W[l2]; u2 = W[l] + W[l3]; u3 = W[l] + W[l1]; u4 = W[l] + W[l4]; u5 = W[l] + W[l5]; u6 = W[l] + W[l6];
    dudx = (u3 - u1) * tix; roty = (roty - dudx) * xvol;
    dudx = (u2 - u4) * tjy; rotx = (rotx + dudx) * xvol;
    dudx = (u5 - u6) * tkz; div = div + dudx * xvol;
    rot = rotx * rotx + roty * roty + rotz * rotz;
  } else {
    const long ndm = p[NDIMDX_MTR_];
    const long minc[3] = {p[NIJK_MTR_], p[NIJK_MTR_ + 1], p[NIJK_MTR_ + 2]};
    /* mcell: the cell whose metrics are used (the ghost planes take those of the last cell of the range, see below) */
    const long lt = (long)(mcell[0] + p[NIJK_MTR_ + 3] - 1) * minc[0] + (long)(mcell[1] + p[NIJK_MTR_ + 3] - 1) * minc[1] +
                    (long)(mcell[2] + p[NIJK_MTR_ + 4] - 1) * minc[2];
    const int twod = typ == 3, homo = typ == 1;
    const double tix = ti[lt], tiy = ti[lt + ndm], tiz = (twod || homo) ? 0. : ti[lt + 2 * ndm];
    const double tix1 = ti[lt + minc[0]], tiy1 = ti[lt + minc[0] + ndm], tiz1 = (twod || homo) ? 0. : ti[lt + minc[0] + 2 * ndm];
    const double tjx = tj[lt], tjy = tj[lt + ndm], tjz = (twod || homo) ? 0. : tj[lt + 2 * ndm];
    const double tjx1 = tj[lt + minc[1]], tjy1 = tj[lt + minc[1] + ndm], tjz1 = (twod || homo) ? 0. : tj[lt + minc[1] + 2 * ndm];
    double tkx = 0., tky = 0., tkz = 0., tkx1 = 0., tky1 = 0., tkz1 = 0.;
    if (homo) tkz = tk[lt + 2 * ndm];
    else if (!twod) {
      tkx = tk[lt]; tky = tk[lt + ndm]; tkz = tk[lt + 2 * ndm];
      tkx1 = tk[lt + minc[2]]; tky1 = tk[lt + minc[2] + ndm]; tkz1 = tk[lt + minc[2] + 2 * ndm];
    }
    double t;
    t = sqrt(tix * tix + tiy * tiy + tiz * tiz); t = t + sqrt(tix1 * tix1 + tiy1 * tiy1 + tiz1 * tiz1); tc[0] = dmax(.5 * t, 1.e-27);
    t = sqrt(tjx * tjx + tjy * tjy + tjz * tjz); t = t + sqrt(tjx1 * tjx1 + tjy1 * tjy1 + tjz1 * tjz1); tc[1] = dmax(.5 * t, 1.e-27);
    if (homo) tc[2] = dmax(fabs(tkz), 1.e-27);
    else {
      t = sqrt(tkx * tkx + tky * tky + tkz * tkz); t = t + sqrt(tkx1 * tkx1 + tky1 * tky1 + tkz1 * tkz1); tc[2] = dmax(.5 * t, 1.e-27);
    }
    const double xvol = 0.5 / vol[lt];
    double u1, u2, u3, u4, u5 = 0., u6 = 0., dudx, rotx = 0., roty = 0., rotz;
    /* gradient component along m of a velocity: the general sum, or its two- / one-term forms */
#define GX (twod || homo ? (u3 * tix1 - u1 * tix + u2 * tjx1 - u4 * tjx) : (u3 * tix1 - u1 * tix + u2 * tjx1 - u4 * tjx + u5 * tkx1 - u6 * tkx))
#define GY (twod || homo ? (u3 * tiy1 - u1 * tiy + u2 * tjy1 - u4 * tjy) : (u3 * tiy1 - u1 * tiy + u2 * tjy1 - u4 * tjy + u5 * tky1 - u6 * tky))
#define GZ (homo ? (u5 - u6) * tkz : (u3 * tiz1 - u1 * tiz + u2 * tjz1 - u4 * tjz + u5 * tkz1 - u6 * tkz))
    u1 = U[l] + U[l2]; u2 = U[l] + U[l3]; u3 = U[l] + U[l1]; u4 = U[l] + U[l4];
    if (!twod) { u5 = U[l] + U[l5]; u6 = U[l] + U[l6]; }
    dudx = GX; div = dudx * xvol;
    rotz = -GY;
    if (!twod) roty = GZ;
    u1 = V[l] + V[l2]; u2 = V[l] + V[l3]; u3 = V[l] + V[l1]; u4 = V[l] + V[l4];
    if (!twod) { u5 = V[l] + V[l5]; u6 = V[l] + V[l6]; }
    dudx = GX; rotz = (rotz + dudx) * xvol;
    dudx = GY; div = div + dudx * xvol;
    if (!twod) {
      dudx = GZ; rotx = -dudx;
      u1 = W[l] + W[l2]; u2 = W[l] + W[l3]; u3 = W[l] + W[l1]; u4 = W[l] + W[l4]; u5 = W[l] + W[l5]; u6 = W[l] + W[l6];
      dudx = GX; roty = (roty - dudx) * xvol;
      dudx = GY; rotx = (rotx + dudx) * xvol;
      dudx = GZ; div = div + dudx * xvol;
      rot = rotx * rotx + roty * roty + rotz * rotz;
    } else
      rot = rotz * rotz;
#undef GX
#undef GY
#undef GZ
  }
  for (int d = 0; d < 3; d++) {
    if (!(dirs & (1 << d))) continue;
    double mach = sqrt(gam * rgp * rop[l + 4 * nd]);
    mach = 10.0 * (1.0 / tc[d]) * (1.0 / mach) * div;
    mach = 0.5 * (1.0 - tanh(2.5 + mach));
    const double p2 = rop[l - inc[d]] * rop[l - inc[d] + 4 * nd], p0 = rop[l] * rop[l + 4 * nd];
    const double p1 = rop[l + inc[d]] * rop[l + inc[d] + 4 * nd];
    double w = fabs(p2 - 2. * p0 + p1) / fabs(p2 + 2. * p0 + p1);
    w = dmin(2.0 * div * div / (div * div + rot + 1e-15) * w * mach, 1.0);
    wsc[l + d * nd] = 1.0 - w;
  }
}

void fo_sciacovelli(const i32 *p, const double *pr, const i32 *ind, const double *rop, const double *ti,
                    const double *tj, const double *tk, const double *vol, double *wig)
{
  const long nd = p[NDIMDX];
  double *wsc = wig + (p[KFLUDOM] == 2 ? 3 * nd : 0);
  const int twod = p[ITYPZONE] == 3;
  const int all = twod ? 3 : 7;
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const int c[3] = {i, j, k};
        shock_cell(p, pr, rop, ti, tj, tk, vol, wsc, i, j, k, all, c);
      }
  /* The loops over the ghost planes of the max sides (loopPlanI/J/K_begin.for) set the cell index only: the metric
   * index keeps the value the main loop left, that of its last cell (sciacovelli.for:96-127). */
  const int last[3] = {ind[1], ind[3], ind[5]};
  if (ind[1] == p[IJKV])
    for (int k = ind[4]; k <= ind[5]; k++)
      for (int j = ind[2]; j <= ind[3]; j++) shock_cell(p, pr, rop, ti, tj, tk, vol, wsc, ind[1] + 1, j, k, 1, last);
  if (ind[3] == p[IJKV + 1])
    for (int k = ind[4]; k <= ind[5]; k++)
      for (int i = ind[0]; i <= ind[1]; i++) shock_cell(p, pr, rop, ti, tj, tk, vol, wsc, i, ind[3] + 1, k, 2, last);
  if (!twod && ind[5] == p[IJKV + 2])
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) shock_cell(p, pr, rop, ti, tj, tk, vol, wsc, i, j, ind[5] + 1, 4, last);
}

typedef struct {
  int kflu, iflow;
  int slope;        /* 1 o1, 2 o3, 3 minmod (roe only: ROE/fluroe_select.for); 6 o3sc (every scheme) */
  double tmin_1;    /* minmod: no slope below 0.01 Tinf (template_FluxAndBalance.for:136) */
  double gam, rgp, gam1, gam3, c1, c2, c3, opt0, c4, c5, c6, cvisq, psiroe;
  double tcx, tcy, tcz, si, sj, sk, volinv;
  /* Spalart-Allmaras (iflow 3): turbulent conductivity factor, Sutherland constants of mulam.for, 1/sigma */
  double gam4, cmus1, temp01, coesut_sa, sigma_1;
  long nd, inc[3];
  /* curvilinear zones (zone type 0): metric arrays ti,tj,tk(NDIMDX_MTR,3), vol(NDIMDX_MTR) */
  int ityp;
  int homo;   /* zone type 1 */
  const double *t[3], *vol;
  long ndm, minc[3];
} fluxctx;

/* statement function of Fast/FastS/FastS/Compute/template_FluxAndBalance.for:92 */
static inline double avmin(double c, double r)
{
  const double sg = copysign(1., c);
  return sg * fmax(0., fmin(fabs(c), sg * r));
}

/* Right (qm) and left (qp) states of one variable q on the face between nm and l:
 * Fast/FastS/FastS/Compute/Slope/o3_slope_var.for, o1_slope_var.for, minmod_slope_var.for.  t_l = temperature of l. */
#define SLOPE_STATES(c, q, t_l, l, nm, nm2, np, qm_, qp_, csc)                              \
  do {                                                                                     \
    if ((c)->slope == 6) {   /* o3sc_slope_var.for: o3 states weighted by the shock sensor of the face */ \
      qm_ = (c)->c4 * (q)[l] + (c)->c5 * (q)[nm] + (c)->c6 * (q)[np];                      \
      qp_ = (c)->c4 * (q)[nm] + (c)->c6 * (q)[nm2] + (c)->c5 * (q)[l];                     \
      qm_ = (q)[l] * (1. - (csc)) + qm_ * (csc);                                           \
      qp_ = (q)[nm] * (1. - (csc)) + qp_ * (csc);                                          \
    } else                                                                                 \
    if ((c)->slope == 1) { qm_ = (q)[l]; qp_ = (q)[nm]; }                                  \
    else if ((c)->slope == 3) {                                                            \
      const double roff_ = .5 * fmax(0., fmin((t_l) * (c)->tmin_1 - .5, 1.));              \
      const double delm_ = (q)[l] - (q)[nm], delp_ = (q)[np] - (q)[l], delq_ = (q)[nm] - (q)[nm2]; \
      const double slp_ = roff_ * avmin(delm_, delp_), slq_ = roff_ * avmin(delq_, delm_); \
      qm_ = (q)[l] - slp_; qp_ = (q)[nm] + slq_;                                           \
    } else {                                                                               \
      qm_ = (c)->c4 * (q)[l] + (c)->c5 * (q)[nm] + (c)->c6 * (q)[np];                      \
      qp_ = (c)->c4 * (q)[nm] + (c)->c6 * (q)[nm2] + (c)->c5 * (q)[l];                     \
    }                                                                                      \
  } while (0)

#define NIJK_MTR 5
#define NDIMDX_MTR 43
/* 0-based metric index (Fast/FastC/FastC/formule_mtr_param.h) */
#define INDMTR(p, i, j, k)                                                                        \
  ((long)((i) + (p)[NIJK_MTR + 3] - 1) * (p)[NIJK_MTR] + (long)((j) + (p)[NIJK_MTR + 3] - 1) * (p)[NIJK_MTR + 1] + \
   (long)((k) + (p)[NIJK_MTR + 4] - 1) * (p)[NIJK_MTR + 2])

/* Flux through the face between cell l-inc[dir] and cell l, direction dir (0 i, 1 j, 2 k).
 * Inviscid part: left / right states (SLOPE_STATES: o3; o1 and minmod with roe) then
 *   ausmpred  Fast/FastS/FastS/Compute/AUSM/3dcart/fluFaceEuler_o3_3dcart_i.for + etat_GD.for
 *             + Vit_ent/qn_3dcart_i.for + Vit_ent/fluvector_3dcart_i.for
 *   senseur   Fast/FastS/FastS/Compute/SENSOR/3dcart/fluFaceEuler_o3_3dcart_i.for
 *   roe       Fast/FastS/FastS/Compute/ROE/3dcart/fluFaceEuler_o3_3dcart_i.for + etat_roe_GD.for
 *             + Vit_ent/fludiffer_3dcart_i.for
 * Viscous part (iflow 2): Fast/FastS/FastS/Compute/fluvisq_3dcart_{i,j,k}.for, mut_interface.for */
static inline void face_flux(const fluxctx *c, const double *rop, const double *wig,
                             const double *xmut, long l, int dir, double *flu)
{
  const long nd = c->nd, inc = c->inc[dir];
  const long nm = l - inc, nm2 = l - 2 * inc, np = l + inc;
  const double tc = dir == 0 ? c->tcx : (dir == 1 ? c->tcy : c->tcz);
  const double s = dir == 0 ? c->si : (dir == 1 ? c->sj : c->sk);
  double qm[5], qp[5];
  /* shock sensor of the face: the arrays after the wiggle sensor's when the scheme has one (sl_i / sl_j / sl_k of
   * template_FluxAndBalance.for:175-182) */
  const double csc = c->slope == 6 ? wig[l + dir * nd + (c->kflu == 2 ? 3 * nd : 0)] : 1.;
  for (int e = 0; e < 5; e++) {
    const double *q = rop + e * nd;
    SLOPE_STATES(c, q, rop[l + 4 * nd], l, nm, nm2, np, qm[e], qp[e], csc);
  }
  const int vn = 1 + dir; /* normal velocity component */
  double flu1, flu2, flu3, flu4, flu5, flu6 = 0., u_sa = 0., tdu_sa = 0.;
  if (c->kflu == 5) {
    const double gam = c->gam, rgp = c->rgp, gam1 = c->gam1, psiroe = c->psiroe;
    double r1 = qp[0], p1 = r1 * qp[4] * rgp;
    double h1 = gam1 * qp[4] * rgp + .5 * (qp[1] * qp[1] + qp[2] * qp[2] + qp[3] * qp[3]);
    double r2 = qm[0], p2 = r2 * qm[4] * rgp;
    double h2 = gam1 * qm[4] * rgp + .5 * (qm[1] * qm[1] + qm[2] * qm[2] + qm[3] * qm[3]);
    double qp1 = sqrt(r1), qm1 = sqrt(r2);
    double r = qp1 * qm1;
    double r_1 = 1. / (qp1 + qm1);
    double u = (qp1 * qp[1] + qm1 * qm[1]) * r_1;
    double v = (qp1 * qp[2] + qm1 * qm[2]) * r_1;
    double w = (qp1 * qp[3] + qm1 * qm[3]) * r_1;
    double h = (qp1 * h1 + qm1 * h2) * r_1;
    double q = .5 * (u * u + v * v + w * w);
    double cc = sqrt((gam - 1.) * fabs(h - q));
    double qn1 = qp[vn] * tc, qn2 = qm[vn] * tc;
    double du = (qm[1] - qp[1]) * s, dv = (qm[2] - qp[2]) * s, dw = (qm[3] - qp[3]) * s;
    double dp = p2 - p1, dqn = qn2 - qn1;
    double s_1 = 1. / dmax(s, 1.e-30);
    double nn = tc * s_1;
    double vel[3] = {u, v, w};
    double qn = vel[dir] * nn;
    double f1 = qn;
    double f4 = psiroe * (fabs(f1) + cc);
    double f2 = .5 + dsign(.5, fabs(f1) - f4);
    double f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (r2 - r1 - dp / (cc * cc)) * s;
    double dd[3] = {du, dv, dw};
    double dn[3] = {du, dv, dw};
    dn[dir] = dd[dir] - nn * dqn;
    flu1 = f3 * (f1);
    flu2 = f3 * (f1 * u + r * dn[0]);
    flu3 = f3 * (f1 * v + r * dn[1]);
    flu4 = f3 * (f1 * w + r * dn[2]);
    flu5 = f3 * (f1 * q + r * (u * du + v * dv + w * dw - qn * dqn));
    f1 = qn + cc;
    f2 = .5 + dsign(.5, fabs(f1) - f4);
    f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (dp + r * cc * dqn * s_1) / (2. * cc * cc) * s * f3;
    double vp[3] = {u, v, w};
    vp[dir] = vel[dir] + nn * cc;
    flu1 = flu1 + f1;
    flu2 = flu2 + f1 * vp[0];
    flu3 = flu3 + f1 * vp[1];
    flu4 = flu4 + f1 * vp[2];
    flu5 = flu5 + f1 * (h + qn * cc);
    f1 = qn - cc;
    f2 = .5 + dsign(.5, fabs(f1) - f4);
    f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (dp - r * cc * dqn * s_1) / (2. * cc * cc) * f3 * s;
    double vm[3] = {u, v, w};
    vm[dir] = vel[dir] - nn * cc;
    flu1 = flu1 + f1;
    flu2 = flu2 + f1 * vm[0];
    flu3 = flu3 + f1 * vm[1];
    flu4 = flu4 + f1 * vm[2];
    flu5 = flu5 + f1 * (h - qn * cc);
    double rou1 = r1 * qn1, rou2 = r2 * qn2, p1p2 = p1 + p2;
    double pt[3] = {0., 0., 0.};
    flu1 = (flu1 + rou1 + rou2) * 0.5;
    if (dir == 0) flu2 = (flu2 + rou1 * qp[1] + rou2 * qm[1] + tc * p1p2) * 0.5;
    else flu2 = (flu2 + rou1 * qp[1] + rou2 * qm[1]) * 0.5;
    if (dir == 1) flu3 = (flu3 + rou1 * qp[2] + rou2 * qm[2] + tc * p1p2) * 0.5;
    else flu3 = (flu3 + rou1 * qp[2] + rou2 * qm[2]) * 0.5;
    if (dir == 2) flu4 = (flu4 + rou1 * qp[3] + rou2 * qm[3] + tc * p1p2) * 0.5;
    else flu4 = (flu4 + rou1 * qp[3] + rou2 * qm[3]) * 0.5;
    flu5 = (flu5 + rou1 * h1 + rou2 * h2) * 0.5;
    (void)pt;
  } else {
    const double rgp = c->rgp, gam1 = c->gam1;
    double r1 = qp[0], rou1 = r1 * qp[1], rov1 = r1 * qp[2], row1 = r1 * qp[3];
    double p1 = r1 * qp[4] * rgp;
    double h1 = gam1 * p1 + .5 * (rou1 * qp[1] + rov1 * qp[2] + row1 * qp[3]);
    double r2 = qm[0], rou2 = r2 * qm[1], rov2 = r2 * qm[2], row2 = r2 * qm[3];
    double p2 = r2 * qm[4] * rgp;
    double h2 = gam1 * p2 + .5 * (rou2 * qm[1] + rov2 * qm[2] + row2 * qm[3]);
    double qn1 = qp[vn] * tc, qn2 = qm[vn] * tc;
    double u, tdu;
    if (c->kflu == 1) {
      double cc = rgp * c->gam * rop[l + 4 * nd];
      double son = sqrt(qn1 * qn1 / cc);
      double tam = c->c3 * son + s;
      double tam1 = dmax(0., tam) * c->c2;
      u = 0.25 * (qn1 + qn2) - tam1 * (p2 - p1);
      tdu = dmax(fabs(u), c->c1 * s);
    } else {
      double wg = wig[l + dir * nd];
      u = 0.25 * (qn1 + qn2) - c->c2 * s * (p2 - p1) * (c->opt0 * wg + 1. - c->opt0);
      tdu = dmax(fabs(u), c->c1 * s) * wg;
    }
    double p1p2 = (p1 + p2) * 0.5;
    flu1 = u * (r1 + r2) - tdu * (r2 - r1);
    if (dir == 0) flu2 = u * (rou1 + rou2) - tdu * (rou2 - rou1) + tc * p1p2;
    else flu2 = u * (rou1 + rou2) - tdu * (rou2 - rou1);
    if (dir == 1) flu3 = u * (rov1 + rov2) - tdu * (rov2 - rov1) + tc * p1p2;
    else flu3 = u * (rov1 + rov2) - tdu * (rov2 - rov1);
    if (dir == 2) flu4 = u * (row1 + row2) - tdu * (row2 - row1) + tc * p1p2;
    else flu4 = u * (row1 + row2) - tdu * (row2 - row1);
    flu5 = u * (h1 + h2) - tdu * (h2 - h1);
    u_sa = u; tdu_sa = tdu;
  }
  if (c->iflow == 3 && c->kflu == 5) {
    /* roe: first-order upwinding with the Roe-averaged normal velocity (ROE/fluFaceSA.for,
     * ROE/3dcart/fluFaceSA_o3_3dcart_{i,j,k}.for, Vit_ent/qn_3dcart_i.for) */
    const double *q6 = rop + 5 * nd;
    double qp1 = rop[nm], qm1 = rop[l];
    const double qn1 = rop[nm + vn * nd] * tc, qn2 = rop[l + vn * nd] * tc;
    const double qm6 = q6[l] * qm1 * qn2, qp6 = q6[nm] * qp1 * qn1;
    qp1 = sqrt(qp1); qm1 = sqrt(qm1);
    const double r_1 = 1. / (qp1 + qm1);
    const double vroe = r_1 * (qp1 * rop[nm + vn * nd] + qm1 * rop[l + vn * nd]);
    const double s_1 = 1. / dmax(s, 1.e-30);
    const double tdu = dmax(c->c1 / s_1, fabs(vroe * tc));
    flu6 = 0.5 * (qm6 + qp6 - tdu * (q6[l] * rop[l] - q6[nm] * rop[nm]));
  } else if (c->iflow == 3) {   /* convective flux of rho*nu_tilde: AUSM/fluFaceSA.for, SENSOR/fluFaceSA.for */
    const double *q6 = rop + 5 * nd;
    const double qm_ = c->c4 * q6[l] + c->c5 * q6[nm] + c->c6 * q6[np];
    const double qp_ = c->c4 * q6[nm] + c->c6 * q6[nm2] + c->c5 * q6[l];
    const double qm6 = qm_ * qm[0], qp6 = qp_ * qp[0];
    flu6 = u_sa * (qm6 + qp6) - tdu_sa * (qm6 - qp6);
  }

  if (c->iflow >= 2) {
    const long inci = c->inc[0], incj = c->inc[1], inck = c->inc[2];
    const double *U = rop + nd, *V = rop + 2 * nd, *W = rop + 3 * nd, *T = rop + 4 * nd;
    const double tcx = c->tcx, tcy = c->tcy, tcz = c->tcz;
    const long il = l - inc, ir = l;
    double f1, f2, f3, f4, f5, f6, fv, fv5, div;
    double xmutvol = (xmut[ir] + xmut[il]) * c->volinv;
    double xktvol = xmutvol * c->gam3;
    double xmulam = 0.;
    if (c->iflow == 3) {   /* mut_prandtltb_interface.for: laminar and turbulent parts of the conductivity */
      xmulam = c->coesut_sa * sqrt(T[il] * c->temp01) / (1. + c->cmus1 / T[il]);
      xmulam = xmulam + c->coesut_sa * sqrt(T[ir] * c->temp01) / (1. + c->cmus1 / T[ir]);
      const double xmutot = xmut[ir] + xmut[il];
      const double xmutur = xmutot - xmulam;
      xktvol = (xmulam * c->gam3 + xmutur * c->gam4) * c->volinv;
      xmutvol = xmutot * c->volinv;
    }
    if (dir == 0) {
      const long l220 = l - inci - incj, l020 = l - incj, l210 = l - inci + incj, l010 = l + incj;
      const long l202 = l - inci - inck, l002 = l - inck, l201 = l - inci + inck, l001 = l + inck;
      const double tix = tcx, tjy = .25 * tcy, tkz = .25 * tcz;
      f1 = U[il]; f2 = U[ir]; fv = f1 + f2;
      f3 = fv + U[l220] + U[l020]; f4 = fv + U[l210] + U[l010];
      f5 = fv + U[l202] + U[l002]; f6 = fv + U[l201] + U[l001];
      double gradU_nx = (f2 - f1) * tix, gradU_ny = (f4 - f3) * tjy, gradU_nz = (f6 - f5) * tkz;
      f1 = V[il]; f2 = V[ir]; fv = f1 + f2;
      f3 = fv + V[l220] + V[l020]; f4 = fv + V[l210] + V[l010];
      double gradV_nx = (f2 - f1) * tix, gradV_ny = (f4 - f3) * tjy;
      f1 = W[il]; f2 = W[ir]; fv = f1 + f2;
      f5 = fv + W[l202] + W[l002]; f6 = fv + W[l201] + W[l001];
      double gradW_nx = (f2 - f1) * tix, gradW_nz = (f6 - f5) * tkz;
      f1 = T[il]; f2 = T[ir];
      double gradT_nx = (f2 - f1) * tix;
      div = (gradU_nx + gradV_ny + gradW_nz) * c->cvisq;
      f1 = gradU_nx - div; f2 = gradU_ny + gradV_nx; f3 = gradU_nz + gradW_nx;
      fv = -(2. * f1 * tcx) * xmutvol; fv5 = fv * (U[ir] + U[il]); flu2 = flu2 + fv;
      fv = -(f2 * tcx) * xmutvol; fv5 = fv5 + fv * (V[ir] + V[il]); flu3 = flu3 + fv;
      fv = -(f3 * tcx) * xmutvol; fv5 = fv5 + fv * (W[ir] + W[il]); flu4 = flu4 + fv;
      fv = fv5 * 0.5 - (gradT_nx * tcx) * xktvol; flu5 = flu5 + fv;
    } else if (dir == 1) {
      const long l022 = l - incj - inck, l002 = l - inck, l021 = l - incj + inck, l001 = l + inck;
      const long l220 = l - inci - incj, l200 = l - inci, l120 = l + inci - incj, l100 = l + inci;
      const double tiy = tcy, tjz = .25 * tcz, tkx = .25 * tcx;
      f1 = U[il]; f2 = U[ir]; fv = f1 + f2;
      f5 = fv + U[l220] + U[l200]; f6 = fv + U[l120] + U[l100];
      double gradU_nx = (f6 - f5) * tkx, gradU_ny = (f2 - f1) * tiy;
      f1 = V[il]; f2 = V[ir]; fv = f1 + f2;
      f3 = fv + V[l022] + V[l002]; f4 = fv + V[l021] + V[l001];
      f5 = fv + V[l220] + V[l200]; f6 = fv + V[l120] + V[l100];
      double gradV_nx = (f6 - f5) * tkx, gradV_ny = (f2 - f1) * tiy, gradV_nz = (f4 - f3) * tjz;
      f1 = W[il]; f2 = W[ir]; fv = f1 + f2;
      f3 = fv + W[l022] + W[l002]; f4 = fv + W[l021] + W[l001];
      double gradW_ny = (f2 - f1) * tiy, gradW_nz = (f4 - f3) * tjz;
      f1 = T[il]; f2 = T[ir];
      double gradT_ny = (f2 - f1) * tiy;
      div = (gradU_nx + gradV_ny + gradW_nz) * c->cvisq;
      f2 = gradU_ny + gradV_nx; f4 = gradV_ny - div; f5 = gradV_nz + gradW_ny;
      fv = -(f2 * tcy) * xmutvol; fv5 = fv * (U[ir] + U[il]); flu2 = flu2 + fv;
      fv = -(2. * f4 * tcy) * xmutvol; fv5 = fv5 + fv * (V[ir] + V[il]); flu3 = flu3 + fv;
      fv = -(f5 * tcy) * xmutvol; fv5 = fv5 + fv * (W[ir] + W[il]); flu4 = flu4 + fv;
      fv = fv5 * 0.5 - (gradT_ny * tcy) * xktvol; flu5 = flu5 + fv;
    } else {
      const long l200 = l - inci, l202 = l - inci - inck, l102 = l + inci - inck, l100 = l + inci;
      const long l022 = l - incj - inck, l020 = l - incj, l012 = l + incj - inck, l010 = l + incj;
      const double tiz = tcz, tky = 0.25 * tcy, tjx = 0.25 * tcx;
      f1 = U[il]; f2 = U[ir]; fv = f1 + f2;
      f3 = fv + U[l202] + U[l200]; f4 = fv + U[l102] + U[l100];
      double gradU_nx = (f4 - f3) * tjx, gradU_nz = (f2 - f1) * tiz;
      f1 = V[il]; f2 = V[ir]; fv = f1 + f2;
      f5 = fv + V[l022] + V[l020]; f6 = fv + V[l012] + V[l010];
      double gradV_ny = (f6 - f5) * tky, gradV_nz = (f2 - f1) * tiz;
      f1 = W[il]; f2 = W[l]; fv = f1 + f2;
      f3 = fv + W[l202] + W[l200]; f4 = fv + W[l102] + W[l100];
      f5 = fv + W[l022] + W[l020]; f6 = fv + W[l012] + W[l010];
      double gradW_nx = (f4 - f3) * tjx, gradW_ny = (f6 - f5) * tky, gradW_nz = (f2 - f1) * tiz;
      f1 = T[il]; f2 = T[ir];
      double gradT_nz = (f2 - f1) * tiz;
      div = (gradU_nx + gradV_ny + gradW_nz) * c->cvisq;
      f3 = gradU_nz + gradW_nx; f5 = gradV_nz + gradW_ny; f6 = gradW_nz - div;
      fv = -f3 * tcz * xmutvol; fv5 = fv * (U[ir] + U[il]); flu2 = flu2 + fv;
      fv = -f5 * tcz * xmutvol; fv5 = fv5 + fv * (V[ir] + V[il]); flu3 = flu3 + fv;
      fv = -2. * f6 * tcz * xmutvol; fv5 = fv5 + fv * (W[ir] + W[il]); flu4 = flu4 + fv;
      fv = fv5 * 0.5 - gradT_nz * tcz * xktvol; flu5 = flu5 + fv;
    }
    if (c->iflow == 3) {   /* SA/fluvisq_SA_3dcart_{i,j,k}.for, SA/mut_interface.for */
      const double *R = rop, *N = rop + 5 * nd;
      const double g = (N[ir] - N[il]) * tc;
      double xm = xmulam + N[il] * R[il] + N[ir] * R[ir];
      xm = xm * c->volinv * c->sigma_1;
      if (dir == 1) flu6 = flu6 - xm * (g * tc);
      else flu6 = flu6 - xm * g * tc;
    }
  }
  flu[0] = flu1; flu[1] = flu2; flu[2] = flu3; flu[3] = flu4; flu[4] = flu5;
  if (c->iflow == 3) flu[5] = flu6;
}

/* Same face flux on a curvilinear zone (zone type 0): lt = metric index of cell l.
 *   normal, surface       Normale/normale_3dfull_{i,j,k}.for
 *   normal velocity       Vit_ent/qn_3dfull_i.for
 *   ausmpred / senseur    AUSM|SENSOR/3dfull/fluFaceEuler_o3_3dfull_i.for, Vit_ent/fluvector_3dfull_i.for
 *   roe                   ROE/3dfull/fluFaceEuler_o3_3dfull_i.for, Vit_ent/fludiffer_3dfull_i.for
 *   viscous               fluvisq_3dfull_{i,j,k}.for, mut_interface.for
 * The three directions of the viscous fragment are one pattern: with a = dir+1, b = dir+2 (cyclic),
 * the cross sums f3..f6 take the neighbours along a then b and the metrics of those directions. */
static inline void face_flux_full(const fluxctx *c, const double *rop, const double *wig,
                                  const double *xmut, long l, long lt, int dir, double *flu)
{
  const long nd = c->nd, inc = c->inc[dir], ndm = c->ndm;
  const long nm = l - inc, nm2 = l - 2 * inc, np = l + inc;
  const double *Tn = c->t[dir];
  /* two-dimensional zone (type 3): ti, tj carry 2 components, there is no k direction; its fragments
   * (Normale/normale_2d_i.for, etat_GD_2d.for, Vit_ent/qn_2d_i.for, fluvector_2d_i.for, fluvisq_2d_i.for, ...)
   * are the 3dfull ones without the z / w / k terms, i.e. with those terms equal to zero */
  const int twod = c->ityp == 3;
  const double tcx = Tn[lt], tcy = Tn[lt + ndm], tcz = twod ? 0. : Tn[lt + 2 * ndm];
  const double s = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
  double qm[5], qp[5];
  const double csc = c->slope == 6 ? wig[l + dir * nd + (c->kflu == 2 ? 3 * nd : 0)] : 1.;
  for (int e = 0; e < 5; e++) {
    const double *q = rop + e * nd;
    SLOPE_STATES(c, q, rop[l + 4 * nd], l, nm, nm2, np, qm[e], qp[e], csc);
  }
  double flu1, flu2, flu3, flu4, flu5, flu6 = 0., u_sa = 0., tdu_sa = 0.;
  if (c->kflu == 5) {
    const double gam = c->gam, rgp = c->rgp, gam1 = c->gam1, psiroe = c->psiroe;
    double r1 = qp[0], p1 = r1 * qp[4] * rgp;
    double h1 = gam1 * qp[4] * rgp + .5 * (qp[1] * qp[1] + qp[2] * qp[2] + qp[3] * qp[3]);
    double r2 = qm[0], p2 = r2 * qm[4] * rgp;
    double h2 = gam1 * qm[4] * rgp + .5 * (qm[1] * qm[1] + qm[2] * qm[2] + qm[3] * qm[3]);
    double qp1 = sqrt(r1), qm1 = sqrt(r2);
    double r = qp1 * qm1;
    double r_1 = 1. / (qp1 + qm1);
    double u = (qp1 * qp[1] + qm1 * qm[1]) * r_1;
    double v = (qp1 * qp[2] + qm1 * qm[2]) * r_1;
    double w = (qp1 * qp[3] + qm1 * qm[3]) * r_1;
    double h = (qp1 * h1 + qm1 * h2) * r_1;
    double q = .5 * (u * u + v * v + w * w);
    double cc = sqrt((gam - 1.) * fabs(h - q));
    double qn1 = qp[1] * tcx + qp[2] * tcy + qp[3] * tcz;
    double qn2 = qm[1] * tcx + qm[2] * tcy + qm[3] * tcz;
    double du = (qm[1] - qp[1]) * s, dv = (qm[2] - qp[2]) * s, dw = (qm[3] - qp[3]) * s;
    double dp = p2 - p1, dqn = qn2 - qn1;
    double s_1 = 1.0 / dmax(s, 1.E-30);
    double nx = tcx * s_1, ny = tcy * s_1, nz = tcz * s_1;
    double qn = u * nx + v * ny + w * nz;
    double f1 = qn;
    double f4 = psiroe * (fabs(f1) + cc);
    double f2 = 0.5 + dsign(0.5, fabs(f1) - f4);
    double f3 = (f2 - 1.0) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (r2 - r1 - dp / (cc * cc)) * s;
    flu1 = f3 * (f1);
    flu2 = f3 * (f1 * u + r * (du - nx * dqn));
    flu3 = f3 * (f1 * v + r * (dv - ny * dqn));
    flu4 = f3 * (f1 * w + r * (dw - nz * dqn));
    flu5 = f3 * (f1 * q + r * (u * du + v * dv + w * dw - qn * dqn));
    f1 = qn + cc;
    f2 = 0.5 + dsign(0.5, fabs(f1) - f4);
    f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    /* Vit_ent/fludiffer_2d_j.for:31 multiplies by f3 before the surface */
    if (twod && dir == 1) f1 = (dp + r * cc * dqn * s_1) / (2. * cc * cc) * f3 * s;
    else f1 = (dp + r * cc * dqn * s_1) / (2. * cc * cc) * s * f3;
    flu1 = flu1 + f1;
    flu2 = flu2 + f1 * (u + nx * cc);
    flu3 = flu3 + f1 * (v + ny * cc);
    flu4 = flu4 + f1 * (w + nz * cc);
    flu5 = flu5 + f1 * (h + qn * cc);
    f1 = qn - cc;
    f2 = 0.5 + dsign(0.5, fabs(f1) - f4);
    f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (dp - r * cc * dqn * s_1) / (2. * cc * cc) * f3 * s;
    flu1 = flu1 + f1;
    flu2 = flu2 + f1 * (u - nx * cc);
    flu3 = flu3 + f1 * (v - ny * cc);
    flu4 = flu4 + f1 * (w - nz * cc);
    flu5 = flu5 + f1 * (h - qn * cc);
    double rou1 = r1 * qn1, rou2 = r2 * qn2, p1p2 = p1 + p2;
    flu1 = (flu1 + rou1 + rou2) * 0.5;
    flu2 = (flu2 + rou1 * qp[1] + rou2 * qm[1] + tcx * p1p2) * 0.5;
    flu3 = (flu3 + rou1 * qp[2] + rou2 * qm[2] + tcy * p1p2) * 0.5;
    flu4 = (flu4 + rou1 * qp[3] + rou2 * qm[3] + tcz * p1p2) * 0.5;
    flu5 = (flu5 + rou1 * h1 + rou2 * h2) * 0.5;
  } else {
    const double rgp = c->rgp, gam1 = c->gam1;
    double r1 = qp[0], rou1 = r1 * qp[1], rov1 = r1 * qp[2], row1 = r1 * qp[3];
    double p1 = r1 * qp[4] * rgp;
    double h1 = gam1 * p1 + .5 * (rou1 * qp[1] + rov1 * qp[2] + row1 * qp[3]);
    double r2 = qm[0], rou2 = r2 * qm[1], rov2 = r2 * qm[2], row2 = r2 * qm[3];
    double p2 = r2 * qm[4] * rgp;
    double h2 = gam1 * p2 + .5 * (rou2 * qm[1] + rov2 * qm[2] + row2 * qm[3]);
    double qn1 = qp[1] * tcx + qp[2] * tcy + qp[3] * tcz;
    double qn2 = qm[1] * tcx + qm[2] * tcy + qm[3] * tcz;
    double u, tdu;
    if (c->kflu == 1) {
      double cc = rgp * c->gam * rop[l + 4 * nd];
      double son = sqrt(qn1 * qn1 / cc);
      double tam = c->c3 * son + s;
      double tam1 = dmax(0., tam) * c->c2;
      u = 0.25 * (qn1 + qn2) - tam1 * (p2 - p1);
      tdu = dmax(fabs(u), c->c1 * s);
    } else {
      double wg = wig[l + dir * nd];
      u = 0.25 * (qn1 + qn2) - c->c2 * s * (p2 - p1) * (c->opt0 * wg + 1. - c->opt0);
      tdu = dmax(fabs(u), c->c1 * s) * wg;
    }
    double p1p2 = (p1 + p2) * 0.5;
    flu1 = u * (r1 + r2) - tdu * (r2 - r1);
    flu2 = u * (rou1 + rou2) - tdu * (rou2 - rou1) + tcx * p1p2;
    flu3 = u * (rov1 + rov2) - tdu * (rov2 - rov1) + tcy * p1p2;
    flu4 = u * (row1 + row2) - tdu * (row2 - row1) + tcz * p1p2;
    flu5 = u * (h1 + h2) - tdu * (h2 - h1);
    u_sa = u; tdu_sa = tdu;
  }
  if (c->iflow == 3 && c->kflu == 5) {
    /* ROE/fluFaceSA.for with Vit_ent/qn_3dfull_i.for: first-order states, Roe-averaged normal velocity */
    const double *q6 = rop + 5 * nd;
    double qp1 = rop[nm], qm1 = rop[l];
    double qp2 = rop[nm + nd], qp3 = rop[nm + 2 * nd], qp4 = rop[nm + 3 * nd];
    const double qm2 = rop[l + nd], qm3 = rop[l + 2 * nd], qm4 = rop[l + 3 * nd];
    double qn1 = qp2 * tcx + qp3 * tcy + qp4 * tcz;
    const double qn2 = qm2 * tcx + qm3 * tcy + qm4 * tcz;
    const double qm6 = q6[l] * qm1 * qn2, qp6 = q6[nm] * qp1 * qn1;
    qp1 = sqrt(qp1); qm1 = sqrt(qm1);
    const double r_1 = 1. / (qp1 + qm1);
    qp2 = r_1 * (qp1 * qp2 + qm1 * qm2); qp3 = r_1 * (qp1 * qp3 + qm1 * qm3); qp4 = r_1 * (qp1 * qp4 + qm1 * qm4);
    qn1 = qp2 * tcx + qp3 * tcy + qp4 * tcz;
    const double s_1 = 1. / dmax(s, 1.e-30);
    const double tdu = dmax(c->c1 / s_1, fabs(qn1));
    flu6 = 0.5 * (qm6 + qp6 - tdu * (q6[l] * rop[l] - q6[nm] * rop[nm]));
  } else if (c->iflow == 3) {   /* AUSM/fluFaceSA.for, SENSOR/fluFaceSA.for */
    const double *q6 = rop + 5 * nd;
    const double qm_ = c->c4 * q6[l] + c->c5 * q6[nm] + c->c6 * q6[np];
    const double qp_ = c->c4 * q6[nm] + c->c6 * q6[nm2] + c->c5 * q6[l];
    const double qm6 = qm_ * qm[0], qp6 = qp_ * qp[0];
    flu6 = u_sa * (qm6 + qp6) - tdu_sa * (qm6 - qp6);
  }
  if (c->iflow >= 2) {
    const int a = (dir + 1) % 3, b = (dir + 2) % 3;
    const long ia = c->inc[a], ib = c->inc[b];
    const long md = c->minc[dir], ma = c->minc[a], mb = c->minc[b];
    const double *Ta = c->t[a], *Tb = c->t[b];
    const long il = l - inc, ir = l;
    double tn[3], tn1[3], ta[3], ta1[3], tb[3], tb1[3];
    const int has_a = !(twod && a == 2), has_b = !(twod && b == 2);
    for (int m = 0; m < 3; m++) {
      tn[m] = tn1[m] = ta[m] = ta1[m] = tb[m] = tb1[m] = 0.;
      if (twod && m == 2) continue;
      tn[m] = .5 * (Tn[lt + m * ndm] + Tn[lt - md + m * ndm]);
      tn1[m] = .5 * (Tn[lt + m * ndm] + Tn[lt + md + m * ndm]);
      if (has_a) {
        ta[m] = .125 * (Ta[lt + m * ndm] + Ta[lt - md + m * ndm]);
        ta1[m] = .125 * (Ta[lt + ma + m * ndm] + Ta[lt - md + ma + m * ndm]);
      }
      if (has_b) {
        tb[m] = .125 * (Tb[lt + m * ndm] + Tb[lt - md + m * ndm]);
        tb1[m] = .125 * (Tb[lt + mb + m * ndm] + Tb[lt - md + mb + m * ndm]);
      }
    }
    double g[5][3];   /* gradients of u, v, w, T (and nu_tilde: SA/fluvisq_SA_3dfull_i.for) */
    for (int e = 1; e <= (c->iflow == 3 ? 5 : 4); e++) {
      const double *q = rop + e * nd;
      double f1 = q[il], f2 = q[ir];
      double f3 = has_a ? q[ir] + q[il] + q[il - ia] + q[ir - ia] : 0.;
      double f4 = has_a ? q[ir] + q[il] + q[il + ia] + q[ir + ia] : 0.;
      double f5 = has_b ? q[ir] + q[il] + q[il - ib] + q[ir - ib] : 0.;
      double f6 = has_b ? q[ir] + q[il] + q[il + ib] + q[ir + ib] : 0.;
      for (int m = 0; m < 3; m++)
        g[e - 1][m] = (f2 * tn1[m] - f1 * tn[m]) + (f4 * ta1[m] - f3 * ta[m]) + (f6 * tb1[m] - f5 * tb[m]);
      if (c->homo)   /* fluvisq_3dhomo_{i,j,k}.for: the one k-face surface factored out of the z derivative */
        g[e - 1][2] = dir == 0 ? (f6 - f5) * tb[2] : (dir == 1 ? (f4 - f3) * ta[2] : (f2 - f1) * tn[2]);
    }
    const double *U = rop + nd, *V = rop + 2 * nd, *W = rop + 3 * nd;
    double div = (g[0][0] + g[1][1] + g[2][2]) * c->cvisq;
    double f1 = g[0][0] - div, f2 = g[0][1] + g[1][0], f3 = g[0][2] + g[2][0];
    double f4 = g[1][1] - div, f5 = g[1][2] + g[2][1], f6 = g[2][2] - div;
    double volinv = 1. / (c->vol[lt] + c->vol[lt - md]);
    double xmutvol = (xmut[ir] + xmut[il]) * volinv;
    double xktvol = xmutvol * c->gam3;
    double xmulam = 0.;
    if (c->iflow == 3) {   /* mut_prandtltb_interface.for */
      const double *T = rop + 4 * nd;
      xmulam = c->coesut_sa * sqrt(T[il] * c->temp01) / (1. + c->cmus1 / T[il]);
      xmulam = xmulam + c->coesut_sa * sqrt(T[ir] * c->temp01) / (1. + c->cmus1 / T[ir]);
      const double xmutot = xmut[ir] + xmut[il];
      const double xmutur = xmutot - xmulam;
      xktvol = (xmulam * c->gam3 + xmutur * c->gam4) * volinv;
      xmutvol = xmutot * volinv;
    }
    double fv, fv5;
    fv = -(2. * f1 * tcx + f2 * tcy + f3 * tcz) * xmutvol;
    fv5 = fv * (U[ir] + U[il]);
    flu2 = flu2 + fv;
    fv = -(f2 * tcx + 2. * f4 * tcy + f5 * tcz) * xmutvol;
    fv5 = fv5 + fv * (V[ir] + V[il]);
    flu3 = flu3 + fv;
    fv = -(f3 * tcx + f5 * tcy + 2. * f6 * tcz) * xmutvol;
    fv5 = fv5 + fv * (W[ir] + W[il]);
    flu4 = flu4 + fv;
    fv = fv5 * 0.5 - (g[3][0] * tcx + g[3][1] * tcy + g[3][2] * tcz) * xktvol;
    flu5 = flu5 + fv;
    if (c->iflow == 3) {   /* SA/mut_interface.for, SA/assemble_tauij_3dfull.for, SA/fluvisq_SA_2d_i.for */
      const double *R = rop, *N = rop + 5 * nd;
      double xm = xmulam + N[il] * R[il] + N[ir] * R[ir];
      xm = xm * volinv * c->sigma_1;
      if (c->homo && dir == 2) flu6 = flu6 - xm * g[4][2] * tcz;   /* SA/fluvisq_SA_3dhomo_k.for: this order */
      else flu6 = flu6 - xm * (g[4][0] * tcx + g[4][1] * tcy + g[4][2] * tcz);
    }
  }
  flu[0] = flu1; flu[1] = flu2; flu[2] = flu3; flu[3] = flu4; flu[4] = flu5;
  if (c->iflow == 3) flu[5] = flu6;
}

static void fluxctx_init(fluxctx *c, const i32 *p, const double *pr, const double *ti,
                         const double *tj, const double *tk, const double *vol)
{
  c->ityp = p[ITYPZONE];
  c->homo = p[ITYPZONE] == 1;
  c->t[0] = ti; c->t[1] = tj; c->t[2] = tk; c->vol = vol;
  c->ndm = p[NDIMDX_MTR];
  c->minc[0] = p[NIJK_MTR]; c->minc[1] = p[NIJK_MTR + 1]; c->minc[2] = p[NIJK_MTR + 2];
  /* constants: Fast/FastS/FastS/Compute/template_FluxAndBalance.for:108-156 */
  c->kflu = p[KFLUDOM]; c->iflow = p[IFLOW];
  c->slope = p[SLOPE] == 6 ? 6 : (p[KFLUDOM] == 5 ? p[SLOPE] : 2);
  c->tmin_1 = 100. / pr[TINF];
  c->tcx = ti[0]; c->tcy = tj[0]; c->tcz = p[ITYPZONE] == 3 ? 0. : tk[0];
  c->si = fabs(c->tcx); c->sj = fabs(c->tcy); c->sk = fabs(c->tcz);
  c->volinv = 0.5 / vol[0];
  double gam = pr[GAMMA];
  c->gam = gam;
  c->rgp = pr[CVINF] * (gam - 1.);
  c->gam1 = gam / (gam - 1.);
  c->gam3 = c->gam1 / pr[PRANDT] * c->rgp;
  c->gam4 = c->gam1 / pr[38] * c->rgp;                 /* PRANDT_TUR */
  c->cmus1 = pr[CS]; c->temp01 = 1. / pr[TEMP0];
  c->coesut_sa = pr[XMUL0] * (1. + c->cmus1 * c->temp01);
  c->sigma_1 = 1. / (2. / 3.);
  double roref = pr[ROINF], uref = pr[VINF];
  c->psiroe = pr[PSIROE];
  c->c1 = 0.02 * uref;
  c->c2 = 0.02 / (uref * roref);
  c->c3 = -2.;
  c->opt0 = (double)(float)p[SENSORTYPE];
  double c6 = 1. / 6.;
  c->c4 = 5. * c6; c->c5 = 2. * c6; c->c6 = -1. * c6;
  c->cvisq = 1. / 3;
  c->nd = p[NDIMDX];
  c->inc[0] = 1; c->inc[1] = p[NIJK]; c->inc[2] = (long)p[NIJK] * p[NIJK + 1];
}

/* Flux balance over the range ind (one cache block, no neighbouring thread): order of the
 * face loops and of the += / -= pairs as Fast/FastS/FastS/Compute/template_FluxAndBalance.for:192-271,
 * assemble_drodm_plus_vec1.for, flu_send_scater_vec1.for */
static void flux_own(const i32 *p, const double *pr, const i32 *ind, const double *rop, double *drodm,
                     const double *wig, const double *xmut, const double *ti, const double *tj,
                     const double *tk, const double *vol, int own_lo, int own_hi);
void fo_flux(const i32 *p, const double *pr, const i32 *ind, const double *rop, double *drodm,
             const double *wig, const double *xmut, const double *ti, const double *tj,
             const double *tk, const double *vol)
{
  flux_own(p, pr, ind, rop, drodm, wig, xmut, ti, tj, tk, vol, -(1 << 30), 1 << 30);
}

/* fo_flux on a k-slab of a zone shared between threads: contributions to cells whose k lies outside
 * [own_lo, own_hi] are left to the thread that owns them (it evaluates the same face), so every cell receives
 * its six contributions in the order of the single-thread sweep. */
static void flux_own(const i32 *p, const double *pr, const i32 *ind, const double *rop, double *drodm,
                     const double *wig, const double *xmut, const double *ti, const double *tj,
                     const double *tk, const double *vol, int own_lo, int own_hi)
{
  GEOM(p)
  fluxctx c;
  fluxctx_init(&c, p, pr, ti, tj, tk, vol);
  double f[6];
  const int neq = p[NEQ];
#define FACE(l, i, j, k, dir)                                                                     \
  do {                                                                                            \
    if (FULL3D(c.ityp) || c.ityp == 3) face_flux_full(&c, rop, wig, xmut, (l), INDMTR(p, i, j, k), (dir), f); \
    else face_flux(&c, rop, wig, xmut, (l), (dir), f);                                            \
  } while (0)
  const int twod = c.ityp == 3;
  for (int k = ind[4]; k <= ind[5]; k++) {
    for (int j = ind[2]; j <= ind[3]; j++) {
      for (int dir = twod ? 1 : 2; dir >= 0; dir--)
        for (int i = ind[0]; i <= ind[1]; i++) {
          long l = INDDM(i, j, k), l0 = l - c.inc[dir];
          FACE(l, i, j, k, dir);
          const int below = dir != 2 || k - 1 >= own_lo;
          for (int e = 0; e < neq; e++) {
            drodm[l + e * nd] = drodm[l + e * nd] + f[e];
            if (below) drodm[l0 + e * nd] = drodm[l0 + e * nd] - f[e];
          }
        }
      {
        long l = INDDM(ind[1] + 1, j, k), ls = l - 1;
        FACE(l, ind[1] + 1, j, k, 0);
        for (int e = 0; e < neq; e++) {
          drodm[ls + e * nd] = drodm[ls + e * nd] - f[e];
          drodm[l + e * nd] = drodm[l + e * nd] + f[e];
        }
      }
    }
    for (int i = ind[0]; i <= ind[1]; i++) {
      long l = INDDM(i, ind[3] + 1, k), ls = l - c.inc[1];
      FACE(l, i, ind[3] + 1, k, 1);
      for (int e = 0; e < neq; e++) {
        drodm[ls + e * nd] = drodm[ls + e * nd] - f[e];
        drodm[l + e * nd] = drodm[l + e * nd] + f[e];
      }
    }
  }
  for (int j = ind[2]; j <= ind[3] && !twod; j++)
    for (int i = ind[0]; i <= ind[1]; i++) {
      long l = INDDM(i, j, ind[5] + 1), ls = l - c.inc[2];
      FACE(l, i, j, ind[5] + 1, 2);
      for (int e = 0; e < neq; e++) {
        drodm[ls + e * nd] = drodm[ls + e * nd] - f[e];
        if (ind[5] + 1 <= own_hi) drodm[l + e * nd] = drodm[l + e * nd] + f[e];
      }
    }
#undef FACE
}

/* Runge-Kutta stage update in conservative form, back to (rho,u,v,w,T)
 * (Fast/FastS/FastS/Compute/core3ark3.for:21-24,70-80,108-154) */
void fo_core3ark3(const i32 *p, const double *pr, int nitcfg, const i32 *ind, double *rop_1,
                  const double *rop, const double *drodm, const double *coe)
{
  GEOM(p)
  const double cv = pr[CVINF], cvinv = 1. / cv, cvinv2 = 0.5 * cvinv;
  double coefW = 1.;
  int b = p[RK];
  if (b == 2) coefW = nitcfg == 1 ? 1. : 0.5;
  else if (b == 3) coefW = nitcfg == 1 ? 0.5 : (nitcfg == 2 ? 0.9106836025229591 : 0.3660254037844387);
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        long l = INDDM(i, j, k);
        double r_1 = coefW * coe[l];
        double tab1 = drodm[l] * r_1, tab2 = drodm[l + nd] * r_1, tab3 = drodm[l + 2 * nd] * r_1;
        double tab4 = drodm[l + 3 * nd] * r_1, tab5 = drodm[l + 4 * nd] * r_1;
        double ro_old = rop[l], u_old = rop[l + nd], v_old = rop[l + 2 * nd];
        double w_old = rop[l + 3 * nd], t_old = rop[l + 4 * nd];
        double ro_new = rop[l] + tab1;
        rop_1[l] = ro_new;
        r_1 = 1. / ro_new;
        double un = (ro_old * u_old + tab2) * r_1;
        double vn = (ro_old * v_old + tab3) * r_1;
        double wn = (ro_old * w_old + tab4) * r_1;
        if (p[ITYPZONE] == 3) wn = 0.;   /* core3ark3.for:156-183 */
        rop_1[l + nd] = un; rop_1[l + 2 * nd] = vn; rop_1[l + 3 * nd] = wn;
        double roe_old = ro_old * (cv * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
        rop_1[l + 4 * nd] = (roe_old + tab5) * r_1 * cvinv - cvinv2 * (un * un + vn * vn + wn * wn);
        if (p[NEQ] == 6) {   /* core3ark3.for:196-236: nu_tilde clipped to [0, ratiom * nu_laminar] */
          const double cmus1 = pr[CS], temp01 = 1. / pr[TEMP0], coesut = pr[XMUL0] * (1. + cmus1 * temp01);
          const double tab6 = drodm[l + 5 * nd] * (coefW * coe[l]);
          const double nu_old = rop[l + 5 * nd];
          double anulam = coesut * sqrt(rop_1[l + 4 * nd] * temp01) / (1. + cmus1 / rop_1[l + 4 * nd]);
          anulam = anulam * r_1 * pr[18];
          double t6 = (ro_old * nu_old + tab6) * r_1;
          t6 = dmin(t6, anulam);
          rop_1[l + 5 * nd] = dmax(t6, 0.);
        }
      }
}

/* Spalart-Allmaras total viscosity mu + mu_t, limited to (1 + ratiom) mu
 * (Fast/FastS/FastS/Compute/SA/vispalart.for, mulam.for, SA/chi.for, SA/xmut.for) */
void fo_vispalart(const i32 *p, const double *pr, const i32 *ind, const double *rop, double *xmut)
{
  GEOM(p)
  const double cmus1 = pr[CS], temp01 = 1. / pr[TEMP0], coesut = pr[XMUL0] * (1. + cmus1 * temp01);
  const double SA_CV1 = 357.911;
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k);
        const double T = rop[l + 4 * nd];
        const double amulam = coesut * sqrt(T * temp01) / (1. + cmus1 / T);
        const double anulam = rop[l] / amulam;
        const double anutild = dmax(rop[l + 5 * nd], 1.e-27);
        const double chi = dmax(anutild * anulam, 1.e-7);
        const double fvv1 = 1. / (1. + SA_CV1 / (chi * chi * chi));
        const double amut = rop[l] * anutild * fvv1;
        const double xmuprov = dmax(amut, 0.);
        xmut[l] = amulam + dmin(xmuprov, pr[18] * amulam);
      }
}

/* Source term of the nu_tilde equation on a Cartesian zone: production - destruction + the
 * cb2 gradient term, multiplied by the cell volume (Fast/FastS/FastS/Compute/SA/spsource_SA.for:185-232,
 * SA/sourceSA_grad_3dcart.for, SA/sourceSA_prod_dest.for; SA_LOW_RE = SA_ROT_CORR = 0 are the switches
 * read from Parameter_int).  With the implicit scheme, at the first sub-iteration, the diagonal of the
 * 6th equation is stored in coe(:,6) (SA/sourceSA_LU.for). */
void fo_spsource_SA(const i32 *p, const double *pr, int nitcfg, const i32 *ind, const double *rop, double *coe,
                    const double *ti, const double *tj, const double *tk, const double *vol, const double *dlng,
                    double *drodm)
{
  GEOM(p)
  const double SA_CKARM = 0.41, SA_CB1 = 0.1355, SA_CB2 = 0.622, SA_CV1 = 357.911, SA_CW2 = 0.3, SA_CW3 = 64.;
  const double SA_CW4 = 0.21, SA_CW5 = 1.5, SA_CROT = 2.0, SA_SIGMA = (2. / 3.);
  const double sw_lre = (double)p[130], sw_rot = (double)p[131];
  const double cmus1 = pr[CS], temp01 = 1. / pr[TEMP0], coesut = pr[XMUL0] * (1. + cmus1 * temp01);
  const double c1 = SA_CB2 / SA_SIGMA;
  const double cw1 = (SA_CB1 / SA_CKARM / SA_CKARM) + (1. + SA_CB2) / SA_SIGMA;
  const long incj = ni, inck = ninj;
  const int cart = p[ITYPZONE] == 2, twod = p[ITYPZONE] == 3, homo = p[ITYPZONE] == 1;
  const double tix = ti[0], tjy = tj[0], tkz = cart ? tk[0] : 0.;
  double xvol = 0.5 / vol[0];
  const long ndm = p[NDIMDX_MTR];
  const long minc[3] = {p[NIJK_MTR], p[NIJK_MTR + 1], p[NIJK_MTR + 2]};
  const double *T[3] = {ti, tj, tk};
  const int store = p[ITYPCP] <= 1 && nitcfg <= 1;
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k);
        const long l1 = l + 1, l2 = l - 1, l3 = l + incj, l4 = l - incj;
        const long l5 = twod ? l : l + inck, l6 = twod ? l : l - inck;   /* no k neighbours on a 2-D zone */
        const long lt = cart ? 0 : INDMTR(p, i, j, k);
        /* surface vectors of the six faces of the cell: t0[d][m] at lt, t1[d][m] at lt + inc_mtr(d)
         * (SA/sourceSA_grad_3d.for:2-21, sourceSA_grad_2d.for: no z component, no k faces) */
        double t0[3][3], t1[3][3];
        if (!cart) {
          for (int d = 0; d < 3; d++)
            for (int m = 0; m < 3; m++) {
              const int have = twod ? (d < 2 && m < 2) : 1;
              t0[d][m] = have ? T[d][lt + m * ndm] : 0.;
              t1[d][m] = have ? T[d][lt + minc[d] + m * ndm] : 0.;
            }
          xvol = 0.5 / vol[lt];
        }
#define GRAD(u1, u2, u3, u4, u5, u6, m) \
  (twod ? ((u3) * t1[0][m] - (u1) * t0[0][m] + (u2) * t1[1][m] - (u4) * t0[1][m])    \
        : ((u3) * t1[0][m] - (u1) * t0[0][m] + (u2) * t1[1][m] - (u4) * t0[1][m] + (u5) * t1[2][m] - (u6) * t0[2][m]))
        double g[3][3];   /* g[v][d]: d(u,v,w)/d(x,y,z) times 2*vol */
        for (int v = 0; v < 3; v++) {
          const double *q = rop + (1 + v) * nd;
          const double u1 = (q[l] + q[l2]), u2 = (q[l] + q[l3]), u3 = (q[l] + q[l1]);
          const double u4 = (q[l] + q[l4]), u5 = (q[l] + q[l5]), u6 = (q[l] + q[l6]);
          if (cart) { g[v][0] = (u3 - u1) * tix; g[v][1] = (u2 - u4) * tjy; g[v][2] = (u5 - u6) * tkz; }
          else if (twod && v == 2) { g[v][0] = g[v][1] = g[v][2] = 0.; }
          else for (int m = 0; m < 3; m++) g[v][m] = (twod && m == 2) ? 0. : GRAD(u1, u2, u3, u4, u5, u6, m);
          if (homo) g[v][2] = (u5 - u6) * t0[2][2];   /* SA/sourceSA_grad_3dhomo.for */
        }
        const double rotx = g[2][1] - g[1][2], roty = g[0][2] - g[2][0], rotz = g[1][0] - g[0][1];
        const double rot = (twod ? fabs(rotz) : sqrt(rotx * rotx + roty * roty + rotz * rotz)) * xvol;
        const double S11 = g[0][0], S22 = g[1][1], S33 = g[2][2];
        const double S12 = 0.5 * (g[0][1] + g[1][0]), S13 = 0.5 * (g[0][2] + g[2][0]), S23 = 0.5 * (g[1][2] + g[2][1]);
        double St = twod ? S11 * S11 + S22 * S22 + 2 * (S12 * S12)
                         : S11 * S11 + S22 * S22 + S33 * S33 + 2 * (S12 * S12) + 2 * (S13 * S13) + 2 * (S23 * S23);
        St = sqrt(2 * St) * xvol;
        const double *N = rop + 5 * nd, *R = rop;
        double u1 = (N[l] + N[l2]), u2 = (N[l] + N[l3]), u3 = (N[l] + N[l1]);
        double u4 = (N[l] + N[l4]), u5 = (N[l] + N[l5]), u6 = (N[l] + N[l6]);
        double dudx, dudy, dudz;
        if (cart) { dudx = (u3 - u1) * tix; dudy = (u2 - u4) * tjy; dudz = (u5 - u6) * tkz; }
        else { dudx = GRAD(u1, u2, u3, u4, u5, u6, 0); dudy = GRAD(u1, u2, u3, u4, u5, u6, 1); dudz = twod ? 0. : GRAD(u1, u2, u3, u4, u5, u6, 2); }
        if (homo) dudz = (u5 - u6) * t0[2][2];
        const double dudz0 = dudz;
        u1 = N[l] * R[l] + N[l2] * R[l2]; u2 = N[l] * R[l] + N[l3] * R[l3]; u3 = N[l] * R[l] + N[l1] * R[l1];
        u4 = N[l] * R[l] + N[l4] * R[l4]; u5 = N[l] * R[l] + N[l5] * R[l5]; u6 = N[l] * R[l] + N[l6] * R[l6];
        if (cart) { dudx = dudx * (u3 - u1) * tix; dudy = dudy * (u2 - u4) * tjy; dudz = dudz * (u5 - u6) * tkz; }
        else {
          dudx = dudx * GRAD(u1, u2, u3, u4, u5, u6, 0); dudy = dudy * GRAD(u1, u2, u3, u4, u5, u6, 1);
          dudz = twod ? 0. : dudz * GRAD(u1, u2, u3, u4, u5, u6, 2);
          if (homo) dudz = dudz0 * (u5 - u6) * t0[2][2];
        }
        const double anvisc = twod ? c1 * (dudx + dudy) * xvol * xvol : c1 * (dudx + dudy + dudz) * xvol * xvol;
        const double T = rop[l + 4 * nd];
        const double amulam = coesut * sqrt(T * temp01) / (1. + cmus1 / T);
        const double anulam = rop[l] / amulam;
        const double anutild = dmax(N[l], 1.e-27);
        const double chi = dmax(anutild * anulam, 1.e-7);
        const double f1 = 1. / (1. + SA_CV1 / (chi * chi * chi));
        const double f2 = 1. - chi / (1. + chi * f1);
        const double dist = dmax(dlng[l], 1.e-27);
        double stild = rot + (anutild * f2) / (SA_CKARM * SA_CKARM * dist * dist);
        stild = stild + sw_rot * SA_CROT * dmin(0.00000000000000000001, St - rot);
        const double prod = rop[l] * SA_CB1 * stild * anutild;
        stild = dmax(stild, 0.00000000000000000001);
        double r = anutild / (stild * SA_CKARM * SA_CKARM * dist * dist);
        r = dmin(r, 10.);
        const double r5 = r * r * r * r * r - 1.;
        double gg = r * (1. + SA_CW2 * r5);
        const double cw2_lre = SA_CW4 + SA_CW5 / ((1. + chi / 40.) * (1. + chi / 40.));
        gg = sw_lre * r * (1. + cw2_lre * r5) + (1. - sw_lre) * gg;
        const double g6 = gg * gg * gg * gg * gg * gg;   /* s**6 */
        const double fwg = gg * pow((1. + SA_CW3) / (g6 + SA_CW3), 1. / 6.);
        const double destruc = rop[l] * cw1 * fwg * (anutild / dist) * (anutild / dist);
        const double tsource = (prod + anvisc - destruc);
        if (store) {
          const double tsourceb = prod - destruc;
          double amutild = anutild * rop[l];
          amutild = dmax(amutild, 0.0000000000000001);
          const double tsourcenu = vol[lt] * tsourceb / amutild;
          coe[l + 5 * nd] = coe[l + 4 * nd] - dmin(tsourcenu, 0.) * coe[l];
        }
        drodm[l + 5 * nd] = drodm[l + 5 * nd] + vol[lt] * tsource;
      }
#undef GRAD
}

/* Ghost range of a BC window, no corner overflow (npass = 0)
 * (Fast/FastC/FastC/BC/indice_cl_sdm.for:46-99).  Returns 1 when the window is skipped. */
int fo_indice_cl(int idir, const i32 *p, const i32 *fen, const i32 *sdm, i32 *cl)
{
  const int ific = p[NIJK + 3], ifick = p[NIJK + 4];
  if (!(sdm[0] <= fen[1] && sdm[1] >= fen[0] && sdm[2] <= fen[3] && sdm[3] >= fen[2] &&
        sdm[4] <= fen[5] && sdm[5] >= fen[4]))
    return 1;
  for (int d = 0; d < 3; d++) {
    cl[2 * d] = fen[2 * d] > sdm[2 * d] ? fen[2 * d] : sdm[2 * d];
    cl[2 * d + 1] = fen[2 * d + 1] < sdm[2 * d + 1] ? fen[2 * d + 1] : sdm[2 * d + 1];
  }
  switch (idir) {
    case 1: cl[1] = cl[0] - 1; cl[0] = cl[0] - ific; break;
    case 2: cl[1] = cl[0] + ific; cl[0] = cl[0] + 1; break;
    case 3: cl[3] = cl[2] - 1; cl[2] = cl[2] - ific; break;
    case 4: cl[3] = cl[2] + ific; cl[2] = cl[2] + 1; break;
    case 5: cl[5] = cl[4] - 1; cl[4] = cl[4] - ifick; break;
    default: cl[5] = cl[4] + ifick; cl[4] = cl[4] + 1; break;
  }
  return 0;
}

/* BCautoperiod: ghost = value one period inside
 * (Fast/FastS/FastS/BC/bvbs_periodique.for:150-171, BCPeriodic_i.for, BCPeriodic.for) */
void fo_bc_periodic(const i32 *p, int idir, const i32 *cl, double *rop)
{
  GEOM(p)
  long incp;
  switch (idir) {
    case 1: incp = p[IJKV]; break;
    case 2: incp = -(long)p[IJKV]; break;
    case 3: incp = (long)p[IJKV + 1] * ni; break;
    case 4: incp = -(long)p[IJKV + 1] * ni; break;
    case 5: incp = (long)p[IJKV + 2] * ninj; break;
    default: incp = -(long)p[IJKV + 2] * ninj; break;
  }
  for (int k = cl[4]; k <= cl[5]; k++)
    for (int j = cl[2]; j <= cl[3]; j++)
      for (int i = cl[0]; i <= cl[1]; i++) {
        long l = INDDM(i, j, k);
        for (int e = 0; e < p[NEQ]; e++) rop[l + e * nd] = rop[l + incp + e * nd];
      }
}

/* Geometry of a ghost window for the BCs that look along the face normal: for every face cell
 * (a,b) the ghost cells g_1, g_2 (rank 1 next to the face) and the interior cells in_1, in_2, ...
 * inc = stride pointing from a ghost cell towards the interior (inci/incj/inck of the
 * reference loops, e.g. Fast/FastS/FastS/BC/bvbs_farfield.for:150-153,198-201). */
typedef struct {
  int d;        /* normal direction 0,1,2 */
  int first;    /* index of the first interior cell along d (1 or iv/jv/kv) */
  int sgn;      /* +1 on a min face, -1 on a max face: ghost rank r sits at first - sgn*r */
  int nr;       /* number of ghost ranks in the window */
} fo_face;

static fo_face fo_face_of(const i32 *p, int idir)
{
  fo_face f;
  f.d = (idir - 1) / 2;
  f.sgn = (idir % 2 == 1) ? 1 : -1;
  f.first = f.sgn > 0 ? 1 : p[IJKV + f.d];
  f.nr = f.d == 2 ? p[NIJK + 4] : p[NIJK + 3];
  return f;
}

/* BCExtrapolate: every ghost rank = first interior cell
 * (Fast/FastS/FastS/BC/bvbs_extrapolate.for, BCExtrapolate_i.for, BCExtrapolate.for with
 * c1 = 1, vmax = -1e30 from BCExtrapolate_init.for) */
static void fo_bc_extrapolate_c(const i32 *p, int idir, const i32 *cl, double *rop, double c1);
void fo_bc_extrapolate(const i32 *p, int idir, const i32 *cl, double *rop) { fo_bc_extrapolate_c(p, idir, cl, rop, 1.); }

/* c1 = 1 on the state; c1 = 0 when the routine is applied to the Newton residual (lrhs = 1,
 * BCExtrapolate_init.for:2-6): the ghost residual becomes (signed) zero */
static void fo_bc_extrapolate_c(const i32 *p, int idir, const i32 *cl, double *rop, double c1)
{
  GEOM(p)
  const fo_face f = fo_face_of(p, idir);
  const double vmax = -1.e30;
  for (int k = cl[4]; k <= cl[5]; k++)
    for (int j = cl[2]; j <= cl[3]; j++)
      for (int i = cl[0]; i <= cl[1]; i++) {
        int q[3] = {i, j, k};
        q[f.d] = f.first;
        long l = INDDM(i, j, k), l1 = INDDM(q[0], q[1], q[2]);
        for (int e = 0; e < p[NEQ]; e++) rop[l + e * nd] = dmax(vmax, rop[l1 + e * nd]) * c1;
      }
}

/* BCExtrapolateRANS (RANS / LES interface): only nu_tilde is extrapolated, bounded from below by RONUTILDEINF; the
 * mean-flow ghost values are left to the join (Fast/FastS/FastS/BC/bvbs_extrapolate_ransles.for,
 * BCExtrapolateRansLes_init.for, BCExtrapolateRansLesSA_i.for; with 5 equations the routine changes nothing) */
void fo_bc_extrapolate_nut(const i32 *p, const double *pr, int idir, const i32 *cl, double *rop)
{
  GEOM(p)
  if (p[NEQ] != 6) return;
  const fo_face f = fo_face_of(p, idir);
  const double vmax = pr[8];
  for (int k = cl[4]; k <= cl[5]; k++)
    for (int j = cl[2]; j <= cl[3]; j++)
      for (int i = cl[0]; i <= cl[1]; i++) {
        int q[3] = {i, j, k};
        q[f.d] = f.first;
        long l = INDDM(i, j, k), l1 = INDDM(q[0], q[1], q[2]);
        rop[l + 5 * nd] = dmax(vmax, rop[l1 + 5 * nd]) * 1.;
      }
}

/* Adiabatic no-slip wall on a fixed mesh: ghost rank r mirrors interior rank r with the
 * velocity reversed (Fast/FastS/FastS/BC/bvbs_wall_viscous_adia.for:118-160,
 * BCWallViscous_i.for with c_ale = 0: rop(l,2) = 2*ventx*c_ale - u). */
void fo_bc_wall_viscous(const i32 *p, int idir, const i32 *cl, double *rop)
{
  GEOM(p)
  const fo_face f = fo_face_of(p, idir);
  const int ref = 2 * f.first - f.sgn;   /* iref = 2*ind_loop(2)+1 (min) or 2*ind_loop(1)-1 (max) */
  for (int k = cl[4]; k <= cl[5]; k++)
    for (int j = cl[2]; j <= cl[3]; j++)
      for (int i = cl[0]; i <= cl[1]; i++) {
        int q[3] = {i, j, k};
        q[f.d] = ref - q[f.d];
        long l = INDDM(i, j, k), lm = INDDM(q[0], q[1], q[2]);
        rop[l] = rop[lm];
        rop[l + nd] = 0. - rop[lm + nd];
        rop[l + 2 * nd] = 0. - rop[lm + 2 * nd];
        rop[l + 3 * nd] = 0. - rop[lm + 3 * nd];
        rop[l + 4 * nd] = rop[lm + 4 * nd];
        if (p[NEQ] == 6) rop[l + 5 * nd] = -rop[lm + 5 * nd];   /* BCWallViscous_SA.for */
      }
}

/* Face metric seen by the BC routines for the face cell (i,j,k) (first interior cell of the
 * column): tijk(lmtr, ic)*ci_mtr, ... of Fast/FastC/FastC/Compute/shape_tab_mtr.for:14-52.  On a
 * Cartesian zone (neq_mtr = 1) only the component along the face normal is kept; on a curvilinear
 * zone (neq_mtr = 3) the three components of the boundary face are read at
 * lmtr = indmtr(face) with face = first cell (min side) or first cell + 1 (max side). */
static void fo_bc_metric(const i32 *p, int idir, int i, int j, int k, const double *ti, const double *tj,
                         const double *tk, double *tc)
{
  if (FULL3D(p[ITYPZONE]) || p[ITYPZONE] == 3) {
    const double *t = idir <= 2 ? ti : (idir <= 4 ? tj : tk);
    int q[3] = {i, j, k};
    if (idir % 2 == 0) q[(idir - 1) / 2] += 1;
    const long lm = INDMTR(p, q[0], q[1], q[2]), ndm = p[NDIMDX_MTR];
    tc[0] = t[lm] * 1.; tc[1] = t[lm + ndm] * 1.;
    tc[2] = p[ITYPZONE] == 3 ? t[lm + ndm] * 0. : t[lm + 2 * ndm] * 1.;   /* neq_mtr = 2: kc = 2, ck_mtr = 0 */
    return;
  }
  if (idir <= 2) { tc[0] = ti[0] * 1.; tc[1] = ti[0] * 0.; tc[2] = ti[0] * 0.; }
  else if (idir <= 4) { tc[0] = tj[0] * 0.; tc[1] = tj[0] * 1.; tc[2] = tj[0] * 0.; }
  else { tc[0] = tk[0] * 0.; tc[1] = tk[0] * 0.; tc[2] = tk[0] * 1.; }
}

/* Slip wall: the velocity of the mirror cell is reflected about the face
 * (Fast/FastS/FastS/BC/bvbs_wall_inviscid.for, BCWallInviscid_i.for,
 * Fast/FastC/FastC/BC/BCWallInviscid.for:1-49 with c_ale = 0) */
void fo_bc_wall_inviscid(const i32 *p, int idir, const i32 *cl, double *rop, const double *ti,
                         const double *tj, const double *tk)
{
  GEOM(p)
  const fo_face f = fo_face_of(p, idir);
  const int ref = 2 * f.first - f.sgn;
  double tc[3];
  for (int k = cl[4]; k <= cl[5]; k++)
    for (int j = cl[2]; j <= cl[3]; j++)
      for (int i = cl[0]; i <= cl[1]; i++) {
        int q[3] = {i, j, k};
        q[f.d] = f.first;
        fo_bc_metric(p, idir, q[0], q[1], q[2], ti, tj, tk, tc);
        q[f.d] = ref - (f.d == 0 ? i : (f.d == 1 ? j : k));
        long l = INDDM(i, j, k), lm = INDDM(q[0], q[1], q[2]);
        double tcx = tc[0], tcy = tc[1], tcz = tc[2];
        double surf = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
        surf = dmax(surf, 1e-30);
        double s_1 = 1. / surf;
        tcx = tcx * s_1; tcy = tcy * s_1; tcz = tcz * s_1;
        double u = rop[lm + nd] - 0., v = rop[lm + 2 * nd] - 0., w = rop[lm + 3 * nd] - 0.;
        double qn = (u * tcx + v * tcy + w * tcz);
        double ut = u - qn * tcx, vt = v - qn * tcy, wt = w - qn * tcz;
        double ua = 2. * ut - u, va = 2. * vt - v, wa = 2. * wt - w;
        rop[l] = rop[lm];
        rop[l + nd] = 0. + ua;
        rop[l + 2 * nd] = 0. + va;
        rop[l + 3 * nd] = 0. + wa;
        rop[l + 4 * nd] = rop[lm + 4 * nd];
        if (p[NEQ] == 6) rop[l + 5 * nd] = rop[lm + 5 * nd];    /* BCWallInviscidSA_i.for */
      }
}

/* Far field: characteristic (non-reflecting) state in the first ghost rank from the first interior
 * cell and the reference state, parabolic extrapolation with a positivity guard in the next ranks
 * (Fast/FastS/FastS/BC/bvbs_farfield.for:118-196, BCFarfield_i.for, BCFarfield.for,
 * non_reflection.for:1-153, BC_nextrank_i.for, BC_nextrank.for; c_ale = 0). */
#define VXINF 19
#define VYINF 20
#define VZINF 21
#define PINF 4
static void fo_bc_farfield_m(const i32 *p, const double *pr, int idir, const i32 *cl, double *rop, const double *ti,
                             const double *tj, const double *tk, int mode, const double *data, int dsize,
                             const int *inc_bc);
void fo_bc_farfield(const i32 *p, const double *pr, int idir, const i32 *cl, double *rop, const double *ti,
                    const double *tj, const double *tk)
{
  fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, 0, 0, 0, 0);
}

/* mode 0: BCFarfield.  mode 1: BCInflow without data = the same characteristic state from the reference state
 * (bvbs_inflow.for, BCInflow_i.for, BCInflow_init.for), nu_tilde always taken from the state (BCInflowSA_i.for).
 * mode 2: BCInflowSupersonic without data: the first ghost rank takes the reference state
 * (bvbs_inflow_supersonic.for:127-135, BCInflowSupersonic_i.for), the next ranks are extrapolated as for the
 * far field.  mode 2 with data: the first ghost rank takes the conservative state stored per face cell of the BC
 * window, dsize values per variable (bvbs_inflow_supersonic_fich.for, BCInflowSupersonic_fich_firstrank.for,
 * BCInflowSupersonic_fich_firstrank_SA.for).
 * mode 4: BCOutflow (bvbs_outflow.for, BCOutflow.for), five equations (with six the reference reads a variable it
 * never sets, BCOutflow_firstrank.for:9). */
static void fo_bc_farfield_m(const i32 *p, const double *pr, int idir, const i32 *cl, double *rop, const double *ti,
                             const double *tj, const double *tk, int mode, const double *data, int dsize,
                             const int *inc_bc)
{
  GEOM(p)
  const fo_face f = fo_face_of(p, idir);
  const long strides[3] = {1, ni, ninj};
  const long inc = f.sgn * strides[f.d];
  const double snorm = (idir % 2 == 0) ? 1. : -1.;
  const double gam1 = pr[GAMMA] - 1., gam1_1 = 1. / gam1, cvinv = 1. / pr[CVINF];
  const double c4 = 5. / 6., c5 = 2. / 6., c6 = -1. / 6.;
  const double c0 = 1. / c6, c1 = -(c4 + c5) * c0, c2 = -c6 * c0, c3 = (2. - c5 - c4) * c0;
  /* default state of BCzone.cpp:26-38; dsize < 0: the BC carries its own conservative state (6 values, :40-48) */
  double st[5], ronut = pr[8];
  st[0] = pr[ROINF]; st[1] = pr[VXINF] * pr[ROINF]; st[2] = pr[VYINF] * pr[ROINF]; st[3] = pr[VZINF] * pr[ROINF];
  st[4] = pr[PINF] / (pr[GAMMA] - 1.) + 0.5 * (pr[VXINF] * st[1] + pr[VYINF] * st[2] + pr[VZINF] * st[3]);
  if (dsize < 0) {
    for (int e = 0; e < 5; e++) st[e] = data[e];
    ronut = data[5];
    data = 0;
  }
  double roe = st[0], rue = st[1], rve = st[2], rwe = st[3], ete = st[4];
  double tc[3];
  /* transverse window */
  int lo[3] = {cl[0], cl[2], cl[4]}, hi[3] = {cl[1], cl[3], cl[5]};
  lo[f.d] = hi[f.d] = f.first;
  for (int k = lo[2]; k <= hi[2]; k++)
    for (int j = lo[1]; j <= hi[1]; j++)
      for (int i = lo[0]; i <= hi[0]; i++) {
        const long l1 = INDDM(i, j, k);      /* first interior cell */
        long l = l1 - inc;                   /* first ghost rank */
        fo_bc_metric(p, idir, i, j, k, ti, tj, tk, tc);
        if (mode == 3) {
          /* BCInj1: stagnation pressure pa, stagnation enthalpy ha and direction d0 imposed, the outgoing
           * characteristic taken from inside, Newton on the normal velocity (bvbs_inflow_newton.for, BCInj1_i.for,
           * BCInflow_newton_{1,2,3}_firstrank.for, linearization_state.for; fixed mesh: qen = 0) */
          const int q[3] = {i, j, k};
          const int da = f.d == 0 ? 1 : 0, db = f.d == 2 ? 1 : 2;
          const long li = (q[da] - inc_bc[1]) + (long)(q[db] - inc_bc[2]) * inc_bc[0];
          const double d0x = data[li], d0y = data[li + dsize], d0z = data[li + 2 * dsize];
          const double pa = data[li + 3 * dsize], ha = data[li + 4 * dsize];
          const double gam = pr[GAMMA], gamm1 = gam - 1., gam2 = gam / gamm1, rgp = pr[CVINF] * gamm1;
          const double tcx = tc[0] * snorm, tcy = tc[1] * snorm, tcz = tc[2] * snorm;
          const double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz), s_1 = 1. / sn;
          double tnx, tny, tnz;
          if (sn == 0.) { tnx = 1.; tny = 0.; tnz = 0.; } else { tnx = tcx * s_1; tny = tcy * s_1; tnz = tcz * s_1; }
          const double roi = rop[l1], ui = rop[l1 + nd], vi = rop[l1 + 2 * nd], wi = rop[l1 + 3 * nd], Ti = rop[l1 + 4 * nd];
          const double pi = roi * rgp * Ti, p0 = pi;
          const double qen = 0.;
          const double vni = ui * tnx + vi * tny + wi * tnz;
          const double roc0 = sqrt(roi * gam * p0);
          const double usd0n = 1. / (d0x * tnx + d0y * tny + d0z * tnz), usd0n2 = usd0n * usd0n;
          const double wni = vni - qen, ri = pi + roc0 * wni;
          double wng = ui * tnx + vi * tny + wi * tnz - qen;
          double residug = 1.e+20;
          int nitnwt = 0;
          while (residug > pr[37] && nitnwt < p[65]) {
            nitnwt++;
            residug = 0.;
            double b = 1. - ((wng + qen) * (wng + qen)) * usd0n2 / (2. * ha);
            b = dmax(0.2, b);
            const double pg = pa * pow(b, gam2);
            const double rog = gam2 * pg / (ha * b);
            const double fn = pg + roc0 * wng - ri;
            const double df = roc0 - rog * (wng + qen) * usd0n2;
            const double dwng = -fn / df;
            wng = wng + dwng;
            residug = dmax(residug, fabs(dwng / wng));
          }
          const double vng = wng + qen, vg = vng * usd0n;
          double b = 1. - (vng * vng) * usd0n2 / (2. * ha);
          b = dmax(0.2, b);
          const double pg = pa * pow(b, gam2), rog = gam2 * pg / (ha * b), Tg = pg / (rog * rgp);
          rop[l] = rog; rop[l + nd] = vg * d0x; rop[l + 2 * nd] = vg * d0y; rop[l + 3 * nd] = vg * d0z;
          rop[l + 4 * nd] = Tg;
          if (p[NEQ] == 6) rop[l + 5 * nd] = data[li + 5 * dsize];
        } else if (mode == 5) {
          /* BCInjMFR: direction d0, mass flow rate per unit area qp and stagnation enthalpy ha imposed
           * (bvbs_injmfr.for, BCInjMFR.for, linearization_state.for; fixed mesh: qen = 0).  A Newton iteration on the
           * pressure starts a flow at rest, otherwise the normal velocity is a root of a quadratic. */
          const int q[3] = {i, j, k};
          const int da = f.d == 0 ? 1 : 0, db = f.d == 2 ? 1 : 2;
          const long li = (q[da] - inc_bc[1]) + (long)(q[db] - inc_bc[2]) * inc_bc[0];
          const double d0x = data[li], d0y = data[li + dsize], d0z = data[li + 2 * dsize];
          const double qp = data[li + 3 * dsize], ha = data[li + 4 * dsize];
          const double gam = pr[GAMMA], gamm1 = gam - 1., gam2 = gam / gamm1, rgp = pr[CVINF] * gamm1;
          const double tcx = tc[0] * snorm, tcy = tc[1] * snorm, tcz = tc[2] * snorm;
          const double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz), s_1 = 1. / sn;
          double tnx, tny, tnz;
          if (sn == 0.) { tnx = 1.; tny = 0.; tnz = 0.; } else { tnx = tcx * s_1; tny = tcy * s_1; tnz = tcz * s_1; }
          const double roi = rop[l1], Ti = rop[l1 + 4 * nd];
          const double pi = roi * rgp * Ti;
          const double ro0 = roi, u0 = rop[l1 + nd], v0 = rop[l1 + 2 * nd], w0 = rop[l1 + 3 * nd], p0 = ro0 * rgp * Ti;
          const double qen = 0.;
          const double roc0 = sqrt(ro0 * gam * p0);
          const double usd0n = 1 / (d0x * tnx + d0y * tny + d0z * tnz), usd0n2 = usd0n * usd0n;
          double wn = u0 * tnx + v0 * tny + w0 * tnz - qen;
          if (fabs(wn) < 1.e-6) {
            double residug = 1.e+20;
            int nitnwt = 0;
            double mach = sqrt((u0 * u0) + (v0 * v0) + (w0 * w0));
            mach = mach * ro0 / roc0;
            const double pt = p0 * pow(1 + 0.2 * (mach * mach), gam2);
            const double ri = pi + roc0 * wn;
            while (residug > pr[37] && nitnwt < p[65]) {
              nitnwt++;
              residug = 0.;
              const double b = dmax(0.2, 1. - ((wn + qen) * (wn + qen)) * usd0n2 / (2. * ha));
              const double pg = pt * pow(b, gam2);
              const double rog = gam2 * pg / (ha * b);
              const double fn = pg + roc0 * wn - ri;
              const double df = roc0 - rog * (wn + qen) * usd0n2;
              const double dwng = -fn / df;
              wn = wn + dwng;
              residug = dmax(residug, fabs(dwng / wn));
            }
            const double qn = wn + qen;
            const double b = dmax(0.2, 1. - (qn * qn) * usd0n2 / (2. * ha));
            const double pg = pt * pow(b, gam2), rog = gam2 * pg / (ha * b), Tg = pg / (rog * rgp);
            const double w = qn * usd0n;
            rop[l] = rog; rop[l + nd] = w * d0x; rop[l + 2 * nd] = w * d0y; rop[l + 3 * nd] = w * d0z;
            rop[l + 4 * nd] = Tg;
          } else {
            const double coefa = 0.5 * qp * usd0n2 / gam2 + roc0;
            const double coefb = p0 + roc0 * (u0 * tnx + v0 * tny + w0 * tnz - qen);
            const double coefc = -ha * qp / gam2;
            const double delta = coefb * coefb - 4. * coefa * coefc;
            const double qn = (-coefb + sqrt(delta)) / (2. * coefa) + qen;
            const double w = qn * fabs(usd0n);
            rop[l] = qp / qn; rop[l + nd] = w * d0x; rop[l + 2 * nd] = w * d0y; rop[l + 3 * nd] = w * d0z;
            rop[l + 4 * nd] = (ha - 0.5 * w * w) / (pr[CVINF] * gam);
          }
          if (p[NEQ] == 6) rop[l + 5 * nd] = data[li + 5 * dsize];
        } else if (mode == 2 && data) {
          const int q[3] = {i, j, k};
          const int da = f.d == 0 ? 1 : 0, db = f.d == 2 ? 1 : 2;
          const long li = (q[da] - inc_bc[1]) + (long)(q[db] - inc_bc[2]) * inc_bc[0];
          const double ro = data[li], ru = data[li + dsize], rv = data[li + 2 * dsize], rw = data[li + 3 * dsize];
          const double re = data[li + 4 * dsize];
          rop[l] = ro; rop[l + nd] = ru / ro; rop[l + 2 * nd] = rv / ro; rop[l + 3 * nd] = rw / ro;
          rop[l + 4 * nd] = (re - 0.5 * (ru * ru + rv * rv + rw * rw) / ro) / (ro * pr[CVINF]);
          if (p[NEQ] == 6) rop[l + 5 * nd] = data[li + 5 * dsize] / ro;
        } else if (mode == 2) {
          const double ro_state = st[0], u_state = st[1] / st[0], v_state = st[2] / st[0], w_state = st[3] / st[0];
          const double t_state = (st[4] - 0.5 * ro_state * (u_state * u_state + v_state * v_state + w_state * w_state)) /
                                 (ro_state * pr[CVINF]);
          rop[l] = ro_state; rop[l + nd] = u_state; rop[l + 2 * nd] = v_state; rop[l + 3 * nd] = w_state;
          rop[l + 4 * nd] = t_state;
          if (p[NEQ] == 6) rop[l + 5 * nd] = ronut / st[0];
        } else {
          if (mode == 4) {   /* BCOutflow.for:1-13: the outside state is the inside one but for the pressure */
            roe = rop[l1]; rue = rop[l1 + nd] * roe; rve = rop[l1 + 2 * nd] * roe; rwe = rop[l1 + 3 * nd] * roe;
            const double roe_inv = 1. / st[0];
            ete = st[4] - 0.5 * (st[1] * st[1] + st[2] * st[2] + st[3] * st[3]) * roe_inv +
                  0.5 * (rue * rop[l1 + nd] + rve * rop[l1 + 2 * nd] + rwe * rop[l1 + 3 * nd]);
          }
          double tcx = tc[0] * snorm, tcy = tc[1] * snorm, tcz = tc[2] * snorm;
          double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
          double roi = rop[l1], rui = rop[l1 + nd] * roi, rvi = rop[l1 + 2 * nd] * roi, rwi = rop[l1 + 3 * nd] * roi;
          double eti = rop[l1 + 4 * nd] * pr[CVINF] * roi + .5 * (rui * rui + rvi * rvi + rwi * rwi) / roi;
          double qen = (0. * tcx + 0. * tcy + 0. * tcz * 0.) * 0.;
          double qref1 = .5 * (roe + roi), qref2 = .5 * (rue + rui), qref3 = .5 * (rve + rvi);
          double qref4 = .5 * (rwe + rwi), qref5 = .5 * (ete + eti);
          double r = qref1, ri = 1. / r, u = qref2 * ri, v = qref3 * ri, w = qref4 * ri;
          double pp = gam1 * (qref5 - .5 * r * (u * u + v * v + w * w));
          double c = sqrt(pr[GAMMA] * pp * ri), ci = 1. / c, s_1 = 1. / sn;
          double qn = tcx * u + tcy * v + tcz * w - qen;
          double tnx, tny;
          if (sn == 0.) { tnx = 1.; tny = 0.; } else { tnx = tcx / sn; tny = tcy / sn; }
          double eigv1 = qn, eigv2 = qn, eigv3 = qn, eigv4 = qn + sn * c, eigv5 = qn - sn * c;
          double qvar1 = roe, qvar2 = rue, qvar3 = rve, qvar4 = rwe, qvar5 = ete;
          double svar1 = qvar1, svar2 = (qvar2 - u * qvar1) * ri, svar3 = (qvar3 - v * qvar1) * ri;
          double svar4 = (qvar4 - w * qvar1) * ri;
          double svar5 = gam1 * (.5 * (u * u + v * v + w * w) * qvar1 - u * qvar2 - v * qvar3 - w * qvar4 + qvar5);
          double rvar1 = svar1 - ci * ci * svar5;
          double rvar2 = tcz * s_1 * (tnx * svar2 + tny * svar3) - svar4;
          double rvar3 = -tny * svar2 + tnx * svar3;
          double rvar4 = .5 * s_1 * (tcx * svar2 + tcy * svar3 + tcz * svar4) + .5 * ri * ci * svar5;
          double rvar5 = -.5 * s_1 * (tcx * svar2 + tcy * svar3 + tcz * svar4) + .5 * ri * ci * svar5;
          const double roext = rvar1, ruext = rvar2, rvext = rvar3, rwext = rvar4, etext = rvar5;
          qvar1 = roi; qvar2 = rui; qvar3 = rvi; qvar4 = rwi; qvar5 = eti;
          svar1 = qvar1; svar2 = (qvar2 - u * qvar1) * ri; svar3 = (qvar3 - v * qvar1) * ri;
          svar4 = (qvar4 - w * qvar1) * ri;
          svar5 = gam1 * (.5 * (u * u + v * v + w * w) * qvar1 - u * qvar2 - v * qvar3 - w * qvar4 + qvar5);
          rvar1 = svar1 - ci * ci * svar5;
          rvar2 = tcz * s_1 * (tnx * svar2 + tny * svar3) - svar4;
          rvar3 = -tny * svar2 + tnx * svar3;
          rvar4 = .5 * s_1 * (tcx * svar2 + tcy * svar3 + tcz * svar4) + .5 * ri * ci * svar5;
          rvar5 = -.5 * s_1 * (tcx * svar2 + tcy * svar3 + tcz * svar4) + .5 * ri * ci * svar5;
          if (eigv1 < 0.) rvar1 = roext;
          if (eigv2 < 0.) rvar2 = ruext;
          if (eigv3 < 0.) rvar3 = rvext;
          if (eigv4 < 0.) rvar4 = rwext;
          if (eigv5 < 0.) rvar5 = etext;
          svar1 = rvar1 + r * ci * (rvar4 + rvar5);
          svar2 = tcz * s_1 * tnx * rvar2 - tny * rvar3 + tcx * s_1 * (rvar4 - rvar5);
          svar3 = tcz * s_1 * tny * rvar2 + tnx * rvar3 + tcy * s_1 * (rvar4 - rvar5);
          svar4 = -rvar2 + tcz * s_1 * (rvar4 - rvar5);
          svar5 = r * c * (rvar4 + rvar5);
          qvar5 = .5 * (u * u + v * v + w * w) * svar1 + r * (u * svar2 + v * svar3 + w * svar4) + gam1_1 * svar5;
          if (mode == 4) {   /* BCOutflow.for:17-33; the next ranks copy the first one (BC_extrap_i.for) */
            rop[l] = rop[l1];
            const double roinv = 1. / rop[l];
            if (qn < 0) {
              rop[l + nd] = rop[l1 + nd] - qn / (sn * sn) * tcx;
              rop[l + 2 * nd] = rop[l1 + 2 * nd] - qn / (sn * sn) * tcy;
              rop[l + 3 * nd] = rop[l1 + 3 * nd] - qn / (sn * sn) * tcz;
            } else {
              rop[l + nd] = rop[l1 + nd]; rop[l + 2 * nd] = rop[l1 + 2 * nd]; rop[l + 3 * nd] = rop[l1 + 3 * nd];
            }
            rop[l + 4 * nd] = (qvar5 * roinv - .5 * (rop[l + nd] * rop[l + nd] + rop[l + 2 * nd] * rop[l + 2 * nd] +
                                                      rop[l + 3 * nd] * rop[l + 3 * nd])) * cvinv;
            for (int rk = 2; rk <= f.nr; rk++)
              for (int e = 0; e < 5; e++) rop[l1 - rk * inc + e * nd] = rop[l + e * nd];
            continue;
          }
          double roinv = 1. / svar1;
          u = u + r * svar2 * roinv; v = v + r * svar3 * roinv; w = w + r * svar4 * roinv;
          rop[l] = svar1; rop[l + nd] = u; rop[l + 2 * nd] = v; rop[l + 3 * nd] = w;
          rop[l + 4 * nd] = (qvar5 * roinv - .5 * (u * u + v * v + w * w)) * cvinv;
          if (p[NEQ] == 6)   /* BCFarfield_firstrank_SA.for, BCFarfield_init.for: nue = state(6)/state(1) */
            rop[l + 5 * nd] = (mode == 1 || eigv1 < 0.) ? ronut / roe : rop[l1 + 5 * nd];
        }
        for (int r = 2; r <= f.nr; r++) {   /* next ranks, outwards */
          l = l1 - r * inc;
          const long l0 = l + inc, l1b = l + 2 * inc, l2 = l + 3 * inc;
          const int neq = p[NEQ];
          rop[l] = rop[l0] * c3 + rop[l1b] * c1 + rop[l2] * c2;
          /* positivity guard: BC_nextrank.for (5 equations), BC_nextrankSA.for (6) */
          const int guard = neq == 6 ? rop[l] <= 0.3 * rop[l0] : fabs(rop[l] / rop[l0] - 1.) >= 0.7;
          if (guard) {
            for (int e = 0; e < neq; e++) rop[l + e * nd] = rop[l0 + e * nd];
          } else {
            for (int e = 1; e < neq; e++) rop[l + e * nd] = rop[l0 + e * nd] * c3 + rop[l1b + e * nd] * c1 + rop[l2 + e * nd] * c2;
          }
        }
      }
}

/* Window data of a BC (BCzone.cpp:257, 570-592, INCREMENT_BC.h): the first array stored for the window in
 * Parameter_real, one value per face cell of the BC window, first transverse direction fastest; a leading
 * array of size 1 is the mesh-velocity coefficient and is skipped.  Returns NULL without data. */
static int nb_of(const i32 *p) { return p[p[PT_BC]]; }
static const double *fo_bc_data(const i32 *p, const double *pr, int ndf, int *inc_bc)
{
  const int pt_bcs = p[PT_BC], nb = p[pt_bcs], pt = p[pt_bcs + 1 + ndf], idir = p[pt + 1];
  const i32 *fen = p + pt + 2;
  if (p[pt + 8] == 0) return 0;
  const double *d = pr + p[pt_bcs + 1 + ndf + nb];
  if (p[pt + 9] == 1) d += 1;
  if (idir <= 2) { inc_bc[0] = fen[3] - fen[2] + 1; inc_bc[1] = fen[2]; inc_bc[2] = fen[4]; }
  else if (idir <= 4) { inc_bc[0] = fen[1] - fen[0] + 1; inc_bc[1] = fen[0]; inc_bc[2] = fen[4]; }
  else { inc_bc[0] = fen[1] - fen[0] + 1; inc_bc[1] = fen[0]; inc_bc[2] = fen[2]; }
  return d;
}

/* Pressure outlet: first ghost rank from the interior state and the imposed pressure along the face normal
 * (Fast/FastS/FastS/BC/bvbs_outpres.for:151-197, BCOutpres_i.for, linearization_state.for), the next ranks copy
 * the first one (BC_extrap_i.for); nu_tilde copied from the interior (BCOutpresSA_i.for). */
static void fo_bc_outpres_m(const i32 *p, const double *pr, int idir, const i32 *cl, double *rop, const double *ti,
                            const double *tj, const double *tk, const double *pext, const int *inc_bc, int mfr);
void fo_bc_outpres(const i32 *p, const double *pr, int idir, const i32 *cl, double *rop, const double *ti,
                   const double *tj, const double *tk, const double *pext, const int *inc_bc)
{
  fo_bc_outpres_m(p, pr, idir, cl, rop, ti, tj, tk, pext, inc_bc, 0);
}

/* mfr = 1: BCOutMFR, the imposed quantity is the mass flow rate per unit area qp; the density of the ghost cell is a
 * root of the quadratic that the C0 and C+ characteristics give (bvbs_outmfr.for, BCOutMFR.for with qen = 0). */
static void fo_bc_outpres_m(const i32 *p, const double *pr, int idir, const i32 *cl, double *rop, const double *ti,
                            const double *tj, const double *tk, const double *pext, const int *inc_bc, int mfr)
{
  GEOM(p)
  const fo_face f = fo_face_of(p, idir);
  const long strides[3] = {1, ni, ninj};
  const long inc = f.sgn * strides[f.d];
  const double snorm = (idir % 2 == 0) ? 1. : -1.;
  const double gam = pr[GAMMA], gam1 = pr[GAMMA] - 1., rgp = pr[CVINF] * gam1;
  double tc[3];
  int lo[3] = {cl[0], cl[2], cl[4]}, hi[3] = {cl[1], cl[3], cl[5]};
  lo[f.d] = hi[f.d] = f.first;
  const int da = f.d == 0 ? 1 : 0, db = f.d == 2 ? 1 : 2;
  for (int k = lo[2]; k <= hi[2]; k++)
    for (int j = lo[1]; j <= hi[1]; j++)
      for (int i = lo[0]; i <= hi[0]; i++) {
        const int q[3] = {i, j, k};
        const long l1 = INDDM(i, j, k);
        long l = l1 - inc;
        const long li = (q[da] - inc_bc[1]) + (long)(q[db] - inc_bc[2]) * inc_bc[0];
        fo_bc_metric(p, idir, i, j, k, ti, tj, tk, tc);
        const double tcx = tc[0] * snorm, tcy = tc[1] * snorm, tcz = tc[2] * snorm;
        const double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
        const double s_1 = 1. / sn;
        double tnx, tny, tnz;
        if (sn == 0.) { tnx = 1.; tny = 0.; tnz = 0.; } else { tnx = tcx * s_1; tny = tcy * s_1; tnz = tcz * s_1; }
        const double roi = rop[l1], ui = rop[l1 + nd], vi = rop[l1 + 2 * nd], wi = rop[l1 + 3 * nd], Ti = rop[l1 + 4 * nd];
        const double pi = roi * rgp * Ti;
        const double ro0 = roi, p0 = ro0 * rgp * Ti;
        const double roc0 = sqrt(ro0 * gam * p0);
        if (mfr) {
          const double u0 = ui, v0 = vi, w0 = wi, qen = 0.;
          double c0_2 = roc0 / ro0;
          c0_2 = c0_2 * c0_2;
          const double coefa = c0_2;
          const double coefb = -roc0 * (u0 * tnx + v0 * tny + w0 * tnz - qen) - c0_2 * roi;
          const double coefc = roc0 * pext[li];
          const double delta = coefb * coefb - 4 * coefa * coefc;
          rop[l] = (-coefb + sqrt(delta)) / (2 * coefa);
          const double dp = c0_2 * (rop[l] - ro0);
          const double dqn = dp / roc0;
          rop[l + nd] = ui - dqn * tnx;
          rop[l + 2 * nd] = vi - dqn * tny;
          rop[l + 3 * nd] = wi - dqn * tnz;
          rop[l + 4 * nd] = (dp + pi) / (rop[l] * rgp);
          if (p[NEQ] == 6) rop[l + 5 * nd] = rop[l1 + 5 * nd];
          for (int r = 2; r <= f.nr; r++)
            for (int e = 0; e < p[NEQ]; e++) rop[l1 - r * inc + e * nd] = rop[l + e * nd];
          continue;
        }
        const double dqn = (pext[li] - pi) / roc0;
        const double rog = roi + ro0 * ro0 * dqn / roc0;
        rop[l] = rog;
        rop[l + nd] = ui - dqn * tnx;
        rop[l + 2 * nd] = vi - dqn * tny;
        rop[l + 3 * nd] = wi - dqn * tnz;
        rop[l + 4 * nd] = pext[li] / (rog * rgp);
        if (p[NEQ] == 6) rop[l + 5 * nd] = rop[l1 + 5 * nd];
        for (int r = 2; r <= f.nr; r++) {
          l = l1 - r * inc;
          for (int e = 0; e < p[NEQ]; e++) rop[l + e * nd] = rop[l1 - inc + e * nd];
        }
      }
}

/* Isothermal no-slip wall: mirror cell, velocity reversed, temperature 2 Tw - T, density from the wall density
 * rho_w = rho T / Tw at zero normal pressure gradient (Fast/FastS/FastS/BC/bvbs_wall_viscous_isothermal.for:151-197,
 * BCWallViscousIsothermal_firstrank.for with c_ale = 0; nu_tilde reversed, BCWallViscousIsothermal_firstrank_SA.for). */
void fo_bc_wall_isothermal(const i32 *p, int idir, const i32 *cl, double *rop, const double *tw, const int *inc_bc)
{
  GEOM(p)
  const fo_face f = fo_face_of(p, idir);
  const int ref = 2 * f.first - f.sgn;
  const int da = f.d == 0 ? 1 : 0, db = f.d == 2 ? 1 : 2;
  for (int k = cl[4]; k <= cl[5]; k++)
    for (int j = cl[2]; j <= cl[3]; j++)
      for (int i = cl[0]; i <= cl[1]; i++) {
        int q[3] = {i, j, k};
        const long li = (q[da] - inc_bc[1]) + (long)(q[db] - inc_bc[2]) * inc_bc[0];
        q[f.d] = ref - q[f.d];
        const long l = INDDM(i, j, k), lm = INDDM(q[0], q[1], q[2]);
        const double rho_w = rop[lm] * rop[lm + 4 * nd] / tw[li];
        rop[l] = 2. * rho_w - rop[lm];
        const double u = rop[lm + nd], v = rop[lm + 2 * nd], w = rop[lm + 3 * nd];
        rop[l + nd] = 0. - u;
        rop[l + 2 * nd] = 0. - v;
        rop[l + 3 * nd] = 0. - w;
        rop[l + 4 * nd] = 2. * tw[li] - rop[lm + 4 * nd];
        if (p[NEQ] == 6) rop[l + 5 * nd] = -rop[lm + 5 * nd];
      }
}

/* Edge ghosts = mean of the two neighbouring face ghosts, corner ghosts = mean of three
 * (Fast/FastS/FastS/BC/correct_coins.for:23-420), whole zone range. */
void fo_correct_coins(const i32 *p, double *rop)
{
  GEOM(p)
  const int iv = p[IJKV], jv = p[IJKV + 1], kv = p[IJKV + 2];
  const int neq = p[NEQ];
  const int ends[3][2] = {{1, iv}, {1, jv}, {1, kv}};
#define MIN2(a, b) ((a) < (b) ? (a) : (b))
  for (int idir = 0; idir < 2; idir++) { /* faces imin, imax */
    int is = idir == 0 ? -1 : 1, i0 = ends[0][idir];
    for (int jd = 0; jd < 2; jd++) { /* bands jmin, jmax */
      int js = jd == 0 ? -1 : 1, j0 = ends[1][jd];
      for (int ne = 0; ne < neq; ne++)
        for (int k = 1; k <= kv; k++)
          for (int j = 1; j <= ific; j++)
            for (int i = 1; i <= ific; i++) {
              int i1 = MIN2(ific, i + 1), j1 = MIN2(ific, j + 1);
              long l1 = INDDM(i0 + i * is, j0 + j * js, k);
              long l11 = INDDM(i0 + i1 * is, j0, k);
              long l12 = INDDM(i0, j0 + j1 * js, k);
              rop[l1 + ne * nd] = (rop[l11 + ne * nd] + rop[l12 + ne * nd]) * 0.5;
            }
    }
    for (int kd = 0; kd < 2; kd++) { /* bands kmin, kmax */
      int ks = kd == 0 ? -1 : 1, k0 = ends[2][kd];
      for (int ne = 0; ne < neq; ne++)
        for (int k = 1; k <= ifick; k++)
          for (int j = 1; j <= jv; j++)
            for (int i = 1; i <= ific; i++) {
              int k1 = MIN2(ifick, k + 1), i1 = MIN2(ific, i + 1);
              long l1 = INDDM(i0 + i * is, j, k0 + k * ks);
              long l11 = INDDM(i0, j, k0 + k1 * ks);
              long l12 = INDDM(i0 + i1 * is, j, k0);
              rop[l1 + ne * nd] = (rop[l11 + ne * nd] + rop[l12 + ne * nd]) * 0.5;
            }
    }
  }
  for (int jdir = 0; jdir < 2; jdir++) { /* faces jmin, jmax: bands kmin, kmax */
    int js = jdir == 0 ? -1 : 1, j0 = ends[1][jdir];
    for (int kd = 0; kd < 2; kd++) {
      int ks = kd == 0 ? -1 : 1, k0 = ends[2][kd];
      for (int ne = 0; ne < neq; ne++)
        for (int k = 1; k <= ifick; k++)
          for (int j = 1; j <= ific; j++)
            for (int i = 1; i <= iv; i++) {
              int k1 = MIN2(ifick, k + 1), j1 = MIN2(ific, j + 1);
              long l1 = INDDM(i, j0 + j * js, k0 + k * ks);
              long l11 = INDDM(i, j0, k0 + k1 * ks);
              long l12 = INDDM(i, j0 + j1 * js, k0);
              rop[l1 + ne * nd] = (rop[l11 + ne * nd] + rop[l12 + ne * nd]) * 0.5;
            }
    }
  }
  const double c3 = 1. / 3.;
  for (int kd = 0; kd < 2; kd++)
    for (int jd = 0; jd < 2; jd++)
      for (int id = 0; id < 2; id++) {
        int is = id == 0 ? -1 : 1, js = jd == 0 ? -1 : 1, ks = kd == 0 ? -1 : 1;
        int i0 = ends[0][id], j0 = ends[1][jd], k0 = ends[2][kd];
        for (int ne = 0; ne < neq; ne++)
          for (int k = 1; k <= ifick; k++)
            for (int j = 1; j <= ific; j++)
              for (int i = 1; i <= ific; i++) {
                int i1 = MIN2(ific, i + 1), j1 = MIN2(ific, j + 1), k1 = MIN2(ifick, k + 1);
                long l = INDDM(i0 + i * is, j0 + j * js, k0 + k * ks);
                long l11 = INDDM(i0, j0, k0 + k1 * ks);
                long l12 = INDDM(i0 + i1 * is, j0, k0);
                long l13 = INDDM(i0, j0 + j1 * js, k0);
                rop[l + ne * nd] = (rop[l11 + ne * nd] + rop[l12 + ne * nd] + rop[l13 + ne * nd]) * c3;
              }
      }
}

/* BC dispatch over the table stored from Parameter_int[PT_BC]
 * (Fast/FastS/FastS/BC/BCzone.cpp:193-226,560-568; table layout
 * Fast/FastC/FastC/PyTree.py:2958-3014), then the edge/corner fix
 * (Fast/FastS/FastS/Compute/bcs.cpp:30-41). Returns -1 on an unsupported BC tag. */
int fo_bczone(const i32 *p, const double *pr, double *rop, const double *ti, const double *tj, const double *tk)
{
  const int pt_bcs = p[PT_BC], nb = p[pt_bcs];
  i32 sdm[6] = {1, p[IJKV], 1, p[IJKV + 1], 1, p[IJKV + 2]}, cl[6];
  for (int ndf = 0; ndf < nb; ndf++) {
    int pt = p[pt_bcs + 1 + ndf], type = p[pt], idir = p[pt + 1];
    if (fo_indice_cl(idir, p, p + pt + 2, sdm, cl)) continue;
    if (type == 11) fo_bc_periodic(p, idir, cl, rop);
    else if (type == 22) continue;
    else if (type == 0) fo_bc_extrapolate(p, idir, cl, rop);
    else if (type == 14) fo_bc_extrapolate_nut(p, pr, idir, cl, rop);
    else if (type == 1 && p[pt + 8] == 0) fo_bc_farfield(p, pr, idir, cl, rop, ti, tj, tk);
    else if (type == 13 && p[pt + 8] == 0) fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, 1, 0, 0, 0);
    else if (type == 10 && p[pt + 8] == 0 && p[NEQ] == 5) fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, 4, 0, 0, 0);
    else if (type == 2 && p[pt + 8] == 0) fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, 2, 0, 0, 0);
    else if ((type == 1 || type == 13 || type == 2 || (type == 10 && p[NEQ] == 5)) &&
             (p[pt + 8] == 21 || (p[pt + 8] == 1 && p[pt + 9] == 6))) {
      /* the BC's own uniform state: one array of 6 conservative values, or a copy of a ReferenceState node
       * (21 scalars, the state at positions 3..7 and 16), BCzone.cpp:40-48 */
      const double *d = pr + p[pt_bcs + 1 + ndf + nb_of(p)];
      double stt[6];
      if (p[pt + 8] == 21) { for (int e = 0; e < 5; e++) stt[e] = d[3 + e]; stt[5] = d[16]; }
      else for (int e = 0; e < 6; e++) stt[e] = d[e];
      fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, type == 1 ? 0 : (type == 13 ? 1 : (type == 10 ? 4 : 2)), stt, -1, 0);
    }
    else if (type == 17 && p[pt + 8] == p[NEQ]) {   /* BCzone.cpp:594-618: d0x, d0y, d0z, pa, ha [, nu_tilde] */
      int inc_bc[3];
      const int save = p[pt + 9];
      const double *d = fo_bc_data(p, pr, ndf, inc_bc);
      if (save == 1) return -1;
      fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, 3, d, save, inc_bc);
    }
    else if (type == 20 && p[pt + 8] == p[NEQ]) {   /* BCzone.cpp:377-398: d0x, d0y, d0z, qp, ha [, nu_tilde] */
      int inc_bc[3];
      const int save = p[pt + 9];
      const double *d = fo_bc_data(p, pr, ndf, inc_bc);
      if (save == 1) return -1;
      fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, 5, d, save, inc_bc);
    }
    else if (type == 21 && p[pt + 8] == 1 && p[pt + 9] > 1) {   /* BCzone.cpp:400-415: qp per face cell */
      int inc_bc[3];
      const double *d = fo_bc_data(p, pr, ndf, inc_bc);
      fo_bc_outpres_m(p, pr, idir, cl, rop, ti, tj, tk, d, inc_bc, 1);
    }
    else if (type == 2 && p[pt + 8] == p[NEQ] && p[pt + 9] > 6) {   /* BCzone.cpp:421-448: arrays of more than 6 values */
      int inc_bc[3];
      const int save = p[pt + 9];
      const double *d = fo_bc_data(p, pr, ndf, inc_bc);
      fo_bc_farfield_m(p, pr, idir, cl, rop, ti, tj, tk, 2, d, save, inc_bc);
    }
    else if (type == 3 || (type == 4 && p[IFLOW] == 1) || type == 5) fo_bc_wall_inviscid(p, idir, cl, rop, ti, tj, tk);
    else if ((type == 6 || type == 4) && p[IFLOW] > 1) fo_bc_wall_viscous(p, idir, cl, rop);
    else if ((type == 16 || (type == 32 && p[IFLOW] > 1)) && p[pt + 8] >= 1 && p[pt + 8] <= 2) {
      int inc_bc[3];
      const double *d = fo_bc_data(p, pr, ndf, inc_bc);
      if (type == 16) fo_bc_outpres(p, pr, idir, cl, rop, ti, tj, tk, d, inc_bc);
      else fo_bc_wall_isothermal(p, idir, cl, rop, d, inc_bc);
    }
    else return -1;
  }
  fo_correct_coins(p, rop);
  return 0;
}

/* Wall-face flux correction of the roe scheme (Fast/FastS/FastS/BC/bfl3.for, CL_correction_flu.for): for every
 * wall window, in the order of the BC table, the Euler roe flux that the balance put through the wall face is
 * taken back (ROE/<type>/corr_fluroe_euler_o3_<type>.for with assemble_drodm_corr.for: iflow = 1, five
 * equations) and the wall flux is put in its place: pressure only on a no-slip wall (tags 4, 6: bflwall.for),
 * pressure + the mean normal mass flux on a slip wall (tag 3: bflwallslip.for).  c_ale = 0 (fixed mesh). */
void fo_bfl3(const i32 *p, const double *pr, const double *rop, double *drodm, const double *wig,
             const double *xmut, const double *ti, const double *tj, const double *tk, const double *vol)
{
  GEOM(p)
  if (p[KFLUDOM] != 5) return;
  const int pt_bcs = p[PT_BC], nb = p[pt_bcs];
  if (nb == 0) return;
  const i32 sdm[6] = {1, p[IJKV], 1, p[IJKV + 1], 1, p[IJKV + 2]};
  i32 pe[256];                 /* Euler view of the zone for the correction flux (iflow = 1) */
  memcpy(pe, p, sizeof(i32) * 137);
  pe[IFLOW] = 1;
  fluxctx c;
  fluxctx_init(&c, pe, pr, ti, tj, tk, vol);
  const double rgp = pr[CVINF] * (pr[GAMMA] - 1.), gam2 = pr[GAMMA] / (pr[GAMMA] - 1.);
  const long strides[3] = {1, ni, ninj};
  for (int ndf = 0; ndf < nb; ndf++) {
    const int pt = p[pt_bcs + 1 + ndf], type = p[pt], idir = p[pt + 1];
    if (type != 3 && type != 4 && type != 6 && type != 7 && type != 12) continue;
    i32 cl[6];
    if (fo_indice_cl(idir, p, p + pt + 2, sdm, cl)) continue;
    const int d = (idir - 1) / 2, maxside = idir % 2 == 0;
    if (!maxside) { cl[2 * d] = 1; cl[2 * d + 1] = 1; }
    else cl[2 * d + 1] = cl[2 * d];
    const long inc = strides[d], shift = maxside ? inc : 0;
    /* 1. take the roe flux of the wall face back */
    {
      const double sens = maxside ? 1. : -1.;
      double f[6];
      for (int k = cl[4]; k <= cl[5]; k++)
        for (int j = cl[2]; j <= cl[3]; j++)
          for (int i = cl[0]; i <= cl[1]; i++) {
            const long l = INDDM(i, j, k), l0 = l - shift;
            if (FULL3D(c.ityp) || c.ityp == 3) face_flux_full(&c, rop, wig, xmut, l, INDMTR(p, i, j, k), d, f);
            else face_flux(&c, rop, wig, xmut, l, d, f);
            for (int e = 0; e < 5; e++) drodm[l0 + e * nd] = drodm[l0 + e * nd] + f[e] * sens;
          }
    }
    /* 2. wall flux */
    {
      const double sens = maxside ? -1. : 1.;
      const int slip = type == 3;
      if (type != 3 && type != 4 && type != 6) continue;   /* tags 7, 12: walls with data, not restated */
      for (int k = cl[4]; k <= cl[5]; k++)
        for (int j = cl[2]; j <= cl[3]; j++)
          for (int i = cl[0]; i <= cl[1]; i++) {
            const long l = INDDM(i, j, k), iadrf = l - inc, l0 = l - shift;
            /* face metric at the loop cell: its min face is the wall face on both sides */
            double tc[3];
            {
              int q[3] = {i, j, k};
              if (maxside) q[d] -= 1;   /* fo_bc_metric adds 1 on a max face */
              fo_bc_metric(p, idir, q[0], q[1], q[2], ti, tj, tk, tc);
            }
            const double tcx = tc[0], tcy = tc[1], tcz = tc[2];
            const double qen = (0. * tcx + 0. * tcy + 0. * tcz * 0.) * 0.;
            double r = 0.5 * (rop[l] + rop[iadrf]);
            double u = 0.5 * (rop[l + nd] + rop[iadrf + nd]);
            double v = 0.5 * (rop[l + 2 * nd] + rop[iadrf + 2 * nd]);
            double w = 0.5 * (rop[l + 3 * nd] + rop[iadrf + 3 * nd]);
            const double u_int = tcx * u + tcy * v + tcz * w - qen;
            double fl[6] = {0., 0., 0., 0., 0., 0.};
            if (!slip) {
              const double pp = 0.5 * (rop[l + 4 * nd] * rop[l] + rop[iadrf + 4 * nd] * rop[iadrf]) * rgp;
              fl[0] = 0.;
              fl[1] = tcx * pp + u_int * u * r;
              fl[2] = tcy * pp + u_int * v * r;
              fl[3] = tcz * pp + u_int * w * r;
              fl[4] = pp * qen;
            } else {
              if (p[NEQ] == 6) { r = rop[iadrf]; u = rop[iadrf + nd]; v = rop[iadrf + 2 * nd]; w = rop[iadrf + 3 * nd]; }
              double pp = rop[iadrf + 4 * nd] * rop[iadrf] * rgp;
              const double h = gam2 * pp + r * (u * u + v * v + w * w) * 0.5;
              pp = 0.5 * (rop[l + 4 * nd] * rop[l] + rop[iadrf + 4 * nd] * rop[iadrf]) * rgp;
              fl[0] = u_int * r;
              fl[1] = tcx * pp + u_int * r * u;
              fl[2] = tcy * pp + u_int * r * v;
              fl[3] = tcz * pp + u_int * r * w;
              fl[4] = u_int * h + pp * qen;
              if (p[NEQ] == 6) fl[5] = u_int * r * rop[iadrf + 5 * nd];
            }
            for (int e = 0; e < p[NEQ]; e++) drodm[l0 + e * nd] = drodm[l0 + e * nd] + fl[e] * sens;
          }
    }
  }
}

/* One sub-iteration of one zone with a single cache block and no neighbouring thread:
 * ranges of Fast/FastC/FastC/HPC_LAYER/indice_boucle_ssdom.for:108-168 (first block = last
 * block), sequence of Fast/FastS/FastS/Compute/navier_stokes_struct.for:199-275,295-301,
 * 310-345,467-475 and src_term.for:59-70,258-272.  rop_in is the stage input, rop_out the
 * stage output. */
void fo_tstb3_implicit(const i32 *p, const double *pr, int first_it, const i32 *ind, const double *rop,
                       const double *xmut, double *coe, const double *ti, const double *tj, const double *tk,
                       const double *vol);
void fo_subiter(const i32 *p, const double *pr, int nitcfg, const double *rop_in, double *rop_out,
                double *drodm, double *coe, double *xmut, double *wig, const double *ti,
                const double *tj, const double *tk, const double *vol)
{
  const int iv = p[IJKV], jv = p[IJKV + 1], kv = p[IJKV + 2];
  const i32 sdm[6] = {1, iv, 1, jv, 1, kv};
  const int gk = p[NIJK + 4] > 0;   /* no ghost layer along k on a two-dimensional zone */
  const i32 grad[6] = {0, iv + 1, 0, jv + 1, 1 - gk, kv + gk};
  if (nitcfg == 1) {
    if (p[IFLOW] == 2) fo_invist(p, pr, grad, rop_in, xmut);
    if (p[IFLOW] == 3) {   /* navier_stokes_struct.for:237-245: total viscosity on the whole zone */
      const int f = p[NIJK + 3], fk = p[NIJK + 4];
      const i32 all[6] = {1 - f, iv + f, 1 - f, jv + f, 1 - fk, kv + fk};
      fo_vispalart(p, pr, all, rop_in, xmut);
    }
    double volinv = pr[DTC] / vol[0]; /* Fast/FastS/FastS/Compute/tstb3_global.for:64-78 */
    GEOM(p)
    if (p[46] == 1) {
      /* local time step with the explicit scheme (cptst3.for:77-82 -> tstb3_expli.for + dtloc.for): the same
       * CFL-limited step as coe(:,1) of the implicit scheme, on the interior cells */
      double *tmp = (double *)malloc(sizeof(double) * 5 * (size_t)nd);
      fo_tstb3_implicit(p, pr, 0, sdm, rop_in, xmut, tmp, ti, tj, tk, vol);
      for (int k = 1; k <= kv; k++)
        for (int j = 1; j <= jv; j++)
          for (int i = 1; i <= iv; i++) coe[INDDM(i, j, k)] = tmp[INDDM(i, j, k)];
      free(tmp);
    } else
    for (int k = 1; k <= kv; k++)
      for (int j = 1; j <= jv; j++)
        for (int i = 1; i <= iv; i++)
          coe[INDDM(i, j, k)] = p[ITYPZONE] != 2 ? pr[DTC] / vol[INDMTR(p, i, j, k)] : volinv;
  }
  fo_init_rhs(p, nitcfg, sdm, drodm);
  if (p[IFLOW] == 3)   /* src_term.for:87-96: dist = second slot of the viscosity array */
    fo_spsource_SA(p, pr, nitcfg, sdm, rop_in, coe, ti, tj, tk, vol, xmut + p[NDIMDX], drodm);
  if (p[SLOPE] == 6) fo_sciacovelli(p, pr, sdm, rop_in, ti, tj, tk, vol, wig);   /* src_term.for:246-255: every sub-iteration */
  if (p[KFLUDOM] == 2 && nitcfg == 1) fo_wiggle(p, sdm, rop_in, wig);
  fo_flux(p, pr, sdm, rop_in, drodm, wig, xmut, ti, tj, tk, vol);
  fo_bfl3(p, pr, rop_in, drodm, wig, xmut, ti, tj, tk, vol);
  fo_core3ark3(p, pr, nitcfg, sdm, rop_out, rop_in, drodm, coe);
}

/* One explicit time step of one stand-alone zone (all BCs local): sub-iterations with the
 * array roles of Fast/FastS/FastS/Compute/gsdr3.cpp:137-145, BCs on the stage output.
 * On return ro_p1 holds t^{n+1}; the caller rotates the levels
 * (Fast/FastS/FastS/Compute/computePT.cpp:673-677).  stage1_rhs (may be NULL) receives a copy
 * of drodm after the first stage. */
int fo_step(const i32 *p, const double *pr, double *ro, double *ro_p1, double *drodm, double *coe,
            double *xmut, double *wig, const double *ti, const double *tj, const double *tk,
            const double *vol, double *stage1_rhs)
{
  const int nssiter = p[NSSITER];
  for (int nstep = 1; nstep <= nssiter; nstep++) {
    double *in = (nstep % 2 == 1) ? ro : ro_p1;
    double *out = (nstep % 2 == 1) ? ro_p1 : ro;
    fo_subiter(p, pr, nstep, in, out, drodm, coe, xmut, wig, ti, tj, tk, vol);
    if (nstep == 1 && stage1_rhs) memcpy(stage1_rhs, drodm, sizeof(double) * p[NEQ] * (size_t)p[NDIMDX]);
    if (fo_bczone(p, pr, out, ti, tj, tk)) return -1;
  }
  return 0;
}

/* fo_subiter with the k-planes dealt to `nthreads` OpenMP threads in contiguous slabs (the decomposition the
 * reference's OpenMP layer would make for a thread topology (1,1,n), Fast/FastC/FastC/HPC_LAYER/WORK_DISTRIBUTION_BEGIN.for):
 * every phase of the sub-iteration is cell-local except the flux balance, whose scatter is restricted to the
 * slab (flux_own), so the result equals fo_subiter's bit for bit.  Global time step only. */
void fo_subiter_mt(const i32 *p, const double *pr, int nitcfg, const double *rop_in, double *rop_out,
                   double *drodm, double *coe, double *xmut, double *wig, const double *ti,
                   const double *tj, const double *tk, const double *vol, int nthreads)
{
  const int iv = p[IJKV], jv = p[IJKV + 1], kv = p[IJKV + 2];
  if (nthreads <= 1 || kv < 2 * nthreads || p[46] == 1 || p[ITYPZONE] == 3) {
    fo_subiter(p, pr, nitcfg, rop_in, rop_out, drodm, coe, xmut, wig, ti, tj, tk, vol);
    return;
  }
  const int gk = p[NIJK + 4] > 0;
  const int f = p[NIJK + 3], fk = p[NIJK + 4];
  GEOM(p)
#pragma omp parallel num_threads(nthreads)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    const int t = 0, nt = 1;
#endif
    const int klo = 1 + (int)((long)kv * t / nt), khi = (int)((long)kv * (t + 1) / nt);
    /* slab extended to the ghost planes on the first and last thread */
    const int glo = t == 0 ? 1 - gk : klo, ghi = t == nt - 1 ? kv + gk : khi;
    const int alo = t == 0 ? 1 - fk : klo, ahi = t == nt - 1 ? kv + fk : khi;
    const i32 sdm[6] = {1, iv, 1, jv, klo, khi};
    if (nitcfg == 1) {
      const i32 grad[6] = {0, iv + 1, 0, jv + 1, glo, ghi};
      const i32 all[6] = {1 - f, iv + f, 1 - f, jv + f, alo, ahi};
      if (p[IFLOW] == 2) fo_invist(p, pr, grad, rop_in, xmut);
      if (p[IFLOW] == 3) fo_vispalart(p, pr, all, rop_in, xmut);
      const double volinv = pr[DTC] / vol[0];
      for (int k = klo; k <= khi; k++)
        for (int j = 1; j <= jv; j++)
          for (int i = 1; i <= iv; i++)
            coe[INDDM(i, j, k)] = p[ITYPZONE] != 2 ? pr[DTC] / vol[INDMTR(p, i, j, k)] : volinv;
    }
    fo_init_rhs(p, nitcfg, sdm, drodm);
#pragma omp barrier
    if (p[IFLOW] == 3) fo_spsource_SA(p, pr, nitcfg, sdm, rop_in, coe, ti, tj, tk, vol, xmut + p[NDIMDX], drodm);
    if (p[SLOPE] == 6) fo_sciacovelli(p, pr, sdm, rop_in, ti, tj, tk, vol, wig);
    if (p[KFLUDOM] == 2 && nitcfg == 1) fo_wiggle(p, sdm, rop_in, wig);
#pragma omp barrier
    flux_own(p, pr, sdm, rop_in, drodm, wig, xmut, ti, tj, tk, vol, klo, khi);
#pragma omp barrier
#pragma omp single
    fo_bfl3(p, pr, rop_in, drodm, wig, xmut, ti, tj, tk, vol);
    fo_core3ark3(p, pr, nitcfg, sdm, rop_out, rop_in, drodm, coe);
  }
}

int fo_step_mt(const i32 *p, const double *pr, double *ro, double *ro_p1, double *drodm, double *coe,
               double *xmut, double *wig, const double *ti, const double *tj, const double *tk,
               const double *vol, double *stage1_rhs, int nthreads)
{
  const int nssiter = p[NSSITER];
  for (int nstep = 1; nstep <= nssiter; nstep++) {
    double *in = (nstep % 2 == 1) ? ro : ro_p1;
    double *out = (nstep % 2 == 1) ? ro_p1 : ro;
    fo_subiter_mt(p, pr, nstep, in, out, drodm, coe, xmut, wig, ti, tj, tk, vol, nthreads);
    if (nstep == 1 && stage1_rhs) memcpy(stage1_rhs, drodm, sizeof(double) * p[NEQ] * (size_t)p[NDIMDX]);
    if (fo_bczone(p, pr, out, ti, tj, tk)) return -1;
  }
  return 0;
}

/* sum of squares and max abs of coe*drodm over interior cells, per equation
 * (accumulation order of Fast/FastS/FastS/Compute/cprdu3s1_incl.for:47-71) */
void fo_residual(const i32 *p, const double *drodm, const double *coe, double *sumsq, double *maxabs)
{
  GEOM(p)
  for (int e = 0; e < 5; e++) { sumsq[e] = 0.; maxabs[e] = 0.; }
  for (int k = 1; k <= p[IJKV + 2]; k++)
    for (int j = 1; j <= p[IJKV + 1]; j++)
      for (int i = 1; i <= p[IJKV]; i++) {
        long l = INDDM(i, j, k);
        for (int e = 0; e < 5; e++) {
          double r = coe[l] * drodm[l + e * nd];
          double xp = fabs(r);
          if (xp > maxabs[e]) maxabs[e] = xp;
          sumsq[e] = sumsq[e] + r * r;
        }
      }
}

/* Kinetic energy and enstrophy sums of a Cartesian zone, Fast/FastS/FastS/POST/post.for:30-62 with one thread and one
 * cache block: velocity gradient * volume by the scatter of cp_gradu_3dcart.for:72-120 (every face between cells l-inc
 * and l carries the 4-point average (q(l)+q(l-inc))*c1 - (q(l+inc)+q(l-2inc))*c2, c1 = 7/12, c2 = 1/12
 * (post.for:26-27, grad_{i,j,k}_3dcart.for), taken from cell l and given to cell l-inc), divided by the volume
 * (cp_gradu.for:88-101), then cptaylor.for:52-74: sums over the interior cells, k, j, i with i fastest, of
 * vol, 0.5*rho*|u|^2*vol and 0.5*rho*|curl u|^2*vol.  out = {enstrophy sum, kinetic-energy sum, volume}. */
int fo_tgv_sums(const i32 *p, const double *rop, const double *ti, const double *tj, const double *tk,
                const double *vol, double *out)
{
  GEOM(p)
  if (p[ITYPZONE] != 2) return -1;
  const double c1 = 7. / 12, c2 = 1. / 12;
  const double t[3] = {ti[0], tj[0], tk[0]};
  const long inc[3] = {1, ni, ninj};
  const double volinv = 1. / vol[0];
  double tke = 0., enst = 0., cpt = 0.;
  for (int k = 1; k <= p[IJKV + 2]; k++)
    for (int j = 1; j <= p[IJKV + 1]; j++)
      for (int i = 1; i <= p[IJKV]; i++) {
        long l = INDDM(i, j, k);
        double g[3][3];   /* g[component][direction] */
        for (int c = 0; c < 3; c++)
          for (int d = 0; d < 3; d++) {
            const double *q = rop + (c + 1) * nd;
            double um = (q[l] + q[l - inc[d]]) * c1 - (q[l + inc[d]] + q[l - 2 * inc[d]]) * c2;
            long lp = l + inc[d];
            double up = (q[lp] + q[lp - inc[d]]) * c1 - (q[lp + inc[d]] + q[lp - 2 * inc[d]]) * c2;
            double a = 0. - um * t[d];
            a = a + up * t[d];
            g[c][d] = a * volinv;
          }
        cpt = cpt + vol[0];
        double rho = rop[l], u1 = rop[l + nd], u2 = rop[l + 2 * nd], u3 = rop[l + 3 * nd];
        tke = tke + 0.5 * rho * (u1 * u1 + u2 * u2 + u3 * u3) * vol[0];
        double w1 = g[2][1] - g[1][2], w2 = g[0][2] - g[2][0], w3 = g[1][0] - g[0][1];
        enst = enst + 0.5 * rho * (w1 * w1 + w2 * w2 + w3 * w3) * vol[0];
      }
  out[0] = enst; out[1] = tke; out[2] = cpt;
  return 0;
}

/* Residual accumulation per sub-block, Fast/FastS/FastS/Compute/cprdu3s1.for:60-79 (one thread: every sub-block
 * in turn) and cprdu3s1_incl.for:14-141 (all sub-blocks computed): the zone is cut in ijkv_lu(d) =
 * max(1, IJKV(d) / SIZE_SSDOM(d)) sub-blocks per direction (Fast/FastS/FastS/Compute/residus_navier.h:12-16), the last
 * one of a direction extended to the zone edge; sub-block no_rdm = i_lu + (j_lu-1)*ijkv_lu(1) + ... holds the sum of
 * squares and the largest magnitude of drodm over its cells, cells visited k, j, i (i fastest).
 * sums / maxs: [ndim_rdm][neq]. */
#define SIZE_SSDOM 30
void fo_cprdu3s1(const i32 *p, const double *drodm, double *sums, double *maxs)
{
  GEOM(p)
  const int neq = p[NEQ];
  int lu[3];
  for (int d = 0; d < 3; d++) { lu[d] = p[IJKV + d] / p[SIZE_SSDOM + d]; if (lu[d] < 1) lu[d] = 1; }
  for (int k_lu = 1; k_lu <= lu[2]; k_lu++)
    for (int j_lu = 1; j_lu <= lu[1]; j_lu++)
      for (int i_lu = 1; i_lu <= lu[0]; i_lu++) {
        const int no = (i_lu - 1) + (j_lu - 1) * lu[0] + (k_lu - 1) * lu[0] * lu[1];
        int lo[3] = {1 + (i_lu - 1) * p[SIZE_SSDOM], 1 + (j_lu - 1) * p[SIZE_SSDOM + 1], 1 + (k_lu - 1) * p[SIZE_SSDOM + 2]};
        int hi[3] = {i_lu * p[SIZE_SSDOM], j_lu * p[SIZE_SSDOM + 1], k_lu * p[SIZE_SSDOM + 2]};
        if (i_lu == lu[0]) hi[0] = p[IJKV];
        if (j_lu == lu[1]) hi[1] = p[IJKV + 1];
        if (k_lu == lu[2]) hi[2] = p[IJKV + 2];
        double xs[6] = {0., 0., 0., 0., 0., 0.}, xm[6] = {0., 0., 0., 0., 0., 0.};
        for (int k = lo[2]; k <= hi[2]; k++)
          for (int j = lo[1]; j <= hi[1]; j++)
            for (int i = lo[0]; i <= hi[0]; i++) {
              long l = INDDM(i, j, k);
              for (int e = 0; e < neq; e++) {
                double v = drodm[l + e * nd];
                double xp = fabs(v);
                if (xp > xm[e]) xm[e] = xp;
                xs[e] = xs[e] + v * v;
              }
            }
        for (int e = 0; e < neq; e++) { sums[no * neq + e] = xs[e]; maxs[no * neq + e] = xm[e]; }
      }
}

/* =====================================================================================
 * Implicit time step: BDF Newton residual + LU-SGS (nb_relax = 1, lu_match = 0, 5 equations,
 * fixed mesh, global time step).  The LU range is the zone interior (no wall boundary).
 * ===================================================================================== */
#define NEQ_COE 39
#define DTLOC 46
#define NB_RELAX 71
#define LU_MATCH 75

/* dt/vol and the LU coefficients: coe(:,1) = dt/vol (Fast/FastS/FastS/Compute/tstb3_global.for:64-78),
 * coe(:,2..4) = spectral radii of the i, j, k directions (convective + viscous), coe(:,5) = diagonal
 * xid + coe1*(coe2+coe3+coe4)/2 (tstb3c.for:62-76,86-176,262-311).  Range = the whole zone,
 * ghost cells included (ind_coe of indice_boucle_ssdom.for:108-112 with one cache block).
 * Local time step (DTLOC = 1, time_step_nature='local'): coe(:,1) = CFL * min(convective, viscous) limit of
 * the cell, xid = 1 (tstb3_impli.for:57-71,86-176, dtloc.for). */
void fo_tstb3_implicit(const i32 *p, const double *pr, int first_it, const i32 *ind, const double *rop,
                       const double *xmut, double *coe, const double *ti, const double *tj, const double *tk,
                       const double *vol)
{
  GEOM(p)
  const int dtloc = p[DTLOC] == 1;
  const double cut0x = dtloc ? 1e-30 : 1e-20;
  const double gamma = pr[GAMMA], rgp = pr[CVINF] * (gamma - 1.), prandt = pr[PRANDT];
  double xid = 1.5;
  if (first_it == 0 || dtloc) xid = 1.;
  double gam1 = gamma / prandt;
  if (p[IFLOW] == 1) gam1 = dtloc ? cut0x : 0.;
  const double gam1_1 = 1. / gam1;
  const double gam2 = gamma * rgp;
  const long ndm = p[NDIMDX_MTR];
  const long minc[3] = {p[NIJK_MTR], p[NIJK_MTR + 1], p[NIJK_MTR + 2]};
  const int twod = p[ITYPZONE] == 3;   /* tstb3_impli.for:315-380: the general formulas without z / w / k terms */
  const int full = FULL3D(p[ITYPZONE]) || twod;
  const double *T[3] = {ti, tj, tk};
  double tcc[3] = {ti[0], tj[0], twod ? 0. : tk[0]}, tc2c[3];
  for (int d = 0; d < 3; d++) tc2c[d] = sqrt(tcc[d] * tcc[d]);
  const double volinv = pr[DTC] / vol[0];
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k);
        const long lt = full ? INDMTR(p, i, j, k) : 0;
        coe[l] = full ? pr[DTC] / vol[lt] : volinv;
        const double r = rop[l], detj = dmax(vol[lt], cut0x);
        const double u = rop[l + nd], v = rop[l + 2 * nd], w = rop[l + 3 * nd];
        const double c = sqrt(gam2 * rop[l + 4 * nd]);
        const double vmu = gam1 * xmut[l], volu = detj * r, xmvis = vmu / volu;
        double tc2s[3] = {0., 0., 0.};
        for (int d = 0; d < (twod ? 2 : 3); d++) {
          double qn, tc2;
          if (twod) {
            const double *t = T[d];
            const long lp = lt + minc[d];
            const double tcx = .5 * (t[lt] + t[lp]), tcy = .5 * (t[lt + ndm] + t[lp + ndm]);
            tc2 = .5 * (sqrt(t[lt] * t[lt] + t[lt + ndm] * t[lt + ndm]) + sqrt(t[lp] * t[lp] + t[lp + ndm] * t[lp + ndm]));
            qn = tcx * u + tcy * v;
          } else if (full) {
            const double *t = T[d];
            const long lp = lt + minc[d];
            const double tcx = .5 * (t[lt] + t[lp]), tcy = .5 * (t[lt + ndm] + t[lp + ndm]);
            const double tcz = .5 * (t[lt + 2 * ndm] + t[lp + 2 * ndm]);
            tc2 = .5 * (sqrt(t[lt] * t[lt] + t[lt + ndm] * t[lt + ndm] + t[lt + 2 * ndm] * t[lt + 2 * ndm]) +
                        sqrt(t[lp] * t[lp] + t[lp + ndm] * t[lp + ndm] + t[lp + 2 * ndm] * t[lp + 2 * ndm]));
            qn = tcx * u + tcy * v + tcz * w;
          } else {
            tc2 = tc2c[d];
            qn = tcc[d] * (d == 0 ? u : (d == 1 ? v : w));
          }
          const double cn = c * tc2, vm = fabs(qn) + cn;
          coe[l + (1 + d) * nd] = vm * 2. + xmvis * tc2 * tc2 * 4.;
          tc2s[d] = tc2;
        }
        if (dtloc) {   /* dtloc.for */
          const double dd = 1. / (tc2s[0] * tc2s[0] + tc2s[1] * tc2s[1] + tc2s[2] * tc2s[2]);
          const double dtvis = 0.5 * detj * dd * gam1_1 / xmut[l] * r;
          const double sp = sqrt(u * u + v * v + w * w) + c;
          const double dtconv = sqrt(dd) / sp;
          coe[l] = pr[15] * dmin(dtconv, dtvis);
        }
        if (twod) coe[l + 4 * nd] = xid + coe[l] * (coe[l + nd] + coe[l + 2 * nd]) * 0.5;
        else coe[l + 4 * nd] = xid + coe[l] * (coe[l + nd] + coe[l + 2 * nd] + coe[l + 3 * nd]) * 0.5;
      }
}

/* Newton residual of the BDF scheme (Fast/FastS/FastS/Compute/core3as2.for:37-45,232-266,300-338):
 * drodm = dt/vol * balance + c1 W(iterate) + c2 W(n) + c3 W(n-1), BDF1 at the very first time step. */
void fo_core3as2(const i32 *p, const double *pr, int nitcfg, int first_it, const i32 *ind, const double *rop,
                 const double *rop_n, const double *rop_n1, double *drodm, const double *coe)
{
  GEOM(p)
  const double cv = pr[CVINF];
  double c1, c2, c3;
  if (first_it == 0 || p[DTLOC] == 1) { c1 = -1.; c2 = 1.; c3 = 0.; }
  else { c1 = -1.5; c2 = 2.; c3 = -0.5; }
  const double c4 = c1 + c2;
  const int neq = p[NEQ];
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k);
#define Q(a, e) (a)[l + (e) * nd]
        if (neq == 6) {   /* core3as2.for:49-66,97-99,187-190: nu_tilde equation; local time step: no BDF terms */
          if (nitcfg == 1 && p[DTLOC] == 1) {
            for (int e = 0; e < 6; e++) Q(drodm, e) = coe[l] * Q(drodm, e);
            continue;
          }
          if (nitcfg == 1) Q(drodm, 5) = coe[l] * Q(drodm, 5) + c4 * Q(rop, 5) * Q(rop, 0) + c3 * Q(rop_n1, 5) * Q(rop_n1, 0);
          else Q(drodm, 5) = coe[l] * Q(drodm, 5) + c1 * Q(rop, 5) * Q(rop, 0) + c2 * Q(rop_n, 5) * Q(rop_n, 0) +
                             c3 * Q(rop_n1, 5) * Q(rop_n1, 0);
        }
        const double roe = Q(rop, 0) * (cv * Q(rop, 4) + 0.5 * (Q(rop, 1) * Q(rop, 1) + Q(rop, 2) * Q(rop, 2) + Q(rop, 3) * Q(rop, 3)));
        const double roe_n1 = Q(rop_n1, 0) * (cv * Q(rop_n1, 4) + 0.5 * (Q(rop_n1, 1) * Q(rop_n1, 1) + Q(rop_n1, 2) * Q(rop_n1, 2) + Q(rop_n1, 3) * Q(rop_n1, 3)));
        if (nitcfg == 1) {
          Q(drodm, 0) = coe[l] * Q(drodm, 0) + c4 * Q(rop, 0) + c3 * Q(rop_n1, 0);
          for (int e = 1; e <= 3; e++)
            Q(drodm, e) = coe[l] * Q(drodm, e) + c4 * Q(rop, e) * Q(rop, 0) + c3 * Q(rop_n1, e) * Q(rop_n1, 0);
          Q(drodm, 4) = coe[l] * Q(drodm, 4) + c4 * roe + c3 * roe_n1;
        } else {
          const double roe_n = Q(rop_n, 0) * (cv * Q(rop_n, 4) + 0.5 * (Q(rop_n, 1) * Q(rop_n, 1) + Q(rop_n, 2) * Q(rop_n, 2) + Q(rop_n, 3) * Q(rop_n, 3)));
          Q(drodm, 0) = coe[l] * Q(drodm, 0) + c1 * Q(rop, 0) + c2 * Q(rop_n, 0) + c3 * Q(rop_n1, 0);
          for (int e = 1; e <= 3; e++)
            Q(drodm, e) = coe[l] * Q(drodm, e) + c1 * Q(rop, e) * Q(rop, 0) + c2 * Q(rop_n, e) * Q(rop_n, 0) +
                          c3 * Q(rop_n1, e) * Q(rop_n1, 0);
          Q(drodm, 4) = coe[l] * Q(drodm, 4) + c1 * roe + c2 * roe_n + c3 * roe_n1;
        }
#undef Q
      }
}

typedef struct {
  const i32 *p;
  double gam1, gamm1, gam2, cp, signe;
  long nd, inc[3], ndm, minc[3];
  int full, twod, sgn, neq;   /* sgn = +1 lower sweep (neighbour l - inc), -1 upper sweep (neighbour l + inc) */
  const double *t[3];
  double tcc[3];
} luctx;

/* b = [A(q_m) + signe*rho_d(m) I] dW(m) for the neighbour m = l -/+ inc(d) of cell l:
 *   Cartesian zone    Fast/FastS/FastS/Compute/LU/lu_i_3dcart.for, lu_j_3dcart.for, lu_k_3dcart.for
 *   curvilinear zone  LU/lu_i_3dfull.for, lu_j_3dfull.for, lu_k_3dfull.for (metric = mean of the two faces
 *                     of the neighbour along d) */
static inline void lu_neighbour(const luctx *c, int d, long l, long lt, const double *rop, const double *dw,
                                const double *coe, double *b)
{
  const long nd = c->nd;
  const long l1 = l - c->sgn * c->inc[d];
  const double signe = c->signe, gamm1 = c->gamm1, gam2 = c->gam2, gam1 = c->gam1, cp = c->cp;
  const double u = rop[l1 + nd], v = rop[l1 + 2 * nd], w = rop[l1 + 3 * nd], t = rop[l1 + 4 * nd];
  const double q2 = .5 * (u * u + v * v + w * w);
  const double h = cp * t + q2;
  const double ph2 = gamm1 * q2;
  const double cr = coe[l1 + (1 + d) * nd] * signe;
  const double d1 = dw[l1], d2 = dw[l1 + nd], d3 = dw[l1 + 2 * nd], d4 = dw[l1 + 3 * nd], d5 = dw[l1 + 4 * nd];
  const double d6 = c->neq == 6 ? dw[l1 + 5 * nd] : 0.;   /* scalar 6th equation: LU/lu_i_3dcart_SA.for:37 */
  if (c->full) {
    const double *tt = c->t[d];
    const long ndm = c->ndm;
    long lt1, lt2;
    if (c->sgn > 0) { lt1 = lt - c->minc[d]; lt2 = lt; }
    else { lt1 = lt + c->minc[d]; lt2 = lt + 2 * c->minc[d]; }
    const double tcx = 0.5 * (tt[lt2] + tt[lt1]);
    const double tcy = 0.5 * (tt[lt2 + ndm] + tt[lt1 + ndm]);
    /* two-dimensional zone: LU/lu_i_2d.for, lu_j_2d.for = the general terms with tcz = 0, w = 0, dW4 = 0 */
    const double tcz = c->twod ? 0. : 0.5 * (tt[lt2 + 2 * ndm] + tt[lt1 + 2 * ndm]);
    const double qn = tcx * u + tcy * v + tcz * w;
    const double b11 = cr, b12 = tcx, b13 = tcy, b14 = tcz;
    const double b21 = (tcx * ph2 - u * qn), b22 = (qn - tcx * gam2 * u) + cr;
    const double b23 = (tcy * u - tcx * gamm1 * v), b24 = (tcz * u - tcx * gamm1 * w), b25 = tcx * gamm1;
    const double b31 = (tcy * ph2 - v * qn), b32 = (tcx * v - tcy * gamm1 * u);
    const double b33 = (qn - tcy * gam2 * v) + cr, b34 = (tcz * v - tcy * gamm1 * w), b35 = tcy * gamm1;
    const double b41 = (tcz * ph2 - w * qn), b42 = (tcx * w - tcz * gamm1 * u);
    const double b43 = (tcy * w - tcz * gamm1 * v), b44 = (qn - tcz * gam2 * w) + cr, b45 = tcz * gamm1;
    const double b51 = qn * (ph2 - h), b52 = (tcx * h - gamm1 * u * qn), b53 = (tcy * h - gamm1 * v * qn);
    const double b54 = (tcz * h - gamm1 * w * qn), b55 = (gam1 * qn) + cr;
    b[0] = b11 * d1 + b12 * d2 + b13 * d3 + b14 * d4;
    b[1] = b21 * d1 + b22 * d2 + b23 * d3 + b24 * d4 + b25 * d5;
    b[2] = b31 * d1 + b32 * d2 + b33 * d3 + b34 * d4 + b35 * d5;
    b[3] = b41 * d1 + b42 * d2 + b43 * d3 + b44 * d4 + b45 * d5;
    b[4] = b51 * d1 + b52 * d2 + b53 * d3 + b54 * d4 + b55 * d5;
    b[5] = (qn + cr) * d6;
    return;
  }
  const double tc = c->tcc[d];
  if (d == 0) {
    const double qn = tc * u;
    const double b11 = cr, b12 = tc, b21 = (tc * ph2 - u * qn), b22 = (qn - tc * gam2 * u) + cr;
    const double b23 = (-tc * gamm1 * v), b24 = (-tc * gamm1 * w), b25 = tc * gamm1;
    const double b31 = (-v * qn), b32 = (tc * v), b33 = (qn) + cr;
    const double b41 = (-w * qn), b42 = (tc * w), b44 = (qn) + cr;
    const double b51 = qn * (ph2 - h), b52 = (tc * h - gamm1 * u * qn), b53 = (-gamm1 * v * qn);
    const double b54 = (-gamm1 * w * qn), b55 = (gam1 * qn) + cr;
    b[0] = b11 * d1 + b12 * d2;
    b[1] = b21 * d1 + b22 * d2 + b23 * d3 + b24 * d4 + b25 * d5;
    b[2] = b31 * d1 + b32 * d2 + b33 * d3;
    b[3] = b41 * d1 + b42 * d2 + b44 * d4;
    b[4] = b51 * d1 + b52 * d2 + b53 * d3 + b54 * d4 + b55 * d5;
    b[5] = (qn + cr) * d6;
  } else if (d == 1) {
    const double qn = tc * v;
    const double b11 = cr, b13 = tc, b21 = (-u * qn), b22 = qn + cr, b23 = tc * u;
    const double b31 = (tc * ph2 - v * qn), b32 = (-tc * gamm1 * u), b33 = (qn - tc * gam2 * v) + cr;
    const double b34 = (-tc * gamm1 * w), b35 = tc * gamm1;
    const double b41 = (-w * qn), b43 = (tc * w), b44 = (qn) + cr;
    const double b51 = qn * (ph2 - h), b52 = (-gamm1 * u * qn), b53 = (tc * h - gamm1 * v * qn);
    const double b54 = (-gamm1 * w * qn), b55 = (gam1 * qn) + cr;
    b[0] = b11 * d1 + b13 * d3;
    b[1] = b21 * d1 + b22 * d2 + b23 * d3;
    b[2] = b31 * d1 + b32 * d2 + b33 * d3 + b34 * d4 + b35 * d5;
    b[3] = b41 * d1 + b43 * d3 + b44 * d4;
    b[4] = b51 * d1 + b52 * d2 + b53 * d3 + b54 * d4 + b55 * d5;
    b[5] = (qn + cr) * d6;
  } else {
    const double qn = tc * w;
    const double b11 = cr, b14 = tc, b21 = -u * qn, b22 = qn + cr, b24 = tc * u;
    const double b31 = -v * qn, b33 = qn + cr, b34 = tc * v;
    const double b41 = tc * ph2 - w * qn, b42 = -tc * gamm1 * u, b43 = -tc * gamm1 * v;
    const double b44 = (qn - tc * gam2 * w) + cr, b45 = tc * gamm1;
    const double b51 = qn * (ph2 - h), b52 = -gamm1 * u * qn, b53 = -gamm1 * v * qn;
    const double b54 = tc * h - gamm1 * w * qn, b55 = gam1 * qn + cr;
    b[0] = b11 * d1 + b14 * d4;
    b[1] = b21 * d1 + b22 * d2 + b24 * d4;
    b[2] = b31 * d1 + b33 * d3 + b34 * d4;
    b[3] = b41 * d1 + b42 * d2 + b43 * d3 + b44 * d4 + b45 * d5;
    b[4] = b51 * d1 + b52 * d2 + b53 * d3 + b54 * d4 + b55 * d5;
     b[5] = (qn + cr) * d6;
  }
}

static void luctx_init(luctx *c, const i32 *p, const double *pr, const double *ti, const double *tj,
                       const double *tk, int sgn)
{
  c->p = p;
  c->gam1 = pr[GAMMA]; c->gamm1 = c->gam1 - 1.; c->cp = pr[GAMMA] * pr[CVINF]; c->gam2 = c->gamm1 - 1.;
  c->sgn = sgn; c->signe = sgn > 0 ? 0.5 : -0.5;
  c->nd = p[NDIMDX];
  c->inc[0] = 1; c->inc[1] = p[NIJK]; c->inc[2] = (long)p[NIJK] * p[NIJK + 1];
  c->ndm = p[NDIMDX_MTR];
  c->minc[0] = p[NIJK_MTR]; c->minc[1] = p[NIJK_MTR + 1]; c->minc[2] = p[NIJK_MTR + 2];
  c->twod = p[ITYPZONE] == 3;
  c->full = FULL3D(p[ITYPZONE]) || c->twod;
  c->neq = p[NEQ];
  c->t[0] = ti; c->t[1] = tj; c->t[2] = tk;
  c->tcc[0] = ti[0]; c->tcc[1] = tj[0]; c->tcc[2] = c->twod ? 0. : tk[0];
}

/* LU-SGS on the range ind, in place in dw (the reference's drodm_in and drodm_out are the same array
 * when nb_relax = 1 and lu_match = 0, Fast/FastS/FastS/Compute/lhs.cpp:151-153), followed plane by
 * plane by the Newton update into rop_1.
 *   lower sweep  LU/invlu_l.for:90-215 (curvilinear: k then j then i), :346-476 (Cartesian: j then k then i)
 *   upper sweep  LU/invlu_u.for:93-263, :439-615 (multiply by the diagonal, k then j then i, divide)
 *   update       LU/mjr_newton.for
 * A cell on the first plane / line / point of a sweep skips the neighbours outside the range. */
void fo_invlu(const i32 *p, const double *pr, const i32 *ind, double *dw, const double *rop, double *rop_1,
              const double *ti, const double *tj, const double *tk, const double *coe)
{
  GEOM(p)
  luctx c;
  double b[6];
  const int neq = p[NEQ];
  const double *coe6 = coe + 5 * nd;   /* diagonal of the nu_tilde equation (SA/sourceSA_LU.for) */
  const double cvinv = 1. / pr[CVINF], cvinv2 = 0.5 * cvinv;
  const double *coe5 = coe + 4 * nd;
  /* ---- lower ---- */
  luctx_init(&c, p, pr, ti, tj, tk, +1);
  const int order_lo[2] = {c.full ? 2 : 1, c.full ? 1 : 2};
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++) {
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
        const double xal = coe[l] * c.signe;
        for (int s = 0; s < 2; s++) {
          const int d = order_lo[s];
          if ((d == 1 && j == ind[2]) || (d == 2 && k == ind[4])) continue;
          lu_neighbour(&c, d, l, lt, rop, dw, coe, b);
          for (int e = 0; e < neq; e++) dw[l + e * nd] = dw[l + e * nd] + b[e] * xal;
        }
      }
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
        if (i > ind[0]) {
          const double xal = coe[l] * c.signe;
          lu_neighbour(&c, 0, l, lt, rop, dw, coe, b);
          for (int e = 0; e < neq; e++) dw[l + e * nd] = dw[l + e * nd] + b[e] * xal;
        }
        const double diag = 1. / coe5[l];
        for (int e = 0; e < 5; e++) dw[l + e * nd] = dw[l + e * nd] * diag;
        if (neq == 6) dw[l + 5 * nd] = dw[l + 5 * nd] / coe6[l];   /* LU/lu_dinv_SA.for */
      }
    }
  /* ---- upper + Newton update ---- */
  luctx_init(&c, p, pr, ti, tj, tk, -1);
  for (int k = ind[5]; k >= ind[4]; k--) {
    for (int j = ind[3]; j >= ind[2]; j--) {
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
        const double xal = coe[l] * c.signe;
        const double diag = coe5[l];
        for (int e = 0; e < 5; e++) dw[l + e * nd] = dw[l + e * nd] * diag;
        if (neq == 6) dw[l + 5 * nd] = dw[l + 5 * nd] * coe6[l];   /* LU/lu_d_SA.for */
        if (k < ind[5]) {
          lu_neighbour(&c, 2, l, lt, rop, dw, coe, b);
          for (int e = 0; e < neq; e++) dw[l + e * nd] = dw[l + e * nd] + b[e] * xal;
        }
        if (j < ind[3]) {
          lu_neighbour(&c, 1, l, lt, rop, dw, coe, b);
          for (int e = 0; e < neq; e++) dw[l + e * nd] = dw[l + e * nd] + b[e] * xal;
        }
      }
      for (int i = ind[1]; i >= ind[0]; i--) {
        const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
        if (i < ind[1]) {
          const double xal = coe[l] * c.signe;
          lu_neighbour(&c, 0, l, lt, rop, dw, coe, b);
          for (int e = 0; e < neq; e++) dw[l + e * nd] = dw[l + e * nd] + b[e] * xal;
        }
        const double diag = 1. / coe5[l];
        for (int e = 0; e < 5; e++) dw[l + e * nd] = dw[l + e * nd] * diag;
        if (neq == 6) dw[l + 5 * nd] = dw[l + 5 * nd] / coe6[l];
      }
    }
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k);
        const double ro_old = rop[l], u_old = rop[l + nd], v_old = rop[l + 2 * nd], w_old = rop[l + 3 * nd];
        const double t_old = rop[l + 4 * nd];
        rop_1[l] = rop[l] + dw[l];
        const double r_1 = 1. / rop_1[l];
        rop_1[l + nd] = (ro_old * u_old + dw[l + nd]) * r_1;
        rop_1[l + 2 * nd] = (ro_old * v_old + dw[l + 2 * nd]) * r_1;
        rop_1[l + 3 * nd] = (ro_old * w_old + dw[l + 3 * nd]) * r_1;
        const double roe_old = ro_old * (pr[CVINF] * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
        rop_1[l + 4 * nd] = (roe_old + dw[l + 4 * nd]) * r_1 * cvinv -
                            cvinv2 * (rop_1[l + nd] * rop_1[l + nd] + rop_1[l + 2 * nd] * rop_1[l + 2 * nd] +
                                      rop_1[l + 3 * nd] * rop_1[l + 3 * nd]);
        if (neq == 6) {   /* LU/mjr_newton_SA.for: nu_tilde clipped to [0, ratiom * nu] */
          const double cmus1 = pr[CS], temp01 = 1. / pr[TEMP0], coesut = pr[XMUL0] * (1. + cmus1 * temp01);
          const double nu_old = rop[l + 5 * nd];
          double anulam = coesut * sqrt(rop_1[l + 4 * nd] * temp01) / (1. + cmus1 / rop_1[l + 4 * nd]);
          anulam = anulam * r_1 * pr[18];
          double t6 = (ro_old * nu_old + dw[l + 5 * nd]) * r_1;
          t6 = dmin(t6, anulam);
          rop_1[l + 5 * nd] = dmax(t6, 0.);
        }
      }
  }
}

/* Largest, smallest and summed local CFL number of a zone (Fast/FastS/FastS/Compute/cpcfl3.for:276-440 viscous
 * models, :635-790 Euler; fixed mesh), called at the first sub-iteration of a verification time step
 * (cptst3.for:114-120).  cfl[0] = max, cfl[1] = min, cfl[2] = sum over the interior cells in the order of the
 * reference loops (k, j, i); the caller divides the sum by the number of cells (computePT.cpp:627-643). */
void fo_cpcfl3(const i32 *p, const double *pr, const double *rop, const double *xmut, const double *ti,
               const double *tj, const double *tk, const double *vol, double *cfl)
{
  GEOM(p)
  const int typ = p[ITYPZONE], visc = p[IFLOW] >= 2;
  const double rgp = pr[CVINF] * (pr[GAMMA] - 1.);
  const double gam1 = pr[GAMMA] / pr[PRANDT], gam2 = pr[GAMMA] * rgp;
  const double dt = pr[DTC];
  const long ndm = p[NDIMDX_MTR];
  const long inci = p[NIJK_MTR], incj = p[NIJK_MTR + 1], inck = p[NIJK_MTR + 2];
  double surf_c = 0.;
  if (typ == 2) {
    const double ddi = 2. * sqrt(ti[0] * ti[0]), ddj = 2. * sqrt(tj[0] * tj[0]), ddk = 2. * sqrt(tk[0] * tk[0]);
    surf_c = 1. / dmax(dmax(ddi, ddj), ddk);
  }
  for (int k = 1; k <= p[IJKV + 2]; k++)
    for (int j = 1; j <= p[IJKV + 1]; j++)
      for (int i = 1; i <= p[IJKV]; i++) {
        const long l = INDDM(i, j, k);
        double surf = surf_c, volc = vol[0];
        double w = rop[l + 3 * nd];
        if (FULL3D(typ)) {
          const long lt = INDMTR(p, i, j, k);
          double ddi = sqrt(ti[lt + inci] * ti[lt + inci] + ti[lt + inci + ndm] * ti[lt + inci + ndm] +
                            ti[lt + inci + 2 * ndm] * ti[lt + inci + 2 * ndm]);
          ddi = ddi + sqrt(ti[lt] * ti[lt] + ti[lt + ndm] * ti[lt + ndm] + ti[lt + 2 * ndm] * ti[lt + 2 * ndm]);
          double ddj = sqrt(tj[lt + incj] * tj[lt + incj] + tj[lt + incj + ndm] * tj[lt + incj + ndm] +
                            tj[lt + incj + 2 * ndm] * tj[lt + incj + 2 * ndm]);
          ddj = ddj + sqrt(tj[lt] * tj[lt] + tj[lt + ndm] * tj[lt + ndm] + tj[lt + 2 * ndm] * tj[lt + 2 * ndm]);
          double ddk = sqrt(tk[lt + inck] * tk[lt + inck] + tk[lt + inck + ndm] * tk[lt + inck + ndm] +
                            tk[lt + inck + 2 * ndm] * tk[lt + inck + 2 * ndm]);
          ddk = ddk + sqrt(tk[lt] * tk[lt] + tk[lt + ndm] * tk[lt + ndm] + tk[lt + 2 * ndm] * tk[lt + 2 * ndm]);
          surf = 1. / dmax(dmax(ddi, ddj), ddk);
          volc = vol[lt];
        } else if (typ == 3) {
          const long lt = INDMTR(p, i, j, k);
          double ddi = sqrt(ti[lt + inci] * ti[lt + inci] + ti[lt + inci + ndm] * ti[lt + inci + ndm]);
          ddi = ddi + sqrt(ti[lt] * ti[lt] + ti[lt + ndm] * ti[lt + ndm]);
          double ddj = sqrt(tj[lt + incj] * tj[lt + incj] + tj[lt + incj + ndm] * tj[lt + incj + ndm]);
          ddj = ddj + sqrt(tj[lt] * tj[lt] + tj[lt + ndm] * tj[lt + ndm]);
          surf = 1. / dmax(ddi, ddj);
          volc = vol[lt];
        }
        const double u = rop[l + nd], v = rop[l + 2 * nd], r = rop[l];
        const double c = sqrt(gam2 * rop[l + 4 * nd]);
        const double ur2 = typ == 3 ? u * u + v * v : u * u + v * v + w * w;
        const double sp = sqrt(ur2);
        const double xinvspc = 1. / (sp + c);
        const double conv = 2. * volc * surf * xinvspc;
        double comp = conv;
        if (visc) {
          const double xinvkt = gam1 / xmut[l];
          const double vis = 2. * r * volc * volc * surf * surf * xinvkt;
          comp = vis < conv ? vis : conv;
        }
        const double cfloc = dt / comp;
        cfl[0] = dmax(cfl[0], cfloc);
        cfl[1] = cfloc < cfl[1] ? cfloc : cfl[1];
        cfl[2] = cfl[2] + cfloc;
      }
}

/* BCzone with lrhs = 1 on the Newton residual (Fast/FastS/FastS/Compute/lhs.cpp:53-68,
 * Fast/FastS/FastS/BC/BCzone.cpp:93-191,193-226,451-523).  On every zone face that carries a wall-type BC
 * (tags 3..7) the ghost residual is first zeroed over the whole face (extrapolation with c1 = 0), then every BC of that
 * face is applied to the residual (walls as walls: bvbs_wall_* with lrhs = 1 only switch the mesh velocity
 * off; every other type as that same zeroing), the LU range grows by one ghost layer on that face, and edges
 * and corners are averaged.  lu receives the LU range; returns -1 for BC types that are not restated. */
int fo_bczone_rhs(const i32 *p, double *drodm, const double *ti, const double *tj, const double *tk, i32 *lu)
{
  const int pt_bcs = p[PT_BC], nb = p[pt_bcs];
  const i32 sdm[6] = {1, p[IJKV], 1, p[IJKV + 1], 1, p[IJKV + 2]};
  int lskip[6] = {1, 1, 1, 1, 1, 1}, any = 0;
  i32 cl[6];
  for (int ndf = 0; ndf < nb; ndf++) {
    const int pt = p[pt_bcs + 1 + ndf], type = p[pt], idir = p[pt + 1];
    if (type == 12 || type == 18 || type == 30) return -1;
    if (type >= 3 && type <= 7) { lskip[idir - 1] = 0; any = 1; }
  }
  for (int ndf = 0; ndf < nb; ndf++) {   /* BCInflow keeps its own treatment of the residual: not restated */
    const int pt = p[pt_bcs + 1 + ndf];
    if ((p[pt] == 13 || p[pt] == 32) && !lskip[p[pt + 1] - 1]) return -1;
  }
  for (int d = 0; d < 3; d++) {
    lu[2 * d] = sdm[2 * d] - (1 - lskip[2 * d]);
    lu[2 * d + 1] = sdm[2 * d + 1] + (1 - lskip[2 * d + 1]);
  }
  if (!any) return 0;
  for (int f = 0; f < 6; f++) {
    if (lskip[f]) continue;
    i32 fen[6] = {sdm[0], sdm[1], sdm[2], sdm[3], sdm[4], sdm[5]};
    const int d = f / 2;
    if (f % 2 == 0) { fen[2 * d] = 1; fen[2 * d + 1] = 1; }
    else fen[2 * d] = fen[2 * d + 1];
    if (fo_indice_cl(f + 1, p, fen, sdm, cl)) continue;
    fo_bc_extrapolate_c(p, f + 1, cl, drodm, 0.);
  }
  for (int ndf = 0; ndf < nb; ndf++) {
    const int pt = p[pt_bcs + 1 + ndf], type = p[pt], idir = p[pt + 1];
    if (lskip[idir - 1]) continue;
    if (fo_indice_cl(idir, p, p + pt + 2, sdm, cl)) continue;
    if (type == 0 || type == 1 || type == 2 || type == 10 || type == 11 || type == 14 || type == 16 || type == 17 ||
        type == 20 || type == 21)
      fo_bc_extrapolate_c(p, idir, cl, drodm, 0.);
    else if (type == 22) continue;
    else if (type == 3 || (type == 4 && p[IFLOW] == 1) || type == 5) fo_bc_wall_inviscid(p, idir, cl, drodm, ti, tj, tk);
    else if ((type == 6 || type == 4) && p[IFLOW] > 1) fo_bc_wall_viscous(p, idir, cl, drodm);
    else return -1;
  }
  fo_correct_coins(p, drodm);
  return 0;
}

/* Newton update of one cell from its increment (LU/mjr_newton.for), 5 equations */
static inline void mjr_newton5(const double *pr, long nd, long l, const double *rop, const double *dw, double *rop_1)
{
  const double cvinv = 1. / pr[CVINF], cvinv2 = 0.5 * cvinv;
  const double ro_old = rop[l], u_old = rop[l + nd], v_old = rop[l + 2 * nd], w_old = rop[l + 3 * nd];
  const double t_old = rop[l + 4 * nd];
  rop_1[l] = rop[l] + dw[l];
  const double r_1 = 1. / rop_1[l];
  rop_1[l + nd] = (ro_old * u_old + dw[l + nd]) * r_1;
  rop_1[l + 2 * nd] = (ro_old * v_old + dw[l + 2 * nd]) * r_1;
  rop_1[l + 3 * nd] = (ro_old * w_old + dw[l + 3 * nd]) * r_1;
  const double roe_old = ro_old * (pr[CVINF] * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
  rop_1[l + 4 * nd] = (roe_old + dw[l + 4 * nd]) * r_1 * cvinv -
                      cvinv2 * (rop_1[l + nd] * rop_1[l + nd] + rop_1[l + 2 * nd] * rop_1[l + 2 * nd] +
                                rop_1[l + 3 * nd] * rop_1[l + 3 * nd]);
}

/* LU/mjr_newton_SA.for: the same, then nu_tilde clipped to [0, ratiom * nu] */
static inline void mjr_newton_n(const double *pr, int neq, long nd, long l, const double *rop, const double *dw, double *rop_1)
{
  const double ro_old = rop[l], nu_old = neq == 6 ? rop[l + 5 * nd] : 0.;
  mjr_newton5(pr, nd, l, rop, dw, rop_1);
  if (neq == 6) {
    const double cmus1 = pr[CS], temp01 = 1. / pr[TEMP0], coesut = pr[XMUL0] * (1. + cmus1 * temp01);
    const double r_1 = 1. / rop_1[l];
    double anulam = coesut * sqrt(rop_1[l + 4 * nd] * temp01) / (1. + cmus1 / rop_1[l + 4 * nd]);
    anulam = anulam * r_1 * pr[18];
    double t6 = (ro_old * nu_old + dw[l + 5 * nd]) * r_1;
    t6 = dmin(t6, anulam);
    rop_1[l + 5 * nd] = dmax(t6, 0.);
  }
}

/* LU-SSOR with nb_relax >= 2 relaxations on the range ind of a three-dimensional zone, 5 equations, one thread whose
 * window is the whole zone (so the window index of the reference's work arrays is the cell index):
 * Fast/FastS/FastS/Compute/LU/invlu.for:195-340 calling invlussor1_l.for (first lower sweep: the increment starts from
 * the right-hand side drodm_in), invlussor_l.for (later lower sweeps: it starts from what the upper sweep left) and
 * invlussor_u.for.  Per cell and sweep: neighbour terms in the order i, j, k, each added to the increment and to the
 * accumulator `ssor` (mjr_drodm_IO_ssor.for, mjr_drodm_ssor.for), then the division by the diagonal (lu_dinv.for); one
 * plane behind the sweep the increment is reset to drodm_in + ssor and ssor to zero (mjr_ssor.for), which is what the
 * next sweep starts from.  The last upper sweep updates the state plane by plane (mjr_newton.for) on the interior sdm.
 * dout, ssor: work arrays of neq * NDIMDX doubles. */
int fo_invlussor(const i32 *p, const double *pr, const i32 *ind, const i32 *sdm, const double *drodm_in, double *dout,
                 double *ssor, const double *rop, double *rop_1, const double *ti, const double *tj, const double *tk,
                 const double *coe)
{
  GEOM(p)
  const int neq = p[NEQ];
  if (neq != 5 && !(neq == 6 && p[IFLOW] == 3)) return -1;
  const int nrelax = p[NB_RELAX];
  const double *coe5 = coe + 4 * nd, *coe6 = coe + 5 * nd;   /* 6th equation: its own diagonal (lu_dinv_SA.for) */
  double b[6];
  luctx c;
  for (int m = 1; m <= nrelax; m++) {
    const int last = m == nrelax;
    /* ---- lower ---- */
    luctx_init(&c, p, pr, ti, tj, tk, +1);
    for (int k = ind[4]; k <= ind[5]; k++) {
      for (int j = ind[2]; j <= ind[3]; j++)
        for (int i = ind[0]; i <= ind[1]; i++) {
          const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
          const double xal = coe[l] * c.signe;
          const int has[3] = {i > ind[0], j > ind[2], k > ind[4]};
          int first = 1;
          for (int d = 0; d < 3; d++) {
            if (!has[d]) continue;
            lu_neighbour(&c, d, l, lt, rop, dout, coe, b);
            for (int e = 0; e < neq; e++) {
              if (m == 1 && first) { dout[l + e * nd] = drodm_in[l + e * nd] + b[e] * xal; ssor[l + e * nd] = b[e] * xal; }
              else { dout[l + e * nd] = dout[l + e * nd] + b[e] * xal; ssor[l + e * nd] = ssor[l + e * nd] + b[e] * xal; }
            }
            first = 0;
          }
          const double diag = 1. / coe5[l];
          if (m == 1 && first) {   /* first cell of the sweep: lu_dinv_IO.for */
            for (int e = 0; e < neq; e++) { dout[l + e * nd] = e < 5 ? drodm_in[l + e * nd] * diag : drodm_in[l + e * nd] / coe6[l]; ssor[l + e * nd] = 0.; }
          } else
            for (int e = 0; e < neq; e++) dout[l + e * nd] = e < 5 ? dout[l + e * nd] * diag : dout[l + e * nd] / coe6[l];
          if (k > ind[4]) {
            const long l1 = l - ninj;
            for (int e = 0; e < neq; e++) { dout[l1 + e * nd] = drodm_in[l1 + e * nd] + ssor[l1 + e * nd]; ssor[l1 + e * nd] = 0.; }
          }
        }
    }
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l1 = INDDM(i, j, ind[5]);
        for (int e = 0; e < neq; e++) { dout[l1 + e * nd] = drodm_in[l1 + e * nd] + ssor[l1 + e * nd]; ssor[l1 + e * nd] = 0.; }
      }
    /* ---- upper ---- */
    luctx_init(&c, p, pr, ti, tj, tk, -1);
    for (int k = ind[5]; k >= ind[4]; k--) {
      for (int j = ind[3]; j >= ind[2]; j--)
        for (int i = ind[1]; i >= ind[0]; i--) {
          const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
          const double xal = coe[l] * c.signe;
          const int has[3] = {i < ind[1], j < ind[3], k < ind[5]};
          for (int d = 0; d < 3; d++) {
            if (!has[d]) continue;
            lu_neighbour(&c, d, l, lt, rop, dout, coe, b);
            for (int e = 0; e < neq; e++) { dout[l + e * nd] = dout[l + e * nd] + b[e] * xal; ssor[l + e * nd] = ssor[l + e * nd] + b[e] * xal; }
          }
          const double diag = 1. / coe5[l];
          for (int e = 0; e < neq; e++) dout[l + e * nd] = e < 5 ? dout[l + e * nd] * diag : dout[l + e * nd] / coe6[l];
        }
      if (last && k >= sdm[4] && k <= sdm[5])
        for (int j = sdm[2]; j <= sdm[3]; j++)
          for (int i = sdm[0]; i <= sdm[1]; i++) mjr_newton_n(pr, neq, nd, INDDM(i, j, k), rop, dout, rop_1);
      if (!last && k < ind[5])
        for (int j = ind[3]; j >= ind[2]; j--)
          for (int i = ind[1]; i >= ind[0]; i--) {
            const long l1 = INDDM(i, j, k + 1);
            for (int e = 0; e < neq; e++) { dout[l1 + e * nd] = drodm_in[l1 + e * nd] + ssor[l1 + e * nd]; ssor[l1 + e * nd] = 0.; }
          }
    }
    if (!last)
      for (int j = ind[3]; j >= ind[2]; j--)
        for (int i = ind[1]; i >= ind[0]; i--) {
          const long l1 = INDDM(i, j, ind[4]);
          for (int e = 0; e < neq; e++) { dout[l1 + e * nd] = drodm_in[l1 + e * nd] + ssor[l1 + e * nd]; ssor[l1 + e * nd] = 0.; }
        }
  }
  return 0;
}

/* LU-SGS as the reference runs it with lu_match = 1 and one thread per zone (LU/invlu.for:98-109): invlu_overlap_l.for
 * (the neighbour terms in the order i, j, k, increments built in a work array from the kept residual: mjr_drodm_IO.for,
 * mjr_drodm.for, lu_dinv.for, lu_dinv_IO.for for the first cell) and invlu_overlap_u.for (plane by plane: multiplication
 * by the diagonal lu_d.for, sweep, division, then the Newton update mjr_newton.for on the interior sdm).  With several
 * threads the reference also lets the ranges of neighbouring thread blocks overlap (BC/BCzone.cpp:129-149): the result
 * then depends on the number of threads; this is the single-thread result. */
int fo_invlu_overlap(const i32 *p, const double *pr, const i32 *ind, const i32 *sdm, const double *drodm_in, double *dout,
                     const double *rop, double *rop_1, const double *ti, const double *tj, const double *tk,
                     const double *coe)
{
  GEOM(p)
  const int neq = p[NEQ];
  if (neq != 5 && !(neq == 6 && p[IFLOW] == 3)) return -1;
  const double *coe5 = coe + 4 * nd, *coe6 = coe + 5 * nd;
  double b[6];
  luctx c;
  luctx_init(&c, p, pr, ti, tj, tk, +1);
  for (int k = ind[4]; k <= ind[5]; k++)
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
        const double xal = coe[l] * c.signe;
        const int has[3] = {i > ind[0], j > ind[2], k > ind[4]};
        int first = 1;
        for (int d = 0; d < 3; d++) {
          if (!has[d]) continue;
          lu_neighbour(&c, d, l, lt, rop, dout, coe, b);
          for (int e = 0; e < neq; e++)
            dout[l + e * nd] = (first ? drodm_in[l + e * nd] : dout[l + e * nd]) + b[e] * xal;
          first = 0;
        }
        const double diag = 1. / coe5[l];
        for (int e = 0; e < neq; e++) {
          const double v = first ? drodm_in[l + e * nd] : dout[l + e * nd];
          dout[l + e * nd] = e < 5 ? v * diag : v / coe6[l];
        }
      }
  luctx_init(&c, p, pr, ti, tj, tk, -1);
  for (int k = ind[5]; k >= ind[4]; k--) {
    for (int j = ind[2]; j <= ind[3]; j++)
      for (int i = ind[0]; i <= ind[1]; i++) {
        const long l = INDDM(i, j, k);
        const double diag = coe5[l];
        for (int e = 0; e < neq; e++) dout[l + e * nd] = dout[l + e * nd] * (e < 5 ? diag : coe6[l]);
      }
    for (int j = ind[3]; j >= ind[2]; j--)
      for (int i = ind[1]; i >= ind[0]; i--) {
        const long l = INDDM(i, j, k), lt = c.full ? INDMTR(p, i, j, k) : 0;
        const double xal = coe[l] * c.signe;
        const int has[3] = {i < ind[1], j < ind[3], k < ind[5]};
        for (int d = 0; d < 3; d++) {
          if (!has[d]) continue;
          lu_neighbour(&c, d, l, lt, rop, dout, coe, b);
          for (int e = 0; e < neq; e++) dout[l + e * nd] = dout[l + e * nd] + b[e] * xal;
        }
        const double diag = 1. / coe5[l];
        for (int e = 0; e < neq; e++) dout[l + e * nd] = e < 5 ? dout[l + e * nd] * diag : dout[l + e * nd] / coe6[l];
      }
    if (k >= sdm[4] && k <= sdm[5])
      for (int j = sdm[2]; j <= sdm[3]; j++)
        for (int i = sdm[0]; i <= sdm[1]; i++) mjr_newton_n(pr, neq, nd, INDDM(i, j, k), rop, dout, rop_1);
  }
  return 0;
}

/* Selective frequency damping (Fast/FastS/FastS/Compute/sfd.for), once per implicit time step on the interior of the
 * level the last sub-iteration has written (lhs.cpp:181-185): the filtered state rof relaxes towards the state, then,
 * from time step SFD_INIT_IT on, the state is pulled towards the filtered state.  coe(:,1) = dt/vol. */
void fo_sfd(const i32 *p, const double *pr, int nitrun, double *rop, double *rof, const double *coe, const double *vol)
{
  GEOM(p)
  const double delta = pr[36], chi = pr[35], cd = chi * delta;
  const int cart = p[ITYPZONE] == 2, neq = p[NEQ];
  for (int ne = 0; ne < neq; ne++)
    for (int k = 1; k <= p[IJKV + 2]; k++)
      for (int j = 1; j <= p[IJKV + 1]; j++)
        for (int i = 1; i <= p[IJKV]; i++) {
          const long l = INDDM(i, j, k), lvo = cart ? 0 : INDMTR(p, i, j, k);
          const double xsi = exp(-(chi + 1. / delta) * coe[l] * vol[lvo]);
          rof[l + ne * nd] = 1. / (1. + cd) * ((1. - xsi) * rop[l + ne * nd] + (cd + xsi) * rof[l + ne * nd]);
        }
  if (nitrun < p[63]) return;
  for (int ne = 0; ne < neq; ne++)
    for (int k = 1; k <= p[IJKV + 2]; k++)
      for (int j = 1; j <= p[IJKV + 1]; j++)
        for (int i = 1; i <= p[IJKV]; i++) {
          const long l = INDDM(i, j, k), lvo = cart ? 0 : INDMTR(p, i, j, k);
          const double xsi = exp(-(chi + 1. / delta) * coe[l] * vol[lvo]);
          rop[l + ne * nd] = 1. / (xsi + cd) * ((1. + cd) * xsi * rop[l + ne * nd] + (cd * (1. - xsi)) * rof[l + ne * nd]);
        }
}

/* One Newton sub-iteration of one zone, one cache block, one thread: sequence of
 * Fast/FastS/FastS/Compute/navier_stokes_struct.for:199-275 (viscosity and coefficients at the first
 * sub-iteration, on the whole zone), :295-301 (drodm = 0), :310-345 (fluxes), :406-419 (core3as2) and
 * Fast/FastS/FastS/Compute/lhs.cpp:53-68,135-161 (BCs on the residual, invlu).  rop_ss = current Newton
 * iterate (ro at the first sub-iteration, ro_p1 afterwards); the new iterate is written to ro_p1.
 * newton_rhs (may be NULL) receives the Newton residual before the LU solve, the array whose norms the
 * reference tabulates (Fast/FastS/FastS/Compute/residus_navier.h, cprdu3s1_incl.for:47-71).
 * rhs_only != 0: the verification slot of a step (last of the NSSITER slots on a verification step,
 * Fast/FastC/FastC/Compute/souszones_list.cpp:81-83 lexit_lu): the residual is evaluated and the LU solve skipped. */
int fo_subiter_implicit_blocks(const i32 *p, const double *pr, int nitcfg, int first_it, const double *ro,
                               const double *ro_m1, double *ro_p1, double *drodm, double *coe, double *xmut,
                               double *wig, const double *ti, const double *tj, const double *tk, const double *vol,
                               double *newton_rhs, int rhs_only, int nblk, const i32 *blk);

int fo_subiter_implicit(const i32 *p, const double *pr, int nitcfg, int first_it, const double *ro,
                        const double *ro_m1, double *ro_p1, double *drodm, double *coe, double *xmut,
                        double *wig, const double *ti, const double *tj, const double *tk, const double *vol,
                        double *newton_rhs, int rhs_only)
{
  return fo_subiter_implicit_blocks(p, pr, nitcfg, first_it, ro, ro_m1, ro_p1, drodm, coe, xmut, wig, ti, tj, tk, vol,
                                    newton_rhs, rhs_only, 0, NULL);
}

/* The same with the LU-SGS solve cut in the blocks of the reference's OpenMP threads: blk holds nblk ranges of
 * interior cells (ind_dm_thread of Fast/FastS/FastS/Compute/lhs.cpp:129-131, from indice_boucle_lu.for); every thread
 * calls invlu on its own block (lhs.cpp:149-161), extended by the ghost layer on the faces of the zone where the block
 * touches a wall (BC/BCzone.cpp:99-127), and invlu skips the neighbour terms across the borders of its range: the
 * blocks are independent of each other, so solving them one after the other gives what the threads give.  The
 * right-hand side (one thread here) does not depend on the blocks.  nblk <= 1: one block, the zone. */
int fo_subiter_implicit_blocks(const i32 *p, const double *pr, int nitcfg, int first_it, const double *ro,
                               const double *ro_m1, double *ro_p1, double *drodm, double *coe, double *xmut,
                               double *wig, const double *ti, const double *tj, const double *tk, const double *vol,
                               double *newton_rhs, int rhs_only, int nblk, const i32 *blk)
{
  const int iv = p[IJKV], jv = p[IJKV + 1], kv = p[IJKV + 2];
  const int f = p[NIJK + 3], fk = p[NIJK + 4];
  const i32 sdm[6] = {1, iv, 1, jv, 1, kv};
  const i32 all[6] = {1 - f, iv + f, 1 - f, jv + f, 1 - fk, kv + fk};
  if (p[NB_RELAX] < 1) return -1;
  if (p[NEQ] != 5 && !(p[NEQ] == 6 && p[IFLOW] == 3)) return -1;
  const double *rop_ss = nitcfg == 1 ? ro : ro_p1;
  if (nitcfg == 1) {
    if (p[IFLOW] == 2) fo_invist(p, pr, all, rop_ss, xmut);
    if (p[IFLOW] == 3) fo_vispalart(p, pr, all, rop_ss, xmut);
    fo_tstb3_implicit(p, pr, first_it, all, rop_ss, xmut, coe, ti, tj, tk, vol);
  }
  fo_init_rhs(p, nitcfg, sdm, drodm);
  if (p[IFLOW] == 3) {   /* source term and, at the first sub-iteration, coe(:,6), on the interior plus one
                          * ghost layer (ind_ssa of indice_boucle_ssdom.for:116-117,156 with one cache block) */
    const int gk = fk > 0;
    const i32 ssa[6] = {0, iv + 1, 0, jv + 1, 1 - gk, kv + gk};
    fo_spsource_SA(p, pr, nitcfg, ssa, rop_ss, coe, ti, tj, tk, vol, xmut + p[NDIMDX], drodm);
  }
  if (p[SLOPE] == 6) fo_sciacovelli(p, pr, sdm, rop_ss, ti, tj, tk, vol, wig);
  if (p[KFLUDOM] == 2 && nitcfg == 1) fo_wiggle(p, sdm, rop_ss, wig);
  fo_flux(p, pr, sdm, rop_ss, drodm, wig, xmut, ti, tj, tk, vol);
  fo_bfl3(p, pr, rop_ss, drodm, wig, xmut, ti, tj, tk, vol);
  fo_core3as2(p, pr, nitcfg, first_it, sdm, rop_ss, ro, ro_m1, drodm, coe);
  if (newton_rhs) memcpy(newton_rhs, drodm, sizeof(double) * p[NEQ] * (size_t)p[NDIMDX]);
  if (rhs_only) return 0;
  i32 lu[6];
  if (fo_bczone_rhs(p, drodm, ti, tj, tk, lu)) return -1;
  if (p[NB_RELAX] > 1 || p[LU_MATCH] != 0) {
    if (nblk > 1 && p[LU_MATCH] != 0) return -1;
    const size_t n = (size_t)p[NEQ] * p[NDIMDX];
    double *w = (double *)calloc(2 * n, sizeof(double));
    if (!w) return -1;
    if (nblk > 1 && blk) {
      /* every thread relaxes its own block on work arrays of its own window (Compute/ssor.cpp:52-70, LU/prep_lussor.h;
       * window index formule_ssor_param.h) and updates the state of its own cells (ind_loop_sdm = its block): here the
       * shared arrays, whose cells outside a block its sweeps never read */
      int rc = 0;
      for (int b = 0; b < nblk && !rc; b++) {
        i32 r[6];
        for (int d = 0; d < 3; d++) {
          r[2 * d] = blk[6 * b + 2 * d] == sdm[2 * d] ? lu[2 * d] : blk[6 * b + 2 * d];
          r[2 * d + 1] = blk[6 * b + 2 * d + 1] == sdm[2 * d + 1] ? lu[2 * d + 1] : blk[6 * b + 2 * d + 1];
        }
        rc = fo_invlussor(p, pr, r, blk + 6 * b, drodm, w, w + n, rop_ss, ro_p1, ti, tj, tk, coe);
      }
      free(w);
      return rc;
    }
    const int rc = (p[LU_MATCH] != 0 && p[NB_RELAX] == 1) ? fo_invlu_overlap(p, pr, lu, sdm, drodm, w, rop_ss, ro_p1, ti, tj, tk, coe)
                                    : fo_invlussor(p, pr, lu, sdm, drodm, w, w + n, rop_ss, ro_p1, ti, tj, tk, coe);
    free(w);
    return rc;
  }
  if (nblk > 1 && blk) {
    for (int b = 0; b < nblk; b++) {
      i32 r[6];
      for (int d = 0; d < 3; d++) {
        r[2 * d] = blk[6 * b + 2 * d] == sdm[2 * d] ? lu[2 * d] : blk[6 * b + 2 * d];
        r[2 * d + 1] = blk[6 * b + 2 * d + 1] == sdm[2 * d + 1] ? lu[2 * d + 1] : blk[6 * b + 2 * d + 1];
      }
      fo_invlu(p, pr, r, drodm, rop_ss, ro_p1, ti, tj, tk, coe);
    }
    return 0;
  }
  fo_invlu(p, pr, lu, drodm, rop_ss, ro_p1, ti, tj, tk, coe);
  return 0;
}
