// Device-side numerics of the FastS face flux for Cartesian (zone type 2) and curvilinear (zone
// type 0) zones, gfx950.
//
// Every expression keeps the operand order of the reference fragment it implements and the
// translation unit is built with -ffp-contract=off, so a face flux is bit-identical to the
// one the reference computes: the oscillation sensor of the `senseur` scheme is a 0/1 switch
// driven by signs of differences, and any rounding-level deviation in the state can flip it.
//
// Reference fragments (paths under Fast/FastS/FastS/Compute/):
//   MUSCL o3 states        Slope/o3_slope_var.for:2-3
//   left/right states      etat_GD.for, etat_roe_GD.for
//   ausmpred               AUSM/3dcart/fluFaceEuler_o3_3dcart_{i,j,k}.for, Vit_ent/fluvector_3dcart_*.for
//   senseur                SENSOR/3dcart/fluFaceEuler_o3_3dcart_{i,j,k}.for
//   roe + Harten fix       ROE/3dcart/fluFaceEuler_o3_3dcart_*.for, Vit_ent/fludiffer_3dcart_*.for
//   viscous flux           fluvisq_3dcart_{i,j,k}.for, mut_interface.for
//   sensor                 SENSOR/wiggle.for, SENSOR/wiggle_{i,j,k}.for
//   curvilinear zones      AUSM|SENSOR|ROE/3dfull/fluFaceEuler_o3_3dfull_{i,j,k}.for,
//                          Vit_ent/fluvector_3dfull_*.for, Vit_ent/fludiffer_3dfull_*.for,
//                          fluvisq_3dfull_{i,j,k}.for
//   constants              template_FluxAndBalance.for:108-156
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct FsZone {
  // geometry (cells, ghost layers included)
  int ni, nj, nk, ific, ifick, iv, jv, kv;
  long long nd;            // NDIMDX: stride between two variables of one time level
  long long incj, inck;    // ni, ni*nj
  int kflu, iflow, neq, rk, nssiter, sensortype;
  // fluid / scheme constants
  double gam, rgp, gam1, gam3, c1, c2, c3, opt0, c4, c5, c6, cvisq, psiroe;
  double tmin_1;      // 100 / Tinf (minmod limiter)
  int slope;          // 1 o1, 2 o3, 3 minmod (1 and 3: roe only), 6 o3sc (kernels of their own, fasts_stage_o3sc.hip)
  double tcx, tcy, tcz, si, sj, sk, volinv;
  double coe;              // dt/vol (constant on a Cartesian zone with a global time step)
  double vol0;             // cell volume of a Cartesian zone
  double dt, cv, cvinv, cvinv2, coesut, cmus1;
  // Spalart-Allmaras (iflow 3): turbulent conductivity factor gam1/Pr_t*R, Sutherland constants in the form
  // of mulam.for (mu = sa_coesut*sqrt(T*temp01)/(1+cmus1/T)), limiter ratio mu_t/mu, model switches
  double gam4, sa_coesut, temp01, ratiom, sa_low_re, sa_rot_corr;
  // curvilinear zones (ityp 0): device arrays ti,tj,tk(ndm,3) and vol(ndm), dimensioned on the nodes;
  // metric index of cell (i,j,k) = (i+ific-1) + (j+ific-1)*mincj + (k+ifick-1)*minck
  int split;               // some of its cells feed halo messages: the stage kernel is launched in two parts (the tiles
                           // holding those cells first), so the zone is cut in more, thinner k chunks
  int ityp;
  int twod;                // two-dimensional zone (one cell along k, no k faces); handled as a curvilinear zone
                           // whose surface vectors have a zero z component and whose tk is zero
  int homo;                // k-homogeneous zone (zone type 1, an extruded mesh): handled as a curvilinear zone whose
                           // metrics have no stride along k, ti / tj a zero z component and tk only a z component;
                           // the z derivatives of the viscous flux and of the SA source take the factored form of
                           // the reference's 3dhomo fragments
  long long ndm, mincj, minck;
  const double *mt[3];
  const double *vol;
};

#define FS_DEV static __device__ __forceinline__

FS_DEV double fs_max(double a, double b) { return a > b ? a : b; }
FS_DEV double fs_min(double a, double b) { return a < b ? a : b; }
FS_DEV double fs_sign(double a, double b) { return copysign(fabs(a), b); }

// Sutherland law (invist.for:49-61)
FS_DEV double fs_sutherland(const FsZone &z, double t1) { return z.coesut * sqrt(t1) * t1 / (t1 + z.cmus1); }

// sensor of one face: Q(e, offset) returns variable e at cell l + offset*inc
template <class Q> FS_DEV double fs_wiggle(const Q &q)
{
  double test = 0.;
#pragma unroll
  for (int e = 0; e < 5; e++) {
    double a2 = q(e, -2), a1 = q(e, -1), a0 = q(e, 0), ap = q(e, 1);
    double qm = (a1 - a2) * (a0 - a1);
    double qp = (a0 - a1) * (ap - a0);
    double f = fs_sign(1., qm) + fs_sign(1., qp);
    test = e == 0 ? f : fs_min(test, f);
  }
  return test < -1e-12 ? 1. : 0.;
}

// statement function avmin of template_FluxAndBalance.for:92
FS_DEV double fs_avmin(double c, double r)
{
  const double sg = copysign(1., c);
  return sg * fs_max(0., fs_min(fabs(c), sg * r));
}

// Right (qm) and left (qp) states on the face between the cells of qm1_ and q0.  o3: Slope/o3_slope_var.for:2-3.
// roe alone also has first-order states (slope 1, Slope/o1_slope_var.for) and the minmod limiter (slope 3,
// Slope/minmod_slope_var.for: no slope where the temperature of the right cell is below 0.01 Tinf).
// csc >= 0 (slope 6, o3sc: Slope/o3sc_slope_var.for): the o3 states weighted by the shock sensor csc of the face, the rest
// of the weight going to the cell values.
template <int KFLU>
FS_DEV void fs_states(const FsZone &z, const double *qm2, const double *qm1_, const double *q0, const double *qp1_,
                      double *qm, double *qp, double csc = -1.)
{
  if (csc >= 0.) {
#pragma unroll
    for (int e = 0; e < 5; e++) {
      qm[e] = z.c4 * q0[e] + z.c5 * qm1_[e] + z.c6 * qp1_[e];
      qp[e] = z.c4 * qm1_[e] + z.c6 * qm2[e] + z.c5 * q0[e];
      qm[e] = q0[e] * (1. - csc) + qm[e] * csc;
      qp[e] = qm1_[e] * (1. - csc) + qp[e] * csc;
    }
    return;
  }
  if (KFLU == 5 && z.slope != 2) {
    if (z.slope == 1) {
#pragma unroll
      for (int e = 0; e < 5; e++) { qm[e] = q0[e]; qp[e] = qm1_[e]; }
    } else {
      const double roff = .5 * fs_max(0., fs_min(q0[4] * z.tmin_1 - .5, 1.));
#pragma unroll
      for (int e = 0; e < 5; e++) {
        const double delm = q0[e] - qm1_[e], delp = qp1_[e] - q0[e], delq = qm1_[e] - qm2[e];
        const double slp = roff * fs_avmin(delm, delp), slq = roff * fs_avmin(delq, delm);
        qm[e] = q0[e] - slp;
        qp[e] = qm1_[e] + slq;
      }
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 5; e++) {
    qm[e] = z.c4 * q0[e] + z.c5 * qm1_[e] + z.c6 * qp1_[e];
    qp[e] = z.c4 * qm1_[e] + z.c6 * qm2[e] + z.c5 * q0[e];
  }
}

// Inviscid flux through the face between cell l-inc and cell l in direction DIR.
// qm2..qp1[e] = variable e at l-2inc, l-inc, l, l+inc; wg = sensor of the face (senseur only).
// sa (may be NULL): receives u, tdu and the right / left density states, the inputs of the rho*nu_tilde flux
// (AUSM/fluFaceSA.for); ausmpred and senseur only.
template <int KFLU, int DIR>
FS_DEV void fs_flux_inviscid(const FsZone &z, const double *qm2, const double *qm1_, const double *q0,
                             const double *qp1_, double wg, double *flu, double *sa = nullptr, double csc = -1.)
{
  const double tc = DIR == 0 ? z.tcx : (DIR == 1 ? z.tcy : z.tcz);
  const double s = DIR == 0 ? z.si : (DIR == 1 ? z.sj : z.sk);
  double qm[5], qp[5];
  fs_states<KFLU>(z, qm2, qm1_, q0, qp1_, qm, qp, csc);
  constexpr int vn = 1 + DIR;
  double flu1, flu2, flu3, flu4, flu5;
  if (KFLU == 5) {
    const double gam = z.gam, rgp = z.rgp, gam1 = z.gam1, psiroe = z.psiroe;
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
    double s_1 = 1. / fs_max(s, 1.e-30);
    double nn = tc * s_1;
    double vel = DIR == 0 ? u : (DIR == 1 ? v : w);
    double qn = vel * nn;
    double f1 = qn;
    double f4 = psiroe * (fabs(f1) + cc);
    double f2 = .5 + fs_sign(.5, fabs(f1) - f4);
    double f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (r2 - r1 - dp / (cc * cc)) * s;
    double dnu = DIR == 0 ? du - nn * dqn : du;
    double dnv = DIR == 1 ? dv - nn * dqn : dv;
    double dnw = DIR == 2 ? dw - nn * dqn : dw;
    flu1 = f3 * (f1);
    flu2 = f3 * (f1 * u + r * dnu);
    flu3 = f3 * (f1 * v + r * dnv);
    flu4 = f3 * (f1 * w + r * dnw);
    flu5 = f3 * (f1 * q + r * (u * du + v * dv + w * dw - qn * dqn));
    f1 = qn + cc;
    f2 = .5 + fs_sign(.5, fabs(f1) - f4);
    f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (dp + r * cc * dqn * s_1) / (2. * cc * cc) * s * f3;
    flu1 = flu1 + f1;
    flu2 = flu2 + f1 * (DIR == 0 ? u + nn * cc : u);
    flu3 = flu3 + f1 * (DIR == 1 ? v + nn * cc : v);
    flu4 = flu4 + f1 * (DIR == 2 ? w + nn * cc : w);
    flu5 = flu5 + f1 * (h + qn * cc);
    f1 = qn - cc;
    f2 = .5 + fs_sign(.5, fabs(f1) - f4);
    f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (dp - r * cc * dqn * s_1) / (2. * cc * cc) * f3 * s;
    flu1 = flu1 + f1;
    flu2 = flu2 + f1 * (DIR == 0 ? u - nn * cc : u);
    flu3 = flu3 + f1 * (DIR == 1 ? v - nn * cc : v);
    flu4 = flu4 + f1 * (DIR == 2 ? w - nn * cc : w);
    flu5 = flu5 + f1 * (h - qn * cc);
    double rou1 = r1 * qn1, rou2 = r2 * qn2, p1p2 = p1 + p2;
    flu1 = (flu1 + rou1 + rou2) * 0.5;
    if (DIR == 0) flu2 = (flu2 + rou1 * qp[1] + rou2 * qm[1] + tc * p1p2) * 0.5;
    else flu2 = (flu2 + rou1 * qp[1] + rou2 * qm[1]) * 0.5;
    if (DIR == 1) flu3 = (flu3 + rou1 * qp[2] + rou2 * qm[2] + tc * p1p2) * 0.5;
    else flu3 = (flu3 + rou1 * qp[2] + rou2 * qm[2]) * 0.5;
    if (DIR == 2) flu4 = (flu4 + rou1 * qp[3] + rou2 * qm[3] + tc * p1p2) * 0.5;
    else flu4 = (flu4 + rou1 * qp[3] + rou2 * qm[3]) * 0.5;
    flu5 = (flu5 + rou1 * h1 + rou2 * h2) * 0.5;
  } else {
    const double rgp = z.rgp, gam1 = z.gam1;
    double r1 = qp[0], rou1 = r1 * qp[1], rov1 = r1 * qp[2], row1 = r1 * qp[3];
    double p1 = r1 * qp[4] * rgp;
    double h1 = gam1 * p1 + .5 * (rou1 * qp[1] + rov1 * qp[2] + row1 * qp[3]);
    double r2 = qm[0], rou2 = r2 * qm[1], rov2 = r2 * qm[2], row2 = r2 * qm[3];
    double p2 = r2 * qm[4] * rgp;
    double h2 = gam1 * p2 + .5 * (rou2 * qm[1] + rov2 * qm[2] + row2 * qm[3]);
    double qn1 = qp[vn] * tc, qn2 = qm[vn] * tc;
    double u, tdu;
    if (KFLU == 1) {
      double cc = rgp * z.gam * q0[4];
      double son = sqrt(qn1 * qn1 / cc);
      double tam = z.c3 * son + s;
      double tam1 = fs_max(0., tam) * z.c2;
      u = 0.25 * (qn1 + qn2) - tam1 * (p2 - p1);
      tdu = fs_max(fabs(u), z.c1 * s);
    } else {
      u = 0.25 * (qn1 + qn2) - z.c2 * s * (p2 - p1) * (z.opt0 * wg + 1. - z.opt0);
      tdu = fs_max(fabs(u), z.c1 * s) * wg;
    }
    double p1p2 = (p1 + p2) * 0.5;
    flu1 = u * (r1 + r2) - tdu * (r2 - r1);
    if (DIR == 0) flu2 = u * (rou1 + rou2) - tdu * (rou2 - rou1) + tc * p1p2;
    else flu2 = u * (rou1 + rou2) - tdu * (rou2 - rou1);
    if (DIR == 1) flu3 = u * (rov1 + rov2) - tdu * (rov2 - rov1) + tc * p1p2;
    else flu3 = u * (rov1 + rov2) - tdu * (rov2 - rov1);
    if (DIR == 2) flu4 = u * (row1 + row2) - tdu * (row2 - row1) + tc * p1p2;
    else flu4 = u * (row1 + row2) - tdu * (row2 - row1);
    flu5 = u * (h1 + h2) - tdu * (h2 - h1);
    if (sa) { sa[0] = u; sa[1] = tdu; sa[2] = qm[0]; sa[3] = qp[0]; }
  }
  flu[0] = flu1; flu[1] = flu2; flu[2] = flu3; flu[3] = flu4; flu[4] = flu5;
}

// Inviscid flux of a curvilinear zone: (tcx,tcy,tcz) = surface vector of the face between the cells
// of qm1_ and q0.  Same expressions for the three directions.
// f3_first: operand order of the j faces of a two-dimensional zone (Vit_ent/fludiffer_2d_j.for:31).
template <int KFLU>
FS_DEV void fs_flux_inviscid_full(const FsZone &z, double tcx, double tcy, double tcz, const double *qm2,
                                  const double *qm1_, const double *q0, const double *qp1_, double wg, double *flu,
                                  bool f3_first = false, double *sa = nullptr, double csc = -1.)
{
  const double s = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
  double qm[5], qp[5];
  fs_states<KFLU>(z, qm2, qm1_, q0, qp1_, qm, qp, csc);
  double flu1, flu2, flu3, flu4, flu5;
  if (KFLU == 5) {
    const double gam = z.gam, rgp = z.rgp, gam1 = z.gam1, psiroe = z.psiroe;
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
    double s_1 = 1. / fs_max(s, 1.e-30);
    double nx = tcx * s_1, ny = tcy * s_1, nz = tcz * s_1;
    double qn = u * nx + v * ny + w * nz;
    double f1 = qn;
    double f4 = psiroe * (fabs(f1) + cc);
    double f2 = .5 + fs_sign(.5, fabs(f1) - f4);
    double f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    f1 = (r2 - r1 - dp / (cc * cc)) * s;
    flu1 = f3 * (f1);
    flu2 = f3 * (f1 * u + r * (du - nx * dqn));
    flu3 = f3 * (f1 * v + r * (dv - ny * dqn));
    flu4 = f3 * (f1 * w + r * (dw - nz * dqn));
    flu5 = f3 * (f1 * q + r * (u * du + v * dv + w * dw - qn * dqn));
    f1 = qn + cc;
    f2 = .5 + fs_sign(.5, fabs(f1) - f4);
    f3 = (f2 - 1.) * (f1 * f1 / f4 + f4) * 0.5 - f2 * fabs(f1);
    if (f3_first) f1 = (dp + r * cc * dqn * s_1) / (2. * cc * cc) * f3 * s;
    else f1 = (dp + r * cc * dqn * s_1) / (2. * cc * cc) * s * f3;
    flu1 = flu1 + f1;
    flu2 = flu2 + f1 * (u + nx * cc);
    flu3 = flu3 + f1 * (v + ny * cc);
    flu4 = flu4 + f1 * (w + nz * cc);
    flu5 = flu5 + f1 * (h + qn * cc);
    f1 = qn - cc;
    f2 = .5 + fs_sign(.5, fabs(f1) - f4);
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
    const double rgp = z.rgp, gam1 = z.gam1;
    double r1 = qp[0], rou1 = r1 * qp[1], rov1 = r1 * qp[2], row1 = r1 * qp[3];
    double p1 = r1 * qp[4] * rgp;
    double h1 = gam1 * p1 + .5 * (rou1 * qp[1] + rov1 * qp[2] + row1 * qp[3]);
    double r2 = qm[0], rou2 = r2 * qm[1], rov2 = r2 * qm[2], row2 = r2 * qm[3];
    double p2 = r2 * qm[4] * rgp;
    double h2 = gam1 * p2 + .5 * (rou2 * qm[1] + rov2 * qm[2] + row2 * qm[3]);
    double qn1 = qp[1] * tcx + qp[2] * tcy + qp[3] * tcz;
    double qn2 = qm[1] * tcx + qm[2] * tcy + qm[3] * tcz;
    double u, tdu;
    if (KFLU == 1) {
      double cc = rgp * z.gam * q0[4];
      double son = sqrt(qn1 * qn1 / cc);
      double tam = z.c3 * son + s;
      double tam1 = fs_max(0., tam) * z.c2;
      u = 0.25 * (qn1 + qn2) - tam1 * (p2 - p1);
      tdu = fs_max(fabs(u), z.c1 * s);
    } else {
      u = 0.25 * (qn1 + qn2) - z.c2 * s * (p2 - p1) * (z.opt0 * wg + 1. - z.opt0);
      tdu = fs_max(fabs(u), z.c1 * s) * wg;
    }
    double p1p2 = (p1 + p2) * 0.5;
    flu1 = u * (r1 + r2) - tdu * (r2 - r1);
    flu2 = u * (rou1 + rou2) - tdu * (rou2 - rou1) + tcx * p1p2;
    flu3 = u * (rov1 + rov2) - tdu * (rov2 - rov1) + tcy * p1p2;
    flu4 = u * (row1 + row2) - tdu * (row2 - row1) + tcz * p1p2;
    flu5 = u * (h1 + h2) - tdu * (h2 - h1);
    if (sa) { sa[0] = u; sa[1] = tdu; sa[2] = qm[0]; sa[3] = qp[0]; }
  }
  flu[0] = flu1; flu[1] = flu2; flu[2] = flu3; flu[3] = flu4; flu[4] = flu5;
}

// Viscous part of a curvilinear zone, added to flu[1..4].  The three reference fragments are one
// pattern: with a = DIR+1, b = DIR+2 (cyclic) the cross sums take the neighbours along a, then b.
//   q(e, da, db, dn)  variable e at offset dn along DIR (0 = right cell ir, -1 = left cell il) and
//                     (da, db) along a and b
//   T(c, m, o)        component m of the surface vector of direction c (0 = DIR, 1 = a, 2 = b) at
//                     metric offset o: 0 = lt, 1 = lt - md, 2 = lt + md, 3 = lt + m(c), 4 = lt - md + m(c)
//   vol_r, vol_l      volumes of ir and il;  (tcx,tcy,tcz) = surface vector of the face
// SA = true (Spalart-Allmaras): conductivity with the turbulent Prandtl number (mut_prandtltb_interface.for) and
// the diffusion of nu_tilde added to flu[5] (SA/fluvisq_SA_3dfull_i.for, SA/assemble_tauij_3dfull.for,
// SA/mut_interface.for); q(5, ...) = nu_tilde, q(0, 0, 0, dn) = density, q(4, 0, 0, dn) = temperature.
// HOMO (k-homogeneous zone; DIR = direction of the face): the z derivative as fluvisq_3dhomo_{i,j,k}.for write it, the difference of
// the two sums along k times the one k-face surface, where the general fragment multiplies first.
template <bool SA = false, int DIR = 0, bool HOMO = false, class Q, class TT>
FS_DEV void fs_flux_viscous_full(const FsZone &z, const Q &q, const TT &T, double vol_r, double vol_l, double mul,
                                 double mur, double tcx, double tcy, double tcz, double *flu)
{
  double tn[3], tn1[3], ta[3], ta1[3], tb[3], tb1[3];
#pragma unroll
  for (int m = 0; m < 3; m++) {
    tn[m] = .5 * (T(0, m, 0) + T(0, m, 1));
    tn1[m] = .5 * (T(0, m, 0) + T(0, m, 2));
    ta[m] = .125 * (T(1, m, 0) + T(1, m, 1));
    ta1[m] = .125 * (T(1, m, 3) + T(1, m, 4));
    tb[m] = .125 * (T(2, m, 0) + T(2, m, 1));
    tb1[m] = .125 * (T(2, m, 3) + T(2, m, 4));
  }
  double g[SA ? 5 : 4][3];
  double sum_lr[SA ? 5 : 4];
#pragma unroll
  for (int e = 1; e <= (SA ? 5 : 4); e++) {
    const double qr = q(e, 0, 0, 0), ql = q(e, 0, 0, -1);
    double f1 = ql, f2 = qr;
    double f3 = qr + ql + q(e, -1, 0, -1) + q(e, -1, 0, 0);
    double f4 = qr + ql + q(e, 1, 0, -1) + q(e, 1, 0, 0);
    double f5 = qr + ql + q(e, 0, -1, -1) + q(e, 0, -1, 0);
    double f6 = qr + ql + q(e, 0, 1, -1) + q(e, 0, 1, 0);
#pragma unroll
    for (int m = 0; m < 3; m++)
      g[e - 1][m] = (f2 * tn1[m] - f1 * tn[m]) + (f4 * ta1[m] - f3 * ta[m]) + (f6 * tb1[m] - f5 * tb[m]);
    if (HOMO) g[e - 1][2] = DIR == 0 ? (f6 - f5) * tb[2] : (DIR == 1 ? (f4 - f3) * ta[2] : (f2 - f1) * tn[2]);
    sum_lr[e - 1] = qr + ql;
  }
  double div = (g[0][0] + g[1][1] + g[2][2]) * z.cvisq;
  double f1 = g[0][0] - div, f2 = g[0][1] + g[1][0], f3 = g[0][2] + g[2][0];
  double f4 = g[1][1] - div, f5 = g[1][2] + g[2][1], f6 = g[2][2] - div;
  double volinv = 1. / (vol_r + vol_l);
  double xmutvol = (mur + mul) * volinv;
  double xktvol = xmutvol * z.gam3;
  double xmulam = 0.;
  if (SA) {
    xmulam = z.sa_coesut * sqrt(q(4, 0, 0, -1) * z.temp01) / (1. + z.cmus1 / q(4, 0, 0, -1));
    xmulam = xmulam + z.sa_coesut * sqrt(q(4, 0, 0, 0) * z.temp01) / (1. + z.cmus1 / q(4, 0, 0, 0));
    const double xmutot = mur + mul;
    const double xmutur = xmutot - xmulam;
    xktvol = (xmulam * z.gam3 + xmutur * z.gam4) * volinv;
    xmutvol = xmutot * volinv;
  }
  double fv, fv5;
  fv = -(2. * f1 * tcx + f2 * tcy + f3 * tcz) * xmutvol;
  fv5 = fv * sum_lr[0];
  flu[1] = flu[1] + fv;
  fv = -(f2 * tcx + 2. * f4 * tcy + f5 * tcz) * xmutvol;
  fv5 = fv5 + fv * sum_lr[1];
  flu[2] = flu[2] + fv;
  fv = -(f3 * tcx + f5 * tcy + 2. * f6 * tcz) * xmutvol;
  fv5 = fv5 + fv * sum_lr[2];
  flu[3] = flu[3] + fv;
  fv = fv5 * 0.5 - (g[3][0] * tcx + g[3][1] * tcy + g[3][2] * tcz) * xktvol;
  flu[4] = flu[4] + fv;
  if (SA) {
    double xm = xmulam + q(5, 0, 0, -1) * q(0, 0, 0, -1) + q(5, 0, 0, 0) * q(0, 0, 0, 0);
    xm = xm * volinv * (1. / (2. / 3.));
    if (HOMO && DIR == 2) flu[5] = flu[5] - xm * g[4][2] * tcz;   // SA/fluvisq_SA_3dhomo_k.for: this order
    else flu[5] = flu[5] - xm * (g[4][0] * tcx + g[4][1] * tcy + g[4][2] * tcz);
  }
}

// Viscous part added to flu[1..4].  il = l - inc(DIR), ir = l.  The accessor gives
//   q(e, da, db, dn): variable e (1 u, 2 v, 3 w, 4 T) at offset dn along DIR (0 = ir, -1 = il)
//   and (da, db) along the two transverse directions in the order (i,j,k) minus DIR.
// mul/mur = viscosity at il / ir.
// xmutvol / xktvol = face viscosity and conductivity over the volume (mut_interface.for,
// mut_prandtltb_interface.for).
template <int DIR, class Q>
FS_DEV void fs_flux_viscous_x(const FsZone &z, const Q &q, double xmutvol, double xktvol, double *flu)
{
  const double tcx = z.tcx, tcy = z.tcy, tcz = z.tcz;
  double f1, f2, f3, f4, f5, f6, fv, fv5, div;
  const double Uil = q(1, 0, 0, -1), Uir = q(1, 0, 0, 0);
  const double Vil = q(2, 0, 0, -1), Vir = q(2, 0, 0, 0);
  const double Wil = q(3, 0, 0, -1), Wir = q(3, 0, 0, 0);
  const double Til = q(4, 0, 0, -1), Tir = q(4, 0, 0, 0);
  if (DIR == 0) {  // transverse (a,b) = (j,k)
    const double tix = tcx, tjy = .25 * tcy, tkz = .25 * tcz;
    f1 = Uil; f2 = Uir; fv = f1 + f2;
    f3 = fv + q(1, -1, 0, -1) + q(1, -1, 0, 0);   // l220, l020
    f4 = fv + q(1, 1, 0, -1) + q(1, 1, 0, 0);     // l210, l010
    f5 = fv + q(1, 0, -1, -1) + q(1, 0, -1, 0);   // l202, l002
    f6 = fv + q(1, 0, 1, -1) + q(1, 0, 1, 0);     // l201, l001
    double gradU_nx = (f2 - f1) * tix, gradU_ny = (f4 - f3) * tjy, gradU_nz = (f6 - f5) * tkz;
    f1 = Vil; f2 = Vir; fv = f1 + f2;
    f3 = fv + q(2, -1, 0, -1) + q(2, -1, 0, 0);
    f4 = fv + q(2, 1, 0, -1) + q(2, 1, 0, 0);
    double gradV_nx = (f2 - f1) * tix, gradV_ny = (f4 - f3) * tjy;
    f1 = Wil; f2 = Wir; fv = f1 + f2;
    f5 = fv + q(3, 0, -1, -1) + q(3, 0, -1, 0);
    f6 = fv + q(3, 0, 1, -1) + q(3, 0, 1, 0);
    double gradW_nx = (f2 - f1) * tix, gradW_nz = (f6 - f5) * tkz;
    double gradT_nx = (Tir - Til) * tix;
    div = (gradU_nx + gradV_ny + gradW_nz) * z.cvisq;
    f1 = gradU_nx - div; f2 = gradU_ny + gradV_nx; f3 = gradU_nz + gradW_nx;
    fv = -(2. * f1 * tcx) * xmutvol; fv5 = fv * (Uir + Uil); flu[1] = flu[1] + fv;
    fv = -(f2 * tcx) * xmutvol; fv5 = fv5 + fv * (Vir + Vil); flu[2] = flu[2] + fv;
    fv = -(f3 * tcx) * xmutvol; fv5 = fv5 + fv * (Wir + Wil); flu[3] = flu[3] + fv;
    fv = fv5 * 0.5 - (gradT_nx * tcx) * xktvol; flu[4] = flu[4] + fv;
  } else if (DIR == 1) {  // transverse (a,b) = (i,k)
    const double tiy = tcy, tjz = .25 * tcz, tkx = .25 * tcx;
    f1 = Uil; f2 = Uir; fv = f1 + f2;
    f5 = fv + q(1, -1, 0, -1) + q(1, -1, 0, 0);   // l220, l200
    f6 = fv + q(1, 1, 0, -1) + q(1, 1, 0, 0);     // l120, l100
    double gradU_nx = (f6 - f5) * tkx, gradU_ny = (f2 - f1) * tiy;
    f1 = Vil; f2 = Vir; fv = f1 + f2;
    f3 = fv + q(2, 0, -1, -1) + q(2, 0, -1, 0);   // l022, l002
    f4 = fv + q(2, 0, 1, -1) + q(2, 0, 1, 0);     // l021, l001
    f5 = fv + q(2, -1, 0, -1) + q(2, -1, 0, 0);
    f6 = fv + q(2, 1, 0, -1) + q(2, 1, 0, 0);
    double gradV_nx = (f6 - f5) * tkx, gradV_ny = (f2 - f1) * tiy, gradV_nz = (f4 - f3) * tjz;
    f1 = Wil; f2 = Wir; fv = f1 + f2;
    f3 = fv + q(3, 0, -1, -1) + q(3, 0, -1, 0);
    f4 = fv + q(3, 0, 1, -1) + q(3, 0, 1, 0);
    double gradW_ny = (f2 - f1) * tiy, gradW_nz = (f4 - f3) * tjz;
    double gradT_ny = (Tir - Til) * tiy;
    div = (gradU_nx + gradV_ny + gradW_nz) * z.cvisq;
    f2 = gradU_ny + gradV_nx; f4 = gradV_ny - div; f5 = gradV_nz + gradW_ny;
    fv = -(f2 * tcy) * xmutvol; fv5 = fv * (Uir + Uil); flu[1] = flu[1] + fv;
    fv = -(2. * f4 * tcy) * xmutvol; fv5 = fv5 + fv * (Vir + Vil); flu[2] = flu[2] + fv;
    fv = -(f5 * tcy) * xmutvol; fv5 = fv5 + fv * (Wir + Wil); flu[3] = flu[3] + fv;
    fv = fv5 * 0.5 - (gradT_ny * tcy) * xktvol; flu[4] = flu[4] + fv;
  } else {  // transverse (a,b) = (i,j)
    const double tiz = tcz, tky = 0.25 * tcy, tjx = 0.25 * tcx;
    f1 = Uil; f2 = Uir; fv = f1 + f2;
    f3 = fv + q(1, -1, 0, -1) + q(1, -1, 0, 0);   // l202, l200
    f4 = fv + q(1, 1, 0, -1) + q(1, 1, 0, 0);     // l102, l100
    double gradU_nx = (f4 - f3) * tjx, gradU_nz = (f2 - f1) * tiz;
    f1 = Vil; f2 = Vir; fv = f1 + f2;
    f5 = fv + q(2, 0, -1, -1) + q(2, 0, -1, 0);   // l022, l020
    f6 = fv + q(2, 0, 1, -1) + q(2, 0, 1, 0);     // l012, l010
    double gradV_ny = (f6 - f5) * tky, gradV_nz = (f2 - f1) * tiz;
    f1 = Wil; f2 = Wir; fv = f1 + f2;
    f3 = fv + q(3, -1, 0, -1) + q(3, -1, 0, 0);
    f4 = fv + q(3, 1, 0, -1) + q(3, 1, 0, 0);
    f5 = fv + q(3, 0, -1, -1) + q(3, 0, -1, 0);
    f6 = fv + q(3, 0, 1, -1) + q(3, 0, 1, 0);
    double gradW_nx = (f4 - f3) * tjx, gradW_ny = (f6 - f5) * tky, gradW_nz = (f2 - f1) * tiz;
    double gradT_nz = (Tir - Til) * tiz;
    div = (gradU_nx + gradV_ny + gradW_nz) * z.cvisq;
    f3 = gradU_nz + gradW_nx; f5 = gradV_nz + gradW_ny; f6 = gradW_nz - div;
    fv = -f3 * tcz * xmutvol; fv5 = fv * (Uir + Uil); flu[1] = flu[1] + fv;
    fv = -f5 * tcz * xmutvol; fv5 = fv5 + fv * (Vir + Vil); flu[2] = flu[2] + fv;
    fv = -2. * f6 * tcz * xmutvol; fv5 = fv5 + fv * (Wir + Wil); flu[3] = flu[3] + fv;
    fv = fv5 * 0.5 - gradT_nz * tcz * xktvol; flu[4] = flu[4] + fv;
  }
}

template <int DIR, class Q>
FS_DEV void fs_flux_viscous(const FsZone &z, const Q &q, double mul, double mur, double *flu)
{
  double xmutvol = (mur + mul) * z.volinv;
  double xktvol = xmutvol * z.gam3;
  fs_flux_viscous_x<DIR>(z, q, xmutvol, xktvol, flu);
}

// laminar viscosity in the form the SA routines use (mulam.for)
FS_DEV double fs_mulam_sa(const FsZone &z, double t) { return z.sa_coesut * sqrt(t * z.temp01) / (1. + z.cmus1 / t); }

// Runge-Kutta stage update of one cell in conservative form, back to primitive variables
// (core3ark3.for:108-154).  q = stage input, d = balance, out = stage output.
FS_DEV void fs_rk_update_c(const FsZone &z, double coefW, double coe, const double *q, const double *d, double *out)
{
  double r_1 = coefW * coe;
  double tab1 = d[0] * r_1, tab2 = d[1] * r_1, tab3 = d[2] * r_1, tab4 = d[3] * r_1, tab5 = d[4] * r_1;
  double ro_old = q[0], u_old = q[1], v_old = q[2], w_old = q[3], t_old = q[4];
  double ro_new = q[0] + tab1;
  r_1 = 1. / ro_new;
  double un = (ro_old * u_old + tab2) * r_1;
  double vn = (ro_old * v_old + tab3) * r_1;
  double wn = (ro_old * w_old + tab4) * r_1;
  double roe_old = ro_old * (z.cv * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
  out[0] = ro_new; out[1] = un; out[2] = vn; out[3] = wn;
  out[4] = (roe_old + tab5) * r_1 * z.cvinv - z.cvinv2 * (un * un + vn * vn + wn * wn);
}

FS_DEV void fs_rk_update(const FsZone &z, double coefW, const double *q, const double *d, double *out)
{
  fs_rk_update_c(z, coefW, z.coe, q, d, out);
}

// Newton residual of the implicit (BDF) scheme for one cell, Fast/FastS/FastS/Compute/core3as2.for:37-45,
// 232-266,300-338:  r = coe*d + c1 W(q) + c2 W(qn) + c3 W(qn1), W = conservative variables; at the
// first sub-iteration the iterate q is the level n itself and the two terms are merged (c4 = c1+c2).
// q = current iterate, qn / qn1 = levels n and n-1, d = flux balance, coe = dt/vol.
struct FsBdf {
  const double *ro_n, *ro_n1;   // levels n and n-1 (device)
  double c1, c2, c3;            // BDF1: -1, 1, 0   BDF2: -1.5, 2, -0.5
  int nit1;                     // 1 at the first Newton sub-iteration
  const double *coe1;           // local time step: dt of every cell (coe(:,1)); NULL = global dt/vol
  int box[6];                   // i1, i2, j1, j2, k1, k2: the cells whose residual is written (a sub-zone of the zone)
};

// Thread t of side `side` (0..5 = imin, imax, jmin, jmax, kmin, kmax) of the box: the cell whose lower face along the
// normal is the face of the box (the first cell of the box on a min side, the cell past the box on a max side).  False
// when the cell the flux lands on lies outside the interior of the zone, or t past the face.
template <class BoxT>
FS_DEV bool box_face_cell(const FsZone &z, const BoxT &box, int side, long long t, int &i, int &j, int &k)
{
  const int d = side >> 1, high = side & 1;
  const int a = d == 0 ? 1 : 0, b = d == 2 ? 1 : 2;
  const int na = box.b[2 * a + 1] - box.b[2 * a] + 1, nb = box.b[2 * b + 1] - box.b[2 * b] + 1;
  if (t >= (long long)na * nb) return false;
  int c[3];
  c[a] = box.b[2 * a] + (int)(t % na);
  c[b] = box.b[2 * b] + (int)(t / na);
  c[d] = high ? box.b[2 * d + 1] + 1 : box.b[2 * d];
  const int ends[3] = {z.iv, z.jv, z.kv};
  if (high ? c[d] > ends[d] : c[d] - 1 < 1) return false;
  i = c[0]; j = c[1]; k = c[2];
  return true;
}

FS_DEV bool fs_in_box(const FsBdf &b, int i, int j, int k)
{
  return i >= b.box[0] && i <= b.box[1] && j >= b.box[2] && j <= b.box[3] && k >= b.box[4] && k <= b.box[5];
}

FS_DEV void fs_bdf_residual(const FsZone &z, const FsBdf &b, double coe, const double *q, const double *qn,
                            const double *qn1, const double *d, double *r)
{
  const double cv = z.cv, c1 = b.c1, c2 = b.c2, c3 = b.c3;
  const double roe = q[0] * (cv * q[4] + 0.5 * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
  const double roe_n1 = qn1[0] * (cv * qn1[4] + 0.5 * (qn1[1] * qn1[1] + qn1[2] * qn1[2] + qn1[3] * qn1[3]));
  if (b.nit1) {
    const double c4 = c1 + c2;
    r[0] = coe * d[0] + c4 * q[0] + c3 * qn1[0];
#pragma unroll
    for (int e = 1; e <= 3; e++) r[e] = coe * d[e] + c4 * q[e] * q[0] + c3 * qn1[e] * qn1[0];
    r[4] = coe * d[4] + c4 * roe + c3 * roe_n1;
  } else {
    const double roe_n = qn[0] * (cv * qn[4] + 0.5 * (qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]));
    r[0] = coe * d[0] + c1 * q[0] + c2 * qn[0] + c3 * qn1[0];
#pragma unroll
    for (int e = 1; e <= 3; e++)
      r[e] = coe * d[e] + c1 * q[e] * q[0] + c2 * qn[e] * qn[0] + c3 * qn1[e] * qn1[0];
    r[4] = coe * d[4] + c1 * roe + c2 * roe_n + c3 * roe_n1;
  }
}
