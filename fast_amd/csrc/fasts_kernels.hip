// HIP kernels of the FastS time step around the stage kernels (viscosity, ghost fill, joins, residual norms,
// CFL diagnostic), gfx950 (MI355X).
// Built with -ffp-contract=off (see fasts_flux.h for why results must be bit-reproducible).
#include "fasts_kernels.h"
#include "fasts_flux.h"

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
FS_DEV long long fs_inddm(const FsZone &z, int i, int j, int k)
{
  // 0-based linear index of cell (i,j,k), interior starting at 1 (formule_param.h:4-7)
  return (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
}

// ------------------------------------------------------------------------------------------
// stage-1 auxiliary: laminar viscosity on interior + first ghost layer
// (navier_stokes_struct.for:213-225 with ind_grad)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_invist(FsZone z, int depth, const double *__restrict__ rop,
                                                double *__restrict__ xmut)
{
  // the (i,j) cells of a plane are numbered row after row, so that every workgroup but the last of a plane is full
  const int wi = z.iv + 2 * depth;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= wi * (z.jv + 2 * depth)) return;
  const int i = idx % wi + 1 - depth;  // 1-depth .. iv+depth
  const int j = idx / wi + 1 - depth;
  const int k = blockIdx.y + 1 - (depth < z.ifick ? depth : z.ifick);   // no ghost layer along k on a 2-D zone
  long long l = fs_inddm(z, i, j, k);
  xmut[l] = fs_sutherland(z, rop[4 * z.nd + l]);
}

// ------------------------------------------------------------------------------------------
// ghost fill: BC windows, then edges and corners
// ------------------------------------------------------------------------------------------
// Characteristic far-field state in ghost cell l from the interior cell l1 and the reference state
// (Fast/FastS/FastS/BC/BCFarfield.for, non_reflection.for:1-153; fixed mesh: qen = 0).
// tc = face metric scaled by snorm (only the normal component is non zero on a Cartesian zone).
// Returns true when the first characteristic enters the domain (eigv1 < 0).
// OUT: BCOutflow (BCOutflow.for): the outside state is the inside one but for the pressure of st; the ghost cell keeps
// the inside density, loses the entering part of the normal velocity and takes the temperature of the characteristic
// energy.
template <bool OUT = false>
FS_DEV bool fs_bc_farfield_first(const FsZone &z, const double *st, double tcx, double tcy, double tcz,
                                 double *rop, long long l, long long l1)
{
  const long long nd = z.nd;
  const double gam1 = z.gam - 1., gam1_1 = 1. / gam1, cvinv = 1. / z.cv;
  double roe = st[0], rue = st[1], rve = st[2], rwe = st[3], ete = st[4];
  if (OUT) {
    roe = rop[l1]; rue = rop[l1 + nd] * roe; rve = rop[l1 + 2 * nd] * roe; rwe = rop[l1 + 3 * nd] * roe;
    const double roe_inv = 1. / st[0];
    ete = st[4] - 0.5 * (st[1] * st[1] + st[2] * st[2] + st[3] * st[3]) * roe_inv +
          0.5 * (rue * rop[l1 + nd] + rve * rop[l1 + 2 * nd] + rwe * rop[l1 + 3 * nd]);
  }
  double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
  double roi = rop[l1], rui = rop[l1 + nd] * roi, rvi = rop[l1 + 2 * nd] * roi, rwi = rop[l1 + 3 * nd] * roi;
  double eti = rop[l1 + 4 * nd] * z.cv * roi + .5 * (rui * rui + rvi * rvi + rwi * rwi) / roi;
  const double qen = 0.;
  double qref1 = .5 * (roe + roi), qref2 = .5 * (rue + rui), qref3 = .5 * (rve + rvi);
  double qref4 = .5 * (rwe + rwi), qref5 = .5 * (ete + eti);
  double r = qref1, ri = 1. / r, u = qref2 * ri, v = qref3 * ri, w = qref4 * ri;
  double pp = gam1 * (qref5 - .5 * r * (u * u + v * v + w * w));
  double c = sqrt(z.gam * pp * ri), ci = 1. / c, s_1 = 1. / sn;
  double qn = tcx * u + tcy * v + tcz * w - qen;
  double tnx, tny;
  if (sn == 0.) { tnx = 1.; tny = 0.; } else { tnx = tcx / sn; tny = tcy / sn; }
  double eigv1 = qn, eigv4 = qn + sn * c, eigv5 = qn - sn * c;
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
  if (eigv1 < 0.) { rvar1 = roext; rvar2 = ruext; rvar3 = rvext; }
  if (eigv4 < 0.) rvar4 = rwext;
  if (eigv5 < 0.) rvar5 = etext;
  svar1 = rvar1 + r * ci * (rvar4 + rvar5);
  svar2 = tcz * s_1 * tnx * rvar2 - tny * rvar3 + tcx * s_1 * (rvar4 - rvar5);
  svar3 = tcz * s_1 * tny * rvar2 + tnx * rvar3 + tcy * s_1 * (rvar4 - rvar5);
  svar4 = -rvar2 + tcz * s_1 * (rvar4 - rvar5);
  svar5 = r * c * (rvar4 + rvar5);
  qvar5 = .5 * (u * u + v * v + w * w) * svar1 + r * (u * svar2 + v * svar3 + w * svar4) + gam1_1 * svar5;
  if (OUT) {
    const double rog = rop[l1], roinv = 1. / rog;
    double ug = rop[l1 + nd], vg = rop[l1 + 2 * nd], wg = rop[l1 + 3 * nd];
    if (qn < 0) { ug = ug - qn / (sn * sn) * tcx; vg = vg - qn / (sn * sn) * tcy; wg = wg - qn / (sn * sn) * tcz; }
    rop[l] = rog; rop[l + nd] = ug; rop[l + 2 * nd] = vg; rop[l + 3 * nd] = wg;
    rop[l + 4 * nd] = (qvar5 * roinv - .5 * (ug * ug + vg * vg + wg * wg)) * cvinv;
    return eigv1 < 0.;
  }
  double roinv = 1. / svar1;
  u = u + r * svar2 * roinv; v = v + r * svar3 * roinv; w = w + r * svar4 * roinv;
  rop[l] = svar1; rop[l + nd] = u; rop[l + 2 * nd] = v; rop[l + 3 * nd] = w;
  rop[l + 4 * nd] = (qvar5 * roinv - .5 * (u * u + v * v + w * w)) * cvinv;
  return eigv1 < 0.;
}

// Face metric seen by the BC routines for the column whose first interior cell is q
// (Fast/FastC/FastC/Compute/shape_tab_mtr.for:14-52).  Cartesian zone: only the component along the
// face normal is kept.  Curvilinear zone: the surface vector of the boundary face, stored with the
// first interior cell on a min face and with the cell after it on a max face.
FS_DEV void fs_bc_metric(const FsZone &z, const FsBcWin &w, int d, const int *q, double *tc)
{
  if (z.ityp == 0) {
    int p[3] = {q[0], q[1], q[2]};
    if (!(w.idir & 1)) p[d] += 1;
    const long long lm = (long long)(p[0] + z.ific - 1) + (long long)(p[1] + z.ific - 1) * z.mincj +
                         (long long)(p[2] + z.ifick - 1) * z.minck;
    const double *__restrict__ t = z.mt[d];
    tc[0] = t[lm] * 1.; tc[1] = t[lm + z.ndm] * 1.; tc[2] = t[lm + 2 * z.ndm] * 1.;
    return;
  }
  tc[0] = w.tcn * 0.; tc[1] = w.tcn * 0.; tc[2] = w.tcn * 0.;
  tc[d] = w.tcn * 1.;
}

// Periodic and extrapolation windows are pure copies (source = periodic image, bvbs_periodique.for:150-171,
// or first interior cell, BCExtrapolate.for with c1 = 1): one thread per ghost cell, i fastest.  A kernel of
// its own, so that this latency-bound copy keeps a small register footprint and a high occupancy.
__global__ void __launch_bounds__(256) k_bc_copy(FsZone z, int nwin, const FsBcWin *__restrict__ wins,
                                                 double *__restrict__ rop)
{
  const FsBcWin *w = wins + blockIdx.y;
  const int type = w->type;
  if (type != FS_BC_PERIODIC && type != FS_BC_EXTRAPOLATE && type != FS_BC_ZERO && type != FS_BC_EXTRAP_NUT) return;
  const int lo0 = w->lo[0], lo1 = w->lo[1], lo2 = w->lo[2];
  const int n0 = w->hi[0] - lo0 + 1, n1 = w->hi[1] - lo1 + 1, n2 = w->hi[2] - lo2 + 1;
  const int idir = w->idir;
  const long long shift = w->shift, nd = z.nd;
  const long long nc = (long long)n0 * n1 * n2;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nc; t += (long long)gridDim.x * blockDim.x) {
    int i = lo0 + (int)(t % n0), j = lo1 + (int)((t / n0) % n1), k = lo2 + (int)(t / ((long long)n0 * n1));
    const long long l = fs_inddm(z, i, j, k);
    long long ls = l + shift;
    if (type != FS_BC_PERIODIC) {
      switch (idir) {
        case 1: i = 1; break; case 2: i = z.iv; break;
        case 3: j = 1; break; case 4: j = z.jv; break;
        case 5: k = 1; break; default: k = z.kv; break;
      }
      ls = fs_inddm(z, i, j, k);
    }
    if (type == FS_BC_EXTRAP_NUT) {   // only nu_tilde, bounded from below (BCExtrapolateRansLesSA_i.for)
      if (z.neq == 6) rop[5 * nd + l] = fs_max(w->nue, rop[5 * nd + ls]) * 1.;
    } else if (type == FS_BC_ZERO) { for (int e = 0; e < z.neq; e++) rop[e * nd + l] = rop[e * nd + ls] * 0.; }
    else { for (int e = 0; e < z.neq; e++) rop[e * nd + l] = rop[e * nd + ls]; }
  }
}

// One block-row per window, one thread per face cell of the window; the thread walks the ghost
// ranks outwards (rank 2 of the far field reads rank 1).
//   BCautoperiod       ghost = rop(l + shift)                     bvbs_periodique.for:150-171
//   BCExtrapolate      ghost = first interior cell                 BCExtrapolate.for (c1 = 1)
//   no-slip wall       mirror cell, velocity reversed              BCWallViscous_i.for (c_ale = 0)
//   slip wall          mirror cell, velocity reflected             FastC/BC/BCWallInviscid.for
//   far field          characteristic state, then extrapolation    BCFarfield.for, BC_nextrank.for
//   pressure outlet    imposed pressure along the normal, ranks copy  BCOutpres_i.for, BC_extrap_i.for
//   isothermal wall    mirror cell, T = 2 Tw - T, rho from rho_w      BCWallViscousIsothermal_firstrank.for
__global__ void __launch_bounds__(256) k_bc_windows(FsZone z, int nwin, const FsBcWin *__restrict__ wins,
                                                    double *__restrict__ rop)
{
  const FsBcWin w = wins[blockIdx.y];
  const int d = (w.idir - 1) >> 1, sgn = (w.idir & 1) ? 1 : -1;
  const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;          // transverse directions, fastest first
  const int na = w.hi[da] - w.lo[da] + 1, nb = w.hi[db] - w.lo[db] + 1, nr = w.hi[d] - w.lo[d] + 1;
  const int first = sgn > 0 ? 1 : (d == 0 ? z.iv : (d == 1 ? z.jv : z.kv));
  const long long stride = d == 0 ? 1 : (d == 1 ? z.incj : z.inck);
  const long long inc = sgn * stride;                          // from a ghost cell towards the interior
  const long long nd = z.nd;
  const long long n = (long long)na * nb;
  if (w.type == FS_BC_PERIODIC || w.type == FS_BC_EXTRAPOLATE || w.type == FS_BC_ZERO || w.type == FS_BC_EXTRAP_NUT)
    return;   // k_bc_copy
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    int q[3];
    q[da] = w.lo[da] + (int)(t % na);
    q[db] = w.lo[db] + (int)(t / na);
    q[d] = first;
    const long long l1 = fs_inddm(z, q[0], q[1], q[2]);        // first interior cell
    if (w.type == FS_BC_PERIODIC || w.type == FS_BC_EXTRAPOLATE) {
      // handled above, one thread per ghost cell
    } else if (w.type == FS_BC_WALL_NOSLIP) {
      for (int r = 1; r <= nr; r++) {
        const long long l = l1 - r * inc, lm = l1 + (r - 1) * inc;
        rop[l] = rop[lm];
        rop[l + nd] = 0. - rop[lm + nd];
        rop[l + 2 * nd] = 0. - rop[lm + 2 * nd];
        rop[l + 3 * nd] = 0. - rop[lm + 3 * nd];
        rop[l + 4 * nd] = rop[lm + 4 * nd];
        if (z.neq == 6) rop[l + 5 * nd] = -rop[lm + 5 * nd];   // BCWallViscous_SA.for
      }
    } else if (w.type == FS_BC_WALL_SLIP) {
      double tc[3];
      fs_bc_metric(z, w, d, q, tc);
      for (int r = 1; r <= nr; r++) {
        const long long l = l1 - r * inc, lm = l1 + (r - 1) * inc;
        double tcx = tc[0], tcy = tc[1], tcz = tc[2];
        double surf = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
        surf = fs_max(surf, 1e-30);
        const double s_1 = 1. / surf;
        tcx = tcx * s_1; tcy = tcy * s_1; tcz = tcz * s_1;
        const double u = rop[lm + nd] - 0., v = rop[lm + 2 * nd] - 0., ww = rop[lm + 3 * nd] - 0.;
        const double qn = (u * tcx + v * tcy + ww * tcz);
        const double ut = u - qn * tcx, vt = v - qn * tcy, wt = ww - qn * tcz;
        const double ua = 2. * ut - u, va = 2. * vt - v, wa = 2. * wt - ww;
        rop[l] = rop[lm];
        rop[l + nd] = 0. + ua;
        rop[l + 2 * nd] = 0. + va;
        rop[l + 3 * nd] = 0. + wa;
        rop[l + 4 * nd] = rop[lm + 4 * nd];
        if (z.neq == 6) rop[l + 5 * nd] = rop[lm + 5 * nd];    // BCWallInviscidSA_i.for
      }
    } else if (w.type == FS_BC_WALL_ISOTHERMAL) {   // BCWallViscousIsothermal_firstrank.for, c_ale = 0
      const double tw = w.data[(q[da] - w.dlo[0]) + (long long)(q[db] - w.dlo[1]) * w.dn];
      for (int r = 1; r <= nr; r++) {
        const long long l = l1 - r * inc, lm = l1 + (r - 1) * inc;
        const double rho_w = rop[lm] * rop[lm + 4 * nd] / tw;
        const double u = rop[lm + nd], v = rop[lm + 2 * nd], ww = rop[lm + 3 * nd];
        const double tm = rop[lm + 4 * nd];
        rop[l] = 2. * rho_w - rop[lm];
        rop[l + nd] = 0. - u;
        rop[l + 2 * nd] = 0. - v;
        rop[l + 3 * nd] = 0. - ww;
        rop[l + 4 * nd] = 2. * tw - tm;
        if (z.neq == 6) rop[l + 5 * nd] = -rop[lm + 5 * nd];   // BCWallViscousIsothermal_firstrank_SA.for
      }
    } else if (w.type == FS_BC_OUTPRES || w.type == FS_BC_OUTMFR) {   // BCOutpres_i.for / BCOutMFR.for, linearization_state.for, BC_extrap_i.for
      const double pext = w.data[(q[da] - w.dlo[0]) + (long long)(q[db] - w.dlo[1]) * w.dn];
      const double snorm = (w.idir & 1) ? -1. : 1.;
      double tc[3];
      fs_bc_metric(z, w, d, q, tc);
      const double tcx = tc[0] * snorm, tcy = tc[1] * snorm, tcz = tc[2] * snorm;
      const double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
      const double s_1 = 1. / sn;
      double tnx = 1., tny = 0., tnz = 0.;
      if (sn != 0.) { tnx = tcx * s_1; tny = tcy * s_1; tnz = tcz * s_1; }
      const double roi = rop[l1], ui = rop[l1 + nd], vi = rop[l1 + 2 * nd], wi = rop[l1 + 3 * nd], Ti = rop[l1 + 4 * nd];
      const double pi = roi * z.rgp * Ti;
      const double roc0 = sqrt(roi * z.gam * pi);
      double g[6];
      if (w.type == FS_BC_OUTMFR) {   // pext = mass flow rate per unit area: density from the C0 and C+ characteristics
        double c0_2 = roc0 / roi;
        c0_2 = c0_2 * c0_2;
        const double coefa = c0_2;
        const double coefb = -roc0 * (ui * tnx + vi * tny + wi * tnz - 0.) - c0_2 * roi;
        const double coefc = roc0 * pext;
        const double delta = coefb * coefb - 4 * coefa * coefc;
        g[0] = (-coefb + sqrt(delta)) / (2 * coefa);
        const double dp = c0_2 * (g[0] - roi);
        const double dqn = dp / roc0;
        g[1] = ui - dqn * tnx; g[2] = vi - dqn * tny; g[3] = wi - dqn * tnz;
        g[4] = (dp + pi) / (g[0] * z.rgp);
      } else {
        const double dqn = (pext - pi) / roc0;
        const double rog = roi + roi * roi * dqn / roc0;
        g[0] = rog; g[1] = ui - dqn * tnx; g[2] = vi - dqn * tny; g[3] = wi - dqn * tnz;
        g[4] = pext / (rog * z.rgp);
      }
      g[5] = z.neq == 6 ? rop[l1 + 5 * nd] : 0.;
      for (int r = 1; r <= nr; r++)
        for (int e = 0; e < z.neq; e++) rop[l1 - r * inc + e * nd] = g[e];
    } else if (w.type == FS_BC_OUTFLOW) {   // bvbs_outflow.for: BCOutflow_i.for, then BC_extrap_i.for (5 equations)
      const double snorm = (w.idir & 1) ? -1. : 1.;
      double tc[3];
      fs_bc_metric(z, w, d, q, tc);
      fs_bc_farfield_first<true>(z, w.st, tc[0] * snorm, tc[1] * snorm, tc[2] * snorm, rop, l1 - inc, l1);
      for (int r = 2; r <= nr; r++)
        for (int e = 0; e < 5; e++) rop[l1 - r * inc + e * nd] = rop[l1 - inc + e * nd];
    } else {  // FS_BC_FARFIELD, FS_BC_INFLOW (same characteristic state, BCInflow_i.for), FS_BC_SUPERSONIC
      const double snorm = (w.idir & 1) ? -1. : 1.;
      if (w.type == FS_BC_INJ1) {
        // stagnation pressure pa, stagnation enthalpy ha and direction d0 imposed, outgoing characteristic from
        // inside, Newton on the normal velocity (BCInj1_i.for, BCInflow_newton_{1,2,3}_firstrank.for; qen = 0)
        const long long l = l1 - inc;
        const double *dt = w.data + (q[da] - w.dlo[0]) + (long long)(q[db] - w.dlo[1]) * w.dn;
        const long long ds = w.dsize;
        const double d0x = dt[0], d0y = dt[ds], d0z = dt[2 * ds], pa = dt[3 * ds], ha = dt[4 * ds];
        const double gam2 = z.gam / (z.gam - 1.);
        double tc[3];
        fs_bc_metric(z, w, d, q, tc);
        const double tcx = tc[0] * snorm, tcy = tc[1] * snorm, tcz = tc[2] * snorm;
        const double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz), s_1 = 1. / sn;
        double tnx = 1., tny = 0., tnz = 0.;
        if (sn != 0.) { tnx = tcx * s_1; tny = tcy * s_1; tnz = tcz * s_1; }
        const double roi = rop[l1], ui = rop[l1 + nd], vi = rop[l1 + 2 * nd], wi = rop[l1 + 3 * nd], Ti = rop[l1 + 4 * nd];
        const double pi = roi * z.rgp * Ti;
        const double qen = 0.;
        const double vni = ui * tnx + vi * tny + wi * tnz;
        const double roc0 = sqrt(roi * z.gam * pi);
        const double usd0n = 1. / (d0x * tnx + d0y * tny + d0z * tnz), usd0n2 = usd0n * usd0n;
        const double wni = vni - qen, ri = pi + roc0 * wni;
        double wng = ui * tnx + vi * tny + wi * tnz - qen;
        double residug = 1.e+20;
        int nitnwt = 0;
        while (residug > w.tol && nitnwt < w.nit) {
          nitnwt++;
          residug = 0.;
          double b = 1. - ((wng + qen) * (wng + qen)) * usd0n2 / (2. * ha);
          b = fs_max(0.2, b);
          const double pg = pa * pow(b, gam2);
          const double rog = gam2 * pg / (ha * b);
          const double fn = pg + roc0 * wng - ri;
          const double df = roc0 - rog * (wng + qen) * usd0n2;
          const double dwng = -fn / df;
          wng = wng + dwng;
          residug = fs_max(residug, fabs(dwng / wng));
        }
        const double vng = wng + qen, vg = vng * usd0n;
        double b = 1. - (vng * vng) * usd0n2 / (2. * ha);
        b = fs_max(0.2, b);
        const double pg = pa * pow(b, gam2), rog = gam2 * pg / (ha * b), Tg = pg / (rog * z.rgp);
        rop[l] = rog; rop[l + nd] = vg * d0x; rop[l + 2 * nd] = vg * d0y; rop[l + 3 * nd] = vg * d0z;
        rop[l + 4 * nd] = Tg;
        if (z.neq == 6) rop[l + 5 * nd] = dt[5 * ds];
      } else if (w.type == FS_BC_INJMFR) {
        // direction d0, mass flow rate per unit area qp and stagnation enthalpy ha imposed (BCInjMFR.for; qen = 0): a
        // Newton iteration on the pressure starts a state at rest, otherwise the normal velocity solves a quadratic
        const long long l = l1 - inc;
        const double *dt = w.data + (q[da] - w.dlo[0]) + (long long)(q[db] - w.dlo[1]) * w.dn;
        const long long ds = w.dsize;
        const double d0x = dt[0], d0y = dt[ds], d0z = dt[2 * ds], qp = dt[3 * ds], ha = dt[4 * ds];
        const double gam2 = z.gam / (z.gam - 1.);
        double tc[3];
        fs_bc_metric(z, w, d, q, tc);
        const double tcx = tc[0] * snorm, tcy = tc[1] * snorm, tcz = tc[2] * snorm;
        const double sn = sqrt(tcx * tcx + tcy * tcy + tcz * tcz), s_1 = 1. / sn;
        double tnx = 1., tny = 0., tnz = 0.;
        if (sn != 0.) { tnx = tcx * s_1; tny = tcy * s_1; tnz = tcz * s_1; }
        const double ro0 = rop[l1], u0 = rop[l1 + nd], v0 = rop[l1 + 2 * nd], w0 = rop[l1 + 3 * nd], T0 = rop[l1 + 4 * nd];
        const double p0 = ro0 * z.rgp * T0, pi = p0;
        const double qen = 0.;
        const double roc0 = sqrt(ro0 * z.gam * p0);
        const double usd0n = 1 / (d0x * tnx + d0y * tny + d0z * tnz), usd0n2 = usd0n * usd0n;
        double wn = u0 * tnx + v0 * tny + w0 * tnz - qen;
        if (fabs(wn) < 1.e-6) {
          double residug = 1.e+20;
          int nitnwt = 0;
          double mach = sqrt((u0 * u0) + (v0 * v0) + (w0 * w0));
          mach = mach * ro0 / roc0;
          const double pt = p0 * pow(1 + 0.2 * (mach * mach), gam2);
          const double ri = pi + roc0 * wn;
          while (residug > w.tol && nitnwt < w.nit) {
            nitnwt++;
            residug = 0.;
            const double b = fs_max(0.2, 1. - ((wn + qen) * (wn + qen)) * usd0n2 / (2. * ha));
            const double pg = pt * pow(b, gam2);
            const double rog = gam2 * pg / (ha * b);
            const double fn = pg + roc0 * wn - ri;
            const double df = roc0 - rog * (wn + qen) * usd0n2;
            const double dwng = -fn / df;
            wn = wn + dwng;
            residug = fs_max(residug, fabs(dwng / wn));
          }
          const double qn = wn + qen;
          const double b = fs_max(0.2, 1. - (qn * qn) * usd0n2 / (2. * ha));
          const double pg = pt * pow(b, gam2), rog = gam2 * pg / (ha * b), Tg = pg / (rog * z.rgp);
          const double vg = qn * usd0n;
          rop[l] = rog; rop[l + nd] = vg * d0x; rop[l + 2 * nd] = vg * d0y; rop[l + 3 * nd] = vg * d0z;
          rop[l + 4 * nd] = Tg;
        } else {
          const double coefa = 0.5 * qp * usd0n2 / gam2 + roc0;
          const double coefb = p0 + roc0 * (u0 * tnx + v0 * tny + w0 * tnz - qen);
          const double coefc = -ha * qp / gam2;
          const double delta = coefb * coefb - 4. * coefa * coefc;
          const double qn = (-coefb + sqrt(delta)) / (2. * coefa) + qen;
          const double vg = qn * fabs(usd0n);
          rop[l] = qp / qn; rop[l + nd] = vg * d0x; rop[l + 2 * nd] = vg * d0y; rop[l + 3 * nd] = vg * d0z;
          rop[l + 4 * nd] = (ha - 0.5 * vg * vg) / (z.cv * z.gam);
        }
        if (z.neq == 6) rop[l + 5 * nd] = dt[5 * ds];
      } else if (w.type == FS_BC_SUPERSONIC && w.data) {   // first rank = the window's own conservative state
        const long long l = l1 - inc;                 // (BCInflowSupersonic_fich_firstrank.for)
        const double *dt = w.data + (q[da] - w.dlo[0]) + (long long)(q[db] - w.dlo[1]) * w.dn;
        const double ro = dt[0], ru = dt[w.dsize], rv = dt[2 * (long long)w.dsize], rw = dt[3 * (long long)w.dsize];
        const double re = dt[4 * (long long)w.dsize];
        rop[l] = ro; rop[l + nd] = ru / ro; rop[l + 2 * nd] = rv / ro; rop[l + 3 * nd] = rw / ro;
        rop[l + 4 * nd] = (re - 0.5 * (ru * ru + rv * rv + rw * rw) / ro) / (ro * z.cv);
        if (z.neq == 6) rop[l + 5 * nd] = dt[5 * (long long)w.dsize] / ro;
      } else if (w.type == FS_BC_SUPERSONIC) {   // first rank = reference state (BCInflowSupersonic_i.for)
        const long long l = l1 - inc;
        const double ro_state = w.st[0], u_state = w.st[1] / w.st[0], v_state = w.st[2] / w.st[0], w_state = w.st[3] / w.st[0];
        const double t_state = (w.st[4] - 0.5 * ro_state * (u_state * u_state + v_state * v_state + w_state * w_state)) /
                               (ro_state * z.cv);
        rop[l] = ro_state; rop[l + nd] = u_state; rop[l + 2 * nd] = v_state; rop[l + 3 * nd] = w_state;
        rop[l + 4 * nd] = t_state;
        if (z.neq == 6) rop[l + 5 * nd] = w.nue;
      } else {
        double tc[3];
        fs_bc_metric(z, w, d, q, tc);
        const bool inflow = fs_bc_farfield_first(z, w.st, tc[0] * snorm, tc[1] * snorm, tc[2] * snorm, rop, l1 - inc, l1);
        // BCFarfield_firstrank_SA.for; BCInflowSA_i.for always takes the state value
        if (z.neq == 6) rop[l1 - inc + 5 * nd] = (inflow || w.type == FS_BC_INFLOW) ? w.nue : rop[l1 + 5 * nd];
      }
      const double c4 = 5. / 6., c5 = 2. / 6., c6 = -1. / 6.;
      const double c0 = 1. / c6, c1 = -(c4 + c5) * c0, c2 = -c6 * c0, c3 = (2. - c5 - c4) * c0;
      for (int r = 2; r <= nr; r++) {   // BC_nextrank.for
        const long long l = l1 - r * inc, l0 = l + inc, l1b = l + 2 * inc, l2 = l + 3 * inc;
        const double ro = rop[l0] * c3 + rop[l1b] * c1 + rop[l2] * c2;
        rop[l] = ro;
        // positivity guard: BC_nextrank.for (5 equations), BC_nextrankSA.for (6)
        const bool guard = z.neq == 6 ? ro <= 0.3 * rop[l0] : fabs(ro / rop[l0] - 1.) >= 0.7;
        if (guard) {
          for (int e = 0; e < z.neq; e++) rop[l + e * nd] = rop[l0 + e * nd];
        } else {
          for (int e = 1; e < z.neq; e++) rop[l + e * nd] = rop[l0 + e * nd] * c3 + rop[l1b + e * nd] * c1 + rop[l2 + e * nd] * c2;
        }
      }
    }
  }
}

// 12 edges (mean of two face ghosts) and 8 corners (mean of three), correct_coins.for:23-420.
// blockIdx.y = edge/corner id 0..19.  box = the range of interior cells the fix is made for (correct_coins.for:16-21):
// an edge is treated where the box reaches both of its faces, along the cells of the box; a corner where it reaches
// the three.
__global__ void __launch_bounds__(256) k_bc_edges(FsZone z, FsBox box, double *__restrict__ rop)
{
  const int id = blockIdx.y;
  const int f = z.ific, fk = z.ifick;
  const int ends[3][2] = {{1, z.iv}, {1, z.jv}, {1, z.kv}};
  const int nfic[3] = {f, f, fk};
  const long long nd = z.nd;
  if (id < 12) {
    // edge family: 0-3 (i,j) ghosts, k interior; 4-7 (i,k), j interior; 8-11 (j,k), i interior
    const int fam = id / 4, s0 = (id % 4) & 1, s1 = ((id % 4) >> 1) & 1;
    const int da = fam == 2 ? 1 : 0;            // first ghost direction
    const int db = fam == 0 ? 1 : 2;            // second ghost direction
    const int dc = fam == 0 ? 2 : (fam == 1 ? 1 : 0);  // interior direction
    if (box.b[2 * da + s0] != ends[da][s0] || box.b[2 * db + s1] != ends[db][s1]) return;
    const int c_lo = box.b[2 * dc];
    const int na = nfic[da], nb = nfic[db], nc = box.b[2 * dc + 1] - c_lo + 1;
    const long long n = (long long)na * nb * nc;
    const int sa = s0 ? 1 : -1, sb = s1 ? 1 : -1;
    const int a0 = ends[da][s0], b0 = ends[db][s1];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
      int c = c_lo + (int)(t % nc);
      int a = 1 + (int)((t / nc) % na);
      int b = 1 + (int)(t / ((long long)nc * na));
      int a1 = a + 1 < na ? a + 1 : na, b1 = b + 1 < nb ? b + 1 : nb;
      int p1[3], p11[3], p12[3];
      p1[da] = a0 + a * sa;   p1[db] = b0 + b * sb;   p1[dc] = c;
      // the reference pairs (ghost a1, face b0) with (face a0, ghost b1) for (i,j) edges and
      // (face a0, ghost b1) with (ghost a1, face b0) for the others: the sum is the same two cells
      p11[da] = a0 + a1 * sa; p11[db] = b0;           p11[dc] = c;
      p12[da] = a0;           p12[db] = b0 + b1 * sb; p12[dc] = c;
      long long l1 = fs_inddm(z, p1[0], p1[1], p1[2]);
      long long l11 = fs_inddm(z, p11[0], p11[1], p11[2]);
      long long l12 = fs_inddm(z, p12[0], p12[1], p12[2]);
      for (int e = 0; e < z.neq; e++) {
        double x11 = rop[e * nd + l11], x12 = rop[e * nd + l12];
        // operand order of correct_coins.for: (i,j) edges add l11 (ghost i) first; (i,k) and
        // (j,k) edges add the k-ghost cell first
        rop[e * nd + l1] = fam == 0 ? (x11 + x12) * 0.5 : (x12 + x11) * 0.5;
      }
    }
  } else {
    const int c = id - 12;
    const int si = (c & 1) ? 1 : -1, sj = (c & 2) ? 1 : -1, sk = (c & 4) ? 1 : -1;
    const int i0 = ends[0][c & 1], j0 = ends[1][(c >> 1) & 1], k0 = ends[2][(c >> 2) & 1];
    if (box.b[c & 1] != i0 || box.b[2 + ((c >> 1) & 1)] != j0 || box.b[4 + ((c >> 2) & 1)] != k0) return;
    const long long n = (long long)f * f * fk;
    const double c3 = 1. / 3.;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
      int i = 1 + (int)(t % f), j = 1 + (int)((t / f) % f), k = 1 + (int)(t / (f * f));
      int i1 = i + 1 < f ? i + 1 : f, j1 = j + 1 < f ? j + 1 : f, k1 = k + 1 < fk ? k + 1 : fk;
      long long l = fs_inddm(z, i0 + i * si, j0 + j * sj, k0 + k * sk);
      long long l11 = fs_inddm(z, i0, j0, k0 + k1 * sk);
      long long l12 = fs_inddm(z, i0 + i1 * si, j0, k0);
      long long l13 = fs_inddm(z, i0, j0 + j1 * sj, k0);
      for (int e = 0; e < z.neq; e++)
        rop[e * nd + l] = (rop[e * nd + l11] + rop[e * nd + l12] + rop[e * nd + l13]) * c3;
    }
  }
}


// ------------------------------------------------------------------------------------------
// Directional shock sensor of the o3sc slope (SENSOR/sciacovelli.for, shock_sensor.for; velocity divergence and squared
// vorticity as SA/div_rot_{3dcart,3dfull,3dhomo,2d}.for, face sizes as SA/metric_*.for): wsc(l, dir) = weight of the o3
// states on the face between l - inc(dir) and l.  Interior cells: the three directions; the plane of ghost cells past
// each max side: the direction of that side, with the metrics of the LAST interior cell (the reference's plane loops
// set the cell index only, sciacovelli.for:96-127).  MODE 0 Cartesian, 1 curvilinear, 2 two-dimensional, 3 k-homogeneous.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) k_sciacovelli(FsZone z, const double *__restrict__ rop, double *__restrict__ wsc)
{
  constexpr bool TWOD = MODE == 2, HOMO = MODE == 3;
  const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y, k = 1 + blockIdx.z;
  if (i > z.iv + 1) return;
  const bool pi = i == z.iv + 1, pj = j == z.jv + 1, pk = !TWOD && k == z.kv + 1;
  if ((int)pi + (int)pj + (int)pk > 1) return;
  const int dirs = pi ? 1 : (pj ? 2 : (pk ? 4 : (TWOD ? 3 : 7)));
  const long long nd = z.nd;
  const long long inc[3] = {1, z.incj, TWOD ? 0 : z.inck};
  const long long l = fs_inddm(z, i, j, k);
  const long long l1 = l + 1, l2 = l - 1, l3 = l + inc[1], l4 = l - inc[1], l5 = l + inc[2], l6 = l - inc[2];
  const double *__restrict__ U = rop + nd, *__restrict__ V = rop + 2 * nd, *__restrict__ W = rop + 3 * nd;
  double div, rot, tc[3];
  if (MODE == 0) {
    const double tix = z.tcx, tjy = z.tcy, tkz = z.tcz;
    tc[0] = fabs(tix); tc[1] = fabs(tjy); tc[2] = fabs(tkz);
    const double xvol = 0.5 / z.vol0;
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
    const bool plane = pi || pj || pk;
    const int mi = plane ? z.iv : i, mj = plane ? z.jv : j, mk = plane ? z.kv : k;
    const long long ndm = z.ndm;
    const long long lt = (long long)(mi + z.ific - 1) + (long long)(mj + z.ific - 1) * z.mincj + (long long)(mk + z.ifick - 1) * z.minck;
    const double *__restrict__ ti = z.mt[0], *__restrict__ tj = z.mt[1], *__restrict__ tk = z.mt[2];
    const double tix = ti[lt], tiy = ti[lt + ndm], tiz = ti[lt + 2 * ndm];
    const double tix1 = ti[lt + 1], tiy1 = ti[lt + 1 + ndm], tiz1 = ti[lt + 1 + 2 * ndm];
    const double tjx = tj[lt], tjy = tj[lt + ndm], tjz = tj[lt + 2 * ndm];
    const double tjx1 = tj[lt + z.mincj], tjy1 = tj[lt + z.mincj + ndm], tjz1 = tj[lt + z.mincj + 2 * ndm];
    const double tkx = tk[lt], tky = tk[lt + ndm], tkz = tk[lt + 2 * ndm];
    const double tkx1 = tk[lt + z.minck], tky1 = tk[lt + z.minck + ndm], tkz1 = tk[lt + z.minck + 2 * ndm];
    double t;
    t = sqrt(tix * tix + tiy * tiy + tiz * tiz); t = t + sqrt(tix1 * tix1 + tiy1 * tiy1 + tiz1 * tiz1); tc[0] = fs_max(.5 * t, 1.e-27);
    t = sqrt(tjx * tjx + tjy * tjy + tjz * tjz); t = t + sqrt(tjx1 * tjx1 + tjy1 * tjy1 + tjz1 * tjz1); tc[1] = fs_max(.5 * t, 1.e-27);
    if (HOMO) tc[2] = fs_max(fabs(tkz), 1.e-27);
    else {
      t = sqrt(tkx * tkx + tky * tky + tkz * tkz); t = t + sqrt(tkx1 * tkx1 + tky1 * tky1 + tkz1 * tkz1); tc[2] = fs_max(.5 * t, 1.e-27);
    }
    const double xvol = 0.5 / z.vol[lt];
    double u1, u2, u3, u4, u5 = 0., u6 = 0., dudx, rotx = 0., roty = 0., rotz;
    // the sums of the two-dimensional and k-homogeneous fragments are the general ones without the terms a zero
    // metric multiplies (the same bits), except along z on a k-homogeneous zone, where the difference is taken first
    auto gx = [&]() { return (TWOD || HOMO) ? (u3 * tix1 - u1 * tix + u2 * tjx1 - u4 * tjx) : (u3 * tix1 - u1 * tix + u2 * tjx1 - u4 * tjx + u5 * tkx1 - u6 * tkx); };
    auto gy = [&]() { return (TWOD || HOMO) ? (u3 * tiy1 - u1 * tiy + u2 * tjy1 - u4 * tjy) : (u3 * tiy1 - u1 * tiy + u2 * tjy1 - u4 * tjy + u5 * tky1 - u6 * tky); };
    auto gz = [&]() { return HOMO ? (u5 - u6) * tkz : (u3 * tiz1 - u1 * tiz + u2 * tjz1 - u4 * tjz + u5 * tkz1 - u6 * tkz); };
    u1 = U[l] + U[l2]; u2 = U[l] + U[l3]; u3 = U[l] + U[l1]; u4 = U[l] + U[l4];
    if (!TWOD) { u5 = U[l] + U[l5]; u6 = U[l] + U[l6]; }
    dudx = gx(); div = dudx * xvol;
    rotz = -gy();
    if (!TWOD) roty = gz();
    u1 = V[l] + V[l2]; u2 = V[l] + V[l3]; u3 = V[l] + V[l1]; u4 = V[l] + V[l4];
    if (!TWOD) { u5 = V[l] + V[l5]; u6 = V[l] + V[l6]; }
    dudx = gx(); rotz = (rotz + dudx) * xvol;
    dudx = gy(); div = div + dudx * xvol;
    if (!TWOD) {
      dudx = gz(); rotx = -dudx;
      u1 = W[l] + W[l2]; u2 = W[l] + W[l3]; u3 = W[l] + W[l1]; u4 = W[l] + W[l4]; u5 = W[l] + W[l5]; u6 = W[l] + W[l6];
      dudx = gx(); roty = (roty - dudx) * xvol;
      dudx = gy(); rotx = (rotx + dudx) * xvol;
      dudx = gz(); div = div + dudx * xvol;
      rot = rotx * rotx + roty * roty + rotz * rotz;
    } else
      rot = rotz * rotz;
  }
  for (int d = 0; d < 3; d++) {
    if (!(dirs & (1 << d))) continue;
    double mach = sqrt(z.gam * z.rgp * rop[l + 4 * nd]);
    mach = 10.0 * (1.0 / tc[d]) * (1.0 / mach) * div;
    mach = 0.5 * (1.0 - tanh(2.5 + mach));
    const double p2 = rop[l - inc[d]] * rop[l - inc[d] + 4 * nd], p0 = rop[l] * rop[l + 4 * nd];
    const double p1 = rop[l + inc[d]] * rop[l + inc[d] + 4 * nd];
    double w = fabs(p2 - 2. * p0 + p1) / fabs(p2 + 2. * p0 + p1);
    w = fs_min(2.0 * div * div / (div * div + rot + 1e-15) * w * mach, 1.0);
    wsc[l + d * nd] = 1.0 - w;
  }
}

// ------------------------------------------------------------------------------------------
// joins: ghost[rcv] = donor[dnr] for every variable (type-0 'direct' transfer,
// setInterpTransfersFast.cpp:440-935), pack / unpack for joins that cross contexts
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_join_copy(int npts, int neq, long long nd_r, long long nd_d,
                                                   const int *__restrict__ rcv, const int *__restrict__ dnr,
                                                   double *__restrict__ rop_r, const double *__restrict__ rop_d)
{
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)npts * neq;
       t += (long long)gridDim.x * blockDim.x) {
    int e = (int)(t / npts), n = (int)(t % npts);
    rop_r[e * nd_r + rcv[n]] = rop_d[e * nd_d + dnr[n]];
  }
}

__global__ void __launch_bounds__(256) k_halo_pack(int npts, int neq, long long nd, const int *__restrict__ pts,
                                                   const double *__restrict__ rop, double *__restrict__ buf)
{
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)npts * neq;
       t += (long long)gridDim.x * blockDim.x) {
    int e = (int)(t / npts), n = (int)(t % npts);
    buf[t] = rop[e * nd + pts[n]];
  }
}

__global__ void __launch_bounds__(256) k_halo_unpack(int npts, int neq, long long nd, const int *__restrict__ pts,
                                                     double *__restrict__ rop, const double *__restrict__ buf)
{
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)npts * neq;
       t += (long long)gridDim.x * blockDim.x) {
    int e = (int)(t / npts), n = (int)(t % npts);
    rop[e * nd + pts[n]] = buf[t];
  }
}

// ------------------------------------------------------------------------------------------
// residual norms: per-block partial sums in a fixed order, then one block folds the partials
// in index order, so the result does not depend on scheduling.
// ------------------------------------------------------------------------------------------
// part: 12 doubles per (j,k) row, [0..5] sums of squares, [6..11] largest magnitudes (5 or 6 equations)
__global__ void __launch_bounds__(256) k_residual_partial(FsZone z, int use_coe, const double *__restrict__ drodm,
                                                          double *__restrict__ part, const double *__restrict__ coe1)
{
  __shared__ double ssum[6][256];
  __shared__ double smax[6][256];
  const int j = blockIdx.x + 1, k = blockIdx.y + 1;
  const int neq = z.neq;
  double s[6] = {0, 0, 0, 0, 0, 0}, m[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x + 1; i <= z.iv; i += blockDim.x) {
    long long l = fs_inddm(z, i, j, k);
    double coe = z.coe;
    if (!use_coe) coe = 1.;
    else if (coe1) coe = coe1[l];
    else if (z.ityp == 0)   // dt/vol of the cell (tstb3_global.for:64-78)
      coe = z.dt / z.vol[(long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj +
                         (long long)(k + z.ifick - 1) * z.minck];
    for (int e = 0; e < neq; e++) {
      double r = use_coe ? coe * drodm[e * z.nd + l] : drodm[e * z.nd + l];
      s[e] = s[e] + r * r;
      m[e] = fs_max(m[e], fabs(r));
    }
  }
  for (int e = 0; e < 6; e++) { ssum[e][threadIdx.x] = s[e]; smax[e][threadIdx.x] = m[e]; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int e = 0; e < 6; e++) {
        ssum[e][threadIdx.x] = ssum[e][threadIdx.x] + ssum[e][threadIdx.x + w];
        smax[e][threadIdx.x] = fs_max(smax[e][threadIdx.x], smax[e][threadIdx.x + w]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    for (int e = 0; e < 6; e++) { part[b * 12 + e] = ssum[e][0]; part[b * 12 + 6 + e] = smax[e][0]; }
  }
}

// Norms of the Newton residual per block of SIZE_SSDOM cells (the last block of a direction takes the remainder):
// one workgroup per block, 12 doubles each as in k_residual_partial (cprdu3s1_incl.for:15-140).
__global__ void __launch_bounds__(256) k_residual_blocks(FsZone z, FsBox nb, const double *__restrict__ drodm,
                                                         double *__restrict__ out)
{
  __shared__ double ssum[6][256];
  __shared__ double smax[6][256];
  const int nbi = nb.b[0], nbj = nb.b[1], nbk = nb.b[2];
  const int b = blockIdx.x;
  const int bi = b % nbi, bj = (b / nbi) % nbj, bk = b / (nbi * nbj);
  const int i0 = 1 + bi * nb.b[3], j0 = 1 + bj * nb.b[4], k0 = 1 + bk * nb.b[5];
  const int i1 = bi == nbi - 1 ? z.iv : (bi + 1) * nb.b[3];
  const int j1 = bj == nbj - 1 ? z.jv : (bj + 1) * nb.b[4];
  const int k1 = bk == nbk - 1 ? z.kv : (bk + 1) * nb.b[5];
  const int ni = i1 - i0 + 1, nj = j1 - j0 + 1;
  const long long n = (long long)ni * nj * (k1 - k0 + 1);
  const int neq = z.neq;
  double s[6] = {0, 0, 0, 0, 0, 0}, m[6] = {0, 0, 0, 0, 0, 0};
  for (long long t = threadIdx.x; t < n; t += blockDim.x) {
    const int i = i0 + (int)(t % ni), j = j0 + (int)((t / ni) % nj), k = k0 + (int)(t / ((long long)ni * nj));
    const long long l = fs_inddm(z, i, j, k);
    for (int e = 0; e < neq; e++) {
      const double r = drodm[e * z.nd + l];
      s[e] = s[e] + r * r;
      m[e] = fs_max(m[e], fabs(r));
    }
  }
  for (int e = 0; e < 6; e++) { ssum[e][threadIdx.x] = s[e]; smax[e][threadIdx.x] = m[e]; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int e = 0; e < 6; e++) {
        ssum[e][threadIdx.x] = ssum[e][threadIdx.x] + ssum[e][threadIdx.x + w];
        smax[e][threadIdx.x] = fs_max(smax[e][threadIdx.x], smax[e][threadIdx.x + w]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int e = 0; e < 6; e++) { out[(long long)b * 12 + e] = ssum[e][0]; out[(long long)b * 12 + 6 + e] = smax[e][0]; }
}

__global__ void __launch_bounds__(256) k_residual_final(long long nblk, const double *__restrict__ part,
                                                        double *__restrict__ out)
{
  __shared__ double ssum[6][256];
  __shared__ double smax[6][256];
  double s[6] = {0, 0, 0, 0, 0, 0}, m[6] = {0, 0, 0, 0, 0, 0};
  for (long long b = threadIdx.x; b < nblk; b += blockDim.x)
    for (int e = 0; e < 6; e++) { s[e] = s[e] + part[b * 12 + e]; m[e] = fs_max(m[e], part[b * 12 + 6 + e]); }
  for (int e = 0; e < 6; e++) { ssum[e][threadIdx.x] = s[e]; smax[e][threadIdx.x] = m[e]; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int e = 0; e < 6; e++) {
        ssum[e][threadIdx.x] = ssum[e][threadIdx.x] + ssum[e][threadIdx.x + w];
        smax[e][threadIdx.x] = fs_max(smax[e][threadIdx.x], smax[e][threadIdx.x + w]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int e = 0; e < 6; e++) { out[e] = ssum[e][0]; out[6 + e] = smax[e][0]; }
}

// ------------------------------------------------------------------------------------------
// local CFL number (Compute/cpcfl3.for:276-440 viscous, :635-790 Euler, fixed mesh): per-block
// [max, min, sum] in a fixed order, folded by one block.  A two-dimensional zone runs with its metrics
// expanded by zero z components and a zero tk: the terms it adds are exact zeros.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cfl_partial(FsZone z, double gam1, double gam2, int visc,
                                                     const double *__restrict__ rop, const double *__restrict__ xmut,
                                                     double *__restrict__ part)
{
  __shared__ double sred[3][256];
  const int j = blockIdx.x + 1, k = blockIdx.y + 1;
  double mx = 0., mn = 100000000., sm = 0.;
  double surf_c = 0.;
  if (z.ityp != 0) {
    const double ddi = 2. * sqrt(z.tcx * z.tcx), ddj = 2. * sqrt(z.tcy * z.tcy), ddk = 2. * sqrt(z.tcz * z.tcz);
    surf_c = 1. / fs_max(fs_max(ddi, ddj), ddk);
  }
  for (int i = threadIdx.x + 1; i <= z.iv; i += blockDim.x) {
    const long long l = fs_inddm(z, i, j, k), nd = z.nd;
    double surf = surf_c, volc = z.vol0;
    if (z.ityp == 0) {
      const long long lt = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj +
                           (long long)(k + z.ifick - 1) * z.minck;
      const long long ndm = z.ndm, inc[3] = {1, z.mincj, z.minck};
      double dd[3];
      for (int d = 0; d < 3; d++) {
        const double *__restrict__ t = z.mt[d];
        const long long lp = lt + inc[d];
        double a = sqrt(t[lp] * t[lp] + t[lp + ndm] * t[lp + ndm] + t[lp + 2 * ndm] * t[lp + 2 * ndm]);
        a = a + sqrt(t[lt] * t[lt] + t[lt + ndm] * t[lt + ndm] + t[lt + 2 * ndm] * t[lt + 2 * ndm]);
        dd[d] = a;
      }
      surf = 1. / fs_max(fs_max(dd[0], dd[1]), dd[2]);
      volc = z.vol[lt];
    }
    const double u = rop[l + nd], v = rop[l + 2 * nd], w = rop[l + 3 * nd], r = rop[l];
    const double c = sqrt(gam2 * rop[l + 4 * nd]);
    const double ur2 = u * u + v * v + w * w;
    const double sp = sqrt(ur2);
    const double xinvspc = 1. / (sp + c);
    const double conv = 2. * volc * surf * xinvspc;
    double comp = conv;
    if (visc) {
      const double xinvkt = gam1 / xmut[l];
      const double vis = 2. * r * volc * volc * surf * surf * xinvkt;
      comp = fs_min(vis, conv);
    }
    const double cfloc = z.dt / comp;
    mx = fs_max(mx, cfloc); mn = fs_min(mn, cfloc); sm = sm + cfloc;
  }
  sred[0][threadIdx.x] = mx; sred[1][threadIdx.x] = mn; sred[2][threadIdx.x] = sm;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      sred[0][threadIdx.x] = fs_max(sred[0][threadIdx.x], sred[0][threadIdx.x + w]);
      sred[1][threadIdx.x] = fs_min(sred[1][threadIdx.x], sred[1][threadIdx.x + w]);
      sred[2][threadIdx.x] = sred[2][threadIdx.x] + sred[2][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    part[b * 3] = sred[0][0]; part[b * 3 + 1] = sred[1][0]; part[b * 3 + 2] = sred[2][0];
  }
}

__global__ void __launch_bounds__(256) k_cfl_final(long long nblk, double inv_cells, const double *__restrict__ part,
                                                   double *__restrict__ out)
{
  __shared__ double sred[3][256];
  double mx = 0., mn = 100000000., sm = 0.;
  for (long long b = threadIdx.x; b < nblk; b += blockDim.x) {
    mx = fs_max(mx, part[b * 3]); mn = fs_min(mn, part[b * 3 + 1]); sm = sm + part[b * 3 + 2];
  }
  sred[0][threadIdx.x] = mx; sred[1][threadIdx.x] = mn; sred[2][threadIdx.x] = sm;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      sred[0][threadIdx.x] = fs_max(sred[0][threadIdx.x], sred[0][threadIdx.x + w]);
      sred[1][threadIdx.x] = fs_min(sred[1][threadIdx.x], sred[1][threadIdx.x + w]);
      sred[2][threadIdx.x] = sred[2][threadIdx.x] + sred[2][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sred[0][0]; out[1] = sred[1][0]; out[2] = sred[2][0] * inv_cells; }
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// kinetic energy and enstrophy of a Cartesian zone (POST/post.for:30-62, cp_gradu_3dcart.for, cptaylor.for): per
// (j,k) row [enstrophy, kinetic energy] summed over i in a fixed order, then one block folds the rows in index order
// and ADDS the zone's three sums to out = {enstrophy sum, kinetic-energy sum, volume}.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tgv_partial(FsZone z, const double *__restrict__ rop, double *__restrict__ part)
{
  __shared__ double se[256], st[256];
  const int j = blockIdx.x + 1, k = blockIdx.y + 1;
  const double c1 = 7. / 12, c2 = 1. / 12;       // post.for:26-27
  const double t[3] = {z.tcx, z.tcy, z.tcz};
  const long long inc[3] = {1, z.incj, z.inck};
  const double vol = z.vol0, volinv = 1. / vol;
  double e = 0., tk = 0.;
  for (int i = threadIdx.x + 1; i <= z.iv; i += blockDim.x) {
    const long long l = fs_inddm(z, i, j, k);
    double g[3][3];   // g[component][direction]: velocity gradient (cp_gradu_3dcart.for:72-120, cp_gradu.for:88-101)
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const double *q = rop + (c + 1) * z.nd;
        const double um = (q[l] + q[l - inc[d]]) * c1 - (q[l + inc[d]] + q[l - 2 * inc[d]]) * c2;
        const long long lp = l + inc[d];
        const double up = (q[lp] + q[lp - inc[d]]) * c1 - (q[lp + inc[d]] + q[lp - 2 * inc[d]]) * c2;
        double a = 0. - um * t[d];
        a = a + up * t[d];
        g[c][d] = a * volinv;
      }
    const double rho = rop[l], u1 = rop[l + z.nd], u2 = rop[l + 2 * z.nd], u3 = rop[l + 3 * z.nd];
    tk = tk + 0.5 * rho * (u1 * u1 + u2 * u2 + u3 * u3) * vol;      // cptaylor.for:62
    const double w1 = g[2][1] - g[1][2], w2 = g[0][2] - g[2][0], w3 = g[1][0] - g[0][1];
    e = e + 0.5 * rho * (w1 * w1 + w2 * w2 + w3 * w3) * vol;         // cptaylor.for:68
  }
  se[threadIdx.x] = e; st[threadIdx.x] = tk;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { se[threadIdx.x] = se[threadIdx.x] + se[threadIdx.x + w]; st[threadIdx.x] = st[threadIdx.x] + st[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    part[2 * b] = se[0]; part[2 * b + 1] = st[0];
  }
}

__global__ void __launch_bounds__(256) k_tgv_final(long long nrow, double zone_volume, const double *__restrict__ part,
                                                   double *__restrict__ out)
{
  __shared__ double se[256], st[256];
  double e = 0., tk = 0.;
  for (long long b = threadIdx.x; b < nrow; b += blockDim.x) { e = e + part[2 * b]; tk = tk + part[2 * b + 1]; }
  se[threadIdx.x] = e; st[threadIdx.x] = tk;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { se[threadIdx.x] = se[threadIdx.x] + se[threadIdx.x + w]; st[threadIdx.x] = st[threadIdx.x] + st[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = out[0] + se[0]; out[1] = out[1] + st[0]; out[2] = out[2] + zone_volume; }
}

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

__global__ void __launch_bounds__(256) k_sfd(FsZone z, double chi, double delta, int pull, double *__restrict__ rop,
                                             double *__restrict__ rof, const double *__restrict__ coe1)
{
  const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y, k = 1 + blockIdx.z;
  if (i > z.iv) return;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  double vol = z.vol0;
  if (z.ityp == 0)
    vol = z.vol[(long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck];
  const double cd = chi * delta;
  // dt / vol: the coefficient array of the implicit scheme or of the local time step, else the global time step
  const double c1 = coe1 ? coe1[l] : (z.ityp == 0 ? z.dt / vol : z.coe);
  const double xsi = exp(-(chi + 1. / delta) * c1 * vol);
  for (int e = 0; e < z.neq; e++) {
    const long long m = l + e * z.nd;
    const double f = 1. / (1. + cd) * ((1. - xsi) * rop[m] + (cd + xsi) * rof[m]);
    rof[m] = f;
    if (pull) rop[m] = 1. / (xsi + cd) * ((1. + cd) * xsi * rop[m] + (cd * (1. - xsi)) * f);
  }
}

void fsk_sfd(const FsZone &z, double chi, double delta, int pull, double *rop, double *rof, const double *coe1, hipStream_t s)
{
  dim3 g((unsigned)((z.iv + 255) / 256), (unsigned)z.jv, (unsigned)z.kv);
  hipLaunchKernelGGL(k_sfd, g, dim3(256), 0, s, z, chi, delta, pull, rop, rof, coe1);
}

void fsk_invist(const FsZone &z, int depth, const double *rop, double *xmut, hipStream_t s)
{
  dim3 g(cdiv((long long)(z.iv + 2 * depth) * (z.jv + 2 * depth), 256), z.kv + 2 * (depth < z.ifick ? depth : z.ifick));
  hipLaunchKernelGGL(k_invist, g, dim3(256), 0, s, z, depth, rop, xmut);
}

void fsk_bc_windows(const FsZone &z, int nwin, const FsBcWin *dev_wins, int kinds, long long max_cells, double *rop,
                    hipStream_t s)
{
  if (nwin <= 0) return;
  unsigned gx = cdiv(max_cells, 256);
  if (gx > 1024) gx = 1024;
  if (kinds & 1) hipLaunchKernelGGL(k_bc_copy, dim3(gx, nwin), dim3(256), 0, s, z, nwin, dev_wins, rop);
  if (kinds & 2) hipLaunchKernelGGL(k_bc_windows, dim3(gx, nwin), dim3(256), 0, s, z, nwin, dev_wins, rop);
}

void fsk_bc_edges(const FsZone &z, double *rop, hipStream_t s, const int *box)
{
  int m = z.iv > z.jv ? z.iv : z.jv;
  if (z.kv > m) m = z.kv;
  unsigned gx = cdiv((long long)m * z.ific * z.ific, 256);
  FsBox b = {{1, z.iv, 1, z.jv, 1, z.kv}};
  if (box) for (int q = 0; q < 6; q++) b.b[q] = box[q];
  hipLaunchKernelGGL(k_bc_edges, dim3(gx, 20), dim3(256), 0, s, z, b, rop);
}

void fsk_join_copy(int npts, int neq, long long nd_r, long long nd_d, const int *rcv, const int *dnr,
                   double *rop_r, const double *rop_d, hipStream_t s)
{
  if (npts <= 0) return;
  unsigned gx = cdiv((long long)npts * neq, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_join_copy, dim3(gx), dim3(256), 0, s, npts, neq, nd_r, nd_d, rcv, dnr, rop_r, rop_d);
}

void fsk_halo_pack(int npts, int neq, long long nd, const int *pts, const double *rop, double *buf, hipStream_t s)
{
  if (npts <= 0) return;
  unsigned gx = cdiv((long long)npts * neq, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_halo_pack, dim3(gx), dim3(256), 0, s, npts, neq, nd, pts, rop, buf);
}

void fsk_halo_unpack(int npts, int neq, long long nd, const int *pts, double *rop, const double *buf, hipStream_t s)
{
  if (npts <= 0) return;
  unsigned gx = cdiv((long long)npts * neq, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_halo_unpack, dim3(gx), dim3(256), 0, s, npts, neq, nd, pts, rop, buf);
}

void fsk_cfl(const FsZone &z, double gam1, double gam2, const double *rop, const double *xmut, double *part,
             double *out, hipStream_t s)
{
  hipLaunchKernelGGL(k_cfl_partial, dim3(z.jv, z.kv), dim3(256), 0, s, z, gam1, gam2, z.iflow >= 2 ? 1 : 0, rop, xmut, part);
  hipLaunchKernelGGL(k_cfl_final, dim3(1), dim3(256), 0, s, (long long)z.jv * z.kv,
                     1. / ((double)z.iv * z.jv * z.kv), (const double *)part, out);
}

void fsk_residual(const FsZone &z, int use_coe, const double *drodm, double *part, double *out, hipStream_t s,
                  const double *coe1)
{
  hipLaunchKernelGGL(k_residual_partial, dim3(z.jv, z.kv), dim3(256), 0, s, z, use_coe, drodm, part, coe1);
  hipLaunchKernelGGL(k_residual_final, dim3(1), dim3(256), 0, s, (long long)z.jv * z.kv, part, out);
}

void fsk_sciacovelli(const FsZone &z, const double *rop, double *wsc, hipStream_t s)
{
  dim3 g((unsigned)((z.iv + 1 + 255) / 256), (unsigned)(z.jv + 1), (unsigned)(z.twod ? 1 : z.kv + 1));
  if (z.ityp != 0) hipLaunchKernelGGL(k_sciacovelli<0>, g, dim3(256), 0, s, z, rop, wsc);
  else if (z.twod) hipLaunchKernelGGL(k_sciacovelli<2>, g, dim3(256), 0, s, z, rop, wsc);
  else if (z.homo) hipLaunchKernelGGL(k_sciacovelli<3>, g, dim3(256), 0, s, z, rop, wsc);
  else hipLaunchKernelGGL(k_sciacovelli<1>, g, dim3(256), 0, s, z, rop, wsc);
}

void fsk_residual_blocks(const FsZone &z, const int nblk[3], const int size[3], const double *drodm, double *out,
                         hipStream_t s)
{
  FsBox nb = {{nblk[0], nblk[1], nblk[2], size[0], size[1], size[2]}};
  hipLaunchKernelGGL(k_residual_blocks, dim3((unsigned)(nblk[0] * nblk[1] * nblk[2])), dim3(256), 0, s, z, nb, drodm, out);
}

void fsk_tgv_sums(const FsZone &z, const double *rop, double *part, double *out, hipStream_t s)
{
  hipLaunchKernelGGL(k_tgv_partial, dim3(z.jv, z.kv), dim3(256), 0, s, z, rop, part);
  // the reference adds the cell volume once per cell (cptaylor.for:54); on a Cartesian zone that is cells * volume up
  // to the rounding of the running sum
  hipLaunchKernelGGL(k_tgv_final, dim3(1), dim3(256), 0, s, (long long)z.jv * z.kv,
                     (double)((long long)z.iv * z.jv * z.kv) * z.vol0, part, out);
}

// ---- streaming ceiling ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_triad(long long n2, double2 *__restrict__ a, const double2 *__restrict__ b,
                                               const double2 *__restrict__ c, double s)
{
  // four 16-byte loads of each input in flight per lane, many short workgroups: the shape that streamed fastest on
  // MI355X among those tried (profiles/triad_sweep.hip)
  constexpr int U = 4;
  const long long stride = (long long)gridDim.x * blockDim.x * U;
  for (long long i0 = (long long)blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n2; i0 += stride) {
    double2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long i = i0 + u * 256;
      if (i < n2) { x[u] = b[i]; y[u] = c[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long i = i0 + u * 256;
      if (i < n2) a[i] = make_double2(x[u].x + s * y[u].x, x[u].y + s * y[u].y);
    }
  }
}

void fsk_triad(long long n, double *a, const double *b, const double *c, double s, hipStream_t st)
{
  const long long n2 = n / 2;
  long long g = (n2 + 1023) / 1024;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(k_triad, dim3((unsigned)g), dim3(256), 0, st, n2, (double2 *)a, (const double2 *)b, (const double2 *)c, s);
}
