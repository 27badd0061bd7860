// One Runge-Kutta stage on a curvilinear zone (zone type 0: surface vectors and volume stored per
// cell), gfx950.  Built with -ffp-contract=off; every expression keeps the operand order of the
// reference fragments listed in fasts_flux.h, so the balance is bit-identical to the reference's
// with one cache block.
//
// Work decomposition.  A block of 256 threads covers a 32 x 8 patch of (i,j) columns and marches
// along k over a chunk of planes.  Each thread evaluates the three "minus" faces of its cell
// (k-1/2, j-1/2, i-1/2); the "plus" faces are the minus faces of the neighbours: the i+1/2 flux comes
// from the next lane (wave shuffle), the j+1/2 flux from the next row (through LDS, double
// buffered: one barrier per plane), the k+1/2 flux is the k-1/2 flux of the next step of the march.
// The last lane of a row and the last row of a patch only produce faces for their neighbours, so
// neighbouring patches overlap by one column / one row: 31 x 7 cells are owned per patch.
//
// Accumulation order of a cell = the order in which the reference's scatter loops reach it:
//   drodm*coefH + Fk- + Fj- + Fi- - Fi+ - Fj+ - Fk+       (template_FluxAndBalance.for:192-271)
// The own column's four k-planes are carried in registers.  On laminar three-dimensional zones the planes k-1, k, k+1
// of the patch (2-cell margin) sit in a ring of three LDS slots, one new plane per step, and the transverse neighbours
// are read from there; metrics and viscosity, and the neighbours of the other variants, come from global memory
// through the vector L1 / L2.
#include "fasts_kernels.h"
#include "fasts_flux.h"

#include <type_traits>

namespace {

constexpr int TI = 32, TJ = 8, OI = TI - 1, OJ = TJ - 1;

template <int DIR> FS_DEV long long inc_of(const FsZone &z) { return DIR == 0 ? 1 : (DIR == 1 ? z.incj : z.inck); }
template <int DIR> FS_DEV long long minc_of(const FsZone &z) { return DIR == 0 ? 1 : (DIR == 1 ? z.mincj : z.minck); }

// Face between cell l - inc(DIR) and cell l.  qm2..qp1 = the four cells of the MUSCL stencil.
// FIRST: the sensor of the face is evaluated from the stencil and stored (0/1 byte); later stages read it.
// TWOD: two-dimensional zone; the k stride of the field is taken as 0 (the cross terms along k then read the
// cell itself and meet zero metrics: they vanish exactly, as in the reference's 2d fragments fluvisq_2d_{i,j}.for).
// Readers of the field around the thread's own cell, rd(e, di, dj, dk) = variable e at (i+di, j+dj, k+dk).
// RdGlobal: from global memory (the k stride is 0 on a two-dimensional zone).  RdLds: from the three planes
// k-1, k, k+1 of the patch staged in LDS (LI x LJ cells per variable and plane, halo of 2 cells).
constexpr int LI = TI + 4, LJ = TJ + 4;
struct RdGlobal {
  const double *__restrict__ rop;
  long long nd, l, incj, inck;
  __device__ __forceinline__ double operator()(int e, int di, int dj, int dk) const { return rop[e * nd + l + di + dj * incj + dk * inck]; }
};
struct RdLds {
  const double *pm, *p0, *pp;   // own cell in the planes k-1, k, k+1 (variable 0)
  __device__ __forceinline__ double operator()(int e, int di, int dj, int dk) const
  {
    const double *p = dk < 0 ? pm : (dk == 0 ? p0 : pp);
    return p[(e * LJ + dj) * LI + di];
  }
};

template <int KFLU, int IFLOW, int DIR, bool FIRST, bool TWOD, bool HOMO = false, class RD>
FS_DEV void face_full(const FsZone &z, const RD &rd, const double *__restrict__ xmut,
                      unsigned char *__restrict__ wig, long long l, long long lt, const double *qm2,
                      const double *qm1, const double *q0, const double *qp1, double *flu, double csc = -1.)
{
  const long long nd = z.nd, ndm = z.ndm;
  const double *__restrict__ Tn = z.mt[DIR];
  const double tcx = Tn[lt], tcy = Tn[lt + ndm], tcz = Tn[lt + 2 * ndm];
  double wg = 0.;
  if (KFLU == 2) {
    if (FIRST) {
      auto qs = [&](int e, int o) -> double { return o == -2 ? qm2[e] : (o == -1 ? qm1[e] : (o == 0 ? q0[e] : qp1[e])); };
      wg = fs_wiggle(qs);
      wig[DIR * nd + l] = wg != 0. ? 1 : 0;
    } else {
      wg = (double)wig[DIR * nd + l];
    }
  }
  double sa[4];
  fs_flux_inviscid_full<KFLU>(z, tcx, tcy, tcz, qm2, qm1, q0, qp1, wg, flu, TWOD && DIR == 1,
                              (IFLOW == 3 && KFLU != 5) ? sa : nullptr, csc);
  if (IFLOW == 3 && KFLU == 5) {
    // roe: first-order upwinding with the Roe-averaged normal velocity (ROE/fluFaceSA.for, Vit_ent/qn_3dfull_i.for)
    double qp1r = qm1[0], qm1r = q0[0];
    double qp2 = qm1[1], qp3 = qm1[2], qp4 = qm1[3];
    double qn1 = qp2 * tcx + qp3 * tcy + qp4 * tcz;
    const double qn2 = q0[1] * tcx + q0[2] * tcy + q0[3] * tcz;
    const double qm6 = q0[5] * qm1r * qn2, qp6 = qm1[5] * qp1r * qn1;
    qp1r = sqrt(qp1r); qm1r = sqrt(qm1r);
    const double r_1 = 1. / (qp1r + qm1r);
    qp2 = r_1 * (qp1r * qp2 + qm1r * q0[1]); qp3 = r_1 * (qp1r * qp3 + qm1r * q0[2]); qp4 = r_1 * (qp1r * qp4 + qm1r * q0[3]);
    qn1 = qp2 * tcx + qp3 * tcy + qp4 * tcz;
    const double s = sqrt(tcx * tcx + tcy * tcy + tcz * tcz);
    const double s_1 = 1. / fs_max(s, 1.e-30);
    const double tdu = fs_max(z.c1 / s_1, fabs(qn1));
    flu[5] = 0.5 * (qm6 + qp6 - tdu * (q0[5] * q0[0] - qm1[5] * qm1[0]));
  } else if (IFLOW == 3) {   // AUSM/fluFaceSA.for, SENSOR/fluFaceSA.for
    const double qm = z.c4 * q0[5] + z.c5 * qm1[5] + z.c6 * qp1[5];
    const double qp = z.c4 * qm1[5] + z.c6 * qm2[5] + z.c5 * q0[5];
    const double qm6 = qm * sa[2], qp6 = qp * sa[3];
    flu[5] = sa[0] * (qm6 + qp6) - sa[1] * (qm6 - qp6);
  }
  if (IFLOW >= 2) {
    constexpr int A = (DIR + 1) % 3, B = (DIR + 2) % 3;
    const long long inc = inc_of<DIR>(z);
    const long long md = minc_of<DIR>(z), ma = minc_of<A>(z), mb = minc_of<B>(z);
    auto q = [&](int e, int da, int db, int dn) -> double {
      if (da == 0 && db == 0) return dn == 0 ? q0[e] : qm1[e];
      int o[3];
      o[DIR] = dn; o[A] = da; o[B] = db;
      return rd(e, o[0], o[1], o[2]);
    };
    auto T = [&](int c, int m, int o) -> double {
      const double *__restrict__ t = c == 0 ? Tn : (c == 1 ? z.mt[A] : z.mt[B]);
      const long long mc = c == 1 ? ma : mb;
      const long long off = o == 0 ? 0 : (o == 1 ? -md : (o == 2 ? md : (o == 3 ? mc : mc - md)));
      return t[lt + off + m * ndm];
    };
    fs_flux_viscous_full<IFLOW == 3, DIR, HOMO>(z, q, T, z.vol[lt], z.vol[lt - md], xmut[l - inc], xmut[l], tcx, tcy, tcz, flu);
  }
}

// Spalart-Allmaras source term of cell l on a curvilinear / two-dimensional zone: vol * (production - destruction
// + cb2 term) (SA/spsource_SA.for:100-130, SA/sourceSA_grad_3d.for, sourceSA_grad_2d.for, sourceSA_prod_dest.for).
// qc = state of the cell; tsb / amu (may be null): production - destruction and the clipped rho*nu_tilde
// (SA/sourceSA_LU.for).
template <bool TWOD, bool HOMO = false>
FS_DEV double sa_source_full(const FsZone &z, const double *__restrict__ rop, const double *__restrict__ dlng, long long l,
                             long long lt, const double *qc, double *tsb = nullptr, double *amu = nullptr)
{
  const double SA_CKARM = 0.41, SA_CB1 = 0.1355, SA_CB2 = 0.622, SA_CV1 = 357.911, SA_CW2 = 0.3, SA_CW3 = 64.;
  const double SA_CW4 = 0.21, SA_CW5 = 1.5, SA_CROT = 2.0, SA_SIGMA = (2. / 3.);
  const double c1 = SA_CB2 / SA_SIGMA;
  const double cw1 = (SA_CB1 / SA_CKARM / SA_CKARM) + (1. + SA_CB2) / SA_SIGMA;
  const long long nd = z.nd, incj = z.incj, inck = TWOD ? 0 : z.inck, ndm = z.ndm;
  const long long l1 = l + 1, l2 = l - 1, l3 = l + incj, l4 = l - incj, l5 = l + inck, l6 = l - inck;
  double t0[3][3], t1[3][3];   // surface vectors of the faces at lt and at lt + inc_mtr(d)
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const long long md = d == 0 ? 1 : (d == 1 ? z.mincj : z.minck);
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const bool have = !(TWOD && (d == 2 || m == 2));
      t0[d][m] = have ? z.mt[d][lt + m * ndm] : 0.;
      t1[d][m] = have ? z.mt[d][lt + md + m * ndm] : 0.;
    }
  }
  auto grad = [&](double u1, double u2, double u3, double u4, double u5, double u6, int m) -> double {
    if (TWOD) return u3 * t1[0][m] - u1 * t0[0][m] + u2 * t1[1][m] - u4 * t0[1][m];
    return u3 * t1[0][m] - u1 * t0[0][m] + u2 * t1[1][m] - u4 * t0[1][m] + u5 * t1[2][m] - u6 * t0[2][m];
  };
  const double xvol = 0.5 / z.vol[lt];
  double g[3][3];
#pragma unroll
  for (int v = 0; v < 3; v++) {
    const double *__restrict__ q = rop + (1 + v) * nd;
    const double c = qc[1 + v];
    const double u1 = (c + q[l2]), u2 = (c + q[l3]), u3 = (c + q[l1]);
    const double u4 = (c + q[l4]), u5 = (c + q[l5]), u6 = (c + q[l6]);
#pragma unroll
    for (int m = 0; m < 3; m++) g[v][m] = (TWOD && (m == 2 || v == 2)) ? 0. : grad(u1, u2, u3, u4, u5, u6, m);
    if (HOMO) g[v][2] = (u5 - u6) * t0[2][2];   // SA/sourceSA_grad_3dhomo.for
  }
  const double rotx = g[2][1] - g[1][2], roty = g[0][2] - g[2][0], rotz = g[1][0] - g[0][1];
  const double rot = (TWOD ? fabs(rotz) : sqrt(rotx * rotx + roty * roty + rotz * rotz)) * xvol;
  const double S11 = g[0][0], S22 = g[1][1], S33 = g[2][2];
  const double S12 = 0.5 * (g[0][1] + g[1][0]), S13 = 0.5 * (g[0][2] + g[2][0]), S23 = 0.5 * (g[1][2] + g[2][1]);
  double St = TWOD ? S11 * S11 + S22 * S22 + 2 * (S12 * S12)
                   : S11 * S11 + S22 * S22 + S33 * S33 + 2 * (S12 * S12) + 2 * (S13 * S13) + 2 * (S23 * S23);
  St = sqrt(2 * St) * xvol;
  const double *__restrict__ N = rop + 5 * nd, *__restrict__ R = rop;
  const double nc = qc[5], rc = qc[0];
  const double n1 = N[l1], n2 = N[l2], n3 = N[l3], n4 = N[l4], n5 = N[l5], n6 = N[l6];
  double u1 = (nc + n2), u2 = (nc + n3), u3 = (nc + n1), u4 = (nc + n4), u5 = (nc + n5), u6 = (nc + n6);
  double dudx = grad(u1, u2, u3, u4, u5, u6, 0), dudy = grad(u1, u2, u3, u4, u5, u6, 1);
  double dudz = TWOD ? 0. : grad(u1, u2, u3, u4, u5, u6, 2);
  if (HOMO) dudz = (u5 - u6) * t0[2][2];
  const double dudz0 = dudz;
  u1 = nc * rc + n2 * R[l2]; u2 = nc * rc + n3 * R[l3]; u3 = nc * rc + n1 * R[l1];
  u4 = nc * rc + n4 * R[l4]; u5 = nc * rc + n5 * R[l5]; u6 = nc * rc + n6 * R[l6];
  dudx = dudx * grad(u1, u2, u3, u4, u5, u6, 0); dudy = dudy * grad(u1, u2, u3, u4, u5, u6, 1);
  dudz = TWOD ? 0. : dudz * grad(u1, u2, u3, u4, u5, u6, 2);
  if (HOMO) dudz = dudz0 * (u5 - u6) * t0[2][2];
  const double anvisc = TWOD ? c1 * (dudx + dudy) * xvol * xvol : c1 * (dudx + dudy + dudz) * xvol * xvol;
  const double amulam = fs_mulam_sa(z, qc[4]);
  const double anulam = rc / amulam;
  const double anutild = fs_max(nc, 1.e-27);
  const double chi = fs_max(anutild * anulam, 1.e-7);
  const double f1 = 1. / (1. + SA_CV1 / (chi * chi * chi));
  const double f2 = 1. - chi / (1. + chi * f1);
  const double dist = fs_max(dlng[l], 1.e-27);
  double stild = rot + (anutild * f2) / (SA_CKARM * SA_CKARM * dist * dist);
  stild = stild + z.sa_rot_corr * SA_CROT * fs_min(0.00000000000000000001, St - rot);
  const double prod = rc * SA_CB1 * stild * anutild;
  stild = fs_max(stild, 0.00000000000000000001);
  double r = anutild / (stild * SA_CKARM * SA_CKARM * dist * dist);
  r = fs_min(r, 10.);
  const double r5 = r * r * r * r * r - 1.;
  double gg = r * (1. + SA_CW2 * r5);
  const double cw2_lre = SA_CW4 + SA_CW5 / ((1. + chi / 40.) * (1. + chi / 40.));
  gg = z.sa_low_re * r * (1. + cw2_lre * r5) + (1. - z.sa_low_re) * gg;
  const double g6 = gg * gg * gg * gg * gg * gg;
  const double fwg = gg * pow((1. + SA_CW3) / (g6 + SA_CW3), 1. / 6.);
  const double destruc = rc * cw1 * fwg * (anutild / dist) * (anutild / dist);
  const double tsource = (prod + anvisc - destruc);
  if (tsb) { *tsb = prod - destruc; *amu = fs_max(anutild * rc, 0.0000000000000001); }
  return z.vol[lt] * tsource;
}

// BDF = false: Runge-Kutta stage.  BDF = true: Newton residual of the implicit scheme (rop_out unused).
// TWOD: one plane, no k faces: the cell is finished in the same step (+Fj- +Fi- -Fi+ -Fj+).
// LQ: the planes k-1, k, k+1 of the patch (halo included) are staged in LDS, one new plane per step of the march, and
// the transverse neighbours are read from there instead of the vector L1 (three-dimensional zones, 5 variables).
// Spalart-Allmaras with the sensor scheme on a three-dimensional zone wants 300-314 registers: held to 256 (two waves per
// SIMD instead of one, 28-92 bytes of scratch per lane) it runs 6.89 -> 6.39 ms per stage at 256^3; the same limit on the
// ausmpred variants (256 registers already) only adds scratch: 5.71 -> 7.28 ms.
template <int KFLU, int IFLOW, bool FIRST, bool BDF, bool TWOD, bool HOMO = false, bool LQ = false>
__global__ void __launch_bounds__(256, (IFLOW == 3 && !TWOD && KFLU == 2) ? 2 : 1)
k_stage_full(FsZone z, double coefH, double coefW, int kchunk, int keep, const double *__restrict__ rop,
             double *__restrict__ rop_out, double *__restrict__ drodm, const double *__restrict__ xmut,
             unsigned char *__restrict__ wig, FsBdf bdf, const double *__restrict__ dlng,
             const int *__restrict__ blist, int npx, int npy, int npz)
{
  constexpr int NQ = IFLOW == 3 ? 6 : 5;   // Spalart-Allmaras: 6th variable nu_tilde
  __shared__ double sfj[2][NQ][TJ][TI];
  __shared__ double spl[LQ ? 3 : 1][LQ ? NQ * LJ * LI : 1];
  static_assert(!LQ || (!TWOD && NQ == 5), "LDS planes: three-dimensional zones with 5 variables");
  const int ti = threadIdx.x & 31, tj = threadIdx.x >> 5;
  // Workgroups go to the 8 XCDs in turn, in the order of their linear index, and each XCD has its own L2.  The index
  // is remapped so that an XCD gets a run of consecutive patches (i fastest, then j, then the k chunk): neighbouring
  // patches overlap by their halo (36 x 12 cells staged for 31 x 7 owned) and read the same metrics at their borders,
  // and now mostly meet in one L2 (256^3: the Euler variants 1.55 -> 1.50 ms per stage; the laminar ones, which are not
  // bound by memory, unchanged at 2.18 ms).
  // blist: a subset of the patches (a one-dimensional grid over the list; patch number = px + npx * (py + npy * pz)):
  // the stage of a zone whose cells feed halo messages runs in two launches, those patches first.
  int pbx, pby, pbz;
  {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned q8 = total / 8, r8 = total % 8, x = lin % 8;
    lin = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + lin / 8;
    if (blist) lin = (unsigned)blist[lin];
    pbx = (int)(lin % (unsigned)npx); pby = (int)((lin / (unsigned)npx) % (unsigned)npy); pbz = (int)(lin / (unsigned)(npx * npy));
  }
  const int i = 1 + pbx * OI + ti, j = 1 + pby * OJ + tj;
  const int k0 = 1 + pbz * kchunk;
  const int k1 = k0 + kchunk - 1 < z.kv ? k0 + kchunk - 1 : z.kv;
  const bool active = i <= z.iv + 1 && j <= z.jv + 1;     // columns this thread may read
  const bool own_i = ti < OI && i <= z.iv, own_j = tj < OJ && j <= z.jv;
  const bool own = own_i && own_j;                         // cells this thread updates
  const bool do_fi = active && own_j, do_fj = active && own_i;
  const long long nd = z.nd;
  // clamp the column of inactive threads onto a valid one: they compute nothing, but the shuffles below
  // are executed by every lane
  const int ic = active ? i : 1, jc = active ? j : 1;
  long long l = (long long)(ic + z.ific - 1) + (long long)(jc + z.ific - 1) * z.incj + (long long)(k0 + z.ifick - 1) * z.inck;
  long long lt = (long long)(ic + z.ific - 1) + (long long)(jc + z.ific - 1) * z.mincj + (long long)(k0 + z.ifick - 1) * z.minck;
  double qa[NQ], qb[NQ], qc[NQ], qd[NQ];   // own column at k-2, k-1, k, k+1
#pragma unroll
  for (int e = 0; e < NQ; e++) {
    qa[e] = TWOD ? 0. : rop[e * nd + l - 2 * z.inck];
    qb[e] = TWOD ? 0. : rop[e * nd + l - z.inck];
    qc[e] = rop[e * nd + l];
    qd[e] = 0.;
  }
  double part[NQ] = {};
  // stage plane kk of the patch (cells i0-2 .. i0+TI+1, j0-2 .. j0+TJ+1, clipped to the array) into slot sl
  auto stage_plane = [&](int kk, int sl) {
    const int i0 = 1 + pbx * OI, j0 = 1 + pby * OJ;
    for (int idx = threadIdx.x; idx < LI * LJ; idx += 256) {
      const int jj = idx / LI, ii = idx - jj * LI;
      const int gi = i0 - 2 + ii, gj = j0 - 2 + jj;
      if (gi > z.iv + z.ific || gj > z.jv + z.ific) continue;
      const long long lg = (long long)(gi + z.ific - 1) + (long long)(gj + z.ific - 1) * z.incj + (long long)(kk + z.ifick - 1) * z.inck;
#pragma unroll
      for (int e = 0; e < NQ; e++) spl[sl][e * LJ * LI + idx] = rop[e * nd + lg];
    }
  };
  int sm = 0, s0 = 1, sp = 2;   // slots of the planes k-1, k, k+1
  if (LQ) { stage_plane(k0 - 1, 0); stage_plane(k0, 1); }
  for (int k = k0; k <= (TWOD ? k0 : k1 + 1); k++) {
    const bool last = !TWOD && k > k1;
    if (!TWOD) {
#pragma unroll
      for (int e = 0; e < NQ; e++) qd[e] = rop[e * nd + l + z.inck];
    }
    if (LQ) {   // plane k+1 takes the slot plane k-2 has left (last read before the barrier of the previous step)
      stage_plane(k + 1, sp);
      __syncthreads();
    }
    const int own_off = (tj + 2) * LI + ti + 2;
    const RdLds rdl = {LQ ? &spl[sm][own_off] : nullptr, LQ ? &spl[s0][own_off] : nullptr, LQ ? &spl[sp][own_off] : nullptr};
    const RdGlobal rdg = {rop, nd, l, z.incj, TWOD ? 0 : z.inck};
    double fk[NQ] = {};
    // finish: the balance of a cell is complete (3-D: cell k-1 once its k+1/2 face is known; 2-D: the cell itself)
    auto finish = [&](long long lb, long long ltb, const double *qcell, const double *d, int kb) {
        double o[NQ];
        const double coe = z.dt / z.vol[ltb];
        if (BDF) {
          if (!fs_in_box(bdf, i, j, kb)) return;   // a sub-zone: the other cells keep their residual
          double qn[NQ] = {}, qn1[NQ];
#pragma unroll
          for (int e = 0; e < NQ; e++) qn1[e] = bdf.ro_n1[e * nd + lb];
          if (!bdf.nit1) {
#pragma unroll
            for (int e = 0; e < NQ; e++) qn[e] = bdf.ro_n[e * nd + lb];
          }
          const double coeb = bdf.coe1 ? bdf.coe1[lb] : coe;
          if (NQ == 6 && bdf.nit1 && bdf.coe1) {   // local time step, first sub-iteration (core3as2.for:57-66)
#pragma unroll
            for (int e = 0; e < NQ; e++) o[e] = coeb * d[e];
          } else {
            fs_bdf_residual(z, bdf, coeb, qcell, qn, qn1, d, o);
            if (NQ == 6) {
              if (bdf.nit1) o[NQ - 1] = coeb * d[NQ - 1] + (bdf.c1 + bdf.c2) * qcell[NQ - 1] * qcell[0] + bdf.c3 * qn1[NQ - 1] * qn1[0];
              else o[NQ - 1] = coeb * d[NQ - 1] + bdf.c1 * qcell[NQ - 1] * qcell[0] + bdf.c2 * qn[NQ - 1] * qn[0] +
                               bdf.c3 * qn1[NQ - 1] * qn1[0];
            }
          }
#pragma unroll
          for (int e = 0; e < NQ; e++) drodm[e * nd + lb] = o[e];
        } else {
          if (keep) {   // the balance of the last Runge-Kutta stage of a step is read by nothing
#pragma unroll
            for (int e = 0; e < NQ; e++) drodm[e * nd + lb] = d[e];
          }
          fs_rk_update_c(z, coefW, coe, qcell, d, o);
          if (TWOD) o[3] = 0.;   // core3ark3.for:156-183
          if (NQ == 6) {   // nu_tilde clipped to [0, ratiom * nu_laminar] (core3ark3.for:196-236)
            const double r_1 = 1. / o[0];
            const double tab6 = d[NQ - 1] * (coefW * coe);
            double anulam = fs_mulam_sa(z, o[4]);
            anulam = anulam * r_1 * z.ratiom;
            double t6 = (qcell[0] * qcell[NQ - 1] + tab6) * r_1;
            t6 = fs_min(t6, anulam);
            o[NQ - 1] = fs_max(t6, 0.);
          }
#pragma unroll
          for (int e = 0; e < NQ; e++) rop_out[e * nd + lb] = o[e];
        }
    };
    if (own && !TWOD) {
      if constexpr (LQ) face_full<KFLU, IFLOW, 2, FIRST, TWOD, HOMO>(z, rdl, xmut, wig, l, lt, qa, qb, qc, qd, fk);
      else face_full<KFLU, IFLOW, 2, FIRST, TWOD, HOMO>(z, rdg, xmut, wig, l, lt, qa, qb, qc, qd, fk);
      if (k > k0) {   // the cell below is complete
        double d[NQ];
#pragma unroll
        for (int e = 0; e < NQ; e++) d[e] = part[e] - fk[e];
        finish(l - z.inck, lt - z.minck, qb, d, k - 1);
      }
    }
    if (!last) {
      double fim[NQ] = {}, fjm[NQ] = {};
      if (do_fj) {
        double m2[NQ], m1[NQ], p1[NQ];
        if constexpr (LQ) {
#pragma unroll
          for (int e = 0; e < NQ; e++) { m2[e] = rdl(e, 0, -2, 0); m1[e] = rdl(e, 0, -1, 0); p1[e] = rdl(e, 0, 1, 0); }
          face_full<KFLU, IFLOW, 1, FIRST, TWOD, HOMO>(z, rdl, xmut, wig, l, lt, m2, m1, qc, p1, fjm);
        } else {
#pragma unroll
          for (int e = 0; e < NQ; e++) {
            m2[e] = rop[e * nd + l - 2 * z.incj]; m1[e] = rop[e * nd + l - z.incj]; p1[e] = rop[e * nd + l + z.incj];
          }
          face_full<KFLU, IFLOW, 1, FIRST, TWOD, HOMO>(z, rdg, xmut, wig, l, lt, m2, m1, qc, p1, fjm);
        }
      }
      const int buf = (k - k0) & 1;
#pragma unroll
      for (int e = 0; e < NQ; e++) sfj[buf][e][tj][ti] = fjm[e];
      if (do_fi) {
        double m2[NQ], m1[NQ], p1[NQ];
        if constexpr (LQ) {
#pragma unroll
          for (int e = 0; e < NQ; e++) { m2[e] = rdl(e, -2, 0, 0); m1[e] = rdl(e, -1, 0, 0); p1[e] = rdl(e, 1, 0, 0); }
          face_full<KFLU, IFLOW, 0, FIRST, TWOD, HOMO>(z, rdl, xmut, wig, l, lt, m2, m1, qc, p1, fim);
        } else {
#pragma unroll
          for (int e = 0; e < NQ; e++) {
            m2[e] = rop[e * nd + l - 2]; m1[e] = rop[e * nd + l - 1]; p1[e] = rop[e * nd + l + 1];
          }
          face_full<KFLU, IFLOW, 0, FIRST, TWOD, HOMO>(z, rdg, xmut, wig, l, lt, m2, m1, qc, p1, fim);
        }
      }
      double fip[NQ];
#pragma unroll
      for (int e = 0; e < NQ; e++) fip[e] = __shfl_down(fim[e], 1);
      __syncthreads();
      if (own) {
        double src = 0.;
        if (NQ == 6) src = sa_source_full<TWOD, HOMO>(z, rop, dlng, l, lt, qc);
#pragma unroll
        for (int e = 0; e < NQ; e++) {
          double dr = (FIRST || BDF) ? 0. : drodm[e * nd + l] * coefH;
          if (NQ == 6 && e == 5) dr = dr + src;
          part[e] = dr + fk[e] + fjm[e] + fim[e] - fip[e] - sfj[buf][e][tj + 1][ti];
        }
        if (TWOD) finish(l, lt, qc, part, k);
      }
    }
#pragma unroll
    for (int e = 0; e < NQ; e++) { qa[e] = qb[e]; qb[e] = qc[e]; qc[e] = qd[e]; }
    l += z.inck; lt += z.minck;
    { const int t = sm; sm = s0; s0 = sp; sp = t; }
  }
}

// coe(:,6) of the implicit Spalart-Allmaras scheme on a curvilinear / two-dimensional zone (SA/sourceSA_LU.for),
// interior plus one ghost layer (none along k on a two-dimensional zone)
template <bool TWOD, bool HOMO = false>
__global__ void __launch_bounds__(256) k_sa_coe6_full(FsZone z, const double *__restrict__ rop,
                                                      const double *__restrict__ dlng, double *__restrict__ coe)
{
  const int gk = TWOD ? 0 : 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, k = blockIdx.z + 1 - gk;
  if (i > z.iv + 1) return;
  const long long nd = z.nd;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  const long long lt = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck;
  double qc[6];
#pragma unroll
  for (int e = 0; e < 6; e++) qc[e] = rop[e * nd + l];
  double tsb, amu;
  sa_source_full<TWOD, HOMO>(z, rop, dlng, l, lt, qc, &tsb, &amu);
  const double tsourcenu = z.vol[lt] * tsb / amu;
  coe[l + 5 * nd] = coe[l + 4 * nd] - fs_min(tsourcenu, 0.) * coe[l];
}

// o3sc slope on curvilinear / two-dimensional / k-homogeneous zones, 5 equations: the fluxes of the faces of one
// direction per launch, one thread per face, operands from memory through the face function of k_stage_full (with the
// weight of the shock sensor), into an array of 15 doubles per cell; then one thread per cell adds them up in the order
// k_stage_full has and ends the stage the same way.  (All six faces of a cell in one thread want some 900 registers.)
template <int KFLU, int IFLOW, int DIR, bool FIRST, bool TWOD, bool HOMO>
__global__ void __launch_bounds__(256) k_faces_o3sc_full(FsZone z, const double *__restrict__ rop,
                                                         const double *__restrict__ xmut, unsigned char *__restrict__ wig,
                                                         const double *__restrict__ wsc, double *__restrict__ flux)
{
  // faces of direction DIR: their right cells run to n + 1 along DIR
  const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y, k = 1 + blockIdx.z;
  if (i > z.iv + (DIR == 0 ? 1 : 0) || j > z.jv + (DIR == 1 ? 1 : 0) || k > z.kv + (DIR == 2 ? 1 : 0)) return;
  const long long nd = z.nd;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  const long long lt = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck;
  const long long st = DIR == 0 ? 1 : (DIR == 1 ? z.incj : z.inck);
  double qm2[5], qm1[5], q0[5], qp1[5], f[5];
#pragma unroll
  for (int e = 0; e < 5; e++) {
    qm2[e] = rop[e * nd + l - 2 * st]; qm1[e] = rop[e * nd + l - st]; q0[e] = rop[e * nd + l]; qp1[e] = rop[e * nd + l + st];
  }
  const RdGlobal rdg = {rop, nd, l, z.incj, TWOD ? 0 : z.inck};
  face_full<KFLU, IFLOW, DIR, FIRST, TWOD, HOMO>(z, rdg, xmut, wig, l, lt, qm2, qm1, q0, qp1, f, wsc[l + DIR * nd]);
#pragma unroll
  for (int e = 0; e < 5; e++) flux[(DIR * 5 + e) * nd + l] = f[e];
}

template <bool FIRST, bool BDF, bool TWOD>
__global__ void __launch_bounds__(256) k_balance_o3sc_full(FsZone z, double coefH, double coefW, int keep,
                                                           const double *__restrict__ rop, double *__restrict__ rop_out,
                                                           double *__restrict__ drodm, const double *__restrict__ flux,
                                                           FsBdf bdf)
{
  const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y, k = 1 + blockIdx.z;
  if (i > z.iv) return;
  const long long nd = z.nd;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  const long long lt = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck;
  double q[5], d[5], o[5];
#pragma unroll
  for (int e = 0; e < 5; e++) {
    q[e] = rop[e * nd + l];
    const double dr = (FIRST || BDF) ? 0. : drodm[e * nd + l] * coefH;
    double a = dr + (TWOD ? 0. : flux[(10 + e) * nd + l]);
    a = a + flux[(5 + e) * nd + l]; a = a + flux[e * nd + l];
    a = a - flux[e * nd + l + 1]; a = a - flux[(5 + e) * nd + l + z.incj];
    d[e] = TWOD ? a : a - flux[(10 + e) * nd + l + z.inck];
  }
  const double coe = z.dt / z.vol[lt];
  if (BDF) {
    double qn[5] = {0., 0., 0., 0., 0.}, qn1[5];
#pragma unroll
    for (int e = 0; e < 5; e++) qn1[e] = bdf.ro_n1[e * nd + l];
    if (!bdf.nit1) {
#pragma unroll
      for (int e = 0; e < 5; e++) qn[e] = bdf.ro_n[e * nd + l];
    }
    const double coeb = bdf.coe1 ? bdf.coe1[l] : coe;
    fs_bdf_residual(z, bdf, coeb, q, qn, qn1, d, o);
#pragma unroll
    for (int e = 0; e < 5; e++) drodm[e * nd + l] = o[e];
  } else {
    if (keep) {
#pragma unroll
      for (int e = 0; e < 5; e++) drodm[e * nd + l] = d[e];
    }
    fs_rk_update_c(z, coefW, coe, q, d, o);
    if (TWOD) o[3] = 0.;   // core3ark3.for:156-183
#pragma unroll
    for (int e = 0; e < 5; e++) rop_out[e * nd + l] = o[e];
  }
}

template <int KFLU, int IFLOW, bool TWOD, bool HOMO>
__global__ void __launch_bounds__(128) k_box_faces_full(FsZone z, FsBox box, const double *__restrict__ rop,
                                                        const double *__restrict__ xmut, unsigned char *__restrict__ wig,
                                                        double *__restrict__ drodm)
{
  constexpr int NQ = IFLOW == 3 ? 6 : 5;
  const int side = blockIdx.y;
  if (TWOD && side >= 4) return;
  int i, j, k;
  if (!box_face_cell(z, box, side, (long long)blockIdx.x * blockDim.x + threadIdx.x, i, j, k)) return;
  const long long nd = z.nd;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  const long long lt = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck;
  const int d = side >> 1;
  const long long st = d == 0 ? 1 : (d == 1 ? z.incj : z.inck);
  double qm2[NQ], qm1[NQ], q0[NQ], qp1[NQ], f[NQ];
#pragma unroll
  for (int e = 0; e < NQ; e++) {
    qm2[e] = rop[e * nd + l - 2 * st]; qm1[e] = rop[e * nd + l - st]; q0[e] = rop[e * nd + l]; qp1[e] = rop[e * nd + l + st];
  }
  const RdGlobal rdg = {rop, nd, l, z.incj, TWOD ? 0 : z.inck};
  if (d == 0) face_full<KFLU, IFLOW, 0, false, TWOD, HOMO>(z, rdg, xmut, wig, l, lt, qm2, qm1, q0, qp1, f);
  else if (d == 1) face_full<KFLU, IFLOW, 1, false, TWOD, HOMO>(z, rdg, xmut, wig, l, lt, qm2, qm1, q0, qp1, f);
  else if (!TWOD) face_full<KFLU, IFLOW, 2, false, TWOD, HOMO>(z, rdg, xmut, wig, l, lt, qm2, qm1, q0, qp1, f);
  if (side & 1) {
    // the cell past the box takes the flux of the five mean-flow equations only (SA/flu_send_scater_vec1.for gives the
    // 6th to the cell inside alone)
#pragma unroll
    for (int e = 0; e < 5; e++) drodm[e * nd + l] = drodm[e * nd + l] + f[e];
  } else {
#pragma unroll
    for (int e = 0; e < NQ; e++) drodm[e * nd + l - st] = drodm[e * nd + l - st] - f[e];
  }
}

// cells of the layer around the box that are interior cells of the zone: drodm(:,6) += vol * source (no layer along k
// on a two-dimensional zone)
template <bool TWOD, bool HOMO = false>
__global__ void __launch_bounds__(256) k_sa_ring_full(FsZone z, FsBox box, const double *__restrict__ rop,
                                                      const double *__restrict__ dlng, double *__restrict__ drodm)
{
  const int gk = TWOD ? 0 : 1;
  const int i = box.b[0] - 1 + blockIdx.x * blockDim.x + threadIdx.x, j = box.b[2] - 1 + blockIdx.y, k = box.b[4] - gk + blockIdx.z;
  if (i > box.b[1] + 1) return;
  if (i >= box.b[0] && i <= box.b[1] && j >= box.b[2] && j <= box.b[3] && k >= box.b[4] && k <= box.b[5]) return;
  if (i < 1 || i > z.iv || j < 1 || j > z.jv || k < 1 || k > z.kv) return;
  const long long nd = z.nd;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  const long long lt = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck;
  double qc[6];
#pragma unroll
  for (int e = 0; e < 6; e++) qc[e] = rop[e * nd + l];
  drodm[5 * nd + l] = drodm[5 * nd + l] + sa_source_full<TWOD, HOMO>(z, rop, dlng, l, lt, qc);
}

}  // namespace

int fsk_stage_o3sc_full(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                        double *drodm, const double *xmut, unsigned char *wig, const double *wsc, double *flux,
                        const FsBdf *bdf, int keep_balance, hipStream_t s)
{
  if (z.ityp != 0 || z.neq != 5 || !wsc || !flux || !z.vol || (z.kflu == 2 && !wig)) return -1;
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, nullptr};
  if (bdf) b = *bdf;
  const unsigned gx = (unsigned)((z.iv + 1 + 255) / 256);
#define FS_F(K, F, D, FIRST_, T2, HO, g) hipLaunchKernelGGL((k_faces_o3sc_full<K, F, D, FIRST_, T2, HO>), g, dim3(256), 0, s, z, rop, xmut, wig, wsc, flux)
#define FS_FACES(K, F, FIRST_, T2, HO)                                                             \
  do {                                                                                             \
    if (!T2) FS_F(K, F, 2, FIRST_, T2, HO, dim3(gx, (unsigned)z.jv, (unsigned)z.kv + 1));         \
    FS_F(K, F, 1, FIRST_, T2, HO, dim3(gx, (unsigned)z.jv + 1, (unsigned)z.kv));                  \
    FS_F(K, F, 0, FIRST_, T2, HO, dim3(gx, (unsigned)z.jv, (unsigned)z.kv));                      \
  } while (0)
#define FS_GO(K, F, FIRST_)                                                                        \
  do {                                                                                             \
    if (z.twod) FS_FACES(K, F, FIRST_, true, false);                                               \
    else if (z.homo) FS_FACES(K, F, FIRST_, false, true);                                          \
    else FS_FACES(K, F, FIRST_, false, false);                                                     \
  } while (0)
#define FS_CASE(K, F) if (z.kflu == K && z.iflow == F) { if (first) FS_GO(K, F, true); else FS_GO(K, F, false); ok = true; }
  bool ok = false;
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
#undef FS_CASE
#undef FS_GO
#undef FS_FACES
#undef FS_F
  if (!ok) return -1;
  dim3 g((unsigned)((z.iv + 255) / 256), (unsigned)z.jv, (unsigned)z.kv);
#define FS_B(FIRST_, BDF_)                                                                                        \
  do {                                                                                                            \
    if (z.twod) hipLaunchKernelGGL((k_balance_o3sc_full<FIRST_, BDF_, true>), g, dim3(256), 0, s, z, coefH, coefW, keep_balance, rop, rop_out, drodm, flux, b); \
    else hipLaunchKernelGGL((k_balance_o3sc_full<FIRST_, BDF_, false>), g, dim3(256), 0, s, z, coefH, coefW, keep_balance, rop, rop_out, drodm, flux, b);      \
  } while (0)
  if (bdf) { if (first) FS_B(true, true); else FS_B(false, true); }
  else { if (first) FS_B(true, false); else FS_B(false, false); }
#undef FS_B
  return 0;
}

void fsk_box_faces_full(const FsZone &z, const int *box, const double *rop, const double *xmut, unsigned char *wig,
                        double *drodm, hipStream_t s)
{
  FsBox b;
  int m = 1;
  for (int q = 0; q < 6; q++) b.b[q] = box[q];
  for (int d = 0; d < 3; d++) {
    const int a = d == 0 ? 1 : 0, c = d == 2 ? 1 : 2;
    const int n = (box[2 * a + 1] - box[2 * a] + 1) * (box[2 * c + 1] - box[2 * c] + 1);
    if (n > m) m = n;
  }
  dim3 g((unsigned)((m + 127) / 128), 6);
#define FS_GO(K, F)                                                                                                     \
  do {                                                                                                                  \
    if (z.twod) hipLaunchKernelGGL((k_box_faces_full<K, F, true, false>), g, dim3(128), 0, s, z, b, rop, xmut, wig, drodm); \
    else if (z.homo) hipLaunchKernelGGL((k_box_faces_full<K, F, false, true>), g, dim3(128), 0, s, z, b, rop, xmut, wig, drodm); \
    else hipLaunchKernelGGL((k_box_faces_full<K, F, false, false>), g, dim3(128), 0, s, z, b, rop, xmut, wig, drodm);   \
  } while (0)
#define FS_CASE(K, F) if (z.kflu == K && z.iflow == F) { FS_GO(K, F); return; }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2) FS_CASE(1, 3) FS_CASE(2, 3) FS_CASE(5, 3)
#undef FS_CASE
#undef FS_GO
}

void fsk_sa_ring_source_full(const FsZone &z, const int *box, const double *rop, const double *dist, double *drodm,
                             hipStream_t s)
{
  FsBox b;
  for (int q = 0; q < 6; q++) b.b[q] = box[q];
  const int gk = z.twod ? 0 : 1;
  dim3 g((unsigned)((box[1] - box[0] + 3 + 255) / 256), (unsigned)(box[3] - box[2] + 3), (unsigned)(box[5] - box[4] + 1 + 2 * gk));
  if (z.twod) hipLaunchKernelGGL(k_sa_ring_full<true>, g, dim3(256), 0, s, z, b, rop, dist, drodm);
  else if (z.homo) hipLaunchKernelGGL((k_sa_ring_full<false, true>), g, dim3(256), 0, s, z, b, rop, dist, drodm);
  else hipLaunchKernelGGL(k_sa_ring_full<false>, g, dim3(256), 0, s, z, b, rop, dist, drodm);
}

void fsk_sa_coe6_full(const FsZone &z, const double *rop, const double *dist, double *coe, hipStream_t s)
{
  const int gk = z.twod ? 0 : 1;
  dim3 g((z.iv + 2 + 255) / 256, z.jv + 2, z.kv + 2 * gk);
  if (z.twod) hipLaunchKernelGGL(k_sa_coe6_full<true>, g, dim3(256), 0, s, z, rop, dist, coe);
  else if (z.homo) hipLaunchKernelGGL((k_sa_coe6_full<false, true>), g, dim3(256), 0, s, z, rop, dist, coe);
  else hipLaunchKernelGGL(k_sa_coe6_full<false>, g, dim3(256), 0, s, z, rop, dist, coe);
}

// Variant of the kernel for one (scheme, model, stage kind): two-dimensional zones; laminar three-dimensional zones with
// the state planes in LDS (k-homogeneous or general); Euler and Spalart-Allmaras three-dimensional zones (the Euler fluxes
// of a k-homogeneous zone are the general ones).
template <int K, int F, bool FIRST_, bool BDF_>
static void launch_full(const FsZone &z, dim3 g, hipStream_t s, double coefH, double coefW, int kchunk, int keep_balance,
                        const double *rop, double *rop_out, double *drodm, const double *xmut, unsigned char *wig,
                        const FsBdf &b, const double *dist, const int *blist, int nlist)
{
  const int npx = (int)g.x, npy = (int)g.y, npz = (int)g.z;
  if (blist) g = dim3((unsigned)nlist, 1, 1);
#define FS_L(...) hipLaunchKernelGGL((k_stage_full<__VA_ARGS__>), g, dim3(256), 0, s, z, coefH, coefW, kchunk, keep_balance, rop, \
                                     rop_out, drodm, xmut, wig, b, dist, blist, npx, npy, npz)
  if (z.twod) { FS_L(K, F, FIRST_, BDF_, true); return; }
  if constexpr (F == 2) {
    if (z.homo) FS_L(K, F, FIRST_, BDF_, false, true, true);
    else FS_L(K, F, FIRST_, BDF_, false, false, true);
  } else if constexpr (F == 3) {
    if (z.homo) FS_L(K, F, FIRST_, BDF_, false, true);
    else FS_L(K, F, FIRST_, BDF_, false);
  } else {
    FS_L(K, F, FIRST_, BDF_, false);
  }
#undef FS_L
}

FsTiling fsk_stage_full_tiling(const FsZone &z)
{
  FsTiling t;
  t.ti = OI; t.tj = OJ; t.pace = 0;
  t.kchunk = 16;   // 256^3 senseur laminar: 2.18 ms per stage; 8: 2.21, 32: 2.26, 64: 2.47 (workgroups in flight per CU)
  if (const char *ev = getenv("FASTS_HIP_KCHUNK_FULL")) {
    int v = atoi(ev);
    if (v >= 1) t.kchunk = v;
  }
  if (t.kchunk > z.kv) t.kchunk = z.kv;
  t.nti = (z.iv + OI - 1) / OI; t.ntj = (z.jv + OJ - 1) / OJ; t.ntk = (z.kv + t.kchunk - 1) / t.kchunk;
  return t;
}

int fsk_stage_full(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                   double *drodm, const double *xmut, unsigned char *wig, const FsBdf *bdf, const double *dist,
                   int keep_balance, const int *blist, int nlist, hipStream_t s)
{
  if (z.ityp != 0 || !z.vol || !z.mt[0] || !z.mt[1] || !z.mt[2]) return -1;
  if (z.iflow == 3 && (z.neq != 6 || !dist)) return -1;
  if (blist && nlist <= 0) return 0;
  const FsTiling t = fsk_stage_full_tiling(z);
  const int kchunk = t.kchunk;
  dim3 g(t.nti, t.ntj, t.ntk);
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, nullptr};
  if (bdf) b = *bdf;
#define FS_GO(K, F, FIRST_, BDF_) launch_full<K, F, FIRST_, BDF_>(z, g, s, coefH, coefW, kchunk, keep_balance, rop, rop_out, drodm, xmut, wig, b, dist, blist, nlist)
#define FS_CASE(K, F)                                                                                         \
  if (z.kflu == K && z.iflow == F) {                                                                          \
    if (bdf) { if (first) FS_GO(K, F, true, true); else FS_GO(K, F, false, true); }                           \
    else { if (first) FS_GO(K, F, true, false); else FS_GO(K, F, false, false); }                             \
    return 0;                                                                                                 \
  }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
  FS_CASE(1, 3) FS_CASE(2, 3) FS_CASE(5, 3)
#undef FS_CASE
#undef FS_GO
  return -1;
}
