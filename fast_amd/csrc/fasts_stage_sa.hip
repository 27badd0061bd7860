// One Runge-Kutta stage of the Spalart-Allmaras model (6 equations) on a Cartesian zone, gfx950.
// Built with -ffp-contract=off; expressions keep the operand order of the reference fragments
//   total viscosity     Fast/FastS/FastS/Compute/SA/vispalart.for, mulam.for, SA/chi.for, SA/xmut.for
//   source term         SA/spsource_SA.for:185-232, SA/sourceSA_grad_3dcart.for, SA/sourceSA_prod_dest.for
//   rho*nu_tilde flux   AUSM/fluFaceSA.for, SENSOR/fluFaceSA.for, SA/fluvisq_SA_3dcart_{i,j,k}.for, SA/mut_interface.for
//   conductivity        mut_prandtltb_interface.for
//   update              core3ark3.for:196-236 (nu_tilde clipped to [0, ratiom*nu])
// The five mean-flow fluxes are those of the laminar path (fasts_flux.h).
//
// Work decomposition: the one of the curvilinear stage kernel.  32 x 8 thread patch marching in k, each
// thread evaluates the three minus-faces of its cell, i+1/2 by shuffle, j+1/2 through double-buffered
// LDS rows, k+1/2 carried to the next step; 31 x 7 cells owned per patch.  Accumulation order of a cell:
//   drodm*coefH (+ vol*source for the 6th equation) + Fk- + Fj- + Fi- - Fi+ - Fj+ - Fk+
// (src_term.for:59-96 before the flux loops of template_FluxAndBalance.for:192-271).
// The fw function of the destruction term takes a 1/6 power: the device pow is not bit-identical to the
// host one, so this kernel matches the CPU restatement to rounding, not bit for bit.
#include <cstdlib>
#include "fasts_kernels.h"
#include "fasts_flux.h"

namespace {

constexpr int TI = 32, TJ = 8, OI = TI - 1, OJ = TJ - 1, NQ = 6;

template <int DIR> FS_DEV long long inc_of(const FsZone &z) { return DIR == 0 ? 1 : (DIR == 1 ? z.incj : z.inck); }

// Addressing: a cell is (pointer common to the wave) + (byte offset of the lane inside its k-plane, 32 bits, fixed
// while the patch marches in k).  Everything that moves -- variable, plane, neighbour -- goes into the common pointer,
// which lives in scalar registers: a load then needs no vector instruction and no vector register pair for its address.
FS_DEV double ldu(const double *ub, unsigned &vo) { asm("" : "+v"(vo)); return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(ub) + vo); }
FS_DEV void stu(double *ub, unsigned &vo, double v) { asm("" : "+v"(vo)); *reinterpret_cast<double *>(reinterpret_cast<char *>(ub) + vo) = v; }

// Face between cell l - inc(DIR) and cell l; qm2..qp1 = the four cells of the MUSCL stencil (6 variables).
template <int KFLU, int DIR, bool FIRST>
FS_DEV void face_sa(const FsZone &z, const double *__restrict__ rop, const double *__restrict__ xmut,
                    unsigned char *__restrict__ wig, unsigned &vo, const double *qm2, const double *qm1,
                    const double *q0, const double *qp1, double *flu)
{
  const long long nd = z.nd;
  double wg = 0.;
  if (KFLU == 2) {
    if (FIRST) {
      auto qs = [&](int e, int o) -> double { return o == -2 ? qm2[e] : (o == -1 ? qm1[e] : (o == 0 ? q0[e] : qp1[e])); };
      wg = fs_wiggle(qs);
      (wig + DIR * nd)[vo >> 3] = wg != 0. ? 1 : 0;
    } else {
      wg = (double)(wig + DIR * nd)[vo >> 3];
    }
  }
  double sa[4];
  fs_flux_inviscid<KFLU, DIR>(z, qm2, qm1, q0, qp1, wg, flu, KFLU == 5 ? nullptr : sa);
  if (KFLU == 5) {
    // roe: first-order upwinding with the Roe-averaged normal velocity (ROE/fluFaceSA.for,
    // ROE/3dcart/fluFaceSA_o3_3dcart_{i,j,k}.for)
    constexpr int VN = 1 + DIR;
    const double tc = DIR == 0 ? z.tcx : (DIR == 1 ? z.tcy : z.tcz);
    const double s = DIR == 0 ? z.si : (DIR == 1 ? z.sj : z.sk);
    double qp1r = qm1[0], qm1r = q0[0];
    const double qn1 = qm1[VN] * tc, qn2 = q0[VN] * tc;
    const double qm6 = q0[5] * qm1r * qn2, qp6 = qm1[5] * qp1r * qn1;
    qp1r = sqrt(qp1r); qm1r = sqrt(qm1r);
    const double r_1 = 1. / (qp1r + qm1r);
    const double vroe = r_1 * (qp1r * qm1[VN] + qm1r * q0[VN]);
    const double s_1 = 1. / fs_max(s, 1.e-30);
    const double tdu = fs_max(z.c1 / s_1, fabs(vroe * tc));
    flu[5] = 0.5 * (qm6 + qp6 - tdu * (q0[5] * q0[0] - qm1[5] * qm1[0]));
  } else {   // convective flux of rho*nu_tilde
    const double qm = z.c4 * q0[5] + z.c5 * qm1[5] + z.c6 * qp1[5];
    const double qp = z.c4 * qm1[5] + z.c6 * qm2[5] + z.c5 * q0[5];
    const double qm6 = qm * sa[2], qp6 = qp * sa[3];
    flu[5] = sa[0] * (qm6 + qp6) - sa[1] * (qm6 - qp6);
  }
  const long long inc = inc_of<DIR>(z);
  const long long ia = DIR == 0 ? z.incj : 1;
  const long long ib = DIR == 2 ? z.incj : z.inck;
  auto q = [&](int e, int da, int db, int dn) -> double {
    if (da == 0 && db == 0) return dn == 0 ? q0[e] : qm1[e];
    return ldu(rop + (e * nd + dn * inc + da * ia + db * ib), vo);
  };
  const double Til = qm1[4], Tir = q0[4];
  double xmulam = fs_mulam_sa(z, Til);
  xmulam = xmulam + fs_mulam_sa(z, Tir);
  const double xmutot = ldu(xmut, vo) + ldu(xmut - inc, vo);
  const double xmutur = xmutot - xmulam;
  const double xktvol = (xmulam * z.gam3 + xmutur * z.gam4) * z.volinv;
  const double xmutvol = xmutot * z.volinv;
  fs_flux_viscous_x<DIR>(z, q, xmutvol, xktvol, flu);
  {   // diffusion of nu_tilde
    const double tc = DIR == 0 ? z.tcx : (DIR == 1 ? z.tcy : z.tcz);
    const double g = (q0[5] - qm1[5]) * tc;
    double xm = xmulam + qm1[5] * qm1[0] + q0[5] * q0[0];
    xm = xm * z.volinv * (1. / (2. / 3.));
    if (DIR == 1) flu[5] = flu[5] - xm * (g * tc);
    else flu[5] = flu[5] - xm * g * tc;
  }
}

// vol * (production - destruction + cb2 term) of cell l; qc = state of the cell
// tsb (may be null) receives production - destruction, amu the clipped rho*nu_tilde (SA/sourceSA_LU.for)
// rop, dlng: pointers of the cell's k-plane; vo: byte offset of the cell in it
FS_DEV double sa_source(const FsZone &z, const double *__restrict__ rop, const double *__restrict__ dlng, unsigned &vo,
                        const double *qc, double *tsb = nullptr, double *amu = nullptr)
{
  const double SA_CKARM = 0.41, SA_CB1 = 0.1355, SA_CB2 = 0.622, SA_CV1 = 357.911, SA_CW2 = 0.3, SA_CW3 = 64.;
  const double SA_CW4 = 0.21, SA_CW5 = 1.5, SA_CROT = 2.0, SA_SIGMA = (2. / 3.);
  const double c1 = SA_CB2 / SA_SIGMA;
  const double cw1 = (SA_CB1 / SA_CKARM / SA_CKARM) + (1. + SA_CB2) / SA_SIGMA;
  const long long nd = z.nd, incj = z.incj, inck = z.inck;
  const long long l1 = 1, l2 = -1, l3 = incj, l4 = -incj, l5 = inck, l6 = -inck;
  const double tix = z.tcx, tjy = z.tcy, tkz = z.tcz;
  const double xvol = z.volinv;   // 0.5/vol
  double g[3][3];
#pragma unroll
  for (int v = 0; v < 3; v++) {
    const double *__restrict__ q = rop + (1 + v) * nd;
    const double c = qc[1 + v];
    const double u1 = (c + ldu(q + l2, vo)), u2 = (c + ldu(q + l3, vo)), u3 = (c + ldu(q + l1, vo));
    const double u4 = (c + ldu(q + l4, vo)), u5 = (c + ldu(q + l5, vo)), u6 = (c + ldu(q + l6, vo));
    g[v][0] = (u3 - u1) * tix; g[v][1] = (u2 - u4) * tjy; g[v][2] = (u5 - u6) * tkz;
  }
  const double rotx = g[2][1] - g[1][2], roty = g[0][2] - g[2][0], rotz = g[1][0] - g[0][1];
  const double rot = sqrt(rotx * rotx + roty * roty + rotz * rotz) * xvol;
  const double S11 = g[0][0], S22 = g[1][1], S33 = g[2][2];
  const double S12 = 0.5 * (g[0][1] + g[1][0]), S13 = 0.5 * (g[0][2] + g[2][0]), S23 = 0.5 * (g[1][2] + g[2][1]);
  double St = S11 * S11 + S22 * S22 + S33 * S33 + 2 * (S12 * S12) + 2 * (S13 * S13) + 2 * (S23 * S23);
  St = sqrt(2 * St) * xvol;
  const double *__restrict__ N = rop + 5 * nd, *__restrict__ R = rop;
  const double nc = qc[5], rc = qc[0];
  const double n1 = ldu(N + l1, vo), n2 = ldu(N + l2, vo), n3 = ldu(N + l3, vo), n4 = ldu(N + l4, vo), n5 = ldu(N + l5, vo), n6 = ldu(N + l6, vo);
  double u1 = (nc + n2), u2 = (nc + n3), u3 = (nc + n1), u4 = (nc + n4), u5 = (nc + n5), u6 = (nc + n6);
  double dudx = (u3 - u1) * tix, dudy = (u2 - u4) * tjy, dudz = (u5 - u6) * tkz;
  u1 = nc * rc + n2 * ldu(R + l2, vo); u2 = nc * rc + n3 * ldu(R + l3, vo); u3 = nc * rc + n1 * ldu(R + l1, vo);
  u4 = nc * rc + n4 * ldu(R + l4, vo); u5 = nc * rc + n5 * ldu(R + l5, vo); u6 = nc * rc + n6 * ldu(R + l6, vo);
  dudx = dudx * (u3 - u1) * tix; dudy = dudy * (u2 - u4) * tjy; dudz = dudz * (u5 - u6) * tkz;
  const double anvisc = c1 * (dudx + dudy + dudz) * xvol * xvol;
  const double amulam = fs_mulam_sa(z, qc[4]);
  const double anulam = rc / amulam;
  const double anutild = fs_max(nc, 1.e-27);
  const double chi = fs_max(anutild * anulam, 1.e-7);
  const double f1 = 1. / (1. + SA_CV1 / (chi * chi * chi));
  const double f2 = 1. - chi / (1. + chi * f1);
  const double dist = fs_max(ldu(dlng, vo), 1.e-27);
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
  return z.vol0 * tsource;
}

// BDF = false: Runge-Kutta stage.  BDF = true: Newton residual of the implicit scheme, 6 equations
// (core3as2.for:49-231; rop_out unused).
template <int KFLU, bool FIRST, bool BDF>
__global__ void __launch_bounds__(256)
k_stage_sa(FsZone z, double coefH, double coefW, int kchunk, int keep, const double *__restrict__ rop,
           double *__restrict__ rop_out, double *__restrict__ drodm, const double *__restrict__ xmut,
           const double *__restrict__ dlng, unsigned char *__restrict__ wig, FsBdf bdf)
{
  __shared__ double sfj[2][NQ][TJ][TI];
  const int ti = threadIdx.x & 31, tj = threadIdx.x >> 5;
  const int i = 1 + blockIdx.x * OI + ti, j = 1 + blockIdx.y * OJ + tj;
  const int k0 = 1 + blockIdx.z * kchunk;
  const int k1 = k0 + kchunk - 1 < z.kv ? k0 + kchunk - 1 : z.kv;
  const bool active = i <= z.iv + 1 && j <= z.jv + 1;
  const bool own_i = ti < OI && i <= z.iv, own_j = tj < OJ && j <= z.jv;
  const bool own = own_i && own_j;
  const bool do_fi = active && own_j, do_fj = active && own_i;
  const long long nd = z.nd;
  const int ic = active ? i : 1, jc = active ? j : 1;
  // byte offset of the lane's cell inside a k-plane (fixed), first cell of the current plane (common to the workgroup)
  unsigned vo = (unsigned)(((long long)(ic + z.ific - 1) + (long long)(jc + z.ific - 1) * z.incj) * 8);
  long long lk = (long long)(k0 + z.ifick - 1) * z.inck;
  double qa[NQ], qb[NQ], qc[NQ], qd[NQ];
#pragma unroll
  for (int e = 0; e < NQ; e++) {
    qa[e] = ldu(rop + (e * nd + lk - 2 * z.inck), vo);
    qb[e] = ldu(rop + (e * nd + lk - z.inck), vo);
    qc[e] = ldu(rop + (e * nd + lk), vo);
    qd[e] = 0.;
  }
  double part[NQ] = {0., 0., 0., 0., 0., 0.};
  for (int k = k0; k <= k1 + 1; k++) {
    const bool last = k > k1;
    const double *__restrict__ rk = rop + lk;
    const double *__restrict__ xk = xmut + lk;
    unsigned char *__restrict__ wk = wig + lk;
#pragma unroll
    for (int e = 0; e < NQ; e++) qd[e] = ldu(rk + (e * nd + z.inck), vo);
    double fk[NQ] = {0., 0., 0., 0., 0., 0.};
    if (own) {
      face_sa<KFLU, 2, FIRST>(z, rk, xk, wk, vo, qa, qb, qc, qd, fk);
      if (k > k0) {
        const long long lb = lk - z.inck;   // plane of the cell that is finished here
        double d[NQ], o[NQ];
#pragma unroll
        for (int e = 0; e < NQ; e++) d[e] = part[e] - fk[e];
        if (BDF) {
          double qn[NQ] = {0., 0., 0., 0., 0., 0.}, qn1[NQ];
#pragma unroll
          for (int e = 0; e < NQ; e++) qn1[e] = ldu(bdf.ro_n1 + (e * nd + lb), vo);
          if (!bdf.nit1) {
#pragma unroll
            for (int e = 0; e < NQ; e++) qn[e] = ldu(bdf.ro_n + (e * nd + lb), vo);
          }
          const double coe = bdf.coe1 ? ldu(bdf.coe1 + lb, vo) : z.coe;
          if (bdf.nit1 && bdf.coe1) {   // local time step, first sub-iteration: no BDF terms (core3as2.for:57-66)
#pragma unroll
            for (int e = 0; e < NQ; e++) o[e] = coe * d[e];
          } else {
            fs_bdf_residual(z, bdf, coe, qb, qn, qn1, d, o);
            if (bdf.nit1) o[5] = coe * d[5] + (bdf.c1 + bdf.c2) * qb[5] * qb[0] + bdf.c3 * qn1[5] * qn1[0];
            else o[5] = coe * d[5] + bdf.c1 * qb[5] * qb[0] + bdf.c2 * qn[5] * qn[0] + bdf.c3 * qn1[5] * qn1[0];
          }
          if (fs_in_box(bdf, i, j, k - 1)) {   // a sub-zone: the other cells keep their residual
#pragma unroll
            for (int e = 0; e < NQ; e++) stu(drodm + (e * nd + lb), vo, o[e]);
          }
        } else {
        if (keep) {   // the balance of the last Runge-Kutta stage of a step is read by nothing
#pragma unroll
          for (int e = 0; e < NQ; e++) stu(drodm + (e * nd + lb), vo, d[e]);
        }
        fs_rk_update(z, coefW, qb, d, o);
#pragma unroll
        for (int e = 0; e < 5; e++) stu(rop_out + (e * nd + lb), vo, o[e]);
        {   // nu_tilde: conservative update, clipped to [0, ratiom * nu_laminar] (core3ark3.for:196-236)
          const double r_1 = 1. / o[0];
          const double tab6 = d[5] * (coefW * z.coe);
          double anulam = fs_mulam_sa(z, o[4]);
          anulam = anulam * r_1 * z.ratiom;
          double t6 = (qb[0] * qb[5] + tab6) * r_1;
          t6 = fs_min(t6, anulam);
          stu(rop_out + (5 * nd + lb), vo, fs_max(t6, 0.));
        }
        }
      }
    }
    if (!last) {
      double fim[NQ] = {0., 0., 0., 0., 0., 0.}, fjm[NQ] = {0., 0., 0., 0., 0., 0.};
      if (do_fj) {
        double m2[NQ], m1[NQ], p1[NQ];
#pragma unroll
        for (int e = 0; e < NQ; e++) {
          m2[e] = ldu(rk + (e * nd - 2 * z.incj), vo); m1[e] = ldu(rk + (e * nd - z.incj), vo); p1[e] = ldu(rk + (e * nd + z.incj), vo);
        }
        face_sa<KFLU, 1, FIRST>(z, rk, xk, wk, vo, m2, m1, qc, p1, fjm);
      }
      const int buf = (k - k0) & 1;
#pragma unroll
      for (int e = 0; e < NQ; e++) sfj[buf][e][tj][ti] = fjm[e];
      if (do_fi) {
        double m2[NQ], m1[NQ], p1[NQ];
#pragma unroll
        for (int e = 0; e < NQ; e++) {
          m2[e] = ldu(rk + (e * nd - 2), vo); m1[e] = ldu(rk + (e * nd - 1), vo); p1[e] = ldu(rk + (e * nd + 1), vo);
        }
        face_sa<KFLU, 0, FIRST>(z, rk, xk, wk, vo, m2, m1, qc, p1, fim);
      }
      double fip[NQ];
#pragma unroll
      for (int e = 0; e < NQ; e++) fip[e] = __shfl_down(fim[e], 1);
      __syncthreads();
      if (own) {
        const double src = sa_source(z, rk, dlng + lk, vo, qc);
#pragma unroll
        for (int e = 0; e < NQ; e++) {
          double dr = (FIRST || BDF) ? 0. : ldu(drodm + (e * nd + lk), vo) * coefH;
          if (e == 5) dr = dr + src;
          part[e] = dr + fk[e] + fjm[e] + fim[e] - fip[e] - sfj[buf][e][tj + 1][ti];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < NQ; e++) { qa[e] = qb[e]; qb[e] = qc[e]; qc[e] = qd[e]; }
    lk += z.inck;
  }
}

// total viscosity mu + mu_t on the whole zone, ghost cells included
__global__ void __launch_bounds__(256) k_vispalart(FsZone z, const double *__restrict__ rop, double *__restrict__ xmut)
{
  const int si = blockIdx.x * blockDim.x + threadIdx.x, sj = blockIdx.y, sk = blockIdx.z;
  if (si >= z.ni) return;
  const long long nd = z.nd, l = si + (long long)sj * z.incj + (long long)sk * z.inck;
  const double SA_CV1 = 357.911;
  const double amulam = fs_mulam_sa(z, rop[l + 4 * nd]);
  const double anulam = rop[l] / amulam;
  const double anutild = fs_max(rop[l + 5 * nd], 1.e-27);
  const double chi = fs_max(anutild * anulam, 1.e-7);
  const double fvv1 = 1. / (1. + SA_CV1 / (chi * chi * chi));
  const double amut = rop[l] * anutild * fvv1;
  const double xmuprov = fs_max(amut, 0.);
  xmut[l] = amulam + fs_min(xmuprov, z.ratiom * amulam);
}

// Diagonal of the nu_tilde equation of the implicit scheme, coe(:,6) = coe(:,5) - min(vol*(P-D)/(rho nu_tilde), 0)*coe(:,1)
// (SA/sourceSA_LU.for), on the interior plus one ghost layer (ind_ssa of indice_boucle_ssdom.for with one cache
// block): the LU range reaches the first ghost layer on wall faces.
__global__ void __launch_bounds__(256) k_sa_coe6(FsZone z, const double *__restrict__ rop, const double *__restrict__ dlng,
                                                 double *__restrict__ coe)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, k = blockIdx.z;   // 0 .. iv+1
  if (i > z.iv + 1) return;
  const long long nd = z.nd;
  const long long lk = (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;   // common to the workgroup
  unsigned vo = (unsigned)(i + z.ific - 1) * 8u;
  const long long l = lk + (i + z.ific - 1);
  double qc[NQ];
#pragma unroll
  for (int e = 0; e < NQ; e++) qc[e] = rop[e * nd + l];
  double tsb, amu;
  sa_source(z, rop + lk, dlng + lk, vo, qc, &tsb, &amu);
  const double tsourcenu = z.vol0 * tsb / amu;
  coe[l + 5 * nd] = coe[l + 4 * nd] - fs_min(tsourcenu, 0.) * coe[l];
}

template <int KFLU>
__global__ void __launch_bounds__(128) k_box_faces_sa(FsZone z, FsBox box, const double *__restrict__ rop,
                                                      const double *__restrict__ xmut, unsigned char *__restrict__ wig,
                                                      double *__restrict__ drodm)
{
  const int side = blockIdx.y;
  int i, j, k;
  if (!box_face_cell(z, box, side, (long long)blockIdx.x * blockDim.x + threadIdx.x, i, j, k)) return;
  const long long nd = z.nd;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  const int d = side >> 1;
  const long long st = d == 0 ? 1 : (d == 1 ? z.incj : z.inck);
  double qm2[NQ], qm1[NQ], q0[NQ], qp1[NQ], f[NQ];
#pragma unroll
  for (int e = 0; e < NQ; e++) {
    qm2[e] = rop[e * nd + l - 2 * st]; qm1[e] = rop[e * nd + l - st]; q0[e] = rop[e * nd + l]; qp1[e] = rop[e * nd + l + st];
  }
  // (the k-plane differs from lane to lane here: its pointer is no common one, the loads take vector addresses)
  const long long lk = (long long)(k + z.ifick - 1) * z.inck;
  unsigned vo = (unsigned)((l - lk) * 8);
  if (d == 0) face_sa<KFLU, 0, false>(z, rop + lk, xmut + lk, wig + lk, vo, qm2, qm1, q0, qp1, f);
  else if (d == 1) face_sa<KFLU, 1, false>(z, rop + lk, xmut + lk, wig + lk, vo, qm2, qm1, q0, qp1, f);
  else face_sa<KFLU, 2, false>(z, rop + lk, xmut + lk, wig + lk, vo, qm2, qm1, q0, qp1, f);
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

// cells of the layer around the box that are interior cells of the zone: drodm(:,6) += vol * source
__global__ void __launch_bounds__(256) k_sa_ring(FsZone z, FsBox box, const double *__restrict__ rop,
                                                 const double *__restrict__ dlng, double *__restrict__ drodm)
{
  const int i = box.b[0] - 1 + blockIdx.x * blockDim.x + threadIdx.x, j = box.b[2] - 1 + blockIdx.y, k = box.b[4] - 1 + blockIdx.z;
  if (i > box.b[1] + 1) return;
  if (i >= box.b[0] && i <= box.b[1] && j >= box.b[2] && j <= box.b[3] && k >= box.b[4] && k <= box.b[5]) return;
  if (i < 1 || i > z.iv || j < 1 || j > z.jv || k < 1 || k > z.kv) return;
  const long long nd = z.nd;
  const long long l = (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
  double qc[NQ];
#pragma unroll
  for (int e = 0; e < NQ; e++) qc[e] = rop[e * nd + l];
  const long long lk = (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;   // common to the workgroup
  unsigned vo = (unsigned)(i + z.ific - 1) * 8u;
  drodm[5 * nd + l] = drodm[5 * nd + l] + sa_source(z, rop + lk, dlng + lk, vo, qc);
}

}  // namespace

void fsk_box_faces_sa(const FsZone &z, const int *box, const double *rop, const double *xmut, unsigned char *wig,
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
  if (z.kflu == 1) hipLaunchKernelGGL(k_box_faces_sa<1>, g, dim3(128), 0, s, z, b, rop, xmut, wig, drodm);
  else if (z.kflu == 2) hipLaunchKernelGGL(k_box_faces_sa<2>, g, dim3(128), 0, s, z, b, rop, xmut, wig, drodm);
  else hipLaunchKernelGGL(k_box_faces_sa<5>, g, dim3(128), 0, s, z, b, rop, xmut, wig, drodm);
}

void fsk_sa_ring_source(const FsZone &z, const int *box, const double *rop, const double *dist, double *drodm, hipStream_t s)
{
  FsBox b;
  for (int q = 0; q < 6; q++) b.b[q] = box[q];
  dim3 g((unsigned)((box[1] - box[0] + 3 + 255) / 256), (unsigned)(box[3] - box[2] + 3), (unsigned)(box[5] - box[4] + 3));
  hipLaunchKernelGGL(k_sa_ring, g, dim3(256), 0, s, z, b, rop, dist, drodm);
}

void fsk_sa_coe6(const FsZone &z, const double *rop, const double *dist, double *coe, hipStream_t s)
{
  dim3 g((z.iv + 2 + 255) / 256, z.jv + 2, z.kv + 2);
  hipLaunchKernelGGL(k_sa_coe6, g, dim3(256), 0, s, z, rop, dist, coe);
}

void fsk_vispalart(const FsZone &z, const double *rop, double *xmut, hipStream_t s)
{
  dim3 g((z.ni + 255) / 256, z.nj, z.nk);
  hipLaunchKernelGGL(k_vispalart, g, dim3(256), 0, s, z, rop, xmut);
}

int fsk_stage_sa(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                 double *drodm, const double *xmut, const double *dist, unsigned char *wig, const FsBdf *bdf,
                 int keep_balance, hipStream_t s)
{
  if (z.ityp != 2 || z.neq != 6 || z.iflow != 3 || !dist) return -1;
  if ((long long)z.inck * 8 >= (1LL << 32)) return -1;   // the kernels address a cell by a 32-bit byte offset inside its k-plane
  int kchunk = 16;   // 256^3, ms per stage: 8 planes 2.16, 16 2.14, 32 2.24, 64 2.44, 128 2.69
  if (const char *ev = getenv("FASTS_HIP_KCHUNK_SA")) { const int v = atoi(ev); if (v >= 1) kchunk = v; }
  if (kchunk > z.kv) kchunk = z.kv;
  dim3 g((z.iv + OI - 1) / OI, (z.jv + OJ - 1) / OJ, (z.kv + kchunk - 1) / kchunk);
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, nullptr};
  if (bdf) b = *bdf;
#define FS_GO(K, FIRST_, BDF_)                                                                                  \
  hipLaunchKernelGGL((k_stage_sa<K, FIRST_, BDF_>), g, dim3(256), 0, s, z, coefH, coefW, kchunk, keep_balance, rop, rop_out, drodm, \
                     xmut, dist, wig, b)
#define FS_CASE(K)                                                                                              \
  if (z.kflu == K) {                                                                                            \
    if (bdf) { if (first) FS_GO(K, true, true); else FS_GO(K, false, true); }                                   \
    else { if (first) FS_GO(K, true, false); else FS_GO(K, false, false); }                                     \
    return 0;                                                                                                   \
  }
  FS_CASE(1) FS_CASE(2) FS_CASE(5)
#undef FS_CASE
#undef FS_GO
  return -1;
}
