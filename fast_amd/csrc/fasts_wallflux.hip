// Wall-face flux correction of the roe scheme and the deferred end of a stage, gfx950.
// Built with -ffp-contract=off; expressions keep the operand order of the reference fragments
//   Fast/FastS/FastS/BC/bfl3.for, CL_correction_flu.for           which windows, in which order
//   Compute/ROE/<type>/corr_fluroe_euler_o3_<type>.for            the Euler roe flux of the wall face, taken back
//   BC/bflwall.for, BC/bflwallslip.for                            the wall flux put in its place
//   Compute/core3ark3.for, core3as2.for                           update / Newton residual of the corrected cells
//
// With roe, the reference first assembles the balance of every cell, then corrects the cells next to a wall,
// then updates.  The stage kernels fuse balance and update, so a zone that needs the correction runs them in
// Runge-Kutta form with the balance kept in drodm, k_roe_wall corrects drodm on the rows next to the walls
// (one launch per window, in the order of the BC table: a corner cell belongs to two windows), and k_finish
// redoes the end of the stage from the corrected balance (explicit scheme: on those rows only; implicit
// scheme: the Newton residual of every cell).
#include "fasts_kernels.h"
#include "fasts_flux.h"

namespace {

FS_DEV long long wcell(const FsZone &z, int i, int j, int k)
{
  return (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
}
FS_DEV long long wmtr(const FsZone &z, int i, int j, int k)
{
  return (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck;
}

// row = cells whose min face along d is the wall face: the first interior cells on a min side, the first
// ghost cells on a max side (bfl3.for:79-93).  slip: tag 3 (bflwallslip), else tags 4 / 6 (bflwall).
struct FsWallRow { int d, maxside, slip, lo[3], hi[3]; };

template <bool FULL>
__global__ void __launch_bounds__(128) k_roe_wall(FsZone z, FsWallRow w, const double *__restrict__ rop,
                                                  double *__restrict__ drodm)
{
  const int n0 = w.hi[0] - w.lo[0] + 1, n1 = w.hi[1] - w.lo[1] + 1, n2 = w.hi[2] - w.lo[2] + 1;
  const long long n = (long long)n0 * n1 * n2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = w.lo[0] + (int)(t % n0), j = w.lo[1] + (int)((t / n0) % n1), k = w.lo[2] + (int)(t / ((long long)n0 * n1));
  const int d = w.d;
  const long long nd = z.nd, inc = d == 0 ? 1 : (d == 1 ? z.incj : z.inck);
  const long long l = wcell(z, i, j, k), l0 = w.maxside ? l - inc : l, iadrf = l - inc;
  double tc[3] = {0., 0., 0.};
  if (FULL) {
    const long long lt = wmtr(z, i, j, k);
    tc[0] = z.mt[d][lt]; tc[1] = z.mt[d][lt + z.ndm]; tc[2] = z.mt[d][lt + 2 * z.ndm];
  } else {
    tc[d] = d == 0 ? z.tcx : (d == 1 ? z.tcy : z.tcz);
  }
  // 1. the Euler roe flux of the wall face (MUSCL o3 from the ghost cells), taken back
  {
    double qm2[5], qm1[5], q0[5], qp1[5], f[5];
#pragma unroll
    for (int e = 0; e < 5; e++) {
      qm2[e] = rop[e * nd + l - 2 * inc]; qm1[e] = rop[e * nd + l - inc];
      q0[e] = rop[e * nd + l]; qp1[e] = rop[e * nd + l + inc];
    }
    if (FULL) fs_flux_inviscid_full<5>(z, tc[0], tc[1], tc[2], qm2, qm1, q0, qp1, 0., f, z.twod && d == 1);
    else if (d == 0) fs_flux_inviscid<5, 0>(z, qm2, qm1, q0, qp1, 0., f);
    else if (d == 1) fs_flux_inviscid<5, 1>(z, qm2, qm1, q0, qp1, 0., f);
    else fs_flux_inviscid<5, 2>(z, qm2, qm1, q0, qp1, 0., f);
    const double sens = w.maxside ? 1. : -1.;
#pragma unroll
    for (int e = 0; e < 5; e++) drodm[e * nd + l0] = drodm[e * nd + l0] + f[e] * sens;
  }
  // 2. wall flux (fixed mesh: qen = 0)
  {
    const double sens = w.maxside ? -1. : 1.;
    const double tcx = FULL ? tc[0] * 1. : tc[0], tcy = FULL ? tc[1] * 1. : tc[1], tcz = FULL ? tc[2] * 1. : tc[2];
    const double qen = (0. * tcx + 0. * tcy + 0. * tcz * 0.) * 0.;
    double r = 0.5 * (rop[l] + rop[iadrf]);
    double u = 0.5 * (rop[l + nd] + rop[iadrf + nd]);
    double v = 0.5 * (rop[l + 2 * nd] + rop[iadrf + 2 * nd]);
    double ww = 0.5 * (rop[l + 3 * nd] + rop[iadrf + 3 * nd]);
    const double u_int = tcx * u + tcy * v + tcz * ww - qen;
    double fl[6] = {0., 0., 0., 0., 0., 0.};
    if (!w.slip) {
      const double pp = 0.5 * (rop[l + 4 * nd] * rop[l] + rop[iadrf + 4 * nd] * rop[iadrf]) * z.rgp;
      fl[1] = tcx * pp + u_int * u * r;
      fl[2] = tcy * pp + u_int * v * r;
      fl[3] = tcz * pp + u_int * ww * r;
      fl[4] = pp * qen;
    } else {
      if (z.neq == 6) { r = rop[iadrf]; u = rop[iadrf + nd]; v = rop[iadrf + 2 * nd]; ww = rop[iadrf + 3 * nd]; }
      double pp = rop[iadrf + 4 * nd] * rop[iadrf] * z.rgp;
      const double h = z.gam1 * pp + r * (u * u + v * v + ww * ww) * 0.5;
      pp = 0.5 * (rop[l + 4 * nd] * rop[l] + rop[iadrf + 4 * nd] * rop[iadrf]) * z.rgp;
      fl[0] = u_int * r;
      fl[1] = tcx * pp + u_int * r * u;
      fl[2] = tcy * pp + u_int * r * v;
      fl[3] = tcz * pp + u_int * r * ww;
      fl[4] = u_int * h + pp * qen;
      if (z.neq == 6) fl[5] = u_int * r * rop[iadrf + 5 * nd];
    }
    for (int e = 0; e < z.neq; e++) drodm[e * nd + l0] = drodm[e * nd + l0] + fl[e] * sens;
  }
}

// End of a stage for the cells of a box: Runge-Kutta update (BDF = false: rop_out from rop_in and drodm) or
// Newton residual (BDF = true: drodm <- coe*drodm + BDF terms, rop_in = current iterate).
template <bool BDF>
__global__ void __launch_bounds__(128) k_finish(FsZone z, FsWallRow w, double coefW, const double *__restrict__ rop_in,
                                                double *__restrict__ rop_out, double *__restrict__ drodm, FsBdf bdf)
{
  const int n0 = w.hi[0] - w.lo[0] + 1, n1 = w.hi[1] - w.lo[1] + 1, n2 = w.hi[2] - w.lo[2] + 1;
  const long long n = (long long)n0 * n1 * n2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = w.lo[0] + (int)(t % n0), j = w.lo[1] + (int)((t / n0) % n1), k = w.lo[2] + (int)(t / ((long long)n0 * n1));
  const long long nd = z.nd, l = wcell(z, i, j, k);
  const int neq = z.neq;
  double q[6] = {0., 0., 0., 0., 0., 0.}, d[6] = {0., 0., 0., 0., 0., 0.}, o[6];
  for (int e = 0; e < neq; e++) { q[e] = rop_in[e * nd + l]; d[e] = drodm[e * nd + l]; }
  double coe = z.ityp == 0 ? z.dt / z.vol[wmtr(z, i, j, k)] : z.coe;
  if (!BDF && bdf.coe1) coe = bdf.coe1[l];   // explicit scheme with a local time step: coe(:,1) of the cell
  if (BDF) {
    double qn[6] = {0., 0., 0., 0., 0., 0.}, qn1[6] = {0., 0., 0., 0., 0., 0.};
    for (int e = 0; e < neq; e++) qn1[e] = bdf.ro_n1[e * nd + l];
    if (!bdf.nit1) for (int e = 0; e < neq; e++) qn[e] = bdf.ro_n[e * nd + l];
    const double coeb = bdf.coe1 ? bdf.coe1[l] : coe;
    if (neq == 6 && bdf.nit1 && bdf.coe1) {
      for (int e = 0; e < 6; e++) o[e] = coeb * d[e];
    } else {
      fs_bdf_residual(z, bdf, coeb, q, qn, qn1, d, o);
      if (neq == 6) {
        if (bdf.nit1) o[5] = coeb * d[5] + (bdf.c1 + bdf.c2) * q[5] * q[0] + bdf.c3 * qn1[5] * qn1[0];
        else o[5] = coeb * d[5] + bdf.c1 * q[5] * q[0] + bdf.c2 * qn[5] * qn[0] + bdf.c3 * qn1[5] * qn1[0];
      }
    }
    for (int e = 0; e < neq; e++) drodm[e * nd + l] = o[e];
  } else {
    fs_rk_update_c(z, coefW, coe, q, d, o);
    if (z.twod) o[3] = 0.;
    if (neq == 6) {
      const double r_1 = 1. / o[0];
      const double tab6 = d[5] * (coefW * coe);
      double anulam = fs_mulam_sa(z, o[4]);
      anulam = anulam * r_1 * z.ratiom;
      double t6 = (q[0] * q[5] + tab6) * r_1;
      t6 = fs_min(t6, anulam);
      o[5] = fs_max(t6, 0.);
    }
    for (int e = 0; e < neq; e++) rop_out[e * nd + l] = o[e];
  }
}

FsWallRow make_row(const FsZone &z, int idir, const int *glo, const int *ghi, int slip)
{
  FsWallRow w;
  w.d = (idir - 1) / 2; w.maxside = idir % 2 == 0; w.slip = slip;
  for (int a = 0; a < 3; a++) { w.lo[a] = glo[a]; w.hi[a] = ghi[a]; }
  const int ijkv[3] = {z.iv, z.jv, z.kv};
  w.lo[w.d] = w.hi[w.d] = w.maxside ? ijkv[w.d] + 1 : 1;
  return w;
}

long long row_cells(const FsWallRow &w)
{
  return (long long)(w.hi[0] - w.lo[0] + 1) * (w.hi[1] - w.lo[1] + 1) * (w.hi[2] - w.lo[2] + 1);
}

}  // namespace

// One wall window: glo/ghi = its ghost-cell window (interior-based indices), tag = BC tag of the reference.
// ---- faces of a sub-zone box: the flux of a face between cell l - inc(DIR) and cell l of a Cartesian zone, operands
// from memory (the stage kernel takes them from its tile) ------------------------------------------------------------
namespace {

// csc >= 0: weight of the o3 states (o3sc slope).  wg_first != NULL: first sub-iteration of the senseur scheme, the
// sensor of the face is evaluated here and returned instead of read.
template <int KFLU, int IFLOW, int DIR>
FS_DEV void face_cart(const FsZone &z, const double *__restrict__ rop, const double *__restrict__ xmut,
                      const unsigned char *__restrict__ wig, long long l, double *flu, double csc = -1.,
                      double *wg_first = nullptr)
{
  const long long nd = z.nd;
  const long long inc[3] = {1, z.incj, z.inck};
  const long long st = inc[DIR];
  double qm2[5], qm1[5], q0[5], qp1[5];
#pragma unroll
  for (int e = 0; e < 5; e++) {
    qm2[e] = rop[e * nd + l - 2 * st]; qm1[e] = rop[e * nd + l - st]; q0[e] = rop[e * nd + l]; qp1[e] = rop[e * nd + l + st];
  }
  // sensor of the face: bit DIR of the byte of cell l (written at the first sub-iteration, which covers the zone)
  double wg = 0.;
  if (KFLU == 2) {
    if (wg_first) {
      auto qs = [&](int e, int o) -> double { return o == -2 ? qm2[e] : (o == -1 ? qm1[e] : (o == 0 ? q0[e] : qp1[e])); };
      wg = fs_wiggle(qs);
      *wg_first = wg;
    } else
      wg = (double)((wig[l] >> DIR) & 1u);
  }
  fs_flux_inviscid<KFLU, DIR>(z, qm2, qm1, q0, qp1, wg, flu, nullptr, csc);
  if (IFLOW == 2) {
    constexpr int A = DIR == 0 ? 1 : 0, B = DIR == 2 ? 1 : 2;   // the transverse directions, in the order (i, j, k)
    auto q = [&](int e, int da, int db, int dn) -> double { return rop[e * nd + l + dn * st + da * inc[A] + db * inc[B]]; };
    fs_flux_viscous<DIR>(z, q, xmut[l - st], xmut[l], flu);
  }
}

template <int KFLU, int IFLOW>
__global__ void __launch_bounds__(128) k_box_faces_cart(FsZone z, FsBox box, const double *__restrict__ rop,
                                                        const double *__restrict__ xmut,
                                                        const unsigned char *__restrict__ wig, double *__restrict__ drodm)
{
  const int side = blockIdx.y;
  int i, j, k;
  if (!box_face_cell(z, box, side, (long long)blockIdx.x * blockDim.x + threadIdx.x, i, j, k)) return;
  const long long l = wcell(z, i, j, k), nd = z.nd;
  const int d = side >> 1;
  double f[5];
  if (d == 0) face_cart<KFLU, IFLOW, 0>(z, rop, xmut, wig, l, f);
  else if (d == 1) face_cart<KFLU, IFLOW, 1>(z, rop, xmut, wig, l, f);
  else face_cart<KFLU, IFLOW, 2>(z, rop, xmut, wig, l, f);
  if (side & 1) {
#pragma unroll
    for (int e = 0; e < 5; e++) drodm[e * nd + l] = drodm[e * nd + l] + f[e];
  } else {
    const long long l0 = l - (d == 0 ? 1 : (d == 1 ? z.incj : z.inck));
#pragma unroll
    for (int e = 0; e < 5; e++) drodm[e * nd + l0] = drodm[e * nd + l0] - f[e];
  }
}

}  // namespace

// ---- o3sc slope on Cartesian zones: one thread per cell, its six face fluxes from memory (every face is evaluated by
// the two cells it separates, with the same operands), balance and end of the stage as the tiled kernel has them -----------
namespace {

template <int KFLU, int IFLOW, bool FIRST, bool BDF>
__global__ void __launch_bounds__(256) k_stage_o3sc_cart(FsZone z, double coefH, double coefW, int keep,
                                                         const double *__restrict__ rop, double *__restrict__ rop_out,
                                                         double *__restrict__ drodm, const double *__restrict__ xmut,
                                                         unsigned char *__restrict__ wig, const double *__restrict__ wsc,
                                                         FsBdf bdf)
{
  const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x, j = 1 + blockIdx.y, k = 1 + blockIdx.z;
  if (i > z.iv) return;
  const long long nd = z.nd, l = wcell(z, i, j, k);
  const long long li = l + 1, lj = l + z.incj, lk = l + z.inck;
  double fkm[5], fjm[5], fim[5], fip[5], fjp[5], fkp[5];
  double w[6] = {0., 0., 0., 0., 0., 0.};
  constexpr bool SEN = KFLU == 2 && FIRST;
  face_cart<KFLU, IFLOW, 2>(z, rop, xmut, wig, l, fkm, wsc[l + 2 * nd], SEN ? &w[2] : nullptr);
  face_cart<KFLU, IFLOW, 1>(z, rop, xmut, wig, l, fjm, wsc[l + nd], SEN ? &w[1] : nullptr);
  face_cart<KFLU, IFLOW, 0>(z, rop, xmut, wig, l, fim, wsc[l], SEN ? &w[0] : nullptr);
  face_cart<KFLU, IFLOW, 0>(z, rop, xmut, wig, li, fip, wsc[li], SEN ? &w[3] : nullptr);
  face_cart<KFLU, IFLOW, 1>(z, rop, xmut, wig, lj, fjp, wsc[lj + nd], SEN ? &w[4] : nullptr);
  face_cart<KFLU, IFLOW, 2>(z, rop, xmut, wig, lk, fkp, wsc[lk + 2 * nd], SEN ? &w[5] : nullptr);
  if (SEN) {
    // the byte of a cell holds the sensors of its three lower faces; the ghost cells past a max side have no thread:
    // their byte (the sensor of the boundary face alone) is written by the cell before them
    wig[l] = (unsigned char)((w[0] != 0. ? 1u : 0u) | (w[1] != 0. ? 2u : 0u) | (w[2] != 0. ? 4u : 0u));
    if (i == z.iv) wig[li] = w[3] != 0. ? 1 : 0;
    if (j == z.jv) wig[lj] = w[4] != 0. ? 2 : 0;
    if (k == z.kv) wig[lk] = w[5] != 0. ? 4 : 0;
  }
  double q[5], d[5], o[5];
#pragma unroll
  for (int e = 0; e < 5; e++) {
    q[e] = rop[e * nd + l];
    const double dr = (FIRST || BDF) ? 0. : drodm[e * nd + l] * coefH;
    double a = dr + fkm[e];
    a = a + fjm[e]; a = a + fim[e];
    a = a - fip[e]; a = a - fjp[e];
    d[e] = a - fkp[e];
  }
  if (BDF) {
    double qn[5] = {0., 0., 0., 0., 0.}, qn1[5];
#pragma unroll
    for (int e = 0; e < 5; e++) qn1[e] = bdf.ro_n1[e * nd + l];
    if (!bdf.nit1) {
#pragma unroll
      for (int e = 0; e < 5; e++) qn[e] = bdf.ro_n[e * nd + l];
    }
    const double coe1 = bdf.coe1 ? bdf.coe1[l] : z.coe;
    fs_bdf_residual(z, bdf, coe1, q, qn, qn1, d, o);
#pragma unroll
    for (int e = 0; e < 5; e++) drodm[e * nd + l] = o[e];
  } else {
    if (keep) {
#pragma unroll
      for (int e = 0; e < 5; e++) drodm[e * nd + l] = d[e];
    }
    fs_rk_update(z, coefW, q, d, o);
#pragma unroll
    for (int e = 0; e < 5; e++) rop_out[e * nd + l] = o[e];
  }
}

}  // namespace

int fsk_stage_o3sc_cart(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                        double *drodm, const double *xmut, unsigned char *wig, const double *wsc, const FsBdf *bdf,
                        int keep_balance, hipStream_t s)
{
  if (z.ityp == 0 || z.neq != 5 || !wsc || (z.kflu == 2 && !wig)) return -1;
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, nullptr};
  if (bdf) b = *bdf;
  dim3 g((unsigned)((z.iv + 255) / 256), (unsigned)z.jv, (unsigned)z.kv);
#define FS_GO(K, F, FIRST_, BDF_) hipLaunchKernelGGL((k_stage_o3sc_cart<K, F, FIRST_, BDF_>), g, dim3(256), 0, s, z, coefH, coefW, \
                                                     keep_balance, rop, rop_out, drodm, xmut, wig, wsc, b)
#define FS_CASE(K, F)                                                                                         \
  if (z.kflu == K && z.iflow == F) {                                                                          \
    if (bdf) { if (first) FS_GO(K, F, true, true); else FS_GO(K, F, false, true); }                           \
    else { if (first) FS_GO(K, F, true, false); else FS_GO(K, F, false, false); }                             \
    return 0;                                                                                                 \
  }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
#undef FS_CASE
#undef FS_GO
  return -1;
}

void fsk_box_faces_cart(const FsZone &z, const int *box, const double *rop, const double *xmut, const unsigned char *wig,
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
#define FS_GO(K, F) hipLaunchKernelGGL((k_box_faces_cart<K, F>), g, dim3(128), 0, s, z, b, rop, xmut, wig, drodm)
  if (z.kflu == 1) { if (z.iflow == 2) FS_GO(1, 2); else FS_GO(1, 1); }
  else if (z.kflu == 2) { if (z.iflow == 2) FS_GO(2, 2); else FS_GO(2, 1); }
  else { if (z.iflow == 2) FS_GO(5, 2); else FS_GO(5, 1); }
#undef FS_GO
}

void fsk_roe_wall(const FsZone &z, int idir, const int *glo, const int *ghi, int tag, const double *rop, double *drodm,
                  hipStream_t s)
{
  const FsWallRow w = make_row(z, idir, glo, ghi, tag == 3);
  const long long n = row_cells(w);
  if (n <= 0) return;
  const unsigned g = (unsigned)((n + 127) / 128);
  if (z.ityp == 0) hipLaunchKernelGGL(k_roe_wall<true>, dim3(g), dim3(128), 0, s, z, w, rop, drodm);
  else hipLaunchKernelGGL(k_roe_wall<false>, dim3(g), dim3(128), 0, s, z, w, rop, drodm);
}

// Runge-Kutta update of the interior cells next to one wall window, from the corrected balance.
void fsk_finish_wall_rows(const FsZone &z, int idir, const int *glo, const int *ghi, double coefW, const double *rop_in,
                          double *rop_out, double *drodm, hipStream_t s)
{
  FsWallRow w = make_row(z, idir, glo, ghi, 0);
  const int ijkv[3] = {z.iv, z.jv, z.kv};
  w.lo[w.d] = w.hi[w.d] = w.maxside ? ijkv[w.d] : 1;      // the interior row
  const long long n = row_cells(w);
  if (n <= 0) return;
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, nullptr};
  hipLaunchKernelGGL(k_finish<false>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, z, w, coefW, rop_in, rop_out,
                     drodm, b);
}

// Runge-Kutta update of every interior cell with its own time step coe1 (time_step_nature='local' with the
// explicit scheme, core3ark3.for with coe(l,1) from tstb3_expli.for), from the balance kept in drodm.
void fsk_finish_rk_local(const FsZone &z, double coefW, const double *coe1, const double *rop_in, double *rop_out,
                         double *drodm, hipStream_t s)
{
  FsWallRow w;
  w.d = 0; w.maxside = 0; w.slip = 0;
  w.lo[0] = w.lo[1] = w.lo[2] = 1;
  w.hi[0] = z.iv; w.hi[1] = z.jv; w.hi[2] = z.kv;
  const long long n = row_cells(w);
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, coe1};
  hipLaunchKernelGGL(k_finish<false>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, z, w, coefW, rop_in, rop_out,
                     drodm, b);
}

// Newton residual of every interior cell from the (corrected) balance kept in drodm.
void fsk_finish_bdf(const FsZone &z, const double *rop_in, double *drodm, const FsBdf &bdf, hipStream_t s)
{
  FsWallRow w;
  w.d = 0; w.maxside = 0; w.slip = 0;
  w.lo[0] = w.lo[1] = w.lo[2] = 1;
  w.hi[0] = z.iv; w.hi[1] = z.jv; w.hi[2] = z.kv;
  const long long n = row_cells(w);
  hipLaunchKernelGGL(k_finish<true>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, z, w, 0., rop_in, (double *)nullptr,
                     drodm, bdf);
}
