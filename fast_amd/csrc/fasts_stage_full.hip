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
// The own column's four k-planes are carried in registers; everything else (transverse neighbours,
// metrics, viscosity) is read from global memory and served by the vector L1 / L2.
#include "fasts_kernels.h"
#include "fasts_flux.h"

namespace {

constexpr int TI = 32, TJ = 8, OI = TI - 1, OJ = TJ - 1;

template <int DIR> FS_DEV long long inc_of(const FsZone &z) { return DIR == 0 ? 1 : (DIR == 1 ? z.incj : z.inck); }
template <int DIR> FS_DEV long long minc_of(const FsZone &z) { return DIR == 0 ? 1 : (DIR == 1 ? z.mincj : z.minck); }

// Face between cell l - inc(DIR) and cell l.  qm2..qp1 = the four cells of the MUSCL stencil.
// FIRST: the sensor of the face is evaluated from the stencil and stored (0/1 byte); later stages read it.
// TWOD: two-dimensional zone; the k stride of the field is taken as 0 (the cross terms along k then read the
// cell itself and meet zero metrics: they vanish exactly, as in the reference's 2d fragments fluvisq_2d_{i,j}.for).
template <int KFLU, int IFLOW, int DIR, bool FIRST, bool TWOD>
FS_DEV void face_full(const FsZone &z, const double *__restrict__ rop, const double *__restrict__ xmut,
                      unsigned char *__restrict__ wig, long long l, long long lt, const double *qm2,
                      const double *qm1, const double *q0, const double *qp1, double *flu)
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
  fs_flux_inviscid_full<KFLU>(z, tcx, tcy, tcz, qm2, qm1, q0, qp1, wg, flu, TWOD && DIR == 1);
  if (IFLOW == 2) {
    constexpr int A = (DIR + 1) % 3, B = (DIR + 2) % 3;
    const long long inc = inc_of<DIR>(z), ia = (TWOD && A == 2) ? 0 : inc_of<A>(z), ib = (TWOD && B == 2) ? 0 : inc_of<B>(z);
    const long long md = minc_of<DIR>(z), ma = minc_of<A>(z), mb = minc_of<B>(z);
    auto q = [&](int e, int da, int db, int dn) -> double {
      if (da == 0 && db == 0) return dn == 0 ? q0[e] : qm1[e];
      return rop[e * nd + l + dn * inc + da * ia + db * ib];
    };
    auto T = [&](int c, int m, int o) -> double {
      const double *__restrict__ t = c == 0 ? Tn : (c == 1 ? z.mt[A] : z.mt[B]);
      const long long mc = c == 1 ? ma : mb;
      const long long off = o == 0 ? 0 : (o == 1 ? -md : (o == 2 ? md : (o == 3 ? mc : mc - md)));
      return t[lt + off + m * ndm];
    };
    fs_flux_viscous_full(z, q, T, z.vol[lt], z.vol[lt - md], xmut[l - inc], xmut[l], tcx, tcy, tcz, flu);
  }
}

// BDF = false: Runge-Kutta stage.  BDF = true: Newton residual of the implicit scheme (rop_out unused).
// TWOD: one plane, no k faces: the cell is finished in the same step (+Fj- +Fi- -Fi+ -Fj+).
template <int KFLU, int IFLOW, bool FIRST, bool BDF, bool TWOD>
__global__ void __launch_bounds__(256)
k_stage_full(FsZone z, double coefH, double coefW, int kchunk, const double *__restrict__ rop,
             double *__restrict__ rop_out, double *__restrict__ drodm, const double *__restrict__ xmut,
             unsigned char *__restrict__ wig, FsBdf bdf)
{
  __shared__ double sfj[2][5][TJ][TI];
  const int ti = threadIdx.x & 31, tj = threadIdx.x >> 5;
  const int i = 1 + blockIdx.x * OI + ti, j = 1 + blockIdx.y * OJ + tj;
  const int k0 = 1 + blockIdx.z * kchunk;
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
  double qa[5], qb[5], qc[5], qd[5];   // own column at k-2, k-1, k, k+1
#pragma unroll
  for (int e = 0; e < 5; e++) {
    qa[e] = TWOD ? 0. : rop[e * nd + l - 2 * z.inck];
    qb[e] = TWOD ? 0. : rop[e * nd + l - z.inck];
    qc[e] = rop[e * nd + l];
    qd[e] = 0.;
  }
  double part[5] = {0., 0., 0., 0., 0.};
  for (int k = k0; k <= (TWOD ? k0 : k1 + 1); k++) {
    const bool last = !TWOD && k > k1;
    if (!TWOD) {
#pragma unroll
      for (int e = 0; e < 5; e++) qd[e] = rop[e * nd + l + z.inck];
    }
    double fk[5] = {0., 0., 0., 0., 0.};
    // finish: the balance of a cell is complete (3-D: cell k-1 once its k+1/2 face is known; 2-D: the cell itself)
    auto finish = [&](long long lb, long long ltb, const double *qcell, const double *d) {
        double o[5];
        const double coe = z.dt / z.vol[ltb];
        if (BDF) {
          double qn[5] = {0., 0., 0., 0., 0.}, qn1[5];
#pragma unroll
          for (int e = 0; e < 5; e++) qn1[e] = bdf.ro_n1[e * nd + lb];
          if (!bdf.nit1) {
#pragma unroll
            for (int e = 0; e < 5; e++) qn[e] = bdf.ro_n[e * nd + lb];
          }
          fs_bdf_residual(z, bdf, bdf.coe1 ? bdf.coe1[lb] : coe, qcell, qn, qn1, d, o);
#pragma unroll
          for (int e = 0; e < 5; e++) drodm[e * nd + lb] = o[e];
        } else {
#pragma unroll
          for (int e = 0; e < 5; e++) drodm[e * nd + lb] = d[e];
          fs_rk_update_c(z, coefW, coe, qcell, d, o);
          if (TWOD) o[3] = 0.;   // core3ark3.for:156-183
#pragma unroll
          for (int e = 0; e < 5; e++) rop_out[e * nd + lb] = o[e];
        }
    };
    if (own && !TWOD) {
      face_full<KFLU, IFLOW, 2, FIRST, TWOD>(z, rop, xmut, wig, l, lt, qa, qb, qc, qd, fk);
      if (k > k0) {   // the cell below is complete
        double d[5];
#pragma unroll
        for (int e = 0; e < 5; e++) d[e] = part[e] - fk[e];
        finish(l - z.inck, lt - z.minck, qb, d);
      }
    }
    if (!last) {
      double fim[5] = {0., 0., 0., 0., 0.}, fjm[5] = {0., 0., 0., 0., 0.};
      if (do_fj) {
        double m2[5], m1[5], p1[5];
#pragma unroll
        for (int e = 0; e < 5; e++) {
          m2[e] = rop[e * nd + l - 2 * z.incj]; m1[e] = rop[e * nd + l - z.incj]; p1[e] = rop[e * nd + l + z.incj];
        }
        face_full<KFLU, IFLOW, 1, FIRST, TWOD>(z, rop, xmut, wig, l, lt, m2, m1, qc, p1, fjm);
      }
      const int buf = (k - k0) & 1;
#pragma unroll
      for (int e = 0; e < 5; e++) sfj[buf][e][tj][ti] = fjm[e];
      if (do_fi) {
        double m2[5], m1[5], p1[5];
#pragma unroll
        for (int e = 0; e < 5; e++) {
          m2[e] = rop[e * nd + l - 2]; m1[e] = rop[e * nd + l - 1]; p1[e] = rop[e * nd + l + 1];
        }
        face_full<KFLU, IFLOW, 0, FIRST, TWOD>(z, rop, xmut, wig, l, lt, m2, m1, qc, p1, fim);
      }
      double fip[5];
#pragma unroll
      for (int e = 0; e < 5; e++) fip[e] = __shfl_down(fim[e], 1);
      __syncthreads();
      if (own) {
#pragma unroll
        for (int e = 0; e < 5; e++) {
          const double dr = (FIRST || BDF) ? 0. : drodm[e * nd + l] * coefH;
          part[e] = dr + fk[e] + fjm[e] + fim[e] - fip[e] - sfj[buf][e][tj + 1][ti];
        }
        if (TWOD) finish(l, lt, qc, part);
      }
    }
#pragma unroll
    for (int e = 0; e < 5; e++) { qa[e] = qb[e]; qb[e] = qc[e]; qc[e] = qd[e]; }
    l += z.inck; lt += z.minck;
  }
}

}  // namespace

int fsk_stage_full(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                   double *drodm, const double *xmut, unsigned char *wig, const FsBdf *bdf, hipStream_t s)
{
  if (z.ityp != 0 || !z.vol || !z.mt[0] || !z.mt[1] || !z.mt[2]) return -1;
  int kchunk = 32;
  if (const char *ev = getenv("FASTS_HIP_KCHUNK_FULL")) {
    int v = atoi(ev);
    if (v >= 1) kchunk = v;
  }
  if (kchunk > z.kv) kchunk = z.kv;
  dim3 g((z.iv + OI - 1) / OI, (z.jv + OJ - 1) / OJ, (z.kv + kchunk - 1) / kchunk);
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, nullptr};
  if (bdf) b = *bdf;
#define FS_GO(K, F, FIRST_, BDF_)                                                                             \
  do {                                                                                                        \
    if (z.twod)                                                                                               \
      hipLaunchKernelGGL((k_stage_full<K, F, FIRST_, BDF_, true>), g, dim3(256), 0, s, z, coefH, coefW, kchunk, rop,  \
                         rop_out, drodm, xmut, wig, b);                                                       \
    else                                                                                                      \
      hipLaunchKernelGGL((k_stage_full<K, F, FIRST_, BDF_, false>), g, dim3(256), 0, s, z, coefH, coefW, kchunk, rop, \
                         rop_out, drodm, xmut, wig, b);                                                       \
  } while (0)
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
