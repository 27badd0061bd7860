// Implicit time step on the device: LU coefficients, LU-SGS sweeps and Newton update, gfx950.
// Built with -ffp-contract=off; expressions keep the operand order of the reference fragments
//   coefficients   Fast/FastS/FastS/Compute/tstb3_global.for:64-78, tstb3c.for:62-76,86-176,262-311
//   neighbour term LU/lu_{i,j,k}_3dcart.for, LU/lu_{i,j,k}_3dfull.for
//   sweeps         LU/invlu_l.for, LU/invlu_u.for (nb_relax = 1, lu_match = 0), LU/lu_d.for, lu_dinv.for
//   update         LU/mjr_newton.for
//
// The reference sweeps a zone in lexicographic order (k, j, then a serial i loop).  A cell only depends
// on its three predecessors (i-1, j-1, k-1 in the lower sweep; i+1, j+1, k+1 in the upper one), so all
// cells of a hyperplane i+j+k = const are independent: one launch per hyperplane, cells of a plane in
// parallel, the planes in stream order.  Per cell the operations are those of the reference in the
// same order, so the result is bit-identical to the lexicographic sweep.
#include "fasts_kernels.h"
#include "fasts_flux.h"

namespace {

struct LuRange { int i0, i1, j0, j1, k0, k1; };

FS_DEV long long lcell(const FsZone &z, int i, int j, int k)
{
  return (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
}
FS_DEV long long lmtr(const FsZone &z, int i, int j, int k)
{
  return (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.mincj + (long long)(k + z.ifick - 1) * z.minck;
}

// dt/vol, spectral radii and the LU diagonal of every cell of the zone, ghost cells included
// (storage indices si, sj, sk from 0).
template <bool FULL>
__global__ void __launch_bounds__(256) k_coe_implicit(FsZone z, double xid, double gam1, double gam2,
                                                       const double *__restrict__ rop,
                                                       const double *__restrict__ xmut, double *__restrict__ coe)
{
  const int si = blockIdx.x * blockDim.x + threadIdx.x, sj = blockIdx.y, sk = blockIdx.z;
  if (si >= z.ni) return;
  const long long nd = z.nd;
  const long long l = si + (long long)sj * z.incj + (long long)sk * z.inck;
  const long long lt = FULL ? si + (long long)sj * z.mincj + (long long)sk * z.minck : 0;
  const double cut0x = 1e-20;
  const double coe1 = FULL ? z.dt / z.vol[lt] : z.coe;
  const double r = rop[l], u = rop[l + nd], v = rop[l + 2 * nd], w = rop[l + 3 * nd];
  const double c = sqrt(gam2 * rop[l + 4 * nd]);
  const double detj = fs_max(FULL ? z.vol[lt] : z.vol0, cut0x);
  const double vmu = gam1 * xmut[l];
  const double volu = detj * r;
  const double xmvis = vmu / volu;
  double cs[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    double qn, tc2;
    if (FULL) {
      const double *__restrict__ t = z.mt[d];
      const long long lp = lt + (d == 0 ? 1 : (d == 1 ? z.mincj : z.minck)), ndm = z.ndm;
      const double tcx = .5 * (t[lt] + t[lp]), tcy = .5 * (t[lt + ndm] + t[lp + ndm]);
      const double tcz = .5 * (t[lt + 2 * ndm] + t[lp + 2 * ndm]);
      tc2 = .5 * (sqrt(t[lt] * t[lt] + t[lt + ndm] * t[lt + ndm] + t[lt + 2 * ndm] * t[lt + 2 * ndm]) +
                  sqrt(t[lp] * t[lp] + t[lp + ndm] * t[lp + ndm] + t[lp + 2 * ndm] * t[lp + 2 * ndm]));
      qn = tcx * u + tcy * v + tcz * w;
    } else {
      const double tc = d == 0 ? z.tcx : (d == 1 ? z.tcy : z.tcz);
      tc2 = sqrt(tc * tc);
      qn = tc * (d == 0 ? u : (d == 1 ? v : w));
    }
    const double cn = c * tc2;
    const double vm = fabs(qn) + cn;
    cs[d] = vm * 2. + xmvis * tc2 * tc2 * 4.;
  }
  coe[l] = coe1;
  coe[l + nd] = cs[0]; coe[l + 2 * nd] = cs[1]; coe[l + 3 * nd] = cs[2];
  coe[l + 4 * nd] = xid + coe1 * (cs[0] + cs[1] + cs[2]) * 0.5;
}

struct LuConst { double gam1, gamm1, gam2, cp, cvinv, cvinv2, cv; };

// b = [A(q_m) + signe*rho_d(m) I] dW(m), m = l - SGN*inc(D): SGN = +1 lower sweep, -1 upper sweep.
// rs = state array the neighbour is read from.
template <bool FULL, int D, int SGN>
FS_DEV void lu_neighbour(const FsZone &z, const LuConst &c, long long l, long long lt, const double *__restrict__ rs,
                         const double *__restrict__ dw, const double *__restrict__ coe, double *b)
{
  const long long nd = z.nd;
  const long long inc = D == 0 ? 1 : (D == 1 ? z.incj : z.inck);
  const long long l1 = l - SGN * inc;
  const double signe = SGN > 0 ? 0.5 : -0.5;
  const double gamm1 = c.gamm1, gam2 = c.gam2, gam1 = c.gam1, cp = c.cp;
  const double u = rs[l1 + nd], v = rs[l1 + 2 * nd], w = rs[l1 + 3 * nd], t = rs[l1 + 4 * nd];
  const double q2 = .5 * (u * u + v * v + w * w);
  const double h = cp * t + q2;
  const double ph2 = gamm1 * q2;
  const double cr = coe[l1 + (1 + D) * nd] * signe;
  const double d1 = dw[l1], d2 = dw[l1 + nd], d3 = dw[l1 + 2 * nd], d4 = dw[l1 + 3 * nd], d5 = dw[l1 + 4 * nd];
  if (FULL) {
    const double *__restrict__ tt = z.mt[D];
    const long long ndm = z.ndm, minc = D == 0 ? 1 : (D == 1 ? z.mincj : z.minck);
    const long long lt1 = SGN > 0 ? lt - minc : lt + minc;
    const long long lt2 = SGN > 0 ? lt : lt + 2 * minc;
    const double tcx = 0.5 * (tt[lt2] + tt[lt1]);
    const double tcy = 0.5 * (tt[lt2 + ndm] + tt[lt1 + ndm]);
    const double tcz = 0.5 * (tt[lt2 + 2 * ndm] + tt[lt1 + 2 * ndm]);
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
    return;
  }
  const double tc = D == 0 ? z.tcx : (D == 1 ? z.tcy : z.tcz);
  if (D == 0) {
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
  } else if (D == 1) {
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
  }
}

// cell of hyperplane h handled by this thread: threads run over j (x) and k (y, offset kbase)
FS_DEV bool plane_cell(const LuRange &r, int h, int kbase, int &i, int &j, int &k)
{
  j = r.j0 + blockIdx.x * blockDim.x + threadIdx.x;
  k = kbase + blockIdx.y;
  i = r.i0 + h - (j - r.j0) - (k - r.k0);
  return j <= r.j1 && k <= r.k1 && i >= r.i0 && i <= r.i1;
}

// Lower sweep, hyperplane h: dW*(l) = [dW(l) + xal*(sum of neighbour terms)] / diag, neighbours in
// the reference's order (Cartesian: j, k, i; curvilinear: k, j, i).
template <bool FULL>
__global__ void __launch_bounds__(64) k_lu_lower(FsZone z, LuConst c, LuRange r, int h, int kbase,
                                                 const double *__restrict__ rop, double *__restrict__ dw,
                                                 const double *__restrict__ coe)
{
  int i, j, k;
  if (!plane_cell(r, h, kbase, i, j, k)) return;
  const long long nd = z.nd, l = lcell(z, i, j, k), lt = FULL ? lmtr(z, i, j, k) : 0;
  const double xal = coe[l] * 0.5;
  double a[5], b[5];
#pragma unroll
  for (int e = 0; e < 5; e++) a[e] = dw[l + e * nd];
  if (FULL) {
    if (k > r.k0) { lu_neighbour<FULL, 2, 1>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
    if (j > r.j0) { lu_neighbour<FULL, 1, 1>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
  } else {
    if (j > r.j0) { lu_neighbour<FULL, 1, 1>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
    if (k > r.k0) { lu_neighbour<FULL, 2, 1>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
  }
  if (i > r.i0) { lu_neighbour<FULL, 0, 1>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
  const double diag = 1. / coe[l + 4 * nd];
#pragma unroll
  for (int e = 0; e < 5; e++) dw[l + e * nd] = a[e] * diag;
}

// Upper sweep, hyperplane h, followed by the Newton update of the cell:
//   dW(l) = [diag*dW*(l) - xal'*(neighbour terms k+1, j+1, i+1)] / diag,   W_new = W(rop) + dW -> rop_new.
// rop_kn = state array of the k+1 neighbour: the reference updates rop plane by plane while it sweeps,
// so when the iterate is updated in place (sub-iterations after the first) the k+1 neighbour already
// holds the new state, the j+1 and i+1 neighbours the old one.
template <bool FULL>
__global__ void __launch_bounds__(64) k_lu_upper(FsZone z, LuConst c, LuRange r, int h, int kbase,
                                                 const double *__restrict__ rop, const double *rop_kn,
                                                 double *rop_new, double *__restrict__ dw,
                                                 const double *__restrict__ coe)
{
  int i, j, k;
  if (!plane_cell(r, h, kbase, i, j, k)) return;
  const long long nd = z.nd, l = lcell(z, i, j, k), lt = FULL ? lmtr(z, i, j, k) : 0;
  const double xal = coe[l] * -0.5;
  const double c5 = coe[l + 4 * nd];
  double a[5], b[5];
#pragma unroll
  for (int e = 0; e < 5; e++) a[e] = dw[l + e * nd] * c5;
  if (k < r.k1) { lu_neighbour<FULL, 2, -1>(z, c, l, lt, rop_kn, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
  if (j < r.j1) { lu_neighbour<FULL, 1, -1>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
  if (i < r.i1) { lu_neighbour<FULL, 0, -1>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < 5; e++) a[e] = a[e] + b[e] * xal; }
  const double diag = 1. / c5;
#pragma unroll
  for (int e = 0; e < 5; e++) { a[e] = a[e] * diag; dw[l + e * nd] = a[e]; }
  // Newton update (mjr_newton.for)
  const double ro_old = rop[l], u_old = rop[l + nd], v_old = rop[l + 2 * nd], w_old = rop[l + 3 * nd];
  const double t_old = rop[l + 4 * nd];
  const double rn = ro_old + a[0];
  const double r_1 = 1. / rn;
  const double un = (ro_old * u_old + a[1]) * r_1;
  const double vn = (ro_old * v_old + a[2]) * r_1;
  const double wn = (ro_old * w_old + a[3]) * r_1;
  const double roe_old = ro_old * (c.cv * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
  rop_new[l] = rn; rop_new[l + nd] = un; rop_new[l + 2 * nd] = vn; rop_new[l + 3 * nd] = wn;
  rop_new[l + 4 * nd] = (roe_old + a[4]) * r_1 * c.cvinv - c.cvinv2 * (un * un + vn * vn + wn * wn);
}

__global__ void __launch_bounds__(256) k_copy_range(FsZone z, LuRange r, const double *__restrict__ src,
                                                    double *__restrict__ dst)
{
  const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x, j = r.j0 + blockIdx.y, k = r.k0 + blockIdx.z;
  if (i > r.i1) return;
  const long long l = lcell(z, i, j, k);
#pragma unroll
  for (int e = 0; e < 5; e++) dst[l + e * z.nd] = src[l + e * z.nd];
}

}  // namespace

void fsk_coe_implicit(const FsZone &z, int first_it, double gamma, double prandtl, double rgp, const double *rop,
                      const double *xmut, double *coe, hipStream_t s)
{
  double xid = 1.5;
  if (first_it == 0) xid = 1.;
  double gam1 = gamma / prandtl;
  if (z.iflow == 1) gam1 = 0.;
  const double gam2 = gamma * rgp;
  dim3 g((z.ni + 255) / 256, z.nj, z.nk);
  if (z.ityp == 0) hipLaunchKernelGGL((k_coe_implicit<true>), g, dim3(256), 0, s, z, xid, gam1, gam2, rop, xmut, coe);
  else hipLaunchKernelGGL((k_coe_implicit<false>), g, dim3(256), 0, s, z, xid, gam1, gam2, rop, xmut, coe);
}

// LU-SGS on the interior of the zone + Newton update.  rop = current iterate; the new iterate is
// written to rop_out.  When rop_out == rop (in-place update, sub-iterations after the first) the new
// state is built in scratch (a full time level) and copied back.
int fsk_lusgs(const FsZone &z, double gamma, const double *rop, double *rop_out, double *scratch, double *dw,
              const double *coe, hipStream_t s)
{
  LuConst c;
  c.gam1 = gamma; c.gamm1 = gamma - 1.; c.cp = gamma * z.cv; c.gam2 = c.gamm1 - 1.;
  c.cv = z.cv; c.cvinv = 1. / z.cv; c.cvinv2 = 0.5 * c.cvinv;
  LuRange r = {1, z.iv, 1, z.jv, 1, z.kv};
  const int nh = z.iv + z.jv + z.kv - 2;
  const bool full = z.ityp == 0;
  const bool inplace = rop_out == rop;
  if (inplace && !scratch) return -1;
  double *rop_new = inplace ? scratch : rop_out;
  const double *rop_kn = inplace ? scratch : rop;
  const unsigned gx = (unsigned)((z.jv + 63) / 64);
  for (int h = 0; h < nh; h++) {
    // k values that can hold a cell of this plane
    int klo = h - (z.iv - 1) - (z.jv - 1); if (klo < 0) klo = 0;
    int khi = h < z.kv - 1 ? h : z.kv - 1;
    dim3 g(gx, (unsigned)(khi - klo + 1));
    if (full) hipLaunchKernelGGL((k_lu_lower<true>), g, dim3(64), 0, s, z, c, r, h, r.k0 + klo, rop, dw, coe);
    else hipLaunchKernelGGL((k_lu_lower<false>), g, dim3(64), 0, s, z, c, r, h, r.k0 + klo, rop, dw, coe);
  }
  for (int h = nh - 1; h >= 0; h--) {
    int klo = h - (z.iv - 1) - (z.jv - 1); if (klo < 0) klo = 0;
    int khi = h < z.kv - 1 ? h : z.kv - 1;
    dim3 g(gx, (unsigned)(khi - klo + 1));
    if (full) hipLaunchKernelGGL((k_lu_upper<true>), g, dim3(64), 0, s, z, c, r, h, r.k0 + klo, rop, rop_kn, rop_new, dw, coe);
    else hipLaunchKernelGGL((k_lu_upper<false>), g, dim3(64), 0, s, z, c, r, h, r.k0 + klo, rop, rop_kn, rop_new, dw, coe);
  }
  if (inplace) {
    dim3 g((unsigned)((z.iv + 255) / 256), (unsigned)z.jv, (unsigned)z.kv);
    hipLaunchKernelGGL(k_copy_range, g, dim3(256), 0, s, z, r, (const double *)scratch, rop_out);
  }
  return 0;
}
