// Implicit time step on the device: LU coefficients, LU-SGS sweeps and Newton update, gfx950.
// Built with -ffp-contract=off; expressions keep the operand order of the reference fragments
//   coefficients   Fast/FastS/FastS/Compute/tstb3_global.for:64-78, tstb3c.for:62-76,86-176,262-311
//   neighbour term LU/lu_{i,j,k}_3dcart.for, LU/lu_{i,j,k}_3dfull.for
//   sweeps         LU/invlu_l.for, LU/invlu_u.for (nb_relax = 1, lu_match = 0), LU/lu_d.for, lu_dinv.for;
//                  LU/invlu_overlap_l.for, invlu_overlap_u.for (lu_match = 1) with the LU-SSOR kernels below
//   update         LU/mjr_newton.for
//
// The reference sweeps a zone in lexicographic order (k, j, then a serial i loop).  A cell only depends
// on its three predecessors (i-1, j-1, k-1 in the lower sweep; i+1, j+1, k+1 in the upper one), so all
// cells of a hyperplane i+j+k = const are independent: one launch per hyperplane, cells of a plane in
// parallel, the planes in stream order.  Per cell the operations are those of the reference in the
// same order, so the result is bit-identical to the lexicographic sweep.
#include "fasts_kernels.h"
#include "fasts_flux.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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
// DTLOC: local time step, coe(:,1) = CFL * min(convective, viscous) limit of the cell
// (tstb3_impli.for:57-71, dtloc.for); cut0x = 1e-30 there, 1e-20 with the global time step.
template <bool FULL, bool DTLOC>
__global__ void __launch_bounds__(256) k_coe_implicit(FsZone z, double xid, double gam1, double gam2, double cfl,
                                                       const double *__restrict__ rop,
                                                       const double *__restrict__ xmut, double *__restrict__ coe)
{
  const int si = blockIdx.x * blockDim.x + threadIdx.x, sj = blockIdx.y, sk = blockIdx.z;
  if (si >= z.ni) return;
  const long long nd = z.nd;
  const long long l = si + (long long)sj * z.incj + (long long)sk * z.inck;
  const long long lt = FULL ? si + (long long)sj * z.mincj + (long long)sk * z.minck : 0;
  const double cut0x = DTLOC ? 1e-30 : 1e-20;
  double coe1 = FULL ? z.dt / z.vol[lt] : z.coe;
  const double r = rop[l], u = rop[l + nd], v = rop[l + 2 * nd], w = rop[l + 3 * nd];
  const double c = sqrt(gam2 * rop[l + 4 * nd]);
  const double detj = fs_max(FULL ? z.vol[lt] : z.vol0, cut0x);
  const double vmu = gam1 * xmut[l];
  const double volu = detj * r;
  const double xmvis = vmu / volu;
  double cs[3], t2[3];
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
    t2[d] = tc2;
  }
  if (DTLOC) {
    const double gam1_1 = 1. / gam1;
    const double dd = 1. / (t2[0] * t2[0] + t2[1] * t2[1] + t2[2] * t2[2]);
    const double dtvis = 0.5 * detj * dd * gam1_1 / xmut[l] * r;
    const double sp = sqrt(u * u + v * v + w * w) + c;
    const double dtconv = sqrt(dd) / sp;
    coe1 = cfl * fs_min(dtconv, dtvis);
  }
  coe[l] = coe1;
  coe[l + nd] = cs[0]; coe[l + 2 * nd] = cs[1]; coe[l + 3 * nd] = cs[2];
  coe[l + 4 * nd] = xid + coe1 * (cs[0] + cs[1] + cs[2]) * 0.5;
}

struct LuConst { double gam1, gamm1, gam2, cp, cvinv, cvinv2, cv; };

// b = [A(q_m) + signe*rho_d(m) I] dW(m) for the neighbour m = l - SGN*inc(D) of cell l (SGN = +1 lower
// sweep, -1 upper sweep), from the neighbour's state (u,v,w,t), its increment d1..d5 and its spectral
// radius crad = coe(m, 2+D); lt = metric index of cell l.
// NE = 6: Spalart-Allmaras, the 6th equation is a decoupled scalar, b6 = (qn + signe*rho_d) dW6
// (LU/lu_{i,j,k}_3dcart_SA.for).
// Matrix of the neighbour term, M[5*r + c] = b(r+1)(c+1) of the reference's fragments (entries the fragment of the
// direction does not have are left unset and never read); M[25] = qn + signe*rho_d, the factor of the 6th equation.
// It depends on the neighbour's state and radius only, not on its increment: the i recurrence of the sweeps computes
// it once per cell, outside the serial chain.
template <bool FULL, int D, int SGN, int NE = 5>
FS_DEV void lu_mat(const FsZone &z, const LuConst &c, long long lt, double u, double v, double w, double t, double crad,
                   double *M)
{
  const double signe = SGN > 0 ? 0.5 : -0.5;
  const double gamm1 = c.gamm1, gam2 = c.gam2, gam1 = c.gam1, cp = c.cp;
  const double q2 = .5 * (u * u + v * v + w * w);
  const double h = cp * t + q2;
  const double ph2 = gamm1 * q2;
  const double cr = crad * signe;
  if (FULL) {
    const double *__restrict__ tt = z.mt[D];
    const long long ndm = z.ndm, minc = D == 0 ? 1 : (D == 1 ? z.mincj : z.minck);
    const long long lt1 = SGN > 0 ? lt - minc : lt + minc;
    const long long lt2 = SGN > 0 ? lt : lt + 2 * minc;
    const double tcx = 0.5 * (tt[lt2] + tt[lt1]);
    const double tcy = 0.5 * (tt[lt2 + ndm] + tt[lt1 + ndm]);
    const double tcz = 0.5 * (tt[lt2 + 2 * ndm] + tt[lt1 + 2 * ndm]);
    const double qn = tcx * u + tcy * v + tcz * w;
    M[0] = cr; M[1] = tcx; M[2] = tcy; M[3] = tcz;
    M[5] = (tcx * ph2 - u * qn); M[6] = (qn - tcx * gam2 * u) + cr;
    M[7] = (tcy * u - tcx * gamm1 * v); M[8] = (tcz * u - tcx * gamm1 * w); M[9] = tcx * gamm1;
    M[10] = (tcy * ph2 - v * qn); M[11] = (tcx * v - tcy * gamm1 * u);
    M[12] = (qn - tcy * gam2 * v) + cr; M[13] = (tcz * v - tcy * gamm1 * w); M[14] = tcy * gamm1;
    M[15] = (tcz * ph2 - w * qn); M[16] = (tcx * w - tcz * gamm1 * u);
    M[17] = (tcy * w - tcz * gamm1 * v); M[18] = (qn - tcz * gam2 * w) + cr; M[19] = tcz * gamm1;
    M[20] = qn * (ph2 - h); M[21] = (tcx * h - gamm1 * u * qn); M[22] = (tcy * h - gamm1 * v * qn);
    M[23] = (tcz * h - gamm1 * w * qn); M[24] = (gam1 * qn) + cr;
    if (NE == 6) M[25] = qn + cr;
    return;
  }
  const double tc = D == 0 ? z.tcx : (D == 1 ? z.tcy : z.tcz);
  if (D == 0) {
    const double qn = tc * u;
    M[0] = cr; M[1] = tc; M[5] = (tc * ph2 - u * qn); M[6] = (qn - tc * gam2 * u) + cr;
    M[7] = (-tc * gamm1 * v); M[8] = (-tc * gamm1 * w); M[9] = tc * gamm1;
    M[10] = (-v * qn); M[11] = (tc * v); M[12] = (qn) + cr;
    M[15] = (-w * qn); M[16] = (tc * w); M[18] = (qn) + cr;
    M[20] = qn * (ph2 - h); M[21] = (tc * h - gamm1 * u * qn); M[22] = (-gamm1 * v * qn);
    M[23] = (-gamm1 * w * qn); M[24] = (gam1 * qn) + cr;
    if (NE == 6) M[25] = qn + cr;
  } else if (D == 1) {
    const double qn = tc * v;
    M[0] = cr; M[2] = tc; M[5] = (-u * qn); M[6] = qn + cr; M[7] = tc * u;
    M[10] = (tc * ph2 - v * qn); M[11] = (-tc * gamm1 * u); M[12] = (qn - tc * gam2 * v) + cr;
    M[13] = (-tc * gamm1 * w); M[14] = tc * gamm1;
    M[15] = (-w * qn); M[17] = (tc * w); M[18] = (qn) + cr;
    M[20] = qn * (ph2 - h); M[21] = (-gamm1 * u * qn); M[22] = (tc * h - gamm1 * v * qn);
    M[23] = (-gamm1 * w * qn); M[24] = (gam1 * qn) + cr;
    if (NE == 6) M[25] = qn + cr;
  } else {
    const double qn = tc * w;
    M[0] = cr; M[3] = tc; M[5] = -u * qn; M[6] = qn + cr; M[8] = tc * u;
    M[10] = -v * qn; M[12] = qn + cr; M[13] = tc * v;
    M[15] = tc * ph2 - w * qn; M[16] = -tc * gamm1 * u; M[17] = -tc * gamm1 * v;
    M[18] = (qn - tc * gam2 * w) + cr; M[19] = tc * gamm1;
    M[20] = qn * (ph2 - h); M[21] = -gamm1 * u * qn; M[22] = -gamm1 * v * qn;
    M[23] = tc * h - gamm1 * w * qn; M[24] = gam1 * qn + cr;
    if (NE == 6) M[25] = qn + cr;
  }
}

// b = M dW, the sums of the fragment of the direction (LU/lu_{i,j,k}_3dcart.for, lu_{i,j,k}_3dfull.for)
template <bool FULL, int D, int NE = 5>
FS_DEV void lu_mul(const double *M, double d1, double d2, double d3, double d4, double d5, double *b, double d6 = 0.)
{
  if (FULL) {
    b[0] = M[0] * d1 + M[1] * d2 + M[2] * d3 + M[3] * d4;
    b[1] = M[5] * d1 + M[6] * d2 + M[7] * d3 + M[8] * d4 + M[9] * d5;
    b[2] = M[10] * d1 + M[11] * d2 + M[12] * d3 + M[13] * d4 + M[14] * d5;
    b[3] = M[15] * d1 + M[16] * d2 + M[17] * d3 + M[18] * d4 + M[19] * d5;
    b[4] = M[20] * d1 + M[21] * d2 + M[22] * d3 + M[23] * d4 + M[24] * d5;
  } else if (D == 0) {
    b[0] = M[0] * d1 + M[1] * d2;
    b[1] = M[5] * d1 + M[6] * d2 + M[7] * d3 + M[8] * d4 + M[9] * d5;
    b[2] = M[10] * d1 + M[11] * d2 + M[12] * d3;
    b[3] = M[15] * d1 + M[16] * d2 + M[18] * d4;
    b[4] = M[20] * d1 + M[21] * d2 + M[22] * d3 + M[23] * d4 + M[24] * d5;
  } else if (D == 1) {
    b[0] = M[0] * d1 + M[2] * d3;
    b[1] = M[5] * d1 + M[6] * d2 + M[7] * d3;
    b[2] = M[10] * d1 + M[11] * d2 + M[12] * d3 + M[13] * d4 + M[14] * d5;
    b[3] = M[15] * d1 + M[17] * d3 + M[18] * d4;
    b[4] = M[20] * d1 + M[21] * d2 + M[22] * d3 + M[23] * d4 + M[24] * d5;
  } else {
    b[0] = M[0] * d1 + M[3] * d4;
    b[1] = M[5] * d1 + M[6] * d2 + M[8] * d4;
    b[2] = M[10] * d1 + M[12] * d3 + M[13] * d4;
    b[3] = M[15] * d1 + M[16] * d2 + M[17] * d3 + M[18] * d4 + M[19] * d5;
    b[4] = M[20] * d1 + M[21] * d2 + M[22] * d3 + M[23] * d4 + M[24] * d5;
  }
  if (NE == 6) b[5] = M[25] * d6;
}

template <bool FULL, int D, int SGN, int NE = 5>
FS_DEV void lu_term(const FsZone &z, const LuConst &c, long long lt, double u, double v, double w, double t,
                    double d1, double d2, double d3, double d4, double d5, double crad, double *b, double d6 = 0.)
{
  double M[26];
  lu_mat<FULL, D, SGN, NE>(z, c, lt, u, v, w, t, crad, M);
  lu_mul<FULL, D, NE>(M, d1, d2, d3, d4, d5, b, d6);
}

// rs = state array the neighbour is read from, dw = increments, coe = coefficient array
template <bool FULL, int D, int SGN, int NE = 5>
FS_DEV void lu_neighbour(const FsZone &z, const LuConst &c, long long l, long long lt, const double *__restrict__ rs,
                         const double *__restrict__ dw, const double *__restrict__ coe, double *b)
{
  const long long nd = z.nd;
  const long long inc = D == 0 ? 1 : (D == 1 ? z.incj : z.inck);
  const long long l1 = l - SGN * inc;
  lu_term<FULL, D, SGN, NE>(z, c, lt, rs[l1 + nd], rs[l1 + 2 * nd], rs[l1 + 3 * nd], rs[l1 + 4 * nd], dw[l1], dw[l1 + nd],
                            dw[l1 + 2 * nd], dw[l1 + 3 * nd], dw[l1 + 4 * nd], coe[l1 + (1 + D) * nd], b,
                            NE == 6 ? dw[l1 + 5 * nd] : 0.);
}

// ------------------------------------------------------------------------------------------------------
// Sweeps over hyperplanes of (unit of U consecutive cells of an i-line, j, k): a wave handles 64 / U units, lane =
// (unit, cell).  The loads and the j/k neighbour terms are done one cell per lane (the U lanes of a unit read U * 8
// contiguous bytes of every array); the i recurrence then runs in U steps, the increment of cell m-1 reaching cell m
// through a DPP row shift inside the unit.  U = 8 for one sweep of a zone (launches bound by their latency), 16 when
// the blocks of the reference's threads are swept side by side (launches bound by memory).
// ------------------------------------------------------------------------------------------------------
#ifdef FS_LU_STAMP
// development aid: time stamps of wave (0,0) of every lower-sweep launch (profiles/lu_stamps.py)
__device__ long long fs_lu_stamp_buf[1024 * 8];
#define LUSTAMP(slot, wait)                                                                               \
  {                                                                                                       \
    if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && h < 1024)                               \
      fs_lu_stamp_buf[h * 8 + (slot)] = (slot) == 0 || (slot) == 7 ? (long long)wall_clock64() : (long long)clock64(); \
  }
extern "C" int fs_lu_stamp_fetch(long long *host)
{
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fs_lu_stamp_buf), sizeof(long long) * 1024 * 8);
}
// the same for the one-launch sweeps: wall-clock stamps of the wave that works the task in the middle of the order
// (lower sweep), five per step
#define FLOWSTAMP(slot)                                                                                   \
  {                                                                                                       \
    asm volatile("" ::: "memory");                                                                        \
    if (n == ntasks / 2 && threadIdx.x == 0 && s < 128) fs_lu_stamp_buf[s * 8 + (slot)] = (long long)wall_clock64(); \
    asm volatile("" ::: "memory");                                                                        \
  }
#else
#define LUSTAMP(slot, wait)
#define FLOWSTAMP(slot)
#endif

// Value of the previous / next lane by a DPP row shift (a VALU move, no trip through the LDS crossbar as
// __shfl_up / __shfl_down take).  Rows are 16 lanes, units 8 or 16: the first (last) lane of a unit receives the value of a
// lane of the neighbouring unit or zero, and never uses it.
FS_DEV double lane_prev(double x)
{
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);   // row_shr:1, lanes without a source read 0
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
FS_DEV double lane_next(double x)
{
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true);   // row_shl:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// Units on 128-byte lines.  A unit of 16 cells is one line of every array when it starts on a multiple of 16 cells in
// memory, and straddles two otherwise (the sweeps of the blocks are bound by memory: FETCH_SIZE was 1.9 x the bytes they
// use).  The i-lines of a zone start p0 + a (j - j0) + b (k - k0) cells after such a multiple, modulo U (a, b: the row
// and plane strides modulo U).  Shifting the units of line (j, k) down by that many cells -- without the modulo, so
// that the shift never decreases from a line to the next -- puts every unit on a line and keeps the order of the
// hyperplanes: the j - 1 and k - 1 neighbours of the cells of unit cn then sit in units cn - 1 or cn of their lines,
// never cn + 1.  The price is (p0 + a (nj - 1) + b (nk - 1)) / U more chunk columns, i.e. hyperplanes.  Measured
// (ms per implicit step, 3 Newton sub-iterations, shift off / on): 252^3 in 64 blocks, a = b = 0, no extra hyperplane:
// 17.2 / 13.9; 256^3 (a = 4) in 200 blocks, 14 % more hyperplanes: 14.4 / 13.6; in 128 blocks, 12 %: 14.6 / 13.8; in 64
// blocks, 17 %: 15.9 / 15.7; in 32 blocks, 20 %: 18.0 / 18.6; two-dimensional 1024 x 256 zone (small launches, bound by
// latency) in 200 blocks, 11 %: 1.44 / 1.65.  So enable = 1 shifts when no hyperplane is added, or at most 15 % on a
// three-dimensional zone; enable = 2 whenever the stride between variables keeps the alignment (tests of the
// general case).
// All zero: units start at the first cell of the range.
struct LuSkew { int p0, a, b; };
// chunk columns of a block with its skew
// FASTS_HIP_LU_ALIGN: 0, 1 (default) or 2, + 4 for the plain workgroup order; anything else is the default.  Read at
// every solve (a tuning aid that the tests switch between cases); two getenv calls per solve are not measurable.
static int lu_align_setting()
{
  const char *e = getenv("FASTS_HIP_LU_ALIGN");
  const int a = e ? atoi(e) : 1;
  return (a < 0 || a > 7 || (a & 3) == 3) ? 1 : a;
}

template <int U>
__host__ __device__ inline int lu_nchunk(const LuSkew &s, int ni, int nj, int nk)
{
  return (ni + s.p0 + s.a * (nj - 1) + s.b * (nk - 1) + U - 1) / U;
}
template <int U>
__host__ __device__ inline LuSkew lu_skew(const FsZone &z, int i0, int i1, int j0, int j1, int k0, int k1, int enable)
{
  LuSkew s = {0, 0, 0};
  if (!enable || z.nd % U != 0) return s;
  const int a = (int)(z.incj % U), b = (int)(z.inck % U);
  const int p0 = ((i0 + z.ific - 1) + (j0 + z.ific - 1) * a + (k0 + z.ifick - 1) * b) % U;
  const int nj = j1 - j0 + 1, nk = k1 - k0 + 1;
  const int plain = (i1 - i0 + U) / U;                                       // chunk columns without the shift
  const int extra = (i1 - i0 + 1 + p0 + a * (nj - 1) + b * (nk - 1) + U - 1) / U - plain;
  const bool too_many = extra > 0 && (z.nk == 1 || 20 * extra > 3 * (plain + nj + nk - 2));
  if (enable == 1 && too_many) return s;
  s.p0 = p0; s.a = a; s.b = b;
  return s;
}

// Range, hyperplane and first k-plane of a workgroup.  MULTI: tab holds one range per blockIdx.z -- the blocks that the
// reference's OpenMP threads sweep independently of each other (each thread calls invlu on its own range and skips the
// neighbour terms across its borders, Fast/FastS/FastS/Compute/lhs.cpp:149-161, LU/invlu_l.for:90-215) -- and n numbers
// the launches of a sweep: the lower sweep of a block runs its hyperplanes upwards from 0, the upper sweep downwards
// from its last one, so that blocks of different sizes all start at launch 0; false when the block has no hyperplane
// for launch n.  Otherwise (one sweep of the range r0) the host passes the chunk count, the hyperplane and the first
// k-plane: these launches are bound by their latency, and working the three out here costs them 0.6 us each.
template <int U, bool UPPER, bool MULTI>
FS_DEV bool lu_block(const FsZone &z, const LuRange &r0, const LuRange *__restrict__ tab, int nchunk0, int n, int kbase0,
                     LuRange &r, LuSkew &sk, int &nchunk, int &h, int &kbase, int &bx, int &by)
{
  bx = blockIdx.x; by = blockIdx.y;
  if (!MULTI) {
    r = r0; sk.p0 = 0; sk.a = 0; sk.b = 0; nchunk = nchunk0; h = n; kbase = kbase0;
    return true;
  }
  // Workgroups go to the 8 XCDs in turn, in the order of their linear index.  All the workgroups of a block are given
  // to one XCD, the same in every launch: what a launch reads of the previous hyperplane of its block (the increments
  // just written, the states and coefficients just read) is then in that XCD's L2 where it is still held.
  int bz = blockIdx.z;
  if ((gridDim.z & 7) == 0 && !(nchunk0 & 4)) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned slot = lin >> 3, nb8 = gridDim.z >> 3;
    bz = (int)((slot % nb8) * 8 + (lin & 7));
    const unsigned w = slot / nb8;
    bx = (int)(w % gridDim.x); by = (int)(w / gridDim.x);
  }
  r = tab[bz];
  const int nj = r.j1 - r.j0 + 1, nk = r.k1 - r.k0 + 1;
  sk = lu_skew<U>(z, r.i0, r.i1, r.j0, r.j1, r.k0, r.k1, nchunk0 & 3);   // with a block table the chunk-count argument carries the switches
  nchunk = lu_nchunk<U>(sk, r.i1 - r.i0 + 1, nj, nk);
  const int nhc = nchunk + nj + nk - 2;
  h = UPPER ? nhc - 1 - n : n;
  if (h < 0 || h >= nhc) return false;
  const int klo = h - (nchunk - 1) - (nj - 1);
  kbase = r.k0 + (klo < 0 ? 0 : klo);
  return true;
}

// Unit and cell of a lane: chunk column cn, line j and cell i of hyperplane h in plane k.  With a skew the chunk columns
// of a plane that hold cells of the hyperplane start at cn_min (those before it lie left of the range or past its
// last line); without, at 0.
template <int U, bool MULTI>
FS_DEV void lu_lane(const LuRange &r, const LuSkew &sk, int h, int k, int bx, int &cn, int &j, int &i, int &m)
{
  const int u = threadIdx.x / U;
  m = threadIdx.x % U;
  const int kp = k - r.k0;
  int cn_min = 0;
  if (MULTI) {
    const int t = sk.p0 + sk.a * (h - kp) + sk.b * kp - U + 1;      // cn (U + a) >= t: the unit reaches the first cell
    if (t > 0) cn_min = (t + U + sk.a - 1) / (U + sk.a);
    const int cj = h - kp - (r.j1 - r.j0);                          // cn >= cj: line j <= j1
    if (cj > cn_min) cn_min = cj;
  }
  cn = cn_min + bx * (64 / U) + u;
  j = r.j0 + h - cn - kp;
  i = r.i0 - (sk.p0 + sk.a * (j - r.j0) + sk.b * kp) + cn * U + m;
}

template <bool FULL, int NE, int U = 8, bool MULTI = false>
__global__ void __launch_bounds__(64) k_lu_lower_wave(FsZone z, LuConst c, LuRange r0, int nchunk0, int n, int kbase0,
                                                      const double *__restrict__ rop, double *__restrict__ dw,
                                                      const double *__restrict__ coe, const LuRange *__restrict__ tab)
{
  LuRange r;
  LuSkew sk;
  int nchunk, h, kbase, bx, by;
  if (!lu_block<U, false, MULTI>(z, r0, tab, nchunk0, n, kbase0, r, sk, nchunk, h, kbase, bx, by)) return;
  // units of a wave: the same k-plane, consecutive chunks of the hyperplane (so consecutive j-lines, descending)
  const int k = kbase + by;
  int cn, j, i, m;
  lu_lane<U, MULTI>(r, sk, h, k, bx, cn, j, i, m);
  const bool active = j >= r.j0 && j <= r.j1 && k <= r.k1 && cn < nchunk && i >= r.i0 && i <= r.i1;
  const long long nd = z.nd;
  const long long l = active ? lcell(z, i, j, k) : lcell(z, r.i0, r.j0, r.k0);
  const long long lt = FULL ? (active ? lmtr(z, i, j, k) : lmtr(z, r.i0, r.j0, r.k0)) : 0;
  const bool has_prev = active && i > r.i0;
  double A[NE] = {}, P[NE] = {}, S[4] = {0., 0., 0., 0.};
  double CR = 0., XAL = 0., DG = 0., C6 = 1.;   // C6 = diagonal of the 6th equation, coe(:,6)
  double MI[26] = {};
  LUSTAMP(0, false) LUSTAMP(1, false)
  if (active) {
    XAL = coe[l] * 0.5;
    double b[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) A[e] = dw[l + e * nd];
    const bool has_j = j > r.j0, has_k = k > r.k0;
    if (FULL) {
      if (has_k) { lu_neighbour<FULL, 2, 1, NE>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL; }
      if (has_j) { lu_neighbour<FULL, 1, 1, NE>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL; }
    } else {
      if (has_j) { lu_neighbour<FULL, 1, 1, NE>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL; }
      if (has_k) { lu_neighbour<FULL, 2, 1, NE>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL; }
    }
    DG = 1. / coe[l + 4 * nd];
    if (NE == 6) C6 = coe[l + 5 * nd];
    if (has_prev) {
      S[0] = rop[l - 1 + nd]; S[1] = rop[l - 1 + 2 * nd]; S[2] = rop[l - 1 + 3 * nd]; S[3] = rop[l - 1 + 4 * nd];
      CR = coe[l - 1 + nd];
      lu_mat<FULL, 0, 1, NE>(z, c, lt, S[0], S[1], S[2], S[3], CR, MI);   // matrix of the i predecessor, once
      if (m == 0) {   // the cell before the unit belongs to an earlier hyperplane: final
#pragma unroll
        for (int e = 0; e < NE; e++) P[e] = dw[l - 1 + e * nd];
      }
    }
  }
  LUSTAMP(2, true)
  // P of lane m holds: before its step, the increment of cell m-1 (lane 0: preloaded); after, its own
  auto cell = [&](const double *pin) {
    if (has_prev) {
      double b[NE];
      lu_mul<FULL, 0, NE>(MI, pin[0], pin[1], pin[2], pin[3], pin[4], b, NE == 6 ? pin[NE - 1] : 0.);
#pragma unroll
      for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL;
    }
#pragma unroll
    for (int e = 0; e < 5; e++) { P[e] = A[e] * DG; }
    if (NE == 6) P[NE - 1] = A[NE - 1] / C6;
  };
  if (m == 0 && active) {   // first cell of a unit: its predecessor belongs to an earlier hyperplane
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = P[e];
    cell(pin);
  }
  for (int st = 1; st < U; st++) {
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = lane_prev(P[e]);
    if (m == st && active) cell(pin);
  }
  LUSTAMP(3, false)
  if (active) {
#pragma unroll
    for (int e = 0; e < NE; e++) dw[l + e * nd] = P[e];
  }
  LUSTAMP(4, true) LUSTAMP(7, false)
}

template <bool FULL, int NE, int U = 8, bool MULTI = false>
__global__ void __launch_bounds__(64) k_lu_upper_wave(FsZone z, LuConst c, LuRange r0, int nchunk0, int n, int kbase0,
                                                      const double *__restrict__ rop, const double *rop_kn,
                                                      double *rop_new, double *__restrict__ dw,
                                                      const double *__restrict__ coe, const LuRange *__restrict__ tab)
{
  LuRange r;
  LuSkew sk;
  int nchunk, h, kbase, bx, by;
  if (!lu_block<U, true, MULTI>(z, r0, tab, nchunk0, n, kbase0, r, sk, nchunk, h, kbase, bx, by)) return;
  // units of a wave: the same k-plane, consecutive chunks of the hyperplane (so consecutive j-lines, descending)
  const int k = kbase + by;
  int cn, j, i, m;
  lu_lane<U, MULTI>(r, sk, h, k, bx, cn, j, i, m);
  const bool active = j >= r.j0 && j <= r.j1 && k <= r.k1 && cn < nchunk && i >= r.i0 && i <= r.i1;
  const long long nd = z.nd;
  const long long l = active ? lcell(z, i, j, k) : lcell(z, r.i0, r.j0, r.k0);
  const long long lt = FULL ? (active ? lmtr(z, i, j, k) : lmtr(z, r.i0, r.j0, r.k0)) : 0;
  const bool has_next = active && i < r.i1;
  const bool last_of_unit = m == U - 1 || i == r.i1;     // its i+1 neighbour lies outside the unit
  double A[NE] = {}, P[NE] = {}, S[4] = {0., 0., 0., 0.};
  double CR = 0., XAL = 0., DG = 0., C6 = 1.;   // C6 = diagonal of the 6th equation, coe(:,6)
  double MI[26] = {};
  double q[NE] = {1., 0., 0., 0., 1.};
  if (active) {
    XAL = coe[l] * -0.5;
    const double c5 = coe[l + 4 * nd];
    double b[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) { A[e] = dw[l + e * nd] * c5; q[e] = rop[l + e * nd]; }
    if (NE == 6) { C6 = coe[l + 5 * nd]; A[NE - 1] = dw[l + 5 * nd] * C6; }   // LU/lu_d_SA.for
    if (k < r.k1) { lu_neighbour<FULL, 2, -1, NE>(z, c, l, lt, rop_kn, dw, coe, b); for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL; }
    if (j < r.j1) { lu_neighbour<FULL, 1, -1, NE>(z, c, l, lt, rop, dw, coe, b); for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL; }
    DG = 1. / c5;
    if (has_next) {
      S[0] = rop[l + 1 + nd]; S[1] = rop[l + 1 + 2 * nd]; S[2] = rop[l + 1 + 3 * nd]; S[3] = rop[l + 1 + 4 * nd];
      CR = coe[l + 1 + nd];
      lu_mat<FULL, 0, -1, NE>(z, c, lt, S[0], S[1], S[2], S[3], CR, MI);   // matrix of the i predecessor, once
      if (last_of_unit) {
#pragma unroll
        for (int e = 0; e < NE; e++) P[e] = dw[l + 1 + e * nd];
      }
    }
  }
  // a cell whose successor lies in the next unit has it preloaded in P; the last cell of the range has none
  auto cell = [&](const double *pin) {
    if (has_next) {
      double b[NE];
      lu_mul<FULL, 0, NE>(MI, pin[0], pin[1], pin[2], pin[3], pin[4], b, NE == 6 ? pin[NE - 1] : 0.);
#pragma unroll
      for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL;
    }
#pragma unroll
    for (int e = 0; e < 5; e++) { P[e] = A[e] * DG; }
    if (NE == 6) P[NE - 1] = A[NE - 1] / C6;
  };
  if (m == U - 1 && active) {
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = P[e];
    cell(pin);
  }
  for (int st = U - 2; st >= 0; st--) {
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = lane_next(P[e]);
    if (m == st && active) cell(pin);
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < NE; e++) dw[l + e * nd] = P[e];
    // Newton update (mjr_newton.for)
    const double ro_old = q[0], u_old = q[1], v_old = q[2], w_old = q[3], t_old = q[4];
    const double rn = ro_old + P[0];
    const double r_1 = 1. / rn;
    const double un = (ro_old * u_old + P[1]) * r_1;
    const double vn = (ro_old * v_old + P[2]) * r_1;
    const double wn = (ro_old * w_old + P[3]) * r_1;
    const double roe_old = ro_old * (c.cv * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
    rop_new[l] = rn; rop_new[l + nd] = un; rop_new[l + 2 * nd] = vn; rop_new[l + 3 * nd] = wn;
    const double tn = (roe_old + P[4]) * r_1 * c.cvinv - c.cvinv2 * (un * un + vn * vn + wn * wn);
    rop_new[l + 4 * nd] = tn;
    if (NE == 6) {   // LU/mjr_newton_SA.for: nu_tilde clipped to [0, ratiom * nu]
      double anulam = fs_mulam_sa(z, tn);
      anulam = anulam * r_1 * z.ratiom;
      double t6 = (ro_old * q[NE - 1] + P[NE - 1]) * r_1;
      t6 = fs_min(t6, anulam);
      rop_new[l + 5 * nd] = fs_max(t6, 0.);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// LU-SSOR, nb_relax >= 2 relaxations (LU/invlu.for:195-340, invlussor1_l.for, invlussor_l.for, invlussor_u.for),
// 5 equations, three-dimensional zones.  Same hyperplanes and the same lane layout as the sweeps above.  Per cell
// and sweep the neighbour terms are added in the order i, j, k to the increment (din = right-hand side, kept; dout
// = increments) and to the accumulator ssor (mjr_drodm_IO_ssor.for, mjr_drodm_ssor.for), then the increment is
// divided by the diagonal (lu_dinv.for).  FIRSTM: first lower sweep, the increment starts from din.  Between two
// sweeps the reference puts din + ssor into the increment and zeroes ssor (mjr_ssor.for, one plane behind the sweep,
// when every successor of a cell has read its increment); a sweep reads the increments of cells it has already
// visited only, so here the next sweep forms din + ssor when it reaches the cell and starts its accumulator at zero:
// the same sums, without a pass over the zone between two sweeps (0.6 ms each at 256^3).
// LAST: last upper sweep, followed by the Newton update of the cells inside sdm (mjr_newton.for).
// ------------------------------------------------------------------------------------------------------
template <bool FULL, bool FIRSTM, int NE = 5, int U = 8, bool MULTI = false>
__global__ void __launch_bounds__(64) k_ssor_lower_wave(FsZone z, LuConst c, LuRange r0, int nchunk0, int n, int kbase0,
                                                        const double *__restrict__ rop, const double *__restrict__ din,
                                                        double *__restrict__ dout, double *__restrict__ ssor,
                                                        const double *__restrict__ coe, const LuRange *__restrict__ tab)
{
  LuRange r;
  LuSkew sk;
  int nchunk, h, kbase, bx, by;
  if (!lu_block<U, false, MULTI>(z, r0, tab, nchunk0, n, kbase0, r, sk, nchunk, h, kbase, bx, by)) return;
  // units of a wave: the same k-plane, consecutive chunks of the hyperplane (so consecutive j-lines, descending)
  const int k = kbase + by;
  int cn, j, i, m;
  lu_lane<U, MULTI>(r, sk, h, k, bx, cn, j, i, m);
  const bool active = j >= r.j0 && j <= r.j1 && k <= r.k1 && cn < nchunk && i >= r.i0 && i <= r.i1;
  const long long nd = z.nd;
  const long long l = active ? lcell(z, i, j, k) : lcell(z, r.i0, r.j0, r.k0);
  const long long lt = FULL ? (active ? lmtr(z, i, j, k) : lmtr(z, r.i0, r.j0, r.k0)) : 0;
  const bool has_prev = active && i > r.i0;
  const bool has_j = active && j > r.j0, has_k = active && k > r.k0;
  double A[NE] = {}, S[NE] = {}, P[NE] = {}, BJ[NE] = {}, BK[NE] = {}, ST[4] = {0., 0., 0., 0.};
  double CR = 0., XAL = 0., DG = 0., C6 = 1.;   // C6 = diagonal of the 6th equation, coe(:,6)
  double MI[26] = {};
  if (active) {
    XAL = coe[l] * 0.5;
    double b[NE];
#pragma unroll
    // later sweeps start from the right-hand side plus what the previous sweep accumulated, with an empty accumulator
    // (mjr_ssor.for, which the reference applies one plane behind the previous sweep; nothing reads a cell in between)
    for (int e = 0; e < NE; e++) { A[e] = FIRSTM ? din[l + e * nd] : din[l + e * nd] + ssor[l + e * nd]; S[e] = 0.; }
    if (has_j) { lu_neighbour<FULL, 1, 1, NE>(z, c, l, lt, rop, dout, coe, b); for (int e = 0; e < NE; e++) BJ[e] = b[e] * XAL; }
    if (has_k) { lu_neighbour<FULL, 2, 1, NE>(z, c, l, lt, rop, dout, coe, b); for (int e = 0; e < NE; e++) BK[e] = b[e] * XAL; }
    DG = 1. / coe[l + 4 * nd];
    if (NE == 6) C6 = coe[l + 5 * nd];
    if (has_prev) {
      ST[0] = rop[l - 1 + nd]; ST[1] = rop[l - 1 + 2 * nd]; ST[2] = rop[l - 1 + 3 * nd]; ST[3] = rop[l - 1 + 4 * nd];
      CR = coe[l - 1 + nd];
      lu_mat<FULL, 0, 1, NE>(z, c, lt, ST[0], ST[1], ST[2], ST[3], CR, MI);   // matrix of the i predecessor, once
      if (m == 0) {
#pragma unroll
        for (int e = 0; e < NE; e++) P[e] = dout[l - 1 + e * nd];
      }
    }
  }
  auto cell = [&](const double *pin) {
    bool any = false;
    if (has_prev) {
      double b[NE];
      lu_mul<FULL, 0, NE>(MI, pin[0], pin[1], pin[2], pin[3], pin[4], b, NE == 6 ? pin[NE - 1] : 0.);
#pragma unroll
      for (int e = 0; e < NE; e++) { const double t = b[e] * XAL; A[e] = A[e] + t; S[e] = FIRSTM ? t : S[e] + t; }
      any = true;
    }
    if (has_j) {
#pragma unroll
      for (int e = 0; e < NE; e++) { A[e] = A[e] + BJ[e]; S[e] = (FIRSTM && !any) ? BJ[e] : S[e] + BJ[e]; }
      any = true;
    }
    if (has_k) {
#pragma unroll
      for (int e = 0; e < NE; e++) { A[e] = A[e] + BK[e]; S[e] = (FIRSTM && !any) ? BK[e] : S[e] + BK[e]; }
    }
#pragma unroll
    for (int e = 0; e < 5; e++) P[e] = A[e] * DG;
    if (NE == 6) P[NE - 1] = A[NE - 1] / C6;   // LU/lu_dinv_SA.for
  };
  if (m == 0 && active) {   // first cell of a unit: its predecessor belongs to an earlier hyperplane
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = P[e];
    cell(pin);
  }
  for (int st = 1; st < U; st++) {
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = lane_prev(P[e]);
    if (m == st && active) cell(pin);
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < NE; e++) { dout[l + e * nd] = P[e]; ssor[l + e * nd] = S[e]; }
  }
}

// LUD: the upper sweep of LU-SGS with lu_match = 1 (LU/invlu_overlap_u.for): the increment of the lower sweep is
// multiplied by the diagonal first (lu_d.for), the terms follow in the order i, j, k, no accumulator.
template <bool FULL, bool LAST, bool LUD = false, int NE = 5, int U = 8, bool MULTI = false>
__global__ void __launch_bounds__(64) k_ssor_upper_wave(FsZone z, LuConst c, LuRange r0, LuRange sdm, int nchunk0, int n, int kbase0,
                                                        const double *__restrict__ rop, const double *rop_kn, double *rop_new,
                                                        const double *__restrict__ din, double *__restrict__ dout,
                                                        double *__restrict__ ssor, const double *__restrict__ coe,
                                                        const LuRange *__restrict__ tab)
{
  LuRange r;
  LuSkew sk;
  int nchunk, h, kbase, bx, by;
  if (!lu_block<U, true, MULTI>(z, r0, tab, nchunk0, n, kbase0, r, sk, nchunk, h, kbase, bx, by)) return;
  // units of a wave: the same k-plane, consecutive chunks of the hyperplane (so consecutive j-lines, descending)
  const int k = kbase + by;
  int cn, j, i, m;
  lu_lane<U, MULTI>(r, sk, h, k, bx, cn, j, i, m);
  const bool active = j >= r.j0 && j <= r.j1 && k <= r.k1 && cn < nchunk && i >= r.i0 && i <= r.i1;
  const long long nd = z.nd;
  const long long l = active ? lcell(z, i, j, k) : lcell(z, r.i0, r.j0, r.k0);
  const long long lt = FULL ? (active ? lmtr(z, i, j, k) : lmtr(z, r.i0, r.j0, r.k0)) : 0;
  const bool has_next = active && i < r.i1;
  const bool has_j = active && j < r.j1, has_k = active && k < r.k1;
  const bool last_of_unit = m == U - 1 || i == r.i1;
  double A[NE] = {}, S[NE] = {}, P[NE] = {}, BJ[NE] = {}, BK[NE] = {}, ST[4] = {0., 0., 0., 0.};
  double CR = 0., XAL = 0., DG = 0., C6 = 1.;
  double MI[26] = {};
  double q[NE] = {1., 0., 0., 0., 1.};
  if (active) {
    XAL = coe[l] * -0.5;
    double b[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) { A[e] = LUD ? dout[l + e * nd] : din[l + e * nd] + ssor[l + e * nd]; S[e] = 0.; q[e] = rop[l + e * nd]; }
    if (NE == 6) C6 = coe[l + 5 * nd];
    if (LUD) {
      const double c5 = coe[l + 4 * nd];
#pragma unroll
      for (int e = 0; e < 5; e++) A[e] = A[e] * c5;
      if (NE == 6) A[NE - 1] = A[NE - 1] * C6;   // LU/lu_d_SA.for
    }
    if (has_j) { lu_neighbour<FULL, 1, -1, NE>(z, c, l, lt, rop, dout, coe, b); for (int e = 0; e < NE; e++) BJ[e] = b[e] * XAL; }
    if (has_k) { lu_neighbour<FULL, 2, -1, NE>(z, c, l, lt, LAST ? rop_kn : rop, dout, coe, b); for (int e = 0; e < NE; e++) BK[e] = b[e] * XAL; }
    DG = 1. / coe[l + 4 * nd];
    if (has_next) {
      ST[0] = rop[l + 1 + nd]; ST[1] = rop[l + 1 + 2 * nd]; ST[2] = rop[l + 1 + 3 * nd]; ST[3] = rop[l + 1 + 4 * nd];
      CR = coe[l + 1 + nd];
      lu_mat<FULL, 0, -1, NE>(z, c, lt, ST[0], ST[1], ST[2], ST[3], CR, MI);   // matrix of the i predecessor, once
      if (last_of_unit) {
#pragma unroll
        for (int e = 0; e < NE; e++) P[e] = dout[l + 1 + e * nd];
      }
    }
  }
  // a cell whose successor lies in the next unit has it preloaded in P; the last cell of the range has none
  auto cell = [&](const double *pin) {
    if (has_next) {
      double b[NE];
      lu_mul<FULL, 0, NE>(MI, pin[0], pin[1], pin[2], pin[3], pin[4], b, NE == 6 ? pin[NE - 1] : 0.);
#pragma unroll
      for (int e = 0; e < NE; e++) { const double t = b[e] * XAL; A[e] = A[e] + t; S[e] = S[e] + t; }
    }
    if (has_j) {
#pragma unroll
      for (int e = 0; e < NE; e++) { A[e] = A[e] + BJ[e]; S[e] = S[e] + BJ[e]; }
    }
    if (has_k) {
#pragma unroll
      for (int e = 0; e < NE; e++) { A[e] = A[e] + BK[e]; S[e] = S[e] + BK[e]; }
    }
#pragma unroll
    for (int e = 0; e < 5; e++) P[e] = A[e] * DG;
    if (NE == 6) P[NE - 1] = A[NE - 1] / C6;
  };
  if (m == U - 1 && active) {
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = P[e];
    cell(pin);
  }
  for (int st = U - 2; st >= 0; st--) {
    double pin[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) pin[e] = lane_next(P[e]);
    if (m == st && active) cell(pin);
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < NE; e++) { dout[l + e * nd] = P[e]; if (!LUD) ssor[l + e * nd] = S[e]; }
    if (LAST && i >= sdm.i0 && i <= sdm.i1 && j >= sdm.j0 && j <= sdm.j1 && k >= sdm.k0 && k <= sdm.k1) {
      const double ro_old = q[0], u_old = q[1], v_old = q[2], w_old = q[3], t_old = q[4];
      const double rn = ro_old + P[0];
      const double r_1 = 1. / rn;
      const double un = (ro_old * u_old + P[1]) * r_1;
      const double vn = (ro_old * v_old + P[2]) * r_1;
      const double wn = (ro_old * w_old + P[3]) * r_1;
      const double roe_old = ro_old * (c.cv * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
      rop_new[l] = rn; rop_new[l + nd] = un; rop_new[l + 2 * nd] = vn; rop_new[l + 3 * nd] = wn;
      const double tn = (roe_old + P[4]) * r_1 * c.cvinv - c.cvinv2 * (un * un + vn * vn + wn * wn);
      rop_new[l + 4 * nd] = tn;
      if (NE == 6) {   // LU/mjr_newton_SA.for: nu_tilde clipped to [0, ratiom * nu]
        double anulam = fs_mulam_sa(z, tn);
        anulam = anulam * r_1 * z.ratiom;
        double t6 = (ro_old * q[NE - 1] + P[NE - 1]) * r_1;
        t6 = fs_min(t6, anulam);
        rop_new[l + 5 * nd] = fs_max(t6, 0.);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Dataflow form of the two LU-SGS sweeps: ONE launch per sweep instead of one per hyperplane.  A task is a pencil of 8
// consecutive j-lines of one k-plane; a wave takes a task and walks along i in steps: at step s the unit (8 cells) of its
// line number q (in sweep order) is chunk s - q of that line, lane = (line, cell).  The three predecessors of that unit
// are then: the chunk before it on its line and the same chunk of the line before -- both worked by this wave one step
// earlier -- and the same chunk of the line in the plane before, worked at step s by the pencil (jb, k -+ 1).  The first
// line of a pencil follows the last line of pencil jb -+ 1, which reaches that chunk at its step s + 7.  Every wave
// publishes the number of steps it has finished in a word of its task; a wave polls the words of its two predecessor
// pencils before it reads their increments.  Tasks are handed out in the order of their first possible step by an atomic
// counter, so whatever the number of waves the chip runs at once, the predecessors of a task are finished or held by a
// running wave.  Per cell the operations are those of the per-hyperplane kernels above, in the same order.
// Visibility between waves of one launch (the L2 caches of the 8 XCDs are private, a CU's L1 is never refreshed): every
// increment (and, in the upper sweep, every new state) is stored write-through (sc1), the storing wave waits for its
// stores (vmcnt(0)) before one lane publishes the step count by an sc1 store; the reader polls that word by sc1 loads and
// reads increments and new states by sc1 loads only (they bypass the L1).  Everything else the sweeps read is written
// by no wave of the launch.
// ------------------------------------------------------------------------------------------------------
struct FlowTask { int jb, k, fid, b; };      // group of lines, k-plane, index of the progress word, range swept
struct FlowBox { LuRange r; int njb, pad; }; // a range swept on its own (the zone, a thread block, a sub-zone box) and its pencils per plane

FS_DEV double ld_sc1(const double *p)
{
  return __hip_atomic_load(const_cast<double *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FS_DEV void st_sc1(double *p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

enum { FLOW_COUNTER = 0, FLOW_ABORT = 1, FLOW_FLAGS = 16 };
constexpr unsigned FLOW_SPIN_LIMIT = 1u << 24;   // polls of about a microsecond each: some 20 s

// Lane 0 waits for *fa >= na, lane 1 for *fb >= nb (0: nothing to wait for).  When the first poll has landed every
// older store of the wave has too: the step count `done` of the wave's own task is published then (0: nothing to
// publish).  false: gave up (a predecessor never came, or another wave gave up): the caller leaves the kernel.
FS_DEV bool flow_wait(const unsigned *fa, unsigned na, const unsigned *fb, unsigned nb, unsigned *mine, unsigned done,
                      unsigned *sync, unsigned *sticky)
{
  const int lane = threadIdx.x;
  const unsigned *p = lane == 0 ? fa : fb;
  const unsigned need = lane == 0 ? na : (lane == 1 ? nb : 0u);
  bool ok = need == 0u;
  if (!ok) ok = __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (done && lane == 0) __hip_atomic_store(mine, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__all(ok)) return true;
  bool alive = true;
  for (unsigned spins = 1; alive; spins++) {
    __builtin_amdgcn_s_sleep(1);
    if (!ok) ok = __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
    if (__all(ok)) break;
    bool stop = spins > FLOW_SPIN_LIMIT;
    if ((spins & 255u) == 255u && lane == 0)
      stop = stop || __hip_atomic_load(sync + FLOW_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (__any(stop)) alive = false;
  }
  if (!alive && lane == 0) {
    __hip_atomic_store(sync + FLOW_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return alive;
}

// K3D = false: two-dimensional zone, no plane before: the k term is not formed
template <bool FULL, int NE, int U = 8, bool K3D = true>
__global__ void __launch_bounds__(64) k_lu_lower_flow(FsZone z, LuConst c, const FlowBox *__restrict__ boxes,
                                                      const FlowTask *__restrict__ tasks, int ntasks, unsigned *sync,
                                                      unsigned *sticky, const double *__restrict__ rop, double *dw,
                                                      const double *__restrict__ coe)
{
  constexpr int LPP = 64 / U;                       // lines of a pencil
  const int lane = threadIdx.x, jj = lane / U, m = lane % U;
  unsigned *flags = sync + FLOW_FLAGS;
  const long long nd = z.nd;
  // Every lane adds one to the counter (the compiler turns that into one add of 64 by one lane): a task index that
  // depended on which lanes are active at this point would not survive the restructuring of the loops around it.
  for (bool alive = true; alive;) {
    const int n = (int)(__builtin_amdgcn_readfirstlane(
                      __hip_atomic_fetch_add(sync + FLOW_COUNTER, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 6);
    if (n >= ntasks) break;
    const FlowTask t = tasks[n];
    const LuRange r = boxes[t.b].r;
    const int njb = boxes[t.b].njb;
    const int ni = r.i1 - r.i0 + 1, nchunk = (ni + U - 1) / U, nsteps = nchunk + LPP - 1;
    const long long lsafe = lcell(z, r.i0, r.j0, r.k0), ltsafe = FULL ? lmtr(z, r.i0, r.j0, r.k0) : 0;
    const int j = r.j0 + LPP * t.jb + jj, k = t.k;
    const bool line_ok = j <= r.j1;
    const bool has_j = j > r.j0, has_k = K3D && k > r.k0;
    const unsigned *fk = flags + t.fid - njb, *fj = flags + t.fid - 1;
    unsigned *mine = flags + t.fid;
    const long long lrow = lcell(z, r.i0, line_ok ? j : r.j0, k) + m, ltrow = FULL ? lmtr(z, r.i0, line_ok ? j : r.j0, k) + m : 0;
    const long long dj_ = has_j ? z.incj : 0, dk_ = has_k ? z.inck : 0;   // a cell without a neighbour reads itself, and drops the term
    // cell of this lane at step s: lanes without one read the first cell of the range and store nothing
    auto geom = [&](int s, bool &active, bool &has_prev, long long &l, long long &lt) {
      const int cn = s - jj, di = cn * U;
      active = line_ok && cn >= 0 && cn < nchunk && di + m < ni;
      has_prev = active && di + m > 0;
      l = active ? lrow + di : lsafe;
      lt = FULL ? (active ? ltrow + di : ltsafe) : 0;
    };
    // what a step reads that no wave of the launch writes, loaded one step ahead
    double sA[NE], sxal, sc5, sc6 = 1., sk[4] = {0., 0., 0., 0.}, ck = 0., sj[4], cj, si[4], ci;
    auto load_stat = [&](int s) {
      bool active, has_prev; long long l, lt;
      geom(s, active, has_prev, l, lt);
      const long long lj = l - dj_, lk = l - dk_, li = has_prev ? l - 1 : l;
      sxal = coe[l]; sc5 = coe[l + 4 * nd];
      if (NE == 6) sc6 = coe[l + 5 * nd];
#pragma unroll
      for (int e = 0; e < NE; e++) sA[e] = dw[l + e * nd];
#pragma unroll
      for (int e = 0; e < 4; e++) { if (K3D) sk[e] = rop[lk + (e + 1) * nd]; sj[e] = rop[lj + (e + 1) * nd]; si[e] = rop[li + (e + 1) * nd]; }
      if (K3D) ck = coe[lk + 3 * nd];
      cj = coe[lj + 2 * nd]; ci = coe[li + nd];
    };
    load_stat(0);
    for (int s = 0; s <= nsteps && alive; s++) {   // pass nsteps publishes the last count only
      bool active, has_prev; long long l, lt;
      geom(s, active, has_prev, l, lt);
      const long long lj = l - dj_, lk = l - dk_, li = (has_prev && m == 0) ? l - 1 : l;
      // the plane before has finished this step; the pencil before has finished the chunk its last line hands over
      const unsigned need_k = (K3D && k > r.k0 && s < nsteps) ? (unsigned)(s + 1) : 0u;
      const unsigned need_j = (t.jb > 0 && s < nchunk) ? (unsigned)(s + LPP) : 0u;
      FLOWSTAMP(0)
      alive = flow_wait(fk, need_k, fj, need_j, mine, (unsigned)s, sync, sticky);
      FLOWSTAMP(1)
      if (!alive || s == nsteps) continue;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double A[NE], P[NE], dk[NE], dj[NE];
#pragma unroll
      for (int e = 0; e < NE; e++) { if (K3D) dk[e] = ld_sc1(dw + lk + e * nd); dj[e] = ld_sc1(dw + lj + e * nd); P[e] = ld_sc1(dw + li + e * nd); }
      // the matrices of the three neighbours while the increments are on their way, then the loads of the next step
      double MI[26], MJ[26], MK[26];
      const double XAL = sxal * 0.5, DG = 1. / sc5, C6 = sc6;
#pragma unroll
      for (int e = 0; e < NE; e++) A[e] = sA[e];
      if (K3D) lu_mat<FULL, 2, 1, NE>(z, c, lt, sk[0], sk[1], sk[2], sk[3], ck, MK);
      lu_mat<FULL, 1, 1, NE>(z, c, lt, sj[0], sj[1], sj[2], sj[3], cj, MJ);
      lu_mat<FULL, 0, 1, NE>(z, c, lt, si[0], si[1], si[2], si[3], ci, MI);
      if (s + 1 < nsteps) load_stat(s + 1);
      {
        double bk[NE], bj[NE];
        if (K3D) lu_mul<FULL, 2, NE>(MK, dk[0], dk[1], dk[2], dk[3], dk[4], bk, NE == 6 ? dk[NE - 1] : 0.);
        lu_mul<FULL, 1, NE>(MJ, dj[0], dj[1], dj[2], dj[3], dj[4], bj, NE == 6 ? dj[NE - 1] : 0.);
        if (FULL) {
          if (K3D && has_k) { for (int e = 0; e < NE; e++) A[e] = A[e] + bk[e] * XAL; }
          if (has_j) { for (int e = 0; e < NE; e++) A[e] = A[e] + bj[e] * XAL; }
        } else {
          if (has_j) { for (int e = 0; e < NE; e++) A[e] = A[e] + bj[e] * XAL; }
          if (K3D && has_k) { for (int e = 0; e < NE; e++) A[e] = A[e] + bk[e] * XAL; }
        }
      }
      FLOWSTAMP(2)
      auto cell = [&](const double *pin) {
        if (has_prev) {
          double b[NE];
          lu_mul<FULL, 0, NE>(MI, pin[0], pin[1], pin[2], pin[3], pin[4], b, NE == 6 ? pin[NE - 1] : 0.);
#pragma unroll
          for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL;
        }
#pragma unroll
        for (int e = 0; e < 5; e++) { P[e] = A[e] * DG; }
        if (NE == 6) P[NE - 1] = A[NE - 1] / C6;
      };
      if (m == 0 && active) {
        double pin[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) pin[e] = P[e];
        cell(pin);
      }
#pragma unroll
      for (int st = 1; st < U; st++) {
        double pin[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) pin[e] = lane_prev(P[e]);
        if (m == st && active) cell(pin);
      }
      FLOWSTAMP(3)
      if (active) {
#pragma unroll
        for (int e = 0; e < NE; e++) st_sc1(dw + l + e * nd, P[e]);
      }
      FLOWSTAMP(4)
    }
  }
}

// Upper sweep and Newton update: the pencils in the opposite order, the lines of a pencil downwards, the chunks of a
// line from the last to the first.  The state of the k + 1 neighbour is the new one when the update is in place
// (rop_kn = the array the new iterate is built in): handed over like the increments.
template <bool FULL, int NE, int U = 8, bool K3D = true>
__global__ void __launch_bounds__(64) k_lu_upper_flow(FsZone z, LuConst c, const FlowBox *__restrict__ boxes,
                                                      const FlowTask *__restrict__ tasks, int ntasks, unsigned *sync,
                                                      unsigned *sticky, const double *__restrict__ rop, const double *rop_kn,
                                                      double *rop_new, double *dw, const double *__restrict__ coe)
{
  constexpr int LPP = 64 / U;
  const int lane = threadIdx.x, jj = lane / U, m = lane % U, q7 = LPP - 1 - jj;
  unsigned *flags = sync + FLOW_FLAGS;
  const long long nd = z.nd;
  for (bool alive = true; alive;) {
    const int n = (int)(__builtin_amdgcn_readfirstlane(
                      __hip_atomic_fetch_add(sync + FLOW_COUNTER, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 6);
    if (n >= ntasks) break;
    const FlowTask t = tasks[ntasks - 1 - n];
    const LuRange r = boxes[t.b].r;
    const int njb = boxes[t.b].njb;
    const int ni = r.i1 - r.i0 + 1, nchunk = (ni + U - 1) / U, nsteps = nchunk + LPP - 1;
    const long long lsafe = lcell(z, r.i0, r.j0, r.k0), ltsafe = FULL ? lmtr(z, r.i0, r.j0, r.k0) : 0;
    const int j = r.j0 + LPP * t.jb + jj, k = t.k;
    const bool line_ok = j <= r.j1;
    const bool has_j = j < r.j1, has_k = K3D && k < r.k1;
    const unsigned *fk = flags + t.fid + njb, *fj = flags + t.fid + 1;
    unsigned *mine = flags + t.fid;
    const long long lrow = lcell(z, r.i0, line_ok ? j : r.j0, k) + m, ltrow = FULL ? lmtr(z, r.i0, line_ok ? j : r.j0, k) + m : 0;
    const long long dj_ = has_j ? z.incj : 0, dk_ = has_k ? z.inck : 0;
    auto geom = [&](int s, bool &active, bool &has_next, long long &l, long long &lt) {
      const int cn = nchunk - 1 - (s - q7), di = cn * U;
      active = line_ok && cn >= 0 && cn < nchunk && di + m < ni;
      has_next = active && di + m < ni - 1;
      l = active ? lrow + di : lsafe;
      lt = FULL ? (active ? ltrow + di : ltsafe) : 0;
    };
    double sA[NE], sq[NE], sxal, sc5, sc6 = 1., ck = 0., sj[4], cj, si[4], ci;
    auto load_stat = [&](int s) {
      bool active, has_next; long long l, lt;
      geom(s, active, has_next, l, lt);
      const long long lj = l + dj_, lk = l + dk_, li = has_next ? l + 1 : l;
      sxal = coe[l]; sc5 = coe[l + 4 * nd];
      if (NE == 6) sc6 = coe[l + 5 * nd];
#pragma unroll
      for (int e = 0; e < NE; e++) { sA[e] = dw[l + e * nd]; sq[e] = rop[l + e * nd]; }
#pragma unroll
      for (int e = 0; e < 4; e++) { sj[e] = rop[lj + (e + 1) * nd]; si[e] = rop[li + (e + 1) * nd]; }
      if (K3D) ck = coe[lk + 3 * nd];
      cj = coe[lj + 2 * nd]; ci = coe[li + nd];
    };
    load_stat(0);
    for (int s = 0; s <= nsteps && alive; s++) {   // pass nsteps publishes the last count only
      bool active, has_next; long long l, lt;
      geom(s, active, has_next, l, lt);
      const bool last_of_unit = m == U - 1 || !has_next;
      const long long lj = l + dj_, lk = l + dk_, li = (has_next && last_of_unit) ? l + 1 : l;
      const unsigned need_k = (K3D && k < r.k1 && s < nsteps) ? (unsigned)(s + 1) : 0u;
      const unsigned need_j = (t.jb < njb - 1 && s < nchunk) ? (unsigned)(s + LPP) : 0u;
      alive = flow_wait(fk, need_k, fj, need_j, mine, (unsigned)s, sync, sticky);
      if (!alive || s == nsteps) continue;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double A[NE], P[NE], dk[NE], dj[NE], sk[4], q[NE];
#pragma unroll
      for (int e = 0; e < 4; e++) sk[e] = K3D ? ld_sc1(rop_kn + lk + (e + 1) * nd) : 0.;
#pragma unroll
      for (int e = 0; e < NE; e++) { if (K3D) dk[e] = ld_sc1(dw + lk + e * nd); dj[e] = ld_sc1(dw + lj + e * nd); P[e] = ld_sc1(dw + li + e * nd); }
      double MI[26], MJ[26], MK[26];
      const double XAL = sxal * -0.5, c5 = sc5, DG = 1. / c5, C6 = sc6, ckk = ck;
#pragma unroll
      for (int e = 0; e < NE; e++) { A[e] = sA[e] * c5; q[e] = sq[e]; }
      if (NE == 6) A[NE - 1] = sA[NE - 1] * C6;   // LU/lu_d_SA.for
      lu_mat<FULL, 1, -1, NE>(z, c, lt, sj[0], sj[1], sj[2], sj[3], cj, MJ);
      lu_mat<FULL, 0, -1, NE>(z, c, lt, si[0], si[1], si[2], si[3], ci, MI);
      if (s + 1 < nsteps) load_stat(s + 1);
      {
        double bk[NE], bj[NE];
        if (K3D) {
          lu_mat<FULL, 2, -1, NE>(z, c, lt, sk[0], sk[1], sk[2], sk[3], ckk, MK);
          lu_mul<FULL, 2, NE>(MK, dk[0], dk[1], dk[2], dk[3], dk[4], bk, NE == 6 ? dk[NE - 1] : 0.);
        }
        lu_mul<FULL, 1, NE>(MJ, dj[0], dj[1], dj[2], dj[3], dj[4], bj, NE == 6 ? dj[NE - 1] : 0.);
        if (K3D && has_k) { for (int e = 0; e < NE; e++) A[e] = A[e] + bk[e] * XAL; }
        if (has_j) { for (int e = 0; e < NE; e++) A[e] = A[e] + bj[e] * XAL; }
      }
      auto cell = [&](const double *pin) {
        if (has_next) {
          double b[NE];
          lu_mul<FULL, 0, NE>(MI, pin[0], pin[1], pin[2], pin[3], pin[4], b, NE == 6 ? pin[NE - 1] : 0.);
#pragma unroll
          for (int e = 0; e < NE; e++) A[e] = A[e] + b[e] * XAL;
        }
#pragma unroll
        for (int e = 0; e < 5; e++) { P[e] = A[e] * DG; }
        if (NE == 6) P[NE - 1] = A[NE - 1] / C6;
      };
      if (m == U - 1 && active) {
        double pin[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) pin[e] = P[e];
        cell(pin);
      }
#pragma unroll
      for (int st = U - 2; st >= 0; st--) {
        double pin[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) pin[e] = lane_next(P[e]);
        if (m == st && active) cell(pin);
      }
      if (active) {
#pragma unroll
        for (int e = 0; e < NE; e++) st_sc1(dw + l + e * nd, P[e]);
        // Newton update (mjr_newton.for)
        const double ro_old = q[0], u_old = q[1], v_old = q[2], w_old = q[3], t_old = q[4];
        const double rn = ro_old + P[0];
        const double r_1 = 1. / rn;
        const double un = (ro_old * u_old + P[1]) * r_1;
        const double vn = (ro_old * v_old + P[2]) * r_1;
        const double wn = (ro_old * w_old + P[3]) * r_1;
        const double roe_old = ro_old * (c.cv * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
        st_sc1(rop_new + l, rn); st_sc1(rop_new + l + nd, un); st_sc1(rop_new + l + 2 * nd, vn); st_sc1(rop_new + l + 3 * nd, wn);
        const double tn = (roe_old + P[4]) * r_1 * c.cvinv - c.cvinv2 * (un * un + vn * vn + wn * wn);
        st_sc1(rop_new + l + 4 * nd, tn);
        if (NE == 6) {   // LU/mjr_newton_SA.for: nu_tilde clipped to [0, ratiom * nu]
          double anulam = fs_mulam_sa(z, tn);
          anulam = anulam * r_1 * z.ratiom;
          double t6 = (ro_old * q[NE - 1] + P[NE - 1]) * r_1;
          t6 = fs_min(t6, anulam);
          st_sc1(rop_new + l + 5 * nd, fs_max(t6, 0.));
        }
      }
    }
  }
}

// LU-SSOR sweeps (nb_relax >= 2) and LU-SGS with lu_match = 1 as one launch per sweep: the pencils, step counts and
// hand-overs of the two kernels above with the cell operations of k_ssor_lower_wave / k_ssor_upper_wave (neighbour terms
// in the order i, j, k inside the recurrence, accumulator ssor).  The increments dout are what pencils hand to each
// other; din, ssor and the states are read and written by the owner of a cell alone.
template <bool FULL, bool UPPER, bool FIRSTM, bool LAST, bool LUD, int NE>
__global__ void __launch_bounds__(64) k_ssor_flow(FsZone z, LuConst c, const FlowBox *__restrict__ boxes, LuRange sdm,
                                                  const FlowTask *__restrict__ tasks, int ntasks, unsigned *sync,
                                                  unsigned *sticky, const double *__restrict__ rop, const double *rop_kn,
                                                  double *rop_new, const double *__restrict__ din, double *dout,
                                                  double *ssor, const double *__restrict__ coe)
{
  constexpr int U = 8, LPP = 8, SGN = UPPER ? -1 : 1;
  const int lane = threadIdx.x, jj = lane / U, m = lane % U;
  const int q = UPPER ? LPP - 1 - jj : jj;          // rank of the line in the order of the sweep
  const int mfirst = UPPER ? U - 1 : 0;             // cell of a unit the recurrence starts from
  unsigned *flags = sync + FLOW_FLAGS;
  const long long nd = z.nd;
  for (bool alive = true; alive;) {
    const int n = (int)(__builtin_amdgcn_readfirstlane(
                      __hip_atomic_fetch_add(sync + FLOW_COUNTER, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 6);
    if (n >= ntasks) break;
    const FlowTask t = tasks[UPPER ? ntasks - 1 - n : n];
    const LuRange r = boxes[t.b].r;
    const int njb = boxes[t.b].njb;
    const int ni = r.i1 - r.i0 + 1, nchunk = (ni + U - 1) / U, nsteps = nchunk + LPP - 1;
    const long long lsafe = lcell(z, r.i0, r.j0, r.k0), ltsafe = FULL ? lmtr(z, r.i0, r.j0, r.k0) : 0;
    const int j = r.j0 + LPP * t.jb + jj, k = t.k;
    const bool line_ok = j <= r.j1;
    const bool has_j = UPPER ? j < r.j1 : j > r.j0, has_k = UPPER ? k < r.k1 : k > r.k0;
    const bool pred_pencil = UPPER ? t.jb < njb - 1 : t.jb > 0;
    const unsigned *fk = flags + t.fid - SGN * njb, *fj = flags + t.fid - SGN;
    unsigned *mine = flags + t.fid;
    const long long lrow = lcell(z, r.i0, line_ok ? j : r.j0, k) + m, ltrow = FULL ? lmtr(z, r.i0, line_ok ? j : r.j0, k) + m : 0;
    const long long dj_ = has_j ? SGN * z.incj : 0, dk_ = has_k ? SGN * z.inck : 0;   // no neighbour: reads itself, drops the term
    const bool in_sdm_jk = j >= sdm.j0 && j <= sdm.j1 && k >= sdm.k0 && k <= sdm.k1;
    auto geom = [&](int s, int &i, bool &active, bool &has_prev, long long &l, long long &lt) {
      const int cn = UPPER ? nchunk - 1 - (s - q) : s - q, di = cn * U;
      i = r.i0 + di + m;
      active = line_ok && cn >= 0 && cn < nchunk && di + m < ni;
      has_prev = active && (UPPER ? di + m < ni - 1 : di + m > 0);
      l = active ? lrow + di : lsafe;
      lt = FULL ? (active ? ltrow + di : ltsafe) : 0;
    };
    // loaded one step ahead: what a step reads that no other wave of the launch writes
    double sA[NE], sS[NE], sq[NE], sxal, sc5, sc6 = 1., sk[4], ck, sj[4], cj, si[4], ci;
    auto load_stat = [&](int s) {
      int i; bool active, has_prev; long long l, lt;
      geom(s, i, active, has_prev, l, lt);
      const long long lj = l - dj_, lk = l - dk_, li = has_prev ? l - SGN : l;
      sxal = coe[l]; sc5 = coe[l + 4 * nd];
      if (NE == 6) sc6 = coe[l + 5 * nd];
#pragma unroll
      for (int e = 0; e < NE; e++) {
        if (UPPER && LUD) sA[e] = dout[l + e * nd];        // the cell's own increment of the lower sweep (the launch before)
        else { sA[e] = din[l + e * nd]; if (UPPER || !FIRSTM) sS[e] = ssor[l + e * nd]; }
        if (UPPER && LAST) sq[e] = rop[l + e * nd];
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (!(UPPER && LAST)) sk[e] = rop[lk + (e + 1) * nd];
        sj[e] = rop[lj + (e + 1) * nd]; si[e] = rop[li + (e + 1) * nd];
      }
      ck = coe[lk + 3 * nd]; cj = coe[lj + 2 * nd]; ci = coe[li + nd];
    };
    load_stat(0);
    for (int s = 0; s <= nsteps && alive; s++) {   // pass nsteps publishes the last count only
      int i; bool active, has_prev; long long l, lt;
      geom(s, i, active, has_prev, l, lt);
      const long long lj = l - dj_, lk = l - dk_, li = (has_prev && m == mfirst) ? l - SGN : l;
      const unsigned need_k = (has_k && s < nsteps) ? (unsigned)(s + 1) : 0u;
      const unsigned need_j = (pred_pencil && s < nchunk) ? (unsigned)(s + LPP) : 0u;
      alive = flow_wait(fk, need_k, fj, need_j, mine, (unsigned)s, sync, sticky);
      if (!alive || s == nsteps) continue;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double A[NE], S[NE], P[NE], dk[NE], dj[NE], skn[4], qs[NE];
#pragma unroll
      for (int e = 0; e < NE; e++) { dk[e] = ld_sc1(dout + lk + e * nd); dj[e] = ld_sc1(dout + lj + e * nd); P[e] = ld_sc1(dout + li + e * nd); }
#pragma unroll
      for (int e = 0; e < 4; e++) skn[e] = (UPPER && LAST) ? ld_sc1(rop_kn + lk + (e + 1) * nd) : sk[e];
      double MI[26], MJ[26], MK[26];
      const double XAL = sxal * (UPPER ? -0.5 : 0.5), c5 = sc5, DG = 1. / c5, C6 = sc6, ckk = ck;
#pragma unroll
      for (int e = 0; e < NE; e++) {
        if (UPPER && LUD) A[e] = sA[e];
        else A[e] = (!UPPER && FIRSTM) ? sA[e] : sA[e] + sS[e];
        S[e] = 0.;
        if (UPPER && LAST) qs[e] = sq[e];
      }
      if (UPPER && LUD) {
#pragma unroll
        for (int e = 0; e < 5; e++) A[e] = A[e] * c5;
        if (NE == 6) A[NE - 1] = A[NE - 1] * C6;   // LU/lu_d_SA.for
      }
      lu_mat<FULL, 1, SGN, NE>(z, c, lt, sj[0], sj[1], sj[2], sj[3], cj, MJ);
      lu_mat<FULL, 0, SGN, NE>(z, c, lt, si[0], si[1], si[2], si[3], ci, MI);
      if (s + 1 < nsteps) load_stat(s + 1);
      lu_mat<FULL, 2, SGN, NE>(z, c, lt, skn[0], skn[1], skn[2], skn[3], ckk, MK);
      double BJ[NE], BK[NE];
      {
        double b[NE];
        lu_mul<FULL, 1, NE>(MJ, dj[0], dj[1], dj[2], dj[3], dj[4], b, NE == 6 ? dj[NE - 1] : 0.);
#pragma unroll
        for (int e = 0; e < NE; e++) BJ[e] = b[e] * XAL;
        lu_mul<FULL, 2, NE>(MK, dk[0], dk[1], dk[2], dk[3], dk[4], b, NE == 6 ? dk[NE - 1] : 0.);
#pragma unroll
        for (int e = 0; e < NE; e++) BK[e] = b[e] * XAL;
      }
      auto cell = [&](const double *pin) {
        bool any = false;
        if (has_prev) {
          double b[NE];
          lu_mul<FULL, 0, NE>(MI, pin[0], pin[1], pin[2], pin[3], pin[4], b, NE == 6 ? pin[NE - 1] : 0.);
#pragma unroll
          for (int e = 0; e < NE; e++) { const double tt = b[e] * XAL; A[e] = A[e] + tt; S[e] = (!UPPER && FIRSTM) ? tt : S[e] + tt; }
          any = true;
        }
        if (has_j) {
#pragma unroll
          for (int e = 0; e < NE; e++) { A[e] = A[e] + BJ[e]; S[e] = (!UPPER && FIRSTM && !any) ? BJ[e] : S[e] + BJ[e]; }
          any = true;
        }
        if (has_k) {
#pragma unroll
          for (int e = 0; e < NE; e++) { A[e] = A[e] + BK[e]; S[e] = (!UPPER && FIRSTM && !any) ? BK[e] : S[e] + BK[e]; }
        }
#pragma unroll
        for (int e = 0; e < 5; e++) P[e] = A[e] * DG;
        if (NE == 6) P[NE - 1] = A[NE - 1] / C6;   // LU/lu_dinv_SA.for
      };
      if (m == mfirst && active) {
        double pin[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) pin[e] = P[e];
        cell(pin);
      }
#pragma unroll
      for (int st = 1; st < U; st++) {
        double pin[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) pin[e] = UPPER ? lane_next(P[e]) : lane_prev(P[e]);
        if (m == (UPPER ? U - 1 - st : st) && active) cell(pin);
      }
      if (active) {
#pragma unroll
        for (int e = 0; e < NE; e++) { st_sc1(dout + l + e * nd, P[e]); if (!(UPPER && LUD)) ssor[l + e * nd] = S[e]; }
        if (UPPER && LAST && in_sdm_jk && i >= sdm.i0 && i <= sdm.i1) {
          // Newton update (mjr_newton.for)
          const double ro_old = qs[0], u_old = qs[1], v_old = qs[2], w_old = qs[3], t_old = qs[4];
          const double rn = ro_old + P[0];
          const double r_1 = 1. / rn;
          const double un = (ro_old * u_old + P[1]) * r_1;
          const double vn = (ro_old * v_old + P[2]) * r_1;
          const double wn = (ro_old * w_old + P[3]) * r_1;
          const double roe_old = ro_old * (c.cv * t_old + 0.5 * (u_old * u_old + v_old * v_old + w_old * w_old));
          st_sc1(rop_new + l, rn); st_sc1(rop_new + l + nd, un); st_sc1(rop_new + l + 2 * nd, vn); st_sc1(rop_new + l + 3 * nd, wn);
          const double tn = (roe_old + P[4]) * r_1 * c.cvinv - c.cvinv2 * (un * un + vn * vn + wn * wn);
          st_sc1(rop_new + l + 4 * nd, tn);
          if (NE == 6) {   // LU/mjr_newton_SA.for: nu_tilde clipped to [0, ratiom * nu]
            double anulam = fs_mulam_sa(z, tn);
            anulam = anulam * r_1 * z.ratiom;
            double t6 = (ro_old * qs[NE - 1] + P[NE - 1]) * r_1;
            t6 = fs_min(t6, anulam);
            st_sc1(rop_new + l + 5 * nd, fs_max(t6, 0.));
          }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_copy_range(FsZone z, LuRange r, const double *__restrict__ src,
                                                    double *__restrict__ dst)
{
  const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x, j = r.j0 + blockIdx.y, k = r.k0 + blockIdx.z;
  if (i > r.i1) return;
  const long long l = lcell(z, i, j, k);
  for (int e = 0; e < z.neq; e++) dst[l + e * z.nd] = src[l + e * z.nd];
}

}  // namespace

void fsk_coe_implicit(const FsZone &z, int first_it, int dtloc, double cfl, double gamma, double prandtl, double rgp,
                      const double *rop, const double *xmut, double *coe, hipStream_t s)
{
  double xid = 1.5;
  if (first_it == 0 || dtloc) xid = 1.;
  double gam1 = gamma / prandtl;
  if (z.iflow == 1) gam1 = dtloc ? 1e-30 : 0.;
  const double gam2 = gamma * rgp;
  dim3 g((z.ni + 255) / 256, z.nj, z.nk);
#define FS_GO(F, D) hipLaunchKernelGGL((k_coe_implicit<F, D>), g, dim3(256), 0, s, z, xid, gam1, gam2, cfl, rop, xmut, coe)
  if (z.ityp == 0) { if (dtloc) FS_GO(true, true); else FS_GO(true, false); }
  else { if (dtloc) FS_GO(false, true); else FS_GO(false, false); }
#undef FS_GO
}

// LU-SGS on the interior of the zone + Newton update.  rop = current iterate; the new iterate is
// written to rop_out.  When rop_out == rop (in-place update, sub-iterations after the first) the new
// state is built in scratch (a full time level) and copied back.
// nblocks > 1: the range is cut in blocks that are swept independently of each other, as the OpenMP threads of the
// reference do (blocks = their LU ranges on the host, 6 integers each; blocks_dev = the same table on the device); the
// blocks tile `range`.  A launch then advances every block by one hyperplane (blockIdx.z = block).
template <int U>
static int lusgs_sweeps(const FsZone &z, const int *range, int nblocks, const int *blocks, const int *blocks_dev, double gamma,
                        const double *rop, double *rop_out, double *scratch, double *dw, const double *coe, hipStream_t s,
                        bool apart)
{
  LuConst c;
  c.gam1 = gamma; c.gamm1 = gamma - 1.; c.cp = gamma * z.cv; c.gam2 = c.gamm1 - 1.;
  c.cv = z.cv; c.cvinv = 1. / z.cv; c.cvinv2 = 0.5 * c.cvinv;
  LuRange r = {range[0], range[1], range[2], range[3], range[4], range[5]};
  const int ni = r.i1 - r.i0 + 1, nj = r.j1 - r.j0 + 1, nk = r.k1 - r.k0 + 1;
  if (ni < 1 || nj < 1 || nk < 1) return -1;
  const bool full = z.ityp == 0;
  const bool inplace = rop_out == rop;
  if (inplace && !scratch) return -1;
  double *rop_new = inplace ? scratch : rop_out;
  const double *rop_kn = inplace ? scratch : rop;
  const bool sa = z.neq == 6;
  const int wpu = 64 / U;
  // units on 128-byte lines: 0 off, 1 where it pays (default), 2 wherever possible; + 4: workgroups in their plain order
  // instead of every block on one XCD (256^3, ms per implicit step, plain / one XCD per block: 64 blocks 17.3 / 15.9, 128
  // blocks 16.9 / 14.6, with two LU-SSOR relaxations in 64 blocks 37.9 / 35.5)
  const int align = lu_align_setting();
  const bool multi = nblocks > 1 && blocks && blocks_dev;
  const int nb = multi ? nblocks : 1;
  const int *br = multi ? blocks : range;
  const LuRange *tab = multi ? reinterpret_cast<const LuRange *>(blocks_dev) : nullptr;
  int max_nhc = 0;
  for (int b = 0; b < nb; b++) {
    const int *q = br + 6 * b;
    const int bi = q[1] - q[0] + 1, bj = q[3] - q[2] + 1, bk = q[5] - q[4] + 1;
    if (bi < 1 || bj < 1 || bk < 1) return -1;
    const LuSkew sk = lu_skew<U>(z, q[0], q[1], q[2], q[3], q[4], q[5], multi ? (align & 3) : 0);
    const int nhc = lu_nchunk<U>(sk, bi, bj, bk) + bj + bk - 2;
    if (nhc > max_nhc) max_nhc = nhc;
  }
  for (int pass = 0; pass < 2; pass++)
    for (int n = 0; n < max_nhc; n++) {
      // grid of launch n: the widest hyperplane among the blocks that have one.  One sweep of the range (no table):
      // the launch walks hyperplane h of the range, its chunk count and first k-plane go in as arguments
      unsigned gx = 0, gy = 0;
      int h1 = 0, nchunk1 = 0, kbase1 = 0;
      for (int b = 0; b < nb; b++) {
        const int *q = br + 6 * b;
        const int bi = q[1] - q[0] + 1, bj = q[3] - q[2] + 1, bk = q[5] - q[4] + 1;
        const LuSkew sk = lu_skew<U>(z, q[0], q[1], q[2], q[3], q[4], q[5], multi ? (align & 3) : 0);
        const int nchunk = lu_nchunk<U>(sk, bi, bj, bk);
        const int nhc = nchunk + bj + bk - 2;
        const int h = pass == 0 ? n : nhc - 1 - n;
        if (h < 0 || h >= nhc) continue;
        int klo = h - (nchunk - 1) - (bj - 1); if (klo < 0) klo = 0;
        const int khi = h < bk - 1 ? h : bk - 1;
        // chunk columns of one plane that can hold cells of a hyperplane (the kernel starts at the first of them)
        int width = (bi + U - 2) / (U + sk.a) + 2; if (width > nchunk) width = nchunk;
        const unsigned x = (unsigned)((width + wpu - 1) / wpu), y = (unsigned)(khi - klo + 1);
        if (x > gx) gx = x;
        if (y > gy) gy = y;
        h1 = h; nchunk1 = nchunk; kbase1 = q[4] + klo;
      }
      if (!gx || !gy) continue;
      dim3 g(gx, gy, (unsigned)nb);
#define FS_LOW(F, N)                                                                                                            \
  do {                                                                                                                          \
    if (multi) hipLaunchKernelGGL((k_lu_lower_wave<F, N, U, true>), g, dim3(64), 0, s, z, c, r, align, n, 0, rop, dw, coe, tab);     \
    else hipLaunchKernelGGL((k_lu_lower_wave<F, N, U, false>), g, dim3(64), 0, s, z, c, r, nchunk1, h1, kbase1, rop, dw, coe, tab); \
  } while (0)
#define FS_UPP(F, N)                                                                                                            \
  do {                                                                                                                          \
    if (multi) hipLaunchKernelGGL((k_lu_upper_wave<F, N, U, true>), g, dim3(64), 0, s, z, c, r, align, n, 0, rop, rop_kn, rop_new, dw, coe, tab); \
    else hipLaunchKernelGGL((k_lu_upper_wave<F, N, U, false>), g, dim3(64), 0, s, z, c, r, nchunk1, h1, kbase1, rop, rop_kn, rop_new, dw, coe, tab); \
  } while (0)
      if (pass == 0) {
        if (sa && full) FS_LOW(true, 6); else if (sa) FS_LOW(false, 6); else if (full) FS_LOW(true, 5); else FS_LOW(false, 5);
      } else {
        if (sa && full) FS_UPP(true, 6); else if (sa) FS_UPP(false, 6); else if (full) FS_UPP(true, 5); else FS_UPP(false, 5);
      }
#undef FS_LOW
#undef FS_UPP
    }
  if (inplace && multi && apart) {   // the new iterate exists inside the boxes only
    for (int b = 0; b < nb; b++) {
      const int *q = br + 6 * b;
      const LuRange rb = {q[0], q[1], q[2], q[3], q[4], q[5]};
      dim3 g((unsigned)((q[1] - q[0] + 1 + 255) / 256), (unsigned)(q[3] - q[2] + 1), (unsigned)(q[5] - q[4] + 1));
      hipLaunchKernelGGL(k_copy_range, g, dim3(256), 0, s, z, rb, (const double *)scratch, rop_out);
    }
  } else if (inplace) {
    dim3 g((unsigned)((ni + 255) / 256), (unsigned)nj, (unsigned)nk);
    hipLaunchKernelGGL(k_copy_range, g, dim3(256), 0, s, z, r, (const double *)scratch, rop_out);
  }
  return 0;
}

// Task table and synchronisation words of the dataflow sweeps of one zone (built at the first solve of a set of ranges)
struct FsLuFlow {
  std::vector<int> key;             // the ranges the tables were built for, and the lines of a pencil
  int ntasks = 0;
  int concurrent = 1;               // sweep launches of other zones expected on the device at the same time, this one included
  FlowTask *tasks_dev = nullptr;
  FlowBox *boxes_dev = nullptr;
  unsigned *sync_dev = nullptr;     // [counter, abort, ..., one progress word per task] zeroed before every sweep, then the
  size_t sync_bytes = 0;            // word that stays set once a wave has given up
};

FsLuFlow *fsk_luflow_new() { return new FsLuFlow; }
void fsk_luflow_concurrency(FsLuFlow *f, int n) { if (f) f->concurrent = n < 1 ? 1 : n; }
void fsk_luflow_free(FsLuFlow *f)
{
  if (!f) return;
  (void)hipFree(f->tasks_dev); (void)hipFree(f->boxes_dev); (void)hipFree(f->sync_dev);
  delete f;
}
int fsk_luflow_gave_up(const FsLuFlow *f)
{
  if (!f || !f->sync_dev) return 0;
  unsigned w = 0;
  if (hipMemcpy(&w, reinterpret_cast<const char *>(f->sync_dev) + f->sync_bytes, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return w != 0u;
}

// FASTS_HIP_LU_FLOW: 1 (default) one launch per sweep, 0 one launch per hyperplane.  FASTS_HIP_LU_FLOW_WAVES: waves of a
// sweep launch = pencils in work at a time (256^3, ms per step of 3 Newton sub-iterations: 256 waves 31.9, 512 21.0,
// 768 20.5, 1024 21.4, 1536 24.2, 2048 24.6: more waves share the issue slots of the waves on the critical path).
static int flow_setting(const char *name, int dflt)
{
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

// nb ranges swept independently of each other (6 integers each).  The tables are rebuilt when the ranges change; the
// frees wait for the device, so a sweep that still reads the old tables has ended by then.
static int flow_prepare(FsLuFlow *f, int nb, const int *ranges, int lpp, hipStream_t s)
{
  std::vector<int> key(ranges, ranges + 6 * (size_t)nb);
  key.push_back(lpp);
  if (f->tasks_dev && key == f->key) return 0;
  // order of the first possible step: pencil jb of a range starts as many steps after pencil jb - 1 as a pencil has
  // lines, plane k one step after plane k - 1; the ranges side by side
  std::vector<FlowBox> boxes((size_t)nb);
  std::vector<int> fid0((size_t)nb);
  int ntasks = 0, tmax = 0;
  for (int b = 0; b < nb; b++) {
    const int *q = ranges + 6 * b;
    const int nj = q[3] - q[2] + 1, nk = q[5] - q[4] + 1;
    if (q[1] < q[0] || nj < 1 || nk < 1) return -1;
    boxes[b].r = {q[0], q[1], q[2], q[3], q[4], q[5]};
    boxes[b].njb = (nj + lpp - 1) / lpp; boxes[b].pad = 0;
    fid0[b] = ntasks;
    ntasks += boxes[b].njb * nk;
    const int t = lpp * (boxes[b].njb - 1) + nk - 1;
    if (t > tmax) tmax = t;
  }
  std::vector<FlowTask> tasks;
  tasks.reserve((size_t)ntasks);
  for (int T = 0; T <= tmax; T++)
    for (int b = 0; b < nb; b++) {
      const int njb = boxes[b].njb, nk = boxes[b].r.k1 - boxes[b].r.k0 + 1;
      for (int jb = 0; jb < njb; jb++) {
        const int kk = T - lpp * jb;
        if (kk < 0 || kk >= nk) continue;
        tasks.push_back({jb, boxes[b].r.k0 + kk, fid0[b] + kk * njb + jb, b});
      }
    }
  if ((int)tasks.size() != ntasks) return -1;
  (void)hipFree(f->tasks_dev); (void)hipFree(f->boxes_dev); (void)hipFree(f->sync_dev);
  f->tasks_dev = nullptr; f->boxes_dev = nullptr; f->sync_dev = nullptr;
  f->sync_bytes = (((size_t)FLOW_FLAGS + (size_t)ntasks) * sizeof(unsigned) + 15) / 16 * 16;
  if (hipMalloc(&f->tasks_dev, sizeof(FlowTask) * (size_t)ntasks) != hipSuccess) return -1;
  if (hipMalloc(&f->boxes_dev, sizeof(FlowBox) * (size_t)nb) != hipSuccess) return -1;
  if (hipMalloc(&f->sync_dev, f->sync_bytes + 16) != hipSuccess) return -1;
  // On the stream of the solve, and waited for: a copy or a fill on the null stream is not ordered with a non-blocking
  // stream (a fill that lands while the first sweep runs would zero its task counter and progress words).
  if (hipMemcpyAsync(f->tasks_dev, tasks.data(), sizeof(FlowTask) * (size_t)ntasks, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
  if (hipMemcpyAsync(f->boxes_dev, boxes.data(), sizeof(FlowBox) * (size_t)nb, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
  if (hipMemsetAsync(f->sync_dev, 0, f->sync_bytes + 16, s) != hipSuccess) return -1;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  f->key = key;
  f->ntasks = ntasks;
  return 0;
}

// nb ranges (the range of the zone, or the blocks / boxes swept independently of each other); copy back of an in-place
// update: the ranges themselves
static int lusgs_flow(FsLuFlow *f, const FsZone &z, int nb, const int *ranges, double gamma, const double *rop, double *rop_out,
                      double *scratch, double *dw, const double *coe, hipStream_t s)
{
  LuConst c;
  c.gam1 = gamma; c.gamm1 = gamma - 1.; c.cp = gamma * z.cv; c.gam2 = c.gamm1 - 1.;
  c.cv = z.cv; c.cvinv = 1. / z.cv; c.cvinv2 = 0.5 * c.cvinv;
  const bool full = z.ityp == 0, sa = z.neq == 6;
  const bool inplace = rop_out == rop;
  if (inplace && !scratch) return -1;
  double *rop_new = inplace ? scratch : rop_out;
  const double *rop_kn = inplace ? scratch : rop;
  // cells of a unit: 8 lines x 8 cells, or 16 x 4 (FASTS_HIP_LU_FLOW_UNIT=4: half the recurrence per step, 6 % more steps;
  // measured slower, 256^3: 23.7 against 20.5 ms per step -- twice the hand-overs and half-used 64-byte segments)
  const int unit = flow_setting("FASTS_HIP_LU_FLOW_UNIT", 8) == 4 ? 4 : 8;
  if (flow_prepare(f, nb, ranges, 64 / unit, s) != 0) return -1;
  // pencils in work at a time: 768 for one sweep on the device; when the sweeps of several zones run side by side on streams of
  // their own, about 1 000 waves in all (four channel blocks of 128 x 256 x 128 cells on four streams, ms per step: 192 waves per
  // launch 41.7, 256 38.3, 384 38.8, 768 42.7)
  int waves = flow_setting("FASTS_HIP_LU_FLOW_WAVES", 2304 / (2 * f->concurrent + 1));
  if (waves < 1) waves = 1;
  if (waves > f->ntasks) waves = f->ntasks;
  unsigned *sticky = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(f->sync_dev) + f->sync_bytes);
  const dim3 g((unsigned)waves);
  if (hipMemsetAsync(f->sync_dev, 0, f->sync_bytes, s) != hipSuccess) return -1;
#define FS_LOW(F, N)                                                                                                         \
  do {                                                                                                                       \
    if (unit == 4) hipLaunchKernelGGL((k_lu_lower_flow<F, N, 4>), g, dim3(64), 0, s, z, c, f->boxes_dev, f->tasks_dev, f->ntasks, f->sync_dev, sticky, rop, dw, coe); \
    else if (z.twod) hipLaunchKernelGGL((k_lu_lower_flow<F, N, 8, false>), g, dim3(64), 0, s, z, c, f->boxes_dev, f->tasks_dev, f->ntasks, f->sync_dev, sticky, rop, dw, coe); \
    else hipLaunchKernelGGL((k_lu_lower_flow<F, N, 8>), g, dim3(64), 0, s, z, c, f->boxes_dev, f->tasks_dev, f->ntasks, f->sync_dev, sticky, rop, dw, coe); \
  } while (0)
  if (sa && full) FS_LOW(true, 6); else if (sa) FS_LOW(false, 6); else if (full) FS_LOW(true, 5); else FS_LOW(false, 5);
#undef FS_LOW
  auto debug_dump = [&](const char *what) {   // FASTS_HIP_LU_FLOW_DEBUG: wait for the sweep and show the words of a sweep that gave up
    if (!flow_setting("FASTS_HIP_LU_FLOW_DEBUG", 0)) return;
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> w(f->sync_bytes / sizeof(unsigned) + 4);
    (void)hipMemcpy(w.data(), f->sync_dev, f->sync_bytes + 16, hipMemcpyDeviceToHost);
    if (!w[FLOW_ABORT]) return;
    fprintf(stderr, "lusgs_flow %s gave up: %d ranges, first %d-%d %d-%d %d-%d, %d tasks, %d waves, counter %u; steps done per task:\n",
            what, nb, ranges[0], ranges[1], ranges[2], ranges[3], ranges[4], ranges[5], f->ntasks, waves, w[FLOW_COUNTER]);
    for (int q = 0; q < f->ntasks && q < 1600; q++) fprintf(stderr, " %3u%s", w[FLOW_FLAGS + q], q % 32 == 31 ? "\n" : "");
    fprintf(stderr, "\n");
  };
  debug_dump("lower sweep");
  if (hipMemsetAsync(f->sync_dev, 0, f->sync_bytes, s) != hipSuccess) return -1;
#define FS_UPP(F, N)                                                                                                         \
  do {                                                                                                                       \
    if (unit == 4) hipLaunchKernelGGL((k_lu_upper_flow<F, N, 4>), g, dim3(64), 0, s, z, c, f->boxes_dev, f->tasks_dev, f->ntasks, f->sync_dev, sticky, rop, rop_kn, rop_new, dw, coe); \
    else if (z.twod) hipLaunchKernelGGL((k_lu_upper_flow<F, N, 8, false>), g, dim3(64), 0, s, z, c, f->boxes_dev, f->tasks_dev, f->ntasks, f->sync_dev, sticky, rop, rop_kn, rop_new, dw, coe); \
    else hipLaunchKernelGGL((k_lu_upper_flow<F, N, 8>), g, dim3(64), 0, s, z, c, f->boxes_dev, f->tasks_dev, f->ntasks, f->sync_dev, sticky, rop, rop_kn, rop_new, dw, coe); \
  } while (0)
  if (sa && full) FS_UPP(true, 6); else if (sa) FS_UPP(false, 6); else if (full) FS_UPP(true, 5); else FS_UPP(false, 5);
#undef FS_UPP
  debug_dump("upper sweep");
  if (inplace) {
    for (int b = 0; b < nb; b++) {
      const int *q = ranges + 6 * b;
      const LuRange rb = {q[0], q[1], q[2], q[3], q[4], q[5]};
      dim3 gc((unsigned)((q[1] - q[0] + 1 + 255) / 256), (unsigned)(q[3] - q[2] + 1), (unsigned)(q[5] - q[4] + 1));
      hipLaunchKernelGGL(k_copy_range, gc, dim3(256), 0, s, z, rb, (const double *)scratch, rop_out);
    }
  }
  return 0;
}

int fsk_lusgs(const FsZone &z, const int *range, int nblocks, const int *blocks, const int *blocks_dev, double gamma,
              const double *rop, double *rop_out, double *scratch, double *dw, const double *coe, hipStream_t s, bool apart,
              FsLuFlow *flow)
{
  // One launch per sweep: one sweep of the range; the boxes of the sub-zones (apart); the thread blocks (they tile the
  // range, so the copy back of the blocks is the copy back of the range) while they are few: the hyperplane launches of
  // many blocks are wide enough to be bound by issue and memory like the one-launch form, and their 16-cell units use
  // whole lines.  Measured, ms per step, one launch per hyperplane / per sweep: 256^3 in 16 blocks 21.4 / 16.0, 64 blocks
  // 15.4 / 15.7, 128 blocks 13.0 / 16.4; the 1024 x 256 C-mesh in 24 blocks 4.06 / 3.75; sub-zones of 32^3 cells,
  // epsi_newton 0.3: 45.5 / 30.9.  FASTS_HIP_LU_FLOW_BLOCKS: 0 never for thread blocks, 1 always, else below 48 blocks.
  if (flow && flow_setting("FASTS_HIP_LU_FLOW_CHECK", 0) != 0 && nblocks <= 1) {
    // development aid: the one-launch solve, then the per-hyperplane solve from the same input, compared on the host
    const size_t n = (size_t)z.neq * (size_t)z.nd, nb = n * sizeof(double);
    const bool inplace = rop_out == rop;
    double *dw0 = nullptr, *rop0 = nullptr;
    if (hipMalloc(&dw0, nb) != hipSuccess || hipMalloc(&rop0, nb) != hipSuccess) return -1;
    // (every copy on the stream of the solve: a copy on the null stream is not ordered with a non-blocking stream)
    (void)hipMemcpyAsync(dw0, dw, nb, hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(rop0, rop_out, nb, hipMemcpyDeviceToDevice, s);
    const int *rg = (apart && nblocks == 1 && blocks) ? blocks : range;
    if (lusgs_flow(flow, z, 1, rg, gamma, rop, rop_out, scratch, dw, coe, s) != 0) return -1;
    std::vector<double> dwA(n), roA(n), dwB(n), roB(n);
    (void)hipMemcpyAsync(dwA.data(), dw, nb, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(roA.data(), rop_out, nb, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(dw, dw0, nb, hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(rop_out, rop0, nb, hipMemcpyDeviceToDevice, s);
    const int rc = lusgs_sweeps<8>(z, rg, 0, nullptr, nullptr, gamma, rop, rop_out, scratch, dw, coe, s, false);
    (void)hipMemcpyAsync(dwB.data(), dw, nb, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(roB.data(), rop_out, nb, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(dw0); (void)hipFree(rop0);
    size_t bad_dw = 0, bad_ro = 0;
    static int shown = 0;
    for (size_t q = 0; q < n; q++) {
      const bool b1 = memcmp(&dwA[q], &dwB[q], 8) != 0, b2 = memcmp(&roA[q], &roB[q], 8) != 0;
      bad_dw += b1; bad_ro += b2;
      if ((b1 || b2) && shown < 40) {
        shown++;
        const long long l = (long long)(q % (size_t)z.nd);
        const int e = (int)(q / (size_t)z.nd);
        const int k = (int)(l / z.inck), j = (int)((l % z.inck) / z.incj), i = (int)(l % z.incj);
        fprintf(stderr, "  flow check: eq %d cell (%d,%d,%d) [interior index %d,%d,%d] dw %.17g / %.17g  state %.17g / %.17g\n", e, i, j, k,
                i - z.ific + 1, j - z.ific + 1, k - z.ifick + 1, dwA[q], dwB[q], roA[q], roB[q]);
      }
    }
    if (bad_dw || bad_ro)
      fprintf(stderr, "flow check: range %d-%d %d-%d %d-%d inplace %d: %zu increments and %zu state values differ (one launch per sweep / per hyperplane)\n",
              rg[0], rg[1], rg[2], rg[3], rg[4], rg[5], (int)inplace, bad_dw, bad_ro);
    return rc;
  }
  if (flow && flow_setting("FASTS_HIP_LU_FLOW", 1) != 0) {
    if (nblocks <= 1) return lusgs_flow(flow, z, 1, (apart && nblocks == 1 && blocks) ? blocks : range, gamma, rop, rop_out, scratch, dw, coe, s);
    const int fb = flow_setting("FASTS_HIP_LU_FLOW_BLOCKS", -1);
    if (blocks && (apart || fb == 1 || (fb != 0 && nblocks < 48)))
      return lusgs_flow(flow, z, nblocks, blocks, gamma, rop, rop_out, scratch, dw, coe, s);
  }
  // units of U consecutive cells of an i-line: the recurrence of a launch has U steps, a sweep ni/U + nj + nk - 2
  // launches of (cells of the hyperplane)/64 waves.  One sweep of the zone is bound by the latency of a launch:
  // measured at 256^3, 3 Newton sub-iterations: U = 4 40.2 ms per step (twice the waves in the large hyperplanes of
  // mid-sweep), U = 8 31.5 ms, U = 16 33.5 ms; two-dimensional NACA0012 case 15.0 / 13.9 / 16.0 ms.  With thread blocks
  // a launch carries a hyperplane of every block and is bound by memory instead: a unit of 16 cells is a whole
  // 128-byte line of every array where 8 cells use half of each line they touch (256^3 in the blocks of 64 threads:
  // 23.4 ms per step with U = 8, 17.2 ms with U = 16; NACA0012 in 24 blocks 5.6 / 4.1 ms).
  if (nblocks > 1) return lusgs_sweeps<16>(z, range, nblocks, blocks, blocks_dev, gamma, rop, rop_out, scratch, dw, coe, s, apart);
  return lusgs_sweeps<8>(z, range, nblocks, blocks, blocks_dev, gamma, rop, rop_out, scratch, dw, coe, s, false);
}

// LU-SSOR with nb_relax >= 2 on the range `range`, Newton update on the cells of `sdm` (the interior of the zone).
// din = Newton residual (kept); work = 2 * neq * nd doubles (increments, accumulator).  In-place update as in fsk_lusgs.
// overlap (nb_relax = 1): LU-SGS as the reference runs it with lu_match = 1 (LU/invlu.for:98-109, invlu_overlap_l.for,
// invlu_overlap_u.for): one lower and one upper sweep with the neighbour terms in the order i, j, k, the increments
// in the work array, the Newton update on `sdm`.
// nblocks > 1 (not with overlap): the relaxations run block by block as the reference's threads run them, each on work
// arrays of its own window (Compute/ssor.cpp, LU/prep_lussor.h) -- here the shared arrays, no sweep reading across the
// borders of its block; see fsk_lusgs.
template <int U>
static int lussor_sweeps(const FsZone &z, const int *range, const int *sdm_, int nblocks, const int *blocks, const int *blocks_dev,
                         int nb_relax, int overlap, double gamma, const double *rop, double *rop_out, double *scratch,
                         const double *din, double *work, const double *coe, hipStream_t s)
{
  if ((z.neq != 5 && z.neq != 6) || (overlap ? nb_relax != 1 : nb_relax < 2)) return -1;
  const bool sa = z.neq == 6;
  LuConst c;
  c.gam1 = gamma; c.gamm1 = gamma - 1.; c.cp = gamma * z.cv; c.gam2 = c.gamm1 - 1.;
  c.cv = z.cv; c.cvinv = 1. / z.cv; c.cvinv2 = 0.5 * c.cvinv;
  LuRange r = {range[0], range[1], range[2], range[3], range[4], range[5]};
  LuRange sdm = {sdm_[0], sdm_[1], sdm_[2], sdm_[3], sdm_[4], sdm_[5]};
  const int ni = r.i1 - r.i0 + 1, nj = r.j1 - r.j0 + 1, nk = r.k1 - r.k0 + 1;
  if (ni < 1 || nj < 1 || nk < 1) return -1;
  const bool full = z.ityp == 0;
  const bool inplace = rop_out == rop;
  if (inplace && !scratch) return -1;
  double *rop_new = inplace ? scratch : rop_out;
  const double *rop_kn = inplace ? scratch : rop;
  double *dout = work, *ssor = work + (size_t)z.neq * z.nd;
  const int wpu = 64 / U;
  // units on 128-byte lines: 0 off, 1 where it pays (default), 2 wherever possible; + 4: workgroups in their plain order
  // instead of every block on one XCD (256^3, ms per implicit step, plain / one XCD per block: 64 blocks 17.3 / 15.9, 128
  // blocks 16.9 / 14.6, with two LU-SSOR relaxations in 64 blocks 37.9 / 35.5)
  const int align = lu_align_setting();
  const bool multi = nblocks > 1 && blocks && blocks_dev && !overlap;
  const int nb = multi ? nblocks : 1;
  const int *br = multi ? blocks : range;
  const LuRange *tab = multi ? reinterpret_cast<const LuRange *>(blocks_dev) : nullptr;
  int max_nhc = 0;
  for (int b = 0; b < nb; b++) {
    const int *q = br + 6 * b;
    const int bi = q[1] - q[0] + 1, bj = q[3] - q[2] + 1, bk = q[5] - q[4] + 1;
    if (bi < 1 || bj < 1 || bk < 1) return -1;
    const LuSkew sk = lu_skew<U>(z, q[0], q[1], q[2], q[3], q[4], q[5], multi ? (align & 3) : 0);
    const int nhc = lu_nchunk<U>(sk, bi, bj, bk) + bj + bk - 2;
    if (nhc > max_nhc) max_nhc = nhc;
  }
  const dim3 gr((unsigned)((ni + 255) / 256), (unsigned)nj, (unsigned)nk);
  if (inplace) {
    // cells of the range outside sdm keep their state: the copy back below covers the whole range
    hipLaunchKernelGGL(k_copy_range, gr, dim3(256), 0, s, z, r, rop, scratch);
  }
  for (int m = 1; m <= nb_relax; m++) {
    const bool last = m == nb_relax;
    for (int pass = 0; pass < 2; pass++) {
      for (int n = 0; n < max_nhc; n++) {
        unsigned gx = 0, gy = 0;
        int h1 = 0, nchunk1 = 0, kbase1 = 0;
        for (int b = 0; b < nb; b++) {
          const int *q = br + 6 * b;
          const int bi = q[1] - q[0] + 1, bj = q[3] - q[2] + 1, bk = q[5] - q[4] + 1;
          const LuSkew sk = lu_skew<U>(z, q[0], q[1], q[2], q[3], q[4], q[5], multi ? (align & 3) : 0);
          const int nchunk = lu_nchunk<U>(sk, bi, bj, bk);
          const int nhc = nchunk + bj + bk - 2;
          const int h = pass == 0 ? n : nhc - 1 - n;
          if (h < 0 || h >= nhc) continue;
          int klo = h - (nchunk - 1) - (bj - 1); if (klo < 0) klo = 0;
          const int khi = h < bk - 1 ? h : bk - 1;
          int width = (bi + U - 2) / (U + sk.a) + 2; if (width > nchunk) width = nchunk;
          const unsigned x = (unsigned)((width + wpu - 1) / wpu), y = (unsigned)(khi - klo + 1);
          if (x > gx) gx = x;
          if (y > gy) gy = y;
          h1 = h; nchunk1 = nchunk; kbase1 = q[4] + klo;
        }
        if (!gx || !gy) continue;
        dim3 g(gx, gy, (unsigned)nb);
#define FS_LO2(F, M1, N)                                                                                               \
  do {                                                                                                                 \
    if (multi) hipLaunchKernelGGL((k_ssor_lower_wave<F, M1, N, U, true>), g, dim3(64), 0, s, z, c, r, align, n, 0, rop, din, dout, ssor, coe, tab); \
    else hipLaunchKernelGGL((k_ssor_lower_wave<F, M1, N, U, false>), g, dim3(64), 0, s, z, c, r, nchunk1, h1, kbase1, rop, din, dout, ssor, coe, tab); \
  } while (0)
#define FS_LO(F, M1) do { if (sa) FS_LO2(F, M1, 6); else FS_LO2(F, M1, 5); } while (0)
#define FS_UPX2(F, L, D, N)                                                                                            \
  do {                                                                                                                 \
    if (multi) hipLaunchKernelGGL((k_ssor_upper_wave<F, L, D, N, U, true>), g, dim3(64), 0, s, z, c, r, sdm, align, n, 0, rop, rop_kn, rop_new, din, dout, ssor, coe, tab); \
    else hipLaunchKernelGGL((k_ssor_upper_wave<F, L, D, N, U, false>), g, dim3(64), 0, s, z, c, r, sdm, nchunk1, h1, kbase1, rop, rop_kn, rop_new, din, dout, ssor, coe, tab); \
  } while (0)
#define FS_UPX(F, L, D) do { if (sa) FS_UPX2(F, L, D, 6); else FS_UPX2(F, L, D, 5); } while (0)
#define FS_UP(F, L) FS_UPX(F, L, false)
#define FS_UPD(F) FS_UPX(F, true, true)
        if (overlap) {
          if (pass == 0) { if (full) FS_LO(true, true); else FS_LO(false, true); }
          else { if (full) FS_UPD(true); else FS_UPD(false); }
        } else if (pass == 0) {
          if (full) { if (m == 1) FS_LO(true, true); else FS_LO(true, false); }
          else { if (m == 1) FS_LO(false, true); else FS_LO(false, false); }
        } else {
          if (full) { if (last) FS_UP(true, true); else FS_UP(true, false); }
          else { if (last) FS_UP(false, true); else FS_UP(false, false); }
        }
#undef FS_LO
#undef FS_LO2
#undef FS_UP
#undef FS_UPD
#undef FS_UPX
#undef FS_UPX2
      }
    }
  }
  if (inplace) hipLaunchKernelGGL(k_copy_range, gr, dim3(256), 0, s, z, r, (const double *)scratch, rop_out);
  return 0;
}

// The relaxations of fsk_lussor as one launch per sweep over nb ranges (the range, or the thread blocks)
static int lussor_flow(FsLuFlow *f, const FsZone &z, const int *range, const int *sdm_, int nb, const int *ranges, int nb_relax,
                       int overlap, double gamma, const double *rop, double *rop_out, double *scratch, const double *din,
                       double *work, const double *coe, hipStream_t s)
{
  if ((z.neq != 5 && z.neq != 6) || (overlap ? nb_relax != 1 : nb_relax < 2)) return -1;
  LuConst c;
  c.gam1 = gamma; c.gamm1 = gamma - 1.; c.cp = gamma * z.cv; c.gam2 = c.gamm1 - 1.;
  c.cv = z.cv; c.cvinv = 1. / z.cv; c.cvinv2 = 0.5 * c.cvinv;
  const LuRange r = {range[0], range[1], range[2], range[3], range[4], range[5]};
  const LuRange sdm = {sdm_[0], sdm_[1], sdm_[2], sdm_[3], sdm_[4], sdm_[5]};
  const int ni = r.i1 - r.i0 + 1, nj = r.j1 - r.j0 + 1, nk = r.k1 - r.k0 + 1;
  if (ni < 1 || nj < 1 || nk < 1) return -1;
  const bool full = z.ityp == 0, sa = z.neq == 6;
  const bool inplace = rop_out == rop;
  if (inplace && !scratch) return -1;
  double *rop_new = inplace ? scratch : rop_out;
  const double *rop_kn = inplace ? scratch : rop;
  double *dout = work, *ssor = work + (size_t)z.neq * z.nd;
  if (flow_prepare(f, nb, ranges, 8, s) != 0) return -1;
  int waves = flow_setting("FASTS_HIP_LU_FLOW_WAVES", 2304 / (2 * f->concurrent + 1));
  if (waves < 1) waves = 1;
  if (waves > f->ntasks) waves = f->ntasks;
  unsigned *sticky = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(f->sync_dev) + f->sync_bytes);
  const dim3 g((unsigned)waves);
  const dim3 gr((unsigned)((ni + 255) / 256), (unsigned)nj, (unsigned)nk);
  // cells of the range outside sdm keep their state: the copy back below covers the whole range
  if (inplace) hipLaunchKernelGGL(k_copy_range, gr, dim3(256), 0, s, z, r, rop, scratch);
#define FS_GO4(F, UP, M1, L, D, N) hipLaunchKernelGGL((k_ssor_flow<F, UP, M1, L, D, N>), g, dim3(64), 0, s, z, c, f->boxes_dev, sdm, f->tasks_dev, f->ntasks, f->sync_dev, sticky, rop, rop_kn, rop_new, din, dout, ssor, coe)
#define FS_GO3(F, UP, M1, L, D) do { if (sa) FS_GO4(F, UP, M1, L, D, 6); else FS_GO4(F, UP, M1, L, D, 5); } while (0)
#define FS_GO(UP, M1, L, D) do { if (full) FS_GO3(true, UP, M1, L, D); else FS_GO3(false, UP, M1, L, D); } while (0)
  for (int m = 1; m <= nb_relax; m++) {
    const bool last = m == nb_relax;
    if (hipMemsetAsync(f->sync_dev, 0, f->sync_bytes, s) != hipSuccess) return -1;
    if (overlap || m == 1) FS_GO(false, true, false, false); else FS_GO(false, false, false, false);
    if (hipMemsetAsync(f->sync_dev, 0, f->sync_bytes, s) != hipSuccess) return -1;
    if (overlap) FS_GO(true, false, true, true);
    else if (last) FS_GO(true, false, true, false);
    else FS_GO(true, false, false, false);
  }
#undef FS_GO
#undef FS_GO3
#undef FS_GO4
  if (inplace) hipLaunchKernelGGL(k_copy_range, gr, dim3(256), 0, s, z, r, (const double *)scratch, rop_out);
  return 0;
}

int fsk_lussor(const FsZone &z, const int *range, const int *sdm, int nblocks, const int *blocks, const int *blocks_dev,
               int nb_relax, int overlap, double gamma, const double *rop, double *rop_out, double *scratch, const double *din,
               double *work, const double *coe, hipStream_t s, FsLuFlow *flow)
{
  // one launch per sweep, as fsk_lusgs takes it: one sweep of the range, or few thread blocks
  if (flow && flow_setting("FASTS_HIP_LU_FLOW", 1) != 0) {
    const bool multi = nblocks > 1 && blocks && !overlap;
    const int fb = flow_setting("FASTS_HIP_LU_FLOW_BLOCKS", -1);
    if (!multi) return lussor_flow(flow, z, range, sdm, 1, range, nb_relax, overlap, gamma, rop, rop_out, scratch, din, work, coe, s);
    // (two relaxations, ms per step, one launch per hyperplane / per sweep: 256^3 74.0 / 46.9, the 1024 x 256 C-mesh 29.6 / 24.1,
    // the C-mesh in 24 blocks 8.6 / 8.9, 256^3 in 64 blocks 34.4 / 35.4)
    if (fb == 1 || (fb != 0 && nblocks < 20))
      return lussor_flow(flow, z, range, sdm, nblocks, blocks, nb_relax, overlap, gamma, rop, rop_out, scratch, din, work, coe, s);
  }
  // unit length as in fsk_lusgs: 8 cells for one sweep of the zone, 16 with thread blocks
  if (nblocks > 1 && !overlap)
    return lussor_sweeps<16>(z, range, sdm, nblocks, blocks, blocks_dev, nb_relax, overlap, gamma, rop, rop_out, scratch, din, work, coe, s);
  return lussor_sweeps<8>(z, range, sdm, 0, nullptr, nullptr, nb_relax, overlap, gamma, rop, rop_out, scratch, din, work, coe, s);
}
