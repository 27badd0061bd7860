// Tiled Runge-Kutta stage kernel (flux balance + update) for Cartesian zones, gfx950.
//
// Work decomposition ("2.5-D marching"): a workgroup of 4 waves owns a tile of 64 (i) x 4 (j)
// cell columns and marches through a chunk of k-planes.  One wave = one j-row, one lane = one
// i, so every global access of a wave is a contiguous 512-byte row segment.
//   * each lane keeps its own column q(k-1..k+2) in registers (MUSCL stencil along k);
//   * the in-plane neighbourhood lives in LDS: a ring of three planes of (u,v,w) with a
//     2-cell halo (68 x 8 cells) for the cross-derivative terms of the viscous flux, and one
//     plane of (rho, T, mu) for the current k;
//   * every face flux is evaluated once: a lane computes the faces k+1/2, j-1/2 and i-1/2 of its
//     cell; the k-1/2 flux is carried over from the previous plane, the i+1/2 flux comes from
//     the neighbouring lane by a wave shuffle, the j+1/2 flux from the next row through LDS.
//     The faces on the far tile edges (i0+64 and j0+4) are evaluated by one extra masked pass
//     each, rotated over the waves.
// The six fluxes are then summed in the order +Fk- +Fj- +Fi- -Fi+ -Fj+ -Fk+ (the order in which
// the reference's scatter loops reach a cell, template_FluxAndBalance.for:192-271) and the
// Runge-Kutta update (core3ark3.for:108-154) is applied in the same kernel, so a stage reads
// the state once and writes the balance and the new state once.
#include "fasts_kernels.h"
#include "fasts_flux.h"

namespace {

constexpr int TI = 64, TJ = 4, HALO = 2;
constexpr int PW = TI + 2 * HALO;      // 68
constexpr int PH = TJ + 2 * HALO;      // 8
constexpr int PLANE = PW * PH;         // 544 cells
constexpr int NTHREADS = TI * TJ;      // 256
constexpr int NSTAGE = (PLANE + NTHREADS - 1) / NTHREADS;  // 3 cells per thread per plane-variable

struct Smem {
  double uvw[3][3][PLANE];   // [ring slot][u,v,w][cell]
  double rtx[3][PLANE];      // rho, T, mu of the current plane
  double fj[5][TJ + 1][TI];  // j-face fluxes of rows 0..TJ
  double fi_extra[5][TJ];    // i-face flux at tile column TI
};

FS_DEV int lidx(int ti, int tj) { return (tj + HALO) * PW + (ti + HALO); }
FS_DEV int slot_of(int kk) { return (kk + 3) % 3; }  // kk >= -2

// inviscid + viscous flux of an i- or j-face whose right cell is tile cell (ti,tj) of plane k;
// all operands from LDS.  Returns the sensor value used (computed when FIRST).
template <int KFLU, int IFLOW, int DIR, bool FIRST>
FS_DEV double face_ij(const FsZone &z, const Smem &sm, int k, int ti, int tj, double wg_in, double *flu)
{
  static_assert(DIR == 0 || DIR == 1, "in-plane faces only");
  const int sk = slot_of(k);
  const int c = lidx(ti, tj);
  const int st = DIR == 0 ? 1 : PW;
  double qm2[5], qm1[5], q0[5], qp1[5];
  qm2[0] = sm.rtx[0][c - 2 * st]; qm1[0] = sm.rtx[0][c - st]; q0[0] = sm.rtx[0][c]; qp1[0] = sm.rtx[0][c + st];
  qm2[4] = sm.rtx[1][c - 2 * st]; qm1[4] = sm.rtx[1][c - st]; q0[4] = sm.rtx[1][c]; qp1[4] = sm.rtx[1][c + st];
#pragma unroll
  for (int e = 1; e <= 3; e++) {
    const double *p = sm.uvw[sk][e - 1];
    qm2[e] = p[c - 2 * st]; qm1[e] = p[c - st]; q0[e] = p[c]; qp1[e] = p[c + st];
  }
  double wg = wg_in;
  if (KFLU == 2 && FIRST) {
    auto qs = [&](int e, int o) -> double { return o == -2 ? qm2[e] : (o == -1 ? qm1[e] : (o == 0 ? q0[e] : qp1[e])); };
    wg = fs_wiggle(qs);
  }
  fs_flux_inviscid<KFLU, DIR>(z, qm2, qm1, q0, qp1, wg, flu);
  if (IFLOW == 2) {
    const int skm = slot_of(k - 1), skp = slot_of(k + 1);
    // transverse directions: DIR 0 -> (a,b) = (j,k); DIR 1 -> (a,b) = (i,k)
    auto q = [&](int e, int da, int db, int dn) -> double {
      const int cc = c + dn * st + da * (DIR == 0 ? PW : 1);
      if (e == 4) return sm.rtx[1][cc];
      const int s = db == 0 ? sk : (db < 0 ? skm : skp);
      return sm.uvw[s][e - 1][cc];
    };
    fs_flux_viscous<DIR>(z, q, sm.rtx[2][c - st], sm.rtx[2][c], flu);
  }
  return wg;
}

template <int KFLU, int IFLOW, bool FIRST>
__global__ void __launch_bounds__(NTHREADS, 2)
k_stage_tile(FsZone z, double coefH, double coefW, int kchunk, int ntile_i, int ntile_j, int nblk,
             const double *__restrict__ rop, double *__restrict__ rop_out, double *__restrict__ drodm,
             const double *__restrict__ xmut, double *__restrict__ wig)
{
  __shared__ Smem sm;
  // blocks that share an XCD (same blockIdx % 8) get consecutive tiles, so halo rows of
  // neighbouring tiles meet in one L2
  int bid = blockIdx.x;
  {
    const int q8 = nblk / 8, r8 = nblk % 8, x = bid % 8;
    bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + bid / 8;
  }
  const int bi = bid % ntile_i;
  const int bj = (bid / ntile_i) % ntile_j;
  const int bk = bid / (ntile_i * ntile_j);
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  const int i0 = bi * TI, j0 = bj * TJ;             // tile cell (0,0) = interior cell (i0+1, j0+1)
  const int k0 = bk * kchunk + 1;
  const int k1 = (k0 + kchunk - 1 < z.kv) ? k0 + kchunk - 1 : z.kv;
  const int i = i0 + lane + 1, j = j0 + row + 1;
  const long long nd = z.nd, incj = z.incj, inck = z.inck;
  const bool cell_ok = (i <= z.iv) && (j <= z.jv);
  const bool fi_ok = (i <= z.iv + 1) && (j <= z.jv);    // lane evaluates its i-1/2 face
  const bool fj_ok = (i <= z.iv) && (j <= z.jv + 1);    // lane evaluates its j-1/2 face
  // own column, clamped into the array for lanes outside the zone (their values are unused)
  const int ic = i <= z.iv + 2 ? i : z.iv + 2, jc = j <= z.jv + 2 ? j : z.jv + 2;
  const long long lcol = (long long)(ic + z.ific - 1) + (long long)(jc + z.ific - 1) * incj +
                         (long long)(z.ifick - 1) * inck;   // + k*inck
  // staging map: thread handles plane cells c = tid + n*256
  int s_lds[NSTAGE];
  long long s_g[NSTAGE];
#pragma unroll
  for (int n = 0; n < NSTAGE; n++) {
    int c = threadIdx.x + n * NTHREADS;
    s_lds[n] = c < PLANE ? c : -1;
    int cc = c < PLANE ? c : 0;
    int tj = cc / PW - HALO, ti = cc % PW - HALO;
    int gi = i0 + ti + 1, gj = j0 + tj + 1;
    gi = gi <= z.iv + 2 ? gi : z.iv + 2;
    gj = gj <= z.jv + 2 ? gj : z.jv + 2;
    s_g[n] = (long long)(gi + z.ific - 1) + (long long)(gj + z.ific - 1) * incj + (long long)(z.ifick - 1) * inck;
  }
  auto stage_uvw = [&](int kk) {   // direct global -> LDS of plane kk (u,v,w)
    const int s = slot_of(kk);
#pragma unroll
    for (int n = 0; n < NSTAGE; n++)
      if (s_lds[n] >= 0) {
        const long long g = s_g[n] + (long long)kk * inck;
#pragma unroll
        for (int e = 0; e < 3; e++) sm.uvw[s][e][s_lds[n]] = rop[(e + 1) * nd + g];
      }
  };
  auto stage_rtx = [&](int kk) {
#pragma unroll
    for (int n = 0; n < NSTAGE; n++)
      if (s_lds[n] >= 0) {
        const long long g = s_g[n] + (long long)kk * inck;
        sm.rtx[0][s_lds[n]] = rop[g];
        sm.rtx[1][s_lds[n]] = rop[4 * nd + g];
        if (IFLOW == 2) sm.rtx[2][s_lds[n]] = xmut[g];
      }
  };

  // ---- prologue: planes k0-1, k0, k0+1 in LDS; own column k0-2 .. k0+1 in registers ----------
  stage_uvw(k0 - 1); stage_uvw(k0); stage_uvw(k0 + 1);
  stage_rtx(k0);
  double qa[5], qb[5], qc[5], qd[5];   // own column at k-1, k, k+1, k+2 (shifted each step)
  double xm_k = 0., xm_kp1 = 0.;
#pragma unroll
  for (int e = 0; e < 5; e++) {
    qa[e] = rop[e * nd + lcol + (long long)(k0 - 2) * inck];
    qb[e] = rop[e * nd + lcol + (long long)(k0 - 1) * inck];
    qc[e] = rop[e * nd + lcol + (long long)(k0)*inck];
    qd[e] = rop[e * nd + lcol + (long long)(k0 + 1) * inck];
  }
  if (IFLOW == 2) { xm_k = xmut[lcol + (long long)(k0 - 1) * inck]; xm_kp1 = xmut[lcol + (long long)k0 * inck]; }
  __syncthreads();

  const int ct = lidx(lane, row);
  // flux through the k-face between planes kk and kk+1 of the own column:
  // qm2..qp1 = column values at kk-1, kk, kk+1, kk+2 ; viscosity at kk, kk+1
  auto kface = [&](int kk, const double *m2, const double *m1, const double *p0, const double *p1, double mul,
                   double mur, double *flu) {
    const long long lf = lcol + (long long)(kk + 1) * inck;   // the face belongs to cell kk+1
    double wg = 0.;
    if (KFLU == 2) {
      if (FIRST) {
        auto qs = [&](int e, int o) -> double { return o == -2 ? m2[e] : (o == -1 ? m1[e] : (o == 0 ? p0[e] : p1[e])); };
        wg = fs_wiggle(qs);
        if (cell_ok) wig[2 * nd + lf] = wg;
      } else
        wg = wig[2 * nd + lf];
    }
    fs_flux_inviscid<KFLU, 2>(z, m2, m1, p0, p1, wg, flu);
    if (IFLOW == 2) {
      const int sl = slot_of(kk), sr = slot_of(kk + 1);
      auto q = [&](int e, int da, int db, int dn) -> double {
        if (e == 4) return dn == 0 ? p0[4] : m1[4];
        return sm.uvw[dn == 0 ? sr : sl][e - 1][ct + da + db * PW];
      };
      fs_flux_viscous<2>(z, q, mul, mur, flu);
    }
  };

  double fkm[5];
  kface(k0 - 1, qa, qb, qc, qd, xm_k, xm_kp1, fkm);   // flux at k0 - 1/2

  for (int k = k0; k <= k1; k++) {
    // shift the column window: now qa..qd = k-1, k, k+1, k+2
#pragma unroll
    for (int e = 0; e < 5; e++) { qa[e] = qb[e]; qb[e] = qc[e]; qc[e] = qd[e]; }
    xm_k = xm_kp1;
    const long long lk = lcol + (long long)k * inck;
#pragma unroll
    for (int e = 0; e < 5; e++) qd[e] = rop[e * nd + lk + 2 * inck];
    if (IFLOW == 2) xm_kp1 = xmut[lk + inck];
    // operands of the balance that do not depend on the fluxes
    double d[5];
#pragma unroll
    for (int e = 0; e < 5; e++) d[e] = FIRST ? 0. : drodm[e * nd + lk] * coefH;
    double wgi = 0., wgj = 0.;
    if (KFLU == 2 && !FIRST) { wgi = wig[lk]; wgj = wig[nd + lk]; }
    // next planes, staged through registers and written after the fluxes are exchanged
    double s_uvw[NSTAGE][3], s_rtx[NSTAGE][3];
    const bool more = k < k1;
    if (more) {
#pragma unroll
      for (int n = 0; n < NSTAGE; n++)
        if (s_lds[n] >= 0) {
          const long long g2 = s_g[n] + (long long)(k + 2) * inck, g1 = s_g[n] + (long long)(k + 1) * inck;
#pragma unroll
          for (int e = 0; e < 3; e++) s_uvw[n][e] = rop[(e + 1) * nd + g2];
          s_rtx[n][0] = rop[g1];
          s_rtx[n][1] = rop[4 * nd + g1];
          if (IFLOW == 2) s_rtx[n][2] = xmut[g1];
        }
    }

    double fkp[5], fjm[5] = {0., 0., 0., 0., 0.}, fim[5] = {0., 0., 0., 0., 0.}, fip[5], fjp[5];
    kface(k, qa, qb, qc, qd, xm_k, xm_kp1, fkp);

    if (fj_ok) {
      double w = face_ij<KFLU, IFLOW, 1, FIRST>(z, sm, k, lane, row, wgj, fjm);
      if (KFLU == 2 && FIRST) wig[nd + lk] = w;
#pragma unroll
      for (int e = 0; e < 5; e++) sm.fj[e][row][lane] = fjm[e];
    }
    {
      double w = 0.;
      if (fi_ok) {
        w = face_ij<KFLU, IFLOW, 0, FIRST>(z, sm, k, lane, row, wgi, fim);
        if (KFLU == 2 && FIRST) wig[lk] = w;
      }
    }
    // far-edge faces of the tile: one masked pass each, rotated over the waves
    const int duty = (k - k0) & 3;
    if (row == duty) {
      // row j0+TJ+1 of the zone: tile row TJ
      const int je = j0 + TJ + 1;
      if (je <= z.jv + 1 && i <= z.iv) {
        const long long le = lk + (long long)(TJ - row) * incj;
        double wge = 0.;
        if (KFLU == 2 && !FIRST) wge = wig[nd + le];
        double fe[5];
        double w = face_ij<KFLU, IFLOW, 1, FIRST>(z, sm, k, lane, TJ, wge, fe);
        if (KFLU == 2 && FIRST) wig[nd + le] = w;
#pragma unroll
        for (int e = 0; e < 5; e++) sm.fj[e][TJ][lane] = fe[e];
      }
    }
    if (row == ((duty + 2) & 3)) {
      // column i0+TI+1 of the zone: tile column TI, one lane per row
      const int ie = i0 + TI + 1, jr = j0 + lane + 1;
      if (lane < TJ && ie <= z.iv + 1 && jr <= z.jv) {
        const long long le = (long long)(ie + z.ific - 1) + (long long)(jr + z.ific - 1) * incj +
                             (long long)(k + z.ifick - 1) * inck;
        double wge = 0.;
        if (KFLU == 2 && !FIRST) wge = wig[le];
        double fe[5];
        double w = face_ij<KFLU, IFLOW, 0, FIRST>(z, sm, k, TI, lane, wge, fe);
        if (KFLU == 2 && FIRST) wig[le] = w;
#pragma unroll
        for (int e = 0; e < 5; e++) sm.fi_extra[e][lane] = fe[e];
      }
    }
    __syncthreads();
    // i+1/2 flux: neighbour lane, or the tile-edge value for the last lane
#pragma unroll
    for (int e = 0; e < 5; e++) {
      double v = __shfl_down(fim[e], 1, 64);
      fip[e] = lane == TI - 1 ? sm.fi_extra[e][row] : v;
      fjp[e] = sm.fj[e][row + 1][lane];
    }
    if (cell_ok) {
#pragma unroll
      for (int e = 0; e < 5; e++) {
        double a = d[e];
        a = a + fkm[e]; a = a + fjm[e]; a = a + fim[e];
        a = a - fip[e]; a = a - fjp[e]; a = a - fkp[e];
        d[e] = a;
        drodm[e * nd + lk] = a;
      }
      double o[5];
      fs_rk_update(z, coefW, qb, d, o);
#pragma unroll
      for (int e = 0; e < 5; e++) rop_out[e * nd + lk] = o[e];
    }
#pragma unroll
    for (int e = 0; e < 5; e++) fkm[e] = fkp[e];
    __syncthreads();   // every read of ring slot k-1, of the rho/T/mu plane and of the flux rows is done
    if (more) {
      const int s = slot_of(k + 2);
#pragma unroll
      for (int n = 0; n < NSTAGE; n++)
        if (s_lds[n] >= 0) {
#pragma unroll
          for (int e = 0; e < 3; e++) sm.uvw[s][e][s_lds[n]] = s_uvw[n][e];
          sm.rtx[0][s_lds[n]] = s_rtx[n][0];
          sm.rtx[1][s_lds[n]] = s_rtx[n][1];
          if (IFLOW == 2) sm.rtx[2][s_lds[n]] = s_rtx[n][2];
        }
      __syncthreads();
    }
  }
}

template <int KFLU, int IFLOW>
void launch(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
            double *drodm, const double *xmut, double *wig, hipStream_t s)
{
  // the face i = iv+1 is evaluated by lane iv - i0 of the last tile, or by its edge pass
  const int nti = (z.iv + TI - 1) / TI;
  const int ntj = (z.jv + TJ - 1) / TJ;
  int kchunk = 32;
  if (z.kv < kchunk) kchunk = z.kv;
  const int ntk = (z.kv + kchunk - 1) / kchunk;
  const int nblk = nti * ntj * ntk;
  if (first)
    hipLaunchKernelGGL((k_stage_tile<KFLU, IFLOW, true>), dim3(nblk), dim3(NTHREADS), 0, s, z, coefH, coefW, kchunk,
                       nti, ntj, nblk, rop, rop_out, drodm, xmut, wig);
  else
    hipLaunchKernelGGL((k_stage_tile<KFLU, IFLOW, false>), dim3(nblk), dim3(NTHREADS), 0, s, z, coefH, coefW, kchunk,
                       nti, ntj, nblk, rop, rop_out, drodm, xmut, wig);
}

}  // namespace

int fsk_stage_tiled(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                    double *drodm, const double *xmut, double *wig, hipStream_t s)
{
#define FS_CASE(K, F)                                                                       \
  if (z.kflu == K && z.iflow == F) {                                                        \
    launch<K, F>(z, coefH, coefW, first, rop, rop_out, drodm, xmut, wig, s);                \
    return 0;                                                                               \
  }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
#undef FS_CASE
  return -1;
}
