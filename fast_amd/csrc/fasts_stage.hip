// Tiled Runge-Kutta stage kernel (flux balance + update) for Cartesian zones, gfx950.
//
// Work decomposition ("2.5-D marching"): a workgroup of 4 waves owns a tile of 64 (i) x 4 (j)
// cell columns and marches through a chunk of k-planes.  One wave = one j-row, one lane = one
// i, so every global access of a wave is a contiguous 512-byte row segment.
//   * each lane keeps rho and T of its own column at k-1..k+2 in registers (MUSCL stencil
//     along k);
//   * the in-plane neighbourhood lives in LDS: a ring of four planes of (u,v,w) with a 2-cell
//     halo (68 x 8 cells; planes k-1, k, k+1 in use, k+2 being filled) for the MUSCL states and
//     the cross-derivative terms of the viscous flux, and one plane of (rho, T, mu) for the
//     current k;
//   * every face flux is evaluated once: a lane computes the faces j-1/2, i-1/2 and k+1/2 of its
//     cell; the k-1/2 flux is carried over from the previous plane, the i+1/2 flux comes from
//     the neighbouring lane by a wave shuffle, the j+1/2 flux from the next row through LDS.
//     The faces on the far tile edges (i0+64 and j0+4) are evaluated by one extra masked pass
//     each, rotated over the waves.
// The six fluxes are then summed in the order +Fk- +Fj- +Fi- -Fi+ -Fj+ -Fk+ (the order in which
// the reference's scatter loops reach a cell, template_FluxAndBalance.for:192-271) and the
// Runge-Kutta update (core3ark3.for:108-154) is applied in the same kernel, so a stage reads
// the state once and writes the balance and the new state once.
//
// Addressing: every global access is (wave-uniform base pointer) + (32-bit per-lane byte offset
// inside one k-plane), so an address costs one VGPR and the uniform part lives in SGPRs.
#include "fasts_kernels.h"
#include "fasts_flux.h"

namespace {

constexpr int TI = 64, TJ = 4, HALO = 2;
constexpr int PW = TI + 2 * HALO;      // 68
constexpr int PH = TJ + 2 * HALO;      // 8
constexpr int PLANE = PW * PH;         // 544 cells
constexpr int NTHREADS = TI * TJ;      // 256
constexpr int NHALO = PLANE - NTHREADS; // 288 halo cells around the tile
constexpr int NSTAGE = (NHALO + NTHREADS - 1) / NTHREADS;  // 2 halo cells per thread per plane-variable

struct Smem {
  double uvw[4][3][PLANE];   // [ring slot][u,v,w][cell]
  double rtx[3][PLANE];      // rho, T, mu of the current plane
  double fj[5][TJ + 1][TI];  // j-face fluxes of rows 0..TJ
  double fi_extra[5][TJ];    // i-face flux at tile column TI
};

FS_DEV int lidx(int ti, int tj) { return (tj + HALO) * PW + (ti + HALO); }
FS_DEV int slot_of(int kk) { return (kk + 4) & 3; }  // kk >= -4

FS_DEV double ldg(const double *base, unsigned boff)
{
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + boff);
}
FS_DEV void stg(double *base, unsigned boff, double v)
{
  *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + boff) = v;
}

// inviscid + viscous flux of an i- or j-face whose right cell is tile cell c of plane k; all
// operands from LDS.  Returns the sensor value used (computed when FIRST).
template <int KFLU, int IFLOW, int DIR, bool FIRST>
FS_DEV double face_ij(const FsZone &z, const Smem &sm, int k, int c, double wg_in, double *flu)
{
  static_assert(DIR == 0 || DIR == 1, "in-plane faces only");
  const int sk = slot_of(k);
  constexpr int st = DIR == 0 ? 1 : PW;
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
             const double *__restrict__ xmut, unsigned char *__restrict__ wig)
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
  const long long nd = z.nd, inck = z.inck;
  const int incj = (int)z.incj;
  const bool cell_ok = (i <= z.iv) && (j <= z.jv);
  const bool fi_ok = (i <= z.iv + 1) && (j <= z.jv);    // lane evaluates its i-1/2 face
  const bool fj_ok = (i <= z.iv) && (j <= z.jv + 1);    // lane evaluates its j-1/2 face
  // in-plane byte offset of the own column, clamped into the array for lanes outside the zone
  const int ic = i <= z.iv + 2 ? i : z.iv + 2, jc = j <= z.jv + 2 ? j : z.jv + 2;
  const unsigned ob = (unsigned)((ic + z.ific - 1) + (jc + z.ific - 1) * incj) * 8u;
  // wave-uniform base of plane kk of variable e
  auto P = [&](const double *a, int e, int kk) -> const double * {
    return a + ((long long)e * nd + (long long)(kk + z.ifick - 1) * inck);
  };
  auto PW_ = [&](double *a, int e, int kk) -> double * {
    return a + ((long long)e * nd + (long long)(kk + z.ifick - 1) * inck);
  };
  // face sensors are 0 or 1: kept as one byte per cell and direction
  auto ldw = [&](int d, int kk, unsigned boff) -> double {
    const unsigned char *b = wig + ((long long)d * nd + (long long)(kk + z.ifick - 1) * inck);
    return (double)b[boff >> 3];
  };
  auto stw = [&](int d, int kk, unsigned boff, double w) {
    unsigned char *b = wig + ((long long)d * nd + (long long)(kk + z.ifick - 1) * inck);
    b[boff >> 3] = (unsigned char)(w != 0.);
  };
  // staging map.  The 256 tile cells of a plane are written to LDS by their own lanes from
  // registers; the 288 halo cells (rows -2,-1,TJ,TJ+1 over the full width, then columns
  // -2,-1,TI,TI+1 of the tile rows) are fetched by thread h and, for h < 32, thread h + 256.
  unsigned sb[NSTAGE];
  int sl[NSTAGE];
#pragma unroll
  for (int n = 0; n < NSTAGE; n++) {
    int h = threadIdx.x + n * NTHREADS;
    if (h >= NHALO) h = 0;
    int tj, ti;
    if (h < 4 * PW) {
      const int rr = h / PW;
      tj = rr < HALO ? rr - HALO : rr - HALO + TJ;
      ti = h % PW - HALO;
    } else {
      const int x = h - 4 * PW;
      tj = x / (2 * HALO);
      const int xx = x % (2 * HALO);
      ti = xx < HALO ? xx - HALO : xx - HALO + TI;
    }
    sl[n] = lidx(ti, tj);
    int gi = i0 + ti + 1, gj = j0 + tj + 1;
    gi = gi <= z.iv + 2 ? gi : z.iv + 2;
    gj = gj <= z.jv + 2 ? gj : z.jv + 2;
    sb[n] = (unsigned)((gi + z.ific - 1) + (gj + z.ific - 1) * incj) * 8u;
  }
  const bool s_last = threadIdx.x + (NSTAGE - 1) * NTHREADS < NHALO;   // second halo cell exists
  const int ct = lidx(lane, row);
  auto stage_uvw = [&](int kk) {   // direct global -> LDS of plane kk (u,v,w)
    const int s = slot_of(kk);
#pragma unroll
    for (int e = 0; e < 3; e++) sm.uvw[s][e][ct] = ldg(P(rop, e + 1, kk), ob);
#pragma unroll
    for (int n = 0; n < NSTAGE; n++)
      if (n < NSTAGE - 1 || s_last) {
#pragma unroll
        for (int e = 0; e < 3; e++) sm.uvw[s][e][sl[n]] = ldg(P(rop, e + 1, kk), sb[n]);
      }
  };
  auto stage_rtx = [&](int kk) {
    sm.rtx[0][ct] = ldg(P(rop, 0, kk), ob);
    sm.rtx[1][ct] = ldg(P(rop, 4, kk), ob);
    if (IFLOW == 2) sm.rtx[2][ct] = ldg(P(xmut, 0, kk), ob);
#pragma unroll
    for (int n = 0; n < NSTAGE; n++)
      if (n < NSTAGE - 1 || s_last) {
        sm.rtx[0][sl[n]] = ldg(P(rop, 0, kk), sb[n]);
        sm.rtx[1][sl[n]] = ldg(P(rop, 4, kk), sb[n]);
        if (IFLOW == 2) sm.rtx[2][sl[n]] = ldg(P(xmut, 0, kk), sb[n]);
      }
  };

  // ---- prologue: planes k0-1, k0, k0+1 in LDS; rho,T of the own column k0-2 .. k0+1 in registers -
  stage_uvw(k0 - 1); stage_uvw(k0); stage_uvw(k0 + 1);
  stage_rtx(k0);
  // own column: rho and T at k-1, k, k+1, k+2 (u,v,w of k-1..k+1 are read from the LDS ring)
  double r_a, r_b, r_c, r_d, t_a, t_b, t_c, t_d;
  double xm_k = 0., xm_kp1 = 0.;
  r_a = ldg(P(rop, 0, k0 - 2), ob); t_a = ldg(P(rop, 4, k0 - 2), ob);
  r_b = ldg(P(rop, 0, k0 - 1), ob); t_b = ldg(P(rop, 4, k0 - 1), ob);
  r_c = ldg(P(rop, 0, k0), ob);     t_c = ldg(P(rop, 4, k0), ob);
  r_d = ldg(P(rop, 0, k0 + 1), ob); t_d = ldg(P(rop, 4, k0 + 1), ob);
  double far[3];   // u,v,w of the plane outside the ring (k0-2 here, k+2 in the loop)
#pragma unroll
  for (int e = 0; e < 3; e++) far[e] = ldg(P(rop, e + 1, k0 - 2), ob);
  if (IFLOW == 2) { xm_k = ldg(P(xmut, 0, k0 - 1), ob); xm_kp1 = ldg(P(xmut, 0, k0), ob); }
  // sensor values of the faces of the next plane are fetched one step ahead (stages 2 and 3)
  double wgi_n = 0., wgj_n = 0., wgk_n = 0.;
  if (KFLU == 2 && !FIRST) {
    wgk_n = ldw(2, k0, ob);
    wgi_n = ldw(0, k0, ob);
    wgj_n = ldw(1, k0, ob);
  }
  __syncthreads();

  // flux through the k-face between planes kk and kk+1 of the own column; the four column
  // states m2, m1, p0, p1 are planes kk-1, kk, kk+1, kk+2; viscosity at kk, kk+1
  auto kface = [&](int kk, const double *m2, const double *m1, const double *p0, const double *p1, double mul,
                   double mur, double wg_in, double *flu) {
    double wg = wg_in;
    if (KFLU == 2 && FIRST) {
      auto qs = [&](int e, int o) -> double { return o == -2 ? m2[e] : (o == -1 ? m1[e] : (o == 0 ? p0[e] : p1[e])); };
      wg = fs_wiggle(qs);
      if (cell_ok) stw(2, kk + 1, ob, wg);   // the face belongs to cell kk+1
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
  {
    double m2[5], m1[5], p0[5], p1[5];
    m2[0] = r_a; m2[4] = t_a; m1[0] = r_b; m1[4] = t_b; p0[0] = r_c; p0[4] = t_c; p1[0] = r_d; p1[4] = t_d;
#pragma unroll
    for (int e = 0; e < 3; e++) {
      m2[e + 1] = far[e];
      m1[e + 1] = sm.uvw[slot_of(k0 - 1)][e][ct];
      p0[e + 1] = sm.uvw[slot_of(k0)][e][ct];
      p1[e + 1] = sm.uvw[slot_of(k0 + 1)][e][ct];
    }
    kface(k0 - 1, m2, m1, p0, p1, xm_k, xm_kp1, wgk_n, fkm);   // flux at k0 - 1/2
  }
  // tile-edge passes: byte offsets of row j0+TJ+1 (own lane) and of column i0+TI+1 (row = lane)
  const unsigned ob_je = ob + (unsigned)((TJ - row) * incj) * 8u;
  const unsigned ob_ie = (unsigned)((i0 + TI + z.ific) + (j0 + (lane & (TJ - 1)) + z.ific) * incj) * 8u;
  const bool edge_j = (j0 + TJ + 1 <= z.jv + 1) && i <= z.iv;
  const bool edge_i = lane < TJ && (i0 + TI + 1 <= z.iv + 1) && (j0 + lane + 1 <= z.jv);

  for (int k = k0; k <= k1; k++) {
    // shift the rho/T window: r_a..r_d = k-1, k, k+1, k+2
    r_a = r_b; r_b = r_c; r_c = r_d; t_a = t_b; t_b = t_c; t_c = t_d;
    xm_k = xm_kp1;
    r_d = ldg(P(rop, 0, k + 2), ob); t_d = ldg(P(rop, 4, k + 2), ob);
#pragma unroll
    for (int e = 0; e < 3; e++) far[e] = ldg(P(rop, e + 1, k + 2), ob);
    if (IFLOW == 2) xm_kp1 = ldg(P(xmut, 0, k + 1), ob);
    const double wgi = wgi_n, wgj = wgj_n;
    double wgk = 0.;
    if (KFLU == 2 && !FIRST) {
      wgk = ldw(2, k + 1, ob);
      wgi_n = ldw(0, k + 1, ob);
      wgj_n = ldw(1, k + 1, ob);
    }
    // sensors of the tile-edge faces handled by this wave in this step
    const int duty = (k - k0) & 3;
    const bool duty_j = row == duty && edge_j;
    const bool duty_i = row == ((duty + 2) & 3) && edge_i;
    double wge_j = 0., wge_i = 0.;
    if (KFLU == 2 && !FIRST) {
      if (duty_j) wge_j = ldw(1, k, ob_je);
      if (duty_i) wge_i = ldw(0, k, ob_ie);
    }
    // plane k+2 of (u,v,w) goes to the free ring slot as soon as the j-face is done; plane k+1 of
    // (rho,T,mu) is requested before the k-face and replaces plane k after the faces are done
    double s_uvw[NSTAGE][3], s_rtx[NSTAGE][3];
    const bool more = k < k1;
    if (more) {
#pragma unroll
      for (int n = 0; n < NSTAGE; n++)
        if (n < NSTAGE - 1 || s_last) {
#pragma unroll
          for (int e = 0; e < 3; e++) s_uvw[n][e] = ldg(P(rop, e + 1, k + 2), sb[n]);
        }
    }

    double fkp[5], fjm[5] = {0., 0., 0., 0., 0.}, fim[5] = {0., 0., 0., 0., 0.}, fip[5], fjp[5];
    if (fj_ok) {
      double w = face_ij<KFLU, IFLOW, 1, FIRST>(z, sm, k, ct, wgj, fjm);
      if (KFLU == 2 && FIRST) stw(1, k, ob, w);
#pragma unroll
      for (int e = 0; e < 5; e++) sm.fj[e][row][lane] = fjm[e];
    }
    if (more) {
      const int s = slot_of(k + 2);
#pragma unroll
      for (int e = 0; e < 3; e++) sm.uvw[s][e][ct] = far[e];
#pragma unroll
      for (int n = 0; n < NSTAGE; n++)
        if (n < NSTAGE - 1 || s_last) {
#pragma unroll
          for (int e = 0; e < 3; e++) sm.uvw[s][e][sl[n]] = s_uvw[n][e];
        }
    }
    if (fi_ok) {
      double w = face_ij<KFLU, IFLOW, 0, FIRST>(z, sm, k, ct, wgi, fim);
      if (KFLU == 2 && FIRST) stw(0, k, ob, w);
    }
    // far-edge faces of the tile (row j0+TJ+1 and column i0+TI+1 of the zone): one masked pass
    // each, rotated over the waves
    if (duty_j) {
      double fe[5];
      double w = face_ij<KFLU, IFLOW, 1, FIRST>(z, sm, k, lidx(lane, TJ), wge_j, fe);
      if (KFLU == 2 && FIRST) stw(1, k, ob_je, w);
#pragma unroll
      for (int e = 0; e < 5; e++) sm.fj[e][TJ][lane] = fe[e];
    }
    if (duty_i) {
      double fe[5];
      double w = face_ij<KFLU, IFLOW, 0, FIRST>(z, sm, k, lidx(TI, lane), wge_i, fe);
      if (KFLU == 2 && FIRST) stw(0, k, ob_ie, w);
#pragma unroll
      for (int e = 0; e < 5; e++) sm.fi_extra[e][lane] = fe[e];
    }
    if (more) {
#pragma unroll
      for (int n = 0; n < NSTAGE; n++)
        if (n < NSTAGE - 1 || s_last) {
          s_rtx[n][0] = ldg(P(rop, 0, k + 1), sb[n]);
          s_rtx[n][1] = ldg(P(rop, 4, k + 1), sb[n]);
          if (IFLOW == 2) s_rtx[n][2] = ldg(P(xmut, 0, k + 1), sb[n]);
        }
    }
    // previous balance of the cell (stages 2, 3): in flight while the k-face is evaluated
    double dr[5];
    if (!FIRST) {
#pragma unroll
      for (int e = 0; e < 5; e++) dr[e] = ldg(P(drodm, e, k), ob);
    }
    // k-face last: its far operands (plane k+2) were requested at the top of the step
    double qk[5];   // state of the cell itself, for the update
    {
      double m2[5], p0[5], p1[5];
      m2[0] = r_a; m2[4] = t_a; qk[0] = r_b; qk[4] = t_b; p0[0] = r_c; p0[4] = t_c; p1[0] = r_d; p1[4] = t_d;
#pragma unroll
      for (int e = 0; e < 3; e++) {
        m2[e + 1] = sm.uvw[slot_of(k - 1)][e][ct];
        qk[e + 1] = sm.uvw[slot_of(k)][e][ct];
        p0[e + 1] = sm.uvw[slot_of(k + 1)][e][ct];
        p1[e + 1] = far[e];
      }
      kface(k, m2, qk, p0, p1, xm_k, xm_kp1, wgk, fkp);
    }
    __syncthreads();   // faces of plane k done: the rho/T/mu plane is free, flux rows are complete
    if (more) {
      sm.rtx[0][ct] = r_c;
      sm.rtx[1][ct] = t_c;
      if (IFLOW == 2) sm.rtx[2][ct] = xm_kp1;
#pragma unroll
      for (int n = 0; n < NSTAGE; n++)
        if (n < NSTAGE - 1 || s_last) {
          sm.rtx[0][sl[n]] = s_rtx[n][0];
          sm.rtx[1][sl[n]] = s_rtx[n][1];
          if (IFLOW == 2) sm.rtx[2][sl[n]] = s_rtx[n][2];
        }
    }
    // i+1/2 flux: neighbour lane, or the tile-edge value for the last lane
#pragma unroll
    for (int e = 0; e < 5; e++) {
      double v = __shfl_down(fim[e], 1, 64);
      fip[e] = lane == TI - 1 ? sm.fi_extra[e][row] : v;
      fjp[e] = sm.fj[e][row + 1][lane];
    }
    if (cell_ok) {
      double d[5];
#pragma unroll
      for (int e = 0; e < 5; e++) {
        double a = FIRST ? 0. : dr[e] * coefH;
        a = a + fkm[e]; a = a + fjm[e]; a = a + fim[e];
        a = a - fip[e]; a = a - fjp[e]; a = a - fkp[e];
        d[e] = a;
        stg(PW_(drodm, e, k), ob, a);
      }
      double o[5];
      fs_rk_update(z, coefW, qk, d, o);
#pragma unroll
      for (int e = 0; e < 5; e++) stg(PW_(rop_out, e, k), ob, o[e]);
    }
#pragma unroll
    for (int e = 0; e < 5; e++) fkm[e] = fkp[e];
    __syncthreads();   // staged planes visible, flux rows consumed
  }
}

template <int KFLU, int IFLOW>
void launch(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
            double *drodm, const double *xmut, unsigned char *wig, hipStream_t s)
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
                    double *drodm, const double *xmut, unsigned char *wig, hipStream_t s)
{
  // one k-plane must stay below 4 GiB / 8 so that in-plane byte offsets fit 32 bits
  if ((long long)z.ni * z.nj >= (1ll << 28)) return -2;
#define FS_CASE(K, F)                                                                       \
  if (z.kflu == K && z.iflow == F) {                                                        \
    launch<K, F>(z, coefH, coefW, first, rop, rop_out, drodm, xmut, wig, s);                \
    return 0;                                                                               \
  }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
#undef FS_CASE
  return -1;
}
