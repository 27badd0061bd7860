// Runge-Kutta stage kernel (flux balance + update) for Cartesian zones, gfx950 -- planes staged
// by LDS-DMA.
//
// A workgroup of 4 waves owns a tile of 32 (i) x 8 (j) cell columns and marches through a chunk
// of k-planes.  Half a wave = one j-row, so a wave access is two contiguous 256-byte row segments.
//
// Staging.  The in-plane neighbourhood of a tile (2-cell halo, 36 x 12 cells) lives in LDS:
//   * a ring of four planes of (u,v,w): planes k-1, k, k+1 are in use while plane k+2 arrives;
//   * two planes of (rho, T, mu): plane k in use while plane k+1 arrives.
// The arriving planes are written by global_load_lds (no destination registers): each lane moves one
// dword, a wave instruction fills 256 contiguous bytes of the plane image, the source address is per
// lane (clamped into the array for tiles that overhang the zone).  They are requested at the top of
// step k and first read in step k+1, so they have a whole step of arithmetic to arrive.  Everything
// else a step needs from global memory (rho,T of the own column two planes up, the previous balance,
// the sensor bytes) is also requested one step ahead.  The mid-step barrier therefore waits for LDS
// traffic only; the end-of-step __syncthreads() is the single point where memory is waited for.
//
// Arithmetic.  Step k evaluates the k-face between planes k-1 and k (own-column MUSCL states from
// registers for k-2, k-1 and from LDS for k, k+1), finishes cell (k-1) with it, then evaluates the i-
// and j-faces of plane k.  Every face flux is evaluated once: i+1/2 comes from the neighbour lane by a
// shuffle, j+1/2 from the next row through LDS, the faces on the far tile edges by one masked pass
// each, rotated over the waves.  The six fluxes of a cell are summed in the order
// +Fk- +Fj- +Fi- -Fi+ -Fj+ -Fk+ (the order in which the reference's scatter loops reach a cell,
// template_FluxAndBalance.for:192-271) and the Runge-Kutta update (core3ark3.for:108-154) follows in
// the same kernel.
#include "fasts_kernels.h"
#include "fasts_flux.h"
#include <cstdint>
#include <cstdlib>

#ifdef FS_STAMP
// development aid: per-phase shader-clock stamps of the four waves of workgroup 0
__device__ long long fs_stamp_buf[4 * 40 * 8];
#define STAMP(slot)                                                                              \
  if (blockIdx.x == 0 && lane == 0 && (k - k0) < 40) fs_stamp_buf[(wave * 40 + (k - k0)) * 8 + (slot)] = clock64();
#else
#define STAMP(slot)
#endif

namespace {

constexpr int TI = 32, TJ = 8, HALO = 2;
constexpr int PW = TI + 2 * HALO;      // 36
constexpr int PH = TJ + 2 * HALO;      // 12
constexpr int PLANE = PW * PH;         // 432 cells
constexpr int NTHREADS = TI * TJ;      // 256
constexpr int PLANE_DW = 2 * PLANE;    // 864 dwords per plane and variable
constexpr int NDMA = (PLANE_DW + NTHREADS - 1) / NTHREADS;   // 4 rounds of 256 dwords

struct alignas(16) Smem {
  double uvw[4][3][PLANE];   // [ring slot][u,v,w][cell]
  double rtx[2][3][PLANE];   // [buffer][rho,T,mu][cell]
  double fj[5][TJ + 1][TI];  // j-face fluxes of rows 0..TJ
  double fi_extra[5][TJ];    // i-face flux at tile column TI
};

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

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
  const int sk = slot_of(k), cur = k & 1;
  constexpr int st = DIR == 0 ? 1 : PW;
  double qm2[5], qm1[5], q0[5], qp1[5];
  const double *ro = sm.rtx[cur][0], *tt = sm.rtx[cur][1];
  qm2[0] = ro[c - 2 * st]; qm1[0] = ro[c - st]; q0[0] = ro[c]; qp1[0] = ro[c + st];
  qm2[4] = tt[c - 2 * st]; qm1[4] = tt[c - st]; q0[4] = tt[c]; qp1[4] = tt[c + st];
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
      if (e == 4) return tt[cc];
      const int s = db == 0 ? sk : (db < 0 ? skm : skp);
      return sm.uvw[s][e - 1][cc];
    };
    fs_flux_viscous<DIR>(z, q, sm.rtx[cur][2][c - st], sm.rtx[cur][2][c], flu);
  }
  return wg;
}

// BDF = false: Runge-Kutta stage (drodm = balance, rop_out = updated state).
// BDF = true: Newton residual of the implicit scheme (drodm = dt/vol * balance + BDF terms, rop_out unused).
template <int KFLU, int IFLOW, bool FIRST, bool WIDE, bool BDF>
__global__ void __launch_bounds__(NTHREADS, 2)
k_stage_dma(FsZone z, double coefH, double coefW, int kchunk, int ntile_i, int ntile_j, int nblk,
            const double *__restrict__ rop, double *__restrict__ rop_out, double *__restrict__ drodm,
            const double *__restrict__ xmut, unsigned char *__restrict__ wig, const int *__restrict__ blist,
            FsBdf bdf)
{
  __shared__ Smem sm;
  // blocks that share an XCD (same blockIdx % 8) get consecutive tiles, so halo rows of
  // neighbouring tiles meet in one L2
  int bid = blockIdx.x;
  {
    const int q8 = nblk / 8, r8 = nblk % 8, x = bid % 8;
    bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + bid / 8;
  }
  if (blist) bid = blist[bid];   // a subset of the tiles (tile number = bi + nti*(bj + ntj*bk))
  const int bi = bid % ntile_i;
  const int bj = (bid / ntile_i) % ntile_j;
  const int bk = bid / (ntile_i * ntile_j);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ti = lane & (TI - 1), tj = 2 * wave + (lane >> 5);
  const int i0 = bi * TI, j0 = bj * TJ;             // tile cell (0,0) = interior cell (i0+1, j0+1)
  const int k0 = bk * kchunk + 1;
  const int k1 = (k0 + kchunk - 1 < z.kv) ? k0 + kchunk - 1 : z.kv;
  const int i = i0 + ti + 1, j = j0 + tj + 1;
  const long long nd = z.nd, inck = z.inck;
  const int incj = (int)z.incj;
  const int ni = z.ni, nj = z.nj;
  const bool cell_ok = (i <= z.iv) && (j <= z.jv);
  const bool fi_ok = (i <= z.iv + 1) && (j <= z.jv);    // lane evaluates its i-1/2 face
  const bool fj_ok = (i <= z.iv) && (j <= z.jv + 1);    // lane evaluates its j-1/2 face
  // in-plane byte offset of a cell given by interior-based indices, clamped into the array
  auto boff = [&](int ii, int jj) -> unsigned {
    int si = ii + z.ific - 1, sj = jj + z.ific - 1;
    si = si < 0 ? 0 : (si > ni - 1 ? ni - 1 : si);
    sj = sj < 0 ? 0 : (sj > nj - 1 ? nj - 1 : sj);
    return (unsigned)(si + sj * incj) * 8u;
  };
  const unsigned ob = boff(i, j);
  // wave-uniform base of plane kk of variable e
  auto P = [&](const double *a, int e, int kk) -> const double * {
    return a + ((long long)e * nd + (long long)(kk + z.ifick - 1) * inck);
  };
  auto PW_ = [&](double *a, int e, int kk) -> double * {
    return a + ((long long)e * nd + (long long)(kk + z.ifick - 1) * inck);
  };
  // face sensors are 0 or 1: one byte per cell and direction.  Raw bytes are fetched a step ahead
  // and widened when used.
  auto ldw = [&](int d, int kk, unsigned bo) -> unsigned {
    const unsigned char *b = wig + ((long long)d * nd + (long long)(kk + z.ifick - 1) * inck);
    return b[bo >> 3];
  };
  auto stw = [&](int d, int kk, unsigned bo, double w) {
    unsigned char *b = wig + ((long long)d * nd + (long long)(kk + z.ifick - 1) * inck);
    b[bo >> 3] = (unsigned char)(w != 0.);
  };

  // ---- LDS-DMA map ------------------------------------------------------------------------------
  // WIDE: lane q = wave*64 + lane moves the 16 bytes of cells (2q, 2q+1) of a plane image, one round
  // of 216 lanes per plane (needs even row length and 16-byte aligned arrays).  Otherwise dword
  // d = (n*4 + wave)*64 + lane, four rounds.
  unsigned goff[WIDE ? 1 : NDMA];
  bool dma_act;
  if constexpr (WIDE) {
    int q = wave * 64 + lane;
    dma_act = q < PLANE / 2;
    if (!dma_act) q = PLANE / 2 - 1;
    const int c = 2 * q, r = c / PW, cc = c % PW;
    int si = i0 + cc - HALO + z.ific, sj = j0 + r - HALO + z.ific;
    si = si < 0 ? 0 : (si > ni - 2 ? ni - 2 : si);
    sj = sj < 0 ? 0 : (sj > nj - 1 ? nj - 1 : sj);
    goff[0] = (unsigned)(si + sj * incj) * 8u;
  } else {
#pragma unroll
    for (int n = 0; n < NDMA; n++) {
      int d = (n * 4 + wave) * 64 + lane;
      if (d >= PLANE_DW) d = PLANE_DW - 1;
      const int c = d >> 1, r = c / PW, cc = c % PW;
      goff[n] = boff(i0 + cc - HALO + 1, j0 + r - HALO + 1) + (unsigned)(d & 1) * 4u;
    }
    dma_act = (3 * 4 + wave) * 64 + lane < PLANE_DW;   // lanes of the partial last round
  }
  auto dma_plane = [&](const double *gbase, double *lds_plane) {
    const char *g = reinterpret_cast<const char *>(gbase);
#ifdef FS_EXP_NODMA   // timing experiment only (wrong results): no plane staging
    if (k0 > 0) return;
#endif
    if constexpr (WIDE) {
      char *l = reinterpret_cast<char *>(lds_plane) + wave * 1024;
      if (dma_act) __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + goff[0]), (lds_ptr_t)(l), 16, 0, 0);
    } else {
      char *l = reinterpret_cast<char *>(lds_plane) + wave * 256;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + goff[0]), (lds_ptr_t)(l), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + goff[1]), (lds_ptr_t)(l + 1024), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + goff[2]), (lds_ptr_t)(l + 2048), 4, 0, 0);
      if (dma_act) __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + goff[3]), (lds_ptr_t)(l + 3072), 4, 0, 0);
    }
  };
  auto dma_uvw = [&](int kk) {
    const int s = slot_of(kk);
#pragma unroll
    for (int e = 0; e < 3; e++) dma_plane(P(rop, e + 1, kk), sm.uvw[s][e]);
  };
  auto dma_rtx = [&](int kk) {
    const int b = kk & 1;
    dma_plane(P(rop, 0, kk), sm.rtx[b][0]);
    dma_plane(P(rop, 4, kk), sm.rtx[b][1]);
    if (IFLOW == 2) dma_plane(P(xmut, 0, kk), sm.rtx[b][2]);
  };

  // ---- prologue ----------------------------------------------------------------------------------
  dma_uvw(k0 - 1); dma_uvw(k0); dma_uvw(k0 + 1);
  dma_rtx(k0);
  const int ct = lidx(ti, tj);
  double qa[5], qb[5];            // own column at k-2, k-1
  double mu_b = 0.;               // viscosity of the own column at k-1
#pragma unroll
  for (int e = 0; e < 5; e++) { qa[e] = ldg(P(rop, e, k0 - 2), ob); qb[e] = ldg(P(rop, e, k0 - 1), ob); }
  if (IFLOW == 2) mu_b = ldg(P(xmut, 0, k0 - 1), ob);
  double r_n = ldg(P(rop, 0, k0 + 1), ob), t_n = ldg(P(rop, 4, k0 + 1), ob);   // rho, T at k+1
  double dr_n[5] = {0., 0., 0., 0., 0.};   // previous balance of cell k (requested a step ahead)
  if (!FIRST) {
#pragma unroll
    for (int e = 0; e < 5; e++) dr_n[e] = ldg(P(drodm, e, k0), ob);
  }
  // tile-edge passes: byte offsets of row j0+TJ+1 (column = lane) and of column i0+TI+1 (row = lane)
  const unsigned ob_je = boff(i0 + (lane & (TI - 1)) + 1, j0 + TJ + 1);
  const unsigned ob_ie = boff(i0 + TI + 1, j0 + (lane & (TJ - 1)) + 1);
  const bool edge_j = lane < TI && (j0 + TJ + 1 <= z.jv + 1) && (i0 + lane + 1 <= z.iv);
  const bool edge_i = lane < TJ && (i0 + TI + 1 <= z.iv + 1) && (j0 + lane + 1 <= z.jv);
  unsigned wgi_n = 0, wgj_n = 0, wgk_n = 0, wge_n = 0;   // sensor bytes of the next step
  if (KFLU == 2 && !FIRST) {
    wgk_n = ldw(2, k0, ob);
    wgi_n = ldw(0, k0, ob);
    wgj_n = ldw(1, k0, ob);
    if (wave == 0 && edge_j) wge_n = ldw(1, k0, ob_je);
    if (wave == 2 && edge_i) wge_n = ldw(0, k0, ob_ie);
  }
  double part[5] = {0., 0., 0., 0., 0.};   // +Fk- +Fj- +Fi- -Fi+ -Fj+ of cell k-1 (and xi*drodm)
  __syncthreads();

  for (int k = k0; k <= k1 + 1; k++) {
    const bool last = k > k1;          // finishing step: only the k-face k1+1/2 and cell k1
    const int cur = k & 1;
    STAMP(0)
    // ---- requests for the next step --------------------------------------------------------------
    double r_nn = 0., t_nn = 0., dr_nn[5] = {0., 0., 0., 0., 0.};
    unsigned wgi_nn = 0, wgj_nn = 0, wgk_nn = 0, wge_nn = 0;
    const int duty = (k - k0) & 3, duty_nn = (k + 1 - k0) & 3;
    if (!last) {
      dma_uvw(k + 2);
      dma_rtx(k + 1);
#ifndef FS_EXP_NOREG
      r_nn = ldg(P(rop, 0, k + 2), ob); t_nn = ldg(P(rop, 4, k + 2), ob);
#endif
      if (!FIRST && !BDF) {
#pragma unroll
        for (int e = 0; e < 5; e++) dr_nn[e] = ldg(P(drodm, e, k + 1), ob);
      }
      if (KFLU == 2 && !FIRST) {
        wgk_nn = ldw(2, k + 1, ob);
        wgi_nn = ldw(0, k + 1, ob);
        wgj_nn = ldw(1, k + 1, ob);
        if (wave == duty_nn && edge_j) wge_nn = ldw(1, k + 1, ob_je);
        if (wave == ((duty_nn + 2) & 3) && edge_i) wge_nn = ldw(0, k + 1, ob_ie);
      }
    }
    STAMP(1)
    // ---- own column at k (LDS) and k+1 ------------------------------------------------------------
    double qc[5], qd[5];
    qc[0] = sm.rtx[cur][0][ct]; qc[4] = sm.rtx[cur][1][ct];
    const double mu_c = IFLOW == 2 ? sm.rtx[cur][2][ct] : 0.;
    qd[0] = r_n; qd[4] = t_n;
#pragma unroll
    for (int e = 0; e < 3; e++) {
      qc[e + 1] = sm.uvw[slot_of(k)][e][ct];
      qd[e + 1] = sm.uvw[slot_of(k + 1)][e][ct];
    }
    // ---- k-face between planes k-1 and k ----------------------------------------------------------
    double fk[5];
    {
      double wg = (double)wgk_n;
      if (KFLU == 2 && FIRST) {
        auto qs = [&](int e, int o) -> double { return o == -2 ? qa[e] : (o == -1 ? qb[e] : (o == 0 ? qc[e] : qd[e])); };
        wg = fs_wiggle(qs);
        if (cell_ok) stw(2, k, ob, wg);   // the face belongs to cell k
      }
      fs_flux_inviscid<KFLU, 2>(z, qa, qb, qc, qd, wg, fk);
      if (IFLOW == 2) {
        const int sl = slot_of(k - 1), sr = slot_of(k);
        auto q = [&](int e, int da, int db, int dn) -> double {
          if (e == 4) return dn == 0 ? qc[4] : qb[4];
          return sm.uvw[dn == 0 ? sr : sl][e - 1][ct + da + db * PW];
        };
        fs_flux_viscous<2>(z, q, mu_b, mu_c, fk);
      }
    }
    STAMP(2)
    // ---- finish cell k-1 --------------------------------------------------------------------------
    if (k > k0 && cell_ok) {
      double d[5];
#pragma unroll
      for (int e = 0; e < 5; e++) d[e] = part[e] - fk[e];
      if (BDF) {
        double qn[5] = {0., 0., 0., 0., 0.}, qn1[5], r[5];
#pragma unroll
        for (int e = 0; e < 5; e++) qn1[e] = ldg(P(bdf.ro_n1, e, k - 1), ob);
        if (!bdf.nit1) {
#pragma unroll
          for (int e = 0; e < 5; e++) qn[e] = ldg(P(bdf.ro_n, e, k - 1), ob);
        }
        const double coe1 = bdf.coe1 ? ldg(P(bdf.coe1, 0, k - 1), ob) : z.coe;
        fs_bdf_residual(z, bdf, coe1, qb, qn, qn1, d, r);
#pragma unroll
        for (int e = 0; e < 5; e++) stg(PW_(drodm, e, k - 1), ob, r[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 5; e++) stg(PW_(drodm, e, k - 1), ob, d[e]);
        double o[5];
        fs_rk_update(z, coefW, qb, d, o);
#pragma unroll
        for (int e = 0; e < 5; e++) stg(PW_(rop_out, e, k - 1), ob, o[e]);
      }
    }
    STAMP(3)
    // ---- i- and j-faces of plane k ------------------------------------------------------------------
    double fjm[5] = {0., 0., 0., 0., 0.}, fim[5] = {0., 0., 0., 0., 0.};
    if (!last) {
      if (fj_ok) {
        double w = face_ij<KFLU, IFLOW, 1, FIRST>(z, sm, k, ct, (double)wgj_n, fjm);
        if (KFLU == 2 && FIRST) stw(1, k, ob, w);
#pragma unroll
        for (int e = 0; e < 5; e++) sm.fj[e][tj][ti] = fjm[e];
      }
      if (fi_ok) {
        double w = face_ij<KFLU, IFLOW, 0, FIRST>(z, sm, k, ct, (double)wgi_n, fim);
        if (KFLU == 2 && FIRST) stw(0, k, ob, w);
      }
      // far-edge faces of the tile (row j0+TJ+1 and column i0+TI+1 of the zone): one masked pass
      // each, rotated over the waves
      if (wave == duty && edge_j) {
        double fe[5];
        double w = face_ij<KFLU, IFLOW, 1, FIRST>(z, sm, k, lidx(lane, TJ), (double)wge_n, fe);
        if (KFLU == 2 && FIRST) stw(1, k, ob_je, w);
#pragma unroll
        for (int e = 0; e < 5; e++) sm.fj[e][TJ][lane] = fe[e];
      }
      if (wave == ((duty + 2) & 3) && edge_i) {
        double fe[5];
        double w = face_ij<KFLU, IFLOW, 0, FIRST>(z, sm, k, lidx(TI, lane), (double)wge_n, fe);
        if (KFLU == 2 && FIRST) stw(0, k, ob_ie, w);
#pragma unroll
        for (int e = 0; e < 5; e++) sm.fi_extra[e][lane] = fe[e];
      }
    }
    STAMP(4)
    // flux rows complete: LDS traffic only, the requests of this step stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(5)
    if (!last) {
#pragma unroll
      for (int e = 0; e < 5; e++) {
        const double v = __shfl_down(fim[e], 1, 64);
        const double fip = ti == TI - 1 ? sm.fi_extra[e][tj] : v;
        const double fjp = sm.fj[e][tj + 1][ti];
        double a = (FIRST || BDF) ? 0. : dr_n[e] * coefH;
        a = a + fk[e]; a = a + fjm[e]; a = a + fim[e];
        a = a - fip; a = a - fjp;
        part[e] = a;
      }
    }
    // ---- shift the column window ------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < 5; e++) { qa[e] = qb[e]; qb[e] = qc[e]; dr_n[e] = dr_nn[e]; }
    mu_b = mu_c;
    r_n = r_nn; t_n = t_nn;
    wgi_n = wgi_nn; wgj_n = wgj_nn; wgk_n = wgk_nn; wge_n = wge_nn;
    STAMP(6)
    __syncthreads();   // arriving planes and requests complete, flux rows consumed
    STAMP(7)
  }
}

template <int KFLU, int IFLOW>
void launch(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
            double *drodm, const double *xmut, unsigned char *wig, const int *blist, int nlist, const FsBdf *bdf,
            hipStream_t s)
{
  const FsTiling t = fsk_stage_dma_tiling(z);
  const int nblk = blist ? nlist : t.nti * t.ntj * t.ntk;
  if (nblk <= 0) return;
  // 16-byte DMA needs pairs of cells that are 16-byte aligned in every row and plane
  const bool wide = z.ni % 2 == 0 && z.nd % 2 == 0 && z.ific % 2 == 0 && ((uintptr_t)rop % 16 == 0) &&
                    (xmut == nullptr || (uintptr_t)xmut % 16 == 0);
  FsBdf b = {nullptr, nullptr, 0., 0., 0., 0, nullptr};
  if (bdf) b = *bdf;
#define FS_LAUNCH(FIRST_, WIDE_, BDF_)                                                                       \
  hipLaunchKernelGGL((k_stage_dma<KFLU, IFLOW, FIRST_, WIDE_, BDF_>), dim3(nblk), dim3(NTHREADS), 0, s, z, coefH, coefW, \
                     t.kchunk, t.nti, t.ntj, nblk, rop, rop_out, drodm, xmut, wig, blist, b)
  if (bdf) {
    if (first) { if (wide) FS_LAUNCH(true, true, true); else FS_LAUNCH(true, false, true); }
    else       { if (wide) FS_LAUNCH(false, true, true); else FS_LAUNCH(false, false, true); }
  } else {
    if (first) { if (wide) FS_LAUNCH(true, true, false); else FS_LAUNCH(true, false, false); }
    else       { if (wide) FS_LAUNCH(false, true, false); else FS_LAUNCH(false, false, false); }
  }
#undef FS_LAUNCH
}

}  // namespace

#ifdef FS_STAMP
extern "C" int fs_stamp_fetch(long long *host)
{
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fs_stamp_buf), sizeof(long long) * 4 * 40 * 8);
}
#endif

FsTiling fsk_stage_dma_tiling(const FsZone &z)
{
  FsTiling t;
  t.ti = TI; t.tj = TJ;
  // the face i = iv+1 is evaluated by lane iv - i0 of the last tile, or by its edge pass
  t.nti = (z.iv + TI - 1) / TI;
  t.ntj = (z.jv + TJ - 1) / TJ;
  t.kchunk = 32;
  if (const char *ev = getenv("FASTS_HIP_KCHUNK")) t.kchunk = atoi(ev) > 0 ? atoi(ev) : t.kchunk;   // tuning aid
  if (z.kv < t.kchunk) t.kchunk = z.kv;
  t.ntk = (z.kv + t.kchunk - 1) / t.kchunk;
  return t;
}

int fsk_stage_dma(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                  double *drodm, const double *xmut, unsigned char *wig, const int *blist, int nlist, const FsBdf *bdf,
                  hipStream_t s)
{
  // one k-plane must stay below 4 GiB / 8 so that in-plane byte offsets fit 32 bits
  if ((long long)z.ni * z.nj >= (1ll << 28)) return -2;
#define FS_CASE(K, F)                                                                       \
  if (z.kflu == K && z.iflow == F) {                                                        \
    launch<K, F>(z, coefH, coefW, first, rop, rop_out, drodm, xmut, wig, blist, nlist, bdf, s);  \
    return 0;                                                                               \
  }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
#undef FS_CASE
  return -1;
}
