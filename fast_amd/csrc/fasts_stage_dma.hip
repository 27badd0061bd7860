// Runge-Kutta stage kernel (flux balance + update) for Cartesian zones, gfx950 -- planes staged
// by LDS-DMA, waves specialised by role.
//
// A workgroup of 12 waves owns a tile of 64 (i) x 8 (j) cell columns and marches through a chunk
// of k-planes.  One workgroup fills a CU: three waves per SIMD at <= 168 registers (a CU does not
// admit two 6-wave workgroups of that size, it does admit one of 12 waves).
//   * waves 0..7 ("cell waves"): a wave = one j-row of 64 cells, one contiguous 512-byte segment per
//     access.  Each lane evaluates the three minus-side faces of its cell and finishes the cell.
//   * wave 8 ("j-edge wave"): the 64 j-faces on the far j edge of the tile (row TJ).
//   * wave 9 ("i-edge wave"): the 8 i-faces on the far i edge of the tile (column TI).
//   * waves 10, 11: staging only.
//   Waves 8..11 feed the plane ring (below), so that the cell waves issue no staging requests and all
//   eight of them do the same amount of work per plane: the workgroup barriers then wait for nobody
//   in particular.
//
// Staging.  The in-plane neighbourhood of a tile (2-cell halo, 68 x 12 cells) lives in LDS:
//   * a ring of four planes of (u,v,w): planes k-1, k, k+1 are in use while plane k+2 arrives;
//   * a ring of three planes of (rho, T): planes k, k+1 in use while plane k+2 arrives;
//   * two planes of mu: plane k in use while plane k+1 arrives.
// The arriving planes are written by global_load_lds (no destination registers): each lane moves 16
// bytes (4 when rows are not 16-byte aligned; then all twelve waves share the requests), the source
// address is per lane (clamped into the array for tiles that overhang the zone).  They are requested
// at the top of step k and first read in step k+1, so they have a whole step of arithmetic to arrive.
// Everything else a step needs from global memory (the previous balance, the sensor byte) is also
// requested one step ahead.  The mid-step barrier
// therefore waits for LDS traffic only; the end-of-step __syncthreads() is the single point where
// memory is waited for.
//
// Arithmetic.  Step k evaluates the k-face between planes k-1 and k (own-column MUSCL states from
// registers for k-2, k-1 and from LDS for k, k+1), finishes cell (k-1) with it, then evaluates the i-
// and j-faces of plane k.  Every face flux is evaluated once: i+1/2 comes from the neighbour lane by a
// DPP shift, j+1/2 from the next row through LDS, the faces on the far tile edges from the edge
// waves through LDS.  The six fluxes of a cell are summed in the order
// +Fk- +Fj- +Fi- -Fi+ -Fj+ -Fk+ (the order in which the reference's scatter loops reach a cell,
// template_FluxAndBalance.for:192-271) and the Runge-Kutta update (core3ark3.for:108-154) follows in
// the same kernel.
#include "fasts_kernels.h"
#include "fasts_flux.h"
#include <cstdint>
#include <cstdlib>

#ifndef FS_X_PRIO
#define FS_X_PRIO 0
#endif
#ifdef FS_X_LEGS
// development aid (timing only, results wrong): bits 8.. of the pace argument switch one leg of the kernel off:
// 1 the stores of the cell waves, 2 their loads, 4 the plane requests, 8 the face arithmetic
#define XF(bit) (((pace) >> 8) & (bit))
#define PACE ((pace) & 0xff)
#else
#define XF(bit) 0
#define PACE (pace)
#endif
#ifdef FS_STAMP
// development aid: per-phase shader-clock stamps of the waves of workgroup 0 (profiles/stamps.py)
__device__ long long fs_stamp_buf[12 * 40 * 8];
#define STAMP(slot)                                                                              \
  if (blockIdx.x == 0 && lane == 0 && (k - k0) < 40) fs_stamp_buf[(wave * 40 + (k - k0)) * 8 + (slot)] = clock64();
#else
#define STAMP(slot)
#endif

namespace {

constexpr int TI = 64, TJ = 8, HALO = 2;
constexpr int PW = TI + 2 * HALO;      // 68
constexpr int PH = TJ + 2 * HALO;      // 12
constexpr int PLANE = PW * PH;         // 816 cells
constexpr int NCELLW = TJ;             // cell waves, one per tile row
constexpr int NWAVES = 12;             // + j-edge wave + i-edge wave + two staging waves
constexpr int NTHREADS = 64 * NWAVES;  // 768
constexpr int NSTGW = NWAVES - NCELLW; // waves that issue the staging requests when rows are 16-byte aligned
constexpr int PLANE_DW = 2 * PLANE;    // 1632 dwords per plane and variable
constexpr int PLANE_QW = PLANE / 2;    // 408 16-byte units per plane and variable
constexpr int NDMA_WIDE = (PLANE_QW + 64 * NSTGW - 1) / (64 * NSTGW);   // 2 rounds of the four staging waves
constexpr int NDMA_DW = (PLANE_DW + NTHREADS - 1) / NTHREADS;         // 3 rounds of all twelve waves

struct alignas(16) Smem {
  double uvw[4][3][PLANE];   // [ring slot][u,v,w][cell]
  double rt[3][2][PLANE];    // [ring slot][rho,T][cell]
  double mu[2][PLANE];       // [buffer][cell]
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

// value of the next lane (lane + 1) of the wave; the last lane gets its own value back
FS_DEV double from_next_lane(double x)
{
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Loads of the cell waves that are requested one step ahead.  Issued by hand, so that the compiler does not count
// them: it would place its wait where a branch of the step has issued no younger request, i.e. wait for every
// outstanding request (vmcnt(0): the requests it has just issued and all the stores of the step before).  The
// destination registers count as written at the statement, so nothing may touch them before wait_ahead<N>,
// which names them: N = the requests this wave has issued after them.
// The destination is declared read-write: the request then goes into the very register the last wait_ahead has
// released, and the value stays in it until the next wait_ahead (no copy of a register that is still being loaded).
FS_DEV void ld_ahead(double &dst, const double *base, unsigned boff)
{
  asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(dst) : "v"(boff), "s"(base) : "memory");
}
FS_DEV void ld_ahead_byte(unsigned &dst, const unsigned char *base, unsigned off)
{
  asm volatile("global_load_ubyte %0, %1, %2" : "+v"(dst) : "v"(off), "s"(base) : "memory");
}
// A store of the cell waves placed by hand: after the value dep exists (it is not stored), so that the stores of a
// step leave one by one between the face fluxes and not in one burst.
FS_DEV void st_late(double *base, unsigned boff, double v, double dep)
{
  asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(boff), "v"(v), "s"(base), "v"(dep));
}

// nyounger (wave-uniform) = 0, 5 or 10.  One statement, so that the registers have one point of definition: with a
// statement per count the compiler joins them by copies, which it puts ahead of the waits.
FS_DEV void wait_ahead(double (&d)[5], unsigned &w, int nyounger)
{
  asm volatile("s_cmp_lt_u32 %6, 5\n\t"
               "s_cbranch_scc1 .Lw0_%=\n\t"
               "s_cmp_lt_u32 %6, 10\n\t"
               "s_cbranch_scc1 .Lw5_%=\n\t"
               "s_waitcnt vmcnt(10)\n\t"
               "s_branch .Lwe_%=\n"
               ".Lw5_%=:\n\t"
               "s_waitcnt vmcnt(5)\n\t"
               "s_branch .Lwe_%=\n"
               ".Lw0_%=:\n\t"
               "s_waitcnt vmcnt(0) ; wait_ahead\n"
               ".Lwe_%=:"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(w) : "s"(nyounger) : "memory", "scc");
}

// inviscid + viscous flux of an i- or j-face whose right cell is tile cell c of plane k; all
// operands from LDS (rs = ring slot of rho, T of plane k).  Returns the sensor value used (computed when FIRST).
template <int KFLU, int IFLOW, int DIR, bool FIRST>
FS_DEV double face_ij(const FsZone &z, const Smem &sm, int k, int rs, int c, double wg_in, double *flu, int xf = 0)
{
  static_assert(DIR == 0 || DIR == 1, "in-plane faces only");
  const int sk = slot_of(k), cur = k & 1;
  constexpr int st = DIR == 0 ? 1 : PW;
  double qm2[5], qm1[5], q0[5], qp1[5];
  const double *ro = sm.rt[rs][0], *tt = sm.rt[rs][1];
  qm2[0] = ro[c - 2 * st]; qm1[0] = ro[c - st]; q0[0] = ro[c]; qp1[0] = ro[c + st];
  qm2[4] = tt[c - 2 * st]; qm1[4] = tt[c - st]; q0[4] = tt[c]; qp1[4] = tt[c + st];
#pragma unroll
  for (int e = 1; e <= 3; e++) {
    const double *p = sm.uvw[sk][e - 1];
    qm2[e] = p[c - 2 * st]; qm1[e] = p[c - st]; q0[e] = p[c]; qp1[e] = p[c + st];
  }
  double wg = wg_in;
#ifdef FS_X_LEGS
  if (xf & 8) {
#pragma unroll
    for (int e = 0; e < 5; e++) flu[e] = qm1[e] + q0[e];
    return wg;
  }
#endif
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
    fs_flux_viscous<DIR>(z, q, sm.mu[cur][c - st], sm.mu[cur][c], flu);
  }
  return wg;
}

// BDF = false: Runge-Kutta stage (drodm = balance, rop_out = updated state).
// BDF = true: Newton residual of the implicit scheme (drodm = dt/vol * balance + BDF terms, rop_out unused).
template <int KFLU, int IFLOW, bool FIRST, bool WIDE, bool BDF>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_stage_dma(FsZone z, double coefH, double coefW, int kchunk, int ntile_i, int ntile_j, int nblk, int pace, int keep,
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
  // tile cell of the lane: cell waves (ti, tj) in 0..TI-1 x 0..TJ-1; j-edge wave (ti, TJ); i-edge wave (TI, tj)
  const int ti = wave == NCELLW + 1 ? TI : lane;
  const int tj = wave < NCELLW ? wave : (wave == NCELLW ? TJ : (lane & (TJ - 1)));
  const int i0 = bi * TI, j0 = bj * TJ;             // tile cell (0,0) = interior cell (i0+1, j0+1)
  const int k0 = bk * kchunk + 1;
  const int k1 = (k0 + kchunk - 1 < z.kv) ? k0 + kchunk - 1 : z.kv;
  const int i = i0 + ti + 1, j = j0 + tj + 1;
  const long long nd = z.nd, inck = z.inck;
  const int incj = (int)z.incj;
  const int ni = z.ni, nj = z.nj;
  // in-plane byte offset of a cell given by interior-based indices, clamped into the array
  auto boff = [&](int ii, int jj) -> unsigned {
    int si = ii + z.ific - 1, sj = jj + z.ific - 1;
    si = si < 0 ? 0 : (si > ni - 1 ? ni - 1 : si);
    sj = sj < 0 ? 0 : (sj > nj - 1 ? nj - 1 : sj);
    return (unsigned)(si + sj * incj) * 8u;
  };
  const unsigned ob = boff(i, j);
  const int ct = lidx(ti, tj);
  // wave-uniform base of plane kk of variable e
  auto P = [&](const double *a, int e, int kk) -> const double * {
    return a + ((long long)e * nd + (long long)(kk + z.ifick - 1) * inck);
  };
  auto PW_ = [&](double *a, int e, int kk) -> double * {
    return a + ((long long)e * nd + (long long)(kk + z.ifick - 1) * inck);
  };
  // face sensors are 0 or 1: one byte per cell holds those of its three minus-side faces (bit 0: i, 1: j, 2: k).
  // The byte of a cell is written by the lane that evaluates those faces: the cell's own lane, or, for a ghost
  // cell just outside the zone (i = iv+1 or j = jv+1), the lane or edge wave that evaluates its one face.  It is
  // fetched a step ahead.
  auto ldw = [&](int kk, unsigned bo) -> unsigned {
    const unsigned char *b = wig + (long long)(kk + z.ifick - 1) * inck;
    return b[bo >> 3];
  };
  auto stw = [&](int kk, unsigned bo, unsigned bits) {
    unsigned char *b = wig + (long long)(kk + z.ifick - 1) * inck;
    b[bo >> 3] = (unsigned char)bits;
  };

  // ---- LDS-DMA map ------------------------------------------------------------------------------
  // WIDE: waves 8..11 move the plane image in 16-byte units (cells 2q, 2q+1), unit
  // q = (r*4 + wave-8)*64 + lane in round r (needs even row length and 16-byte aligned arrays).
  // Otherwise every wave moves dwords, dword d = (r*12 + wave)*64 + lane in round r.
  constexpr int NDMA = WIDE ? NDMA_WIDE : NDMA_DW;
  const bool dma_wave = WIDE ? wave >= NCELLW : true;
  unsigned goff[NDMA];
  bool dma_act[NDMA];
#pragma unroll
  for (int r = 0; r < NDMA; r++) {
    if constexpr (WIDE) {
      int q = (r * NSTGW + (wave - NCELLW)) * 64 + lane;
      dma_act[r] = dma_wave && q < PLANE_QW;
      if (q < 0 || q >= PLANE_QW) q = PLANE_QW - 1;
      const int c = 2 * q, rr = c / PW, cc = c % PW;
      int si = i0 + cc - HALO + z.ific, sj = j0 + rr - HALO + z.ific;
      si = si < 0 ? 0 : (si > ni - 2 ? ni - 2 : si);
      sj = sj < 0 ? 0 : (sj > nj - 1 ? nj - 1 : sj);
      goff[r] = (unsigned)(si + sj * incj) * 8u;
    } else {
      int d = (r * NWAVES + wave) * 64 + lane;
      dma_act[r] = d < PLANE_DW;
      if (d >= PLANE_DW) d = PLANE_DW - 1;
      const int c = d >> 1, rr = c / PW, cc = c % PW;
      goff[r] = boff(i0 + cc - HALO + 1, j0 + rr - HALO + 1) + (unsigned)(d & 1) * 4u;
    }
  }
  // paced = true: a pause after every request, so that the requests of the staging waves spread over the step and
  // the loads and stores of the cell waves do not queue behind a burst of them
  auto dma_plane = [&](const double *gbase, double *lds_plane, bool paced = false) {
    const char *g = reinterpret_cast<const char *>(gbase);
#pragma unroll
    for (int r = 0; r < NDMA; r++) {
      if constexpr (WIDE) {
        char *l = reinterpret_cast<char *>(lds_plane) + (r * NSTGW + (wave - NCELLW)) * 1024;
        if (dma_act[r] && !XF(4)) __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + goff[r]), (lds_ptr_t)(l), 16, 0, 0);
        if (paced) for (int q = 0; q < PACE; q++) __builtin_amdgcn_s_sleep(1);
      } else {
        char *l = reinterpret_cast<char *>(lds_plane) + (r * NWAVES + wave) * 256;
        if (dma_act[r]) __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + goff[r]), (lds_ptr_t)(l), 4, 0, 0);
      }
    }
  };
  auto dma_uvw = [&](int kk, bool paced = false) {
    const int s = slot_of(kk);
#pragma unroll
    for (int e = 0; e < 3; e++) dma_plane(P(rop, e + 1, kk), sm.uvw[s][e], paced);
  };
  auto dma_rt = [&](int kk, int slot, bool paced = false) {
    dma_plane(P(rop, 0, kk), sm.rt[slot][0], paced);
    dma_plane(P(rop, 4, kk), sm.rt[slot][1], paced);
  };
  auto dma_mu = [&](int kk, bool paced = false) {
    if (IFLOW == 2) dma_plane(P(xmut, 0, kk), sm.mu[kk & 1], paced);
  };

  // ---- prologue: the first planes of the ring ------------------------------------------------------
  if (dma_wave) {
    dma_uvw(k0 - 1); dma_uvw(k0); dma_uvw(k0 + 1);
    dma_rt(k0, 0); dma_rt(k0 + 1, 1);
    dma_mu(k0);
  }
  int rs = 0;   // ring slot of (rho, T) of plane k; plane k+1 sits in the next slot, plane k+2 arrives in the third
  auto rnext = [](int s_) { return s_ == 2 ? 0 : s_ + 1; };

  if (wave < NCELLW) {
    // ================================ cell waves =================================================
    const bool cell_ok = (i <= z.iv) && (j <= z.jv);
    const bool fi_ok = (i <= z.iv + 1) && (j <= z.jv);    // lane evaluates its i-1/2 face
    const bool fj_ok = (i <= z.iv) && (j <= z.jv + 1);    // lane evaluates its j-1/2 face
    double qa[5], qb[5];            // own column at k-2, k-1
    double mu_b = 0.;               // viscosity of the own column at k-1
#pragma unroll
    for (int e = 0; e < 5; e++) { qa[e] = ldg(P(rop, e, k0 - 2), ob); qb[e] = ldg(P(rop, e, k0 - 1), ob); }
    if (IFLOW == 2) mu_b = ldg(P(xmut, 0, k0 - 1), ob);
    double dr_n[5] = {0., 0., 0., 0., 0.};   // previous balance of cell k (requested a step ahead)
    if (!FIRST) {
#pragma unroll
      for (int e = 0; e < 5; e++) dr_n[e] = ldg(P(drodm, e, k0), ob);
    }
    unsigned wb_n = 0;   // sensor byte of the cell at plane k (requested a step ahead)
    if (KFLU == 2 && !FIRST) wb_n = ldw(k0, ob);
    double part[5] = {0., 0., 0., 0., 0.};   // +Fk- +Fj- +Fi- -Fi+ -Fj+ of cell k-1 (and xi*drodm)
    // AHEAD: the requests for the previous balance and the sensor byte of the next plane go out by hand (ld_ahead)
    // and are waited for with a count that leaves the stores of the step in flight
    // Not with roe: those kernels spill registers, and a spill of a register that is being loaded stores garbage.
    constexpr bool AHEAD = WIDE && !BDF && !FIRST && KFLU != 5;
    // LATE: the stores of the new state leave between the face fluxes of the step (st_late) instead of in one burst
    constexpr bool LATE = AHEAD;
    const bool row_ok = __builtin_amdgcn_readfirstlane((int)(j <= z.jv)) != 0;   // a row outside the zone stores nothing
    int nst_prev = 0;   // store instructions of this wave in the step before
    __syncthreads();

#if FS_X_PRIO == 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef FS_X_UNROLL
#pragma unroll FS_X_UNROLL
#endif
    for (int k = k0; k <= k1 + 1; k++) {
      const bool last = k > k1;          // finishing step: only the k-face k1+1/2 and cell k1
      const int cur = k & 1;
#if FS_X_PRIO == 2
      if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#elif FS_X_PRIO == 3
      if ((wave >= 4) == ((k & 1) != 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
      STAMP(0)
      // the uniform constants of the zone are re-read as scalars in every step: their products (c2*s, c1*s, tc/4 ...)
      // would otherwise be parked in vector registers for the whole loop, which the register budget cannot afford
      FsZone zl = z;
      asm volatile("" : "+s"(zl.tcx), "+s"(zl.tcy), "+s"(zl.tcz), "+s"(zl.si), "+s"(zl.sj), "+s"(zl.sk));
      asm volatile("" : "+s"(zl.c1), "+s"(zl.c2), "+s"(zl.volinv), "+s"(zl.gam3), "+s"(zl.rgp), "+s"(zl.gam1));
      double dr_nn[5] = {0., 0., 0., 0., 0.};
      unsigned wb_nn = 0, wb_new = 0;
      const int rs1 = rnext(rs);
      double xdr[5] = {0., 0., 0., 0., 0.};   // xi * previous balance of cell k
      if constexpr (AHEAD) {
        // what was requested in the step before has landed; the stores issued after it need not have
        wait_ahead(dr_n, wb_n, nst_prev);
#pragma unroll
        for (int e = 0; e < 5; e++) xdr[e] = dr_n[e] * coefH;
        // the products exist before the next requests reuse the registers
        asm volatile("" : "+v"(xdr[0]), "+v"(xdr[1]), "+v"(xdr[2]), "+v"(xdr[3]), "+v"(xdr[4]));
      } else if (!FIRST && !BDF) {
#pragma unroll
        for (int e = 0; e < 5; e++) xdr[e] = dr_n[e] * coefH;
      }
      unsigned wb_c = wb_n;   // sensor byte of the cell at plane k
      if constexpr (AHEAD) asm volatile("v_mov_b32 %0, %1" : "=v"(wb_c) : "v"(wb_n));
      STAMP(1)
      // ---- own column at k (LDS) and k+1 ------------------------------------------------------------
      double qc[5], qd[5];
      qc[0] = sm.rt[rs][0][ct]; qc[4] = sm.rt[rs][1][ct];
      const double mu_c = IFLOW == 2 ? sm.mu[cur][ct] : 0.;
      qd[0] = sm.rt[rs1][0][ct]; qd[4] = sm.rt[rs1][1][ct];
#pragma unroll
      for (int e = 0; e < 3; e++) {
        qc[e + 1] = sm.uvw[slot_of(k)][e][ct];
        qd[e + 1] = sm.uvw[slot_of(k + 1)][e][ct];
      }
      // ---- k-face between planes k-1 and k ----------------------------------------------------------
      double fk[5];
      {
        double wg = (double)((wb_c >> 2) & 1u);
        if (XF(8)) {
#pragma unroll
          for (int e = 0; e < 5; e++) fk[e] = qb[e] + qc[e] + qa[e] + qd[e];
        } else {
        if (KFLU == 2 && FIRST) {
          auto qs = [&](int e, int o) -> double { return o == -2 ? qa[e] : (o == -1 ? qb[e] : (o == 0 ? qc[e] : qd[e])); };
          wg = fs_wiggle(qs);
          wb_new = wg != 0. ? 4u : 0u;   // the face belongs to cell k
          // the plane above the zone has no i/j faces: its byte holds the k sensor only
          if (last && k == z.kv + 1 && cell_ok) stw(k, ob, wb_new);
        }
        fs_flux_inviscid<KFLU, 2>(zl, qa, qb, qc, qd, wg, fk);
        if (IFLOW == 2) {
          const int sl = slot_of(k - 1), sr = slot_of(k);
          auto q = [&](int e, int da, int db, int dn) -> double {
            if (e == 4) return dn == 0 ? qc[4] : qb[4];
            return sm.uvw[dn == 0 ? sr : sl][e - 1][ct + da + db * PW];
          };
          fs_flux_viscous<2>(zl, q, mu_b, mu_c, fk);
        }
        }
      }
      STAMP(2)
      // ---- requests for the next step, ahead of the stores of this one: a wait for these loads then leaves the
      // stores in flight (the memory counter retires in order) ----------------------------------------------------
      if (!last) {
        if (!WIDE) { dma_uvw(k + 2); dma_rt(k + 2, rnext(rs1)); dma_mu(k + 1); }
        if constexpr (AHEAD) {
          if (!XF(2))
#pragma unroll
          for (int e = 0; e < 5; e++) ld_ahead(dr_n[e], P(drodm, e, k + 1), ob);
          if (KFLU == 2) ld_ahead_byte(wb_n, wig + (long long)(k + 1 + z.ifick - 1) * inck, ob >> 3);
        } else {
          if (!FIRST && !BDF) {
#pragma unroll
            for (int e = 0; e < 5; e++) dr_nn[e] = ldg(P(drodm, e, k + 1), ob);
          }
          if (KFLU == 2 && !FIRST) wb_nn = ldw(k + 1, ob);
        }
      }
      // ---- finish cell k-1 --------------------------------------------------------------------------
      double d[5], o[5] = {0., 0., 0., 0., 0.};
#pragma unroll
      for (int e = 0; e < 5; e++) d[e] = part[e] - fk[e];
      if (k > k0 && cell_ok && !XF(1)) {
        if (BDF) {
          if (fs_in_box(bdf, i, j, k - 1)) {
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
          }
        } else {
          if (keep) {   // the balance of the last Runge-Kutta stage of a step is read by nothing
#pragma unroll
            for (int e = 0; e < 5; e++) {
              if constexpr (LATE) st_late(PW_(drodm, e, k - 1), ob, d[e], d[e]);
              else stg(PW_(drodm, e, k - 1), ob, d[e]);
            }
          }
          fs_rk_update(z, coefW, qb, d, o);
          if constexpr (!LATE) {
#pragma unroll
            for (int e = 0; e < 5; e++) stg(PW_(rop_out, e, k - 1), ob, o[e]);
          }
        }
      }
      nst_prev = (k > k0 && row_ok && !BDF && !XF(1)) ? (keep ? 10 : 5) : 0;
#if FS_X_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
      STAMP(3)
      // ---- balance of cell k, in the reference's order: xi*drodm + Fk- + Fj- + Fi- ... ----------------
      double acc[5] = {0., 0., 0., 0., 0.}, fin[5] = {0., 0., 0., 0., 0.};
      if (!last) {
#pragma unroll
        for (int e = 0; e < 5; e++) {
          acc[e] = xdr[e] + fk[e];
        }
        // ---- j-face, then i-face of plane k ----------------------------------------------------------
        {
          double f[5] = {0., 0., 0., 0., 0.};
          if (fj_ok) {
            double w = face_ij<KFLU, IFLOW, 1, FIRST>(zl, sm, k, rs, ct, (double)((wb_c >> 1) & 1u), f, XF(8));
            if (KFLU == 2 && FIRST) wb_new |= w != 0. ? 2u : 0u;
#pragma unroll
            for (int e = 0; e < 5; e++) sm.fj[e][tj][ti] = f[e];
          }
#pragma unroll
          for (int e = 0; e < 5; e++) acc[e] = acc[e] + f[e];
        }
        if (LATE && k > k0 && cell_ok && !XF(1)) {
          st_late(PW_(rop_out, 0, k - 1), ob, o[0], acc[0]);
          st_late(PW_(rop_out, 1, k - 1), ob, o[1], acc[1]);
        }
        STAMP(4)
        {
          double f[5] = {0., 0., 0., 0., 0.};
          if (fi_ok) {
            double w = face_ij<KFLU, IFLOW, 0, FIRST>(zl, sm, k, rs, ct, (double)(wb_c & 1u), f, XF(8));
            if (KFLU == 2 && FIRST) wb_new |= w != 0. ? 1u : 0u;
          }
          if (KFLU == 2 && FIRST && (fi_ok || fj_ok)) stw(k, ob, wb_new);
#pragma unroll
          for (int e = 0; e < 5; e++) { acc[e] = acc[e] + f[e]; fin[e] = from_next_lane(f[e]); }
        }
        if (LATE && k > k0 && cell_ok && !XF(1)) {
          st_late(PW_(rop_out, 2, k - 1), ob, o[2], acc[2]);
          st_late(PW_(rop_out, 3, k - 1), ob, o[3], acc[3]);
          st_late(PW_(rop_out, 4, k - 1), ob, o[4], acc[4]);
        }
      } else if (LATE && k > k0 && cell_ok && !XF(1)) {   // finishing step: no faces to wait for
#pragma unroll
        for (int e = 0; e < 5; e++) st_late(PW_(rop_out, e, k - 1), ob, o[e], d[e]);
      }
      STAMP(5)
      // flux rows complete: LDS traffic only, the requests of this step stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      STAMP(6)
      if (!last) {
#pragma unroll
        for (int e = 0; e < 5; e++) {
          const double fip = ti == TI - 1 ? sm.fi_extra[e][tj] : fin[e];
          const double fjp = sm.fj[e][tj + 1][ti];
          double a = acc[e];
          a = a - fip; a = a - fjp;
          part[e] = a;
        }
      }
      // ---- shift the column window ------------------------------------------------------------------
#pragma unroll
      for (int e = 0; e < 5; e++) { qa[e] = qb[e]; qb[e] = qc[e]; }
      if constexpr (!AHEAD) {
#pragma unroll
        for (int e = 0; e < 5; e++) dr_n[e] = dr_nn[e];
        wb_n = wb_nn;
      }
      mu_b = mu_c;
      rs = rs1;
      STAMP(7)
      // flux rows consumed; the staging waves have waited for the arriving planes.  The cell waves' own loads are
      // waited for where they are used and their stores are never waited for.
      if (WIDE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      } else
        __syncthreads();
    }
  } else {
    // ================================ edge waves =================================================
    // one in-plane face per lane and plane: wave 8 the j-faces of tile row TJ, wave 9 the i-faces of tile
    // column TI, waves 10 and 11 none; all four request the arriving planes
    const bool jedge = wave == NCELLW;
    const bool act = jedge ? (j <= z.jv + 1 && i <= z.iv) : (wave == NCELLW + 1 && lane < TJ && i <= z.iv + 1 && j <= z.jv);
    const int dir = jedge ? 1 : 0;
    // the cell of an edge face belongs to the next tile, whose cell wave writes its sensor byte; only a ghost cell
    // just outside the zone has no such owner
    const bool ghost = jedge ? j == z.jv + 1 : i == z.iv + 1;
    unsigned wg_n = 0;
    if (KFLU == 2 && !FIRST && act) wg_n = (ldw(k0, ob) >> dir) & 1u;
    __syncthreads();
    for (int k = k0; k <= k1 + 1; k++) {
      const bool last = k > k1;
      unsigned wg_nn = 0;
      STAMP(0)
      if (!last) {
        dma_rt(k + 2, rnext(rnext(rs)), true);
        dma_mu(k + 1, true);
        dma_uvw(k + 2, true);
        STAMP(1)
        if (KFLU == 2 && !FIRST && act) wg_nn = (ldw(k + 1, ob) >> dir) & 1u;
        if (act) {
          double f[5];
          if (jedge) {
            double w = face_ij<KFLU, IFLOW, 1, FIRST>(z, sm, k, rs, ct, (double)wg_n, f, XF(8));
            if (KFLU == 2 && FIRST && ghost) stw(k, ob, w != 0. ? 2u : 0u);
#pragma unroll
            for (int e = 0; e < 5; e++) sm.fj[e][TJ][ti] = f[e];
          } else {
            double w = face_ij<KFLU, IFLOW, 0, FIRST>(z, sm, k, rs, ct, (double)wg_n, f, XF(8));
            if (KFLU == 2 && FIRST && ghost) stw(k, ob, w != 0. ? 1u : 0u);
#pragma unroll
            for (int e = 0; e < 5; e++) sm.fi_extra[e][tj] = f[e];
          }
        }
      }
      STAMP(5)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      STAMP(6)
      wg_n = wg_nn;
      rs = rnext(rs);
      STAMP(7)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the arriving planes have landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
}

template <int KFLU, int IFLOW>
void launch(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
            double *drodm, const double *xmut, unsigned char *wig, const int *blist, int nlist, const FsBdf *bdf,
            int keep, hipStream_t s)
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
                     t.kchunk, t.nti, t.ntj, nblk, t.pace, keep, rop, rop_out, drodm, xmut, wig, blist, b)
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
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fs_stamp_buf), sizeof(long long) * 12 * 40 * 8);
}
#endif

FsTiling fsk_stage_dma_tiling(const FsZone &z)
{
  FsTiling t;
  t.ti = TI; t.tj = TJ;
  // the face i = iv+1 is evaluated by lane iv - i0 of the last tile, or by its i-edge wave
  t.nti = (z.iv + TI - 1) / TI;
  t.ntj = (z.jv + TJ - 1) / TJ;
  // one workgroup per CU is resident (LDS): the k chunks are cut so that the tiles of the zone give about one
  // workgroup per CU, the fewest plane-ring fills (256^3: 128 tiles x 2 chunks of 128 planes, 0.710 ms per stage
  // against 0.738 with chunks of 32)
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    ncu = n;
  }
  int nchunks = ncu / (t.nti * t.ntj);
  if (nchunks < 1) nchunks = 1;
  // a zone whose stage runs in two parts around its halo messages: at least 8 chunks, so that the chunks that touch
  // no k face (and the tiles that touch no i / j face) are most of the zone and run while the messages travel
  if (z.split && nchunks < 8) nchunks = 8;
  t.kchunk = (z.kv + nchunks - 1) / nchunks;
  if (t.kchunk < 16) t.kchunk = 16;
  if (const char *ev = getenv("FASTS_HIP_KCHUNK")) t.kchunk = atoi(ev) > 0 ? atoi(ev) : t.kchunk;   // tuning aid
  if (z.kv < t.kchunk) t.kchunk = z.kv;
  t.pace = 1;
  if (const char *ev = getenv("FASTS_HIP_PACE")) t.pace = atoi(ev);   // tuning aid
#ifdef FS_X_LEGS
  if (const char *ev = getenv("FASTS_HIP_XLEGS")) t.pace = (t.pace & 0xff) | (atoi(ev) << 8);
#endif
  t.ntk = (z.kv + t.kchunk - 1) / t.kchunk;
  return t;
}

int fsk_stage_dma(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
                  double *drodm, const double *xmut, unsigned char *wig, const int *blist, int nlist, const FsBdf *bdf,
                  int keep_balance, hipStream_t s)
{
  // one k-plane must stay below 4 GiB / 8 so that in-plane byte offsets fit 32 bits
  if ((long long)z.ni * z.nj >= (1ll << 28)) return -2;
#define FS_CASE(K, F)                                                                       \
  if (z.kflu == K && z.iflow == F) {                                                        \
    launch<K, F>(z, coefH, coefW, first, rop, rop_out, drodm, xmut, wig, blist, nlist, bdf, keep_balance, s);  \
    return 0;                                                                               \
  }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
#undef FS_CASE
  return -1;
}
