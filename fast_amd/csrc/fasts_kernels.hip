// HIP kernels of the FastS explicit time step on Cartesian zones, gfx950 (MI355X).
// Built with -ffp-contract=off (see fasts_flux.h for why results must be bit-reproducible).
#include "fasts_kernels.h"
#include "fasts_flux.h"

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
FS_DEV long long fs_inddm(const FsZone &z, int i, int j, int k)
{
  // 0-based linear index of cell (i,j,k), interior starting at 1 (formule_param.h:4-7)
  return (long long)(i + z.ific - 1) + (long long)(j + z.ific - 1) * z.incj + (long long)(k + z.ifick - 1) * z.inck;
}

template <int DIR> FS_DEV long long fs_inc(const FsZone &z) { return DIR == 0 ? 1 : (DIR == 1 ? z.incj : z.inck); }

// flux through the face between cell lc - inc(DIR) and cell lc, global-memory operands
template <int KFLU, int IFLOW, int DIR>
FS_DEV void face_flux_g(const FsZone &z, const double *__restrict__ rop, const double *__restrict__ xmut,
                        const double *__restrict__ wig, long long lc, double *flu)
{
  const long long inc = fs_inc<DIR>(z), nd = z.nd;
  double qm2[5], qm1[5], q0[5], qp1[5];
#pragma unroll
  for (int e = 0; e < 5; e++) {
    const double *q = rop + e * nd + lc;
    qm2[e] = q[-2 * inc]; qm1[e] = q[-inc]; q0[e] = q[0]; qp1[e] = q[inc];
  }
  double wg = 0.;
  if (KFLU == 2) wg = wig[DIR * nd + lc];
  fs_flux_inviscid<KFLU, DIR>(z, qm2, qm1, q0, qp1, wg, flu);
  if (IFLOW == 2) {
    const long long ia = DIR == 0 ? z.incj : 1;
    const long long ib = DIR == 2 ? z.incj : z.inck;
    auto q = [&](int e, int da, int db, int dn) -> double {
      return rop[e * nd + lc + dn * inc + da * ia + db * ib];
    };
    fs_flux_viscous<DIR>(z, q, xmut[lc - inc], xmut[lc], flu);
  }
}

// ------------------------------------------------------------------------------------------
// stage-1 auxiliaries: laminar viscosity on interior + first ghost layer
// (navier_stokes_struct.for:213-225 with ind_grad), sensor on all faces 1..N+1
// (src_term.for:258-272, SENSOR/wiggle2001.for)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_invist(FsZone z, const double *__restrict__ rop, double *__restrict__ xmut)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. iv+1
  const int j = blockIdx.y;                             // 0 .. jv+1
  const int k = blockIdx.z;                             // 0 .. kv+1
  if (i > z.iv + 1) return;
  long long l = fs_inddm(z, i, j, k);
  xmut[l] = fs_sutherland(z, rop[4 * z.nd + l]);
}

__global__ void __launch_bounds__(256) k_wiggle(FsZone z, const double *__restrict__ rop, double *__restrict__ wig)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;  // 1 .. iv+1
  const int j = blockIdx.y + 1;                             // 1 .. jv+1
  const int k = blockIdx.z + 1;                             // 1 .. kv+1
  if (i > z.iv + 1) return;
  const long long l = fs_inddm(z, i, j, k), nd = z.nd;
  const bool in_i = i <= z.iv, in_j = j <= z.jv, in_k = k <= z.kv;
  if (in_j && in_k) {
    auto q = [&](int e, int o) -> double { return rop[e * nd + l + o]; };
    wig[l] = fs_wiggle(q);
  }
  if (in_i && in_k) {
    const long long inc = z.incj;
    auto q = [&](int e, int o) -> double { return rop[e * nd + l + o * inc]; };
    wig[nd + l] = fs_wiggle(q);
  }
  if (in_i && in_j) {
    const long long inc = z.inck;
    auto q = [&](int e, int o) -> double { return rop[e * nd + l + o * inc]; };
    wig[2 * nd + l] = fs_wiggle(q);
  }
}

// ------------------------------------------------------------------------------------------
// one Runge-Kutta stage: balance of the six face fluxes of a cell + update.
// Accumulation order per cell = the order in which the reference's scatter loops reach it
// with one cache block: +Fk-, +Fj-, +Fi-, -Fi+, -Fj+, -Fk+
// (template_FluxAndBalance.for:192-271, assemble_drodm_plus_vec1.for).
// ------------------------------------------------------------------------------------------
template <int KFLU, int IFLOW>
__global__ void __launch_bounds__(256) k_stage_ref(FsZone z, double coefH, double coefW, int first,
                                                   const double *__restrict__ rop, double *__restrict__ rop_out,
                                                   double *__restrict__ drodm, const double *__restrict__ xmut,
                                                   const double *__restrict__ wig)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const int j = blockIdx.y + 1;
  const int k = blockIdx.z + 1;
  if (i > z.iv) return;
  const long long l = fs_inddm(z, i, j, k), nd = z.nd;
  double d[5], f[5];
#pragma unroll
  for (int e = 0; e < 5; e++) d[e] = first ? 0. : drodm[e * nd + l] * coefH;
  face_flux_g<KFLU, IFLOW, 2>(z, rop, xmut, wig, l, f);
#pragma unroll
  for (int e = 0; e < 5; e++) d[e] = d[e] + f[e];
  face_flux_g<KFLU, IFLOW, 1>(z, rop, xmut, wig, l, f);
#pragma unroll
  for (int e = 0; e < 5; e++) d[e] = d[e] + f[e];
  face_flux_g<KFLU, IFLOW, 0>(z, rop, xmut, wig, l, f);
#pragma unroll
  for (int e = 0; e < 5; e++) d[e] = d[e] + f[e];
  face_flux_g<KFLU, IFLOW, 0>(z, rop, xmut, wig, l + 1, f);
#pragma unroll
  for (int e = 0; e < 5; e++) d[e] = d[e] - f[e];
  face_flux_g<KFLU, IFLOW, 1>(z, rop, xmut, wig, l + z.incj, f);
#pragma unroll
  for (int e = 0; e < 5; e++) d[e] = d[e] - f[e];
  face_flux_g<KFLU, IFLOW, 2>(z, rop, xmut, wig, l + z.inck, f);
#pragma unroll
  for (int e = 0; e < 5; e++) d[e] = d[e] - f[e];
  double q[5], o[5];
#pragma unroll
  for (int e = 0; e < 5; e++) { q[e] = rop[e * nd + l]; drodm[e * nd + l] = d[e]; }
  fs_rk_update(z, coefW, q, d, o);
#pragma unroll
  for (int e = 0; e < 5; e++) rop_out[e * nd + l] = o[e];
}

// ------------------------------------------------------------------------------------------
// ghost fill: BC windows, then edges and corners
// ------------------------------------------------------------------------------------------
// One block-row per window; window w fills cells [lo..hi] in each direction from
// rop(l + shift) (BCautoperiod, bvbs_periodique.for:150-171) or from the nearest interior cell
// (BCExtrapolate order 0).
__global__ void __launch_bounds__(256) k_bc_windows(FsZone z, int nwin, const FsBcWin *__restrict__ wins,
                                                    double *__restrict__ rop)
{
  const FsBcWin w = wins[blockIdx.y];
  const int n0 = w.hi[0] - w.lo[0] + 1, n1 = w.hi[1] - w.lo[1] + 1, n2 = w.hi[2] - w.lo[2] + 1;
  const long long n = (long long)n0 * n1 * n2;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    int i = w.lo[0] + (int)(t % n0);
    int j = w.lo[1] + (int)((t / n0) % n1);
    int k = w.lo[2] + (int)(t / ((long long)n0 * n1));
    long long l = fs_inddm(z, i, j, k), ls;
    if (w.type == 11) ls = l + w.shift;
    else {  // order-0 extrapolation: first interior cell along the face normal
      int ii = i, jj = j, kk = k;
      switch (w.idir) {
        case 1: ii = 1; break; case 2: ii = z.iv; break;
        case 3: jj = 1; break; case 4: jj = z.jv; break;
        case 5: kk = 1; break; default: kk = z.kv; break;
      }
      ls = fs_inddm(z, ii, jj, kk);
    }
    for (int e = 0; e < z.neq; e++) rop[e * z.nd + l] = rop[e * z.nd + ls];
  }
}

// 12 edges (mean of two face ghosts) and 8 corners (mean of three), correct_coins.for:23-420.
// blockIdx.y = edge/corner id 0..19.
__global__ void __launch_bounds__(256) k_bc_edges(FsZone z, double *__restrict__ rop)
{
  const int id = blockIdx.y;
  const int f = z.ific, fk = z.ifick;
  const int ends[3][2] = {{1, z.iv}, {1, z.jv}, {1, z.kv}};
  const int nfic[3] = {f, f, fk};
  const long long nd = z.nd;
  if (id < 12) {
    // edge family: 0-3 (i,j) ghosts, k interior; 4-7 (i,k), j interior; 8-11 (j,k), i interior
    const int fam = id / 4, s0 = (id % 4) & 1, s1 = ((id % 4) >> 1) & 1;
    const int da = fam == 2 ? 1 : 0;            // first ghost direction
    const int db = fam == 0 ? 1 : 2;            // second ghost direction
    const int dc = fam == 0 ? 2 : (fam == 1 ? 1 : 0);  // interior direction
    const int na = nfic[da], nb = nfic[db], nc = ends[dc][1];
    const long long n = (long long)na * nb * nc;
    const int sa = s0 ? 1 : -1, sb = s1 ? 1 : -1;
    const int a0 = ends[da][s0], b0 = ends[db][s1];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
      int c = 1 + (int)(t % nc);
      int a = 1 + (int)((t / nc) % na);
      int b = 1 + (int)(t / ((long long)nc * na));
      int a1 = a + 1 < na ? a + 1 : na, b1 = b + 1 < nb ? b + 1 : nb;
      int p1[3], p11[3], p12[3];
      p1[da] = a0 + a * sa;   p1[db] = b0 + b * sb;   p1[dc] = c;
      // the reference pairs (ghost a1, face b0) with (face a0, ghost b1) for (i,j) edges and
      // (face a0, ghost b1) with (ghost a1, face b0) for the others: the sum is the same two cells
      p11[da] = a0 + a1 * sa; p11[db] = b0;           p11[dc] = c;
      p12[da] = a0;           p12[db] = b0 + b1 * sb; p12[dc] = c;
      long long l1 = fs_inddm(z, p1[0], p1[1], p1[2]);
      long long l11 = fs_inddm(z, p11[0], p11[1], p11[2]);
      long long l12 = fs_inddm(z, p12[0], p12[1], p12[2]);
      for (int e = 0; e < z.neq; e++) {
        double x11 = rop[e * nd + l11], x12 = rop[e * nd + l12];
        // operand order of correct_coins.for: (i,j) edges add l11 (ghost i) first; (i,k) and
        // (j,k) edges add the k-ghost cell first
        rop[e * nd + l1] = fam == 0 ? (x11 + x12) * 0.5 : (x12 + x11) * 0.5;
      }
    }
  } else {
    const int c = id - 12;
    const int si = (c & 1) ? 1 : -1, sj = (c & 2) ? 1 : -1, sk = (c & 4) ? 1 : -1;
    const int i0 = ends[0][c & 1], j0 = ends[1][(c >> 1) & 1], k0 = ends[2][(c >> 2) & 1];
    const long long n = (long long)f * f * fk;
    const double c3 = 1. / 3.;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
      int i = 1 + (int)(t % f), j = 1 + (int)((t / f) % f), k = 1 + (int)(t / (f * f));
      int i1 = i + 1 < f ? i + 1 : f, j1 = j + 1 < f ? j + 1 : f, k1 = k + 1 < fk ? k + 1 : fk;
      long long l = fs_inddm(z, i0 + i * si, j0 + j * sj, k0 + k * sk);
      long long l11 = fs_inddm(z, i0, j0, k0 + k1 * sk);
      long long l12 = fs_inddm(z, i0 + i1 * si, j0, k0);
      long long l13 = fs_inddm(z, i0, j0 + j1 * sj, k0);
      for (int e = 0; e < z.neq; e++)
        rop[e * nd + l] = (rop[e * nd + l11] + rop[e * nd + l12] + rop[e * nd + l13]) * c3;
    }
  }
}

// ------------------------------------------------------------------------------------------
// joins: ghost[rcv] = donor[dnr] for every variable (type-0 'direct' transfer,
// setInterpTransfersFast.cpp:440-935), pack / unpack for joins that cross contexts
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_join_copy(int npts, int neq, long long nd_r, long long nd_d,
                                                   const int *__restrict__ rcv, const int *__restrict__ dnr,
                                                   double *__restrict__ rop_r, const double *__restrict__ rop_d)
{
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)npts * neq;
       t += (long long)gridDim.x * blockDim.x) {
    int e = (int)(t / npts), n = (int)(t % npts);
    rop_r[e * nd_r + rcv[n]] = rop_d[e * nd_d + dnr[n]];
  }
}

__global__ void __launch_bounds__(256) k_halo_pack(int npts, int neq, long long nd, const int *__restrict__ pts,
                                                   const double *__restrict__ rop, double *__restrict__ buf)
{
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)npts * neq;
       t += (long long)gridDim.x * blockDim.x) {
    int e = (int)(t / npts), n = (int)(t % npts);
    buf[t] = rop[e * nd + pts[n]];
  }
}

__global__ void __launch_bounds__(256) k_halo_unpack(int npts, int neq, long long nd, const int *__restrict__ pts,
                                                     double *__restrict__ rop, const double *__restrict__ buf)
{
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)npts * neq;
       t += (long long)gridDim.x * blockDim.x) {
    int e = (int)(t / npts), n = (int)(t % npts);
    rop[e * nd + pts[n]] = buf[t];
  }
}

// ------------------------------------------------------------------------------------------
// residual norms: per-block partial sums in a fixed order, then one block folds the partials
// in index order, so the result does not depend on scheduling.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_residual_partial(FsZone z, const double *__restrict__ drodm,
                                                          double *__restrict__ part)
{
  __shared__ double ssum[5][256];
  __shared__ double smax[5][256];
  const int j = blockIdx.x + 1, k = blockIdx.y + 1;
  double s[5] = {0, 0, 0, 0, 0}, m[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x + 1; i <= z.iv; i += blockDim.x) {
    long long l = fs_inddm(z, i, j, k);
    for (int e = 0; e < 5; e++) {
      double r = z.coe * drodm[e * z.nd + l];
      s[e] = s[e] + r * r;
      m[e] = fs_max(m[e], fabs(r));
    }
  }
  for (int e = 0; e < 5; e++) { ssum[e][threadIdx.x] = s[e]; smax[e][threadIdx.x] = m[e]; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int e = 0; e < 5; e++) {
        ssum[e][threadIdx.x] = ssum[e][threadIdx.x] + ssum[e][threadIdx.x + w];
        smax[e][threadIdx.x] = fs_max(smax[e][threadIdx.x], smax[e][threadIdx.x + w]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    for (int e = 0; e < 5; e++) { part[b * 10 + e] = ssum[e][0]; part[b * 10 + 5 + e] = smax[e][0]; }
  }
}

__global__ void __launch_bounds__(256) k_residual_final(long long nblk, const double *__restrict__ part,
                                                        double *__restrict__ out)
{
  __shared__ double ssum[5][256];
  __shared__ double smax[5][256];
  double s[5] = {0, 0, 0, 0, 0}, m[5] = {0, 0, 0, 0, 0};
  for (long long b = threadIdx.x; b < nblk; b += blockDim.x)
    for (int e = 0; e < 5; e++) { s[e] = s[e] + part[b * 10 + e]; m[e] = fs_max(m[e], part[b * 10 + 5 + e]); }
  for (int e = 0; e < 5; e++) { ssum[e][threadIdx.x] = s[e]; smax[e][threadIdx.x] = m[e]; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int e = 0; e < 5; e++) {
        ssum[e][threadIdx.x] = ssum[e][threadIdx.x] + ssum[e][threadIdx.x + w];
        smax[e][threadIdx.x] = fs_max(smax[e][threadIdx.x], smax[e][threadIdx.x + w]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int e = 0; e < 5; e++) { out[e] = ssum[e][0]; out[5 + e] = smax[e][0]; }
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

void fsk_invist(const FsZone &z, const double *rop, double *xmut, hipStream_t s)
{
  dim3 g(cdiv(z.iv + 2, 256), z.jv + 2, z.kv + 2);
  hipLaunchKernelGGL(k_invist, g, dim3(256), 0, s, z, rop, xmut);
}

void fsk_wiggle(const FsZone &z, const double *rop, double *wig, hipStream_t s)
{
  dim3 g(cdiv(z.iv + 1, 256), z.jv + 1, z.kv + 1);
  hipLaunchKernelGGL(k_wiggle, g, dim3(256), 0, s, z, rop, wig);
}

template <int KFLU, int IFLOW>
static void launch_stage_ref(const FsZone &z, double coefH, double coefW, int first, const double *rop,
                             double *rop_out, double *drodm, const double *xmut, const double *wig, hipStream_t s)
{
  int bx = z.iv >= 256 ? 256 : (z.iv >= 128 ? 128 : 64);
  dim3 g(cdiv(z.iv, bx), z.jv, z.kv);
  hipLaunchKernelGGL((k_stage_ref<KFLU, IFLOW>), g, dim3(bx), 0, s, z, coefH, coefW, first, rop, rop_out, drodm,
                     xmut, wig);
}

int fsk_stage(const FsZone &z, double coefH, double coefW, int first, const double *rop, double *rop_out,
              double *drodm, const double *xmut, const double *wig, hipStream_t s)
{
#define FS_CASE(K, F)                                                                         \
  if (z.kflu == K && z.iflow == F) {                                                          \
    launch_stage_ref<K, F>(z, coefH, coefW, first, rop, rop_out, drodm, xmut, wig, s);        \
    return 0;                                                                                 \
  }
  FS_CASE(1, 1) FS_CASE(1, 2) FS_CASE(2, 1) FS_CASE(2, 2) FS_CASE(5, 1) FS_CASE(5, 2)
#undef FS_CASE
  return -1;
}

void fsk_bc_windows(const FsZone &z, int nwin, const FsBcWin *dev_wins, long long max_cells, double *rop, hipStream_t s)
{
  if (nwin <= 0) return;
  unsigned gx = cdiv(max_cells, 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(k_bc_windows, dim3(gx, nwin), dim3(256), 0, s, z, nwin, dev_wins, rop);
}

void fsk_bc_edges(const FsZone &z, double *rop, hipStream_t s)
{
  int m = z.iv > z.jv ? z.iv : z.jv;
  if (z.kv > m) m = z.kv;
  unsigned gx = cdiv((long long)m * z.ific * z.ific, 256);
  hipLaunchKernelGGL(k_bc_edges, dim3(gx, 20), dim3(256), 0, s, z, rop);
}

void fsk_join_copy(int npts, int neq, long long nd_r, long long nd_d, const int *rcv, const int *dnr,
                   double *rop_r, const double *rop_d, hipStream_t s)
{
  if (npts <= 0) return;
  unsigned gx = cdiv((long long)npts * neq, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_join_copy, dim3(gx), dim3(256), 0, s, npts, neq, nd_r, nd_d, rcv, dnr, rop_r, rop_d);
}

void fsk_halo_pack(int npts, int neq, long long nd, const int *pts, const double *rop, double *buf, hipStream_t s)
{
  if (npts <= 0) return;
  unsigned gx = cdiv((long long)npts * neq, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_halo_pack, dim3(gx), dim3(256), 0, s, npts, neq, nd, pts, rop, buf);
}

void fsk_halo_unpack(int npts, int neq, long long nd, const int *pts, double *rop, const double *buf, hipStream_t s)
{
  if (npts <= 0) return;
  unsigned gx = cdiv((long long)npts * neq, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_halo_unpack, dim3(gx), dim3(256), 0, s, npts, neq, nd, pts, rop, buf);
}

void fsk_residual(const FsZone &z, const double *drodm, double *part, double *out, hipStream_t s)
{
  hipLaunchKernelGGL(k_residual_partial, dim3(z.jv, z.kv), dim3(256), 0, s, z, drodm, part);
  hipLaunchKernelGGL(k_residual_final, dim3(1), dim3(256), 0, s, (long long)z.jv * z.kv, part, out);
}
