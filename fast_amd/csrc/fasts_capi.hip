// C ABI of libfasts_hip.so: context, zones, the sub-iteration sequence and data movement.
// Declarations and the reference interfaces they replace: include/fasts_hip.h.
#include "fasts_ctx.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// Parameter_int / Parameter_real slots (Fast/FastS/FastS/param_solver.h:77-306)
enum { NIJK = 0, NIJK_MTR = 5, IJKV = 20, ITYPZONE = 25, IFLOW = 27, ITYPCP = 29, KFLUDOM = 33, LALE = 34, NEQ = 36,
       NEQ_IJ = 37, NEQ_K = 38, NDIMDX = 41, NDIMDX_MTR = 43, DTLOC = 46, RK = 52, EXPLOC = 54, NSSITER = 58,
       SLOPE = 64, PT_BC = 70, SIZE_SSDOM = 30, SA_LOW_RE = 130, SA_ROT_CORR = 131, SA_DIST = 132, SFD = 62, SFD_INIT_IT = 63, NB_RELAX = 71, IMPLICITSOLVER = 74, LU_MATCH = 75, SENSORTYPE = 85 };
enum { DTC = 0, GAMMA = 1, CVINF = 2, ROINF = 3, PINF = 4, VINF = 5, PRANDT = 10, XMUL0 = 11, TEMP0 = 12, CS = 13,
       RONUTILDEINF = 8, EPSI_NEWTON = 14, CFL = 15, SA_RATIOM_R = 18, PRANDT_TUR = 38, VXINF = 19, VYINF = 20, VZINF = 21,
       PSIROE = 34 };

}  // namespace

static std::string g_create_err;

// Scratch of the allocation-free entry point fs_tgv_sums_static: two doubles per (j,k) row of the largest zone it
// accepts (4 MiB).  Every GPU the library runs on holds its own copy at its own address, looked up once per device in
// fs_create; host code uses that address, never the array's name.
#define FS_STATIC_ROWS (1 << 18)
#define FS_MAX_DEVICES 64
static __device__ __align__(256) double fs_static_scratch[2 * FS_STATIC_ROWS];
static double *g_static_scratch[FS_MAX_DEVICES];

static int bc_kinds(const std::vector<FsBcWin> &v)
{
  int k = 0;
  for (auto &w : v)
    k |= (w.type == FS_BC_PERIODIC || w.type == FS_BC_EXTRAPOLATE || w.type == FS_BC_ZERO || w.type == FS_BC_EXTRAP_NUT) ? 1 : 2;
  return k;
}

extern "C" const char *fs_version(void) { return "fasts_hip 0.1 gfx950 fp64 contract=off"; }

extern "C" const char *fs_last_error(const fs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int fs_create(fs_ctx **out, int device)
{
  if (!out) return FS_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("fs_create: no HIP device available (") + hipGetErrorString(e) + ")";
    return FS_ERR_HIP;
  }
  if (device < 0 || device >= ndev) {
    g_create_err = "fs_create: device index out of range";
    return FS_ERR_ARG;
  }
  fs_ctx *c = new fs_ctx;
  c->device = device;
  e = hipSetDevice(device);
  if (e == hipSuccess && device < FS_MAX_DEVICES && !g_static_scratch[device]) {
    void *sp = nullptr;
    e = hipGetSymbolAddress(&sp, HIP_SYMBOL(fs_static_scratch));   // this device's copy of the array
    if (e == hipSuccess) g_static_scratch[device] = (double *)sp;
  }
  if (e == hipSuccess) e = hipStreamCreate(&c->own_stream);
  if (e != hipSuccess) {
    g_create_err = std::string("fs_create: ") + hipGetErrorString(e);
    delete c;
    return FS_ERR_HIP;
  }
  c->stream = c->own_stream;
  *out = c;
  return FS_OK;
}

extern "C" void fs_destroy(fs_ctx *ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &zn : ctx->zones) {
    for (int r = 0; r < 3; r++) (void)hipFree(zn.dev[r]);
    for (int r = 0; r < 3; r++) if (zn.pinned[r]) { if (hipHostUnregister(zn.pinned[r]) != hipSuccess) (void)hipGetLastError(); }
    (void)hipFree(zn.drodm); (void)hipFree(zn.xmut); (void)hipFree(zn.wig);
    (void)hipFree(zn.res_part); (void)hipFree(zn.res_out); (void)hipFree(zn.bc_dev); (void)hipFree(zn.bc_data_dev);
    (void)hipFree(zn.blk_res); (void)hipFree(zn.wsc); (void)hipFree(zn.wflux);
    for (auto &bb : zn.box_bcs) { (void)hipFree(bb.dev); (void)hipFree(bb.tiles_dev); (void)hipFree(bb.lu_dev); fsk_luflow_free(bb.flow); }
    (void)hipFree(zn.shell_dev); (void)hipFree(zn.inner_dev); (void)hipFree(zn.mtr_dev);
    (void)hipFree(zn.coe); (void)hipFree(zn.scratch); (void)hipFree(zn.ssor_work); (void)hipFree(zn.lu_blocks_dev); (void)hipFree(zn.dist); (void)hipFree(zn.rof); (void)hipFree(zn.rhs_face_dev); (void)hipFree(zn.rhs_bc_dev);
    fsk_luflow_free(zn.lu_flow);
    if (zn.ev_part) (void)hipEventDestroy(zn.ev_part);
    if (zn.ev_lu_fork) (void)hipEventDestroy(zn.ev_lu_fork);
    if (zn.ev_lu_join) (void)hipEventDestroy(zn.ev_lu_join);
  }
  (void)hipFree(ctx->diag_out);
  fs_comm_release(ctx);
  for (hipStream_t st : ctx->lu_streams) (void)hipStreamDestroy(st);
  for (auto &j : ctx->joins) { (void)hipFree(j.rcv_dev); (void)hipFree(j.dnr_dev); }
  for (auto &s : ctx->sends) (void)hipFree(s.pts_dev);
  for (auto &p : ctx->ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

// Host memory for the caller's level buffers that the host link copies in place (page-locked, known to the
// runtime): fs_upload / fs_download then run at the rate of the link instead of through a pageable staging copy.
extern "C" int fs_host_alloc(int device, void **ptr, size_t nbytes)
{
  if (!ptr || nbytes == 0) return FS_ERR_ARG;
  *ptr = nullptr;
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return FS_ERR_HIP; }
  if (hipHostMalloc(ptr, nbytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *ptr = nullptr; return FS_ERR_HIP; }
  return FS_OK;
}

extern "C" int fs_host_free(void *ptr)
{
  if (!ptr) return FS_OK;
  if (hipHostFree(ptr) != hipSuccess) { (void)hipGetLastError(); return FS_ERR_HIP; }
  return FS_OK;
}

extern "C" int fs_set_stream(fs_ctx *ctx, void *hip_stream)
{
  if (!ctx) return FS_ERR_ARG;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return FS_OK;
}

extern "C" int fs_set_verif(fs_ctx *ctx, int modulo_verif)
{
  if (!ctx || modulo_verif <= 0) return FS_ERR_ARG;
  ctx->modulo_verif = modulo_verif;
  return FS_OK;
}

extern "C" int fs_zone_set_distance(fs_ctx *ctx, int zone, const double *dist, int n)
{
  if (!ctx || !dist) return ctx ? fail(ctx, FS_ERR_ARG, "fs_zone_set_distance: null argument") : FS_ERR_ARG;
  if (zone < 0 || zone >= (int)ctx->zones.size()) return fail(ctx, FS_ERR_ARG, "fs_zone_set_distance: zone index out of range");
  if (ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_zone_set_distance: call before fs_upload");
  Zone &zn = ctx->zones[zone];
  if ((long long)n != (long long)zn.z.ni * zn.z.nj * zn.z.nk) return fail(ctx, FS_ERR_ARG, "fs_zone_set_distance: one value per cell, ghost cells included");
  zn.dist_host = dist;
  return FS_OK;
}

extern "C" int fs_zone_set_filtered(fs_ctx *ctx, int zone, double *rof, long long n)
{
  if (!ctx || !rof) return ctx ? fail(ctx, FS_ERR_ARG, "fs_zone_set_filtered: null argument") : FS_ERR_ARG;
  if (zone < 0 || zone >= (int)ctx->zones.size()) return fail(ctx, FS_ERR_ARG, "fs_zone_set_filtered: zone index out of range");
  if (ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_zone_set_filtered: call before fs_upload");
  Zone &zn = ctx->zones[zone];
  if (zn.pint[SFD] != 1) return fail(ctx, FS_ERR_STATE, "fs_zone_set_filtered: the zone does not use selective frequency damping");
  if (n != (long long)zn.z.neq * zn.z.nd) return fail(ctx, FS_ERR_ARG, "fs_zone_set_filtered: NEQ * NDIMDX values expected");
  zn.rof_host = rof;
  return FS_OK;
}

// Blocks of interior cells that the reference's OpenMP threads take in a zone of iv x jv x kv cells (omp_mode 0: every zone
// is cut among all the threads, Fast/FastC/FastC/HPC_LAYER/distributeThreads.cpp:90-95 -> indice_boucle_lu.for:94-335):
// a p x q cut with p a divisor of the thread count not above 8, along k and j where both directions offer enough slices
// of lmin cells, else along j and i; the pair with the smallest load imbalance wins (an uneven cut along the second
// direction weighs 2 or 5 times; the imbalance is evaluated in single precision as the reference does); when no pair
// fits and a cut was possible in principle the first thread takes the zone, and with fewer slices than threads there are
// as many blocks as slices.  Threads left without cells are dropped.
static std::vector<int32_t> lu_thread_blocks(int iv, int jv, int kv, int nthreads, int lmin)
{
  const int size[3] = {iv, jv, kv};
  std::vector<int32_t> out;
  auto whole = [&]() { out = {1, iv, 1, jv, 1, kv}; return out; };
  const int n = nthreads;
  if (n <= 1) return whole();
  int slices[3];
  for (int d = 0; d < 3; d++) slices[d] = size[d] / lmin > 1 ? size[d] / lmin : 1;
  if (kv == 1) slices[2] = 1;
  std::vector<int> divisors = {1};
  for (int q = 2; q <= 8; q++) if (n % q == 0) divisors.push_back(q);
  // cost of cutting direction a in ca and direction b in cb
  auto imbalance = [&](int a, int ca, int b, int cb, double weight) {
    const float fa = (float)(size[a] % ca) / (float)(size[a] / ca), fb = (float)(size[b] % cb) / (float)(size[b] / cb);
    return (double)fa + (double)fb * weight;
  };
  // best (count along a, count along b); small_first: the pair with the divisor along a is tried first
  auto best_cut = [&](int a, int b, double weight, bool small_first, int &ca_best, int &cb_best) {
    double cost_best = 100000.;
    bool found = false;
    for (int p1 : divisors) {
      const int p2 = n / p1;
      const int pairs[2][2] = {{small_first ? p1 : p2, small_first ? p2 : p1}, {small_first ? p2 : p1, small_first ? p1 : p2}};
      for (auto &pr : pairs) {
        if (pr[0] <= slices[a] && pr[1] <= slices[b]) {
          const double cost = imbalance(a, pr[0], b, pr[1], weight);
          if (cost < cost_best) { cost_best = cost; ca_best = pr[0]; cb_best = pr[1]; found = true; }
        }
      }
    }
    return found;
  };
  int counts[3] = {0, 0, 0};
  bool have = false, failed = false;
  int ca = 0, cb = 0;
  if (slices[2] * slices[1] >= n) {
    if (best_cut(2, 1, 2., false, ca, cb)) { counts[0] = 1; counts[1] = cb; counts[2] = ca; have = true; }
    else failed = true;
  }
  if (!have && slices[0] * slices[1] >= n) {
    if (best_cut(1, 0, 5., true, ca, cb)) { counts[0] = cb; counts[1] = ca; counts[2] = 1; have = true; }
    else failed = true;
  }
  if (!have) {
    if (failed) return whole();
    if (slices[0] * slices[1] > slices[1] * slices[2]) { counts[0] = slices[0]; counts[1] = slices[1]; counts[2] = 1; }
    else { counts[0] = 1; counts[1] = slices[1]; counts[2] = slices[2]; }
  }
  auto part = [&](int d, int q, int32_t &lo, int32_t &hi) {   // q-th (from 0) of counts[d] parts of direction d
    const int c = counts[d], sz = size[d] / c, rest = size[d] % c;
    if (q < rest) { lo = 1 + (sz + 1) * q; hi = (sz + 1) * (q + 1); }
    else { lo = 1 + sz * q + rest; hi = sz * (q + 1) + rest; }
  };
  if (counts[0] == 1) {
    const int nb = n < counts[1] * counts[2] ? n : counts[1] * counts[2];
    for (int t = 0; t < nb; t++) {
      int32_t b[6] = {1, iv, 0, 0, 0, 0};
      part(1, t % counts[1], b[2], b[3]); part(2, t / counts[1], b[4], b[5]);
      out.insert(out.end(), b, b + 6);
    }
  } else {
    const int nb = n < counts[0] * counts[1] ? n : counts[0] * counts[1];
    for (int t = 0; t < nb; t++) {
      int32_t b[6] = {0, 0, 0, 0, 1, kv};
      part(0, t % counts[0], b[0], b[1]); part(1, t / counts[0], b[2], b[3]);
      out.insert(out.end(), b, b + 6);
    }
  }
  return out;
}

extern "C" int fs_zone_set_lu_blocks(fs_ctx *ctx, int zone, int nblocks, const int32_t *ind_sdm);

extern "C" int fs_set_lu_threads(fs_ctx *ctx, int nthreads)
{
  if (!ctx || nthreads < 1) return ctx ? fail(ctx, FS_ERR_ARG, "fs_set_lu_threads: at least one thread") : FS_ERR_ARG;
  if (ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_set_lu_threads: call before fs_upload");
  if (!ctx->implicit) return fail(ctx, FS_ERR_STATE, "fs_set_lu_threads: the zones do not use the implicit scheme");
  for (int zi = 0; zi < (int)ctx->zones.size(); zi++) {
    const Zone &zn = ctx->zones[zi];
    const std::vector<int32_t> b = lu_thread_blocks(zn.z.iv, zn.z.jv, zn.z.kv, nthreads, 10);
    const int rc = fs_zone_set_lu_blocks(ctx, zi, (int)(b.size() / 6), b.data());
    if (rc != FS_OK) return rc;
  }
  return FS_OK;
}

extern "C" int fs_zone_lu_blocks(fs_ctx *ctx, int zone, int32_t *ind_lu, int capacity)
{
  if (!ctx || capacity < 0 || (capacity > 0 && !ind_lu)) return ctx ? fail(ctx, FS_ERR_ARG, "fs_zone_lu_blocks: bad argument") : FS_ERR_ARG;
  if (zone < 0 || zone >= (int)ctx->zones.size()) return fail(ctx, FS_ERR_ARG, "fs_zone_lu_blocks: zone index out of range");
  const Zone &zn = ctx->zones[zone];
  const int nb = (int)(zn.lu_blocks.size() / 6);
  for (int q = 0; q < 6 * nb && q < 6 * capacity; q++) ind_lu[q] = zn.lu_blocks[q];
  return nb;
}

extern "C" int fs_zone_set_lu_blocks(fs_ctx *ctx, int zone, int nblocks, const int32_t *ind_sdm)
{
  if (!ctx || (nblocks > 0 && !ind_sdm)) return ctx ? fail(ctx, FS_ERR_ARG, "fs_zone_set_lu_blocks: null argument") : FS_ERR_ARG;
  if (zone < 0 || zone >= (int)ctx->zones.size()) return fail(ctx, FS_ERR_ARG, "fs_zone_set_lu_blocks: zone index out of range");
  if (ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_zone_set_lu_blocks: call before fs_upload");
  Zone &zn = ctx->zones[zone];
  if (!ctx->implicit) return fail(ctx, FS_ERR_STATE, "fs_zone_set_lu_blocks: the zone does not use the implicit scheme");
  if (nblocks <= 1) { zn.lu_blocks.clear(); return FS_OK; }
  if (nblocks > 4096) return fail(ctx, FS_ERR_ARG, "fs_zone_set_lu_blocks: at most 4096 blocks");
  if (zn.pint[LU_MATCH] != 0)
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_set_lu_blocks: thread blocks with lu_match = 1 are not supported");
  // the blocks tile the interior: each inside it, pairwise disjoint, volumes adding up to the zone's
  const int hi[3] = {zn.z.iv, zn.z.jv, zn.z.kv};
  long long cells = 0;
  for (int b = 0; b < nblocks; b++) {
    const int32_t *q = ind_sdm + 6 * b;
    long long v = 1;
    for (int d = 0; d < 3; d++) {
      if (q[2 * d] < 1 || q[2 * d + 1] > hi[d] || q[2 * d + 1] < q[2 * d])
        return fail(ctx, FS_ERR_ARG, "fs_zone_set_lu_blocks: block " + std::to_string(b) + " is empty or leaves the interior of the zone");
      v *= q[2 * d + 1] - q[2 * d] + 1;
    }
    cells += v;
    for (int a = 0; a < b; a++) {
      const int32_t *o = ind_sdm + 6 * a;
      bool apart = false;
      for (int d = 0; d < 3; d++) apart = apart || q[2 * d] > o[2 * d + 1] || o[2 * d] > q[2 * d + 1];
      if (!apart) return fail(ctx, FS_ERR_ARG, "fs_zone_set_lu_blocks: blocks " + std::to_string(a) + " and " + std::to_string(b) + " overlap");
    }
  }
  if (cells != (long long)hi[0] * hi[1] * hi[2]) return fail(ctx, FS_ERR_ARG, "fs_zone_set_lu_blocks: the blocks do not cover the interior of the zone");
  // LU range of a block: its own bounds, and on a face of the zone that of the zone (one ghost layer more over a
  // wall, Fast/FastS/FastS/BC/BCzone.cpp:99-127: the test is made per thread block)
  zn.lu_blocks.assign((size_t)6 * nblocks, 0);
  for (int b = 0; b < nblocks; b++)
    for (int d = 0; d < 3; d++) {
      const int32_t lo = ind_sdm[6 * b + 2 * d], up = ind_sdm[6 * b + 2 * d + 1];
      zn.lu_blocks[6 * b + 2 * d] = lo == 1 ? zn.lu[2 * d] : lo;
      zn.lu_blocks[6 * b + 2 * d + 1] = up == hi[d] ? zn.lu[2 * d + 1] : up;
    }
  return FS_OK;
}

// ---- sub-zones of 'implicit_local': the host logic on plain arrays (fasts_subzones.hip) ----------------------------
extern "C" int fs_subzone_skip_lu(int nitmax, int mx_ssiter, int32_t *iskip_lu)
{
  if (!iskip_lu || mx_ssiter < 1 || nitmax > mx_ssiter) return FS_ERR_ARG;
  fsz_skip_lu(nitmax, mx_ssiter, iskip_lu);
  return FS_OK;
}

extern "C" int fs_subzone_boxes(const int32_t *ijkv, const int32_t *size_ssdom, int iseuil, const int32_t *it_target,
                                int32_t *no_lu, int32_t *count, int32_t *ind_dm, int max_boxes)
{
  if (!ijkv || !size_ssdom || !it_target || !no_lu || !count || !ind_dm || max_boxes < 0) return FS_ERR_ARG;
  int nblk[3], size[3], iv[3];
  for (int d = 0; d < 3; d++) {
    if (ijkv[d] < 1 || size_ssdom[d] < 1) return FS_ERR_ARG;
    iv[d] = ijkv[d]; size[d] = size_ssdom[d];
    nblk[d] = ijkv[d] / size_ssdom[d] > 1 ? ijkv[d] / size_ssdom[d] : 1;
  }
  std::vector<int32_t> boxes;
  const int nb = fsz_boxes(nblk, size, iv, iseuil, it_target, no_lu, count, boxes);
  if (nb > max_boxes) return FS_ERR_ARG;
  memcpy(ind_dm, boxes.data(), sizeof(int32_t) * boxes.size());
  return nb;
}

extern "C" int fs_subzone_residual(const int32_t *ijkv, const int32_t *size_ssdom, int nssiter, int neq, int threed,
                                   int nstep, int it_bloc, double epsi, const double *norms, const int32_t *no_lu,
                                   int nis, int32_t *it_lu, int32_t *it_target, int32_t *it_target_old, int32_t *it_temp,
                                   double *rdm)
{
  if (!ijkv || !size_ssdom || !norms || !no_lu || !it_lu || !it_target || !it_target_old || !it_temp || !rdm ||
      nssiter < 1 || neq < 5 || neq > 6 || nstep < 1 || nstep > nssiter || it_bloc < 1 || it_bloc > nssiter)
    return FS_ERR_ARG;
  FsSubZones s;
  int iv[3] = {ijkv[0], ijkv[1], ijkv[2]}, sz[3] = {size_ssdom[0], size_ssdom[1], size_ssdom[2]};
  for (int d = 0; d < 3; d++) if (iv[d] < 1 || sz[d] < 1) return FS_ERR_ARG;
  fsz_init(s, iv, sz, nssiter, neq, threed != 0);
  const size_t n = (size_t)s.n;
  s.it_lu.assign(it_lu, it_lu + n); s.it_target.assign(it_target, it_target + n);
  s.it_target_old.assign(it_target_old, it_target_old + n); s.it_temp.assign(it_temp, it_temp + n);
  s.rdm.assign(rdm, rdm + s.rdm.size());
  s.it_bloc = it_bloc;
  fsz_residual(s, nstep, epsi, norms, no_lu, nis);
  memcpy(it_lu, s.it_lu.data(), sizeof(int32_t) * n); memcpy(it_target, s.it_target.data(), sizeof(int32_t) * n);
  memcpy(it_target_old, s.it_target_old.data(), sizeof(int32_t) * n); memcpy(it_temp, s.it_temp.data(), sizeof(int32_t) * n);
  memcpy(rdm, s.rdm.data(), sizeof(double) * s.rdm.size());
  return FS_OK;
}

extern "C" int fs_set_local_newton(fs_ctx *ctx, int lssiter_loc)
{
  if (!ctx || (lssiter_loc != 0 && lssiter_loc != 1)) return FS_ERR_ARG;
  if (ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_set_local_newton: must be called before fs_upload");
  ctx->local_newton = lssiter_loc;
  return FS_OK;
}

extern "C" int fs_set_first_it(fs_ctx *ctx, int first_it)
{
  if (!ctx || (first_it != 0 && first_it != 1)) return FS_ERR_ARG;
  ctx->first_it = first_it;
  return FS_OK;
}

// Ghost range of a BC window on the whole zone, no corner overflow
// (Fast/FastC/FastC/BC/indice_cl_sdm.for:46-99 with npass = 0)
static bool bc_window(const int32_t *p, int idir, const int32_t *fen, FsBcWin &w)
{
  const int ific = p[NIJK + 3], ifick = p[NIJK + 4];
  const int sdm[6] = {1, p[IJKV], 1, p[IJKV + 1], 1, p[IJKV + 2]};
  if (!(sdm[0] <= fen[1] && sdm[1] >= fen[0] && sdm[2] <= fen[3] && sdm[3] >= fen[2] && sdm[4] <= fen[5] &&
        sdm[5] >= fen[4]))
    return false;
  int cl[6];
  for (int d = 0; d < 3; d++) {
    cl[2 * d] = fen[2 * d] > sdm[2 * d] ? fen[2 * d] : sdm[2 * d];
    cl[2 * d + 1] = fen[2 * d + 1] < sdm[2 * d + 1] ? fen[2 * d + 1] : sdm[2 * d + 1];
  }
  switch (idir) {
    case 1: cl[1] = cl[0] - 1; cl[0] = cl[0] - ific; break;
    case 2: cl[1] = cl[0] + ific; cl[0] = cl[0] + 1; break;
    case 3: cl[3] = cl[2] - 1; cl[2] = cl[2] - ific; break;
    case 4: cl[3] = cl[2] + ific; cl[2] = cl[2] + 1; break;
    case 5: cl[5] = cl[4] - 1; cl[4] = cl[4] - ifick; break;
    default: cl[5] = cl[4] + ifick; cl[4] = cl[4] + 1; break;
  }
  for (int d = 0; d < 3; d++) { w.lo[d] = cl[2 * d]; w.hi[d] = cl[2 * d + 1]; }
  return true;
}

extern "C" int fs_zone_add(fs_ctx *ctx, const int32_t *pint, int n_int, const double *preal, int n_real,
                           double *ro, double *ro_m1, double *ro_p1, const double *mtr, int n_mtr)
{
  if (!ctx || !pint || !preal || !ro || !ro_m1 || !ro_p1 || !mtr) return ctx ? fail(ctx, FS_ERR_ARG, "fs_zone_add: null argument") : FS_ERR_ARG;
  if (n_int < 137 || n_real < 41) return fail(ctx, FS_ERR_ARG, "fs_zone_add: Parameter_int/real too short");
  if (ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_zone_add: zones must be registered before fs_upload");
  const int32_t *p = pint;
  if (p[ITYPZONE] < 0 || p[ITYPZONE] > 3) return fail(ctx, FS_ERR_ARG, "fs_zone_add: zone type must be 0, 1, 2 or 3");
  const bool twod = p[ITYPZONE] == 3, homo = p[ITYPZONE] == 1;
  if (p[ITYPCP] != 2 && p[ITYPCP] != 1) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: temporal scheme must be explicit or implicit");
  const bool implicit = p[ITYPCP] == 1;
  if (!ctx->zones.empty() && implicit != ctx->implicit)
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: explicit and implicit zones cannot share a context");
  if (implicit && (p[NB_RELAX] < 1 || p[IMPLICITSOLVER] != 0))
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: the implicit scheme is LU-SGS / LU-SSOR (no GMRES)");
  const bool sa = p[IFLOW] == 3 && p[NEQ] == 6;
  if (!sa && (p[NEQ] != 5 || p[IFLOW] > 2)) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: governing equations must be Euler, NSLaminar or NSTurbulent (Spalart-Allmaras)");
  if (sa && (p[47] > 1 || p[SA_DIST] != 1))
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: the Spalart-Allmaras model is implemented as plain SA (no DES variant), "
                "with a TurbulentDistance field");
  // slopes of the reference's flux tables: o3 and o3sc for every scheme; o1 and minmod exist for roe alone
  // (ROE/fluroe_select.for); o3sc is implemented for 5 equations
  if (p[SLOPE] != 2 && p[SLOPE] != 6 && !(p[KFLUDOM] == 5 && (p[SLOPE] == 1 || p[SLOPE] == 3)))
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: slope must be 'o3' or 'o3sc' (or, with roe, 'o1' / 'minmod')");
  if (p[SLOPE] == 6 && sa)
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: the o3sc slope is implemented for Euler and NSLaminar");
  if (p[SFD] != 0 && p[SFD] != 1) return fail(ctx, FS_ERR_ARG, "fs_zone_add: SFD must be 0 or 1");
  if (p[SFD] == 1 && !implicit && p[NSSITER] < 2)
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: selective frequency damping runs after the last stage but one: it needs two stages");
  if (p[KFLUDOM] != 1 && p[KFLUDOM] != 2 && p[KFLUDOM] != 5) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: scheme must be ausmpred, senseur or roe");
  if (p[LALE] != 0 || p[EXPLOC] != 0) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: moving mesh / local explicit time stepping are not implemented");
  if (p[DTLOC] != 0 && p[DTLOC] != 1)
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: DTLOC must be 0 or 1");
  if (!implicit && (p[RK] < 1 || p[RK] > 3 || p[NSSITER] != p[RK])) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: Runge-Kutta order must be 1..3");
  if (implicit && p[NSSITER] < 2) return fail(ctx, FS_ERR_ARG, "fs_zone_add: an implicit step needs at least one Newton sub-iteration (NSSITER >= 2)");
  if (p[NIJK + 3] != 2 || p[NIJK + 4] != (twod ? 0 : 2))
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: two ghost layers in every direction are required (none along k on a two-dimensional zone)");
  if (n_mtr < 4) return fail(ctx, FS_ERR_ARG, "fs_zone_add: metric array too short");
  if (p[ITYPZONE] == 0) {
    // metric arrays dimensioned on the nodes, strides (1, ni+1, (ni+1)(nj+1)) (nature_geom_dom.for:186-503)
    const long long nn = (long long)(p[NIJK] + 1) * (p[NIJK + 1] + 1) * (p[NIJK + 2] + 1);
    if (p[NDIMDX_MTR] != nn || p[NEQ_IJ] != 3 || p[NEQ_K] != 3 || p[NIJK_MTR] != 1 || p[NIJK_MTR + 1] != p[NIJK] + 1 ||
        p[NIJK_MTR + 2] != (p[NIJK] + 1) * (p[NIJK + 1] + 1) || p[NIJK_MTR + 3] != p[NIJK + 3] || p[NIJK_MTR + 4] != p[NIJK + 4])
      return fail(ctx, FS_ERR_ARG, "fs_zone_add: metric layout of a curvilinear zone must be node-dimensioned with 3 components per direction");
    if ((long long)n_mtr < 10 * nn) return fail(ctx, FS_ERR_ARG, "fs_zone_add: metric array too short for a curvilinear zone");
  }
  if (homo) {
    // ti, tj(NDIMDX_MTR, 2), tk(NDIMDX_MTR, 1) and vol on the nodes of one k-plane (nature_geom_dom.for:468-479)
    const long long nn = (long long)(p[NIJK] + 1) * (p[NIJK + 1] + 1);
    if (p[NDIMDX_MTR] != nn || p[NEQ_IJ] != 2 || p[NEQ_K] != 1 || p[NIJK_MTR] != 1 || p[NIJK_MTR + 1] != p[NIJK] + 1 ||
        p[NIJK_MTR + 2] != 0 || p[NIJK_MTR + 3] != 2 || p[NIJK_MTR + 4] != 0)
      return fail(ctx, FS_ERR_ARG, "fs_zone_add: metric layout of a k-homogeneous zone must be that of one node plane, 2 + 2 + 1 components");
    if ((long long)n_mtr < 6 * nn) return fail(ctx, FS_ERR_ARG, "fs_zone_add: metric array too short for a k-homogeneous zone");
  }
  if (twod) {
    // ti, tj(NDIMDX_MTR, 2) and vol on the nodes of one k-plane, no tk (nature_geom_dom.for:186-198)
    const long long nn = (long long)(p[NIJK] + 1) * (p[NIJK + 1] + 1);
    if (p[NIJK + 2] != 1 || p[IJKV + 2] != 1 || p[NDIMDX_MTR] != nn || p[NEQ_IJ] != 2 || p[NEQ_K] != 0 || p[NIJK_MTR] != 1 ||
        p[NIJK_MTR + 1] != p[NIJK] + 1 || p[NIJK_MTR + 2] != 0 || p[NIJK_MTR + 3] != 2 || p[NIJK_MTR + 4] != 0)
      return fail(ctx, FS_ERR_ARG, "fs_zone_add: metric layout of a two-dimensional zone must be node-dimensioned with 2 components per direction");
    if ((long long)n_mtr < 5 * nn) return fail(ctx, FS_ERR_ARG, "fs_zone_add: metric array too short for a two-dimensional zone");
  }
  if ((long long)p[NIJK] * p[NIJK + 1] * p[NIJK + 2] > p[NDIMDX]) return fail(ctx, FS_ERR_ARG, "fs_zone_add: NDIMDX smaller than the zone");
  if (p[IJKV] < 4 || p[IJKV + 1] < 4 || (!twod && p[IJKV + 2] < 4)) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: at least 4 interior cells per direction");

  Zone zn;
  zn.pint.assign(pint, pint + n_int);
  zn.preal.assign(preal, preal + n_real);
  zn.host[0] = ro; zn.host[1] = ro_m1; zn.host[2] = ro_p1;
  FsZone &z = zn.z;
  z.ni = p[NIJK]; z.nj = p[NIJK + 1]; z.nk = p[NIJK + 2]; z.ific = p[NIJK + 3]; z.ifick = p[NIJK + 4];
  z.iv = p[IJKV]; z.jv = p[IJKV + 1]; z.kv = p[IJKV + 2];
  z.nd = p[NDIMDX]; z.incj = z.ni; z.inck = (long long)z.ni * z.nj;
  z.kflu = p[KFLUDOM]; z.iflow = p[IFLOW]; z.neq = p[NEQ]; z.rk = p[RK]; z.nssiter = p[NSSITER];
  z.sensortype = p[SENSORTYPE];
  z.split = 0;
  // constants, same expressions as Fast/FastS/FastS/Compute/template_FluxAndBalance.for:108-156
  const double *r = preal;
  z.tcx = mtr[0]; z.tcy = mtr[1]; z.tcz = mtr[2];
  z.si = std::fabs(z.tcx); z.sj = std::fabs(z.tcy); z.sk = std::fabs(z.tcz);
  z.volinv = 0.5 / mtr[3];
  z.gam = r[GAMMA];
  z.rgp = r[CVINF] * (z.gam - 1.);
  z.gam1 = z.gam / (z.gam - 1.);
  z.gam3 = z.gam1 / r[PRANDT] * z.rgp;
  z.psiroe = r[PSIROE];
  z.c1 = 0.02 * r[VINF];
  z.c2 = 0.02 / (r[VINF] * r[ROINF]);
  z.c3 = -2.;
  z.opt0 = (double)p[SENSORTYPE];
  double c6 = 1. / 6.;
  z.c4 = 5. * c6; z.c5 = 2. * c6; z.c6 = -1. * c6;
  z.cvisq = 1. / 3;
  z.slope = p[SLOPE];
  z.tmin_1 = 100. / r[6];   // TINF (template_FluxAndBalance.for:136)
  z.dt = r[DTC];
  z.coe = r[DTC] / mtr[3];  // tstb3_global.for:72-78
  z.vol0 = mtr[3];
  // a two-dimensional zone runs through the curvilinear kernels: same formulas with zero z metrics and no k faces
  // a k-homogeneous zone too: general formulas, zero-padded metrics of one plane, no metric stride along k
  z.ityp = (twod || homo) ? 0 : p[ITYPZONE]; z.twod = twod ? 1 : 0; z.homo = homo ? 1 : 0; z.ndm = 0; z.mincj = z.minck = 0;
  z.mt[0] = z.mt[1] = z.mt[2] = z.vol = nullptr;
  if (z.ityp == 0) {
    z.ndm = p[NDIMDX_MTR]; z.mincj = p[NIJK_MTR + 1]; z.minck = p[NIJK_MTR + 2];
    // the scalars of the Cartesian path are not used by the curvilinear kernels
    z.tcx = z.tcy = z.tcz = z.si = z.sj = z.sk = z.volinv = z.coe = z.vol0 = 0.;
  }
  z.cv = r[CVINF]; z.cvinv = 1. / z.cv; z.cvinv2 = 0.5 * z.cvinv;  // core3ark3.for:44-46
  z.gam4 = z.gam1 / r[PRANDT_TUR] * z.rgp;
  z.temp01 = 1. / r[TEMP0];
  z.sa_coesut = r[XMUL0] * (1. + r[CS] * z.temp01);
  z.ratiom = r[SA_RATIOM_R];
  z.sa_low_re = (double)p[SA_LOW_RE]; z.sa_rot_corr = (double)p[SA_ROT_CORR];
  {  // invist.for:49-51
    double cmus1 = r[CS], temp01 = 1. / r[TEMP0];
    z.cmus1 = cmus1;
    z.coesut = r[XMUL0] * (1. + cmus1 * temp01) * std::sqrt(temp01);
  }
  // BC table (Fast/FastC/FastC/PyTree.py:2958-3014)
  const int pt_bcs = p[PT_BC];
  if (pt_bcs <= 0 || pt_bcs >= n_int) return fail(ctx, FS_ERR_ARG, "fs_zone_add: PT_BC out of range");
  const int nb = p[pt_bcs];
  if (nb < 0 || (long long)pt_bcs + 1 + 2 * (long long)nb > n_int)
    return fail(ctx, FS_ERR_ARG, "fs_zone_add: BC table overruns Parameter_int");
  for (int ndf = 0; ndf < nb; ndf++) {   // every window record lies inside Parameter_int, data sizes included
    const int pt = p[pt_bcs + 1 + ndf];
    if (pt <= 0 || (long long)pt + 9 > n_int || p[pt + 8] < 0 || (long long)pt + 9 + p[pt + 8] > n_int)
      return fail(ctx, FS_ERR_ARG, "fs_zone_add: BC table overruns Parameter_int");
  }
  for (int ndf = 0; ndf < nb; ndf++) {
    const int pt = p[pt_bcs + 1 + ndf];
    FsBcWin w;
    w.type = p[pt]; w.idir = p[pt + 1];
    if (w.type == 22) continue;  // BCOverlap: filled by the join transfer
    // dispatch of Fast/FastS/FastS/BC/BCzone.cpp:262-283,451-461,513-523,578-586
    const int tag = w.type;
    // DEFAULT_STATE of BCzone.cpp:26-48: a far field / inflow may carry its own uniform conservative state, as one
    // array of 6 values or as the 21 scalars of a ReferenceState node (state at positions 3..7 and 16)
    const bool own_state = (tag == 1 || tag == 13 || tag == 2 || tag == 10) &&
                           (p[pt + 8] == 21 || (p[pt + 8] == 1 && pt + 9 < n_int && p[pt + 9] == 6));
    if (tag == 11 || tag == 0 || tag == 1 || tag == 14) w.type = tag;   // 14: BCExtrapolateRANS, nu_tilde only
    else if ((tag == 13 || tag == 2) && (p[pt + 8] == 0 || own_state)) w.type = tag;   // BCInflow / BCInflowSupersonic on a uniform state
    else if (tag == 2 && p[pt + 8] == p[NEQ] && pt + 9 < n_int && p[pt + 9] > 6) w.type = tag;   // its own state: BCzone.cpp:421-448
    // BCOutflow on a uniform state; with 6 equations the reference reads a variable it never sets (BCOutflow_firstrank.for:9)
    else if (tag == 10 && p[NEQ] == 5 && (p[pt + 8] == 0 || own_state)) w.type = tag;
    else if (tag == 17 && p[pt + 8] == p[NEQ]) w.type = tag;   // BCInj1: d0x, d0y, d0z, pa, ha [, nu_tilde], BCzone.cpp:594-618
    else if (tag == 20 && p[pt + 8] == p[NEQ]) w.type = tag;   // BCInjMFR: d0x, d0y, d0z, qp, ha [, nu_tilde], BCzone.cpp:377-398
    else if (tag == 21 && p[pt + 8] == 1) w.type = tag;        // BCOutMFR: qp, BCzone.cpp:400-415
    else if (tag == 3 || tag == 5 || (tag == 4 && p[IFLOW] == 1)) w.type = FS_BC_WALL_SLIP;
    else if ((tag == 4 || tag == 6) && p[IFLOW] > 1) w.type = FS_BC_WALL_NOSLIP;
    else if (tag == 16 || (tag == 32 && p[IFLOW] > 1)) w.type = tag;   // BCOutpres, BCWallViscousIsothermal: window data
    else {
      char msg[128];
      snprintf(msg, sizeof msg, "fs_zone_add: boundary condition tag %d is not implemented", tag);
      return fail(ctx, FS_ERR_UNSUPPORTED, msg);
    }
    if (w.idir < 1 || w.idir > 6) return fail(ctx, FS_ERR_ARG, "fs_zone_add: BC window is not on a zone face");
    long long data_off = -1;
    w.data = nullptr; w.dlo[0] = w.dlo[1] = 0; w.dn = 1; w.dsize = 0;
    w.nit = p[65]; w.tol = r[37];   // NIT_INFLOW, EPSI_INFLOW
    if ((tag == 2 && p[pt + 8] != 0 && !own_state) || tag == 17 || tag == 20 || tag == 21) {
      // NEQ arrays of one size, one value per face cell of the window: rho, rho u, rho v, rho w, rho E [, rho nu_tilde]
      const int nbdata = p[pt + 8], size = p[pt + 9];
      const long long off = p[pt_bcs + 1 + ndf + nb];
      const int32_t *fen = p + pt + 2;
      const int da = w.idir <= 2 ? 1 : 0, db = w.idir <= 4 ? 2 : 1;
      const int na = fen[2 * da + 1] - fen[2 * da] + 1, nbb = fen[2 * db + 1] - fen[2 * db] + 1;
      if (pt + 8 + nbdata >= n_int || size != na * nbb || p[pt + 8 + nbdata] != nbdata * size || off < 0 ||
          off + (long long)nbdata * size > n_real)
        return fail(ctx, FS_ERR_ARG, "fs_zone_add: BC window data do not match the BC window");
      data_off = (long long)zn.bc_data.size();
      zn.bc_data.insert(zn.bc_data.end(), r + off, r + off + (long long)nbdata * size);
      w.dlo[0] = fen[2 * da]; w.dlo[1] = fen[2 * db]; w.dn = na; w.dsize = size;
    } else if (tag == 16 || tag == 32) {
      // BCzone.cpp:257,545-551,577-585: one array of window data, optionally after/before a mesh-velocity
      // coefficient of size 1; stored in Parameter_real at the pointer that follows the BC pointers
      const int nbdata = p[pt + 8];
      if (nbdata < 1 || nbdata > 2 || pt + 8 + nbdata >= n_int)
        return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: BCOutpres / BCWallViscousIsothermal need one array of window data");
      long long off = p[pt_bcs + 1 + ndf + nb];
      int size = p[pt + 9];
      if (size == 1) { off += 1; size = nbdata == 2 ? p[pt + 10] - p[pt + 9] : 0; }
      const int32_t *fen = p + pt + 2;
      const int da = w.idir <= 2 ? 1 : 0, db = w.idir <= 4 ? 2 : 1;
      const int na = fen[2 * da + 1] - fen[2 * da] + 1, nbb = fen[2 * db + 1] - fen[2 * db] + 1;
      if (size != na * nbb || off < 0 || off + size > n_real)
        return fail(ctx, FS_ERR_ARG, "fs_zone_add: BC window data do not match the BC window");
      data_off = (long long)zn.bc_data.size();
      zn.bc_data.insert(zn.bc_data.end(), r + off, r + off + size);
      w.dlo[0] = fen[2 * da]; w.dlo[1] = fen[2 * db]; w.dn = na;
    } else if (p[pt + 8] != 0 && tag != 11 && !own_state)
      return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: data on this boundary condition are not implemented");
    if (!bc_window(p, w.idir, p + pt + 2, w)) continue;
    // reference state of BCzone.cpp:26-38 and the face metric along the normal
    w.st[0] = r[ROINF]; w.st[1] = r[VXINF] * r[ROINF]; w.st[2] = r[VYINF] * r[ROINF]; w.st[3] = r[VZINF] * r[ROINF];
    w.st[4] = r[PINF] / (r[GAMMA] - 1.) + 0.5 * (r[VXINF] * w.st[1] + r[VYINF] * w.st[2] + r[VZINF] * w.st[3]);
    w.tcn = w.idir <= 2 ? mtr[0] : (w.idir <= 4 ? mtr[1] : mtr[2]);
    w.nue = tag == 14 ? r[RONUTILDEINF] : r[RONUTILDEINF] / w.st[0];
    if (own_state) {
      const long long off = p[pt_bcs + 1 + ndf + nb];
      const int nval = p[pt + 8] == 21 ? 21 : 6;
      if (off < 0 || off + nval > n_real) return fail(ctx, FS_ERR_ARG, "fs_zone_add: BC state data overrun Parameter_real");
      const double *d = r + off;
      if (nval == 21) { for (int e = 0; e < 5; e++) w.st[e] = d[3 + e]; w.nue = d[16] / w.st[0]; }
      else { for (int e = 0; e < 5; e++) w.st[e] = d[e]; w.nue = d[5] / w.st[0]; }
    }
    switch (w.idir) {  // bvbs_periodique.for: incp = IJKV(dir) * stride, towards the interior
      case 1: w.shift = z.iv; break;
      case 2: w.shift = -(long long)z.iv; break;
      case 3: w.shift = (long long)z.jv * z.incj; break;
      case 4: w.shift = -(long long)z.jv * z.incj; break;
      case 5: w.shift = (long long)z.kv * z.inck; break;
      default: w.shift = -(long long)z.kv * z.inck; break;
    }
    long long cells = (long long)(w.hi[0] - w.lo[0] + 1) * (w.hi[1] - w.lo[1] + 1) * (w.hi[2] - w.lo[2] + 1);
    if (cells > zn.bc_max_cells) zn.bc_max_cells = cells;
    zn.bc.push_back(w);
    zn.bc_data_off.push_back(data_off);
    if (p[KFLUDOM] == 5 && (tag == 3 || tag == 4 || tag == 6)) {   // bfl3.for:58-63 (tag 5 is not corrected)
      Zone::RoeWall rw;
      rw.tag = tag; rw.idir = w.idir;
      for (int a = 0; a < 3; a++) { rw.lo[a] = w.lo[a]; rw.hi[a] = w.hi[a]; }
      zn.roe_walls.push_back(rw);
    }
  }
  if (p[SLOPE] == 6 && !zn.roe_walls.empty())
    return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: the o3sc slope with the roe scheme is implemented on zones without walls "
                "(the wall-flux correction of the reference has o3sc variants of its own)");
  if (implicit) {
    // BCzone with lrhs = 1 (Fast/FastS/FastS/BC/BCzone.cpp:93-191,193-226,262-264): faces that carry a wall
    // BC (tags 3..7) get the ghost residual zeroed over the whole face (extrapolation with c1 = 0), then their BCs
    // applied to it (walls as walls, the rest as the same zeroing); the LU range grows by one ghost layer there
    int lskip[6] = {1, 1, 1, 1, 1, 1};
    for (int ndf = 0; ndf < nb; ndf++) {
      const int pt = p[pt_bcs + 1 + ndf], tag = p[pt], idir = p[pt + 1];
      if (tag >= 3 && tag <= 7 && idir >= 1 && idir <= 6) lskip[idir - 1] = 0;
    }
    const int ijkv[3] = {z.iv, z.jv, z.kv};
    for (int d = 0; d < 3; d++) {
      zn.lu[2 * d] = 1 - (1 - lskip[2 * d]);
      zn.lu[2 * d + 1] = ijkv[d] + (1 - lskip[2 * d + 1]);
    }
    for (int f = 0; f < 6; f++) {
      if (lskip[f]) continue;
      int32_t fen[6] = {1, z.iv, 1, z.jv, 1, z.kv};
      const int d = f / 2;
      if (f % 2 == 0) fen[2 * d + 1] = 1; else fen[2 * d] = fen[2 * d + 1];
      FsBcWin w;
      memset(&w, 0, sizeof w);
      w.type = FS_BC_ZERO; w.idir = f + 1;
      if (!bc_window(p, w.idir, fen, w)) continue;
      long long cells = (long long)(w.hi[0] - w.lo[0] + 1) * (w.hi[1] - w.lo[1] + 1) * (w.hi[2] - w.lo[2] + 1);
      if (cells > zn.rhs_max_cells) zn.rhs_max_cells = cells;
      zn.rhs_face.push_back(w);
    }
    for (int ndf = 0; ndf < nb; ndf++) {
      const int pt = p[pt_bcs + 1 + ndf], tag = p[pt], idir = p[pt + 1];
      if (tag == 22 || idir < 1 || idir > 6 || lskip[idir - 1]) continue;
      if (tag == 13 || tag == 32)
        return fail(ctx, FS_ERR_UNSUPPORTED, "fs_zone_add: BCInflow / BCWallViscousIsothermal on a zone face that also carries an adiabatic or slip wall is not implemented with the implicit scheme");
      FsBcWin w;
      memset(&w, 0, sizeof w);
      w.idir = idir;
      if (tag == 3 || tag == 5 || (tag == 4 && p[IFLOW] == 1)) w.type = FS_BC_WALL_SLIP;
      else if ((tag == 4 || tag == 6) && p[IFLOW] > 1) w.type = FS_BC_WALL_NOSLIP;
      else w.type = FS_BC_ZERO;
      if (!bc_window(p, idir, p + pt + 2, w)) continue;
      w.tcn = idir <= 2 ? mtr[0] : (idir <= 4 ? mtr[1] : mtr[2]);
      long long cells = (long long)(w.hi[0] - w.lo[0] + 1) * (w.hi[1] - w.lo[1] + 1) * (w.hi[2] - w.lo[2] + 1);
      if (cells > zn.rhs_max_cells) zn.rhs_max_cells = cells;
      zn.rhs_bc.push_back(w);
    }
  }
  if (z.ityp == 0) {   // last step that can fail: nothing to release on the error paths above
    (void)hipSetDevice(ctx->device);
    const size_t nb = sizeof(double) * 10 * (size_t)z.ndm;
    HIPCHK(ctx, hipMalloc(&zn.mtr_dev, nb));
    hipError_t e;
    if (twod) {   // expand [ti(2) | tj(2) | vol] to [ti(3) | tj(3) | tk(3) | vol] with zero z components and zero tk
      std::vector<double> full(10 * (size_t)z.ndm, 0.);
      const size_t n = (size_t)z.ndm;
      memcpy(&full[0], mtr, sizeof(double) * 2 * n);
      memcpy(&full[3 * n], mtr + 2 * n, sizeof(double) * 2 * n);
      memcpy(&full[9 * n], mtr + 4 * n, sizeof(double) * n);
      e = hipMemcpy(zn.mtr_dev, full.data(), nb, hipMemcpyHostToDevice);
    } else if (homo) {   // expand [ti(2) | tj(2) | tk(1) | vol]: zero z components of ti, tj, zero x, y components of tk
      std::vector<double> full(10 * (size_t)z.ndm, 0.);
      const size_t n = (size_t)z.ndm;
      memcpy(&full[0], mtr, sizeof(double) * 2 * n);
      memcpy(&full[3 * n], mtr + 2 * n, sizeof(double) * 2 * n);
      memcpy(&full[8 * n], mtr + 4 * n, sizeof(double) * n);
      memcpy(&full[9 * n], mtr + 5 * n, sizeof(double) * n);
      e = hipMemcpy(zn.mtr_dev, full.data(), nb, hipMemcpyHostToDevice);
    } else
      e = hipMemcpy(zn.mtr_dev, mtr, nb, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(zn.mtr_dev);
    (void)hipFree(zn.coe); (void)hipFree(zn.scratch); (void)hipFree(zn.ssor_work); (void)hipFree(zn.lu_blocks_dev); (void)hipFree(zn.dist); (void)hipFree(zn.rof); (void)hipFree(zn.rhs_face_dev); (void)hipFree(zn.rhs_bc_dev);
      return fail(ctx, FS_ERR_HIP, std::string("fs_zone_add: metric upload: ") + hipGetErrorString(e));
    }
    z.mt[0] = zn.mtr_dev; z.mt[1] = zn.mtr_dev + 3 * z.ndm; z.mt[2] = zn.mtr_dev + 6 * z.ndm;
    z.vol = zn.mtr_dev + 9 * z.ndm;
  }
  ctx->implicit = implicit;
  ctx->zones.push_back(std::move(zn));
  return (int)ctx->zones.size() - 1;
}

static int check_pts(fs_ctx *ctx, int zone, const int32_t *pts, int npts, const char *who)
{
  if (zone < 0 || zone >= (int)ctx->zones.size()) return fail(ctx, FS_ERR_ARG, std::string(who) + ": zone index out of range");
  const FsZone &z = ctx->zones[zone].z;
  const long long n = (long long)z.ni * z.nj * z.nk;
  for (int i = 0; i < npts; i++)
    if (pts[i] < 0 || pts[i] >= n) return fail(ctx, FS_ERR_ARG, std::string(who) + ": cell index outside the zone");
  return FS_OK;
}

extern "C" int fs_join_add(fs_ctx *ctx, int rcv_zone, int donor_zone, const int32_t *rcv_pts,
                           const int32_t *donor_pts, int npts)
{
  if (!ctx || !rcv_pts || npts <= 0) return ctx ? fail(ctx, FS_ERR_ARG, "fs_join_add: bad argument") : FS_ERR_ARG;
  int rc = check_pts(ctx, rcv_zone, rcv_pts, npts, "fs_join_add");
  if (rc) return rc;
  if (donor_zone >= 0) {   // every check before the first allocation: nothing to release on the error paths
    if (!donor_pts) return fail(ctx, FS_ERR_ARG, "fs_join_add: donor list missing");
    rc = check_pts(ctx, donor_zone, donor_pts, npts, "fs_join_add");
    if (rc) return rc;
    if (ctx->zones[donor_zone].z.neq != ctx->zones[rcv_zone].z.neq) return fail(ctx, FS_ERR_ARG, "fs_join_add: zones differ in NEQ");
  }
  Join j;
  j.rcv_zone = rcv_zone; j.npts = npts;
  (void)hipSetDevice(ctx->device);
  HIPCHK(ctx, hipMalloc(&j.rcv_dev, sizeof(int) * npts));
  HIPCHK(ctx, hipMemcpy(j.rcv_dev, rcv_pts, sizeof(int) * npts, hipMemcpyHostToDevice));
  if (donor_zone >= 0) {
    j.donor_zone = donor_zone; j.peer_tag = -1;
    HIPCHK(ctx, hipMalloc(&j.dnr_dev, sizeof(int) * npts));
    HIPCHK(ctx, hipMemcpy(j.dnr_dev, donor_pts, sizeof(int) * npts, hipMemcpyHostToDevice));
  } else {
    j.donor_zone = -1; j.peer_tag = -donor_zone - 1;
  }
  ctx->joins.push_back(j);
  return FS_OK;
}

extern "C" int fs_send_add(fs_ctx *ctx, int donor_zone, int peer_tag, const int32_t *donor_pts, int npts)
{
  if (!ctx || !donor_pts || npts <= 0 || peer_tag < 0) return ctx ? fail(ctx, FS_ERR_ARG, "fs_send_add: bad argument") : FS_ERR_ARG;
  int rc = check_pts(ctx, donor_zone, donor_pts, npts, "fs_send_add");
  if (rc) return rc;
  Send s;
  s.donor_zone = donor_zone; s.peer_tag = peer_tag; s.npts = npts;
  (void)hipSetDevice(ctx->device);
  HIPCHK(ctx, hipMalloc(&s.pts_dev, sizeof(int) * npts));
  HIPCHK(ctx, hipMemcpy(s.pts_dev, donor_pts, sizeof(int) * npts, hipMemcpyHostToDevice));
  s.pts_host.assign(donor_pts, donor_pts + npts);
  ctx->sends.push_back(s);
  return FS_OK;
}

// A sweep launch whose waves stopped waiting for each other (fasts_lusgs.hip: a defect, or a device in trouble) leaves
// a zone half swept: reported at the next point where the host waits for the stream anyway.
static int check_sweeps(fs_ctx *ctx, const char *who)
{
  for (auto &zn : ctx->zones) {
    bool gave_up = zn.lu_flow && fsk_luflow_gave_up(zn.lu_flow) != 0;
    for (auto &bb : zn.box_bcs) gave_up = gave_up || (bb.flow && fsk_luflow_gave_up(bb.flow) != 0);
    if (gave_up)
      return fail(ctx, FS_ERR_HIP, std::string(who) + ": an LU-SGS sweep launch gave up waiting for a predecessor pencil; the "
                  "fields of the context are not to be trusted (FASTS_HIP_LU_FLOW=0 runs one launch per hyperplane)");
  }
  return FS_OK;
}

static int build_shell_lists(fs_ctx *ctx);
static int local_setup(fs_ctx *ctx);

extern "C" int fs_upload(fs_ctx *ctx)
{
  if (!ctx) return FS_ERR_ARG;
  if (ctx->zones.empty()) return fail(ctx, FS_ERR_STATE, "fs_upload: no zone registered");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!ctx->uploaded)   // every precondition before the first allocation: a refused call leaves nothing behind
    for (auto &zn : ctx->zones)
      if (zn.z.neq == 6 && !zn.dist_host)
        return fail(ctx, FS_ERR_STATE, "fs_upload: a Spalart-Allmaras zone needs fs_zone_set_distance first");
  if (!ctx->uploaded)
    for (auto &zn : ctx->zones)
      if (zn.pint[SFD] == 1 && !zn.rof_host)
        return fail(ctx, FS_ERR_STATE, "fs_upload: a zone with selective frequency damping needs fs_zone_set_filtered first");
  if (!ctx->uploaded && ctx->local_newton) {
    if (!ctx->implicit) return fail(ctx, FS_ERR_STATE, "fs_upload: local Newton sub-iterations need the implicit scheme");
    for (auto &zn : ctx->zones)
      if (!zn.lu_blocks.empty() || zn.pint[NB_RELAX] > 1 || zn.pint[LU_MATCH] != 0 || !zn.roe_walls.empty() || zn.z.slope == 6 ||
          zn.pint[SIZE_SSDOM] < 1 || zn.pint[SIZE_SSDOM + 1] < 1 || zn.pint[SIZE_SSDOM + 2] < 1)
        return fail(ctx, FS_ERR_UNSUPPORTED, "fs_upload: local Newton sub-iterations are implemented with one LU-SGS sweep "
                    "per sub-zone (no thread blocks, nb_relax = 1, lu_match = 0, no roe scheme with walls, no o3sc slope)");
  }
  for (auto &zn : ctx->zones) {
    const size_t nb = sizeof(double) * (size_t)zn.z.neq * (size_t)zn.z.nd;
    if (!ctx->uploaded) {
      for (int r = 0; r < 3; r++) HIPCHK(ctx, hipMalloc(&zn.dev[r], nb));
      HIPCHK(ctx, hipMalloc(&zn.drodm, nb));
      HIPCHK(ctx, hipMemsetAsync(zn.drodm, 0, nb, ctx->stream));
      HIPCHK(ctx, hipMalloc(&zn.xmut, sizeof(double) * (size_t)zn.z.nd));
      HIPCHK(ctx, hipMemsetAsync(zn.xmut, 0, sizeof(double) * (size_t)zn.z.nd, ctx->stream));
      if (zn.z.slope == 6) {
        const size_t nw = sizeof(double) * 3 * (size_t)zn.z.nd;
        HIPCHK(ctx, hipMalloc(&zn.wsc, nw));
        HIPCHK(ctx, hipMemsetAsync(zn.wsc, 0, nw, ctx->stream));
        HIPCHK(ctx, hipMalloc(&zn.wflux, 5 * nw));
        HIPCHK(ctx, hipMemsetAsync(zn.wflux, 0, 5 * nw, ctx->stream));
      }
      if (zn.z.kflu == 2) {
        // face sensors: 0/1 values, one byte per cell and direction (curvilinear and SA kernels) or one byte per
        // cell holding the three of them (Cartesian kernel)
        const size_t nw = 3 * (size_t)zn.z.nd;
        HIPCHK(ctx, hipMalloc(&zn.wig, nw));
        HIPCHK(ctx, hipMemsetAsync(zn.wig, 0, nw, ctx->stream));
      }
      if (zn.z.neq == 6) {
        const size_t nbd = sizeof(double) * (size_t)zn.z.nd;
        HIPCHK(ctx, hipMalloc(&zn.dist, nbd));
        HIPCHK(ctx, hipMemsetAsync(zn.dist, 0, nbd, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(zn.dist, zn.dist_host, sizeof(double) * (size_t)zn.z.ni * zn.z.nj * zn.z.nk,
                                   hipMemcpyHostToDevice, ctx->stream));
      }
      if (ctx->implicit || zn.pint[DTLOC] == 1) {   // coe(:,1..5|6); an explicit local time step uses coe(:,1) only
        HIPCHK(ctx, hipMalloc(&zn.coe, nb));
        HIPCHK(ctx, hipMemsetAsync(zn.coe, 0, nb, ctx->stream));
      }
      if (ctx->implicit) {
        HIPCHK(ctx, hipMalloc(&zn.scratch, nb));
        HIPCHK(ctx, hipMemsetAsync(zn.scratch, 0, nb, ctx->stream));
        if (!zn.lu_blocks.empty()) {
          HIPCHK(ctx, hipMalloc(&zn.lu_blocks_dev, sizeof(int) * zn.lu_blocks.size()));
          HIPCHK(ctx, hipMemcpy(zn.lu_blocks_dev, zn.lu_blocks.data(), sizeof(int) * zn.lu_blocks.size(), hipMemcpyHostToDevice));
        }
        if (zn.pint[NB_RELAX] > 1 || zn.pint[LU_MATCH] != 0) {   // increments and accumulator of the SSOR relaxations (Compute/ssor.cpp)
          HIPCHK(ctx, hipMalloc(&zn.ssor_work, 2 * nb));
          HIPCHK(ctx, hipMemsetAsync(zn.ssor_work, 0, 2 * nb, ctx->stream));
        }
      }
      HIPCHK(ctx, hipMalloc(&zn.res_part, sizeof(double) * 12 * (size_t)zn.z.jv * zn.z.kv));
      HIPCHK(ctx, hipMalloc(&zn.res_out, sizeof(double) * (12 * (size_t)zn.z.nssiter + 3)));
      if (!zn.bc_data.empty()) {
        HIPCHK(ctx, hipMalloc(&zn.bc_data_dev, sizeof(double) * zn.bc_data.size()));
        HIPCHK(ctx, hipMemcpy(zn.bc_data_dev, zn.bc_data.data(), sizeof(double) * zn.bc_data.size(), hipMemcpyHostToDevice));
        for (size_t b = 0; b < zn.bc.size(); b++)
          if (zn.bc_data_off[b] >= 0) zn.bc[b].data = zn.bc_data_dev + zn.bc_data_off[b];
      }
      if (!zn.bc.empty()) {
        HIPCHK(ctx, hipMalloc(&zn.bc_dev, sizeof(FsBcWin) * zn.bc.size()));
        HIPCHK(ctx, hipMemcpy(zn.bc_dev, zn.bc.data(), sizeof(FsBcWin) * zn.bc.size(), hipMemcpyHostToDevice));
      }
      if (!zn.rhs_face.empty()) {
        HIPCHK(ctx, hipMalloc(&zn.rhs_face_dev, sizeof(FsBcWin) * zn.rhs_face.size()));
        HIPCHK(ctx, hipMemcpy(zn.rhs_face_dev, zn.rhs_face.data(), sizeof(FsBcWin) * zn.rhs_face.size(), hipMemcpyHostToDevice));
      }
      if (!zn.rhs_bc.empty()) {
        HIPCHK(ctx, hipMalloc(&zn.rhs_bc_dev, sizeof(FsBcWin) * zn.rhs_bc.size()));
        HIPCHK(ctx, hipMemcpy(zn.rhs_bc_dev, zn.rhs_bc.data(), sizeof(FsBcWin) * zn.rhs_bc.size(), hipMemcpyHostToDevice));
      }
    }
    // The level buffers belong to the caller (numpy arrays of the pyTree: pageable memory, copied through a staging
    // buffer at a third of the link's rate).  Registered once, they are copied in place.  FASTS_HIP_PIN=0: leave them;
    // a refusal of the runtime (locked-memory limit) is not an error either.
    {
      static const bool pin = !(getenv("FASTS_HIP_PIN") && atoi(getenv("FASTS_HIP_PIN")) == 0);
      for (int r = 0; r < 3 && pin; r++) {
        bool have = false;
        for (int q = 0; q < 3; q++) have = have || zn.pinned[q] == (void *)zn.host[r];
        if (have) continue;
        if (hipHostRegister(zn.host[r], nb, hipHostRegisterDefault) == hipSuccess) {
          for (int q = 0; q < 3; q++) if (!zn.pinned[q]) { zn.pinned[q] = zn.host[r]; break; }
        } else
          (void)hipGetLastError();
      }
    }
    for (int r = 0; r < 3; r++)
      HIPCHK(ctx, hipMemcpyAsync(zn.dev[r], zn.host[r], nb, hipMemcpyHostToDevice, ctx->stream));
    if (zn.rof_host) {
      if (!zn.rof) HIPCHK(ctx, hipMalloc(&zn.rof, nb));
      HIPCHK(ctx, hipMemcpyAsync(zn.rof, zn.rof_host, nb, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (!ctx->diag_out) HIPCHK(ctx, hipMalloc(&ctx->diag_out, sizeof(double) * 3));
  if (!ctx->uploaded) {
    int rc = build_shell_lists(ctx);
    if (rc) return rc;
    if (ctx->local_newton) {
      rc = local_setup(ctx);
      if (rc) return rc;
    }
  }
  ctx->uploaded = true;
  return FS_OK;
}

extern "C" int fs_download(fs_ctx *ctx, int mask)
{
  if (!ctx) return FS_ERR_ARG;
  if (!ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_download: nothing on the device yet");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int bits[3] = {FS_LEVEL_N, FS_LEVEL_M1, FS_LEVEL_P1};
  for (auto &zn : ctx->zones) {
    const size_t nb = sizeof(double) * (size_t)zn.z.neq * (size_t)zn.z.nd;
    for (int r = 0; r < 3; r++)
      if (mask & bits[r]) HIPCHK(ctx, hipMemcpyAsync(zn.host[r], zn.dev[r], nb, hipMemcpyDeviceToHost, ctx->stream));
    if ((mask & FS_LEVEL_FILTERED) && zn.rof)
      HIPCHK(ctx, hipMemcpyAsync(zn.rof_host, zn.rof, nb, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return check_sweeps(ctx, "fs_download");
}

static int level_role(int level)
{
  return level == FS_LEVEL_N ? 0 : (level == FS_LEVEL_M1 ? 1 : (level == FS_LEVEL_P1 ? 2 : -1));
}

// stage input / output roles (Fast/FastS/FastS/Compute/gsdr3.cpp:137-145): explicit stages alternate between N
// and P1; Newton sub-iterations read N (first) or P1 and always write P1
static void stage_roles(const fs_ctx *ctx, int nstep, int &in, int &out)
{
  if (ctx->implicit) { in = nstep == 1 ? 0 : 2; out = 2; }
  else if (nstep % 2 == 1) { in = 0; out = 2; }
  else { in = 2; out = 0; }
}

static void rk_coefs(int rk, int nstep, double &coefH, double &coefW)
{
  // init_rhs.for:31-32,58-76 ; core3ark3.for:21-24,70-80
  coefH = 0.; coefW = 1.;
  if (rk == 3) {
    coefH = nstep == 2 ? -0.6830127018922193 : (nstep == 3 ? -4. / 3. : 0.);
    coefW = nstep == 1 ? 0.5 : (nstep == 2 ? 0.9106836025229591 : 0.3660254037844387);
  } else if (rk == 2) {
    coefH = nstep == 2 ? -1. : 0.;
    coefW = nstep == 1 ? 1. : 0.5;
  }
}

static int ensure_ready(fs_ctx *ctx, const char *who)
{
  if (!ctx) return FS_ERR_ARG;
  if (!ctx->uploaded) return fail(ctx, FS_ERR_STATE, std::string(who) + ": call fs_upload first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return FS_OK;
}

// Tiles of the stage kernel that contain donor cells of joins to other contexts: their results are
// what the halo messages carry, so they can be computed (and sent) first.
static int build_shell_lists(fs_ctx *ctx)
{
  for (size_t nz = 0; nz < ctx->zones.size(); nz++) {
    Zone &zn = ctx->zones[nz];
    const FsZone &z = zn.z;
    for (auto &s : ctx->sends) if (s.donor_zone == (int)nz) zn.has_sends = true;
    for (auto &j : ctx->joins) if (j.donor_zone < 0 && j.rcv_zone == (int)nz) zn.has_recvs = true;
    // implicit zones, roe + walls and the local time step run their stage in one piece (later kernels of the stage read
    // the whole balance); so do Spalart-Allmaras zones on Cartesian meshes (their stage kernel takes no tile list)
    if (ctx->implicit || (z.iflow == 3 && z.ityp != 0) || !zn.roe_walls.empty() || zn.pint[DTLOC] == 1 || z.slope == 6 ||
        zn.pint[SFD] == 1) continue;   // (a filtered zone changes its cells after the stage kernel)
    zn.z.split = zn.has_sends ? 1 : 0;
    // tiles of the Cartesian stage kernel, patches (owned cells) of the curvilinear one
    const FsTiling t = z.ityp == 0 ? fsk_stage_full_tiling(zn.z) : fsk_stage_dma_tiling(zn.z);
    const int nblk = t.nti * t.ntj * t.ntk;
    std::vector<char> mark((size_t)nblk, 0);
    bool any = false;
    for (auto &s : ctx->sends) {
      if (s.donor_zone != (int)nz) continue;
      any = true;
      for (int n = 0; n < s.npts; n++) {
        const long long l = s.pts_host[n];
        const int i = (int)(l % z.ni) - z.ific + 1, j = (int)((l / z.ni) % z.nj) - z.ific + 1;
        const int k = (int)(l / ((long long)z.ni * z.nj)) - z.ifick + 1;
        if (i < 1 || i > z.iv || j < 1 || j > z.jv || k < 1 || k > z.kv) continue;   // ghost donors: not computed
        const int bi = (i - 1) / t.ti, bj = (j - 1) / t.tj, bk = (k - 1) / t.kchunk;
        mark[(size_t)bi + (size_t)t.nti * (bj + (size_t)t.ntj * bk)] = 1;
      }
    }
    if (!any) continue;
    std::vector<int> shell, inner;
    for (int b = 0; b < nblk; b++) (mark[b] ? shell : inner).push_back(b);
    zn.n_shell = (int)shell.size(); zn.n_inner = (int)inner.size();
    if (zn.n_shell) {
      HIPCHK(ctx, hipMalloc(&zn.shell_dev, sizeof(int) * shell.size()));
      HIPCHK(ctx, hipMemcpy(zn.shell_dev, shell.data(), sizeof(int) * shell.size(), hipMemcpyHostToDevice));
    }
    if (zn.n_inner) {
      HIPCHK(ctx, hipMalloc(&zn.inner_dev, sizeof(int) * inner.size()));
      HIPCHK(ctx, hipMemcpy(zn.inner_dev, inner.data(), sizeof(int) * inner.size(), hipMemcpyHostToDevice));
    }
  }
  return FS_OK;
}

// ---- 'implicit_local': sub-zones --------------------------------------------------------------------------------------
// BC windows of sub-iteration nstep cut to its boxes: a window is applied where a box reaches its face, over the cells
// of the box (Fast/FastC/FastC/BC/indice_cl_sdm.for:46-99 with the box as ind_sdm; Fast/FastS/FastS/Compute/bcs.cpp:14-41)
static int local_bc_tables(fs_ctx *ctx, Zone &zn, int nstep)
{
  const FsZone &z = zn.z;
  const std::vector<int32_t> &bx = zn.sub.boxes[(size_t)nstep - 1];
  const int ends[6] = {1, z.iv, 1, z.jv, 1, z.kv};
  std::vector<FsBcWin> wins;
  long long max_cells = 0;
  for (size_t b = 0; b + 6 <= bx.size(); b += 6)
    for (const FsBcWin &w0 : zn.bc) {
      const int f = w0.idir - 1, d = f / 2;
      if (bx[b + f] != ends[f]) continue;
      FsBcWin w = w0;
      bool empty = false;
      for (int t = 0; t < 3; t++) {
        if (t == d) continue;
        w.lo[t] = std::max(w0.lo[t], (int)bx[b + 2 * t]);
        w.hi[t] = std::min(w0.hi[t], (int)bx[b + 2 * t + 1]);
        if (w.lo[t] > w.hi[t]) empty = true;
      }
      if (empty) continue;
      const long long cells = (long long)(w.hi[0] - w.lo[0] + 1) * (w.hi[1] - w.lo[1] + 1) * (w.hi[2] - w.lo[2] + 1);
      if (cells > max_cells) max_cells = cells;
      wins.push_back(w);
    }
  Zone::BoxBcs &bb = zn.box_bcs[(size_t)nstep - 1];
  bb.n = (int)wins.size(); bb.kinds = bc_kinds(wins); bb.max_cells = max_cells;
  if (wins.size() > bb.cap) {
    (void)hipFree(bb.dev);
    bb.dev = nullptr; bb.cap = 0;
    HIPCHK(ctx, hipMalloc(&bb.dev, sizeof(FsBcWin) * wins.size()));
    bb.cap = wins.size();
  }
  if (!wins.empty()) HIPCHK(ctx, hipMemcpy(bb.dev, wins.data(), sizeof(FsBcWin) * wins.size(), hipMemcpyHostToDevice));
  // LU ranges: a box that reaches a wall face of the zone takes the ghost layer there (BC/BCzone.cpp:99-127)
  bb.lu_ranges.clear();
  for (size_t b = 0; b + 6 <= bx.size(); b += 6)
    for (int q = 0; q < 6; q++) bb.lu_ranges.push_back(bx[b + q] == ends[q] ? zn.lu[q] : bx[b + q]);
  if (bb.lu_ranges.size() > bb.lu_cap) {
    (void)hipFree(bb.lu_dev);
    bb.lu_dev = nullptr; bb.lu_cap = 0;
    HIPCHK(ctx, hipMalloc(&bb.lu_dev, sizeof(int) * bb.lu_ranges.size()));
    bb.lu_cap = bb.lu_ranges.size();
  }
  if (!bb.lu_ranges.empty())
    HIPCHK(ctx, hipMemcpy(bb.lu_dev, bb.lu_ranges.data(), sizeof(int) * bb.lu_ranges.size(), hipMemcpyHostToDevice));
  // tiles (Cartesian kernel) or patches (curvilinear kernel) that hold cells of a box: the stage of a box launches those
  // alone; the Cartesian Spalart-Allmaras kernel takes no list and runs its whole grid, cells outside the box masked
  bb.tile_off.clear(); bb.tile_cnt.clear();
  if (!(z.iflow == 3 && z.ityp != 0)) {
    const FsTiling t = z.ityp == 0 ? fsk_stage_full_tiling(z) : fsk_stage_dma_tiling(z);
    std::vector<int> tiles;
    for (size_t b = 0; b + 6 <= bx.size(); b += 6) {
      bb.tile_off.push_back((int)tiles.size());
      for (int bk = (bx[b + 4] - 1) / t.kchunk; bk <= (bx[b + 5] - 1) / t.kchunk; bk++)
        for (int bj = (bx[b + 2] - 1) / t.tj; bj <= (bx[b + 3] - 1) / t.tj; bj++)
          for (int bi = (bx[b] - 1) / t.ti; bi <= (bx[b + 1] - 1) / t.ti; bi++) tiles.push_back(bi + t.nti * (bj + t.ntj * bk));
      bb.tile_cnt.push_back((int)tiles.size() - bb.tile_off.back());
    }
    if (tiles.size() > bb.tiles_cap) {
      (void)hipFree(bb.tiles_dev);
      bb.tiles_dev = nullptr; bb.tiles_cap = 0;
      HIPCHK(ctx, hipMalloc(&bb.tiles_dev, sizeof(int) * tiles.size()));
      bb.tiles_cap = tiles.size();
    }
    if (!tiles.empty()) HIPCHK(ctx, hipMemcpy(bb.tiles_dev, tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice));
  }
  return FS_OK;
}

// State of the sub-zones as the reference's warmup leaves it (Fast/FastS/FastS/PyTree.py:391-394: souszones_list for
// every sub-iteration with no list yet: counters reset, then the boxes of the initial targets)
static int local_setup(fs_ctx *ctx)
{
  const int nssiter = ctx->zones[0].z.nssiter;
  ctx->iskip_lu.assign(2 * (size_t)nssiter, 0);
  fsz_skip_lu(nssiter, nssiter, ctx->iskip_lu.data());       // init_ssiter_bloc.for:50-53
  fsz_skip_lu(nssiter - 1, nssiter, ctx->iskip_lu.data());
  for (auto &zn : ctx->zones) {
    const FsZone &z = zn.z;
    const int ijkv[3] = {z.iv, z.jv, z.kv};
    const int ss[3] = {zn.pint[SIZE_SSDOM], zn.pint[SIZE_SSDOM + 1], zn.pint[SIZE_SSDOM + 2]};
    fsz_init(zn.sub, ijkv, ss, nssiter, z.neq, z.ifick != 0);
    fsz_reset(zn.sub);
    zn.box_bcs.assign((size_t)nssiter, Zone::BoxBcs());
    HIPCHK(ctx, hipMalloc(&zn.blk_res, sizeof(double) * 12 * (size_t)zn.sub.n * nssiter));
    for (int nstep = 1; nstep <= nssiter; nstep++) {
      fsz_evaluate(zn.sub, nstep, ctx->iskip_lu.data());
      int rc = local_bc_tables(ctx, zn, nstep);
      if (rc) return rc;
    }
  }
  return FS_OK;
}

// The block norms of the sub-iterations of a finished verification step come back from the device and go through the
// bookkeeping of cprdu3s1, in the order the sub-iterations ran (residus_navier.h:12-43; it_bloc: Compute/ssor.cpp:14-20)
static int local_finalize(fs_ctx *ctx)
{
  bool any = false;
  for (auto &zn : ctx->zones) any = any || !zn.pending.empty();
  if (!any) return FS_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto &zn : ctx->zones) {
    if (zn.pending.empty()) continue;
    const size_t per = 12 * (size_t)zn.sub.n;
    std::vector<double> norms(per * zn.pending.size());
    HIPCHK(ctx, hipMemcpy(norms.data(), zn.blk_res, sizeof(double) * norms.size(), hipMemcpyDeviceToHost));
    for (size_t q = 0; q < zn.pending.size(); q++) {
      const Zone::PendingNorms &pn = zn.pending[q];
      zn.sub.it_bloc += 1;
      fsz_residual(zn.sub, pn.nstep, zn.preal[EPSI_NEWTON], norms.data() + per * q, pn.no_lu.data(), pn.nis);
    }
    zn.pending.clear();
    // norms of the zone per slot: the blocks added up in their order (DISPLAY/display_ss_iteration.cpp:112-136)
    const int nss = zn.sub.nssiter, neq = zn.sub.neq;
    zn.res_host.assign(12 * (size_t)zn.sub.it_bloc, 0.);
    for (int it = 0; it < zn.sub.it_bloc; it++)
      for (int ne = 0; ne < neq; ne++)
        for (int b = 0; b < zn.sub.n; b++) {
          const size_t base = (size_t)it + (size_t)nss * ((size_t)ne + (size_t)neq * (2 * (size_t)b));
          zn.res_host[12 * (size_t)it + ne] += zn.sub.rdm[base];
          zn.res_host[12 * (size_t)it + 6 + ne] = std::max(zn.res_host[12 * (size_t)it + 6 + ne], zn.sub.rdm[base + (size_t)nss * neq]);
        }
    zn.res_slots = zn.sub.it_bloc;
    zn.have_res = zn.sub.it_bloc > 0;
  }
  return FS_OK;
}

// souszones_list before sub-iteration nstep of time step nitrun (Fast/FastC/FastC/Compute/souszones_list.cpp:161-196 with
// lssiter_loc = 1, Fast/FastS/FastS/Compute/computePT.cpp:551-577): on a verification step the counters restart at the
// first sub-iteration; on the step after one (and on the first step) the boxes of this sub-iteration are evaluated
// again from the targets.  Once per (nitrun, nstep).
static int local_prepare(fs_ctx *ctx, int nitrun, int nstep)
{
  if (ctx->local_nitrun == nitrun && ctx->local_nstep == nstep) return FS_OK;
  ctx->local_nitrun = nitrun; ctx->local_nstep = nstep;
  const int kmod = ctx->modulo_verif;
  const bool verif = nitrun % kmod == 0, eval = (nitrun - 1) % kmod == 0;
  if (nstep == 1 && (verif || eval)) {
    int rc = local_finalize(ctx);
    if (rc) return rc;
  }
  if (verif && nstep == 1)
    for (auto &zn : ctx->zones) fsz_reset(zn.sub);
  if (eval) {
    // the tables of this sub-iteration may still be read by the step before
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &zn : ctx->zones) {
      fsz_evaluate(zn.sub, nstep, ctx->iskip_lu.data());
      int rc = local_bc_tables(ctx, zn, nstep);
      if (rc) return rc;
    }
  }
  return FS_OK;
}

// part < 0: the whole stage.  part 0: viscosity + the shell tiles.  part 1: the remaining tiles
// (+ residual norms).  Zones that are not split (no joins to other contexts, curvilinear zones, implicit
// scheme) run in one piece: in part 0 if other contexts wait for their cells, else in part 1.
static int stage_impl(fs_ctx *ctx, int nitrun, int nstep, int part, const char *who)
{
  int rc = ensure_ready(ctx, who);
  if (rc) return rc;
  std::vector<hipEvent_t> lu_joins;
  int n_lu = 0;
  if (nstep < 1 || nstep > ctx->zones[0].z.nssiter) return fail(ctx, FS_ERR_ARG, std::string(who) + ": sub-iteration out of range");
  if (ctx->local_newton) {
    rc = local_prepare(ctx, nitrun, nstep);
    if (rc) return rc;
  }
  for (auto &zn : ctx->zones) {
    const FsZone &z = zn.z;
    if (nstep < 1 || nstep > z.nssiter) return fail(ctx, FS_ERR_ARG, std::string(who) + ": sub-iteration out of range");
    // the last slot of an implicit step is the reference's verification pass: skipped on ordinary steps
    // (Fast/FastS/FastS/Compute/computePT.cpp:579-585); on a verification step it evaluates the Newton residual of
    // the final iterate and its norms, and advances nothing (lexit_lu, Fast/FastC/FastC/Compute/souszones_list.cpp:81-83)
    // with local Newton sub-iterations the first time step is no verification step of its own
    // (souszones_list.cpp:166-196: the second pass alone runs, and leaves lssiter_verif = 0)
    const bool verif = ctx->local_newton ? nitrun % ctx->modulo_verif == 0
                                         : (nitrun % ctx->modulo_verif == 0 || nitrun == 1 || ctx->modulo_verif == 1);
    const bool slot = ctx->implicit && nstep == z.nssiter;
    if (slot && !verif) continue;
    // the boxes this zone works on at this sub-iteration (one task per box, rhs.cpp:9-58); none: the zone sits it out
    const int whole_box[6] = {1, z.iv, 1, z.jv, 1, z.kv};
    const int32_t *boxes = nullptr;
    int nbox = 1;
    if (ctx->local_newton) {
      boxes = zn.sub.boxes[(size_t)nstep - 1].data();
      nbox = (int)(zn.sub.boxes[(size_t)nstep - 1].size() / 6);
      if (nbox == 0) continue;
      if (nstep == 1 && (nbox != 1 || memcmp(boxes, whole_box, sizeof whole_box) != 0))
        return fail(ctx, FS_ERR_STATE, std::string(who) + ": the first sub-iteration must cover the whole zone");
    }
    int in, out;
    stage_roles(ctx, nstep, in, out);
    double coefH, coefW;
    rk_coefs(z.rk, nstep, coefH, coefW);
    const bool split = part >= 0 && zn.n_shell > 0;
    // a zone that cannot be split runs in one piece: in part 0 when messages wait for its cells, else in part 1.
    // A zone whose ghost cells are filled from other contexts also runs in part 0 when its stage reads the level the
    // messages are unpacked into (Newton sub-iterations after the first: input = output = P1), so that it never
    // overlaps fs_halo_unpack on the other stream.
    const bool early = zn.has_sends || (zn.has_recvs && in == out);
    if (part >= 0 && !split && part != (early ? 0 : 1)) continue;
    const bool opens = part <= 0 || !split;      // this call starts the stage of this zone
    const bool closes = part != 0 || !split;     // this call ends it
    const bool dtloc = zn.pint[DTLOC] == 1;
    FsBdf bdf = {zn.dev[0], zn.dev[1], -1.5, 2., -0.5, nstep == 1, dtloc ? zn.coe : nullptr, {1, z.iv, 1, z.jv, 1, z.kv}};
    if (ctx->first_it == 0 || dtloc) { bdf.c1 = -1.; bdf.c2 = 1.; bdf.c3 = 0.; }   // core3as2.for:37-45
    const FsBdf *pbdf = ctx->implicit ? &bdf : nullptr;
    if (opens && nstep == 1) {
      if (z.iflow == 2) fsk_invist(z, (ctx->implicit || dtloc) ? z.ific : 1, zn.dev[in], zn.xmut, ctx->stream);
      if (z.iflow == 3) fsk_vispalart(z, zn.dev[in], zn.xmut, ctx->stream);
      if (!ctx->implicit && dtloc) {
        // tstb3_expli.for + dtloc.for: the CFL-limited step of every cell = coe(:,1) of the implicit coefficients
        const double *r = zn.preal.data();
        fsk_coe_implicit(z, 0, 1, r[CFL], r[GAMMA], r[PRANDT], r[CVINF] * (r[GAMMA] - 1.), zn.dev[in], zn.xmut, zn.coe,
                         ctx->stream);
      }
      if (ctx->implicit) {
        const double *r = zn.preal.data();
        fsk_coe_implicit(z, ctx->first_it, dtloc, r[CFL], r[GAMMA], r[PRANDT], r[CVINF] * (r[GAMMA] - 1.), zn.dev[in],
                         zn.xmut, zn.coe, ctx->stream);
        if (z.iflow == 3) {
          if (z.ityp == 0) fsk_sa_coe6_full(z, zn.dev[in], zn.dist, zn.coe, ctx->stream);
          else fsk_sa_coe6(z, zn.dev[in], zn.dist, zn.coe, ctx->stream);
        }
      }
    }
    if (ctx->timing && opens) {
      if (!zn.ev_part) HIPCHK(ctx, hipEventCreate(&zn.ev_part));
      HIPCHK(ctx, hipEventRecord(zn.ev_part, ctx->stream));
    }
    int krc = 0;
    // roe + walls: the stage kernel runs in Runge-Kutta form and leaves the balance in drodm; the wall rows are
    // corrected and the end of the stage redone below.  Implicit scheme: balance from zero at every
    // sub-iteration, the (unused) update goes to the spare level.
    const bool deferred = !zn.roe_walls.empty();
    const FsBdf *kbdf = deferred ? nullptr : pbdf;
    const int kfirst = deferred && ctx->implicit ? 1 : (nstep == 1);
    double *kout = deferred && ctx->implicit ? zn.scratch : zn.dev[out];
    // the balance of the last Runge-Kutta stage is dead: the next step starts from zero (init_rhs.for), the norms
    // are those of the first stage; kept when a later kernel of this stage reads it (wall-flux correction, local
    // time step) and for a one-stage scheme, whose only stage is also the first
    const int keep = ctx->implicit || deferred || dtloc || nstep < z.nssiter || nstep == 1;
    for (int b = 0; b < nbox && !krc; b++) {
    if (boxes) for (int q = 0; q < 6; q++) bdf.box[q] = boxes[6 * b + q];   // cells outside the box keep their residual
    // a box smaller than the zone: only the tiles that hold some of its cells are launched
    const bool partial_box = boxes && memcmp(boxes + 6 * b, whole_box, sizeof whole_box) != 0 &&
                             !zn.box_bcs[(size_t)nstep - 1].tile_cnt.empty();
    const int *box_tiles = partial_box ? zn.box_bcs[(size_t)nstep - 1].tiles_dev + zn.box_bcs[(size_t)nstep - 1].tile_off[(size_t)b] : nullptr;
    const int box_ntiles = partial_box ? zn.box_bcs[(size_t)nstep - 1].tile_cnt[(size_t)b] : 0;
    if (z.slope == 6) {
      // o3sc: the shock sensor of the state this sub-iteration starts from (src_term.for:246-255), then kernels of its own
      fsk_sciacovelli(z, zn.dev[in], zn.wsc, ctx->stream);
      krc = z.ityp == 0 ? fsk_stage_o3sc_full(z, coefH, coefW, kfirst, zn.dev[in], kout, zn.drodm, zn.xmut,
                                              (unsigned char *)zn.wig, zn.wsc, zn.wflux, kbdf, keep, ctx->stream)
                        : fsk_stage_o3sc_cart(z, coefH, coefW, kfirst, zn.dev[in], kout, zn.drodm, zn.xmut,
                                              (unsigned char *)zn.wig, zn.wsc, kbdf, keep, ctx->stream);
    } else if (z.iflow == 3 && z.ityp != 0)
      krc = fsk_stage_sa(z, coefH, coefW, kfirst, zn.dev[in], kout, zn.drodm, zn.xmut, zn.dist,
                         (unsigned char *)zn.wig, kbdf, keep, ctx->stream);
    else if (z.ityp == 0) {
      const int *blist = !split ? nullptr : (part == 0 ? zn.shell_dev : zn.inner_dev);
      int nlist = !split ? 0 : (part == 0 ? zn.n_shell : zn.n_inner);
      if (partial_box) { blist = box_tiles; nlist = box_ntiles; }
      if (!split || nlist > 0)
        krc = fsk_stage_full(z, coefH, coefW, kfirst, zn.dev[in], kout, zn.drodm, zn.xmut,
                             (unsigned char *)zn.wig, kbdf, zn.dist, keep, blist, nlist, ctx->stream);
    } else {
      const int *blist = !split ? nullptr : (part == 0 ? zn.shell_dev : zn.inner_dev);
      int nlist = !split ? 0 : (part == 0 ? zn.n_shell : zn.n_inner);
      if (partial_box) { blist = box_tiles; nlist = box_ntiles; }
      if (!split || nlist > 0)
        krc = fsk_stage_dma(z, coefH, coefW, kfirst, zn.dev[in], kout, zn.drodm, zn.xmut,
                            (unsigned char *)zn.wig, blist, nlist, kbdf, keep, ctx->stream);
    }
    if (boxes && verif && !krc && memcmp(boxes + 6 * b, whole_box, sizeof whole_box) != 0) {
      // What the reference's loops leave in the balance of the cells around the box: the source of the 6th equation,
      // then the fluxes of the faces of the box.  Only the norms of a verification step ever read those cells.
      const int box[6] = {boxes[6 * b], boxes[6 * b + 1], boxes[6 * b + 2], boxes[6 * b + 3], boxes[6 * b + 4], boxes[6 * b + 5]};
      if (z.iflow == 3) {
        if (z.ityp == 0) fsk_sa_ring_source_full(z, box, zn.dev[in], zn.dist, zn.drodm, ctx->stream);
        else fsk_sa_ring_source(z, box, zn.dev[in], zn.dist, zn.drodm, ctx->stream);
      }
      if (z.ityp == 0) fsk_box_faces_full(z, box, zn.dev[in], zn.xmut, (unsigned char *)zn.wig, zn.drodm, ctx->stream);
      else if (z.iflow == 3) fsk_box_faces_sa(z, box, zn.dev[in], zn.xmut, (unsigned char *)zn.wig, zn.drodm, ctx->stream);
      else fsk_box_faces_cart(z, box, zn.dev[in], zn.xmut, (const unsigned char *)zn.wig, zn.drodm, ctx->stream);
    }
    }
    if (krc)
      return fail(ctx, FS_ERR_UNSUPPORTED, std::string(who) + ": no kernel for this scheme / model");
    if (deferred) {
      for (auto &rw : zn.roe_walls) fsk_roe_wall(z, rw.idir, rw.lo, rw.hi, rw.tag, zn.dev[in], zn.drodm, ctx->stream);
      if (ctx->implicit) fsk_finish_bdf(z, zn.dev[in], zn.drodm, bdf, ctx->stream);
      else if (!dtloc)
        for (auto &rw : zn.roe_walls)
          fsk_finish_wall_rows(z, rw.idir, rw.lo, rw.hi, coefW, zn.dev[in], zn.dev[out], zn.drodm, ctx->stream);
    }
    // explicit scheme with a local time step: the stage kernel has left the balance in drodm (and an update with
    // the global step that is discarded); every cell is updated with its own coe(:,1)
    if (dtloc && !ctx->implicit)
      fsk_finish_rk_local(z, coefW, zn.coe, zn.dev[in], zn.dev[out], zn.drodm, ctx->stream);
    if (!closes) continue;
    if (ctx->timing) {   // one timed interval per stage, from its first launch to its last
      if (ctx->ev_used == ctx->ev.size()) {
        hipEvent_t a, b;
        HIPCHK(ctx, hipEventCreate(&a));
        HIPCHK(ctx, hipEventCreate(&b));
        ctx->ev.emplace_back(a, b);
      }
      std::swap(ctx->ev[ctx->ev_used].first, zn.ev_part);
      HIPCHK(ctx, hipEventRecord(ctx->ev[ctx->ev_used].second, ctx->stream));
      ctx->ev_used++;
    }
    // residual norms (cprdu3s1.for): every Newton sub-iteration of a verification step; explicit scheme: the
    // first-stage increment
    if (verif && ctx->local_newton) {
      // norms per block; the bookkeeping of cprdu3s1 follows on the host once the step is over, with the block list
      // as it stands now
      fsk_residual_blocks(z, zn.sub.nblk, zn.sub.size, zn.drodm, zn.blk_res + 12 * (size_t)zn.sub.n * zn.pending.size(),
                          ctx->stream);
      zn.pending.push_back({nstep, zn.sub.nisdom_residu[(size_t)nstep - 1], zn.sub.no_lu});
    } else if (verif && (ctx->implicit || nstep == 1)) {
      fsk_residual(z, ctx->implicit ? 0 : 1, zn.drodm, zn.res_part, zn.res_out + 12 * (nstep - 1), ctx->stream,
                   (!ctx->implicit && dtloc) ? zn.coe : nullptr);
      zn.have_res = true;
      zn.res_slots = nstep;
    }
    if (nstep == 1 && verif) {
      // local CFL number of the state the step starts from (cptst3.for:114-120: first sub-iteration of a
      // verification step), with the viscosity this stage has just evaluated
      const double *r = zn.preal.data();
      fsk_cfl(z, r[GAMMA] / r[PRANDT], r[GAMMA] * (r[CVINF] * (r[GAMMA] - 1.)), zn.dev[in], zn.xmut, zn.res_part,
              zn.res_out + 12 * z.nssiter, ctx->stream);
      zn.have_cfl = true;
    }
    // selective frequency damping, once per time step in its last slot that runs (lhs.cpp:181-185): on a verification
    // step after the residual of the verification slot (no join or BC follows, gsdr3.cpp:343,417), else after the LU
    // solve of the last Newton sub-iteration, below
    const bool sfd = zn.pint[SFD] == 1 && zn.rof;
    const int sfd_pull = nitrun >= zn.pint[SFD_INIT_IT];
    if (sfd && slot)
      fsk_sfd(z, zn.preal[35], zn.preal[36], sfd_pull, zn.dev[out], zn.rof, zn.coe, ctx->stream);
    // the explicit loop runs the same condition (no verification slot: nitcfg = nssiter - 1): the state is filtered
    // after the last stage but one, before its joins and BCs
    if (sfd && !ctx->implicit && nstep == z.nssiter - 1)
      fsk_sfd(z, zn.preal[35], zn.preal[36], sfd_pull, zn.dev[out], zn.rof, dtloc ? zn.coe : nullptr, ctx->stream);
    if (ctx->implicit && !slot) {   // LU-SGS solve of the Newton system, new iterate into P1 (lhs.cpp:53-68,135-161)
      if (!zn.rhs_face.empty()) {   // wall faces: BCs on the residual, two ordered passes, then edges/corners
        fsk_bc_windows(z, (int)zn.rhs_face.size(), zn.rhs_face_dev, bc_kinds(zn.rhs_face), zn.rhs_max_cells, zn.drodm, ctx->stream);
        fsk_bc_windows(z, (int)zn.rhs_bc.size(), zn.rhs_bc_dev, bc_kinds(zn.rhs_bc), zn.rhs_max_cells, zn.drodm, ctx->stream);
        fsk_bc_edges(z, zn.drodm, ctx->stream);
      }
      hipStream_t ls = ctx->stream;
      if (ctx->lu_nstreams < 0) {
        const char *ev = getenv("FASTS_HIP_LU_STREAMS");
        ctx->lu_nstreams = ev ? atoi(ev) : 4;
        if (ctx->lu_nstreams > 8) ctx->lu_nstreams = 8;
        if (ctx->lu_nstreams < 2 || ctx->zones.size() < 2) ctx->lu_nstreams = 0;
        for (int q = 0; q < ctx->lu_nstreams; q++) {
          hipStream_t st;
          HIPCHK(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
          ctx->lu_streams.push_back(st);
        }
      }
      if (ctx->lu_nstreams > 0) {
        if (!zn.ev_lu_fork) {
          HIPCHK(ctx, hipEventCreateWithFlags(&zn.ev_lu_fork, hipEventDisableTiming));
          HIPCHK(ctx, hipEventCreateWithFlags(&zn.ev_lu_join, hipEventDisableTiming));
        }
        ls = ctx->lu_streams[n_lu % ctx->lu_nstreams];
        HIPCHK(ctx, hipEventRecord(zn.ev_lu_fork, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ls, zn.ev_lu_fork, 0));
      }
      int lrc = 0;
      const int side_by_side = ctx->lu_nstreams > 0 ? std::min<int>(ctx->lu_nstreams, (int)ctx->zones.size()) : 1;
      if (ctx->local_newton) {
        // one sweep pair over the boxes, side by side like the blocks of the reference's threads: neighbour terms across
        // the borders of a box are left out (LU/invlu_l.for:90-215), the Newton update touches the boxes alone
        Zone::BoxBcs &bb = zn.box_bcs[(size_t)nstep - 1];
        if (!bb.flow) bb.flow = fsk_luflow_new();
        fsk_luflow_concurrency(bb.flow, side_by_side);
        lrc = fsk_lusgs(z, nbox > 1 ? zn.lu : bb.lu_ranges.data(), nbox, bb.lu_ranges.data(), bb.lu_dev, zn.preal[GAMMA],
                        zn.dev[in], zn.dev[out], zn.scratch, zn.drodm, zn.coe, ls, true, bb.flow);
      } else {
      const int sdm[6] = {1, z.iv, 1, z.jv, 1, z.kv};
      if (!zn.lu_flow) zn.lu_flow = fsk_luflow_new();
      fsk_luflow_concurrency(zn.lu_flow, side_by_side);
      // lu_match = 1: the sweeps of invlu_overlap_{l,u}.for (with nb_relax > 1, 5 equations: the LU-SSOR sweeps whatever
      // lu_match is, LU/invlu.for:195-340).  The reference also copies the Newton residual into the
      // ghost cells of match joins first (lhs.cpp:1-15); no sweep reads them (the range grows into ghost cells only
      // over walls, whose BC then overwrites them, and between the thread blocks of a zone): nothing to do here
      lrc = (zn.pint[NB_RELAX] > 1 || zn.pint[LU_MATCH] != 0)
                          ? fsk_lussor(z, zn.lu, sdm, (int)(zn.lu_blocks.size() / 6), zn.lu_blocks.data(), zn.lu_blocks_dev, zn.pint[NB_RELAX],
                                       zn.pint[LU_MATCH] != 0 && zn.pint[NB_RELAX] == 1, zn.preal[GAMMA], zn.dev[in],
                                       zn.dev[out], zn.scratch, zn.drodm, zn.ssor_work, zn.coe, ls, zn.lu_flow)
                          : fsk_lusgs(z, zn.lu, (int)(zn.lu_blocks.size() / 6), zn.lu_blocks.data(), zn.lu_blocks_dev, zn.preal[GAMMA], zn.dev[in],
                                      zn.dev[out], zn.scratch, zn.drodm, zn.coe, ls, false, zn.lu_flow);
      }
      if (sfd && !verif && nstep == z.nssiter - 1)
        fsk_sfd(z, zn.preal[35], zn.preal[36], sfd_pull, zn.dev[out], zn.rof, zn.coe, ls);
      if (ctx->lu_nstreams > 0) {
        HIPCHK(ctx, hipEventRecord(zn.ev_lu_join, ls));
        lu_joins.push_back(zn.ev_lu_join);
        n_lu++;
      }
      if (lrc) {
        for (hipEvent_t e : lu_joins) (void)hipStreamWaitEvent(ctx->stream, e, 0);
        return fail(ctx, FS_ERR_STATE, std::string(who) + ": LU-SGS scratch level missing");
      }
    }
  }
  for (hipEvent_t e : lu_joins) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, e, 0));
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_stage_part(fs_ctx *ctx, int nitrun, int nstep, int part)
{
  if (part != 0 && part != 1) return ctx ? fail(ctx, FS_ERR_ARG, "fs_stage_part: part must be 0 or 1") : FS_ERR_ARG;
  return stage_impl(ctx, nitrun, nstep, part, "fs_stage_part");
}

extern "C" int fs_set_comm_stream(fs_ctx *ctx, void *hip_stream)
{
  if (!ctx) return FS_ERR_ARG;
  // with the library's own transport the pack, the messages and the unpack share the communicator's stream: another
  // stream here would let a message leave before its buffer is packed
  if (ctx->comm) return fail(ctx, FS_ERR_STATE, "fs_set_comm_stream: the context has a communicator (fs_comm_init), its stream is fixed");
  ctx->comm_stream = (hipStream_t)hip_stream;
  return FS_OK;
}

extern "C" int fs_stage(fs_ctx *ctx, int nitrun, int nstep)
{
  int rc = stage_impl(ctx, nitrun, nstep, -1, "fs_stage");
  if (rc) return rc;
  return FS_OK;
}

static int local_joins_on(fs_ctx *ctx, int role_of_zone(const void *, int), const void *arg)
{
  for (auto &j : ctx->joins) {
    if (j.donor_zone < 0) continue;
    Zone &zr = ctx->zones[j.rcv_zone], &zd = ctx->zones[j.donor_zone];
    int rr = role_of_zone(arg, j.rcv_zone), rd = role_of_zone(arg, j.donor_zone);
    fsk_join_copy(j.npts, zr.z.neq, zr.z.nd, zd.z.nd, j.rcv_dev, j.dnr_dev, zr.dev[rr], zd.dev[rd], ctx->stream);
  }
  return FS_OK;
}

struct NstepArg { const fs_ctx *ctx; int nstep; };

static int role_from_nstep(const void *arg, int)
{
  const NstepArg *a = (const NstepArg *)arg;
  int in, out;
  stage_roles(a->ctx, a->nstep, in, out);
  return out;
}

static int role_fixed(const void *arg, int) { return *(const int *)arg; }

extern "C" int fs_local_joins(fs_ctx *ctx, int nstep)
{
  int rc = ensure_ready(ctx, "fs_local_joins");
  if (rc) return rc;
  NstepArg a = {ctx, nstep};
  local_joins_on(ctx, role_from_nstep, &a);
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

// nstep >= 1 with local Newton sub-iterations: the BCs of the boxes of that sub-iteration (bcs.cpp:14-41), their edges
// and corners box by box (correct_coins with the range of the task)
static void bcs_on_role(fs_ctx *ctx, int role, int nstep = 0)
{
  for (auto &zn : ctx->zones) {
    if (ctx->local_newton && nstep >= 1) {
      const Zone::BoxBcs &bb = zn.box_bcs[(size_t)nstep - 1];
      fsk_bc_windows(zn.z, bb.n, bb.dev, bb.kinds, bb.max_cells, zn.dev[role], ctx->stream);
      const std::vector<int32_t> &bx = zn.sub.boxes[(size_t)nstep - 1];
      for (size_t b = 0; b + 6 <= bx.size(); b += 6) {
        const int box[6] = {bx[b], bx[b + 1], bx[b + 2], bx[b + 3], bx[b + 4], bx[b + 5]};
        fsk_bc_edges(zn.z, zn.dev[role], ctx->stream, box);
      }
      continue;
    }
    fsk_bc_windows(zn.z, (int)zn.bc.size(), zn.bc_dev, bc_kinds(zn.bc), zn.bc_max_cells, zn.dev[role], ctx->stream);
    fsk_bc_edges(zn.z, zn.dev[role], ctx->stream);
  }
}

extern "C" int fs_bcs(fs_ctx *ctx, int nstep)
{
  int rc = ensure_ready(ctx, "fs_bcs");
  if (rc) return rc;
  int in, out;
  stage_roles(ctx, nstep, in, out);
  bcs_on_role(ctx, out, nstep);
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_rotate(fs_ctx *ctx)
{
  if (!ctx) return FS_ERR_ARG;
  for (auto &zn : ctx->zones) {  // computePT.cpp:673-677
    double *d = zn.dev[1]; zn.dev[1] = zn.dev[0]; zn.dev[0] = zn.dev[2]; zn.dev[2] = d;
    double *h = zn.host[1]; zn.host[1] = zn.host[0]; zn.host[0] = zn.host[2]; zn.host[2] = h;
  }
  ctx->first_it = 1;   // Fast/FastS/FastS/PyTree.py:235
  return FS_OK;
}

extern "C" int fs_halo_count(fs_ctx *ctx, int peer_tag, int is_send)
{
  if (!ctx) return FS_ERR_ARG;
  long long n = 0;
  if (is_send) {
    for (auto &s : ctx->sends) if (s.peer_tag == peer_tag) n += (long long)s.npts * ctx->zones[s.donor_zone].z.neq;
  } else {
    for (auto &j : ctx->joins) if (j.donor_zone < 0 && j.peer_tag == peer_tag) n += (long long)j.npts * ctx->zones[j.rcv_zone].z.neq;
  }
  return (int)n;
}

extern "C" int fs_halo_pack(fs_ctx *ctx, int nstep, int peer_tag, double *dev_buf, int n_doubles)
{
  int rc = ensure_ready(ctx, "fs_halo_pack");
  if (rc) return rc;
  if (n_doubles != fs_halo_count(ctx, peer_tag, 1)) return fail(ctx, FS_ERR_ARG, "fs_halo_pack: buffer size mismatch");
  int in, out;
  if (nstep <= 0) out = level_role(-nstep); else stage_roles(ctx, nstep, in, out);
  if (out < 0) return fail(ctx, FS_ERR_ARG, "fs_halo_pack: bad level");
  long long off = 0;
  for (auto &s : ctx->sends) {
    if (s.peer_tag != peer_tag) continue;
    Zone &zd = ctx->zones[s.donor_zone];
    fsk_halo_pack(s.npts, zd.z.neq, zd.z.nd, s.pts_dev, zd.dev[out], dev_buf + off,
                  ctx->comm_stream ? ctx->comm_stream : ctx->stream);
    off += (long long)s.npts * zd.z.neq;
  }
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_halo_unpack(fs_ctx *ctx, int nstep, int peer_tag, const double *dev_buf, int n_doubles)
{
  int rc = ensure_ready(ctx, "fs_halo_unpack");
  if (rc) return rc;
  if (n_doubles != fs_halo_count(ctx, peer_tag, 0)) return fail(ctx, FS_ERR_ARG, "fs_halo_unpack: buffer size mismatch");
  int in, out;
  if (nstep <= 0) out = level_role(-nstep); else stage_roles(ctx, nstep, in, out);
  if (out < 0) return fail(ctx, FS_ERR_ARG, "fs_halo_unpack: bad level");
  long long off = 0;
  for (auto &j : ctx->joins) {
    if (j.donor_zone >= 0 || j.peer_tag != peer_tag) continue;
    Zone &zr = ctx->zones[j.rcv_zone];
    fsk_halo_unpack(j.npts, zr.z.neq, zr.z.nd, j.rcv_dev, zr.dev[out], dev_buf + off,
                    ctx->comm_stream ? ctx->comm_stream : ctx->stream);
    off += (long long)j.npts * zr.z.neq;
  }
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_compute(fs_ctx *ctx, int nitrun, int nstep_deb, int nstep_fin, int nit_c, double *t_trans)
{
  int rc = ensure_ready(ctx, "fs_compute");
  if (rc) return rc;
  if (t_trans) *t_trans = 0.;
  const int nssiter = ctx->zones[0].z.nssiter;
  if (nstep_deb < 1 || nstep_fin > nssiter || nstep_deb > nstep_fin || nit_c < 1)
    return fail(ctx, FS_ERR_ARG, "fs_compute: bad sub-iteration range");
  bool remote = false;
  for (auto &j : ctx->joins) if (j.donor_zone < 0) remote = true;
  for (auto &sd : ctx->sends) { (void)sd; remote = true; }
  if (remote && !fs_comm_active(ctx))
    return fail(ctx, FS_ERR_STATE, "fs_compute: joins to other contexts need fs_comm_init / fs_comm_setup, or the split calls "
                "(fs_stage / fs_halo_* / fs_bcs) with a transport of the caller's");
  static const bool overlap = !(getenv("FASTS_HIP_OVERLAP") && strcmp(getenv("FASTS_HIP_OVERLAP"), "0") == 0);
  for (int it = 0; it < nit_c; it++) {
    const int nitrun_loc = nitrun * nit_c + it;  // computePT.cpp:543
    for (int nstep = nstep_deb; nstep <= nstep_fin; nstep++) {
      const bool slot = ctx->implicit && nstep == nssiter;
      if (remote && !slot) {
        // stage in two parts: the tiles that feed halo messages first; pack -> RCCL send/recv -> unpack run on the
        // comm stream while the rest of the stage runs here (the reference transfers after the whole stage,
        // Fast/FastS/FastS/Compute/gsdr3.cpp:251-349)
        if (overlap) {
          rc = stage_impl(ctx, nitrun_loc, nstep, 0, "fs_compute");
          if (rc) return rc;
          rc = fs_comm_exchange_begin(ctx, nstep);  // ordered after part 0 by an event, returns at once
          if (rc) return rc;
          rc = stage_impl(ctx, nitrun_loc, nstep, 1, "fs_compute");
        } else {
          rc = stage_impl(ctx, nitrun_loc, nstep, -1, "fs_compute");
          if (rc) return rc;
          rc = fs_comm_exchange_begin(ctx, nstep);
        }
        if (rc) return rc;
        rc = fs_comm_exchange_end(ctx);             // the context's stream waits for the unpack
      } else
        rc = fs_stage(ctx, nitrun_loc, nstep);
      if (rc) return rc;
      if (slot) continue;   // verification slot: no update, no ghost fill
      NstepArg a = {ctx, nstep};
      local_joins_on(ctx, role_from_nstep, &a);
      int in, out;
      stage_roles(ctx, nstep, in, out);
      bcs_on_role(ctx, out, nstep);
    }
    // explicit: the result of the last sub-iteration sits in P1 only when their number is odd
    // (Fast/FastS/FastS/PyTree.py:233: switchPointers__(zones, case, nitmax%2+2)); implicit: always (:232)
    if (nstep_fin == nssiter && (ctx->implicit || nssiter % 2 == 1)) fs_rotate(ctx);
    else if (nstep_fin == nssiter) ctx->first_it = 1;
  }
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_fill_ghostcells(fs_ctx *ctx, int level)
{
  int rc = ensure_ready(ctx, "fs_fill_ghostcells");
  if (rc) return rc;
  int role = level_role(level);
  if (role < 0) return fail(ctx, FS_ERR_ARG, "fs_fill_ghostcells: bad level");
  local_joins_on(ctx, role_fixed, &role);
  bcs_on_role(ctx, role);
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_apply_bc(fs_ctx *ctx, int nstep, int level)
{
  (void)nstep;
  int rc = ensure_ready(ctx, "fs_apply_bc");
  if (rc) return rc;
  int role = level_role(level);
  if (role < 0) return fail(ctx, FS_ERR_ARG, "fs_apply_bc: bad level");
  bcs_on_role(ctx, role);
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_cfl(fs_ctx *ctx, int zone, double *cfl)
{
  int rc = ensure_ready(ctx, "fs_cfl");
  if (rc) return rc;
  if (zone < 0 || zone >= (int)ctx->zones.size() || !cfl) return fail(ctx, FS_ERR_ARG, "fs_cfl: bad argument");
  Zone &zn = ctx->zones[zone];
  if (!zn.have_cfl) return fail(ctx, FS_ERR_STATE, "fs_cfl: no verification sub-iteration has run yet");
  HIPCHK(ctx, hipMemcpyAsync(cfl, zn.res_out + 12 * zn.z.nssiter, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FS_OK;
}

// L2 / Loo of one slot of the residual table, Fast/FastS/FastS/DISPLAY/display_ss_iteration.cpp:98,137-141 (the
// reciprocal of the cell count is evaluated in single precision there)
static void residual_norms(const Zone &zn, const double *out, int neq, double *l2, double *loo)
{
  const double xpt_1 = 1. / (double)(float)((long long)zn.z.iv * zn.z.jv * zn.z.kv);
  for (int e = 0; e < neq; e++) {
    l2[e] = e < zn.z.neq ? std::sqrt(out[e] * xpt_1) / zn.z.dt : 0.;
    loo[e] = e < zn.z.neq ? out[6 + e] / zn.z.dt : 0.;
  }
}

extern "C" int fs_residuals(fs_ctx *ctx, int zone, double *l2, double *loo, int neq)
{
  int rc = ensure_ready(ctx, "fs_residuals");
  if (rc) return rc;
  if (zone < 0 || zone >= (int)ctx->zones.size() || !l2 || !loo || neq < 5 || neq > 6) return fail(ctx, FS_ERR_ARG, "fs_residuals: bad argument");
  Zone &zn = ctx->zones[zone];
  if (ctx->local_newton) {
    rc = local_finalize(ctx);
    if (rc) return rc;
  }
  if (!zn.have_res) return fail(ctx, FS_ERR_STATE, "fs_residuals: no verification sub-iteration has run yet");
  double out[12];
  if (ctx->local_newton) memcpy(out, zn.res_host.data(), sizeof out);
  else {
    HIPCHK(ctx, hipMemcpyAsync(out, zn.res_out, sizeof out, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  residual_norms(zn, out, neq, l2, loo);
  return FS_OK;
}

extern "C" int fs_residual_history(fs_ctx *ctx, int zone, double *l2, double *loo, int nslot, int neq)
{
  int rc = ensure_ready(ctx, "fs_residual_history");
  if (rc) return rc;
  if (zone < 0 || zone >= (int)ctx->zones.size() || !l2 || !loo || neq < 5 || neq > 6 || nslot < 1)
    return fail(ctx, FS_ERR_ARG, "fs_residual_history: bad argument");
  Zone &zn = ctx->zones[zone];
  if (ctx->local_newton) {
    rc = local_finalize(ctx);
    if (rc) return rc;
  }
  if (!zn.have_res) return fail(ctx, FS_ERR_STATE, "fs_residual_history: no verification step has run yet");
  const int n = zn.res_slots < nslot ? zn.res_slots : nslot;
  std::vector<double> out(12 * (size_t)n);
  if (ctx->local_newton) memcpy(out.data(), zn.res_host.data(), sizeof(double) * out.size());
  else {
    HIPCHK(ctx, hipMemcpyAsync(out.data(), zn.res_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  for (int i = 0; i < n; i++) residual_norms(zn, out.data() + 12 * i, neq, l2 + (size_t)i * neq, loo + (size_t)i * neq);
  return n;
}

// Kinetic energy and enstrophy (Fast/FastS/FastS/POST/computePT_enstrophy.cpp:176-218): sums over the zones of the
// context; the caller divides by the volume (after adding the sums of the other ranks of a multi-GPU job).
static int tgv_sums_on(fs_ctx *ctx, int role, double *dev_out, double *scratch, long long scratch_rows, const char *who)
{
  for (auto &zn : ctx->zones) {
    if (zn.z.ityp != 2 || zn.z.twod) return fail(ctx, FS_ERR_UNSUPPORTED, std::string(who) + ": Cartesian zones only");
    if (scratch && (long long)zn.z.jv * zn.z.kv > scratch_rows)
      return fail(ctx, FS_ERR_ARG, std::string(who) + ": zone has more (j,k) rows than the static scratch buffer holds");
  }
  HIPCHK(ctx, hipMemsetAsync(dev_out, 0, sizeof(double) * 3, ctx->stream));
  for (auto &zn : ctx->zones) fsk_tgv_sums(zn.z, zn.dev[role], scratch ? scratch : zn.res_part, dev_out, ctx->stream);
  HIPCHK(ctx, hipGetLastError());
  return FS_OK;
}

extern "C" int fs_tgv_sums(fs_ctx *ctx, int level, double *sums)
{
  int rc = ensure_ready(ctx, "fs_tgv_sums");
  if (rc) return rc;
  const int role = level_role(level);
  if (role < 0 || !sums) return fail(ctx, FS_ERR_ARG, "fs_tgv_sums: bad argument");
  rc = tgv_sums_on(ctx, role, ctx->diag_out, nullptr, 0, "fs_tgv_sums");
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(sums, ctx->diag_out, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FS_OK;
}

extern "C" int fs_tgv_sums_static(fs_ctx *ctx, int level, double *dev_sums)
{
  int rc = ensure_ready(ctx, "fs_tgv_sums_static");
  if (rc) return rc;
  const int role = level_role(level);
  if (role < 0 || !dev_sums) return fail(ctx, FS_ERR_ARG, "fs_tgv_sums_static: bad argument");
  double *scratch = ctx->device < FS_MAX_DEVICES ? g_static_scratch[ctx->device] : nullptr;
  if (!scratch) return fail(ctx, FS_ERR_STATE, "fs_tgv_sums_static: no static scratch buffer on this device");
  return tgv_sums_on(ctx, role, dev_sums, scratch, FS_STATIC_ROWS, "fs_tgv_sums_static");
}

extern "C" int fs_subzone_list(fs_ctx *ctx, int zone, int nstep, int32_t *boxes, int max_boxes)
{
  int rc = ensure_ready(ctx, "fs_subzone_list");
  if (rc) return rc;
  if (!ctx->local_newton) return fail(ctx, FS_ERR_STATE, "fs_subzone_list: the context runs no local Newton sub-iterations");
  if (zone < 0 || zone >= (int)ctx->zones.size() || !boxes || nstep < 1 || nstep > ctx->zones[zone].z.nssiter)
    return fail(ctx, FS_ERR_ARG, "fs_subzone_list: bad argument");
  const std::vector<int32_t> &bx = ctx->zones[zone].sub.boxes[(size_t)nstep - 1];
  const int nb = (int)(bx.size() / 6);
  if (nb > max_boxes) return fail(ctx, FS_ERR_ARG, "fs_subzone_list: more boxes than the caller has room for");
  if (nb) memcpy(boxes, bx.data(), sizeof(int32_t) * bx.size());
  return nb;
}

extern "C" int fs_subzone_targets(fs_ctx *ctx, int zone, int32_t *it_target, int max_blocks)
{
  int rc = ensure_ready(ctx, "fs_subzone_targets");
  if (rc) return rc;
  if (!ctx->local_newton) return fail(ctx, FS_ERR_STATE, "fs_subzone_targets: the context runs no local Newton sub-iterations");
  if (zone < 0 || zone >= (int)ctx->zones.size() || !it_target || ctx->zones[zone].sub.n > max_blocks)
    return fail(ctx, FS_ERR_ARG, "fs_subzone_targets: bad argument");
  rc = local_finalize(ctx);
  if (rc) return rc;
  const FsSubZones &sz = ctx->zones[zone].sub;
  memcpy(it_target, sz.it_target.data(), sizeof(int32_t) * (size_t)sz.n);
  return sz.n;
}

extern "C" int fs_sync(fs_ctx *ctx)
{
  if (!ctx) return FS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return check_sweeps(ctx, "fs_sync");
}

// Streaming rate of the device this context runs on: triad a = b + s*c over three arrays of n doubles, timed with HIP
// events over `reps` launches after one warm-up launch; *gbs = 24 n reps / time.  Allocates and frees its arrays.
extern "C" int fs_stream_triad(fs_ctx *ctx, long long n_doubles, int reps, double *gbs)
{
  if (!ctx) return FS_ERR_ARG;
  if (!gbs || n_doubles < 1024 || n_doubles % 2 || reps < 1) return fail(ctx, FS_ERR_ARG, "fs_stream_triad: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double *a = nullptr, *b = nullptr, *c = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = FS_OK;
  const size_t nb = sizeof(double) * (size_t)n_doubles;
  if (hipMalloc(&a, nb) != hipSuccess || hipMalloc(&b, nb) != hipSuccess || hipMalloc(&c, nb) != hipSuccess ||
      hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
    rc = fail(ctx, FS_ERR_HIP, "fs_stream_triad: allocation failed");
  if (!rc) {
    (void)hipMemsetAsync(b, 0, nb, ctx->stream); (void)hipMemsetAsync(c, 0, nb, ctx->stream);
    fsk_triad(n_doubles, a, b, c, 0.5, ctx->stream);
    (void)hipEventRecord(e0, ctx->stream);
    for (int r = 0; r < reps; r++) fsk_triad(n_doubles, a, b, c, 0.5, ctx->stream);
    (void)hipEventRecord(e1, ctx->stream);
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f)
      rc = fail(ctx, FS_ERR_HIP, "fs_stream_triad: timing failed");
    else
      *gbs = 24. * (double)n_doubles * reps / (ms * 1e-3) / 1e9;
  }
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return rc;
}

extern "C" int fs_stage_kernel_ms(fs_ctx *ctx, int reset, double *avg_ms, int *n_launch)
{
  if (!ctx) return FS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.;
  for (size_t i = 0; i < ctx->ev_used; i++) {
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[i].first, ctx->ev[i].second));
    tot += ms;
  }
  if (avg_ms) *avg_ms = ctx->ev_used ? tot / (double)ctx->ev_used : 0.;
  if (n_launch) *n_launch = (int)ctx->ev_used;
  if (reset) { ctx->ev_used = 0; ctx->timing = true; }
  return FS_OK;
}

extern "C" int fs_debug_fetch(fs_ctx *ctx, int zone, int which, double *host, int n_doubles)
{
  int rc = ensure_ready(ctx, "fs_debug_fetch");
  if (rc) return rc;
  if (zone < 0 || zone >= (int)ctx->zones.size() || !host) return fail(ctx, FS_ERR_ARG, "fs_debug_fetch: bad argument");
  Zone &zn = ctx->zones[zone];
  if (which != 0 && which != 2) return fail(ctx, FS_ERR_ARG, "fs_debug_fetch: which must be 0 (drodm) or 2 (xmut)");
  const double *src = which == 0 ? zn.drodm : zn.xmut;
  long long n = which == 0 ? (long long)zn.z.neq * zn.z.nd : zn.z.nd;
  if (!src) return fail(ctx, FS_ERR_STATE, "fs_debug_fetch: array not allocated for this scheme");
  if (n_doubles != n) return fail(ctx, FS_ERR_ARG, "fs_debug_fetch: size mismatch");
  HIPCHK(ctx, hipMemcpyAsync(host, src, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FS_OK;
}

// ---- shared with fasts_comm.hip --------------------------------------------------------------------------------
int fs_stage_roles_out(const fs_ctx *ctx, int nstep) { int in, out; stage_roles(ctx, nstep, in, out); return out; }
int fs_level_role(int level) { return level_role(level); }
int fs_ensure_ready(fs_ctx *ctx, const char *who) { return ensure_ready(ctx, who); }

extern "C" int fs_zone_set_global_id(fs_ctx *ctx, int zone, int gid)
{
  if (!ctx) return FS_ERR_ARG;
  if (zone < 0 || zone >= (int)ctx->zones.size() || gid < 0) return fail(ctx, FS_ERR_ARG, "fs_zone_set_global_id: bad argument");
  ctx->zones[zone].gid = gid;
  return FS_OK;
}

extern "C" int fs_send_add_remote(fs_ctx *ctx, int donor_zone, int peer_rank, const int32_t *donor_pts, int rcv_gid,
                                  const int32_t *rcv_pts, int npts)
{
  if (!ctx || !rcv_pts || rcv_gid < 0) return ctx ? fail(ctx, FS_ERR_ARG, "fs_send_add_remote: bad argument") : FS_ERR_ARG;
  int rc = fs_send_add(ctx, donor_zone, peer_rank, donor_pts, npts);
  if (rc) return rc;
  ctx->sends.back().rcv_gid = rcv_gid;
  ctx->sends.back().rcv_host.assign(rcv_pts, rcv_pts + npts);
  return FS_OK;
}

// ---- joins from the flat connectivity arrays ------------------------------------------------------------------------
// tc/Parameter_int and tc/Parameter_real as the reference reads them (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp):
//   [0] initialised flag, [1] number of point-to-point blocks, [2] sizecomIBC + that many entries, then sizecomID + that
//   many entries (:114-118); the pointer `ech` of block ip2p = 1.. sits at [ip2p + sizecomIBC + sizecomID + 3] (:220);
//   a block: [ech] destination rank, [ech+1] nrac, [ech+2] nrac_inst, [ech+3] timelevel, 2*timelevel entries, then
//   arrays of nrac entries read at shift_rac + nrac*k, shift_rac = ech + 4 + 2*timelevel + irac (:474-478,522-528):
//   k = 1 NbInterpD, 3 ibcType (< 0: interpolation data, no immersed boundary), 5 donor zone, 6 -> donor points,
//   7 -> [ntype, counts..., types...], 8 -> coefficients (Parameter_real), and one entry further on 10 receiver
//   points count, 11 receiver zone, 12 -> receiver points, 13 nvars, 14 rotation (:646-653,748-758).
// Only what a match join needs is accepted: steady racs, ibcType < 0, types 0 / 1 with unit coefficients, no rotation.
extern "C" int fs_joins_set(fs_ctx *ctx, const int32_t *pint_tc, int n_int, const double *preal_tc, int n_real, int rank)
{
  if (!ctx) return FS_ERR_ARG;
  if (!pint_tc || !preal_tc || n_int < 4) return fail(ctx, FS_ERR_ARG, "fs_joins_set: bad argument");
  if (ctx->uploaded) return fail(ctx, FS_ERR_STATE, "fs_joins_set: call before fs_upload");
  auto at = [&](long long i, int &v) { if (i < 0 || i >= n_int) return false; v = pint_tc[i]; return true; };
  int np2p, szibc, szid;
  if (!at(1, np2p) || !at(2, szibc) || szibc < 0 || !at(3 + (long long)szibc, szid) || szid < 0)
    return fail(ctx, FS_ERR_ARG, "fs_joins_set: truncated header");
  const long long shift_graph = (long long)szibc + szid + 3;
  const int nz = (int)ctx->zones.size();
  for (int ip2p = 1; ip2p <= np2p; ip2p++) {
    int ech, dest, nrac, nrac_inst, timelevel;
    if (!at(ip2p + shift_graph, ech) || !at(ech, dest) || !at((long long)ech + 1, nrac) || !at((long long)ech + 2, nrac_inst) ||
        !at((long long)ech + 3, timelevel) || nrac < 0 || timelevel < 0)
      return fail(ctx, FS_ERR_ARG, "fs_joins_set: truncated point-to-point block");
    if (nrac_inst != 0) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_joins_set: unsteady joins (nrac_inst > 0) are not supported");
    for (int irac = 0; irac < nrac; irac++) {
      const long long sr = (long long)ech + 4 + 2LL * timelevel + irac;
      int ninterp, ibctype, nod, pdon, ptyp, pcoef, nrcv, nor, prcv, nvars, rot;
      if (!at(sr + nrac, ninterp) || !at(sr + 3LL * nrac, ibctype) || !at(sr + 5LL * nrac, nod) || !at(sr + 6LL * nrac, pdon) ||
          !at(sr + 7LL * nrac, ptyp) || !at(sr + 8LL * nrac, pcoef) || !at(sr + 10LL * nrac + 1, nrcv) ||
          !at(sr + 11LL * nrac + 1, nor) || !at(sr + 12LL * nrac + 1, prcv) || !at(sr + 13LL * nrac + 1, nvars) ||
          !at(sr + 14LL * nrac + 1, rot))
        return fail(ctx, FS_ERR_ARG, "fs_joins_set: truncated rac table");
      if (ibctype >= 0) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_joins_set: immersed-boundary racs are not supported");
      if (rot != 0) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_joins_set: azimuthal periodicity (rotation) is not supported");
      if (nvars != 5 && nvars != 6) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_joins_set: only 5 or 6 variables per point");
      if (nrcv <= 0 || pdon < 0 || prcv < 0 || ptyp < 0 || (long long)pdon + nrcv > n_int || (long long)prcv + nrcv > n_int)
        return fail(ctx, FS_ERR_ARG, "fs_joins_set: point lists outside the array");
      if (nod < 0 || nod >= nz) return fail(ctx, FS_ERR_ARG, "fs_joins_set: donor zone index outside the zones of the context");
      if (nvars != ctx->zones[nod].z.neq)
        return fail(ctx, FS_ERR_ARG, "fs_joins_set: a rac carries " + std::to_string(nvars) + " variables per point, its donor zone has " +
                                         std::to_string(ctx->zones[nod].z.neq) + " equations");
      // interpolation types: [ntype, points per type..., type of every point]; a match join copies one donor
      int ntype;
      if (!at(ptyp, ntype) || ntype < 1 || (long long)ptyp + 1 + ntype + nrcv > n_int)
        return fail(ctx, FS_ERR_ARG, "fs_joins_set: type table outside the array");
      long long tot = 0;
      for (int t = 0; t < ntype; t++) tot += pint_tc[ptyp + 1 + t];
      if (tot != nrcv) return fail(ctx, FS_ERR_ARG, "fs_joins_set: type counts do not add up to the number of points");
      const int32_t *types = pint_tc + ptyp + 1 + ntype;
      for (int i = 0; i < nrcv; i++)
        if (types[i] != 0 && types[i] != 1) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_joins_set: only match joins (one donor per point)");
      if (ninterp != nrcv || pcoef < 0 || (long long)pcoef + ninterp > n_real)
        return fail(ctx, FS_ERR_ARG, "fs_joins_set: coefficient list outside the array");
      for (int i = 0; i < ninterp; i++)
        if (preal_tc[pcoef + i] != 1.) return fail(ctx, FS_ERR_UNSUPPORTED, "fs_joins_set: only match joins (unit coefficients)");
      int rc;
      if (dest == rank) {
        if (nor < 0 || nor >= nz) return fail(ctx, FS_ERR_ARG, "fs_joins_set: receiver zone index outside the zones of the context");
        rc = fs_join_add(ctx, nor, nod, pint_tc + prcv, pint_tc + pdon, nrcv);
      } else {
        // receiver zone = its index on the destination rank, as the reference ships it (setInterpTransfersFast.cpp:1185-1206)
        rc = fs_send_add_remote(ctx, nod, dest, pint_tc + pdon, nor, pint_tc + prcv, nrcv);
      }
      if (rc) return rc;
    }
  }
  return FS_OK;
}
