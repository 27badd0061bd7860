// Context and zone records shared by the translation units of the C ABI (fasts_capi.hip, fasts_comm.hip).
#pragma once
#include "../../include/fasts_hip.h"
#include "fasts_kernels.h"
#include "fasts_subzones.h"

#include <string>
#include <vector>

struct Zone {
  FsZone z;
  int gid = -1;                // job-wide zone number (fs_zone_set_global_id), names the receiver of a remote join
  std::vector<int32_t> pint;
  std::vector<double> preal;
  double *host[3] = {nullptr, nullptr, nullptr};  // roles: 0 = N, 1 = M1, 2 = P1
  void *pinned[3] = {nullptr, nullptr, nullptr};  // host level buffers registered with the runtime (fs_upload), so that
                                                  // fs_upload / fs_download move them at the rate of the host link
  double *dev[3] = {nullptr, nullptr, nullptr};
  double *drodm = nullptr, *xmut = nullptr, *wig = nullptr;
  double *wsc = nullptr;       // o3sc slope: shock sensor of the faces, 3 doubles per cell
  double *wflux = nullptr;     // o3sc slope: face fluxes of the three directions, 15 doubles per cell
  double *mtr_dev = nullptr;   // curvilinear zones: [ti(ndm,3) | tj | tk | vol]
  // implicit scheme: coe(nd,5) = dt/vol, spectral radii i/j/k, LU diagonal; one spare time level
  double *coe = nullptr, *scratch = nullptr;
  double *ssor_work = nullptr;   // nb_relax > 1: increments and accumulator of the SSOR relaxations, 2 * neq * nd
  double *dist = nullptr;      // SA model: wall distance of every cell (fs_zone_set_distance)
  double *rof = nullptr, *rof_host = nullptr;   // selective frequency damping: filtered state (fs_zone_set_filtered)
  const double *dist_host = nullptr;
  // implicit scheme with wall BCs: windows applied to the Newton residual before the LU solve (whole-face
  // extrapolation first, then the BCs of the wall faces) and the LU range (one ghost layer over wall faces)
  std::vector<FsBcWin> rhs_face, rhs_bc;
  FsBcWin *rhs_face_dev = nullptr, *rhs_bc_dev = nullptr;
  long long rhs_max_cells = 0;
  int lu[6] = {0, 0, 0, 0, 0, 0};
  // LU ranges of the blocks swept independently of each other (fs_zone_set_lu_blocks), 6 integers each; empty = the
  // whole range `lu` in one lexicographic sweep
  std::vector<int> lu_blocks;
  int *lu_blocks_dev = nullptr;
  FsLuFlow *lu_flow = nullptr;   // one-launch-per-sweep form of the LU-SGS sweeps (created at the first implicit solve)
  double *res_part = nullptr, *res_out = nullptr;   // res_out: 12 residual sums / maxima per sub-iteration, then the CFL triple
  int res_slots = 0;              // sub-iterations of the last verification step whose norms are in res_out
  bool have_cfl = false;
  hipEvent_t ev_lu_fork = nullptr, ev_lu_join = nullptr;
  FsBcWin *bc_dev = nullptr;
  std::vector<double> bc_data;          // window data of the BCs that carry their own, concatenated
  std::vector<long long> bc_data_off;   // per entry of bc: offset in bc_data, -1 without data
  double *bc_data_dev = nullptr;
  std::vector<FsBcWin> bc;
  long long bc_max_cells = 0;
  bool have_res = false;
  // stage-kernel tiles that hold donor cells of joins to other contexts ("shell") and the others
  int *shell_dev = nullptr, *inner_dev = nullptr;
  int n_shell = 0, n_inner = 0;
  bool has_sends = false;         // some of its cells are donors of joins to other contexts
  bool has_recvs = false;         // some of its ghost cells are filled from other contexts (fs_halo_unpack)
  // roe + walls: windows whose wall-face flux is corrected after the balance (BC/bfl3.for), in BC-table order
  struct RoeWall { int tag, idir, lo[3], hi[3]; };
  std::vector<RoeWall> roe_walls;
  hipEvent_t ev_part = nullptr;   // start of a split stage (timing)
  // 'implicit_local': blocks, targets and boxes; norms of the blocks at the sub-iterations of a verification step (on the
  // device until the step is over, then folded into sub.rdm on the host); BC windows cut to the boxes, per sub-iteration
  FsSubZones sub;
  double *blk_res = nullptr;
  struct PendingNorms { int nstep, nis; std::vector<int32_t> no_lu; };
  std::vector<PendingNorms> pending;
  struct BoxBcs {
    FsBcWin *dev = nullptr; size_t cap = 0; int n = 0, kinds = 0; long long max_cells = 0;
    // tiles of the stage kernel that hold cells of each box: one list after the other on the device
    int *tiles_dev = nullptr; size_t tiles_cap = 0;
    std::vector<int> tile_off, tile_cnt;
    // LU ranges of the boxes (a wall face takes its ghost layer), host and device
    std::vector<int> lu_ranges; int *lu_dev = nullptr; size_t lu_cap = 0;
    FsLuFlow *flow = nullptr;   // one-launch-per-sweep tables of these ranges (rebuilt when the boxes change)
  };
  std::vector<BoxBcs> box_bcs;
  std::vector<double> res_host;   // 12 norms per slot of the last verification step, summed over the blocks
};

struct Join {
  int rcv_zone, donor_zone, peer_tag, npts;
  int *rcv_dev = nullptr, *dnr_dev = nullptr;
};

struct Send {
  int donor_zone, peer_tag, npts;
  int rcv_gid = -1;                   // global id of the receiver zone on the peer (library transport only)
  std::vector<int32_t> rcv_host;      // its ghost cells, shipped to the peer once by fs_comm_setup
  int *pts_dev = nullptr;
  std::vector<int32_t> pts_host;
};

struct fs_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipStream_t comm_stream = nullptr;   // halo pack / unpack launch here when set (fs_set_comm_stream)
  std::vector<Zone> zones;
  std::vector<Join> joins;
  std::vector<Send> sends;
  bool uploaded = false;
  void *comm = nullptr;         // state of the library's own transport (fasts_comm.hip), NULL without fs_comm_init
  double *diag_out = nullptr;   // device: the three sums of fs_tgv_sums
  int modulo_verif = 200;
  // LU-SGS sweeps of different zones are independent and latency-bound (one small launch per hyperplane): they run
  // on a few streams of their own, forked from and joined back into the context's stream by events
  std::vector<hipStream_t> lu_streams;
  int lu_nstreams = -1;    // -1: not decided yet (FASTS_HIP_LU_STREAMS, default 4; 0 or 1 = everything on one stream)
  bool implicit = false;      // ITYPCP 1: BDF + LU-SGS Newton sub-iterations (all zones of a context agree)
  int first_it = 0;           // FIRST_IT of the reference: 0 until one time step is complete (BDF1, xid = 1)
  int local_newton = 0;       // lssiter_loc = 1 ('implicit_local'): Newton sub-iterations on sub-zones
  std::vector<int32_t> iskip_lu;            // schedule of the sub-iterations, iskip_lu(nssiter, 2)
  int local_nitrun = -1, local_nstep = -1;  // the sub-iteration the sub-zone lists were last prepared for
  std::string err;
  // stage-kernel timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t ev_used = 0;
};

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return FS_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

static inline int fail(fs_ctx *ctx, int code, const std::string &msg)
{
  ctx->err = msg;
  return code;
}

// shared between the translation units
int fs_stage_roles_out(const fs_ctx *ctx, int nstep);          // role (0 N, 1 M1, 2 P1) a sub-iteration writes
int fs_level_role(int level);                                   // FS_LEVEL_* -> role, -1 when invalid
int fs_ensure_ready(fs_ctx *ctx, const char *who);
void fs_comm_release(fs_ctx *ctx);                              // called by fs_destroy
bool fs_comm_active(const fs_ctx *ctx);
// halo messages of one sub-iteration (nstep >= 1: its output level; nstep <= 0: time level -nstep): begin = after what
// the context's stream holds so far, pack -> send/recv -> unpack on the comm stream; end = the context's stream waits
int fs_comm_exchange_begin(fs_ctx *ctx, int nstep);
int fs_comm_exchange_end(fs_ctx *ctx);
