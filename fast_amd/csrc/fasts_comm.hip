// Halo transport of the library: RCCL point-to-point messages between the contexts (one per GPU / rank) of a job.
// Replaces the MPI side of K_FASTC::setInterpTransfersInter / getTransfersInter
// (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:944-1582,1590-1719): per sub-iteration and per peer one packed
// buffer of donor values goes out and one comes in (ncclSend / ncclRecv in one group, straight between device
// buffers over xGMI); the receiver index lists, which the reference ships with every message (:1185-1206), are
// shipped once, by fs_comm_setup.  RCCL is opened at run time (dlopen) by fs_comm_init: a single-GPU job never loads it.
#include "fasts_ctx.h"

#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <map>

namespace {

struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::string g_rccl_err;

bool rccl_open()
{
  if (g_rccl.h) return true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) { g_rccl_err = std::string("cannot open librccl: ") + dlerror(); return false; }
#define FS_SYM(field, name)                                                   \
  g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                      \
  if (!g_rccl.field) { g_rccl_err = std::string("librccl lacks ") + name; return false; }
  FS_SYM(GetUniqueId, "ncclGetUniqueId") FS_SYM(CommInitRank, "ncclCommInitRank") FS_SYM(CommDestroy, "ncclCommDestroy")
  FS_SYM(GroupStart, "ncclGroupStart") FS_SYM(GroupEnd, "ncclGroupEnd") FS_SYM(Send, "ncclSend") FS_SYM(Recv, "ncclRecv")
  FS_SYM(AllGather, "ncclAllGather") FS_SYM(AllReduce, "ncclAllReduce") FS_SYM(GetErrorString, "ncclGetErrorString")
#undef FS_SYM
  g_rccl.h = h;
  return true;
}

// RCCL prints a version banner on the standard output of rank 0 when it initialises.  Callers of this library own
// their standard output (a benchmark prints one line of JSON there), so the banner is sent to the standard error for
// the duration of the two initialising calls.
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) (void)dup2(2, 1); }
  ~StdoutToStderr() { if (saved >= 0) { fflush(stdout); (void)dup2(saved, 1); close(saved); } }
};

struct Peer {
  int rank = 0;
  long long nsend = 0, nrecv = 0;          // doubles per message
  double *sbuf = nullptr, *rbuf = nullptr; // device
};

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  bool ready = false;                      // fs_comm_setup done
  hipStream_t stream = nullptr;            // pack -> messages -> unpack
  hipEvent_t ev_go = nullptr, ev_done = nullptr;
  std::vector<Peer> peers;                 // ascending rank
  double *red = nullptr;                   // device scratch of fs_comm_allreduce
};

#define NCCLCHK(ctx, call)                                                                    \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      (ctx)->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_);                     \
      return FS_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

Comm *comm_of(const fs_ctx *ctx) { return (Comm *)ctx->comm; }

// Inside ncclGroupStart / ncclGroupEnd the first error is kept and the group is always closed: a return from the
// middle would leave every later call of this thread queued in a group that never ends.
struct Group {
  ncclResult_t first = ncclSuccess;
  const char *what = "";
  Group() { note(g_rccl.GroupStart(), "ncclGroupStart"); }
  void note(ncclResult_t r, const char *w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; what = w; } }
  // closes the group; 0 or FS_ERR_HIP with the text of the first error
  int end(fs_ctx *ctx)
  {
    note(g_rccl.GroupEnd(), "ncclGroupEnd");
    if (first == ncclSuccess) return FS_OK;
    ctx->err = std::string(what) + ": " + g_rccl.GetErrorString(first);
    return FS_ERR_HIP;
  }
};

// device buffers of one call, freed on every way out
struct DevInts {
  std::vector<int32_t *> p;
  ~DevInts() { for (int32_t *q : p) (void)hipFree(q); }
  hipError_t get(int32_t **out, size_t n) { hipError_t e = hipMalloc(out, sizeof(int32_t) * n); if (e == hipSuccess) p.push_back(*out); return e; }
};

}  // namespace

bool fs_comm_active(const fs_ctx *ctx) { return ctx->comm && comm_of(ctx)->ready; }

void fs_comm_release(fs_ctx *ctx)
{
  Comm *c = comm_of(ctx);
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto &p : c->peers) { (void)hipFree(p.sbuf); (void)hipFree(p.rbuf); }
  (void)hipFree(c->red);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->ev_go) (void)hipEventDestroy(c->ev_go);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  ctx->comm = nullptr;
  ctx->comm_stream = nullptr;
}

extern "C" int fs_comm_unique_id(void *id, int nbytes)
{
  if (!id || nbytes < NCCL_UNIQUE_ID_BYTES) return FS_ERR_ARG;
  if (!rccl_open()) return FS_ERR_HIP;
  ncclUniqueId u;
  {
    StdoutToStderr quiet;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return FS_ERR_HIP;
  }
  memcpy(id, &u, NCCL_UNIQUE_ID_BYTES);
  return FS_OK;
}

extern "C" int fs_comm_init(fs_ctx *ctx, int rank, int world, const void *id, int nbytes)
{
  if (!ctx) return FS_ERR_ARG;
  if (!id || nbytes < NCCL_UNIQUE_ID_BYTES || world < 1 || rank < 0 || rank >= world)
    return fail(ctx, FS_ERR_ARG, "fs_comm_init: bad argument");
  if (ctx->comm) return fail(ctx, FS_ERR_STATE, "fs_comm_init: the context already has a communicator");
  if (!rccl_open()) return fail(ctx, FS_ERR_HIP, "fs_comm_init: " + g_rccl_err);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Comm *c = new Comm;
  c->rank = rank; c->world = world;
  ctx->comm = c;
  ncclUniqueId u;
  memcpy(&u, id, NCCL_UNIQUE_ID_BYTES);
  {
    StdoutToStderr quiet;
    NCCLCHK(ctx, g_rccl.CommInitRank(&c->comm, world, u, rank));
  }
  HIPCHK(ctx, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(ctx, hipEventCreateWithFlags(&c->ev_go, hipEventDisableTiming));
  HIPCHK(ctx, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  HIPCHK(ctx, hipMalloc(&c->red, sizeof(double) * 64));
  ctx->comm_stream = c->stream;
  return FS_OK;
}

// Handshake, once: every rank tells every peer which ghost cells its messages fill.  Wire format per message, int32:
// [nrac | (receiver zone global id, npts, variables per point, receiver cells[npts]) x nrac], the integer part of the
// reference's message (setInterpTransfersFast.cpp:1185-1206: nrac, then per join NoR, the number of variables and the
// receiver list).  Every rank takes part in every collective of the call whatever it finds wrong on the way, and the
// ranks agree on the outcome at the end (one all-reduce): a rank that returned early would leave the others waiting.
extern "C" int fs_comm_setup(fs_ctx *ctx)
{
  int rc = fs_ensure_ready(ctx, "fs_comm_setup");
  if (rc) return rc;
  Comm *c = comm_of(ctx);
  if (!c) return fail(ctx, FS_ERR_STATE, "fs_comm_setup: call fs_comm_init first");
  if (c->ready) return fail(ctx, FS_ERR_STATE, "fs_comm_setup: already done");
  const int W = c->world;
  int bad = FS_OK;                 // first local problem; reported after the collectives
  std::string bad_text;
  auto problem = [&](int code, const std::string &text) { if (!bad) { bad = code; bad_text = text; } };
  std::map<int, int> gid_zone;
  for (size_t nz = 0; nz < ctx->zones.size(); nz++)
    if (ctx->zones[nz].gid >= 0) gid_zone[ctx->zones[nz].gid] = (int)nz;
  // outgoing metadata per destination (a send that cannot be described is left out and fails the call below)
  std::vector<std::vector<int32_t>> meta(W);
  for (auto &s : ctx->sends) {
    if (s.peer_tag < 0 || s.peer_tag >= W) { problem(FS_ERR_ARG, "fs_comm_setup: a send names a rank outside the job"); continue; }
    if (s.rcv_gid < 0) { problem(FS_ERR_STATE, "fs_comm_setup: sends must be registered with fs_send_add_remote"); continue; }
    auto &m = meta[s.peer_tag];
    if (m.empty()) m.push_back(0);
    m[0]++;
    m.push_back(s.rcv_gid); m.push_back(s.npts); m.push_back(ctx->zones[s.donor_zone].z.neq);
    m.insert(m.end(), s.rcv_host.begin(), s.rcv_host.end());
  }
  // sizes: row r of the matrix = int32 counts rank r sends to every rank
  std::vector<int32_t> mine(W), all((size_t)W * W);
  for (int p = 0; p < W; p++) mine[p] = (int32_t)meta[p].size();
  DevInts tmp;
  int32_t *d_mine = nullptr, *d_all = nullptr;
  HIPCHK(ctx, tmp.get(&d_mine, W));
  HIPCHK(ctx, tmp.get(&d_all, (size_t)W * W));
  HIPCHK(ctx, hipMemcpyAsync(d_mine, mine.data(), sizeof(int32_t) * W, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(ctx, g_rccl.AllGather(d_mine, d_all, W, ncclInt32, c->comm, c->stream));
  HIPCHK(ctx, hipMemcpyAsync(all.data(), d_all, sizeof(int32_t) * W * W, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(ctx, hipStreamSynchronize(c->stream));
  // metadata messages
  std::vector<int32_t *> d_out(W, nullptr), d_in(W, nullptr);
  std::vector<std::vector<int32_t>> in(W);
  for (int p = 0; p < W; p++) {
    if (mine[p] > 0) {
      HIPCHK(ctx, tmp.get(&d_out[p], mine[p]));
      HIPCHK(ctx, hipMemcpyAsync(d_out[p], meta[p].data(), sizeof(int32_t) * mine[p], hipMemcpyHostToDevice, c->stream));
    }
    const int32_t nin = all[(size_t)p * W + c->rank];
    if (nin > 0) { HIPCHK(ctx, tmp.get(&d_in[p], nin)); in[p].resize(nin); }
  }
  {
    Group g;
    for (int p = 0; p < W; p++) {
      if (d_out[p]) g.note(g_rccl.Send(d_out[p], mine[p], ncclInt32, p, c->comm, c->stream), "ncclSend");
      if (d_in[p]) g.note(g_rccl.Recv(d_in[p], in[p].size(), ncclInt32, p, c->comm, c->stream), "ncclRecv");
    }
    rc = g.end(ctx);
    if (rc) return rc;
  }
  for (int p = 0; p < W; p++)
    if (d_in[p]) HIPCHK(ctx, hipMemcpyAsync(in[p].data(), d_in[p], sizeof(int32_t) * in[p].size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(ctx, hipStreamSynchronize(c->stream));
  // receive side: ghost-cell joins in the order the peer packs them
  for (int p = 0; p < W && !bad; p++) {
    if (in[p].empty()) continue;
    const std::vector<int32_t> &m = in[p];
    size_t pos = 1;
    for (int q = 0; q < m[0] && !bad; q++) {
      if (pos + 3 > m.size()) { problem(FS_ERR_ARG, "fs_comm_setup: truncated join table from a peer"); break; }
      const int gid = m[pos], npts = m[pos + 1], nvars = m[pos + 2];
      pos += 3;
      if (npts <= 0 || pos + (size_t)npts > m.size()) { problem(FS_ERR_ARG, "fs_comm_setup: truncated join table from a peer"); break; }
      auto it = gid_zone.find(gid);
      if (it == gid_zone.end()) { problem(FS_ERR_ARG, "fs_comm_setup: a peer sends ghost values for a zone that is not on this rank"); break; }
      if (nvars != ctx->zones[it->second].z.neq) {
        problem(FS_ERR_ARG, "fs_comm_setup: a peer sends " + std::to_string(nvars) + " variables per point to a zone with " +
                                std::to_string(ctx->zones[it->second].z.neq) + " equations");
        break;
      }
      rc = fs_join_add(ctx, it->second, -p - 1, m.data() + pos, nullptr, npts);
      if (rc) { problem(rc, ctx->err); break; }
      pos += npts;
    }
  }
  // message buffers
  for (int p = 0; p < W && !bad; p++) {
    Peer pr;
    pr.rank = p;
    pr.nsend = fs_halo_count(ctx, p, 1); pr.nrecv = fs_halo_count(ctx, p, 0);
    if (!pr.nsend && !pr.nrecv) continue;
    if (pr.nsend && hipMalloc(&pr.sbuf, sizeof(double) * pr.nsend) != hipSuccess) problem(FS_ERR_HIP, "fs_comm_setup: no memory for a send buffer");
    if (pr.nrecv && hipMalloc(&pr.rbuf, sizeof(double) * pr.nrecv) != hipSuccess) problem(FS_ERR_HIP, "fs_comm_setup: no memory for a receive buffer");
    c->peers.push_back(pr);      // released by fs_comm_release also when the call fails
  }
  // the ranks agree: one failure fails the call everywhere
  double worst = bad ? 1. : 0.;
  HIPCHK(ctx, hipMemcpyAsync(c->red, &worst, sizeof(double), hipMemcpyHostToDevice, c->stream));
  NCCLCHK(ctx, g_rccl.AllReduce(c->red, c->red, 1, ncclDouble, ncclMax, c->comm, c->stream));
  HIPCHK(ctx, hipMemcpyAsync(&worst, c->red, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(ctx, hipStreamSynchronize(c->stream));
  if (bad) return fail(ctx, bad, bad_text);
  if (worst != 0.) return fail(ctx, FS_ERR_STATE, "fs_comm_setup: another rank of the job failed in this call");
  // zones with remote ghost cells: scheduling of split stages
  for (auto &j : ctx->joins) if (j.donor_zone < 0) ctx->zones[j.rcv_zone].has_recvs = true;
  c->ready = true;
  return FS_OK;
}

int fs_comm_exchange_begin(fs_ctx *ctx, int nstep)
{
  Comm *c = comm_of(ctx);
  if (!c || !c->ready) return fail(ctx, FS_ERR_STATE, "halo exchange: call fs_comm_init and fs_comm_setup first");
  HIPCHK(ctx, hipEventRecord(c->ev_go, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(c->stream, c->ev_go, 0));
  for (auto &p : c->peers)
    if (p.nsend) { int rc = fs_halo_pack(ctx, nstep, p.rank, p.sbuf, (int)p.nsend); if (rc) return rc; }
  {
    Group g;
    for (auto &p : c->peers) {
      if (p.nsend) g.note(g_rccl.Send(p.sbuf, (size_t)p.nsend, ncclDouble, p.rank, c->comm, c->stream), "ncclSend");
      if (p.nrecv) g.note(g_rccl.Recv(p.rbuf, (size_t)p.nrecv, ncclDouble, p.rank, c->comm, c->stream), "ncclRecv");
    }
    const int grc = g.end(ctx);
    if (grc) return grc;
  }
  for (auto &p : c->peers)
    if (p.nrecv) { int rc = fs_halo_unpack(ctx, nstep, p.rank, p.rbuf, (int)p.nrecv); if (rc) return rc; }
  HIPCHK(ctx, hipEventRecord(c->ev_done, c->stream));
  return FS_OK;
}

int fs_comm_exchange_end(fs_ctx *ctx)
{
  Comm *c = comm_of(ctx);
  if (!c || !c->ready) return fail(ctx, FS_ERR_STATE, "halo exchange: call fs_comm_init and fs_comm_setup first");
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
  return FS_OK;
}

extern "C" int fs_halo_exchange(fs_ctx *ctx, int nstep)
{
  int rc = fs_ensure_ready(ctx, "fs_halo_exchange");
  if (rc) return rc;
  rc = fs_comm_exchange_begin(ctx, nstep);
  if (rc) return rc;
  return fs_comm_exchange_end(ctx);
}

extern "C" int fs_comm_allreduce(fs_ctx *ctx, double *vals, int n, int op)
{
  if (!ctx) return FS_ERR_ARG;
  Comm *c = comm_of(ctx);
  if (!c) return fail(ctx, FS_ERR_STATE, "fs_comm_allreduce: call fs_comm_init first");
  if (!vals || n < 1 || n > 64 || op < 0 || op > 2) return fail(ctx, FS_ERR_ARG, "fs_comm_allreduce: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(c->red, vals, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(ctx, g_rccl.AllReduce(c->red, c->red, n, ncclDouble, op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin), c->comm, c->stream));
  HIPCHK(ctx, hipMemcpyAsync(vals, c->red, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(ctx, hipStreamSynchronize(c->stream));
  return FS_OK;
}
