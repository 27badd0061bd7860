"""Test transport: halo messages over torch.distributed's gloo backend, so that several ranks can be rehearsed on
the CPU or share one GPU (RCCL, the product's transport inside libfasts_hip.so, wants one GPU per rank).  Drives the
product's split native calls (fs_stage_part / fs_halo_pack / fs_halo_unpack) through the `transport=` hook of
fast_amd.Mpi.warmup.  Import torch before fast_amd in a process that uses this module."""
import torch
import torch.distributed as dist


class HaloExchange:
    """Per-peer send / receive buffers and the message round of one sub-iteration."""

    def __init__(self, plan, device, neq=5, sync=None, counts=None):
        """device: where pack / unpack work.  Device buffers are staged through pinned host memory; `sync` (a
        callable that waits for the pack kernels) is then called before the copy out."""
        self.plan = plan
        self.sbuf, self.rbuf, self.hs, self.hr = {}, {}, {}, {}
        self.sync = sync
        dev = torch.device(device)
        self.staged = dev.type == 'cuda'
        if counts is None:
            counts = lambda p, send: plan.count(p, send, neq)
        for p in plan.peers():
            self.sbuf[p] = torch.empty(counts(p, True), dtype=torch.float64, device=dev)
            self.rbuf[p] = torch.empty(counts(p, False), dtype=torch.float64, device=dev)
            if self.staged:
                self.hs[p] = torch.empty(self.sbuf[p].numel(), dtype=torch.float64).pin_memory()
                self.hr[p] = torch.empty(self.rbuf[p].numel(), dtype=torch.float64).pin_memory()

    def run(self, pack, unpack):
        """pack(peer, send_tensor) fills the outgoing values; all messages of the round are posted together;
        unpack(peer, recv_tensor) scatters what arrived."""
        peers = self.plan.peers()
        if not peers:
            return
        for p in peers:
            if self.sbuf[p].numel():
                pack(p, self.sbuf[p])
        if self.staged:
            if self.sync is not None:
                self.sync()
            for p in peers:
                self.hs[p].copy_(self.sbuf[p])      # stream-ordered after the pack kernels, blocking
        wire_s = self.hs if self.staged else self.sbuf
        wire_r = self.hr if self.staged else self.rbuf
        me = dist.get_rank()
        ops = []
        for p in peers:
            if p == me:                             # loop-back message
                wire_r[p].copy_(wire_s[p])
                continue
            if wire_s[p].numel():
                ops.append(dist.P2POp(dist.isend, wire_s[p], p))
            if wire_r[p].numel():
                ops.append(dist.P2POp(dist.irecv, wire_r[p], p))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        if self.staged:
            for p in peers:
                self.rbuf[p].copy_(self.hr[p])
        for p in peers:
            if self.rbuf[p].numel():
                unpack(p, self.rbuf[p])


class GlooTransport:
    """The object fast_amd.Mpi.warmup(transport=...) takes: kernels on one explicit stream, pack -> messages ->
    unpack on a second one so that they overlap the tiles of a stage that feed no message."""
    native = False

    def __init__(self, device=0, overlap=True):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.overlap = overlap
        torch.cuda.set_device(device)
        self.main = torch.cuda.Stream(device=device)
        self.comm = torch.cuda.Stream(device=device)
        self.halo = None
        self.ev = None

    def all_gather_object(self, out, obj):
        dist.all_gather_object(out, obj)

    def attach(self, ctx):
        ctx.set_stream(self.main.cuda_stream)
        ctx.set_comm_stream(self.comm.cuda_stream)

    def ready(self, ctx, plan):
        with torch.cuda.stream(self.comm):
            self.halo = HaloExchange(plan, torch.device('cuda', self.device), counts=lambda p, send: ctx.halo_count(p, send))

    def mark(self, ctx):
        self.ev = self.main.record_event()

    def exchange(self, ctx, plan, nstep):
        if self.overlap and self.ev is not None:
            self.comm.wait_event(self.ev)
        else:
            self.comm.wait_stream(self.main)
        pack = lambda p, b: ctx.halo_pack(nstep, p, b.data_ptr(), b.numel())
        unpack = lambda p, b: ctx.halo_unpack(nstep, p, b.data_ptr(), b.numel())
        with torch.cuda.stream(self.comm):
            self.halo.run(pack, unpack)
        self.main.wait_stream(self.comm)
