"""Multi-rank driver: one rank per GPU, OpenFOAM-style sub-domains, shared-point exchange.

Replaces the reference's syncTools::syncPointList / returnReduce calls inside the loop
(src/smoothMesh.C:134,142,402,429,455,472 -> exchange "A"; :2374 -> exchange "F"; :1567,2396 ->
the 2-scalar reduction).  The library (include/smgpu.h, smgpu_iter_begin/mid/end) packs and
combines on the device; this module only builds the slot tables and moves the packed buffers with
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

Exchange pattern: every pair of ranks that shares points exchanges, in ascending global point id,
one record per shared point -- a single all_to_all_single per exchange (two per iteration) plus one
all_gather of {residual, nFrozenPoints}.  No collective touches non-shared data.
"""
import os

import numpy as np

A_DOUBLES = 13  # SMGPU_HALO_A_DOUBLES
L_DOUBLES = 14  # SMGPU_HALO_L_DOUBLES (layer treatment / boundary point smoothing record, layout in include/smgpu.h)


class HaloTables:
    """Slot tables of one rank from every rank's processor-patch point lists."""

    def __init__(self, rank, pointProcAddressing, patch_lists):
        """patch_lists[r] = {neighbour: sorted unique global ids of the points on rank r's processor patch to it}
        (SubDomain.processor_patch_point_lists).  Who shares a point with whom follows OpenFOAM's globalPoints
        (decompose.shared_point_groups): copies connected through processor patches -- not "every rank that holds the id": the
        two sides of a baffle on different ranks are different shared points, or none."""
        from .decompose import shared_point_components
        self.rank = rank
        n = len(patch_lists)
        node_rank, node_gid, comp, size = shared_point_components(patch_lists)
        mine = (node_rank == rank) & (size >= 2)         # (a rank has ONE local point per id: it is in at most one group per id)
        others = np.isin(comp, comp[mine]) & (node_rank != rank)
        shared_with = {int(o): np.sort(node_gid[others & (node_rank == o)]) for o in np.unique(node_rank[others])}
        allg = np.unique(np.concatenate(list(shared_with.values()))) if shared_with else np.zeros(0, np.int64)
        # local ids of the shared points, ordered by global id
        order = np.argsort(pointProcAddressing, kind="stable")
        sorted_g = pointProcAddressing[order]
        pos = np.searchsorted(sorted_g, allg)
        assert np.array_equal(sorted_g[pos], allg), "shared point not present in pointProcAddressing"
        self.sharedGlobal = allg
        self.sharedLocal = order[pos].astype(np.int32)
        # the group (connected component) of each shared point: every rank labels the components of the same all-gathered lists the
        # same way, so (global id, component) names a shared point on all its sharers -- the global id alone does not (a baffle
        # between ranks: one mesh point, two shared points)
        mg, mc = node_gid[mine], comp[mine]
        o2 = np.argsort(mg, kind="stable")
        self.sharedComp = mc[o2][np.searchsorted(mg[o2], allg)].astype(np.int64) if len(allg) else np.zeros(0, np.int64)
        self.counts = np.zeros(n, np.int64)         # send == recv counts per peer (symmetric lists)
        send = []
        base = {}
        run = 0
        for o in range(n):
            if o in shared_with:
                idx = np.searchsorted(allg, shared_with[o]).astype(np.int32)
                send.append(idx)
                self.counts[o] = len(idx)
                base[o] = (run, idx)
                run += len(idx)
        self.sendShared = np.concatenate(send).astype(np.int32) if send else np.zeros(0, np.int32)
        self.nSend = self.nRecv = int(run)
        # combine table: sharers ascending by rank, -1 marks this rank
        nsh = len(allg)
        per = [[] for _ in range(nsh)]
        for o in sorted(list(shared_with.keys()) + [rank]):
            if o == rank:
                for i in range(nsh):
                    per[i].append(-1)
            else:
                b, idx = base[o]
                for k, i in enumerate(idx):
                    per[i].append(b + k)
        self.combOffsets = np.zeros(nsh + 1, np.int32)
        if nsh:
            np.cumsum([len(p) for p in per], out=self.combOffsets[1:])
        self.combSlots = np.array([s for p in per for s in p], dtype=np.int32) if nsh else np.zeros(0, np.int32)


def own_fold_default():
    """the syncTools::syncPointList model (include/smgpu.h smgpu_set_sync_variant): False = the master's fold handed to every
    sharer (default), True = every sharer folds onto its own value (SMGPU_SYNC_VARIANT=own, the engine reads the same variable)"""
    return os.environ.get("SMGPU_SYNC_VARIANT", "") == "own"


def combine_shared(tables, own, recv, op, own_fold=None):
    """syncTools::syncPointList for the shared points of one rank, on the host (set-up only).  own: (nShared, k) this
    rank's values, recv: (nRecv, k) the other sharers' values in recv-slot order.  op: "max" (maxEqOp), "sum" (plusEqOp,
    ascending rank order), "maxmag" (maxMagSqrEqOp: globalMeshData::syncData folds the sharers' values once, starting from
    the lowest rank's, in ascending rank order -- the larger magnitude wins, a tie keeps the lower rank's -- and every
    sharer receives that value; own_fold: every sharer folds the others onto its own value instead)."""
    if own_fold is None:
        own_fold = own_fold_default()
    out = own.copy()
    off, slots = tables.combOffsets, tables.combSlots
    for i in range(len(own)):
        sl = slots[off[i]:off[i + 1]]
        if op == "max":
            for s in sl:
                if s >= 0:
                    out[i] = np.maximum(out[i], recv[s])
        elif op == "sum":
            acc = np.zeros_like(own[i])
            for s in sl:
                acc = acc + (own[i] if s < 0 else recv[s])
            out[i] = acc
        elif op == "maxmag":
            x = own[i] if (own_fold or sl[0] < 0) else recv[sl[0]]
            for s in (sl if own_fold else sl[1:]):
                if s >= 0 or not own_fold:
                    y = own[i] if s < 0 else recv[s]
                    # magSqr left to right in plain IEEE doubles (no BLAS / FMA), as the reference evaluates it
                    mx = float(x[0]) * float(x[0]) + float(x[1]) * float(x[1]) + float(x[2]) * float(x[2])
                    my = float(y[0]) * float(y[0]) + float(y[1]) * float(y[1]) + float(y[2]) * float(y[2])
                    x = x if mx >= my else y
            out[i] = x
        else:
            raise ValueError(op)
    return out


def setup_layers_stepwise(engines, exchange, lp, minEdgeLength):
    """The reference's layer set-up under -parallel (SM.C:2215-2221) on `engines` (this process's ranks), with its
    syncPointList calls done by `exchange(field_index, list of own arrays, op) -> list of combined arrays`."""
    E = type(engines[0])
    res = [e.layers_begin(lp, minEdgeLength) for e in engines]
    on, maxIter = res[0]
    assert all(r == res[0] for r in res)
    if not on:
        return False

    def sync(field, op):
        own = [e.layers_shared_get(field) for e in engines]
        for e, c in zip(engines, exchange(own, op)):
            e.layers_shared_set(field, c)

    for _ in range(maxIter):
        for e in engines:
            e.layers_step(E.LAYERS_HOPS_SWEEP)
        sync(E.LAYERS_F_HOPS, "max")                       # OBB.C:124-130
    for e in engines:
        e.layers_step(E.LAYERS_NORMALS_ACCUMULATE)
    sync(E.LAYERS_F_NORMALS_COUNT, "sum")                  # OBB.C:184-198
    for e in engines:
        e.layers_step(E.LAYERS_NORMALS_FINISH)
    for it in range(1, maxIter + 1):
        for e in engines:
            e.layers_step(E.LAYERS_PROPAGATE_SWEEP, it)
        sync(E.LAYERS_F_NORMALS, "maxmag")                 # OBB.C:359-365
    for e in engines:
        e.layers_step(E.LAYERS_FINISH)
    return True


def setup_boundary_stepwise(engines, exchange, reduce_stats, bp, minEdgeLength, layerEdgeLength=None):
    """The reference's boundary point smoothing set-up under -parallel (SM.C:2080-2253) on `engines` (this process's ranks):
    `reduce_stats(list of (minEdge, bb)) -> (minEdge, bb)` over ALL ranks (returnReduce, SM.C:1528-1535), `exchange` as in
    setup_layers_stepwise.  Returns the list of classification summaries (one per engine)."""
    E = type(engines[0])
    mn, bb = reduce_stats([e.boundary_stats() for e in engines])
    perimeter = bb[1] - bb[0] + bb[3] - bb[2] + bb[5] + bb[4]              # SM.C:1538, "+ bbMinZ" as written
    infos = [e.boundary_begin(bp, minEdgeLength, mn, perimeter, layerEdgeLength) for e in engines]
    assert all(i["enabled"] == infos[0]["enabled"] for i in infos)
    if not infos[0]["enabled"]:
        return infos

    def sync(field, op):
        own = [e.boundary_shared_get(field) for e in engines]
        for e, c in zip(engines, exchange(own, op)):
            e.boundary_shared_set(field, c)

    for _ in range(2):                                                      # SM.C:2218
        for e in engines:
            e.boundary_step(E.BOUNDARY_HOPS_SWEEP)
        sync(E.BOUNDARY_F_HOPS, "max")                                      # OBB.C:124-130
    for e in engines:
        e.boundary_step(E.BOUNDARY_TABLES)
    for e in engines:
        e.boundary_step(E.BOUNDARY_NORMALS_ACCUMULATE)
    sync(E.BOUNDARY_F_NORMALS_COUNT, "sum")                                 # OBB.C:184-198
    for e in engines:
        e.boundary_step(E.BOUNDARY_NORMALS_FINISH)
    return infos


def l_view(t, width):
    """the exchange-L records in use: `width` doubles per slot, packed from the start of the (slots, L_DOUBLES) buffer"""
    n = t.shape[0]
    return t.view(-1)[: n * width].view(n, width)


def push_layout(rank, counts_of):
    """Slot layout of the peer-store transport for `rank`, from every rank's slot counts per rank (counts_of[r][o] = slots of rank r
    towards rank o; symmetric).  Returns (peers, count, remoteBase, myIndexAtPeer): the ranks `rank` shares points with
    (ascending), the slots per peer, the first slot -- in the PEER's receive numbering, which groups its slots by source rank in
    ascending order -- of this rank's records, and this rank's position among the peer's peers (its flag word there)."""
    mine = counts_of[rank]
    peers = [o for o in range(len(mine)) if mine[o] > 0]
    cnt, base, my_index = [], [], []
    for o in peers:
        theirs = counts_of[o]
        assert theirs[rank] == mine[o], "asymmetric shared-point lists"
        cnt.append(int(mine[o]))
        base.append(int(sum(theirs[:rank])))
        my_index.append(sum(1 for r in range(rank) if theirs[r] > 0))
    return peers, cnt, base, my_index


class PushBuffers:
    """Receive buffers and flag words of the peer-store transport (include/smgpu.h, smgpu_push_desc): device memory other
    processes can map.  Allocated through the engine's library (uncached device memory + hipIpc handle); `handles` is what
    travels to the peers, `open_peer` maps theirs."""

    FLAG_BYTES = 4 * 2 * 64

    def __init__(self, device_index, nRecv):
        import ctypes as C
        from . import _ffi
        self._C, self._lib, self._dev = C, _ffi.lib(), int(device_index)
        self._opened = []
        n = max(int(nRecv), 1)
        self.ptr, self.handles = {}, {}
        for name, nbytes in (("recvA", n * A_DOUBLES * 8), ("recvL", n * L_DOUBLES * 8), ("recvF", n * 4), ("flags", self.FLAG_BYTES)):
            p, h = C.c_void_p(), (C.c_char * 64)()
            if self._lib.smgpu_push_alloc(self._dev, nbytes, C.byref(p), h):
                raise RuntimeError(self._lib.smgpu_last_error().decode())
            self.ptr[name], self.handles[name] = p.value, bytes(h)

    def open_peer(self, handles):
        C = self._C
        out = {}
        for name, raw in handles.items():
            p = C.c_void_p()
            if self._lib.smgpu_push_open(self._dev, C.create_string_buffer(raw, 64), C.byref(p)):
                raise RuntimeError(self._lib.smgpu_last_error().decode())
            out[name] = p.value
            self._opened.append(p.value)
        return out

    def close(self):
        for p in self._opened:
            self._lib.smgpu_push_close(self._C.c_void_p(p))
        self._opened = []
        for p in self.ptr.values():
            self._lib.smgpu_push_free(self._C.c_void_p(p))
        self.ptr = {}


class _RankState:
    """engine + exchange buffers of one rank (torch tensors on the engine's device)"""

    def __init__(self, sub, tables, engine, torch, device, exchange_stream=None, push=None):
        self.sub, self.t, self.eng = sub, tables, engine
        self.push = push     # PushBuffers: the receive side lives in IPC-mappable memory instead of torch tensors
        f64, i32 = torch.float64, torch.int32
        self.sendA = torch.zeros((max(tables.nSend, 1), A_DOUBLES), dtype=f64, device=device)
        self.recvA = torch.zeros((max(tables.nRecv, 1), A_DOUBLES), dtype=f64, device=device)
        self.sendF = torch.zeros(max(tables.nSend, 1), dtype=i32, device=device)
        self.recvF = torch.zeros(max(tables.nRecv, 1), dtype=i32, device=device)
        self.localStats = torch.zeros(2, dtype=f64, device=device)
        self.sendL = torch.zeros((max(tables.nSend, 1), L_DOUBLES), dtype=f64, device=device)
        self.recvL = torch.zeros((max(tables.nRecv, 1), L_DOUBLES), dtype=f64, device=device)
        rA, rF, rL = ((push.ptr["recvA"], push.ptr["recvF"], push.ptr["recvL"]) if push is not None
                      else (self.recvA.data_ptr(), self.recvF.data_ptr(), self.recvL.data_ptr()))
        engine.halo_configure(tables.sharedLocal, tables.sendShared, tables.nRecv, tables.combOffsets, tables.combSlots,
                              self.sendA.data_ptr(), rA, self.sendF.data_ptr(), rF,
                              self.localStats.data_ptr(), exchangeStream=exchange_stream,
                              sendL=self.sendL.data_ptr(), recvL=rL)


class DistributedSmoother:
    """The reference's loop (SM.C:2257-2437) on this rank's sub-domain, in step with all other ranks."""

    def __init__(self, sub, device=0, engine_factory=None, torch_device=None, probe_slots=0, overlap=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.sub = sub
        self.probe_slots = int(probe_slots)
        self.xstream = None
        self.layers = False
        self.boundary = False
        self.own_fold = own_fold_default()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert sub.rank == self.rank and sub.nRanks == self.world
        if torch_device is None:
            torch_device = torch.device("cuda", device)
        self.device = torch_device
        cands = [None] * self.world
        dist.all_gather_object(cands, sub.processor_patch_point_lists())
        self.tables = HaloTables(self.rank, sub.pointProcAddressing, cands)
        if engine_factory is None:
            from .engine import SmoothEngine
            # two torch-owned streams: the engine computes on estream (handed over as its caller stream); the exchanges are
            # enqueued on estream as well (in order) or on xstream (overlapped, ordered by the engine's events)
            self.estream = torch.cuda.Stream(torch_device)
            # (high priority: the exchange kernels are a few workgroups that the iteration waits for; next to a launch that fills
            # the chip they ran 2.5 times longer at normal priority)
            self.xstream = torch.cuda.Stream(torch_device, priority=-1)
            engine = SmoothEngine(sub.mesh, device=device, stream=self.estream.cuda_stream)
            if os.environ.get("SMOOTHMESH_SHARE_GPU") and self.world > 1:
                # several ranks on one device (a debugging arrangement): every rank's persistent walk replay needs all of its
                # workgroups resident at once, so each takes its share of the chip
                engine.set_device_share(-(-self.world // max(1, torch.cuda.device_count())))
            self.overlap = bool(overlap)
            xs = self.xstream.cuda_stream if self.overlap else None
        else:
            engine, xs = engine_factory(sub.mesh), None
        self.engine = engine
        # SMOOTHMESH_EXCHANGE=push: the peer-store transport (the ranks' kernels store the records into each other's receive
        # slots; nothing is exchanged by the host).  Needs the real engine, one node, in-order arrangement.
        self.pushbuf = None
        want_push = os.environ.get("SMOOTHMESH_EXCHANGE", "") == "push" and engine_factory is None and torch_device.type == "cuda"
        if want_push:
            self.overlap, xs = False, None
            self.pushbuf = PushBuffers(device, self.tables.nRecv)
        self.state = _RankState(sub, self.tables, engine, torch, torch_device, xs, push=self.pushbuf)
        self.counts = [int(c) for c in self.tables.counts]
        if want_push:
            self._open_push()
        self.allStats = torch.zeros((self.world, 2), dtype=torch.float64, device=torch_device)
        if self.probe_slots:
            z = lambda shape, dt: (torch.zeros(shape, dtype=dt, device=torch_device), torch.zeros(shape, dtype=dt, device=torch_device))
            self._probe = {torch.float64: z((self.probe_slots, A_DOUBLES), torch.float64), torch.int32: z((self.probe_slots,), torch.int32),
                           "L": z((self.probe_slots, L_DOUBLES), torch.float64)}
        self.direct = self._open_direct(engine_factory is None)

    def _open_push(self):
        """exchange the IPC handles and slot counts, map the peers' buffers, describe them to the engine"""
        dist, t = self.dist, self.tables
        if self.world == 1:
            if not self.probe_slots:
                return
            # measurement aid (scripts/probe_rank_of_8.py): the tables are those of one rank among several, the "peers" are this
            # rank itself -- send slot k lands in the own receive slot k, every peer's flag words are the own ones
            peers = [o for o in range(len(t.counts)) if t.counts[o] > 0]
            cnt = [int(t.counts[o]) for o in peers]
            base = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32) if cnt else np.zeros(0, np.int32)
            mine = self.pushbuf.ptr
            self.engine.set_push(cnt, base, list(range(len(peers))), [mine["recvA"]] * len(peers), [mine["recvL"]] * len(peers),
                                 [mine["recvF"]] * len(peers), [mine["flags"]] * len(peers), mine["flags"])
            return
        everyone = [None] * self.world
        dist.all_gather_object(everyone, {"handles": self.pushbuf.handles, "counts": [int(c) for c in t.counts]})
        peers, cnt, base, my_index = push_layout(self.rank, [e["counts"] for e in everyone])
        rA, rL, rF, fl = [], [], [], []
        for o in peers:
            m = self.pushbuf.open_peer(everyone[o]["handles"])
            rA.append(m["recvA"]); rL.append(m["recvL"]); rF.append(m["recvF"]); fl.append(m["flags"])
        self.engine.set_push(cnt, base, my_index, rA, rL, rF, fl, self.pushbuf.ptr["flags"])
        dist.barrier()                                          # nobody stores into a buffer its owner has not finished setting up

    def _open_direct(self, own_engine):
        """grouped ncclSend / ncclRecv on the engine's stream for the per-iteration exchanges (rccl_direct.py) when the process
        group is RCCL and every rank's self-check against all_to_all_single passes; None = the torch collective"""
        torch, dist = self.torch, self.dist
        # self.direct_status: why the per-iteration exchanges travel the way they do (bench.py prints it)
        if self.pushbuf is not None:
            self.direct_status = "off: peer-store transport"
            return None
        if not own_engine or dist.get_backend() != "nccl" or self.device.type != "cuda" or os.environ.get("SMOOTHMESH_EXCHANGE", "rccl") == "torch":
            self.direct_status = "off: " + ("SMOOTHMESH_EXCHANGE=torch" if os.environ.get("SMOOTHMESH_EXCHANGE", "rccl") == "torch" else
                                            f"process group backend {dist.get_backend()}" if own_engine else "stand-in engine")
            return None
        if self.world == 1 and not self.probe_slots:
            self.direct_status = "off: one rank"
            return None
        from . import rccl_direct
        try:
            rccl_direct._find_library(torch)
            have = 1
        except OSError:
            have = 0
        flag = torch.tensor([have], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag.item()):
            self.direct_status = "fell_back: librccl.so not found on some rank"
            return None
        # (the first real multi-GPU run of this path happens outside the builder's reach: a rank on which the second communicator
        # cannot be brought up, or whose self-check raises, must not take the job down -- every rank then agrees on the torch
        # collective instead)
        d, ok = None, 1
        try:
            d = rccl_direct.RcclDirect(torch, dist, self.device)
        except Exception as ex:  # noqa: BLE001
            print(f"[smoothmesh_amd] rank {self.rank}: direct RCCL exchange unavailable ({type(ex).__name__}: {ex}); using all_to_all_single", flush=True)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank has its communicator, or nobody uses one
        counts = [self.probe_slots] if self.world == 1 else self.counts
        if not int(flag.item()):
            self.direct_status = "fell_back: second communicator could not be created on some rank"
        elif d.self_check(counts, self.device):   # (self_check agrees on its result across the ranks itself)
            self.direct_status = "pass"
            return d
        else:
            self.direct_status = "fell_back: self-check against all_to_all_single failed on some rank"
        if d is not None:
            try:
                d.close()
            except Exception:  # noqa: BLE001
                pass
        return None

    def transport_info(self):
        """how the shared-point records travel in this job, for the bench line: {"backend", "transport": direct | torch | push,
        "ranks_seen": the size of the communicator that carries them (ncclCommCount of the send / recv communicator, else the
        process group's world size), "self_check": pass | fell_back: why | off: why}"""
        backend = self.dist.get_backend()
        if self.pushbuf is not None:
            t, seen = "push", self.world
        elif self.direct is not None:
            t, seen = "direct", self.direct.ranks_seen
        else:
            t, seen = "torch", self.dist.get_world_size()
        return {"backend": "RCCL (torch backend nccl)" if backend == "nccl" else backend + " (debug transport, host-staged)", "transport": t,
                "ranks_seen": int(seen), "self_check": getattr(self, "direct_status", "off"), "exchange_stream": bool(getattr(self, "overlap", False)),
                "had_error": getattr(self, "had_error", None)}

    def close(self):
        """Orderly shutdown, to be called on every rank BEFORE dist.barrier() / destroy_process_group(): drain the engine's and
        the exchange stream (nothing may still be using the second communicator), destroy that communicator, release the
        engine.  Leaving this to __del__ would destroy the communicator after the process group is gone, with work possibly
        still in flight -- an exit-time hang or abort inside ncclCommDestroy on a real multi-GPU run."""
        for st in (getattr(self, "estream", None), getattr(self, "xstream", None)):
            if st is not None:
                st.synchronize()
        d, self.direct = getattr(self, "direct", None), None
        if d is not None:
            d.close()
        pb, self.pushbuf = getattr(self, "pushbuf", None), None
        if pb is not None:
            if self.world > 1:
                self.dist.barrier()       # every rank has stopped storing into its peers' buffers
            try:
                self.engine.clear_push()
            except Exception:             # noqa: BLE001 -- engine already closed
                pass
            pb.close()
        eng = getattr(self, "engine", None)
        if eng is not None and hasattr(eng, "close"):
            eng.close()

    def global_min_edge(self):
        """getMeshStats + returnReduce(minOp), SM.C:1527"""
        dev = "cpu" if self._staged() else self.device
        t = self.torch.tensor([self.engine.mesh_stats()[0]], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def set_params(self, p):
        self.engine.set_params(p)

    def set_sync_variant(self, variant):
        """"master" (default) / "own": the syncTools::syncPointList model of the magnitude folds, in the engine's combines and in
        the host-side combines of the set-ups (include/smgpu.h smgpu_set_sync_variant)"""
        self.own_fold = {"master": False, "own": True}[variant]
        self.engine.set_sync_variant(variant)

    def set_layers(self, lp, minEdgeLength):
        """boundary layer treatment (-layerPatches under mpirun): step-wise set-up, the reference's syncPointList calls
        done with the same all_to_all as the per-iteration exchanges (values travel as doubles; set-up only)"""
        torch, t = self.torch, self.tables
        cpu = self._staged() or self.device.type == "cpu"

        def exchange(own, op):
            o = own[0]
            k = o.shape[1]
            send = torch.from_numpy(np.ascontiguousarray(o[t.sendShared])).reshape(-1, k) if t.nSend else torch.zeros((0, k), dtype=torch.float64)
            recv = torch.zeros((t.nRecv, k), dtype=torch.float64)
            if self.world > 1:
                if cpu:
                    self.dist.all_to_all_single(recv, send, self.counts, self.counts)
                else:
                    r = recv.to(self.device)
                    self.dist.all_to_all_single(r, send.to(self.device), self.counts, self.counts)
                    recv = r.cpu()
            return [combine_shared(t, o, recv.numpy(), op, self.own_fold)]
        self.layers = setup_layers_stepwise([self.engine], exchange, lp, minEdgeLength)
        return self.layers

    def set_boundary_smoothing(self, bp, minEdgeLength, layerEdgeLength=None):
        """boundary point smoothing (constant/geometry/*.obj under mpirun): step-wise set-up; the reductions of getMeshStats
        (SM.C:1528-1538) with all_gather_object, the syncPointList calls with the all_to_all of the exchanges.  After
        set_layers when both are used.  Returns this rank's classification summary."""
        torch, t = self.torch, self.tables
        cpu = self._staged() or self.device.type == "cpu"

        def exchange(own, op):
            o = own[0]
            k = o.shape[1]
            send = torch.from_numpy(np.ascontiguousarray(o[t.sendShared])).reshape(-1, k) if t.nSend else torch.zeros((0, k), dtype=torch.float64)
            recv = torch.zeros((t.nRecv, k), dtype=torch.float64)
            if self.world > 1:
                if cpu:
                    self.dist.all_to_all_single(recv, send, self.counts, self.counts)
                else:
                    r = recv.to(self.device)
                    self.dist.all_to_all_single(r, send.to(self.device), self.counts, self.counts)
                    recv = r.cpu()
            return [combine_shared(t, o, recv.numpy(), op, self.own_fold)]

        def reduce_stats(stats):
            mn, bb = stats[0]
            allv = [None] * self.world
            self.dist.all_gather_object(allv, (float(mn), [float(x) for x in bb]))
            b = np.array([v[1] for v in allv])
            return min(v[0] for v in allv), np.array([b[:, 0].min(), b[:, 1].max(), b[:, 2].min(), b[:, 3].max(), b[:, 4].min(), b[:, 5].max()])
        info = setup_boundary_stepwise([self.engine], exchange, reduce_stats, bp, minEdgeLength, layerEdgeLength)[0]
        self.boundary = bool(info["enabled"])
        return info

    def _staged(self):
        # RCCL moves device buffers directly; gloo (CPU tests, or several debug ranks sharing one GPU)
        # cannot, so device tensors are staged through the host for it
        return self.dist.get_backend() == "gloo" and self.device.type != "cpu"

    def _a2a(self, recv, send, overlap=None):
        """all_to_all of the packed shared-point records on torch's current stream (= the engine's exchange
        stream), then `overlap` (smgpu_iter_interior / smgpu_iter_ahead: the kernels that do not need the
        exchanged data) on the engine's own stream.  The engine orders the two streams with events inside
        smgpu_iter_begin/mid/end, so compute never leaves its queue and the host only pays for the collective
        call (~17 us on this stack; async_op=True + wait() costs ~70 us, torch stream contexts ~40 us)."""
        if self.pushbuf is not None:      # peer-store transport: the kernels have moved the records themselves
            if overlap:
                overlap()
            return
        n = self.tables.nSend
        counts = self.counts
        if self.world == 1 and self.probe_slots:
            # measurement aid (bench.py, SMOOTHMESH_FORCE_DIST=1): a self-exchange of probe_slots dummy records
            # drives the same stream/event/collective machinery a real N>1 run uses
            n = self.probe_slots
            recv, send = self._probe[recv.dtype]
            counts = [n]
        elif self.world == 1:
            n = 0
        # a rank without shared points (a sub-domain that touches no other: n == 0) still takes part in a COLLECTIVE -- the other
        # ranks' all_to_all_single would wait for it for ever; the send / recv groups are pairwise, there it has nothing to do
        collective = self.world > 1 and self.direct is None
        if (n or collective) and self._staged():
            r = self.torch.empty_like(recv[:n], device="cpu")
            self.dist.all_to_all_single(r, send[:n].cpu(), counts, counts)
            recv[:n].copy_(r)
        elif n and self.direct is not None:
            self.direct.exchange(recv.data_ptr(), send.data_ptr(), counts, (recv.numel() // recv.shape[0]) * recv.element_size(),
                                 self.torch.cuda.current_stream(self.device).cuda_stream)
        elif n or collective:
            self.dist.all_to_all_single(recv[:n], send[:n], counts, counts)
        if overlap:
            overlap()

    def _a2a_LA(self, overlap=None):
        """exchange L (when the layer treatment or the boundary point smoothing is on) and exchange A: one send / recv group
        with the direct exchange, two collectives otherwise"""
        st, eng = self.state, self.engine
        if self.pushbuf is not None:
            if overlap:
                overlap()
            return
        withL = self.layers or self.boundary
        if withL and self.direct is not None:
            probe = self.world == 1 and self.probe_slots
            n = self.probe_slots if probe else self.tables.nSend
            if n or self.tables.nRecv:
                if probe:      # measurement aid: a self-exchange of dummy records of the same sizes
                    (ra, sa), (rl, sl) = self._probe[self.torch.float64], self._probe["L"]
                    rl, sl = l_view(rl, eng.l_doubles()), l_view(sl, eng.l_doubles())
                else:
                    ra, sa = st.recvA, st.sendA
                    rl, sl = l_view(st.recvL, eng.l_doubles()), l_view(st.sendL, eng.l_doubles())
                self.direct.exchange_many([(rl.data_ptr(), sl.data_ptr(), (rl.numel() // rl.shape[0]) * rl.element_size()),
                                           (ra.data_ptr(), sa.data_ptr(), (ra.numel() // ra.shape[0]) * ra.element_size())],
                                          [n] if probe else self.counts, self.torch.cuda.current_stream(self.device).cuda_stream)
            if overlap:
                overlap()
            return
        if withL:
            self._a2a(l_view(st.recvL, eng.l_doubles()), l_view(st.sendL, eng.l_doubles()))   # OBB.C:184-198, 490-496
        self._a2a(st.recvA, st.sendA, overlap)   # SM.C:134-148, 402-478

    def _agree_on_error(self):
        """Wait for the engine's stream, read its error word (include/smgpu.h) and let EVERY rank raise when any rank has one: a
        rank raising on its own would leave the others inside the next collective.  The engine clears the word once it has been
        reported, so the handle keeps a sticky `had_error` for later readers (transport_info, bench.py)."""
        torch, eng = self.torch, self.engine
        err = None
        if hasattr(eng, "check_error"):
            try:
                eng.check_error()
            except Exception as ex:  # noqa: BLE001 -- re-raised below, on every rank
                err = ex
        if self.world > 1:
            flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device="cpu" if self._staged() else self.device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            if int(flag.item()) and err is None:
                err = RuntimeError("another rank reported a device error in this call (see its message)")
        if err is not None:
            self.had_error = str(err)
            raise err

    def _gather_stats(self):
        if self._staged():
            out = self.torch.empty((self.world, 2), dtype=self.torch.float64)
            self.dist.all_gather_into_tensor(out.view(-1), self.state.localStats.cpu())
            self.allStats.copy_(out)
        else:
            self.dist.all_gather_into_tensor(self.allStats.view(-1), self.state.localStats)

    def set_overlap(self, overlap):
        """overlap=True: exchanges run on a communication stream next to the engine's exchange-independent
        kernels (ordered with events inside the engine); False: exchanges run in order on the engine's stream.
        Which one is faster depends on the exchange latency of the machine (each cross-stream dependency costs
        ~10 us of iteration time on MI355X, an exposed exchange its full latency): see autotune()."""
        if self.xstream is None or self.pushbuf is not None:   # (peer stores: nothing is enqueued by the host, no second stream)
            return
        self.overlap = bool(overlap)
        self.engine.set_exchange_stream(self.xstream.cuda_stream if self.overlap else None)

    def autotune(self, iters=20, allow_overlap=True):
        """Time `iters` iterations in both arrangements (max over ranks), keep the faster one for all ranks and
        restore the coordinates.  Returns {"overlap": bool, "us_per_iter": {...}}.  allow_overlap=False: the exchange-stream
        arrangement is not a candidate (bench.py: its check against the oracle did not pass on this machine)."""
        import time
        if self.xstream is None or self.pushbuf is not None or (self.world == 1 and not self.probe_slots):
            return {"overlap": getattr(self, "overlap", False), "us_per_iter": {}}
        torch = self.torch
        pts0 = self.engine.get_points()
        timing = {}
        for mode in ((False, True) if allow_overlap else (False,)):
            self.set_overlap(mode)
            self.iterate(3, 0.0)
            torch.cuda.synchronize(self.device)
            self.dist.barrier()
            t0 = time.perf_counter()
            self.iterate(iters, 0.0)
            torch.cuda.synchronize(self.device)
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if self._staged() else self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            timing[mode] = float(t.item()) / iters * 1e6
            self.engine.set_points(pts0)
        best = min(timing, key=timing.get)
        self.set_overlap(best)
        return {"overlap": best, "us_per_iter": {"inorder": timing[False], **({"overlap": timing[True]} if True in timing else {})}}

    def iterate(self, centroidalIters, relTol=0.02):
        if self.xstream is None:
            return self._iterate(centroidalIters, relTol)
        # torch's current stream for the loop: the communication stream (overlap) or the engine's own stream
        # (in order).  Never the HIP null stream: operations on it synchronise with every blocking stream.
        torch = self.torch
        loop = self.xstream if self.overlap else self.estream
        loop.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(loop):
            out = self._iterate(centroidalIters, relTol)
        torch.cuda.current_stream(self.device).wait_stream(loop)
        return out

    def _iterate(self, centroidalIters, relTol):
        torch, st, eng = self.torch, self.state, self.engine
        n = max(centroidalIters, 1)
        done = 0
        if self.pushbuf is not None and self.world > 1:
            # peer stores: the consuming kernels wait for their peers' flags for a BOUNDED time, so the ranks start a call together
            # (host work between two calls -- mesh output, an oracle leg -- may differ by seconds from rank to rank)
            self.dist.barrier()
        if relTol > 0.0:
            hist = torch.zeros((n, 2), dtype=torch.float64, device=self.device)
            for i in range(centroidalIters):
                eng.iter_begin()
                self._a2a_LA(eng.iter_interior)
                eng.iter_mid()
                self._a2a(st.recvF, st.sendF, eng.iter_ahead)   # SM.C:2374
                eng.iter_end()
                self._agree_on_error()                      # (this branch reads the residual on the host every iteration anyway)
                self._gather_stats()                        # SM.C:1567, 2396
                hist[i, 0] = self.allStats[:, 0].max()
                hist[i, 1] = self.allStats[:, 1].sum()
                done += 1
                if float(hist[i, 0].item()) < relTol:       # SM.C:2401
                    break
            h = hist[:done].cpu().numpy()
            return done, h[:, 0].copy(), h[:, 1].astype(np.int64)
        # residual >= 0, so relTol <= 0 can never stop the loop (SM.C:2401): no per-iteration reduction and no
        # host read-back; every rank keeps its local {residual, nFrozenPoints} history on the device and the
        # all-rank values of the log line (SM.C:2396) come from ONE gather after the loop
        local = torch.zeros((n, 2), dtype=torch.float64, device=self.device)
        eng.set_stats_history(local.data_ptr(), n)      # iter_end fills record i: no per-iteration copy
        for i in range(centroidalIters):
            eng.iter_begin()
            self._a2a_LA(eng.iter_interior)
            eng.iter_mid()
            self._a2a(st.recvF, st.sendF, eng.iter_ahead)
            eng.iter_end()
            done += 1
        eng.set_stats_history(None, 0)
        # an error word raised by a kernel of this call (a peer's records that never came, a point without usable neighbours): every
        # rank must learn of it BEFORE the gather below -- a rank that raised on its own would leave the others inside the collective
        self._agree_on_error()
        if self._staged():
            allh = torch.empty((self.world, n, 2), dtype=torch.float64)
            self.dist.all_gather_into_tensor(allh.view(-1), local.cpu().view(-1))
        else:
            allh = torch.empty((self.world, n, 2), dtype=torch.float64, device=self.device)
            self.dist.all_gather_into_tensor(allh.view(-1), local.view(-1))
        allh = allh[:, :done].cpu().numpy()
        return done, allh[:, :, 0].max(axis=0), allh[:, :, 1].sum(axis=0).astype(np.int64)

    def get_points(self):
        return self.engine.get_points()


class LocalMultiSmoother:
    """All sub-domains in ONE process on ONE device (engines side by side), the exchange done by
    device-side tensor indexing instead of RCCL.  Exercises every device-side piece of the multi-rank
    path (pack / combine / or kernels, slot tables) on a single GPU."""

    def __init__(self, subs, device=0, engine_factory=None, torch_device=None, overlap=True):
        import torch
        self.torch = torch
        self.subs = subs
        self.own_fold = own_fold_default()
        if torch_device is None:
            torch_device = torch.device("cuda", device)
        self.device = torch_device
        cands = [s.processor_patch_point_lists() for s in subs]
        self.states = []
        for s in subs:
            t = HaloTables(s.rank, s.pointProcAddressing, cands)
            if engine_factory is None:
                from .engine import SmoothEngine
                cur = torch.cuda.current_stream(torch_device).cuda_stream
                eng = SmoothEngine(s.mesh, device=device) if overlap else SmoothEngine(s.mesh, device=device, stream=cur)
                if overlap:      # the sub-domains' persistent walk replays run side by side on one device: each takes its share
                    eng.set_device_share(len(subs))
                xs = cur if overlap else None
            else:
                eng, xs = engine_factory(s.mesh), None
            self.states.append(_RankState(s, t, eng, torch, torch_device, xs))
        # (src rank, src offset, dst offset, count) copies implementing the all_to_all
        self.copies = []
        n = len(subs)
        offs = [np.concatenate([[0], np.cumsum(st.t.counts)]) for st in self.states]
        for a in range(n):
            for b in range(n):
                c = int(self.states[a].t.counts[b])
                if c:
                    self.copies.append((a, int(offs[a][b]), b, int(offs[b][a]), c))

    def global_min_edge(self):
        return min(st.eng.mesh_stats()[0] for st in self.states)

    def set_params(self, p):
        for st in self.states:
            st.eng.set_params(p)

    def set_sync_variant(self, variant):
        """"master" (default) / "own": see DistributedSmoother.set_sync_variant"""
        self.own_fold = {"master": False, "own": True}[variant]
        for st in self.states:
            st.eng.set_sync_variant(variant)

    def _exchange(self, which):
        for a, so, b, do, c in self.copies:
            src = getattr(self.states[a], "send" + which)
            dst = getattr(self.states[b], "recv" + which)
            if which == "L":     # records of the width in use, packed from the start of the buffers
                w = self.states[a].eng.l_doubles()
                src, dst = l_view(src, w), l_view(dst, w)
            dst[do:do + c].copy_(src[so:so + c])

    def set_layers(self, lp, minEdgeLength):
        """boundary layer treatment on all sub-domains (the reference under mpirun with -layerPatches)"""
        def exchange(own, op):
            # recv slots of rank b from rank a = a's values at a's send slots towards b
            recv = [np.zeros((st.t.nRecv, own[0].shape[1])) for st in self.states]
            for a, so, b, do, c in self.copies:
                recv[b][do:do + c] = own[a][self.states[a].t.sendShared[so:so + c]]
            return [combine_shared(st.t, o, r, op, self.own_fold) for st, o, r in zip(self.states, own, recv)]
        self.layers = setup_layers_stepwise([st.eng for st in self.states], exchange, lp, minEdgeLength)
        return self.layers

    def set_boundary_smoothing(self, bp, minEdgeLength, layerEdgeLength=None):
        """boundary point smoothing on all sub-domains (the reference under mpirun with constant/geometry/*.obj); after
        set_layers when both are used"""
        def exchange(own, op):
            recv = [np.zeros((st.t.nRecv, own[0].shape[1])) for st in self.states]
            for a, so, b, do, c in self.copies:
                recv[b][do:do + c] = own[a][self.states[a].t.sendShared[so:so + c]]
            return [combine_shared(st.t, o, r, op, self.own_fold) for st, o, r in zip(self.states, own, recv)]
        def reduce_stats(stats):
            bb = np.array([s[1] for s in stats])
            return min(s[0] for s in stats), np.array([bb[:, 0].min(), bb[:, 1].max(), bb[:, 2].min(), bb[:, 3].max(), bb[:, 4].min(), bb[:, 5].max()])
        infos = setup_boundary_stepwise([st.eng for st in self.states], exchange, reduce_stats, bp, minEdgeLength, layerEdgeLength)
        self.boundary = bool(infos[0]["enabled"])
        return infos

    def iterate(self, centroidalIters, relTol=0.02):
        res, frz = [], []
        for i in range(centroidalIters):
            for st in self.states:
                st.eng.iter_begin()
            for st in self.states:
                st.eng.iter_interior()
            self._exchange("A")
            if getattr(self, "layers", False) or getattr(self, "boundary", False):
                self._exchange("L")
            for st in self.states:
                st.eng.iter_mid()
            for st in self.states:
                st.eng.iter_ahead()
            self._exchange("F")
            for st in self.states:
                st.eng.iter_end()
            allst = self.torch.stack([st.localStats for st in self.states]).cpu().numpy()
            res.append(float(allst[:, 0].max()))
            frz.append(int(allst[:, 1].sum()))
            if res[-1] < relTol:
                break
        return len(res), np.array(res), np.array(frz, dtype=np.int64)

    def get_points(self):
        return [st.eng.get_points() for st in self.states]
