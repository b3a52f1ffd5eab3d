"""One oversized trace split over the GPUs of a node (BASELINE config[4]: "Single Fq12 exponentiation proof, trace height
2^18, 8xMI355X with RCCL FRI fold"; reference workload src/fields/fq12/exp.rs:638-696).

The library (sbn_split_prover_*, include/sbn.h) does the sharded proving and calls the two collectives an `sbn_comm`
carries.  Transports, one rank per GPU:

  * `RcclComm`: the library's own RCCL transport (sbn_rccl_comm_create, csrc/transport.hip): grouped ncclSend / ncclRecv on
    the prover's streams, no Python in the data path; torch.distributed is used once, to hand rank 0's unique id around.
  * `TorchComm(staged=False)`: the same exchanges through torch.distributed (backend "nccl" = RCCL on ROCm) as a batch of
    isend / irecv on the staging tensors, ordered on the prover's stream (wrapped as a torch ExternalStream); host
    all-gathers (caps, openings, query rows: KB..MB) run on a gloo side group.
  * `TorchComm(staged=True)`: every block goes device -> host -> gloo send/recv -> device, blocking.  Slow, but it works
    with several PROCESSES on ONE GPU (RCCL refuses two ranks per device): the multi-process parity test on a one-GPU box.
  * `LocalGroup`: the ranks are THREADS of this process (sbn_local_comm_create), all on one device or one device each:
    how 8- and 16-rank proofs are tested on a one-GPU box (`prove_local`).

PyTorch is plumbing here: device memory for the staging buffers and the process group.  Per proof and rank the exchanges
move (world-1)/world of that rank's LDE columns, once per plane, one 64-column block per step (bytes: DESIGN.md section 5).
A rank that fails leaves the others waiting in a collective: give the process group a timeout
(`init_process_group(timeout=...)`, as tests/split_worker.py and bench.py do) so that they fail instead of hanging;
`SplitProver.prove` first agrees on a status word, so a rank whose earlier step failed stops all ranks cleanly.
"""
import ctypes as C

import numpy as np

from . import api

_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_AGH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)


class _Comm(C.Structure):
    """sbn_comm (include/sbn.h, ABI 3)."""
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("send_buf", C.c_void_p), ("recv_buf", C.c_void_p), ("send_bytes", C.c_uint64), ("recv_bytes", C.c_uint64),
                ("all_to_all", _A2A), ("all_gather_host", _AGH)]


def exchange_bytes(stark, config, degree_bits, world):
    """(send_bytes, recv_bytes) of device staging memory a rank needs (sbn_split_exchange_bytes)."""
    L = api.lib()
    L.sbn_split_exchange_bytes.argtypes = [C.POINTER(api._AirDesc), C.POINTER(api._Config), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    s, r = C.c_uint64(), C.c_uint64()
    api._check(L.sbn_split_exchange_bytes(C.byref(stark._d), C.byref(config._c), degree_bits, world, C.byref(s), C.byref(r)))
    return s.value, r.value


def own_columns(total, world, rank, block=64):
    """Columns of a `total`-column matrix that rank `rank` owns (prover.hip ColShare: blocks of 64 dealt round-robin)."""
    nblocks = -(-total // block)
    return sum(min(block, total - b * block) for b in range(rank, nblocks, world))


def exchange_bytes_sent(stark, config, degree_bits, world, rank):
    """Bytes one rank sends per proof in the column -> row exchanges of the two commitments (the gathers add < 1 %):
    every other rank gets its rows of this rank's LDE columns, once per plane."""
    m = 2 << degree_bits
    planes = 2 if world >= 4 else 1
    cols = own_columns(stark.num_columns, world, rank) + own_columns(stark.num_permutation_zs(config), world, rank)
    return planes * cols * (m // world) * 8 * (world - 1)


def plan_blocks(send_off, send_len, recv_off, recv_len, world):
    """The callback's four arrays as python lists of (offset, length) pairs (also used by the CPU test of the backends)."""
    return ([(int(send_off[d]), int(send_len[d])) for d in range(world)], [(int(recv_off[s]), int(recv_len[s])) for s in range(world)])


class TorchComm:
    """The collectives of an sbn_comm on torch.distributed.  `send` / `recv` are uint8 tensors (device for the prover;
    the CPU test of this class passes host tensors)."""

    def __init__(self, send, recv, staged=False, group=None, host_group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.send, self.recv, self.staged = send, recv, staged
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # host all-gathers need a CPU-capable group
        if host_group is not None:
            self.host_group = host_group
        elif dist.get_backend(group) == "gloo":
            self.host_group = group
        else:
            self.host_group = dist.new_group(backend="gloo")
        self.bytes_sent = 0

    # -- device blocks -----------------------------------------------------------------------------------------------
    def all_to_all(self, sends, recvs, stream=None):
        """Stream-ordered on `stream` (a hipStream_t as an integer; None on the CPU).  Zero-length blocks are skipped."""
        torch, dist = self.torch, self.dist
        ins = [self.send[o:o + n] for o, n in sends]
        outs = [self.recv[o:o + n] for o, n in recvs]
        self.bytes_sent += sum(n for d, (o, n) in enumerate(sends) if d != self.rank)
        on_gpu = self.send.is_cuda
        ext = torch.cuda.ExternalStream(stream, device=self.send.device) if (on_gpu and stream) else None
        ctx = torch.cuda.stream(ext) if ext is not None else _nullctx()
        with ctx:
            if sends[self.rank][1]:
                outs[self.rank].copy_(ins[self.rank], non_blocking=True)
            if not self.staged:
                # RCCL over xGMI: one batch of point-to-point operations on the staging tensors themselves; the batch waits
                # for the current (= the prover's) stream, and wait() makes that stream wait for the batch -- no host block
                ops = []
                for peer in range(self.world):
                    if peer == self.rank:
                        continue
                    if sends[peer][1]:
                        ops.append(dist.P2POp(dist.isend, ins[peer], peer, group=self.group))
                    if recvs[peer][1]:
                        ops.append(dist.P2POp(dist.irecv, outs[peer], peer, group=self.group))
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
            else:
                if ext is not None:
                    ext.synchronize()                         # blocking transport: the packed blocks must be complete
                host_in = {d: ins[d].cpu() for d in range(self.world) if d != self.rank and sends[d][1]}
                host_out = {s: torch.empty(recvs[s][1], dtype=torch.uint8) for s in range(self.world) if s != self.rank and recvs[s][1]}
                # pairwise exchange in a fixed order: the lower rank of a pair sends first (gloo send/recv are blocking-safe this way)
                for peer in range(self.world):
                    if peer == self.rank:
                        continue
                    first, second = (dist.send, dist.recv) if self.rank < peer else (dist.recv, dist.send)
                    for op in (first, second):
                        buf = host_in.get(peer) if op is dist.send else host_out.get(peer)
                        if buf is not None:
                            op(buf, peer, group=self.host_group)
                for s, t in host_out.items():
                    outs[s].copy_(t)
                if ext is not None:
                    ext.synchronize()
        return 0

    # -- host blocks -------------------------------------------------------------------------------------------------
    def all_gather_host(self, mine):
        """mine: 1-d uint8 numpy array -> [world, len] numpy array."""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            return mine.reshape(1, -1).copy()
        t = torch.from_numpy(np.ascontiguousarray(mine))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t, group=self.host_group)
        return np.stack([o.numpy() for o in outs])


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _bind(L):
    L.sbn_split_prover_create.argtypes = [C.POINTER(api._AirDesc), C.POINTER(api._Config), C.c_uint32, C.POINTER(_Comm), C.POINTER(C.c_void_p)]
    L.sbn_split_prover_destroy.argtypes = [C.c_void_p]
    L.sbn_split_prover_generate_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.sbn_split_prover_load_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.sbn_split_prover_prove.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.sbn_split_prover_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    L.sbn_rccl_unique_id.argtypes = [C.c_void_p]
    L.sbn_rccl_comm_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(_Comm)]
    L.sbn_rccl_comm_destroy.argtypes = [C.POINTER(_Comm)]
    L.sbn_rccl_comm_destroy.restype = None
    L.sbn_local_comm_create.argtypes = [C.c_uint32, C.POINTER(C.c_int), C.c_uint64, C.c_uint64, C.POINTER(_Comm), C.POINTER(C.c_void_p)]
    L.sbn_local_comm_abort.argtypes = [C.c_void_p]
    L.sbn_local_comm_abort.restype = None
    L.sbn_local_comm_destroy.argtypes = [C.c_void_p]
    L.sbn_local_comm_destroy.restype = None
    L.sbn_comm_selftest.argtypes = [C.POINTER(_Comm)]
    return L


class RcclComm:
    """The library's native RCCL transport (sbn_rccl_comm_create): rank 0 draws the unique id, `broadcast_id(bytes) -> bytes`
    hands it to the other ranks (default: torch.distributed broadcast_object_list on `group`).  The transport owns the
    staging buffers; nothing of a proof passes through Python."""

    def __init__(self, send_bytes, recv_bytes, rank, world, broadcast_id=None, group=None):
        L = _bind(api.lib())
        idbuf = (C.c_uint8 * 128)()
        if rank == 0:
            api._check(L.sbn_rccl_unique_id(idbuf))
        if world > 1:
            if broadcast_id is None:
                import torch.distributed as dist
                box = [bytes(idbuf)]
                dist.broadcast_object_list(box, src=0, group=group)
                idbytes = box[0]
            else:
                idbytes = broadcast_id(bytes(idbuf))
            idbuf = (C.c_uint8 * 128).from_buffer_copy(idbytes)
        self.c = _Comm()
        api._check(L.sbn_rccl_comm_create(idbuf, rank, world, send_bytes, recv_bytes, C.byref(self.c)))
        self.rank, self.world, self.bytes_sent = rank, world, 0

    def selftest(self):
        api._check(api.lib().sbn_comm_selftest(C.byref(self.c)))

    def close(self):
        if self.c.ctx:
            api.lib().sbn_rccl_comm_destroy(C.byref(self.c))


class LocalGroup:
    """`world` ranks as threads of this process (sbn_local_comm_create): devices = one device index per rank, or None = all
    ranks on the current device."""

    def __init__(self, world, send_bytes, recv_bytes, devices=None):
        L = _bind(api.lib())
        self.world = world
        self.comms = (_Comm * world)()
        self._h = C.c_void_p()
        devs = (C.c_int * world)(*devices) if devices is not None else None
        api._check(L.sbn_local_comm_create(world, devs, send_bytes, recv_bytes, self.comms, C.byref(self._h)))

    def abort(self):
        if self._h:
            api.lib().sbn_local_comm_abort(self._h)

    def close(self):
        if self._h:
            api.lib().sbn_local_comm_destroy(self._h)
            self._h = C.c_void_p()


class SplitProver:
    """This rank's share of ONE proof over all ranks of `comm` (every rank gets the identical proof).
    Mirrors api.Prover: generate_trace / load_trace, prove, stage_times.
    transport: "torch" (TorchComm; staged=True for the host-staged test form), "rccl" (the library's RCCL transport, id handed
    around with torch.distributed), a ready `RcclComm`, or a ready `_Comm` (a rank of a LocalGroup)."""

    def __init__(self, stark, config, degree_bits, staged=False, device=None, group=None, transport="torch"):
        self.stark, self.config, self.degree_bits = stark, config, degree_bits
        self._err = None
        self._native = None
        L = _bind(api.lib())
        if isinstance(transport, _Comm):
            self._c = transport
            self.comm = None
        elif isinstance(transport, RcclComm):          # the caller's (it closes it)
            self._c = transport.c
            self.comm = transport
        else:
            import torch
            import torch.distributed as dist
            world = dist.get_world_size(group)
            sb, rb = exchange_bytes(stark, config, degree_bits, world)
            if transport == "rccl":
                self._native = self.comm = RcclComm(sb, rb, dist.get_rank(group), world, group=group)
                self._c = self._native.c
            else:
                dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
                self._send = torch.empty(sb, dtype=torch.uint8, device=dev)
                self._recv = torch.empty(rb, dtype=torch.uint8, device=dev)
                self.comm = TorchComm(self._send, self._recv, staged=staged, group=group)

                def a2a(ctx, stream, so, sl, ro, rl):
                    try:
                        sends, recvs = plan_blocks(so, sl, ro, rl, world)
                        return self.comm.all_to_all(sends, recvs, stream=stream)
                    except Exception as e:  # noqa: BLE001 -- an exception must not cross the C boundary
                        self._err = e
                        return 1

                def agh(ctx, send, recv, nbytes):
                    try:
                        mine = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                        out = self.comm.all_gather_host(mine)
                        C.memmove(recv, out.ctypes.data, world * nbytes)
                        return 0
                    except Exception as e:  # noqa: BLE001
                        self._err = e
                        return 1

                self._cb = (_A2A(a2a), _AGH(agh))   # keep the trampolines alive as long as the prover
                self._c = _Comm(C.sizeof(_Comm), 0, None, dist.get_rank(group), world, self._send.data_ptr(), self._recv.data_ptr(), sb, rb, self._cb[0], self._cb[1])
        self._h = C.c_void_p()
        self._status = 0     # first failure of this rank; prove() agrees on it with the other ranks before any exchange
        api._check(L.sbn_split_prover_create(C.byref(stark._d), C.byref(config._c), degree_bits, C.byref(self._c), C.byref(self._h)))

    def _checked(self, rc):
        if rc != 0 and self._err is not None:
            e, self._err = self._err, None
            raise e
        api._check(rc)

    def generate_trace(self, ios):
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        pi = np.zeros(self.stark.num_public_inputs, dtype=np.uint64)
        rc = api.lib().sbn_split_prover_generate_trace(self._h, api._ptr(ios), ios.shape[0], api._ptr(pi))
        self._status = self._status or rc
        self._checked(rc)
        return pi

    def load_trace(self, trace, public_inputs):
        trace = np.ascontiguousarray(trace, dtype=np.uint64)
        pi = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        rc = api.lib().sbn_split_prover_load_trace(self._h, api._ptr(trace), api._ptr(pi), len(pi))
        self._status = self._status or rc
        self._checked(rc)

    def agree(self):
        """All ranks exchange their status word on the host; raises everywhere if any rank has failed so far (a rank that
        walked into the first exchange alone would leave the others hanging in it)."""
        if isinstance(self.comm, TorchComm) and self.comm.world > 1:
            st = self.comm.all_gather_host(np.array([self._status & 0xff], dtype=np.uint8))
            bad = [int(r) for r in range(self.comm.world) if st[r][0]]
            if bad:
                raise api.SbnError(-4, f"split proof abandoned: rank(s) {bad} failed before the first exchange")

    def selftest(self):
        """sbn_comm_selftest on this prover's transport (every rank calls it): a pattern exchange checked on the device; raises
        SbnError naming the receiving rank and the block that differs."""
        L = api.lib()
        L.sbn_comm_selftest.argtypes = [C.POINTER(_Comm)]
        self._checked(L.sbn_comm_selftest(C.byref(self._c)))

    def prove(self):
        self.agree()
        h = C.c_void_p()
        self._checked(api.lib().sbn_split_prover_prove(self._h, C.byref(h)))
        return api._take_proof(h)

    def stage_times(self):
        buf = (C.c_float * 32)()
        k = api.lib().sbn_split_prover_stage_times(self._h, buf, 32)
        return {api.lib().sbn_prover_stage_name(i).decode(): float(buf[i]) for i in range(k)}

    def close(self):
        if self._h:
            api.lib().sbn_split_prover_destroy(self._h)
            self._h = C.c_void_p()
        if self._native is not None:
            self._native.close()
            self._native = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove_local(stark, config, degree_bits, world, ios=None, trace=None, public_inputs=None, devices=None, proofs=1):
    """ONE proof by `world` ranks that are threads of this process (LocalGroup): every rank generates (or loads) the witness,
    proves its share `proofs` times and returns its proof; -> (list of api.Proof per rank, list of stage-time dicts).
    ctypes releases the GIL inside the library, so the ranks really run side by side and meet in the transport's barriers."""
    import threading
    sb, rb = exchange_bytes(stark, config, degree_bits, world)
    grp = LocalGroup(world, sb, rb, devices)
    out, times, errs = [None] * world, [None] * world, [None] * world

    def rank_main(r):
        try:
            if devices is not None:
                api._check(api.lib().sbn_set_thread_device(devices[r]))   # this rank's thread only: the process default stays the caller's
            sp = SplitProver(stark, config, degree_bits, transport=grp.comms[r])
            try:
                if ios is not None:
                    sp.generate_trace(ios)
                else:
                    sp.load_trace(trace, public_inputs)
                for _ in range(proofs):
                    out[r] = sp.prove()
                times[r] = sp.stage_times()
            finally:
                sp.close()
        except Exception as e:  # noqa: BLE001
            errs[r] = e
            grp.abort()          # releases the ranks waiting for this one

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.close()
    for e in errs:
        if e is not None:
            raise e
    return out, times
