"""One oversized trace split over the GPUs of a node (BASELINE config[4]: "Single Fq12 exponentiation proof, trace height
2^18, 8xMI355X with RCCL FRI fold"; reference workload src/fields/fq12/exp.rs:638-696).

The library (sbn_split_prover_*, include/sbn.h) does the sharded proving and calls back into the two collectives an
`sbn_comm` carries; this module supplies them from `torch.distributed`, one process per GPU:

  * `TorchComm(staged=False)`: the device all-to-all runs on RCCL (backend "nccl" on ROCm) straight on the staging
    tensors, over xGMI; host all-gathers (caps, openings, query rows: KB..MB) run on a gloo side group.
  * `TorchComm(staged=True)`: every block goes device -> host -> gloo send/recv -> device.  Slow, but it works with
    several ranks on ONE GPU (RCCL refuses two ranks per device), which is how the parity test runs on a one-GPU box.

PyTorch is plumbing here: device memory for the staging buffers and the process group.  Per proof and rank the all-to-all
moves (world-1)/world of that rank's LDE columns, once per plane (bytes: DESIGN.md section 5).
"""
import ctypes as C

import numpy as np

from . import api

_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_AGH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)


class _Comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("send_buf", C.c_void_p), ("recv_buf", C.c_void_p),
                ("send_bytes", C.c_uint64), ("recv_bytes", C.c_uint64), ("all_to_all", _A2A), ("all_gather_host", _AGH)]


def exchange_bytes(stark, config, degree_bits, world):
    """(send_bytes, recv_bytes) of device staging memory a rank needs (sbn_split_exchange_bytes)."""
    L = api.lib()
    L.sbn_split_exchange_bytes.argtypes = [C.POINTER(api._AirDesc), C.POINTER(api._Config), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    s, r = C.c_uint64(), C.c_uint64()
    api._check(L.sbn_split_exchange_bytes(C.byref(stark._d), C.byref(config._c), degree_bits, world, C.byref(s), C.byref(r)))
    return s.value, r.value


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
    def all_to_all(self, sends, recvs):
        torch, dist = self.torch, self.dist
        ins = [self.send[o:o + n] for o, n in sends]
        outs = [self.recv[o:o + n] for o, n in recvs]
        self.bytes_sent += sum(n for d, (o, n) in enumerate(sends) if d != self.rank)
        if self.world == 1:
            outs[0].copy_(ins[0])
        elif not self.staged:
            dist.all_to_all(outs, ins, group=self.group)          # RCCL over xGMI, on the staging tensors themselves
        else:
            outs[self.rank].copy_(ins[self.rank])
            host_in = {d: ins[d].cpu() for d in range(self.world) if d != self.rank}
            host_out = {s: torch.empty(recvs[s][1], dtype=torch.uint8) for s in range(self.world) if s != self.rank}
            # pairwise exchange in a fixed order: the lower rank of a pair sends first (gloo send/recv are blocking-safe this way)
            for peer in range(self.world):
                if peer == self.rank:
                    continue
                if self.rank < peer:
                    dist.send(host_in[peer], peer, group=self.host_group)
                    dist.recv(host_out[peer], peer, group=self.host_group)
                else:
                    dist.recv(host_out[peer], peer, group=self.host_group)
                    dist.send(host_in[peer], peer, group=self.host_group)
            for s, t in host_out.items():
                outs[s].copy_(t)
        if self.send.is_cuda:
            torch.cuda.synchronize()
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


class SplitProver:
    """This rank's share of ONE proof over all ranks of `comm` (every rank gets the identical proof).
    Mirrors api.Prover: generate_trace / load_trace, prove, stage_times."""

    def __init__(self, stark, config, degree_bits, staged=False, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.stark, self.config, self.degree_bits = stark, config, degree_bits
        world = dist.get_world_size(group)
        sb, rb = exchange_bytes(stark, config, degree_bits, world)
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._send = torch.empty(sb, dtype=torch.uint8, device=dev)
        self._recv = torch.empty(rb, dtype=torch.uint8, device=dev)
        self.comm = TorchComm(self._send, self._recv, staged=staged, group=group)
        self._err = None

        def a2a(ctx, so, sl, ro, rl):
            try:
                return self.comm.all_to_all(*plan_blocks(so, sl, ro, rl, world))
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
        self._c = _Comm(None, dist.get_rank(group), world, self._send.data_ptr(), self._recv.data_ptr(), sb, rb, self._cb[0], self._cb[1])
        L = api.lib()
        L.sbn_split_prover_create.argtypes = [C.POINTER(api._AirDesc), C.POINTER(api._Config), C.c_uint32, C.POINTER(_Comm), C.POINTER(C.c_void_p)]
        L.sbn_split_prover_destroy.argtypes = [C.c_void_p]
        L.sbn_split_prover_generate_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.sbn_split_prover_load_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.sbn_split_prover_prove.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.sbn_split_prover_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        self._h = C.c_void_p()
        api._check(L.sbn_split_prover_create(C.byref(stark._d), C.byref(config._c), degree_bits, C.byref(self._c), C.byref(self._h)))

    def _checked(self, rc):
        if rc != 0 and self._err is not None:
            e, self._err = self._err, None
            raise e
        api._check(rc)

    def generate_trace(self, ios):
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        pi = np.zeros(self.stark.num_public_inputs, dtype=np.uint64)
        self._checked(api.lib().sbn_split_prover_generate_trace(self._h, api._ptr(ios), ios.shape[0], api._ptr(pi)))
        return pi

    def load_trace(self, trace, public_inputs):
        trace = np.ascontiguousarray(trace, dtype=np.uint64)
        pi = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        self._checked(api.lib().sbn_split_prover_load_trace(self._h, api._ptr(trace), api._ptr(pi), len(pi)))

    def prove(self):
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

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
