"""Particle filter sharded over the GPUs of one node: one process per GPU, particles split into
contiguous global ranges, collectives over RCCL (``torch.distributed`` backend "nccl" on ROCm).

Two exchanges exist (stage calls of include/cssm_pf.h):

**Single-collective exchange** -- every weighted observation of an ordinary series:

1. ``shard_propagate_at``   fused propagate + weight + local fixed-point sums of exp(w - c), c being the
   observation's reference level (known without an exchange)
2. ``shard_boundary_pack``  for every peer one segment: header = the rank's 5 words (S, S2, order key of the local max)
   and the totals of its boundary blocks; for the two adjacent ranks also rows = its first (rank below) or last (rank
   above) ``cap`` particles with their cumulative weights
3. ONE all-to-all (equal split; opt-in: all-to-all-v trimmed to what is read)
4. ``shard_adopt_spec``     global max -> c usable?  global cumulative weights -> ll, ess, end slots and runs of the own
   particles; the received rows expanded to the slots the own particles left open; coverage check

Nothing is read by the host per observation.  A capacity miss is resumed in place (``shard_resume``: that observation's
exchange is redone with four times the capacity, the series carries on); a reference level ruled out by the max
(an outlying observation) repeats the series with every level taken from the global max (below).  With ``DistComm`` over RCCL the library enqueues
kernels AND collectives itself (``cssm_pf_shard_series_rccl``), in stretches after each of which one status word is
read, bounded by a timeout (a rank that never joins a collective surfaces as an RCCL error on every rank, not a hang).

The repetition just mentioned (and the first event of an LGCP series, which has no predecessor to predict its level from --
numerics contract v8) run the same exchange with the level taken from the GLOBAL max: an all-gather of the ranks' 5 words and
``shard_sums`` (sums relative to the level the gathered max selects) precede the all-to-all -- two collectives per
observation, still nothing read by the host.

**Exact exchange** -- forced (``exact=True``; tests, and the stage calls a host may drive itself): all-gather of the sums,
``shard_offspring``, all-to-all of the range sizes, ONE host read of those sizes, all-to-all-v of (d + 1) doubles per
candidate, ``shard_adopt``.

Random variates are keyed by the GLOBAL particle id and every sum is an integer sum, so ll, ess
and the ancestor arrays are bit-identical for 1, 2, 4 and 8 ranks.

The orchestration is written over a list of local shards and a communicator object so that the
same code drives (a) one shard per process over RCCL or gloo (``DistComm``) and (b) several
shards inside one process with the exchanges done by tensor copies (``tests/local_comm.py``: how the stage
kernels are tested for world > 1 on a single GPU; test infrastructure, not part of the package).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _abi
from .model import Model


def shard_bounds(n_global: int, world: int, rank: int):
    """Rank r owns global particles / slots [r*ceil(N/R), min((r+1)*ceil(N/R), N))."""
    per = (n_global + world - 1) // world
    lo = min(rank * per, n_global)
    hi = min(lo + per, n_global)
    return lo, hi - lo


class GpuShard:
    """One shard on one GPU, driven through the cssm_pf_shard_* stage calls."""

    def __init__(self, model: Model, n_global: int, rank: int, world: int, seed: int, device: int,
                 lgcp_precision: int = 0):
        self.lib = _abi.load_library()
        self.rank, self.world = rank, world
        self.first, self.n = shard_bounds(n_global, world, rank)
        if self.n < 1:
            raise ValueError("every rank needs at least one particle")
        self.n_global = n_global
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self._desc = model.descriptor(lgcp_precision)
        self._h = C.c_void_p()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _abi.check(self.lib.cssm_pf_create_shard(self._desc.ptr(), n_global, self.first, self.n, seed & (2**64 - 1),
                                                 device, C.c_void_p(stream), C.byref(self._h)))
        self.d = int(self.lib.cssm_pf_dim(self._h))
        kw = dict(device=self.dev)
        self.sums5 = torch.zeros(5, dtype=torch.int64, **kw)
        self.all_sums = torch.zeros(5 * world, dtype=torch.int64, **kw)
        # [send_first | send_count | recv_count | redo flag], one tensor so that one D2H copy reads all of them
        self.meta = torch.zeros(3 * world + 1, dtype=torch.int64, **kw)
        self.send_first, self.send_count = self.meta[:world], self.meta[world:2 * world]
        self.recv_count, self.redo_flag = self.meta[2 * world:3 * world], self.meta[3 * world:]
        self._bufs = {}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.cssm_pf_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def buffer(self, name: str, n_doubles: int) -> torch.Tensor:
        """A persistent exchange buffer of at least n_doubles (grown geometrically, never per step)."""
        b = self._bufs.get(name)
        if b is None or b.numel() < n_doubles:
            b = torch.empty(max(int(n_doubles * 1.25) + 1024, 1), dtype=torch.float64, device=self.dev)
            self._bufs[name] = b
        return b

    def init(self, t0: float):
        _abi.check(self.lib.cssm_pf_shard_init(self._h, float(t0)))

    def set_option(self, option: int, value: int):
        """cssm_pf_set_option on the shard's handle (CSSM_OPT_RESAMPLER = 2: 0 systematic, 1 stratified -- the same on every rank)."""
        _abi.check(self.lib.cssm_pf_set_option(self._h, int(option), int(value)))

    # ---- a whole series, records resident on the device
    def begin(self, t, y, has):
        t = np.ascontiguousarray(t, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        h = np.ones(len(t), dtype=np.uint8) if has is None else np.ascontiguousarray(has, dtype=np.uint8)
        dp = C.POINTER(C.c_double)
        _abi.check(self.lib.cssm_pf_shard_begin(self._h, t.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                                h.ctypes.data_as(C.POINTER(C.c_uint8)), len(t)))

    def begin_more(self, t, y, has):
        """T MORE observations of the running sharded filter (cssm_pf_shard_continue): no new cloud, the series goes on."""
        t = np.ascontiguousarray(t, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        h = np.ones(len(t), dtype=np.uint8) if has is None else np.ascontiguousarray(has, dtype=np.uint8)
        dp = C.POINTER(C.c_double)
        _abi.check(self.lib.cssm_pf_shard_continue(self._h, t.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                                   h.ctypes.data_as(C.POINTER(C.c_uint8)), len(t)))

    KERNELS = ("k_propagate", "k_tile_sums", "k_offspring", "k_reduce_units", "k_boundary_pack", "k_offspring_expand_spec", "collective")

    def profile(self, enable: bool):
        """HIP events around every kernel and every library-issued collective of the shard's stream (a pass of its own: the
        event records perturb the throughput figure)."""
        _abi.check(self.lib.cssm_pf_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        ms = np.zeros(len(self.KERNELS))
        cnt = np.zeros(len(self.KERNELS), dtype=np.uint64)
        _abi.check(self.lib.cssm_pf_profile_read(self._h, ms.ctypes.data_as(C.POINTER(C.c_double)), cnt.ctypes.data_as(C.POINTER(C.c_uint64))))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KERNELS)}

    def propagate_at(self, s: int, with_sums: bool = True):
        """with_sums=False: the single-collective exchange totals the sums itself (boundary_pack)."""
        _abi.check(self.lib.cssm_pf_shard_propagate_at(self._h, int(s), C.c_void_p(self.sums5.data_ptr()) if with_sums else None))

    # ---- single-collective exchange: sums and boundary particles in one all-to-all
    def spec_segment(self, cap: int) -> int:
        return int(self.lib.cssm_pf_shard_spec_segment(self._h, int(cap)))

    def unit_particles(self) -> int:
        """Particles per unit sum (cssm_pf_shard_unit): capacities are rounded up to a multiple of it."""
        return int(self.lib.cssm_pf_shard_unit(self._h))

    def boundary_pack(self, cap: int, send_buf: torch.Tensor):
        _abi.check(self.lib.cssm_pf_shard_boundary_pack(self._h, self.rank, self.world, int(cap), C.c_void_p(send_buf.data_ptr())))

    def adopt_spec(self, recv_buf: torch.Tensor, cap: int):
        self._recv_keepalive = recv_buf
        _abi.check(self.lib.cssm_pf_shard_adopt_spec(self._h, C.c_void_p(recv_buf.data_ptr()), self.rank, self.world, int(cap)))

    def series_native(self, comm_handle, s_begin: int, s_end: int, weighted: np.ndarray, cap: int,
                      send_buf: torch.Tensor, recv_buf: torch.Tensor, single_collective: int = 1):
        """Observations [s_begin, s_end) with the collectives issued by the library itself (cssm_pf_shard_series_rccl).
        single_collective: 1 = one equal-split all-to-all, 2 = one all-to-all-v (whole segments between adjacent ranks
        only, headers otherwise; world > 2), 3 = the same at any world size (tests)."""
        self._recv_keepalive = recv_buf
        w = np.ascontiguousarray(weighted, dtype=np.uint8)
        _abi.check(self.lib.cssm_pf_shard_series_rccl(self._h, comm_handle, self.rank, self.world, int(s_begin), int(s_end),
                                                      w.ctypes.data_as(C.POINTER(C.c_uint8)), int(cap),
                                                      C.c_void_p(self.sums5.data_ptr()), C.c_void_p(self.all_sums.data_ptr()),
                                                      C.c_void_p(send_buf.data_ptr()), C.c_void_p(recv_buf.data_ptr()),
                                                      int(single_collective)))

    # ---- peer-written exchange (cssm_pf_shard_peer_*): segments written straight into the other ranks' receive windows
    def peer_setup(self, cap: int) -> "_abi.PeerHandle":
        h = _abi.PeerHandle()
        _abi.check(self.lib.cssm_pf_shard_peer_setup(self._h, self.rank, self.world, int(cap), C.byref(h)))
        self._peer_cap = int(cap)
        return h

    def peer_connect(self, handles: Sequence["_abi.PeerHandle"]):
        arr = (_abi.PeerHandle * len(handles))(*handles)
        _abi.check(self.lib.cssm_pf_shard_peer_connect(self._h, arr, len(handles)))
        self._peer_ready = True

    def peer_handshake(self, token: int):
        _abi.check(self.lib.cssm_pf_shard_peer_handshake(self._h, int(token)))

    def peer_probe_stale(self) -> int:
        """Probe words of the last handshake that plain loads read wrongly while system-scope loads read them right (diagnostic)."""
        return int(self.lib.cssm_pf_shard_peer_probe_stale(self._h))

    def peer_ready(self, cap: int) -> bool:
        return bool(getattr(self, "_peer_ready", False)) and getattr(self, "_peer_cap", None) == int(cap)

    def peer_close(self):
        self.lib.cssm_pf_shard_peer_close(self._h)
        self._peer_ready = False

    def pack_peer(self, cap: int):
        _abi.check(self.lib.cssm_pf_shard_pack_peer(self._h, self.rank, self.world, int(cap)))

    def pack_rows_peer(self, cap: int):
        """Second stage of the pack: the rows beyond the eager ones that the neighbours' slots need (behind every shard's ``pack_peer``)."""
        _abi.check(self.lib.cssm_pf_shard_pack_rows_peer(self._h, self.rank, self.world, int(cap)))

    def peer_rows(self):
        """(rows written for neighbours, neighbour segments, segments that needed rows beyond the eager ones) since the windows were set
        up -- diagnostics of the peer-written exchange's pack."""
        r, g, b = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _abi.check(self.lib.cssm_pf_shard_peer_rows(self._h, C.byref(r), C.byref(g), C.byref(b)))
        return int(r.value), int(g.value), int(b.value)

    def adopt_peer(self, cap: int):
        _abi.check(self.lib.cssm_pf_shard_adopt_peer(self._h, self.rank, self.world, int(cap)))

    def series_peer(self, s_begin: int, s_end: int, weighted: np.ndarray, cap: int):
        """Observations [s_begin, s_end) on the peer-written exchange, enqueued by the library (cssm_pf_shard_series_peer)."""
        w = np.ascontiguousarray(weighted, dtype=np.uint8)
        _abi.check(self.lib.cssm_pf_shard_series_peer(self._h, self.rank, self.world, int(s_begin), int(s_end),
                                                      w.ctypes.data_as(C.POINTER(C.c_uint8)), int(cap)))

    def resume(self) -> int:
        """After a capacity miss of the single-collective series: the observation that missed (its propagate is done,
        its resampling is not); the sticky bit is cleared and the handle rewound to that point."""
        k = C.c_uint32()
        _abi.check(self.lib.cssm_pf_shard_resume(self._h, C.byref(k)))
        return int(k.value)

    def resume_level(self) -> int:
        """After an observation whose reference level its max ruled out (sticky bit 4): that observation, NOT yet propagated -- the
        bit is cleared and the handle rewound; the host runs it again with its level from the all-gathered max."""
        k = C.c_uint32()
        _abi.check(self.lib.cssm_pf_shard_resume_level(self._h, C.byref(k)))
        return int(k.value)

    def status(self, T: int):
        """(ll, ess, sticky bits, capacity needed per step) of the series just run."""
        ll, ess, bits = C.c_double(), C.c_int32(), C.c_uint32()
        need = np.zeros(T, dtype=np.uint32)
        _abi.check(self.lib.cssm_pf_shard_status(self._h, C.byref(ll), C.byref(ess), C.byref(bits),
                                                 need.ctypes.data_as(C.POINTER(C.c_uint32)), T))
        return ll.value, ess.value, bits.value, need

    def last_device_us(self) -> float:
        """Device time of the last continued series as of its last status read, first kernel of the call to the status kernel, from the
        GPU's constant clock (cssm_pf_last_device_us)."""
        us = C.c_double()
        _abi.check(self.lib.cssm_pf_last_device_us(self._h, C.byref(us)))
        return us.value

    def wait_stats(self):
        """(exchanges counted, mean us the first offspring block waited for every rank's header words, mean us the first expansion block
        waited for them, mean us it then waited for its neighbours' eager-rows flags) of the peer-written exchanges since the cloud was
        drawn, as of the last status read (cssm_pf_shard_wait_stats: the GPU's own 100 MHz clock)."""
        out = (C.c_uint64 * 4)()
        _abi.check(self.lib.cssm_pf_shard_wait_stats(self._h, out))
        n = int(out[0])
        return (n,) + tuple((float(out[k]) * 1e-2 / n) if n else None for k in (1, 2, 3))

    def want_path(self, on: bool):
        _abi.check(self.lib.cssm_pf_shard_want_path(self._h, 1 if on else 0))

    def get_path(self, T: int) -> np.ndarray:
        """This rank's (T + 1) x d rows of `filter`'s path: the rows whose picked slot it owns, zero bits elsewhere."""
        out = np.zeros((T + 1, self.d))
        _abi.check(self.lib.cssm_pf_shard_get_path(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), T))
        return out

    # -- getIntervals over the shards (cssm_pf_shard_summary_*): the collectives between the stages are the caller's
    def summary_begin(self, interval: float):
        if not hasattr(self, "sm_sums"):
            self.sm_sums = torch.zeros(self.d, dtype=torch.float64, device=self.dev)
            self.sm_hist = torch.zeros((self.d + 1) * 512, dtype=torch.int32, device=self.dev)
        _abi.check(self.lib.cssm_pf_shard_summary_begin(self._h, float(interval), C.c_void_p(self.sm_sums.data_ptr()),
                                                        C.c_void_p(self.sm_hist.data_ptr())))

    def summary_hist(self, shift: int):
        _abi.check(self.lib.cssm_pf_shard_summary_hist(self._h, int(shift), C.c_void_p(self.sm_hist.data_ptr())))

    def summary_pick(self, shift: int):
        _abi.check(self.lib.cssm_pf_shard_summary_pick(self._h, int(shift), C.c_void_p(self.sm_hist.data_ptr())))

    def summary_finish(self) -> dict:
        d = self.d
        m, lo, hi = np.zeros(d), np.zeros(d), np.zeros(d)
        em, el, eu = C.c_double(), C.c_double(), C.c_double()
        _p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        _abi.check(self.lib.cssm_pf_shard_summary_finish(self._h, C.c_void_p(self.sm_sums.data_ptr()), _p(m), _p(lo), _p(hi),
                                                         C.byref(em), C.byref(el), C.byref(eu)))
        return dict(state_mean=m, state_lower=lo, state_upper=hi, eta_of_mean=em.value, eta_lower=el.value, eta_upper=eu.value)

    def propagate(self, t, y, has_obs):
        _abi.check(self.lib.cssm_pf_shard_propagate(self._h, float(t), float(y), int(has_obs),
                                                    C.c_void_p(self.sums5.data_ptr())))

    def sums(self):
        _abi.check(self.lib.cssm_pf_shard_sums(self._h, C.c_void_p(self.all_sums.data_ptr()), self.world,
                                               C.c_void_p(self.sums5.data_ptr())))

    def offspring(self):
        _abi.check(self.lib.cssm_pf_shard_offspring(self._h, C.c_void_p(self.all_sums.data_ptr()), self.rank, self.world,
                                                    C.c_void_p(self.send_first.data_ptr()),
                                                    C.c_void_p(self.send_count.data_ptr()),
                                                    C.c_void_p(self.redo_flag.data_ptr())))

    def pack(self, first_host: np.ndarray, count_host: np.ndarray, send_buf: torch.Tensor):
        """Rows for every OTHER rank, destinations back to back (the own range never travels)."""
        f = np.ascontiguousarray(first_host, dtype=np.int64)
        c = np.ascontiguousarray(count_host, dtype=np.int64)
        _abi.check(self.lib.cssm_pf_shard_pack(self._h, self.world, f.ctypes.data_as(C.POINTER(C.c_int64)),
                                               c.ctypes.data_as(C.POINTER(C.c_int64)), self.rank, C.c_void_p(send_buf.data_ptr())))

    def adopt(self, recv_buf: torch.Tensor, n_low: int, n_high: int, self_first: int, self_count: int):
        self._recv_keepalive = recv_buf
        _abi.check(self.lib.cssm_pf_shard_adopt(self._h, C.c_void_p(recv_buf.data_ptr()), int(n_low), int(n_high),
                                                int(self_first), int(self_count)))

    def result(self):
        ll, ess = C.c_double(), C.c_int32()
        _abi.check(self.lib.cssm_pf_shard_result(self._h, C.byref(ll), C.byref(ess)))
        return ll.value, ess.value

    def particles(self) -> np.ndarray:
        out = np.zeros((self.d, self.n))
        _abi.check(self.lib.cssm_pf_get_particles(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def ancestors_local(self) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.uint32)
        _abi.check(self.lib.cssm_pf_get_ancestors(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out


class DistComm:
    """One shard per process; collectives through torch.distributed (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device   # where the small agreement tensors live (the GPU under RCCL, None = CPU under gloo)

    @property
    def peer(self) -> bool:
        """Whether the ordinary exchanges of a series run on the peer-written protocol (no collective per observation).  GPU ranks
        only; CSSM_SHARD_PEER=0 keeps every exchange on the collectives (RCCL issued by the library, or torch.distributed)."""
        import os
        return self.device is not None and os.environ.get("CSSM_SHARD_PEER", "1") != "0"

    def exchange_handles(self, handles: List) -> List:
        """Every rank's peer handle (bytes), in rank order, on every rank."""
        box = [None] * self.world
        self.dist.all_gather_object(box, bytes(handles[0]), group=self.group)
        return [_abi.PeerHandle.from_buffer_copy(b) for b in box]

    def all_gather(self, outs: List[torch.Tensor], ins: List[torch.Tensor]):
        self.dist.all_gather_into_tensor(outs[0], ins[0], group=self.group)

    def all_to_all_counts(self, outs: List[torch.Tensor], ins: List[torch.Tensor]):
        self.dist.all_to_all_single(outs[0], ins[0], group=self.group)

    def all_to_all_v(self, outs, ins, out_splits, in_splits):
        self.dist.all_to_all_single(outs[0], ins[0], output_split_sizes=out_splits[0], input_split_sizes=in_splits[0],
                                    group=self.group)

    def all_to_all_equal(self, outs, ins):
        self.dist.all_to_all_single(outs[0], ins[0], group=self.group)

    def all_reduce_sum(self, tensors: List[torch.Tensor]):
        self.dist.all_reduce(tensors[0], op=self.dist.ReduceOp.SUM, group=self.group)

    def agree_max(self, values: List[int]) -> int:
        """Max over all ranks of a host integer (a decision every rank must take alike)."""
        x = torch.tensor([int(values[0])], dtype=torch.int64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(x.item())

    def combine_rows(self, arrays: List[np.ndarray]) -> np.ndarray:
        """Rows of which exactly one rank holds non-zero bits -> the same array on every rank, bit for bit: the 32-bit halves of
        the bit patterns are summed (RCCL has no bitwise OR; a sum of doubles could turn -0.0 into +0.0)."""
        bits = np.ascontiguousarray(arrays[0]).view(np.uint64)
        halves = np.stack([(bits >> np.uint64(32)).astype(np.int64), (bits & np.uint64(0xffffffff)).astype(np.int64)])
        x = torch.from_numpy(halves).to(self.device if self.device is not None else "cpu")
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM, group=self.group)
        h = x.cpu().numpy().astype(np.uint64)
        return ((h[0] << np.uint64(32)) | h[1]).view(np.float64).reshape(arrays[0].shape)

    def native_comm(self):
        """An RCCL communicator owned by libcssm_pf (its series loop issues the collectives itself), or None.

        Only under the RCCL backend with one GPU per rank.  Rank 0 creates the id; it travels by object broadcast.
        Set CSSM_SHARD_NATIVE=0 to stay on the torch.distributed collectives."""
        import os
        if getattr(self, "_native", False) is not False:
            return self._native
        self._native = None
        if self.device is None or self.dist.get_backend(self.group) != "nccl" or os.environ.get("CSSM_SHARD_NATIVE", "1") == "0":
            return None
        lib = _abi.load_library()
        ok = torch.tensor([1 if lib.cssm_rccl_available() else 0], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            raise RuntimeError("librccl.so could not be loaded by libcssm_pf on some rank; set CSSM_SHARD_NATIVE=0 to run the collectives "
                               "through torch.distributed instead")
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            _abi.check(lib.cssm_rccl_unique_id(ident))
        box = [ident.raw if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        rc = lib.cssm_rccl_comm_create(ident, self.world, self.rank, self.device.index, C.byref(h))
        why = _abi.last_error() if rc != 0 else ""
        good = torch.tensor([1 if rc == 0 else 0], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(good, op=self.dist.ReduceOp.MIN, group=self.group)
        if int(good.item()) == 0:
            # No silent change of protocol: the torch.distributed collectives cost several host-language calls per observation
            # (a different performance regime), so they are chosen explicitly (CSSM_SHARD_NATIVE=0), never fallen back to.
            if rc == 0:
                lib.cssm_rccl_comm_destroy(h)
            raise RuntimeError("libcssm_pf could not create its RCCL communicator on every rank (rank %d: %s); set CSSM_SHARD_NATIVE=0 to "
                               "run the collectives through torch.distributed instead" % (self.rank, why or "ok here, failed on a peer"))
        self._native = h
        return h

    def barrier(self):
        self.dist.barrier(group=self.group)

    def close(self):
        """Destroy the library-owned RCCL communicator (if one was made); call before destroy_process_group."""
        h = getattr(self, "_native", None)
        if h:
            _abi.load_library().cssm_rccl_comm_destroy(h)
        self._native = None


class ShardedFilter:
    """llFilter over sharded particles (ParticleFilter.scala:137-140 on R GPUs).

    ``shards``: the local shards (one under DistComm, R under LocalComm).
    """

    def __init__(self, shards: Sequence, comm):
        self.shards = list(shards)
        self.comm = comm
        self.d = self.shards[0].d

    def init(self, t0: float):
        for s in self.shards:
            s.init(t0)

    def step(self, t: float, y: Optional[float], has_obs: bool = True, lgcp: bool = False):
        """One observation with the exact (host-read) exchange."""
        yv = 0.0 if y is None else y
        for s in self.shards:
            s.propagate(t, yv, 1 if has_obs else 0)
        if has_obs or lgcp:
            self._resample_exact(lgcp)

    def _resample_exact(self, lgcp: bool = False) -> int:
        """Stages after propagate with the host read of the exchange sizes; returns the largest send count."""
        S, comm, d = self.shards, self.comm, self.d
        W = comm.world

        def resample_stage():
            comm.all_gather([s.all_sums for s in S], [s.sums5 for s in S])
            for s in S:
                s.offspring()
            comm.all_to_all_counts([s.recv_count for s in S], [s.send_count for s in S])
            return [s.meta.cpu().numpy() for s in S]   # the one host read of the step: exchange sizes + redo flag

        # the level of the observation comes from the GLOBAL max: an all-gather of the local maxima (only the max keys of the 5
        # words matter), the sums relative to the level it selects, then the ordinary stage.  (Rounds 1-2 first tried the
        # observation's reference level for non-LGCP models and formed the sums again when the max ruled it out; the kernels
        # that form sums relative to the reference level now keep the weights in place of the log-weights, so the exchange
        # whose level is unknown in advance runs the log-weight kernels throughout.)
        comm.all_gather([s.all_sums for s in S], [s.sums5 for s in S])
        for s in S:
            s.sums()
        metas = resample_stage()
        if any(int(m[3 * W]) for m in metas):
            raise RuntimeError("the exact exchange cannot ask for its sums again")
        firsts = [m[:W] for m in metas]
        scount = [m[W:2 * W] for m in metas]
        rcount = [m[2 * W:3 * W] for m in metas]
        row = d + 1
        # a rank's own range stays in place: it is excluded from the exchange
        sx = [c.copy() for c in scount]
        rx = [c.copy() for c in rcount]
        for s, a, b in zip(S, sx, rx):
            a[s.rank] = 0
            b[s.rank] = 0
        send_bufs = [s.buffer("send", int(c.sum()) * row) for s, c in zip(S, sx)]
        recv_bufs = [s.buffer("recv", int(c.sum()) * row) for s, c in zip(S, rx)]
        for s, f, c, b in zip(S, firsts, scount, send_bufs):
            s.pack(f, c, b)
        if W > 1:
            comm.all_to_all_v([b[: int(c.sum()) * row] for b, c in zip(recv_bufs, rx)],
                              [b[: int(c.sum()) * row] for b, c in zip(send_bufs, sx)],
                              [[int(v) * row for v in c] for c in rx], [[int(v) * row for v in c] for c in sx])
        for s, b, c, f, sc in zip(S, recv_bufs, rx, firsts, scount):
            s.adopt(b, int(c[: s.rank].sum()), int(c[s.rank + 1:].sum()), int(f[s.rank]), int(sc[s.rank]))
        return max([int(c.max()) for c in sx] + [0])

    def _resample_spec(self, cap: int):
        """Stages after propagate with the single-collective exchange: sums (segment headers) and boundary particles
        travel in ONE all-to-all; nothing is read by the host."""
        S, comm = self.shards, self.comm
        n = comm.world * S[0].spec_segment(cap)
        send = [s.buffer("send_spec", n)[:n] for s in S]
        recv = [s.buffer("recv_spec", n)[:n] for s in S]
        for s, b in zip(S, send):
            s.boundary_pack(cap, b)
        comm.all_to_all_equal(recv, send)
        for s, b in zip(S, recv):
            s.adopt_spec(b, cap)

    def _resample_peer(self, cap: int):
        """The same exchange on the peer-written protocol: every shard writes its segments into the other shards' windows, then
        every shard resamples behind the flags (shards of ONE process share a stream: the order of the launches is the order
        of the protocol; one shard per process: the flags are)."""
        for s in self.shards:
            s.pack_peer(cap)          # headers + the eager rows
        for s in self.shards:
            s.pack_rows_peer(cap)     # rows beyond them that the neighbours need (their blocks wait for every shard's header)
        for s in self.shards:
            s.adopt_peer(cap)

    def _ensure_peer(self, cap: int) -> bool:
        """Windows of every local shard set up for `cap` and every rank's mapped by every other (once per capacity)."""
        S, comm = self.shards, self.comm
        if not all(hasattr(s, "peer_setup") for s in S) or getattr(self, "peer_refused", None):
            return False
        if all(s.peer_ready(cap) for s in S):
            return True

        def agreed(stage, action):
            """`action` on every local shard; every rank learns whether it worked everywhere (a rank that failed must not leave the
            others waiting in the next collective).  False: the peer-written exchange is off for this filter, on every rank alike,
            and ``peer_refused`` says why -- the series then run on the collective exchange."""
            err = None
            out = []
            for s in S:
                try:
                    out.append(action(s))
                except Exception as e:          # noqa: BLE001 -- any failure of this rank is a reason for all ranks
                    err = f"{stage}: {e}"
                    out.append(None)
            if comm.agree_max([1 if err else 0] * len(S)):
                self.peer_refused = err or f"{stage}: failed on another rank"
                for s in S:
                    s.peer_close()
                return None
            return out

        mine = agreed("setup", lambda s: s.peer_setup(cap))
        if mine is None:
            return False
        every = comm.exchange_handles(mine)
        if agreed("connect", lambda s: s.peer_connect(every)) is None:
            return False
        comm.barrier()
        if len(S) == 1:       # (several shards of one process share a stream: a handshake kernel of one would wait for the next one's)
            # two rounds with different tokens and payloads (`agreed` ends in a collective: every rank has read round one's words before
            # anybody writes round two's): the second finds a window line that a cache kept from the first
            for rnd in ("handshake", "handshake, second round"):
                self._peer_token = getattr(self, "_peer_token", 0) + 1
                if agreed(rnd, lambda s: s.peer_handshake(self._peer_token)) is None:
                    return False
            self.peer_probe_stale = max(s.peer_probe_stale() for s in S)
        comm.barrier()
        return True

    MIN_CAP = 1024
    # capacity >= CAP_SQRT x sqrt(N_global): the number of particles whose runs cross a rank boundary is the deviation of a
    # cumulative weight from its mean, ~ sqrt(N x (N/ESS - 1)) (bench workload, 2^20 particles per rank: median 660 rows,
    # maximum 6237 / 9879 / 14184 at world 2 / 4 / 8 = 4.3 .. 4.9 sqrt(N), tools/need_probe.py).  An observation that needs
    # more is resumed with four times as much.
    CAP_SQRT = 6.0
    NATIVE_STRETCH = 128     # observations the library enqueues between two looks at the sticky bits
    # library-driven series: 1 = equal-split all-to-all of whole segments (default), 2 = all-to-all-v trimmed to what is read
    # (whole segments between adjacent ranks, the 12 header words between all other pairs).  Mode 2 moves (world - 3) segments
    # fewer per rank and observation but has never run on more than one real GPU: it is opt-in (CSSM_SHARD_TRIM=1) until
    # it has; LocalCommTrimmed (tests/local_comm.py) checks its claim -- nothing but the header of a non-adjacent segment is ever
    # read -- on one GPU.
    SINGLE_MODE = 1
    last_resumes = 0
    last_all_to_all = "equal split"

    def _capacity(self) -> int:
        S, comm = self.shards, self.comm
        n_max = -(-S[0].n_global // comm.world)   # ceil(N / world): the same on every rank, and no count can exceed it
        cap = min(max(self.MIN_CAP, int(self.CAP_SQRT * S[0].n_global ** 0.5)), n_max)
        return self._round_cap(cap, n_max)

    def _round_cap(self, cap: int, n_max: int) -> int:
        """Whole units of sums (1024 particles, or a multiple for large shards): the boundary blocks then line up with the sums
        k_propagate / k_tile_sums formed, and k_boundary_pack takes their totals and prefixes from there."""
        # (derived from n_max, which every rank knows, not asked of the local shard: the LAST rank's shard may be smaller and
        #  split into finer units, and all ranks must arrive at the same capacity.  A unit = ceil(tiles / 1024) tiles of 1024.)
        tiles = -(-n_max // 1024)
        unit = 1024 * (-(-tiles // 1024))
        return min(-(-cap // unit) * unit, n_max) if cap >= 1024 else cap

    def filter(self, t, y, has=None, lgcp: bool = False):
        """(ll, path): `filter` of ParticleFilter.scala:152-158 over the shards -- the log-likelihood and, for the initial cloud
        and after every observation, ONE uniformly picked particle (Resampling.sampleOne; row s + 1 = after observation s).  The
        rank that owns the picked global slot records the state; the ranks' rows are combined once, at the end of the series.
        This is what a BootstrapFilter over several GPUs returns (model/package.scala:24; PMMH, model/PMMH.scala:71)."""
        for s in self.shards:
            s.want_path(True)
        try:
            ll, ess = self.ll_filter(t, y, has, lgcp=lgcp)
            path = self.comm.combine_rows([s.get_path(len(t)) for s in self.shards])
        finally:
            for s in self.shards:
                s.want_path(False)
        return ll, path

    def summary(self, interval: float = 0.975) -> dict:
        """getIntervals (model/ParticleFilter.scala:415-424) of the sharded cloud as it stands (after ``ll_filter`` / ``filter`` /
        ``ll_filter_more``): what NativePf.summary returns for the single-GPU cloud of the same N_global -- the order statistics
        bit for bit, the means to ~1e-13.  Eight radix-selection passes, each one all-reduce of (d + 1) x 512 counters, plus one
        all-reduce of d sums: nothing of the cloud leaves its rank."""
        S, comm = self.shards, self.comm
        for s in S:
            s.summary_begin(interval)
        comm.all_reduce_sum([s.sm_sums for s in S])
        for shift in range(56, -8, -8):
            for s in S:
                s.summary_hist(shift)
            comm.all_reduce_sum([s.sm_hist for s in S])
            for s in S:
                s.summary_pick(shift)
        out = [s.summary_finish() for s in S]
        return out[0]

    def ll_filter_more(self, t, y, has=None, lgcp: bool = False):
        """T MORE observations of the filter ``ll_filter`` (or an earlier ``ll_filter_more``) left running: the sharded
        cssm_pf_ll_filter_more.  Capacity misses are resumed in place as ever, and so is an observation whose reference level the
        max rules out (that one observation is run again from the global max: cssm_pf_shard_resume_level); shards without that
        call (the CPU rehearsal's) cannot repeat a continued series from its start: RuntimeError."""
        return self.ll_filter(t, y, has, lgcp=lgcp, cont=True)

    def ll_filter(self, t, y, has=None, lgcp: bool = False, exact: bool = False, cont: bool = False):
        import os
        t = np.asarray(t, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        T = len(t)
        weighted = np.ones(T, dtype=bool) if (has is None or lgcp) else np.asarray(has, dtype=bool)
        S, comm = self.shards, self.comm
        n_max = -(-S[0].n_global // comm.world)
        # Plans, tried in this order until one completes without a sticky bit:
        #   "ref"   the single-collective exchange, every observation's level being its reference level
        #   "max"   the same exchange with every level taken from the GLOBAL max (an all-gather of the local maxima and
        #           shard_sums precede the all-to-all): LGCP series start here (their level IS the max); others get here
        #           when an outlying observation voided "ref" (bit 4)
        #   "exact" the host-read exchange: slots owned by particles of NON-adjacent ranks (weights so degenerate that no
        #           capacity covers them: bit 8 survives the resumes), or forced
        can_single = all(hasattr(s, "boundary_pack") for s in S)
        plans = ["exact"] if (exact or not can_single) else ["ref", "max", "exact"]
        if cont:
            plans = plans[:1]   # (a continued series cannot be repeated from its start)
        # LGCP (numerics contract v8): the level of an event is predicted from the max of the weighted observation before it, so an
        # LGCP series runs the "ref" plan like any other -- except its FIRST event when nothing precedes it (a new cloud): that one
        # takes its level from the global max (all-gather + shard_sums ahead of its all-to-all), whatever the plan.
        have_level0 = bool(getattr(self, "_have_level", False)) if cont else False
        attempt = 0
        while plans:
            plan = plans.pop(0)
            attempt += 1
            all_exact = plan == "exact"
            from_max = plan == "max"
            first_from_max = lgcp and not have_level0 and not all_exact and not from_max
            for s in S:
                s.begin_more(t, y, has) if cont else s.begin(t, y, has)
            cap = None if all_exact else self._capacity()
            # the ordinary exchanges of the "ref" plan: peer-written where the communicator offers it (no collective at all), else
            # the collectives -- issued by the library itself over RCCL where it can
            peer = (plan == "ref") and bool(getattr(comm, "peer", False)) and self._ensure_peer(cap)
            native = None
            if not all_exact and not peer and len(S) == 1 and hasattr(comm, "native_comm"):
                native = comm.native_comm()
            mode = 2 if (self.SINGLE_MODE == 1 and os.environ.get("CSSM_SHARD_TRIM", "0") == "1") else self.SINGLE_MODE
            k, resumes, redo_exchange, redo_cap, escalated, give_up = 0, 0, False, 0, {}, False
            redo_level, level_redos = False, 0
            # (in place only on the peer-written exchange: its packs are gated ON THE DEVICE while the series holds, and the failed exchange
            #  wrote the OTHER receive window -- the rows the observation is propagated from again are intact.  A collective has moved
            #  its segments into the one receive buffer before any rank knows the verdict: there the series is repeated from the max)
            can_redo_level = (plan == "ref") and bool(peer) and all(hasattr(s, "resume_level") for s in S)
            final_status = []

            def look_for_a_miss():
                """A capacity miss is resumable: the observation that missed was left untouched on every rank (all reach the
                same verdict from the segment headers, so bit 8 needs no agreement) and every kernel behind it returned at
                once.  Its exchange is redone with four times the capacity (again four times if it missed before); the
                series behind it keeps the ordinary capacity."""
                nonlocal k, resumes, redo_exchange, redo_cap, give_up, redo_level, level_redos
                if all_exact or resumes >= 64:
                    return False
                seen = [s.status(T) for s in S]
                bits_seen = max(r[2] for r in seen)
                if bits_seen == 4 and can_redo_level and level_redos < 64:
                    # An observation whose reference level its max ruled out holds the series in place just as a capacity miss does
                    # (bit 4 is a function of the segment headers too: every rank reads it).  That ONE observation is run again from
                    # the global max -- propagated again, its log-weights stored, an all-gather of the maxima ahead of its exchange --
                    # and the series goes on behind it on the ordinary plan: no second pass over the series, and the only way for a
                    # CONTINUED series, whose start is gone.
                    ks = [s.resume_level() for s in S]
                    kf = comm.agree_max([max(ks)] * len(S))
                    if any(v != kf for v in ks):
                        raise RuntimeError("ranks disagree on the observation whose level was ruled out")
                    k = kf
                    level_redos += 1
                    redo_level = True
                    return True
                if bits_seen != 8 or cap >= n_max:
                    if k >= T:
                        final_status.extend(seen)   # (the series is complete: this look at it is the one the results are read from)
                    return False
                ks = [s.resume() for s in S]
                kf = comm.agree_max([max(ks)] * len(S))
                if any(v != kf for v in ks):
                    raise RuntimeError("ranks disagree on the observation whose exchange did not fit")
                if escalated.get(kf, 0) >= n_max:
                    give_up = True                   # even whole shards as boundary blocks do not cover it: the exact exchange
                    return False
                k = kf
                escalated[k] = min(escalated.get(k, cap) * 4, n_max)
                redo_cap = self._round_cap(escalated[k], n_max)
                resumes += 1
                redo_exchange = True
                return True

            while k < T and not give_up:
                if peer and len(S) == 1 and hasattr(comm, "dist") and not redo_exchange and not redo_level and not (first_from_max and k == 0):
                    # one shard per process: the library enqueues propagate, pack and offspring + expansion of a whole stretch
                    kend = min(k + self.NATIVE_STRETCH, T)
                    S[0].series_peer(k, kend, weighted, cap)
                    self.last_all_to_all = "none: segments written into the peers' windows (peer-written exchange)"
                    k = kend
                    look_for_a_miss()
                    continue
                if native is not None and not redo_exchange:
                    # the library enqueues kernels and RCCL collectives itself, one stream, no host-language call per
                    # observation -- in stretches, so that a capacity miss is noticed (one status read per stretch) before
                    # the whole tail has been enqueued in vain
                    fm = from_max or (first_from_max and k == 0) or redo_level
                    kend = k + 1 if (fm and not from_max) else min(k + self.NATIVE_STRETCH, T)
                    redo_level = False
                    nb = comm.world * S[0].spec_segment(cap)
                    S[0].series_native(native, k, kend, weighted, cap, S[0].buffer("send_spec", nb)[:nb],
                                       S[0].buffer("recv_spec", nb)[:nb], single_collective=mode + (4 if fm else 0))
                    self.last_all_to_all = ("ncclAllToAllv: whole segments between adjacent ranks, 12 header words between the others"
                                            if (mode == 3 or (mode == 2 and comm.world > 2)) else "ncclAllToAll, equal split of whole segments")
                    k = kend
                    look_for_a_miss()
                    continue
                fm = from_max or (first_from_max and k == 0) or redo_level
                if not redo_exchange:               # (after a resume the observation is already propagated)
                    for s in S:
                        s.propagate_at(k) if (all_exact or fm) else s.propagate_at(k, with_sums=False)
                    if fm and weighted[k]:
                        comm.all_gather([s.all_sums for s in S], [s.sums5 for s in S])   # only the max keys matter
                        for s in S:
                            s.sums()                # the sums, relative to the level the gathered max selects
                if weighted[k]:
                    if all_exact:
                        self._resample_exact(lgcp)
                    elif peer and not redo_exchange and not fm:
                        self._resample_peer(cap)
                    else:
                        self._resample_spec(redo_cap if redo_exchange else cap)   # (a resumed observation: its enlarged capacity)
                redo_exchange, redo_cap, redo_level = False, 0, False
                k += 1
                if native is None and k == T:
                    look_for_a_miss()                # host-driven series: one look at its end
            self.last_resumes = resumes
            self.last_level_redos = level_redos
            self.last_single = not all_exact
            self.last_from_max = from_max
            self.last_native = native is not None
            self.last_peer = bool(peer)
            if give_up:
                if cont:
                    raise RuntimeError("a continued sharded series met an observation no boundary capacity covers (slots owned by particles of "
                                       "non-adjacent ranks): it needs the exact exchange from its start, which a continued series no longer has")
                plans = ["exact"]
                continue
            res = final_status if len(final_status) == len(S) else [s.status(T) for s in S]
            # bit 4 (an observation's reference level was ruled out by the max) means "again from the max", bit 8 a capacity miss that
            # was not resumed.  Both are functions of the segment headers, which every rank holds after the exchange: every rank reads
            # the same bits, no agreement is needed (an all-reduce and a host wait per call, ~80 us of every 20-observation leg).
            bits = max(r[2] for r in res)
            if bits == 0:
                self.last_cap, self.last_attempts = cap, attempt
                self._have_level = have_level0 or bool(np.any(weighted))
                return res[0][0], res[0][1]
            if cont:
                raise RuntimeError(f"a continued sharded series cannot be repeated under another plan (sticky bits {bits}: "
                                   "4 = an observation's reference level was ruled out by the max, 8 = slots owned by non-adjacent ranks)")
            if (bits & 8) and not (bits & 4):
                plans = ["exact"]                    # (a level problem moves on to "max")
        raise RuntimeError("the exact exchange cannot raise a sticky bit")

    def result(self):
        res = [s.result() for s in self.shards]
        return res[0]
