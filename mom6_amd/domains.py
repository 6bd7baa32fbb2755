"""Horizontal domain decomposition and halo exchange: the part of MOM_domains the hot path needs
(reference: src/framework/MOM_domains.F90:66 MOM_domains_init; config_src/infra/FMS2/MOM_domain_infra.F90:130
MOM_domain_type, :171 pass_var, :660 pass_vector, :1141 do_group_pass; MOM_coms_infra.F90:42-66 sum/min/max
across PEs).

One process per GPU, one tile per process.  The transport is torch.distributed point-to-point
(backend "nccl" = RCCL over xGMI on the GPU node; "gloo" for the CPU tests), not a translation of FMS's MPI
code: every group pass is, per direction, one batch of isend/irecv with the (at most two) neighbours, with
the E/W exchange done before the N/S exchange so that corners arrive without extra messages.  With one tile
in a re-entrant direction the wrap-around is a local copy.

Semantics kept from the reference: data domain = compute domain + halo; symmetric
memory adds one extra column/row of u/v/q points at the SW edge that belongs to the compute domain; halos
beyond a closed edge are left untouched; C-grid vector components are exchanged like scalars at their own
staggering (no tripolar fold here).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _abi
from .grid import Grid


def compute_extent(n_global: int, ndiv: int):
    """Start (0-based) and size of each of `ndiv` blocks of `n_global` points, as even as possible with the
    remainder given to the first blocks.  (FMS mpp_compute_extent, which the reference uses through
    MOM_domains, is not vendored; any partition gives the same answers because halo updates carry no
    arithmetic -- the reference's own test.layout criterion.)"""
    base, rem = divmod(n_global, ndiv)
    sizes = [base + (1 if p < rem else 0) for p in range(ndiv)]
    starts = [sum(sizes[:p]) for p in range(ndiv)]
    return starts, sizes


class Domain:
    """MOM_domain_type for one tile of an (npi x npj) layout of a global (NI x NJ) grid."""

    def __init__(self, NI, NJ, layout=(1, 1), rank=0, halo=4, reentrant_x=True, reentrant_y=False, group=None, self_exchange=False,
                 tripolar_n=False):
        self.NI, self.NJ = int(NI), int(NJ)
        self.npi, self.npj = int(layout[0]), int(layout[1])
        self.nranks = self.npi * self.npj
        self.rank = int(rank)
        self.halo = int(halo)
        self.reentrant_x, self.reentrant_y = bool(reentrant_x), bool(reentrant_y)
        self.group = group
        # self_exchange: a re-entrant direction with ONE tile goes through the exchange with this rank as its own neighbour
        # instead of the local wrap (rehearses the native RCCL path on a single GPU)
        self.self_exchange = bool(self_exchange)
        # self_exchange = "y": only y goes through the exchange; x is the tile's own wrap (what a 1 x N layout does: the rehearsal of
        # the non-blocking passes with their interior / edge-band launches on one GPU)
        self.self_exchange_x = bool(self_exchange) and self_exchange != "y"
        self.pi, self.pj = self.rank % self.npi, self.rank // self.npi          # PE numbering: i fastest
        self.i_starts, self.i_sizes = compute_extent(self.NI, self.npi)
        self.j_starts, self.j_sizes = compute_extent(self.NJ, self.npj)
        self.ni, self.nj = self.i_sizes[self.pi], self.j_sizes[self.pj]
        self.i0, self.j0 = self.i_starts[self.pi], self.j_starts[self.pj]      # global 0-based index of (isc, jsc)
        if min(self.i_sizes) < self.halo or min(self.j_sizes) < self.halo:
            raise ValueError("MOM_domains: a tile is narrower than the halo")
        # TRIPOLAR_N (MOM_domains.F90:189): the fold is the northern edge of the northernmost row of tiles; with tiles that span
        # x (the 1 x N latitude bands of this framework) a tile on the fold is its own neighbour across it
        self.tripolar_n = bool(tripolar_n)
        if self.tripolar_n and (self.npi != 1 or not self.reentrant_x or self.reentrant_y or self.NI % 2):
            raise ValueError("MOM_domains: TRIPOLAR_N needs REENTRANT_X, an even NIGLOBAL and one tile in x (layout 1 x N)")
        self.on_fold = self.tripolar_n and self.pj == self.npj - 1
        self._dg = None          # the DeviceGrid of this tile (DeviceGrid.set_domain): enables the packed exchange
        self._bufs = {}

    # ---- neighbours ---------------------------------------------------------------------------------
    def _nbr(self, dpi, dpj):
        """Rank of the neighbour tile in direction (dpi, dpj), or None across a closed edge."""
        pi, pj = self.pi + dpi, self.pj + dpj
        if pi < 0 or pi >= self.npi:
            if not self.reentrant_x:
                return None
            pi %= self.npi
        if pj < 0 or pj >= self.npj:
            if not self.reentrant_y:
                return None
            pj %= self.npj
        return pj * self.npi + pi

    # ---- tile grid ------------------------------------------------------------------------------------
    def tile_grid(self, gg: Grid) -> Grid:
        """The Grid of this tile, with metrics cut out of the one-tile global grid `gg` (whose halos are
        already wrapped/closed), so every rank holds the same numbers a one-tile run holds there."""
        if gg.ni != self.NI or gg.nj != self.NJ or gg.halo != self.halo:
            raise ValueError("tile_grid: global grid does not match the domain")
        t = Grid(ni=self.ni, nj=self.nj, nk=gg.nk, halo=self.halo,
                 reentrant_x=self.reentrant_x and self.npi == 1 and not self.self_exchange_x,
                 reentrant_y=self.reentrant_y and self.npj == 1 and not self.self_exchange,
                 tripolar_n=self.on_fold, first_direction=gg.first_direction, Angstrom_H=gg.Angstrom_H, H_to_Z=gg.H_to_Z, Z_to_H=gg.Z_to_H,
                 g_Earth=gg.g_Earth, Rho0=gg.Rho0)
        for name, a in gg.metrics.items():
            t.set_metric(name, self.cut(a, gg.pos_of(name)))
        return t

    def cut(self, a, pos):
        """The data-domain window of this tile out of a one-tile global array (numpy or torch), any rank."""
        xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
        ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
        h = self.halo
        sl = (slice(self.j0, self.j0 + self.nj + 2 * h + ys), slice(self.i0, self.i0 + self.ni + 2 * h + xs))
        w = a[(..., *sl)]
        return np.ascontiguousarray(w) if isinstance(w, np.ndarray) else w.contiguous()

    # ---- halo update ----------------------------------------------------------------------------------
    def _ranges(self, pos):
        pos = pos & 3      # (without PASS_SCALAR_PAIR)
        xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
        ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
        return xs, ys

    def _fold(self, f, pos_flags):
        """The halo beyond the tripolar fold of a tile on it: its own northern rows turned by half a turn (oracle/domains.c);
        the components of a vector change sign, the members of a SCALAR_PAIR do not."""
        xs, ys = self._ranges(pos_flags)
        h, nj = self.halo, self.nj
        vec = (pos_flags & 3) in (_abi.POS_U, _abi.POS_V) and not (pos_flags & _abi.PASS_SCALAR_PAIR)
        src = torch.flip(f[..., nj:nj + h, :], dims=(-2, -1))
        f[..., h + nj + ys:h + nj + ys + h, :] = -src if vec else src

    def pass_var(self, fields, positions, halo=None):
        """do_group_pass: fill the halos of `fields` (torch tensors, last two axes (j,i), allocated with this
        tile's data-domain shape) from the neighbours / the tile itself."""
        import torch.distributed as dist
        h = self.halo
        w = h if halo is None else min(int(halo), h)
        backend = dist.get_backend(self.group) if (dist.is_available() and dist.is_initialized()) else None
        stage = backend == "gloo"      # gloo moves host memory only
        if getattr(self, "native", False) and all(f.is_cuda for f in fields):
            return self._dg.halo_update(fields, positions)      # the library's group pass (RCCL)
        if self._dg is not None and len(fields) <= 24 and all(f.is_cuda for f in fields):
            return self.pass_ptrs([f.data_ptr() for f in fields], positions, [1 if f.dim() == 2 else f.shape[0] for f in fields],
                                  w, stage)

        def exchange(direction):
            ops, recvs, keep = [], [], []
            for f, pos in zip(fields, positions):
                xs, ys = self._ranges(pos)
                if direction == "x":
                    n, s, lo_nbr, hi_nbr, ax = self.ni, xs, self._nbr(-1, 0), self._nbr(+1, 0), -1
                    rows = slice(h, h + self.nj + ys)          # compute rows only; the y pass brings the corners
                else:
                    n, s, lo_nbr, hi_nbr, ax = self.nj, ys, self._nbr(0, -1), self._nbr(0, +1), -2
                    rows = slice(None)

                def sl(a, b):
                    idx = [slice(None)] * f.dim()
                    idx[ax] = slice(a, b)
                    if direction == "x":
                        idx[-2] = rows
                    return tuple(idx)

                # low halo [h-w, h) <- low neighbour's [n+h-w, n+h)  ;  high halo [h+n+s, h+n+s+w) <- high neighbour's [h+s, h+s+w)
                lo_halo, hi_halo = sl(h - w, h), sl(h + n + s, h + n + s + w)
                to_hi, to_lo = sl(n + h - w, n + h), sl(h + s, h + s + w)
                if hi_nbr == self.rank or lo_nbr == self.rank:
                    # one tile in a re-entrant direction: local wrap
                    f[hi_halo] = f[to_lo]
                    f[lo_halo] = f[to_hi]
                    continue
                # Messages between one pair of ranks are matched in posting order.  When both neighbours are
                # the same rank (two tiles, re-entrant) the peer's first message is its "to-high" block, which
                # lands in my LOW halo: post [send->hi, recv<-lo, send->lo, recv<-hi].
                plan = [("s", hi_nbr, to_hi), ("r", lo_nbr, lo_halo), ("s", lo_nbr, to_lo), ("r", hi_nbr, hi_halo)]
                for kind, nbr, sl_ in plan:
                    if nbr is None:
                        continue
                    if kind == "s":
                        sbuf = f[sl_].contiguous()
                        if stage and sbuf.is_cuda:
                            sbuf = sbuf.cpu()
                        keep.append(sbuf)
                        ops.append(dist.P2POp(dist.isend, sbuf, nbr, group=self.group))
                    else:
                        rbuf = torch.empty(f[sl_].shape, dtype=f.dtype, device="cpu" if (stage and f.is_cuda) else f.device)
                        ops.append(dist.P2POp(dist.irecv, rbuf, nbr, group=self.group))
                        recvs.append((f, sl_, rbuf))
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
                for f, recv_sl, rbuf in recvs:
                    f[recv_sl] = rbuf.to(f.device)

        exchange("x")
        exchange("y")
        if self.on_fold:
            for f, pos in zip(fields, positions):
                self._fold(f, pos)

    def pass_ptrs(self, ptr_list, positions, nk_list, w=None, stage=None):
        """The group pass for device fields: one message per neighbour and direction.  The library packs the send slabs
        of every field of the group into one buffer (mom6hip_halo_pack), torch.distributed moves the buffers (RCCL
        send/recv over xGMI; staged through the host only for the gloo rehearsal), the library unpacks into the halos.
        A direction with a single tile is the library's own wrap kernel.  Everything is enqueued in stream order on the
        current stream; the plan of a group (ctypes argument blocks, offsets, buffers) is built once and reused."""
        import ctypes as C
        import torch.distributed as dist
        from ._lib import check, lib
        dg, L = self._dg, lib()
        if self.on_fold and self.self_exchange:
            raise ValueError("MOM_domains: the self-exchange rehearsal of a tile on the tripolar fold needs the native (RCCL) domain")
        h, nf = self.halo, len(ptr_list)
        if w is None:
            w = h
        if stage is None:
            stage = dist.get_backend(self.group) == "gloo"
        key = (tuple(ptr_list), tuple(positions), tuple(nk_list), w)
        plan = self._bufs.get(key)
        if plan is None:
            ptrs = (C.c_void_p * nf)(*ptr_list)
            pos = (C.c_int32 * nf)(*positions)
            nks = (C.c_int32 * nf)(*nk_list)
            plan = dict(ptrs=ptrs, pos=pos, nks=nks, dirs={})
            for direction, n, lo, hi, ntile in ((0, self.ni, self._nbr(-1, 0), self._nbr(+1, 0), self.npi),
                                               (1, self.nj, self._nbr(0, -1), self._nbr(0, +1), self.npj)):
                if ntile == 1:
                    continue
                s = [self._ranges(p)[direction] for p in positions]
                a0 = {name: (C.c_int32 * nf)(*v) for name, v in (("to_hi", [n + h - w] * nf), ("to_lo", [h + sf for sf in s]),
                                                                  ("lo_halo", [h - w] * nf), ("hi_halo", [h + n + sf for sf in s]))}
                cnt = C.c_int64(0)
                check(L.mom6hip_halo_pack(dg.handle, ptrs, pos, nks, a0["to_hi"], nf, direction, w, None, 1, C.byref(cnt)),
                      "mom6hip_halo_pack")
                bufs = [torch.empty(cnt.value, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device())) for _ in range(4)]
                plan["dirs"][direction] = dict(lo=lo, hi=hi, a0=a0, bufs=bufs)
            self._bufs[key] = plan
        ptrs, pos, nks = plan["ptrs"], plan["pos"], plan["nks"]

        def local_wrap():
            check(L.mom6hip_halo_update(dg.handle, ptrs, pos, nks, nf), "mom6hip_halo_update")

        def exchange(direction):
            d = plan["dirs"][direction]
            lo, hi, a0 = d["lo"], d["hi"], d["a0"]
            s_hi, s_lo, r_lo, r_hi = d["bufs"]

            def pack(which, buf, do_pack):
                check(L.mom6hip_halo_pack(dg.handle, ptrs, pos, nks, a0[which], nf, direction, w, buf.data_ptr(), do_pack, None),
                      "mom6hip_halo_pack")
            if hi is not None:
                pack("to_hi", s_hi, 1)
            if lo is not None:
                pack("to_lo", s_lo, 1)
            ops, staged = [], []

            def snd(buf, nbr):
                b = buf.cpu() if stage else buf
                staged.append(b); ops.append(dist.P2POp(dist.isend, b, nbr, group=self.group))

            def rcv(buf, nbr):
                b = torch.empty(buf.shape, dtype=buf.dtype) if stage else buf
                staged.append((b, buf)); ops.append(dist.P2POp(dist.irecv, b, nbr, group=self.group))

            # posting order as in pass_var: [send->hi, recv<-lo, send->lo, recv<-hi]
            if hi is not None: snd(s_hi, hi)
            if lo is not None: rcv(r_lo, lo)
            if lo is not None: snd(s_lo, lo)
            if hi is not None: rcv(r_hi, hi)
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
                if stage:
                    for item in staged:
                        if isinstance(item, tuple):
                            item[1].copy_(item[0])
            if lo is not None:
                pack("lo_halo", r_lo, 0)
            if hi is not None:
                pack("hi_halo", r_hi, 0)

        did_local = False
        if self.npi == 1:
            if self.reentrant_x or (self.npj == 1 and self.reentrant_y):
                local_wrap(); did_local = True
        else:
            exchange(0)
        if self.npj == 1:
            if self.reentrant_y and not did_local:
                local_wrap()
        else:
            exchange(1)

    # ---- the native (RCCL) domain of the library -----------------------------------------------------------
    @staticmethod
    def native_available():
        """A LOCAL probe (no collective): can this process load RCCL and draw a unique id?  attach_native is itself collective
        (broadcast, ncclCommInitRank), so every rank must agree on the answer of this probe before any rank enters it."""
        import ctypes as C
        from ._lib import lib
        L = lib()
        L.mom6hip_rccl_get_unique_id.argtypes = [C.c_void_p, C.c_int32]
        uid = np.zeros(128, dtype=np.uint8)
        return L.mom6hip_rccl_get_unique_id(uid.ctypes.data, 128) == 0

    def attach_native(self, dg):
        """mom6hip_domain_init_rccl: the group passes and reductions inside library calls become the library's own RCCL
        exchange on its communication stream (mom6_amd/csrc/domain_rccl.hip).  Rank 0 draws the RCCL unique id, the
        process group (any backend) broadcasts it."""
        import ctypes as C
        from ._lib import check, lib
        L = lib()
        L.mom6hip_rccl_get_unique_id.argtypes = [C.c_void_p, C.c_int32]
        L.mom6hip_domain_init_rccl.argtypes = [C.c_void_p, C.POINTER(_abi.DomainStruct), C.c_void_p, C.c_int32]
        uid = np.zeros(128, dtype=np.uint8)
        if self.rank == 0:
            check(L.mom6hip_rccl_get_unique_id(uid.ctypes.data, 128), "mom6hip_rccl_get_unique_id")
        if self.nranks > 1:
            import torch.distributed as dist
            t = torch.from_numpy(uid)
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=self.group)
            uid = t.cpu().numpy().copy()

        def nb(dpi, dpj, ntile, reentrant):
            if ntile == 1:
                return self.rank if ((self.self_exchange_x if dpi else self.self_exchange) and reentrant) else -1
            r = self._nbr(dpi, dpj)
            return -1 if r is None else r
        dom = _abi.DomainStruct(self.nranks, self.rank, nb(-1, 0, self.npi, self.reentrant_x), nb(+1, 0, self.npi, self.reentrant_x),
                                nb(0, -1, self.npj, self.reentrant_y), nb(0, +1, self.npj, self.reentrant_y))
        check(L.mom6hip_domain_init_rccl(dg.handle, C.byref(dom), uid.ctypes.data, 128), "mom6hip_domain_init_rccl")
        self._dg = dg
        dg.domain = self
        self.native = True

    def exchange_timing(self, enable=True):
        """(ms, passes) of the native exchanges since the last call (mom6hip_domain_exchange_timing)."""
        import ctypes as C
        from ._lib import check, lib
        L = lib()
        L.mom6hip_domain_exchange_timing.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        ms, n = C.c_double(0), C.c_int64(0)
        check(L.mom6hip_domain_exchange_timing(self._dg.handle, int(enable), C.byref(ms), C.byref(n)), "mom6hip_domain_exchange_timing")
        return ms.value, n.value

    # ---- reductions (MOM_coms) -------------------------------------------------------------------------
    def sum_across_PEs(self, t: torch.Tensor):
        import torch.distributed as dist
        if self.nranks > 1:
            backend = dist.get_backend(self.group)
            if backend == "gloo" and t.is_cuda:
                c = t.cpu(); dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group); t.copy_(c)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def min_across_PEs(self, t: torch.Tensor):
        import torch.distributed as dist
        if self.nranks > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return t

    def max_across_PEs(self, t: torch.Tensor):
        import torch.distributed as dist
        if self.nranks > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t
