"""Host-side mirror of MOM_tracer_advect (reference: src/tracer/MOM_tracer_advect.F90).

Same names, argument meaning and error behaviour as the reference's public procedures
(`tracer_advect_init`, `advect_tracer`, `tracer_advect_end`); the work is done by libmom6hip's
HIP kernels through the C ABI (include/mom6hip.h).  Fields are either numpy arrays (host: the
drop-in path, staged to HBM and back) or torch CUDA tensors (resident in HBM, no copies).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib

_dp = C.POINTER(C.c_double)


class DeviceGrid:
    """A grid uploaded to the GPU (mom6hip_ctx_t): metrics in HBM plus the library's work space."""

    def __init__(self, grid, device=0, stream=None):
        L = lib()
        check(L.mom6hip_init(int(device)), "mom6hip_init")
        self.grid = grid
        self.stream = stream
        self._h = C.c_void_p()
        check(L.mom6hip_grid_create(C.byref(grid.struct()), C.c_void_p(stream or 0), C.byref(self._h)),
              "mom6hip_grid_create")

    @property
    def handle(self):
        if not self._h:
            raise Mom6HipError("DeviceGrid used after close()")
        return self._h

    def sync(self):
        check(lib().mom6hip_sync(self.handle), "mom6hip_sync")

    def set_timing(self, enable=True):
        check(lib().mom6hip_set_timing(self.handle, int(enable)), "mom6hip_set_timing")

    def advect_timing(self):
        t = _abi.AdvectTiming()
        check(lib().mom6hip_advect_get_timing(self.handle, C.byref(t)), "mom6hip_advect_get_timing")
        return t

    def bt_graph_nodes(self):
        """Kernel nodes of the subcycle graph captured last (mom6hip_bt_graph_nodes)."""
        n = C.c_int64(0)
        L = lib()
        L.mom6hip_bt_graph_nodes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        check(L.mom6hip_bt_graph_nodes(self.handle, C.byref(n)), "mom6hip_bt_graph_nodes")
        return int(n.value)

    def graph_node_floor(self, nodes, points, reps=20):
        """Microseconds per kernel node of a replayed hipGraph of `nodes` dependent one-pass kernels over `points` doubles
        (mom6hip_graph_node_floor): the launch floor the barotropic subcycle is quoted against."""
        us = C.c_double(0.0)
        L = lib()
        L.mom6hip_graph_node_floor.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
        check(L.mom6hip_graph_node_floor(self.handle, int(nodes), int(points), int(reps), C.byref(us)), "mom6hip_graph_node_floor")
        return float(us.value)

    def bt_graph_stats(self):
        """(captures, launches) of the hipGraph of btstep's subcycle (mom6hip_bt_graph_stats)."""
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().mom6hip_bt_graph_stats(self.handle, C.byref(a), C.byref(b)), "mom6hip_bt_graph_stats")
        return a.value, b.value

    def debug_poison_passes(self, poison=True, split_rows=False):
        """mom6hip_debug_poison_passes: NaNs into the halos of every non-blocking pass between its start and its completion (a halo read
        too early then shows in the results whatever the timing); split_rows: the interior / edge-band launches around the passes
        even on one tile."""
        L = lib(); L.mom6hip_debug_poison_passes.argtypes = [C.c_void_p, C.c_int32]
        check(L.mom6hip_debug_poison_passes(self.handle, int(bool(poison)) | (2 if split_rows else 0)), "mom6hip_debug_poison_passes")

    def kernel_timing(self, enable):
        """(ms_total, launches) per timing slot since recording was switched on (mom6hip_kernel_timing)."""
        ms = (C.c_double * 4)(); n = (C.c_int64 * 4)()      # MOM6HIP_KT_SLOTS: cont flux x, y, the barotropic subcycle, pgf_face_kernel
        check(lib().mom6hip_kernel_timing(self.handle, int(bool(enable)), ms, n), "mom6hip_kernel_timing")
        return [(ms[q], n[q]) for q in range(4)]

    def halo_update(self, fields, positions):
        """pass_var / pass_vector on this one-tile domain for torch CUDA tensors."""
        n = len(fields)
        ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in fields])
        pos = (C.c_int32 * n)(*positions)
        nks = (C.c_int32 * n)(*[1 if f.dim() == 2 else f.shape[0] for f in fields])
        check(lib().mom6hip_halo_update(self.handle, ptrs, pos, nks, n), "mom6hip_halo_update")

    def overlap_stats(self, reset=False):
        """mom6hip_overlap_stats: (row-split launches, continuity calls in two phases, passes completed before anything else ran, 0)"""
        L = lib(); L.mom6hip_overlap_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]
        out = (C.c_uint64 * 4)()
        check(L.mom6hip_overlap_stats(self.handle, out, int(bool(reset))), "mom6hip_overlap_stats")
        return tuple(int(x) for x in out)

    def start_group_pass(self, fields, positions):
        """start_group_pass (MOM_domain_infra.F90:1141): the halo update of the fields is in flight until complete_group_pass."""
        n = len(fields)
        ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in fields])
        pos = (C.c_int32 * n)(*positions)
        nks = (C.c_int32 * n)(*[1 if f.dim() == 2 else f.shape[0] for f in fields])
        check(lib().mom6hip_start_group_pass(self.handle, ptrs, pos, nks, n), "mom6hip_start_group_pass")

    def complete_group_pass(self):
        check(lib().mom6hip_complete_group_pass(self.handle), "mom6hip_complete_group_pass")

    def set_domain(self, domain):
        """Attach a multi-tile Domain (mom6_amd/domains.py): the group pass and sum_across_PEs that happen
        inside library calls are then done by the domain over torch.distributed."""
        import torch
        HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32)
        SUM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int32)
        grid = self.grid

        def halo(user, fields, pos, nk, n):
            try:
                # device pointers go straight to the packed exchange (one message per neighbour and direction), in stream
                # order on the current stream
                domain.pass_ptrs([int(fields[f] or 0) for f in range(n)], [int(pos[f]) for f in range(n)], [int(nk[f]) for f in range(n)])
                return 0
            except Exception:      # never let an exception cross the C boundary
                import traceback; traceback.print_exc()
                return 1

        def sumf(user, values, n):
            try:
                t = torch.tensor([values[q] for q in range(n)], dtype=torch.int32)
                if domain.nranks > 1:
                    import torch.distributed as dist
                    if dist.get_backend(domain.group) == "nccl":
                        t = t.cuda()
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=domain.group)
                    t = t.cpu()
                for q in range(n):
                    values[q] = int(t[q])
                return 0
            except Exception:
                import traceback; traceback.print_exc()
                return 1

        MINF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)

        def minf(user, values, n):
            try:
                t = torch.tensor([values[q] for q in range(n)], dtype=torch.float64)
                if domain.nranks > 1:
                    import torch.distributed as dist
                    if dist.get_backend(domain.group) == "nccl":
                        t = t.cuda()
                    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=domain.group)
                    t = t.cpu()
                for q in range(n):
                    values[q] = float(t[q])
                return 0
            except Exception:
                import traceback; traceback.print_exc()
                return 1

        self._cb = (HALO(halo), SUM(sumf), MINF(minf))      # keep the thunks alive
        check(lib().mom6hip_set_min_callback(self.handle, C.cast(self._cb[2], C.c_void_p), None), "mom6hip_set_min_callback")
        check(lib().mom6hip_set_domain_callbacks(self.handle, C.cast(self._cb[0], C.c_void_p),
                                                 C.cast(self._cb[1], C.c_void_p), None), "mom6hip_set_domain_callbacks")
        self.domain = domain
        domain._dg = self
        import torch.distributed as dist
        # torch.distributed's RCCL ops wait for the current stream, which is the stream the library launches on (the
        # null stream); the gloo rehearsal stages through the host with blocking copies.  No host synchronisation needed.
        ordered = self._stream_is_current()
        check(lib().mom6hip_set_callback_stream_ordered(self.handle, int(ordered)), "mom6hip_set_callback_stream_ordered")

    def _stream_is_current(self):
        import torch
        return self.stream in (None, 0) and torch.cuda.current_stream().cuda_stream == 0

    def close(self):
        if self._h:
            lib().mom6hip_grid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TracerAdvectCS:
    """tracer_advect_CS (MOM_tracer_advect.F90:30-40)."""

    def __init__(self, dt, scheme="PLM", use_huynh_stencil_bug=False):
        if scheme not in _abi.ADV_SCHEMES:
            # MOM_tracer_advect.F90:1131-1133
            raise Mom6HipError("MOM_tracer_advect, tracer_advect_init: Unknown TRACER_ADVECTION_SCHEME = " + str(scheme))
        self.dt = float(dt)
        self.scheme = scheme
        self.use_huynh_stencil_bug = bool(use_huynh_stencil_bug)

    def struct(self):
        return _abi.TracerAdvectCS(self.dt, _abi.ADV_SCHEMES[self.scheme], int(self.use_huynh_stencil_bug))


def tracer_advect_init(dt, scheme="PLM", use_huynh_stencil_bug=False):
    """tracer_advect_init (MOM_tracer_advect.F90:1090): DT and TRACER_ADVECTION_SCHEME."""
    return TracerAdvectCS(dt, scheme, use_huynh_stencil_bug)


def _ptr_space(a):
    """(address, memspace) of a numpy array or a torch CUDA tensor."""
    if isinstance(a, np.ndarray):
        if a.dtype != np.float64 or not a.flags.c_contiguous:
            raise Mom6HipError("fields must be contiguous float64")
        return a.ctypes.data, _abi.MEM_HOST
    if str(a.dtype) != "torch.float64" or not a.is_contiguous():
        raise Mom6HipError("fields must be contiguous float64")
    if not a.is_cuda:
        raise Mom6HipError("torch fields must live on the GPU (use numpy arrays for host fields)")
    return a.data_ptr(), _abi.MEM_DEVICE


def advect_tracer(h_end, uhtr, vhtr, OBC, dt, G: DeviceGrid, CS: TracerAdvectCS, Reg, x_first_in=None,
                  vol_prev=None, max_iter_in=None, update_vol_prev=None, uhr_out=None, vhr_out=None,
                  conc_underflow=None):
    """advect_tracer(h_end, uhtr, vhtr, OBC, dt, G, GV, US, CS, Reg, x_first_in, vol_prev, max_iter_in,
    update_vol_prev, uhr_out, vhr_out)  -- MOM_tracer_advect.F90:52.

    `Reg` is the list of tracer arrays Reg%Tr(m)%t (updated in place); GV/US are folded into `G`.
    Returns the AdvectStats of the call.
    """
    if CS is None:
        raise Mom6HipError("MOM_tracer_advect: tracer_advect_init must be called before advect_tracer.")
    if Reg is None:
        raise Mom6HipError("MOM_tracer_advect: register_tracer must be called before advect_tracer.")
    # advect_x / advect_y read of an associated OBC only the tracer registries of its segments (segment%tr_Reg, :442-477, :580-627 and
    # their twins): without one on any segment the advection is that of a closed domain
    ntr = len(Reg)
    st = _abi.AdvectStats()
    if ntr == 0:
        return st
    g = G.grid
    spaces = set()
    shapes = {"h": g.shape3(_abi.POS_H), "u": g.shape3(_abi.POS_U), "v": g.shape3(_abi.POS_V)}

    def P(a, kind, name):
        if a is None:
            return None
        if tuple(a.shape) != shapes[kind]:
            raise Mom6HipError(f"advect_tracer: {name} has shape {tuple(a.shape)}, expected {shapes[kind]}")
        p, s = _ptr_space(a)
        spaces.add(s)
        return C.c_void_p(p)

    ph, pu, pv = P(h_end, "h", "h_end"), P(uhtr, "u", "uhtr"), P(vhtr, "v", "vhtr")
    trp = (C.c_void_p * ntr)(*[P(t, "h", f"Reg%Tr({m+1})%t") for m, t in enumerate(Reg)])
    pvol, puo, pvo = P(vol_prev, "h", "vol_prev"), P(uhr_out, "u", "uhr_out"), P(vhr_out, "v", "vhr_out")
    if len(spaces) != 1:
        raise Mom6HipError("advect_tracer: all fields must be in the same memory space")
    cu = None
    if conc_underflow is not None:
        cu = np.ascontiguousarray(conc_underflow, dtype=np.float64)
        if cu.shape != (ntr,):
            raise Mom6HipError("advect_tracer: conc_underflow must have one entry per tracer")
    cs = CS.struct()
    space = spaces.pop()
    obc = None
    if OBC is not None:
        def tres_ptr(a):      # a reservoir of a segment's registry, (IsdB:IedB, jsd:jed, nz) of the segment, in the memory space of the call
            if hasattr(a, "data_ptr"):
                if space != _abi.MEM_DEVICE:
                    raise Mom6HipError("advect_tracer: the tracer reservoirs of the OBC segments must be in the memory space of the fields")
                return a.data_ptr(), a
            if space != _abi.MEM_HOST:
                raise Mom6HipError("advect_tracer: the tracer reservoirs of the OBC segments must be in the memory space of the fields")
            a = np.ascontiguousarray(a, dtype=np.float64)
            return a.ctypes.data, a
        obc = OBC.struct(lambda a: (0, None), tres_ptr)
        for s in OBC.segment:
            for t in (s.tr_Reg or []):
                a = t.get("tres")
                if a is not None and s.on_pe:
                    hi = s.HI
                    want = ((g.nk, hi["jed"] - hi["jsd"] + 1, hi["IedB"] - hi["IsdB"] + 1) if s.is_E_or_W
                            else (g.nk, hi["JedB"] - hi["JsdB"] + 1, hi["ied"] - hi["isd"] + 1))
                    if tuple(a.shape) != want:
                        raise Mom6HipError(f"advect_tracer: a tracer reservoir of an OBC segment has shape {tuple(a.shape)}, expected {want}")
    L = lib()
    L.mom6hip_advect_tracer_obc.argtypes = ([C.c_void_p] * 4 + [C.c_double, C.c_void_p, C.c_void_p, _dp, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p])
    rc = L.mom6hip_advect_tracer_obc(
        G.handle, ph, pu, pv, float(dt), C.byref(cs), trp,
        None if cu is None else cu.ctypes.data_as(_dp), ntr,
        -1 if x_first_in is None else int(bool(x_first_in)), pvol,
        0 if max_iter_in is None else int(max_iter_in), int(bool(update_vol_prev)), puo, pvo,
        None if obc is None else C.byref(obc), space, C.byref(st))
    check(rc, "advect_tracer")
    return st


def tracer_advect_end(CS):
    """tracer_advect_end (MOM_tracer_advect.F90:1153)."""
    return None
