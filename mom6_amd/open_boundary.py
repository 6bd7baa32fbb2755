"""Host-side mirror of the part of MOM_open_boundary (reference: src/core/MOM_open_boundary.F90) that places open-boundary segments:
open_boundary_config (:429), parse_segment_str (:1612), setup_segment_indices (:1211), setup_u_point_obc / setup_v_point_obc (:1333,
:1473) -- the segment strings of MOM_input ("I=N,J=0:N,FLATHER,ORLANSKI") become an ocean_OBC_type with the reference's member names,
and `struct()` hands the library what continuity_PPM reads of it (mom6hip_obc_t, include/mom6hip.h).

Round 4 provides the OBC branches of continuity_PPM and CorAdCalc (continuity(..., OBC=), CorAdCalc(..., OBC, ...)) and, here,
radiation_open_bdry_conds for the normal component (:2196) and open_boundary_zero_normal_flow (:3374); every other operator still
refuses an associated OBC."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError

_ACTIONS_OPEN = ("FLATHER", "ORLANSKI", "OBLIQUE", "GRADIENT")
_ACTIONS = ("FLATHER", "ORLANSKI", "ORLANSKI_TAN", "ORLANSKI_GRAD", "OBLIQUE", "OBLIQUE_TAN", "OBLIQUE_GRAD", "NUDGED", "NUDGED_TAN", "NUDGED_GRAD",
            "GRADIENT", "SIMPLE", "SIMPLE_TAN", "SIMPLE_GRAD")


def _interpret_int_expr(word, imax):      # :1700-1722
    word = word.strip()
    if not word:
        raise Mom6HipError("MOM_open_boundary.F90, parse_segment_str: Parsed string was empty!")
    try:
        if word == "N":
            return imax
        if word[0] == "N":
            return imax + int(word[2:]) if word[1] == "+" else imax - int(word[2:])
        return int(word)
    except (ValueError, IndexError):
        raise Mom6HipError(f"MOM_open_boundary.F90, parse_segment_str: Problem reading value from string '{word}'.")


def parse_segment_str(ni_global, nj_global, segment_str, reentrant=False):
    """parse_segment_str :1612 -> (l, m, n, [action strings]); the first word tells whether l is an I or a J"""
    words = [w.strip() for w in segment_str.replace(" ", "").split(",")]
    if len(words) < 2 or words[0][:2] not in ("I=", "J="):
        raise Mom6HipError(f"MOM_open_boundary.F90, parse_segment_str: String '{segment_str}' must start with 'I=' or 'J='.")
    u_seg = words[0][:2] == "I="
    if words[1][:2] != ("J=" if u_seg else "I="):
        raise Mom6HipError(f"MOM_open_boundary.F90, parse_segment_str: Second word of string '{segment_str}' must start with "
                           f"'{'J=' if u_seg else 'I='}'.")
    l_max, mn_max = (ni_global, nj_global) if u_seg else (nj_global, ni_global)
    l = _interpret_int_expr(words[0][2:], l_max)
    if l < 0 or l > l_max:
        raise Mom6HipError(f"MOM_open_boundary.F90, parse_segment_str: First value from string '{segment_str}' is outside of the physical domain.")
    rng = words[1][2:].split(":")
    if len(rng) != 2:
        raise Mom6HipError(f"MOM_open_boundary.F90, parse_segment_str: Problem reading value from string '{words[1]}'.")
    m, n = _interpret_int_expr(rng[0], mn_max), _interpret_int_expr(rng[1], mn_max)
    lo, hi = (-10, mn_max + 10) if reentrant else (-1, mn_max + 1)
    for v, what in ((m, "Beginning"), (n, "End")):
        if v < lo or v > hi:
            raise Mom6HipError(f"MOM_open_boundary.F90, parse_segment_str: {what} of range in string '{segment_str}' is outside of the physical domain.")
    if n == m:
        raise Mom6HipError(f"MOM_open_boundary.F90, parse_segment_str: Range in string '{segment_str}' must span one cell.")
    return l, m, n, [w for w in words[2:] if w]


class OBC_segment_type:
    """OBC_segment_type :146-263: the members continuity_PPM reads, with the reference's names"""

    def __init__(self):
        for n in ("Flather", "radiation", "radiation_tan", "radiation_grad", "oblique", "oblique_tan", "oblique_grad", "nudged", "nudged_tan",
                  "nudged_grad", "specified", "specified_tan", "specified_grad", "open", "gradient", "on_pe", "is_E_or_W", "is_N_or_S",
                  "is_E_or_W_2", "is_N_or_S_2"):
            setattr(self, n, False)
        self.direction = _abi.OBC_NONE
        self.HI = {}
        self.Is_obc = self.Ie_obc = self.Js_obc = self.Je_obc = 0
        self.normal_trans = None      # (nk, jsd:jed, IsdB:IedB) for E / W, (nk, JsdB:JedB, isd:ied) for N / S (C order = the Fortran layout)
        self.normal_vel = None
        self.tangential_vel = None    # (nk, JsdB:JedB, IsdB:IedB): the corner points along the segment
        self.tangential_grad = None
        self.nudged_normal_vel = None      # the layout of normal_vel
        self.nudged_tangential_vel = self.nudged_tangential_grad = None      # the layout of tangential_vel (NUDGED_TAN, NUDGED_GRAD)
        self.normal_vel_bt = self.SSH = None      # (jsd:jed, IsdB:IedB) | (JsdB:JedB, isd:ied): the external barotropic velocity and surface height
        self.Velocity_nudging_timescale_in = self.Velocity_nudging_timescale_out = 0.0
        # segment%tr_Reg: None, or a list of dicts(ntr_index = 1-based place of the tracer in the registry, tres = the reservoir on the
        # segment's faces in the layout of normal_vel or None, OBC_inflow_conc; for update_segment_tracer_reservoirs t = the external
        # values in the layout of tres, resrv_lfac_in / resrv_lfac_out = 1.0) -- register_segment_tracer :5213
        self.tr_Reg = None
        self.Tr_InvLscale_in = self.Tr_InvLscale_out = 0.0      # OBC_TRACER_RESERVOIR_LENGTH_SCALE_IN / _OUT, inverted (:655-667)


class ocean_OBC_type:
    """ocean_OBC_type :266-386 as open_boundary_config leaves it on one tile.  grid: the tile's Grid; idg_offset / jdg_offset: global
    index = local index + offset (one tile with isd = 1: -halo)."""

    def __init__(self, grid, segment_strs, idg_offset=None, jdg_offset=None, ni_global=None, nj_global=None, **flags):
        g = self.grid = grid
        # OBC_ZERO_VORTICITY, OBC_FREESLIP_VORTICITY, OBC_COMPUTED_VORTICITY, OBC_SPECIFIED_VORTICITY (:470-500), read by CorAdCalc
        # OBC_ZERO_STRAIN, OBC_FREESLIP_STRAIN, OBC_COMPUTED_STRAIN (:492-506), OBC_ZERO_BIHARMONIC (:518), read by horizontal_viscosity
        for n in ("zero_vorticity", "freeslip_vorticity", "computed_vorticity", "specified_vorticity", "zero_strain", "freeslip_strain",
                  "computed_strain", "zero_biharmonic"):
            setattr(self, n, bool(flags.pop(n, False)))
        if (self.zero_strain and self.freeslip_strain) or (self.zero_strain and self.computed_strain) or (self.freeslip_strain and self.computed_strain):
            raise Mom6HipError("MOM_open_boundary.F90, open_boundary_config: Only one of OBC_ZERO_STRAIN, OBC_FREESLIP_STRAIN, "
                               "OBC_COMPUTED_STRAIN and OBC_IMPORTED_STRAIN can be True at once.")      # :507-514
        # OBC_RAD_VEL_WT, OBC_RADIATION_MAX (:629-640) and the restart fields of the radiation (arrays at u / v points, nk layers, in the
        # memory space of the calls; None with gamma_uv >= 1)
        self.gamma_uv, self.rx_max = float(flags.pop("gamma_uv", 0.3)), float(flags.pop("rx_max", 1.0))
        self.rx_normal = self.ry_normal = None
        # what the oblique segments keep between steps (:355-362), the layouts of rx_normal (u) and ry_normal (v)
        self.rx_oblique_u = self.ry_oblique_u = self.cff_normal_u = self.rx_oblique_v = self.ry_oblique_v = self.cff_normal_v = None
        if flags:
            raise Mom6HipError(f"open_boundary_config: unknown option {sorted(flags)}")
        self.idg_offset = -g.halo if idg_offset is None else idg_offset
        self.jdg_offset = -g.halo if jdg_offset is None else jdg_offset
        self.ieg = g.ni if ni_global is None else ni_global
        self.jeg = g.nj if nj_global is None else nj_global
        self.number_of_segments = len(segment_strs)
        for n in ("open_u_BCs_exist_globally", "open_v_BCs_exist_globally", "Flather_u_BCs_exist_globally", "Flather_v_BCs_exist_globally",
                  "oblique_BCs_exist_globally", "nudged_u_BCs_exist_globally", "nudged_v_BCs_exist_globally", "specified_u_BCs_exist_globally",
                  "specified_v_BCs_exist_globally", "radiation_BCs_exist_globally", "OBC_pe"):
            setattr(self, n, False)
        self.segnum_u = np.zeros(g.shape2(_abi.POS_U), dtype=np.int32)      # OBC%segnum_u(IsdB:IedB, jsd:jed) = OBC_NONE
        self.segnum_v = np.zeros(g.shape2(_abi.POS_V), dtype=np.int32)
        self.segment = []
        for l_seg, sstr in enumerate(segment_strs, start=1):      # :560-580
            seg = OBC_segment_type()
            self.segment.append(seg)
            if sstr.replace(" ", "")[:2] == "I=":
                self._setup_u_point_obc(seg, sstr, l_seg)
            else:
                self._setup_v_point_obc(seg, sstr, l_seg)
        self.OBC_pe = any(s.on_pe for s in self.segment)
        self._keep = None

    # hor_index_type of the tile (local numbering)
    def _bounds(self):
        g = self.grid
        return dict(isd=g.isd, ied=g.ied, jsd=g.jsd, jed=g.jed, IsdB=g.isd - 1, IedB=g.ied, JsdB=g.jsd - 1, JedB=g.jed,
                    isc=g.isc, iec=g.iec, jsc=g.jsc, jec=g.jec, IscB=g.isc - 1, IecB=g.iec, JscB=g.jsc - 1, JecB=g.jec)

    def _setup_segment_indices(self, seg, Is_obc, Ie_obc, Js_obc, Je_obc):      # :1211-1331
        IsgB, IegB = (Ie_obc, Is_obc) if Ie_obc < Is_obc else (Is_obc, Ie_obc)
        JsgB, JegB = (Je_obc, Js_obc) if Je_obc < Js_obc else (Js_obc, Je_obc)
        isg = ieg = jsg = jeg = 0
        if Is_obc > Ie_obc:      # northern boundary
            isg, jsg, ieg, jeg = IsgB + 1, JsgB, IegB, JegB
        if Is_obc < Ie_obc:      # southern
            isg, jsg, ieg, jeg = IsgB + 1, JsgB + 1, IegB, JegB + 1
        if Js_obc < Je_obc:      # eastern
            isg, jsg, ieg, jeg = IsgB, JsgB + 1, IegB, JegB
        if Js_obc > Je_obc:      # western
            isg, jsg, ieg, jeg = IsgB + 1, JsgB + 1, IegB + 1, JegB
        H = dict(IsgB=IsgB, IegB=IegB, JsgB=JsgB, JegB=JegB, isg=isg, ieg=ieg, jsg=jsg, jeg=jeg)
        io, jo = self.idg_offset, self.jdg_offset
        IsgB, IegB, isg, ieg = IsgB - io, IegB - io, isg - io, ieg - io
        JsgB, JegB, jsg, jeg = JsgB - jo, JegB - jo, jsg - jo, jeg - jo
        B = self._bounds()
        clip = lambda v, lo, hi: min(max(v, lo), hi)
        H.update(IsdB=clip(IsgB, B["IsdB"], B["IedB"]), IedB=clip(IegB, B["IsdB"], B["IedB"]), isd=clip(isg, B["isd"], B["ied"]),
                 ied=clip(ieg, B["isd"], B["ied"]), IscB=clip(IsgB, B["IscB"], B["IecB"]), IecB=clip(IegB, B["IscB"], B["IecB"]),
                 isc=clip(isg, B["isc"], B["iec"]), iec=clip(ieg, B["isc"], B["iec"]),
                 JsdB=clip(JsgB, B["JsdB"], B["JedB"]), JedB=clip(JegB, B["JsdB"], B["JedB"]), jsd=clip(jsg, B["jsd"], B["jed"]),
                 jed=clip(jeg, B["jsd"], B["jed"]), JscB=clip(JsgB, B["JscB"], B["JecB"]), JecB=clip(JegB, B["JscB"], B["JecB"]),
                 jsc=clip(jsg, B["jsc"], B["jec"]), jec=clip(jeg, B["jsc"], B["jec"]))
        seg.HI = H

    def _actions(self, seg, actions, u_seg):      # :1365-1440 / :1505-1580
        d = "u" if u_seg else "v"
        for a in actions:
            if a not in _ACTIONS:
                raise Mom6HipError(f"MOM_open_boundary.F90, setup_{d}_point_obc: String '{a}' not understood.")
            if a in _ACTIONS_OPEN:
                seg.open = True
                setattr(self, f"open_{d}_BCs_exist_globally", True)
            if a == "FLATHER":
                seg.Flather = True; setattr(self, f"Flather_{d}_BCs_exist_globally", True)
            elif a == "ORLANSKI":
                seg.radiation = True; self.radiation_BCs_exist_globally = True
            elif a == "ORLANSKI_TAN":
                seg.radiation = True; seg.radiation_tan = True; self.radiation_BCs_exist_globally = True
            elif a == "ORLANSKI_GRAD":
                seg.radiation = True; seg.radiation_grad = True
            elif a == "OBLIQUE":
                seg.oblique = True; self.oblique_BCs_exist_globally = True
            elif a == "OBLIQUE_TAN":
                seg.oblique = True; seg.oblique_tan = True; self.oblique_BCs_exist_globally = True
            elif a == "OBLIQUE_GRAD":
                seg.oblique = True; seg.oblique_grad = True
            elif a == "NUDGED":
                seg.nudged = True; setattr(self, f"nudged_{d}_BCs_exist_globally", True)
            elif a == "NUDGED_TAN":
                seg.nudged_tan = True; setattr(self, f"nudged_{d}_BCs_exist_globally", True)
            elif a == "NUDGED_GRAD":
                seg.nudged_grad = True
            elif a == "GRADIENT":
                seg.gradient = True
            elif a == "SIMPLE":
                seg.specified = True; setattr(self, f"specified_{d}_BCs_exist_globally", True)
            elif a == "SIMPLE_TAN":
                seg.specified_tan = True
            elif a == "SIMPLE_GRAD":
                seg.specified_grad = True
        if seg.oblique and seg.radiation:
            raise Mom6HipError(f"MOM_open_boundary.F90, setup_{d}_point_obc: Orlanski and Oblique OBC options cannot be used together on one segment.")

    def _alloc(self, seg):      # allocate_OBC_segment_data :3618: normal_vel, normal_trans on the segment's own index range
        H, nk = seg.HI, self.grid.nk
        if seg.is_E_or_W:
            shp = (nk, H["jed"] - H["jsd"] + 1, H["IedB"] - H["IsdB"] + 1)
        else:
            shp = (nk, H["JedB"] - H["JsdB"] + 1, H["ied"] - H["isd"] + 1)
        seg.normal_trans = np.zeros(shp); seg.normal_vel = np.zeros(shp); seg.nudged_normal_vel = np.zeros(shp)
        seg.normal_vel_bt = np.zeros(shp[1:]); seg.SSH = np.zeros(shp[1:])
        shq = (nk, H["JedB"] - H["JsdB"] + 1, H["IedB"] - H["IsdB"] + 1)
        seg.tangential_vel = np.zeros(shq); seg.tangential_grad = np.zeros(shq)

    def _setup_u_point_obc(self, seg, sstr, l_seg):      # :1333-1471
        g = self.grid
        I_obc, Js_obc, Je_obc, actions = parse_segment_str(self.ieg, self.jeg, sstr, g.reentrant_y)
        self._setup_segment_indices(seg, I_obc, I_obc, Js_obc, Je_obc)
        I_obc -= self.idg_offset; Js_obc -= self.jdg_offset; Je_obc -= self.jdg_offset
        if Je_obc > Js_obc:
            seg.direction = _abi.OBC_DIRECTION_E
        elif Je_obc < Js_obc:
            seg.direction = _abi.OBC_DIRECTION_W
            Js_obc, Je_obc = Je_obc, Js_obc
        self._actions(seg, actions, True)
        seg.is_E_or_W_2 = True
        B = self._bounds()
        if I_obc <= B["IsdB"] + 1 or I_obc >= B["IedB"] - 1:
            return      # boundary is not on tile
        if Je_obc <= B["JsdB"] or Js_obc >= B["JedB"]:
            return      # segment is not on tile
        seg.on_pe = True; seg.is_E_or_W = True
        for j in range(B["jsd"], B["jed"] + 1):
            if Js_obc < j <= Je_obc:
                self.segnum_u[j - g.jsd, I_obc - (g.isd - 1)] = l_seg
        seg.Is_obc = seg.Ie_obc = I_obc; seg.Js_obc, seg.Je_obc = Js_obc, Je_obc
        self._alloc(seg)

    def _setup_v_point_obc(self, seg, sstr, l_seg):      # :1473-1610
        g = self.grid
        J_obc, Is_obc, Ie_obc, actions = parse_segment_str(self.ieg, self.jeg, sstr, g.reentrant_x)
        self._setup_segment_indices(seg, Is_obc, Ie_obc, J_obc, J_obc)
        J_obc -= self.jdg_offset; Is_obc -= self.idg_offset; Ie_obc -= self.idg_offset
        if Ie_obc > Is_obc:
            seg.direction = _abi.OBC_DIRECTION_S
        elif Ie_obc < Is_obc:
            seg.direction = _abi.OBC_DIRECTION_N
            Is_obc, Ie_obc = Ie_obc, Is_obc
        self._actions(seg, actions, False)
        seg.is_N_or_S_2 = True
        B = self._bounds()
        if J_obc <= B["JsdB"] + 1 or J_obc >= B["JedB"] - 1:
            return
        if Ie_obc <= B["IsdB"] or Is_obc >= B["IedB"]:
            return
        seg.on_pe = True; seg.is_N_or_S = True
        for i in range(B["isd"], B["ied"] + 1):
            if Is_obc < i <= Ie_obc:
                self.segnum_v[J_obc - (g.jsd - 1), i - g.isd] = l_seg
        seg.Is_obc, seg.Ie_obc = Is_obc, Ie_obc; seg.Js_obc = seg.Je_obc = J_obc
        self._alloc(seg)

    def cuda(self):
        """the arrays of the OBC and of its segments as CUDA tensors (what a device-resident caller keeps): in place, returns self"""
        import torch
        T = lambda a: a if a is None or hasattr(a, "data_ptr") else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
        for k in ("rx_normal", "ry_normal", "rx_oblique_u", "ry_oblique_u", "cff_normal_u", "rx_oblique_v", "ry_oblique_v", "cff_normal_v"):
            setattr(self, k, T(getattr(self, k)))
        for s in self.segment:
            for k in ("normal_trans", "normal_vel", "tangential_vel", "tangential_grad", "nudged_normal_vel", "normal_vel_bt", "SSH",
                      "nudged_tangential_vel", "nudged_tangential_grad"):
                setattr(s, k, T(getattr(s, k)))
            for t in (s.tr_Reg or []):
                for k in ("tres", "t"):
                    if t.get(k) is not None:
                        t[k] = T(t[k])
        return self

    def struct(self, to_ptr=None, tres_ptr=None):
        """mom6hip_obc_t for a call.  to_ptr(array) -> address of a segment's data array in the memory space of the call (default: the numpy
        array itself: HOST); tres_ptr: the same for the tracer reservoirs of the registries (default: to_ptr); the struct keeps what it
        points at alive."""
        tres_ptr = tres_ptr or to_ptr
        segs = (_abi.ObcSegment * max(self.number_of_segments, 1))()
        keep = [segs, self.segnum_u, self.segnum_v]
        for n, s in enumerate(self.segment):
            c = segs[n]
            c.direction, c.open, c.specified, c.on_pe = s.direction, int(s.open), int(s.specified), int(s.on_pe)
            c.is_E_or_W, c.is_N_or_S = int(s.is_E_or_W), int(s.is_N_or_S)
            for k in ("IsdB", "IedB", "JsdB", "JedB", "isd", "ied", "jsd", "jed"):
                setattr(c, k, int(s.HI.get(k, 0)))
            c.radiation, c.gradient, c.nudged, c.oblique = int(s.radiation), int(s.gradient), int(s.nudged), int(s.oblique)
            c.Flather = int(s.Flather)
            c.radiation_tan_or_grad = (_abi.OBC_TAN_RADIATION * int(s.radiation_tan) | _abi.OBC_GRAD_RADIATION * int(s.radiation_grad) |
                                       _abi.OBC_TAN_NUDGED * int(s.nudged_tan) | _abi.OBC_GRAD_NUDGED * int(s.nudged_grad) |
                                       _abi.OBC_TAN_OBLIQUE * int(s.oblique_tan) | _abi.OBC_GRAD_OBLIQUE * int(s.oblique_grad))
            c.Velocity_nudging_timescale_in, c.Velocity_nudging_timescale_out = float(s.Velocity_nudging_timescale_in), float(s.Velocity_nudging_timescale_out)
            c.Tr_InvLscale_in, c.Tr_InvLscale_out = float(s.Tr_InvLscale_in), float(s.Tr_InvLscale_out)
            for k in ("normal_trans", "normal_vel", "tangential_vel", "tangential_grad", "nudged_normal_vel", "normal_vel_bt", "SSH",
                      "nudged_tangential_vel", "nudged_tangential_grad"):
                a = getattr(s, k)
                need = {"normal_trans": s.specified, "normal_vel": s.specified or s.radiation or s.gradient or s.oblique, "nudged_normal_vel": s.nudged,
                        "normal_vel_bt": s.Flather, "SSH": s.Flather,
                        "tangential_vel": self.computed_vorticity or self.computed_strain or s.radiation_tan or s.nudged_tan or s.oblique_tan,
                        "tangential_grad": self.specified_vorticity or s.radiation_grad or s.nudged_grad or s.oblique_grad,
                        "nudged_tangential_vel": s.nudged_tan, "nudged_tangential_grad": s.nudged_grad}[k]
                if a is not None and need and s.on_pe:
                    if to_ptr is None:
                        a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); setattr(c, k, a.ctypes.data)
                    else:
                        p, owner = to_ptr(a); keep.append(owner); setattr(c, k, p)
        for n, s in enumerate(self.segment):      # the tracer registries of the segments (read by advect_tracer)
            if s.tr_Reg is None:
                continue
            trs = (_abi.ObcSegmentTracer * max(len(s.tr_Reg), 1))()
            for m, t in enumerate(s.tr_Reg):
                trs[m].ntr_index, trs[m].OBC_inflow_conc = int(t["ntr_index"]), float(t.get("OBC_inflow_conc", 0.0))
                trs[m].resrv_lfac_in, trs[m].resrv_lfac_out = float(t.get("resrv_lfac_in", 1.0)), float(t.get("resrv_lfac_out", 1.0))
                for key in ("tres", "t"):      # the reservoir (updated in place by update_segment_tracer_reservoirs) and the external values
                    a = t.get(key)
                    if a is not None:
                        if tres_ptr is None:
                            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
                                raise Mom6HipError(f"the segments' tr_Reg {key} arrays must be contiguous float64")
                            keep.append(a); setattr(trs[m], key, a.ctypes.data)
                        else:
                            p, owner = tres_ptr(a); keep.append(owner); setattr(trs[m], key, p)
            keep.append(trs)
            segs[n].tr_Reg = C.cast(trs, C.POINTER(_abi.ObcSegmentTracer)); segs[n].ntseg = len(s.tr_Reg)
        o = _abi.Obc()
        o.number_of_segments, o.OBC_pe = self.number_of_segments, int(self.OBC_pe)
        for k in ("open_u_BCs_exist_globally", "open_v_BCs_exist_globally", "specified_u_BCs_exist_globally", "specified_v_BCs_exist_globally",
                  "Flather_u_BCs_exist_globally", "Flather_v_BCs_exist_globally", "zero_vorticity", "freeslip_vorticity", "computed_vorticity",
                  "specified_vorticity", "zero_strain", "freeslip_strain", "computed_strain", "zero_biharmonic"):
            setattr(o, k, int(getattr(self, k)))
        o.segment = C.cast(segs, C.POINTER(_abi.ObcSegment))
        o.segnum_u, o.segnum_v = self.segnum_u.ctypes.data, self.segnum_v.ctypes.data
        o.gamma_uv, o.rx_max = self.gamma_uv, self.rx_max      # (read by the RK2 step; radiation_open_bdry_conds takes them as arguments)
        for k in ("rx_normal", "ry_normal", "rx_oblique_u", "ry_oblique_u", "cff_normal_u", "rx_oblique_v", "ry_oblique_v", "cff_normal_v"):
            a = getattr(self, k)
            if a is not None:
                if to_ptr is None:
                    keep.append(a); setattr(o, k, a.ctypes.data)
                else:
                    ptr, owner = to_ptr(a); keep.append(owner); setattr(o, k, ptr)
        o._keep = keep
        return o


def open_boundary_config(G, segment_strs, **kw):
    """open_boundary_config(G, US, param_file, OBC) :429 for OBC_NUMBER_OF_SEGMENTS = len(segment_strs), OBC_SEGMENT_%%% = segment_strs"""
    return ocean_OBC_type(G.grid if hasattr(G, "grid") else G, list(segment_strs), **kw)


def _obc_call_space(arrays):
    from .tracer_advect import _ptr_space
    spaces, ptrs = set(), []
    for a in arrays:
        if a is None:
            ptrs.append(None); continue
        p, sp = _ptr_space(a); spaces.add(sp); ptrs.append(C.c_void_p(p))
    if len(spaces) != 1:
        raise Mom6HipError("MOM_open_boundary: all fields must be in the same memory space")
    return ptrs, spaces.pop()


def _seg_to_ptr(space):
    def to_ptr(a):      # a segment's own array in the memory space of the call (device: a torch tensor kept on the segment by the caller)
        if hasattr(a, "data_ptr"):
            return a.data_ptr(), a
        a = np.ascontiguousarray(a, dtype=np.float64)
        if space == _abi.MEM_DEVICE:      # not on the device: handed over as absent (the library names what a call needs and does not find)
            return 0, None
        return a.ctypes.data, a
    return to_ptr


def radiation_open_bdry_conds(OBC, u_new, u_old, v_new, v_old, G, dt):
    """radiation_open_bdry_conds(OBC, u_new, u_old, v_new, v_old, G, GV, US, dt) -- :2196: the normal component (Orlanski radiation, the
    gradient condition, nudging), open_boundary_apply_normal_flow and the pass of u_new, v_new; segment%normal_vel and OBC%rx_normal /
    ry_normal are updated in place"""
    from ._lib import check, lib
    if OBC is None:
        return
    ptrs, space = _obc_call_space([OBC.rx_normal, OBC.ry_normal, u_new, u_old, v_new, v_old])
    obc = OBC.struct(_seg_to_ptr(space))
    L = lib()
    L.mom6hip_radiation_open_bdry_conds.argtypes = [C.c_void_p, C.POINTER(_abi.Obc), C.c_double, C.c_double] + [C.c_void_p] * 6 + [C.c_double, C.c_int32]
    check(L.mom6hip_radiation_open_bdry_conds(G.handle, C.byref(obc), OBC.gamma_uv, OBC.rx_max, *ptrs, float(dt), space), "radiation_open_bdry_conds")


def update_segment_tracer_reservoirs(G, uhr, vhr, h, OBC, dt, Reg):
    """update_segment_tracer_reservoirs(G, GV, uhr, vhr, h, OBC, dt, Reg) -- :5373: the reservoirs tr_Reg[m]["tres"] of the segments are
    updated in place (arrays in the memory space of the fields); Reg: the list of tracer arrays"""
    from ._lib import check, lib
    if OBC is None or not any(s.tr_Reg for s in OBC.segment):
        return
    ptrs, space = _obc_call_space([uhr, vhr, h] + list(Reg))

    def tres_ptr(a):
        if hasattr(a, "data_ptr"):
            if space != _abi.MEM_DEVICE:
                raise Mom6HipError("update_segment_tracer_reservoirs: the segments' tracer arrays must be in the memory space of the fields")
            return a.data_ptr(), a
        if space != _abi.MEM_HOST or not (a.dtype == np.float64 and a.flags.c_contiguous):
            raise Mom6HipError("update_segment_tracer_reservoirs: the segments' tracer arrays must be contiguous float64 in the memory space of the fields")
        return a.ctypes.data, a
    obc = OBC.struct(lambda a: (0, None), tres_ptr)
    trp = (C.c_void_p * len(Reg))(*ptrs[3:])
    L = lib()
    L.mom6hip_update_segment_tracer_reservoirs.argtypes = [C.c_void_p] * 4 + [C.POINTER(_abi.Obc), C.c_double, C.c_void_p, C.c_int32, C.c_int32]
    check(L.mom6hip_update_segment_tracer_reservoirs(G.handle, ptrs[0], ptrs[1], ptrs[2], C.byref(obc), float(dt), trp, len(Reg), space),
          "update_segment_tracer_reservoirs")


def open_boundary_zero_normal_flow(OBC, G, u, v):
    """open_boundary_zero_normal_flow(OBC, G, GV, u, v) -- :3374"""
    from ._lib import check, lib
    if OBC is None:
        return
    ptrs, space = _obc_call_space([u, v])
    obc = OBC.struct(_seg_to_ptr(space))
    L = lib()
    L.mom6hip_open_boundary_zero_normal_flow.argtypes = [C.c_void_p, C.POINTER(_abi.Obc), C.c_void_p, C.c_void_p, C.c_int32]
    check(L.mom6hip_open_boundary_zero_normal_flow(G.handle, C.byref(obc), *ptrs, space), "open_boundary_zero_normal_flow")
