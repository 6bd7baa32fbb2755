"""Loader of libmom6hip.so (the product).  There is no CPU fallback: if the library is missing or
no gfx950 device is usable, every operation raises."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOM6HIP_LIB_PATH: another build of the same library (kernel experiments: tools/build_variant.sh)
LIB_PATH = os.environ.get("MOM6HIP_LIB_PATH") or os.path.join(_HERE, "libmom6hip.so")
_LIB = None
_dp = C.POINTER(C.c_double)

# every symbol include/mom6hip.h declares
EXPORTS = (
    "mom6hip_init", "mom6hip_last_error", "mom6hip_grid_create", "mom6hip_grid_destroy",
    "mom6hip_sync", "mom6hip_malloc", "mom6hip_free", "mom6hip_sync_to_device",
    "mom6hip_sync_to_host", "mom6hip_halo_update", "mom6hip_advect_tracer", "mom6hip_set_timing",
    "mom6hip_advect_get_timing", "mom6hip_ale_remap_tracers", "mom6hip_ale_plm_edge_values", "mom6hip_coradcalc", "mom6hip_continuity", "mom6hip_pressureforce_fv_bouss", "mom6hip_pressureforce_fv_nonbouss", "mom6hip_calculate_density", "mom6hip_set_domain_callbacks",
    "mom6hip_barotropic_init", "mom6hip_btcalc", "mom6hip_bt_mass_source", "mom6hip_set_dtbt", "mom6hip_memset_zero", "mom6hip_transfer_stats", "mom6hip_debug_poison_passes", "mom6hip_tracer_hordiff_varmix", "mom6hip_tracer_hordiff_neutral", "mom6hip_tracer_hordiff_epipycnal", "mom6hip_continuity_obc", "mom6hip_coradcalc_obc", "mom6hip_vertvisc_coef_obc", "mom6hip_update_segment_tracer_reservoirs", "mom6hip_advect_tracer_obc", "mom6hip_btstep_obc", "mom6hip_horizontal_viscosity_obc", "mom6hip_set_viscous_bbl_obc", "mom6hip_btcalc_obc", "mom6hip_vertvisc_obc", "mom6hip_radiation_open_bdry_conds", "mom6hip_open_boundary_zero_normal_flow", "mom6hip_overlap_stats", "mom6hip_thickness_diffuse", "mom6hip_mixedlayer_restrat", "mom6hip_mixedlayer_restrat_mu", "mom6hip_start_group_pass", "mom6hip_complete_group_pass", "mom6hip_set_dtbt_eta", "mom6hip_btstep",
    "mom6hip_dyn_split_rk2_init", "mom6hip_step_dyn_split_rk2", "mom6hip_dyn_split_rk2b_init", "mom6hip_step_dyn_split_rk2b",
    "mom6hip_ale_regrid", "mom6hip_ale_remap_set_h_vel", "mom6hip_ale_remap_set_h_vel_via_dz", "mom6hip_ale_remap_velocities", "mom6hip_halo_pack", "mom6hip_set_min_callback", "mom6hip_kernel_timing", "mom6hip_set_callback_stream_ordered", "mom6hip_bt_graph_stats", "mom6hip_bt_graph_nodes",
    "mom6hip_vertvisc_coef", "mom6hip_vertvisc", "mom6hip_vertvisc_remnant", "mom6hip_vertvisc_ntrunc", "mom6hip_vertvisc_and_remnant", "mom6hip_vertvisc_step",
    "mom6hip_hor_visc_init", "mom6hip_horizontal_viscosity", "mom6hip_set_viscous_bbl", "mom6hip_set_viscous_ml",
    "mom6hip_chksum", "mom6hip_reproducing_sum", "mom6hip_write_energy_sums", "mom6hip_depth_list_create", "mom6hip_write_energy_ape", "mom6hip_host_register", "mom6hip_host_unregister", "mom6hip_stage_to_host",
    "mom6hip_stage_query", "mom6hip_stage_wait", "mom6hip_stream_bandwidth", "mom6hip_graph_node_floor", "mom6hip_tracer_hordiff", "mom6hip_rccl_get_unique_id", "mom6hip_domain_init_rccl", "mom6hip_domain_exchange_timing",
)


class Mom6HipError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise Mom6HipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C mom6_amd/csrc). mom6_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch brings its own copy of the ROCm libraries, and a process that loads the system's copy
        # first (through this library) and PyTorch's afterwards ends with two runtimes of which the second finds no device
        # ("no ROCm-capable device is detected").  Where PyTorch is installed it is imported first, so that libmom6hip binds to the
        # copy already in the process.  (A Fortran host has no PyTorch and binds to /opt/rocm.)
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.mom6hip_last_error.restype = C.c_char_p
        L.mom6hip_init.argtypes = [C.c_int]
        L.mom6hip_grid_create.argtypes = [C.POINTER(_abi.GridStruct), C.c_void_p, C.POINTER(C.c_void_p)]
        L.mom6hip_grid_destroy.argtypes = [C.c_void_p]
        L.mom6hip_sync.argtypes = [C.c_void_p]
        L.mom6hip_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_uint64]
        L.mom6hip_free.argtypes = [C.c_void_p]
        L.mom6hip_sync_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.mom6hip_sync_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.mom6hip_halo_update.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.c_int32]
        L.mom6hip_advect_tracer.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
            C.POINTER(_abi.TracerAdvectCS), C.POINTER(C.c_void_p), _dp, C.c_int32, C.c_int32,
            C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
            C.POINTER(_abi.AdvectStats)]
        L.mom6hip_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.mom6hip_advect_get_timing.argtypes = [C.c_void_p, C.POINTER(_abi.AdvectTiming)]
        L.mom6hip_ale_remap_tracers.argtypes = [C.c_void_p, C.POINTER(_abi.RemappingCS), C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_void_p), _dp, C.c_int32, C.c_int32]
        L.mom6hip_coradcalc.argtypes = [C.c_void_p, C.POINTER(_abi.CoriolisAdvCS)] + [C.c_void_p] * 7 + [C.c_int32]
        L.mom6hip_continuity.argtypes = ([C.c_void_p, C.POINTER(_abi.ContinuityCS)] + [C.c_void_p] * 6 + [C.c_double]
                                         + [C.c_void_p] * 6 + [C.POINTER(_abi.BTCont)] + [C.c_void_p] * 2 + [C.c_int32])
        L.mom6hip_pressureforce_fv_bouss.argtypes = ([C.c_void_p, C.POINTER(_abi.PressureForceCS), C.POINTER(_abi.EOS)]
                                                     + [C.c_void_p] * 8 + [C.c_int32])
        L.mom6hip_pressureforce_fv_nonbouss.argtypes = ([C.c_void_p, C.POINTER(_abi.PressureForceCS), C.POINTER(_abi.EOS)]
                                                        + [C.c_void_p] * 4 + [C.c_double] + [C.c_void_p] * 4 + [C.c_int32])
        L.mom6hip_calculate_density.argtypes = ([C.c_void_p, C.POINTER(_abi.EOS)] + [C.c_void_p] * 4
                                                + [C.c_int64, C.c_int32, C.c_double, C.c_int32])
        L.mom6hip_set_domain_callbacks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mom6hip_kernel_timing.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.mom6hip_bt_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mom6hip_set_callback_stream_ordered.argtypes = [C.c_void_p, C.c_int32]
        L.mom6hip_set_min_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mom6hip_halo_pack.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                        C.POINTER(C.c_int64)]
        for n in ("grid", "tracer_advect_cs", "advect_stats", "advect_timing"):
            getattr(L, f"mom6hip_abi_sizeof_{n}").restype = C.c_uint64
        L.mom6hip_abi_offsetof_grid_mask2dT.restype = C.c_uint64
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().mom6hip_last_error().decode("utf-8", "replace")
        raise Mom6HipError(f"{what}: {msg} (rc={rc})")
