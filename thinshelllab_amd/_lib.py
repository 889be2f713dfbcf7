"""ctypes binding of ``libtsl_hip.so`` (C ABI declared in ``include/tsl_hip.h``).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing an
engine object that needs it raises ``TslLibraryError``; if no GPU is visible ``tsl_ctx_create`` fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSL_HIP_LIB") or os.path.join(_HERE, "lib", "libtsl_hip.so")   # TSL_HIP_LIB: A/B builds of the same library (kernel experiments)


class TslLibraryError(RuntimeError):
    pass


class TslError(RuntimeError):
    pass


class ClothDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("NV", C.c_int32), ("NF", C.c_int32), ("v_offset", C.c_int32),
                ("dx", C.c_double), ("mass", C.c_double), ("Kl", C.c_double), ("Ka", C.c_double), ("Kb", C.c_double), ("k_angle", C.c_double),
                ("f2v", C.c_void_p), ("counter_face", C.c_void_p), ("counter_point", C.c_void_p), ("rest_area", C.c_void_p), ("rest_len", C.c_void_p)]


class ElasticDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_verts", C.c_int32), ("n_cells", C.c_int32), ("v_offset", C.c_int32),
                ("mu", C.c_double), ("lam", C.c_double), ("alpha", C.c_double),
                ("tets", C.c_void_p), ("B", C.c_void_p), ("W", C.c_void_p)]


class ContactPair(C.Structure):
    _fields_ = [("b_idx", C.c_int32), ("v_start", C.c_int32), ("v_end", C.c_int32), ("mu_is_param", C.c_int32), ("mu", C.c_double)]


class Body(C.Structure):
    _fields_ = [("v_start", C.c_int32), ("v_end", C.c_int32), ("f_start", C.c_int32), ("f_end", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("tot_NV", C.c_int32), ("tot_NF", C.c_int32),
                ("dt", C.c_double), ("k_contact", C.c_double), ("eps_contact", C.c_double), ("eps_v", C.c_double), ("damping", C.c_double),
                ("max_n_constraints", C.c_int32),
                ("n_cloth", C.c_int32), ("cloths", C.POINTER(ClothDesc)),
                ("n_elastic", C.c_int32), ("elastics", C.POINTER(ElasticDesc)),
                ("n_body", C.c_int32), ("bodies", C.POINTER(Body)),
                ("n_pair", C.c_int32), ("pairs", C.POINTER(ContactPair)),
                ("mass", C.c_void_p), ("gravity", C.c_void_p), ("faces", C.c_void_p), ("frozen", C.c_void_p),
                ("grid_h", C.c_double)]


class StepStats(C.Structure):
    _fields_ = [("newton_iters", C.c_int32), ("ls_evals", C.c_int32), ("cg_iters", C.c_int32), ("solves", C.c_int32),
                ("restarts", C.c_int32), ("fallback", C.c_int32), ("nc", C.c_int32),
                ("last_delta", C.c_double), ("last_alpha", C.c_double), ("energy", C.c_double),
                ("unconverged", C.c_int32), ("attained", C.c_int32), ("factorizations", C.c_int32), ("plans", C.c_int32),
                ("max_rel_residual", C.c_double), ("max_backward_error", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SolveStats(C.Structure):
    _fields_ = [("iters", C.c_int32), ("restarts", C.c_int32), ("flag", C.c_int32), ("rel_residual", C.c_double),
                ("method", C.c_int32), ("attained", C.c_int32), ("backward_error", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/tsl_hip.h declares (tests check the shared object exports all of them)
EXPORTS = [
    "tsl_version", "tsl_last_error", "tsl_ctx_create", "tsl_ctx_destroy", "tsl_set_stream", "tsl_set_param", "tsl_set_frozen",
    "tsl_set_ext_force", "tsl_set_gravity", "tsl_energy", "tsl_assemble", "tsl_solve", "tsl_step", "tsl_contact_detect",
    "tsl_contact_reset", "tsl_update_ref_angle", "tsl_adjoint_step", "tsl_param_grad", "tsl_friction_grad", "tsl_elastic_force", "tsl_matrix_nnzb", "tsl_matrix_export",
    "tsl_constraints_export", "tsl_contact_blocks_export", "tsl_proj_export", "tsl_proj_import", "tsl_set_border", "tsl_spd_project", "tsl_profile_reset", "tsl_profile_read", "tsl_profile_read_events",
    "tsl_bench_spmv", "tsl_bench_direct", "tsl_direct_info", "tsl_direct_counters",
    "tsl_group_create", "tsl_group_destroy", "tsl_group_step", "tsl_group_adjoint_step", "tsl_group_info",
]

_lib = None


def load():
    """Load libtsl_hip.so; raise TslLibraryError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles the HIP runtime this process must share (device pointers come from torch
    # tensors); loading libtsl_hip.so before it would pull a second libamdhip64 into the process.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise TslLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). thinshelllab_amd has no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise TslLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    L.tsl_version.restype = C.c_char_p
    L.tsl_last_error.restype = C.c_char_p
    L.tsl_ctx_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
    L.tsl_ctx_destroy.argtypes = [C.c_void_p]
    L.tsl_ctx_destroy.restype = None
    L.tsl_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.tsl_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.tsl_set_frozen.argtypes = [C.c_void_p, C.c_void_p]
    L.tsl_set_ext_force.argtypes = [C.c_void_p, C.c_void_p]
    L.tsl_set_gravity.argtypes = [C.c_void_p, C.c_void_p]
    L.tsl_energy.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.POINTER(C.c_double)]
    L.tsl_assemble.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
    L.tsl_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SolveStats)]
    L.tsl_step.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.POINTER(StepStats)]
    L.tsl_contact_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.tsl_contact_reset.argtypes = [C.c_void_p]
    L.tsl_update_ref_angle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsl_adjoint_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                   C.POINTER(SolveStats)]
    L.tsl_matrix_nnzb.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.tsl_matrix_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsl_constraints_export.argtypes = [C.c_void_p] + [C.c_void_p] * 7 + [C.c_int32]
    L.tsl_contact_blocks_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.tsl_proj_export.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    L.tsl_proj_import.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsl_set_border.argtypes = [C.c_void_p, C.c_void_p]
    L.tsl_spd_project.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.tsl_profile_reset.argtypes = [C.c_void_p, C.c_int]
    L.tsl_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tsl_profile_read_events.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.tsl_bench_spmv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.tsl_bench_direct.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.tsl_direct_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.tsl_direct_counters.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]
    L.tsl_param_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.tsl_friction_grad.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.tsl_elastic_force.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsl_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]
    L.tsl_group_destroy.argtypes = [C.c_void_p]
    L.tsl_group_destroy.restype = None
    L.tsl_group_step.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(StepStats)]
    L.tsl_group_adjoint_step.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.POINTER(C.c_void_p)] * 5 + [C.POINTER(C.c_double), C.POINTER(SolveStats)]
    L.tsl_group_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    _lib = L
    return L


def check(rc, what=""):
    if rc < 0:
        raise TslError(f"{what}: {load().tsl_last_error().decode()}")
    return rc
