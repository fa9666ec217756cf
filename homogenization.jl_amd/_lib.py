"""ctypes binding of libhmg_hip.so (C ABI: include/hmg.h).  No fallback: if the HIP library has not
been built, importing the compute API fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMG_LIB_PATH") or os.path.join(_HERE, "libhmg_hip.so")   # (HMG_LIB_PATH: A/B runs of two builds)

c_i64 = ctypes.c_int64
c_f64 = ctypes.c_double
c_int = ctypes.c_int
vp = ctypes.c_void_p
p_i64 = ctypes.POINTER(ctypes.c_int64)
p_i32 = ctypes.POINTER(ctypes.c_int32)
p_f64 = ctypes.POINTER(ctypes.c_double)
pp = ctypes.POINTER(ctypes.c_void_p)

EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
EXCHANGE_END_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
P2P_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                          ctypes.POINTER(ctypes.c_int64))

# name -> (restype, argtypes); every symbol declared in include/hmg.h
SIGNATURES = {
    "hmg_last_error": (ctypes.c_char_p, []),
    "hmg_version": (c_int, []),
    "hmg_ctx_create": (c_int, [c_int, vp, pp]),
    "hmg_ctx_create_on_stream": (c_int, [c_int, vp, pp]),
    "hmg_ctx_destroy": (c_int, [vp]),
    "hmg_ctx_sync": (c_int, [vp]),
    "hmg_ctx_release_memory": (c_int, [vp]),
    "hmg_ctx_set_option": (c_int, [vp, ctypes.c_char_p, c_i64]),
    "hmg_ctx_set_option_f64": (c_int, [vp, ctypes.c_char_p, c_f64]),
    "hmg_ctx_scalar_bank": (vp, [vp]),
    "hmg_ctx_apply_timing": (c_int, [vp, p_i64, p_f64, p_f64]),
    "hmg_ctx_counter": (c_i64, [vp, ctypes.c_char_p]),
    "hmg_ctx_apply_timing_level": (c_int, [vp, c_int, p_i64, p_f64, p_f64]),
    "hmg_rhs_axi_grad": (c_int, [vp, p_f64, vp]),
    "hmg_next_rhs": (c_int, [vp, vp, vp]),
    "hmg_local_rhs": (c_int, [vp, vp]),
    "hmg_integrate": (c_int, [vp, c_int, vp, vp, c_i64, p_f64, p_f64]),
    "hmg_grid_create": (c_int, [vp, c_int, c_int, c_i64, p_f64, c_i64, p_i64, pp]),
    "hmg_grid_destroy": (c_int, [vp]),
    "hmg_grid_set_operator": (c_int, [vp, p_f64, c_f64]),
    "hmg_grid_set_lambda": (c_int, [vp, c_f64]),
    "hmg_grid_shrink": (c_int, [vp, c_i64, c_i64]),
    "hmg_grid_reserve_spare": (c_int, [vp, c_int]),
    "hmg_grid_ncells": (c_i64, [vp]),
    "hmg_grid_nnodes": (c_i64, [vp]),
    "hmg_grid_nlevels": (c_int, [vp]),
    "hmg_grid_nf": (c_i64, [vp, c_int]),
    "hmg_grid_ld": (c_i64, [vp, c_int]),
    "hmg_grid_table_i32": (c_int, [vp, c_int, ctypes.c_char_p, p_i32, c_i64, p_i64]),
    "hmg_grid_table_f64": (c_int, [vp, c_int, ctypes.c_char_p, p_f64, c_i64, p_i64]),
    "hmg_vec_create": (c_int, [vp, c_int, pp]),
    "hmg_vec_wrap": (c_int, [vp, c_int, vp, pp]),
    "hmg_vec_destroy": (c_int, [vp]),
    "hmg_vec_device_ptr": (vp, [vp]),
    "hmg_vec_upload": (c_int, [vp, p_f64]),
    "hmg_vec_download": (c_int, [vp, p_f64]),
    "hmg_vec_fill": (c_int, [vp, c_f64]),
    "hmg_vec_fill_random": (c_int, [vp, ctypes.c_uint64, c_i64]),
    "hmg_vec_copy": (c_int, [vp, vp]),
    "hmg_vec_axpy": (c_int, [c_f64, vp, vp]),
    "hmg_vec_xpby": (c_int, [vp, c_f64, vp]),
    "hmg_vec_dot": (c_int, [vp, vp, p_f64]),
    "hmg_vec_norm_unique": (c_int, [vp, p_f64]),
    "hmg_apply": (c_int, [vp, c_int, c_f64, vp, vp]),
    "hmg_apply_ex": (c_int, [vp, c_int, c_f64, vp, vp, vp, c_int]),
    "hmg_residual": (c_int, [vp, c_int, vp, vp, vp]),
    "hmg_constraint": (c_int, [vp, c_int, vp]),
    "hmg_interface_sum": (c_int, [vp, c_int, vp]),
    "hmg_zero_duplicates": (c_int, [vp, c_int, vp]),
    "hmg_restrict": (c_int, [vp, c_int, vp, vp]),
    "hmg_prolong_add": (c_int, [vp, c_int, vp, vp]),
    "hmg_gather_base": (c_int, [vp, vp, p_f64]),
    "hmg_scatter_base": (c_int, [vp, p_f64, vp]),
    "hmg_smooth": (c_int, [vp, c_int, c_int, vp, vp, vp, vp, vp]),
    "hmg_level_tune_placement": (c_int, [vp, c_int, c_int, pp, c_int, c_int, p_f64]),
    "hmg_coarse_setup": (c_int, [vp]),
    "hmg_coarse_solve": (c_int, [vp, vp, vp]),
    "hmg_coarse_last_iterations": (c_int, [vp]),
    "hmg_coarse_misses": (c_i64, [vp]),
    "hmg_vcycle": (c_int, [vp, c_int, c_int, c_int, pp]),
    "hmg_vcycle_down": (c_int, [vp, c_int, c_int, pp]),
    "hmg_vcycle_up": (c_int, [vp, c_int, c_int, pp]),
    "hmg_grid_set_cut": (c_int, [vp, c_i64, c_i64, c_i64, c_i64, p_i64, p_i32, c_i64, p_i64, p_i32, c_i64, p_i64, p_i32]),
    "hmg_grid_set_exchange": (c_int, [vp, EXCHANGE_FN, EXCHANGE_FN, vp, vp, c_i64]),
    "hmg_grid_cut_buffer_doubles": (c_i64, [vp, c_int]),
    "hmg_grid_set_exchange_async": (c_int, [vp, EXCHANGE_FN, EXCHANGE_END_FN]),
    "hmg_grid_set_overlap": (c_int, [vp, c_int]),
    "hmg_grid_set_exchange_p2p": (c_int, [vp, c_int, P2P_FN, P2P_FN, vp, c_i64]),
    "hmg_grid_cut_stage_doubles": (c_i64, [vp]),
    "hmg_grid_exchange_messages": (c_int, [vp, c_int, p_i64, c_i64, p_i64]),
    "hmg_ctx_set_scalar_bank": (c_int, [vp, vp]),
    "hmg_ctx_stream": (vp, [vp]),
    "hmg_comm_unique_id": (c_int, [vp]),
    "hmg_comm_init": (c_int, [vp, c_int, c_int, vp]),
    "hmg_comm_destroy": (c_int, [vp]),
    "hmg_comm_stats": (c_int, [vp, p_i64, p_i64]),
    "hmg_comm_sum_host": (c_int, [vp, p_f64, c_int]),
    "hmg_grid_use_comm": (c_int, [vp]),
    "hmg_checkerboard_mesh_size": (c_int, [c_int, p_i64, p_i64, p_i64]),
    "hmg_checkerboard_mesh": (c_int, [c_int, p_i64, p_f64, c_int, c_int, p_f64, p_i64]),
    "hmg_block_owner": (c_int, [c_int, c_i64, p_f64, c_i64, p_i64, p_i64, c_f64, p_f64, p_i32]),
    "hmg_conductivity_per_element": (c_int, [c_int, c_i64, p_f64, c_i64, p_i64, p_i64, p_f64, p_f64, p_f64]),
    "hmg_grid_create_partition": (c_int, [vp, c_int, c_int, c_i64, p_f64, c_i64, p_i64, p_i32, c_int, c_int, pp]),
    "hmg_grid_create_partition_rehearsal": (c_int, [vp, c_int, c_int, c_i64, p_f64, c_i64, p_i64, p_i32, p_i32, c_int, c_int, pp]),
}

_lib = None


class HmgError(RuntimeError):
    pass


def load():
    """Load libhmg_hip.so and declare every entry point.  Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; "
                "g.build()' or `make -C homogenization.jl_amd/csrc`).  There is no CPU fallback.")
        # torch wheels bundle a HIP runtime with the same SONAME (libamdhip64.so.7) as the system one this
        # library links.  A process holds one of them: whichever is loaded first.  If the host program uses
        # torch on the GPU (multi-GPU layer, torch tensors wrapped as level vectors) torch must come first --
        # then this library runs on torch's runtime and shares its streams; the other order leaves torch
        # without a usable device.  Only acted upon when torch is already imported.
        import sys
        torch = sys.modules.get("torch")
        if torch is not None:
            try:
                if torch.cuda.is_available():
                    torch.cuda.init()
            except Exception:
                pass
        lib = ctypes.CDLL(LIB_PATH)
        # A/B run against an OLDER build (HMG_LIB_PATH=... HMG_LIB_AB=1): entry points added since are absent there; calling one
        # raises a clear error.  Without HMG_LIB_AB a missing symbol fails here, at load time (a stale build must not surface
        # as an AttributeError in the middle of a collective).
        ab = bool(os.environ.get("HMG_LIB_PATH")) and os.environ.get("HMG_LIB_AB") == "1"
        for name, (res, args) in SIGNATURES.items():
            if ab and not hasattr(lib, name):
                def _absent(*_a, _n=name):
                    raise HmgError(f"{_n} is not exported by {LIB_PATH} (older build loaded through HMG_LIB_PATH / HMG_LIB_AB)")
                setattr(lib, name, _absent)
                continue
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise ImportError(f"{LIB_PATH} does not export {name}, which include/hmg.h declares: stale or mismatched "
                                  "build (rebuild, or set HMG_LIB_AB=1 for an A/B run against an older build)") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def fingerprint():
    """What build is this?  sha256 of the library file that gets loaded and of the sources it is made from (csrc/ + include/hmg.h).
    bench.py reports a profile-derived number only next to the build it was measured on (profiles/apply_traffic.json)."""
    import glob
    import hashlib

    def sha(paths):
        h = hashlib.sha256()
        for q in paths:
            h.update(os.path.basename(q).encode())
            with open(q, "rb") as f:
                h.update(f.read())
        return h.hexdigest()
    src = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.hpp")) +
                 glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + glob.glob(os.path.join(_HERE, "csrc", "Makefile")) +
                 [os.path.join(os.path.dirname(_HERE), "include", "hmg.h")])
    return {"lib_sha256": sha([LIB_PATH]) if os.path.exists(LIB_PATH) else None, "src_sha256": sha([q for q in src if os.path.exists(q)])}


def check(rc):
    if rc != 0:
        raise HmgError(load().hmg_last_error().decode())
