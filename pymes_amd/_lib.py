"""ctypes binding of include/pymes_amd.h.

The product library is ``pymes_amd/lib/libpymes_amd.so`` (built by
``__graft_entry__.build()`` / ``make -C pymes_amd/csrc``).  Loading fails loudly when
the library is missing or is not the HIP build: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYMES_AMD_LIBRARY=<path>: another build of the same library (A/B runs of kernel variants); the backend check below applies
DEFAULT_PATH = os.environ.get("PYMES_AMD_LIBRARY") or os.path.join(_HERE, "lib", "libpymes_amd.so")

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)


class ShardBuffers(C.Structure):
    """pymes_shard_buffers of include/pymes_amd.h (device pointers, in this order)."""
    _fields_ = [(k, C.c_void_p) for k in ("ETd", "ETx", "L", "QK", "Tall", "W", "Xvv", "P", "R1", "S")]

c_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); exactly the declarations of include/pymes_amd.h
SIGNATURES = {
    "pymes_last_error": (C.c_char_p, []),
    "pymes_backend": (C.c_char_p, []),
    "pymes_ctx_create": (C.c_int, [c_pp, C.c_int, C.c_int, C.c_int, C.c_uint64]),
    "pymes_ctx_destroy": (C.c_int, [C.c_void_p]),
    "pymes_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pymes_ctx_sync": (C.c_int, [C.c_void_p]),
    "pymes_ctx_workspace": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pymes_malloc": (C.c_int, [C.c_void_p, C.c_uint64, c_pp]),
    "pymes_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pymes_live_allocations": (C.c_int, [c_i64_p]),
    "pymes_dress_generation": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "pymes_phase_enable": (C.c_int, [C.c_int]),
    "pymes_phase_hold": (C.c_int, [C.c_int]),
    "pymes_phase_stats": (C.c_int, [c_i64_p, c_i64_p, c_i64_p, c_i64_p]),
    "pymes_mem_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pymes_graph_begin": (C.c_int, [C.c_void_p]),
    "pymes_graph_end": (C.c_int, [C.c_void_p, c_pp]),
    "pymes_graph_abort": (C.c_int, [C.c_void_p]),
    "pymes_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pymes_graph_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pymes_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "pymes_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "pymes_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "pymes_memset_zero": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "pymes_contract": (C.c_int, [C.c_void_p, C.c_double,
                                 C.c_void_p, C.c_char_p, c_i64_p, c_i64_p,
                                 C.c_void_p, C.c_char_p, c_i64_p, c_i64_p,
                                 C.c_double, C.c_void_p, C.c_char_p, c_i64_p, c_i64_p, C.c_char_p]),
    "pymes_permute": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_char_p, c_i64_p, c_i64_p,
                                C.c_double, C.c_void_p, C.c_char_p, c_i64_p]),
    "pymes_dgemm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64,
                              C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_int64]),
    "pymes_set_V_pqrs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_i64_p]),
    "pymes_set_V_block": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int, c_i64_p]),
    "pymes_V_exchange_asymmetry": (C.c_int, [C.c_void_p, c_double_p]),
    "pymes_exchange_asymmetry": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64_p, c_double_p]),
    "pymes_set_V_from_factors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "pymes_V_block_ptr": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_pp, c_i64_p]),
    "pymes_set_orbital_energies": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_mp2": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, c_double_p]),
    "pymes_ccsd_dress_fock": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_ccsd_dress_V": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "pymes_ccsd_dress_V_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pymes_ccsd_singles_residual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_ccsd_singles_residual_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                      C.c_int, C.c_uint32]),
    "pymes_doubles_residual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "pymes_ladder": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]),
    "pymes_ladder_sym": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "pymes_ladder_sym_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]),
    "pymes_hole_ladder_packed_multi": (C.c_int, [C.c_void_p, c_pp, c_pp, c_pp, C.c_int, C.c_void_p]),
    "pymes_ladder_sym_multi": (C.c_int, [C.c_void_p, c_pp, C.c_int, C.c_void_p, C.c_int]),
    "pymes_ladder_dress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "pymes_pair_layouts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_symmetrised_assemble": (C.c_int, [C.c_void_p] * 7),
    "pymes_hole_ladder_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                           C.c_void_p]),
    "pymes_residual_slab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_slab_prepare_ws": (C.c_int, [C.c_void_p, c_i64_p]),
    "pymes_slab_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32]),
    "pymes_residual_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pymes_residual_finish_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p]),
    "pymes_xvv_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32]),
    "pymes_ccsd_dress_fock_ws": (C.c_int, [C.c_void_p, c_i64_p]),
    "pymes_ccsd_dress_fock_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pymes_ccsd_dress_fock_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_cc_update_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                        C.c_int]),
    "pymes_pairs_supported": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "pymes_pairs_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pymes_pairs_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pymes_ccsd_dress_abcd_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pymes_cc_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                  C.c_int]),
    "pymes_cc_update_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                     C.c_int]),
    "pymes_energy_norms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_double_p]),
    "pymes_energy_norms_start": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "pymes_energy_norms_wait": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "pymes_readback_start": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "pymes_readback_wait": (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_int]),
    "pymes_energy_norms_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           c_double_p]),
    "pymes_ccsd_energy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_double_p]),
    "pymes_ccd_energy": (C.c_int, [C.c_void_p, C.c_void_p, c_double_p]),
    "pymes_ueg_eval_2b": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_ueg_eval_2b_corr": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                         c_double_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_ueg_eval_2b_tab": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pymes_hf_fock_matrix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_fcidump_header": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pymes_fcidump_read_host": (C.c_int, [C.c_char_p, C.c_int, c_double_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_fcidump_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_double_p, C.c_void_p, C.c_void_p, c_i64_p]),
    "pymes_packed_header": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "pymes_packed_load": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p, C.c_void_p, C.c_void_p]),
    "pymes_packed_write": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    "pymes_packed_write_factors": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "pymes_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, c_i64_p, c_double_p, C.c_int64]),
    "pymes_tc_single_contraction": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pymes_tc_double_contraction": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pymes_tc_triple_contraction": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_double_p]),
    "pymes_dots": (C.c_int, [C.c_void_p, C.c_int, c_pp, c_pp, C.c_int64, c_double_p]),
    "pymes_dots_var": (C.c_int, [C.c_void_p, C.c_int, c_pp, c_pp, c_i64_p, c_double_p]),
    "pymes_lincomb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_pp, c_double_p, C.c_int64]),
    "pymes_cmul": (C.c_int, [C.c_void_p] * 7 + [C.c_int64]),
    "pymes_ccsd_residuals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pymes_ccsd_iterate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_ccsd_release": (C.c_int, [C.c_void_p]),
    "pymes_set_collectives": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pymes_shard_buffer_sizes": (C.c_int, [C.c_void_p, C.c_int, c_i64_p]),
    "pymes_ccsd_sharded_residuals": (C.c_int, [C.c_void_p] * 6 + [C.c_uint32, C.c_void_p]),
    "pymes_ccd_sharded_residuals": (C.c_int, [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p]),
    "pymes_set_alltoallv": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pymes_owner_tile_sizes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_i64_p, c_i64_p]),
    "pymes_set_owner_tile_buffers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_ccsd_sharded_finish": (C.c_int, [C.c_void_p] * 6 + [C.POINTER(C.c_int)]),
    "pymes_ccsd_sharded_energy": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "pymes_ccsd_sharded_await": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pymes_cshift_inv": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_double] * 5 + [C.c_void_p, C.c_void_p, C.c_int64]),
    "pymes_eom_sigma_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pymes_eom_sigma_flags": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pymes_eom_sigma_apply": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5),
    "pymes_eom_diagonals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "pymes_scratch_trim": (C.c_int, [C.c_void_p]),
    "pymes_eom_sigma_destroy": (C.c_int, [C.c_void_p]),
    "pymes_diis_mix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, c_pp, c_pp, c_i64_p, c_pp, c_pp]),
    "pymes_diis_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pymes_diis_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_pp, c_pp, c_i64_p, C.c_int, C.c_int, C.c_int]),
    "pymes_gemm_group_begin": (C.c_int, [C.c_void_p]),
    "pymes_gemm_group_end": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p]),
    "pymes_gram": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_pp, c_pp, C.c_int64, c_double_p]),
    "pymes_lincomb_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_pp, C.c_void_p, C.c_void_p, c_pp, C.c_int64]),
    "pymes_lincomb_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_pp, C.c_void_p, C.c_int64]),
    "pymes_stats": (C.c_int, [C.c_void_p, C.c_int, c_i64_p, c_double_p, c_i64_p, c_double_p]),
    "pymes_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "pymes_prof_reset": (C.c_int, [C.c_void_p]),
    "pymes_prof_query": (C.c_int, [C.c_void_p, C.c_int, c_i64_p, c_i64_p, c_double_p, c_double_p]),
}

PYMES_DCD, PYMES_USE_DRESSED, PYMES_SKIP_LADDER, PYMES_SYM_LADDER, PYMES_SYM_RINGS = 1, 2, 4, 8, 16
PYMES_OWNER_TILES = 1 << 21
PYMES_REUSE_LAYOUTS = 32
PYMES_T1_ZERO = 1 << 20
PYMES_SLAB_RINGS_ONLY, PYMES_SLAB_LADDERS_ONLY = 64, 128
PYMES_DRESS_ABIJ_REDUCED = 1 << 16


class PymesError(RuntimeError):
    pass


class Library:
    """Loaded C-ABI library with typed entry points and error checking."""

    BACKEND = "hip-gfx950"       # the only backend the package accepts: there is no CPU fallback

    def __init__(self, path=None):
        path = path or DEFAULT_PATH
        if not os.path.exists(path):
            raise PymesError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C pymes_amd/csrc).  pymes_amd has no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)      # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        self.backend = self.dll.pymes_backend().decode()
        if self.backend != self.BACKEND:
            raise PymesError(f"{path} reports backend '{self.backend}', expected '{self.BACKEND}'")

    def call(self, name, *args):
        rc = getattr(self.dll, name)(*args)
        if rc != 0:
            raise PymesError(f"{name}: {self.dll.pymes_last_error().decode()}")


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


def i64_array(values):
    return (C.c_int64 * len(values))(*[int(v) for v in values]) if values is not None else None


def ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*[int(p) for p in ptrs])


def host_ptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.c_void_p)
