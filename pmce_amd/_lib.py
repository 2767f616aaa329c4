"""ctypes binding of libpmce_hip.so (include/pmce_hip.h).  There is no fallback: if the library is missing or
a symbol is absent, importing / calling fails loudly.  The product path never touches oracle/."""
from __future__ import annotations

import ctypes as C
import os.path as osp

import os

HERE = osp.dirname(osp.abspath(__file__))
# PMCE_LIB_PATH: another build of the SAME library (scripts/build_variant.sh: A/B runs of two revisions inside one GPU session)
LIB_PATH = os.environ.get("PMCE_LIB_PATH") or osp.join(HERE, "libpmce_hip.so")

_f = C.c_void_p      # device pointer (float*/int*)
_i = C.c_int
_l = C.c_longlong
_s = C.c_void_p      # hipStream_t
_fl = C.c_float

# name -> argtypes  (restype is int unless listed in _RESTYPES); mirrors include/pmce_hip.h one-to-one
PROTOTYPES = {
    "pmce_version": [],
    "pmce_build_id": [],
    "pmce_last_error_string": [],
    "pmce_model_create": [_i, _i, _i, C.POINTER(C.c_void_p)],
    "pmce_model_destroy": [C.c_void_p],
    "pmce_model_set_tensor": [C.c_void_p, C.c_char_p, _f],
    "pmce_model_tensor_count": [C.c_void_p],
    "pmce_model_tensor_name": [C.c_void_p, _i],
    "pmce_model_set_regressor_rows": [C.c_void_p, _i],
    "pmce_model_finalize": [C.c_void_p],
    "pmce_model_finalize_on": [C.c_void_p, _s],
    "pmce_model_split_bytes": [C.c_void_p],
    "pmce_model_set_split_arena": [C.c_void_p, C.c_void_p, C.c_size_t],
    "pmce_model_set_gemm_mode_on": [C.c_void_p, _i, _s],
    "pmce_model_workspace_bytes": [C.c_void_p, _i],
    "pmce_model_workspace_offset": [C.c_void_p, _i, C.c_char_p],
    "pmce_lifter_forward": [C.c_void_p, _f, _f, _f, _i, _f, C.c_size_t, _s],
    "pmce_decoder_forward": [C.c_void_p, _f, _f, _f, _f, _i, _f, C.c_size_t, _s],
    "pmce_forward": [C.c_void_p, _f, _f, _f, _f, _f, _f, _i, _f, C.c_size_t, _s],
    "pmce_coevo_block_forward": [C.c_void_p, _i, _f, _f, _f, _f, _f, _i, _f, C.c_size_t, _s],
    "pmce_stream_precompute": [C.c_void_p, _f, _f, _i, _f, _f, _f, C.c_size_t, _s],
    "pmce_stream_forward": [C.c_void_p, _f, _f, _f, _i, _i, _f, _f, _f, _f, _f, C.c_size_t, _s],
    "pmce_window_tokens_f32": [_f, _f, _f, _f, _f, _fl, _f, _f, _i, _i, _i, _i, _i, _s],
    "pmce_window_tokens_ex_f32": [_f, _f, _f, _f, _f, _fl, _f, _f, _i, _i, _i, _i, _i, _i, _s],
    "pmce_window_rows_f32": [_f, _f, _f, _i, _i, _i, _i, _s],
    "pmce_model_set_concurrency": [C.c_void_p, _i],
    "pmce_model_wait_lifter": [C.c_void_p, _s],
    "pmce_model_profile": [C.c_void_p, _i],
    "pmce_model_set_gemm_mode": [C.c_void_p, _i],
    "pmce_model_gemm_mode": [C.c_void_p],
    "pmce_model_share_split_weights": [C.c_void_p, C.c_void_p],
    "pmce_model_set_split_min_batch": [C.c_void_p, _i],
    "pmce_model_set_overflow_policy": [C.c_void_p, _i],
    "pmce_model_get_overflow_policy": [C.c_void_p],
    "pmce_model_get_split_min_batch": [C.c_void_p],
    "pmce_model_get_concurrency": [C.c_void_p],
    "pmce_model_get_split_overlap": [C.c_void_p],
    "pmce_model_set_clock_probe": [C.c_void_p, C.c_void_p],
    "pmce_model_overflowed": [C.c_void_p],
    "pmce_model_clear_overflow": [C.c_void_p],
    "pmce_model_profile_read": [C.c_void_p, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_longlong)],
    "pmce_gemm_nt_f32": [_f, _f, _f, _f, _f, _i, _i, _i, _l, _i, _l, _i, _i, _l, _l, _i, _l, _l, _i, _l, _l, _l, _l, _s],
    "pmce_gemm_set_tuning": [_i, _i],
    "pmce_gemm_pack_split_f16": [_f, _i, _i, _i, _f, _f, _s],
    "pmce_gemm_nt_split_f16": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _l, _l, _i, _i, _s],
    "pmce_split_rows_f16": [_f, _l, _i, _l, _f, _s],
    "pmce_split_rows_scaled_f16": [_f, _l, _i, _l, _f, _f, _s],
    "pmce_gemm_nt_split_f16_rs": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _l, _i, _l, _l, _s],
    "pmce_gemm_pack_split_f16_blk": [_f, _i, _i, _i, _f, _f, _s],
    "pmce_gemm_nt_split_f16_blk": [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _l, _l, _i, _i, _i, _i, _l, _l, _s],
    "pmce_gemm_nt_split_f16_ln": [_f, _f, _i, _f, _f, _f, _i, _i, _f, _f, _fl, _f, _f, _f, _fl, _f, _s],
    "pmce_gemm_nt_split_f16_ex": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _l, _l, _i, _i, _i, _s],
    "pmce_ln_chain_ex_f32": [_f, _l, _i, _f, _f, _fl, _f, _i, _i, _f, _f, _f, _fl, _f, _i, _s],
    "pmce_seq_attention_ex_f32": [_f, _f, _i, _i, _i, _i, _l, _l, _l, _i, _s],
    "pmce_seq_attention_split_supported": [_i, _i],
    "pmce_seq_attention_split_f16": [_f, _f, _i, _i, _i, _i, _l, _l, _l, _s],
    "pmce_gemm_nt_split_f16_rowmap": [_f, _f, _f, _f, _f, _i, _i, _i, _l, _i, _l, _l, _s],
    "pmce_gemm_split_set_tuning": [_i],
    "pmce_embed_tokens_f32": [_f, _f, _f, _f, _f, _f, _l, _i, _i, _s],
    "pmce_embed_ln_f32": [_f, _f, _f, _f, _f, _f, _l, _i, _i, _f, _f, _fl, _f, _i, _s],
    "pmce_ln_chain_f32": [_f, _l, _i, _f, _f, _fl, _f, _i, _i, _f, _f, _f, _fl, _f, _s],
    "pmce_seq_attention_f32": [_f, _f, _i, _i, _i, _i, _l, _l, _l, _s],
    "pmce_lifter_head_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _s],
    "pmce_lifter_head_ex_f32": [_f, _f, _f, _fl, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _s],
    "pmce_gru_step_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _l, _l, _i, _i, _i, _s],
    "pmce_gru_step_split_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _l, _l, _i, _i, _i, _s],
    "pmce_gru_step_split_blk_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _l, _l, _i, _i, _i, _s],
    "pmce_div_scalar_f32": [_f, _f, _l, _fl, _s],
    "pmce_vertex_init_gather_f32": [_f, _f, _f, _i, _i, _s],
    "pmce_ca_image_floats": [],
    "pmce_ca_fold_img_f32": [_f, _f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _s],
    "pmce_joint_prep_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _s],
    "pmce_ca_fold_f32": [_f, _f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _s],
    "pmce_vertex_ca_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _s],
    "pmce_adaln_mlp_f32": [_f, _f, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _s],
    "pmce_ffn_image_floats": [],
    "pmce_ffn_pack_f16": [_f, _f, _f, _s],
    "pmce_adaln_mlp_pk_f32": [_f, _f, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _f, _s],
    "pmce_vertex_ca_mlp_pk_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _s],
    "pmce_vertex_ca_mlp_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _f, _f, _f, _f, _f, _f, _i, _i, _s],
    "pmce_adaln_qkv_f32": [_f, _f, _i, _i, _f, _f, _f, _i, _s],
    "pmce_qkv_image_floats": [],
    "pmce_qkv_pack_f16": [_f, _f, _s],
    "pmce_vertex_sa_f32": [_f, _f, _f, _f, _f, _i, _s],
    "pmce_vertex_sab_scratch_floats": [_i],
    "pmce_vertex_sab_split_f32": [_f, _f, _i, _i, _f, _f, _f, _f, _f, _f, _i, _s],
    "pmce_tokens_kv_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _f, _f, _f, _i, _s],
    "pmce_tkv_image_floats": [],
    "pmce_tkv_pack_f16": [_f, _f, _f, _f, _s],
    "pmce_tokens_kv_pk_f32": [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _f, _f, _f, _f, _f, _i, _f, _s],
    "pmce_joint_stream_f32": [_f, _f, _f, _f, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_int), _f, _f, _f, _i, _i, _i, _s],
    "pmce_build_final_operand_f32": [_f, _f, _f, _i, _i, _s],
    "pmce_build_final_operand_pk_f32": [_f, _f, _f, _i, _i, _i, _s],
    "pmce_j_regress_f32": [_f, _f, _f, _f, _f, _i, _i, _i, _fl, _s],
    "pmce_sample_errors_f32": [_f, _f, _fl, _i, _f, _f, _f, _f, _i, _f, _f, _i, _i, _f, _f, _f, _f, _f, _i, _s],
    "pmce_accel_error_f32": [_f, _f, _f, _f, _i, _i, _s],
    "pmce_assemble_windows_f32": [_f, _f, _f, _f, _f, _i, _i, _i, _s],
    "pmce_prepare_pose2d_f32": [_f, _i, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s],
}
_RESTYPES = {
    "pmce_last_error_string": C.c_char_p,
    "pmce_build_id": C.c_char_p,
    "pmce_model_destroy": None,
    "pmce_model_tensor_name": C.c_char_p,
    "pmce_model_workspace_bytes": C.c_size_t,
    "pmce_model_split_bytes": C.c_size_t,
    "pmce_model_workspace_offset": C.c_longlong,
    "pmce_vertex_sab_scratch_floats": C.c_longlong,
}

_lib = None


class PmceError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built — no CPU fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    # The library takes device pointers and hipStream_t handles that PyTorch created, so both must live in ONE HIP runtime.
    # torch bundles its own libamdhip64; imported first, its copy also satisfies this library's libamdhip64.so.7 dependency.
    # Loaded the other way round the process would hold two runtimes and the library's would see no device.
    import torch  # noqa: F401
    if not osp.exists(LIB_PATH):
        raise PmceError(f"{LIB_PATH} not found: build it first (python -m pmce_amd.build, or __graft_entry__.build()). "
                        "There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: loud by design
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def build_id() -> str:
    """Identity of the sources the loaded library was built from (pmce_amd.build.source_id() at build time)."""
    return (load().pmce_build_id() or b"").decode()


def last_error() -> str:
    return (load().pmce_last_error_string() or b"").decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise PmceError(f"{what or 'libpmce_hip'} failed (rc={rc}): {last_error()}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libpmce_hip needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def split_overlap() -> bool:
    """Kernels of the split-f16 mode may overlap each other (two streams inside a forward, pipeline lanes on their own streams):
    the library contains no packed-fp32 instruction, the one thing f16 matrix instructions disturb on MI355X.
    PMCE_SPLIT_OVERLAP=0 (read here and by pmce_model_create) restores the strictly serial schedule."""
    import os
    return os.environ.get("PMCE_SPLIT_OVERLAP", "1") != "0"
