"""ctypes binding of libuniadapter_hip.so (include/uniadapter.h).

The product path has NO fallback: if the library is missing this module raises at import, and
every wrapper raises RuntimeError(ua_strerror(code)) on a non-zero return.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libuniadapter_hip.so')

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f'{LIB_PATH} not found: build it with `python uni-adapter_amd/csrc/build.py` '
        '(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the hot path.')

_lib = C.CDLL(LIB_PATH)

_p, _i32, _i64, _f, _d = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class GemmEpilogue(C.Structure):
    """ua_gemm_epilogue (include/uniadapter.h)."""
    _fields_ = [('bias', _p), ('scale', _p), ('shift', _p),
                ('rowgroup_add', _p), ('ldg', _i64), ('group_rows', _i32),
                ('act', _i32),
                ('residual', _p), ('ldr', _i64),
                ('pool_out', _p), ('ldp', _i64), ('pool_rows', _i32),
                ('split_out', _p), ('ld_split', _i64), ('split_scale', _f),
                ('row_stats_out', _p), ('ld_stats', _i64), ('ln_mean_rstd', _p), ('ln_colsum', _p),
                ('ln_stats_in', _p), ('ln_stats_ld', _i64), ('ln_parts', _i32), ('ln_n', _i32), ('ln_eps', _f)]


class TtaState(C.Structure):
    """ua_tta_state (include/uniadapter.h)."""
    _fields_ = [('K', _i32), ('M', _i32), ('D', _i32),
                ('eps', _f), ('rho', _f), ('eta', _f), ('logit_scale', _f),
                ('res_learning', _i32), ('res_steps', _i32),
                ('lr', _d), ('beta1', _d), ('beta2', _d), ('adam_eps', _d),
                ('mu', _p), ('var', _p), ('pi', _p), ('c', _p), ('class_counts', _p),
                ('text0', _p),
                ('res', _p), ('adam_m', _p), ('adam_v', _p), ('e', _p), ('norm', _p),
                ('fhat', _p), ('faug', _p), ('clip_logits', _p), ('prob', _p), ('dota_scores', _p),
                ('L', _p), ('joint', _p), ('grad_ll', _p), ('grad_e', _p), ('loss', _p),
                ('inv_var', _p), ('row_const', _p), ('tickets', _p),
                ('align_mfma', _i32), ('t', _p), ('U', _p), ('align_ws', _p), ('align_ws_bytes', C.c_size_t),
                ('replay_lean', _i32)]


# name -> argtypes, exactly the declarations of include/uniadapter.h
SIGNATURES = {
    'ua_gmm_loglik': [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _f, _p, _p, _p, _p],
    'ua_gmm_loglik_bwd': [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _f, _p, _p],
    'ua_gmm_fit': [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _f, _p],
    'ua_gmm_step_fused': [_p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _f, _p],
    'ua_gmm_step_fused_prepare': [_p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _f, _p, _p, _p],
    'ua_clip_logits': [_p, _p, _i32, _i32, _f, _i32, _p, _p, _p, _p, _p, _p],
    'ua_clip_logits_pair': [_p, _p, _p, _i32, _i32, _f, _p, _p, _p, _p, _p],
    'ua_fuse_logits': [_p, _p, _p, _i32, _i32, _f, _f, _i32, _p, _p, _p, _p, _p],
    'ua_text_normalize': [_p, _p, _i32, _i32, _f, _p, _p, _p],
    'ua_align_loss': [_p, _p, _i32, _i32, _p, _p, _p],
    'ua_gmm_prepare_alignment': [_p, _p, _i32, _i32, _i32, _f, _p, _p, _p],
    'ua_align_forward': [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p],
    'ua_align_backward': [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p],
    'ua_residual_adam': [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _d, _d, _d, _d, _p],
    'ua_align_step': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _d, _d, _d, _d, _p, _p, _p, _p, _p],
    'ua_cluster_merge': [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f, _p],
    'ua_cluster_laplacian': [_p, _i32, _i32, _f, _f, _p, _p],
    'ua_cluster_cg_step': [_p, _p, _p, _p, _p, _i32, _i32, _i32, _f, _p, _p],
    'ua_cluster_logits': [_p, _i32, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p],
    'ua_dota_fit': [_p, _p, _p, _p, _p, _i32, _i32, _i32, _f, _p, _p, _p],
    'ua_spd_inverse_small': [_p, _i64, _i32, _p, _i64, _p],
    'ua_gmm_delta_pack': [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p],
    'ua_gmm_delta_merge': [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _i32, _i32, _i32, _p],
    'ua_vec_delta': [_p, _p, _i64, _p, _p],
    'ua_class_shard_pack': [_p, _p, _i32, _i32, _i32, _p, _p],
    'ua_fuse_logits_sharded': [_p, _p, _i32, _i32, _i32, _i32, _i32, _f, _f, _i32, _p, _p, _p, _p, _p, _p],
    'ua_vec_merge': [_p, _p, _i32, _i64, _i64, _f, _p, _p],
    'ua_to_half': [_p, _p, _i64, _p],
    'ua_dota_predict': [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p],
    'ua_fps': [_p, _i32, _i32, _i32, _i32, _f, _p, _p, _p],
    'ua_knn_group': [_p, _p, _p, _i32, _i32, _i32, _i32, _p, _i32, _p, _p],
    'ua_ball_group': [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f, _p, _i32, _p, _p],
    'ua_gemm': [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, C.POINTER(GemmEpilogue), _p],
    'ua_layernorm': [_p, _i64, _p, _p, _f, _p, _i64, _i32, _i32, _i32, _p],
    'ua_layernorm_split': [_p, _i64, _p, _p, _f, _p, _i64, _i32, _i32, _i32, _f, _p],
    'ua_split_f16': [_p, _i64, _p, _i64, _i32, _i32, _i32, _f, _p],
    'ua_gemm_split': [_p, _i64, _f, _p, _i64, _f, _p, _i64, _i32, _i32, _i32, C.POINTER(GemmEpilogue), _p],
    'ua_row_stats_finalize': [_p, _i32, _i64, _i32, _i32, _f, _p, _p],
    'ua_gemm_split_set_persistent': [_i32],
    'ua_attention': [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p],
    'ua_attention_cls': [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p],
    'ua_attention_split': [_p, _i64, _f, _p, _i64, _p, _i64, _f, _i32, _i32, _i32, _i32, _p],
    'ua_attention_split_cls': [_p, _i64, _f, _p, _i64, _p, _i64, _f, _i32, _i32, _i32, _i32, _p],
    'ua_assemble_tokens': [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p],
    'ua_add_pos': [_p, _p, _p, _i32, _i32, _i32, _p],
    'ua_cls_max_concat': [_p, _p, _i32, _i32, _i32, _p],
    'ua_tta_chunk': [C.POINTER(TtaState), _p, _p, _i32, _i64, _i64, _p, _p, _p, _p],
    'ua_gmm_mfma_operand': [_p, _p, _p, _i32, _i32, _i32, _f, _p, C.c_size_t, _p],
    'ua_align_mfma_prepare': [_p, _p, _p, _i32, _i32, _i32, _f, _p, C.c_size_t, _p],
    'ua_align_forward_mfma': [_p, _i32, _i32, _i32, _p, C.c_size_t, _p, _p, _p, _p, _p],
    'ua_align_step_mfma': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _d, _d, _d, _d, _p, C.c_size_t, _p, _p, _p],
    'ua_gmm_loglik_mfma': [_p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, C.c_size_t, _p],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(_lib, _name)       # AttributeError here = header and library out of sync
    _fn.argtypes = _args
    _fn.restype = C.c_int
_lib.ua_gmm_loglik_mfma_workspace_bytes.restype = C.c_size_t
_lib.ua_gmm_loglik_mfma_workspace_bytes.argtypes = [_i32, _i32, _i32, _i32]
_lib.ua_align_mfma_workspace_bytes.restype = C.c_size_t
_lib.ua_align_mfma_workspace_bytes.argtypes = [_i32, _i32, _i32]
_lib.ua_align_mfma_supported.restype = C.c_int
_lib.ua_align_mfma_supported.argtypes = [_i32, _i32, _i32]
_lib.ua_version.restype = C.c_int
_lib.ua_strerror.restype = C.c_char_p
_lib.ua_strerror.argtypes = [C.c_int]


def version():
    return _lib.ua_version()


def align_mfma_supported(K, M, D):
    return bool(_lib.ua_align_mfma_supported(K, M, D))


def align_mfma_workspace_bytes(K, M, D):
    return int(_lib.ua_align_mfma_workspace_bytes(K, M, D))


def mfma_workspace_bytes(B, K, M, D):
    return int(_lib.ua_gmm_loglik_mfma_workspace_bytes(B, K, M, D))


def strerror(code):
    return _lib.ua_strerror(code).decode()


def _check(code, name):
    if code != 0:
        raise RuntimeError(f'{name}: {strerror(code)} (code {code})')


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t, dtype=torch.float32, allow_none=False):
    """Raw device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        if allow_none:
            return None
        raise ValueError('tensor required')
    if not t.is_cuda:
        raise RuntimeError('uni_adapter_amd kernels need tensors on a HIP device (no CPU fallback)')
    if t.dtype != dtype:
        raise TypeError(f'expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError('tensor must be contiguous')
    return t.data_ptr()


def call(name, *args):
    _check(getattr(_lib, name)(*args), name)
