"""ctypes binding of libfcmf_hip.so (C ABI declared in include/fcmf_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails the
product raises.  (The CPU restatement under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfcmf_hip.so")

F32, BF16, F64 = 0, 1, 2
BOX_FAST_TRIG = 0x100      # OR-ed into fcmf_box_bias_*'s coord_dtype (include/fcmf_hip.h FCMF_BOX_FAST_TRIG)
EPI_NONE, EPI_GELU, EPI_TANH, EPI_DGELU, EPI_DTANH, EPI_ADD = 0, 1, 2, 3, 4, 5
ERR_UNSUPPORTED = -3      # FCMF_ERR_UNSUPPORTED: callers with a documented fallback entry point test for it

_c = ctypes
_vp, _i, _i64, _f, _u64 = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_uint64


class AttnDesc(ctypes.Structure):
    """mirror of fcmf_attn_desc"""
    _fields_ = [
        ("dtype", _i), ("G", _i), ("heads", _i), ("d", _i), ("R", _i), ("T1", _i), ("T2", _i), ("group_div", _i),
        ("q_sg", _i64), ("q_sr", _i64), ("k1_sg", _i64), ("k1_st", _i64), ("k2_sg", _i64), ("k2_sr", _i64),
        ("k2_st", _i64), ("o_sg", _i64), ("o_sr", _i64),
        ("q", _vp), ("k1", _vp), ("v1", _vp), ("k2", _vp), ("v2", _vp),
        ("mask", _vp), ("bias", _vp),
        ("scale", _f), ("dropout_p", _f), ("seed", _u64), ("causal", _i), ("head_quirk", _i),
    ]


# name -> argtypes (restype is int unless noted); must list EVERY symbol of include/fcmf_hip.h
SIGNATURES = {
    "fcmf_abi_version": [],
    "fcmf_build_info": [],
    "fcmf_gemm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp],
    "fcmf_gemm_ctx_create": [_c.POINTER(_vp)],
    "fcmf_gemm_ctx_destroy": [_vp],
    "fcmf_gemm_ctx_set_workspace": [_vp, _vp, _i64],
    "fcmf_gemm_ctx_tune": [_vp, _i, _i, _i, _i64],
    "fcmf_gemm_ctx_last_kernel": [_vp],
    "fcmf_gemm_colblocks": [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _i, _i, _i64, _i, _vp],
    "fcmf_quant_fp8_rows": [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _vp],
    "fcmf_gemm_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _vp],
    "fcmf_colsum": [_vp, _vp, _i, _i, _i64, _i, _i, _vp],
    "fcmf_attn_small_fwd": [_c.POINTER(AttnDesc), _vp, _vp, _vp],
    "fcmf_attn_small_bwd": [_c.POINTER(AttnDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fcmf_attn_small_bwd_grouped": [_c.POINTER(AttnDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "fcmf_attn_mfma_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _f, _f, _u64, _vp],
    "fcmf_attn_mfma_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _f, _f,
                           _u64, _vp, _vp],
    "fcmf_add_ln_fwd": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _u64, _i, _vp],
    "fcmf_add_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _u64, _i, _vp],
    "fcmf_add_ln_fwd_fp8": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _u64, _i, _vp, _vp, _vp],
    "fcmf_add_ln_bwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _u64, _i, _vp, _vp, _vp],
    "fcmf_add_ln_bwd_workspace": [_i, _i],
    "fcmf_position_ids": [_vp, _vp, _i, _i, _i, _vp],
    "fcmf_embed_ln_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _u64, _i, _vp],
    "fcmf_embed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fcmf_embed_pos_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "fcmf_embed_pos_type_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "fcmf_box_bias_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "fcmf_box_bias_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "fcmf_box_embedding": [_vp, _i, _vp, _vp, _i, _i, _vp],
    "fcmf_xent_fwd": [_vp, _i64, _vp, _vp, _vp, _i, _i, _i64, _i, _vp],
    "fcmf_xent_bwd": [_vp, _i64, _vp, _vp, _i64, _vp, _f, _i, _i, _i64, _i, _vp],
    "fcmf_xent_mean": [_vp, _vp, _i, _i64, _f, _vp, _vp],
    "fcmf_additive_mask": [_vp, _i64, _vp, _i, _i, _f, _vp],
    "fcmf_cast": [_vp, _vp, _i64, _i, _i, _vp],
    "fcmf_cast_transpose": [_vp, _vp, _i, _i, _vp],
    "fcmf_multi_cast_transpose": [_vp, _vp, _vp, _vp, _i, _vp],
    "fcmf_dropout": [_vp, _vp, _i64, _f, _u64, _i, _vp],
    "fcmf_act_bwd": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "fcmf_sum_axis": [_vp, _vp, _i64, _i, _i64, _i, _vp],
    "fcmf_embed_scale_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _f, _i, _vp],
    "fcmf_embed_scale_bwd": [_vp, _vp, _vp, _i, _i, _i64, _vp, _f, _i, _vp],
    "fcmf_head_gather": [_vp, _vp, _i64, _i, _i, _i, _i, _i, _vp],
    "fcmf_multi_sumsq": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "fcmf_multi_adamw": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _c.POINTER(_f), _c.POINTER(_f), _i, _f, _f,
                         _f, _i, _vp, _f, _vp],
    "fcmf_bertadam": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _vp, _vp],
    "fcmf_conv_im2col": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _vp],
    "fcmf_bn_stats": [_vp, _vp, _i64, _i, _i, _i, _vp],
    "fcmf_bn_stats_workspace": [_i64, _i, _i],
    "fcmf_bn_finalize": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _f, _f, _vp],
    "fcmf_bn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "fcmf_conv_col2im": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fcmf_maxpool3x3s2_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "fcmf_adaptive_avgpool_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fcmf_bn_apply": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i64, _i, _i, _vp],
    "fcmf_bn_apply_pad": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i64, _i, _i, _i, _i, _i, _vp],
    "fcmf_conv_gemm": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fcmf_gemm_dw_batched": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _vp],
    "fcmf_conv_gemm_runs": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fcmf_pack_rgb0": [_vp, _i, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _vp],
    "fcmf_gemm_colstats": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _vp],
    "fcmf_conv_gemm_colstats": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fcmf_gemm_colstats_block_rows": [_vp, _i, _i, _i],
    "fcmf_bn_stats_blocks": [_vp, _vp, _i64, _i, _i, _i, _vp],
    "fcmf_bn_finalize_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _f, _f, _i, _i, _i, _i, _i, _vp],
    "fcmf_maxpool3x3s2": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "fcmf_adaptive_avgpool": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fcmf_dp_unique_id": [_vp],
    "fcmf_dp_comm_create": [_c.POINTER(_vp), _vp, _i, _i],
    "fcmf_dp_comm_destroy": [_vp],
    "fcmf_dp_allreduce_bucket": [_vp, _vp, _i64, _i, _i, _vp],
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` or "
                f"`make -C multimodal-aspect-category-sentiment-analysis_amd/csrc` (hipcc, gfx950). "
                f"fcmf_framework has no CPU/PyTorch fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = (ctypes.c_char_p if name in ("fcmf_build_info", "fcmf_gemm_ctx_last_kernel") else
                          ctypes.c_int64 if name in ("fcmf_add_ln_bwd_workspace", "fcmf_bn_stats_workspace") else ctypes.c_int)
        _lib = l
    return _lib


_ERR = {-1: "bad argument", -2: "kernel launch failed", -3: "unsupported configuration",
        -4: "RCCL unavailable or collective failed"}


def check(rc, what):
    if rc != 0:
        raise HipLibraryError(f"{what} failed: {_ERR.get(rc, rc)}")


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float64:
        return F64
    raise HipLibraryError(f"unsupported dtype {t.dtype}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    """raw hipStream_t of the current torch stream on the current device (the C accessor: torch.cuda.current_stream()
    builds a Stream object per call, ~8 us of host time, which is what an IAOG step of 2200 launches is made of)"""
    return _raw_stream(torch.cuda.current_device())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


# ---- GEMM contexts: one per (device, stream), created on first use ---------------------------------------------------------
SPLITK_WORKSPACE_BYTES = 512 << 20   # >= ksplit*M*N*4 of every weight-gradient GEMM of FCMF-base (66 MB for 768x768 x 28 splits)
_gemm_ctx = {}
_gemm_tuning = dict(tile=int(os.environ.get("FCMF_GEMM_TILE", "0")), kb=32 if os.environ.get("FCMF_GEMM_KB") == "32" else 64,
                    cus=int(os.environ.get("FCMF_GEMM_CUS", "256")), nt_min_mb=int(os.environ.get("FCMF_GEMM_NT_MIN_MB", "0")))


def _apply_tuning(h):
    t = _gemm_tuning
    check(lib().fcmf_gemm_ctx_tune(h, t["tile"], t["kb"], t["cus"], t["nt_min_mb"] << 20), "fcmf_gemm_ctx_tune")


def gemm_ctx(workspace=False):
    """the GEMM context of the current (device, stream): holds the split-K workspace (allocated when a weight-gradient GEMM
    first asks for it), the tuning knobs and the name of the last kernel -- the library itself has no global state"""
    key = (torch.cuda.current_device(), stream())
    ent = _gemm_ctx.get(key)
    if ent is None:
        h = ctypes.c_void_p()
        check(lib().fcmf_gemm_ctx_create(ctypes.byref(h)), "fcmf_gemm_ctx_create")
        _apply_tuning(h)
        ent = _gemm_ctx[key] = [h, None]
    if workspace and ent[1] is None:
        ent[1] = torch.empty(SPLITK_WORKSPACE_BYTES, dtype=torch.uint8, device=torch.device("cuda", key[0]))
        check(lib().fcmf_gemm_ctx_set_workspace(ent[0], ent[1].data_ptr(), ent[1].numel()), "fcmf_gemm_ctx_set_workspace")
    return ent[0]


def set_gemm_tuning(tile=None, kb=None, cus=None, nt_min_mb=None):
    """benchmark / test knobs (see fcmf_gemm_ctx_tune), applied to every existing and future context"""
    for k, v in (("tile", tile), ("kb", kb), ("cus", cus), ("nt_min_mb", nt_min_mb)):
        if v is not None:
            _gemm_tuning[k] = int(v)
    for h, _ in _gemm_ctx.values():
        _apply_tuning(h)


def drop_gemm_workspace():
    """unregister and free the split-K workspaces (tests: the float-atomic path)"""
    for ent in _gemm_ctx.values():
        if ent[1] is not None:
            check(lib().fcmf_gemm_ctx_set_workspace(ent[0], None, 0), "fcmf_gemm_ctx_set_workspace")
            ent[1] = None


def last_gemm_kernel():
    return lib().fcmf_gemm_ctx_last_kernel(gemm_ctx()).decode()


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise HipLibraryError("fcmf_framework ops run on the MI355X only: got a CPU tensor "
                                  "(move the model and batch to 'cuda'; there is no CPU fallback)")
