"""ctypes binding of libmsgm_hip.so (the C ABI declared in include/msgm_hip.h).

Loading is lazy and LOUD: if the shared library is missing or a symbol is
absent the import of any kernel wrapper raises — there is no CPU / eager
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsgm_hip.so")

MSGM_OK = 0
SDE_SGM, SDE_MSGM_SPARSE, SDE_MSGM_DENSE = 0, 1, 2
PROC_REVERSE, PROC_FORWARD = 0, 1
RNG_STREAM_T, RNG_STREAM_EPS, RNG_STREAM_V, RNG_STREAM_DW, RNG_STREAM_ROWS, RNG_STREAM_USER = 0, 1, 2, 3, 4, 16


class MsgmError(RuntimeError):
    pass


class SdeT(C.Structure):
    _fields_ = [("kind", C.c_int32), ("beta_min", C.c_float), ("beta_max", C.c_float), ("T", C.c_float),
                ("t_epsilon", C.c_float), ("G", C.c_void_p), ("L_G", C.c_void_p)]


class ConvFuseT(C.Structure):
    """msgm_conv_fuse_t (include/msgm_hip.h)."""
    _fields_ = [("residual", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("in_act", C.c_int32),
                ("reserved", C.c_int32), ("tapmask_in", C.c_uint16 * 16), ("tapmask_out", C.c_uint16 * 8),
                ("chanstats", C.c_void_p)]


class ReduceJobT(C.Structure):
    """msgm_reduce_job_t (include/msgm_hip.h)."""
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p), ("stride", C.c_int64), ("n_elem", C.c_int64),
                ("n_elem2", C.c_int64), ("block_begin", C.c_int64), ("nslots", C.c_int32), ("C", C.c_int32), ("Ktot", C.c_int32),
                ("koff", C.c_int32), ("rowsP", C.c_int32), ("rows", C.c_int32), ("accumulate", C.c_int32), ("reserved", C.c_int32)]


class PackJobT(C.Structure):
    """msgm_pack_job_t (include/msgm_hip.h)."""
    _fields_ = [("W", C.c_void_p), ("Wp", C.c_void_p), ("sr", C.c_int64), ("sc", C.c_int64), ("st", C.c_int64),
                ("rows", C.c_int32), ("ncols", C.c_int32), ("col_off", C.c_int32), ("taps", C.c_int32),
                ("rowsP", C.c_int32), ("Ktot", C.c_int32), ("kp_off", C.c_int32), ("reserved", C.c_int32)]


class EmbJobT(C.Structure):
    """msgm_emb_job_t (include/msgm_hip.h)."""
    _fields_ = [("W", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("dout", C.c_void_p), ("dW", C.c_void_p),
                ("db", C.c_void_p), ("db2", C.c_void_p), ("co", C.c_int32), ("block_begin", C.c_int32)]


class ConvGeomT(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("N", "Hi", "Wi", "Ho", "Wo", "KH", "KW", "strideH", "padH", "strideW", "padW", "mode", "ups")]


class MlpParamsT(C.Structure):
    _fields_ = [("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("W3", C.c_void_p), ("b3", C.c_void_p), ("W4", C.c_void_p), ("b4", C.c_void_p),
                ("d", C.c_int32), ("premodule", C.c_int32)]


_P, _I64, _I32, _U64, _U32, _F, _D, _SZ = (C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_uint32, C.c_float,
                                          C.c_double, C.c_size_t)

# name -> (restype, argtypes); mirrors include/msgm_hip.h one to one
SIGNATURES = {
    "msgm_version": (C.c_int, []),
    "msgm_error_string": (C.c_char_p, [C.c_int]),
    "msgm_rng_advance": (C.c_int, [_P, _U64, _P]),
    "msgm_fill_uniform": (C.c_int, [_P, _I64, _P, _U32, _P]),
    "msgm_fill_normal": (C.c_int, [_P, _I64, _P, _U32, _P]),
    "msgm_perturb_vp": (C.c_int, [_P, _P, _P, _P, _I64, _I64, C.POINTER(SdeT), _P, _P, _P, _P]),
    "msgm_perturb_vp_at": (C.c_int, [_P, _P, _P, _I64, _I64, C.POINTER(SdeT), _P, _P, _P, _P]),
    "msgm_ssm_prep": (C.c_int, [_P, _P, _P, _P, _I64, _I64, C.POINTER(SdeT), _P, _P, _P]),
    "msgm_forward_step_index": (C.c_int, [_P, _P, _I64, _I32, _F, _P]),
    "msgm_rademacher": (C.c_int, [_P, _I64, _P, _P, _P]),
    "msgm_sde_stage": (C.c_int, [_P, _P, _F, _P, _P, _P, _P, _F, _P, _U64, _P, _P, _I64, _I64, C.POINTER(SdeT), _I32,
                                 _I32, _F, _F, _F, _P, _P, _F, _P, _P, _P]),
    "msgm_time_tick": (C.c_int, [_P, _P, _I64, _F, _P, _P, _I64, _P]),
    "msgm_time_tick_stage": (C.c_int, [_P, _P, _I64, _F, _F, _P, _P, _I64, _P]),
    "msgm_ssm_loss_diag": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, C.POINTER(SdeT), _F, _P]),
    "msgm_lincomb": (C.c_int, [_P, _P, _F, _P, _F, _P, _F, _I64, _P]),
    "msgm_rk4_combine": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "msgm_row_norm": (C.c_int, [_P, _P, _I64, _I64, _P]),
    "msgm_keep_rows": (C.c_int, [_P, _P, _P, _I32, _I64, _I64, _P]),
    "msgm_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _D, _D, _D, _D, _F, _I64, _P, _P]),
    "msgm_counter_inc": (C.c_int, [_P, _P]),
    "msgm_conv_forward": (C.c_int, [C.POINTER(ConvGeomT), _P, _I32, _P, _I32, _P, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _I32, _P]),
    "msgm_conv_forward_fused": (C.c_int, [C.POINTER(ConvGeomT), _P, _I32, _P, _I32, _P, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _I32, C.POINTER(ConvFuseT), _P]),
    "msgm_conv_wino_supported": (C.c_int, [C.POINTER(ConvGeomT), _I32, _I32, _I32]),
    "msgm_wino_pack_weights_batched": (C.c_int, [_P, _I32, _P]),
    "msgm_conv_forward_wino": (C.c_int, [C.POINTER(ConvGeomT), _P, _I32, _P, _I32, _P, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _I32, C.POINTER(ConvFuseT), _P]),
    "msgm_conv_b6_supported": (C.c_int, [C.POINTER(ConvGeomT), _I32, _I32, _I32]),
    "msgm_b6_split_weights": (C.c_int, [_P, _P, C.c_int64, _P]),
    "msgm_conv_forward_b6": (C.c_int, [C.POINTER(ConvGeomT), _P, _I32, _P, _I32, _P, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _I32, C.POINTER(ConvFuseT), _P]),
    "msgm_conv_input_transform_supported": (C.c_int, [C.POINTER(ConvGeomT), _I32, _I32, _I32]),
    "msgm_groupnorm_affine": (C.c_int, [_P, _I32, _P, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _F, _P, C.c_size_t, _P]),
    "msgm_conv_chanstats_slots": (C.c_int32, [C.POINTER(ConvGeomT), _I32, _I32, _I32, _I32]),
    "msgm_conv_wgrad_slabs": (C.c_int, [C.POINTER(ConvGeomT), _P, _P, _I32, _I32, _P, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _SZ,
                                        C.POINTER(ReduceJobT), C.POINTER(C.c_int32), _P]),
    "msgm_slot_reduce_batched": (C.c_int, [_P, _I32, _I64, _P]),
    "msgm_conv_small_cout_supported": (C.c_int, [C.POINTER(ConvGeomT), _I32, _I32, _I32]),
    "msgm_groupnorm_affine_chanstats": (C.c_int, [_P, _I32, _I32, _P, _I32, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _F, _P]),
    "msgm_conv_wgrad": (C.c_int, [C.POINTER(ConvGeomT), _P, _P, _I32, _I32, _P, _I32, _I32, _I32, _P, _I32,
                                  C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), _P]),
    "msgm_conv_wgrad_workspace": (_SZ, [C.POINTER(ConvGeomT), _I32, _I32, _I32, _I32]),
    "msgm_conv_wgrad_det": (C.c_int, [C.POINTER(ConvGeomT), _P, _P, _I32, _I32, _P, _I32, _I32, _I32, _P, _I32,
                                      C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), _P, _SZ, _P]),
    "msgm_colsum_workspace": (_SZ, [_I32, _I32, _I32]),
    "msgm_colsum_det": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _SZ, _P]),
    "msgm_pack_weight": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _I32, _I32, _P]),
    "msgm_unpack_weight": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _P]),
    "msgm_act_dual_forward": (C.c_int, [_I32, _P, _P, _I64, _I32, _P]),
    "msgm_act_dual_backward": (C.c_int, [_I32, _P, _P, _I64, _P]),
    "msgm_colsum": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "msgm_gather_row": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "msgm_add_row": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _F, _P]),
    "msgm_groupnorm_workspace": (_SZ, [_I32, _I32]),
    "msgm_groupnorm_dual_forward": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _SZ, _P]),
    "msgm_groupnorm_dual_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P, _P, _SZ, _P]),
    "msgm_groupnorm_dual_forward2": (C.c_int, [_P, _I32, _P, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P, _SZ, _P]),
    "msgm_groupnorm_dual_backward2": (C.c_int, [_P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F,
                                                _P, _SZ, _P]),
    "msgm_groupnorm_param_slots_bytes": (_SZ, [_I32, _I32, _I32]),
    "msgm_groupnorm_dual_backward_slots": (C.c_int, [_P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F,
                                                     _P, _P, _P, _SZ, _P, _SZ, C.POINTER(ReduceJobT), C.POINTER(C.c_int32), _P]),
    "msgm_emb_bank_forward": (C.c_int, [_P, _I32, _I32, _P, _I32, _I32, _I32, _P]),
    "msgm_emb_bank_backward": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _I32, _I32, _P]),
    "msgm_bmm": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I32, _P]),
    "msgm_bmm_dual": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64,
                                _F, _I32, _P]),
    "msgm_softmax_dual_forward": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P]),
    "msgm_softmax_dual_backward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P]),
    "msgm_pack_weights_batched": (C.c_int, [_P, _I32, _I32, _P]),
    "msgm_rbf_kernel": (C.c_int, [_P, _P, _I64, _I64, _I32, _P, _P, _P]),
    "msgm_attention_supported": (C.c_int, [_I32, _I32]),
    "msgm_attention_forward": (C.c_int, [_P, _P, _I64, _I32, _I32, _F, _P]),
    "msgm_attention_dual_supported": (C.c_int, [_I32, _I32]),
    "msgm_attention_dual_workspace": (_SZ, [_I64, _I32, _I32]),
    "msgm_attention_dual_forward": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _F, _P]),
    "msgm_attention_dual_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _P, _SZ, _P]),
    "msgm_timestep_embedding": (C.c_int, [_P, _P, _I32, _I32, _F, _P]),
    "msgm_timestep_embedding_dual": (C.c_int, [_P, _P, _I32, _I32, _F, _P]),
    "msgm_normalize_dual": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _F, _F, _P]),
    "msgm_flat_to_image": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "msgm_image_to_flat": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "msgm_sum2x2": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "msgm_mlp_forward": (C.c_int, [C.POINTER(MlpParamsT), _P, _P, _P, _I64, _P]),
    "msgm_mlp_em_step": (C.c_int, [C.POINTER(MlpParamsT), _P, _I64, C.POINTER(SdeT), _F, _F, _F, _P, _P, _U64, _P]),
    "msgm_mlp_em_loop": (C.c_int, [C.POINTER(MlpParamsT), _P, _I64, C.POINTER(SdeT), _P, _I32, _F, _F, _P, _U64, _P]),
    "msgm_mlp_ssm_workspace": (_SZ, [_I32, _I32]),
    "msgm_mlp_num_params": (_I64, [_I32, _I32]),
    "msgm_mlp_ssm_partial": (C.c_int, [C.POINTER(MlpParamsT), _P, _P, _P, _P, _P, _I64, C.POINTER(SdeT), _F, _P, _P, _SZ,
                                       C.POINTER(C.c_int32), _P]),
    "msgm_mlp_ssm_reduce": (C.c_int, [_I32, _I32, _P, _I32, _F, _P, _P, _P]),
    "msgm_mlp_ssm_reduce_adam": (C.c_int, [_I32, _I32, _P, _I32, _F, _P, _P, _P, _P, _P, _D, _D, _D, _D, _P, _P, _P]),
    "msgm_mlp_ssm_grad": (C.c_int, [C.POINTER(MlpParamsT), _P, _P, _P, _P, _P, _I64, C.POINTER(SdeT), _F, _P, _P, _P, _P, _SZ, _P]),
    "msgm_ssm_terms": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, C.POINTER(SdeT), _P]),
    "msgm_ssm_loss": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """The loaded library; raises MsgmError when it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MsgmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise MsgmError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise MsgmError(f"{LIB_PATH} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != MSGM_OK:
        raise MsgmError(f"{what} failed: {lib().msgm_error_string(rc).decode()} ({rc})")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous CUDA(=HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MsgmError("libmsgm_hip kernels need device tensors (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise MsgmError("libmsgm_hip kernels need contiguous tensors")
    return t.data_ptr()


def f32(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if t.dtype != torch.float32:
        raise MsgmError(f"{name} must be float32 (got {t.dtype})")
    return t


_HOST_SCALARS = {}


def host_scalar(t) -> float:
    """Value of a one-element tensor (the horizon T: an nn.Parameter(requires_grad=False) on the device upstream,
    MSGM_higherDim.py:728) WITHOUT a device synchronisation per call: read once, cached per (storage address, version
    counter) — an in-place update (load_state_dict) bumps the version and is read again.  Needed because the reference's
    ``sde.T.item()`` (sde_scheme.py:54-57) sits inside code that the trainers capture into a hipGraph."""
    if not torch.is_tensor(t):
        return float(t)
    if t.device.type == "cpu":
        return float(t.item())
    key = (t.device.index, t.data_ptr(), t._version)
    v = _HOST_SCALARS.get(key)
    if v is None:
        if torch.cuda.is_current_stream_capturing():
            raise MsgmError("a device scalar (the horizon T) is read for the first time inside a graph capture; "
                            "run one eager step first")
        v = _HOST_SCALARS[key] = float(t.item())
    return v


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def sde_struct(kind: int, beta_min: float, beta_max: float, T: float, t_epsilon: float,
               G: Optional[torch.Tensor] = None, L_G: Optional[torch.Tensor] = None) -> SdeT:
    return SdeT(kind, float(beta_min), float(beta_max), float(T), float(t_epsilon), ptr(G), ptr(L_G))


class PhiloxState:
    """Device-resident Philox state consumed by kernels that draw noise: uint64[4] = {seed, offset, row_base,
    elem_base}.  ``set_shard(row_base, n)`` places this rank's rows inside the global index space of a data-parallel
    run (csrc/common.h), so a sharded run draws the numbers the single-GPU run draws for the same rows."""

    def __init__(self, seed: int, device, offset: int = 0, row_base: int = 0, n: int = 0):
        # int64 storage, reinterpreted as uint64 by the kernels
        self.state = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, offset, 0, 0], dtype=torch.int64, device=device)
        if row_base:
            self.set_shard(row_base, n)

    def ptr(self):
        return self.state.data_ptr()

    def set_shard(self, row_base: int, n: int) -> None:
        # k_fill / the prep kernels address draws by QUADS of the global index: row-indexed streams (t, the latent rows)
        # by row quads, element-indexed ones by element quads.  A base that is not a multiple of 4 would be truncated
        # and this shard would re-draw the previous rank's numbers.
        if row_base % 4 or (row_base * n) % 4:
            raise MsgmError("shard base: row_base (and row_base * n) must be multiples of 4 — Philox draws are addressed "
                            "by quads of the GLOBAL row / element index")
        self.state[2:].copy_(torch.tensor([row_base, row_base * n], dtype=torch.int64))

    def advance(self, n: int = 1) -> None:
        check(lib().msgm_rng_advance(self.ptr(), n, stream()), "msgm_rng_advance")

    def state_dict(self) -> dict:
        seed, offset, row_base, elem_base = (int(v) for v in self.state.tolist())
        return {"seed": seed, "offset": offset, "row_base": row_base, "elem_base": elem_base}

    def load_state_dict(self, sd: dict) -> None:
        """Restores the STREAM position (seed, offset) only.  The shard base {row_base, elem_base} belongs to the rank
        that loads, not to the rank that wrote the file: in a data-parallel resume every rank loads rank 0's checkpoint
        and must keep drawing the numbers of ITS rows (set_shard), otherwise all ranks would replay rank 0's noise."""
        self.state[:2].copy_(torch.tensor([sd["seed"], sd["offset"]], dtype=torch.int64))
