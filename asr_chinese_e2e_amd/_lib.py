"""Binding of libasr_hip.so (the C ABI declared in include/asr_hip.h): ctypes loads the library and checks every
symbol and the ABI version; launches go through the vectorcall trampolines of csrc/fastcall.c (`fast`).

There is deliberately NO fallback: if the shared library is missing or a symbol is absent the
import of the product path fails loudly.  Build it with
    make -C asr_chinese_e2e_amd/csrc        (or  python -c "import __graft_entry__ as g; g.build()")
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_uint32, c_void_p

# torch must be imported BEFORE the extension is dlopen'ed: libasr_hip.so then binds to the HIP
# runtime instance torch has already loaded, so stream handles and device pointers are shared.
# (Loaded the other way round the process ends up with two runtimes and every launch fails.)
import torch  # noqa: F401  (side effect: loads libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASR_HIP_LIB: another build of the SAME library (the diagnostic twin `make debug` produces); never a different implementation
LIB_PATH = os.environ.get("ASR_HIP_LIB") or os.path.join(_HERE, "libasr_hip.so")

ASR_F32, ASR_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU_MASK = 0, 1, 2
ABI_VERSION = 10
DROP_PRE, DROP_POST = 1, 2

P, I, F, Z, U = c_void_p, c_int, c_float, c_size_t, c_uint32

class TnProblem(ctypes.Structure):
    """asr_tn_problem of include/asr_hip.h."""
    _fields_ = [("dY", P), ("X", P), ("dW", P), ("dbias", P), ("M", I), ("N", I), ("K", I), ("ldy", I), ("ldx", I), ("ldw", I)]


TN_GROUP_MAX = 8


class LnReduceItem(ctypes.Structure):
    """asr_ln_reduce_item of include/asr_hip.h."""
    _fields_ = [("ws", P), ("dgamma", P), ("dbeta", P), ("dbias", P), ("rows", I)]


LN_REDUCE_MAX = 16


class DecLayerPlan(ctypes.Structure):
    """asr_dec_layer_plan of include/asr_hip.h (same field order)."""
    _fields_ = ([(n, I) for n in ("B", "To", "T", "d", "H", "dk", "ff")] + [("drop_p", F), ("seed", U * 5), ("ld_kv_c_T", I)] +
                [(n, P) for n in ("dec_len", "cross_len",
                                  "w_qkv_s", "w_fc_s", "w_q_c", "w_fc_c", "w_1", "w_2",
                                  "b_qkv_s", "b_fc_s", "b_q_c", "b_fc_c", "b_1", "b_2",
                                  "g_s", "be_s", "g_c", "be_c", "g_f", "be_f",
                                  "w_kv_c_T", "x_in",
                                  "qkv_s", "ctx_s", "a_s", "y_s", "lse_s", "rstd_s",
                                  "q_c", "kv_c", "kv_ready_event",
                                  "ctx_c", "a_c", "y_c", "lse_c", "rstd_c",
                                  "h", "o", "y_f", "rstd_f",
                                  "dz_f", "g_o", "g_h", "dx_f",
                                  "dz_c", "g_ac", "g_qc", "g_kvc", "dx_c",
                                  "dz_s", "g_as", "g_qkv", "dx_s",
                                  "dctx", "gb_2", "gb_fc_c", "gb_fc_s",
                                  "part_f", "part_c", "part_s", "delta")] +
                [("delta_bytes", Z), ("d_enc", P), ("wgrad_stream", P), ("aux_cus", I), ("ld_kv_c", I), ("kv_dgrad_cols", I), ("g_kv_group", P), ("ctx_s_lo", P), ("ctx_c_lo", P)])

# name -> (restype, argtypes); order and meaning exactly as in include/asr_hip.h
SIGNATURES = {
    "asr_abi_version": (I, []),
    "asr_last_error": (I, [c_char_p, Z]),
    "asr_stream_fork": (I, [P, P]),
    "asr_stream_arm": (I, [P, P]),
    "asr_stream_arm_pending": (I, []),
    "asr_stream_create": (I, [I, P]),
    "asr_set_option": (I, [c_char_p, I, P]),
    "asr_get_deterministic": (I, []),
    "asr_set_deterministic": (I, [I]),
    "asr_add_ln_fwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, F, U, I, I, P]),
    "asr_add_ln_bwd_workspace_bytes": (Z, [I, I]),
    "asr_add_ln_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, Z, I, I, I, F, U, I, I, P]),
    "asr_add_ln_bwd_reduce_batched": (I, [P, I, I, P]),
    "asr_sdpa_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, F, F, U, P, I, P]),
    "asr_sdpa_bwd_workspace_bytes": (Z, [I, I, I, I, I, I, I, I]),
    "asr_sdpa_bwd": (I, [P, P, P, P, P, P, P, Z, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, F, F, U, P, I, P]),
    "asr_dropout_mask": (I, [P, I, I, F, U, P]),
    "asr_sdpa_dropout_mask": (I, [P, I, I, I, I, F, U, P]),
    "asr_ctc_workspace_bytes": (Z, [I, I, I]),
    "asr_ctc_fwd_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, F, P, I, P, P, Z, I, P]),
    "asr_ctc_greedy_decode": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "asr_ctc_frame_argmax": (I, [P, P, P, I, I, I, I, I, I, P]),
    "asr_ctc_collapse": (I, [P, P, P, I, I, I, P]),
    "asr_decode_attn": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, F, I, P]),
    "asr_logsoftmax_topk": (I, [P, P, P, I, I, I, I, I, P]),
    "asr_ctc_frame_topk": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "asr_ctc_prefix_beam_workspace_bytes": (Z, [I, I, I]),
    "asr_ctc_prefix_beam": (I, [P, P, P, P, P, Z, P, P, P, I, I, I, I, I, I, I, P]),
    "asr_beam_step": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "asr_cache_gather": (I, [P, P, P, I, I, I, I, I, I, P]),
    "asr_xent_fwd_bwd": (I, [P, P, P, P, P, I, I, I, F, F, P, I, P]),
    "asr_dec_preprocess": (I, [P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P]),
    "asr_embed_pe_fwd": (I, [P, P, P, P, F, I, I, I, I, F, U, I, P]),
    "asr_embed_bwd": (I, [P, P, P, P, F, I, I, I, F, U, I, P]),
    "asr_relu_fwd": (I, [P, Z, I, P]),
    "asr_relu_bwd": (I, [P, P, P, P, Z, I, I, I, P]),
    "asr_colsum_workspace_bytes": (Z, [I, I]),
    "asr_colsum": (I, [P, P, P, Z, I, I, I, I, I, P]),
    "asr_cast": (I, [P, P, Z, I, I, P]),
    "asr_sumsq_workspace_bytes": (Z, [Z]),
    "asr_grad_sumsq": (I, [P, Z, P, P, Z, P]),
    "asr_noam_hyper": (I, [P, P, F, F, F, F, F, F, P]),
    "asr_grad_sumsq_noam": (I, [P, Z, P, P, Z, P, P, F, F, F, F, F, F, P]),
    "asr_adam_step": (I, [P, P, P, P, P, Z, P, P, F, F, F, F, I, P]),
    "asr_loss_combine": (I, [P, I, P, P, I, F, F, P, P]),
    "asr_gemm_nt_bf16": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "asr_gemm_small_bf16": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "asr_gemm_f32": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "asr_decoder_layer_fwd": (I, [P, P]),
    "asr_decoder_layer_bwd": (I, [P, P, P, P, P]),
    "asr_gemm_tn_workspace_bytes": (Z, [I, I, I]),
    "asr_gemm_tn_bf16": (I, [P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "asr_gemm_tn_bias_bf16": (I, [P, P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "asr_gemm_tn_grouped_bf16": (I, [P, I, I, P]),
    "asr_transpose_batched_bf16": (I, [P, P, P, I, P]),
    "asr_cer": (I, [P, P, I, I, P, P, I, I, P, P, I, I, I, I, P, P]),
    "asr_logmel_fwd": (I, [P, P, P, P, P, I, I, I, I, P]),
    "asr_utt_norm_lfr_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "asr_utt_norm_augment_lfr_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
}


class AsrHipError(RuntimeError):
    pass


def _check_runtime_env(env=None):
    """Refuse HIP runtime settings under which the multi-stream step cannot make progress.
    ROC_SYSTEM_SCOPE_SIGNAL=0: the step orders its streams by waits on completion signals of other queues (asr_stream_fork /
    asr_stream_arm: hipStreamWaitEvent on pooled events and on a dispatch packet's own completion signal); with system-scope signal
    completion switched off those waits never see the signal and the step hangs without an error (round 3: a run was cut after 7
    silent minutes).  Inherited from a user's environment it would hang the product silently - fail at import instead."""
    env = os.environ if env is None else env
    v = env.get("ROC_SYSTEM_SCOPE_SIGNAL")
    if v is not None and v.strip() == "0":
        raise RuntimeError("ROC_SYSTEM_SCOPE_SIGNAL=0 is set: the two-stream training step of asr_chinese_e2e_amd waits on completion signals "
                           "across HIP queues and would hang silently under it - unset the variable (or set it to 1)")


def _load():
    _check_runtime_env()
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `make -C asr_chinese_e2e_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"{LIB_PATH} does not export {name}: rebuild the extension") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.asr_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"libasr_hip.so ABI version {v}, binding expects {ABI_VERSION}")
    return lib


lib = _load()


def _bind_fast():
    """The launch path: one vectorcall trampoline per entry point (csrc/fastcall.c) instead of a ctypes foreign call
    (~0.3 us against ~6.4 us of host time per launch; the joint step issues ~600).  Same addresses, same argument
    meaning; pointers, stream handles and sizes are passed as plain Python ints (None = NULL)."""
    try:
        alt = os.environ.get("ASR_FASTCALL_DIR")      # another build of the same trampolines (the AddressSanitizer twin: `make asan`)
        if alt:
            import glob
            import importlib.machinery
            import importlib.util
            path = glob.glob(os.path.join(alt, "_asr_fastcall*.so"))[0]
            loader = importlib.machinery.ExtensionFileLoader("_asr_fastcall", path)
            _asr_fastcall = importlib.util.module_from_spec(importlib.util.spec_from_loader("_asr_fastcall", loader))
            loader.exec_module(_asr_fastcall)
        else:
            from . import _asr_fastcall
    except (ImportError, IndexError) as e:
        raise ImportError(f"asr_chinese_e2e_amd/_asr_fastcall*.so not found or not loadable ({e}): build it with "
                          "`make -C asr_chinese_e2e_amd/csrc` (it is part of the required native code; there is no slow path)") from e
    kind = {P: "P", I: "I", F: "F", Z: "Z", U: "U", c_char_p: "P"}

    class _Fast:
        pass
    ns = _Fast()
    for name, (res, args) in SIGNATURES.items():
        addr = ctypes.cast(getattr(lib, name), c_void_p).value
        setattr(ns, name, _asr_fastcall.make(addr, name, "".join(kind[a] for a in args), "Z" if res is Z else "I"))
    return ns


fast = _bind_fast()


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib.asr_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc, what=""):
    """Raise on a negative ASR_E* code, as the reference raises Python exceptions."""
    if rc != 0:
        names = {-1: "ASR_EINVAL", -2: "ASR_EDTYPE", -3: "ASR_EWORKSPACE", -4: "ASR_EHIP"}
        raise AsrHipError(f"{what}: {names.get(rc, rc)}: {last_error()}")
