"""ctypes binding of libwseg.so (include/wseg.h).  The product path has NO CPU fallback: if the
library is missing or no gfx950 device is present, loading raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WSEG_LIB: measurement builds only (tools/stamps.py loads lib/libwseg_stamps<N>.so); the product path is the default
LIB_PATH = os.environ.get("WSEG_LIB") or os.path.join(_HERE, "lib", "libwseg.so")
ABI_VERSION = 5


class LogmelDesc(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("hop", C.c_int32), ("n_mels", C.c_int32), ("n_cols", C.c_int32),
                ("window", C.c_void_p), ("twiddle", C.c_void_p), ("mel_start", C.c_void_p),
                ("mel_count", C.c_void_p), ("mel_offset", C.c_void_p), ("mel_weight", C.c_void_p)]


class ModelConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "n_heads", "enc_layers", "dec_layers", "ffn", "vocab",
                                         "n_mels", "spec_cols", "enc_positions", "dec_positions", "dtype")]


class GenerateParams(C.Structure):
    _fields_ = [("prompt", C.c_int32 * 8), ("prompt_len", C.c_int32), ("eos_token_id", C.c_int32),
                ("pad_token_id", C.c_int32), ("max_length", C.c_int32), ("num_beams", C.c_int32),
                ("length_penalty", C.c_float), ("suppress_tokens", C.c_void_p), ("n_suppress", C.c_int32),
                ("begin_suppress_tokens", C.c_void_p), ("n_begin_suppress", C.c_int32),
                ("n_slots", C.c_int32), ("refill_min", C.c_int32), ("lookahead", C.c_int32),
                ("window_max_length", C.c_void_p), ("top_k", C.c_int32), ("top_p", C.c_float), ("seed", C.c_uint64),
                ("encoder_output", C.c_void_p)]


class GenerateStats(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("n_slots", C.c_int32), ("n_steps", C.c_int32), ("n_admissions", C.c_int32),
                ("slot_steps_active", C.c_int64), ("slot_steps_total", C.c_int64),
                ("queued_slot_steps_active", C.c_int64), ("queued_slot_steps_total", C.c_int64),
                ("kv_units_total", C.c_int32), ("kv_units_peak", C.c_int32), ("n_preemptions", C.c_int32), ("reserved_", C.c_int32)]


# name -> (restype, argtypes); every symbol include/wseg.h declares.
SYMBOLS = {
    "wseg_abi_version": (C.c_int, []),
    "wseg_last_error": (C.c_char_p, []),
    "wseg_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "wseg_logmel_scratch_bytes": (C.c_size_t, [C.POINTER(LogmelDesc), C.c_int32, C.c_int64]),
    "wseg_logmel_f32": (C.c_int, [C.POINTER(LogmelDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "wseg_resample_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_int64, C.c_void_p]),
    "wseg_model_create": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(C.c_void_p)]),
    "wseg_model_destroy": (None, [C.c_void_p]),
    "wseg_model_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "wseg_model_ready": (C.c_int, [C.c_void_p]),
    "wseg_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "wseg_workspace_bytes_kv": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "wseg_convert_operand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "wseg_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "wseg_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(GenerateParams), C.c_void_p, C.c_size_t,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "wseg_debug_first_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "wseg_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 4)]),
    "wseg_last_stats": (C.c_int, [C.c_void_p, C.POINTER(GenerateStats)]),
    "wseg_debug_gemm": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wseg_debug_gemm_out_is_mx": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "wseg_debug_gemm_resid_ln": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wseg_debug_lane_xor": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wseg_profile_begin": (C.c_int, []),
    "wseg_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


class WsegError(RuntimeError):
    pass


class WsegOutOfMemory(WsegError):
    """The decode workspace could not be allocated on the device (the only error Engine.generate answers with fewer slots)."""


def load(require_device=False):
    """dlopen libwseg.so and bind every symbol.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64; it must be in the process BEFORE libwseg.so is dlopen'ed so that
        # both resolve to ONE HIP runtime (two runtimes in one process cannot both own the device: the
        # second one reports "No HIP GPUs are available").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise WsegError(f"{LIB_PATH} not found: build it with `python -m whisperseg_amd.build` "
                            "(there is no CPU fallback for the MI355X path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.wseg_abi_version() != ABI_VERSION:
            raise WsegError("libwseg ABI version mismatch")
        _lib = lib
    if require_device:
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        check(_lib.wseg_device_info(name, 256, C.byref(cus)), _lib)
    return _lib


def check(status, lib=None):
    if status != 0:
        lib = lib or _lib
        msg = lib.wseg_last_error().decode() if lib is not None else ""
        raise WsegError(f"libwseg call failed ({status}): {msg}")
    return status


def stream_ptr():
    """Current torch HIP stream as a raw hipStream_t."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
