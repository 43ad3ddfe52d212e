"""ctypes binding of ``libspeakerguard_hip.so`` (C-ABI declared in include/speakerguard_hip.h).

There is no fallback: if the library is missing, or a call fails, this module raises.  Torch is
used by the callers only to own device memory and streams; every pointer handed over here is
``tensor.data_ptr()``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspeakerguard_hip.so")

SG_LOSS_ENTROPY, SG_LOSS_MARGIN, SG_LOSS_LINEAR = 0, 1, 2
SG_TASK = {"CSI": 0, "SV": 1, "OSI": 2}
SG_FLAG_WAV, SG_FLAG_RAW, SG_FLAG_CMVN = 0, 1, 2

# every symbol include/speakerguard_hip.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "sg_version", "sg_create", "sg_destroy", "sg_last_error", "sg_sync", "sg_xv_load", "sg_xv_set_enroll",
    "sg_xv_num_frames", "sg_input_scale", "sg_xv_mfcc", "sg_xv_cmvn", "sg_xv_forward", "sg_xv_debug_activation",
    "sg_xv_loss_grad", "sg_loss_eval", "sg_pgd_update", "sg_xv_pgd_run", "sg_xv_time_layer",
    "sg_cw2_step", "sg_nes_queries", "sg_nes_grad", "sg_fakebob_step",
    "sg_an_load", "sg_an_num_frames", "sg_an_logmel", "sg_an_forward", "sg_an_debug_activation", "sg_an_loss_grad",
    "sg_an_pgd_run", "sg_an_pgd_run_feco", "sg_conv1d_rows", "sg_wav_finalize", "sg_eer_threshold",
    "sg_xv_mfcc_backward", "sg_xv_cmvn_backward", "sg_feco_kmeans", "sg_feco_kmeans_seeded", "sg_feco_kmeans_compress", "sg_feco_compress_backward_reps", "sg_feco_compress", "sg_feco_compress_backward",
    "sg_an_logmel_backward", "sg_an_configure", "sg_xv_configure", "sg_xv_enroll_override", "sg_health", "sg_set_streamk", "sg_debug_lose_handoffs", "sg_debug_feco_epoch", "sg_feco_set_two_cu", "sg_trace_begin", "sg_trace_end",
)

# stage tags of sg_trace_end (include/speakerguard_hip.h); +l / -l = forward / data-gradient contraction of TDNN layer l
STAGE_NAMES = {10: "mfcc_fwd", 11: "cmvn_fwd", 12: "pool_fwd", 13: "fc1_fwd", 14: "tail", 15: "fc1_bwd", 16: "pool_bwd",
               17: "cmvn_bwd", 18: "mfcc_bwd", 19: "overlap_add"}
STAGE_NAMES.update({l: "tdnn%d_fwd" % l for l in range(1, 6)})
STAGE_NAMES.update({-l: "tdnn%d_dgrad" % l for l in range(1, 6)})
# AudioNet (SG_STAGE_AN_*): conv block l = conv(l + 2) of audionet_csine.py
STAGE_NAMES.update({20: "an_logmel_fwd", 21: "an_prefilter_fwd", 22: "an_pool_fwd", 23: "an_tail", 24: "an_pool_bwd",
                    25: "an_prefilter_bwd", 26: "an_logmel_bwd", 27: "an_overlap_add", 28: "an_feco_fwd", 29: "an_feco_bwd",
                    50: "an_cnn_fwd", 51: "an_cnn_bwd", 52: "an_cnn_fwdbwd"})
STAGE_NAMES.update({30 + l: "an_conv%d_fwd" % (l + 2) for l in range(7)})
STAGE_NAMES.update({40 + l: "an_conv%d_dgrad" % (l + 2) for l in range(7)})


class NativeError(RuntimeError):
    pass


class XvWeights(C.Structure):
    _fields_ = [
        ("tdnn_weight", C.c_void_p * 5), ("tdnn_bias", C.c_void_p * 5),
        ("bn_mean", C.c_void_p * 5), ("bn_var", C.c_void_p * 5),
        ("fc1_weight", C.c_void_p), ("fc1_bias", C.c_void_p), ("emb_mean", C.c_void_p), ("lda", C.c_void_p),
        ("plda_mean", C.c_void_p), ("plda_transform", C.c_void_p), ("plda_psi", C.c_void_p), ("enroll", C.c_void_p),
        ("D", C.c_int32), ("S", C.c_int32), ("bn_eps", C.c_float), ("threshold", C.c_float),
    ]


class AnWeights(C.Structure):
    _fields_ = [
        ("conv1_weight", C.c_void_p), ("conv1_bias", C.c_void_p), ("bn1", C.c_void_p * 4),
        ("conv_weight", C.c_void_p * 7), ("conv_bias", C.c_void_p * 7), ("bn_weight", C.c_void_p * 7),
        ("bn_bias", C.c_void_p * 7), ("bn_mean", C.c_void_p * 7), ("bn_var", C.c_void_p * 7),
        ("fc_weight", C.c_void_p), ("fc_bias", C.c_void_p), ("num_class", C.c_int32), ("bn_eps", C.c_float),
    ]


class LossSpec(C.Structure):
    _fields_ = [("loss", C.c_int32), ("task", C.c_int32), ("targeted", C.c_int32), ("clip_max", C.c_int32),
                ("confidence", C.c_float), ("threshold", C.c_float), ("coef_dev", C.c_void_p)]


class Dither(C.Structure):
    _fields_ = [("dither", C.c_float), ("seed", C.c_uint64), ("index_base", C.c_int64), ("noise_dev", C.c_void_p),
                ("row_base", C.c_int64), ("rep_rows", C.c_int32)]


class PgdParams(C.Structure):
    _fields_ = [("loss", LossSpec), ("step_size", C.c_float), ("max_iter", C.c_int32), ("grad_sign", C.c_int32),
                ("eot_size", C.c_int32), ("eot_batch_size", C.c_int32), ("dither", Dither)]


class FecoParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("max_iter", C.c_int32), ("random_init", C.c_int32), ("seed", C.c_uint64),
                ("index_base", C.c_int64)]


_lib = None


def load():
    """Load the shared library (once).  Raises NativeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "%s is missing: the HIP extension has not been built (run `make` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    # torch first: it ships its own libamdhip64; loaded afterwards it would be a SECOND HIP runtime beside the
    # system one this library links to, and the two do not share devices, streams or allocations (sg_create fails)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "sg_version": (C.c_int, []),
        "sg_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "sg_destroy": (None, [vp]),
        "sg_last_error": (C.c_char_p, [vp]),
        "sg_sync": (C.c_int, [vp, vp]),
        "sg_health": (C.c_int, [vp]),
        "sg_set_streamk": (C.c_int, [vp, i32]),
        "sg_debug_lose_handoffs": (C.c_int, [vp, i32]),
        "sg_debug_feco_epoch": (C.c_int, [vp, C.c_uint32]),
        "sg_xv_configure": (C.c_int, [vp, i32]),
        "sg_feco_set_two_cu": (C.c_int, [vp, i32]),
        "sg_trace_begin": (C.c_int, [vp, i32]),
        "sg_trace_end": (C.c_int, [vp, vp, vp, i32, C.POINTER(i32)]),
        "sg_xv_load": (C.c_int, [vp, C.POINTER(XvWeights)]),
        "sg_xv_set_enroll": (C.c_int, [vp, vp, i32, f32]),
        "sg_xv_enroll_override": (C.c_int, [vp, vp, i32]),
        "sg_xv_num_frames": (i32, [i32]),
        "sg_input_scale": (C.c_int, [vp, vp, i64, vp, vp]),
        "sg_xv_mfcc": (C.c_int, [vp, vp, i32, i32, vp, C.POINTER(Dither), vp, vp]),
        "sg_xv_cmvn": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "sg_xv_forward": (C.c_int, [vp, vp, i32, i32, i32, C.POINTER(Dither), vp, vp, vp, vp, vp]),
        "sg_xv_debug_activation": (C.c_int, [vp, i32, vp, i64, C.POINTER(i32), C.POINTER(i32), vp]),
        "sg_xv_loss_grad": (C.c_int, [vp, vp, vp, i32, i32, i32, C.POINTER(LossSpec), C.POINTER(Dither),
                                      vp, vp, vp, vp, vp]),
        "sg_pgd_update": (C.c_int, [vp, vp, vp, vp, vp, i64, f32, i32, vp]),
        "sg_loss_eval": (C.c_int, [vp, vp, vp, i32, i32, f32, C.POINTER(LossSpec), vp, vp, vp, vp]),
        "sg_xv_pgd_run": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, C.POINTER(PgdParams), vp, vp, vp, vp, vp, vp, vp]),
        "sg_cw2_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp, vp, vp]),
        "sg_nes_queries": (C.c_int, [vp, vp, i32, i32, i32, i32, f32, C.c_uint64, i64, i32, vp, vp, vp, vp]),
        "sg_nes_grad": (C.c_int, [vp, vp, i32, i32, i32, i32, C.c_uint64, i64, i32, vp, i32, f32, i32, vp, vp]),
        "sg_fakebob_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, vp]),
        "sg_an_load": (C.c_int, [vp, C.POINTER(AnWeights)]),
        "sg_an_num_frames": (i32, [i32]),
        "sg_an_logmel": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "sg_an_forward": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
        "sg_an_debug_activation": (C.c_int, [vp, i32, vp, i64, C.POINTER(i32), C.POINTER(i32), vp]),
        "sg_an_loss_grad": (C.c_int, [vp, vp, vp, i32, i32, i32, C.POINTER(LossSpec), vp, vp, vp, vp, vp]),
        "sg_an_pgd_run": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, C.POINTER(PgdParams), vp, vp, vp, vp, vp, vp, vp]),
        "sg_an_pgd_run_feco": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, C.POINTER(PgdParams), C.POINTER(FecoParams), vp, vp, vp,
                                         vp, vp, vp, vp]),
        "sg_xv_time_layer": (C.c_int, [vp, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(C.c_double), C.POINTER(i32), vp]),
        "sg_conv1d_rows": (C.c_int, [vp, vp, vp, vp, vp, vp] + [i32] * 10 + [vp]),
        "sg_wav_finalize": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp]),
        "sg_eer_threshold": (C.c_int, [vp, vp, i32, vp, i32, vp, vp]),
        "sg_xv_mfcc_backward": (C.c_int, [vp, vp, i32, i32, vp, C.POINTER(Dither), vp, vp, vp]),
        "sg_xv_cmvn_backward": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "sg_an_logmel_backward": (C.c_int, [vp, vp, i32, i32, vp, vp, i32, vp]),
        "sg_an_configure": (C.c_int, [vp, i32, i32, i32]),
        "sg_feco_kmeans": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
        "sg_feco_kmeans_seeded": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, C.c_uint64, C.c_int64, vp, vp]),
        "sg_feco_kmeans_compress": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, C.c_uint64, C.c_int64, i32, vp, vp, vp, vp]),
        "sg_feco_compress_backward_reps": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
        "sg_feco_compress": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "sg_feco_compress_backward": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    """Device/host pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


class Context:
    """One ``sg_ctx`` on one GPU."""

    def __init__(self, device_index=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.sg_create(int(device_index), C.byref(h))
        if rc != 0 or not h:
            raise NativeError("sg_create(device=%d) failed with code %d (no usable HIP device?)" % (device_index, rc))
        self.handle = h
        self.device_index = int(device_index)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.sg_last_error(self.handle)
            raise NativeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.handle, *args), name)


def current_stream_ptr(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
