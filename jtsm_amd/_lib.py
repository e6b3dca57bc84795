"""ctypes binding of libjtsm_hip.so — the only way jtsm_amd reaches the GPU kernels.

There is no CPU fallback and no alternative backend: if the shared object is missing or a
call fails, a RuntimeError is raised (the reference raises RuntimeError from its C++
checks too, e.g. detectron2/layers/csrc/ROIAlign/ROIAlign.h:72-74 "Not compiled with GPU
support").  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JTSM_HIP_LIB") or os.path.join(_HERE, "lib", "libjtsm_hip.so")   # (override: A/B sweeps)
_lib = None

NCHW, NHWC = 0, 1


def lib():
    """Load (once) and return the CDLL.  Loading does not need a GPU; compute calls do."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "jtsm_amd: %s not found — build it with `python -m jtsm_amd.build` "
                "(there is no fallback path)" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.jtsm_last_error.restype = C.c_char_p
        _lib.jtsm_version.restype = C.c_char_p
        _lib.jtsm_event_create.restype = C.c_void_p
        _lib.jtsm_event_destroy.restype = None
        _lib.jtsm_conv_set_mid_event.restype = None
        _lib.jtsm_conv_set_splitk_fused.restype = None
        _lib.jtsm_moi_pool_workspace_bytes.restype = C.c_size_t
        for name in ("jtsm_mil_workspace_bytes", "jtsm_oicr_workspace_bytes", "jtsm_conv_workspace_bytes", "jtsm_conv_transpose2x2_workspace_bytes",
                     "jtsm_group_norm_workspace_bytes", "jtsm_semseg_ce_workspace_bytes",
                     "jtsm_conv_bf16x3_wgrad_workspace_bytes", "jtsm_conv_bf16x3_wgrad_bias_workspace_bytes", "jtsm_moi_pool_levels_workspace_bytes",
                     "jtsm_paint_sem_seg_workspace_bytes", "jtsm_mask_bce_workspace_bytes",
                     "jtsm_moi_pool_backward_levels_workspace_bytes", "jtsm_channel_sum_workspace_bytes",
                     "jtsm_pool_f16_workspace_bytes", "jtsm_moi_pool_f16_workspace_bytes",
                     "jtsm_conv_bf16x3_wgrad_group_workspace_bytes", "jtsm_image_labels_workspace_bytes",
                     "jtsm_roi_align_backward_levels_workspace_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_size_t
    if TIMING is not None:
        return _TimedLib(_lib)
    return _lib


# ---- optional per-entry-point timing (bench.py's roofline leg) ------------------------------------------------
# While TIMING is a list, every compute entry point called through `timed_lib()` is bracketed by hipEvents on the
# launch stream and logged as (name, EventSpan, algorithmic bytes or None).  Call sites that know their algorithmic
# traffic announce it with note_bytes() right before the call.  Nothing of this runs in normal operation.
TIMING = None
_pending_bytes = None
_UNTIMED = ("jtsm_event_", "jtsm_last_error", "jtsm_version", "jtsm_device_count", "jtsm_conv_set_mid_event",
            "jtsm_conv_set_splitk_fused", "jtsm_conv_plan", "jtsm_conv_bf16x3_plan", "jtsm_conv_bf16x3_eligible", "jtsm_conv_out_size",
            "jtsm_conv2d_", "jtsm_conv_transpose2x2_")   # (the contractions carry their own, finer instrumentation: layers/conv.py LAUNCH_LOG)


def note_bytes(nbytes):
    """Algorithmic HBM bytes of the NEXT library call (only looked at while TIMING is recording)."""
    global _pending_bytes
    if TIMING is not None:
        _pending_bytes = float(nbytes)


class EventSpan(object):
    __slots__ = ("a", "b")

    def __init__(self, raw):
        self.a, self.b = raw.jtsm_event_create(), raw.jtsm_event_create()

    def ms(self):
        out = C.c_float()
        return out.value if lib().jtsm_event_elapsed_ms(C.c_void_p(self.a), C.c_void_p(self.b), C.byref(out)) == 0 else 0.0

    def __del__(self):
        try:
            for e in (self.a, self.b):
                lib().jtsm_event_destroy(C.c_void_p(e))
        except Exception:
            pass


class _TimedLib(object):
    def __init__(self, raw):
        self._raw = raw

    def __getattr__(self, name):
        fn = getattr(self._raw, name)
        if TIMING is None or name.endswith("_workspace_bytes") or name.startswith(_UNTIMED):
            return fn
        raw = self._raw

        def timed(*args):
            global _pending_bytes
            span = EventSpan(raw)
            st = stream()
            raw.jtsm_event_record(C.c_void_p(span.a), st)
            rc = fn(*args)
            raw.jtsm_event_record(C.c_void_p(span.b), st)
            TIMING.append((name, span, _pending_bytes))
            _pending_bytes = None
            return rc
        return timed


def check(rc, what=""):
    if rc != 0:
        msg = lib().jtsm_last_error().decode("utf-8", "replace")
        raise RuntimeError("jtsm_hip %s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a tensor (or NULL for None) as c_void_p."""
    return C.c_void_p(0 if t is None else t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_device_index = None


def stream():
    """The current HIP stream of this process's device as a void*.  Called once per library launch (~950 per
    training step), so it goes through torch's raw accessor (one C call) rather than building a Stream object; one
    process drives one GPU (bench.py, DDP), whose index is looked up once."""
    global _device_index
    if _raw_stream is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if _device_index is None:
        _device_index = torch.cuda.current_device()
    return C.c_void_p(_raw_stream(_device_index))


def _check_device(t):
    """One process drives ONE GPU (stream() caches its index): a tensor on another device must not be launched on
    this device's stream."""
    if _device_index is not None and t.device.index is not None and t.device.index != _device_index:
        raise RuntimeError("jtsm_amd: this process launches on cuda:%d but got a tensor on %s (one process per GPU)"
                           % (_device_index, t.device))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "jtsm_amd operators run only on the HIP device (got a %s tensor); "
                "there is no CPU path in the product" % t.device)
        if t is not None:
            _check_device(t)


def f32(x):
    return C.c_float(float(x))


def f64(x):
    return C.c_double(float(x))


def is_nhwc(t):
    """True when a logical NCHW tensor is stored channels-last (and not also plain-contiguous
    in a way that makes the two layouts coincide ambiguously for C==1 / H*W==1)."""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not (
        t.is_contiguous() and t.shape[1] != 1 and t.shape[2] * t.shape[3] != 1)
