"""Tensor-level wrappers over the C ABI: torch owns memory and streams, HIP kernels do the arithmetic.

Every function takes contiguous float32 CUDA(=HIP) tensors, passes ``data_ptr()`` + dims + the current
stream across the ABI and returns torch tensors.  No op has a PyTorch fallback.
"""
import os

import torch

from ._lib import check, lib

LRELU_SLOPE = 0.2   # get_activation("lrelu"), utils.py:63
NORM_EPS = 1e-5     # nn.GroupNorm / nn.LayerNorm default eps


MATRIX_MODES = {"fp32": 0, "bf16": 1, "fp32_mfma": 2, "fp32_cand": 3}


class matrix_mode:
    """``with ops.matrix_mode("bf16"):`` -- GEMM arithmetic of the launches enqueued inside (include/urnn_hip.h
    urnn_set_matrix_mode): "fp32" = the reference's semantics on the 16-bit matrix pipe (default), "bf16" = bf16 compute with fp32
    accumulation, "fp32_mfma" = the exact fp32 matrix instructions everywhere (slower), "fp32_cand" = the default except for the
    full-resolution cells' candidate GEMM on the fp32 instruction (a few per cent slower; plain-fp32 torch's long-rollout error)."""

    def __init__(self, mode):
        self.mode = MATRIX_MODES[mode]

    def __enter__(self):
        self.prev = lib().urnn_get_matrix_mode()
        check(lib().urnn_set_matrix_mode(self.mode), "urnn_set_matrix_mode")
        return self

    def __exit__(self, *exc):
        lib().urnn_set_matrix_mode(self.prev)
        return False


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


class _ExactGuard:
    """The launches inside run with URNN_MATRIX_FP32_MFMA when the process is in one of the f16-piece modes."""

    def __enter__(self):
        self.prev = lib().urnn_get_matrix_mode()
        if self.prev in (MATRIX_MODES["fp32"], MATRIX_MODES["fp32_cand"]):
            check(lib().urnn_set_matrix_mode(MATRIX_MODES["fp32_mfma"]), "urnn_set_matrix_mode")
        return self

    def __exit__(self, *exc):
        lib().urnn_set_matrix_mode(self.prev)
        return False


def exact_matrix_if(flag):
    """``with ops.exact_matrix_if(layer_has_a_weight_beyond_f16_range): launch(...)`` -- the f16 x 3 forward arithmetic is finite for
    |weight| < 64 (include/urnn_hip.h); a layer outside that range takes the exact fp32 matrix instruction instead of NaN."""
    return _ExactGuard() if flag else _NO_GUARD


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev_check(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("urnn_amd ops need tensors on the GPU (cuda:N / HIP device); there is no CPU path")
        if t.dtype != torch.float32:
            raise RuntimeError(f"urnn_amd ops are float32 only (got {t.dtype})")
        if not t.is_contiguous():
            raise RuntimeError("urnn_amd ops need contiguous tensors")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def tuning_env(name, default):
    """Development knobs (URNN_TUNE_*: A/B switches of the schedules) are read ONLY when URNN_TUNING=1 is set -- the product path
    takes every default whatever else the environment holds (the library's own knobs exist in -DURNN_TUNING builds only)."""
    if os.environ.get("URNN_TUNING") != "1":
        return default
    return os.environ.get(name, default)


def workspace(nbytes, device):
    """A scratch buffer of ``nbytes`` bytes (caller-owned, as the C ABI requires)."""
    # the first 16 KB of a cell / head workspace are STATUS words that kernels only ever OR into and the cooperative launches' barrier
    # words (include/urnn_hip.h URNN_STATUS_AREA_BYTES): only those need to start at zero -- the rest is scratch every kernel writes
    # before it reads (tens of MB per full-resolution cell)
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    buf[:min(STATUS_AREA_BYTES, int(nbytes))].zero_()
    return buf


class Arena:
    """Scratch buffers OWNED BY ONE engine / trainer, keyed by name (GroupNorm / LayerNorm partials, raw gates, backward
    scratch).  There is deliberately no process-wide pool: a captured hipGraph holds raw pointers into its scratch, so the
    buffers must live exactly as long as the owner's graphs.  Grow-only; growing a buffer bumps ``generation`` (an owner that
    captured graphs compares it and re-captures) and is refused while a capture is running."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._buf = {}
        self.generation = 0

    def get(self, key, nbytes):
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("scratch buffer %r must be sized before graph capture (run the path once eagerly)" % (key,))
            buf = workspace(nbytes, self.device)
            self._buf[key] = buf
            self.generation += 1
        return buf


def _scratch(ws, nbytes, device, key="ws"):
    """ws: None (a fresh buffer for this call), an Arena (its buffer ``key``) or a uint8 tensor of at least nbytes."""
    if ws is None:
        return workspace(nbytes, device)
    if isinstance(ws, Arena):
        return ws.get(key, nbytes)
    if ws.numel() < nbytes:
        raise RuntimeError(f"workspace of {ws.numel()} bytes given, {nbytes} needed")
    return ws


def gru_cell_workspace_bytes(B, F, H, W):
    return lib().urnn_gru_cell_workspace_bytes(B, F, H, W)


def _counter_check(*counters):
    for c in counters:
        if c is None or not c.is_cuda or c.dtype != torch.int32 or c.numel() != 1:
            raise RuntimeError("frame counters are int32 device scalars")


def head_workspace_bytes(B, C, H, W):
    return lib().urnn_head_workspace_bytes(B, C, H, W)


STATUS_AREA_BYTES = 16384                                               # include/urnn_hip.h URNN_STATUS_AREA_BYTES
STATUS_GATES, STATUS_CAND, STATUS_HEAD, STATUS_BARRIER = 1, 2, 4, 8     # include/urnn_hip.h: word 0 of a cell / head workspace
STATUS_NAMES = {STATUS_GATES: "GroupNorm sums of a cell's gates", STATUS_CAND: "GroupNorm sums of a cell's candidate",
                STATUS_HEAD: "LayerNorm sums of the head",
                STATUS_BARRIER: "a cooperative launch's grid barrier gave up (its blocks were not all resident: the GPU is shared, "
                                "partitioned or masked)"}


def workspace_status(ws):
    """Word 0 of a workspace's status area (0 = every norm's statistics were finite).  Reads the device: synchronises."""
    return int(ws[:4].view(torch.int32).item())


def max_abs(values):
    """max |v| of a float32 device tensor as a python float (urnn_max_abs_f32; NaN if any element is NaN)."""
    _dev_check(values)
    out = torch.empty(1, dtype=torch.float32, device=values.device)
    check(lib().urnn_max_abs_f32(_ptr(values), values.numel(), _ptr(out), _stream()), "urnn_max_abs_f32")
    return float(out.item())


# ---- weight packing ----------------------------------------------------------------------------------
def pack_conv(weight, bias):
    """nn.Conv2d (Cout,Cin,1,1) [+ bias] -> packed buffer for stage_conv."""
    _dev_check(weight, bias)
    Cout, Cin = weight.shape[0], weight.shape[1]
    L = lib()
    out = torch.empty(L.urnn_packed_conv_floats(Cin, Cout), dtype=torch.float32, device=weight.device)
    check(L.urnn_pack_conv_f32(_ptr(weight), _ptr(bias), _ptr(out), Cin, Cout, _stream()), "urnn_pack_conv_f32")
    return out


def pack_gru(W1, b1, W2, b2, I, F, skip):
    _dev_check(W1, b1, W2, b2)
    L = lib()
    out = torch.empty(L.urnn_packed_gru_floats(I, F, int(skip)), dtype=torch.float32, device=W1.device)
    check(L.urnn_pack_gru_f32(_ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(out), I, F, int(skip), _stream()),
          "urnn_pack_gru_f32")
    return out


def pack_deconv(weight, bias):
    """nn.ConvTranspose2d (Cin,Cout,2,2) + bias -> packed buffer for deconv2x2."""
    _dev_check(weight, bias)
    Cin, Cout = weight.shape[0], weight.shape[1]
    L = lib()
    out = torch.empty(L.urnn_packed_deconv_floats(Cin, Cout), dtype=torch.float32, device=weight.device)
    check(L.urnn_pack_deconv_f32(_ptr(weight), _ptr(bias), _ptr(out), Cin, Cout, _stream()), "urnn_pack_deconv_f32")
    return out


# ---- forwards ----------------------------------------------------------------------------------------
def stage_conv_stem_applies(B, Cin, Cout, H, W):
    """Can the flat conv to the head's 16 channels take the head's first LayerNorm statistics in its epilogue (urnn_stage_conv_stem_f32)?"""
    return bool(lib().urnn_stage_conv_stem_applies(B, Cin, Cout, H, W))


def stage_conv(x, packed, Cout, pool, out=None, slope=LRELU_SLOPE, head_w=None, head_partial0=None):
    """[AvgPool2](LeakyReLU(conv1x1(x))): x (B,Cin,H,W) -> (B,Cout,H[/2],W[/2]).  ``head_w`` (the head's conv_w) + ``head_partial0``
    (``head_tail_partial``): the decoder's last conv also leaves the head's first LayerNorm partials there (``stage_conv_stem_applies``)."""
    _dev_check(x, packed, out)
    B, Cin, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cout, H // 2, W // 2) if pool else (B, Cout, H, W), dtype=torch.float32, device=x.device)
    if head_w is not None:
        if pool or head_partial0 is None:
            raise ValueError("the head's statistics go with the flat conv and need their partial buffer")
        check(lib().urnn_stage_conv_stem_f32(_ptr(x), _ptr(packed), _ptr(out), B, Cin, Cout, H, W, slope, _ptr(head_w), _ptr(head_partial0),
                                             _stream()), "urnn_stage_conv_stem_f32")
        return out
    check(lib().urnn_stage_conv_f32(_ptr(x), _ptr(packed), _ptr(out), B, Cin, Cout, H, W, 1 if pool else 0, slope,
                                    _stream()), "urnn_stage_conv_f32")
    return out


PHASE_GATES, PHASE_GN1, PHASE_CAND, PHASE_GN2, PHASE_BLEND, PHASE_ALL = 1, 2, 4, 8, 16, 31
PHASE_COOP = 64        # modifier: the whole cell of a small plane as one cooperative launch (include/urnn_hip.h URNN_PHASE_COOP)
PHASE_FUSED_R = 32     # modifier: reset gate recomputed inside the candidate kernel (include/urnn_hip.h URNN_PHASE_FUSED_R)


def gru_cell(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, I, out=None, eps=NORM_EPS, phases=PHASE_ALL, ws=None):
    """ConvGRU (e None) / Skip-ConvGRU step.  x may be None (zeros, I channels).  out may be h (in place).
    ``phases`` enqueues a subset of the cell's five kernels (profiling only).  ``ws``: the cell's scratch (``_scratch``); the
    backward pass reads the forward's scratch, so training passes the same buffer to both."""
    _dev_check(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, out)
    B, F, H, W = h.shape
    if x is not None and tuple(x.shape) != (B, I, H, W):
        raise RuntimeError(f"gru_cell: x has shape {tuple(x.shape)}, expected {(B, I, H, W)}")
    if e is not None and tuple(e.shape) != (B, F, H, W):
        raise RuntimeError(f"gru_cell: e has shape {tuple(e.shape)}, expected {(B, F, H, W)}")
    L = lib()
    ws = _scratch(ws, L.urnn_gru_cell_workspace_bytes(B, F, H, W), h.device)
    if out is None:
        out = torch.empty_like(h)
    check(L.urnn_gru_cell_phases_f32(_ptr(x), _ptr(e), _ptr(h), _ptr(packed), _ptr(gn1_w), _ptr(gn1_b), _ptr(gn2_w),
                                     _ptr(gn2_b), _ptr(out), _ptr(ws), ws.numel(), B, I, F, H, W, eps, int(phases),
                                     _stream()), "urnn_gru_cell_f32")
    return out


def gru_cell_tail_applies(B, F, H, W, Cout, pool):
    """Can the end of a cell on (B, F, H, W) be fused with its consumer, a 1x1 conv F -> Cout (pool: + AvgPool2)?"""
    return bool(lib().urnn_gru_cell_tail_applies(B, F, H, W, Cout, 1 if pool else 0))


def head_tail_partial(B, H, W, device):
    """Buffer for the head's first LayerNorm statistics when the kernel that produces the feature map takes them (gru_cell_tail)."""
    return torch.zeros(lib().urnn_head_tail_partial_floats(B, H, W), dtype=torch.float32, device=device)


def gru_cell_tail(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, I, conv_packed, Cout, pool, conv_out=None, out=None, eps=NORM_EPS,
                  phases=PHASE_ALL, ws=None, head_w=None, head_partial0=None, slope=LRELU_SLOPE):
    """``gru_cell`` whose last kernel (GroupNorm finalize + blend) also runs the 1x1 conv that consumes the new state:
    returns (h', [AvgPool2](LeakyReLU(conv(h')))).  ``head_w`` (the head's stem conv (16,16)) + ``head_partial0``: the launch also
    takes the head's first LayerNorm statistics (``head(..., partial0=head_partial0)`` then skips its first pass)."""
    _dev_check(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, out, conv_packed, conv_out, head_w, head_partial0)
    B, F, H, W = h.shape
    L = lib()
    ws = _scratch(ws, L.urnn_gru_cell_workspace_bytes(B, F, H, W), h.device)
    if out is None:
        out = torch.empty_like(h)
    if conv_out is None:
        conv_out = torch.empty((B, Cout, H // 2, W // 2) if pool else (B, Cout, H, W), dtype=torch.float32, device=h.device)
    check(L.urnn_gru_cell_tail_f32(_ptr(x), _ptr(e), _ptr(h), _ptr(packed), _ptr(gn1_w), _ptr(gn1_b), _ptr(gn2_w), _ptr(gn2_b), _ptr(out),
                                   _ptr(ws), ws.numel(), B, I, F, H, W, eps, int(phases), _ptr(conv_packed), int(Cout), 1 if pool else 0,
                                   slope, _ptr(conv_out), _ptr(head_w), _ptr(head_partial0), _stream()), "urnn_gru_cell_tail_f32")
    return out, conv_out


def gru_cell_strip(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, I, global_pixels, exchange, out=None, eps=NORM_EPS, ws=None):
    """One horizontal strip of a cell step whose plane of ``global_pixels`` pixels is split over ranks (include/urnn_hip.h,
    "Spatial strips").  ``exchange(sums)`` must all-reduce (sum) the float64 device tensor in place; it is called twice."""
    _dev_check(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, out)
    B, F, H, W = h.shape
    L = lib()
    ws = _scratch(ws, L.urnn_gru_cell_workspace_bytes(B, F, H, W), h.device)
    if out is None:
        out = torch.empty_like(h)

    def phase(mask):
        check(L.urnn_gru_cell_strip_f32(_ptr(x), _ptr(e), _ptr(h), _ptr(packed), _ptr(gn1_w), _ptr(gn1_b), _ptr(gn2_w), _ptr(gn2_b),
                                        _ptr(out), _ptr(ws), ws.numel(), B, I, F, H, W, eps, mask, int(global_pixels), _stream()),
              "urnn_gru_cell_strip_f32")

    def swap(which, groups):
        sums = torch.empty((B, groups, 2), dtype=torch.float64, device=h.device)
        for direction in (0, 1):
            check(L.urnn_gru_cell_strip_stats_f32(_ptr(ws), ws.numel(), B, F, H, W, which, direction, _ptr(sums), _stream()),
                  "urnn_gru_cell_strip_stats_f32")
            if direction == 0:
                exchange(sums)

    phase(PHASE_GATES)
    swap(1, 2 * F // 32)
    phase(PHASE_CAND)
    swap(2, F // 32)
    phase(PHASE_GN2 | PHASE_BLEND)
    return out


def head_strip(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, cls_thred, global_pixels, exchange, want_raw=False, eps=NORM_EPS,
               slope=LRELU_SLOPE, ws=None):
    """One strip of the head: ln_w / ln_b are the strip's rows (5,C,H,W); three statistics exchanges (LayerNorm levels)."""
    _dev_check(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b)
    B, C, H, W = feat.shape
    L = lib()
    ws = _scratch(ws, L.urnn_head_workspace_bytes(B, C, H, W), feat.device)
    f32 = dict(dtype=torch.float32, device=feat.device)
    masked, cls = torch.empty((B, H, W), **f32), torch.empty((B, H, W), **f32)
    raw = torch.empty((B, H, W), **f32) if want_raw else None

    def phase(mask):
        check(L.urnn_head_strip_f32(_ptr(feat), _ptr(conv_w), _ptr(ln_w), _ptr(ln_b), _ptr(cls_w), _ptr(cls_b), _ptr(reg_w), _ptr(reg_b),
                                    _ptr(masked), _ptr(cls), _ptr(raw), None, _ptr(ws), ws.numel(), B, C, H, W, float(cls_thred), eps,
                                    slope, mask, int(global_pixels), _stream()), "urnn_head_strip_f32")

    for level in range(3):
        phase(1 << (2 * level))                                   # K1 / K2 / K3: convs + partial sums of the level's norms
        sums = torch.empty((1 if level == 0 else 2, B, 2), dtype=torch.float64, device=feat.device)
        for direction in (0, 1):
            check(L.urnn_head_strip_stats_f32(_ptr(ws), ws.numel(), B, C, H, W, level, direction, _ptr(sums), _stream()),
                  "urnn_head_strip_stats_f32")
            if direction == 0:
                exchange(sums)
        phase(2 << (2 * level))                                   # F1 / F2 / F3
    phase(64)                                                     # K4
    return masked, cls, raw


def deconv2x2(x, packed, Cout, out=None, slope=LRELU_SLOPE):
    """LeakyReLU(ConvTranspose2d(k=2,s=2)(x)): (B,Cin,H,W) -> (B,Cout,2H,2W)."""
    _dev_check(x, packed, out)
    B, Cin, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    check(lib().urnn_deconv2x2_f32(_ptr(x), _ptr(packed), _ptr(out), B, Cin, Cout, H, W, slope, _stream()),
          "urnn_deconv2x2_f32")
    return out


def head(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, cls_thred, out_masked=None, out_cls=None, out_raw=None,
         frame_index=None, want_raw=False, eps=NORM_EPS, slope=LRELU_SLOPE, ws=None, partial0=None, coop=False, frame_next=None):
    """Dual head + mask.  Returns (masked, cls, raw|None), each (B,H,W) unless preallocated (T,B,H,W) buffers
    plus a device ``frame_index`` are given.  ``frame_next`` (captured frame loops): the int32 device word that receives
    ``frame_index + 1`` -- the next head's ``frame_index``; two words used alternately are a frame counter without a kernel of its own
    (urnn_head_rollout_f32)."""
    _dev_check(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, out_masked, out_cls, out_raw)
    B, C, H, W = feat.shape
    L = lib()
    ws = _scratch(ws, L.urnn_head_workspace_bytes(B, C, H, W), feat.device)
    if out_masked is None:
        out_masked = torch.empty((B, H, W), dtype=torch.float32, device=feat.device)
    if out_cls is None:
        out_cls = torch.empty((B, H, W), dtype=torch.float32, device=feat.device)
    if out_raw is None and want_raw:
        out_raw = torch.empty((B, H, W), dtype=torch.float32, device=feat.device)
    if frame_next is not None:
        _dev_check(partial0)
        _counter_check(frame_index, frame_next)
        check(L.urnn_head_rollout_f32(_ptr(feat), _ptr(conv_w), _ptr(ln_w), _ptr(ln_b), _ptr(cls_w), _ptr(cls_b), _ptr(reg_w),
                                      _ptr(reg_b), _ptr(out_masked), _ptr(out_cls), _ptr(out_raw), _ptr(frame_index), _ptr(ws),
                                      ws.numel(), B, C, H, W, float(cls_thred), eps, slope, int(bool(coop) and partial0 is None),
                                      _ptr(partial0), _ptr(frame_next), _stream()), "urnn_head_rollout_f32")
        return out_masked, out_cls, out_raw
    if partial0 is not None:      # the first LayerNorm's statistics were taken by the kernel that produced feat (gru_cell_tail)
        _dev_check(partial0)
        check(L.urnn_head_after_tail_f32(_ptr(feat), _ptr(conv_w), _ptr(ln_w), _ptr(ln_b), _ptr(cls_w), _ptr(cls_b), _ptr(reg_w),
                                         _ptr(reg_b), _ptr(out_masked), _ptr(out_cls), _ptr(out_raw), _ptr(frame_index), _ptr(ws),
                                         ws.numel(), B, C, H, W, float(cls_thred), eps, slope, _ptr(partial0), _stream()),
              "urnn_head_after_tail_f32")
        return out_masked, out_cls, out_raw
    fn, what = (L.urnn_head_coop_f32, "urnn_head_coop_f32") if coop else (L.urnn_head_f32, "urnn_head_f32")   # coop: one cooperative launch
    check(fn(_ptr(feat), _ptr(conv_w), _ptr(ln_w), _ptr(ln_b), _ptr(cls_w), _ptr(cls_b), _ptr(reg_w),
             _ptr(reg_b), _ptr(out_masked), _ptr(out_cls), _ptr(out_raw), _ptr(frame_index), _ptr(ws),
             ws.numel(), B, C, H, W, float(cls_thred), eps, slope, _stream()), what)
    return out_masked, out_cls, out_raw


def preprocess(rain, cumsum, dem, imperv, manhole, dem_min, dem_max, t, nums, rain_max, cumsum_max, out=None, t_dev=None, t_next=None):
    """Per-frame input assembly: returns (B, 2*nums+3, H, W).  rain/cumsum (B,T) scalar or (B,T,H,W) spatial;
    dem/imperv/manhole (B,H,W).  ``t_dev`` (int32 device scalar) overrides ``t`` for graph replay; ``t_next``: the device word that
    receives ``t_dev + 1``, the next frame's ``t_dev`` (urnn_preprocess_rollout_f32)."""
    _dev_check(rain, cumsum, dem, imperv, manhole, out)
    B, H, W = dem.shape
    T = rain.shape[1]
    spatial = 1 if rain.dim() == 4 else 0
    C = 2 * nums + 3
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=dem.device)
    if t_next is not None:
        _counter_check(t_dev, t_next)
        check(lib().urnn_preprocess_rollout_f32(_ptr(rain), _ptr(cumsum), _ptr(dem), _ptr(imperv), _ptr(manhole), float(dem_min),
                                                float(dem_max), _ptr(out), _ptr(t_dev), _ptr(t_next), B, T, nums, H, W, spatial,
                                                float(rain_max), float(cumsum_max), _stream()), "urnn_preprocess_rollout_f32")
        return out
    check(lib().urnn_preprocess_f32(_ptr(rain), _ptr(cumsum), _ptr(dem), _ptr(imperv), _ptr(manhole), float(dem_min),
                                    float(dem_max), _ptr(out), int(t), _ptr(t_dev), B, T, nums, H, W, spatial,
                                    float(rain_max), float(cumsum_max), _stream()), "urnn_preprocess_f32")
    return out


def stage1_static(dem, imperv, manhole, dem_min, dem_max, weight, nums, out=None):
    """Static part of encoder stage 1 for scalar-rain events: S (B,Cout,H,W), once per event."""
    _dev_check(dem, imperv, manhole, weight, out)
    B, H, W = dem.shape
    Cout = weight.shape[0]
    if out is None:
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dem.device)
    check(lib().urnn_stage1_static_f32(_ptr(dem), _ptr(imperv), _ptr(manhole), float(dem_min), float(dem_max), _ptr(weight),
                                       _ptr(out), B, int(nums), Cout, H, W, _stream()), "urnn_stage1_static_f32")
    return out


def stage1_scalar_rain(S, rain, cumsum, weight, bias, t, nums, rain_max, cumsum_max, out=None, t_dev=None, slope=LRELU_SLOPE,
                       t_next=None):
    """preprocess_inputs + Encoder.stage1 for scalar rain: LeakyReLU(S + v_t) -> (B,Cout,H,W).  ``t_next`` as in preprocess."""
    _dev_check(S, rain, cumsum, weight, bias, out)
    B, Cout, H, W = S.shape
    if out is None:
        out = torch.empty_like(S)
    if t_next is not None:
        _counter_check(t_dev, t_next)
        check(lib().urnn_stage1_scalar_rain_rollout_f32(_ptr(S), _ptr(rain), _ptr(cumsum), _ptr(weight), _ptr(bias), _ptr(out),
                                                        _ptr(t_dev), _ptr(t_next), B, rain.shape[1], int(nums), Cout, H, W,
                                                        float(rain_max), float(cumsum_max), slope, _stream()),
              "urnn_stage1_scalar_rain_rollout_f32")
        return out
    check(lib().urnn_stage1_scalar_rain_f32(_ptr(S), _ptr(rain), _ptr(cumsum), _ptr(weight), _ptr(bias), _ptr(out), int(t),
                                            _ptr(t_dev), B, rain.shape[1], int(nums), Cout, H, W, float(rain_max),
                                            float(cumsum_max), slope, _stream()), "urnn_stage1_scalar_rain_f32")
    return out


def advance_counter(counter, delta=1):
    check(lib().urnn_advance_counter(_ptr(counter), int(delta), _stream()), "urnn_advance_counter")
