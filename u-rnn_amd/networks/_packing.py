"""Packed-weight cache shared by the module mirrors: re-pack only when a parameter changed."""


def params_stamp(*params):
    """Cheap change detector: (storage pointer, in-place version counter) per parameter."""
    return tuple((p.data_ptr(), p._version) for p in params if p is not None)


# The default GEMM arithmetic carries weights as two f16 pieces of w * 2^10 (include/urnn_hip.h, URNN_MATRIX_FP32): finite for
# |w| < 2^16 / 2^10 = 64.  A layer with a larger weight runs on the exact fp32 matrix instruction instead (PackedCache.wide).
F16_WEIGHT_LIMIT = 63.9


class PackedCache:
    def __init__(self):
        self._stamp = None
        self._packed = None
        self.wide = False        # some |weight| >= F16_WEIGHT_LIMIT: the layer's launches take URNN_MATRIX_FP32_MFMA
        self.owner_checks = False    # an owner of all the parameters (training.Trainer) refreshes `wide` itself, once per event

    def get(self, params, pack_fn, weights=()):
        """``weights``: the tensors that become f16 pieces; their range is checked when they are (re)packed -- one small
        reduction and a host read per layer and weight change.  Skipped under stream capture and when an owner has taken the
        check over (``owner_checks``: ``training.Trainer.refresh_weight_ranges`` -- one reduction over its flat parameter
        buffer per event instead of a host synchronisation per layer and window)."""
        import torch
        stamp = params_stamp(*params)
        if self._packed is None or stamp != self._stamp:
            self._packed = pack_fn()
            self._stamp = stamp
            if weights and not self.owner_checks and not torch.cuda.is_current_stream_capturing():
                from .. import ops
                m = max(ops.max_abs(w.detach().contiguous()) for w in weights)
                self.wide = not (m < F16_WEIGHT_LIMIT)          # (NaN counts as out of range)
        return self._packed

    def clear(self):
        self._stamp = None
        self._packed = None                 # (wide is kept: it is refreshed by the next un-captured pack)
