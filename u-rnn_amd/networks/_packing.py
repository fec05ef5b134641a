"""Packed-weight cache shared by the module mirrors: re-pack only when a parameter changed."""


def params_stamp(*params):
    """Cheap change detector: (storage pointer, in-place version counter) per parameter."""
    return tuple((p.data_ptr(), p._version) for p in params if p is not None)


class PackedCache:
    def __init__(self):
        self._stamp = None
        self._packed = None

    def get(self, params, pack_fn):
        stamp = params_stamp(*params)
        if self._packed is None or stamp != self._stamp:
            self._packed = pack_fn()
            self._stamp = stamp
        return self._packed

    def clear(self):
        self._stamp = None
        self._packed = None
