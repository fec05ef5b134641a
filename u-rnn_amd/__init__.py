"""urnn_amd -- MI355X-native U-RNN rollout engine (import name; the directory is ``u-rnn_amd``).

Host side mirrors the reference's model API (``ED`` / ``Encoder`` / ``Decoder`` / ``CGRU_cell`` /
``YOLOXHead`` forwards, ``Inference`` rollout); all arithmetic runs in hand-written HIP kernels
for gfx950 behind the C ABI declared in ``include/urnn_hip.h`` (``liburnn_hip.so``).
"""
__version__ = "0.1.0"
