"""Host-side mirror of the reference's ``src/lib/model/networks`` package: same class names, constructor
signatures, forward signatures and ``state_dict`` keys; forwards call the HIP kernels through the C ABI."""
from .ConvRNN import CGRU_cell  # noqa: F401
from .encoder import Encoder  # noqa: F401
from .decoder import Decoder  # noqa: F401
from .head import YOLOXHead  # noqa: F401
from .model import ED  # noqa: F401
from .net_params import get_network_params  # noqa: F401
