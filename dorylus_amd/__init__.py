"""dorylus_amd -- MI355X-native aggregation + transform engine behind the
reference's Engine / ResourceComm boundary (see include/dorylus_hip.h).

Only the hot path lives here: HIP kernels + C-ABI (csrc/), the C++ host mirror of
the reference's Engine stages and file formats (host/), and this thin ctypes layer.
"""
from ._lib import (BACKWARD, FORWARD, GAT, GATMH, GCN, LIB_PATH, SYMBOLS, Context,  # noqa: F401
                   DoryError, load)
from .engine import Chunk, Engine, NativeEngine  # noqa: F401
from .partition import Partition, read_features, read_labels, read_layer_config  # noqa: F401
