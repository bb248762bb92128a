"""cbim_amd — MI355X-native (gfx950) engine for the model/dim3 forward/backward hot path of
yhygao/CBIM-Medical-Image-Segmentation.

Host side: Python mirror of the reference's plugin surface (``model.utils.get_model``,
``training.losses.DiceLoss``) calling hand-written HIP kernels through the C ABI in
``include/cbim_hip.h`` (``libcbim_hip.so``).  See DESIGN.md / INTEGRATION.md.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def set_compute_dtype(dtype):
    """'bf16' (fast mode) or 'fp32' (parity mode) activations; default follows torch.autocast:
    bf16/fp16 autocast -> bf16, otherwise fp32."""
    from . import functional
    functional.set_compute_dtype(dtype)


def release_graph_refs():
    """After destroying the hipGraphs that captured training steps (before capturing again for another model / input shape):
    free the packed-weight tables and buffers that were kept alive for their replays (ops.PackedWeights.release_graphs)."""
    from . import ops
    ops.PACKED.release_graphs()
