"""inference/utils.py of the reference (/root/reference/inference/utils.py:1-44): dispatcher + window index."""


def get_inference(args):
    if args.dimension == "2d":
        raise NotImplementedError("cbim_amd: 2-D inference is outside the model/dim3 hot path")
    if args.dimension == "3d":
        from .inference3d import inference_sliding_window, inference_whole_image
        return inference_sliding_window if args.sliding_window else inference_whole_image
    raise ValueError("Error in image dimension")


def split_idx(half_win, size, i):
    """Window i starts at half_win*i and spans two half-windows; the last one is clamped to the volume end
    (utils.py:29-43)."""
    start = half_win * i
    end = start + half_win * 2
    if end > size:
        start, end = size - half_win * 2, size
    return start, end
