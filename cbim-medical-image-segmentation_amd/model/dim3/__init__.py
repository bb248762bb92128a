from .unet import UNet  # noqa: F401
from .medformer import MedFormer  # noqa: F401
