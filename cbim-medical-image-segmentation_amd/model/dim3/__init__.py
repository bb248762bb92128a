from .unet import UNet  # noqa: F401
from .medformer import MedFormer  # noqa: F401
from .swin_unetr import SwinUNETR  # noqa: F401
from .unetpp import UNetPlusPlus  # noqa: F401
from .attention_unet import AttentionUNet  # noqa: F401
from .vnet import VNet  # noqa: F401
