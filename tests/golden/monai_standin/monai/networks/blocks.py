"""Stand-ins for monai.networks.blocks.{MLPBlock, PatchEmbed, UnetOutBlock, UnetrBasicBlock, UnetrUpBlock}
(MONAI 1.1.0 semantics, module/parameter names included).  See ../__init__.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Conv(nn.Module):
    """monai Convolution(conv_only=True): a module whose only child is `.conv`."""

    def __init__(self, cin, cout, k, stride=1, bias=False, transposed=False):
        super().__init__()
        if transposed:
            self.conv = nn.ConvTranspose3d(cin, cout, kernel_size=k, stride=stride, bias=bias)
        else:
            self.conv = nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2, bias=bias)

    def forward(self, x):
        return self.conv(x)


class UnetResBlock(nn.Module):
    """conv-IN-lrelu-conv-IN (+ 1x1 conv-IN on the residual when channels change) -> add -> lrelu.
    InstanceNorm3d(affine=False, eps=1e-5); LeakyReLU(0.01)."""

    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name):
        super().__init__()
        assert spatial_dims == 3 and norm_name == "instance"
        self.conv1 = _Conv(in_channels, out_channels, kernel_size, stride)
        self.conv2 = _Conv(out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=True)
        self.norm1 = nn.InstanceNorm3d(out_channels)
        self.norm2 = nn.InstanceNorm3d(out_channels)
        self.downsample = in_channels != out_channels or stride != 1
        if self.downsample:
            self.conv3 = _Conv(in_channels, out_channels, 1, stride)
            self.norm3 = nn.InstanceNorm3d(out_channels)

    def forward(self, inp):
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        out = self.norm2(self.conv2(out))
        if self.downsample:
            residual = self.norm3(self.conv3(residual))
        out = out + residual
        return self.lrelu(out)


class UnetrBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=False):
        super().__init__()
        assert res_block
        self.layer = UnetResBlock(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name)

    def forward(self, inp):
        return self.layer(inp)


class UnetrUpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name,
                 res_block=False):
        super().__init__()
        assert res_block
        self.transp_conv = _Conv(in_channels, out_channels, upsample_kernel_size, upsample_kernel_size, transposed=True)
        self.conv_block = UnetResBlock(spatial_dims, out_channels + out_channels, out_channels, kernel_size, 1, norm_name)

    def forward(self, inp, skip):
        out = self.transp_conv(inp)
        out = torch.cat((out, skip), dim=1)
        return self.conv_block(out)


class UnetOutBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, dropout=None):
        super().__init__()
        self.conv = _Conv(in_channels, out_channels, 1, 1, bias=True)

    def forward(self, inp):
        return self.conv(inp)


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=2, in_chans=1, embed_dim=48, norm_layer=None, spatial_dims=3):
        super().__init__()
        assert spatial_dims == 3 and norm_layer is None
        self.patch_size = tuple(patch_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = None

    def forward(self, x):
        _, _, d, h, w = x.size()
        p = self.patch_size
        if w % p[2] != 0:
            x = F.pad(x, (0, p[2] - w % p[2]))
        if h % p[1] != 0:
            x = F.pad(x, (0, 0, 0, p[1] - h % p[1]))
        if d % p[0] != 0:
            x = F.pad(x, (0, 0, 0, 0, 0, p[0] - d % p[0]))
        return self.proj(x)


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0, act="GELU", dropout_mode="vit"):
        super().__init__()
        assert act == "GELU" and dropout_rate == 0.0
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        self.fn = nn.GELU()
        self.drop1 = nn.Dropout(dropout_rate)
        self.drop2 = nn.Dropout(dropout_rate)

    def forward(self, x):
        return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))
