"""Plain torch.nn networks for the end-to-end driver: a ResNet-18 encoder, a Monodepth2-shaped depth decoder with
four disparity outputs, a pose decoder and a PoseCNN.  Random initialisation, no downloads, no torchvision.

Shapes follow the reference's network files: the encoder returns five feature maps with 64, 64, 128, 256 and 512
channels at strides 2 ... 32 after normalising the input with (x - 0.45) / 0.225
(Python/networks/resnet_encoder_torch.py:56-92); the depth decoder walks them from coarse to fine with
nearest-neighbour upsampling, reflection-padded 3x3 convolutions and skip connections and emits a sigmoid disparity
at scales 3, 2, 1, 0 (Python/networks/depth_decoder.py:152-182; the reference's own decoder assumes the
256-channel maps of its FPN backbone, the widths here are Monodepth2's 16 ... 256 for ResNet-18 features); the pose
decoder maps the last feature map of every frame to 0.01 * (axisangle, translation) of shape [B,F,1,3] through a
1x1 squeeze, two padded 3x3 convolutions, a 1x1 convolution and a spatial mean (interface and output scaling of
Python/networks/pose_decoder.py:26-66; that file's unpadded 7x7/5x5/5x5/3x3 stack is sized for the 48 x 160 maps of
its FPN backbone and cannot run on ResNet-18's 6 x 20 map, so the layers are Monodepth2's for ResNet features); PoseCNN is seven stride-2 convolutions on the stacked frames
(Python/networks/pose_cnn.py:13-50).  The convolutions run on stock PyTorch-ROCm (MIOpen); nothing here is part of
the HIP library.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class _BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.down is None else self.down(x)), inplace=True)


class ResNet18Encoder(nn.Module):
    num_ch_enc = (64, 64, 128, 256, 512)

    def __init__(self, num_input_images: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(3 * num_input_images, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        widths, layers, cin = (64, 128, 256, 512), [], 64
        for i, w in enumerate(widths):
            layers.append(nn.Sequential(_BasicBlock(cin, w, 1 if i == 0 else 2), _BasicBlock(w, w, 1)))
            cin = w
        self.layers = nn.ModuleList(layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, image) -> List[torch.Tensor]:
        x = F.relu(self.bn1(self.conv1((image - 0.45) / 0.225)), inplace=True)
        feats = [x]
        x = F.max_pool2d(x, 3, 2, 1)
        for layer in self.layers:
            x = layer(x)
            feats.append(x)
        return feats


class _Conv3x3(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3)

    def forward(self, x):
        return self.conv(F.pad(x, (1, 1, 1, 1), mode="reflect"))


class DepthDecoder(nn.Module):
    """Disparity at scales 0..3 ([B,1,H/2^s,W/2^s]) from the five encoder maps."""
    num_ch_dec = (16, 32, 64, 128, 256)

    def __init__(self, num_ch_enc: Sequence[int] = ResNet18Encoder.num_ch_enc, scales: Sequence[int] = (0, 1, 2, 3)):
        super().__init__()
        self.scales = list(scales)
        self.up0, self.up1, self.heads = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        for i in range(4, -1, -1):
            cin = num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.up0[str(i)] = _Conv3x3(cin, self.num_ch_dec[i])
            self.up1[str(i)] = _Conv3x3(self.num_ch_dec[i] + (num_ch_enc[i - 1] if i > 0 else 0), self.num_ch_dec[i])
        for s in self.scales:
            self.heads[str(s)] = _Conv3x3(self.num_ch_dec[s], 1)

    def forward(self, feats) -> Dict:
        out, x = {}, feats[-1]
        for i in range(4, -1, -1):
            x = F.elu(self.up0[str(i)](x), inplace=True)
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            if i > 0:
                x = torch.cat([x, feats[i - 1]], 1)
            x = F.elu(self.up1[str(i)](x), inplace=True)
            if i in self.scales:
                out[("disp", i)] = torch.sigmoid(self.heads[str(i)](x))
        return out


class PoseDecoder(nn.Module):
    """(axisangle, translation), each [B, num_frames_to_predict_for, 1, 3], from the last feature map of each of
    `num_input_features` inputs (each input is a list of feature maps, as the reference passes them)."""

    def __init__(self, num_input_features: int, num_frames_to_predict_for: int | None = None, num_ch_last: int = 512):
        super().__init__()
        self.n_out = num_input_features - 1 if num_frames_to_predict_for is None else num_frames_to_predict_for
        self.squeeze = nn.Conv2d(num_ch_last, 256, 1)
        self.c1 = nn.Conv2d(256 * num_input_features, 256, 3, 1, 1)
        self.c2 = nn.Conv2d(256, 256, 3, 1, 1)
        self.c3 = nn.Conv2d(256, 6 * self.n_out, 1)

    def forward(self, input_features):
        x = torch.cat([F.relu(self.squeeze(f[-1])) for f in input_features], 1)
        x = F.relu(self.c1(x))
        x = F.relu(self.c2(x))
        x = 0.01 * self.c3(x).mean(3).mean(2).view(-1, self.n_out, 1, 6)
        return x[..., :3], x[..., 3:]


class PoseCNN(nn.Module):
    def __init__(self, num_input_frames: int):
        super().__init__()
        self.n_out = num_input_frames - 1
        chans, cin, convs = (16, 32, 64, 128, 256, 256, 256), 3 * num_input_frames, []
        for i, c in enumerate(chans):
            k = 7 if i == 0 else 5 if i == 1 else 3
            convs += [nn.Conv2d(cin, c, k, 2, k // 2), nn.ReLU(inplace=True)]
            cin = c
        self.net = nn.Sequential(*convs)
        self.pose_conv = nn.Conv2d(256, 6 * self.n_out, 1)

    def forward(self, x):
        x = 0.01 * self.pose_conv(self.net(x)).mean(3).mean(2).view(-1, self.n_out, 1, 6)
        return x[..., :3], x[..., 3:]
