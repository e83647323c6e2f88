"""Building blocks of the StreamMOS BEV / range-view encoder (mirror of networks/backbone.py).

Only the blocks reachable from the shipped configuration are provided.
Sub-module names and registration order reproduce the reference's ``state_dict`` keys exactly, which
is what makes the 474-tensor checkpoint loadable with ``strict=True``; cited per class.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


def _bn(c):
    return nn.BatchNorm2d(c)


def conv3x3(in_planes, out_planes, stride=1, dilation=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=bias)


def conv1x1(in_planes, out_planes, bias=False):
    return nn.Conv2d(in_planes, out_planes, 1, bias=bias)


class DownSample2D(nn.Module):
    """relu(bn(conv3x3 stride s) + maxpool3x3 stride s(bn(conv1x1)))  -- networks/backbone.py:14-34."""

    def __init__(self, in_planes, out_planes, stride=1):
        super().__init__()
        self.conv_branch = nn.Sequential(conv3x3(in_planes, out_planes, stride=stride), _bn(out_planes))
        self.pool_branch = nn.Sequential(conv1x1(in_planes, out_planes), _bn(out_planes),
                                         nn.MaxPool2d(3, stride=stride, padding=1))

    def forward(self, x):
        return F.relu(self.conv_branch(x) + self.pool_branch(x))


class ChannelAtt(nn.Module):
    """Squeeze-excite gate -- networks/backbone.py:87-102 (keys cnet.1 / cnet.3)."""

    def __init__(self, channels, reduction=4):
        super().__init__()
        self.cnet = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(channels, channels // reduction, 1), nn.ReLU(),
                                  nn.Conv2d(channels // reduction, channels, 1), nn.Sigmoid())

    def forward(self, x):
        return x * self.cnet(x)


class BasicBlock(nn.Module):
    """Two conv3x3+BN with a residual, optional channel gate -- networks/backbone.py:136-159."""

    def __init__(self, inplanes, reduction=1, dilation=1, use_att=True):
        super().__init__()
        mid = inplanes // reduction
        self.layer = nn.Sequential(conv3x3(inplanes, mid), _bn(mid), nn.ReLU(),
                                   conv3x3(mid, inplanes, dilation=dilation), _bn(inplanes))
        self.use_att = use_att
        if use_att:
            self.channel_att = ChannelAtt(inplanes, reduction=4)

    def forward(self, x):
        y = self.layer(x)
        if self.use_att:
            y = self.channel_att(y)
        return F.relu(y + x)


class PredBranch(nn.Module):
    """Dropout(0.2, train only) + 1x1 classifier with bias -- networks/backbone.py:188-196."""

    def __init__(self, cin, cout):
        super().__init__()
        self.pred_layer = nn.Sequential(nn.Conv2d(cin, cout, 1))

    def forward(self, x):
        return self.pred_layer(F.dropout(x, p=0.2, training=self.training))


class PointNet(nn.Module):
    """[BN] -> 1x1 conv -> BN -> [ReLU] over (B, C, N, 1) -- networks/backbone.py:199-231."""

    def __init__(self, cin, cout, pre_bn=False, post_act=True):
        super().__init__()
        mods = [_bn(cin)] if pre_bn else []
        mods += [conv1x1(cin, cout), _bn(cout)]
        if post_act:
            mods.append(nn.ReLU())
        self.layer = nn.Sequential(*mods)

    def forward(self, x):
        return self.layer(x)


class PointNetStacker(nn.Module):
    """networks/backbone.py:233-250."""

    def __init__(self, cin, cout, pre_bn=False, post_act=True, stack_num=1):
        super().__init__()
        if stack_num == 1:
            mods = [PointNet(cin, cout, pre_bn, post_act)]
        else:
            mods = [PointNet(cin, cout, pre_bn, True)]
            mods += [PointNet(cout, cout, False, True) for _ in range(stack_num - 2)]
            mods.append(PointNet(cout, cout, False, post_act))
        self.layer = nn.Sequential(*mods)

    def forward(self, x):
        return self.layer(x)


class CatFusion(nn.Module):
    """concat -> dropout(0.2, train only) -> two 1x1 conv+BN+ReLU -- networks/backbone.py:387-413."""

    def __init__(self, in_channel_list, out_channel):
        super().__init__()
        assert len(in_channel_list) >= 2
        self.in_channel_list, self.out_channel = in_channel_list, out_channel
        s = sum(in_channel_list)
        self.merge_layer = nn.Sequential(conv1x1(s, s // 2), _bn(s // 2), nn.ReLU(),
                                         conv1x1(s // 2, out_channel), _bn(out_channel), nn.ReLU())

    def forward(self, *x_list):
        x = F.dropout(torch.cat(x_list, dim=1), p=0.2, training=self.training)
        return self.merge_layer(x)


class BilinearSample(nn.Module):
    """Grid -> point bilinear gather, parameter-free -- networks/backbone.py:453-475.

    grid_feat (BS, C, H, W), grid_coord (BS, N, 2, S) -> (BS, C, N, S).  GPU tensors that do not need
    gradients use the HIP gather; anything else (CPU tensors, training) goes through ``F.grid_sample``
    with the reference's exact formulation.
    """

    def __init__(self, in_dim, scale_rate):
        super().__init__()
        self.scale_rate = scale_rate

    def forward(self, grid_feat, grid_coord):
        needs_grad = torch.is_grad_enabled() and (grid_feat.requires_grad or grid_coord.requires_grad)
        if grid_feat.is_cuda and not needs_grad and grid_coord.shape[-1] == 1 and grid_feat.dtype == torch.float32:
            return ops.bilinear_gather(grid_feat, grid_coord.float(), self.scale_rate).unsqueeze(-1)
        h, w = grid_feat.shape[2], grid_feat.shape[3]
        gx = (2 * grid_coord[:, :, 1] * self.scale_rate[1] / (w - 1)) - 1
        gy = (2 * grid_coord[:, :, 0] * self.scale_rate[0] / (h - 1)) - 1
        return F.grid_sample(grid_feat, torch.stack((gx, gy), dim=-1), mode="bilinear", padding_mode="zeros",
                             align_corners=True)


_REGISTRY = {"BilinearSample": BilinearSample, "CatFusion": CatFusion, "BasicBlock": BasicBlock}


def get_module(param_dic, **kwargs):
    """Config dict -> module (networks/backbone.py:37-44; a lookup table instead of ``eval``)."""
    for key, val in param_dic.items():
        if key != "type" and val is not None:
            kwargs[key] = val
    return _REGISTRY[param_dic["type"]](**kwargs)
