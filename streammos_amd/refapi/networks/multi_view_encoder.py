"""BEV <-> range-view encoder with deformable-attention memory fusion (mirror of
networks/multi_view_encoder.py: ``CENet_Transformer`` :323-458 and what it instantiates).

``state_dict`` quirks reproduced on purpose: the two Unbalance blocks are
registered twice (as attributes and as ``header_bev.1`` / ``res1_bev.1``), and never-called modules keep
their keys (``up1``, ``up2``, ``self_attn`` / ``normx`` of every deformable layer, ``aux_head1-3`` are
computed but unused at inference).
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import backbone
from .. import deep_point
from ..deformattn.modules import MSDeformAttn


def VoxelMaxPool(pcds_feat, pcds_ind, output_size, scale_rate):
    out = deep_point.VoxelMaxPool(pcds_feat=pcds_feat.float(), pcds_ind=pcds_ind, output_size=output_size,
                                  scale_rate=scale_rate)
    return out.to(pcds_feat.dtype)


class AttMerge(nn.Module):
    """Attention-weighted merge of two scales (networks/multi_view_encoder.py:40-82).  Built only because
    the checkpoint carries its tensors (``up1.*``, ``up2.*``); CENet_Transformer.forward never calls it."""

    def __init__(self, cin_low, cin_high, cout, scale_factor):
        super().__init__()
        self.scale_factor, self.cout = scale_factor, cout
        self.att_layer = nn.Sequential(backbone.conv3x3(2 * cout, cout // 2), nn.BatchNorm2d(cout // 2), nn.ReLU(),
                                       backbone.conv3x3(cout // 2, 2, bias=True))
        self.conv_high = nn.Sequential(backbone.conv3x3(cin_high, cout), nn.BatchNorm2d(cout), nn.ReLU())
        self.conv_low = nn.Sequential(backbone.conv3x3(cin_low, cout), nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x_low, x_high):
        up = F.interpolate(x_high, scale_factor=self.scale_factor, mode="bilinear", align_corners=False)
        both = torch.stack((self.conv_low(x_low), self.conv_high(up)), dim=1)
        both = F.dropout(both, p=0.2, training=self.training)
        b, _, c, h, w = both.shape
        gate = F.softmax(self.att_layer(both.view(b, 2 * c, h, w)).view(b, 2, 1, h, w), dim=1)
        return (both * gate).sum(dim=1)


class DeformAttnLayer(nn.Module):
    """cross-attention on the memory stream + FFN (networks/multi_view_encoder.py:285-321)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=3, n_heads=8, n_points=4):
        super().__init__()
        if activation != "relu":
            raise RuntimeError("activation should be relu, not %s" % activation)
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)     # unused, owns checkpoint keys
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropoutx = nn.Dropout(dropout)
        self.normx = nn.LayerNorm(d_model)                                       # unused, owns checkpoint keys
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, query, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        q_in = query if pos is None else query + pos
        att = self.cross_attn(q_in, reference_points, src, spatial_shapes, level_start_index, padding_mask)
        query = self.norm1(query + self.dropout1(att))
        ffn = self.linear2(self.dropout2(F.relu(self.linear1(query))))
        return self.norm2(query + self.dropout3(ffn))


class DeformAttnModule(nn.Module):
    """Stack of layers sharing one set of cell-centre reference points (multi_view_encoder.py:245-273)."""

    def __init__(self, deformattn_layers, num_layers):
        super().__init__()
        self.deformattn_layers = nn.ModuleList([copy.deepcopy(deformattn_layers) for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        per_level = []
        for lvl, (h, w) in enumerate(spatial_shapes):
            h, w = int(h), int(w)
            ys = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32, device=device)
            xs = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32, device=device)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            gy = gy.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            gx = gx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            per_level.append(torch.stack((gx, gy), -1))
        ref = torch.cat(per_level, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, query, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None):
        ref = self.get_reference_points(spatial_shapes, valid_ratios, device=src.device)
        for layer in self.deformattn_layers:
            query = layer(query, src, pos, ref, spatial_shapes, level_start_index, padding_mask)
        return query


class BasicConv2d(nn.Module):
    """conv (no bias) + BN + LeakyReLU(0.01) -- networks/multi_view_encoder.py:460-476."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1, relu=True):
        super().__init__()
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size, stride=stride, padding=padding, dilation=dilation,
                              bias=False)
        self.bn = nn.BatchNorm2d(out_planes)
        self.relu = nn.LeakyReLU() if relu else None

    def forward(self, x):
        x = self.bn(self.conv(x))
        return self.relu(x) if self.relu is not None else x


class Unbalance_BasicBlock(nn.Module):
    """Asymmetric k0 x k1 and k1 x k0 branches, 3x3 fuse, residual -- multi_view_encoder.py:478-497."""

    def __init__(self, inplanes, kernel_size, padding):
        super().__init__()
        (k0, k1), (p0, p1) = kernel_size, padding

        def branch(k, p):
            return nn.Sequential(nn.Conv2d(inplanes, inplanes, k, padding=p, bias=False), nn.BatchNorm2d(inplanes),
                                 nn.ReLU())

        self.layer7x3 = branch((k0, k1), (p0, p1))
        self.layer3x7 = branch((k1, k0), (p1, p0))
        self.layer3x3 = nn.Sequential(nn.Conv2d(inplanes * 2, inplanes, 3, padding=1, bias=False),
                                      nn.BatchNorm2d(inplanes))

    def forward(self, x):
        y = self.layer3x3(torch.cat((self.layer7x3(x), self.layer3x7(x)), dim=1))
        return F.relu(y + x)


class CENet_Transformer(nn.Module):
    def __init__(self, base_block, context_layers, layers, nclasses, use_att):
        super().__init__()
        block = {"BasicBlock": backbone.BasicBlock}[base_block]
        layer = DeformAttnLayer(d_model=128, d_ffn=512, dropout=0.0, n_levels=1, n_heads=4, n_points=4)
        self.deformattn_module = DeformAttnModule(layer, 2)
        self.query_embed = nn.Embedding(64 * 64, 128)        # memory of frame 0 (multi_view_encoder.py:342)

        self.header_unbalance_conv = Unbalance_BasicBlock(32, kernel_size=(7, 3), padding=(3, 1))
        self.res1_unbalance_conv = Unbalance_BasicBlock(64, kernel_size=(5, 3), padding=(2, 1))
        c = context_layers
        self.header_bev = self._make_layer(block, c[0], c[1], layers[0], stride=2, use_att=use_att)
        self.header_bev[1] = self.header_unbalance_conv
        self.header_rv = self._make_layer(block, 32, c[1], layers[0] - 1, stride=1, use_att=use_att)
        self.res1_bev = self._make_layer(block, c[1] * 2, c[2], layers[1], stride=2, use_att=use_att)
        self.res1_bev[1] = self.res1_unbalance_conv
        self.res1_rv = self._make_layer(block, c[1] * 2, c[2], layers[1] - 1, stride=1, use_att=use_att)
        self.res2 = self._make_layer(block, c[2] * 2, c[3], layers[2], stride=2, use_att=use_att)

        fuse2 = c[3] + c[2]
        self.up2 = AttMerge(c[2], c[3], fuse2 // 2, scale_factor=2)
        fuse1 = fuse2 // 2 + c[1]
        self.up1 = AttMerge(c[1], fuse2 // 2, fuse1 // 2, scale_factor=2)
        self.out_channels = fuse1 // 2

        self.aux = True
        self.conv_1 = BasicConv2d(320, 128, kernel_size=3, padding=1)
        self.conv_2 = BasicConv2d(128, self.out_channels, kernel_size=3, padding=1)
        self.aux_head1 = nn.Conv2d(64, nclasses, 1)
        self.aux_head2 = nn.Conv2d(128, nclasses, 1)
        self.aux_head3 = nn.Conv2d(128, nclasses, 1)

        self.bev_grid2point_x0 = backbone.BilinearSample(in_dim=4, scale_rate=(0.5, 0.5))
        self.bev_grid2point_x1 = backbone.BilinearSample(in_dim=4, scale_rate=(0.25, 0.25))

    @staticmethod
    def _make_layer(block, in_planes, out_planes, num_blocks, stride=1, dilation=1, use_att=True):
        mods = [backbone.DownSample2D(in_planes, out_planes, stride=stride)]
        mods += [block(out_planes, dilation=dilation, use_att=False) for _ in range(num_blocks)]
        mods.append(block(out_planes, dilation=dilation, use_att=True))
        return nn.Sequential(*mods)

    def _cross_view(self, bev, bev_xy, sphere, sampler, rv_net, rv_size, bev_size, scale):
        """B2P gather -> P2R scatter -> range-view convs -> R2P gather -> P2B scatter
        (multi_view_encoder.py:395-405 and :410-420)."""
        pts = sampler(bev, bev_xy)
        rv = rv_net(VoxelMaxPool(pcds_feat=pts, pcds_ind=sphere, output_size=rv_size, scale_rate=scale))
        pts = sampler(rv, sphere)
        back = VoxelMaxPool(pcds_feat=pts, pcds_ind=bev_xy, output_size=bev_size, scale_rate=scale)
        return torch.cat((bev, back), dim=1), pts

    def forward(self, x, pcds_cood_cur, pcds_sphere_coord_cur, query_embed_store, use_query_store=False,
                return_query=False):
        x0 = self.header_bev(x)
        x0, _ = self._cross_view(x0, pcds_cood_cur, pcds_sphere_coord_cur, self.bev_grid2point_x0, self.header_rv,
                                 (32, 1024), (256, 256), (0.5, 0.5))
        x1 = self.res1_bev(x0)
        x1, x1_point = self._cross_view(x1, pcds_cood_cur, pcds_sphere_coord_cur, self.bev_grid2point_x1, self.res1_rv,
                                        (16, 512), (128, 128), (0.25, 0.25))
        x2 = self.res2(x1)

        # temporal fusion: the previous frame's x2 (or the learned embedding) queries the current x2
        bs, c, hh, ww = x2.shape
        shapes = torch.as_tensor([[hh, ww]], dtype=torch.long, device=x2.device)
        level_start = torch.zeros((1,), dtype=torch.long, device=x2.device)
        valid_ratios = torch.ones((bs, 1, 2), dtype=x2.dtype, device=x2.device)
        src = x2.flatten(2).transpose(2, 1)
        if use_query_store:
            query = query_embed_store.flatten(2).transpose(2, 1)
        else:
            query = self.query_embed.weight.unsqueeze(0).repeat(bs, 1, 1)
        fused = self.deformattn_module(query, src, shapes, level_start, valid_ratios)
        x2 = fused.transpose(2, 1).reshape(bs, c, hh, ww)

        size = x0.shape[2:]
        res = [F.interpolate(t, size=size, mode="bilinear", align_corners=True) for t in (x0, x1, x2)]
        out = self.conv_2(self.conv_1(torch.cat(res, dim=1)))
        if self.aux:
            res = [self.aux_head1(res[0]), self.aux_head2(res[1]), self.aux_head3(res[2])]
        if return_query:
            return out, x1_point, res[0], res[1], res[2], x2
        return out, res[0], res[1], res[2]
