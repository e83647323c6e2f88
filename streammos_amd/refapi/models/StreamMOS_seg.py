"""``models.StreamMOS_seg.AttNet`` -- stage-2 variant with the boundary-refinement head (mirror of
models/StreamMOS_seg.py:21-220).

Identical backbone; a second point head ``refine`` (CatFusion + PredBranch over the same three point features,
:21-30) predicts ``bf_pred_cls``.  ``stage_forward`` / ``infer`` return the reference's 6-tuple
``(pred_cls, bf_pred_cls, aux0, aux1, aux2, query_embed_store)``; the checkpoint has 488 tensors (+14 ``refine.*``).
Training loss: only on ``bf_pred_cls`` (:164-171) -- the stage-2 script freezes everything except ``refine``.
"""
import torch
import torch.nn as nn

from ..networks import backbone
from . import StreamMOS as _base


class Refine(nn.Module):
    def __init__(self, fusion_mode, point_fusion_channels, point_feat_out_channels, class_num):
        super().__init__()
        fusion = {"CatFusion": backbone.CatFusion}[fusion_mode]
        self.bf_point_post = fusion(in_channel_list=point_fusion_channels, out_channel=point_feat_out_channels)
        self.bf_pred_layer = backbone.PredBranch(point_feat_out_channels, class_num)

    def forward(self, point_feat_tmp_cur, point_bev_feat, point_feat_1):
        return self.bf_pred_layer(self.bf_point_post(point_feat_tmp_cur, point_bev_feat, point_feat_1)).float()


class AttNet(_base.AttNet):
    def build_network(self):
        super().build_network()
        p = self.pModel
        point_channels = p.BEVParam.context_layers[0]
        self.refine = Refine(p.fusion_mode, (point_channels, self.bev_net.out_channels, 64), self.point_feat_out_channels,
                             p.class_num)

    def stage_forward(self, point_feat, pcds_coord, pcds_sphere_coord, query_embed_store=None, use_query_store=False,
                      return_query=False):
        eng = self._engine_for(point_feat)
        if eng is not None:
            return eng.stage_forward(point_feat, pcds_coord, pcds_sphere_coord,
                                     query_embed_store if use_query_store else None)
        bs, t, c, n, _ = point_feat.shape
        cur_xy = pcds_coord[:, 0, :, :2].contiguous()
        cur_sphere = pcds_sphere_coord[:, 0].contiguous()
        pts = self.point_pre(point_feat.view(bs * t, c, n, 1))
        bev = _base.VoxelMaxPool(pcds_feat=pts, pcds_ind=pcds_coord.view(bs * t, n, 3, 1)[:, :, :2].contiguous(),
                                 output_size=self.bev_wl_shape, scale_rate=(1.0, 1.0))
        bev = bev.view(bs, -1, self.bev_wl_shape[0], self.bev_wl_shape[1])
        bev_feat, point_feat_1, aux0, aux1, aux2, memory = self.bev_net(bev, cur_xy, cur_sphere, query_embed_store,
                                                                        use_query_store, True)
        point_bev = self.bev_grid2point(bev_feat, cur_xy)
        pts_cur = pts.view(bs, t, -1, n, 1)[:, 0].contiguous()
        pred_cls = self.pred_layer(self.point_post(pts_cur, point_bev, point_feat_1)).float()
        bf_pred_cls = self.refine(pts_cur, point_bev, point_feat_1)
        return pred_cls, bf_pred_cls, aux0, aux1, aux2, memory

    def single_forward(self, batch, query_embed_store=None, use_query_store=False, return_query=False):
        out = self.stage_forward(batch["pcds_xyzi"], batch["pcds_coord"], batch["pcds_sphere_coord"], query_embed_store,
                                 use_query_store, return_query)
        return self._seg_loss(out[1], batch["pcds_bf_target"]), out[5]

    def forward(self, batch):
        memory, total = None, 0
        for i in range(3):
            step = {k: batch["%s_%d" % (k, i)] for k in ("pcds_bf_target", "pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
            loss, memory = self.single_forward(step, query_embed_store=memory, use_query_store=i > 0, return_query=True)
            total = total + loss
        return total / 3


def freeze_for_stage2(model):
    """The stage-2 recipe of train_StreamMOS_seg.py:165-174: every parameter frozen except the `refine` head (the
    stage-1 checkpoint is loaded with strict=False before).  Returns the trainable parameters."""
    for p in model.parameters():
        p.requires_grad = False
    for p in model.refine.parameters():
        p.requires_grad = True
    return [p for p in model.parameters() if p.requires_grad]
