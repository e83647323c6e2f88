"""``models.StreamMOS.AttNet`` -- the streaming moving-object-segmentation network
(mirror of models/StreamMOS.py:22-202).

Same constructor argument (the ``ModelParam`` config class), the same 474-key ``state_dict`` and the
same ``infer(batch, i, query_embed_store)`` / ``stage_forward(...)`` contracts, so the reference's
``val_StreamMOS.py`` flow -- ``load_state_dict`` -> ``SyncBatchNorm.convert_sync_batchnorm`` -> DDP ->
optimizer -> ``.infer`` -- runs on this class unchanged.  The point<->grid scatters, the bilinear
gathers and the deformable-attention sampler underneath are the HIP kernels of libsmos_hip.so.
"""
import copy

import torch
import torch.nn as nn

from ..networks import backbone, multi_view_encoder
from ..networks.backbone import get_module
from .. import deep_point


def VoxelMaxPool(pcds_feat, pcds_ind, output_size, scale_rate):
    out = deep_point.VoxelMaxPool(pcds_feat=pcds_feat.float(), pcds_ind=pcds_ind, output_size=output_size,
                                  scale_rate=scale_rate)
    return out.to(pcds_feat.dtype)


class AttNet(nn.Module):
    def __init__(self, pModel):
        super().__init__()
        self.pModel = pModel
        vox = pModel.Voxel
        self.bev_shape = list(vox.bev_shape)
        self.rv_shape = list(vox.rv_shape)
        self.bev_wl_shape = self.bev_shape[:2]
        self.dx = (vox.range_x[1] - vox.range_x[0]) / vox.bev_shape[0]
        self.dy = (vox.range_y[1] - vox.range_y[0]) / vox.bev_shape[1]
        self.dz = (vox.range_z[1] - vox.range_z[0]) / vox.bev_shape[2]
        self.point_feat_out_channels = pModel.point_feat_out_channels
        self.build_network()
        self.fast_inference = True      # eval-mode GPU inference runs the fused engine (streammos_amd/engine.py)
        self.engine_layout = "cl"       # "cl" (channels-last, default) or "nchw"
        self.engine_miopen_search = True
        self._engine = None

    def build_network(self):
        p = self.pModel
        context = copy.deepcopy(p.BEVParam.context_layers)
        point_channels = context[0]
        context[0] = p.seq_num * context[0]          # T stacked scans share the BEV input (StreamMOS.py:74)
        self.point_pre = backbone.PointNetStacker(7, point_channels, pre_bn=True, stack_num=2)
        self.bev_net = multi_view_encoder.CENet_Transformer(p.BEVParam.base_block, context,
                                                            copy.deepcopy(p.BEVParam.layers), p.class_num, use_att=True)
        self.bev_grid2point = get_module(p.BEVParam.bev_grid2point, in_dim=self.bev_net.out_channels)
        fusion = {"CatFusion": backbone.CatFusion}[p.fusion_mode]
        self.point_post = fusion(in_channel_list=(point_channels, self.bev_net.out_channels, 64),
                                 out_channel=self.point_feat_out_channels)
        self.pred_layer = backbone.PredBranch(self.point_feat_out_channels, p.class_num)

    # ---- fused inference engine: holds folded copies of the weights, so anything that may change the
    # parameters (or the module tree) drops it; it is rebuilt lazily on the next eval-mode GPU call
    def invalidate_engine(self):
        self._engine = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None             # never pickled / deep-copied: it is a cache
        return state

    def _apply(self, fn, *args, **kwargs):
        self._engine = None
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        self._engine = None
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self._engine = None
        return super().load_state_dict(*args, **kwargs)

    def _engine_for(self, tensor):
        if not self.fast_inference or self.training or not tensor.is_cuda or torch.is_grad_enabled():
            return None
        if self._engine is None or self._engine.device != tensor.device or self._engine.layout != self.engine_layout:
            from ... import engine
            self._engine = engine.InferenceEngine(self, layout=self.engine_layout)
        self._engine.miopen_search = self.engine_miopen_search
        if getattr(self, "engine_sparse_stem", None) is not None:
            self._engine.sparse_stem = bool(self.engine_sparse_stem)
        return self._engine

    # ------------------------------------------------------------------------------------
    def stage_forward(self, point_feat, pcds_coord, pcds_sphere_coord, query_embed_store=None, use_query_store=False,
                      return_query=False):
        """point_feat (BS,T,C,N,1), pcds_coord (BS,T,N,3,1), pcds_sphere_coord (BS,T,N,2,1)
        -> pred_cls (BS,class_num,N,1), three BEV aux maps, new memory (BS,128,64,64)   [StreamMOS.py:86-113]"""
        eng = self._engine_for(point_feat)
        if eng is not None:
            return eng.stage_forward(point_feat, pcds_coord, pcds_sphere_coord,
                                     query_embed_store if use_query_store else None)
        bs, t, c, n, _ = point_feat.shape
        cur_xy = pcds_coord[:, 0, :, :2].contiguous()
        cur_sphere = pcds_sphere_coord[:, 0].contiguous()

        pts = self.point_pre(point_feat.view(bs * t, c, n, 1))
        bev = VoxelMaxPool(pcds_feat=pts, pcds_ind=pcds_coord.view(bs * t, n, 3, 1)[:, :, :2].contiguous(),
                           output_size=self.bev_wl_shape, scale_rate=(1.0, 1.0))
        bev = bev.view(bs, -1, self.bev_wl_shape[0], self.bev_wl_shape[1])
        bev_feat, point_feat_1, aux0, aux1, aux2, memory = self.bev_net(
            bev, cur_xy, cur_sphere, query_embed_store, use_query_store, True)
        point_bev = self.bev_grid2point(bev_feat, cur_xy)

        pts_cur = pts.view(bs, t, -1, n, 1)[:, 0].contiguous()
        fused = self.point_post(pts_cur, point_bev, point_feat_1)
        pred_cls = self.pred_layer(fused).float()
        return pred_cls, aux0, aux1, aux2, memory

    def infer(self, batch, i, query_embed_store=None):
        """One streamed scan.  ``batch`` tensors carry the DataLoader's leading dim of 1, which is squeezed
        here (StreamMOS.py:181-202); frame 0 starts from the learned memory embedding."""
        args = (batch["pcds_xyzi"].squeeze(0), batch["pcds_coord"].squeeze(0), batch["pcds_sphere_coord"].squeeze(0))
        if i == 0:
            return self.stage_forward(*args, return_query=True)
        return self.stage_forward(*args, query_embed_store=query_embed_store, use_query_store=True, return_query=True)

    # ---- training (row f2) --------------------------------------------------------------------
    def _seg_loss(self, pred, target):
        from . import losses
        mode = self.pModel.loss_mode
        if mode == "ohem":
            ce = losses.ohem_cross_entropy(pred, target, top_ratio=0.2, top_weight=4.0, ignore_index=0)
        elif mode == "ce":
            ce = torch.nn.functional.cross_entropy(pred, target.long(), ignore_index=0)
        else:
            raise Exception('loss_mode must in ["ce", "ohem"]')      # "wce" needs the dataset yaml, not on this path
        return ce + 3 * losses.lovasz_softmax(pred, target, ignore=0)

    def single_forward(self, batch, query_embed_store=None, use_query_store=False, return_query=False):
        """One training step of the chain: point loss + the mean of the three BEV auxiliary losses
        (models/StreamMOS.py:124-153)."""
        pred_cls, aux0, aux1, aux2, memory = self.stage_forward(batch["pcds_xyzi"], batch["pcds_coord"],
                                                                 batch["pcds_sphere_coord"], query_embed_store,
                                                                 use_query_store, return_query)
        bs, k = pred_cls.shape[0], pred_cls.shape[1]
        bev_target = batch["pcds_bev_target"].view(bs, -1, 1)
        loss = self._seg_loss(pred_cls, batch["pcds_target"])
        aux = sum(self._seg_loss(a.view(bs, k, -1).unsqueeze(-1), bev_target) for a in (aux0, aux1, aux2))
        return loss + aux / 3, memory

    def forward(self, batch):
        """Three consecutive samples chained through the memory (models/StreamMOS.py:155-179); returns the mean loss."""
        memory, total = None, 0
        for i in range(3):
            step = {k: batch["%s_%d" % (k, i)] for k in ("pcds_target", "pcds_bev_target", "pcds_xyzi", "pcds_coord",
                                                        "pcds_sphere_coord")}
            loss, memory = self.single_forward(step, query_embed_store=memory, use_query_store=i > 0, return_query=True)
            total = total + loss
        return total / 3
