"""Plain-PyTorch fp32 CPU restatement of ``AttNet.stage_forward`` -- TEST INFRASTRUCTURE.

A functional re-expression of the reference's inference graph driven directly by a flat
``state_dict`` (474 keys).  It deliberately shares no code with
``streammos_amd``: scatter, gather and deformable sampling go through ``torch`` primitives /
``oracle.ops_np``-equivalent formulations so that it can check the HIP path independently.

Pinned by tests/golden/e2e_*.npz (outputs of the real reference on the same seeded weights and
synthetic scans) in tests/test_oracle_golden.py.  Reference lines are cited per block.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LN_EPS = 1e-5


class OracleNet:
    def __init__(self, state_dict, bev_hw=(512, 512), seq_num=3):
        self.w = {k: v.detach().to(torch.float32).cpu() if v.is_floating_point() else v.cpu()
                  for k, v in state_dict.items()}
        self.bev_hw = tuple(bev_hw)
        self.seq_num = seq_num

    # ---- leaf helpers -------------------------------------------------------------------
    def bn(self, x, p):
        w = self.w
        return F.batch_norm(x, w[p + ".running_mean"], w[p + ".running_var"], w[p + ".weight"],
                            w[p + ".bias"], False, 0.0, BN_EPS)

    def conv(self, x, p, stride=1, padding=0):
        return F.conv2d(x, self.w[p + ".weight"], self.w.get(p + ".bias"), stride, padding)

    def linear(self, x, p):
        return F.linear(x, self.w[p + ".weight"], self.w[p + ".bias"])

    # ---- blocks -------------------------------------------------------------------------
    def downsample(self, x, p, stride):
        """networks/backbone.py:14-34."""
        a = self.bn(self.conv(x, p + ".conv_branch.0", stride, 1), p + ".conv_branch.1")
        b = self.bn(self.conv(x, p + ".pool_branch.0"), p + ".pool_branch.1")
        b = F.max_pool2d(b, 3, stride, 1)
        return F.relu(a + b)

    def basic_block(self, x, p, use_att):
        """networks/backbone.py:136-159 (+ ChannelAtt :87-102)."""
        y = F.relu(self.bn(self.conv(x, p + ".layer.0", 1, 1), p + ".layer.1"))
        y = self.bn(self.conv(y, p + ".layer.3", 1, 1), p + ".layer.4")
        if use_att:
            g = F.adaptive_avg_pool2d(y, 1)
            g = F.relu(self.conv(g, p + ".channel_att.cnet.1"))
            g = torch.sigmoid(self.conv(g, p + ".channel_att.cnet.3"))
            y = y * g
        return F.relu(y + x)

    def unbalance_block(self, x, p, k):
        """networks/multi_view_encoder.py:478-497; k = (7,3) or (5,3)."""
        a = F.relu(self.bn(self.conv(x, p + ".layer7x3.0", 1, (k[0] // 2, k[1] // 2)), p + ".layer7x3.1"))
        b = F.relu(self.bn(self.conv(x, p + ".layer3x7.0", 1, (k[1] // 2, k[0] // 2)), p + ".layer3x7.1"))
        y = self.bn(self.conv(torch.cat((a, b), 1), p + ".layer3x3.0", 1, 1), p + ".layer3x3.1")
        return F.relu(y + x)

    def stage(self, x, p, n_plain, stride, unbalance=None):
        """_make_layer, networks/multi_view_encoder.py:380-388: DownSample2D, n_plain BasicBlocks
        (index 1 optionally swapped for the Unbalance block, :350,:355), one BasicBlock with attention."""
        x = self.downsample(x, p + ".0", stride)
        for i in range(1, n_plain + 1):
            if unbalance is not None and i == 1:
                x = self.unbalance_block(x, p + ".1", unbalance)
            else:
                x = self.basic_block(x, "%s.%d" % (p, i), False)
        return self.basic_block(x, "%s.%d" % (p, n_plain + 1), True)

    # ---- native-op restatements ---------------------------------------------------------
    @staticmethod
    def scatter_max(feat, ind, out_hw, scale):
        """deep_point.VoxelMaxPool (deep_point/src/point_deep.cpp:19-88): feat (B,C,N,1), ind (B,N,2,1)."""
        b, c, n, _ = feat.shape
        h, w = out_hw
        cy = (ind[:, :, 0, 0].float() * torch.tensor(scale[0], dtype=torch.float32)).double().trunc()
        cx = (ind[:, :, 1, 0].float() * torch.tensor(scale[1], dtype=torch.float32)).double().trunc()
        ok = (cy >= 0) & (cy < h) & (cx >= 0) & (cx < w)
        flat = torch.where(ok, cy * w + cx, torch.full_like(cy, h * w)).long()      # dump slot at h*w
        out = torch.zeros(b, c, h * w + 1, dtype=feat.dtype)
        out.scatter_reduce_(2, flat[:, None, :].expand(b, c, n), feat[..., 0], "amax", include_self=False)
        return out[:, :, :h * w].reshape(b, c, h, w)

    @staticmethod
    def gather_bilinear(grid, coord, scale):
        """networks/backbone.py:453-475."""
        h, w = grid.shape[2], grid.shape[3]
        gx = (2 * coord[:, :, 1] * scale[1] / (w - 1)) - 1
        gy = (2 * coord[:, :, 0] * scale[0] / (h - 1)) - 1
        g = torch.stack((gx, gy), dim=-1)
        return F.grid_sample(grid, g, mode="bilinear", padding_mode="zeros", align_corners=True)

    @staticmethod
    def msda_core(value, hw, loc, attn):
        """Sampler of deformattn/src/cuda/ms_deform_im2col_cuda.cuh:237-299 for one level, written as
        explicit four-tap gathers (not grid_sample): value (N,S,M,D), loc (N,Lq,M,1,P,2), attn (N,Lq,M,1,P)."""
        n, s, m, d = value.shape
        hh, ww = hw
        lq, p = loc.shape[1], loc.shape[4]
        w_im = loc[:, :, :, 0, :, 0] * ww - 0.5          # (N,Lq,M,P)
        h_im = loc[:, :, :, 0, :, 1] * hh - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < hh) & (w_im < ww)
        h0, w0 = torch.floor(h_im), torch.floor(w_im)
        lh, lw = h_im - h0, w_im - w0
        out = torch.zeros(n, lq, m, d, dtype=value.dtype)
        val = value.permute(0, 2, 1, 3)                   # (N,M,S,D)
        for dy, dx, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                           (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            yy, xx = h0 + dy, w0 + dx
            ok = inside & (yy >= 0) & (yy <= hh - 1) & (xx >= 0) & (xx <= ww - 1)
            lin = torch.where(ok, yy * ww + xx, torch.zeros_like(yy)).long()       # (N,Lq,M,P)
            lin = lin.permute(0, 2, 1, 3).reshape(n, m, lq * p)
            tap = torch.gather(val, 2, lin[..., None].expand(n, m, lq * p, d))
            tap = tap.reshape(n, m, lq, p, d).permute(0, 2, 1, 3, 4)                # (N,Lq,M,P,D)
            coef = torch.where(ok, wt * attn[:, :, :, 0, :], torch.zeros_like(wt))
            out = out + (tap * coef[..., None]).sum(3)
        return out.reshape(n, lq, m * d)

    def deform_layer(self, query, src, ref, hw, p):
        """networks/multi_view_encoder.py:313-321 with deformattn/modules/ms_deform_attn.py:78-116."""
        n, lq, c = query.shape
        heads, pts = 4, 4
        a = p + ".cross_attn"
        value = self.linear(src, a + ".value_proj").view(n, -1, heads, c // heads)
        off = self.linear(query, a + ".sampling_offsets").view(n, lq, heads, 1, pts, 2)
        attn = F.softmax(self.linear(query, a + ".attention_weights").view(n, lq, heads, pts), -1)
        attn = attn.view(n, lq, heads, 1, pts)
        norm = torch.tensor([hw[1], hw[0]], dtype=torch.float32)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, None, None, :]
        sampled = self.msda_core(value, hw, loc, attn)
        query = query + self.linear(sampled, a + ".output_proj")
        query = F.layer_norm(query, (c,), self.w[p + ".norm1.weight"], self.w[p + ".norm1.bias"], LN_EPS)
        ffn = self.linear(F.relu(self.linear(query, p + ".linear1")), p + ".linear2")
        return F.layer_norm(query + ffn, (c,), self.w[p + ".norm2.weight"], self.w[p + ".norm2.bias"], LN_EPS)

    # ---- the network --------------------------------------------------------------------
    def bev_net(self, x, bev_xy, sphere, memory):
        """CENet_Transformer.forward, networks/multi_view_encoder.py:390-458."""
        p = "bev_net"
        x0 = self.stage(x, p + ".header_bev", 2, 2, (7, 3))
        x0_pt = self.gather_bilinear(x0, bev_xy, (0.5, 0.5))
        x0_rv = self.scatter_max(x0_pt, sphere, (32, 1024), (0.5, 0.5))
        x0_rv = self.stage(x0_rv, p + ".header_rv", 1, 1)
        x0_pt = self.gather_bilinear(x0_rv, sphere, (0.5, 0.5))
        x0 = torch.cat((x0, self.scatter_max(x0_pt, bev_xy, (256, 256), (0.5, 0.5))), 1)

        x1 = self.stage(x0, p + ".res1_bev", 3, 2, (5, 3))
        x1_pt = self.gather_bilinear(x1, bev_xy, (0.25, 0.25))
        x1_rv = self.scatter_max(x1_pt, sphere, (16, 512), (0.25, 0.25))
        x1_rv = self.stage(x1_rv, p + ".res1_rv", 2, 1)
        x1_pt = self.gather_bilinear(x1_rv, sphere, (0.25, 0.25))
        x1 = torch.cat((x1, self.scatter_max(x1_pt, bev_xy, (128, 128), (0.25, 0.25))), 1)

        x2 = self.stage(x1, p + ".res2", 4, 2)

        b, c, hh, ww = x2.shape
        src = x2.flatten(2).transpose(1, 2)
        if memory is None:
            query = self.w[p + ".query_embed.weight"].unsqueeze(0).repeat(b, 1, 1)
        else:
            query = memory.flatten(2).transpose(1, 2)
        ys = (torch.arange(hh, dtype=torch.float32) + 0.5) / hh
        xs = (torch.arange(ww, dtype=torch.float32) + 0.5) / ww
        ref = torch.stack((xs[None, :].expand(hh, ww), ys[:, None].expand(hh, ww)), -1).reshape(1, hh * ww, 1, 2)
        ref = ref.expand(b, hh * ww, 1, 2)
        for i in range(2):
            query = self.deform_layer(query, src, ref, (hh, ww), "%s.deformattn_module.deformattn_layers.%d" % (p, i))
        x2 = query.transpose(1, 2).reshape(b, c, hh, ww)

        size = x0.shape[2:]
        r0 = F.interpolate(x0, size=size, mode="bilinear", align_corners=True)
        r1 = F.interpolate(x1, size=size, mode="bilinear", align_corners=True)
        r2 = F.interpolate(x2, size=size, mode="bilinear", align_corners=True)
        out = torch.cat((r0, r1, r2), 1)
        out = F.leaky_relu(self.bn(self.conv(out, p + ".conv_1.conv", 1, 1), p + ".conv_1.bn"), 0.01)
        out = F.leaky_relu(self.bn(self.conv(out, p + ".conv_2.conv", 1, 1), p + ".conv_2.bn"), 0.01)
        aux = (self.conv(r0, p + ".aux_head1"), self.conv(r1, p + ".aux_head2"), self.conv(r2, p + ".aux_head3"))
        return out, x1_pt, aux, x2

    def stage_forward(self, xyzi, coord, sphere, memory=None):
        """models/StreamMOS.py:86-113.  xyzi (B,T,7,N,1), coord (B,T,N,3,1), sphere (B,T,N,2,1)."""
        with torch.no_grad():
            b, t, c, n, _ = xyzi.shape
            bev_xy = coord[:, 0, :, :2].contiguous()
            sph = sphere[:, 0].contiguous()
            x = xyzi.reshape(b * t, c, n, 1).float()
            p = "point_pre.layer"
            x = self.bn(x, p + ".0.layer.0")
            x = F.relu(self.bn(self.conv(x, p + ".0.layer.1"), p + ".0.layer.2"))
            pt = F.relu(self.bn(self.conv(x, p + ".1.layer.0"), p + ".1.layer.1"))
            grid = self.scatter_max(pt, coord.reshape(b * t, n, 3, 1)[:, :, :2].contiguous(), self.bev_hw, (1.0, 1.0))
            grid = grid.view(b, -1, self.bev_hw[0], self.bev_hw[1])
            bev, pt1, aux, mem = self.bev_net(grid, bev_xy, sph, memory)
            pt_bev = self.gather_bilinear(bev, bev_xy, (0.5, 0.5))
            pt_cur = pt.view(b, t, -1, n, 1)[:, 0]
            m = "point_post.merge_layer"
            y = torch.cat((pt_cur, pt_bev, pt1), 1)
            y = F.relu(self.bn(self.conv(y, m + ".0"), m + ".1"))
            y = F.relu(self.bn(self.conv(y, m + ".3"), m + ".4"))
            pred = self.conv(y, "pred_layer.pred_layer.0").float()
            return pred, aux[0], aux[1], aux[2], mem


def tta_labels(pred_cls):
    """val_StreamMOS.py:97-98,113: softmax over classes, mean over the TTA batch, argmax -> (N,) int64
    plus the averaged probabilities (N,3)."""
    prob = F.softmax(pred_cls, dim=1).mean(dim=0).permute(2, 1, 0).squeeze(0).contiguous()
    return prob.argmax(dim=1), prob
