"""Fused inference engine for ``AttNet`` (eval mode, fp32, one GPU).

``AttNet.infer`` / ``stage_forward`` keep the reference's module graph for training and for CPU
tensors; on a GPU in eval mode they hand over to this engine, which computes the same function
(models/StreamMOS.py:86-113, networks/multi_view_encoder.py:390-458) with far fewer passes over HBM:

* BatchNorm is folded into the preceding conv's weights once (float64 on the host); what is left of
  every conv -> BN -> ReLU (-> add -> ReLU) chain is one fused epilogue kernel (csrc/epilogue.hip);
* concatenations never run: producers write straight into channel slices of the destination buffer
  (conv epilogues, scatters and gathers all take output strides);
* the three bilinear resizes + ``torch.cat`` in front of ``conv_1`` are one kernel;
* the convs themselves stay on PyTorch-ROCm / MIOpen (NCHW Winograd kernels).

The engine holds *copies* of the folded weights: it is rebuilt whenever the module's parameters may have
changed (``load_state_dict``, ``.to()``, ``.train()``), see ``AttNet._engine_for``.
"""
import os

import torch
import torch.nn.functional as F

from . import ops, profiling

RELU, LEAKY, NONE = ops.ACT_RELU, ops.ACT_LEAKY, ops.ACT_NONE


def _fold(conv_w, conv_b, bn, pre=None):
    """(w', b') with BatchNorm `bn` (eval statistics) applied AFTER the conv and, optionally, a BatchNorm
    `pre` applied BEFORE it (PointNet's input BN, networks/backbone.py:205-210)."""
    w = conv_w.detach().double()
    b = conv_b.detach().double() if conv_b is not None else torch.zeros(w.shape[0], dtype=torch.float64, device=w.device)
    if pre is not None:
        s_in = pre.weight.detach().double() / torch.sqrt(pre.running_var.detach().double() + pre.eps)
        o_in = pre.bias.detach().double() - pre.running_mean.detach().double() * s_in
        b = b + (w.flatten(2).sum(2) * o_in[None, :]).sum(1)
        w = w * s_in[None, :, None, None]
    if bn is not None:
        s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        w = w * s[:, None, None, None]
        b = (b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
    return w.float().contiguous(), b.float().contiguous()


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _cl_w(w):
    return w.contiguous(memory_format=torch.channels_last)


class InferenceEngine:
    def __init__(self, net, layout="cl"):
        """layout "cl": every feature map channels-last (default; fastest under MIOpen's solver search and the natural
        layout of the scatters and of the attention tokens); "nchw": the first version of the engine, kept for A/B."""
        if layout not in ("cl", "nchw"):
            raise ValueError("layout must be 'cl' or 'nchw'")
        self.layout = layout
        self.device = next(net.parameters()).device
        self.bev_hw = tuple(net.bev_wl_shape)
        enc = net.bev_net

        l0, l1 = net.point_pre.layer[0].layer, net.point_pre.layer[1].layer
        self.pp1 = _fold(l0[1].weight, None, l0[2], pre=l0[0])
        self.pp2 = _fold(l1[0].weight, None, l1[1])

        self.header_bev = [self._block(m) for m in enc.header_bev]
        # sparse first stage (csrc/stem.hip): per parity class of the input cell, the kernel taps that can reach an
        # output pixel under stride 2, stacked with the 1x1 pool-branch weights, in MFMA operand order
        self.sparse_stem = os.environ.get("SMOS_SPARSE_STEM", "1") != "0"     # A/B switch (AttNet.engine_sparse_stem overrides)
        self.stem_w = None
        p0 = self.header_bev[0]
        if (p0.kind == "down" and p0.stride == 2 and p0.wa.shape[0] == 32 and p0.wa.shape[1] == 192 and
                tuple(p0.wa.shape[2:]) == (3, 3)):
            self.stem_w = ops.stem_prepare_weights(p0.wa, p0.wp)
        self.header_rv = [self._block(m) for m in enc.header_rv]
        self.res1_bev = [self._block(m) for m in enc.res1_bev]
        self.res1_rv = [self._block(m) for m in enc.res1_rv]
        self.res2 = [self._block(m) for m in enc.res2]

        self.query_embed = enc.query_embed.weight.detach()
        self.layers = []
        for lyr in enc.deformattn_module.deformattn_layers:
            ca = lyr.cross_attn
            self.layers.append(_Obj(
                heads=ca.n_heads, points=ca.n_points,
                value=(ca.value_proj.weight.detach(), ca.value_proj.bias.detach()),
                # one GEMM for offsets and attention logits
                qproj=(torch.cat((ca.sampling_offsets.weight.detach(), ca.attention_weights.weight.detach()), 0).contiguous(),
                       torch.cat((ca.sampling_offsets.bias.detach(), ca.attention_weights.bias.detach()), 0).contiguous()),
                out=(ca.output_proj.weight.detach(), ca.output_proj.bias.detach()),
                norm1=(lyr.norm1.weight.detach(), lyr.norm1.bias.detach(), lyr.norm1.eps),
                lin1=(lyr.linear1.weight.detach(), lyr.linear1.bias.detach()),
                lin2=(lyr.linear2.weight.detach(), lyr.linear2.bias.detach()),
                norm2=(lyr.norm2.weight.detach(), lyr.norm2.bias.detach(), lyr.norm2.eps)))

        # temporal fusion on the own MFMA kernels (csrc/tfusion.hip): value_proj of every layer + the first layer's offset /
        # logit projection as ONE launch, then per layer the sampler and ONE kernel for output_proj -> +query -> LayerNorm ->
        # FFN -> + -> LayerNorm (-> the next layer's projection).  SMOS_TFUSION=0: the library-GEMM form (A/B switch).
        self.tfusion = os.environ.get("SMOS_TFUSION", "1") != "0"
        self._tf_ok = False
        try:
            for i, L in enumerate(self.layers):
                nxt = self.layers[i + 1].qproj if i + 1 < len(self.layers) else None
                L.tf = ops.TfusionLayer(L.out, L.norm1, L.lin1, L.lin2, L.norm2, next_qproj=nxt)
                L.wv_stream = ops.tfusion_pack_linear(L.value[0])
                L.wq_stream = ops.tfusion_pack_linear(L.qproj[0]) if i == 0 else None
            self._tf_ok = all(L.value[0].shape[0] // L.heads == 32 and L.points <= 8 and L.qproj[0].shape[0] % 4 == 0 for L in self.layers)
        except RuntimeError:
            self._tf_ok = False

        self.conv_1 = _fold(enc.conv_1.conv.weight, None, enc.conv_1.bn)
        self.conv_2 = _fold(enc.conv_2.conv.weight, None, enc.conv_2.bn)
        # the three 1x1 aux heads as ONE block-diagonal 1x1 conv over the 320-channel decoder input
        heads = (enc.aux_head1, enc.aux_head2, enc.aux_head3)
        ncls = heads[0].weight.shape[0]
        cin = [h.weight.shape[1] for h in heads]
        w = torch.zeros((3 * ncls, sum(cin), 1, 1), device=self.device)
        off = 0
        for i, h in enumerate(heads):
            w[i * ncls:(i + 1) * ncls, off:off + cin[i]] = h.weight.detach()
            off += cin[i]
        self.aux = (w.contiguous(), torch.cat([h.bias.detach() for h in heads]).contiguous(), ncls)
        self.grid2point_scale = tuple(net.bev_grid2point.scale_rate)
        # conv_1 without the upsampled concatenation (csrc/upconv.hip): direct conv on the fine map's channels, tap GEMMs
        # at source resolution for the two coarser maps; the aux heads likewise run before the (linear) upsampling
        self.upconv = os.environ.get("SMOS_UPCONV", "1") != "0"
        w1 = self.conv_1[0]
        self.conv_1a = w1[:, :cin[0]].contiguous()
        self.conv_1z = (ops.upconv_tap_weights(w1, cin[0], cin[0] + cin[1]), ops.upconv_tap_weights(w1, cin[0] + cin[1], sum(cin)))
        self.aux_split = [(h.weight.detach().contiguous(), h.bias.detach().contiguous()) for h in heads]

        m = net.point_post.merge_layer
        self.post1 = _fold(m[0].weight, None, m[1])
        self.post2 = _fold(m[3].weight, None, m[4])
        p = net.pred_layer.pred_layer[0]
        self.pred = (p.weight.detach().contiguous(), p.bias.detach().contiguous())
        self.refine = None                  # stage-2 model (models/StreamMOS_seg.py:21-30): a second point head
        if hasattr(net, "refine"):
            rm = net.refine.bf_point_post.merge_layer
            rp = net.refine.bf_pred_layer.pred_layer[0]
            self.refine = (_fold(rm[0].weight, None, rm[1]), _fold(rm[3].weight, None, rm[4]),
                           (rp.weight.detach().contiguous(), rp.bias.detach().contiguous()))
        # fused point head (csrc/point_head.hip) when the head has the reference's 192 -> 96 -> 64 -> 3 shape
        self.fused_head = os.environ.get("SMOS_FUSED_HEAD", "1") != "0"
        self.head_w = self.refine_w = None
        try:
            self.head_w = ops.point_head_prepare(self.post1, self.post2, self.pred)
            if self.refine is not None:
                self.refine_w = ops.point_head_prepare(*self.refine)
        except RuntimeError:
            self.head_w = self.refine_w = None
        if layout == "cl":
            for blocks in (self.header_bev, self.header_rv, self.res1_bev, self.res1_rv, self.res2):
                for p in blocks:
                    for k in ("wa", "wp", "wb", "wc", "w1", "w2"):
                        if hasattr(p, k):
                            setattr(p, k, _cl_w(getattr(p, k)))
            self.conv_1a = _cl_w(self.conv_1a)
            self.aux_split = [(_cl_w(w), b) for w, b in self.aux_split]
            self.conv_1 = (_cl_w(self.conv_1[0]), self.conv_1[1])
            self.conv_2 = (_cl_w(self.conv_2[0]), self.conv_2[1])
            self.aux = (_cl_w(self.aux[0]), self.aux[1], self.aux[2])
        # every 2-D convolution of the channels-last engine runs on the library's own implicit-GEMM kernel with the
        # epilogue fused (csrc/conv_igemm.hip).  SMOS_OWN_CONV=0 falls back to MIOpen convs + separate epilogue passes
        # (kept for A/B runs; tools/ubench_conv.py compares the two per layer).
        self.own_conv = os.environ.get("SMOS_OWN_CONV", "1") != "0"
        self.conv_rows_mt = int(os.environ.get("SMOS_CONV_ROWS_MT", "1"))      # largest mt the row-staging conv is used for
        self.conv_rows = int(os.environ.get("SMOS_CONV_ROWS", "3"))      # n > 0: row-staging conv for KW >= n at mt = 1; 0: off
        self.fused_gate_sums = os.environ.get("SMOS_GATE_SUMS", "1") != "0"      # ChannelAtt pool sums from the conv epilogue
        self.wino = os.environ.get("SMOS_WINO", "1") != "0"      # Winograd F(2x2,3x3) for the stride-1 3x3 layers (A/B switch)
        self.wino1d = os.environ.get("SMOS_WINO1D", "1") != "0"  # 1-D Winograd F(2,3) for the k x 3 / 3 x k layers (A/B switch)
        self.pool_fused = os.environ.get("SMOS_POOL_FUSED", "1") != "0"   # DownSample2D pool branch + tail in one launch (A/B switch)
        self.block_call = os.environ.get("SMOS_BLOCK_CALL", "1") != "0"   # BasicBlock = one foreign call (A/B switch; same launches)
        self.wino_chain = os.environ.get("SMOS_WINO_CHAIN", "0") == "1"   # EXPERIMENTAL: runs of BasicBlocks as one dataflow launch
        self._wprep = {}
        self._wino_plan = {}
        self._shapes = None
        self._lsi = None
        self._hw = None
        self.miopen_search = True

    # ---- parameter extraction -----------------------------------------------------------
    def _block(self, m):
        from .refapi.networks import backbone as bb
        from .refapi.networks import multi_view_encoder as mve
        if isinstance(m, bb.DownSample2D):
            wa, ba = _fold(m.conv_branch[0].weight, None, m.conv_branch[1])
            wp, bp = _fold(m.pool_branch[0].weight, None, m.pool_branch[1])
            # maxpool(y + b) == maxpool(y) + b for a per-channel constant b: both biases move behind the pool
            return _Obj(kind="down", wa=wa, wp=wp, bias=(ba + bp).contiguous(), stride=m.conv_branch[0].stride[0])
        if isinstance(m, mve.Unbalance_BasicBlock):
            wa, ba = _fold(m.layer7x3[0].weight, None, m.layer7x3[1])
            wb, bb_ = _fold(m.layer3x7[0].weight, None, m.layer3x7[1])
            wc, bc = _fold(m.layer3x3[0].weight, None, m.layer3x3[1])
            return _Obj(kind="unbalance", wa=wa, ba=ba, pa=m.layer7x3[0].padding, wb=wb, bb=bb_, pb=m.layer3x7[0].padding,
                        wc=wc, bc=bc)
        if isinstance(m, bb.BasicBlock):
            w1, b1 = _fold(m.layer[0].weight, None, m.layer[1])
            w2, b2 = _fold(m.layer[3].weight, None, m.layer[4])
            o = _Obj(kind="basic", w1=w1, b1=b1, w2=w2, b2=b2, att=m.use_att)
            if m.use_att:
                c1, c2 = m.channel_att.cnet[1], m.channel_att.cnet[3]
                o.cw1 = c1.weight.detach().reshape(c1.weight.shape[0], -1).contiguous()
                o.cb1 = c1.bias.detach().contiguous()
                o.cw2 = c2.weight.detach().reshape(c2.weight.shape[0], -1).contiguous()
                o.cb2 = c2.bias.detach().contiguous()
            return o
        raise RuntimeError("InferenceEngine: unexpected module %s" % type(m).__name__)

    # ---- blocks ---------------------------------------------------------------------------
    def _run_block(self, x, p, out=None):
        if p.kind == "down":
            if x.stride(1) == 1 and x.shape[1] > 1:       # channels-last input (the scatter target of stage 0)
                a = F.conv2d(x, p.wa.contiguous(memory_format=torch.channels_last), None, p.stride, 1)
                q = F.conv2d(x, p.wp.contiguous(memory_format=torch.channels_last))
                dst = out if out is not None else torch.empty(a.shape, dtype=a.dtype, device=a.device)
                return ops.downsample_epilogue(a, q, p.bias, p.stride, out=dst)
            a = F.conv2d(x, p.wa, None, p.stride, 1)
            q = F.conv2d(x, p.wp)
            return ops.downsample_epilogue(a, q, p.bias, p.stride, out=out if out is not None else a)
        if p.kind == "unbalance":
            b, c, h, w = x.shape
            both = torch.empty((b, 2 * c, h, w), dtype=x.dtype, device=x.device)
            ops.bias_act(F.conv2d(x, p.wa, None, 1, p.pa), p.ba, RELU, out=both[:, :c])
            ops.bias_act(F.conv2d(x, p.wb, None, 1, p.pb), p.bb, RELU, out=both[:, c:])
            y = F.conv2d(both, p.wc, None, 1, 1)
            return ops.bias_act(y, p.bc, RELU, out=out if out is not None else y, residual=x)
        y = F.conv2d(x, p.w1, None, 1, 1)
        ops.bias_act(y, p.b1, RELU, out=y)
        y2 = F.conv2d(y, p.w2, None, 1, 1)
        dst = out if out is not None else y2
        if p.att:
            return ops.channel_gate_residual(y2, p.b2, p.cw1, p.cb1, p.cw2, p.cb2, x, self._block_ws(p, y2.shape[0] * y2.shape[1]),
                                             out=dst)
        return ops.bias_act(y2, p.b2, RELU, out=dst, residual=x)

    @staticmethod
    def _linear_relu(x, wb):
        """relu(x @ W^T + b) for a folded 1x1 conv (W [Cout, Cin, 1, 1]) on point rows; the bias + ReLU epilogue
        is fused into the hipBLASLt GEMM where the runtime offers it."""
        w = wb[0].view(wb[0].shape[0], -1)
        try:
            return torch._addmm_activation(wb[1], x, w.t(), use_gelu=False)
        except (RuntimeError, AttributeError):
            return torch.relu_(torch.addmm(wb[1], x, w.t()))

    def _point_heads(self, fuse, aux, k, x2, n_live=None):
        """CatFusion + PredBranch as point-major GEMMs: [B*N, 192] -> 96 -> 64 -> 3 (and the stage-2 refine head).
        n_live (device int32, runner only): real points at the front of every sample; the padding tail's logits are zeros."""
        bs, n = fuse.shape[0], fuse.shape[1]
        rows = fuse.view(bs * n, -1)
        a3 = tuple(aux) if isinstance(aux, (tuple, list)) else (aux[:, :k], aux[:, k:2 * k], aux[:, 2 * k:])

        def head(l1, l2, pr, fused):
            if self.fused_head and fused is not None and fuse.stride(1) % 4 == 0:
                return ops.point_head(fuse, fused[0], fused[1], n_live=n_live).unsqueeze(-1)
            z = self._linear_relu(self._linear_relu(rows, l1), l2)
            out = torch.addmm(pr[1], z, pr[0].view(pr[0].shape[0], -1).t())
            return out.view(bs, n, -1).permute(0, 2, 1).contiguous().unsqueeze(-1)

        pred = head(self.post1, self.post2, self.pred, self.head_w)
        if self.refine is not None:
            return (pred, head(*self.refine, self.refine_w)) + a3 + (x2,)
        return (pred,) + a3 + (x2,)

    def _block_ws(self, p, n_floats):
        """Scratch of one channel-attention block (plane sums written by one kernel, read by the next).  Keyed by
        (block, HIP stream, scratch namespace): blocks of different pipeline stages run concurrently on different streams, the same
        block may run on two streams at once (encode(t) on the main stream next to encode(t+1) on the side stream), and
        hipGraphs captured from one engine for several TTA groups replay concurrently with the addresses baked in
        (namespace = group index, ops.set_workspace_namespace in StreamRunner._capture) -- a shared scratch would be a data
        race in each case."""
        table = p.__dict__.get("ws")
        if table is None:
            table = p.__dict__["ws"] = ops.new_block_scratch(self.device)     # registered: release_stream_workspaces purges it
        key = (ops._raw_stream(ops._dev_index(self.device)), ops._ws_namespace)
        ws = table.get(key)
        if ws is None or ws.numel() < n_floats:
            ws = table[key] = torch.zeros(n_floats, dtype=torch.float32, device=self.device)
        return ws

    def _run_stage(self, x, blocks, out=None):
        for i, p in enumerate(blocks):
            x = self._run_block(x, p, out if i == len(blocks) - 1 else None)
        return x

    def _cross_view(self, cat_buf, c, bev_xy, sphere, rv_blocks, rv_hw, scale, point_rows=None):
        """cat_buf[:, :c] holds the BEV feature; fills cat_buf[:, c:] with the range-view branch scattered back
        (multi_view_encoder.py:395-405 / :410-420).  B2P gather + P2R scatter and R2P gather + P2B scatter are
        one kernel each (csrc/point_fused.hip); the channels-last scatter targets are transposed into the NCHW
        maps the convs want.  point_rows (optional [B,N,c] row view) receives the R2P point features."""
        bev = cat_buf[:, :c]
        b = bev.shape[0]
        dev = bev.device
        rv_cl = torch.zeros((b,) + rv_hw + (c,), dtype=torch.float32, device=dev)
        ops.gather_scatter(bev, bev_xy, scale, sphere, scale, out=rv_cl)
        rv = ops.nhwc_to_nchw(rv_cl, torch.empty((b, c) + rv_hw, dtype=torch.float32, device=dev))
        rv = self._run_stage(rv, rv_blocks)
        back = cat_buf[:, c:]
        back_cl = torch.zeros((b,) + tuple(back.shape[2:]) + (c,), dtype=torch.float32, device=dev)
        ops.gather_scatter(rv, sphere, scale, bev_xy, scale, out=back_cl, pts_out=point_rows)
        ops.nhwc_to_nchw(back_cl, back)

    @staticmethod
    def _add_norm(a, b, norm, c):
        """LayerNorm(a + b) in one pass (csrc/epilogue.hip) for the token widths that kernel is built for."""
        if c % 64 == 0 and c <= 512:
            return ops.add_layer_norm(a.contiguous(), b.contiguous(), norm[0], norm[1], norm[2] if len(norm) > 2 else 1e-5)
        return F.layer_norm(a + b, (c,), norm[0], norm[1], norm[2] if len(norm) > 2 else 1e-5)

    def _temporal_fusion(self, x2, memory, channels_last=False):
        """DeformAttnModule (multi_view_encoder.py:426-439, 245-321): the memory stream queries the current map."""
        b, c, hh, ww = x2.shape
        dev = x2.device
        if self._shapes is None or self._hw != (hh, ww):
            self._hw = (hh, ww)
            self._shapes = torch.tensor([[hh, ww]], dtype=torch.long, device=dev)
            self._lsi = torch.zeros((1,), dtype=torch.long, device=dev)
            ys = (torch.arange(hh, dtype=torch.float32, device=dev) + 0.5) / hh
            xs = (torch.arange(ww, dtype=torch.float32, device=dev) + 0.5) / ww
            self._ref = torch.stack((xs[None, :].expand(hh, ww), ys[:, None].expand(hh, ww)), -1).reshape(1, hh * ww, 1, 1, 1, 2)
            self._norm = torch.tensor([ww, hh], dtype=torch.float32, device=dev)
        # a channels-last [B,C,H,W] map is already the [B, H*W, C] token matrix: the permute + reshape below is a view
        src = x2.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        if memory is None:
            query = self.query_embed.unsqueeze(0).expand(b, -1, -1)
        else:
            query = memory.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        lq = hh * ww
        if self.tfusion and self._tf_ok and channels_last and c == 128:
            # five launches: projections | (sampler, layer) x 2
            query = query if query.is_contiguous() else query.contiguous()
            jobs = [(src, L.wv_stream, L.value[1]) for L in self.layers] + [(query, self.layers[0].wq_stream, self.layers[0].qproj[1])]
            outs = []
            for k in range(0, len(jobs), 8):
                outs += ops.tfusion_project(jobs[k:k + 8])
            qp = outs[-1]
            for i, L in enumerate(self.layers):
                sampled = ops.msda_fwd_qp(outs[i].view(b, lq, L.heads, 32), qp.view(b, lq, -1), hh, ww, L.points)
                query, qp = ops.tfusion_layer(sampled, query, L.tf)
            return query.view(b, hh, ww, c).permute(0, 3, 1, 2)
        for L in self.layers:
            h, p = L.heads, L.points
            value = F.linear(src, *L.value).view(b, lq, h, c // h)
            qp = F.linear(query, *L.qproj)
            if c // h == 32 and p <= 8:
                # softmax over the points, offset normalisation and reference points folded into the sampler
                sampled = ops.msda_fwd_qp(value.contiguous(), qp.contiguous(), hh, ww, p)
            else:
                off = qp[..., :h * p * 2].reshape(b, lq, h, 1, p, 2)
                attn = F.softmax(qp[..., h * p * 2:].reshape(b, lq, h, p), -1).view(b, lq, h, 1, p)
                loc = (self._ref + off / self._norm).contiguous()
                sampled = ops.msda_fwd(value.contiguous(), self._shapes, self._lsi, loc, attn.contiguous())
            query = self._add_norm(query, F.linear(sampled, *L.out), L.norm1, c)
            ffn = F.linear(self._linear_relu(query.view(b * lq, c), L.lin1).view(b, lq, -1), *L.lin2)
            query = self._add_norm(query, ffn, L.norm2, c)
        if channels_last:
            return query.contiguous().view(b, hh, ww, c).permute(0, 3, 1, 2)
        return query.transpose(1, 2).reshape(b, c, hh, ww).contiguous()

    # ---- the network ------------------------------------------------------------------------
    @torch.no_grad()
    def stage_forward(self, point_feat, pcds_coord, pcds_sphere_coord, memory=None):
        return self.decode(self.encode(point_feat, pcds_coord, pcds_sphere_coord), memory)

    def _conv_flags(self):
        """MIOpen solver search (measure every applicable solver once per conv shape, then reuse the fastest) --
        what the reference's own test scripts switch on (test_StreamMOS.py:20-23).  +7 % scans/s at the val shape;
        scoped to the engine's calls instead of flipping the process-wide flag."""
        if self.own_conv and self.layout == "cl":
            return ops._NO_GUARD          # every convolution of this engine is an own kernel: nothing for MIOpen to search
        return torch.backends.cudnn.flags(enabled=True, benchmark=self.miopen_search)

    def encode(self, point_feat, pcds_coord, pcds_sphere_coord, n_live=None):
        """n_live (device int32 tensor, runner only): see decode(); the point rows of the padding tail are not produced."""
        with torch.no_grad(), self._conv_flags():
            if self.layout == "cl":
                return self._encode_cl(point_feat, pcds_coord, pcds_sphere_coord, n_live)
            return self._encode(point_feat, pcds_coord, pcds_sphere_coord)

    def decode(self, enc, memory=None, want_aux=True, n_live=None):
        """want_aux=False: the three BEV aux maps (training-time supervision heads, models/StreamMOS.py:106-111) are not
        computed and come back as None -- the streaming runner throws them away (val_StreamMOS.py:97 uses pred_cls only);
        AttNet.infer always returns them.  n_live (device int32 tensor, runner only): the number of real points at the front
        of every sample of the current scan; the point head leaves the padding tail's logits at zero (val_StreamMOS.py:113
        cuts that tail off; datasets/data_StreamMOS.py:568-571 creates it).  AttNet.infer computes all N."""
        return self.decode_heads(enc, self.decode_memory(enc, memory), want_aux, n_live)

    def decode_memory(self, enc, memory=None):
        """The only part that is serial across frames: third BEV stage + deformable-attention fusion with the previous
        frame's memory.  Returns the new memory (= the fused 1/8-resolution map)."""
        with torch.no_grad(), self._conv_flags():
            if self.layout == "cl":
                x2 = enc["x2"] if "x2" in enc else self._stage_cl(enc["x1cat"], self.res2)
                return self._temporal_fusion(x2, memory, channels_last=True)
            return self._temporal_fusion(enc["x2"], memory)

    def decode_heads(self, enc, x2, want_aux=True, n_live=None):
        """Decoder convs, aux heads, bev->point gather and the point heads; nothing here feeds the next frame."""
        with torch.no_grad(), self._conv_flags():
            if self.layout == "cl":
                return self._decode_cl(enc, x2, want_aux, n_live)
            return self._decode(enc, x2)

    # ---- channels-last path -----------------------------------------------------------------------
    def _conv(self, x, w, bias, act, stride=1, residual=None, out=None, chan_sums=None):
        """act(conv(x, w) + bias [+ residual]) for a folded weight w [Cout, Cin, KH, KW] ("same" padding for odd kernels)
        on channels-last maps: one launch of csrc/conv_igemm.hip; the operand-ordered copy of w is made once per (weight,
        mt).  With SMOS_OWN_CONV=0: MIOpen conv + the separate bias / activation / residual pass."""
        cout, cin, kh, kw = w.shape
        # fast accept (no per-launch predicate walk): channel counts the MFMA tiling covers and operands far below the 2 GiB of the
        # 32-bit buffer offsets (<= 2^26 input floats; a slice's pitch is at most twice its channels, Cout at most four times Cin)
        easy = (cin % 32 == 0 and cout % 32 == 0 and cout <= 4 * cin and kh <= 7 and kw <= 7 and (stride == 1 or stride == 2) and
                x.numel() <= (1 << 26))
        if not self.own_conv or not (easy or ops.conv_cl_supported(x, cout, (kh, kw), stride, residual, out)):
            # MIOpen + the separate epilogue pass: channel counts the MFMA tiling does not cover, operands of 2 GiB and more
            if chan_sums is not None:
                raise RuntimeError("InferenceEngine._conv: channel sums need the own conv kernel")
            y = F.conv2d(x, w, None, stride, (kh // 2, kw // 2))
            if bias is None and act == NONE and residual is None and out is None:
                return y
            return ops.bias_act_cl(y, bias, act, out=out if out is not None else y, residual=residual)
        b, _, h, wd = x.shape
        ho, wo = (h + 2 * (kh // 2) - kh) // stride + 1, (wd + 2 * (kw // 2) - kw) // stride + 1
        if self.wino and kh == 3 and kw == 3 and stride == 1 and cin % 16 == 0 and cout % 16 == 0:      # = ops.conv_wino_ok
            # stride-1 3x3: Winograd F(2x2, 3x3) on the matrix cores (csrc/conv_wino.hip): 4 instead of 9 multiply-adds per
            # output and channel pair, fp32 throughout (weights transformed in float64 on the host); 1.35-1.7x the direct
            # kernels per layer (tools/ubench_wino.py), logits within 2e-6 of the reference's (tests/test_gpu_e2e.py)
            plan = self._wino_plan.get(id(w))          # the folded weight tensors live as long as the engine: id() is stable
            if plan is None:
                mb = ops.conv_wino_mb(cout)
                plan = self._wino_plan[id(w)] = (ops.conv_wino_prepare(w, mb), mb, w)
            return ops.conv_wino_cl(x, plan[0], bias, act, cout, mb=plan[1], residual=residual, out=out, chan_sums=chan_sums)
        if self.wino1d and ops.conv_wino1d_ok((kh, kw), stride, cin, cout, residual, chan_sums):
            # the k x 3 / 3 x k branches of the Unbalance blocks: 1-D Winograd F(2, 3) along the 3-tap axis (csrc/conv_wino1d.hip)
            mb = ops.conv_wino_mb(cout)
            key = (w.data_ptr(), "wino1d", mb)
            wp = self._wprep.get(key)
            if wp is None:
                wp = self._wprep[key] = ops.conv_wino1d_prepare(w, mb)
            return ops.conv_wino1d_cl(x, wp, bias, act, cout, (kh, kw), mb=mb, out=out)
        mt = ops.conv_mt(cout, b * ho * wo, residual is not None)
        if self.conv_rows and mt <= self.conv_rows_mt and ops.conv_rows_ok((kh, kw), stride, cin, cout) and kw >= self.conv_rows:
            # the row-staging variant (csrc/conv_rows.hip): per layer within 0 .. -5 % of conv_igemm alone on the GPU, +1 % in the
            # two-stream step (241.9 vs 239.1 scans/s: a fifth of the L1 requests leaves more of the memory path to the other stream)
            key = (w.data_ptr(), "rows", mt)
            wp = self._wprep.get(key)
            if wp is None:
                wp = self._wprep[key] = ops.conv_prepare(w, mt, order="rows")
            return ops.conv_rows_cl(x, wp, bias, act, cout, (kh, kw), mt=mt, residual=residual, out=out, chan_sums=chan_sums)
        key = (w.data_ptr(), mt)
        wp = self._wprep.get(key)
        if wp is None:
            wp = self._wprep[key] = ops.conv_prepare(w, mt)
        return ops.conv_cl(x, wp, bias, act, cout, (kh, kw), stride=stride, mt=mt, residual=residual, out=out, chan_sums=chan_sums)

    def _block_cl(self, x, p, out=None):
        if p.kind == "down":
            a = self._conv(x, p.wa, None, NONE, stride=p.stride)
            cout, cin = p.wp.shape[0], p.wp.shape[1]
            if (self.pool_fused and self.own_conv and cin <= 64 and ops.pool_branch_ok(cin, cout, p.stride) and
                    x.shape[0] * x.shape[2] * x.shape[3] * x.stride(3) * 4 < (1 << 31)):
                # the 1x1 pool-branch conv, the 3x3 max pool and the tail in one launch (csrc/downsample.hip): the full-resolution
                # branch map is never written.  Measured (tools/ubench_pool.py): 64 ch @256^2 /2 0.044 vs 0.047 ms for the two
                # launches, 32 ch @32x1024 /1 0.019 vs 0.023, 64 ch @16x512 /1 0.014 vs 0.020 -- and 128 ch @128^2 /2 0.043 vs
                # 0.037: the 1x1 conv is not as HBM-bound as it looks (2.1 GFLOP x 1.29 halo recompute = 19 us at the matrix
                # peak), so the 128-channel block keeps the two launches
                wq = p.__dict__.get("wq")
                if wq is None:
                    wq = p.__dict__["wq"] = ops.pool_branch_prepare(p.wp)
                return ops.downsample_pool_branch(x, wq, a, p.bias, p.stride, out=out if out is not None else a)
            q = self._conv(x, p.wp, None, NONE)
            return ops.downsample_epilogue_cl(a, q, p.bias, p.stride, out=out if out is not None else a)
        if p.kind == "unbalance":
            b, c, h, w = x.shape
            if self.block_call and self.own_conv and self.wino and self.wino1d and not profiling.enabled():
                plan = p.__dict__.get("plan")
                if plan is None:
                    ok = (c % 16 == 0 and tuple(p.wc.shape) == (c, 2 * c, 3, 3) and
                          ops.conv_wino1d_ok(tuple(p.wa.shape[2:]), 1, c, c) and ops.conv_wino1d_ok(tuple(p.wb.shape[2:]), 1, c, c))
                    plan = p.__dict__["plan"] = ops.UnbalanceBlockPlan(p.wa, p.ba, p.wb, p.bb, p.wc, p.bc) if ok else False
                if plan and 2 * x.numel() <= (1 << 26):
                    return ops.unbalance_block_cl(x, plan, out=out)
            both = ops.empty_cl(b, 2 * c, h, w, x.device)
            self._conv(x, p.wa, p.ba, RELU, out=both[:, :c])
            self._conv(x, p.wb, p.bb, RELU, out=both[:, c:])
            return self._conv(both, p.wc, p.bc, RELU, residual=x, out=out)
        if self.block_call and self.own_conv and self.wino and not profiling.enabled():
            # the block's two or three launches behind ONE foreign call (csrc/blocks.hip; same launches, same arguments).
            # Profiled runs take the launch-by-launch path below so that every launch keeps its label.
            plan = p.__dict__.get("plan")
            if plan is None:
                c = p.w1.shape[0]
                ok = (tuple(p.w1.shape) == (c, c, 3, 3) and tuple(p.w2.shape) == (c, c, 3, 3) and ops.basic_block_ok(c, p.att) and
                      (not p.att or self.fused_gate_sums))
                plan = p.__dict__["plan"] = ops.BasicBlockPlan(p.w1, p.b1, p.w2, p.b2,
                                                               (p.cw1, p.cb1, p.cw2, p.cb2) if p.att else None) if ok else False
            if plan and x.numel() <= (1 << 26):
                ws = None
                if p.att:
                    bsz, c, h, w = x.shape
                    ws = self._block_ws(p, bsz * c * (ops.conv_wino_sum_chunks(h, w) + 1))
                return ops.basic_block_cl(x, plan, out=out, ws=ws)
        y = self._conv(x, p.w1, p.b1, RELU)
        if not p.att:
            return self._conv(y, p.w2, p.b2, RELU, residual=x, out=out)
        bsz, c, h, w = y.shape
        if (self.own_conv and self.fused_gate_sums and c % 32 == 0 and c <= 256 and 1024 % c == 0 and
                ops.conv_cl_supported(y, c, tuple(p.w2.shape[2:]))):
            # the conv's epilogue leaves the per-row-segment channel sums behind: no pass over y2 for the average pool
            wino = self.wino and ops.conv_wino_ok(tuple(p.w2.shape[2:]), 1, c, c)
            chunks = ops.conv_wino_sum_chunks(h, w) if wino else ops.conv_sum_chunks(h, w)
            ws = self._block_ws(p, bsz * c * (chunks + 1))
            sums = ws[:bsz * chunks * c].view(bsz, chunks, c)
            y2 = self._conv(y, p.w2, None, NONE, chan_sums=sums)
            return ops.channel_gate_apply_cl(y2, p.b2, p.cw1, p.cb1, p.cw2, p.cb2, x, sums, ws[bsz * chunks * c:],
                                             out=out if out is not None else y2)
        y2 = self._conv(y, p.w2, None, NONE)
        need = (y2.shape[2] * y2.shape[3] // 512 + 2) * y2.shape[0] * y2.shape[1]
        return ops.channel_gate_residual_cl(y2, p.b2, p.cw1, p.cb1, p.cw2, p.cb2, x, self._block_ws(p, need),
                                            out=out if out is not None else y2)

    def _stage_cl(self, x, blocks, out=None):
        i = 0
        while i < len(blocks):
            run = self._chain_run(x, blocks, i) if self.wino_chain else 0
            if run >= 2:
                x = self._chain_cl(x, blocks[i:i + run], out if i + run == len(blocks) else None)
                i += run
                continue
            x = self._block_cl(x, blocks[i], out if i == len(blocks) - 1 else None)
            i += 1
        return x

    # ---- EXPERIMENTAL (SMOS_WINO_CHAIN=1): consecutive BasicBlocks of a stage as one launch (csrc/conv_wino_chain.hip) ----
    def _chain_run(self, x, blocks, i):
        """Number of consecutive BasicBlocks from blocks[i] on that one smos_conv_wino_chain_cl launch can take: mb = 2, at most
        one work item per CU (where a launch-by-launch layer leaves half of every CU idle anyway), at most 12 layers, a gated
        block only at the end, no profiling (labels are per launch) and no graph capture (the launch counter is host state)."""
        if not (self.own_conv and self.wino and self.fused_gate_sums) or profiling.enabled() or torch.cuda.is_current_stream_capturing():
            return 0
        b, c, h, w = x.shape
        if c % 32 or b * ((h + 7) // 8) * ((w + 31) // 32) * (c // 32) > 256 or x.numel() > (1 << 26):
            return 0
        n = 0
        while i + n < len(blocks) and n < 6:
            p = blocks[i + n]
            if p.kind != "basic" or tuple(p.w1.shape) != (c, c, 3, 3) or tuple(p.w2.shape) != (c, c, 3, 3):
                break
            if p.att and not ops.basic_block_ok(c, True):
                break
            n += 1
            if p.att:
                break
        return n

    def _chain_cl(self, x, blocks, out=None):
        b, c, h, w = x.shape
        dev = x.device
        layers, last = [], blocks[-1]
        for k, p in enumerate(blocks):
            plan = p.__dict__.get("plan")
            if not plan:
                plan = p.__dict__["plan"] = ops.BasicBlockPlan(p.w1, p.b1, p.w2, p.b2, (p.cw1, p.cb1, p.cw2, p.cb2) if p.att else None)
            u1, b1, u2, b2, _ = plan._keep
            layers.append((u1, b1, -1, ops.empty_cl(b, c, h, w, dev), RELU))
            dst = out if (out is not None and k == len(blocks) - 1) else ops.empty_cl(b, c, h, w, dev)
            if p.att:
                layers.append((u2, None, -1, dst, NONE))
            else:
                layers.append((u2, b2, 2 * k, dst, RELU))
        key = (ops._raw_stream(ops._dev_index(self.device)), ops._ws_namespace, len(layers), b, h, w)
        table = last.__dict__.setdefault("chain_ws", {})
        cw = table.get(key)
        if cw is None:
            cw = table[key] = ops.WinoChainWorkspace(len(layers), b, h, w, dev)
        sums = ws = None
        if last.att:
            chunks = ops.conv_wino_sum_chunks(h, w)
            ws = self._block_ws(last, b * c * (chunks + 1))
            sums = ws[:b * chunks * c].view(b, chunks, c)
        y = ops.conv_wino_chain_cl(x, layers, cw, chan_sums=sums)
        if last.att:
            res = layers[-3][3] if len(layers) >= 3 else x            # the gated block's input = the previous block's output
            return ops.channel_gate_apply_cl(y, last.b2, last.cw1, last.cb1, last.cw2, last.cb2, res, sums, ws[b * chunks * c:], out=y)
        return y

    def _cross_view_cl(self, cat_buf, c, bev_xy, sphere, rv_blocks, rv_hw, scale, point_rows=None, n_live=None, rv=None):
        """B2P gather + P2R scatter, range-view convs, R2P gather + P2B scatter straight into cat_buf[:, c:]
        (multi_view_encoder.py:395-405 / :410-420); everything channels-last, nothing transposed."""
        b = cat_buf.shape[0]
        back = cat_buf[:, c:]
        if rv is None:
            rv = ops.empty_cl(b, c, rv_hw[0], rv_hw[1], cat_buf.device)
            ops.zero_views_cl([rv, back])                 # the two scatter-max targets
        ops.gather_scatter_cl(cat_buf[:, :c], bev_xy, scale, sphere, scale, out=rv)
        rv = self._stage_cl(rv, rv_blocks)
        ops.gather_scatter_cl(rv, sphere, scale, bev_xy, scale, out=back, pts_out=point_rows, n_live=n_live)

    def _stem_sparse_cl(self, bev_cl, pcds_coord):
        """header_bev[0] on the occupied cells of a DENSE channels-last grid (csrc/stem.hip); the engine itself scatters
        into compact rows and never builds the dense grid (_encode_cl), this entry serves tests and A/B runs."""
        plan = ops.stem_plan(pcds_coord, bev_cl.shape[1], bev_cl.shape[2])
        return ops.sparse_downsample(bev_cl, plan, self.stem_w, self.header_bev[0].bias, compact=False)

    def _encode_cl(self, point_feat, pcds_coord, pcds_sphere_coord, n_live=None):
        bs, t, cin, n, _ = point_feat.shape
        dev = point_feat.device
        # views, not copies: the cross-view kernels take the strided slices as they lie (smos_gather_scatter_cl_view)
        bev_xy = pcds_coord[:, 0, :, :2, 0]
        sphere = pcds_sphere_coord[:, 0, :, :, 0]
        cpt = self.pp2[0].shape[0]
        hb, wb = self.bev_hw
        c_dec, c1 = self.conv_2[0].shape[0], self.res1_bev[-1].w2.shape[0]
        o1, o2 = cpt, cpt + c_dec
        fuse = torch.empty((bs, n, cpt + c_dec + c1), dtype=torch.float32, device=dev)
        c0 = self.header_bev[-1].w2.shape[0]
        x0cat = ops.empty_cl(bs, 2 * c0, hb // 2, wb // 2, dev)
        x1cat = ops.empty_cl(bs, 2 * c1, hb // 4, wb // 4, dev)
        # the four zero-initialised scatter-max targets of the frame's two cross-view transfers (two range-view maps, the upper
        # channel halves of the two concatenation buffers) in one launch instead of four torch fills
        hw0, hw1 = (32, 1024), (16, 512)               # range-view maps of the two cross-view transfers
        rv0, rv1 = ops.empty_cl(bs, c0, hw0[0], hw0[1], dev), ops.empty_cl(bs, c1, hw1[0], hw1[1], dev)
        ops.zero_views_cl([rv0, x0cat[:, c0:], rv1, x1cat[:, c1:]])
        if self.sparse_stem and self.stem_w is not None:
            # sparse first stage: the occupancy of the grid follows from the coordinates alone, so the point MLP scatters
            # into a compact row table (one row per occupied cell) and the 805 MB dense grid is never built
            plan = ops.stem_plan(pcds_coord, hb, wb, row_floats=t * cpt)
            rows = ops.pointnet_scatter_rows(point_feat.float(), pcds_coord, self.pp1[0], self.pp1[1], self.pp2[0], self.pp2[1],
                                             plan, pts_out=fuse[:, :, :o1], n_live=n_live)
            x = ops.sparse_downsample(rows, plan, self.stem_w, self.header_bev[0].bias, compact=True)
            self._stage_cl(x, self.header_bev[1:], out=x0cat[:, :c0])
        else:
            bev_cl = torch.empty((bs, hb, wb, t * cpt), dtype=torch.float32, device=dev)
            ops.pointnet_scatter(point_feat.float(), pcds_coord, self.pp1[0], self.pp1[1], self.pp2[0], self.pp2[1], bev_cl,
                                 pts_out=fuse[:, :, :o1], zero_fill=True)
            self._stage_cl(bev_cl.permute(0, 3, 1, 2), self.header_bev, out=x0cat[:, :c0])
        self._cross_view_cl(x0cat, c0, bev_xy, sphere, self.header_rv, hw0, (0.5, 0.5), rv=rv0)
        self._stage_cl(x0cat, self.res1_bev, out=x1cat[:, :c1])
        self._cross_view_cl(x1cat, c1, bev_xy, sphere, self.res1_rv, hw1, (0.25, 0.25), point_rows=fuse[:, :, o2:], n_live=n_live,
                            rv=rv1)
        # res2 (the third BEV stage) is independent of the past too, but it runs in decode(): that keeps the two pipeline
        # stages balanced so both HIP streams stay busy (re-measured after the sparse first stage shortened the encoder:
        # res2 on the encode side 158.6 vs 167.5 scans/s)
        return {"x0cat": x0cat, "x1cat": x1cat, "fuse": fuse, "bev_xy": bev_xy, "o1": o1, "o2": o2}

    def _decode_cl(self, enc, x2, want_aux=True, n_live=None):
        x0cat, x1cat, fuse, bev_xy, o1, o2 = enc["x0cat"], enc["x1cat"], enc["fuse"], enc["bev_xy"], enc["o1"], enc["o2"]
        k = self.aux[2]
        if self.upconv:
            y = ops.upconv3x3(self._conv(x0cat, self.conv_1a, None, NONE), self.conv_1[1],
                              [(x1cat, self.conv_1z[0]), (x2, self.conv_1z[1])], LEAKY)
            size = tuple(x0cat.shape[2:])

            def head1x1(src, wb):
                # 3 output channels: as a conv MIOpen falls back to its naive kernel; on channels-last rows it is a GEMM
                bb, cc, hh, ww = src.shape
                rows = src.permute(0, 2, 3, 1).reshape(bb * hh * ww, cc)
                return torch.addmm(wb[1], rows, wb[0].reshape(wb[0].shape[0], -1).t()).view(bb, hh, ww, -1).permute(0, 3, 1, 2)

            aux = (None, None, None)
            if want_aux:
                aux = [head1x1(x0cat, self.aux_split[0])]
                for src, wb in ((x1cat, self.aux_split[1]), (x2, self.aux_split[2])):
                    # a 1x1 convolution commutes with the (linear, weights summing to 1) bilinear resize
                    aux.append(F.interpolate(head1x1(src, wb), size=size, mode="bilinear", align_corners=True))
        else:
            dec_in = ops.upsample_concat_cl([x0cat, x1cat, x2], tuple(x0cat.shape[2:]))
            y = self._conv(dec_in, self.conv_1[0], self.conv_1[1], LEAKY)
            aux = F.conv2d(dec_in, self.aux[0], self.aux[1]) if want_aux else (None, None, None)
        bev_feat = self._conv(y, self.conv_2[0], self.conv_2[1], LEAKY)
        ops.gather_scatter_cl(bev_feat, bev_xy, self.grid2point_scale, pts_out=fuse[:, :, o1:o2], n_live=n_live)
        return self._point_heads(fuse, aux, k, x2, n_live)

    def _encode(self, point_feat, pcds_coord, pcds_sphere_coord):
        """Everything that does NOT depend on the previous frame: point MLP + input scatter, the three BEV stages
        and both cross-view cascades (multi_view_encoder.py:393-423).  The recurrent memory only enters in
        ``decode``, so the encoder of frame t+1 may run while frame t is still being decoded (StreamRunner pipeline)."""
        bs, t, cin, n, _ = point_feat.shape
        dev = point_feat.device
        bev_xy = pcds_coord[:, 0, :, :2, 0].contiguous()
        sphere = pcds_sphere_coord[:, 0, :, :, 0].contiguous()

        # point_pre + input scatter in one kernel; the BEV grid is channels-last, the t = 0 point features go
        # straight into the point-wise fusion buffer [pts(t=0) | bev gather | range-view gather] (point rows)
        cpt = self.pp2[0].shape[0]
        hb, wb = self.bev_hw
        c_dec, c1 = self.conv_2[0].shape[0], self.res1_bev[-1].w2.shape[0]
        o1, o2 = cpt, cpt + c_dec
        fuse = torch.empty((bs, n, cpt + c_dec + c1), dtype=torch.float32, device=dev)
        bev_cl = torch.empty((bs, hb, wb, t * cpt), dtype=torch.float32, device=dev)
        ops.pointnet_scatter(point_feat.float(), pcds_coord, self.pp1[0], self.pp1[1], self.pp2[0], self.pp2[1], bev_cl,
                             pts_out=fuse[:, :, :o1], zero_fill=True)
        bev = bev_cl.permute(0, 3, 1, 2)            # logical NCHW view of the channels-last buffer

        c0 = self.header_bev[-1].w2.shape[0]
        x0cat = torch.empty((bs, 2 * c0, hb // 2, wb // 2), dtype=torch.float32, device=dev)
        self._run_stage(bev, self.header_bev, out=x0cat[:, :c0])
        self._cross_view(x0cat, c0, bev_xy, sphere, self.header_rv, (32, 1024), (0.5, 0.5))

        x1cat = torch.empty((bs, 2 * c1, hb // 4, wb // 4), dtype=torch.float32, device=dev)
        self._run_stage(x0cat, self.res1_bev, out=x1cat[:, :c1])
        self._cross_view(x1cat, c1, bev_xy, sphere, self.res1_rv, (16, 512), (0.25, 0.25), point_rows=fuse[:, :, o2:])

        x2 = self._run_stage(x1cat, self.res2)
        return {"x0cat": x0cat, "x1cat": x1cat, "x2": x2, "fuse": fuse, "bev_xy": bev_xy, "o1": o1, "o2": o2}

    def _decode(self, enc, x2):
        """Decoder and point heads of the NCHW engine (multi_view_encoder.py:441-456, models/StreamMOS.py:105-113)."""
        x0cat, x1cat, fuse, bev_xy, o1, o2 = enc["x0cat"], enc["x1cat"], enc["fuse"], enc["bev_xy"], enc["o1"], enc["o2"]

        dec_in = ops.upsample_concat([x0cat, x1cat, x2], tuple(x0cat.shape[2:]))
        y = F.conv2d(dec_in, self.conv_1[0], None, 1, 1)
        ops.bias_act(y, self.conv_1[1], LEAKY, out=y)
        bev_feat = F.conv2d(y, self.conv_2[0], None, 1, 1)
        ops.bias_act(bev_feat, self.conv_2[1], LEAKY, out=bev_feat)
        aux = F.conv2d(dec_in, self.aux[0], self.aux[1])
        k = self.aux[2]

        ops.gather_scatter(bev_feat, bev_xy, self.grid2point_scale, pts_out=fuse[:, :, o1:o2])
        return self._point_heads(fuse, aux, k, x2)
