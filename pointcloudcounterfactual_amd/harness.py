"""Training / inference step harness with the reference's layer shapes.

NOT a port of the reference's model zoo: it exists so that BASELINE configs[3] (autoencoder training step,
Chamfer + EMD) and configs[4] (batch-sharded counterfactual-style inference) can be measured end to end with the
hot-path kernels in place.  Shapes follow the reference's source text:
  encoder   DGCNN: EdgeConv 6->64, 128->64, 128->128, 256->256 (1x1 conv2d + BN2d [+ LeakyReLU 0.2]), dynamic kNN
            (k=25) before every layer, max over k; cat -> 512; 1x1 conv1d 512->1024; global max
            (src/module/encoders.py:31-59, src/module/layers.py:159-203, configs/.../default_data.yaml:12)
  quantiser 256 codes x book 16 x dim 4, nearest code, straight-through gradient
            (src/module/quantize.py:20-32, src/module/layers.py:220-237, configs/.../vqvae.yaml)
  decoder   PCGen: sample 8 -> 64 (ReLU) -> 1024 (Hardtanh), multiplied by w; 8 component MLPs
            1024->1024->256->16 (BN, ReLU, residual) -> 3; attention over components (gumbel-softmax tau 5);
            graph_filtering k=4 (src/module/decoders.py:39-130, configs/.../pcgen.yaml)
  loss      mean-Chamfer + match_cost + 8 * MSE(w_q, w_e)   (src/train/metrics_and_losses.py:21-90,258-266)
  optimiser AdamW lr 4e-3, weight decay 1e-3                (configs/.../learn/default_learn.yaml)
  w-AE      counterfactual latent step (src/module/w_autoencoders.py:247-262, w_encoders.py:73-107, w_conditional.py:
            15-102, w_decoders.py:71-112; configs/.../w_autoencoder/model/*.yaml): transformer encoder 4 -> 512 (2 layers,
            8 heads, ff 1024, GELU, norm-first) -> z1 (16); conditional prior Linear(40 -> 256*32) and posterior
            transformer (2 layers) on the interpolated class probabilities -> z2 (16); transformer decoder (4 layers,
            ff 1024/1024/1024/512) -> w_recon; nearest code in the codebook -> indices -> embeddings -> PCGen.
The dense layers (convolutions, transformers) are plain PyTorch-ROCm modules (rocBLAS / MIOpen / SDPA); every kNN,
gather, max-pool, nearest-code search, Chamfer and EMD call goes through this package's HIP kernels.
"""

from __future__ import annotations

import itertools

import torch
import torch.nn.functional as F
from torch import nn

from pointcloudcounterfactual_amd import _lib
from pointcloudcounterfactual_amd import neighbour_ops as ops
from pointcloudcounterfactual_amd.edgeconv import FusedEdgeConv
from pointcloudcounterfactual_amd.keops_shim import SquareDistance
from pointcloudcounterfactual_amd.losses import chamfer_emd


class EdgeConv(nn.Module):
    def __init__(self, cin: int, cout: int, act: bool = True) -> None:
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = nn.LeakyReLU(0.2, inplace=True) if act else nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.act(self.bn(self.conv(x)))


class _BNReLURes(torch.autograd.Function):
    """``relu(batch_norm(z)) + res.repeat_interleave(r, 1)[:, :C]`` over ``[B,C,N]`` in two streaming passes forward and
    two backward (``csrc/bnact.hip``) instead of PyTorch's BatchNorm + threshold + add kernels; same values as the
    composition (tests/test_gpu_harness.py).  ``mean`` / ``var`` are the batch statistics (training) or the running
    ones (eval)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, res, r, mean, var, eps, training):  # noqa: ANN001
        b, c, n = z.shape
        y = torch.empty_like(z)
        st = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(_lib.lib.pcc_bn_relu_res_fwd(b, c, n, z.data_ptr(), mean.data_ptr(), var.data_ptr(), float(eps),
                                                    gamma.data_ptr(), beta.data_ptr(),
                                                    res.data_ptr() if res is not None else None,
                                                    res.shape[1] if res is not None else 0, int(r), y.data_ptr(), st),
                       'bn_relu_res_fwd')
        ctx.save_for_backward(z, gamma, beta, mean, var)
        ctx.eps, ctx.training, ctx.r = float(eps), bool(training), int(r)
        ctx.res_shape = None if res is None else tuple(res.shape)
        return y

    @staticmethod
    def backward(ctx, gy):  # noqa: ANN001
        z, gamma, beta, mean, var = ctx.saved_tensors
        b, c, n = z.shape
        gy = gy.contiguous()
        dz = torch.empty_like(z)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        st = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(_lib.lib.pcc_bn_relu_bwd(b, c, n, z.data_ptr(), mean.data_ptr(), var.data_ptr(), ctx.eps,
                                                gamma.data_ptr(), beta.data_ptr(), gy.data_ptr(), int(ctx.training),
                                                dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), st), 'bn_relu_bwd')
        dres = None
        if ctx.res_shape is not None and ctx.needs_input_grad[3]:
            r, used = ctx.r, c // ctx.r
            g = gy.view(b, used, r, n).sum(2) if r > 1 else gy
            dres = g if used == ctx.res_shape[1] else F.pad(g, (0, 0, 0, ctx.res_shape[1] - used))
        return dz, dgamma, dbeta, dres, None, None, None, None, None


class PointsConv(nn.Module):
    def __init__(self, cin: int, cout: int, act: nn.Module | None = None, bn: bool = True, residual: bool = False) -> None:
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, 1, bias=not bn)
        self.bn = nn.BatchNorm1d(cout) if bn else nn.Identity()
        self.act = act if act is not None else nn.Identity()
        self.residual = residual
        self.cin, self.cout = cin, cout

    def _fused_tail(self, z: torch.Tensor, x: torch.Tensor) -> torch.Tensor | None:
        """BatchNorm1d + ReLU (+ residual) through the fused HIP passes when the block has exactly that shape."""
        bn = self.bn
        r = self.cout // self.cin + 1
        if not (isinstance(bn, nn.BatchNorm1d) and isinstance(self.act, nn.ReLU) and z.is_cuda and z.dtype == torch.float32
                and z.is_contiguous() and z.shape[0] * z.shape[1] <= 65535 and bn.affine and bn.track_running_stats
                and (not self.residual or (self.cout % r == 0 and x.is_contiguous()))):
            return None
        b, c, n = z.shape
        if self.training:
            mean = torch.empty(c, device=z.device, dtype=torch.float32)
            var = torch.empty_like(mean)
            with torch.cuda.device(z.device):
                _lib.check(_lib.lib.pcc_bn_stats(b, c, n, z.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                                 torch.cuda.current_stream(z.device).cuda_stream), 'bn_stats')
            with torch.no_grad():  # running statistics exactly as nn.BatchNorm1d keeps them (unbiased variance)
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                cnt = b * n
                bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(var, alpha=m * cnt / max(cnt - 1, 1))
                bn.num_batches_tracked += 1
        else:
            mean, var = bn.running_mean, bn.running_var
        res = x if self.residual else None
        return _BNReLURes.apply(z, bn.weight, bn.bias, res, r if self.residual else 1, mean, var, bn.eps, self.training)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        z = self.conv(x)
        y = self._fused_tail(z, x)
        if y is not None:
            return y
        y = self.act(self.bn(z))
        if self.residual:
            # layers.py:164-166: y + x.repeat_interleave(r, 1)[:, :cout] with r = cout // cin + 1, i.e. output channel o
            # receives input channel o // r.  Same values without materialising the r-fold repeated tensor (536 MB for the
            # 1024 -> 1024 layers at B=32, N=2048): a broadcast add over a [B, cout/r, r, N] view of y.
            r = self.cout // self.cin + 1
            if self.cout % r == 0:
                b, _, n = y.shape
                y = (y.view(b, self.cout // r, r, n) + x[:, : self.cout // r].unsqueeze(2)).view(b, self.cout, n)
            else:
                y = y + x.repeat_interleave(r, 1)[:, : y.shape[1], ...]
        return y


class DGCNNEncoder(nn.Module):
    h_dim = (64, 64, 128, 256)

    def __init__(self, k: int = 25, w_dim: int = 1024, fused: bool = True) -> None:
        """``fused=True`` runs every EdgeConv block through ``FusedEdgeConv`` (same parameters, no [B,2C,N,k] tensor);
        ``fused=False`` composes it as the reference does (get_graph_features -> conv2d -> BN -> act -> max)."""
        super().__init__()
        self.k, self.fused = k, fused
        if fused:
            layers: list[nn.Module] = [FusedEdgeConv(3, self.h_dim[0], act=False)]
            layers += [FusedEdgeConv(cin, cout) for cin, cout in itertools.pairwise(self.h_dim)]
        else:
            layers = [EdgeConv(6, self.h_dim[0], act=False)]
            layers += [EdgeConv(2 * cin, cout) for cin, cout in itertools.pairwise(self.h_dim)]
        self.edge_convolutions = nn.ModuleList(layers)
        self.final_conv = PointsConv(sum(self.h_dim), w_dim, bn=False)
        # test hooks: the kNN graphs of a forward pass can be recorded, and replayed into another composition of the same
        # layers (the dynamic graphs of layers 2-4 are built on features; two compositions that agree to rounding may
        # still break a near tie differently, which the comparison must not mix up with a wrong layer)
        self.recorded_graphs: list[torch.Tensor] | None = None
        self.replay_graphs: list[torch.Tensor] | None = None

    def features(self, cloud: torch.Tensor) -> torch.Tensor:
        x = cloud.transpose(2, 1).contiguous()
        xs = []
        for layer, conv in enumerate(self.edge_convolutions):
            if self.replay_graphs is not None:
                idx = self.replay_graphs[layer]
            else:
                idx = ops.knn(x, self.k)  # dynamic graph every layer
            if self.recorded_graphs is not None:
                self.recorded_graphs.append(idx)
            if self.fused:
                x = conv(x, idx)
            else:
                x = conv(ops.get_graph_features(x, idx, self.k)[1]).max(dim=3)[0]
            xs.append(x)
        return self.final_conv(torch.cat(xs, dim=1).contiguous())

    def forward(self, cloud: torch.Tensor) -> torch.Tensor:
        return ops.global_max_pool(self.features(cloud))


class LDGCNNEncoder(nn.Module):
    """The reference's lighter encoder (``src/module/encoders.py:62-91``): ONE kNN graph on the input cloud, one EdgeConv
    layer, then ``graph_max_pooling`` over that fixed graph before every 1x1 convolution (``conv_dims`` from
    ``lgcnn.yaml``-style configurations), concatenation, final convolution, global max.  kNN, the fused EdgeConv front-end,
    the max over neighbours and the global max are this package's HIP kernels."""

    def __init__(self, k: int = 25, w_dim: int = 1024, conv_dims: tuple[int, ...] = (64, 64, 128, 256),
                 fused: bool = True) -> None:
        super().__init__()
        self.k, self.fused = k, fused
        self.edge_conv = FusedEdgeConv(3, conv_dims[0]) if fused else EdgeConv(6, conv_dims[0])
        self.points_convolutions = nn.ModuleList(
            [PointsConv(a, b, nn.LeakyReLU(0.2)) for a, b in itertools.pairwise(conv_dims)])
        self.final_conv = PointsConv(sum(conv_dims), w_dim, bn=False)

    def forward(self, cloud: torch.Tensor) -> torch.Tensor:
        x = cloud.transpose(2, 1).contiguous()
        idx = ops.knn(x, self.k)
        if self.fused:
            x = self.edge_conv(x, idx)
        else:
            x = self.edge_conv(ops.get_graph_features(x, idx, self.k)[1]).max(dim=3)[0]
        xs = [x]
        for conv in self.points_convolutions:
            x = conv(ops.graph_max_pooling(x, idx, self.k))
            xs.append(x)
        return ops.global_max_pool(self.final_conv(torch.cat(xs, dim=1).contiguous()))


class DGCNNClassifier(nn.Module):
    """classifier.py:18-66 with dgcnn.yaml: k=20, conv (64,64,128,256), feature 512, mlp (512,256), 40 classes."""

    def __init__(self, k: int = 20, n_classes: int = 40) -> None:
        super().__init__()
        self.body = DGCNNEncoder(k=k, w_dim=512)
        self.body.edge_convolutions[0] = FusedEdgeConv(3, 64)
        self.body.final_conv = PointsConv(512, 512, bn=True)
        self.mlp = nn.Sequential(nn.Linear(1024, 512, bias=False), nn.BatchNorm1d(512), nn.LeakyReLU(0.2), nn.Dropout(0.5),
                                 nn.Linear(512, 256, bias=False), nn.BatchNorm1d(256), nn.LeakyReLU(0.2), nn.Dropout(0.5),
                                 nn.Linear(256, n_classes))

    def forward(self, cloud: torch.Tensor) -> torch.Tensor:
        feat = self.body.features(cloud)
        if feat.requires_grad:
            pooled = torch.cat((ops.global_max_pool(feat), feat.mean(dim=2)), 1)
        else:
            pooled = ops.global_max_mean_pool(feat)
        return self.mlp(pooled)


class _TransferGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, from_tensor, _to_tensor):  # noqa: ANN001
        return from_tensor

    @staticmethod
    def backward(ctx, grad):  # noqa: ANN001
        return None, grad.clone()


class PCGenDecoder(nn.Module):
    def __init__(self, w_dim: int = 1024, sample_dim: int = 8, n_components: int = 8, tau: float = 5.0,
                 conv_dims: tuple[int, ...] = (1024, 256, 16), filtering: bool = True) -> None:
        super().__init__()
        self.sample_dim, self.n_components, self.tau, self.filtering = sample_dim, n_components, tau, filtering
        self.map_sample = nn.Sequential(PointsConv(sample_dim, 64, nn.ReLU(inplace=True), bn=False),
                                        PointsConv(64, w_dim, nn.Hardtanh(), bn=False))
        self.group_conv = nn.ModuleList()
        self.group_final = nn.ModuleList()
        for _ in range(n_components):
            dims = [w_dim, *conv_dims]
            self.group_conv.append(nn.Sequential(*[PointsConv(a, b, nn.ReLU(inplace=True), residual=True)
                                                   for a, b in itertools.pairwise(dims)]))
            self.group_final.append(PointsConv(conv_dims[-1], 3, bn=False))
        self.att = PointsConv(conv_dims[-1] * n_components, n_components, bn=False)

    def forward(self, w: torch.Tensor, n_points: int) -> torch.Tensor:
        x = torch.randn(w.shape[0], self.sample_dim, n_points, device=w.device)
        x = w.unsqueeze(2) * self.map_sample(x)
        feats = [g(x) for g in self.group_conv]
        xs = torch.stack([f(h) for f, h in zip(self.group_final, feats, strict=True)], dim=3)
        att = self.att(torch.cat(feats, dim=1).contiguous())
        att = F.gumbel_softmax(att, tau=self.tau, dim=1) if self.training else torch.softmax(att / self.tau, dim=1)
        x = (xs * att.transpose(2, 1).unsqueeze(1)).sum(3)
        return ops.graph_filtering(x) if self.filtering else x


class VQAutoencoder(nn.Module):
    def __init__(self, n_points: int = 2048, k: int = 25, n_codes: int = 256, book_size: int = 16, dim: int = 4,
                 fused: bool = True) -> None:
        super().__init__()
        self.n_points, self.n_codes, self.dim = n_points, n_codes, dim
        self.encoder = DGCNNEncoder(k=k, w_dim=n_codes * dim, fused=fused)
        self.decoder = PCGenDecoder(w_dim=n_codes * dim)
        self.codebook = nn.Parameter(torch.randn(n_codes, book_size, dim))

    def quantize(self, w_q: torch.Tensor) -> torch.Tensor:
        return decode_from_indices(quantize_indices(w_q.detach(), self.codebook.detach()), self.codebook)

    def forward(self, cloud: torch.Tensor) -> dict[str, torch.Tensor]:
        w_q = self.encoder(cloud)
        w_e = self.quantize(w_q)
        w = _TransferGrad.apply(w_e, w_q)
        recon = self.decoder(w, self.n_points).transpose(2, 1).contiguous()
        return {'recon': recon, 'w_q': w_q, 'w_e': w_e}


def quantize_indices(x: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    """``VectorQuantizer.quantize`` (src/module/quantize.py:20-32), index part: nearest entry of each of the ``n_codes``
    books for ``x[B, n_codes * dim]`` -> ``idx[B, n_codes]``.  On the GPU the search runs on the HIP kernel behind the
    PyKeOps call site of the reference (``pykeops_square_distance(x_flat, book_repeated).argmin(axis=2)``)."""
    n_codes, book, dim = codebook.shape
    b = x.shape[0]
    x_flat = x.reshape(b * n_codes, 1, dim).contiguous()
    book_repeated = codebook.repeat(b, 1, 1)
    if x.is_cuda:
        idx_flat = SquareDistance(x_flat, book_repeated).argmin(axis=2)
    else:
        idx_flat = ((x_flat[:, :, None, :] - book_repeated[:, None, :, :]) ** 2).sum(-1).argmin(2, keepdim=True)
    return idx_flat.view(b, n_codes)


def decode_from_indices(idx: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    """``VectorQuantizer.decode_from_indices`` (quantize.py:45-53): ``idx[B, n_codes]`` -> ``w[B, n_codes * dim]``."""
    n_codes, _book, dim = codebook.shape
    b = idx.shape[0]
    book = codebook.repeat(b, 1, 1)
    return book.gather(1, idx.reshape(b * n_codes, 1, 1).expand(-1, -1, dim)).view(b, n_codes * dim)


class CounterfactualWAutoEncoder(nn.Module):
    """The latent step of the reference's counterfactual generation, ``CounterfactualWAutoEncoder.generate_counterfactual``
    (src/module/w_autoencoders.py:247-262) with the transformer encoder / conditional encoder / decoder of the default
    configuration (wae.yaml).  Dense PyTorch; what it hands to the hot path is ``w_recon`` for the nearest-code search."""

    def __init__(self, n_codes: int = 256, dim: int = 4, n_classes: int = 40, z1_dim: int = 16, z2_dim: int = 16,
                 proj_dim: int = 512, n_heads: int = 8, cf_temperature: float = 5.0) -> None:
        super().__init__()
        self.n_codes, self.dim, self.n_classes, self.z1_dim, self.z2_dim = n_codes, dim, n_classes, z1_dim, z2_dim
        self.temperature = cf_temperature

        def enc_layer(ff: int, drop: float) -> nn.Module:
            return nn.TransformerEncoderLayer(d_model=proj_dim, nhead=n_heads, dim_feedforward=ff, dropout=drop,
                                              activation=nn.GELU(), batch_first=True, norm_first=True)

        # TransformerWEncoder (w_encoders.py:73-107)
        self.enc_proj = nn.Linear(dim, proj_dim)
        self.enc_pos = nn.Parameter(torch.randn(1, n_codes, proj_dim))
        self.enc_layers = nn.ModuleList([enc_layer(1024, 0.0), enc_layer(1024, 0.0)])
        self.enc_latent = nn.Linear(proj_dim, 2 * z1_dim)
        # ConditionalPrior + TransformerWConditionalEncoder (w_conditional.py:15-102)
        self.prior = nn.Linear(n_classes, n_codes * 2 * z2_dim)
        self.post_proj = nn.Linear(dim, proj_dim)
        self.post_pos = nn.Parameter(torch.randn(1, n_codes, proj_dim))
        self.post_prob = nn.Linear(n_classes, proj_dim)
        self.post_layers = nn.ModuleList([enc_layer(1024, 0.0), enc_layer(1024, 0.0)])
        self.post_latent = nn.Linear(proj_dim, 2 * z2_dim)
        # TransformerWDecoder (w_decoders.py:71-112)
        self.z1_proj = nn.Linear(z1_dim, proj_dim)
        self.z2_proj = nn.Linear(z2_dim, proj_dim)
        self.dec_pos = nn.Parameter(torch.randn(1, n_codes, proj_dim))
        self.dec_mem_pos = nn.Parameter(torch.randn(1, n_codes, proj_dim))
        self.dec_layers = nn.ModuleList([
            nn.TransformerDecoderLayer(d_model=proj_dim, nhead=n_heads, dim_feedforward=ff, dropout=0.1,
                                       activation=nn.GELU(), batch_first=True, norm_first=True)
            for ff in (1024, 1024, 1024, 512)])
        self.compress = nn.Linear(proj_dim, dim)

    def encode_z1(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        h = self.enc_pos + self.enc_proj(x)
        for layer in self.enc_layers:
            h = layer(h)
        mu, log_var = self.enc_latent(h).chunk(2, 2)
        return mu, log_var

    def posterior(self, probs: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        h = self.post_pos + self.post_proj(x) + self.post_prob(probs).unsqueeze(1)
        for layer in self.post_layers:
            h = layer(h)
        return self.post_latent(h)

    def decode(self, z1: torch.Tensor, z2: torch.Tensor) -> torch.Tensor:
        b = z1.shape[0]
        memory = self.z1_proj(z1) + self.dec_mem_pos
        h = self.z2_proj(z2) + self.dec_pos
        for layer in self.dec_layers:
            h = layer(h, memory)
        return self.compress(h).reshape(b, self.n_codes * self.dim)

    def interpolated_probs(self, logits: torch.Tensor, target_dim: int, target_value: float) -> torch.Tensor:
        old = torch.softmax(logits / self.temperature, dim=1)             # TemperatureScaledSoftmax, cf_temperature 5
        target = torch.zeros_like(old)
        target[:, target_dim] = 1
        return (1 - target_value) * old + target_value * target           # interpolate_probs, :281-284

    def counterfactual_w(self, w_q: torch.Tensor, logits: torch.Tensor, target_dim: int,
                         target_value: float = 1.0) -> tuple[torch.Tensor, torch.Tensor]:
        """-> (w_recon[B, n_codes * dim], probs[B, n_classes]): no sampling (z1 = mu1, z2 = p_mu2 + d_mu2, :258-260)."""
        x = w_q.view(-1, self.n_codes, self.dim)
        mu1, _ = self.encode_z1(x)
        probs = self.interpolated_probs(logits, target_dim, target_value)
        p_mu2, _ = self.prior(probs).view(-1, self.n_codes, 2 * self.z2_dim).chunk(2, 2)
        d_mu2, _ = self.posterior(probs, x).chunk(2, 2)
        return self.decode(mu1, p_mu2 + d_mu2), probs


class CounterfactualVQVAE(VQAutoencoder):
    """``CounterfactualVQVAE.generate_counterfactual`` (src/module/autoencoders.py:168-181): encoder -> w-autoencoder
    latent step on the classifier's logits -> nearest codes -> embeddings -> decoder, one forward pass."""

    def __init__(self, n_points: int = 2048, k: int = 25, n_codes: int = 256, book_size: int = 16, dim: int = 4,
                 n_classes: int = 40, fused: bool = True) -> None:
        super().__init__(n_points=n_points, k=k, n_codes=n_codes, book_size=book_size, dim=dim, fused=fused)
        self.w_autoencoder = CounterfactualWAutoEncoder(n_codes=n_codes, dim=dim, n_classes=n_classes)

    @torch.inference_mode()
    def generate_counterfactual(self, cloud: torch.Tensor, sample_logits: torch.Tensor, target_dim: int,
                                target_value: float = 1.0) -> dict[str, torch.Tensor]:
        w_q = self.encoder(cloud)
        w_recon, probs = self.w_autoencoder.counterfactual_w(w_q, sample_logits, target_dim, target_value)
        idx = quantize_indices(w_recon, self.codebook)
        w = decode_from_indices(idx, self.codebook)
        recon = self.decoder(w, self.n_points).transpose(2, 1).contiguous()
        return {'recon': recon, 'w_q': w_q, 'w_recon': w_recon, 'idx': idx, 'w': w, 'probs': probs}


def autoencoder_loss(out: dict[str, torch.Tensor], ref: torch.Tensor, c_embedding: float = 8.0) -> torch.Tensor:
    """Per-sample loss [B]: mean-Chamfer + approximate EMD + c * MSE(w_q, w_e) (chamfer_emd.yaml)."""
    embed = F.mse_loss(out['w_q'], out['w_e'], reduction='none').mean(dim=1)
    cham, emd = chamfer_emd(out['recon'], ref)  # the reference's ChamferEMD loss as one autograd node
    return cham + emd + c_embedding * embed


def make_optimizer(model: nn.Module) -> torch.optim.Optimizer:
    return torch.optim.AdamW(model.parameters(), lr=4e-3, weight_decay=1e-3)
