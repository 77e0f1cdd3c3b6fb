"""ORACLE (test infrastructure, never shipped, never on the product path).

Independent fp32 torch-CPU restatement of the reference forward path, written against
torch.nn.functional (conv1d / linear) instead of the shifted-matmul form used by
oracle/ref_numpy.py, so that the two restatements cross-check each other
(tests/test_oracle.py requires <=1e-6 relative L2 between them).  It is also the
`cpu_baseline` of bench.py (kind "port"): one compute thread, one utterance per call,
mirroring model/trainer.py:135-139 (intra=inter=1) and
egs/voxceleb/v1/nnet/lib/extract.py:89 (batch of one).

**parity unpinned** for the TDNN layers (see oracle/ref_numpy.py header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPSILON = 1e-3
LRELU_ALPHA = 0.2
VAR2STD_EPSILON = 1e-12


def _get(params, key, default=None):
    if isinstance(params, dict):
        return params.get(key, default)
    d = getattr(params, "dict", None)
    if d is None:
        d = params.__dict__
    return d.get(key, default)


class TorchTdnn(object):
    """Weights converted once (like a restored TF session), then `forward` per call."""

    def __init__(self, weights, params, dtype=torch.float32):
        self.params = params
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))).to(dtype) for k, v in weights.items()}
        ntype = _get(params, "network_type")
        if ntype == "tdnn":                                # model/tdnn.py:36-181
            self.scope, self.widths, self.seg = "tdnn", (5, 5, 7, 1, 1), (6, 7)
        elif ntype == "extended_tdnn":                     # model/tdnn.py:343-591
            self.scope, self.widths, self.seg = "etdnn", (5, 1, 5, 1, 7, 1, 9, 1, 1, 1), (12, 13)
        else:
            raise NotImplementedError("Not implement %s network" % ntype)
        # conv kernels [1,k,cin,cout] (conv2d HWIO) or [k,cin,cout] (conv1d) -> conv1d weight [cout,cin,k]
        self.conv = {}
        for i, wd in enumerate(self.widths, start=1):
            if wd > 1:
                k = self.w["%s/tdnn%d_conv/kernel" % (self.scope, i)]
                if k.dim() == 4:
                    k = k[0]
                self.conv[i] = k.permute(2, 1, 0).contiguous()

    def _bn(self, x, scope):
        w = self.w
        return w[scope + "/gamma"] * (x - w[scope + "/moving_mean"]) / torch.sqrt(
            w[scope + "/moving_variance"] + BN_EPSILON) + w[scope + "/beta"]

    def _act(self, x, scope):
        t = _get(self.params, "network_relu_type")
        if t == "prelu":                                   # model/common.py:27-42
            return F.relu(x) + self.w[scope + "/alpha"] * (x - x.abs()) * 0.5
        if t == "lrelu":
            return F.leaky_relu(x, LRELU_ALPHA)
        return F.relu(x)

    def _dense_block(self, x, name, kind, ep, scope=None):
        base = "%s/%s" % (scope or (self.scope + "/attention"), name)
        x = F.linear(x, self.w["%s/%s_dense/kernel" % (base, name)].t(), self.w["%s/%s_dense/bias" % (base, name)])
        ep["%s_dense" % name] = x
        if kind == 2:
            x = self._bn(x, "%s/%s_bn" % (base, name))
            ep["%s_bn" % name] = x
        if kind in (1, 2):
            x = self._act(x, "%s/%s_relu" % (base, name))
            ep["%s_relu" % name] = x
        if kind == 3:
            x = torch.tanh(x)
            ep["%s_tanh" % name] = x
        return x

    def _stat_pool(self, x):                               # model/pooling.py:27-52
        mean = x.mean(dim=1, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=1)
        var = torch.where(var <= VAR2STD_EPSILON, torch.full_like(var, VAR2STD_EPSILON), var)
        return torch.cat([mean[:, 0], var.sqrt()], dim=1)

    def _attention(self, ep):                              # model/pooling.py:55-240
        p = self.params
        value = ep[_get(p, "att_value_input")]
        key = ep[_get(p, "att_key_input")]
        kn = list(_get(p, "att_key_num_nodes"))
        for i in range(len(kn) - 1):
            key = self._dense_block(key, "att_key%d" % i, 2, ep)
        key = self._dense_block(key, "att_key%d" % (len(kn) - 1), int(_get(p, "att_key_network_type")), ep)
        vn = list(_get(p, "att_value_num_nodes"))
        if vn:
            for i in range(len(vn) - 1):
                value = self._dense_block(value, "att_value%d" % i, 2, ep)
            value = self._dense_block(value, "att_value%d" % (len(vn) - 1), int(_get(p, "att_value_network_type")), ep)
        h = int(_get(p, "att_num_heads"))
        q = self.w[self.scope + "/attention/query"]
        b, l, dv = value.shape
        dk = key.shape[-1]
        if _get(p, "att_split_key"):
            k4 = key.reshape(b, l, h, dk // h).permute(0, 2, 1, 3)
            qk = (k4 * q[None, :, None, :]).sum(-1)                     # [b,h,l]
            dkh = dk // h
        else:
            qk = torch.einsum('bld,hd->bhl', key, q)
            dkh = dk
        if _get(p, "att_use_scale"):
            qk = qk * (1.0 / np.sqrt(float(dkh)))
        w = torch.softmax(qk, dim=-1)
        ep["attention_weights"] = w
        if _get(p, "att_split_value"):
            v4 = value.reshape(b, l, h, dv // h).permute(0, 2, 1, 3)    # [b,h,l,d]
        else:
            v4 = value[:, None].expand(b, h, l, dv)
        mean = (v4 * w[..., None]).sum(2)
        var = (((v4 - mean[:, :, None, :]) ** 2) * w[..., None]).sum(2)
        mean = mean.reshape(b, -1)
        var = var.reshape(b, -1)
        var = torch.where(var <= VAR2STD_EPSILON, torch.full_like(var, VAR2STD_EPSILON), var)
        att = torch.cat([mean, var.sqrt()], dim=1)
        ep["att_output_before_nonlinear"] = att
        if _get(p, "att_apply_nonlinear"):
            att = self._bn(att, self.scope + "/attention/att_post_bn")
            ep["att_post_bn"] = att
            att = self._act(att, self.scope + "/attention/att_post_relu")
            ep["att_post_relu"] = att
        return att

    @torch.no_grad()
    def forward(self, features):
        """features: [b,l,d] -> endpoints (OrderedDict of torch tensors)."""
        p = self.params
        ep = OrderedDict()
        x = torch.as_tensor(np.ascontiguousarray(features)).to(self.dtype)
        sc = self.scope
        for i, wd in enumerate(self.widths, start=1):      # model/tdnn.py:40-130 / :377-540
            s = "%s/tdnn%d" % (sc, i)
            if wd > 1:
                xt = x.transpose(1, 2)                     # [b,c,l] for conv1d
                if xt.shape[2] < wd:
                    x = x.new_zeros((x.shape[0], 0, self.conv[i].shape[0]))
                else:
                    x = F.conv1d(xt, self.conv[i], self.w[s + "_conv/bias"]).transpose(1, 2)
                ep["tdnn%d_conv" % i] = x
            else:
                x = F.linear(x, self.w[s + "_dense/kernel"].t(), self.w[s + "_dense/bias"])
                ep["tdnn%d_dense" % i] = x
            x = self._bn(x, s + "_bn")
            ep["tdnn%d_bn" % i] = x
            x = self._act(x, s + "_relu")
            ep["tdnn%d_relu" % i] = x
        ptype = _get(p, "pooling_type")                    # model/pooling.py:8-25
        if ptype == "statistics_pooling":
            x = self._stat_pool(x)
        elif ptype == "self_attention":
            x = self._attention(ep)
        else:
            raise NotImplementedError("Not implement %s pooling" % ptype)
        ep["pooling"] = x
        a, b = self.seg
        x = F.linear(x, self.w["%s/tdnn%d_dense/kernel" % (sc, a)].t(), self.w["%s/tdnn%d_dense/bias" % (sc, a)])
        ep["tdnn%d_dense" % a] = x
        x = self._bn(x, "%s/tdnn%d_bn" % (sc, a))
        ep["tdnn%d_bn" % a] = x
        x = self._act(x, "%s/tdnn%d_relu" % (sc, a))
        ep["tdnn%d_relu" % a] = x
        x = F.linear(x, self.w["%s/tdnn%d_dense/kernel" % (sc, b)].t(), self.w["%s/tdnn%d_dense/bias" % (sc, b)])
        ep["tdnn%d_dense" % b] = x
        if not _get(p, "last_layer_no_bn", False):
            x = self._bn(x, "%s/tdnn%d_bn" % (sc, b))
            ep["tdnn%d_bn" % b] = x
        if not _get(p, "last_layer_linear", False):
            x = self._act(x, "%s/tdnn%d_relu" % (sc, b))
            ep["tdnn%d_relu" % b] = x
        ep["output"] = x
        if _get(p, "feature_norm", False):                 # model/trainer.py:400-403
            sq = (x * x).sum(-1, keepdim=True)
            ep["output"] = x * (float(_get(p, "feature_scaling_factor")) * torch.rsqrt(torch.clamp(sq, min=1e-12)))
        return ep

    def predict(self, features, dim, node=None):
        """model/trainer.py:886-913."""
        features = np.asarray(features)
        rank = features.ndim
        assert rank in (2, 3)
        if rank == 2:
            features = features[None]
        if features.shape[-1] != dim:
            features = features[:, :, :dim]
        ep = self.forward(features)
        e = ep[node if node else _get(self.params, "embedding_node")].numpy()
        return e[0] if rank == 2 else e


class TorchResnet18(object):
    """fp32 torch restatement of model/resnet.py:152-351 with F.conv2d on NCHW tensors and
    explicit F.pad for TensorFlow's asymmetric 'same' padding (independent of the tap-loop
    formulation in oracle/ref_numpy.py)."""

    def __init__(self, weights, params, dtype=torch.float32):
        self.params = params
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))).to(dtype) for k, v in weights.items()}
        self.scope = "resnet_18"

    def _bn(self, x, name):                                # x NCHW or [b,l,c]
        w, s = self.w, self.scope + "/" + name
        sc = w[s + "/gamma"] / torch.sqrt(w[s + "/moving_variance"] + BN_EPSILON)
        sh = w[s + "/beta"] - w[s + "/moving_mean"] * sc
        if x.dim() == 4:
            return x * sc[None, :, None, None] + sh[None, :, None, None]
        return x * sc + sh

    def _act(self, x, name):
        t = _get(self.params, "network_relu_type")
        if t == "prelu":
            a = self.w[self.scope + "/" + name + "/alpha"]
            if x.dim() == 4:
                a = a[None, :, None, None]
            return F.relu(x) + a * (x - x.abs()) * 0.5
        if t == "lrelu":
            return F.leaky_relu(x, LRELU_ALPHA)
        return F.relu(x)

    def _conv_same(self, x, name, strides=(1, 1)):
        k = self.w[self.scope + "/" + name + "/kernel"]    # HWIO
        kh, kw = k.shape[0], k.shape[1]
        pads = []
        for n, kk, st in ((x.shape[3], kw, strides[1]), (x.shape[2], kh, strides[0])):   # F.pad: last dim first
            out = -(-n // st)
            tot = max((out - 1) * st + kk - n, 0)
            pads += [tot // 2, tot - tot // 2]
        return F.conv2d(F.pad(x, pads), k.permute(3, 2, 0, 1).contiguous(), stride=strides)

    def _block(self, x, name, strides, projection, ep):
        f = self._act(self._bn(self._conv_same(x, name + "_conv0", strides), name + "_bn0"), name + "_relu0")
        f = self._bn(self._conv_same(f, name + "_conv1"), name + "_bn1")
        sc = self._bn(self._conv_same(x, name + "_conv_short", strides), name + "_bn_short") if projection else x
        f = self._act(f + sc, name + "_relu_final")
        ep[name] = f.permute(0, 2, 3, 1)
        return f

    @torch.no_grad()
    def forward(self, features):
        p, sc = self.params, self.scope
        ts = 2 if _get(p, "resnet_time_stride", False) else 1                                # resnet.py:187
        blocks = list(_get(p, "resnet_blocks", [2, 2, 2, 2]))
        ep = OrderedDict()
        x = torch.as_tensor(np.ascontiguousarray(features)).to(self.dtype)[:, None]       # [b,1,l,40]
        x = self._act(self._bn(self._conv_same(x, "conv0_1"), "conv0_bn"), "conv0_relu")
        ep["conv0_relu"] = x.permute(0, 2, 3, 1)
        if _get(p, "resnet_maxpooling", False):                                              # resnet.py:230-231
            x = F.max_pool2d(x, 3, 1, padding=1)                                             # pads with -inf: 'same'
            ep["conv0_max"] = x.permute(0, 2, 3, 1)
        for stage in (1, 2, 3, 4):
            x = self._block(x, "conv%da" % stage, (1, 1) if stage == 1 else (ts, 2), True, ep)
            for i in range(blocks[stage - 1] - 1):
                x = self._block(x, "conv%db_%d" % (stage, i), (1, 1), False, ep)
        k5 = self.w[sc + "/conv5/kernel"].permute(3, 2, 0, 1).contiguous()
        x = F.conv2d(x, k5, self.w[sc + "/conv5/bias"])
        x = self._act(self._bn(x, "conv5_bn"), "conv5_relu")[:, :, :, 0].transpose(1, 2)   # [b,l,512]
        ep["conv5_relu"] = x
        for name in ("dense1", "dense2"):
            x = F.linear(x, self.w["%s/%s/kernel" % (sc, name)].t(), self.w["%s/%s/bias" % (sc, name)])
            x = self._act(self._bn(x, name + "_bn"), name + "_relu")
            ep[name + "_relu"] = x
        if _get(p, "pooling_type") != "statistics_pooling":
            raise NotImplementedError("resnet_18 registers no frame-level endpoints for attention inputs")
        mean = x.mean(dim=1, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=1)
        var = torch.where(var <= VAR2STD_EPSILON, torch.full_like(var, VAR2STD_EPSILON), var)
        x = torch.cat([mean[:, 0], var.sqrt()], dim=1)
        ep["pooling"] = x
        x = F.linear(x, self.w[sc + "/tdnn6_dense/kernel"].t(), self.w[sc + "/tdnn6_dense/bias"])
        ep["tdnn6_dense"] = x
        x = self._bn(x, "tdnn6_bn")
        ep["tdnn6_bn"] = x
        x = self._act(x, "tdnn6_relu")
        ep["tdnn6_relu"] = x
        x = F.linear(x, self.w[sc + "/tdnn7_dense/kernel"].t(), self.w[sc + "/tdnn7_dense/bias"])
        ep["tdnn7_dense"] = x
        if not _get(p, "last_layer_no_bn", False):
            x = self._bn(x, "tdnn7_bn")
            ep["tdnn7_bn"] = x
        if not _get(p, "last_layer_linear", False):
            x = self._act(x, "tdnn7_relu")
            ep["tdnn7_relu"] = x
        ep["output"] = x
        if _get(p, "feature_norm", False):
            sq = (x * x).sum(-1, keepdim=True)
            ep["output"] = x * (float(_get(p, "feature_scaling_factor")) * torch.rsqrt(torch.clamp(sq, min=1e-12)))
        return ep

    def predict(self, features, dim=40, node=None):
        features = np.asarray(features)
        rank = features.ndim
        if rank == 2:
            features = features[None]
        ep = self.forward(features[:, :, :dim])
        e = ep[node if node else _get(self.params, "embedding_node")].numpy()
        return e[0] if rank == 2 else e
