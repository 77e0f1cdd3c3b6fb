"""Synthetic model weights and feature matrices for tests and bench (there are no trained
checkpoints in the reference tree and no network: README.md:101-119 are remote links).

Variable names and shapes are those the reference graph creates (model/tdnn.py:36-181,
model/pooling.py:96-229, model/common.py:27-42,113-225), so a real checkpoint converted to
the same name->array dict is interchangeable.  Distributions follow SURVEY.md 8(d):
kernels Glorot-uniform (the TF default initialiser), non-identity BN statistics so that a
BN-folding mistake cannot hide, query ~ truncated N(0, 0.1) (model/pooling.py:182-183).
"""
from collections import OrderedDict

import numpy as np


def _get(params, key, default=None):
    d = params if isinstance(params, dict) else params.dict
    return d.get(key, default)


def _glorot(rs, shape):
    receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = receptive * shape[-2], receptive * shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)


def _bn(rs, w, scope, n):
    w[scope + "/gamma"] = rs.uniform(0.5, 1.5, n).astype(np.float32)
    w[scope + "/beta"] = (0.1 * rs.standard_normal(n)).astype(np.float32)
    w[scope + "/moving_mean"] = (0.1 * rs.standard_normal(n)).astype(np.float32)
    w[scope + "/moving_variance"] = rs.uniform(0.5, 1.5, n).astype(np.float32)


def _prelu(rs, w, scope, n, params):
    if _get(params, "network_relu_type") == "prelu":
        # init 0.01 in the reference; jittered so a per-channel indexing bug shows
        w[scope + "/alpha"] = (0.01 + 0.2 * rs.uniform(0, 1, n)).astype(np.float32)


_NETS = {"tdnn": ("tdnn", (5, 5, 7, 1, 1), 4, (6, 7)),                                  # model/tdnn.py:36-181
         "extended_tdnn": ("etdnn", (5, 1, 5, 1, 7, 1, 9, 1, 1, 1), 3, (12, 13))}       # model/tdnn.py:343-591


def net_spec(params):
    """(variable scope, frame-layer kernel widths, conv kernel rank, segment layer indices)."""
    return _NETS[_get(params, "network_type", "tdnn")]


def tdnn_layer_dims(params, dim, channels=512):
    """(name, kernel_width, cin, cout) of the affine layers of model/tdnn.py tdnn() / etdnn()."""
    _, widths, _, seg = net_spec(params)
    pool_nodes = int(_get(params, "num_nodes_pooling_layer", 1500))
    last_nodes = int(_get(params, "num_nodes_last_layer", 512))
    pool_dim = pooling_output_dim(params, pool_nodes, channels)
    out, cin = [], dim
    for i, w in enumerate(widths, start=1):
        cout = pool_nodes if i == len(widths) else channels
        out.append(("tdnn%d_%s" % (i, "conv" if w > 1 else "dense"), w, cin, cout))
        cin = cout
    out.append(("tdnn%d_dense" % seg[0], 1, pool_dim, channels))
    out.append(("tdnn%d_dense" % seg[1], 1, channels, last_nodes))
    return out


def _node_width(params, node, channels, pool_nodes):
    """Channel count of a frame-level endpoint used as attention key/value input."""
    nframe = len(net_spec(params)[1])
    if node.startswith("tdnn%d_" % nframe):
        return pool_nodes
    return channels


def pooling_output_dim(params, pool_nodes, channels=512):
    if _get(params, "pooling_type") == "self_attention":
        vn = list(_get(params, "att_value_num_nodes"))
        dv = vn[-1] if vn else _node_width(params, _get(params, "att_value_input"), channels, pool_nodes)
        h = int(_get(params, "att_num_heads"))
        return 2 * (dv if _get(params, "att_split_value") else dv * h)
    return 2 * pool_nodes


def synth_weights(params, dim, seed=0, channels=512):
    """name -> float32 array for every variable of the predict graph.  `channels` is 512 in
    the reference (hard-coded, model/tdnn.py:43); tests shrink it to keep the oracle fast."""
    rs = np.random.RandomState(seed)
    w = OrderedDict()
    pool_nodes = int(_get(params, "num_nodes_pooling_layer", 1500))
    scope, _, rank, seg = net_spec(params)
    for name, k, cin, cout in tdnn_layer_dims(params, dim, channels):
        idx = name[4:name.index("_")]
        vs = scope + "/" + name
        if name.endswith("conv"):
            w[vs + "/kernel"] = _glorot(rs, (1, k, cin, cout) if rank == 4 else (k, cin, cout))
        else:
            w[vs + "/kernel"] = _glorot(rs, (cin, cout))
        w[vs + "/bias"] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
        last = idx == str(seg[1])
        if not (last and _get(params, "last_layer_no_bn", False)):
            _bn(rs, w, "%s/tdnn%s_bn" % (scope, idx), cout)
        if not (last and _get(params, "last_layer_linear", False)):
            _prelu(rs, w, "%s/tdnn%s_relu" % (scope, idx), cout, params)
    if _get(params, "pooling_type") == "self_attention":
        base = scope + "/attention"
        for which in ("key", "value"):
            nodes = list(_get(params, "att_%s_num_nodes" % which))
            cin = _node_width(params, _get(params, "att_%s_input" % which), channels, pool_nodes)
            last_kind = int(_get(params, "att_%s_network_type" % which))
            for i, n in enumerate(nodes):
                name = "att_%s%d" % (which, i)
                kind = 2 if i < len(nodes) - 1 else last_kind
                w["%s/%s/%s_dense/kernel" % (base, name, name)] = _glorot(rs, (cin, n))
                w["%s/%s/%s_dense/bias" % (base, name, name)] = (0.1 * rs.standard_normal(n)).astype(np.float32)
                if kind == 2:
                    _bn(rs, w, "%s/%s/%s_bn" % (base, name, name), n)
                if kind in (1, 2):
                    _prelu(rs, w, "%s/%s/%s_relu" % (base, name, name), n, params)
                cin = n
            if which == "key":
                dk = cin
        h = int(_get(params, "att_num_heads"))
        dq = dk // h if _get(params, "att_split_key") else dk
        w[base + "/query"] = np.clip(0.1 * rs.standard_normal((h, dq)), -0.2, 0.2).astype(np.float32)
        if _get(params, "att_apply_nonlinear"):
            n = pooling_output_dim(params, pool_nodes, channels)
            _bn(rs, w, base + "/att_post_bn", n)
            _prelu(rs, w, base + "/att_post_relu", n, params)
    return w


def synth_resnet_weights(params, seed=0, width=64):
    """Variables of model/resnet.py:152-351 (scope "resnet_18").  `width` = channels of stage 1
    (64 in the reference; stages are width, 2w, 4w, 8w; the dense/segment layers use 8w and
    num_nodes_pooling_layer); tests shrink it."""
    rs = np.random.RandomState(seed)
    w = OrderedDict()
    sc = "resnet_18"
    blocks = list(_get(params, "resnet_blocks", [2, 2, 2, 2]))
    pool_nodes = int(_get(params, "num_nodes_pooling_layer", 1500))
    last_nodes = int(_get(params, "num_nodes_last_layer", 512))

    def conv(name, kh, kw, cin, cout, bias=False):
        w["%s/%s/kernel" % (sc, name)] = _glorot(rs, (kh, kw, cin, cout))
        if bias:
            w["%s/%s/bias" % (sc, name)] = (0.1 * rs.standard_normal(cout)).astype(np.float32)

    def bn_act(bn_name, relu_name, n):
        _bn(rs, w, "%s/%s" % (sc, bn_name), n)
        if relu_name:
            _prelu(rs, w, "%s/%s" % (sc, relu_name), n, params)

    conv("conv0_1", 3, 3, 1, width)
    bn_act("conv0_bn", "conv0_relu", width)
    cin = width
    for stage in (1, 2, 3, 4):
        nf = width << (stage - 1)
        names = ["conv%da" % stage] + ["conv%db_%d" % (stage, i) for i in range(blocks[stage - 1] - 1)]
        for bi, name in enumerate(names):
            conv(name + "_conv0", 3, 3, cin, nf)
            bn_act(name + "_bn0", name + "_relu0", nf)
            conv(name + "_conv1", 3, 3, nf, nf)
            bn_act(name + "_bn1", None, nf)
            if bi == 0:
                conv(name + "_conv_short", 1, 1, cin, nf)
                bn_act(name + "_bn_short", None, nf)
            _prelu(rs, w, "%s/%s_relu_final" % (sc, name), nf, params)
            cin = nf
    c8 = width << 3
    conv("conv5", 1, 5, c8, c8, bias=True)
    bn_act("conv5_bn", "conv5_relu", c8)
    for name, a, b in (("dense1", c8, c8), ("dense2", c8, pool_nodes), ("tdnn6_dense", 2 * pool_nodes, c8),
                       ("tdnn7_dense", c8, last_nodes)):
        w["%s/%s/kernel" % (sc, name)] = _glorot(rs, (a, b))
        w["%s/%s/bias" % (sc, name)] = (0.1 * rs.standard_normal(b)).astype(np.float32)
    bn_act("dense1_bn", "dense1_relu", c8)
    bn_act("dense2_bn", "dense2_relu", pool_nodes)
    bn_act("tdnn6_bn", "tdnn6_relu", c8)
    if not _get(params, "last_layer_no_bn", False):
        _bn(rs, w, sc + "/tdnn7_bn", last_nodes)
    if not _get(params, "last_layer_linear", False):
        _prelu(rs, w, sc + "/tdnn7_relu", last_nodes, params)
    return w


RESNET_PARAMS = {             # egs/voxceleb/v3/nnet_conf/resnet18_softmax_1e-2.json (hot-path keys)
    "seed": 0, "network_type": "resnet_18", "resnet_time_stride": False, "resnet_maxpooling": False,
    "last_layer_linear": False, "loss_func": "softmax", "pooling_type": "statistics_pooling",
    "embedding_node": "tdnn6_dense", "weight_l2_regularizer": 1e-2, "batchnorm_momentum": 0.99,
    "keep_checkpoint_max": 10,
}


def synth_features(num, frames, dim, seed=1234):
    """Post-CMVN-like features: N(0,1) float32.  `frames` is an int (uniform) or a sequence
    of per-utterance lengths.  Returns a list of [T,dim] arrays."""
    rs = np.random.RandomState(seed)
    lens = [int(frames)] * num if np.isscalar(frames) else [int(t) for t in frames]
    assert len(lens) == num
    return [rs.standard_normal((t, dim)).astype(np.float32) for t in lens]


TDNN_STAT_PARAMS = {          # egs/voxceleb/v1/nnet_conf/tdnn_softmax_1e-2.json (hot-path keys)
    "seed": 0, "network_type": "tdnn", "last_layer_linear": False, "loss_func": "softmax",
    "pooling_type": "statistics_pooling", "embedding_node": "tdnn6_dense",
    "weight_l2_regularizer": 1e-2, "batchnorm_momentum": 0.99, "keep_checkpoint_max": 100,
}

TDNN_ATT_PARAMS = dict(TDNN_STAT_PARAMS, **{   # ..._tdnn4_att_pretrain.json:15-28
    "num_nodes_pooling_layer": 1500, "pooling_type": "self_attention",
    "att_key_input": "tdnn4_relu", "att_key_num_nodes": [1500, 1500], "att_key_network_type": 1,
    "att_value_input": "tdnn5_relu", "att_value_num_nodes": [], "att_value_network_type": 0,
    "att_apply_nonlinear": False, "att_use_scale": True, "att_num_heads": 1,
    "att_split_value": True, "att_split_key": True, "att_penalty_term": 0,
})
