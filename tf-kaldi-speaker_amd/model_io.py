"""Model directory contract of the extraction path (egs/voxceleb/v1/nnet/lib/extract.py:42-55,
model/trainer.py:148,277-295):

    <model_dir>/nnet/config.json     hyper-parameters (Params)
    <model_dir>/nnet/feature_dim     one integer
    <model_dir>/nnet/checkpoint      TF-style state file: model_checkpoint_path: "model-<step>"
    <model_dir>/nnet/model-<step>.npz   name -> float32 array, keyed by TF variable name

Weights are either an .npz with the TF variable names as keys (written by `save_model`) or the
TensorFlow checkpoint-V2 bundle itself (`model-<step>.index` + `.data-*`), parsed without
TensorFlow by tf_checkpoint.py (unpinned: no real checkpoint available).
"""
import json
import os
import re

import numpy as np


def save_model(model_dir, params_dict, feature_dim, weights, step=0):
    nnet = os.path.join(model_dir, "nnet")
    os.makedirs(nnet, exist_ok=True)
    with open(os.path.join(nnet, "config.json"), "w") as f:
        json.dump(params_dict, f, indent=2)
    with open(os.path.join(nnet, "feature_dim"), "w") as f:
        f.write("%d\n" % feature_dim)
    name = "model-%d" % step
    np.savez(os.path.join(nnet, name + ".npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in weights.items()})
    with open(os.path.join(nnet, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))
    return nnet


def read_checkpoint_state(nnet_dir):
    """Return the checkpoint basename named by <nnet_dir>/checkpoint, or None."""
    path = os.path.join(nnet_dir, "checkpoint")
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        for line in f:
            m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
            if m:
                return os.path.basename(m.group(1))
    return None


def load_weights(nnet_dir):
    """-> (weights dict, step).  Step is the trailing number of the checkpoint name
    (model/trainer.py:288-289)."""
    name = read_checkpoint_state(nnet_dir)
    if not name:
        return None, None
    path = os.path.join(nnet_dir, name + ".npz")
    step = int(next(re.finditer(r"(\d+)(?!.*\d)", name)).group(0))
    if not os.path.isfile(path):
        prefix = os.path.join(nnet_dir, name)
        if os.path.isfile(prefix + ".index"):
            # a TensorFlow checkpoint-V2 bundle (what Saver.restore reads, model/trainer.py:290): parsed
            # without TensorFlow; optimizer slots / the loss layer are dropped later by name
            from . import tf_checkpoint
            weights = tf_checkpoint.read_bundle(prefix, name_filter=_graph_variable)
            return {k: v.astype(np.float32) for k, v in weights.items()}, step
        return None, None
    with np.load(path, allow_pickle=False) as z:
        weights = {k: z[k] for k in z.files}
    return weights, step


SOFTMAX_KERNEL = "softmax/output/kernel"    # [E, num_speakers]; only extract_angle.py reads it (extract_angle.py:64-67)


def _graph_variable(name):
    if name == SOFTMAX_KERNEL:
        return True
    return _network_variable(name)


def _network_variable(name):
    """Variables of the predict graph live under the network scope; skip optimizer slots
    (".../Momentum", ".../Adam*"), the global step and the loss layer ("softmax/...")."""
    if not name.startswith(("tdnn/", "etdnn/", "resnet_18/")):
        return False
    leaf = name.rsplit("/", 1)[-1]
    return leaf in ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_variance", "alpha", "query")
