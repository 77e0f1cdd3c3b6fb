"""Model directory contract of the extraction path (egs/voxceleb/v1/nnet/lib/extract.py:42-55,
model/trainer.py:148,277-295):

    <model_dir>/nnet/config.json     hyper-parameters (Params)
    <model_dir>/nnet/feature_dim     one integer
    <model_dir>/nnet/checkpoint      TF-style state file: model_checkpoint_path: "model-<step>"
    <model_dir>/nnet/model-<step>.npz   name -> float32 array, keyed by TF variable name

The reference restores a TF checkpoint-V2 bundle through tf.train.Saver; reading that
format without TensorFlow is SURVEY.md 8(f) rank 2 (next).  Until then the weight
container is an .npz with the same variable names, written by `save_model`.
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
    if not os.path.isfile(path):
        if os.path.isfile(os.path.join(nnet_dir, name + ".index")):
            raise NotImplementedError(
                "%s is a TensorFlow checkpoint-V2 bundle; convert it to %s.npz (variable name -> array). "
                "A TF-free bundle reader is not part of this round." % (name, name))
        return None, None
    step = int(next(re.finditer(r"(\d+)(?!.*\d)", name)).group(0))
    with np.load(path, allow_pickle=False) as z:
        weights = {k: z[k] for k in z.files}
    return weights, step
