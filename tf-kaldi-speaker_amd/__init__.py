"""MI355X-native x-vector embedding extractor (hot path of tf-kaldi-speaker).

Host-side mirror of the reference's predict surface (`Trainer`, `Params`, `kaldi_io`,
the `extract.py` driver) over a C-ABI HIP library (`csrc/`, `include/xvec_hip.h`).
"""
from .params import Params, ParamsPlain  # noqa: F401

__all__ = ["Params", "ParamsPlain"]
