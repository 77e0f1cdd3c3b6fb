"""Hyper-parameter bag.  Mirrors `Params` of the reference (misc/utils.py:13-41): a JSON
file becomes attributes, `.dict` gives dict-like access so that the reference idiom
`"key" in params.dict` keeps working (model/tdnn.py:29,114,152,163,173)."""
import json


class Params(object):
    def __init__(self, json_path=None, **kwargs):
        if json_path is not None:
            self.update(json_path)
        self.__dict__.update(kwargs)

    def save(self, json_path):
        with open(json_path, "w") as f:
            json.dump(self.__dict__, f, indent=4)

    def update(self, json_path):
        with open(json_path) as f:
            self.__dict__.update(json.load(f))

    @property
    def dict(self):
        return self.__dict__


class ParamsPlain(object):
    """misc/utils.py:44-62: parameters set by hand (no JSON)."""

    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    @property
    def dict(self):
        return self.__dict__
