"""Minimal stand-in for the OmegaConf calls on the inference path (get_model.py:15,19; INN.py:38): loads a YAML
file with PyYAML into nested attribute/item-accessible nodes.  The reference's configs use no interpolation.
Missing keys yield ``None`` like the pinned omegaconf 2.0.5 does (get_model.py:42 reads ``Training['control']``
unconditionally; only BAIR's config defines it)."""
import yaml


class Node(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __getitem__(self, k):
        return dict.get(self, k)

    def get(self, k, default=None):
        return dict.get(self, k, default)


def _wrap(o):
    if isinstance(o, dict):
        return Node({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def load(path):
    with open(path) as f:
        return _wrap(yaml.safe_load(f))
