"""
graph.py -- the few lines of TF-1.x surface the reference's hot path is driven through:
placeholders, named tensors in a model dict, and `Session.run(fetches, feed_dict)`
(e.g. vdsr/vdsr/experiment_train.py:17-26,134-151).  Nothing is traced or compiled: a fetch
simply asks the owning model object to execute the engine eagerly on the GPU.
"""
import numpy as np
import torch


class Tensor(object):
    """A named handle: either a placeholder (owner None until a model adopts it) or an output
    of a model (`owner.run` knows how to produce `key`)."""

    def __init__(self, name, shape=None, owner=None, key=None):
        self.name = name
        self.shape = shape
        self.owner = owner
        self.key = key

    def __repr__(self):
        return '<srx Tensor %s>' % self.name

    __hash__ = object.__hash__


def placeholder(shape=None, dtype='float32', name=None):
    """tf.placeholder(shape=..., dtype=tf.float32, name=...)."""
    return Tensor(name or 'Placeholder', shape=shape)


def to_device(value, device):
    """Feed values may be numpy arrays (as the reference feeds) or device tensors (zero-copy)."""
    if torch.is_tensor(value):
        t = value.to(device=device, dtype=torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32)).to(device)
    return t.contiguous()


class Session(object):
    """tf.Session() stand-in.  `run` accepts a Tensor, a list/tuple or a dict of Tensors and
    returns the same structure of numpy values (python scalars for 0-d)."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None):
        feed_dict = feed_dict or {}
        if isinstance(fetches, Tensor):
            return self._run_flat([fetches], feed_dict)[0]
        if isinstance(fetches, dict):
            keys = list(fetches.keys())
            vals = self._run_flat([fetches[k] for k in keys], feed_dict)
            return dict(zip(keys, vals))
        return type(fetches)(self._run_flat(list(fetches), feed_dict))

    @staticmethod
    def _run_flat(tensors, feed_dict):
        owners = []
        for t in tensors:
            if t.owner is None:
                raise ValueError('%r is a placeholder that no model consumes; feed it instead' % t)
            if t.owner not in owners:
                owners.append(t.owner)
        results = {}
        for owner in owners:
            keys = [t.key for t in tensors if t.owner is owner]
            results[owner] = owner.run(keys, feed_dict)
        return [results[t.owner][t.key] for t in tensors]
