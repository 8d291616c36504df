"""Attribute bag for one protein (prograph/protein.py of the reference): `Protein(Sequence, **labels)`."""
import numpy as np


class Protein:
    def __init__(self, Sequence, **labels):
        self.Sequence = Sequence
        for name, value in labels.items():
            setattr(self, name, value)

    def __len__(self):
        return len(self.Sequence)

    def __eq__(self, other):
        return self.Sequence == other.Sequence

    def __getitem__(self, keys):
        if isinstance(keys, list):
            return tuple(self.__dict__[k] for k in keys)
        return self.__dict__[keys]

    def __repr__(self):
        def fmt(k, v):
            if isinstance(v, np.ndarray):
                return f"{k}=np.array({list(v)})"
            return f"{k}='{v}'" if isinstance(v, str) else f"{k}={v}"
        return "Protein(" + ",".join(fmt(k, v) for k, v in vars(self).items()) + ")"
