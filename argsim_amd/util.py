"""small functional helpers (counterpart of reference src/util.py)."""
from collections.abc import Mapping
from functools import partial  # noqa: F401  (re-exported like the reference does)


def identity(x):
    return x


def comp(g, f, *fs):
    """comp(g, f, ...)(x) == g(f(...(x)))"""
    if fs:
        f = comp(f, *fs)
    return lambda x: g(f(x))


class Record(Mapping):
    """attribute-style read-only mapping; ``Record(d)``, ``Record(a=1)``, ``Record(r1, r2, b=2)``
    (reference src/util.py:30-58).  ``**record`` works because it is a Mapping."""

    def __init__(self, *records, **entries):
        for rec in records:
            for k, v in rec.items():
                setattr(self, k, v)
        for k, v in entries.items():
            setattr(self, k, v)

    def __repr__(self):
        return repr(self.__dict__)

    def __iter__(self):
        return iter(self.__dict__)

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return getattr(self, key)


def select(record, *keys):
    return Record({k: record[k] for k in keys})
