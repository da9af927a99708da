"""Variable nodes (tramp/variables/sub_variables.py, tramp/base.py:49-233).

A variable is characterised by how many factors feed it (``n_prev``) and how many it feeds
(``n_next``).  On a chain a SISO variable only relays messages between its two factors
(sub_variables.py:21-43); with more neighbours it sends each one the sum of the others' messages
(base.py:186-211, `tramp_message_sum`).  The posterior r = sum b / sum a, v = 1 / sum a
(base.py:152-161) is kernel K11, which also forms the early-stopping norms.
"""
from .base import Variable


class _Degrees(Variable):
    """Variable whose degrees are fixed by the subclass; a degree left open there (``None``) is the
    constructor's second argument, as in the reference's signatures (``SIMOVariable(id, n_next)``)."""
    _prev = _next = None

    def __init__(self, id, degree=None, **named):
        degree = named.pop("n_prev", named.pop("n_next", degree))
        if named:
            raise TypeError(f"unexpected arguments {sorted(named)}")
        open_slots = (self._prev is None) + (self._next is None)
        if open_slots == 0 and degree is not None:
            raise TypeError(f"{type(self).__name__} takes no degree argument")
        if open_slots == 1 and degree is None:
            raise TypeError(f"{type(self).__name__} needs its free degree")
        super().__init__(id=id, n_prev=degree if self._prev is None else self._prev,
                         n_next=degree if self._next is None else self._next)


class SISOVariable(_Degrees):
    "one factor in, one factor out: the interior of a chain"
    _prev, _next = 1, 1


class SILeafVariable(_Degrees):
    "one factor in, none out: an output that gets observed"
    _prev, _next = 1, 0


class SIMOVariable(_Degrees):
    "one factor in, ``n_next`` out"
    _prev = 1


class MISOVariable(_Degrees):
    "``n_prev`` factors in, one out"
    _next = 1


class MILeafVariable(_Degrees):
    "``n_prev`` factors in, none out (e.g. a gradient field that also carries its own prior)"
    _next = 0
