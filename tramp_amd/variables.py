"""Variable nodes (tramp/variables/sub_variables.py, tramp/base.py:49-233).

On the hot path a SISO variable only forwards messages between its two factors
(sub_variables.py:21-43) and forms the posterior r = sum b / sum a, v = 1 / sum a
(base.py:152-161), which kernel K11 computes together with the early-stopping norms.
"""
from .base import Variable


class SISOVariable(Variable):
    def __init__(self, id):
        super().__init__(id=id, n_prev=1, n_next=1)


class SILeafVariable(Variable):
    def __init__(self, id):
        super().__init__(id=id, n_prev=1, n_next=0)


class SIMOVariable(Variable):
    "one incoming factor, n_next outgoing (sub_variables.py:4-7)"

    def __init__(self, id, n_next):
        super().__init__(id=id, n_prev=1, n_next=n_next)


class MISOVariable(Variable):
    "n_prev incoming factors, one outgoing (sub_variables.py:10-13)"

    def __init__(self, id, n_prev):
        super().__init__(id=id, n_prev=n_prev, n_next=1)


class MILeafVariable(Variable):
    "leaf with n_prev incoming factors (sub_variables.py:46-49)"

    def __init__(self, id, n_prev):
        super().__init__(id=id, n_prev=n_prev, n_next=0)
