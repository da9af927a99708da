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
