"""State evolution of the GLMs of the reference's examples from the informed start a0 = 1000 (examples/glm/data/*_mse_curves.py,
*_critical_lines.py): the variance each reaches at a few alpha.  python tools/probes/probe_se_informed.py"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/probes')
import tramp_amd as T
from se_cases import CASES
for name, kw, alphas in CASES:
    for al in alphas:
        try:
            v = T.find_state_evolution_mse("x", 1000, al, T.glm_state_evolution, **kw)
        except Exception as e:
            v = "ERR " + repr(e)[:80]
        print("mine", name, al, v, flush=True)
