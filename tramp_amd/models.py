"""Model / graph algebra (host orchestration, CPU).

Minimal counterpart of tramp/models/{dag_algebra,base_model,generalized_linear_model,
multi_layer_model}.py for tree-structured models built with ``@``: a bipartite DAG
of Factor and Variable nodes held in a networkx DiGraph (networkx >= 2 API; the
reference needs networkx 1.x), topological order, shape and
second-moment propagation, channel -> likelihood conversion for observed leaves,
forward sampling.
"""
import networkx as nx
import numpy as np

from .base import Factor, Variable
from .channels import (ConvChannel, DFTChannel, GaussianChannel, HardTanhChannel, HardSigmoidChannel, SymmetricDoorChannel, SgnChannel, AbsChannel, ReluChannel, LeakyReluChannel, LinearChannel,
                       PiecewiseLinearChannel, get_channel)
from .ensembles import get_ensemble
from .likelihoods import (GaussianLikelihood, SgnLikelihood, AbsLikelihood, ReluLikelihood,
                          LeakyReluLikelihood, HardTanhLikelihood, HardSigmoidLikelihood,
                          SymmetricDoorLikelihood, Likelihood)
from .priors import Prior, get_prior
from .variables import SISOVariable, SILeafVariable


def channel2likelihood(channel, y, y_name):
    "tramp/models/dag_algebra.py:21-43"
    kw = dict(dtype=getattr(channel, "dtype", np.float64), ctx=getattr(channel, "_ctx", None))
    if isinstance(channel, GaussianChannel):
        return GaussianLikelihood(y=y, y_name=y_name, var=channel.var)
    if isinstance(channel, AbsChannel):
        return AbsLikelihood(y=y, y_name=y_name, **kw)
    if isinstance(channel, SgnChannel):
        return SgnLikelihood(y=y, y_name=y_name, **kw)
    if isinstance(channel, ReluChannel):
        return ReluLikelihood(y=y, y_name=y_name, **kw)
    if isinstance(channel, LeakyReluChannel):
        return LeakyReluLikelihood(slope=channel.slope, y=y, y_name=y_name, **kw)
    if isinstance(channel, HardTanhChannel):
        return HardTanhLikelihood(y=y, y_name=y_name, **kw)
    if isinstance(channel, HardSigmoidChannel):
        return HardSigmoidLikelihood(y=y, y_name=y_name, **kw)
    if isinstance(channel, SymmetricDoorChannel):
        return SymmetricDoorLikelihood(width=channel.width, y=y, y_name=y_name, **kw)
    raise NotImplementedError(f"cannot convert {channel} to likelihood")


class DAG:
    """Chain of nodes composed with ``@`` (tramp/models/dag_algebra.py:91-173)."""

    def __init__(self, node_or_nodes):
        self.nodes = list(node_or_nodes) if isinstance(node_or_nodes, (list, tuple)) else [node_or_nodes]

    def __matmul__(self, other):
        if not isinstance(other, DAG):
            other = DAG(other)
        return DAG(self.nodes + other.nodes)

    def to_model(self):
        return Model(self)


class Model:
    """tramp/models/base_model.py:28-148 for SISO chains."""

    def __init__(self, dag):
        if isinstance(dag, DAG):
            nodes = dag.nodes
        else:
            nodes = list(dag)
        g = nx.DiGraph()
        for u, v in zip(nodes[:-1], nodes[1:]):
            if isinstance(u, Factor) == isinstance(v, Factor):
                raise ValueError(f"{u} @ {v}: factors and variables must alternate")
            g.add_edge(u, v)
        if len(nodes) == 1:
            g.add_node(nodes[0])
        if not nx.is_directed_acyclic_graph(g):
            raise ValueError("model graph is not a DAG")
        for node in g.nodes:
            if g.in_degree(node) != node.n_prev or g.out_degree(node) != node.n_next:
                raise ValueError(f"node {node} has {g.in_degree(node)} predecessors / "
                                 f"{g.out_degree(node)} successors, expected {node.n_prev} / {node.n_next}")
        self.dag = g
        self.forward_ordering = list(nx.topological_sort(g))
        self.variables = [n for n in self.forward_ordering if isinstance(n, Variable)]
        self.variable_ids = [v.id for v in self.variables]
        if len(set(self.variable_ids)) != len(self.variable_ids):
            raise ValueError("duplicate variable ids")
        self.factors = [n for n in self.forward_ordering if isinstance(n, Factor)]
        for idx, f in enumerate(self.factors):
            f.id = f"f_{idx}"                                   # base_model.py:23-25
        self.n_variables = len(self.variables)
        self.n_factors = len(self.factors)

    # -- structure helpers for the scheduler
    def prev_factor(self, variable):
        return next(iter(self.dag.predecessors(variable)))

    def next_factor(self, variable):
        succ = list(self.dag.successors(variable))
        return succ[0] if succ else None

    def to_observed(self, observations):
        "tramp/models/dag_algebra.py:243-291: observed leaf + its channel -> likelihood"
        nodes = []
        order = self.forward_ordering
        for i, node in enumerate(order):
            if isinstance(node, Variable) and node.id in observations:
                if node.n_next != 0:
                    raise ValueError(f"{node} not a leaf")
                continue
            if isinstance(node, Factor) and i + 1 < len(order) and \
                    isinstance(order[i + 1], Variable) and order[i + 1].id in observations:
                vid = order[i + 1].id
                nodes.append(channel2likelihood(node, y=observations[vid], y_name=vid))
            else:
                nodes.append(node)
        return Model(DAG(nodes))

    def sample(self, seed=0):
        "Forward sampling (base_model.py:71-94)"
        if seed != 0:
            np.random.seed(seed)
        X = {}
        prev = None
        for node in self.forward_ordering:
            if isinstance(node, Factor):
                if node.n_next == 0:
                    continue
                prev = node.sample() if node.n_prev == 0 else node.sample(prev)
            else:
                X[node.id] = prev
        return X

    def init_shapes(self):
        "base_model.py:96-109 (shapes from the factors' geometry rather than a dry-run sample)"
        shapes = {}
        shape = None
        for node in self.forward_ordering:
            if isinstance(node, Prior):
                shape = (node.size,) if np.isscalar(node.size) else tuple(node.size)
            elif isinstance(node, LinearChannel):
                shape = (node.Nx,)
            elif isinstance(node, ConvChannel):
                shape = tuple(node.shape)
            elif isinstance(node, DFTChannel):
                shape = (2,) + tuple(shape)
            if isinstance(node, Variable):
                shapes[node.id] = shape
                self.dag.nodes[node]["shape"] = shape
        return shapes

    def get_shapes(self):
        return {v.id: self.dag.nodes[v].get("shape") for v in self.variables}

    def init_second_moments(self):
        "base_model.py:111-124"
        tau = None
        for node in self.forward_ordering:
            if isinstance(node, Factor) and node.n_next:
                tau = node.second_moment() if node.n_prev == 0 else node.second_moment(tau)
            elif isinstance(node, Variable):
                self.dag.nodes[node]["tau"] = tau

    def get_second_moments(self):
        return {v.id: self.dag.nodes[v].get("tau") for v in self.variables}


def _get_kwargs(target, kwargs):
    n = len(target) + 1
    return {k[n:]: v for k, v in kwargs.items() if k.startswith(target)}


def glm_generative(N, alpha, ensemble_type, prior_type, output_type, dtype=np.float64, ctx=None, **kwargs):
    "tramp/models/generalized_linear_model.py:17-36"
    M = int(alpha * N)
    F = get_ensemble(ensemble_type, M=M, N=N, **_get_kwargs("ensemble", kwargs)).generate()
    prior = get_prior(size=N, prior_type=prior_type, dtype=dtype, ctx=ctx, **_get_kwargs("prior", kwargs))
    linear = get_channel("linear", W=F, name="F", dtype=dtype, ctx=ctx, keep_W=True)
    okw = _get_kwargs("output", kwargs)
    if output_type != "gaussian":
        okw.update(dtype=dtype, ctx=ctx)
    output = get_channel(output_type, **okw)
    return (prior @ SISOVariable(id="x") @ linear @ SISOVariable(id="z") @ output
            @ SILeafVariable(id="y")).to_model()


def glm_state_evolution(alpha, prior_type, output_type, **kwargs):
    "tramp/models/generalized_linear_model.py:38-55"
    from .likelihoods import get_likelihood
    prior = get_prior(size=None, prior_type=prior_type, **_get_kwargs("prior", kwargs))
    linear = get_channel("marchenko", alpha=alpha, name="F")
    output = get_likelihood(y=None, likelihood_type=output_type, **_get_kwargs("output", kwargs))
    return (prior @ SISOVariable(id="x") @ linear @ SISOVariable(id="z") @ output).to_model()


class MultiLayerModel(Model):
    "tramp/models/multi_layer_model.py:30-49"

    def __init__(self, layers, ids=None):
        if not isinstance(layers[0], Prior):
            raise ValueError("first layer must be a Prior")
        n = len(layers)
        if ids is None:
            ids = [f"t_{l}" for l in range(n)]
            ids[0] = "x"
            if n > 1:
                ids[-1] = "y"
        if len(ids) != n:
            raise ValueError(f"ids should be of length {n}")
        self.layers, self.ids, self.n_layers = layers, ids, n
        nodes = []
        for l, layer in enumerate(layers):
            nodes.append(layer)
            if isinstance(layer, Likelihood):
                break
            V = SILeafVariable if l == n - 1 else SISOVariable
            nodes.append(V(id=ids[l]))
        super().__init__(DAG(nodes))


__all__ = ["DAG", "Model", "MultiLayerModel", "glm_generative", "glm_state_evolution",
           "channel2likelihood", "PiecewiseLinearChannel"]
