"""Model / graph algebra (host orchestration, CPU).

Minimal counterpart of tramp/models/{dag_algebra,base_model,generalized_linear_model,
multi_layer_model}.py for tree-structured models built with ``@`` and ``+``: a bipartite DAG
of Factor and Variable nodes held in a networkx DiGraph (networkx >= 2 API; the
reference needs networkx 1.x), topological order, shape and
second-moment propagation, channel -> likelihood conversion for observed leaves,
forward sampling.
"""
import networkx as nx
import numpy as np

from .base import Factor, Variable
from .channels import (AsymmetricAbsChannel, ConvChannel, DFTChannel, GradientChannel, ReshapeChannel, GaussianChannel, HardTanhChannel, HardSigmoidChannel, SymmetricDoorChannel, SgnChannel, AbsChannel, ReluChannel, LeakyReluChannel, LinearChannel,
                       ComplexLinearChannel, ModulusChannel, PiecewiseLinearChannel, get_channel)
from .ensembles import get_ensemble
from .likelihoods import (GaussianLikelihood, SgnLikelihood, AbsLikelihood, AsymmetricAbsLikelihood, ReluLikelihood,
                          LeakyReluLikelihood, HardTanhLikelihood, HardSigmoidLikelihood,
                          SymmetricDoorLikelihood, ModulusLikelihood, Likelihood)
from .priors import Prior, get_prior
from .variables import SISOVariable, SILeafVariable


def channel2likelihood(channel, y, y_name):
    "tramp/models/dag_algebra.py:21-43"
    kw = dict(dtype=getattr(channel, "dtype", np.float64), ctx=getattr(channel, "_ctx", None))
    if isinstance(channel, GaussianChannel):
        return GaussianLikelihood(y=y, y_name=y_name, var=channel.var, **kw)
    if isinstance(channel, AbsChannel):
        return AbsLikelihood(y=y, y_name=y_name, **kw)
    if isinstance(channel, AsymmetricAbsChannel):
        return AsymmetricAbsLikelihood(y=y, y_name=y_name, shift=channel.shift, **kw)
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
    if isinstance(channel, ModulusChannel):
        return ModulusLikelihood(y=y, y_name=y_name, **kw)                # dag_algebra.py:38-39
    raise NotImplementedError(f"cannot convert {channel} to likelihood")


class _Stub:
    "dangling end of an edge while a graph is being composed (a node's unconnected input or output)"
    __slots__ = ("kind",)

    def __init__(self, kind):
        self.kind = kind


class DAG:
    """Composition algebra of tramp/models/dag_algebra.py:91-130 on a networkx DiGraph.

    A single node enters with one dangling stub per missing predecessor / successor.  ``A + B``
    places two graphs side by side (their dangling ends are listed left to right); ``A @ B``
    additionally splices A's dangling outputs onto B's dangling inputs, pairwise in that order.
    A list of nodes is the chain n0 @ n1 @ ...
    """

    def __init__(self, obj):
        if isinstance(obj, DAG):
            self.g, self.ins, self.outs = obj.g, obj.ins, obj.outs
            return
        if isinstance(obj, (list, tuple)):
            d = DAG(obj[0])
            for node in obj[1:]:
                d = d @ node
            self.g, self.ins, self.outs = d.g, d.ins, d.outs
            return
        g = nx.DiGraph()
        g.add_node(obj)
        self.ins = [_Stub("in") for _ in range(obj.n_prev)]
        self.outs = [_Stub("out") for _ in range(obj.n_next)]
        for st in self.ins:
            g.add_edge(st, obj)
        for st in self.outs:
            g.add_edge(obj, st)
        self.g = g

    @classmethod
    def _of(cls, g, ins, outs):
        d = cls.__new__(cls)
        d.g, d.ins, d.outs = g, ins, outs
        return d

    def __add__(self, other):
        other = other if isinstance(other, DAG) else DAG(other)
        return DAG._of(nx.compose(self.g, other.g), self.ins + other.ins, self.outs + other.outs)

    def __matmul__(self, other):
        other = other if isinstance(other, DAG) else DAG(other)
        g = nx.compose(self.g, other.g)
        k = min(len(self.outs), len(other.ins))
        for out_stub, in_stub in zip(self.outs[:k], other.ins[:k]):
            (src,) = g.predecessors(out_stub)
            (dst,) = g.successors(in_stub)
            g.remove_node(out_stub)
            g.remove_node(in_stub)
            if isinstance(src, Factor) == isinstance(dst, Factor):
                raise ValueError(f"{src} @ {dst}: factors and variables must alternate")
            g.add_edge(src, dst)
        return DAG._of(g, self.ins + other.ins[k:], other.outs + self.outs[k:])

    @property
    def nodes(self):
        return [n for n in nx.topological_sort(self.g) if not isinstance(n, _Stub)]

    def to_model(self):
        return Model(self)

    def to_model_dag(self):
        "tramp/models/dag_algebra.py:215-219: the closed graph as a model DAG (here the same object; .to_model() follows)"
        if self.ins or self.outs:
            raise ValueError(f"model graph has {len(self.ins)} unconnected inputs and {len(self.outs)} unconnected outputs")
        return self


class Model:
    """tramp/models/base_model.py:28-148 for tree-structured (bipartite, acyclic) models."""

    def __init__(self, dag):
        dag = dag if isinstance(dag, DAG) else DAG(list(dag))
        if dag.ins or dag.outs:
            raise ValueError(f"model graph has {len(dag.ins)} unconnected inputs and {len(dag.outs)} "
                             "unconnected outputs")
        g = dag.g
        if not nx.is_directed_acyclic_graph(g):
            raise ValueError("model graph is not a DAG")
        for node in g.nodes:
            if g.in_degree(node) != node.n_prev or g.out_degree(node) != node.n_next:
                raise ValueError(f"node {node} has {g.in_degree(node)} predecessors / "
                                 f"{g.out_degree(node)} successors, expected {node.n_prev} / {node.n_next}")
            for nb in list(g.predecessors(node)) + list(g.successors(node)):
                if isinstance(nb, Factor) == isinstance(node, Factor):
                    raise ValueError(f"{nb} - {node}: factors and variables must alternate")
        if g.number_of_nodes() and not nx.is_tree(g.to_undirected()):
            raise ValueError("model graph is not a tree (loopy or disconnected)")
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
        self.is_chain = all(v.n_prev == 1 and v.n_next <= 1 for v in self.variables) and \
            all(f.n_prev <= 1 and f.n_next <= 1 for f in self.factors)

    # -- structure helpers for the scheduler
    def prev_factors(self, variable):
        return list(self.dag.predecessors(variable))

    def next_factors(self, variable):
        return list(self.dag.successors(variable))

    def prev_factor(self, variable):
        return next(iter(self.dag.predecessors(variable)))

    def next_factor(self, variable):
        succ = list(self.dag.successors(variable))
        return succ[0] if succ else None

    def plot(self, layout=None):
        """the factor graph drawn with networkx / matplotlib (tramp/models/base_model.py:57-58): variables as circles with their ids,
        factors as squares with their class names"""
        import matplotlib.pyplot as plt
        g = self.dag
        pos = layout or nx.spring_layout(g, seed=0)
        fig, ax = plt.subplots(1, 1, figsize=(6, 4))
        nx.draw_networkx_edges(g, pos, ax=ax)
        nx.draw_networkx_nodes(g, pos, nodelist=self.variables, node_shape="o", node_color="w", edgecolors="k", ax=ax)
        nx.draw_networkx_nodes(g, pos, nodelist=self.factors, node_shape="s", node_color="lightgray", edgecolors="k", ax=ax)
        labels = {n: (n.id if isinstance(n, Variable) else type(n).__name__.replace("Channel", "").replace("Prior", "")
                      .replace("Likelihood", "")) for n in g.nodes}
        nx.draw_networkx_labels(g, pos, labels=labels, font_size=8, ax=ax)
        ax.set_axis_off()
        return fig

    def to_observed(self, observations):
        "tramp/models/dag_algebra.py:243-291: observed leaf + its channel -> likelihood"
        g = self.dag.copy()
        for var in self.variables:
            if var.id not in observations:
                continue
            if var.n_next != 0 or var.n_prev != 1:
                raise ValueError(f"{var} not a single-input leaf")
            (channel,) = g.predecessors(var)
            lik = channel2likelihood(channel, y=observations[var.id], y_name=var.id)
            preds = list(g.predecessors(channel))
            g.remove_node(var)
            g.remove_node(channel)
            for p in preds:
                g.add_edge(p, lik)
        return Model(DAG._of(g, [], []))

    def sample(self, seed=0):
        "Forward sampling (base_model.py:71-94): each factor samples its successors from its predecessors"
        if seed != 0:
            np.random.seed(seed)
        X = {}
        for node in self.forward_ordering:
            if not isinstance(node, Factor) or node.n_next == 0:
                continue
            args = [X[p.id] for p in self.dag.predecessors(node)]
            val = node.sample(*args)
            for succ in self.dag.successors(node):
                X[succ.id] = val
        return X

    def _out_shape(self, factor, in_shape):
        if isinstance(factor, Prior):
            return (factor.size,) if np.isscalar(factor.size) else tuple(factor.size)
        if isinstance(factor, ComplexLinearChannel):
            return (2, factor.Nx)                   # planar complex vectors
        if isinstance(factor, ModulusChannel):
            return tuple(in_shape[1:])
        if isinstance(factor, LinearChannel):
            return (factor.Nx,)
        if isinstance(factor, ConvChannel):
            # real=False: planar complex vectors (2,) + shape (channels.py ConvChannel._vshape)
            return tuple(factor._vshape)
        if isinstance(factor, GradientChannel):
            return (factor.d,) + tuple(factor.shape)
        if isinstance(factor, DFTChannel):
            # real=True: z real -> x planar complex; real=False: z is already planar (2,) + shape, x has its shape
            return (2,) + tuple(in_shape) if factor.real else tuple(in_shape)
        if isinstance(factor, ReshapeChannel):
            return tuple(factor.next_shape)
        return in_shape

    def init_shapes(self):
        "base_model.py:96-109 (shapes from the factors' geometry rather than a dry-run sample)"
        shapes = {}
        for node in self.forward_ordering:
            if isinstance(node, Factor) and node.n_next:
                preds = list(self.dag.predecessors(node))
                shape = self._out_shape(node, shapes[preds[0].id] if preds else None)
                for succ in self.dag.successors(node):
                    shapes[succ.id] = shape
                    self.dag.nodes[succ]["shape"] = shape
        return {v.id: shapes[v.id] for v in self.variables}

    def get_shapes(self):
        return {v.id: self.dag.nodes[v].get("shape") for v in self.variables}

    def init_second_moments(self):
        "base_model.py:111-124"
        for node in self.forward_ordering:
            if isinstance(node, Factor) and node.n_next:
                taus = [self.dag.nodes[p]["tau"] for p in self.dag.predecessors(node)]
                tau = node.second_moment(*taus)
                succs = list(self.dag.successors(node))
                for k, succ in enumerate(succs):              # a K-output factor returns one value per output (base_model.py:120-123)
                    self.dag.nodes[succ]["tau"] = tau[k] if node.n_next > 1 else tau

    def get_second_moments(self):
        return {v.id: self.dag.nodes[v].get("tau") for v in self.variables}


def _get_kwargs(target, kwargs):
    n = len(target) + 1
    return {k[n:]: v for k, v in kwargs.items() if k.startswith(target)}


def glm_generative(N, alpha, ensemble_type, prior_type, output_type, dtype=np.float64, ctx=None, **kwargs):
    "tramp/models/generalized_linear_model.py:17-36"
    M = int(alpha * N)
    F = get_ensemble(ensemble_type, M=M, N=N, **_get_kwargs("ensemble", kwargs)).generate()
    # phase retrieval: complex signal as a planar (2, N) array behind a complex matrix (generalized_linear_model.py:26-28)
    size = (2, N) if output_type == "modulus" else N
    prior = get_prior(size=size, prior_type=prior_type, dtype=dtype, ctx=ctx, **_get_kwargs("prior", kwargs))
    linear = get_channel("complex_linear" if output_type == "modulus" else "linear", W=F, name="F", dtype=dtype, ctx=ctx,
                         keep_W=True)
    okw = _get_kwargs("output", kwargs)
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


def _gradient_block(x_shape, prior_var, grad_prior, kw):
    """tramp/models/total_variation_model.py:12-40: a signal x with a Gaussian prior whose gradient field x' has a second
    parent, the prior ``grad_prior`` (variable "x'" with two parents), flattened for the dense channel that follows."""
    from .channels import GradientChannel, ReshapeChannel
    from .priors import GaussianPrior
    from .variables import SIMOVariable, MILeafVariable
    N = int(np.prod(x_shape))
    return (GaussianPrior(size=x_shape, var=prior_var, **kw) @ SIMOVariable(id="x", n_next=2) @ (
        (GradientChannel(shape=x_shape, **kw) + grad_prior) @ MILeafVariable(id="x'", n_prev=2))
    ) @ ReshapeChannel(prev_shape=x_shape, next_shape=N, **kw)


def _sparse_gradient_block(x_shape, prior_var, grad_rho, kw):
    "total_variation_model.py:12-25: Gauss-Bernoulli prior on the gradient"
    from .priors import GaussBernoulliPrior
    return _gradient_block(x_shape, prior_var, GaussBernoulliPrior(size=(len(x_shape),) + x_shape, rho=grad_rho, **kw), kw)


def _tv_block(x_shape, prior_var, grad_scale, kw):
    """total_variation_model.py:28-40: the group penalty MAP_L21NormPrior(axis=0) on the gradient.  (The reference passes
    ``scale=grad_scale`` to a constructor whose argument is ``gamma`` - map_L21_norm_prior.py:41 - and raises; this is the
    model its docstrings describe.)"""
    from .priors import MAP_L21NormPrior
    return _gradient_block(x_shape, prior_var, MAP_L21NormPrior(size=(len(x_shape),) + x_shape, gamma=grad_scale, axis=0, **kw), kw)


def _regression_block(A, y, noise_var, kw):
    "total_variation_model.py:43-48"
    lin = A if isinstance(A, LinearChannel) else LinearChannel(A, name="A", **kw)
    return DAG([lin, SISOVariable(id="z"), GaussianLikelihood(y=y, var=noise_var, **kw)])


def _classification_block(A, y, noise_var, kw):
    "total_variation_model.py:51-56: noisy sign observations, the noise as a hidden GaussianChannel"
    from .channels import GaussianChannel
    from .likelihoods import SgnLikelihood
    lin = A if isinstance(A, LinearChannel) else LinearChannel(A, name="A", **kw)
    return DAG([lin, SISOVariable(id="z"), GaussianChannel(var=noise_var, **kw), SISOVariable(id="a"), SgnLikelihood(y=y, **kw)])


def sparse_gradient_regression(A, y, x_shape, grad_rho, noise_var, prior_var=1.0, dtype=np.float64, ctx=None):
    """tramp/models/total_variation_model.py:59-63: a signal with a Gauss-Bernoulli prior on its gradient, observed
    through the dense matrix A with Gaussian noise.  y: [M] or [instances, M]."""
    kw = dict(dtype=dtype, ctx=ctx)
    return (_sparse_gradient_block(tuple(x_shape), prior_var, grad_rho, kw) @ SISOVariable(id="r")
            @ _regression_block(A, y, noise_var, kw)).to_model()


def sparse_gradient_classification(A, y, x_shape, grad_rho, noise_var, prior_var=1.0, dtype=np.float64, ctx=None):
    "total_variation_model.py:66-70: the same signal model under noisy sign observations y = sgn(A x + noise)"
    kw = dict(dtype=dtype, ctx=ctx)
    return (_sparse_gradient_block(tuple(x_shape), prior_var, grad_rho, kw) @ SISOVariable(id="r")
            @ _classification_block(A, y, noise_var, kw)).to_model()


def tv_regression(A, y, x_shape, grad_scale, noise_var, prior_var=1.0, dtype=np.float64, ctx=None):
    "total_variation_model.py:73-77: total-variation (group-L21) penalty on the gradient, Gaussian observations"
    kw = dict(dtype=dtype, ctx=ctx)
    return (_tv_block(tuple(x_shape), prior_var, grad_scale, kw) @ SISOVariable(id="r")
            @ _regression_block(A, y, noise_var, kw)).to_model()


def tv_classification(A, y, x_shape, grad_scale, noise_var, prior_var=1.0, dtype=np.float64, ctx=None):
    "total_variation_model.py:80-84: total-variation penalty, noisy sign observations"
    kw = dict(dtype=dtype, ctx=ctx)
    return (_tv_block(tuple(x_shape), prior_var, grad_scale, kw) @ SISOVariable(id="r")
            @ _classification_block(A, y, noise_var, kw)).to_model()


class FactorModel(Model):
    """tramp/models/factor_model.py:5-13: a model from a factor DAG whose roots are all priors."""

    def __init__(self, factor_dag):
        if not isinstance(factor_dag, DAG):
            raise TypeError(f"factor_dag {factor_dag} is not a FactorDAG")
        if factor_dag.ins:
            raise ValueError(f"{len(factor_dag.ins)} root node(s) not a prior")
        self.factor_dag = factor_dag
        Model.__init__(self, factor_dag)


def committee(K, N, alpha, ensemble_type, priors, activation1, activation2, noise_var, dtype=np.float64, ctx=None):
    """tramp/models/committee_model.py:8-41: K experts prior_k -> x_k -> F -> z_k -> activation1 -> a_k sharing
    the sensing matrix F, summed by the K-input SumChannel, then activation2 / Gaussian noise / the leaf y.
    F is factorised once; the K LinearChannel nodes share the device factors."""
    from .channels import SumChannel
    if activation1 not in ["abs", "relu", "sgn"]:
        raise ValueError(f"activation1={activation1} must be abs, sgn or relu")
    if activation2 not in [None, "abs", "relu", "sgn"]:
        raise ValueError(f"activation2={activation2} must be None, abs, sgn or relu")
    if not isinstance(priors, list) or len(priors) != K:
        raise ValueError(f"priors must be a list of length {K}")
    M = int(alpha * N)
    F = get_ensemble(ensemble_type, M=M, N=N).generate()
    kw = dict(dtype=dtype, ctx=ctx)
    shared = LinearChannel(F, name="F", keep_W=True, **kw)
    experts = None
    for k, prior in enumerate(priors):
        lin = shared if k == 0 else LinearChannel(F, name="F", keep_W=True, **kw,
                                                  factors=(shared._Uk, shared._sing, shared._Vt, shared.rank))
        expert = DAG([get_prior(size=N, **prior, **kw), SISOVariable(id=f"x_{k}"), lin, SISOVariable(id=f"z_{k}"),
                      get_channel(activation1, **kw), SISOVariable(id=f"a_{k}")])
        experts = expert if experts is None else experts + expert
    dag = experts @ SumChannel(n_prev=K, **kw)
    if activation2 in ["abs", "relu", "sgn"]:
        dag = dag @ SISOVariable(id="a") @ get_channel(activation2, **kw)
    if noise_var:
        dag = dag @ SISOVariable(id="n") @ GaussianChannel(var=noise_var, **kw)
    return (dag @ SILeafVariable(id="y")).to_model()


def sgn_committee(K, N, alpha, ensemble_type, p_pos, noise_var, **kw):
    "committee_model.py:44-54: binary priors, sgn experts, majority vote"
    p_pos = [p_pos] * K if isinstance(p_pos, float) else p_pos
    if len(p_pos) != K:
        raise ValueError(f"p_pos must be a list of length {K}")
    priors = [dict(prior_type="binary", p_pos=p) for p in p_pos]
    return committee(K, N, alpha, ensemble_type, priors, "sgn", "sgn", noise_var, **kw)


def soft_committee(K, N, alpha, ensemble_type, prior_mean, prior_var, noise_var, **kw):
    "committee_model.py:57-73: Gaussian priors, relu experts, linear output"
    prior_mean = [prior_mean] * K if isinstance(prior_mean, float) else prior_mean
    prior_var = [prior_var] * K if isinstance(prior_var, float) else prior_var
    if len(prior_mean) != K or len(prior_var) != K:
        raise ValueError(f"prior_mean and prior_var must be lists of length {K}")
    priors = [dict(prior_type="gaussian", mean=m, var=v) for m, v in zip(prior_mean, prior_var)]
    return committee(K, N, alpha, ensemble_type, priors, "relu", None, noise_var, **kw)


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


__all__ = ["DAG", "Model", "MultiLayerModel", "glm_generative", "glm_state_evolution", "sparse_gradient_regression", "committee", "soft_committee", "sgn_committee",
           "channel2likelihood", "PiecewiseLinearChannel"]
