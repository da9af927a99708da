"""Quick plots of experiment tables and estimates - the helpers scripts written for the reference call (tramp/experiments/plots.py:
plot_function, plot_compare, plot_compare_complex, qplot).  Host matplotlib, imported on first use; nothing here touches the device.

qplot lays a long-format DataFrame out as a grid of panels: `row` / `column` pick the panel, `color` / `marker` / `linestyle` the
look of a curve inside it; every combination of the values of those fields that occurs in the table is one curve.
"""
import itertools

import numpy as np

LINESTYLES = ["-", "--", "-.", ":"]
MARKERS = [".", "x", "+", "o", "v", "^", "<", ">", "s", "D"]


def _plt():
    import matplotlib.pyplot as plt
    return plt


def _as_complex(x):
    x = np.asarray(x)
    return x if np.iscomplexobj(x) else x[0] + 1j * x[1]          # planar (2, n) -> complex (tramp/utils/misc.py:13-27)


def plot_function(f, xlim=(-3, 3)):
    plt = _plt()
    x = np.linspace(xlim[0], xlim[1], 100)
    plt.plot(x, f(x))
    plt.title(f"{f.__name__}")


def plot_compare(x_true, x_pred):
    "the first hundred components of the truth and of the estimate on one axis"
    plt = _plt()
    fig, ax = plt.subplots(1, 1, figsize=(4, 4))
    for label, x in (("x true", x_true), ("x pred", x_pred)):
        x = np.ravel(np.asarray(x))[:100]
        ax.plot(np.arange(x.size), x, label=label)
    ax.legend()
    fig.tight_layout()
    return fig


def plot_compare_complex(x_true, x_pred):
    "phase against phase and modulus against modulus of a complex (or planar) truth and estimate"
    plt = _plt()
    zt, zp = _as_complex(x_true), _as_complex(x_pred)
    fig, axs = plt.subplots(1, 2, sharey=False, sharex=False)
    axs[0].plot(np.angle(zt), np.angle(zp), ".")
    axs[0].set(xlabel="phase x true", ylabel="phase x pred")
    axs[1].plot(np.abs(zt), np.abs(zp), ".")
    axs[1].set(xlabel="modulus x true", ylabel="modulus x pred")
    fig.tight_layout()
    return fig


def _renamed(text, rename):
    for old, new in (rename or {}).items():
        text = text.replace(old, new)
    return text


def qplot(data, x, y, color=None, column=None, row=None, marker=None, linestyle=None, xlog=False, ylog=False, xlim=None,
          ylim=None, y_markers=None, sharex=True, sharey=True, figsize=4, y_legend=False, rename=None, font_size=12,
          usetex=False):
    """y against x from a long-format table, faceted by `row` / `column`, styled by `color` / `marker` / `linestyle`.
    y may be a list of columns, then y_markers gives one line style or marker per column (and marker / linestyle are taken)."""
    plt = _plt()
    ys = list(y) if isinstance(y, list) else [y]
    if isinstance(y, list):
        if not isinstance(y_markers, list) or len(y_markers) != len(ys):
            raise ValueError("y_markers must be a list of same length as y")
        if marker is not None or linestyle is not None:
            raise ValueError("cannot use marker / linestyle in y_multiple mode")
        for m in y_markers:
            if m not in LINESTYLES and m not in MARKERS:
                raise ValueError(f"unkown marker {m}")
    plt.rc("text", usetex=usetex)
    plt.rc("font", family="serif", size=font_size)
    fields = dict(row=row, column=column, color=color, marker=marker, linestyle=linestyle)
    levels = {f: sorted(data[f].unique()) for f in fields.values() if f}
    colors = plt.rcParams["axes.prop_cycle"].by_key()["color"]
    nrows = len(levels[row]) if row else 1
    ncols = len(levels[column]) if column else 1
    if isinstance(figsize, (int, float)):
        figsize = (figsize * ncols, figsize * nrows)
    fig, axs = plt.subplots(nrows, ncols, squeeze=False, figsize=figsize, sharex=sharex, sharey=sharey)
    any_label = False
    names = list(levels)
    for combo in itertools.product(*[levels[f] for f in names]):
        chosen = dict(zip(names, combo))
        sub = data
        for f, val in chosen.items():
            sub = sub[sub[f] == val]
        if sub.empty:
            continue
        ax = axs[levels[row].index(chosen[row]) if row else 0, levels[column].index(chosen[column]) if column else 0]
        style = {}
        if color:
            style["color"] = colors[levels[color].index(chosen[color]) % len(colors)]
        if marker:
            style["marker"] = MARKERS[levels[marker].index(chosen[marker]) % len(MARKERS)]
        if linestyle:
            style["linestyle"] = LINESTYLES[levels[linestyle].index(chosen[linestyle]) % len(LINESTYLES)]
        label = " ".join(f"{f}={chosen[f]}" for f in (color, marker, linestyle) if f)
        title = " ".join(f"{f}={chosen[f]}" for f in (row, column) if f)
        for i, col in enumerate(ys):
            st = dict(style)
            if isinstance(y, list):
                m = y_markers[i]
                st.update(dict(linestyle=m, marker="") if m in LINESTYLES else dict(linestyle="", marker=m))
                lab = f"{label} {col}" if y_legend else (label if i == 0 else "")
            else:
                st.setdefault("linestyle", "-" if not marker else "")
                st.setdefault("marker", "")
                lab = f"{label} {col}" if y_legend else label
            lab = _renamed(lab.strip(), rename)
            any_label = any_label or bool(lab)
            ax.plot(sub[x], sub[col], label=lab or None, **st)
        ax.set(title=_renamed(title, rename))
    for ax in axs.ravel():
        if xlog:
            ax.set_xscale("log")
        if ylog:
            ax.set_yscale("log")
        if xlim:
            ax.set_xlim(xlim)
        if ylim:
            ax.set_ylim(ylim)
        ax.set(xlabel=_renamed(x, rename), ylabel="" if y_legend else _renamed(", ".join(ys), rename))
        if any_label or y_legend:
            ax.legend()
    fig.tight_layout()
    return fig
