"""Host-side render() of the single-env mirrors (optional): writes the same directory
layout and file names as the reference's render() methods -- render/<kind>/<stp>.png plus the text dumps --
from the state the HIP kernels produced.  Plotting is this module's own code (a field panel and a control
panel per frame); matplotlib is imported lazily and only here.

reference render(): rayleigh.py:278-341, mixing.py:267-359, burgers.py:169-213, shkadov.py:267-350,
sloshing.py:247-295."""
import os

import numpy as np


def _plt():
    try:
        import matplotlib
        if not os.environ.get("DISPLAY") and not os.environ.get("MPLBACKEND"):
            matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        return plt
    except ImportError as e:                      # pragma: no cover
        raise RuntimeError("render() needs matplotlib on the host") from e


def _dirs(env, *sub):
    """Create render/ and its sub-directories on the first frame (stp_plot == 0)."""
    env.path = "render"
    out = [os.path.join(env.path, s) for s in sub]
    if env.stp_plot == 0:
        os.makedirs(env.path, exist_ok=True)
        for d in out:
            os.makedirs(d, exist_ok=True)
    return out


def _bar(ax, x, y, w, h, value):
    from matplotlib.patches import Rectangle
    ax.add_patch(Rectangle((x, y), w, h, color="r" if value > 0.0 else "b", fill=True, lw=1))


def _finish(plt, fig, filename, show, dpi=100):
    fig.savefig(filename, dpi=dpi, bbox_inches="tight")
    if show:
        plt.pause(0.0001)
    plt.close(fig)


def _bare(ax, xlim, ylim):
    ax.set_xlim(xlim)
    ax.set_ylim(ylim)
    ax.set_xticks([])
    ax.set_yticks([])


def rayleigh(env, show=False, dump=True):
    plt = _plt()
    tdir, fdir, adir = _dirs(env, "temperature", "field", "action")
    fig = plt.figure(figsize=(5, 6))
    ax = fig.add_subplot(30, 1, (1, 29))
    ax.set_xticks([]); ax.set_yticks([])
    ax.imshow(np.rot90(env.T[1:-1, 1:-1]), cmap="RdBu_r", vmin=env.Tc, vmax=env.Th, extent=[0.0, env.L, 0.0, env.H])
    ax = fig.add_subplot(30, 1, (27, 30))
    _bare(ax, [0.0, env.n_sgts], [-env.C, env.C])
    for i, a in enumerate(env.a):
        _bar(ax, 0.5 + i - 0.12, 0.0, 0.24, 0.98 * a * env.C, a)
    _finish(plt, fig, os.path.join(tdir, "%d.png" % env.stp_plot), show)
    if dump:
        env.dump(os.path.join(fdir, "field_%d.dat" % env.stp_plot), os.path.join(adir, "a_%d.dat" % env.stp_plot),
                 os.path.join(env.path, "nu.dat"))
    env.stp_plot += 1


def mixing(env, show=False, dump=True):
    plt = _plt()
    cdir, fdir, adir = _dirs(env, "concentration", "field", "action")
    u_t = u_b = v_l = v_r = 0.0                                   # get_control, mixing.py:212-234
    a = int(env.a)
    if a == 0: u_b, u_t = env.u_max, -env.u_max
    if a == 1: u_b, u_t = -env.u_max, env.u_max
    if a == 2: v_r, v_l = env.u_max, -env.u_max
    if a == 3: v_r, v_l = -env.u_max, env.u_max
    fig = plt.figure(figsize=(5.5, 5.5))
    gs = fig.add_gridspec(15, 15)
    ax = fig.add_subplot(gs[1:-1, 1:-1])
    ax.set_xticks([]); ax.set_yticks([])
    ax.imshow(np.rot90(env.C[1:-1, 1:-1]), cmap="RdBu_r", vmin=0.0, vmax=env.C0, extent=[0.0, env.L, 0.0, env.H])
    for spec, horizontal, val in ((gs[0, 1:-1], True, u_t), (gs[-1, 1:-1], True, u_b),
                                  (gs[1:-1, 0], False, v_l), (gs[1:-1, -1], False, v_r)):
        ax = fig.add_subplot(spec)
        if horizontal:
            _bare(ax, [-env.u_max, env.u_max], [0.0, 0.1])
            _bar(ax, 0.0, 0.02, 0.98 * val, 0.06, val)
        else:
            _bare(ax, [0.0, 0.1], [-env.u_max, env.u_max])
            _bar(ax, 0.02, 0.0, 0.06, 0.98 * val, val)
    _finish(plt, fig, os.path.join(cdir, "%d.png" % env.stp_plot), show)
    if dump:
        env.dump(os.path.join(fdir, "field_%d.dat" % env.stp_plot), os.path.join(adir, "a_%d.dat" % env.stp_plot))
    env.stp_plot += 1


def _line_frame(plt, x, y, xlim, ylim, ref, figsize=(7, 3), top=(1, 17), bottom=(19, 20), vline=None):
    fig = plt.figure(figsize=figsize)
    ax = fig.add_subplot(20, 1, top)
    _bare(ax, xlim, ylim)
    if vline is not None:
        ax.axvline(x=vline, color="k", lw=1)
    ax.plot(x, np.full_like(x, ref), color="k", lw=1, linestyle="dashed")
    ax.plot(x, y)
    return fig, fig.add_subplot(20, 1, bottom)


def burgers(env, show=False, dump=True):
    plt = _plt()
    gdir, fdir = _dirs(env, "gif", "fields")
    x = np.arange(env.nx) * env.dx
    fig, ax = _line_frame(plt, x, env.u, [0.0, env.L], [env.u_target - 2 * env.sigma, env.u_target + 2 * env.sigma],
                          env.u_target, vline=env.ctrl_pos * env.dx)
    _bare(ax, [-env.amp, env.amp], [0.0, 0.2])
    _bar(ax, 0.0, 0.05, 0.98 * env.a[0] * env.amp, 0.1, env.a[0])
    _finish(plt, fig, os.path.join(gdir, "%d.png" % env.stp_plot), show)
    if dump:
        env.dump(os.path.join(fdir, "%d.dat" % env.stp_plot))
    env.stp_plot += 1


def shkadov(env, show=False, dump=True):
    plt = _plt()
    x = np.arange(env.nx) * env.dx
    if env.render_style == "dynamic":
        hdir, fdir, adir = _dirs(env, "height", "field", "action")
        fig, ax = _line_frame(plt, x, env.h, [0.0, env.L], [0.0, 2.0], 1.0, figsize=(10, 3), top=(1, 15), bottom=(17, 20))
        _bare(ax, [0.0, env.L], [-1.0, 1.0])
        ax.plot(x, np.zeros_like(x), color="k", lw=1, linestyle="dashed")
        for i, u in enumerate(env.u):
            _bar(ax, (env.jet_pos + i * env.jet_space - env.jet_hw) * env.dx, 0.0, (env.jet_hw + 1) * env.dx, u, u)
        _finish(plt, fig, os.path.join(hdir, "%d.png" % env.stp_plot), show, dpi=None)
        if dump:
            env.dump(os.path.join(fdir, "field_%d.dat" % env.stp_plot), os.path.join(adir, "jet_%d.dat" % env.stp_plot))
    else:                                                         # "static": one space-time map per episode
        _dirs(env)
        row = np.asarray(env.h).reshape(1, -1)
        env.h_store = row if env.stp_plot == 0 else np.vstack((env.h_store, row))
        if env.stp_plot == env.n_act - 1:
            fig, ax = plt.subplots(figsize=(10, 3))
            im = ax.imshow(env.h_store, vmin=0.6, vmax=1.4, aspect="auto", origin="lower", cmap="RdBu_r",
                           extent=[0, env.L, 0, env.stp_plot])
            fig.colorbar(im, ax=ax)
            _finish(plt, fig, os.path.join(env.path, "height.png"), show)
    env.stp_plot += 1


def sloshing(env, show=False, dump=True):
    plt = _plt()
    hdir, fdir, adir = _dirs(env, "height", "field", "action")
    x = np.arange(env.nx) * env.dx
    fig, ax = _line_frame(plt, x, env.h[1:env.nx + 1], [0.0, env.L], [0.25, 1.75], 1.0)
    _bare(ax, [-1.0, 1.0], [0.0, 0.2])
    _bar(ax, 0.0, 0.05, 0.98 * env.u[0], 0.1, env.u[0])
    _finish(plt, fig, os.path.join(hdir, "%d.png" % env.stp_plot), show)
    if dump:
        env.dump(os.path.join(fdir, "field_%d.dat" % env.stp_plot), os.path.join(adir, "jet_%d.dat" % env.stp_plot))
    env.stp_plot += 1


def _control_strip(fig, pos, value, xlim=(-1.0, 1.0)):
    ax = fig.add_subplot(*pos)
    _bare(ax, list(xlim), [0.0, 0.2])
    _bar(ax, 0.0, 0.05, 0.98 * value, 0.1, value)


def lorenz(env, show=False, dump=True):
    """lorenz.py:175-248: png/gif/<n>.png (trajectory so far + control), png/history.png at the end of the
    episode, png/lorenz.dat (columns t, x, y, z)."""
    plt = _plt()
    env.path = "png"
    if env.stp_plot == 0:
        os.makedirs(os.path.join(env.path, "gif"), exist_ok=True)
    hx, ht = np.asarray(env.hx), np.asarray(env.ht)
    if env.stp == env.n_act:
        fig, ax = plt.subplots(figsize=(8, 2))
        ax.plot(ht, hx[:, 0])
        ax.set_xlim([0.0, env.t_max]); ax.set_ylim([-20.0, 20.0]); ax.grid()
        fig.savefig(os.path.join(env.path, "history.png"), dpi=100)
        plt.close(fig)
    fig = plt.figure()
    ax = fig.add_subplot(15, 1, (1, 14), projection="3d")
    ax.set_axis_off()
    ax.set_xlim([-20.0, 20.0]); ax.set_ylim([-20.0, 20.0]); ax.set_zlim([0.0, 40.0])
    ax.plot(hx[:, 0], hx[:, 1], hx[:, 2], linewidth=1)
    _control_strip(fig, (15, 1, 15), float(env.actions[env.u]))
    _finish(plt, fig, os.path.join(env.path, "gif", "%d.png" % env.stp_plot), show)
    if dump:
        env.dump(os.path.join(env.path, "lorenz.dat"))
    env.stp_plot += 1


def vortex(env, show=False, dump=True):
    """vortex.py:211-264: png/gif/<n>.png (phase portrait of (ar, ai) + the two controls), png/vortex.dat at
    the last step of the episode (columns t, ar, ai, yr, yi, kmod, kphase)."""
    plt = _plt()
    env.path = "png"
    if env.stp_plot == 0:
        os.makedirs(os.path.join(env.path, "gif"), exist_ok=True)
    hx = np.asarray(env.hx)
    fig = plt.figure()
    ax = fig.add_subplot(30, 1, (1, 26))
    _bare(ax, [-0.04, 0.04], [-0.04, 0.04])
    ax.plot(hx[:, 0], hx[:, 1], linewidth=1)
    _control_strip(fig, (30, 1, (27, 28)), float(env.u[0]))
    _control_strip(fig, (30, 1, (29, 30)), float(env.u[1]))
    fig.savefig(os.path.join(env.path, "gif", "%d.png" % env.stp_plot), dpi=100)
    if show:
        plt.pause(0.0001)
    plt.close(fig)
    if dump and env.stp_plot == env.n_act - 1:
        env.dump(os.path.join(env.path, "vortex.dat"))
    env.stp_plot += 1
