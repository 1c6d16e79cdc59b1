import os, sys, subprocess, numpy as np
code = (
    "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
    "from beacon_amd import vec as V\n"
    "from beacon_amd.envs import packaged_init\n"
    "env = V.VecRayleigh(24, 'cuda:0', 'f64', packaged_init('rayleigh'))\n"
    "assert env.set_variant(1) == 1\n"
    "env.reset()\n"
    "a = np.random.default_rng(5).uniform(-1, 1, (2, 24, 10))\n"
    "for k in range(2): obs, rwd, *_ = env.step(a[k])\n"
    "env.check_status()\n"
    "print(env.kernel_name)\n"
    "np.save(sys.argv[1], np.concatenate([obs.cpu().numpy().ravel(), rwd.cpu().numpy(),"
    " env.get_state().cpu().numpy().ravel(), env.sweeps.cpu().numpy().ravel().astype(float)]))\n"
) % os.getcwd()
outs = []
for tag, extra in (("split", dict(BCN_SCHED="1", BCN_LPT_MIN_BATCH="2")), ("single", dict(BCN_SCHED="0")),
                   ("ticket", dict(BCN_SCHED="2", BCN_SCHED_GRID="5")), ("single2", dict(BCN_SCHED="0"))):
    path = "/tmp/bcn_lpt_%s.npy" % tag
    r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
    print(tag, r.returncode, r.stdout.strip(), r.stderr[-500:])
    outs.append(np.load(path))
n_obs, B = 24 * 192, 24
for name, o in zip(("split", "single", "ticket", "single2"), outs):
    d = np.abs(o - outs[1])
    print(name, "obs/rwd maxdiff", d[:n_obs + B].max(), "sweeps diff", d[-B * 200:].max(), "state", d[n_obs + B:-B * 200].max())
