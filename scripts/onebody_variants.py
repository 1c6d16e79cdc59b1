# diagnostic: the code-generation variants of round 5's table (profiles/r05_twobody_variants.log: the two-body float64 kernel of 110x64
# failed under 14 of 21 of them) on the ONE-body form of round 6: every variant's plugin is compiled on the box with its flags and
# compared with the generic kernel over 12 timesteps (plain launch and ticket scheduler).
import os, sys, warnings
sys.path.insert(0, os.getcwd())
warnings.simplefilter("ignore")
from beacon_amd import build as B, jit, vec as V
VARIANTS = [("as built", []), ("nodppcomb", ["-mllvm", "-amdgpu-dpp-combine=0"]), ("nopostsched", ["-mllvm", "-enable-post-misched=0"]),
            ("nomlicm", ["-mllvm", "-disable-machine-licm"]), ("O2", ["-O2"]), ("O1", ["-O1"]), ("nosdwa", ["-mllvm", "-amdgpu-sdwa-peephole=0"]),
            ("nomsink", ["-mllvm", "-disable-machine-sink"]), ("nopresched", ["-mllvm", "-enable-misched=0"]),
            ("noslotshare", ["-mllvm", "-no-stack-slot-sharing"]), ("nossc", ["-mllvm", "-disable-ssc"]),
            ("noagprspill", ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]), ("noshrinkwrap", ["-mllvm", "-enable-shrink-wrap=0"]),
            ("prealloc", ["-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs=1"]), ("nomcp", ["-mllvm", "-disable-copyprop"]),
            ("nobranchfold", ["-mllvm", "-disable-branch-fold"]), ("notaildup", ["-mllvm", "-disable-tail-duplicate"]),
            ("noblockplace", ["-mllvm", "-disable-block-placement"]), ("noearlytaildup", ["-mllvm", "-disable-early-taildup"]),
            ("nopostra", ["-mllvm", "-disable-postra-machine-licm", "-mllvm", "-disable-postra-machine-sink"]),
            ("nosgprvgpr", ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"])]
base = list(B.FLAGS)
bad = 0
for n, (tag, flags) in enumerate(VARIANTS):
    B.FLAGS[:] = base + flags                       # (the plugin's name carries a hash of the flags)
    jit._LOADED.clear()
    mk = lambda Bn: V.VecRayleigh(Bn, "cuda:0", "f64", None, L=2.2, H=1.28)
    try:
        e = mk(2); name = os.path.basename(e._plugin.path) if getattr(e, "_plugin", None) else None; e.close()
        ok, rep = jit.compare_with_generic(mk, 0, True, ndt=12, batch=3)
    except Exception as ex:      # noqa: BLE001
        ok, rep, name = False, "%s: %s" % (type(ex).__name__, str(ex)[:200]), None
    bad += not ok
    print("%-14s %-44s %s  %s" % (tag, " ".join(flags), "exact" if ok else "FAIL ", rep if not ok else rep.split(";")[0]), flush=True)
print("failing variants:", bad)
