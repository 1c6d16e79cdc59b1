"""A/B builds of libbeacon_hip.so for ONE gpurun call (diagnostic; nothing here ships).

build (here, no GPU):   python scripts/variants.py build tag1=-DFLAG[,-DFLAG2] tag2= ...      (an empty flag list = the tree as it is)
    compiles the translation unit(s) the flags touch (ns2d_fast.hip, ns2d_fast2.hip) once per variant, links them with the objects
    of the other sources (compiled once, cached in /tmp/bcn_variant_objs) into beacon_amd/_variants/<tag>.so
run (on the GPU box):   python scripts/variants.py run "<command>" [tag ...]
    for every variant: puts it in the library's place (with the tree's signature, so that nothing rebuilds), runs the command in a
    child process, prints its last line; the tree's own library is put back at the end."""
import os, shutil, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beacon_amd import build as B
VDIR = os.path.join(B.PKG, "_variants")
CACHE = "/tmp/bcn_variant_objs"
HOT = ("ns2d_fast.hip", "ns2d_fast_f64.hip", "ns2d_fast2.hip")


def compile_one(src, extra, out):
    cmd = [B.hipcc()] + B.FLAGS + B.FILE_FLAGS.get(os.path.basename(src), []) + extra + ["-I", B.INC, "-c", src, "-o", out]
    subprocess.check_call(cmd)


def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    os.makedirs(CACHE, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor
    cold = [s for s in B.sources() if os.path.basename(s) not in HOT]
    with ThreadPoolExecutor(max_workers=4) as ex:      # the other sources: once per content
        def cached(src):
            import hashlib
            h = hashlib.sha256(open(src, "rb").read() + b"".join(open(d, "rb").read() for d in sorted(B._deps()) if d.endswith((".h", ".inc")))).hexdigest()[:16]
            out = os.path.join(CACHE, "%s.%s.o" % (os.path.basename(src)[:-4], h))
            if not os.path.exists(out):
                compile_one(src, [], out)
            return out
        cold_objs = list(ex.map(cached, cold))
    jobs = []
    for spec in specs:
        tag, _, fl = spec.partition("=")
        extra = [f for f in fl.split(",") if f]
        for h in HOT:
            jobs.append((tag, extra, os.path.join(B.CSRC, h), os.path.join(CACHE, "%s.%s.o" % (h[:-4], tag))))
    with ThreadPoolExecutor(max_workers=int(os.environ.get("JOBS", "6"))) as ex:
        list(ex.map(lambda j: compile_one(j[2], j[1], j[3]), jobs))
    for spec in specs:
        tag = spec.partition("=")[0]
        objs = cold_objs + [os.path.join(CACHE, "%s.%s.o" % (h[:-4], tag)) for h in HOT]
        out = os.path.join(VDIR, tag + ".so")
        subprocess.check_call([B.hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", out] + objs)
        print("built", out, flush=True)


def run(command, tags):
    tags = tags or sorted(f[:-3] for f in os.listdir(VDIR) if f.endswith(".so"))
    keep = B.LIB + ".keep"
    if os.path.exists(B.LIB):
        shutil.copy2(B.LIB, keep)
    try:
        for tag in tags:
            shutil.copy2(os.path.join(VDIR, tag + ".so"), B.LIB)
            with open(B.LIB + ".sig", "w") as fh:
                fh.write(B.signature() + "\n")
            r = subprocess.run(command, shell=True, capture_output=True, text=True, env=dict(os.environ, BEACON_NO_BUILD="1", BEACON_JIT="0"))
            lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
            print("%-14s rc=%d  %s" % (tag, r.returncode, lines[-1] if lines else r.stderr.strip()[-300:]), flush=True)
    finally:
        if os.path.exists(keep):
            shutil.move(keep, B.LIB)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2], sys.argv[3:])
