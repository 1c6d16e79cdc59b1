# same-box A/B of two versions of ns2d_fast4_impl.h on the tall / wide grids:
#   scripts/ab_f4.sh <other header>      (works in a copy under /tmp; the other header must travel with the tree, e.g.
#   git show <rev>:beacon_amd/csrc/ns2d_fast4_impl.h > scripts/_old_fast4_impl.h before the gpurun call -- there is no .git on the box)
set -e
old=$(realpath "$1"); shift
rm -rf /tmp/work && cp -r . /tmp/work && cd /tmp/work
export PYTHONPATH=.
cp beacon_amd/csrc/ns2d_fast4_impl.h /tmp/new_fast4.h
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp "$old" beacon_amd/csrc/ns2d_fast4_impl.h; else cp /tmp/new_fast4.h beacon_amd/csrc/ns2d_fast4_impl.h; fi
    echo "== $v"
    for wl in "mixing 100x200" "rayleigh 50x150" "mixing 200x100" "rayleigh 300x50" "rayleigh 110x64 f64"; do
      timeout -k 10 300 python3 scripts/tall_base.py 256 "$wl" 2>&1 | grep -v amdgpu.ids
    done
  done
done
