# same-box A/B of build flags: scripts/ab.sh <script.py and its args> -- <flags of variant 1> -- <flags of variant 2> ...
set -e
cp -r . /tmp/work && cd /tmp/work
cmd=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do cmd+=("$1"); shift; done
while [ $# -gt 0 ]; do
  shift; fl=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do fl+=("$1"); shift; done
  timeout -k 10 400 python3 "${cmd[@]}" "${fl[@]}"
done
