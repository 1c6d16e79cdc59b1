# experiment: exchange-buffer layouts of the one-row-per-lane kernel on the bench workload (each variant rebuilds the library in a scratch copy)
set -e
cp -r . /tmp/work && cd /tmp/work
for x in 0 1 2 0 1 2; do timeout -k 10 300 python3 scripts/kstat.py f32 8 -DBCN_XPACK=$x; done
