set -u
export TMPDIR=/tmp

python tools/eco_time.py --iterations 10 --variant full 2>&1 | grep "global_avgpool_fc\|Average"

