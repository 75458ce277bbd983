#!/bin/bash
# GPU box: the one-launch-per-paint form of config 2 (what a drop-in caller painting buffer by buffer gets: ZH_BENCH_IN_ORDER=1 =
# one kernel node per step, no ZH_CAPTURE_COALESCE) under the variants VERDICT r5 item 3 names -- fewer, fatter workgroups
# (osc_fc = frames per wave), the per-voice constants computed per block instead of loaded from the table (ZH_BENCH_NO_TABLE),
# store flavours -- at the driver's 20 steps and at 1,000, alternating; then the kernel's own duration by rocprofv3 and the
# store-only floor in the same geometry.  -> gpurun_out/r06_percall/ab_percall.txt
set -u
O=gpurun_out/r06_percall; mkdir -p $O
exec > $O/ab_percall.txt 2>&1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f us/step wall, %.2f us/launch HIP events, frac %.3f, value %.3g' % (d['ms_per_step']*1e3, r['launch_ms_hip_events']*1e3, r['frac'], d['value']))"; }
one() { env ZH_BENCH_IN_ORDER=1 "$@" python bench.py --steps $STEPS --warmup $WARM --no-cpu --no-config5 --no-parity 2>/dev/null | line; }
echo "# config 2 (4,096 PulseOsc voices x 1,024 frames), one kernel node per step; three alternating rounds per variant"
for round in 1 2 3; do
 for STEPS in 20 1000; do
  WARM=$([ $STEPS = 20 ] && echo 5 || echo 100)
  echo "## round $round, --steps $STEPS"
  echo -n "default (4 frames per wave, table, sc1 stores):        "; one A=1
  echo -n "osc_fc=8  (512 workgroups):                            "; one ZH_FORMS=osc_fc=8
  echo -n "osc_fc=16 (256 workgroups):                            "; one ZH_FORMS=osc_fc=16
  echo -n "osc_fc=2  (2,048 workgroups):                          "; one ZH_FORMS=osc_fc=2
  echo -n "constants computed per block, no table:                "; one ZH_BENCH_NO_TABLE=1
  echo -n "non-temporal stores:                                   "; one ZH_STORE_MODE=1
  echo -n "sc0 sc1 stores:                                        "; one ZH_STORE_MODE=3
 done
done
echo "# rocprofv3 --kernel-trace --stats, the same form at --steps 20 --warmup 5 (per-kernel average without the inter-launch gaps)"
cd /tmp && export TMPDIR=/tmp
for v in "A=1" "ZH_FORMS=osc_fc=8" "ZH_BENCH_NO_TABLE=1"; do
  d=/tmp/r06pc_$$; rm -rf $d
  env ZH_BENCH_IN_ORDER=1 $v rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-parity --repeats 30 > /dev/null 2>&1
  echo "## $v"; python3 -c "
import csv, glob, sys
for r in csv.DictReader(open(glob.glob('$d/*/*kernel_stats.csv')[0])):
    if 'k_osc_const4' in r['Name'] and int(r['Calls']) > 100:
        print('%-70s calls %5s  average %.0f ns  min %s  max %s' % (r['Name'][:70], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))"
done
cd $GRAFT_REPO_ROOT
echo "# the floor: a store-only kernel in the same geometry, one graph (tools/ubench/store_floor.hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/store_floor.hip -o /tmp/store_floor 2>/dev/null && /tmp/store_floor 4096 2>&1 | grep -E "sm=2 tpb=256|sm=0 tpb=256 fc=4|hipMemset|param loads|divide|no prologue"
