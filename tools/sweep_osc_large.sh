#!/bin/bash
# GPU box: PulseOsc at many voices -- frames per lane (osc_fc) x store mode (ZH_STORE_MODE) x row padding x table / setup form, REPS runs
# each, one line per point: `roofline.frac` of bench.py.  One script for the nine sweeps of round 5 (profiles/r05/osc_large_voice_counts.txt
# holds their output; they differed only in these lists):
#   VOICES="65536 1048576" FCS="0 2 8 16" MODES="2 0 1"                              the first pass
#   VOICES="65536 262144 1048576" FCS="2 3 4 5 6" MODES="1 2" REPS=3                 around the gap
#   VOICES="49152 ... 786432" CFGS="3:1 4:2 3:2 4:1 2:1" REPS=3                      (fc:mode pairs instead of the product)
#   PADS="0 64 256 512 1024 4096"                                                    row padding in voices (bench.py --pad-voices)
#   SETUP=1                                                                          paints not flagged ZH_PAINT_PARAMS_UNCHANGED (no constants table)
#   STEPS=20 WARMUP=5 / STEPS=1000 WARMUP=100                                        a fixed step count (default: 6,553,600 / voices, at least 12)
# usage: [VOICES=..] [FCS=..] [MODES=..] [CFGS=..] [PADS=..] [REPS=n] [SETUP=1] [STEPS=n WARMUP=n] tools/sweep_osc_large.sh [out file]
out=${1:-gpurun_out/sweep_osc_large.txt}; mkdir -p "$(dirname "$out")"; : > "$out"
VOICES=${VOICES:-"65536 1048576"}; FCS=${FCS:-"0 2 8 16"}; MODES=${MODES:-"2 0 1"}; PADS=${PADS:-""}; REPS=${REPS:-1}
if [ -z "${CFGS:-}" ]; then CFGS=""; for fc in $FCS; do for sm in $MODES; do CFGS="$CFGS $fc:$sm"; done; done; fi
one() { v=$1; fc=$2; sm=$3; pad=$4
  steps=${STEPS:-$((6553600 / v))}; [ "$steps" -lt 12 ] && steps=12
  env ${SETUP:+ZH_BENCH_NO_TABLE=1} ZH_STORE_MODE=$sm ZH_FORMS=osc_fc=$fc python bench.py --voices $v ${pad:+--pad-voices $pad} --steps $steps --warmup ${WARMUP:-4} --no-cpu --no-config5 --no-parity 2>/dev/null \
    | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['roofline']['frac'], end=' ')"; }
for v in $VOICES; do for pad in ${PADS:-""}; do for cfg in $CFGS; do
  fc=${cfg%%:*}; sm=${cfg##*:}
  echo -n "voices $v${pad:+ pad $pad}${SETUP:+ setup form} osc_fc=$fc store_mode=$sm: " >> "$out"
  for rep in $(seq $REPS); do one $v $fc $sm "$pad" >> "$out"; done; echo >> "$out"
done; done; done
cat "$out"
