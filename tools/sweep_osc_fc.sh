# GPU box: frames per lane of the chunked oscillator kernel over voice counts, 3 repeats each (min reported)
for v in 16384 65536 262144 524288 1048576; do
  st=100; [ $v -ge 262144 ] && st=40
  for fc in 4 8 16 32 64; do
    best=999999
    for r in 1 2 3; do
      t=$(ZH_FORMS=osc_fc=$fc python bench.py --voices $v --steps $st --warmup 10 --no-cpu --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['launch_ms_hip_events']*1e3)")
      best=$(python -c "print(min($best, $t))")
    done
    echo "V=$v fc=$fc best_us=$best"
  done
done
