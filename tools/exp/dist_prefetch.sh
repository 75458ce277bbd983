for round in 1 2 3; do for V in 4096 32768 131072; do for rc in 1 0; do
  echo -n "voices $V overdrive prefetch $([ $rc = 1 ] && echo off || echo on): "; ZH_FORMS=distortion_rc=$rc ZH_BENCH_ONLY="Distortion overdrive" python tools/bench_modules.py $V 2>/dev/null | grep Distortion | awk '{printf "%s us %s TB/s", $(NF-2), $NF}'; echo
done; done; done
