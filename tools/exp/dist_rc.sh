for V in 32768 131072; do for cfg in "1073741824 0" "0 3" "0 6" "0 8"; do set -- $cfg
  echo -n "voices $V distortion_rows_min=$1 distortion_rc=$2: "; ZH_FORMS=distortion_rows_min=$1,distortion_rc=$2 ZH_BENCH_ONLY="Distortion" python tools/bench_modules.py $V 2>/dev/null | grep Distortion | awk '{printf "%s %s %s us %s TB/s;  ", $1, $2, $(NF-2), $NF}'; echo
done; done
