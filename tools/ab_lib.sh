# GPU box: A/B two builds of the library on the same box.  usage: tools/ab_lib.sh <other.so> -- prints us/step for both
other=$1
run() { python bench.py "$@" --no-cpu --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step']*1e3,2))"; }
ab() { for i in 1 2; do a=$(run "$@"); b=$(ZANG_HIP_LIB=$other run "$@"); echo "$* : default $a us | other $b us"; done; }
ab --steps 800 --warmup 80
ab --voices 65536 --steps 100 --warmup 10
ab --voices 524288 --steps 40 --warmup 4
ab --voices 1048576 --steps 40 --warmup 4
ab --workload nice --voices 131072 --steps 96 --warmup 48
ab --workload nice --voices 4096 --steps 96 --warmup 48
ab --workload nice_mix --voices 131072 --steps 96 --warmup 48
ab --workload nice_mix --voices 1048576 --steps 48 --warmup 48
