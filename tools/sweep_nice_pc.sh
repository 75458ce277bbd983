# GPU box: NiceInstrument kernel forms (k_nice_pc three-wave pipeline vs k_nice) over voice counts
for v in 4096 16384 32768 65536 131072; do for mx in 0 1000000; do
  r=$(ZH_FORMS=nice_pc_max=$mx python bench.py --workload nice --voices $v --steps 96 --warmup 48 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us parity', (d.get('parity') or {}).get('bitexact'))")
  echo "V=$v nice_pc_max=$mx: $r"
done; done
