cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r02i; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload noise_filter_fused --no-cpu --no-parity --no-config5 --steps 192 --repeats 0 > $out/bench.json 2>/dev/null
cp $out/kt/*/*kernel_stats.csv $out/nf_fused_kernel_stats.csv; head -8 $out/nf_fused_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $out/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload noise_filter_fused --no-cpu --no-parity --no-config5 --steps 48 --warmup 0 --repeats 0 --eager > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r02i'
for f in glob.glob(out+'/pmc1/*/*counter_collection.csv'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k,d in acc.items():
        print(k, {c: round(sum(v)/len(v),1) for c,v in d.items()}, 'n=',len(next(iter(d.values()))))
PY
