cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  d=$R/gpurun_out/pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph > /dev/null 2>&1
done
python - <<'PY'
import glob, csv, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'][:40]
        if 'transit' in k: agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k,v in agg.items():
        for c,vals in v.items(): print(k, c, sum(vals)/len(vals), len(vals))
PY
python - <<'PY'
import torch,time
x=torch.empty(1024*150000,dtype=torch.float64,device='cuda')
for _ in range(3): x.zero_()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): x.zero_()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
print('zero_ 1.23GB', dt*1e6,'us', x.numel()*8/dt/1e12,'TB/s')
y=torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): y.copy_(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
print('copy 1.23GB', dt*1e6,'us', 2*x.numel()*8/dt/1e12,'TB/s')
PY
