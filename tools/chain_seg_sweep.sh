for seg in 3 4 5 6 8 10 16; do ALAN_CHAIN_SEG=$seg python - <<'PY'
import os,sys,time
sys.path.insert(0,'.')
import torch as t, bench, alan_amd as alan
prob=bench.build_timeseries_problem("cuda"); s=bench.draw(prob,30)
for _ in range(3): v=s.elbo_nograd(alan.no_checkpoint, graph=True)
t.cuda.synchronize(); t0=time.perf_counter()
for _ in range(200): v=s.elbo_nograd(alan.no_checkpoint, graph=True)
t.cuda.synchronize(); dt=(time.perf_counter()-t0)/200
print("seg", os.environ["ALAN_CHAIN_SEG"], f"{dt*1e6:.1f} us/eval", float(v))
PY
done
