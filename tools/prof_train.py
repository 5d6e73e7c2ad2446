import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench
from alan_amd.training import GraphedStep
mode = sys.argv[1] if len(sys.argv) > 1 else "vi"
prob = bench.build_problem("cuda")
opt = (t.optim.Adam(prob.Q.parameters(), lr=1e-2, capturable=True, maximize=True) if mode == "rws"
       else t.optim.Adam(prob.parameters(), lr=1e-2, capturable=True))
step = GraphedStep(prob, 30, opt, method=mode)
for _ in range(50): step()
t.cuda.synchronize()
