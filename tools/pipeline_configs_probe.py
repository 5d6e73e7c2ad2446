#!/usr/bin/env python3
"""Pipelined throughput of the other workloads: sample() + elbo with fresh particles (SamplingPipeline) by lanes, and
bus_breakdown / timeseries evaluations at K=30 one after another against four lanes.   python3 tools/pipeline_configs_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, alan_amd as alan, bench

p = bench.build_problem("cuda")
for lanes in (1, 2, 3, 4, 6):
    sp = alan.SamplingPipeline(p, 30, alan.no_checkpoint, lanes=lanes, results=4096)
    sp.run(64); t.cuda.synchronize(); t0 = time.perf_counter(); v = sp.run(3000); t.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3000
    print(f"sample+elbo pipelined, {lanes} lane(s): {dt * 1e6:.2f} us per iteration ({1 / dt:.0f} it/s), {len(set(v.tolist()))} distinct ELBOs of 3000", flush=True)
    sp.close()
for name, builder in (("bus_breakdown", bench.build_bus_problem), ("timeseries T=1000", bench.build_timeseries_problem)):
    s = bench.draw(builder("cuda"), 30)
    ref = float(s.elbo_nograd(alan.no_checkpoint, graph=False))
    for lanes in (1, 4):
        pipe = s.pipeline(alan.no_checkpoint, lanes=lanes, results=4096)
        pipe.run(64); t.cuda.synchronize(); t0 = time.perf_counter(); v = pipe.run(2000); t.cuda.synchronize()
        assert float((v - ref).abs().max()) <= 2e-6 * abs(ref)
        print(f"{name} K=30, {lanes} lane(s): {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per evaluation", flush=True)
