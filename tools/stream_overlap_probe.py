#!/usr/bin/env python
"""Tooling: throughput of the same bins split into S independently stepping sub-batches on S HIP
streams, eager vs replayed as one hipGraph (two ping-pong steps per graph)."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import irbpp_amd  # noqa
from bench import make_workload
from irbpp_amd.vec_env import GpuPackingEnv

shapes, seqs, kw = make_workload(sys.argv[1] if len(sys.argv) > 1 else "blockout")
bins, K = 4096, 100
for nsplit in (1, 2, 4, 8):
    per = bins // nsplit
    envs = [GpuPackingEnv(shapes, seqs, per, device="cuda:0", global_offset=i * per, global_bins=bins, **kw)
            for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    obs, nxt, act = [], [], []
    for e, s in zip(envs, streams):
        with torch.cuda.stream(s):
            o = e.reset()
            obs.append(o); nxt.append(torch.empty_like(o))
            act.append(e.policy_minz(o))
            e.set_auto_policy(act[-1])                # fused policy, as in bench.py

    def step_all():
        for i, (e, s) in enumerate(zip(envs, streams)):
            with torch.cuda.stream(s):
                e.step(act[i], obs_out=nxt[i])
                obs[i], nxt[i] = nxt[i], obs[i]

    for _ in range(100):
        step_all()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K):
        step_all()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    eager = bins * K / dt / 1e6
    # hipGraph: fork the sub-batch streams from the capture stream, two steps (A->B, B->A), join
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=cap):
            for s in streams:
                s.wait_stream(cap)
            step_all(); step_all()
            for s in streams:
                cap.wait_stream(s)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K // 2):
        g.replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(nsplit, "streams: eager %.2f M steps/s, graph %.2f M steps/s (%.3f ms/step)" % (eager, bins * K / dt / 1e6, dt / K * 1e3))
    for e in envs:
        e.check_device_error(); e.close()
