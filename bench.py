#!/usr/bin/env python
"""bench.py -- placement-steps/s of the batched packing environment on N MI355X.

One "step" = one batched transition of every bin on this rank: the transition kernel applies the
actions (placement, reward, termination, auto-reset), the trace, polygon and emit kernels produce the next
observation and, fused into the emit kernel, the scripted MINZ policy's action on it.  Inputs
(shape tables, trajectories, heightmaps, observations) are resident in HBM before the timed
region; nothing crosses PCIe inside it.

Workload at N=1: the configuration BASELINE.json's north_star quotes its target on -- BlockOut online (bufferSize=1,
configs[1]'s data) with 8192 bins on one GPU, resolutionA=0.02, resolutionH=0.01, R=4, S=500 (synthetic polycubes,
SURVEY.md 8d); configs[1]'s own 4096 bins are measured beside it (`extra.bins4096_one_gpu`).
Multi-GPU: bins are sharded, no data-path collective; one RCCL all-reduce of the episode
totals after the timed region.  Default: "weak" scaling, 8192 bins per GPU; `--config cfg4|cfg5` are north_star's
sharded configs (8192 k=10 bins / 16384 fine-heightmap bins divided over the GPUs: "strong").
The timed region is repeated in blocks of --steps steps until it adds up to --min-seconds, so the figure does
not depend on how few steps the caller asked for.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import irbpp_amd  # noqa: E402,F401
from irbpp_amd import synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
S = 500


def make_workload(name):
    if name == "blockout":
        shapes = synthetic.blockout_shapes(n_shapes=64, n_rot=4, seed=0)
        seqs = synthetic.make_sequences(shapes.n_shapes, 10000, 160, seed=123)
        kw = dict(resolutionA=0.02, resolutionH=0.01)
    elif name == "blockout_r8":         # cfg 2 with the README command's eight rotations (README.md:100)
        shapes = synthetic.blockout_shapes(n_shapes=64, n_rot=8, seed=0)
        seqs = synthetic.make_sequences(shapes.n_shapes, 10000, 160, seed=123)
        kw = dict(resolutionA=0.02, resolutionH=0.01)
    elif name == "blockout_k10":        # cfg 4: buffered packing, order action scripted = slot 0
        shapes = synthetic.blockout_shapes(n_shapes=64, n_rot=4, seed=0)
        seqs = synthetic.make_sequences(shapes.n_shapes, 10000, 160, seed=123)
        kw = dict(resolutionA=0.02, resolutionH=0.01, bufferSize=10)
    elif name == "general":
        shapes = synthetic.general_shapes(n_shapes=256, n_rot=8, seed=1)
        seqs = synthetic.make_sequences(shapes.n_shapes, 10000, 100, seed=123)
        kw = dict(resolutionA=0.02, resolutionH=0.01)
    elif name == "abc_fine":
        shapes = synthetic.general_shapes(n_shapes=256, n_rot=8, fmin=8, fmax=40, res_h=0.005, seed=1)
        seqs = synthetic.make_sequences(shapes.n_shapes, 10000, 100, seed=123)
        kw = dict(resolutionA=0.02, resolutionH=0.005)
    elif name == "cube":
        shapes = synthetic.cube_shapes()
        seqs = synthetic.make_sequences(shapes.n_shapes, 10000, 100, seed=123)
        kw = dict(resolutionA=0.02, resolutionH=0.01)
    else:
        raise SystemExit(f"unknown workload {name}")
    return shapes, seqs, kw


def algorithmic_bytes_per_step(shapes, hc, k=1):
    """compulsory HBM bytes of one placement-step of one bin.  The master
    heightmap is float64 in HBM (8*Hc in and out); footprint tables are priced at the survey's
    external layout (4-byte height + 1-byte mask per cell); the observation is float32."""
    f = np.array([[t[0].size for t in per_rot] for per_rot in shapes.tables], dtype=np.float64)   # [n, R]
    sum_r = f.sum(axis=1).mean()          # all rotations of the next item (overlap test)
    f_star = f.mean()                     # the placed rotation (heightmap update)
    reads = 8 * hc + 5 * sum_r + 5 * f_star + 4 + 4 * k
    writes = 8 * hc + 4 * (5 * S + 9 + hc) + 5
    if k > 1:
        writes += 4 * (k + hc)
    return float(reads + writes)


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a restatement of the reference's numpy path) on the host cores
# ---------------------------------------------------------------------------------------------
def usable_cores():
    """Cores this process may really use: affinity mask and cgroup quota, not just cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _cpu_worker(args):
    """One OS process = one bin, like wrapper/shmem_vec_env.py:120-157.  Runs for a fixed wall time."""
    name, rank, seconds, impl = args
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    sys.path.insert(0, ROOT)
    shapes, seqs, kw = make_workload(name)
    if impl == "c":
        from oracle.c_oracle import COracleVecEnv as Env
    else:
        from oracle.packing import OracleVecEnv as Env
    env = Env(1, shapes, seqs, global_offset=rank, global_num=1 << 20, **kw)
    obs = env.reset()
    buffered = kw.get("bufferSize", 1) > 1

    def minz(o):
        c = o[:5 * S].reshape(S, 5)
        v = c[:, 4] == 1
        return int(np.argmin(np.where(v, c[:, 3], np.inf))) if v.any() else 0

    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        if buffered:                    # one hierarchical placement = get_action_candidates + step
            obs = env.get_action_candidates([0])
        obs, _, _, _ = env.step([minz(o) for o in obs])
        n += 1
    return n, time.perf_counter() - t0


def _shmem_worker(conn, name, rank, buf):
    """Worker of the lockstep runner below: the command loop of wrapper/shmem_vec_env.py:120-157 --
    receive an action over the pipe, step, auto-reset on done, write the observation into the shared
    buffer, send (reward, done) back."""
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    sys.path.insert(0, ROOT)
    from oracle.c_oracle import COracleVecEnv
    shapes, seqs, kw = make_workload(name)
    env = COracleVecEnv(1, shapes, seqs, global_offset=rank, global_num=1 << 20, **kw)
    dst = np.frombuffer(buf, dtype=np.float32)
    dst[:] = env.reset()[0]
    conn.send(None)
    while True:
        act = conn.recv()
        if act is None:
            break
        obs, rew, done, _ = env.step([act])
        dst[:] = obs[0]
        conn.send((float(rew[0]), bool(done[0])))
    conn.close()


def cpu_lockstep_runner(name, cores, seconds):
    """The reference's own process structure (envs.py:67-99 -> ShmemVecEnv): one worker process per
    env, actions out and (reward, done) back over pipes, observations through shared memory, and ONE
    parent that waits for every env before it chooses the next actions.  Online workloads only."""
    ctx = mp.get_context("fork")
    shapes, _, kw = make_workload(name)
    hx = int(round(0.32 / kw["resolutionH"]))
    obs_len = 5 * S + 9 + hx * hx
    bufs = [ctx.RawArray("f", obs_len) for _ in range(cores)]
    pipes, procs = [], []
    for r in range(cores):
        parent, child = ctx.Pipe()
        p = ctx.Process(target=_shmem_worker, args=(child, name, r, bufs[r]), daemon=True)
        p.start()
        child.close()
        pipes.append(parent)
        procs.append(p)
    for c in pipes:
        c.recv()
    views = [np.frombuffer(b, dtype=np.float32) for b in bufs]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for c, o in zip(pipes, views):                    # step_async (shmem_vec_env.py:70-74)
            cand = o[:5 * S].reshape(S, 5)
            v = cand[:, 4] == 1
            c.send(int(np.argmin(np.where(v, cand[:, 3], np.inf))) if v.any() else 0)
        for c in pipes:                                   # step_wait (:76-81)
            c.recv()
        n += cores
    dt = time.perf_counter() - t0
    for c in pipes:
        c.send(None)
    for p in procs:
        p.join(timeout=5)
    return n / dt, n


def cpu_baseline(name, budget_s=15.0):
    """The oracle on the host cores, one process per bin: the plain-C restatement (the fair CPU
    number) and, for reference, the numpy/python one (what the reference's own Python costs) and the
    C one again under the reference's lockstep parent/worker structure."""
    cores = usable_cores()
    ctx = mp.get_context("fork")
    out = {}
    for impl, secs in (("c", budget_s * 0.5), ("python", budget_s * 0.3)):
        with ctx.Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(name, r, secs, impl) for r in range(cores)])
        out[impl] = (sum(n / t for n, t in res), sum(n for n, _ in res), secs)
    rate, steps, secs = out["c"]
    result = {"value": rate, "unit": "placement-steps/s", "cores": cores, "kind": "port",
              "sample": f"{cores} processes x 1 bin x {secs:.0f} s of the plain-C oracle (oracle/c/), {steps} steps in "
                        f"total, scripted MINZ policy in numpy (process-per-bin like shmem_vec_env; no physics, "
                        f"which flatters the CPU side)",
              "python_port": {"value": out["python"][0], "sample": f"same, numpy/python oracle, {out['python'][2]:.0f} s, "
                                                                   f"{out['python'][1]} steps"}}
    if make_workload(name)[2].get("bufferSize", 1) == 1:
        lock_rate, lock_steps = cpu_lockstep_runner(name, cores, budget_s * 0.2)
        result["lockstep_runner"] = {"value": lock_rate, "sample": f"plain-C oracle, {cores} worker processes + one parent "
                                                                   f"choosing all actions per step over pipes/shared memory "
                                                                   f"(the ShmemVecEnv structure), {lock_steps} steps"}
    return result


def respawn_under_torchrun(a):
    """`python bench.py --gpus N` with no torchrun environment: become N ranks.  Re-executes this
    command under `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1)
    so that a plain launch can never silently measure one GPU and call it N."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def pmc_profile(workload):
    """Counter passes committed under profiles/ (tools/collect_pmc.py) -- only if they were taken on the
    kernel sources this run uses (stamp == irbpp_amd.build.source_hash()); otherwise (None, reason)."""
    from irbpp_amd.build import source_hash
    path = os.path.join(ROOT, "profiles", "pmc_hbm.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_hbm.json"
    try:
        prof = json.load(open(path))
    except Exception as e:                      # noqa: BLE001
        return None, f"unreadable profiles/pmc_hbm.json: {e}"
    if prof.get("kernel_source_sha") != source_hash():
        return None, (f"profiles/pmc_hbm.json was taken on kernel sources {prof.get('kernel_source_sha')}, "
                      f"this run uses {source_hash()}")
    return prof.get("workloads", {}).get(workload), None


def timed_run(env, a, dev, k, barrier, steps, prefill, warmup, min_seconds=0.0, agree=None):
    """prefill (untimed: every bin deep in its own episode, terminal steps and auto-resets in the mix),
    warm-up, then blocks of exactly `steps` timed steps, each between two barriers, until the timed blocks add up to
    `min_seconds` (at least one block; `agree` makes the ranks agree on "enough").  `env` is a GroupedPackingEnv: every
    group of bins is stepped on its own stream, policy kernel then transition, without waiting for the other groups
    (one group = one launch per kernel over all bins).  -> (seconds, timed steps, transition ms per placement summed
    over the groups' own event pairs, episodes finished inside the timed blocks)."""
    G, per = env.num_groups, env.per
    launches = 2 if k > 1 else 1               # transitions per placement
    obs = env.reset()
    cur = [obs[env.rows(g)] for g in range(G)]
    nxt = [torch.empty_like(c) for c in cur]
    act = [torch.empty((per,), dtype=torch.int32, device=dev) for _ in range(G)]
    if k > 1:
        slot0 = torch.zeros((per,), dtype=torch.int32, device=dev)
        loc = [torch.empty((per, env.loc_obs_len), dtype=torch.float32, device=dev) for _ in range(G)]
    torch.cuda.synchronize(dev)

    # The scripted MINZ policy runs fused into the emit kernel (irbpp_set_auto_policy): every observation comes
    # with the action the policy picks on it, in act[g], which the next step() consumes.  The first actions come
    # from the stand-alone policy kernel on the reset observation (same function, tests/test_gpu_parity.py).
    for g in range(G):
        if k == 1:
            env.policy_minz_group(g, cur[g], actions_out=act[g])
        env.groups[g].set_auto_policy(act[g])
        # the ping-pong observation buffers belong to the library from here on (irbpp_register_obs_buffer): it clears
        # the candidate rows a bin no longer has instead of rewriting the zero tail of all S rows every step
        for t in ([loc[g]] if k > 1 else [cur[g], nxt[g]]):
            env.groups[g].register_obs_buffer(t)
    torch.cuda.synchronize(dev)

    def one_step():
        # wait=False: actions and buffers are long-lived and produced on the group's own stream (fused policy)
        for g in range(G):
            if k > 1:                          # one hierarchical placement: candidates of the
                env.get_action_candidates_group(g, slot0, obs_out=loc[g], wait=False)    # chosen buffer slot, then the placement
            env.step_group(g, act[g], obs_out=nxt[g], wait=False)
            cur[g], nxt[g] = nxt[g], cur[g]

    for _ in range(prefill + warmup):
        one_step()
    every = int(os.environ.get("IRBPP_BENCH_TIMING_EVERY", "4" if launches == 1 else "3"))   # odd for k > 1: both kinds of launch get sampled
    done_before = float(env.episode_totals()[0].item())
    elapsed, timed_steps, kms = 0.0, 0, []
    while True:
        for e in env.groups:                       # HIP events around every fourth transition, on its stream (an event
            e.enable_kernel_timing((steps * launches + every - 1) // every, every)   # pair per step costs the stream ~5 %)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        barrier()
        elapsed += time.perf_counter() - t0
        timed_steps += steps
        kms.append(float(sum(e.kernel_times_ms().mean() for e in env.groups)) * launches)
        enough = elapsed >= min_seconds
        if agree is not None:
            enough = agree(enough)
        if enough:
            break
    env.check_device_error()
    for e in env.groups:
        e.enable_kernel_timing(0)
    finished = float(env.episode_totals()[0].item()) - done_before
    return elapsed, timed_steps, float(np.mean(kms)), finished


# BASELINE.json configs as bench modes: (workload, global bins or None = --bins per GPU, default scaling)
CONFIGS = {
    "cfg2": ("blockout", None, "weak"),            # BlockOut online, 4096 bins per GPU (configs[1] to the letter)
    "cfg3": ("general", None, "weak"),             # General dataset, 4096 bins per GPU
    "cfg4": ("blockout_k10", 8192, "strong"),      # BlockOut buffered k=10, 8192 bins sharded over the GPUs
    "cfg5": ("abc_fine", 16384, "strong"),         # ABC fine heightmap, 16384 bins sharded over the GPUs
}


def vecenv_rate(bins, dev, steps=100):
    """placement-steps/s through the reference-facing GpuVecEnv.step the way trainer.py:161-186 drives it: device
    policy -> action.cpu().numpy() -> envs.step() -> (obs on device, reward CPU tensor, done numpy, infos), once with
    the trainer's per-env Python loop over infos and once with the same bookkeeping vectorised."""
    from irbpp_amd.vec_env import GpuVecEnv
    shapes, seqs, kw = make_workload("blockout")
    envs = GpuVecEnv(shapes, seqs, bins, device=dev, **kw)
    state = envs.reset()
    for _ in range(60):
        state, _, _, _ = envs.step(envs.env.policy_minz(state).cpu().numpy())

    def run(per_env_loop, device_actions=False):
        nonlocal state
        finished = 0
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(steps):
            action = envs.env.policy_minz(state)
            state, reward, done, infos = envs.step(action if device_actions else action.cpu().numpy())
            if per_env_loop:
                for i in range(len(infos)):                 # trainer.py:167-178
                    if done[i] and infos[i]["Valid"]:
                        finished += 1
            else:
                finished += int(done.sum())
        torch.cuda.synchronize(dev)
        return bins * steps / (time.perf_counter() - t)

    out = {"bins": bins, "with_trainer_per_env_loop": run(True), "without_per_env_loop": run(False),
           "device_action_tensor_no_loop": run(False, True), "unit": "placement-steps/s",
           "note": "GpuVecEnv.step as ONE group (make_vec_envs' default: the synchronous step is fastest that way) incl. host actions H2D, one pinned D2H of reward/done/info + sync per step"}
    envs.close()
    return out


def actor_loop_rate(bins, dev, steps=100, capacity=64):
    """SURVEY 8f-3, the caller side kept on the device: environment step + scripted policy (stands for Agent.act) + the
    vectorised replay memory's append (irbpp_amd.replay.actor_step over VectorReplayMemory: one tensor set for the N
    per-env memories of main.py:61-63) -- the acting loop of trainer.py:161-186 without its per-env Python loop and without
    a host round trip per step -- and one sample + priority update per env (memory.py:178-204, 206-210)."""
    from irbpp_amd.replay import VectorReplayMemory, actor_step
    from irbpp_amd.vec_env import GpuPackingEnv
    shapes, seqs, kw = make_workload("blockout")
    env = GpuPackingEnv(shapes, seqs, bins, device=dev, **kw)
    mem = VectorReplayMemory(bins, capacity, env.obs_len, device=dev)
    policy = lambda s_, m_: env.policy_minz(s_).to(torch.int64)      # noqa: E731
    state = env.reset()
    for _ in range(capacity + 8):
        state, _, _ = actor_step(env, policy, mem, state)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(steps):
        state, _, _ = actor_step(env, policy, mem, state)
    torch.cuda.synchronize(dev)
    t_act = (time.perf_counter() - t) / steps
    t = time.perf_counter()
    for _ in range(20):
        batch = mem.sample(1)
        mem.update_priorities(batch[0], torch.rand(bins, device=dev) + 0.1)
    torch.cuda.synchronize(dev)
    t_learn = (time.perf_counter() - t) / 20
    env.check_device_error()
    env.close()
    return {"bins": bins, "replay_capacity_per_env": capacity, "actor_steps_per_s": bins / t_act, "ms_per_actor_step": t_act * 1e3,
            "sample_plus_priority_update_ms": t_learn * 1e3, "unit": "placement-steps/s",
            "note": "GpuPackingEnv.step + policy kernel + VectorReplayMemory.append per step, everything on the device, one launch "
                    "group; the reference's structure for the same work is extra.vecenv_step.with_trainer_per_env_loop plus one "
                    "ReplayMemory.append per env in Python"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="steps per timed block")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the timed block of --steps steps until the timed blocks add up to this (0 = one block)")
    ap.add_argument("--prefill", type=int, default=300,
                    help="untimed steps before the warm-up that bring every bin to a steady-state episode mix")
    ap.add_argument("--bins", type=int, default=None, help="bins per GPU (weak scaling; default 8192 = north_star's target "
                    "configuration, 4096 under --config cfg2/cfg3) / ignored for a sharded --config")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="a BASELINE.json config as a bench mode")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: --bins per GPU; strong: the global bins (of --config, else --bins) divided over the GPUs")
    ap.add_argument("--groups", type=int, default=0,
                    help="independent groups of bins, each stepped on its own stream (0 = as many as the library recommends "
                         "for the data and size: vec_env.groups_for, i.e. two or one)")
    ap.add_argument("--tuning", type=int, default=0, help="irbpp_config::tuning bit flags (A/B measurements; results never change)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (other configs, 8192 bins, grouped, VecEnv)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; nccl is RCCL on ROCm "
                    "(gloo + several ranks on one device is only for dry runs of the multi-rank path)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--force-process-group", action="store_true",
                    help="create the torch.distributed process group even for ONE rank, so that the timed loop's barrier / "
                         "agreement and the final all-reduce run through the backend (RCCL) on a one-GPU box")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(a)                          # does not return

    cfg_workload, cfg_global, cfg_scaling = CONFIGS[a.config] if a.config else (None, None, None)
    if a.bins is None:
        a.bins = 4096 if a.config in ("cfg2", "cfg3") else 8192
    workload = a.workload or cfg_workload or "blockout"
    scaling = a.scaling or cfg_scaling or "weak"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if scaling == "strong":
        global_bins = cfg_global or a.bins
        if global_bins % world:
            raise SystemExit(f"strong scaling: {global_bins} global bins do not divide over {world} ranks")
        bins = global_bins // world
    else:
        bins = a.bins

    from irbpp_amd import distributed as D
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(workload, a.cpu_budget)         # before HIP is initialised: the pool forks
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} is running as {world} rank(s): launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {a.gpus}) or let bench.py spawn them itself")
    n_dev = torch.cuda.device_count()
    if a.backend == "nccl" and n_dev < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, this node shows {n_dev} "
                         "(--backend gloo runs several ranks on one device as a dry run only)")
    rank, world, local_rank = D.init_from_env(a.backend, force=a.force_process_group)    # nccl == RCCL on ROCm
    if a.backend != "nccl":
        local_rank %= max(1, n_dev)                        # dry run: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from irbpp_amd.vec_env import GroupedPackingEnv
    shapes, seqs, kw = make_workload(workload)
    if a.tuning:
        kw["tuning"] = a.tuning
    # The bins are stepped as the number of independent groups the library recommends for this data and size
    # (vec_env.groups_for: two groups on two HIP streams where that was measured to pay -- repeatable, because two consecutive
    # streams always get hardware queues of their own -- else one; what GpuVecEnv(num_groups=0) does); --groups overrides.
    # The same bins as ONE launch group are measured beside it (`extra.one_group`).
    from irbpp_amd.vec_env import groups_for
    groups = a.groups if a.groups > 0 else groups_for(workload, bins, device=dev)
    env = GroupedPackingEnv(shapes, seqs, bins, groups, device=dev, **D.shard(rank, world, bins), **kw)
    hc = env.Hx * env.Hy
    k = int(kw.get("bufferSize", 1))

    def barrier():
        D.barrier(dev)                 # torch.cuda.synchronize covers every group's stream

    def agree(flag):                   # every rank stops after the same block
        return D.max_over_ranks(0.0 if flag else 1.0, dev) == 0.0

    my_elapsed, timed_steps, kernel_ms, finished = timed_run(env, a, dev, k, barrier, a.steps, a.prefill, a.warmup,
                                                             a.min_seconds, agree)
    lds_bytes, kernel_name = env.groups[0].kernel_info()
    from irbpp_amd.vec_env import group_stream_report
    probe = group_stream_report(dev)           # what the stream probe behind groups_for saw in this process
    elapsed = D.max_over_ranks(my_elapsed, dev)
    fastest = -D.max_over_ranks(-my_elapsed, dev)
    finished = float(D.reduce_totals(torch.tensor([finished, 0, 0, 0], dtype=torch.float64, device=dev))[0].item())
    tot = D.reduce_totals(env.episode_totals()).cpu().numpy()   # the only exchange: 4 doubles over RCCL
    devices = D.gather_strings(f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)} "
                               f"[{torch.cuda.get_device_properties(local_rank).gcnArchName}] "
                               f"{my_elapsed / timed_steps * 1e3:.4f} ms/step")
    env.close()
    bps = algorithmic_bytes_per_step(shapes, hc, k)

    def side_run(wl, nbins, ngroups=1, min_seconds=0.4):
        """another configuration measured the same way on this rank's device (extras: never `value`)"""
        sh, sq, kw2 = make_workload(wl)
        k2 = int(kw2.get("bufferSize", 1))
        e2 = GroupedPackingEnv(sh, sq, nbins, ngroups, device=dev, **kw2)
        t2, n2, km2, f2 = timed_run(e2, a, dev, k2, lambda: torch.cuda.synchronize(dev), a.steps, a.prefill, a.warmup, min_seconds)
        b2 = algorithmic_bytes_per_step(sh, e2.Hx * e2.Hy, k2)
        e2.close()
        return {"workload": wl, "bins": nbins, "groups": ngroups, "value": nbins * n2 / t2, "ms_per_step": t2 / n2 * 1e3,
                "steps": n2, "kernel_ms": km2, "algorithmic_bytes_per_step": b2,
                "roofline_frac": b2 * nbins / (t2 / n2) / 1e9 / HBM_PEAK_GBS, "episodes_finished_in_timed_region": f2}

    # Extra measurements on one GPU (never `value`): the same bins as four independent groups on four HIP streams
    # (vec_env.GroupedPackingEnv: one group's straggler workgroups overlap the next group's kernels; how much that gives
    # depends on how the runtime maps the streams onto its hardware queues, hence two instances, both reported),
    # north_star's 8192 bins on one GPU, the other BASELINE configs at their per-GPU sizes, and the rate a user of the
    # reference-facing VecEnv API gets.
    extra, grouped = None, None
    if world == 1 and not a.no_extra:
        extra = {}
        if groups != 1:
            extra["one_group"] = side_run(workload, bins, 1)             # the same bins as one launch per kernel
        if workload == "blockout":
            def recommended(wl, nb):
                return side_run(wl, nb, groups_for(wl, nb, device=dev))
            # round-over-round comparable: the definition `value` had until round 4 (configs[1] to the letter, ONE launch group)
            extra["cfg2_4096_one_group"] = side_run("blockout", 4096, 1)
            if bins != 4096:
                extra["bins4096_one_gpu"] = recommended("blockout", 4096)     # BASELINE configs[1] at its own size
            if bins != 8192:
                extra["bins8192_one_gpu"] = recommended("blockout", 8192)
            extra["cfg3_general_4096"] = recommended("general", 4096)
            extra["cfg4_blockout_k10_1024_per_gpu"] = recommended("blockout_k10", 1024)
            extra["cfg5_abc_fine_2048_per_gpu"] = recommended("abc_fine", 2048)
            extra["cfg1_cube_4096"] = recommended("cube", 4096)
            extra["vecenv_step"] = vecenv_rate(4096, dev)
            extra["actor_loop"] = actor_loop_rate(4096, dev)
            extra["blockout_r8_8192"] = recommended("blockout_r8", 8192)      # cfg 2 at the README command's eight rotations (README.md:100)

    if rank == 0:
        total_steps = bins * world * timed_steps
        # Roofline on the whole step: the transition is four kernels per group and the groups overlap, so no single
        # launch duration prices the step's algorithmic bytes; its wall time (gaps included) does
        achieved = bps * bins / (elapsed / timed_steps) / 1e9
        prof, why = pmc_profile(workload)
        traffic, issue = None, None
        if prof is not None:
            # HBM bytes per launch from the PMC passes (taken at prof["bins"] bins per launch, beyond the
            # Infinity Cache); bins are independent, so a launch's traffic scales with their number
            traffic = prof["hbm_bytes_per_launch"] * bins / prof["bins"]
            issue = prof.get("issue")
        out = {
            "metric": "env steps/sec (placements/sec) across N parallel bins",
            "value": total_steps / elapsed, "unit": "placement-steps/s",
            # `steps` = the K of the contract: every timed block is EXACTLY K steps between barrier + synchronize on both sides; the block
            # is repeated until the blocks add up to --min-seconds and `value` / `ms_per_step` are the mean over the blocks
            "n_gpus": world, "steps": a.steps, "steps_per_block": a.steps, "timed_blocks": timed_steps // a.steps,
            "steps_timed_total": timed_steps, "min_seconds": a.min_seconds,
            "warmup": a.warmup, "prefill_steps": a.prefill,
            "ms_per_step": elapsed / timed_steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{workload} {'online' if k == 1 else 'buffered'} (bufferSize={k}), {bins} bins/GPU, resolutionA=0.02 "
                                   f"resolutionH={kw['resolutionH']}, R={shapes.n_rot}, S={S}, scripted MINZ policy"
                                   + (f", stepped as {groups} groups of bins on {groups} HIP streams (vec_env.groups_for)" if groups > 1 else ""),
                       "baseline_config": a.config, "bins_per_gpu": bins, "global_bins": bins * world,
                       "parallelism": f"bins sharded x{world}", "groups_per_gpu": groups,
                       "group_stream_probe": probe},
            "ranks": {"world_size": world, "backend": a.backend if dist.is_initialized() else None,
                      "process_group": bool(dist.is_initialized()), "devices": devices,
                      "ms_per_step_min": fastest / timed_steps * 1e3, "ms_per_step_max": elapsed / timed_steps * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "basis": "algorithmic bytes of one step of this rank's bins / wall time of one step",
                         "kernel": kernel_name, "kernel_ms": kernel_ms, "lds_bytes_per_workgroup": lds_bytes,
                         "kernel_note": "a step is a chain of kernels (transition [behind irbpp_apply_kernel at large launches], trace, "
                                        "polygon, emit), none of which carries the step's bytes alone: kernel_ms = HIP events "
                                        "around the chain on its stream (summed over the groups, which overlap), and `achieved` "
                                        "prices the step's algorithmic bytes on its wall time",
                         "algorithmic_bytes_per_step": bps},
            "episodes": {"finished_in_timed_region": finished, "finished_since_reset": float(tot[0]),
                         "mean_ratio": float(tot[1] / tot[0]) if tot[0] else None,
                         "mean_items": float(tot[2] / tot[0]) if tot[0] else None},
        }
        if why is not None:
            out["roofline"]["traffic_note"] = why
        if issue is not None:
            out["roofline"]["issue"] = issue       # the kernel is instruction-issue bound, not HBM bound: SQ busy shares
        out["value_definition"] = f"{workload}, {bins} bins/GPU x {world} GPU(s), stepped as {groups} group(s) per GPU"
        if extra and "cfg2_4096_one_group" in extra:
            out["value_cfg2_one_group"] = extra["cfg2_4096_one_group"]["value"]     # stable key: 4096 BlockOut bins, one launch group
        if extra:
            out["extra"] = extra
        if cpu is not None:
            out["cpu_baseline"] = cpu
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: librccl announces itself through C stdio ("Librccl path : ..."), which sits
        # in the C library's buffer until flushed -- at exit, i.e. behind a line Python printed earlier
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
