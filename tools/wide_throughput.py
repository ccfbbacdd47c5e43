import sys, time, json, torch
sys.path.insert(0, '.')
import irbpp_amd
from irbpp_amd import synthetic
from irbpp_amd.vec_env import GpuPackingEnv
out = {}
for name, sh in (("general_r4", synthetic.general_shapes(n_shapes=64, n_rot=4, fmin=4, fmax=14, seed=3)),
                 ("blockout_r4", synthetic.blockout_shapes(n_shapes=64, n_rot=4, seed=0))):
    seqs = synthetic.make_sequences(sh.n_shapes, 2000, 160, seed=1)
    for n in (256, 1024, 4096):
        env = GpuPackingEnv(sh, seqs, n, device="cuda:0", resolutionA=0.01, resolutionH=0.01)
        act = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        env.set_auto_policy(act)
        obs = env.reset()
        buf = [torch.empty_like(obs), torch.empty_like(obs)]
        for t in range(60):
            env.step(act, obs_out=buf[t & 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 40
        for t in range(steps):
            env.step(act, obs_out=buf[t & 1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        env.check_device_error()
        out[f"{name}@{n}"] = {"ms_per_step": round(dt * 1e3, 3), "Msteps_per_s": round(n / dt / 1e6, 3)}
        env.close()
print(json.dumps(out))
