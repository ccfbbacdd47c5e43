"""bench.py's bookkeeping that does not need a GPU: the algorithmic-bytes formula of SURVEY.md 8(d)
and the host-core count the CPU baseline is allowed to use."""
import json

import pytest
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_follow_the_survey_formula():
    shapes, _, kw = bench.make_workload("blockout")
    hc, S = 32 * 32, 500
    f = np.array([[t[0].size for t in per_rot] for per_rot in shapes.tables], dtype=np.float64)
    reads = 8 * hc + 5 * f.sum(1).mean() + 5 * f.mean() + 4 + 4
    writes = 8 * hc + 4 * (5 * S + 9 + hc) + 5
    assert bench.algorithmic_bytes_per_step(shapes, hc, 1) == reads + writes
    # SURVEY 8(d) works cfg 2 out at ~25.9 KB with a float32 heightmap and F ~ 12x12 per rotation; the
    # synthetic polycubes average ~98 cells per rotation, and the float64 master tile adds 2*4*Hc
    assert 23_000 < bench.algorithmic_bytes_per_step(shapes, hc, 1) - 8 * hc < 26_000
    # buffered placements also write the order observation [k ids | heightmap]
    k10 = bench.algorithmic_bytes_per_step(shapes, hc, 10)
    assert k10 - bench.algorithmic_bytes_per_step(shapes, hc, 1) == 4 * 9 + 4 * (10 + hc)


def test_every_workload_builds_and_names_its_config():
    for name, n_rot, res_h in (("blockout", 4, 0.01), ("blockout_r8", 8, 0.01), ("blockout_k10", 4, 0.01),
                               ("general", 8, 0.01), ("abc_fine", 8, 0.005), ("cube", 2, 0.01)):
        shapes, seqs, kw = bench.make_workload(name)
        assert shapes.n_rot == n_rot and kw["resolutionH"] == res_h
        assert seqs.dtype == np.int32 and seqs.max() < shapes.n_shapes
        shapes.validate(kw["resolutionH"], kw["resolutionA"])
    assert bench.make_workload("blockout_k10")[2]["bufferSize"] == 10


def test_usable_cores_is_bounded_by_affinity():
    n = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


@pytest.mark.parametrize("which", ["r01_final", os.path.join("r02", "final"), os.path.join("r03", "final")])
def test_committed_bench_line_keeps_the_contract(which):
    line = json.load(open(os.path.join(ROOT, "profiles", which, "bench_default.json")))
    if which.startswith("r03"):                       # round 3: load-independent timing, per-config extras, VecEnv rates
        assert line["steps"] % line["steps_per_block"] == 0 and line["steps"] * line["ms_per_step"] >= 1e3 * line["min_seconds"]
        for key in ("cfg3_general_4096", "cfg4_blockout_k10_1024_per_gpu", "cfg5_abc_fine_2048_per_gpu", "cfg1_cube_4096"):
            assert 0 < line["extra"][key]["roofline_frac"] < 1 and line["extra"][key]["value"] > 1e6
        assert line["extra"]["vecenv_step"]["with_trainer_per_env_loop"] > 1e6
        assert line["ranks"]["ms_per_step_min"] <= line["ranks"]["ms_per_step_max"] and line["scaling"] == "weak"
    if which != "r01_final":                          # round 2: north_star's 8192-bin figure, steady-state mix, ranks
        assert line["extra"]["bins8192_one_gpu"]["value"] > line["value"] * 0.5
        assert line["prefill_steps"] > 0 and line["episodes"]["finished_in_timed_region"] > 0
        assert line["ranks"]["world_size"] == line["n_gpus"] == 1 and "workload" in line["config"]
        assert line["config"]["bins_per_gpu"] == 4096 and line["vs_baseline"] is None
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    r = line["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(line["value"] - 4096 * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1.0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1


def test_gpus_n_never_degrades_to_one_rank():
    """`python bench.py --gpus 2` launched as a plain process (how the driver launches it) must become two
    ranks or fail -- never print a 1-GPU line.  Here (no GPU) both spawned ranks refuse: non-zero exit, no JSON."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--prefill", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0
    assert '"n_gpus"' not in res.stdout
    assert "needs 2 visible GPUs" in res.stderr or "needs 2 visible GPUs" in res.stdout


def test_rank_count_mismatch_is_refused(monkeypatch):
    """A torchrun world of 2 with --gpus 4 is an error, not a silently smaller run."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0 and "is running as 1 rank(s)" in res.stderr


def test_pmc_profile_is_quoted_only_for_matching_kernel_sources(tmp_path, monkeypatch):
    from irbpp_amd.build import source_hash
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    assert bench.pmc_profile("blockout") == (None, "no profiles/pmc_hbm.json")
    doc = {"kernel_source_sha": "0" * 16, "workloads": {"blockout": {"bins": 16384, "hbm_bytes_per_launch": 1.0}}}
    (tmp_path / "profiles" / "pmc_hbm.json").write_text(json.dumps(doc))
    prof, why = bench.pmc_profile("blockout")
    assert prof is None and "was taken on kernel sources" in why
    doc["kernel_source_sha"] = source_hash()
    (tmp_path / "profiles" / "pmc_hbm.json").write_text(json.dumps(doc))
    assert bench.pmc_profile("blockout") == ({"bins": 16384, "hbm_bytes_per_launch": 1.0}, None)
