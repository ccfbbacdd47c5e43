"""Host-side helpers of the package that need no GPU."""
import os


def test_use_hardware_queues_sets_the_runtime_variable_once(monkeypatch):
    """irbpp_amd.use_hardware_queues: GPU_MAX_HW_QUEUES for grouped stepping (profiles/r04/s42), left alone if the caller
    has set it.  (No GPU here: the process never initialises HIP, so the call is always in time.)"""
    import irbpp_amd
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    irbpp_amd.use_hardware_queues(8)
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    irbpp_amd.use_hardware_queues(16)                                  # a value already there stays
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)


def test_groups_for_lattice_data_depends_on_the_hardware_queues(monkeypatch):
    """vec_env.groups_for: free-form data as 2 / 4 groups; lattice data as four groups only from 4096 bins on and only when
    the runtime was asked for eight hardware queues (with four the group streams share queues: profiles/r04/s42)."""
    from irbpp_amd.vec_env import groups_for
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert groups_for("general", 4096) == 2 and groups_for("abc_fine", 2048) == 4 and groups_for("general", 512) == 1
    assert groups_for("lattice", 4096) == 1 and groups_for("blockout_k10", 8192) == 1
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert groups_for("lattice", 4096) == 4 and groups_for("blockout", 8192) == 4
    assert groups_for("lattice", 1024) == 1 and groups_for("lattice", 4098) == 1
    assert groups_for("general", 4096) == 2
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "not a number")
    assert groups_for("lattice", 4096) == 1
