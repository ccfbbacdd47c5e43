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
