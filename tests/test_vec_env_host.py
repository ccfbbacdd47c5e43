"""Host-side helpers of the package that need no GPU."""

def test_groups_for_recommends_two_groups_or_one(monkeypatch):
    """vec_env.groups_for: two groups (two consecutive streams always get hardware queues of their own: profiles/r05/s9) where
    that was measured to pay, one otherwise; never four, and nothing read from the process environment."""
    from irbpp_amd.vec_env import groups_for
    for queues in (None, "4", "8"):
        if queues is None:
            monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
        else:
            monkeypatch.setenv("GPU_MAX_HW_QUEUES", queues)
        assert groups_for("general", 4096) == 2 and groups_for("abc_fine", 2048) == 2 and groups_for("general", 512) == 1
        assert groups_for("lattice", 4096) == 2 and groups_for("blockout", 8192) == 2 and groups_for("cube", 8192) == 2
        assert groups_for("lattice", 2048) == 2 and groups_for("lattice", 1024) == 1 and groups_for("lattice", 4097) == 1
        assert groups_for("blockout_k10", 1024) == 1 and groups_for("lattice", 1024, buffered=True) == 1
        assert groups_for("blockout_k10", 8192) == 2 and groups_for("blockout_k10", 2048) == 2
