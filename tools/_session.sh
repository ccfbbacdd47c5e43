timeout 600 python -m pytest tests/test_replay.py -m gpu -q 2>&1 | tail -5
